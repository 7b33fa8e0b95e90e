// Shared device/host helpers for the sliders_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef unsigned short bf16_t;  // raw storage type used in host-visible signatures

#define SLH_WAVE 64

// 64 B of zeros that out-of-image im2col taps read from (device globals are zero-initialised).
static __device__ __attribute__((aligned(64))) unsigned int slh_zero_page[64];

__device__ __forceinline__ float bf2f(__bf16 x) { return (float)x; }
__device__ __forceinline__ __bf16 f2bf(float x) { return (__bf16)x; }

__device__ __forceinline__ float bfraw2f(unsigned short u) {
    return __builtin_bit_cast(float, ((unsigned int)u) << 16);
}
// round-to-nearest-even fp32 -> bf16 bits (matches torch's CPU/GPU conversion; NaN kept quiet)
__device__ __forceinline__ unsigned short f2bfraw(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ float round_bf16(float f) { return (float)((__bf16)f); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// async 16-byte global -> LDS copy (dest = wave-uniform base + lane*16)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Workgroup barrier that also drains this wave's LDS-DMA.  __syncthreads() alone is NOT enough after glds16: it is a
// workgroup-scope fence + s_barrier, and on gfx9 (non-tgsplit) that fence waits lgkmcnt(0) only, while the LDS write of
// a global_load_lds is tracked by vmcnt.  hipcc's own waitcnt insertion covers straight-line code but was seen to leave
// the loop-carried case (prefetch tile t+1, barrier at the top of the next iteration) without any vmcnt wait - a tile
// could be consumed before it landed (scripts/check_lds_dma_waits.py proves the wait is there in the compiled code).
__device__ __forceinline__ void lds_dma_syncthreads() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// The same copy issued from inline asm, i.e. INVISIBLE to hipcc's s_waitcnt bookkeeping.  An LDS-DMA that the compiler
// can see makes it treat the LGKM counter as out-of-order ("pending flat"), and every ds_read consumer in the loop
// then waits lgkmcnt(0) - which drains the fragment reads a software-pipelined K loop wants to keep in flight
// (checked in the ISA: counted lgkmcnt(4/5) appear only once the LDS-DMA is hidden).  The caller owns the vmcnt
// accounting (counted s_waitcnt vmcnt(N) + s_barrier before the data is read).  M0 (the LDS destination base) is
// written and consumed inside the one statement; kernels that use this helper must not ALSO use the compiler-visible
// glds16 in the same loop (the compiler assumes it owns M0 between its own LDS-DMA instructions).
// lds_byte_addr must be wave-uniform.
__device__ __forceinline__ void glds16_hidden(const void* gsrc, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// ---- host side -------------------------------------------------------------------------------
void slh_set_error(const char* fmt, ...);
#define SLH_CHECK(cond, ...)            \
    do {                                \
        if (!(cond)) {                  \
            slh_set_error(__VA_ARGS__); \
            return -1;                  \
        }                               \
    } while (0)
#define SLH_LAUNCH_CHECK(name)                                              \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            slh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return -2;                                                      \
        }                                                                   \
    } while (0)
