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
// The same function, x * Phi(x), through erfc(z) = t * exp(-z^2 + P(t)), t = 1 / (1 + z / 2) (the classic Chebyshev fit: fractional
// error < 1.2e-7 everywhere; against float64 the product is within 4.4e-6 for |x| < 6 and as good as an fp32 evaluation of the erf
// form beyond) - 10 fma + v_rcp_f32 + v_exp_f32 instead of libm's erff (~50 VALU slots per element: the GEGLU epilogue of a
// 128 x 320 tile spent ~7 us of each of its two rounds on it, one workgroup per CU).  Forward GEGLU only; its result is rounded to
// bf16 next (the reference's gelu output is a bf16 tensor).  -DSLH_GELU_ERFF restores erff (A/B builds).
__device__ __forceinline__ float gelu_erf_fast_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.5f, z, 1.0f));
    float p = 0.17087277f;
    p = __builtin_fmaf(p, t, -0.82215223f); p = __builtin_fmaf(p, t, 1.48851587f); p = __builtin_fmaf(p, t, -1.13520398f);
    p = __builtin_fmaf(p, t, 0.27886807f); p = __builtin_fmaf(p, t, -0.18628806f); p = __builtin_fmaf(p, t, 0.09678418f);
    p = __builtin_fmaf(p, t, 0.37409196f); p = __builtin_fmaf(p, t, 1.00002368f); p = __builtin_fmaf(p, t, -1.26551223f);
    const float e = t * __expf(__builtin_fmaf(-z, z, p));          // erfc(|x| / sqrt 2)
    const float h = 0.5f * e;
    return x * (x >= 0.f ? 1.0f - h : h);
}

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

// ---- weight touch (slh_gemm_desc.pf_* / slh_attn_desc.pf_*) --------------------------------------------------------------------
// Workgroup `idx` of `count` touch workgroups - dispatched behind every workgroup that does the launch's own work - streams its
// share of a byte range a LATER launch will need through plain loads (they allocate in the memory-side cache) and exits.
// 16 independent 16-byte loads per thread in flight (128 KB per workgroup): at ~2 us per HBM miss anything less leaves the touch
// slower than the launch it rides on, and a launch does not end before its touch does.
template <int U = 16>      // independent 16-byte loads per thread (a kernel with a tight register budget takes fewer)
__device__ __forceinline__ void weight_touch(const void* ptr, const long bytes, const int idx, const int count) {
    const uint4* src = (const uint4*)ptr;
    const long n16 = bytes >> 4, stride = (long)count * blockDim.x;
    unsigned acc = 0;
    long i = (long)idx * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x;
    }
    for (; i < n16; i += stride) acc ^= src[i].x;
    asm volatile("" ::"v"(acc));       // the loads are the point
}

// ---- write-through result stores (round 5) ------------------------------------------------------------------------------------
// A kernel's results are consumed by the NEXT launch, on other XCDs as much as on this one: the per-XCD L2s are not coherent, so
// the end of every kernel writes this XCD's dirty lines back (the implicit release) before the next dispatch may start.  Stored
// write-through (sc0 sc1), the bytes leave for the memory side while the kernel is still busy and the end-of-kernel write-back finds
// nothing to do: -0.5 us per launch on the 128 x 128 GEMM tile (2048 x 1280 x 64: 8.4 -> 7.9 us), SDXL 1024^2 pass 24.02 -> 23.82 ms
// with the GEMM row stores and the GroupNorm / LayerNorm stores in this form (same box, three alternations:
// profiles/r05_write_through_stores.txt); the consumer reads from the memory-side cache either way.  16-BYTE STORES ONLY: written
// through, the 8-byte quads of the GEGLU epilogue cost the pass +0.65 ms and those of the attention kernels +0.3 ms (a scalar
// write-through store is one fabric write each) - they stay plain.  nt measured neutral, sc1 alone slightly behind sc0 sc1.
// SLH_WT_MASK (A/B builds, scripts/build_variant_all.sh): bit 0 GEMM row stores, bit 3 GroupNorm / LayerNorm; 0 = all plain.
#ifndef SLH_WT_MASK
#define SLH_WT_MASK 9
#endif
constexpr int SLH_WT_AUX = 17;     // sc0 | sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const void* base) {       // base: wave-uniform
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ void wt_store16(const __amdgpu_buffer_rsrc_t r, const long byte_off, const bf16x8 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (unsigned)byte_off, 0, SLH_WT_AUX);
}
// The resource above spans 0x7FFFFFF0 bytes from the tensor's base and the offset is 32 bits: a row store beyond that is dropped by
// the bounds check (silently).  Every entry point that stores this way refuses a result tensor that does not fit (host side):
inline bool wt_span_ok(long rows, long ld, long cols) { return rows <= 0 || ((rows - 1) * ld + cols) * 2 < 0x7FFFFFF0L; }

// ---- fixed-order cross-workgroup reductions -----------------------------------------------------------------------
// Every reduction whose result feeds bf16 activations (GroupNorm statistics, split-K partial sums) is done in a fixed
// order: fp32 atomics commit in arrival order, the last bits of the sum then differ from run to run, a bf16 rounding
// boundary flips somewhere, and ~1000 dependent kernels later two runs of the SAME pass differ by the whole bf16 error
// budget (measured in rounds 1-2: rel-L2 7e-3 run to run).  The pattern (cdna_hip_programming.md §5 "in-launch split-K
// reduction", sc1 form): every workgroup publishes its partial (value pairs, one 8-byte write-through store each),
// drains its stores, takes a ticket; the workgroup that draws the last ticket reads all partials back (sc1 loads: L2
// of any XCD is bypassed) and combines them in index order.  Which workgroup is last varies, the arithmetic does not.
__device__ __forceinline__ void store_pair_sc1(float* p, float a, float b) {
    const unsigned long long bits = ((unsigned long long)__builtin_bit_cast(unsigned, b) << 32) | __builtin_bit_cast(unsigned, a);
    __hip_atomic_store((unsigned long long*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x2 load_pair_sc1(const float* p) {
    const unsigned long long bits = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x2{__builtin_bit_cast(float, (unsigned)bits), __builtin_bit_cast(float, (unsigned)(bits >> 32))};
}
// the same pair read with only the L1 bypassed (workgroup scope): served by the L2 of THIS XCD - correct only for data whose
// write-through stores were issued on this XCD (split-K reduction: the ticket proves it)
__device__ __forceinline__ f32x2 load_pair_l2(const float* p) {
    const unsigned long long bits = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return f32x2{__builtin_bit_cast(float, (unsigned)bits), __builtin_bit_cast(float, (unsigned)(bits >> 32))};
}
// true in every thread of the workgroup that arrives last of `expected` at *ticket (zeroed by the caller before the
// launch; the last arriver re-arms it).  lds_flag: one int of the kernel's LDS.
__device__ __forceinline__ bool last_arriver(unsigned* ticket, unsigned expected, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's sc1 stores have reached the memory side
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == expected - 1;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// Fixed-order reduction of per-thread, per-channel (s, q) pairs to per-group pairs inside one workgroup.
// lds: [2][rpi * C] floats.  Thread (chunk, rl) owns CH consecutive channels; afterwards the threads t = g * lpg
// (g < groups) hold group g's sums in (S, Q) and the function returns true for exactly those threads.  The order of
// every addition is a function of the launch geometry only - no LDS / global atomics (they commit in arrival order).
template <int CH>
__device__ __forceinline__ bool gn_block_reduce(float* lds, int C, int cg, int groups, int rpi, int lpg, int chunk, int rl,
                                                const float* s, const float* q, float& S, float& Q) {
    float* ls = lds;
    float* lq = lds + rpi * C;
    const int base = rl * C + chunk * CH;
#pragma unroll
    for (int e = 0; e < CH; e += 4) {
        *(f32x4*)(ls + base + e) = f32x4{s[e], s[e + 1], s[e + 2], s[e + 3]};
        *(f32x4*)(lq + base + e) = f32x4{q[e], q[e + 1], q[e + 2], q[e + 3]};
    }
    __syncthreads();
    const int tid = threadIdx.x;
    const int g = tid / lpg, j = tid & (lpg - 1);
    float a = 0.f, b = 0.f;
    if (g < groups) {
        const int E = rpi * cg;
        for (int idx = j; idx < E; idx += lpg) {
            const int r = idx / cg;
            const int o = r * C + g * cg + (idx - r * cg);
            a += ls[o];
            b += lq[o];
        }
    }
    for (int o = lpg >> 1; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    S = a; Q = b;
    return g < groups && j == 0;
}

// Sum of the `count` published pairs partial[(i * groups + g) * 2 ..] of group g = tid / lpg in index order (double);
// valid in the threads t = g * lpg.  The loads are independent sc1 (L2-bypassing) reads of ~1-2 us each: they are issued
// 16 at a time, only the additions are ordered.
__device__ __forceinline__ void gn_combine_partials(const float* partial, int count, int groups, int lpg, double& S, double& Q) {
    const int tid = threadIdx.x;
    const int g = tid / lpg, j = tid & (lpg - 1);
    double a = 0.0, b = 0.0;
    if (g < groups) {
        constexpr int U = 16;
        for (int i0 = j; i0 < count; i0 += U * lpg) {
            f32x2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * lpg;
                v[u] = i < count ? load_pair_sc1(partial + ((long)i * groups + g) * 2) : f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { a += (double)v[u][0]; b += (double)v[u][1]; }
        }
    }
    for (int o = lpg >> 1; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    S = a; Q = b;
}

// Two-level fixed-order reduction of one sample's published pairs (GroupNorm statistics kernels).  The row blocks of a
// sample form clusters of GN_CLUSTER; the last block to arrive in a cluster adds its cluster's pairs (level 1) and
// publishes one pair per group, the last CLUSTER to finish adds those (level 2).  Each level is one batch of 16 loads
// per lane, so the serial tail behind the last workgroup is two L2-bypassing round trips, whatever the tensor size
// (a flat combine of 1024 row blocks cost 50 us of dependent loads).
//   part   : [row_blocks + nclusters][groups][2]   level-1 pairs, then level-2 pairs
//   ticket : [1 + nclusters]                       ticket[0] = level 2, ticket[1 + c] = cluster c
// Returns true in the threads t = g * lpg (g < groups) of the ONE workgroup that ends up with the sample's totals.
constexpr int GN_CLUSTER = 16;
__device__ __forceinline__ bool gn_two_level_reduce(float* part, unsigned* ticket, int row_blocks, int rb, int groups, int lpg,
                                                    bool owner, float S1, float Q1, int* lds_flag, double& S, double& Q) {
    const int tid = threadIdx.x;
    const int g = tid / lpg;
    const int ncl = (row_blocks + GN_CLUSTER - 1) / GN_CLUSTER;
    const int cl = rb / GN_CLUSTER;
    if (owner) store_pair_sc1(part + ((long)rb * groups + g) * 2, S1, Q1);
    const int first = cl * GN_CLUSTER;
    const int members = min(GN_CLUSTER, row_blocks - first);
    if (!last_arriver(ticket + 1 + cl, (unsigned)members, lds_flag)) return false;
    gn_combine_partials(part + (long)first * groups * 2, members, groups, lpg, S, Q);
    if (ncl == 1) return g < groups && (tid & (lpg - 1)) == 0;
    if (g < groups && (tid & (lpg - 1)) == 0) store_pair_sc1(part + ((long)(row_blocks + cl) * groups + g) * 2, (float)S, (float)Q);
    if (!last_arriver(ticket, (unsigned)ncl, lds_flag)) return false;
    gn_combine_partials(part + (long)row_blocks * groups * 2, ncl, groups, lpg, S, Q);
    return g < groups && (tid & (lpg - 1)) == 0;
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
// kernel-name query mode (error.cpp; slh_gemm_kernel_name): the dispatch code runs as for a launch, slh_launch records the chosen
// instantiation instead of launching it, and the launch checks are skipped (no device is needed)
bool slh_name_mode();
void slh_name_record(const char* fmt, ...);
void slh_name_sink_set(char* buf, int cap);
inline const char* slh_tf(bool b) { return b ? "true" : "false"; }
// fmt + args spell the instantiation's template arguments from the SAME constants the template is instantiated with, in the one
// statement that launches it (hipcc's __PRETTY_FUNCTION__ drops the arguments of a function-template pointer, so they are not derived)
template <auto Kern, class A, class... N>
inline void slh_launch(int grid, int block, hipStream_t s, const A& a, const char* fmt, N... n) {
    if (slh_name_mode()) { slh_name_record(fmt, n...); return; }
    hipLaunchKernelGGL(Kern, dim3(grid), dim3(block), 0, s, a);
}
#define SLH_LAUNCH_CHECK(name)                                              \
    do {                                                                    \
        if (slh_name_mode()) break;                                         \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            slh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return -2;                                                      \
        }                                                                   \
    } while (0)
