// Micro-benchmark (development aid, round 6): what does ONE LDS-DMA piece cost the instruction stream of a wave that is issuing MFMAs,
// by addressing form?  One workgroup per CU, WAVES waves (4 = one per SIMD, 8 = two per SIMD); every wave runs ITER rounds of 40
// independent 16x16x32 MFMAs (the half tile of gemm7w_kernel) with P pieces dealt out behind every second MFMA:
//   form 0: no DMA (baseline)
//   form 1: global_load_lds_dwordx4 with a 64-bit VGPR address
//   form 2: global_load_lds_dwordx4 with SGPR base + 32-bit VGPR offset
//   form 3: buffer_load_dwordx4 ... offen lds (resource + 32-bit VGPR offset)
//   form 4: buffer_load_dwordx4 ... lds with ADD_TID_ENABLE (stride 16: lane i fetches base + 16 i; NO address register)
// The window (64 KB per CU) is L2 resident; vmcnt is kept below 32 by a wait every 8 pieces.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form scripts/ubench_dma_issue.hip -o scripts/ubench_dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int FORM, int P>
__global__ __launch_bounds__(512) void k(const char* base, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* win = base + (long)blockIdx.x * 65536 + wave * 8192;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8192;
    const char* vaddr = win + lane * 16;
    const unsigned voff = lane * 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)win, 16, 0x7fffffff, 0x00820000);      // ADD_TID_ENABLE, stride 16
    f32x4 acc[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * lane); b[e] = (__bf16)(0.002f * e); }
    int cnt = 0;
    for (int it = 0; it < iters; ++it) {
        int m = 0;
#pragma unroll
        for (int i = 0; i < 40; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            if ((i & 1) == 1 && m < P) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned dst = lds0 + (m & 7) * 1024;
                const int so = (m & 7) * 1024;
                if (FORM == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr + so), "s"(dst) : "memory");
                if (FORM == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(win + so), "s"(dst) : "memory");
                if (FORM == 3) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(so), "s"(dst) : "memory");
                if (FORM == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %0, %1 lds" ::"s"(rt), "s"(so), "s"(dst) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                ++m;
                if (FORM != 0 && ((++cnt) & 7) == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 40; ++i) s += acc[i];
    if (s[0] == 123.456f) sink[0] = s[1];
}

template <int FORM, int P>
void run(const char* buf, float* sink, int waves) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    hipLaunchKernelGGL((k<FORM, P>), dim3(256), dim3(64 * waves), 0, 0, buf, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<FORM, P>), dim3(256), dim3(64 * waves), 0, 0, buf, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_per_round = ms * 1e3 / iters;
    printf("%5d %5d %5d   %8.3f us per 40-MFMA round   (%6.0f clk @2.1 GHz)\n", FORM, P, waves, us_per_round, us_per_round * 2100);
    fflush(stdout);
}

int main() {
    char* buf;
    CK(hipMalloc(&buf, 256L * 65536));
    CK(hipMemset(buf, 1, 256L * 65536));
    float* sink;
    CK(hipMalloc(&sink, 4));
    printf(" form     P waves\n");
    for (int waves : {4, 8}) {
        run<0, 0>(buf, sink, waves);
        run<1, 5>(buf, sink, waves); run<2, 5>(buf, sink, waves); run<3, 5>(buf, sink, waves);
        run<1, 10>(buf, sink, waves); run<2, 10>(buf, sink, waves); run<3, 10>(buf, sink, waves);
        // (form 4 - ADD_TID_ENABLE addressing - faulted on the first try and is not run)
    }
    return 0;
}
