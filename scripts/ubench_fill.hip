// Micro-benchmark (development aid): how fast can ONE CU pull L2-resident data, and through which path?
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 64-bit per-lane addresses), 1 KB contiguous per instruction
//   mode 1: global_load_dwordx4 into VGPRs (inline asm, never waited per load), 1 KB contiguous per instruction
//   mode 2: LDS-DMA with the SGPR-base + 32-bit VGPR offset addressing form
//   mode 3: buffer_load_dwordx4 ... lds (MUBUF form)
// One workgroup per CU (the LDS request forces it), `waves` waves, each streaming its own DEPTH KB window again and
// again (L2 resident), DEPTH pieces per inner iteration, straight-line code: 2-3 instructions per piece.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_fill.hip -o scripts/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void fill_kernel(const char* base, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* win = base + ((long)blockIdx.x * 16 + wave) * (16 * 1024);       // this wave's window
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (DEPTH * 1024);
    const char* lsrc = win + lane * 16;
    const unsigned voff = lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, 0x7fffffff, 0x00020000);
    f4 keep[DEPTH];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (MODE == 0)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lsrc + d * 1024), "s"(lds0 + d * 1024) : "memory");
            else if (MODE == 1)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(keep[d]) : "v"(lsrc + d * 1024) : "memory");
            else if (MODE == 2)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff + d * 1024), "s"(win), "s"(lds0 + d * 1024) : "memory");
            else
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff + d * 1024), "s"(rsrc), "s"(lds0 + d * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
        if (MODE == 1) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) asm volatile("" ::"v"(keep[d]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (iters < 0) sink[0] = 1.f;
}

template <int MODE, int DEPTH>
void run(const char* buf, float* sink, int waves) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 4096 / DEPTH;
    const size_t lds = 150 * 1024;
    CK(hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(256), dim3(64 * waves), lds, 0, buf, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(256), dim3(64 * waves), lds, 0, buf, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_cu = 3.0 * iters * DEPTH * waves * 1024.0;
    const double gbs = bytes_per_cu / (ms * 1e-3) / 1e9;
    printf("%4d %5d %5d   %10.1f  %10.1f   %8.2f   %8.1f\n", MODE, waves, DEPTH, gbs, gbs / 2.1, gbs * 256 / 1e3,
           ms * 1e-3 / 3 * 2.1e9 / (iters * DEPTH));
    fflush(stdout);
}

int main() {
    char* buf;
    const long total = 256L * 16 * 16 * 1024;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    float* sink;
    CK(hipMalloc(&sink, 4));
    printf("mode waves depth   GB/s_per_CU  B/clk@2.1GHz   chip_TB/s  clk_per_piece_per_wave@2.1GHz\n");
    for (int waves : {1, 4, 8, 16}) {
        run<0, 4>(buf, sink, waves);
        run<0, 16>(buf, sink, waves);
        run<1, 4>(buf, sink, waves);
        run<1, 16>(buf, sink, waves);
        run<2, 16>(buf, sink, waves);
        run<3, 16>(buf, sink, waves);
    }
    return 0;
}
