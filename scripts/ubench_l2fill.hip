// Micro-benchmark (development aid, round 6): the L2 -> LDS fill rate a GEMM's operand stream can count on.
// scripts/ubench_fill.hip streams a 16 KB window per wave, which the CU's 32 KB L1 serves; here every workgroup (one per CU, 4 waves)
// walks a window that does NOT fit the L1 but stays in the XCD's 4 MB L2, by LDS-DMA (global_load_lds_dwordx4, 1 KB per instruction,
// DEPTH pieces in flight per wave):
//   share 0: every CU its own window (win KB each; 32 CUs x win must fit the 4 MB L2)
//   share 1: all CUs walk the SAME window in the same order (a W panel shared by the row tiles of an XCD)
//   share 2: the same window, every CU starting at a different piece (skewed walk)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_l2fill.hip -o scripts/ubench_l2fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void fill_kernel(const char* base, int win_kb, int share, int rounds, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[4 * DEPTH * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long win = (long)win_kb * 1024;
    const char* w0 = base + (share ? 0 : (long)blockIdx.x * win);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (DEPTH * 1024);
    const int npieces = win_kb;                      // 1 KB pieces
    int pc = wave + (share == 2 ? (int)(blockIdx.x >> 3) * 16 : 0);
    pc %= npieces;
    const int total = rounds * (npieces / 4);        // pieces this wave issues
    for (int it = 0; it < total; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* src = w0 + (long)pc * 1024 + lane * 16;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + d * 1024) : "memory");
            pc += 4;
            pc = pc >= npieces ? pc - npieces : pc;
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (rounds < 0) sink[0] = 1.f;
}

template <int DEPTH>
void run(const char* buf, float* sink, int win_kb, int share) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int rounds = (64 * 1024) / win_kb;        // 64 MB per CU per launch
    hipLaunchKernelGGL((fill_kernel<DEPTH>), dim3(256), dim3(256), 0, 0, buf, win_kb, share, rounds, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((fill_kernel<DEPTH>), dim3(256), dim3(256), 0, 0, buf, win_kb, share, rounds, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_cu = 3.0 * rounds * (win_kb / 4) * 4 * 1024.0;
    const double gbs = bytes_per_cu / (ms * 1e-3) / 1e9;
    printf("%6d %6d %6d   %10.1f  %10.1f   %8.2f\n", win_kb, share, DEPTH, gbs, gbs / 2.1, gbs * 256 / 1e3);
    fflush(stdout);
}

int main() {
    char* buf;
    const long total = 256L * 4096 * 1024;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    float* sink;
    CK(hipMalloc(&sink, 4));
    printf("win_KB  share  depth   GB/s_per_CU  B/clk@2.1GHz   chip_TB/s\n");
    for (int share : {0, 1, 2})
        for (int win_kb : {16, 64, 128, 1024, 4096}) {
            if (share == 0 && win_kb > 4096) continue;
            run<8>(buf, sink, win_kb, share);
            run<16>(buf, sink, win_kb, share);
            run<32>(buf, sink, win_kb, share);
        }
    return 0;
}
