// Micro-benchmark (development aid, round 6): L2 -> LDS fill rate of the access SHAPES a GEMM operand stream is made of (the contiguous
// case is scripts/ubench_l2fill.hip).  One workgroup per CU, 4 waves; CU b walks panel (b / 8) % 16 of a [2048 rows][pitch B] matrix - 128
// rows - along the row, 1280 B of each row, again and again (2.6 MB in all: L2-resident), 16 pieces in flight per wave:
//   mode 0: piece = 8 rows x 128 B (full cache lines, the 64-deep K tiles of gemm.hip / gemm5.hip)
//   mode 1: piece = 16 rows x 64 B (half lines, the 32-deep half tiles of gemm7.hip)
//   mode 2: piece = 1 KB contiguous (rows of a packed weight block: 8 rows x 128 B at 128-byte pitch)
//   mode 3: piece = 16 rows x 64 B at 128-byte pitch (halves of packed weight rows)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_l2rows.hip -o scripts/ubench_l2rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const char* base, int pitch, int rounds, float* sink) {
    constexpr int DEPTH = 16;
    __shared__ __attribute__((aligned(16))) char smem[4 * DEPTH * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int panel = (blockIdx.x >> 3) & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (DEPTH * 1024);
    // this wave's 32 rows of the panel; a "column step" = 128 B (modes 0, 2) or 64 B (modes 1, 3) along the row
    const char* p0;
    int rows_per_piece, colstep, ncol;
    if (MODE == 0) { rows_per_piece = 8; colstep = 128; ncol = 10; p0 = base + ((long)(panel * 128 + wave * 32 + (lane >> 3)) * pitch) + (lane & 7) * 16; }
    else if (MODE == 1) { rows_per_piece = 16; colstep = 64; ncol = 20; p0 = base + ((long)(panel * 128 + wave * 32 + (lane >> 2)) * pitch) + (lane & 3) * 16; }
    else if (MODE == 2) { rows_per_piece = 8; colstep = 8192; ncol = 10; p0 = base + (long)panel * 163840 + (wave * 32 + (lane >> 3)) * 128 + (lane & 7) * 16; }
    else { rows_per_piece = 16; colstep = 64; ncol = 20; p0 = base + (long)panel * 163840 + (wave * 32 + (lane >> 2)) * 128 + (lane & 3) * 16; }
    const int pieces_per_col = 32 / rows_per_piece;
    const long rowstep = (MODE >= 2) ? (long)rows_per_piece * 128 : (long)rows_per_piece * pitch;
    int issued = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int c = 0; c < ncol; ++c) {
            for (int q = 0; q < pieces_per_col; ++q) {
                long off;
                if (MODE == 3) off = (long)(c >> 1) * 8192 + (c & 1) * 64 + q * rowstep;
                else off = (long)c * colstep + q * rowstep;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p0 + off), "s"(lds0 + (issued & (DEPTH - 1)) * 1024) : "memory");
                if ((++issued & 7) == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (rounds < 0) sink[0] = 1.f;
}

template <int MODE>
void run(const char* buf, float* sink, int pitch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int rounds = 400;
    hipLaunchKernelGGL((fill_kernel<MODE>), dim3(256), dim3(256), 0, 0, buf, pitch, rounds, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((fill_kernel<MODE>), dim3(256), dim3(256), 0, 0, buf, pitch, rounds, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_cu = 3.0 * rounds * 128 * 1280.0;
    const double gbs = bytes_per_cu / (ms * 1e-3) / 1e9;
    printf("%4d %6d   %10.1f  %10.1f   %8.2f\n", MODE, pitch, gbs, gbs / 2.1, gbs * 256 / 1e3);
    fflush(stdout);
}

int main() {
    char* buf;
    const long total = 64L << 20;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    float* sink;
    CK(hipMalloc(&sink, 4));
    printf("mode  pitch   GB/s_per_CU  B/clk@2.1GHz   chip_TB/s\n");
    for (int pitch : {2560, 2560 + 128, 1280, 5120, 10240, 20480}) { run<0>(buf, sink, pitch); run<1>(buf, sink, pitch); }
    run<2>(buf, sink, 128);
    run<3>(buf, sink, 128);
    return 0;
}
