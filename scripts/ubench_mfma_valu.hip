// Micro-benchmark (development aid, GPU box): do the matrix pipe and the vector ALU of ONE SIMD overlap, and what does a
// v_exp_f32 / v_pk_fma_f32 / v_fma_f32 cost there?  One workgroup per CU; 4 waves (one per SIMD) or 8 (two per SIMD: waves w
// and w+4 share a SIMD).  build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/ubench_mfma_valu.hip -o scripts/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define PIN __builtin_amdgcn_sched_barrier(0)

struct Work {
    f32x16 acc[4];
    bf16x8 a, b;
    float x[16];
    f32x2 y[8];
};

__device__ __forceinline__ void mfma16(Work& w) {   // 16 products, 4 independent chains of 4
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) w.acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a, w.b, w.acc[j], 0, 0, 0);
}
__device__ __forceinline__ void fma64(Work& w) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) w.x[j] = __builtin_fmaf(w.x[j], 0.999f, 0.001f);
}
__device__ __forceinline__ void exp32(Work& w) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) w.x[j] = __builtin_amdgcn_exp2f(w.x[j]);
}
__device__ __forceinline__ void pk32(Work& w) {
    const f32x2 k = {0.999f, 0.999f}, k2 = {0.001f, 0.001f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) w.y[j] = w.y[j] * k + k2;
}
// one MFMA followed by 4 fma + 2 exp, sixteen times: what a perfectly interleaved softmax would look like
__device__ __forceinline__ void interleaved(Work& w) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        w.acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a, w.b, w.acc[i & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) w.x[(i * 4 + j) & 15] = __builtin_fmaf(w.x[(i * 4 + j) & 15], 0.999f, 0.001f);
#pragma unroll
        for (int j = 0; j < 2; ++j) w.x[(i * 2 + j + 8) & 15] = __builtin_amdgcn_exp2f(w.x[(i * 2 + j + 8) & 15]);
        PIN;
    }
}

// attention-like dependent structure: 8 score MFMAs (2 chains of 4) -> 32 fma + 32 exp + 16 cvt on the RESULTS -> 8 MFMAs that
// take the converted values as their B operand
__device__ __forceinline__ void dependent(Work& w) {
    f32x16 s0 = {0}, s1 = {0};
#pragma unroll
    for (int i = 0; i < 4; ++i) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a, w.b, s0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.b, w.a, s1, 0, 0, 0);
    bf16x8 pb[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], 0.01f, -1.f));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], 0.01f, -1.f));
        w.x[r] += e0 + e1;
        pb[r >> 3][r & 7] = (__bf16)e0;
        pb[2 + (r >> 3)][r & 7] = (__bf16)e1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) w.acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a, pb[i], w.acc[0], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w.acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a, pb[i], w.acc[1], 0, 0, 0);
}

template <int mode>
__global__ __launch_bounds__(1024) void bench(int iters, float* out, unsigned long long* clk) {
    Work w;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) w.acc[j][r] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { w.a[e] = (__bf16)(0.001f * lane); w.b[e] = (__bf16)(0.002f * e); }
#pragma unroll
    for (int j = 0; j < 16; ++j) w.x[j] = -0.01f * (lane + j);
#pragma unroll
    for (int j = 0; j < 8; ++j) w.y[j] = f32x2{0.01f * lane, 0.02f * j};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const bool second = wave >= 4;
    for (int it = 0; it < ((mode == 6 || mode == 7 || mode == 9) ? 0 : iters); ++it) {
        if constexpr (mode == 0 || mode == 10) mfma16(w);
        if constexpr (mode == 1 || mode == 11 || mode == 12) fma64(w);
        if constexpr (mode == 2 || mode == 13) exp32(w);
        if constexpr (mode == 3) pk32(w);
        if constexpr (mode == 4) interleaved(w);
        if constexpr (mode >= 15) dependent(w);
        if constexpr (mode == 5 || mode == 8 || mode == 14) { mfma16(w); PIN; fma64(w); PIN; exp32(w); }   // the same work in blocks
        PIN;
    }
    // 8 waves, waves w and w+4 on one SIMD: separate loops per role so that no value is merged between the roles
    if constexpr (mode == 6 || mode == 7 || mode == 9) {
        if (second) {
            for (int it = 0; it < iters; ++it) {
                fma64(w); PIN; exp32(w); PIN;
                if constexpr (mode == 9) { mfma16(w); PIN; }
            }
        } else if (mode != 7) {
            for (int it = 0; it < iters; ++it) {
                mfma16(w); PIN;
                if constexpr (mode == 9) { fma64(w); PIN; exp32(w); PIN; }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += w.acc[j][0] + w.acc[j][7];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += w.x[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += w.y[j][0] + w.y[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) { clk[wave * 2] = t1 - t0; clk[wave * 2 + 1] = r1 - r0; }
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 32 * 8);
    const char* names[] = {"16 mfma (4 waves)", "64 v_fma_f32 (4 waves)", "32 v_exp_f32 (4 waves)", "32 v_pk_fma_f32 (4 waves)",
                           "16 x [mfma, 4 fma, 2 exp] interleaved (4 waves)", "16 mfma ; 64 fma ; 32 exp in blocks (4 waves)",
                           "8 waves: w0-3 16 mfma | w4-7 64 fma + 32 exp", "8 waves: w4-7 64 fma + 32 exp, w0-3 idle",
                           "8 waves: all [16 mfma ; 64 fma ; 32 exp] free running", "8 waves: same, w4-7 in the opposite phase order",
                           "8 waves: all 16 mfma", "8 waves: all 64 v_fma_f32", "16 waves: all 64 v_fma_f32", "8 waves: all 32 v_exp_f32",
                           "16 waves: all [16 mfma ; 64 fma ; 32 exp] free running",
                           "4 waves: [8 mfma -> 32 fma+32 exp+16 cvt on results -> 8 mfma]", "8 waves: the same dependent structure",
                           "12 waves: the same dependent structure", "16 waves: the same dependent structure"};
    const int threads[] = {256, 256, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 512, 1024, 256, 512, 768, 1024};
    const int iters = 4000;
    typedef void (*kern_t)(int, float*, unsigned long long*);
    const kern_t kerns[] = {bench<0>, bench<1>, bench<2>, bench<3>, bench<4>, bench<5>, bench<6>, bench<7>, bench<8>, bench<9>, bench<10>, bench<11>, bench<12>, bench<13>, bench<14>, bench<15>, bench<16>, bench<17>, bench<18>};
    for (int mode = 0; mode < 19; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kerns[mode], dim3(256), dim3(threads[mode]), 0, 0, 100, out, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kerns[mode], dim3(256), dim3(threads[mode]), 0, 0, iters, out, clk);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[32]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        // s_memtime = shader clock, s_memrealtime = 100 MHz
        printf("%-58s %8.1f ns/iter  wave0 %6.0f shader cycles/iter (clock %.2f GHz)  wave4 %6.0f\n", names[mode],
               ms * 1e6 / iters, (double)h[0] / iters, h[1] ? (double)h[0] / h[1] * 0.1 : 0.0, (double)h[8] / iters);
    }
    return 0;
}
