// Flash-style attention backward for head_dim 64 (loss.backward() through Attention, train_lora_xl.py:345).
// gfx950 only; same MFMA/LDS idioms as attention.hip (scores recomputed, never materialised in HBM).
//
//   delta[q]  = sum_d dO[q][d] * O[q][d]
//   P         = exp2(S*c - lse2[q]),  S = Q K^T,  c = scale*log2(e)
//   dP        = dO V^T ;  dS = P * (dP - delta[q])
//   dQ = scale * dS K ;  dK = scale * dS^T Q ;  dV = P^T dO
//
// Two passes, both free of atomics and deterministic:
//   dq kernel  : one wave per 32 queries, loops over key tiles   (tiles: K, V natural; K^T for the dQ product)
//   dkv kernel : one wave per 32 keys,    loops over query tiles (tiles: Q, dO natural; Q^T, dO^T transposed)
// As in the forward the first MFMA of each pair is issued with swapped operands and permuted rows so that the
// probabilities land in registers in exactly the k-order the second MFMA's B operand needs.
// Cross-attention (text K/V carry no gradient in the reference) runs the dq kernel only.
#include "common.h"
#include "../../include/sliders_hip.h"

namespace {

__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ int perm_row(int r) {  // swap bits 2 and 3
    return (r & 3) | (((r >> 3) & 1) << 2) | (((r >> 2) & 1) << 3) | (r & 16);
}

__global__ __launch_bounds__(256) void attn_delta_kernel(const slh_attn_bwd_desc p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, q, h)
    const long total = (long)p.B * p.Tq * p.H;
    if (idx >= total) return;
    const int h = (int)(idx % p.H);
    const long bq = idx / p.H;
    const int b = (int)(bq / p.Tq), q = (int)(bq - (long)b * p.Tq);
    const __bf16* o = (const __bf16*)p.o + bq * p.ldo + h * 64;
    const __bf16* g = (const __bf16*)p.d_o + bq * p.lddo + h * 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bf16x8 a = *(const bf16x8*)(o + i * 8);
        const bf16x8 c = *(const bf16x8*)(g + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)c[e];
    }
    p.delta[((long)b * p.H + h) * p.Tq + q] = s;
}

// ---- dQ ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const slh_attn_bwd_desc p) {
    __shared__ __attribute__((aligned(16))) char smem[6 * 8192];
    char* sK = smem;            // [2][64 kv][128 B]
    char* sV = smem + 16384;    // [2][64 kv][128 B]
    char* sKT = smem + 32768;   // [2][64 d ][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int frow = lane >> 3, fslot = lane & 7;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const __bf16* K = (const __bf16*)p.k;
    const __bf16* V = (const __bf16*)p.v;
    const __bf16* KT = (const __bf16*)p.kt;

    int qrow = q0 + lrow;
    const bool qvalid = qrow < p.Tq;
    qrow = qvalid ? qrow : p.Tq - 1;
    bf16x8 qf[4], gf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *(const bf16x8*)((const __bf16*)p.q + ((long)b * p.Tq + qrow) * p.ldq + h * 64 + ks * 16 + lhi * 8);
        gf[ks] = *(const bf16x8*)((const __bf16*)p.d_o + ((long)b * p.Tq + qrow) * p.lddo + h * 64 + ks * 16 + lhi * 8);
    }
    const float lse2 = p.lse[((long)b * p.H + h) * p.Tq + qrow];
    const float delta = p.delta[((long)b * p.H + h) * p.Tq + qrow];
    const float c = p.scale * 1.4426950408889634f;
    const int prow = perm_row(lrow);

    const int nt = (p.Tk + 63) / 64;
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave + 4 * i) * 8 + frow;
            const int ks = fslot ^ ((row >> 1) & 7);
            int kv = t * 64 + row;
            kv = kv < p.Tk ? kv : p.Tk - 1;
            glds16(K + ((long)b * p.Tk + kv) * p.ldk + h * 64 + ks * 8, sK + buf * 8192 + (wave + 4 * i) * 1024);
            glds16(V + ((long)b * p.Tk + kv) * p.ldv + h * 64 + ks * 8, sV + buf * 8192 + (wave + 4 * i) * 1024);
            glds16(KT + (((long)b * p.H + h) * 64 + row) * p.ldkt + t * 64 + ks * 8,
                   sKT + buf * 8192 + (wave + 4 * i) * 1024);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char* cK = sK + (t & 1) * 8192;
        const char* cV = sV + (t & 1) * 8192;
        const char* cKT = sKT + (t & 1) * 8192;
        bf16x8 ds[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(cK + lds_off(kt * 32 + prow, ks * 2 + lhi));
                const bf16x8 vf = *(const bf16x8*)(cV + lds_off(kt * 32 + prow, ks * 2 + lhi));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, gf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = t * 64 + kt * 32 + 16 * (r >> 3) + 8 * lhi + (r & 7);
                float pv = exp2f(s[r] * c - lse2);
                if (kv >= p.Tk) pv = 0.f;
                ds[kt][r >> 3][r & 7] = (__bf16)(pv * (dp[r] - delta));
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const bf16x8 kf = *(const bf16x8*)(cKT + lds_off(dt * 32 + lrow, kstep * 2 + lhi));
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, ds[kstep >> 1][kstep & 1], acc[dt], 0, 0, 0);
            }
    }
    if (qvalid) {
        __bf16* O = (__bf16*)p.dq + ((long)b * p.Tq + qrow) * p.lddq + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__bf16)(acc[dt][qd * 4 + e] * p.scale);
                *(bf16x4*)(O + dt * 32 + qd * 8 + lhi * 4) = v;
            }
    }
}

// ---- dK, dV --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const slh_attn_bwd_desc p) {
    __shared__ __attribute__((aligned(16))) char smem[8 * 8192];
    char* sQ = smem;             // [2][64 q][128 B]
    char* sG = smem + 16384;     // [2][64 q][128 B]   dO
    char* sQT = smem + 32768;    // [2][64 d][128 B]
    char* sGT = smem + 49152;    // [2][64 d][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int frow = lane >> 3, fslot = lane & 7;
    const int h = blockIdx.y, b = blockIdx.z;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const __bf16* Q = (const __bf16*)p.q;
    const __bf16* G = (const __bf16*)p.d_o;
    const __bf16* QT = (const __bf16*)p.qt;
    const __bf16* GT = (const __bf16*)p.dot;

    int krow = k0 + lrow;
    const bool kvalid = krow < p.Tk;
    krow = kvalid ? krow : p.Tk - 1;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8*)((const __bf16*)p.k + ((long)b * p.Tk + krow) * p.ldk + h * 64 + ks * 16 + lhi * 8);
        vf[ks] = *(const bf16x8*)((const __bf16*)p.v + ((long)b * p.Tk + krow) * p.ldv + h * 64 + ks * 16 + lhi * 8);
    }
    const float c = p.scale * 1.4426950408889634f;
    const int prow = perm_row(lrow);
    const int nt = (p.Tq + 63) / 64;
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave + 4 * i) * 8 + frow;
            const int ks = fslot ^ ((row >> 1) & 7);
            int q = t * 64 + row;
            q = q < p.Tq ? q : p.Tq - 1;
            glds16(Q + ((long)b * p.Tq + q) * p.ldq + h * 64 + ks * 8, sQ + buf * 8192 + (wave + 4 * i) * 1024);
            glds16(G + ((long)b * p.Tq + q) * p.lddo + h * 64 + ks * 8, sG + buf * 8192 + (wave + 4 * i) * 1024);
            glds16(QT + (((long)b * p.H + h) * 64 + row) * p.ldqt + t * 64 + ks * 8,
                   sQT + buf * 8192 + (wave + 4 * i) * 1024);
            glds16(GT + (((long)b * p.H + h) * 64 + row) * p.ldqt + t * 64 + ks * 8,
                   sGT + buf * 8192 + (wave + 4 * i) * 1024);
        }
    };
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const int bufo = (t & 1) * 8192;
        // per-query lse2 / delta straight from global (L1-resident; the arrays are padded by 64 floats)
        const float* cL = p.lse + ((long)b * p.H + h) * p.Tq + t * 64;
        const float* cD = p.delta + ((long)b * p.H + h) * p.Tq + t * 64;
        bf16x8 pb[2][2], dsb[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qa = *(const bf16x8*)(sQ + bufo + lds_off(qt * 32 + prow, ks * 2 + lhi));
                const bf16x8 ga = *(const bf16x8*)(sG + bufo + lds_off(qt * 32 + prow, ks * 2 + lhi));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);     // S[q][kv = lane]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf[ks], dp, 0, 0, 0);   // dP[q][kv = lane]
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int qb = qt * 32 + 16 * hf + 8 * lhi;
                float lv[8], dl[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { lv[e] = cL[qb + e]; dl[e] = cD[qb + e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = hf * 8 + e;
                    float pv = exp2f(s[r] * c - lv[e]);
                    if (t * 64 + qb + e >= p.Tq) pv = 0.f;
                    pb[qt][hf][e] = (__bf16)pv;
                    dsb[qt][hf][e] = (__bf16)(pv * (dp[r] - dl[e]));
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const bf16x8 ga = *(const bf16x8*)(sGT + bufo + lds_off(dt * 32 + lrow, kstep * 2 + lhi));
                const bf16x8 qa = *(const bf16x8*)(sQT + bufo + lds_off(dt * 32 + lrow, kstep * 2 + lhi));
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, pb[kstep >> 1][kstep & 1], dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, dsb[kstep >> 1][kstep & 1], dk[dt], 0, 0, 0);
            }
    }
    if (kvalid) {
        __bf16* DK = (__bf16*)p.dk + ((long)b * p.Tk + krow) * p.lddk + h * 64;
        __bf16* DV = (__bf16*)p.dv + ((long)b * p.Tk + krow) * p.lddv + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 a, g;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = (__bf16)(dk[dt][qd * 4 + e] * p.scale);
                    g[e] = (__bf16)dv[dt][qd * 4 + e];
                }
                *(bf16x4*)(DK + dt * 32 + qd * 8 + lhi * 4) = a;
                *(bf16x4*)(DV + dt * 32 + qd * 8 + lhi * 4) = g;
            }
    }
}

}  // namespace

extern "C" int slh_attn_bwd(const slh_attn_bwd_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->q && d->k && d->v && d->o && d->d_o && d->lse && d->delta && d->dq && d->kt,
              "slh_attn_bwd: null pointer");
    SLH_CHECK(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 8 == 0 && d->lddo % 8 == 0 &&
                  d->lddq % 4 == 0 && d->ldkt % 64 == 0,
              "slh_attn_bwd: alignment");
    SLH_CHECK(d->ldkt >= ((d->Tk + 63) / 64) * 64, "slh_attn_bwd: KT must be padded to a multiple of 64 keys");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)d->B * d->Tq * d->H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *d);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((d->Tq + 127) / 128, d->H, d->B), dim3(256), 0, s, *d);
    if (d->need_dkv) {
        SLH_CHECK(d->qt && d->dot && d->dk && d->dv, "slh_attn_bwd: dK/dV need qt, dot, dk, dv");
        SLH_CHECK(d->ldqt % 64 == 0 && d->ldqt >= ((d->Tq + 63) / 64) * 64 && d->lddk % 4 == 0 && d->lddv % 4 == 0,
                  "slh_attn_bwd: QT/dOT padding");
        hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((d->Tk + 127) / 128, d->H, d->B), dim3(256), 0, s, *d);
    }
    SLH_LAUNCH_CHECK("slh_attn_bwd");
    return 0;
}
