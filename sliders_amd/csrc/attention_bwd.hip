// Flash-style attention backward (loss.backward() through Attention, train_lora_xl.py:345), head dims 64 (SDXL)
// and 40 / 80 / 160 (SD-1.x, as DT = ceil(D/64) d-tiles with zero-filled columns >= D).  gfx950 only; same
// MFMA/LDS idioms as attention.hip (scores recomputed, never materialised in HBM).
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
__device__ __forceinline__ bf16x8 load_row_chunk(const __bf16* base, int col, int D) {
    if (col < D) return *(const bf16x8*)(base + col);
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
    return z;
}

__global__ __launch_bounds__(256) void attn_delta_kernel(const slh_attn_bwd_desc p, int D) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, q, h)
    const long total = (long)p.B * p.Tq * p.H;
    if (idx >= total) return;
    const int h = (int)(idx % p.H);
    const long bq = idx / p.H;
    const int b = (int)(bq / p.Tq), q = (int)(bq - (long)b * p.Tq);
    const __bf16* o = (const __bf16*)p.o + bq * p.ldo + h * D;
    const __bf16* g = (const __bf16*)p.d_o + bq * p.lddo + h * D;
    float s = 0.f;
    for (int i = 0; i < D; i += 8) {
        const bf16x8 a = *(const bf16x8*)(o + i);
        const bf16x8 c = *(const bf16x8*)(g + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)c[e];
    }
    p.delta[((long)b * p.H + h) * p.Tq + q] = s;
}

// XCD-aware block order shared by the backward kernels (see attn_fwd_kernel): virtual block id whose consecutive values
// (same head, next 128-row block) land on the same XCD's L2
__device__ __forceinline__ int xcd_virtual_block() {
    const int vb = blockIdx.x, nblk = gridDim.x;
    const int qd = nblk >> 3, rm = nblk & 7;
    const int xcd = vb & 7, idx = vb >> 3;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
}

// ---- dQ ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const slh_attn_bwd_desc p) {
    __shared__ __attribute__((aligned(16))) char smem[6 * DT * 8192];
    char* sK = smem;                      // [2][DT][64 kv][128 B]
    char* sV = smem + 2 * DT * 8192;      // [2][DT][64 kv][128 B]
    char* sKT = smem + 4 * DT * 8192;     // [2][DT][64 d ][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int frow = lane >> 3, fslot = lane & 7;
    const int vb = xcd_virtual_block();
    const int nqb = (p.Tq + 127) / 128;
    const int h = (vb / nqb) % p.H, b = vb / (nqb * p.H);
    const int q0 = (vb % nqb) * 128 + wave * 32;
    const int D = p.D > 0 ? p.D : 64;
    const __bf16* K = (const __bf16*)p.k;
    const __bf16* V = (const __bf16*)p.v;
    const __bf16* KT = (const __bf16*)p.kt;

    int qrow = q0 + lrow;
    const bool qvalid = qrow < p.Tq;
    qrow = qvalid ? qrow : p.Tq - 1;
    bf16x8 qf[DT * 4], gf[DT * 4];
#pragma unroll
    for (int ks = 0; ks < DT * 4; ++ks) {
        qf[ks] = load_row_chunk((const __bf16*)p.q + ((long)b * p.Tq + qrow) * p.ldq + h * D, ks * 16 + lhi * 8, D);
        gf[ks] = load_row_chunk((const __bf16*)p.d_o + ((long)b * p.Tq + qrow) * p.lddo + h * D, ks * 16 + lhi * 8, D);
    }
    const float lse2 = p.lse[((long)b * p.H + h) * p.Tq + qrow];
    const float delta = p.delta[((long)b * p.H + h) * p.Tq + qrow];
    const float c = p.scale * 1.4426950408889634f;
    const int prow = perm_row(lrow);

    const int nt = (p.Tk + 63) / 64;
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave + 4 * i) * 8 + frow;
                const int ks = fslot ^ ((row >> 1) & 7);
                int kv = t * 64 + row;
                kv = kv < p.Tk ? kv : p.Tk - 1;
                const int col = dt * 64 + ks * 8;
                const bool ok = col < D;
                glds16(ok ? K + ((long)b * p.Tk + kv) * p.ldk + h * D + col : (const __bf16*)slh_zero_page,
                       sK + (buf * DT + dt) * 8192 + (wave + 4 * i) * 1024);
                glds16(ok ? V + ((long)b * p.Tk + kv) * p.ldv + h * D + col : (const __bf16*)slh_zero_page,
                       sV + (buf * DT + dt) * 8192 + (wave + 4 * i) * 1024);
                glds16(KT + (((long)b * p.H + h) * (64 * DT) + dt * 64 + row) * p.ldkt + t * 64 + ks * 8,
                       sKT + (buf * DT + dt) * 8192 + (wave + 4 * i) * 1024);
            }
        }
    };
    f32x16 acc[2 * DT];
#pragma unroll
    for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dd][r] = 0.f;

    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        lds_dma_syncthreads();       // tile t landed (all waves), everybody is past tile t-1
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char* cK = sK + (t & 1) * DT * 8192;
        const char* cV = sV + (t & 1) * DT * 8192;
        const char* cKT = sKT + (t & 1) * DT * 8192;
        bf16x8 ds[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(cK + dt * 8192 + lds_off(kt * 32 + prow, ks * 2 + lhi));
                    const bf16x8 vf = *(const bf16x8*)(cV + dt * 8192 + lds_off(kt * 32 + prow, ks * 2 + lhi));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[dt * 4 + ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, gf[dt * 4 + ks], dp, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = t * 64 + kt * 32 + 16 * (r >> 3) + 8 * lhi + (r & 7);
                float pv = __builtin_amdgcn_exp2f(s[r] * c - lse2);   // raw v_exp_f32 (argument <= 0), like the forward
                if (kv >= p.Tk) pv = 0.f;
                ds[kt][r >> 3][r & 7] = (__bf16)(pv * (dp[r] - delta));
            }
        }
#pragma unroll
        for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const bf16x8 kf = *(const bf16x8*)(cKT + (dd >> 1) * 8192 + lds_off((dd & 1) * 32 + lrow, kstep * 2 + lhi));
                acc[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, ds[kstep >> 1][kstep & 1], acc[dd], 0, 0, 0);
            }
    }
    if (qvalid) {
        __bf16* O = (__bf16*)p.dq + ((long)b * p.Tq + qrow) * p.lddq + h * D;
#pragma unroll
        for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int dcol = dd * 32 + qd * 8 + lhi * 4;
                if (dcol < D) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (__bf16)(acc[dd][qd * 4 + e] * p.scale);
                    *(bf16x4*)(O + dcol) = v;
                }
            }
    }
}

// ---- dK, dV --------------------------------------------------------------------------------------------
// NBUF = 2: double-buffered query tiles; NBUF = 1 (DT = 3: 4 x 3 x 8 KB per buffer) single-buffered
template <int DT, int NBUF>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const slh_attn_bwd_desc p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * NBUF * DT * 8192];
    char* sQ = smem;                          // [NBUF][DT][64 q][128 B]
    char* sG = smem + NBUF * DT * 8192;       // dO
    char* sQT = smem + 2 * NBUF * DT * 8192;  // [NBUF][DT][64 d][128 B]
    char* sGT = smem + 3 * NBUF * DT * 8192;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int frow = lane >> 3, fslot = lane & 7;
    const int vb = xcd_virtual_block();
    const int nkb = (p.Tk + 127) / 128;
    const int h = (vb / nkb) % p.H, b = vb / (nkb * p.H);
    const int k0 = (vb % nkb) * 128 + wave * 32;
    const int D = p.D > 0 ? p.D : 64;
    const __bf16* Q = (const __bf16*)p.q;
    const __bf16* G = (const __bf16*)p.d_o;
    const __bf16* QT = (const __bf16*)p.qt;
    const __bf16* GT = (const __bf16*)p.dot;

    int krow = k0 + lrow;
    const bool kvalid = krow < p.Tk;
    krow = kvalid ? krow : p.Tk - 1;
    bf16x8 kf[DT * 4], vf[DT * 4];
#pragma unroll
    for (int ks = 0; ks < DT * 4; ++ks) {
        kf[ks] = load_row_chunk((const __bf16*)p.k + ((long)b * p.Tk + krow) * p.ldk + h * D, ks * 16 + lhi * 8, D);
        vf[ks] = load_row_chunk((const __bf16*)p.v + ((long)b * p.Tk + krow) * p.ldv + h * D, ks * 16 + lhi * 8, D);
    }
    const float c = p.scale * 1.4426950408889634f;
    const int prow = perm_row(lrow);
    const int nt = (p.Tq + 63) / 64;
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave + 4 * i) * 8 + frow;
                const int ks = fslot ^ ((row >> 1) & 7);
                int q = t * 64 + row;
                q = q < p.Tq ? q : p.Tq - 1;
                const int col = dt * 64 + ks * 8;
                const bool ok = col < D;
                const int o = (buf * DT + dt) * 8192 + (wave + 4 * i) * 1024;
                glds16(ok ? Q + ((long)b * p.Tq + q) * p.ldq + h * D + col : (const __bf16*)slh_zero_page, sQ + o);
                glds16(ok ? G + ((long)b * p.Tq + q) * p.lddo + h * D + col : (const __bf16*)slh_zero_page, sG + o);
                const long trow = (((long)b * p.H + h) * (64 * DT) + dt * 64 + row) * p.ldqt + t * 64 + ks * 8;
                glds16(QT + trow, sQT + o);
                glds16(GT + trow, sGT + o);
            }
        }
    };
    f32x16 dk[2 * DT], dv[2 * DT];
#pragma unroll
    for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dd][r] = 0.f; dv[dd][r] = 0.f; }

    if (NBUF == 2) stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        if (NBUF == 2) {
            lds_dma_syncthreads();    // tile t landed (all waves), everybody is past tile t-1
            if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        } else {
            __syncthreads();          // everybody finished reading the single buffer
            stage(0, t);
            lds_dma_syncthreads();    // tile t landed for all waves
        }
        const int bufo = (NBUF == 2 ? (t & 1) : 0) * DT * 8192;
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
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 qa = *(const bf16x8*)(sQ + bufo + dt * 8192 + lds_off(qt * 32 + prow, ks * 2 + lhi));
                    const bf16x8 ga = *(const bf16x8*)(sG + bufo + dt * 8192 + lds_off(qt * 32 + prow, ks * 2 + lhi));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[dt * 4 + ks], s, 0, 0, 0);     // S[q][kv = lane]
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf[dt * 4 + ks], dp, 0, 0, 0);   // dP[q][kv = lane]
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
                    float pv = __builtin_amdgcn_exp2f(s[r] * c - lv[e]);
                    if (t * 64 + qb + e >= p.Tq) pv = 0.f;
                    pb[qt][hf][e] = (__bf16)pv;
                    dsb[qt][hf][e] = (__bf16)(pv * (dp[r] - dl[e]));
                }
            }
        }
#pragma unroll
        for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const int o = bufo + (dd >> 1) * 8192 + lds_off((dd & 1) * 32 + lrow, kstep * 2 + lhi);
                const bf16x8 ga = *(const bf16x8*)(sGT + o);
                const bf16x8 qa = *(const bf16x8*)(sQT + o);
                dv[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, pb[kstep >> 1][kstep & 1], dv[dd], 0, 0, 0);
                dk[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, dsb[kstep >> 1][kstep & 1], dk[dd], 0, 0, 0);
            }
    }
    if (kvalid) {
        __bf16* DK = (__bf16*)p.dk + ((long)b * p.Tk + krow) * p.lddk + h * D;
        __bf16* DV = (__bf16*)p.dv + ((long)b * p.Tk + krow) * p.lddv + h * D;
#pragma unroll
        for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int dcol = dd * 32 + qd * 8 + lhi * 4;
                if (dcol < D) {
                    bf16x4 a, g;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = (__bf16)(dk[dd][qd * 4 + e] * p.scale);
                        g[e] = (__bf16)dv[dd][qd * 4 + e];
                    }
                    *(bf16x4*)(DK + dcol) = a;
                    *(bf16x4*)(DV + dcol) = g;
                }
            }
    }
}

}  // namespace

extern "C" int slh_attn_bwd(const slh_attn_bwd_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->q && d->k && d->v && d->o && d->d_o && d->lse && d->delta && d->dq && d->kt,
              "slh_attn_bwd: null pointer");
    const int D = d->D > 0 ? d->D : 64;
    SLH_CHECK(D % 8 == 0 && D <= 192, "slh_attn_bwd: head_dim %d unsupported", D);
    SLH_CHECK(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 8 == 0 && d->lddo % 8 == 0 &&
                  d->lddq % 4 == 0 && d->ldkt % 64 == 0,
              "slh_attn_bwd: alignment");
    SLH_CHECK(d->ldkt >= ((d->Tk + 63) / 64) * 64, "slh_attn_bwd: KT must be padded to a multiple of 64 keys");
    hipStream_t s = (hipStream_t)stream;
    const int DT = (D + 63) / 64;
    const long total = (long)d->B * d->Tq * d->H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *d, D);
    const dim3 gq((unsigned)((long)((d->Tq + 127) / 128) * d->H * d->B));
    if (DT == 1) hipLaunchKernelGGL(attn_bwd_dq_kernel<1>, gq, dim3(256), 0, s, *d);
    else if (DT == 2) hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, gq, dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel<3>, gq, dim3(256), 0, s, *d);
    if (d->need_dkv) {
        SLH_CHECK(d->qt && d->dot && d->dk && d->dv, "slh_attn_bwd: dK/dV need qt, dot, dk, dv");
        SLH_CHECK(d->ldqt % 64 == 0 && d->ldqt >= ((d->Tq + 63) / 64) * 64 && d->lddk % 4 == 0 && d->lddv % 4 == 0,
                  "slh_attn_bwd: QT/dOT padding");
        const dim3 gk((unsigned)((long)((d->Tk + 127) / 128) * d->H * d->B));
        if (DT == 1) hipLaunchKernelGGL((attn_bwd_dkv_kernel<1, 2>), gk, dim3(256), 0, s, *d);
        else if (DT == 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, 2>), gk, dim3(256), 0, s, *d);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<3, 1>), gk, dim3(256), 0, s, *d);
    }
    SLH_LAUNCH_CHECK("slh_attn_bwd");
    return 0;
}
