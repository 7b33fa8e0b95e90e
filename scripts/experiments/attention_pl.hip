// EXPERIMENT (round 6), NOT part of libsliders_hip.so: measured equal to the shipped kernel (profiles/r06_attn_pipelined.txt), kept
// for the record.  To try it again: copy this file and attention_pl.h into sliders_amd/csrc, add attention_pl.hip to SRCS_HIP with
// `-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form`, and dispatch `slh_attn_pl_launch(d, ring, s)` from launch_fwd<1> (attention.hip)
// where slh_attn_pl_ok(d, ring) and the launch is not one the key-split form takes.
//
// Attention forward, D = 64, whole key tiles: the SOFTWARE-PIPELINED form of attn_fwd_kernel<4, 1, false> (attention.hip).
// gfx950 only.  Same arithmetic in the same order - the results are bit-identical to those kernels - but a wave's instruction
// stream is built so that the matrix pipe and the vector ALU of its SIMD work at the same time:
//
//   iteration t of a wave:   MFMA   P(t-1) V(t-1) -> O          (8 products)      VALU   softmax of S(t): max, exp2, row sum,
//                                   K(t+1) Q^T    -> S(t+1)     (8 products)             bf16 P(t)
//
// The sixteen products of an iteration do not depend on its softmax (they use the previous tile's P and produce the next tile's
// scores), so each is followed by one slice of the ~150 vector instructions, pinned in that order with sched_barrier: a wave that
// runs [8 MFMA; softmax; 8 MFMA] in blocks overlaps with its SIMD neighbours by ~25 % only, the interleaved stream of one wave
// runs both pipes (scripts/ubench_mfma_valu.hip, profiles/r03_ubench_mfma_valu.txt: 680 cycles against 1040 for 16 MFMA + 64 fma +
// 32 exp).  Costs two score tiles and two P tiles in registers (~190 VGPRs: two waves per SIMD).
//
// Workgroup = four query tiles of 32 (one wave each) sharing the key tiles through a ring of S LDS buffers, each one pair
// {K(t+1), V^T(t-1)} - what iteration t multiplies.
//
// Replaces diffusers-0.20.2 Attention + XFormersAttnProcessor (trainscripts/textsliders/train_lora_xl.py:80,
// train_util.py:242-247): softmax(Q K^T / sqrt(d)) V per (sample, head).
#include "common.h"
#include "attention_pl.h"

namespace {

__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

#define PL_PIN __builtin_amdgcn_sched_barrier(0)
#ifndef PL_PROBE
#define PL_PROBE 0
#endif
// A value is computed HERE: without a use the optimiser sees inside the slice, it sinks a slice's arithmetic into the block where
// the next iteration reads it (behind the rescale branch) and the interleaving is gone.  Empty statements, no instruction.
#define PL_HERE(x) asm volatile("" : "+v"(x))
#define PL_HERE2(x, y) asm volatile("" : "+v"(x), "+v"(y))

struct PlLane {
    int lrow, lhi, prow;
    float c;
};

// One pipelined iteration.  cK / cV: the K tile of step t+1 and the V^T tile of step t-1 (this iteration's buffer);
// s_cur = S(t) (complete), s_nxt <- S(t+1); pb_prev = P(t-1), pb_cur <- P(t).
__device__ __forceinline__ void pl_step(const char* cK, const char* cV, const bf16x8 (&qf)[4], const f32x16 (&s_cur)[2],
                                        f32x16 (&s_nxt)[2], const bf16x8 (&pb_prev)[2][2], bf16x8 (&pb_cur)[2][2], f32x16 (&o)[2],
                                        float& m_run, float& l_run, const PlLane& L) {
    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // product i = 4 j + w: w = 0, 1: O^T block w += V^T P^T over key step j; w = 2, 3: S^T block (w - 2) += K Q^T over d step j.
    // (the O products first: the rescale at the end of the iteration waits for them; the score products end the iteration and the
    // barrier + staging of the next one stand between them and the first read of S)
    auto frag = [&](int i) -> bf16x8 {
        const int j = i >> 2, w = i & 3;
        return w < 2 ? *(const bf16x8*)(cV + lds_off(w * 32 + L.lrow, j * 2 + L.lhi))
                     : *(const bf16x8*)(cK + lds_off((w - 2) * 32 + L.prow, j * 2 + L.lhi));
    };
    bf16x8 fr[16];
#if PL_PROBE & 2
    auto frag2 = [&](int i) -> bf16x8 { return qf[i & 3]; };
#define frag frag2
#endif
    fr[0] = frag(0); fr[1] = frag(1); fr[2] = frag(2);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, alpha = 1.f, mc = 0.f, ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i + 3 < 16) fr[i + 3] = frag(i + 3);
        const int j = i >> 2, w = i & 3;
#if !(PL_PROBE & 8)
        if (w < 2) o[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], pb_prev[j >> 1][j & 1], o[w], 0, 0, 0);
        else if (j == 0) s_nxt[w - 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], qf[0], kZero16, 0, 0, 0);
        else s_nxt[w - 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], qf[j], s_nxt[w - 2], 0, 0, 0);
#else
        asm volatile("" ::"v"(fr[i]));
#endif
        // ---- the vector slice that rides behind product i ----
#if PL_PROBE & 4
        if (false) {
#else
        if (i == 0) {
#endif
            a0 = s_cur[0][0]; a1 = s_cur[0][8];
#pragma unroll
            for (int r = 1; r < 8; ++r) { a0 = fmaxf(a0, s_cur[0][r]); a1 = fmaxf(a1, s_cur[0][8 + r]); }
            PL_HERE2(a0, a1);
        } else if (i == 1) {
            a2 = s_cur[1][0]; a3 = s_cur[1][8];
#pragma unroll
            for (int r = 1; r < 8; ++r) { a2 = fmaxf(a2, s_cur[1][r]); a3 = fmaxf(a3, s_cur[1][8 + r]); }
            PL_HERE2(a2, a3);
        } else if (i == 2) {
            float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
            // the other half of the wave holds the other 32 keys of this query: v_permlane32_swap leaves the low half's value in
            // both halves of one register and the high half's in the other
            // (from asm: with the builtin and one value in both operands hipcc dropped the second result - the maximum came out as
            // the low half's alone)
            float lo = mx, hi = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
            mx = fmaxf(lo, hi);
            const float m_new = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * L.c);
            mc = m_new * L.c;
            m_run = m_new;
            PL_HERE2(alpha, mc);
        } else if (!(PL_PROBE & 4)) {
            // sixteen pairs of scores over slices 3 .. 15: two pairs behind products 3, 4, 5, one pair behind each later one
            const int q0 = i <= 5 ? (i - 3) * 2 : i - 6 + 6, nq = i <= 5 ? 2 : 1;
#pragma unroll
            for (int q = q0; q < q0 + nq; ++q) {
                const int kt = q >> 3, r = (q & 7) * 2;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kt][r + e], L.c, -mc));
                    ps += pv;
                    pb_cur[kt][(r + e) >> 3][(r + e) & 7] = (__bf16)pv;
                }
                PL_HERE(ps);
                if ((r & 7) == 6) PL_HERE(pb_cur[kt][r >> 3]);       // an operand of the next iteration's products is complete
            }
        }
        PL_PIN;
    }
    PL_HERE2(s_nxt[0], s_nxt[1]);
    l_run = l_run * alpha + ps;
    // the running maximum settles after the first tiles: rescale only when some row of the wave moved (alpha == 1 exactly otherwise)
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
    }
}

// S = ring depth: pair t = {K tile of step t+1, V^T tile of step t-1} lives in buffer t mod S and is requested S-1 steps before
// the step that multiplies it.  (One step is ~1.2 us with two waves per SIMD and an L2 -> LDS copy takes longer than that to
// land under load: with the request one step ahead every step began by waiting for it - taking the copies out of the S = 2 form
// took 24 of its 140 us at T = 4096, profiles/r06_attn_pipelined.txt.)
template <int S>
__global__ __launch_bounds__(256, 2) void attn_fwd_pl_kernel(const slh_attn_desc p) {
    constexpr int NW = 4, NG = 8 / NW;                   // four query tiles of 32 share the key tiles; pieces per wave and tile
    constexpr int U = S % 2 == 0 ? S : 2 * S;            // steps per loop trip: buffer index and register roles repeat
    __shared__ __attribute__((aligned(16))) char smem[S * 16384];       // [S buffers][K tile 8 KB | V^T tile 8 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int qi = __builtin_amdgcn_readfirstlane(tid >> 6);
    PlLane L;
    L.lrow = lane & 31; L.lhi = lane >> 5;
    L.prow = (L.lrow & 3) | (((L.lrow >> 3) & 1) << 2) | (((L.lrow >> 2) & 1) << 3) | (L.lrow & 16);    // key row order of the S^T A operand
    L.c = p.scale * 1.4426950408889634f;
    const int frow = lane >> 3, fslot = lane & 7;
    int vb = blockIdx.x;
    const int nqb = p.Tq / (32 * NW);
    {
        // XCD-aware order (as attn_fwd_kernel): the query blocks that share one head's K / V meet in one L2
        const int nblk = gridDim.x;
        const int qd = nblk >> 3, rm = nblk & 7;
        const int xcd = vb & 7, idx = vb >> 3;
        vb = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int qb = vb % nqb, hb = vb / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int qrow = qb * (32 * NW) + qi * 32 + L.lrow;
    const int vt_heads = p.vt_batch_heads > 0 ? p.vt_batch_heads : p.H;
    const __bf16* Q = (const __bf16*)p.q;
    const __bf16* K = (const __bf16*)p.k;
    const __bf16* VT = (const __bf16*)p.vt;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Q + ((long)b * p.Tq + qrow) * p.ldq + h * 64 + ks * 16 + L.lhi * 8);

    const int nt = p.Tk >> 6;                            // key tiles (a multiple of U, >= S: slh_attn_pl_ok)
    // staging: wave qi copies the 8-row groups qi and qi + 4 of a tile.  Past the ends both streams read the zero page with
    // zero strides: every pair is the same number of pieces, so one counted wait serves every step.
    const __bf16* kp;
    const __bf16* vp;
    long kgrp, kstep, vgrp, vstep;
    {
        const int row = qi * 8 + frow;
        const int ks = fslot ^ ((row >> 1) & 7);          // the same for row + 32
        kp = K + ((long)b * p.Tk + row) * p.ldk + h * 64 + ks * 8;
        kgrp = (long)(8 * NW) * p.ldk; kstep = 64L * p.ldk;
        vp = (const __bf16*)slh_zero_page;                // (pairs -1 and 0 have no V^T tile: zeros, multiplied by P(-1) = 0)
        vgrp = 0; vstep = 0;
    }
    const __bf16* const v0 = VT + (((long)b * vt_heads + h) * 64 + qi * 8 + frow) * p.ldvt + (fslot ^ (((qi * 8 + frow) >> 1) & 7)) * 8;
    const unsigned base = lds_addr_of(smem);
    auto stage = [&](const int buf) {
#pragma unroll
        for (int i = 0; i < NG; ++i) glds16_hidden(kp + i * kgrp, base + buf * 16384 + (qi + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < NG; ++i) glds16_hidden(vp + i * vgrp, base + buf * 16384 + 8192 + (qi + NW * i) * 1024);
        kp += kstep;
        vp += vstep;
    };
    // prologue: pairs -1 .. S-2 (pair -1 = {K(0)} in buffer S-1)
    int kleft = nt;                                       // K tiles not yet requested
    auto next_pair = [&](const int buf, const bool v_begins) {
        if (kleft == 0) { kp = (const __bf16*)slh_zero_page; kgrp = 0; kstep = 0; }
        if (v_begins) { vp = v0; vgrp = (long)(8 * NW) * p.ldvt; vstep = 64; }
        stage(buf);
        kleft -= kleft > 0 ? 1 : 0;
    };
    next_pair(S - 1, false);                              // {K(0)}
    next_pair(0, false);                                  // {K(1)}
#pragma unroll
    for (int j = 1; j <= S - 2; ++j) next_pair(j, j == 1);       // {K(j+1), V^T(j-1)}
    if (S == 2) { vp = v0; vgrp = (long)(8 * NW) * p.ldvt; vstep = 64; }      // (every pair the loop requests carries a V^T tile)
    int vleft = nt - (S - 2);                             // V^T tiles not yet requested
    f32x16 o[2], sA[2], sB[2];
    bf16x8 pA[2][2], pB[2][2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] = 0.f;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 8; ++e) pB[x][y][e] = (__bf16)0.f;
    float m_run = -1e30f, l_run = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));     // the Q loads are waited for here, not inside the loop
    // pair t has landed when at most S-2 younger pairs are in flight (2 NG pieces each)
#define PL_WAIT_PAIR() asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * 2 * NG) : "memory")
    // S(0) = K(0) Q^T, not overlapped with anything
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * 2 * NG) : "memory");
    __builtin_amdgcn_s_barrier();
    {
        const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const char* cK = smem + (S - 1) * 16384;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(cK + lds_off(kt * 32 + L.prow, ks * 2 + L.lhi));
                if (ks == 0) sA[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0], kZero16, 0, 0, 0);
                else sA[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sA[kt], 0, 0, 0);
            }
    }
    for (int t = 0; t < nt; t += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // step t + u multiplies pair t + u (buffer u mod S) and requests pair t + u + S - 1 into the buffer step t + u - 1 left
            PL_WAIT_PAIR();
            __builtin_amdgcn_s_barrier();
            if (kleft == 0) { kp = (const __bf16*)slh_zero_page; kgrp = 0; kstep = 0; }
            if (vleft == 0) { vp = (const __bf16*)slh_zero_page; vgrp = 0; vstep = 0; }
            stage((u + S - 1) % S);
            kleft -= kleft > 0 ? 1 : 0;
            vleft -= vleft > 0 ? 1 : 0;
            const char* buf = smem + (u % S) * 16384;
            if (u % 2 == 0) pl_step(buf, buf + 8192, qf, sA, sB, pB, pA, o, m_run, l_run, L);
            else pl_step(buf, buf + 8192, qf, sB, sA, pA, pB, o, m_run, l_run, L);
        }
    }
    // the last tile's P V: V^T(nt-1) is in pair nt
    PL_WAIT_PAIR();
    __builtin_amdgcn_s_barrier();
    {
        const char* cV = smem + (nt % S) * 16384 + 8192;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int kstep2 = 0; kstep2 < 4; ++kstep2) {
                const bf16x8 vf = *(const bf16x8*)(cV + lds_off(dd * 32 + L.lrow, kstep2 * 2 + L.lhi));
                o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pB[kstep2 >> 1][kstep2 & 1], o[dd], 0, 0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the zero-page pieces of the pairs past the end)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    __bf16* O = (__bf16*)p.o + ((long)b * p.Tq + qrow) * p.ldo + h * 64;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)(o[dd][qd * 4 + e] * inv);
            *(bf16x4*)(O + dd * 32 + qd * 8 + L.lhi * 4) = v;
        }
    if (p.lse && L.lhi == 0) p.lse[((long)b * p.H + h) * p.Tq + qrow] = m_run * L.c + log2f(l_tot);
}

}  // namespace

// form = ring depth (2 or 4)
bool slh_attn_pl_ok(const slh_attn_desc* d, int form) {
    const int D = d->D > 0 ? d->D : 64;
    if (D != 64 || (d->Tk & 63) != 0 || d->Tq % 128 != 0) return false;
    if (form != 2 && form != 4) return false;
    const int nt = d->Tk / 64;
    return nt % form == 0 && nt >= form;
}

int slh_attn_pl_blocks(const slh_attn_desc* d, int form) {
    (void)form;
    return (d->Tq / 128) * d->H * d->B;
}

int slh_attn_pl_launch(const slh_attn_desc* d, int form, hipStream_t s) {
    SLH_CHECK(slh_attn_pl_ok(d, form), "slh_attn_fwd: the pipelined form (ring of %d) cannot run this launch", form);
    const int grid = slh_attn_pl_blocks(d, form);
    if (form == 2) slh_launch<attn_fwd_pl_kernel<2>>(grid, 256, s, *d, "attn_fwd_pl_kernel<2>");
    else slh_launch<attn_fwd_pl_kernel<4>>(grid, 256, s, *d, "attn_fwd_pl_kernel<4>");
    SLH_LAUNCH_CHECK("slh_attn_fwd (pipelined)");
    return 0;
}
