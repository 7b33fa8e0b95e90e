// Flash-style attention forward, bf16 MFMA 32x32x16, fp32 online softmax kept entirely in registers.  gfx950 only.
// Head dims: 64 (every SDXL layer) and 40 / 80 / 160 (SD-1.x), handled as DT = ceil(D/64) 64-wide d-tiles with the
// columns >= D zero-filled on the way into LDS / registers.
//
// Replaces diffusers-0.20.2 Attention + XFormersAttnProcessor (enabled at trainscripts/textsliders/
// train_lora.py:68, config `other.use_xformers`): softmax(Q K^T / sqrt(d)) V per (sample, head).
//
// Layout trick (no cross-lane softmax): the score tile is computed TRANSPOSED, S^T = K Q^T, so a lane owns
// one query column and its 32 scores of a 64-key tile sit in its own registers; the key rows fed to the MFMA
// A operand are permuted (bits 2 and 3 of the row index swapped) so that the C-layout register order is
// exactly the k-order the B operand of the second MFMA (O^T = V^T P^T) expects - P never leaves registers.
// V is consumed transposed (VT[d][kv], produced by slh_transpose_heads) so both LDS tiles are read with
// conflict-free swizzled ds_read_b128 exactly like the GEMM.
#include "common.h"
#include "../../include/sliders_hip.h"

namespace {

__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

// NW waves per workgroup: 4 (128 queries) or 2 (64 queries, used when the grid would not fill the chip)
// TAIL: Tk is not a multiple of 64 (cross-attention, Tk = 77): the key columns past Tk of the last tile are masked.  A
// template flag because hipcc if-converts the mask into ~50 selects per tile that every tile of every launch would execute.
// (Measured and not adopted - the A/B branches were taken out of this file in round 6, `git show 46d9a74:sliders_amd/csrc/attention.hip`
// has them: all fragment reads of a phase issued ahead of its MFMAs (~140 VGPRs, 3 waves per SIMD instead of 4: 412 vs 406 us over the
// pass shapes - co-resident waves already hide the LDS latency in front of every MFMA), row sums on the MFMA, s_setprio around the score
// MFMAs, packed fp32 softmax arithmetic; numbers in profiles/r03_attn_variants.txt and docs/ROUND_NOTES.md.)
constexpr int ATTN_OCC41 = 4;
template <int NW, int DT, bool TAIL>
__global__ __launch_bounds__(64 * NW, DT == 1 ? (NW == 4 ? ATTN_OCC41 : 3) : ((DT == 2 && NW == 4) ? 2 : 1)) void attn_fwd_kernel(const slh_attn_desc p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * DT * 8192];
    char* sK = smem;                    // [2][DT][64 kv][128 B]
    char* sV = smem + 2 * DT * 8192;    // [2][DT][64 d ][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int frow = lane >> 3, fslot = lane & 7;
    // XCD-aware block order (block i runs on XCD i % 8, each with its own 4 MB L2): every XCD gets a contiguous run of the
    // (sample, head, query block) sequence with the query block fastest, so the query blocks that share one head's K / V
    // meet in ONE L2 instead of all eight (counter pass of the round-3 tree: 103 MB fetched per T = 4096 launch for 31 MB of
    // Q, K, V; L2 hit 52 %)
    int vb = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int qd = nblk >> 3, rm = nblk & 7;
        const int xcd = vb & 7, idx = vb >> 3;
        vb = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int nqb = (p.Tq + 32 * NW - 1) / (32 * NW);
    const int qb = vb % nqb, hb = vb / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qb * (32 * NW) + wave * 32;
    const int D = p.D > 0 ? p.D : 64;
    const int vt_heads = p.vt_batch_heads > 0 ? p.vt_batch_heads : p.H;

    const __bf16* Q = (const __bf16*)p.q;
    const __bf16* K = (const __bf16*)p.k;
    const __bf16* VT = (const __bf16*)p.vt;

    // Q fragments (B operand): lane holds Q[q][16*ks + 8*hi .. +8]; columns >= D are zero
    int qrow = q0 + lrow;
    const bool qvalid = qrow < p.Tq;
    qrow = qvalid ? qrow : p.Tq - 1;
    bf16x8 qf[DT * 4];
#pragma unroll
    for (int ks = 0; ks < DT * 4; ++ks) {
        const int col = ks * 16 + lhi * 8;
        if (col < D) qf[ks] = *(const bf16x8*)(Q + ((long)b * p.Tq + qrow) * p.ldq + h * D + col);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)0.f;
        }
    }

    const int nt = (p.Tk + 63) / 64;
    // Whole key tiles only (!TAIL): the per-lane source addresses of this wave's staging rows are computed once and advanced
    // by one key tile per iteration, and the copies are issued from inline asm (glds16_hidden) so that hipcc keeps counting
    // the fragment reads instead of draining lgkmcnt in front of every MFMA.  (The recomputed form cost a scalar load of the
    // zero page address + ~25 VALU + two exec-mask branches per tile, right behind the barrier.)
    const __bf16* kp[DT];
    const __bf16* vp[DT];
    long kstep[DT], khalf[DT];
    if constexpr (!TAIL) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int row = wave * 8 + frow;
            const int ks = fslot ^ ((row >> 1) & 7);       // the same for rows row + 8 * NW * i
            const int col = dt * 64 + ks * 8;
            const bool live = col < D;
            kp[dt] = live ? K + ((long)b * p.Tk + row) * p.ldk + h * D + col : (const __bf16*)slh_zero_page;
            kstep[dt] = live ? 64L * p.ldk : 0L;
            khalf[dt] = live ? (long)(8 * NW) * p.ldk : 0L;
            vp[dt] = VT + (((long)b * vt_heads + h) * (64 * DT) + dt * 64 + row) * p.ldvt + ks * 8;
        }
    }
    const unsigned sK_addr = lds_addr_of(smem), sV_addr = sK_addr + 2 * DT * 8192;
    auto stage_fast = [&](int buf) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int i = 0; i < 8 / NW; ++i) {
                const unsigned off = (buf * DT + dt) * 8192 + (wave + NW * i) * 1024;
                glds16_hidden(kp[dt] + i * khalf[dt], sK_addr + off);
                glds16_hidden(vp[dt] + (long)i * (8 * NW) * p.ldvt, sV_addr + off);
            }
            kp[dt] += kstep[dt];
            vp[dt] += 64;
        }
    };
    auto stage = [&](int buf, int t) {
        if constexpr (!TAIL) { stage_fast(buf); return; }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int i = 0; i < 8 / NW; ++i) {
                const int row = (wave + NW * i) * 8 + frow;
                const int ks = fslot ^ ((row >> 1) & 7);
                int kv = t * 64 + row;
                kv = kv < p.Tk ? kv : p.Tk - 1;
                const int col = dt * 64 + ks * 8;
                const __bf16* ksrc = col < D ? K + ((long)b * p.Tk + kv) * p.ldk + h * D + col
                                             : (const __bf16*)slh_zero_page;
                glds16(ksrc, sK + (buf * DT + dt) * 8192 + (wave + NW * i) * 1024);
                glds16(VT + (((long)b * vt_heads + h) * (64 * DT) + dt * 64 + row) * p.ldvt + t * 64 + ks * 8,
                       sV + (buf * DT + dt) * 8192 + (wave + NW * i) * 1024);
            }
        }
    };

    f32x16 o[2 * DT];
#pragma unroll
    for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * 1.4426950408889634f;
    // permuted key row for the A operand of S^T (swap bits 2 and 3)
    const int prow = (lrow & 3) | (((lrow >> 3) & 1) << 2) | (((lrow >> 2) & 1) << 3) | (lrow & 16);

    stage(0, 0);
    if constexpr (!TAIL) {
        // make hipcc wait for the Q fragments HERE: the copies above are invisible to its counter model, so a Q load it
        // still counts as pending at the loop entry becomes a vmcnt(0) in front of the first MFMA of EVERY iteration -
        // which would also drain the prefetch of the next tile
#pragma unroll
        for (int ks = 0; ks < DT * 4; ++ks) asm volatile("" ::"v"(qf[ks]));
    }
    for (int t = 0; t < nt; ++t) {
        lds_dma_syncthreads();       // tile t landed (all waves), everybody is past tile t-1
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char* cK = sK + (t & 1) * DT * 8192;
        const char* cV = sV + (t & 1) * DT * 8192;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(cK + dt * 8192 + lds_off(kt * 32 + prow, ks * 2 + lhi));
                    // first product of the tile takes a literal-zero C operand instead of 16 zeroed registers
                    if (dt == 0 && ks == 0) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0], kZero16, 0, 0, 0);
                    else s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[dt * 4 + ks], s[kt], 0, 0, 0);
                }
        }
        // s[kt][r] = S[q = lrow][kv = t*64 + kt*32 + 16*(r>>3) + 8*lhi + (r&7)]
        if (TAIL && t == nt - 1) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * 64 + kt * 32 + 16 * (r >> 3) + 8 * lhi + (r & 7);
                    if (kv >= p.Tk) s[kt][r] = -1e30f;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        // raw v_exp_f32 (arguments <= 0; results below 2^-126 flush to 0, which is what a probability that small is
        // worth) - the libm exp2f wraps every call in a denormal-range fixup that quadruples the VALU work here
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        bf16x8 pb[2][2];
        // scalar fp32 softmax arithmetic (this file is built with -fno-slp-vectorize): v_pk_fma/add/mul_f32 beside MFMAs cost more
        // than the two plain instructions they replace (same-box A/B, T = 4096: 149.9 -> 141.0 us)
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c, -mc));
                ps += pv;
                pb[kt][r >> 3][r & 7] = (__bf16)pv;
            }
        l_run = l_run * alpha + ps;
        // the running maximum settles after the first tiles: rescale the accumulators only when some row of the
        // wave actually moved (alpha == 1 exactly otherwise, so skipping is bit-identical)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
        }
#pragma unroll
        for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const bf16x8 vf = *(const bf16x8*)(cV + (dd >> 1) * 8192 + lds_off((dd & 1) * 32 + lrow, kstep * 2 + lhi));
                o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kstep >> 1][kstep & 1], o[dd], 0, 0, 0);
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qvalid) {
        __bf16* O = (__bf16*)p.o + ((long)b * p.Tq + qrow) * p.ldo + h * D;
#pragma unroll
        for (int dd = 0; dd < 2 * DT; ++dd)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int dcol = dd * 32 + qd * 8 + lhi * 4;
                if (dcol < D) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (__bf16)(o[dd][qd * 4 + e] * inv);
                    *(bf16x4*)(O + dcol) = v;
                }
            }
        if (p.lse && lhi == 0) p.lse[((long)b * p.H + h) * p.Tq + qrow] = m_run * c + log2f(l_tot);
    }
}

// ---- key-split form for grids that do not fill the SIMDs evenly (D = 64, whole key tiles, Tk % 128 == 0) -----------------------
// SDXL at 1024^2: the 32 x 32 level has 1280 wave tiles of 32 queries for 1024 SIMDs - a quarter of the SIMDs run two waves, and the
// time per key tile grows with the waves per SIMD (1400 + 1400 N cycles): the launch takes what TWO rounds take.  Here a workgroup is
// 4 waves = 2 query tiles x 2 key halves: wave (qi, kh) runs queries qi over the key tiles of half kh, so 2560 half-length waves
// spread as 2-3 per SIMD (1.5 rounds), and the two halves of a query tile meet in LDS at the end (fixed order: half 0 + half 1).
// LDS per half: K double-buffered, V^T single-buffered (48 KB per workgroup -> three workgroups per CU, all 640 resident): the K
// tile of step t+1 and the V^T tile of step t are requested at the top of step t (V^T first, so a counted vmcnt retires it while
// K stays in flight), and a second barrier in front of P.V publishes V^T.
__global__ __launch_bounds__(256, 3) void attn_fwd_ks_kernel(const slh_attn_desc p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 24576];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = wave & 1, kh = wave >> 1;
    const int lrow = lane & 31, lhi = lane >> 5;
    const int frow = lane >> 3, fslot = lane & 7;
    int vb = blockIdx.x;
    const int nqb = p.Tq >> 6;
    {
        // the launch's own workgroups come first; any behind them stream weights for a later product (slh_attn_desc.pf_*) and leave
        const int nblk = nqb * p.H * p.B;
        if (vb >= nblk) {
            weight_touch<8>(p.pf_ptr, p.pf_bytes, vb - nblk, (int)gridDim.x - nblk);
            return;
        }
        const int qd = nblk >> 3, rm = nblk & 7;
        const int xcd = vb & 7, idx = vb >> 3;
        vb = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int qb = vb % nqb, hb = vb / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int qrow = qb * 64 + qi * 32 + lrow;
    const int vt_heads = p.vt_batch_heads > 0 ? p.vt_batch_heads : p.H;
    const __bf16* Q = (const __bf16*)p.q;
    const __bf16* K = (const __bf16*)p.k;
    const __bf16* VT = (const __bf16*)p.vt;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Q + ((long)b * p.Tq + qrow) * p.ldq + h * 64 + ks * 16 + lhi * 8);

    const int nth = (p.Tk >> 6) >> 1;                    // key tiles per half
    // staging: the two waves of a half share its tiles; wave qi copies the 8-row groups qi, qi + 2, qi + 4, qi + 6
    const __bf16* kp;
    const __bf16* vp;
    long khalf;
    {
        const int row = qi * 8 + frow;
        const int ks = fslot ^ ((row >> 1) & 7);          // the same for rows row + 16 i
        kp = K + ((long)b * p.Tk + (long)kh * nth * 64 + row) * p.ldk + h * 64 + ks * 8;
        khalf = 16L * p.ldk;
        vp = VT + (((long)b * vt_heads + h) * 64 + row) * p.ldvt + (long)kh * nth * 64 + ks * 8;
    }
    const unsigned base = lds_addr_of(smem) + kh * 24576;
    long kstep = 64L * p.ldk;
    auto stage_k = [&](const int buf) {      // (past the last tile: four pieces from the zero page into the idle buffer - the waits stay uniform)
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_hidden(kp + i * khalf, base + buf * 8192 + (qi + 2 * i) * 1024);
        kp += kstep;
    };
    auto stage_v = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_hidden(vp + (long)i * 16 * p.ldvt, base + 16384 + (qi + 2 * i) * 1024);
        vp += 64;
    };

    f32x16 o[2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * 1.4426950408889634f;
    const int prow = (lrow & 3) | (((lrow >> 3) & 1) << 2) | (((lrow >> 2) & 1) << 3) | (lrow & 16);
    const char* sKh = smem + kh * 24576;
    const char* cV = sKh + 16384;

    stage_v();
    stage_k(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));     // the Q loads are waited for here, not inside the loop
    for (int t = 0; t < nth; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // K(t) landed for every wave; every wave is past P.V of step t-1
        if (t > 0) stage_v();                        // V^T(t) over V^T(t-1) ...
        if (t + 1 == nth) { kp = (const __bf16*)slh_zero_page; khalf = 0; kstep = 0; }
        stage_k((t + 1) & 1);                        // ... then K(t+1): the counted wait below retires V^T and leaves K in flight
        const char* cK = sKh + (t & 1) * 8192;
        f32x16 sc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(cK + lds_off(kt * 32 + prow, ks * 2 + lhi));
                if (ks == 0) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0], kZero16, 0, 0, 0);
                else sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc[kt], 0, 0, 0);
            }
        float mx = sc[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        bf16x8 pb[2][2];
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][r], c, -mc));
                ps += pv;
                pb[kt][r >> 3][r & 7] = (__bf16)pv;
            }
        l_run = l_run * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
        }
        // V^T(t): this wave's pieces have landed (the four K(t+1) pieces issued behind them may still fly), then everybody's
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const bf16x8 vf = *(const bf16x8*)(cV + lds_off(dd * 32 + lrow, kstep * 2 + lhi));
                o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kstep >> 1][kstep & 1], o[dd], 0, 0, 0);
            }
    }
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    // ---- the two key halves of a query tile meet: half 1 leaves (O, m, l) in LDS, half 0 combines in a fixed order and stores --
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zero-page pieces of the step past the end)
    __syncthreads();                                  // every wave is done with the tiles
    float* ex = (float*)smem + qi * (34 * 64);
    if (kh == 1) {
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[(dd * 16 + r) * 64 + lane] = o[dd][r];
        ex[32 * 64 + lane] = m_run;
        ex[33 * 64 + lane] = l_tot;
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = ex[32 * 64 + lane], l1 = ex[33 * 64 + lane];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f((m_run - m) * c), a1 = __builtin_amdgcn_exp2f((m1 - m) * c);
    l_tot = l_tot * a0 + l1 * a1;
    const float inv = 1.f / l_tot;
    __bf16* O = (__bf16*)p.o + ((long)b * p.Tq + qrow) * p.ldo + h * 64;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = qd * 4 + e;
                v[e] = (__bf16)((o[dd][r] * a0 + ex[(dd * 16 + r) * 64 + lane] * a1) * inv);
            }
            *(bf16x4*)(O + dd * 32 + qd * 8 + lhi * 4) = v;
        }
    if (p.lse && lhi == 0) p.lse[((long)b * p.H + h) * p.Tq + qrow] = m * c + log2f(l_tot);
}

// src [B][T][ld], head h = columns [h*D, (h+1)*D) -> dst [B][H][Dp][ldt], Dp = 64*ceil(D/64); rows d >= D and
// tokens >= T are written as zeros.  grid = (ldt/64, H * Dp/64, B)
__device__ __forceinline__ void transpose_heads_body(const slh_transpose_desc& p, int D, int DT, int bx, int by, int bz) {
    __shared__ __bf16 tile[64][72];
    const int tid = threadIdx.x;
    const int t0 = bx * 64, h = by / DT, dt = by - h * DT, b = bz;
    const __bf16* src = (const __bf16*)p.src;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int tok = idx >> 3, ch = idx & 7;
        const int col = dt * 64 + ch * 8;
        bf16x8 v;
        if (t0 + tok < p.T && col < D) v = *(const bf16x8*)(src + ((long)b * p.T + t0 + tok) * p.ld + h * D + col);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
        }
        *(bf16x8*)&tile[tok][ch * 8] = v;
    }
    __syncthreads();
    const int d = tid >> 2, tc = (tid & 3) * 16;
    __bf16* dst = (__bf16*)p.dst + (((long)b * p.H + h) * (64 * DT) + dt * 64 + d) * p.ldt + t0 + tc;
    bf16x8 o0, o1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o0[e] = tile[tc + e][d]; o1[e] = tile[tc + 8 + e][d]; }
    *(bf16x8*)dst = o0;
    *(bf16x8*)(dst + 8) = o1;
}

__global__ __launch_bounds__(256) void transpose_heads_kernel(const slh_transpose_desc p, int D, int DT) {
    transpose_heads_body(p, D, DT, blockIdx.x, blockIdx.y, blockIdx.z);
}

// n problems in one launch (slh_batch_desc): workgroup -> problem by bisection of the prefix sums
__global__ __launch_bounds__(256) void transpose_heads_batch_kernel(const slh_transpose_desc* table, const int* prefix, int n) {
    const int bid = blockIdx.x;
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= bid) lo = mid; else hi = mid;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const slh_transpose_desc d = table[lo];
    const int D = d.D > 0 ? d.D : 64;
    const int DT = (D + 63) / 64;
    const int local = bid - prefix[lo];
    const int gx = d.ldt / 64, gy = d.H * DT;
    const int bx = local % gx, rest = local / gx;
    transpose_heads_body(d, D, DT, bx, rest % gy, rest / gy);
}

// the key-split form can run the launch (D = 64, whole key tiles in two equal halves) / is the form launch_fwd picks for it
static bool attn_ks_ok(const slh_attn_desc* d, int DT) {
    return DT == 1 && (d->Tk & 63) == 0 && (d->D == 0 || d->D == 64) && d->Tq % 64 == 0 && d->Tk % 128 == 0 && d->Tk >= 256;
}
static bool attn_ks_form(const slh_attn_desc* d, int DT) {
    static const int knob_ks = getenv("SLH_ATTN_KS") ? atoi(getenv("SLH_ATTN_KS")) : 1;        // A/B: 0 = never
    const long blocks4 = (long)((d->Tq + 127) / 128) * d->H * d->B, blocks2 = (long)(d->Tq / 64) * d->H * d->B;
    return knob_ks && attn_ks_ok(d, DT) && blocks4 < 512 && blocks2 > 128 && blocks2 <= 768;
}

template <int DT>
int launch_fwd(const slh_attn_desc* d, hipStream_t s) {
    const long blocks4 = (long)((d->Tq + 127) / 128) * d->H * d->B;
    const bool tail = (d->Tk & 63) != 0;
    static const int knob_nw2 = getenv("SLH_ATTN_NW2") ? atoi(getenv("SLH_ATTN_NW2")) : 0;     // A/B: 64-query workgroups everywhere
    // 128-query workgroups once they fill every CU twice; below that the 64-query form (same waves per SIMD at most, finer
    // placement: T = 1024 with 40 heads 31.2 -> 28.9 us, same box).  (Capping the workgroups per CU with unused dynamic LDS so
    // that the dispatcher must spread them was tried: the hardware already places them evenly, the cap only delays backfill.)
    // key-split form (attn_fwd_ks_kernel above): D = 64, whole key tiles in two equal halves, and a 64-query grid that neither
    // fills every SIMD twice (>= 1024 workgroups) nor is so small that one round of full-length waves is cheaper
    static const int knob_ks = getenv("SLH_ATTN_KS") ? atoi(getenv("SLH_ATTN_KS")) : 1;        // A/B: 0 = never
    const long blocks2 = (long)(d->Tq / 64) * d->H * d->B;
    if (attn_ks_form(d, DT) || (knob_ks == 2 && attn_ks_ok(d, DT))) {      // 2 = whenever the shape allows (A/B)
        // weight touch (pf_*): up to 64 workgroups behind the launch's own; with three workgroups per CU they are resident beside them
        const int pf_blocks = (d->pf_ptr && d->pf_bytes >= 16) ? 64 : 0;
        hipLaunchKernelGGL(attn_fwd_ks_kernel, dim3((unsigned)blocks2 + pf_blocks), dim3(256), 0, s, *d);
        SLH_LAUNCH_CHECK("slh_attn_fwd (key split)");
        return 0;
    }
    if (blocks4 >= 512 && !knob_nw2) {
        const dim3 grid((unsigned)blocks4);
        if (tail) hipLaunchKernelGGL((attn_fwd_kernel<4, DT, true>), grid, dim3(256), 0, s, *d);
        else hipLaunchKernelGGL((attn_fwd_kernel<4, DT, false>), grid, dim3(256), 0, s, *d);
    } else {
        const dim3 grid((unsigned)((long)((d->Tq + 63) / 64) * d->H * d->B));
        if (tail) hipLaunchKernelGGL((attn_fwd_kernel<2, DT, true>), grid, dim3(128), 0, s, *d);
        else hipLaunchKernelGGL((attn_fwd_kernel<2, DT, false>), grid, dim3(128), 0, s, *d);
    }
    SLH_LAUNCH_CHECK("slh_attn_fwd");
    return 0;
}

}  // namespace

extern "C" int slh_attn_fwd_carries_touch(const slh_attn_desc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->Tq <= 0 || d->Tk <= 0) return 0;
    const int D = d->D > 0 ? d->D : 64;
    return attn_ks_form(d, (D + 63) / 64) ? 1 : 0;
}

extern "C" int slh_attn_fwd(const slh_attn_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->q && d->k && d->vt && d->o, "slh_attn_fwd: null pointer");
    SLH_CHECK(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "slh_attn_fwd: bad shape");
    const int D = d->D > 0 ? d->D : 64;
    SLH_CHECK(D % 8 == 0 && D <= 192, "slh_attn_fwd: head_dim %d unsupported (multiple of 8, <= 192)", D);
    SLH_CHECK(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 64 == 0 && d->ldo % 4 == 0, "slh_attn_fwd: alignment");
    SLH_CHECK(d->ldvt >= ((d->Tk + 63) / 64) * 64, "slh_attn_fwd: VT must be padded to a multiple of 64 keys");
    SLH_CHECK(!d->pf_ptr || (((uintptr_t)d->pf_ptr & 15) == 0 && d->pf_bytes >= 0), "slh_attn_fwd: pf_ptr must be 16-byte aligned");
    const int DT = (D + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    if (DT == 1) return launch_fwd<1>(d, s);
    if (DT == 2) return launch_fwd<2>(d, s);
    return launch_fwd<3>(d, s);
}

static int transpose_check(const slh_transpose_desc* d) {
    SLH_CHECK(d && d->src && d->dst, "slh_transpose_heads: null pointer");
    SLH_CHECK(d->ld % 8 == 0 && d->ldt % 64 == 0 && d->ldt >= d->T, "slh_transpose_heads: alignment");
    SLH_CHECK(d->B > 0 && d->H > 0 && d->T > 0, "slh_transpose_heads: empty problem");
    const int D = d->D > 0 ? d->D : 64;
    SLH_CHECK(D % 8 == 0 && D <= 192, "slh_transpose_heads: head_dim %d unsupported", D);
    return 0;
}

extern "C" int slh_transpose_heads_blocks(const slh_transpose_desc* d) {
    if (transpose_check(d)) return -1;
    const int D = d->D > 0 ? d->D : 64;
    return (d->ldt / 64) * d->H * ((D + 63) / 64) * d->B;
}

extern "C" int slh_transpose_heads_batch(const slh_batch_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->table && d->prefix && d->n > 0 && d->total > 0, "slh_transpose_heads_batch: empty batch / null pointer");
    hipLaunchKernelGGL(transpose_heads_batch_kernel, dim3(d->total), dim3(256), 0, (hipStream_t)stream,
                       (const slh_transpose_desc*)d->table, d->prefix, d->n);
    SLH_LAUNCH_CHECK("slh_transpose_heads_batch");
    return 0;
}

extern "C" int slh_transpose_heads(const slh_transpose_desc* d, slh_stream_t stream) {
    if (transpose_check(d)) return -1;
    const int D = d->D > 0 ? d->D : 64;
    const int DT = (D + 63) / 64;
    dim3 grid(d->ldt / 64, d->H * DT, d->B);
    hipLaunchKernelGGL(transpose_heads_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d, D, DT);
    SLH_LAUNCH_CHECK("slh_transpose_heads");
    return 0;
}
