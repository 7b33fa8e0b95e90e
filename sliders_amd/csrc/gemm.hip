// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution with fused LoRA-up, bias, time-embedding, residual
// and GEGLU epilogues.  gfx950 (CDNA4) only.
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T )
//
// Structure: 256 threads = 4 waves (2 x 2), block tile (64*MI) x (64*NI), K-step 64, double-buffered LDS
// filled with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip), one barrier per K-step.
// LDS rows are 128 B (64 bf16); the 16-byte slot index is XOR-swizzled with (row>>1)&7 so that the 16-lane
// groups of ds_read_b128 touch 16 distinct 16-B bank slots.  global_load_lds writes lane-linear, so the
// swizzle is applied to the per-lane SOURCE address (inverse permutation) and again on the read.
// MFMA operand roles are swapped (W rows feed the A operand, activation rows the B operand) so that each
// lane ends up with 4 consecutive output columns of ONE output row per accumulator quad: 8-byte stores,
// per-lane LoRA T[m][0..3], per-lane residual reads.
//
// Reference semantics replaced: torch.nn.functional.linear / conv2d inside diffusers-0.20.2
// (ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D.conv, Upsample2D.conv, Attention.to_q/k/v/out,
// GEGLU.proj, FeedForward.net.2, Transformer2DModel.proj_in/out) plus the LoRA branch of
// trainscripts/textsliders/lora.py:108-112.
#include "gemm_common.h"
#include <cstdlib>
#include <cstring>

using namespace slh_gemm_detail;

namespace {

// WM = waves along M (2: 4-wave workgroup, tile 64*MI x 64*NI; 4: 8-wave workgroup, tile 128*MI x 64*NI - twice
// the W-tile reuse per byte pulled from L2, which is what bounds these kernels)
// waves per SIMD the LDS footprint allows (workgroups per CU x waves per workgroup / 4 SIMDs); handed to the
// register allocator as the occupancy target so the epilogue's temporaries cannot cost a resident wave
constexpr int gemm_waves_per_simd(int MI, int NI, int STAGES, bool LORA, int WM) {
    const int lds = STAGES * (32 * MI * WM + 64 * NI + (LORA ? 32 : 0)) * 128;
    int blocks = (160 * 1024) / lds;
    int w = blocks * 2 * WM / 4;
    if (LORA && MI * NI == 1 && WM == 2 && w > 4) w = 4;   // 64x64 + fused LoRA needs ~110 registers: 4 waves, not 5
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}


template <int MI, int NI, int MODE, int STAGES, bool LORA, int WM, bool XA = false>
__global__ __launch_bounds__(128 * WM, gemm_waves_per_simd(MI, NI, STAGES, LORA, WM)) void gemm_kernel(const GemmArgs p) {
    constexpr int NW = 2 * WM;
    constexpr int BM = 32 * MI * WM;
    constexpr int BN = 64 * NI;
    constexpr int XI = BM / (8 * NW);  // glds instructions per wave for the X tile (8 rows each)
    constexpr int WI = BN / (8 * NW);
    static_assert(XI >= 1 && WI >= 1, "tile too small for this many waves");
    // LORA: the rank-r down matrix A (lora_down, [r][K], r <= 12, zero-padded to 32 rows) rides along as a third
    // operand tile; every wave multiplies it with the X fragments it already holds, so T = X.A^T of the block's own
    // rows is available in registers for the epilogue without a separate pass over X (lora.py:108-112 fused).
    constexpr int LROWS = LORA ? 32 : 0;
    static_assert(2 * WM * (2048 * NI + (LORA ? 2048 * MI : 0)) + 4 * BN * 4 <= STAGES * (BM + BN + LROWS) * 128,
                  "epilogue patches + adapter exchange + column vectors must fit the operand stages");
    __shared__ __attribute__((aligned(16))) char smem[STAGES * (BM + BN + LROWS) * 128];
    char* sX = smem;                        // [STAGES][BM][128 B]
    char* sW = smem + STAGES * BM * 128;    // [STAGES][BN][128 B]
    char* sL = smem + STAGES * (BM + BN) * 128;   // [STAGES][32][128 B]

    // the last pf_blocks workgroups of the grid (slh_gemm_desc.pf_*); only in the ring tile's instantiations: slh_gemm gives the
    // hint to no other tile, and the 16 loads in flight per thread do not fit the 80-register budgets of the small tiles
    if constexpr (STAGES == 4 && WM == 4 && MI == 1 && NI == 2) {
        if (gemm_weight_touch(p)) return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n, ks_id;
    int kt_begin = 0, nk = p.K / BK;
    {
        gemm_map_tile(p, tile_m, tile_n, ks_id);
        if (p.splitk > 1) {
            kt_begin = ks_id * p.kper;          // K tiles per slice: computed once, by slh_gemm, which also makes every slice non-empty
            nk = min(nk, kt_begin + p.kper) - kt_begin;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane fill geometry --------------------------------------------------------------
    const int frow = lane >> 3;   // row within the 8-row group written by one glds instruction
    const int fslot = lane & 7;   // physical 16-B slot in the row
    const int cin = p.ca0 + p.ca1;

    long xrow_off0[XI];           // dense: element offset of the row in source 0 / 1
    long xrow_off1[XI];
    int xb[XI], xoy[XI], xox[XI]; // conv: sample / output pixel of the row
    int xks[XI];                  // logical k-slot this lane fetches (inverse swizzle)
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (wave + NW * i) * 8 + frow;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        xks[i] = fslot ^ ((row >> 1) & 7);
        if (MODE == 0) {
            xrow_off0[i] = (long)m * p.lda0;
            xrow_off1[i] = (long)m * p.lda1;
        } else {
            const int hw = p.ho * p.wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / p.wo;
            xb[i] = b; xoy[i] = oy; xox[i] = rem - oy * p.wo;
        }
    }
    const __bf16* wptr[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (wave + NW * i) * 8 + frow;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        wptr[i] = p.w_packed ? p.w + ((long)(n >> 6) * (p.K >> 6)) * 4096 + ((n & 63) << 6) + (fslot << 3)
                             : p.w + (long)n * p.ldw + ((fslot ^ ((row >> 1) & 7)) << 3);
    }
    const int wkstep = p.w_packed ? 4096 : BK;   // elements between consecutive K tiles of one W row group

    // 2-slot loop: compiler-visible LDS-DMA.  (Issued from inline asm like the deep ring, hipcc counts
    // lgkmcnt in front of the MFMAs instead of waiting lgkmcnt(0), but the pass time did not change in a same-box A/B, and the
    // compiler's own vmcnt(0) in front of the epilogue barriers, which scripts/check_lds_dma_waits.py relies on, goes away.)
    auto stage_copy = [&](const void* src, void* lds_dst) {
        glds16(src, lds_dst);
    };
    auto stage = [&](int buf, int kt) {
        const int k0 = kt * BK;
        char* dX = sX + buf * (BM * 128);
        char* dW = sW + buf * (BN * 128);
        if (MODE == 0) {
            const bool s1 = k0 >= p.ca0;
            const __bf16* base = s1 ? p.a1 : p.a0;
            const int kk = s1 ? k0 - p.ca0 : k0;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const __bf16* src = base + (s1 ? xrow_off1[i] : xrow_off0[i]) + kk + (xks[i] << 3);
                stage_copy(src, dX + (wave + NW * i) * 1024);
            }
        } else {
            const int tap = k0 / cin;
            const int c0 = k0 - tap * cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const bool s1 = c0 >= p.ca0;
            const __bf16* base = s1 ? p.a1 : p.a0;
            const int ld = s1 ? p.lda1 : p.lda0;
            const int cc = s1 ? c0 - p.ca0 : c0;
            const int sh = p.src_xform ? 1 : 0;
            const int HL = p.hs << sh, WL = p.ws << sh;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int iy = xoy[i] * p.stride + ky - 1;
                const int ix = xox[i] * p.stride + kx - 1;
                bool ok = (iy >= 0) & (iy < HL) & (ix >= 0) & (ix < WL);
                if (p.src_xform == 2) ok = ok & (((iy | ix) & 1) == 0);
                const int sy = iy >> sh, sx = ix >> sh;
                const long pix = ((long)xb[i] * p.hs + sy) * p.ws + sx;
                const __bf16* src = ok ? base + pix * ld + cc + (xks[i] << 3)
                                       : (const __bf16*)slh_zero_page;
                stage_copy(src, dX + (wave + NW * i) * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            stage_copy(wptr[i] + (long)kt * wkstep, dW + (wave + NW * i) * 1024);
        }
        if (LORA) {
            const int row = (wave & 3) * 8 + frow;   // with 8 waves the upper four re-issue the same rows (benign)
            const __bf16* src = row < p.lora_rank
                                    ? p.lora_down + (long)row * p.K + k0 + ((fslot ^ ((row >> 1) & 7)) << 3)
                                    : (const __bf16*)slh_zero_page;
            stage_copy(src, sL + buf * (32 * 128) + (wave & 3) * 1024);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 accl[MI];   // LORA: accl[i][r'] = T[m = ...i*32 + lrow][rank index (r'&3) + 8*(r'>>2) + 4*lhi]
    if (LORA) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accl[i][r] = 0.f;
    }
    const int lrow = lane & 31, lhi = lane >> 5;

    float ln_mean[MI], ln_rstd[MI];
    f32x2 ln_pairs[MI][LN_MAXC];
    const bool ln_on = MODE == 0 && !LORA && p.ln_in != nullptr;   // (never with a fused adapter: slh_gemm rejects it)
    if (MODE == 0 && ln_on) gemm_ln_request<MI>(p, m0 + wm * (32 * MI), lrow, ln_pairs);
    auto ln_finish = [&]() {
        if (MODE == 0 && ln_on) gemm_ln_finish<MI>(p, m0 + wm * (32 * MI), lrow, tile_n == 0 && wn == 0 && lhi == 0, ln_pairs, ln_mean, ln_rstd);
    };

    auto compute = [&](int buf) {
        const char* cX = sX + buf * (BM * 128);
        const char* cW = sW + buf * (BN * 128);
        // fragments are double-buffered in registers: the ds_read_b128 of k-step ks+1 are issued before the MFMAs
        // of k-step ks, so LDS latency hides under the matrix pipe instead of being exposed 4x per K tile
        const char* cL = sL + buf * (32 * 128);
        bf16x8 xf[2][MI], wf[2][NI], lf[2];
        auto load_frags = [&](int set, int ks) {
            if (LORA && (ks & 1) == wn) lf[set] = *(const bf16x8*)(cL + lds_off(lrow, ks * 2 + lhi));
#pragma unroll
            for (int i = 0; i < MI; ++i)
                xf[set][i] = *(const bf16x8*)(cX + lds_off(wm * (32 * MI) + i * 32 + lrow, ks * 2 + lhi));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                wf[set][j] = *(const bf16x8*)(cW + lds_off(wn * (32 * NI) + j * 32 + lrow, ks * 2 + lhi));
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) load_frags((ks + 1) & 1, ks + 1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][j], xf[ks & 1][i], acc[i][j], 0, 0, 0);
            if (LORA && (ks & 1) == wn) {     // the two waves that share these rows split the adapter's K steps (see epilogue)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    accl[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf[ks & 1], xf[ks & 1][i], accl[i], 0, 0, 0);
            }
        }
    };

    if constexpr (STAGES == 2) {
        stage(0, kt_begin);
        ln_finish();
        for (int kt = 0; kt < nk; ++kt) {
            lds_dma_syncthreads();  // drains this wave's glds (explicit vmcnt(0)) and orders all waves
            if (kt + 1 < nk) stage((kt + 1) & 1, kt_begin + kt + 1);
            compute(kt & 1);
        }
    } else {
        // ---- deep LDS ring (STAGES = 3 or 4 slots), for launches that leave ONE workgroup per CU -----------------
        // A CU with a single resident workgroup hides nothing behind other workgroups: with the 2-slot loop above
        // every K tile pays a full L2/HBM round trip (measured: fill alone = 1340 clk per 32 KB tile, i.e. 24 B/clk
        // of the 64 B/clk the CU can pull) and the first ds_read latency of each tile.  Here
        //   * S-2 whole K tiles are kept in flight (LDS-DMA outstanding across barriers, counted vmcnt),
        //   * the MFMA fragments are software-pipelined across k-steps AND across K tiles (the fragments of step
        //     ks+1 - or of the next tile's step 0 - are requested before the MFMAs of step ks issue),
        //   * ONE raw s_barrier per K tile sits in the middle of the tile (after k-step 1).
        // Protocol for K tile g (slot g % S), all counts per wave:
        //   ks0, ks1 : MFMAs of (g,0), (g,1); fragment reads of (g,1), (g,2)
        //   sync     : s_waitcnt vmcnt((S-3)*L) -> this wave's share of tile g+1 has landed (tiles g+2.. stay in
        //              flight); s_barrier -> every wave's share has, and every wave is past its last read of tile g-1
        //   ks2, ks3 : MFMAs of (g,2), (g,3); fragment reads of (g,3), (g+1,0); LDS-DMA of tile g+S-1 into the slot
        //              tile g-1 occupied, half of the pieces behind each k-step's reads
        // The source pointers of the L pieces are running pointers (advanced once per K tile; re-based when the
        // implicit GEMM moves to the next filter tap or the second concat source), so a piece costs one 64-bit add.
        constexpr int S = STAGES;
        constexpr int L = XI + WI + (LORA ? 1 : 0);   // LDS-DMA instructions per wave per K tile
        constexpr int H0 = (L + 1) / 2;               // pieces issued behind k-step 2 (the rest behind k-step 3)
        const char* xsrc[XI];
        int xadv[XI];
        const char* wsrc[WI];
        const char* lsrc = nullptr;
        int ladv = 0;
#pragma unroll
        for (int i = 0; i < WI; ++i) wsrc[i] = (const char*)(wptr[i] + (long)kt_begin * wkstep);
        if (LORA) {
            const int row = (wave & 3) * 8 + frow;
            const bool ok = row < p.lora_rank;
            lsrc = ok ? (const char*)(p.lora_down + (long)row * p.K + kt_begin * BK + ((fslot ^ ((row >> 1) & 7)) << 3))
                      : (const char*)slh_zero_page;
            ladv = ok ? 128 : 0;
        }
        // K tile the next issue belongs to (relative to this workgroup's K slice); its channel offset / filter tap
        int i_kt = 0, i_c0 = kt_begin * BK, i_tap = 0;
        if (MODE == 1) { i_tap = i_c0 / cin; i_c0 -= i_tap * cin; }
        bool i_first = true;
        const unsigned lds0 = lds_addr_of(smem);
        const int wkbytes = wkstep * 2;

        // (re)base the X pointers of the tile at (i_tap, i_c0); called when a tap or a concat source begins
        auto rebase_x = [&]() {
            const bool s1 = i_c0 >= p.ca0;
            const __bf16* base = s1 ? p.a1 : p.a0;
            const int cc = s1 ? i_c0 - p.ca0 : i_c0;
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    xsrc[i] = (const char*)(base + (s1 ? xrow_off1[i] : xrow_off0[i]) + cc + (xks[i] << 3));
                    xadv[i] = 128;
                }
            } else {
                const int ld = s1 ? p.lda1 : p.lda0;
                const int ky = i_tap / 3, kx = i_tap - ky * 3;
                const int sh = p.src_xform ? 1 : 0;
                const int HL = p.hs << sh, WL = p.ws << sh;
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const int iy = xoy[i] * p.stride + ky - 1;
                    const int ix = xox[i] * p.stride + kx - 1;
                    bool ok = (iy >= 0) & (iy < HL) & (ix >= 0) & (ix < WL);
                    if (p.src_xform == 2) ok = ok & (((iy | ix) & 1) == 0);
                    const int sy = iy >> sh, sx = ix >> sh;
                    const long pix = ((long)xb[i] * p.hs + sy) * p.ws + sx;
                    xsrc[i] = ok ? (const char*)(base + pix * ld + cc + (xks[i] << 3)) : (const char*)slh_zero_page;
                    xadv[i] = ok ? 128 : 0;
                }
            }
        };
        // LDS-DMA piece j (order: X, W, LoRA) of K tile i_kt into ring slot `slot`
        auto piece = [&](const int j, const int slot) {
            if (j < XI) {
                glds16_hidden(xsrc[j], lds0 + slot * (BM * 128) + (wave + NW * j) * 1024);
                xsrc[j] += xadv[j];
            } else if (j < XI + WI) {
                const int i = j - XI;
                glds16_hidden(wsrc[i], lds0 + STAGES * BM * 128 + slot * (BN * 128) + (wave + NW * i) * 1024);
                wsrc[i] += wkbytes;
            } else {
                glds16_hidden(lsrc, lds0 + STAGES * (BM + BN) * 128 + slot * (32 * 128) + (wave & 3) * 1024);
                lsrc += ladv;
            }
        };
        auto tile_begin = [&]() {
            if (i_first || i_c0 == 0 || i_c0 == p.ca0) rebase_x();
            i_first = false;
        };
        auto tile_end = [&]() {
            ++i_kt;
            i_c0 += BK;
            if (MODE == 1 && i_c0 == cin) { i_c0 = 0; ++i_tap; }
        };
        auto issue_all = [&](const int slot) {
            tile_begin();
#pragma unroll
            for (int j = 0; j < L; ++j) piece(j, slot);
            tile_end();
        };

        bf16x8 xf[2][MI], wf[2][NI], lf[2];
        auto load_frags = [&](const int set, const int slot, const int ks) {
            const char* cX = sX + slot * (BM * 128);
            const char* cW = sW + slot * (BN * 128);
            if (LORA && set == wn) lf[set] = *(const bf16x8*)(sL + slot * (32 * 128) + lds_off(lrow, ks * 2 + lhi));   // set == ks & 1
#pragma unroll
            for (int i = 0; i < MI; ++i)
                xf[set][i] = *(const bf16x8*)(cX + lds_off(wm * (32 * MI) + i * 32 + lrow, ks * 2 + lhi));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                wf[set][j] = *(const bf16x8*)(cW + lds_off(wn * (32 * NI) + j * 32 + lrow, ks * 2 + lhi));
        };
        // the MFMAs of one k-step; with ISSUE, the LDS-DMA pieces [p0, p1) of the tile being staged are dealt out behind
        // them one at a time, so each piece's ~7 scalar/vector instructions issue in the shadow of a 32-cycle MFMA
        constexpr int NMF = MI * NI + (LORA ? MI : 0);
        auto mfmas = [&](const int set, const bool with_pieces, const int p0, const int p1, const int slot) {
            const int per = with_pieces ? (p1 - p0 + NMF - 1) / NMF : 0;
            int m = 0;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI + (LORA ? 1 : 0); ++j) {
                    if (j < NI) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][j], xf[set][i], acc[i][j], 0, 0, 0);
                    else if (set == wn) accl[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf[set], xf[set][i], accl[i], 0, 0, 0);
                    if (with_pieces) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < per; ++q)
                            if (p0 + m * per + q < p1) piece(p0 + m * per + q, slot);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ++m;
                }
            }
        };

        // prologue: tiles 0 .. S-2 in flight, tile 0 landed, its first fragments requested
        for (int t = 0; t < S - 1; ++t) {
            if (t < nk) issue_all(t);
        }
        ln_finish();
        if (nk >= S - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_frags(0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): nothing pending when the loop is entered
        int cur = 0;
        // one K tile; MORE: a next tile exists, ISSUE: tile g+S-1 exists, KEEP: tile g+2 exists (its LDS-DMA may stay in
        // flight across the barrier).  The flags are compile-time so that the steady-state body is straight-line code:
        // hipcc's waitcnt insertion falls back to lgkmcnt(0) at control-flow joins, which would expose the latency of
        // the fragment reads it was asked to keep in flight.
        auto body = [&](auto more_c, auto issue_c, auto keep_c) {
            constexpr bool MORE = decltype(more_c)::value, ISSUE = decltype(issue_c)::value, KEEP = decltype(keep_c)::value;
            const int nxt = cur == S - 1 ? 0 : cur + 1;
            const int prv = cur == 0 ? S - 1 : cur - 1;
            __builtin_amdgcn_sched_barrier(0);
            load_frags(1, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0, false, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(0, cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(1, false, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MORE) {
                if constexpr (S >= 4 && KEEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 3) * L) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            load_frags(1, cur, 3);
            if constexpr (ISSUE) tile_begin();
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0, ISSUE, 0, H0, prv);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MORE) load_frags(0, nxt, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(1, ISSUE, H0, L, prv);
            if constexpr (ISSUE) tile_end();
            __builtin_amdgcn_sched_barrier(0);
            // the fragments requested before the last four MFMAs have landed long before those MFMAs have issued; saying
            // so here leaves nothing pending across the loop edge (where the compiler would otherwise wait lgkmcnt(0)
            // AFTER the next k-step's reads have been issued)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            cur = nxt;
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        int g = 0;
        for (; g + S - 1 < nk; ++g) body(T_{}, T_{}, T_{});
        if constexpr (S >= 4) {
            if (g + 2 < nk) { body(T_{}, F_{}, T_{}); ++g; }
        }
        if (g + 1 < nk) { body(T_{}, F_{}, F_{}); ++g; }
        if (g < nk) body(F_{}, F_{}, F_{});
    }

    // XA: the instantiation that carries the fused cross-attention (FEAT bit 8, gemm_common.h)
    gemm_epilogue<MI, NI, MODE, LORA, NW, 2, XA ? 15 : 7>(p, smem, acc, accl, ln_mean, ln_rstd, ln_on, tile_m, tile_n, ks_id, wave, wm, wn);
}

template <int MI, int NI, int MODE, bool LORA, int WM>
int launch_gemm3(const GemmArgs& a, int stages, hipStream_t s) {
    const int grid = a.tiles_m * a.tiles_n * (a.splitk > 1 ? a.splitk : 1) + a.pf_blocks;
    constexpr int stage_bytes = (32 * MI * WM + 64 * NI + (LORA ? 32 : 0)) * 128;
    constexpr bool can3 = 3 * stage_bytes <= 160 * 1024;
    constexpr bool can4 = 4 * stage_bytes <= 160 * 1024;
#define SLH_LAUNCH_RING(ST_)                                                                                                  \
    slh_launch<gemm_kernel<MI, NI, MODE, ST_, LORA, WM>>(grid, 128 * WM, s, a, "gemm_kernel<%d, %d, %d, %d, %s, %d, false>", MI, NI, \
                                                         MODE, (int)(ST_), slh_tf(LORA), WM)
    if (stages == 4 && !can4) stages = 3;          // the deepest ring that fits the 160 KB of LDS
    if (stages == 3 && !can3) stages = 2;
    if (stages == 4)
        SLH_LAUNCH_RING((can4 ? 4 : 2));
    else if (stages == 3)
        SLH_LAUNCH_RING((can3 ? 3 : 2));
    else
        SLH_LAUNCH_RING(2);
#undef SLH_LAUNCH_RING
    SLH_LAUNCH_CHECK("slh_gemm");
    return 0;
}

// query projection + cross-attention (slh_gemm_desc.xa_k): the 128 x 128 8-wave ring tile (256 registers per lane to work with;
// the double-buffered loop's 128 would spill the scores)
int launch_gemm_xa(const GemmArgs& a, hipStream_t s) {
    const int grid = a.tiles_m * a.tiles_n + a.pf_blocks;
    slh_launch<gemm_kernel<1, 2, 0, 4, false, 4, true>>(grid, 512, s, a, "gemm_kernel<1, 2, 0, 4, false, 4, true>");
    SLH_LAUNCH_CHECK("slh_gemm (query projection + cross-attention)");
    return 0;
}

template <int MI, int NI, int WM>
int launch_gemm(const GemmArgs& a, int mode, int stages, hipStream_t s) {
    const bool lora = a.lora_down != nullptr;
    if (mode == 0) return lora ? launch_gemm3<MI, NI, 0, true, WM>(a, stages, s) : launch_gemm3<MI, NI, 0, false, WM>(a, stages, s);
    return lora ? launch_gemm3<MI, NI, 1, true, WM>(a, stages, s) : launch_gemm3<MI, NI, 1, false, WM>(a, stages, s);
}

}  // namespace

int slh_gemm5_launch(const slh_gemm_desc* d, slh_stream_t stream);      // gemm5.hip
int slh_gemm7_launch(const slh_gemm_desc* d, slh_stream_t stream);      // gemm7.hip

static int slh_ncu() {
    static const int ncu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        return prop.multiProcessorCount;
    }();
    return ncu;
}

// tile choice: explicit (d->tile, set by the planner from the tuned table) or a fill-the-chip heuristic
static void pick_tile(const slh_gemm_desc* d, int& MI, int& NI, int& WM) {
    MI = 2; NI = 2; WM = 2;
    if (d->tile) {
        MI = (d->tile >> 4) & 15; NI = d->tile & 15;
        const int wcode = (d->tile >> 12) & 15;
        WM = wcode == 4 ? 4 : 2;
        if (wcode == 8) WM = 8;      // ping-pong K loops (gemm8p.hip): MI = 4, NI = 2 -> 256 x 256; MI = 1, NI = 3..5 -> 128 x 64*NI
        if (MI) return;
        MI = 2; NI = 2; WM = 2;
    }
    // untuned shape: the pattern of the measured table (sliders_amd/tuning/gfx950_sdxl_128.json) - the 8-wave 128x128
    // tile once it yields enough workgroups (or fewer, but with a long K loop to amortise them), 64x64 otherwise
    const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
    if (d->geglu == 1) {
        NI = 2; MI = 1; WM = t128 >= 192 ? 4 : 2;
    } else if (t128 >= 256 || (t128 >= 160 && d->K >= 5760)) {
        MI = 1; NI = 2; WM = 4;
    } else {
        MI = 1; NI = 1; WM = 2;
    }
}

// L2 blocking: number of m-tile groups G (see the kernel's tile mapping).  An XCD's chunk of the grouped sequence reads
// its groups' X rows once if a group fits the L2 budget, and sweeps W once per group it touches, so the fabric
// traffic is about  X * max(1, 8/G) + G * W ; pick the power of two that minimises it.
static int pick_group_m(const slh_gemm_desc* d, int tiles_m) {
    static const char* env = getenv("SLIDERS_GEMM_GROUPS");   // measurement aid: force G (1 = ungrouped order)
    if (tiles_m <= 1) return 1;
    int G = 1;
    if (env && atoi(env) > 0) {
        G = atoi(env);
    } else {
        const int cin = d->ca0 + d->ca1;
        double xb = 2.0 * d->M * cin;
        if (d->mode == 1) xb = 2.0 * d->batch * d->hs * d->ws * cin;     // unique source pixels of the implicit GEMM
        const double wb = 2.0 * d->N * d->K;
        const double budget = 2.75e6;                                    // of the 4 MB L2: the rest holds W panels + C
        double best = -1.0;
        for (int g = 1; g <= tiles_m; g *= 2) {
            if (xb / g > budget && 2 * g <= tiles_m) continue;
            const double cost = xb * (g < 8 ? 8.0 / g : 1.0) + g * wb;
            if (best < 0.0 || cost < best) { best = cost; G = g; }
        }
    }
    if (G > tiles_m) G = tiles_m;
    return (tiles_m + G - 1) / G;
}

// the name of the kernel instantiation slh_gemm would launch for d, as rocprofv3 prints it ("gemm8pb_kernel<1, 5, 0, false>"): the
// whole of slh_gemm runs - descriptor checks, tile choice, ring-depth fallback - with the launch itself replaced by a record of the
// selected template (common.h: slh_launch).  No device needed, nothing launched.  0 / the descriptor's error.
extern "C" int slh_gemm_kernel_name(const slh_gemm_desc* d, char* buf, int cap) {
    SLH_CHECK(buf && cap >= 16, "slh_gemm_kernel_name: buffer");
    slh_name_sink_set(buf, cap);
    const int rc = slh_gemm(d, nullptr);
    slh_name_sink_set(nullptr, 0);
    if (rc == 0 && !buf[0]) { slh_set_error("slh_gemm_kernel_name: internal: no launch site recorded a name"); return -3; }
    return rc;
}

// (MI<<8)|(NI<<4)|mode of the kernel instantiation slh_gemm would launch: gemm_kernel<MI, NI, mode>
extern "C" int slh_gemm_variant(const slh_gemm_desc* d) {
    int MI, NI, WM;
    pick_tile(d, MI, NI, WM);
    return (WM << 12) | (MI << 8) | (NI << 4) | (d->mode & 15);
}

extern "C" int slh_gemm(const slh_gemm_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->a0 && d->w && d->c, "slh_gemm: null pointer");
    SLH_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "slh_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    SLH_CHECK((d->tile >> 20) == 0, "slh_gemm: tile 0x%x uses reserved bits (20 and up must be zero)", d->tile);
    SLH_CHECK(d->K % 64 == 0, "slh_gemm: K=%d must be a multiple of 64", d->K);
    SLH_CHECK(d->N % 4 == 0, "slh_gemm: N=%d must be a multiple of 4", d->N);
    SLH_CHECK(d->ca0 % 64 == 0 && d->ca1 % 64 == 0, "slh_gemm: channel counts must be multiples of 64");
    SLH_CHECK((d->a1 != nullptr) == (d->ca1 > 0), "slh_gemm: a1/ca1 mismatch");
    SLH_CHECK(d->lda0 % 8 == 0 && d->lda1 % 8 == 0 && d->ldw % 8 == 0 && d->ldc % 4 == 0,
              "slh_gemm: leading dimensions must keep 16-byte loads / 8-byte stores aligned");
    SLH_CHECK(wt_span_ok(d->M, d->ldc, d->geglu == 2 ? 2 * d->N : d->geglu ? d->N / 2 : d->N),
              "slh_gemm: the result tensor (M=%d rows of ldc=%d) reaches past 2 GiB from its base: the write-through row stores address it "
              "with 32-bit offsets - split the batch", d->M, d->ldc);
    const int cin = d->ca0 + d->ca1;
    if (d->mode == 0) {
        SLH_CHECK(cin == d->K, "slh_gemm: dense K=%d != ca0+ca1=%d", d->K, cin);
    } else {
        SLH_CHECK(d->mode == 1, "slh_gemm: bad mode %d", d->mode);
        SLH_CHECK(d->K == 9 * cin, "slh_gemm: conv K=%d != 9*Cin=%d", d->K, 9 * cin);
        SLH_CHECK(d->stride == 1 || d->stride == 2, "slh_gemm: bad stride");
        SLH_CHECK(d->src_xform >= 0 && d->src_xform <= 2, "slh_gemm: bad src_xform");
        SLH_CHECK(d->M == d->batch * d->ho * d->wo, "slh_gemm: conv M mismatch");
    }
    if (d->lora_down) {
        SLH_CHECK(!d->lora_t && d->lora_up && d->lora_scale && !d->geglu, "slh_gemm: fused lora_down excludes an external T / geglu");
        if (d->lora_up_rmajor) {
            // backward-data form: lora_down = the k-major copy of the up matrices ([rank][K], block-diagonal over a fused
            // q|k|v group), lora_up = the down matrices as stored ([rank][N]); U = dY . B leaves through lora_t_out
            SLH_CHECK(d->lora_groups == 1 && (d->lora_rank == 4 || d->lora_rank == 8 || d->lora_rank == 12),
                      "slh_gemm: fused r-major lora needs groups = 1, rank in {4, 8, 12}");
        } else {
            SLH_CHECK(d->lora_groups >= 1 && d->lora_groups <= 3 && d->lora_rank == 4 * d->lora_groups &&
                          d->N % d->lora_groups == 0 && (d->N / d->lora_groups) % 4 == 0,
                      "slh_gemm: fused lora needs rank = 4 * groups");
        }
        if (d->lora_t_out) SLH_CHECK(d->ld_t >= d->lora_rank && d->ld_t % 4 == 0, "slh_gemm: ld_t for lora_t_out");
    }
    if (d->lora_t) {
        SLH_CHECK(d->lora_up && d->lora_scale, "slh_gemm: lora pointers");
        SLH_CHECK(d->lora_groups >= 1 && d->lora_groups <= 3 && d->N % d->lora_groups == 0 &&
                      d->ld_t >= 4 * d->lora_groups && d->ld_t % 4 == 0,
                  "slh_gemm: bad lora grouping");
        if (d->lora_up_rmajor)
            SLH_CHECK(d->lora_groups == 1 && (d->lora_rank == 4 || d->lora_rank == 8 || d->lora_rank == 12) &&
                          d->ld_t >= d->lora_rank,
                      "slh_gemm: r-major lora_up needs groups=1 and rank in {4,8,12}");
        SLH_CHECK((d->N / d->lora_groups) % 4 == 0, "slh_gemm: lora group width");
    }
    if (d->rowbias) SLH_CHECK(d->rows_per_sample > 0 && d->ld_rowbias % 4 == 0, "slh_gemm: rowbias");
    if (d->residual) SLH_CHECK(d->ld_res % 4 == 0, "slh_gemm: ld_res");
    if (d->geglu == 1) SLH_CHECK(d->N % 64 == 0 && !d->lora_t && !d->residual && !d->rowbias, "slh_gemm: geglu constraints");
    if (d->geglu == 3)
        SLH_CHECK(d->N % 32 == 0 && !d->lora_t && !d->lora_down && !d->residual && !d->rowbias && !d->geglu_pre && !d->vt_out && !d->ln_out,
                  "slh_gemm: geglu = 3 (16 | 16 weight blocks) needs N %% 32 == 0 and excludes adapters, residual, row bias, geglu_pre, vt_out, ln_out");
    SLH_CHECK(d->geglu >= 0 && d->geglu <= 3, "slh_gemm: geglu is 0, 1 / 3 (forward epilogue, 32 | 32 or 16 | 16 weight blocks) or 2 (backward form)");

    if (((d->tile >> 12) & 15) == 7) return slh_gemm7_launch(d, stream);      // the four-wave tiles (gemm7.hip): their own descriptor checks
    if (((d->tile >> 12) & 15) == 5) return slh_gemm5_launch(d, stream);      // the 64 x 160 tile (gemm5.hip): its own descriptor checks
    int MI = 2, NI = 2, WM = 2;
    pick_tile(d, MI, NI, WM);
    if (WM == 8) {
        SLH_CHECK((MI == 4 && NI == 2) || (MI == 1 && NI >= 3 && NI <= 5),
                  "slh_gemm: ping-pong tiles are 256 x 256 (0x8042) or 128 x 64*NI (0x801<NI>, NI = 3..5)");
        SLH_CHECK(!d->lora_down || MI == 1, "slh_gemm: the 256 x 256 tile does not take a fused adapter (lora_down)");
        SLH_CHECK(!d->vt_out || (MI == 4 && NI == 2) || (MI == 1 && NI == 4 && d->mode == 0),
                  "slh_gemm: vt_out on the ping-pong tiles needs 256 x 256 (0x8042) or the dense 128 x 256 tile (0x8014)");
    } else {
        SLH_CHECK((MI == 1 || MI == 2) && (NI == 1 || NI == 2), "slh_gemm: bad tile");
        SLH_CHECK(WM == 2 || NI == 2 || MI == 1, "slh_gemm: 8-wave tiles are 128x64, 128x128 or 256x128");
    }
    SLH_CHECK(d->w_layout == 0 || d->w_layout == 1, "slh_gemm: bad w_layout");
    if (d->geglu == 1) SLH_CHECK(NI == 2, "slh_gemm: geglu needs NI=2");

    GemmArgs a;
    a.a0 = (const __bf16*)d->a0; a.a1 = (const __bf16*)d->a1; a.w = (const __bf16*)d->w;
    a.bias = (const __bf16*)d->bias; a.rowbias = (const __bf16*)d->rowbias; a.lora_t = d->lora_t;
    a.lora_up = (const __bf16*)d->lora_up; a.lora_scale = d->lora_scale;
    a.residual = (const __bf16*)d->residual; a.c = (__bf16*)d->c;
    a.lora_down = (const __bf16*)d->lora_down; a.lora_t_out = d->lora_t_out;
    a.lda0 = d->lda0; a.lda1 = d->lda1; a.ca0 = d->ca0; a.ca1 = d->ca1;
    a.hs = d->hs; a.ws = d->ws; a.src_xform = d->src_xform; a.stride = d->stride; a.ho = d->ho; a.wo = d->wo;
    a.ldw = d->ldw; a.M = d->M; a.N = d->N; a.K = d->K;
    a.ld_rowbias = d->ld_rowbias; a.rows_per_sample = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
    a.ld_t = d->ld_t; a.lora_cols_per_group = (d->lora_t || d->lora_down) ? d->N / d->lora_groups : 1;
    a.ld_res = d->ld_res; a.ldc = d->ldc; a.geglu = d->geglu;
    a.lora_rank = d->lora_rank > 0 ? d->lora_rank : 4; a.lora_up_rmajor = d->lora_up_rmajor;
    a.w_packed = d->w_layout;
    a.vt = (__bf16*)d->vt_out; a.vt_col0 = d->vt_col0; a.vt_D = d->vt_D; a.vt_heads = d->vt_heads;
    a.vt_tokens = d->vt_tokens; a.vt_ld = d->vt_ld;
    if (d->vt_out) {
        SLH_CHECK(d->vt_D > 0 && d->vt_D % 64 == 0 && d->vt_col0 % 128 == 0 && d->vt_col0 < d->N && d->vt_heads > 0 &&
                      (d->N - d->vt_col0) == d->vt_heads * d->vt_D && d->vt_tokens % 8 == 0 && d->M % d->vt_tokens == 0 &&
                      d->vt_ld % 8 == 0 && d->vt_ld >= d->vt_tokens && !d->geglu && ((uintptr_t)d->vt_out & 15) == 0,
                  "slh_gemm: vt_out constraints");
    }
    a.ln_out = d->ln_out; a.ln_in = d->ln_in; a.ln_s = d->ln_s; a.ln_b = d->ln_b;
    a.ln_in_chunks = d->ln_in_chunks; a.ln_eps = d->ln_eps;
    if (d->ln_out) {
        SLH_CHECK(NI == 2 && d->N % 64 == 0 && !d->geglu && !d->vt_out,
                  "slh_gemm: ln_out needs a 128-column tile (NI = 2), N %% 64 == 0, no GEGLU / vt_out");
        SLH_CHECK(((uintptr_t)d->ln_out & 7) == 0, "slh_gemm: ln_out alignment");
    }
    if (d->ln_in) {
        SLH_CHECK(d->mode == 0 && !d->a1 && d->ln_s && d->ln_b && !d->bias && !d->lora_t,
                  "slh_gemm: ln_in needs a dense single-source product, ln_s / ln_b, no bias (folded into ln_b), no external T");
        if (d->lora_down)
            SLH_CHECK(WM == 8 && MI == 1 && NI <= 4 && d->ln_lora_s && d->ln_lora_c && !d->lora_up_rmajor && !d->lora_t_out && !d->ln_mr_out &&
                          ((d->tile >> 16) & 15) <= 1,
                      "slh_gemm: ln_in with a fused adapter runs on the ping-pong 128 x 192 / 128 x 256 tiles (0x8013, 0x8014) and needs "
                      "ln_lora_s / ln_lora_c (lora_down = A . gamma); forward form, no split-K");
        SLH_CHECK(d->ln_in_chunks >= 1 && d->ln_in_chunks <= 20 && d->K % d->ln_in_chunks == 0 &&
                      (d->K == 64 * d->ln_in_chunks || d->K == 80 * d->ln_in_chunks),
                  "slh_gemm: ln_in_chunks must be K / 64 (producer on a 64 / 128-column tile) or K / 80 (producer on the 64 x 160 tile), <= 20");
        SLH_CHECK(((uintptr_t)d->ln_in & 7) == 0 && ((uintptr_t)d->ln_s & 15) == 0 && ((uintptr_t)d->ln_b & 15) == 0,
                  "slh_gemm: ln_in / ln_s / ln_b alignment");
    }
    a.splitk = (d->tile >> 16) & 15;
    a.c32 = d->splitk_c32;
    a.t32 = d->splitk_t32;
    a.ticket = (unsigned long long*)d->splitk_ticket;
    a.ln_mr_out = d->ln_mr_out;
    a.geglu_pre = (__bf16*)d->geglu_pre; a.ld_pre = d->ld_pre;
    a.vt_also_c = d->vt_also_c;
    SLH_CHECK(!d->vt_also_c || d->vt_out, "slh_gemm: vt_also_c without vt_out");
    SLH_CHECK(!d->ln_mr_out || d->ln_in, "slh_gemm: ln_mr_out without ln_in");
    SLH_CHECK(!d->geglu_pre || (d->geglu && d->ld_pre >= (d->geglu == 2 ? 2 * d->N : d->N) && d->ld_pre % 4 == 0 &&
                                ((uintptr_t)d->geglu_pre & 7) == 0),
              "slh_gemm: geglu_pre needs the GEGLU epilogue, ld_pre >= N (2N in the backward form), 8-byte alignment");
    if (d->geglu == 2)
        SLH_CHECK(d->geglu_pre && d->N % 32 == 0 && d->ldc >= 2 * d->N && d->ldc % 4 == 0 && !d->bias && !d->rowbias && !d->residual &&
                      !d->lora_t && !d->lora_down && !d->vt_out && !d->ln_in && !d->ln_out,
                  "slh_gemm: geglu = 2 (backward form) needs geglu_pre, N %% 32 == 0, ldc >= 2N and a bare product");
    // same-XCD slab reads (gemm_common.h, split-K epilogue): gfx950 only, SLIDERS_SPLITK_LOCAL=0 turns them off
    static const int splitk_local_ok = [] {
        const char* e = getenv("SLIDERS_SPLITK_LOCAL");
        if (e && atoi(e) == 0) return 0;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
    }();
    a.splitk_local = splitk_local_ok;
    a.kper = d->K / 64;
    if (a.splitk > 1) {
        // every slice must be non-empty: each publishes its whole partial tile, the last one to arrive reads them all
        const int nk = d->K / 64;
        const int per = (nk + a.splitk - 1) / a.splitk;
        a.splitk = (nk + per - 1) / per;
        a.kper = per;
        SLH_CHECK((a.splitk - 1) * per < nk, "slh_gemm: internal: empty K slice (%d slices of %d over %d K tiles)", a.splitk, per, nk);
    }
    if (a.splitk > 1) {
        SLH_CHECK(d->splitk_c32, "slh_gemm: split-K needs the fp32 slab workspace splitk_c32");
        SLH_CHECK(d->splitk_slabs >= a.splitk, "slh_gemm: split-K into %d slices but the workspace holds %d slabs", a.splitk,
                  d->splitk_slabs);
        SLH_CHECK(d->splitk_ticket, "slh_gemm: split-K needs the arrival tickets splitk_ticket (zeroed once)");
        SLH_CHECK((long)a.splitk * ((d->M + 255) / 256 * 256L) * ((d->N + 255) / 256 * 256L) * 4 < (1L << 31),
                  "slh_gemm: split-K slabs beyond 2 GB");
        SLH_CHECK(((uintptr_t)d->splitk_ticket & 7) == 0 && d->N % 4 == 0, "slh_gemm: split-K needs N %% 4 == 0 and 8-byte aligned tickets");
        SLH_CHECK(!d->lora_down || a.t32, "slh_gemm: split-K with a fused adapter needs the slab workspace splitk_t32");
        SLH_CHECK(((uintptr_t)d->splitk_c32 & 15) == 0 && ((uintptr_t)d->splitk_t32 & 15) == 0, "slh_gemm: slab alignment");
    } else {
        a.splitk = 1;
    }
    a.store16 = (d->ldc % 8 == 0) && (((uintptr_t)d->c & 15) == 0);
    a.ln_lora_s = d->ln_lora_s; a.ln_lora_c = d->ln_lora_c;
    a.xa_k = (const __bf16*)d->xa_k; a.xa_vt = (const __bf16*)d->xa_vt;
    a.xa_tk = d->xa_tk; a.xa_tq = d->xa_tq; a.xa_ldk = d->xa_ldk; a.xa_ldvt = d->xa_ldvt; a.xa_vt_heads = d->xa_vt_heads;
    a.xa_scale = d->xa_scale;
    if (d->xa_k) {
        const int st = (d->tile >> 8) & 15;
        SLH_CHECK(WM == 4 && MI == 1 && NI == 2 && st == 4 && a.splitk == 1,
                  "slh_gemm: the fused cross-attention runs on the 128 x 128 8-wave ring tile (0x4412), no split-K");
        SLH_CHECK(d->mode == 0 && !d->lora_down && !d->lora_t && !d->residual && !d->rowbias && !d->geglu && !d->ln_out && !d->vt_out,
                  "slh_gemm: xa_k excludes adapters, residual, row bias, GEGLU, ln_out, vt_out");
        SLH_CHECK(d->xa_vt && d->N % 64 == 0 && d->xa_tk >= 1 && d->xa_tk <= 96 && d->xa_tq > 0 && d->xa_tq % 128 == 0 &&
                      d->M % d->xa_tq == 0 && d->xa_ldvt >= 128 && d->xa_ldvt % 8 == 0 && d->xa_ldk % 8 == 0 &&
                      d->xa_vt_heads >= d->N / 64 && ((uintptr_t)d->xa_k & 15) == 0 && ((uintptr_t)d->xa_vt & 15) == 0,
                  "slh_gemm: xa_* need head dim 64 (N %% 64 == 0), 1 <= xa_tk <= 96, xa_tq %% 128 == 0, M %% xa_tq == 0, "
                  "xa_ldvt >= 128 (two 64-key tiles are staged), 16-byte aligned keys / values");
    }
    a.pf_ptr = nullptr; a.pf_bytes = 0; a.pf_blocks = 0;
    if (WM == 8) {
        const int bm = MI == 1 ? 128 : 256, bn = 64 * NI * (MI == 4 ? 2 : 1);
        a.tiles_m = (d->M + bm - 1) / bm;
        a.tiles_n = (d->N + bn - 1) / bn;
        // the slab workspace is sized by contract (include/sliders_hip.h: roundup(M, 256) x roundup(N, 128) floats per slice)
        if (a.splitk > 1)
            SLH_CHECK((long)a.tiles_m * bm * a.tiles_n * bn <= ((d->M + 255) / 256 * 256L) * ((d->N + 127) / 128 * 128L),
                      "slh_gemm: split-K slabs of %d x %d tiles exceed the workspace contract for M=%d N=%d", bm, bn, d->M, d->N);
        a.group_m = pick_group_m(d, a.tiles_m);
        return launch_gemm8p(a, d->mode, MI == 4 ? 0 : NI, (hipStream_t)stream);
    }
    a.tiles_m = (d->M + 32 * MI * WM - 1) / (32 * MI * WM);
    a.tiles_n = (d->N + 64 * NI - 1) / (64 * NI);
    a.group_m = pick_group_m(d, a.tiles_m);
    if (d->pf_ptr && d->pf_bytes >= 16 && WM == 4 && MI == 1 && NI == 2 && ((d->tile >> 8) & 15) == 4) {
        // the weight touch is a hint: taken only on the 128 x 128 ring tile (0x4412: one workgroup per CU) and only when the launch
        // leaves CUs idle
        SLH_CHECK(((uintptr_t)d->pf_ptr & 15) == 0, "slh_gemm: pf_ptr must be 16-byte aligned");
        const int idle = slh_ncu() - a.tiles_m * a.tiles_n * (a.splitk > 1 ? a.splitk : 1);
        if (idle >= 16) { a.pf_ptr = d->pf_ptr; a.pf_bytes = (long)d->pf_bytes; a.pf_blocks = idle < 64 ? idle : 64; }
    }
    hipStream_t s = (hipStream_t)stream;
    const int stages = (d->tile >> 8) & 15;   // tile = (WM<<12)|(stages<<8)|(MI<<4)|NI ; stages 0/2 = double buffer
    if (d->xa_k) return launch_gemm_xa(a, s);
    if (WM == 4) {
        if (MI == 2) return launch_gemm<2, 2, 4>(a, d->mode, stages, s);   // 256 x 128, 8 waves
        if (NI == 1) return launch_gemm<1, 1, 4>(a, d->mode, stages, s);   // 128 x 64, 8 waves
        return launch_gemm<1, 2, 4>(a, d->mode, stages, s);                // 128 x 128, 8 waves
    }
    if (MI == 2 && NI == 2) return launch_gemm<2, 2, 2>(a, d->mode, stages, s);
    if (MI == 2 && NI == 1) return launch_gemm<2, 1, 2>(a, d->mode, stages, s);
    if (MI == 1 && NI == 2) return launch_gemm<1, 2, 2>(a, d->mode, stages, s);
    return launch_gemm<1, 1, 2>(a, d->mode, stages, s);
}
