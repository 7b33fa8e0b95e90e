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
#include "common.h"
#include <cstdlib>
#include <cstdint>
#include <type_traits>
#include "../../include/sliders_hip.h"

namespace {

struct GemmArgs {
    const __bf16* a0; const __bf16* a1; const __bf16* w;
    const __bf16* bias; const __bf16* rowbias; const float* lora_t; const __bf16* lora_up;
    const float* lora_scale; const __bf16* residual; __bf16* c;
    const __bf16* lora_down; float* lora_t_out;
    int lda0, lda1, ca0, ca1;
    int hs, ws, src_xform, stride, ho, wo;
    int ldw, M, N, K;
    int ld_rowbias, rows_per_sample, ld_t, lora_cols_per_group, ld_res, ldc, geglu;
    int lora_rank, lora_up_rmajor, w_packed;
    int tiles_m, tiles_n, group_m;
    float* c32; float* t32;   // split-K: fp32 partial sums (zeroed by the caller), see slh_gemm_desc.splitk_c32
    int splitk;
    unsigned long long* ticket;   // split-K arrival tickets, one per output tile (slh_gemm_desc.splitk_ticket)
    __bf16* vt; int vt_col0, vt_D, vt_heads, vt_tokens, vt_ld;   // head-transposed store of the V columns (slh_gemm_desc.vt_out)
    int store16;  // c and ldc allow 16-byte row stores
    // LayerNorm folded into the product (slh_gemm_desc.ln_*): producer side writes per-row chunk statistics of its
    // bf16-rounded output, consumer side normalises the A operand algebraically (weights pre-scaled by gamma)
    float* ln_out; const float* ln_in; const float* ln_s; const float* ln_b;
    float* ln_mr_out;     // consumer side: the merged (mean, rstd) of every row, for the LayerNorm backward (slh_gemm_desc.ln_mr_out)
    int vt_also_c;        // the head-transposed columns are written to c as well (slh_gemm_desc.vt_also_c)
    __bf16* geglu_pre; int ld_pre;   // GEGLU: the bf16 pre-activation [M][N] kept for the backward (slh_gemm_desc.geglu_pre)
    int ln_in_chunks; float ln_eps;
    int probe;   // diagnostics (slh_gemm_desc.reserved_): 1 skip tile refills, 2 skip MFMA work, 4 skip the epilogue,
                 // 8 skip the first tile fill, 16 return at once
};

constexpr int BK = 64;

// swizzled byte offset of (row, 16-byte slot) inside a [rows][64] bf16 LDS tile
__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

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

template <int MI, int NI, int MODE, int STAGES, bool LORA, int WM>
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

    if (p.probe & 16) return;   // diagnostics: launch + dispatch cost only
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping: block b runs on XCD b%8 (own 4 MB L2); give each XCD a contiguous chunk of a
    // grouped tile sequence: groups of group_m consecutive m-tiles, inside a group m fastest, then n.  The host
    // sizes the groups so that one group's X rows stay resident in the XCD's L2 while the W panels stream past
    // once (measured before grouping: L2 hit rate 62-78 %, fabric fetches 6-14x the unique operand bytes).
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int ks_id = 0;                 // split-K: consecutive block ids = the K slices of one tile (same XCD, same L2)
    if (p.splitk > 1) { ks_id = bid % p.splitk; bid = bid / p.splitk; }
    int tile_m, tile_n;
    {
        const int gsz = p.group_m * p.tiles_n;
        const int g = bid / gsz;
        const int first_m = g * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - first_m);
        const int r = bid - g * gsz;
        tile_n = r / gm;
        tile_m = first_m + r - tile_n * gm;
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

    // 2-slot loop: compiler-visible LDS-DMA.  (Issued from inline asm like the deep ring - SLH_GEMM_HIDDEN_STAGE - hipcc counts
    // lgkmcnt in front of the MFMAs instead of waiting lgkmcnt(0), but the pass time did not change in a same-box A/B, and the
    // compiler's own vmcnt(0) in front of the epilogue barriers, which scripts/check_lds_dma_waits.py relies on, goes away.)
    auto stage_copy = [&](const void* src, void* lds_dst) {
#ifdef SLH_GEMM_HIDDEN_STAGE
        glds16_hidden(src, lds_addr_of(lds_dst));
#else
        glds16(src, lds_dst);
#endif
    };
#ifdef SLH_GEMM_PROBE_W
    bool probe_first_tile = true;
#endif
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
#ifdef SLH_GEMM_PROBE_W        // ablation build (scripts/build_variant.sh): probe bit 32 = no W refills after the first tile
            if ((p.probe & 32) && !probe_first_tile) continue;
#endif
            stage_copy(wptr[i] + (long)kt * wkstep, dW + (wave + NW * i) * 1024);
        }
        if (LORA) {
            const int row = (wave & 3) * 8 + frow;   // with 8 waves the upper four re-issue the same rows (benign)
            const __bf16* src = row < p.lora_rank
                                    ? p.lora_down + (long)row * p.K + k0 + ((fslot ^ ((row >> 1) & 7)) << 3)
                                    : (const __bf16*)slh_zero_page;
            stage_copy(src, sL + buf * (32 * 128) + (wave & 3) * 1024);
        }
#ifdef SLH_GEMM_PROBE_W
        probe_first_tile = false;
#endif
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
    int kt_begin = 0, nk = p.K / BK;
    if (p.splitk > 1) {
        const int per = (nk + p.splitk - 1) / p.splitk;
        kt_begin = ks_id * per;
        nk = min(nk, kt_begin + per) - kt_begin;
        if (nk <= 0) return;       // uniform over the workgroup
    }
    const int lrow = lane & 31, lhi = lane >> 5;

    // LayerNorm of the A operand, folded (consumer side).  The producer of A left per-row (mean, M2) pairs of 64-column
    // chunks (ln_in, laid out [chunk][row], written by its epilogue below); they are requested here, ahead of the first operand tiles, merged
    // in a fixed order (Chan's merge for equal counts: cancellation-free) into the row's mean and 1/sigma, and applied in the epilogue:
    //   LN(x) . W^T = rstd * (x . W'^T - mean * s) + b',   W' = W * gamma, s = row sums of W', b' = bias + W . beta
    constexpr int LN_MAXC = 20;
    float ln_mean[MI], ln_rstd[MI];
    f32x2 ln_pairs[MI][MODE == 0 ? LN_MAXC : 1];
    const bool ln_on = MODE == 0 && !LORA && p.ln_in != nullptr;   // (never with a fused adapter: slh_gemm rejects it)
    if (MODE == 0 && ln_on) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int m = m0 + wm * (32 * MI) + i * 32 + lrow;
            m = m < p.M ? m : p.M - 1;
            const f32x2* src = (const f32x2*)p.ln_in + m;        // chunk-major [chunks][M]: the 32 rows of a wave-load are contiguous
#pragma unroll
            for (int c = 0; c < LN_MAXC; ++c)
                if (c < p.ln_in_chunks) ln_pairs[i][c] = src[(long)c * p.M];
        }
    }
    auto ln_finish = [&]() {
        if (MODE == 0 && ln_on) {
            // equal-sized chunks, one pass over the pairs with the chunk means shifted by the first one (d_c = mean_c - mean_0):
            //   mean = mean_0 + S/k,  M2 = sum M2_c + n_chunk * (sum d_c^2 - S^2/k),  S = sum d_c
            // = Chan's merge for equal counts; two interleaved accumulator sets keep the dependent chains short; fixed order
            const float nc = (float)(p.K / p.ln_in_chunks);
            const float inv_chunks = 1.f / (float)p.ln_in_chunks;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float m0v = ln_pairs[i][0][0];
                float sa = 0.f, sb = 0.f, pa = 0.f, pb = 0.f, qa = ln_pairs[i][0][1], qb = 0.f;
#pragma unroll
                for (int c = 1; c < LN_MAXC; c += 2) {
                    if (c < p.ln_in_chunks) { const float dl = ln_pairs[i][c][0] - m0v; sa += dl; pa += dl * dl; qa += ln_pairs[i][c][1]; }
                    if (c + 1 < p.ln_in_chunks && c + 1 < LN_MAXC) {
                        const float dl = ln_pairs[i][c + 1 < LN_MAXC ? c + 1 : c][0] - m0v;
                        sb += dl; pb += dl * dl; qb += ln_pairs[i][c + 1 < LN_MAXC ? c + 1 : c][1];
                    }
                }
                const float S = sa + sb;
                const float dm = S * inv_chunks;
                ln_mean[i] = m0v + dm;
                const float M2 = (qa + qb) + nc * fmaxf((pa + pb) - S * dm, 0.f);
                ln_rstd[i] = 1.0f / sqrtf(M2 / (float)p.K + p.ln_eps);
                if (p.ln_mr_out && tile_n == 0 && wn == 0 && lhi == 0) {
                    const int m = m0 + wm * (32 * MI) + i * 32 + lrow;
                    if (m < p.M) *(f32x2*)(p.ln_mr_out + (long)m * 2) = f32x2{ln_mean[i], ln_rstd[i]};
                }
            }
        }
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
        if (!(p.probe & 8)) stage(0, kt_begin);
        ln_finish();
        for (int kt = 0; kt < nk; ++kt) {
            lds_dma_syncthreads();  // drains this wave's glds (explicit vmcnt(0)) and orders all waves
            if (kt + 1 < nk && !(p.probe & 1)) stage((kt + 1) & 1, kt_begin + kt + 1);
            if (!(p.probe & 2)) compute(kt & 1);
        }
        if (p.probe & 4) return;
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
#if defined(SLH_RING_PROBE) && (SLH_RING_PROBE & 1)
            if (i_kt >= S - 1) return;                 // ablation: no LDS-DMA after the prologue
#endif
#ifdef SLH_GEMM_PROBE_W        // ablation build: probe bit 32 = no W pieces, bit 64 = no X pieces after the prologue
            if (i_kt >= S - 1 && (((p.probe & 32) && j >= XI && j < XI + WI) || ((p.probe & 64) && j < XI))) return;
#endif
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
#if defined(SLH_RING_PROBE) && (SLH_RING_PROBE & 4)
            return;                                    // ablation: no fragment reads
#endif
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
#if defined(SLH_RING_PROBE) && (SLH_RING_PROBE & 2)
                    if (j < NI) asm volatile("" ::"v"(wf[set][j]), "v"(xf[set][i]));      // ablation: no MFMA (fragments stay live)
                    else asm volatile("" ::"v"(lf[set]), "v"(xf[set][i]));
#else
                    if (j < NI) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][j], xf[set][i], acc[i][j], 0, 0, 0);
                    else if (set == wn) accl[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf[set], xf[set][i], accl[i], 0, 0, 0);
#endif
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

    // ---- epilogue ---------------------------------------------------------------------------------
    // acc[i][j][r] = C[m = m0 + wm*32*MI + i*32 + lrow][n = n0 + wn*32*NI + j*32 + (r&3) + 8*(r>>2) + 4*lhi]
    if (p.splitk > 1) {
        // split-K, reduced inside the launch in a FIXED order (no fp32 atomics: they commit in arrival order, the sum's last
        // bits - hence bf16 roundings downstream - would differ from run to run).  Every K slice publishes its partial tile
        // in ITS OWN fp32 slab (write-through stores), drains them and takes a ticket of the tile; the slice that draws the
        // last ticket adds all slabs IN SLICE ORDER (its own included, read back like the others: which slice is last
        // varies, the arithmetic does not) and runs the ordinary epilogue below.  The other slices are done.
        // The ticket also counts arrivals per XCD (7-bit fields above the 8-bit count): when every slice ran on the reader's
        // XCD - the launch order puts the slices of a tile on consecutive block ids of one XCD - the partials are read from
        // that XCD's L2 (workgroup-scope loads: only the L1 is bypassed); otherwise from memory (agent scope).
        // Slab layout = the accumulator layout: slab[slice][tile][wave][block i,j][q][lane] holds 4 floats, so every store /
        // load instruction of a wave moves 1 KB of consecutive addresses and nothing needs a bounds check (rows / columns past
        // M / N are padding inside the slab).  (8-byte row-major pieces, written through, cost ~50 us per launch.)
        constexpr int TILE_BYTES = BM * BN * 4;
        const unsigned slab_bytes = (unsigned)(p.tiles_m * p.tiles_n) * TILE_BYTES;
        const __amdgpu_buffer_rsrc_t slabs = __builtin_amdgcn_make_buffer_rsrc(p.c32, 0, (int)(p.splitk * slab_bytes), 0x00020000);
        unsigned lane_off = (unsigned)(tile_m * p.tiles_n + tile_n) * TILE_BYTES + wave * (MI * NI * 4096) + lane * 16;
        asm volatile("" : "+v"(lane_off));      // not to be formed ahead of the K loop and carried through it
        int lrow_s = lrow;
        asm volatile("" : "+v"(lrow_s));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slabs,
                                                           ks_id * slab_bytes + lane_off + ((i * NI + j) * 4 + q) * 1024, 0, 16 /* sc1 */);
                }
            const int m = m0 + wm * (32 * MI) + i * 32 + lrow_s;
            if (m >= p.M) continue;
            if (LORA) {
                // T of the tile's rows: each wave of a row pair holds the partial of its own k-steps -> 2 slabs per slice, per
                // column tile (every column tile reduces its own copy: its last slice cannot wait for another tile's slices)
                // rank index of accl[i][r]: (r&3) + 8*(r>>2) + 4*lhi  ->  ranks 0-3 / 8-11 in the lhi=0 half, 4-7 in lhi=1
                float* ts = p.t32 + (((long)tile_n * 2 * p.splitk + ks_id * 2 + wn) * p.M + m) * p.ld_t;
                if (4 * lhi < p.lora_rank) {
                    store_pair_sc1(ts + 4 * lhi, accl[i][0], accl[i][1]);
                    store_pair_sc1(ts + 4 * lhi + 2, accl[i][2], accl[i][3]);
                }
                if (lhi == 0 && 8 < p.lora_rank) {
                    store_pair_sc1(ts + 8, accl[i][4], accl[i][5]);
                    store_pair_sc1(ts + 10, accl[i][6], accl[i][7]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's partials have reached the memory side
        __syncthreads();                                       // ... and every wave is done with the operand stages
        unsigned long long& arrival = *(unsigned long long*)smem;
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID
        if (tid == 0) {
            unsigned long long* ticket = p.ticket + (long)tile_m * p.tiles_n + tile_n;
            const unsigned long long mine = 1ull + (1ull << (8 + 7 * xcc));
            const unsigned long long seen = __hip_atomic_fetch_add(ticket, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
            if ((int)(seen & 255) == p.splitk) __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrival = seen;
        }
        __syncthreads();
        const unsigned long long seen = arrival;
        if ((int)(seen & 255) != p.splitk) return;
        const bool local = (int)((seen >> (8 + 7 * xcc)) & 127) == p.splitk;     // uniform over the workgroup
        // slice 0 is loaded straight into the accumulators (their contents are in the slabs now), every further slice as
        // batches of independent 16-byte loads (one 32-row block): the serial part is one round trip per slice and block
        auto reduce = [&](auto kLocal) {
            constexpr int kAux = decltype(kLocal)::value ? 1 /* sc0: this XCD's L2 */ : 16 /* sc1: memory */;
            auto ld = [&](const float* q) { return decltype(kLocal)::value ? load_pair_l2(q) : load_pair_sc1(q); };
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            slabs, lane_off + ((i * NI + j) * 4 + q) * 1024, 0, kAux));
                        acc[i][j][q * 4] = v[0]; acc[i][j][q * 4 + 1] = v[1]; acc[i][j][q * 4 + 2] = v[2]; acc[i][j][q * 4 + 3] = v[3];
                    }
                for (int k = 1; k < p.splitk; ++k) {
                    f32x4 t[NI * 4];
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            t[j * 4 + q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                slabs, k * slab_bytes + lane_off + ((i * NI + j) * 4 + q) * 1024, 0, kAux));
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[i][j][q * 4] += t[j * 4 + q][0]; acc[i][j][q * 4 + 1] += t[j * 4 + q][1];
                            acc[i][j][q * 4 + 2] += t[j * 4 + q][2]; acc[i][j][q * 4 + 3] += t[j * 4 + q][3];
                        }
                }
                const int m = m0 + wm * (32 * MI) + i * 32 + lrow_s;
                if (m >= p.M) continue;
                if (LORA) {
                    // this wave's own half (wn) of every slice, in slice order; the halves meet in the epilogue's exchange
                    const long tstride = 2L * p.M * p.ld_t;
                    const float* ts = p.t32 + (((long)tile_n * 2 * p.splitk + wn) * p.M + m) * p.ld_t;
                    const bool r0 = 4 * lhi < p.lora_rank, r8 = lhi == 0 && 8 < p.lora_rank;
                    f32x2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f}, t2 = {0.f, 0.f}, t3 = {0.f, 0.f};
                    for (int k = 0; k < p.splitk; ++k) {
                        const float* tk = ts + k * tstride;
                        const f32x2 a0 = r0 ? ld(tk + 4 * lhi) : f32x2{0.f, 0.f}, a1 = r0 ? ld(tk + 4 * lhi + 2) : f32x2{0.f, 0.f};
                        const f32x2 a2 = r8 ? ld(tk + 8) : f32x2{0.f, 0.f}, a3 = r8 ? ld(tk + 10) : f32x2{0.f, 0.f};
                        t0 += a0; t1 += a1; t2 += a2; t3 += a3;
                    }
                    accl[i][0] = t0[0]; accl[i][1] = t0[1]; accl[i][2] = t1[0]; accl[i][3] = t1[1];
                    accl[i][4] = t2[0]; accl[i][5] = t2[1]; accl[i][6] = t3[0]; accl[i][7] = t3[1];
                }
            }
        };
        if (local) reduce(std::true_type{});
        else reduce(std::false_type{});
    }
    // ---- per-column epilogue vectors (bias, per-sample row bias, LayerNorm-fold s / b'), once per workgroup through LDS -------
    // Read per accumulator quad from global memory they were 4-16 SERIAL round trips at the end of every tile (each load sat
    // behind a condition: hipcc waits vmcnt(0) at the join); a tile's columns share them, so one thread per column fetches its
    // four values in one round trip, behind the patches of the store staging.  A row bias qualifies when the tile's rows belong
    // to one sample (rows_per_sample a multiple of the tile height); otherwise it stays a per-row load below.
    constexpr int S = 4 * NI;                  // 16-byte slots per staged row
    constexpr int LOG2S = NI == 2 ? 3 : 2;
    float* sCol = (float*)(smem + NW * (32 * S * 16) + (LORA ? NW * MI * 2048 : 0));     // [4][BN]
    const bool rb_tile = p.rowbias != nullptr && p.rows_per_sample % BM == 0;
    {
        const int n = n0 + tid;
        const bool nok = tid < BN && n < p.N;
        const __bf16* zb = (const __bf16*)slh_zero_page;
        const float* zf = (const float*)slh_zero_page;
        const float c0 = (float)*((p.bias && nok) ? p.bias + n : zb);
        const float c1 = (float)*((rb_tile && nok) ? p.rowbias + (long)(m0 / p.rows_per_sample) * p.ld_rowbias + n : zb);
        const float c2 = *((ln_on && nok) ? p.ln_s + n : zf);
        const float c3 = *((ln_on && nok) ? p.ln_b + n : zf);
        __syncthreads();                       // every wave is done reading the operand stages being reused below
        if (tid < BN) { sCol[tid] = c0; sCol[BN + tid] = c1; sCol[2 * BN + tid] = c2; sCol[3 * BN + tid] = c3; }
    }
    if (p.geglu == 2) {
        // Backward of GEGLU fused into the backward-data product of the Linear behind it (ff.net.2): the accumulators are
        // d(ff) - rounded to bf16 as the unfused path stores it - and leave as d(proj) in proj's blocked column order, computed
        // from the forward's pre-activation (slh_elementwise GEGLU_BWD arithmetic, one launch and one HBM round trip of
        // d(ff) less):  d_h = dd * bf16(g * Phi(g)),  d_g = bf16(dd * h) * (Phi(g) + g * phi(g)).
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * (32 * MI) + i * 32 + lrow;
            const bool mok = m < p.M;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                bf16x4 h4[4], g4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {        // the block's pre-activation quads, requested together
                    const int n = n0 + wn * (32 * NI) + j * 32 + q * 8 + lhi * 4;
                    const __bf16* src = (mok && n < p.N) ? p.geglu_pre + (long)m * p.ld_pre + (n >> 5) * 64 + (n & 31)
                                                         : (const __bf16*)slh_zero_page;
                    h4[q] = *(const bf16x4*)src;
                    g4[q] = *(const bf16x4*)(src + ((mok && n < p.N) ? 32 : 0));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * (32 * NI) + j * 32 + q * 8 + lhi * 4;
                    if (!mok || n >= p.N) continue;
                    bf16x4 dh, dg;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = (float)g4[q][e], hh = (float)h4[q][e], dd = round_bf16(acc[i][j][q * 4 + e]);
                        const float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752f));
                        const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
                        dh[e] = (__bf16)(dd * round_bf16(g * cdf));
                        dg[e] = (__bf16)(round_bf16(dd * hh) * (cdf + g * pdf));
                    }
                    __bf16* po = p.c + (long)m * p.ldc + (n >> 5) * 64 + (n & 31);
                    *(bf16x4*)po = dh;
                    *(bf16x4*)(po + 32) = dg;
                }
            }
        }
        return;
    }
    if (p.geglu) {
        // GEGLU: W rows are stored in 64-row blocks [32 value rows | 32 gate rows]; NI is 2 here, so
        // sub-tile j=0 holds the values and j=1 the gates of the same 32 output columns.
        __syncthreads();
        if (NI == 2) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (32 * MI) + i * 32 + lrow;
                if (m >= p.M) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + q * 8 + lhi * 4;          // row index of the value rows
                    const int nout = ((n0 + wn * 64) >> 1) + q * 8 + lhi * 4;
                    if (n + 32 >= p.N) continue;
                    float a[4], g[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = acc[i][0][q * 4 + e]; g[e] = acc[i][NI - 1][q * 4 + e]; }
                    const int cl = n - n0;                                  // column inside the tile
                    if (MODE == 0 && ln_on) {
                        const f32x4 sa = *(const f32x4*)(sCol + 2 * BN + cl), sg = *(const f32x4*)(sCol + 2 * BN + cl + 32);
                        const f32x4 ba = *(const f32x4*)(sCol + 3 * BN + cl), bg = *(const f32x4*)(sCol + 3 * BN + cl + 32);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = ln_rstd[i] * (a[e] - ln_mean[i] * sa[e]) + ba[e];
                            g[e] = ln_rstd[i] * (g[e] - ln_mean[i] * sg[e]) + bg[e];
                        }
                    }
                    if (p.bias) {
                        const f32x4 ba = *(const f32x4*)(sCol + cl), bg = *(const f32x4*)(sCol + cl + 32);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] += ba[e]; g[e] += bg[e]; }
                    }
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // reference rounds proj(x) to bf16 before chunk/gelu
                        const float av = round_bf16(a[e]), gv = round_bf16(g[e]);
                        o[e] = (__bf16)(av * round_bf16(gelu_erf_f(gv)));
                    }
                    *(bf16x4*)(p.c + (long)m * p.ldc + nout) = o;
                    if (p.geglu_pre) {      // training: proj(x) itself, in the column order of this product, for the GEGLU backward
                        *(bf16x4*)(p.geglu_pre + (long)m * p.ld_pre + n) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
                        *(bf16x4*)(p.geglu_pre + (long)m * p.ld_pre + n + 32) = bf16x4{(__bf16)g[0], (__bf16)g[1], (__bf16)g[2], (__bf16)g[3]};
                    }
                }
            }
        }
        return;
    }

    // The MFMA result layout gives a lane 4 consecutive columns of ONE row, so a direct store touches 32 rows per
    // instruction with 8-byte pieces (store-issue bound).  Each wave therefore transposes its 32 x (32*NI) sub-tile
    // through a private, swizzled LDS patch and writes whole 64/128-byte row segments with 16-byte stores.
    if (LORA) {
        // T = x . A^T of a row block was accumulated half by each of the two waves that own those rows (wn = 0 took the
        // even k-steps, wn = 1 the odd ones: the adapter costs half an MFMA per k-step and wave instead of one); the
        // partials are exchanged through LDS behind the staging patches.  Only registers 0-7 of accl carry ranks < 12.
        float* ex = (float*)(smem + NW * (32 * S * 16));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) ex[(wave * MI * 8 + i * 8 + r) * 64 + lane] = accl[i][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) accl[i][r] += ex[((wave ^ 1) * MI * 8 + i * 8 + r) * 64 + lane];
    } else {
        __syncthreads();                       // the column vectors are in LDS
    }
    char* sE = smem + wave * (32 * S * 16);
    const float lscale = (LORA || p.lora_t != nullptr) ? *p.lora_scale : 0.f;
    const int ncol0 = n0 + wn * (32 * NI);
    // Fused adapter, forward form (lora_up [N][4]): the up-projection  scale * T . B^T  is one more MFMA per accumulator
    // tile.  T (ranks x rows, already in the accumulator layout of a B operand up to a fixed permutation of the rank
    // index) is scaled and rounded to bf16 - the reference's down-projection output is a bf16 tensor too - and the A
    // operand holds, for output column n of group g, B[n][0..3] at the k positions of ranks 4g..4g+3 and zeros elsewhere.
    // k index of a lane: 8*lhi + e  <->  rank: lhi = 0: e < 4 -> e, e >= 4 -> 8 + (e - 4);  lhi = 1: e < 4 -> 4 + e, else unused
    const bool mfma_up = LORA && !p.lora_up_rmajor;
    if (mfma_up) {
        bf16x8 ua[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = ncol0 + j * 32 + lrow;
            bf16x4 u4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            int g = -1;
            if (n < p.N) { u4 = *(const bf16x4*)(p.lora_up + (long)n * 4); g = n / p.lora_cols_per_group; }
            const bool lo = lhi == 0 ? g == 0 : g == 1;        // e < 4: ranks 0-3 (lhi 0) / 4-7 (lhi 1)
            const bool hi = lhi == 0 && g == 2;                // e >= 4: ranks 8-11 (lhi 0 only)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ua[j][e] = lo ? u4[e] : (__bf16)0.f;
                ua[j][4 + e] = hi ? u4[e] : (__bf16)0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            bf16x8 tb;
#pragma unroll
            for (int e = 0; e < 8; ++e) tb[e] = (__bf16)(lscale * accl[i][e]);
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[j], tb, acc[i][j], 0, 0, 0);
        }
    }
    const bool have_t = (LORA || p.lora_t != nullptr) && !mfma_up;      // the per-element forms below
    // wave-uniform: this wave's columns belong to the V block that slh_attn_fwd wants head-transposed
    const bool to_vt = p.vt != nullptr && ncol0 >= p.vt_col0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mbase = m0 + wm * (32 * MI) + i * 32;
        const int m = mbase + lrow;
        const bool mok = m < p.M;
        float ln_k = 0.f, ln_sum = 0.f, ln_sq = 0.f;
        f32x4 tv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) tv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (LORA && (have_t || p.lora_t_out)) {
            // ranks 0-3 sit in registers 0-3 of the lhi=0 half, 4-7 in registers 0-3 of the lhi=1 half, 8-11 in
            // registers 4-7 of the lhi=0 half: one exchange with lane^32 gives every lane all of its row's T
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = accl[i][e], x1 = accl[i][4 + e];
                const float y0 = __shfl_xor(x0, 32, 64), y1 = __shfl_xor(x1, 32, 64);
                tv[0][e] = lhi == 0 ? x0 : y0;
                tv[1][e] = lhi == 1 ? x0 : y0;
                tv[2][e] = lhi == 0 ? x1 : y1;
            }
            if (mok && p.lora_t_out && tile_n == 0 && wn == 0 && lhi == 0) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    if (g * 4 < p.lora_rank) *(f32x4*)(p.lora_t_out + (long)m * p.ld_t + g * 4) = tv[g];
            }
        } else if (!LORA && p.lora_t && mok) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
                if (g * 4 < p.ld_t) tv[g] = *(const f32x4*)(p.lora_t + (long)m * p.ld_t + g * 4);
        }
        // a row bias whose tile spans samples stays a per-row load
        const __bf16* rb = (p.rowbias && !rb_tile && mok) ? p.rowbias + (long)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
        const int hb = (lrow >> LOG2S) & 1;
        bf16x4 okeep[NI][4];          // vt_also_c: the rounded quads, for the row-major store behind the transposed one
        // the residual quads of the whole 32-row block are requested together (one round trip instead of one per quad)
        bf16x4 res4[NI][4];
        if (p.residual) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = ncol0 + j * 32 + q * 8 + lhi * 4;
                    res4[j][q] = *((mok && n < p.N) ? (const bf16x4*)(p.residual + (long)m * p.ld_res + n) : (const bf16x4*)slh_zero_page);
                }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = ncol0 + j * 32 + q * 8 + lhi * 4;
                const int cl = n - n0;                                      // column inside the tile
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
                if (mok && n < p.N) {
                    if (MODE == 0 && ln_on) {
                        const f32x4 s4 = *(const f32x4*)(sCol + 2 * BN + cl), b4 = *(const f32x4*)(sCol + 3 * BN + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ln_rstd[i] * (v[e] - ln_mean[i] * s4[e]) + b4[e];
                    }
                    if (p.bias) {
                        const f32x4 b4 = *(const f32x4*)(sCol + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += b4[e];
                    }
                    if (rb_tile) {
                        const f32x4 b4 = *(const f32x4*)(sCol + BN + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += b4[e];
                    }
                    if (rb) {
                        const bf16x4 b4 = *(const bf16x4*)(rb + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)b4[e];
                    }
                    if (have_t && !p.lora_up_rmajor) {
                        const int g = n / p.lora_cols_per_group;
                        const f32x4 t = g == 0 ? tv[0] : (g == 1 ? tv[1] : tv[2]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bf16x4 u = *(const bf16x4*)(p.lora_up + (long)(n + e) * 4);
                            v[e] += lscale * (t[0] * (float)u[0] + t[1] * (float)u[1] +
                                              t[2] * (float)u[2] + t[3] * (float)u[3]);
                        }
                    } else if (have_t) {
                        // backward-data form: the "up" matrix is lora_down as stored, [rank][N], rank 4..12
                        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            if (g * 4 < p.lora_rank) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const bf16x4 u = *(const bf16x4*)(p.lora_up + (long)(g * 4 + r) * p.N + n);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) s4[e] += tv[g][r] * (float)u[e];
                                }
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += lscale * s4[e];
                    }
                    if (p.residual) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)res4[j][q][e];
                    }
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                if (p.ln_out) {                 // statistics of the stored (rounded) values, shifted by the lane's first one
                    if (j == 0 && q == 0) { ln_k = (float)o[0]; ln_sum = 0.f; ln_sq = 0.f; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float dlt = (float)o[e] - ln_k; ln_sum += dlt; ln_sq += dlt * dlt; }
                }
                if (to_vt) {
                    // transposed patch sT[n_local][m_local] (32*NI rows of 32 bf16): column n of the tile becomes a
                    // 64-byte run along m
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *(__bf16*)(sE + ((j * 32 + q * 8 + lhi * 4 + e) * 32 + lrow) * 2) = o[e];
                    okeep[j][q] = o;
                    continue;
                }
                // logical 16-byte slot j*4+q, 8-byte half lhi of row lrow; slot ^ row and half ^ row-bit keep both
                // the 8-byte writes and the 16-byte row reads off each other's banks
                const int slot = (j * 4 + q) ^ (lrow & (S - 1));
                *(bf16x4*)(sE + lrow * (S * 16) + slot * 16 + ((lhi ^ hb) << 3)) = o;
            }
        }
        if (NI == 2 && p.ln_out) {
            // LayerNorm statistics of this row's 64 columns (producer side of the folded LayerNorm): the lane holds 32 of
            // them, its partner lane^32 the other 32; (mean, M2) from sums shifted by a sample of the row (no cancellation), merged (Chan)
            const float dm = ln_sum * (1.f / (16 * NI));
            const float mu = ln_k + dm;
            const float m2 = fmaxf(ln_sq - ln_sum * dm, 0.f);
            const float mu_o = __shfl_xor(mu, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
            const float delta = mu_o - mu;
            if (lhi == 0 && mok && ncol0 < p.N)
                *(f32x2*)(p.ln_out + ((long)(ncol0 >> 6) * p.M + m) * 2) =
                    f32x2{mu + 0.5f * delta, m2 + m2_o + delta * delta * (8.f * NI)};
        }
        __builtin_amdgcn_wave_barrier();       // same-wave LDS ops retire in order; only the compiler must not reorder
        if (to_vt) {
            const int Dp = (p.vt_D + 63) & ~63;
#pragma unroll
            for (int it = 0; it < 2 * NI; ++it) {
                const int idx = it * 64 + lane;
                const int nl = idx >> 2, seg = idx & 3;
                const bf16x8 t8 = *(const bf16x8*)(sE + nl * 64 + seg * 16);
                const int m2 = mbase + seg * 8, n2 = ncol0 + nl;
                if (m2 < p.M && n2 < p.N) {
                    const int nv = n2 - p.vt_col0;
                    const int hh = nv / p.vt_D, dd = nv - hh * p.vt_D;
                    const int bb = m2 / p.vt_tokens, tt = m2 - bb * p.vt_tokens;
                    *(bf16x8*)(p.vt + (((long)bb * p.vt_heads + hh) * Dp + dd) * p.vt_ld + tt) = t8;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (!p.vt_also_c) continue;
            // ... and row-major into c as well (training: the backward reads V / dO in both layouts): the same quads through the
            // row patch
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int slot = (j * 4 + q) ^ (lrow & (S - 1));
                    *(bf16x4*)(sE + lrow * (S * 16) + slot * 16 + ((lhi ^ hb) << 3)) = okeep[j][q];
                }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int it = 0; it < S / 2; ++it) {
            const int idx = it * 64 + lane;
            const int row = idx / S, slot = idx % S;
            bf16x8 t8 = *(const bf16x8*)(sE + row * (S * 16) + ((slot ^ (row & (S - 1))) << 4));
            if ((row >> LOG2S) & 1) t8 = __builtin_shufflevector(t8, t8, 4, 5, 6, 7, 0, 1, 2, 3);
            const int m2 = mbase + row, n2 = ncol0 + slot * 8;
            if (m2 < p.M && n2 < p.N) {
                __bf16* dst = p.c + (long)m2 * p.ldc + n2;
                if (n2 + 8 <= p.N) {
                    if (p.store16) {
                        *(bf16x8*)dst = t8;
                    } else {
                        *(bf16x4*)dst = __builtin_shufflevector(t8, t8, 0, 1, 2, 3);
                        *(bf16x4*)(dst + 4) = __builtin_shufflevector(t8, t8, 4, 5, 6, 7);
                    }
                } else {
                    *(bf16x4*)dst = __builtin_shufflevector(t8, t8, 0, 1, 2, 3);   // N % 4 == 0: exactly 4 columns left
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MI, int NI, int MODE, bool LORA, int WM>
int launch_gemm3(const GemmArgs& a, int stages, hipStream_t s) {
    const int grid = a.tiles_m * a.tiles_n * (a.splitk > 1 ? a.splitk : 1);
    constexpr int stage_bytes = (32 * MI * WM + 64 * NI + (LORA ? 32 : 0)) * 128;
    constexpr bool can3 = 3 * stage_bytes <= 160 * 1024;
    constexpr bool can4 = 4 * stage_bytes <= 160 * 1024;
    if (stages == 4 && !can4) stages = 3;          // the deepest ring that fits the 160 KB of LDS
    if (stages == 3 && !can3) stages = 2;
    if (stages == 4)
        hipLaunchKernelGGL((gemm_kernel<MI, NI, MODE, (can4 ? 4 : 2), LORA, WM>), dim3(grid), dim3(128 * WM), 0, s, a);
    else if (stages == 3)
        hipLaunchKernelGGL((gemm_kernel<MI, NI, MODE, (can3 ? 3 : 2), LORA, WM>), dim3(grid), dim3(128 * WM), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_kernel<MI, NI, MODE, 2, LORA, WM>), dim3(grid), dim3(128 * WM), 0, s, a);
    SLH_LAUNCH_CHECK("slh_gemm");
    return 0;
}

template <int MI, int NI, int WM>
int launch_gemm(const GemmArgs& a, int mode, int stages, hipStream_t s) {
    const bool lora = a.lora_down != nullptr;
    if (mode == 0) return lora ? launch_gemm3<MI, NI, 0, true, WM>(a, stages, s) : launch_gemm3<MI, NI, 0, false, WM>(a, stages, s);
    return lora ? launch_gemm3<MI, NI, 1, true, WM>(a, stages, s) : launch_gemm3<MI, NI, 1, false, WM>(a, stages, s);
}

}  // namespace

// tile choice: explicit (d->tile, set by the planner from the tuned table) or a fill-the-chip heuristic
static void pick_tile(const slh_gemm_desc* d, int& MI, int& NI, int& WM) {
    MI = 2; NI = 2; WM = 2;
    if (d->tile) {
        MI = (d->tile >> 4) & 15; NI = d->tile & 15;
        WM = ((d->tile >> 12) & 15) == 4 ? 4 : 2;
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

// (MI<<8)|(NI<<4)|mode of the kernel instantiation slh_gemm would launch: gemm_kernel<MI, NI, mode>
extern "C" int slh_gemm_variant(const slh_gemm_desc* d) {
    int MI, NI, WM;
    pick_tile(d, MI, NI, WM);
    return (WM << 12) | (MI << 8) | (NI << 4) | (d->mode & 15);
}

extern "C" int slh_gemm(const slh_gemm_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->a0 && d->w && d->c, "slh_gemm: null pointer");
    SLH_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "slh_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    SLH_CHECK(d->K % 64 == 0, "slh_gemm: K=%d must be a multiple of 64", d->K);
    SLH_CHECK(d->N % 4 == 0, "slh_gemm: N=%d must be a multiple of 4", d->N);
    SLH_CHECK(d->ca0 % 64 == 0 && d->ca1 % 64 == 0, "slh_gemm: channel counts must be multiples of 64");
    SLH_CHECK((d->a1 != nullptr) == (d->ca1 > 0), "slh_gemm: a1/ca1 mismatch");
    SLH_CHECK(d->lda0 % 8 == 0 && d->lda1 % 8 == 0 && d->ldw % 8 == 0 && d->ldc % 4 == 0,
              "slh_gemm: leading dimensions must keep 16-byte loads / 8-byte stores aligned");
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
    SLH_CHECK(d->geglu >= 0 && d->geglu <= 2, "slh_gemm: geglu is 0, 1 (forward epilogue) or 2 (backward form)");

    int MI = 2, NI = 2, WM = 2;
    pick_tile(d, MI, NI, WM);
    SLH_CHECK((MI == 1 || MI == 2) && (NI == 1 || NI == 2), "slh_gemm: bad tile");
    SLH_CHECK(WM == 2 || NI == 2 || MI == 1, "slh_gemm: 8-wave tiles are 128x64, 128x128 or 256x128");
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
        SLH_CHECK(d->mode == 0 && !d->a1 && d->ln_s && d->ln_b && !d->bias && !d->lora_down && !d->lora_t,
                  "slh_gemm: ln_in needs a dense single-source product, ln_s / ln_b, no bias (folded into ln_b), no adapter");
        SLH_CHECK(d->ln_in_chunks >= 1 && d->ln_in_chunks <= 20 && d->K == 64 * d->ln_in_chunks,
                  "slh_gemm: ln_in_chunks must be K / 64 (<= 20)");
        SLH_CHECK(((uintptr_t)d->ln_in & 7) == 0 && ((uintptr_t)d->ln_s & 15) == 0 && ((uintptr_t)d->ln_b & 15) == 0,
                  "slh_gemm: ln_in / ln_s / ln_b alignment");
    }
    a.probe = d->reserved_;
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
    if (a.splitk > 1) {
        // every slice must be non-empty: each publishes its whole partial tile, the last one to arrive reads them all
        const int nk = d->K / 64;
        const int per = (nk + a.splitk - 1) / a.splitk;
        a.splitk = (nk + per - 1) / per;
    }
    if (a.splitk > 1) {
        SLH_CHECK(d->splitk_c32, "slh_gemm: split-K needs the fp32 slab workspace splitk_c32");
        SLH_CHECK(d->splitk_slabs >= a.splitk, "slh_gemm: split-K into %d slices but the workspace holds %d slabs", a.splitk,
                  d->splitk_slabs);
        SLH_CHECK(d->splitk_ticket, "slh_gemm: split-K needs the arrival tickets splitk_ticket (zeroed once)");
        SLH_CHECK((long)a.splitk * ((d->M + 255) / 256 * 256L) * ((d->N + 127) / 128 * 128L) * 4 < (1L << 31),
                  "slh_gemm: split-K slabs beyond 2 GB");
        SLH_CHECK(((uintptr_t)d->splitk_ticket & 7) == 0 && d->N % 4 == 0, "slh_gemm: split-K needs N %% 4 == 0 and 8-byte aligned tickets");
        SLH_CHECK(!d->lora_down || a.t32, "slh_gemm: split-K with a fused adapter needs the slab workspace splitk_t32");
        SLH_CHECK(((uintptr_t)d->splitk_c32 & 15) == 0 && ((uintptr_t)d->splitk_t32 & 15) == 0, "slh_gemm: slab alignment");
    } else {
        a.splitk = 1;
    }
    a.store16 = (d->ldc % 8 == 0) && (((uintptr_t)d->c & 15) == 0);
    a.tiles_m = (d->M + 32 * MI * WM - 1) / (32 * MI * WM);
    a.tiles_n = (d->N + 64 * NI - 1) / (64 * NI);
    a.group_m = pick_group_m(d, a.tiles_m);
    hipStream_t s = (hipStream_t)stream;
    const int stages = (d->tile >> 8) & 15;   // tile = (WM<<12)|(stages<<8)|(MI<<4)|NI ; stages 0/2 = double buffer
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
