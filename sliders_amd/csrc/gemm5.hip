// 64 x 160 tile of the bf16 MFMA GEMM (round 5): the M = 2048, N = 1280 products of the no-grad passes as ONE full round of the chip.
//
// Why another tile.  The 128 x 128 ring tile (gemm.hip) covers 2048 x 1280 with 160 workgroups on 256 CUs and walks K at ~880
// clocks per 64-deep K tile: per K tile a CU moves 32 KB of operands into LDS, reads 96 KB of fragments back out of it (8 waves of
// 32 x 64: 1.5 KB per MFMA) and issues 512 clocks of MFMAs per SIMD - the LDS array is busy ~625 of those clocks.  A 64 x 160 tile
// makes 32 x 8 = 256 workgroups = every CU, each staging 28 KB per K tile and reading 56 KB of fragments (4 waves of 32 x 80 on the
// 16 x 16 x 32 MFMA: 0.7 KB per MFMA, 5 column blocks per wave share two row fragments) for 340 clocks of MFMAs per SIMD.  Same
// 4-slot LDS ring and the same one-barrier-per-K-tile protocol as gemm.hip (LDS-DMA hidden from the compiler, counted vmcnt,
// fragments software-pipelined across k-steps and K tiles), same XCD-aware grouped tile order, same packed-weight layout.
//
// Scope: dense single-source products with packed weights, M % 64 == 0, N % 160 == 0, K % 64 == 0; epilogue = bias + residual, one
// rounding, 16-byte write-through row stores through a per-wave LDS patch, and the producer side of a folded LayerNorm with
// 80-COLUMN chunks (ln_out [N/80][M][2]: the consumer merges equal-sized chunks of any width), or its consumer side (ln_in: the
// chunk statistics of a row are merged by its four lanes behind the K loop, round 6); optionally ONE fused rank-4 adapter
// (lora.py:108-112: the down matrix rides as 8 extra rows of the W tile - 4 of them zero -, T = x . A^T costs one MFMA per row block
// and K tile in each wave, the up-projection one MFMA per accumulator block in the epilogue).  No split-K, no GEGLU, no row bias.
// Replaces F.linear inside diffusers' Attention.to_out[0] / Transformer2DModel.proj_out / FeedForward.net[2] as called from
// trainscripts/textsliders/train_util.py:242-247.
#include "gemm_common.h"

using namespace slh_gemm_detail;

namespace {

constexpr int G5_BM = 64, G5_BN = 160;
constexpr int G5_WROWS = G5_BN + 8;                   // W tile rows in LDS: 160 + the adapter's 8 (rows 160-163 = lora_down, 164-167 zero)
constexpr int G5_STAGE = (G5_BM + G5_WROWS) * 128;   // 29 KB per ring slot
constexpr int G5_XI = 2, G5_WI = 5, G5_L = G5_XI + G5_WI;   // LDS-DMA instructions per wave per K tile (8 rows of 128 B each)
constexpr int G5_PATCH_LD = 176;                      // bytes per row of a wave's 32 x 80 bf16 store patch (160 + 16: rows 16-byte aligned)

struct G5Args {
    const __bf16* a; const __bf16* w; const __bf16* bias; const __bf16* residual; __bf16* c; float* ln_out;
    const __bf16* lora_down; const __bf16* lora_up; const float* lora_scale; float* lora_t_out;
    const float* ln_in; const float* ln_s; const float* ln_b; float* ln_mr_out;
    int lda, ldc, ld_res, M, N, K, tiles_m, tiles_n, group_m, ld_t, skew;
    int ln_in_chunks; float ln_eps;
};
constexpr int G5_LNC = 6;      // chunk pairs a lane requests per row: chunk 0 (the shift) + its quarter of up to 20 chunks

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// (built with -mllvm -amdgpu-mfma-vgpr-form, Makefile: one wave per SIMD gives hipcc a 512-register budget, and it then picks the
// AGPR form of the MFMAs and copies every accumulator through VGPRs around them)
// G5_S: ring slots (tile code bits 8-11: 4 = 116 KB of LDS, 5 = 145 KB: one more K tile of LDS-DMA in flight)
template <bool LORA, int G5_S>
__global__ __launch_bounds__(256) void gemm5_kernel(const G5Args p) {
    __shared__ __attribute__((aligned(16))) char smem[G5_S * G5_STAGE];
    char* sX = smem;                              // [S][64][128 B]
    char* sW = smem + G5_S * G5_BM * 128;         // [S][168][128 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    {
        // XCD-aware grouped order (gemm_common.h): XCD x owns a contiguous run of the sequence [groups of group_m row tiles: m fastest, then n]
        const int nblk = (int)gridDim.x;
        const int bid = gemm_remap_bid(nblk);
        const int gsz = p.group_m * p.tiles_n;
        const int g = bid / gsz;
        const int first_m = g * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - first_m);
        const int r = bid - g * gsz;
        tile_n = r / gm;
        tile_m = first_m + r - tile_n * gm;
    }
    const int m0 = tile_m * G5_BM, n0 = tile_n * G5_BN;
    const int nk = p.K >> 6;

    // ---- per-lane fill geometry: one LDS-DMA instruction writes 8 rows x 128 B, lane -> (row = lane / 8, physical slot = lane % 8)
    const int frow = lane >> 3, fslot = lane & 7;
    const char* xsrc[G5_XI];
    const char* wsrc[G5_WI];
    // Skew (p.skew): the workgroups that share an operand panel inside an XCD run in step, so without it they all ask the L2 for the
    // SAME 1 KB piece at the same moment.  Row tile tm starts its walk over the W panel's 20 pieces at piece 4 * (tm % 5), column
    // tile tn its walk over the X panel's 8 pieces at piece 4 * (tn % 2): same pieces per K tile, a different order per sibling.
    const int wrot = p.skew ? 4 * (tile_m % 5) : 0, xrot = p.skew ? 4 * (tile_n & 1) : 0;
    unsigned xdst[G5_XI], wdst[G5_WI];       // byte offset of the piece inside its ring slot's X / W tile
#pragma unroll
    for (int i = 0; i < G5_XI; ++i) {
        const int pc = (wave + 4 * i + xrot) & 7;
        const int row = pc * 8 + frow;
        xdst[i] = pc * 1024;
        xsrc[i] = (const char*)(p.a + (long)(m0 + row) * p.lda + ((fslot ^ ((row >> 1) & 7)) << 3));
    }
#pragma unroll
    for (int i = 0; i < G5_WI; ++i) {
        int pc = wave + 4 * i + wrot;
        pc = pc >= 20 ? pc - 20 : pc;
        const int n = n0 + pc * 8 + frow;       // (the swizzle key of a packed row has period 16: a tile may start at row 0 / 32 of a block)
        wdst[i] = pc * 1024;
        wsrc[i] = (const char*)(p.w + ((long)(n >> 6) * (p.K >> 6)) * 4096 + ((n & 63) << 6) + (fslot << 3));
    }
    // fused adapter: wave 3 stages one more piece per K tile, rows 160-167 of the W tile = lora_down[0..3][k tile] + 4 zero rows (its
    // counted waits are one piece deeper: `extra`, wave-uniform)
    const bool extra = LORA && wave == 3;
    const char* lsrc = (const char*)slh_zero_page;
    int ladv = 0;
    if (LORA && frow < 4) {
        lsrc = (const char*)(p.lora_down + (long)frow * p.K + ((fslot ^ ((frow >> 1) & 7)) << 3));
        ladv = 128;
    }
    // folded LayerNorm, consumer side: the chunk statistics of this wave's rows are requested HERE, ahead of every LDS-DMA piece
    // (loads retire in order: older ordinary loads never make a counted wait on the pieces too short), and merged behind the K loop
    const int r16 = lane & 15, g4 = lane >> 4;
    f32x2 lnp[2][G5_LNC];
    if (p.ln_in) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x2* src = (const f32x2*)p.ln_in + m0 + wm * 32 + i * 16 + r16;        // chunk-major [chunks][M]
#pragma unroll
            for (int c = 0; c < G5_LNC; ++c) {
                const int ch = c == 0 ? 0 : g4 + 4 * (c - 1);
                lnp[i][c] = src[(long)(ch < p.ln_in_chunks ? ch : 0) * p.M];
            }
        }
    }
    const unsigned lds0 = lds_addr_of(smem);
    auto piece = [&](const int j, const int slot) {
        if (j < G5_XI) {
            glds16_hidden(xsrc[j], lds0 + slot * (G5_BM * 128) + xdst[j]);
            xsrc[j] += 128;
        } else if (j < G5_L) {
            const int i = j - G5_XI;
            glds16_hidden(wsrc[i], lds0 + G5_S * G5_BM * 128 + slot * (G5_WROWS * 128) + wdst[i]);
            wsrc[i] += 8192;
        } else {
            glds16_hidden(lsrc, lds0 + G5_S * G5_BM * 128 + slot * (G5_WROWS * 128) + G5_BN * 128);
            lsrc += ladv;
        }
    };

    f32x4_t acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragments: lane (g4 = lane / 16, r16 = lane % 16) holds row r16 of a 16-row block, k = 32 * ks + 8 * g4 .. + 7 (16 bytes)
    bf16x8 xf[2][2], wf[2][5], lf[2];
    f32x4_t accl[2];     // LORA: accl[i][e] = T[m = ..i*16 + r16][rank 4*g4 + e]; ranks 0-3 (the lanes with g4 == 0) are live
    accl[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accl[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int lrow = G5_BN + (r16 < 4 ? r16 : 4 + (r16 & 3));      // adapter row of this lane's A-operand row (>= 4: a zero row)
    auto load_frags = [&](const int set, const int slot, const int ks) {
        const char* cX = sX + slot * (G5_BM * 128);
        const char* cW = sW + slot * (G5_WROWS * 128);
#pragma unroll
        for (int i = 0; i < 2; ++i) xf[set][i] = *(const bf16x8*)(cX + lds_off(wm * 32 + i * 16 + r16, ks * 4 + g4));
#pragma unroll
        for (int j = 0; j < 5; ++j) wf[set][j] = *(const bf16x8*)(cW + lds_off(wn * 80 + j * 16 + r16, ks * 4 + g4));
        if (LORA) lf[set] = *(const bf16x8*)(cW + lds_off(lrow, ks * 4 + g4));
    };
    // the 10 MFMAs of one k-step (W rows feed the A operand: a lane ends up with 4 consecutive output columns of one row);
    // with ISSUE the 7 LDS-DMA pieces of the tile being staged are dealt out one behind each of the first MFMAs
    auto mfmas = [&](const int set, const bool with_pieces, const int slot) {
        int m = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][j], xf[set][i], acc[i][j], 0, 0, 0);
                if (with_pieces && m < G5_L) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(m, slot);
                    __builtin_amdgcn_sched_barrier(0);
                }
                ++m;
            }
        }
        if (LORA) {
#pragma unroll
            for (int i = 0; i < 2; ++i) accl[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf[set], xf[set][i], accl[i], 0, 0, 0);
        }
    };

    // prologue: tiles 0 .. S-2 in flight, tile 0 landed, its first fragments requested
    for (int t = 0; t < G5_S - 1; ++t) {
        if (t < nk) {
#pragma unroll
            for (int j = 0; j < G5_L; ++j) piece(j, t);
            if (extra) piece(G5_L, t);
        }
    }
    if (nk >= G5_S - 1) {
        if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G5_S - 2) * (G5_L + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G5_S - 2) * G5_L) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0)
    int cur = 0;
    // K tile g in ring slot cur.  Protocol (per wave): k-step 0 = [fragment reads of (g, 1)] [MFMAs of (g, 0)]; then
    // s_waitcnt vmcnt((S-3) * L): this wave's share of tile g+1 has landed (g+2 stays in flight); s_barrier: every wave's share has,
    // and every wave is past its last read of tile g-1; k-step 1 = [fragment reads of (g+1, 0)] [MFMAs of (g, 1) with the LDS-DMA of
    // tile g+S-1 into the slot of tile g-1 dealt out between them].  MORE / ISSUE / KEEP are compile-time (see gemm.hip).
    auto body = [&](auto more_c, auto issue_c, auto keep_c) {
        // keep_c: K tiles beyond g+1 whose LDS-DMA has been issued and may stay in flight across the barrier (S-3 in the steady state)
        constexpr bool MORE = decltype(more_c)::value, ISSUE = decltype(issue_c)::value;
        constexpr int KEEP = decltype(keep_c)::value;
        const int nxt = cur == G5_S - 1 ? 0 : cur + 1;
        const int prv = cur == 0 ? G5_S - 1 : cur - 1;
        __builtin_amdgcn_sched_barrier(0);
        load_frags(1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, false, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * (G5_L + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * G5_L) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) load_frags(0, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, ISSUE, prv);
        if constexpr (ISSUE) {
            if (extra) piece(G5_L, prv);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);        // nothing pending across the loop edge
        cur = nxt;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int g = 0;
    for (; g + G5_S - 1 < nk; ++g) body(T_{}, T_{}, std::integral_constant<int, G5_S - 3>{});
    // the tail: no tile left to stage; tile g+1 must have landed, the j tiles behind it that exist stay in flight
    if constexpr (G5_S >= 5) {
        if (g + 3 < nk) { body(T_{}, F_{}, std::integral_constant<int, 2>{}); ++g; }
    }
    if (g + 2 < nk) { body(T_{}, F_{}, std::integral_constant<int, 1>{}); ++g; }
    if (g + 1 < nk) { body(T_{}, F_{}, std::integral_constant<int, 0>{}); ++g; }
    if (g < nk) body(F_{}, F_{}, std::integral_constant<int, 0>{});

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    // acc[i][j][e] = C[m = m0 + wm*32 + i*16 + r16][n = n0 + wn*80 + j*16 + 4*g4 + e]
    const int mrow = m0 + wm * 32 + r16;
    const int ncol = n0 + wn * 80 + 4 * g4;
    bf16x4 bq[5], rq[2][5];
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 5; ++j) bq[j] = *(const bf16x4*)(p.bias + ncol + j * 16);
    }
    if (p.residual) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) rq[i][j] = *(const bf16x4*)(p.residual + (long)(mrow + i * 16) * p.ld_res + ncol + j * 16);
    }
    if (p.ln_in) {
        // LN(x) . W^T = rstd (x . W'^T - mean s) + b'   (W' = W gamma, s = row sums of W', b' = bias + W beta).  Equal-sized chunks
        // merged with the chunk means shifted by the first one (gemm_common.h: gemm_ln_finish): every lane its quarter of the chunks,
        // the quarters added across the row's four lanes in a fixed order - the arithmetic of the 128-row tiles (gemm7.hip)
        const float nc = (float)(p.K / p.ln_in_chunks), inv_chunks = 1.f / (float)p.ln_in_chunks;
        float ln_mean[2], ln_rstd[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float m0v = lnp[i][0][0];
            float sm = 0.f, pq = 0.f, q = 0.f;
#pragma unroll
            for (int c = 1; c < G5_LNC; ++c) {
                const int ch = g4 + 4 * (c - 1);
                if (ch < p.ln_in_chunks) { const float dl = lnp[i][c][0] - m0v; sm += dl; pq += dl * dl; q += lnp[i][c][1]; }
            }
            sm += __shfl_xor(sm, 16, 64); pq += __shfl_xor(pq, 16, 64); q += __shfl_xor(q, 16, 64);
            sm += __shfl_xor(sm, 32, 64); pq += __shfl_xor(pq, 32, 64); q += __shfl_xor(q, 32, 64);
            const float dm = sm * inv_chunks;
            ln_mean[i] = m0v + dm;
            const float M2 = q + nc * fmaxf(pq - sm * dm, 0.f);
            ln_rstd[i] = 1.0f / sqrtf(M2 / (float)p.K + p.ln_eps);
            if (p.ln_mr_out && tile_n == 0 && wn == 0 && g4 == 0) *(f32x2*)(p.ln_mr_out + (long)(mrow + i * 16) * 2) = f32x2{ln_mean[i], ln_rstd[i]};
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const f32x4_t s4 = *(const f32x4_t*)(p.ln_s + ncol + j * 16), b4 = *(const f32x4_t*)(p.ln_b + ncol + j * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = ln_rstd[i] * (acc[i][j][e] - ln_mean[i] * s4[e]) + b4[e];
        }
    }
    if (LORA) {
        // up-projection: acc += B[n][0..3] . bf16(scale * T[m][0..3]) - one more MFMA per accumulator block, its 32-deep k axis carrying
        // the rank index (k = 8 * g4 + e <-> rank 4 * g4 + e: T sits in the accumulator layout of a B operand already); the reference's
        // down-projection output is a bf16 tensor too (lora.py:108-112), same rounding as the 32 x 32 tiles' fused form
        const float lscale = *p.lora_scale;
        if (p.lora_t_out && tile_n == 0 && wn == 0 && g4 == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) *(f32x4_t*)(p.lora_t_out + (long)(mrow + i * 16) * p.ld_t) = accl[i];
        }
        bf16x8 tb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) tb[i][e] = (g4 == 0 && e < 4) ? (__bf16)(lscale * accl[i][e]) : (__bf16)0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const bf16x4 u4 = *(const bf16x4*)(p.lora_up + (long)(n0 + wn * 80 + j * 16 + r16) * 4);
            bf16x8 ua;
#pragma unroll
            for (int e = 0; e < 8; ++e) ua[e] = (g4 == 0 && e < 4) ? u4[e & 3] : (__bf16)0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua, tb[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();       // every wave is past its last fragment read (and every LDS-DMA has landed: vmcnt(0) in the last bodies)
    char* sE = smem + wave * (32 * G5_PATCH_LD);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float ln_k = 0.f, ln_s = 0.f, ln_q = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][j][e];
                if (p.bias) v += (float)bq[j][e];
                if (p.residual) v += (float)rq[i][j][e];
                o[e] = (__bf16)v;
            }
            if (p.ln_out) {      // statistics of the stored (rounded) values of this row's 80 columns, shifted by a sample of the row
                if (j == 0) ln_k = __shfl((float)o[0], r16, 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dlt = (float)o[e] - ln_k; ln_s += dlt; ln_q += dlt * dlt; }
            }
            *(bf16x4*)(sE + (i * 16 + r16) * G5_PATCH_LD + j * 32 + g4 * 8) = o;
        }
        if (p.ln_out) {
            ln_s += __shfl_xor(ln_s, 16, 64); ln_q += __shfl_xor(ln_q, 16, 64);
            ln_s += __shfl_xor(ln_s, 32, 64); ln_q += __shfl_xor(ln_q, 32, 64);
            if (g4 == 0) {
                const float dm = ln_s * (1.f / 80.f);
                const int chunk = tile_n * 2 + wn;
                *(f32x2*)(p.ln_out + ((long)chunk * p.M + mrow + i * 16) * 2) = f32x2{ln_k + dm, fmaxf(ln_q - ln_s * dm, 0.f)};
            }
        }
    }
    __builtin_amdgcn_wave_barrier();       // same-wave LDS ops retire in order; only the compiler must not reorder
    const __amdgpu_buffer_rsrc_t crs = wt_rsrc(p.c);
#pragma unroll
    for (int it = 0; it < 5; ++it) {       // 32 rows x 10 sixteen-byte segments = 5 per lane
        const int item = it * 64 + lane;
        const int row = item / 10, seg = item - row * 10;
        const bf16x8 v8 = *(const bf16x8*)(sE + row * G5_PATCH_LD + seg * 16);
        wt_store16(crs, ((long)(m0 + wm * 32 + row) * p.ldc + n0 + wn * 80 + seg * 8) * 2, v8);
    }
}

}  // namespace

// group_m of the grouped tile order: an XCD's run of tiles_m * tiles_n / 8 tiles is gm row tiles x (run / gm) column tiles; pick the
// gm (power of two) that minimises the operand rows it pulls through its L2 (64 * gm of X + 160 * run / gm of W)
static int g5_group_m(int tiles_m, int tiles_n) {
    const int run = (tiles_m * tiles_n + 7) / 8;
    int best = 1;
    long best_cost = -1;
    for (int gm = 1; gm <= tiles_m; gm *= 2) {
        const int gn = (run + gm - 1) / gm;
        const long cost = 64L * gm + 160L * (gn < tiles_n ? gn : tiles_n);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = gm; }
    }
    return best;
}

// called by slh_gemm (gemm.hip) for tile codes whose bits 12-15 are 5; 1 where slh_gemm5 can run the descriptor (the planner asks)
extern "C" int slh_gemm5_ok(const slh_gemm_desc* d) {
    if (!d || !d->a0 || !d->w || !d->c) return 0;
    if (d->mode != 0 || d->a1 || d->ca1 || d->w_layout != 1) return 0;
    if (d->M <= 0 || d->M % 64 || d->N <= 0 || d->N % 160 || d->K < 64 || d->K % 64 || d->ca0 != d->K) return 0;
    if (d->lora_t || d->rowbias || d->geglu || d->vt_out || d->xa_k || d->geglu_pre) return 0;
    if (d->ln_in) {          // consumer side of a folded LayerNorm (as the 128-row tiles: gemm7.hip), without an adapter
        if (!d->ln_s || !d->ln_b || d->bias || d->lora_down || ((uintptr_t)d->ln_in & 7) || ((uintptr_t)d->ln_s & 15) || ((uintptr_t)d->ln_b & 15)) return 0;
        if (d->ln_in_chunks < 1 || d->ln_in_chunks > 20 || !(d->K == 64 * d->ln_in_chunks || d->K == 80 * d->ln_in_chunks)) return 0;
        if (d->ln_mr_out && ((uintptr_t)d->ln_mr_out & 7)) return 0;
    } else if (d->ln_mr_out) {
        return 0;
    }
    if (d->lora_down) {      // one fused rank-4 adapter, forward form
        if (!d->lora_up || !d->lora_scale || d->lora_up_rmajor || d->lora_groups != 1 || d->lora_rank != 4 || d->ln_lora_s) return 0;
        if (((uintptr_t)d->lora_down & 15) || ((uintptr_t)d->lora_up & 7)) return 0;
        if (d->lora_t_out && (d->ld_t < 4 || d->ld_t % 4 || ((uintptr_t)d->lora_t_out & 15))) return 0;
    } else if (d->lora_t_out) {
        return 0;
    }
    if (d->lda0 % 8 || d->ldc % 8 || ((uintptr_t)d->c & 15) || ((uintptr_t)d->a0 & 15) || ((uintptr_t)d->w & 15)) return 0;
    if (d->residual && (d->ld_res % 4 || ((uintptr_t)d->residual & 7))) return 0;
    if (d->bias && ((uintptr_t)d->bias & 7)) return 0;
    if (d->ln_out && ((uintptr_t)d->ln_out & 7)) return 0;
    return 1;
}

int slh_gemm5_launch(const slh_gemm_desc* d, slh_stream_t stream) {
    SLH_CHECK(slh_gemm5_ok(d),
              "slh_gemm: the 64 x 160 tile (0x5xxx) runs dense single-source products with packed weights, M %% 64 == 0, N %% 160 == 0, "
              "bias / residual / ln_out / ln_in / one fused rank-4 adapter only (M=%d N=%d K=%d)", d ? d->M : 0, d ? d->N : 0, d ? d->K : 0);
    SLH_CHECK(((d->tile >> 16) & 15) <= 1, "slh_gemm: the 64 x 160 tile has no split-K");
    G5Args a;
    a.a = (const __bf16*)d->a0; a.w = (const __bf16*)d->w; a.bias = (const __bf16*)d->bias; a.residual = (const __bf16*)d->residual;
    a.c = (__bf16*)d->c; a.ln_out = d->ln_out;
    a.lora_down = (const __bf16*)d->lora_down; a.lora_up = (const __bf16*)d->lora_up; a.lora_scale = d->lora_scale;
    a.lora_t_out = d->lora_t_out; a.ld_t = d->ld_t;
    a.ln_in = d->ln_in; a.ln_s = d->ln_s; a.ln_b = d->ln_b; a.ln_mr_out = d->ln_mr_out; a.ln_in_chunks = d->ln_in_chunks; a.ln_eps = d->ln_eps;
    a.lda = d->lda0; a.ldc = d->ldc; a.ld_res = d->ld_res; a.M = d->M; a.N = d->N; a.K = d->K;
    a.tiles_m = d->M / G5_BM; a.tiles_n = d->N / G5_BN;
    a.group_m = g5_group_m(a.tiles_m, a.tiles_n);
    static const int skew_knob = getenv("SLH_G5_SKEW") ? atoi(getenv("SLH_G5_SKEW")) : 1;      // A/B: 0 = every sibling walks its panels in the same order
    a.skew = skew_knob;
    const int grid = a.tiles_m * a.tiles_n;
    const hipStream_t st = (hipStream_t)stream;
    if (((d->tile >> 8) & 15) == 5) {      // 5 ring slots
        if (d->lora_down) slh_launch<gemm5_kernel<true, 5>>(grid, 256, st, a, "gemm5_kernel<true, 5>");
        else slh_launch<gemm5_kernel<false, 5>>(grid, 256, st, a, "gemm5_kernel<false, 5>");
    } else {
        if (d->lora_down) slh_launch<gemm5_kernel<true, 4>>(grid, 256, st, a, "gemm5_kernel<true, 4>");
        else slh_launch<gemm5_kernel<false, 4>>(grid, 256, st, a, "gemm5_kernel<false, 4>");
    }
    SLH_LAUNCH_CHECK("slh_gemm (64 x 160 tile)");
    return 0;
}
