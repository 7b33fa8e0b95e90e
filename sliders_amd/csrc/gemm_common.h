// Shared device code of the bf16 MFMA GEMM kernels (gemm.hip: 2-slot / ring K loops; gemm8p.hip: 256 x 256 ping-pong K loop):
// argument block, LDS swizzle, XCD-aware tile mapping, the folded-LayerNorm prologue and the fused epilogue.  gfx950 only.
#pragma once
#include "common.h"
#include <cstdint>
#include <type_traits>
#include "../../include/sliders_hip.h"

namespace slh_gemm_detail {

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
    int kper;                     // split-K: K tiles per slice (the last slice may be shorter, never empty)
    int splitk_local;             // split-K: the last slice may read same-XCD partials through this XCD's L2 (see the epilogue)
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
    // cross-attention behind the query projection (slh_gemm_desc.xa_*)
    const __bf16* xa_k; const __bf16* xa_vt; int xa_tk, xa_tq, xa_ldk, xa_ldvt, xa_vt_heads; float xa_scale;
    const float* ln_lora_s; const float* ln_lora_c;   // LayerNorm fold of the fused adapter's down-projection (slh_gemm_desc.ln_lora_*)
    const void* pf_ptr; long pf_bytes; int pf_blocks;   // weight touch by the launch's last pf_blocks workgroups (slh_gemm_desc.pf_*)
};

constexpr int BK = 64;

// swizzled byte offset of (row, 16-byte slot) inside a [rows][64] bf16 LDS tile
__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

// ---- XCD-aware tile mapping: block b runs on XCD b%8 (own 4 MB L2); give each XCD a contiguous chunk of a
// grouped tile sequence: groups of group_m consecutive m-tiles, inside a group m fastest, then n.  The host
// sizes the groups so that one group's X rows stay resident in the XCD's L2 while the W panels stream past
// once (measured before grouping: L2 hit rate 62-78 %, fabric fetches 6-14x the unique operand bytes).
// split-K: consecutive block ids = the K slices of one tile (same XCD, same L2).
// block id -> position in the XCD-grouped launch sequence (XCD x owns a contiguous chunk of it)
__device__ __forceinline__ int gemm_remap_bid(const int nblk) {      // nblk: workgroups that compute tiles (the grid minus pf_blocks)
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// position in the grouped tile sequence -> (tile_m, tile_n)
__device__ __forceinline__ void gemm_tile_of(const GemmArgs& p, const int t, int& tile_m, int& tile_n) {
    const int gsz = p.group_m * p.tiles_n;
    const int g = t / gsz;
    const int first_m = g * p.group_m;
    const int gm = min(p.group_m, p.tiles_m - first_m);
    const int r = t - g * gsz;
    tile_n = r / gm;
    tile_m = first_m + r - tile_n * gm;
}

__device__ __forceinline__ void gemm_map_tile(const GemmArgs& p, int& tile_m, int& tile_n, int& ks_id) {
    int bid = gemm_remap_bid((int)gridDim.x - p.pf_blocks);
    ks_id = 0;
    if (p.splitk > 1) { ks_id = bid % p.splitk; bid = bid / p.splitk; }
    gemm_tile_of(p, bid, tile_m, tile_n);
}

// ---- weight touch (slh_gemm_desc.pf_*): the launch's LAST pf_blocks workgroups - dispatched behind every tile workgroup, i.e. onto
// CUs the grid leaves idle - stream a byte range a later launch will need through plain loads (they allocate in the memory-side
// cache) and exit.  Returns true in those workgroups.
__device__ __forceinline__ bool gemm_weight_touch(const GemmArgs& p) {
    const int first = (int)gridDim.x - p.pf_blocks;
    if (p.pf_blocks <= 0 || (int)blockIdx.x < first) return false;
    weight_touch(p.pf_ptr, p.pf_bytes, (int)blockIdx.x - first, p.pf_blocks);      // common.h
    return true;
}

// ---- LayerNorm of the A operand, folded (consumer side) ---------------------------------------------------------------------
// The producer of A left per-row (mean, M2) pairs of 64-column chunks (ln_in, laid out [chunk][row], written by its epilogue
// below); they are requested ahead of the first operand tiles (gemm_ln_request), merged in a fixed order (Chan's merge for equal
// counts: cancellation-free) into the row's mean and 1/sigma (gemm_ln_finish) and applied in the epilogue:
//   LN(x) . W^T = rstd * (x . W'^T - mean * s) + b',   W' = W * gamma, s = row sums of W', b' = bias + W . beta
constexpr int LN_MAXC = 20;

template <int MI>
__device__ __forceinline__ void gemm_ln_request(const GemmArgs& p, int row0, int lrow, f32x2 (&pairs)[MI][LN_MAXC]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int m = row0 + i * 32 + lrow;
        m = m < p.M ? m : p.M - 1;
        const f32x2* src = (const f32x2*)p.ln_in + m;        // chunk-major [chunks][M]: the 32 rows of a wave-load are contiguous
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c)
            if (c < p.ln_in_chunks) pairs[i][c] = src[(long)c * p.M];
    }
}

// The same requests issued from inline asm, i.e. invisible to hipcc's waitcnt bookkeeping (ping-pong kernels: the loads go out ahead
// of the prologue's LDS-DMA - older in the in-order vmcnt queue, retired by the prologue's own counted wait - and hipcc, which
// does not see the DMA, would otherwise drain everything with vmcnt(0) in front of the first use).  Unconditional loads at a
// clamped chunk index.  The caller waits (counted vmcnt covering these loads) and then calls gemm_ln_landed.
template <int MI>
__device__ __forceinline__ void gemm_ln_request_hidden(const GemmArgs& p, int row0, int lrow, f32x2 (&pairs)[MI][LN_MAXC]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int m = row0 + i * 32 + lrow;
        m = m < p.M ? m : p.M - 1;
        const f32x2* src = (const f32x2*)p.ln_in + m;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const f32x2* q = src + (long)(c < p.ln_in_chunks ? c : 0) * p.M;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(pairs[i][c]) : "v"(q) : "memory");
        }
    }
}
template <int MI>
__device__ __forceinline__ void gemm_ln_landed(f32x2 (&pairs)[MI][LN_MAXC]) {      // pins the first uses behind the caller's wait
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) asm volatile("" : "+v"(pairs[i][c]));
}

// row0: first row of the wave's sub-tile; writer: this lane also stores the merged pairs (ln_mr_out)
template <int MI>
__device__ __forceinline__ void gemm_ln_finish(const GemmArgs& p, int row0, int lrow, bool writer, const f32x2 (&pairs)[MI][LN_MAXC],
                                               float (&ln_mean)[MI], float (&ln_rstd)[MI]) {
    // equal-sized chunks, one pass over the pairs with the chunk means shifted by the first one (d_c = mean_c - mean_0):
    //   mean = mean_0 + S/k,  M2 = sum M2_c + n_chunk * (sum d_c^2 - S^2/k),  S = sum d_c
    // = Chan's merge for equal counts; two interleaved accumulator sets keep the dependent chains short; fixed order
    const float nc = (float)(p.K / p.ln_in_chunks);
    const float inv_chunks = 1.f / (float)p.ln_in_chunks;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const float m0v = pairs[i][0][0];
        float sa = 0.f, sb = 0.f, pa = 0.f, pb = 0.f, qa = pairs[i][0][1], qb = 0.f;
#pragma unroll
        for (int c = 1; c < LN_MAXC; c += 2) {
            if (c < p.ln_in_chunks) { const float dl = pairs[i][c][0] - m0v; sa += dl; pa += dl * dl; qa += pairs[i][c][1]; }
            if (c + 1 < p.ln_in_chunks && c + 1 < LN_MAXC) {
                const float dl = pairs[i][c + 1 < LN_MAXC ? c + 1 : c][0] - m0v;
                sb += dl; pb += dl * dl; qb += pairs[i][c + 1 < LN_MAXC ? c + 1 : c][1];
            }
        }
        const float S = sa + sb;
        const float dm = S * inv_chunks;
        ln_mean[i] = m0v + dm;
        const float M2 = (qa + qb) + nc * fmaxf((pa + pb) - S * dm, 0.f);
        ln_rstd[i] = 1.0f / sqrtf(M2 / (float)p.K + p.ln_eps);
        if (p.ln_mr_out && writer) {
            const int m = row0 + i * 32 + lrow;
            if (m < p.M) *(f32x2*)(p.ln_mr_out + (long)m * 2) = f32x2{ln_mean[i], ln_rstd[i]};
        }
    }
}

// LDS bytes the epilogue needs (store patches + adapter exchange + column vectors); the kernels assert their operand stages cover it
constexpr int gemm_epilogue_lds(int MI, int NI, int NW, int WN, bool LORA) {
    return NW * (2048 * NI + (LORA ? 2048 * MI : 0)) + 4 * (32 * NI * WN) * 4;
}

// ---- epilogue -----------------------------------------------------------------------------------------------------------------
// Shared by every K loop.  The workgroup's NW waves form a (NW / WN) x WN grid; wave (wm, wn) owns the MI x NI blocks of 32 x 32
//   acc[i][j][r] = C[m = m0 + wm*32*MI + i*32 + lrow][n = n0 + wn*32*NI + j*32 + (r&3) + 8*(r>>2) + 4*lhi]
// (MFMA operand roles swapped: W rows feed the A operand, activation rows the B operand).  Must be entered by all waves of the
// workgroup with the operand stages in `smem` no longer in use by the K loop's LDS-DMA (they are recycled as staging patches).
// FEAT: optional forms a K loop's tiles can take - 1 the head-transposed V store (vt_out), 2 the 32 | 32 GEGLU forms (geglu = 1, 2),
// 4 the LayerNorm chunk statistics (ln_out), 8 cross-attention behind the query projection (xa_k; MI = 1, NI = 2 only), 16 the folded
// LayerNorm (ln_in) together with a fused adapter; compiled out
// where slh_gemm never routes them (registers and code of the odd-NI tiles)
template <int MI, int NI, int MODE, bool LORA, int NW, int WN, int FEAT = 7>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, char* smem, f32x16 (&acc)[MI][NI], f32x16 (&accl)[MI],
                                              const float (&ln_mean)[MI], const float (&ln_rstd)[MI], const bool ln_on,
                                              const int tile_m, const int tile_n, const int ks_id, const int wave, const int wm,
                                              const int wn) {
    constexpr int BM = 32 * MI * (NW / WN);
    constexpr int BN = 32 * NI * WN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int lrow = lane & 31, lhi = lane >> 5;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

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
        // (The L2-local read is a measured shortcut of THIS chip, outside the memory model's guarantees: it relies on the slab
        // lines never being resident in the reader's L1 - L1 is invalidated per dispatch and every slab line is read once - and on
        // HW_REG_XCC_ID naming the L2 a workgroup's write-through stores passed.  slh_gemm enables it on gfx950 only;
        // SLIDERS_SPLITK_LOCAL=0 switches to the agent-scope loads everywhere; tests/test_kernels_gpu.py stresses both.)
        const bool local = p.splitk_local && (int)((seen >> (8 + 7 * xcc)) & 127) == p.splitk;     // uniform over the workgroup
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
    // swizzle of the store staging patch: physical 16-byte slot of (row, logical slot), and whether the row's 8-byte halves are swapped;
    // XOR forms for power-of-two S (NI = 1, 2, 4), a rotation for the others (NI = 5: S = 20)
    constexpr bool SPOW2 = (S & (S - 1)) == 0;
    constexpr int LOG2S = S == 4 ? 2 : (S == 8 ? 3 : (S == 16 ? 4 : 0));
    auto patch_slot = [](const int row, const int slot) { return SPOW2 ? (slot ^ (row & (S - 1))) : (slot + row) % S; };
    auto patch_hb = [](const int row) { return SPOW2 ? ((row >> LOG2S) & 1) : ((row / S) & 1); };
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
    if ((FEAT & 2) && p.geglu == 2) {
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
    if (p.geglu == 3) {
        // GEGLU, 32-row weight blocks [16 value rows | 16 gate rows] (slh_gemm_desc.geglu = 3; any NI): inside one 32 x 32
        // accumulator block the quads q = 0, 1 are values and q = 2, 3 the gates of the same 16 output columns, in the same lane.
        // The results leave like those of the ordinary path: through a wave-private staging patch as whole 16-byte row segments,
        // written through (round 5; the 8-byte quad per lane and row - 32 cache lines per store instruction - measured +2..3 us per
        // launch of GEGLU.proj and cannot be written through: common.h).  SLH_GEGLU_STORE8 (A/B builds) keeps the quads.
        __syncthreads();
        constexpr int SG = 2 * NI;               // 16-byte slots of a staged output row: 16 output columns per 32 x 32 block
        constexpr int GROW = SG * 16 + 16;       // patch row stride; + 16: the rows of one quad write fall on distinct banks
        static_assert(32 * GROW <= 32 * S * 16, "the GEGLU staging patch lives inside the wave's store patch");
        char* sG = smem + wave * (32 * S * 16);
        const __amdgpu_buffer_rsrc_t g_rsrc = wt_rsrc(p.c);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mbase = m0 + wm * (32 * MI) + i * 32;
            const int m = mbase + lrow;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nb = n0 + wn * (32 * NI) + j * 32;               // first weight row of the block
                if (nb >= p.N) continue;                                    // N % 32 == 0: blocks are whole (wave-uniform)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cl = nb - n0 + q * 8 + lhi * 4;               // tile column of the value quad (gate: + 16)
                    float a[4], g[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = acc[i][j][q * 4 + e]; g[e] = acc[i][j][(q + 2) * 4 + e]; }
                    if (MODE == 0 && ln_on) {
                        const f32x4 sa = *(const f32x4*)(sCol + 2 * BN + cl), sg = *(const f32x4*)(sCol + 2 * BN + cl + 16);
                        const f32x4 ba = *(const f32x4*)(sCol + 3 * BN + cl), bg = *(const f32x4*)(sCol + 3 * BN + cl + 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = ln_rstd[i] * (a[e] - ln_mean[i] * sa[e]) + ba[e];
                            g[e] = ln_rstd[i] * (g[e] - ln_mean[i] * sg[e]) + bg[e];
                        }
                    }
                    if (p.bias) {
                        const f32x4 ba = *(const f32x4*)(sCol + cl), bg = *(const f32x4*)(sCol + cl + 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] += ba[e]; g[e] += bg[e]; }
                    }
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = round_bf16(a[e]), gv = round_bf16(g[e]);   // the reference rounds proj(x) to bf16 before chunk / gelu
                        o[e] = (__bf16)(av * round_bf16(gelu_erf_fast_f(gv)));
                    }
                    *(bf16x4*)(sG + lrow * GROW + j * 32 + q * 16 + lhi * 8) = o;
                }
            }
            __builtin_amdgcn_wave_barrier();       // same-wave LDS ops retire in order; only the compiler must not reorder
#pragma unroll
            for (int it = 0; it < SG / 2; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / SG, slot = idx - row * SG;
                const bf16x8 t8 = *(const bf16x8*)(sG + row * GROW + slot * 16);
                const int m2 = mbase + row;
                const int nblk = n0 + wn * (32 * NI) + (slot >> 1) * 32;               // the weight-row block this slot came from
                const long off = (long)m2 * p.ldc + ((n0 + wn * (32 * NI)) >> 1) + slot * 8;
                if (m2 < p.M && nblk < p.N) {
                    if (p.store16) {
                        wt_store16(g_rsrc, off * 2, t8);
                    } else {
                        *(bf16x4*)(p.c + off) = __builtin_shufflevector(t8, t8, 0, 1, 2, 3);
                        *(bf16x4*)(p.c + off + 4) = __builtin_shufflevector(t8, t8, 4, 5, 6, 7);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if ((FEAT & 2) && p.geglu) {
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
                        o[e] = (__bf16)(av * round_bf16(gelu_erf_fast_f(gv)));
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
    bool ln_done = false;
    if constexpr ((FEAT & 16) != 0 && LORA && MODE == 0) {
        if (ln_on) {
            // Linear(LayerNorm(x)) + LoRA(LayerNorm(x)) from the raw rows: the main product is normalised HERE, ahead of the
            // up-projection MFMA that adds the adapter term into the same accumulators, and the down-projection T (lora_down
            // holds A . gamma) by the same algebra with the adapter's own row sums / offsets:
            //   y = rstd (x W'^T - mean s) + b' + scale * B . bf16(rstd (x A'^T - mean sA) + cA)
            ln_done = true;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = wn * (32 * NI) + j * 32 + q * 8 + lhi * 4;
                        const f32x4 s4 = *(const f32x4*)(sCol + 2 * BN + cl), b4 = *(const f32x4*)(sCol + 3 * BN + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][q * 4 + e] = ln_rstd[i] * (acc[i][j][q * 4 + e] - ln_mean[i] * s4[e]) + b4[e];
                    }
                // rank index of accl[i][e]: e < 4 -> e + 4 lhi, e >= 4 -> 8 + (e - 4) in the lhi = 0 half (unused in the other)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int rank = e < 4 ? e + 4 * lhi : 4 + e;
                    const bool live = rank < p.lora_rank && (e < 4 || lhi == 0);
                    const float sa = p.ln_lora_s[live ? rank : 0], ca = p.ln_lora_c[live ? rank : 0];
                    accl[i][e] = live ? ln_rstd[i] * (accl[i][e] - ln_mean[i] * sa) + ca : accl[i][e];
                }
            }
        }
    }
    bool xa = false;
    if constexpr ((FEAT & 8) != 0 && MI == 1 && NI == 2 && !LORA) {
        if (p.xa_k != nullptr) {
            // ---- cross-attention of this wave's 32 rows x one head, in registers (slh_gemm_desc.xa_*) ---------------------------
            // The accumulators are Q^T of the head in the C layout: lane (lrow, lhi) holds, for query row lrow, the head dims
            //   dim(j, r) = 32 j + 8 (r >> 2) + 4 lhi + (r & 3).
            // A sum over dims does not care in which order the 16 dims of a k-step sit in the operand registers as long as both
            // operands agree, so registers 8 (s & 1) .. + 7 of block j = s >> 1 ARE the B operand of k-step s (dims 16 s + 8 (e >> 2)
            // + 4 lhi + (e & 3), e = 0..7) and the key rows of the A operand are gathered in the same order (two 8-byte pieces
            // per k-step).  The scores come out transposed (S^T = K Q^T: lane = query, registers = keys), the softmax is a
            // reduction over registers + one exchange with lane ^ 32, and the probabilities feed the second product (O^T = V^T
            // P^T) by the same trick; O^T lands in the accumulator layout Q^T came in, so the ordinary store path below writes it.
            xa = true;
            const int mrow0 = m0 + wm * 32;
            const int b = min(mrow0, p.M - 1) / p.xa_tq;
            const int head = (n0 + wn * 64) >> 6;
            const int tk = p.xa_tk;
            // K and V^T of the workgroup's two heads go through LDS (LDS-DMA, 8 lanes per 128-byte row, the GEMM's slot swizzle):
            // gathered straight from memory - every lane its own key row, 8 bytes at a time, rows 332 KB apart in the batched K / V
            // matrix - the two small products cost as much as the separate attention launch they replace (measured: pass time
            // unchanged).  Behind the store patches and column vectors: per head [96 keys][128 B] + 2 x [64 dims][128 B] (keys
            // 0-63 | 64-127).
            constexpr int XA_BASE = (NW * (32 * S * 16) + 4 * BN * 4 + 1023) & ~1023, XA_K = 96 * 128, XA_HEAD = XA_K + 2 * 8192;
            {
                const int frow = lane >> 3, fslot = lane & 7;
                const unsigned xs0 = lds_addr_of(smem) + XA_BASE;
                const int head0 = n0 >> 6;
#pragma unroll
                for (int it = 0; it < 7; ++it) {
                    const int idx = wave + NW * it;                 // 56 groups of 8 rows: [head][12 K groups | 16 V groups]
                    const int hh = idx >= 28 ? 1 : 0, g = idx - 28 * hh;
                    const int hd = min(head0 + hh, p.N / 64 - 1);
                    const __bf16* src;
                    unsigned dst = xs0 + hh * XA_HEAD;
                    if (g < 12) {
                        const int row = g * 8 + frow;
                        const int key = row < tk ? row : tk - 1;
                        src = p.xa_k + ((long)b * tk + key) * p.xa_ldk + hd * 64 + ((fslot ^ ((row >> 1) & 7)) << 3);
                        dst += g * 1024;
                    } else {
                        const int t = (g - 12) >> 3, row = ((g - 12) & 7) * 8 + frow;
                        src = p.xa_vt + (((long)b * p.xa_vt_heads + hd) * 64 + row) * p.xa_ldvt + t * 64 + ((fslot ^ ((row >> 1) & 7)) << 3);
                        dst += XA_K + (g - 12) * 1024;
                    }
                    glds16_hidden(src, dst);
                }
            }
            // Q: LayerNorm fold / bias applied, rounded to bf16 (the unfused path stores it as a bf16 tensor)
            bf16x8 qf[4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wn * 64 + j * 32 + q * 8 + lhi * 4;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[0][j][q * 4 + e];
                    if (MODE == 0 && ln_on) {
                        const f32x4 s4 = *(const f32x4*)(sCol + 2 * BN + cl), b4 = *(const f32x4*)(sCol + 3 * BN + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ln_rstd[0] * (v[e] - ln_mean[0] * s4[e]) + b4[e];
                    }
                    if (p.bias) {
                        const f32x4 b4 = *(const f32x4*)(sCol + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += b4[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) qf[2 * j + (q >> 1)][(q & 1) * 4 + e] = (__bf16)v[e];
                }
            const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            lds_dma_syncthreads();                 // the heads' K / V^T have landed (every wave's share)
            const char* xk = smem + XA_BASE + wn * XA_HEAD;
            const char* xv = xk + XA_K;
            f32x16 sc[3];
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                sc[kb] = kZero16;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    // dims 16 s + 4 lhi + {0..3} and 16 s + 8 + 4 lhi + {0..3} of key row 32 kb + lrow
                    const bf16x4 k0 = *(const bf16x4*)(xk + lds_off(kb * 32 + lrow, 2 * s_) + 8 * lhi);
                    const bf16x4 k1 = *(const bf16x4*)(xk + lds_off(kb * 32 + lrow, 2 * s_ + 1) + 8 * lhi);
                    const bf16x8 kf = __builtin_shufflevector(k0, k1, 0, 1, 2, 3, 4, 5, 6, 7);
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s_], sc[kb], 0, 0, 0);
                }
            }
            // sc[kb][r] = S[query lrow][key 32 kb + 8 (r >> 2) + 4 lhi + (r & 3)]
            const float cs = p.xa_scale * 1.4426950408889634f;
            float mx = -1e30f;
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
                    if (key >= tk) sc[kb][r] = -1e30f;
                    mx = fmaxf(mx, sc[kb][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mc = mx * cs;
            float ps = 0.f;
            bf16x8 pb[6];
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r], cs, -mc));
                    ps += pv;
                    pb[2 * kb + (r >> 3)][r & 7] = (__bf16)pv;
                }
            const float inv = 1.f / (ps + __shfl_xor(ps, 32, 64));
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                f32x16 o = kZero16;
#pragma unroll
                for (int s2 = 0; s2 < 6; ++s2) {
                    // keys 16 s2 + 4 lhi + {0..3} and + 8 of dim row 32 db + lrow; key tile s2 >> 2, slots 2 (s2 & 3), 2 (s2 & 3) + 1
                    const char* vt_ = xv + (s2 >> 2) * 8192 + 8 * lhi;
                    const bf16x4 v0 = *(const bf16x4*)(vt_ + lds_off(db * 32 + lrow, 2 * (s2 & 3)));
                    const bf16x4 v1 = *(const bf16x4*)(vt_ + lds_off(db * 32 + lrow, 2 * (s2 & 3) + 1));
                    const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s2], o, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][db][r] = o[r] * inv;
            }
        }
    }
    const bool epi_ln = ln_on && !xa && !ln_done, epi_bias = p.bias != nullptr && !xa;     // (consumed above where xa / ln_done)
    const __amdgpu_buffer_rsrc_t c_rsrc = wt_rsrc(p.c);
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
    const bool to_vt = (FEAT & 1) && p.vt != nullptr && ncol0 >= p.vt_col0;
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
        const int hb = patch_hb(lrow);
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
                    if (MODE == 0 && epi_ln) {
                        const f32x4 s4 = *(const f32x4*)(sCol + 2 * BN + cl), b4 = *(const f32x4*)(sCol + 3 * BN + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ln_rstd[i] * (v[e] - ln_mean[i] * s4[e]) + b4[e];
                    }
                    if (epi_bias) {
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
                if ((FEAT & 4) && p.ln_out) {   // statistics of the stored (rounded) values, shifted by the lane's first one
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
                const int slot = patch_slot(lrow, j * 4 + q);
                *(bf16x4*)(sE + lrow * (S * 16) + slot * 16 + ((lhi ^ hb) << 3)) = o;
            }
        }
        if ((FEAT & 4) && NI == 2 && p.ln_out) {
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
                    const int slot = patch_slot(lrow, j * 4 + q);
                    *(bf16x4*)(sE + lrow * (S * 16) + slot * 16 + ((lhi ^ hb) << 3)) = okeep[j][q];
                }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int it = 0; it < S / 2; ++it) {
            const int idx = it * 64 + lane;
            const int row = idx / S, slot = idx % S;
            bf16x8 t8 = *(const bf16x8*)(sE + row * (S * 16) + (patch_slot(row, slot) << 4));
            if (patch_hb(row)) t8 = __builtin_shufflevector(t8, t8, 4, 5, 6, 7, 0, 1, 2, 3);
            const int m2 = mbase + row, n2 = ncol0 + slot * 8;
            if (m2 < p.M && n2 < p.N) {
                __bf16* dst = p.c + (long)m2 * p.ldc + n2;
                if (n2 + 8 <= p.N) {
                    if (p.store16) {
                        if constexpr (SLH_WT_MASK & 1) wt_store16(c_rsrc, ((long)m2 * p.ldc + n2) * 2, t8);   // write-through (common.h)
                        else *(bf16x8*)dst = t8;
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

// gemm8p.hip: the 256 x 256 ping-pong K loop (slh_gemm_desc.tile code 0x8xxx)
int launch_gemm8p(const GemmArgs& a, int mode, int ni_b, hipStream_t s);   // ni_b: 0 = 256 x 256, 3..5 = 128 x 64*ni_b

}  // namespace slh_gemm_detail
