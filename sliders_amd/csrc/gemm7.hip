// 128 x (32 WB) tiles of the bf16 MFMA GEMM on FOUR compute waves with big register tiles, and a 256 x 320 tile on eight (round 6; tile
// codes 0x7<S><XB><WB>: 0x7648, 0x7645; 0x748a).
//
// Why another K loop.  scripts/ubench_l2fill.hip (profiles/r06_l2_fill_ubench.txt): with nothing else going on a CU pulls 56-65 B/clk of
// L2-resident data into LDS by LDS-DMA - the rate of its address path (one 1 KB wave-instruction per 16 clocks) - while the GEMM K loops
// of rounds 2-5 sustain 26-35.  A K tile costs three things that each occupy the CU for about as long: the address path (bytes staged
// / 64 per clock), the LDS array (bytes staged / 128 + fragment bytes read / 256) and the matrix pipe (16 clocks per 16 x 16 x 32
// MFMA per SIMD), and a loop in which they overlap imperfectly lands near the sum of two of them.  The 64 x 160 tile (gemm5.hip) stages
// 29 KB and reads 56 KB of fragments for 340 MFMA clocks per K tile: ~450 / ~450 / 340 - measured 830.  The vendor library's kernels for
// the shapes it wins (2048 x 3840 x 1280: 27 us against our 37) are 160 x 256 macro tiles on four waves of 80 x 128.  This file is
// that shape of kernel in the construction of gemm5.hip: ONE wave per SIMD, each wave a (16 XB) x (16 WB) register tile of 16 x 16 x 32
// MFMAs (XB = 4, WB = 8: 128 accumulator registers, 12 fragments feed 32 MFMAs = 0.375 KB of LDS reads per MFMA against 0.7 / 1.5 on
// the 64 x 160 / 128 x 128 tiles; a 128 x 256 tile stages 48 KB per 64-deep K step for 1024 MFMA clocks: 750 / 750 / 1024).
//
// Ring of HALF K tiles.  Three 64-deep K tiles of such a tile are all the LDS holds, and a 3-slot ring has to drain its DMA queue at every
// barrier (round 5's removed 256 x 160 experiment).  The ring unit here is a 32-deep half tile - rows of 64 bytes - so S = 5 / 6 slots
// of 24-28 KB fit and 2-3 half tiles stay in flight across every barrier (counted vmcnt, never 0 in the loop).  One LDS-DMA
// instruction fills 16 rows x 64 B; lane i fetches (row i / 4, logical 16-byte slot (i % 4) ^ f(row / 4 % 4)), f = {0, 2, 3, 1}: with that
// XOR every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS table) touches 16 distinct 16-byte bank groups.  The packed weights
// (64-row x 64-k blocks, rows of 128 B with the 16-byte slots XOR-swizzled by (row >> 1) & 7) are read as they are: the two halves of a
// row sit 64 B apart, which side first depends on bit 2 of the row's key, so a lane alternates +-64 / 8192 -+ 64 byte steps.
//
// Per half tile g (ring slot cur), per wave:  [first half of the MFMAs of g]  s_waitcnt vmcnt(KEEP * pieces): my share of g+1 has
// landed  s_barrier: everyone's has, and everyone is past its last read of g-1  [fragment reads of g+1]  [second half of the MFMAs of g
// with the LDS-DMA pieces of half tile g+S-1 - into the slot of g-1 - dealt out one behind each]  s_waitcnt lgkmcnt(0).
//
// Epilogue forms: bias, residual, producer side of a folded LayerNorm (64- or 80-column chunks), consumer side (ln_in, also with the fused
// adapter's own fold), fused rank-4..12 adapter of up to 3 column groups (lora.py:108-112; the down matrix rides as 16 extra W rows, each
// wave of a row pair accumulates T for half of its row blocks), the V third of a fused q|k|v projection head-transposed (vt_out), GEGLU
// in 16 | 16 weight blocks (geglu = 3).  Results leave as 16-byte write-through row segments through per-wave LDS patches.
// Replaces F.linear inside diffusers' Attention (to_q | to_k | to_v) / FeedForward (GEGLU.proj) as called from
// trainscripts/textsliders/train_util.py:242-247.
#include "gemm_common.h"

using namespace slh_gemm_detail;

namespace {

struct G7Args {
    const __bf16* a; const __bf16* w; const __bf16* bias; const __bf16* residual; __bf16* c;
    float* ln_out; const float* ln_in; const float* ln_s; const float* ln_b; const float* ln_lora_s; const float* ln_lora_c; float* ln_mr_out;
    const __bf16* lora_down; const __bf16* lora_up; const float* lora_scale; float* lora_t_out;
    __bf16* vt;
    int lda, ldc, ld_res, M, N, K, tiles_m, tiles_n, group_m, ld_t;
    int ln_in_chunks; float ln_eps;
    int lora_rank, lora_cols_per_group;
    int vt_col0, vt_D, vt_heads, vt_tokens, vt_ld, vt_also_c;
    int geglu;
};

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// XOR key of a row's four 16-byte slots inside its 64-byte LDS row: q = (row >> 2) & 3 -> {0, 2, 3, 1}
__device__ __forceinline__ int g7_key(const int q) { return (0x78 >> (2 * q)) & 3; }

constexpr int G7_LNC = 6;      // chunk pairs a lane requests per row: chunk 0 (the shift) + its quarter of up to 20 chunks

template <int XB, int WB, int S, bool LORA>
__global__ __launch_bounds__(512) void gemm7_kernel(const G7Args p) {
    static_assert(XB % 2 == 0 && WB >= 4 && S >= 4, "tile");
    constexpr int BM = 32 * XB, BN = 32 * WB;
    constexpr int WROWS = BN + (LORA ? 16 : 0);
    constexpr int XBYTES = BM * 64, SLOT = (BM + WROWS) * 64;
    constexpr int NXP = (2 * XB) / 4;          // X pieces per wave and half tile
    constexpr int NWP = (2 * WB) / 4;          // W pieces every wave issues
    constexpr int WREM = (2 * WB) % 4;         // waves < WREM issue one more
    constexpr int L = NXP + NWP;
    static_assert(S * SLOT <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[S * SLOT];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    {
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
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K >> 5;                   // half tiles

    const int r16 = lane & 15, g4 = lane >> 4;
    const bool ln_on = p.ln_in != nullptr;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    if (wave >= 4) {
        // ================= loader waves: nothing but the LDS-DMA stream of the ring =================================================
        // One LDS-DMA instruction = 16 rows x 64 B; lane -> (row = lane / 4, physical slot = lane % 4).  Loader lw = wave - 4 takes the
        // pieces lw, lw + 4, ... of every half tile: NXP of the X rows, NWP (+ 1 for lw < WREM) of the W rows, loader 3 the adapter's 16 rows.
        const int lw = wave - 4;
        const int frow = lane >> 2, fps = lane & 3;
        const int fsl = fps ^ g7_key(frow >> 2);   // the logical 16-byte slot (k = 8 * fsl .. + 7 of the half tile) this lane fetches
        const char* xsrc[NXP];
        const char* wsrc[NWP + 1];
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int row = (lw + 4 * i) * 16 + frow;
            xsrc[i] = (const char*)(p.a + (long)(m0 + row) * p.lda + (fsl << 3));
        }
#pragma unroll
        for (int i = 0; i < NWP + 1; ++i) {
            int pc = lw + 4 * i;
            pc = pc < 2 * WB ? pc : 2 * WB - 1;      // (the slot past the end belongs to loaders < WREM only; the others never issue it)
            const int n = n0 + pc * 16 + frow;
            wsrc[i] = (const char*)(p.w + ((long)(n >> 6) * (p.K >> 6)) * 4096 + ((n & 63) << 6) + ((fsl ^ ((n >> 1) & 7)) << 3));
        }
        // second half of a packed 128-byte row: 64 bytes up or down, by bit 2 of the row's key = bit 3 of the row (tiles start at
        // multiples of 16); from the second half to the first half of the next 64-deep block: 8192 minus that
        const int wd = (frow & 8) ? -64 : 64;
        const bool extra = (lw < WREM) || (LORA && lw == 3);      // wave-uniform: one piece more per half tile
        const char* lsrc = (const char*)slh_zero_page;
        int ladv = 0;
        if (LORA && frow < p.lora_rank) {
            lsrc = (const char*)(p.lora_down + (long)frow * p.K + (fsl << 3));
            ladv = 64;
        }
        const unsigned lds0 = lds_addr_of(smem);
        auto stage = [&](const int t) {              // all of this loader's pieces of half tile t, into ring slot t % S
            const unsigned base = lds0 + (t % S) * SLOT;
            const int wstep = (t & 1) ? 8192 - wd : wd;
#pragma unroll
            for (int j = 0; j < NXP; ++j) {
                glds16_hidden(xsrc[j], base + (lw + 4 * j) * 1024);
                xsrc[j] += 64;
            }
#pragma unroll
            for (int i = 0; i < NWP; ++i) {
                glds16_hidden(wsrc[i], base + XBYTES + (lw + 4 * i) * 1024);
                wsrc[i] += wstep;
            }
            if (LORA && lw == 3) {
                glds16_hidden(lsrc, base + XBYTES + BN * 64);
                lsrc += ladv;
            } else if (extra) {
                glds16_hidden(wsrc[NWP], base + XBYTES + (lw + 4 * NWP) * 1024);
                wsrc[NWP] += wstep;
            }
        };
        auto wait_keep = [&](auto keep_c) {      // this loader's pieces of all but the last KEEP half tiles have landed
            constexpr int KEEP = decltype(keep_c)::value;
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * (L + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * L) : "memory");
        };
        // prologue: half tiles 0 .. S-2 requested; barrier P = tile 0 has landed
        for (int t = 0; t < S - 1; ++t) stage(t);
        wait_keep(std::integral_constant<int, S - 2>{});
        __builtin_amdgcn_s_barrier();
        // barrier B_g (g = 0 .. nk-2) = half tile g+1 has landed.  A compute wave requests its fragments of tile t between B_(t-1) and
        // B_t and has certainly received them when it arrives at B_(t+1) (its MFMAs of tile t consumed them): behind B_(g-1) the slot of
        // tile g-2 is free and takes tile g+S-2 - S-3 half tiles stay in flight across every barrier, and no compute wave ever waits
        // on LDS for the ring's sake.
        int g = 0;
        if (S - 1 < nk) { wait_keep(std::integral_constant<int, S - 3>{}); __builtin_amdgcn_s_barrier(); g = 1; }      // B_0: nothing to refill yet
        for (; g + S - 2 < nk; ++g) {
            stage(g + S - 2);
            wait_keep(std::integral_constant<int, S - 3>{});
            __builtin_amdgcn_s_barrier();
        }
        // the last S-3 barriers: nothing left to request; tile g+1 must have landed, the nk-2-g behind it stay in flight
        auto tail = [&](auto k_c, auto&& self) {
            constexpr int KK = decltype(k_c)::value;
            if constexpr (KK >= 0) {
                wait_keep(k_c);
                __builtin_amdgcn_s_barrier();
                self(std::integral_constant<int, KK - 1>{}, self);
            }
        };
        tail(std::integral_constant<int, S - 4>{}, tail);
        __syncthreads();                             // the epilogue's barriers, arrived at and left
        if (LORA) __syncthreads();
        return;
    }

    // ================= compute waves: fragment reads + MFMAs, no vector-memory instruction inside the loop ============================
    // folded LayerNorm, consumer side: the four lanes of a row split its chunk pairs (ordinary loads: this wave's vmcnt is its own)
    float ln_mean[XB], ln_rstd[XB];
    if (ln_on) {
        const float nc = (float)(p.K / p.ln_in_chunks), inv_chunks = 1.f / (float)p.ln_in_chunks;
#pragma unroll
        for (int i = 0; i < XB; ++i) {
            const int m = m0 + (wm * XB + i) * 16 + r16;
            const f32x2* src = (const f32x2*)p.ln_in + m;        // chunk-major [chunks][M]
            f32x2 lnp[G7_LNC];
#pragma unroll
            for (int c = 0; c < G7_LNC; ++c) {
                const int ch = c == 0 ? 0 : g4 + 4 * (c - 1);
                lnp[c] = src[(long)(ch < p.ln_in_chunks ? ch : 0) * p.M];
            }
            // equal-sized chunks merged with the chunk means shifted by the first one (gemm_common.h: gemm_ln_finish): every lane its
            // quarter of the chunks, the quarters added across the row's four lanes in a fixed order
            const float m0v = lnp[0][0];
            float sm = 0.f, pq = 0.f, q = 0.f;
#pragma unroll
            for (int c = 1; c < G7_LNC; ++c) {
                const int ch = g4 + 4 * (c - 1);
                if (ch < p.ln_in_chunks) { const float dl = lnp[c][0] - m0v; sm += dl; pq += dl * dl; q += lnp[c][1]; }
            }
            sm += __shfl_xor(sm, 16, 64); pq += __shfl_xor(pq, 16, 64); q += __shfl_xor(q, 16, 64);
            sm += __shfl_xor(sm, 32, 64); pq += __shfl_xor(pq, 32, 64); q += __shfl_xor(q, 32, 64);
            const float dm = sm * inv_chunks;
            ln_mean[i] = m0v + dm;
            const float M2 = q + nc * fmaxf(pq - sm * dm, 0.f);
            ln_rstd[i] = 1.0f / sqrtf(M2 / (float)p.K + p.ln_eps);
            if (p.ln_mr_out && tile_n == 0 && wn == 0 && g4 == 0) *(f32x2*)(p.ln_mr_out + (long)m * 2) = f32x2{ln_mean[i], ln_rstd[i]};
        }
    }

    f32x4_t acc[XB][WB];
#pragma unroll
    for (int i = 0; i < XB; ++i)
#pragma unroll
        for (int j = 0; j < WB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fused adapter: wave (wm, wn) accumulates T = x . A^T for its row blocks i = 2 ii + wn (the partner wave of the row pair takes the
    // others; exchanged through LDS behind the loop).  accl[ii][e] = T[row block, r16][rank 4 g4 + e]
    f32x4_t accl[XB / 2];
#pragma unroll
    for (int ii = 0; ii < XB / 2; ++ii) accl[ii] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragments: lane (g4, r16) holds row r16 of a 16-row block, k = 8 g4 .. + 7 of the half tile (16 bytes)
    const int foff = r16 * 64 + ((g4 ^ g7_key(r16 >> 2)) << 4);
    const int fx = foff + wm * (XB * 1024), fw = foff + XBYTES + wn * (WB * 1024), fl = foff + XBYTES + BN * 64;
    // X fragments in two sets (all of them feed every column of MFMAs), W fragments in ONE: the MFMAs run column by column (j outer), so
    // a column's fragment is dead behind its XB MFMAs and the next half tile's takes its registers right there (128 x 320: 160
    // accumulator + 40 + 2 x 16 fragment registers fit the 256 of a two-waves-per-SIMD kernel; two full sets would not)
    bf16x8 xf[2][XB], wf[WB], lf;
    int cur = 0;
    // half tile g: X fragments in set SET.  MORE: behind barrier B_g (half tile g+1 has landed, ring slot nxt) the X fragments of g+1 are
    // requested one behind each of the first MFMAs and W block j's behind the last MFMA of column j.  No waitcnt by hand: every LDS
    // access of a compute wave is visible to hipcc, which counts lgkmcnt in front of each consumer.
    auto body = [&](auto set_c, auto more_c) {
        constexpr int SET = decltype(set_c)::value;
        constexpr bool MORE = decltype(more_c)::value;
        const int nxt = cur == S - 1 ? 0 : cur + 1;
        const char* nb = smem + nxt * SLOT;
        if constexpr (MORE) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < WB; ++j) {
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[SET][i], acc[i][j], 0, 0, 0);
                if (MORE && j == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    xf[1 - SET][i] = *(const bf16x8*)(nb + fx + i * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MORE) {
                __builtin_amdgcn_sched_barrier(0);
                wf[j] = *(const bf16x8*)(nb + fw + j * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (LORA) {
#pragma unroll
            for (int ii = 0; ii < XB / 2; ++ii) {
                const bf16x8 xs = wn ? xf[SET][2 * ii + 1] : xf[SET][2 * ii];      // both named, the scalar wn selects
                accl[ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, xs, accl[ii], 0, 0, 0);
            }
            if (MORE) {
                __builtin_amdgcn_sched_barrier(0);
                lf = *(const bf16x8*)(nb + fl);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    __builtin_amdgcn_s_barrier();                  // P: half tile 0 has landed
#pragma unroll
    for (int i = 0; i < XB; ++i) xf[0][i] = *(const bf16x8*)(smem + fx + i * 1024);
#pragma unroll
    for (int j = 0; j < WB; ++j) wf[j] = *(const bf16x8*)(smem + fw + j * 1024);
    if (LORA) lf = *(const bf16x8*)(smem + fl);
    for (int g = 0; g + 2 < nk; g += 2) {
        body(P0{}, T_{});
        body(P1{}, T_{});
    }
    body(P0{}, T_{});
    body(P1{}, F_{});

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    // acc[i][j][e] = C[m = m0 + (wm XB + i) 16 + r16][n = n0 + (wn WB + j) 16 + 4 g4 + e]
    constexpr int EX_BYTES = LORA ? 4 * (XB / 2) * 1024 : 0;      // adapter exchange: [wave][ii][lane] f32x4
    __syncthreads();       // every wave is past its last fragment read; every LDS-DMA has landed (vmcnt(0) in the last bodies)
    f32x4_t tl[XB];        // LORA: T of all of this wave's row blocks
    if (LORA) {
        f32x4_t* ex = (f32x4_t*)smem;
#pragma unroll
        for (int ii = 0; ii < XB / 2; ++ii) ex[(wave * (XB / 2) + ii) * 64 + lane] = accl[ii];
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < XB / 2; ++ii) {
            const f32x4_t o = ex[((wave ^ 1) * (XB / 2) + ii) * 64 + lane];
            tl[2 * ii] = wn ? o : accl[ii];
            tl[2 * ii + 1] = wn ? accl[ii] : o;
        }
    }
    const int mrow = m0 + wm * (XB * 16) + r16;            // + 16 i
    const int ncw = n0 + wn * (WB * 16);                   // first column (W row) of this wave
    const int ncol = ncw + 4 * g4;                         // + 16 j

    if (ln_on) {
        // LN(x) . W^T = rstd (x . W'^T - mean s) + b'   (W' = W gamma, s = row sums of W', b' = bias + W beta)
#pragma unroll
        for (int j = 0; j < WB; ++j) {
            const f32x4_t s4 = *(const f32x4_t*)(p.ln_s + ncol + j * 16), b4 = *(const f32x4_t*)(p.ln_b + ncol + j * 16);
#pragma unroll
            for (int i = 0; i < XB; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = ln_rstd[i] * (acc[i][j][e] - ln_mean[i] * s4[e]) + b4[e];
        }
    }
    if (LORA) {
        // up-projection: acc += B[n][0..3] . bf16(scale T[m][ranks of n's group]) as one more MFMA per accumulator block, its 32-deep k
        // axis carrying the rank index (k = 8 g4 + e <-> rank 4 g4 + e, e < 4: T sits in the operand layout already)
        const float lscale = *p.lora_scale;
        if (ln_on && p.ln_lora_s) {      // the adapter's own fold: lora_down holds A . gamma
            const int rk = 4 * g4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool live = rk + e < p.lora_rank;
                const float sa = p.ln_lora_s[live ? rk + e : 0], ca = p.ln_lora_c[live ? rk + e : 0];
#pragma unroll
                for (int i = 0; i < XB; ++i) tl[i][e] = live ? ln_rstd[i] * (tl[i][e] - ln_mean[i] * sa) + ca : tl[i][e];
            }
        }
        if (p.lora_t_out && tile_n == 0 && wn == 0 && 4 * g4 < p.lora_rank) {
#pragma unroll
            for (int i = 0; i < XB; ++i) *(f32x4_t*)(p.lora_t_out + (long)(mrow + i * 16) * p.ld_t + 4 * g4) = tl[i];
        }
        bf16x8 tb[XB];
#pragma unroll
        for (int i = 0; i < XB; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) tb[i][e] = e < 4 ? (__bf16)(lscale * tl[i][e]) : (__bf16)0.f;
#pragma unroll
        for (int j = 0; j < WB; ++j) {
            const int n = ncw + j * 16 + r16;
            const bf16x4 u4 = *(const bf16x4*)(p.lora_up + (long)n * 4);
            const bool mine = n / p.lora_cols_per_group == g4;
            bf16x8 ua;
#pragma unroll
            for (int e = 0; e < 8; ++e) ua[e] = (mine && e < 4) ? u4[e & 3] : (__bf16)0.f;
#pragma unroll
            for (int i = 0; i < XB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua, tb[i], acc[i][j], 0, 0, 0);
        }
    }
    bf16x4 bq[WB];
    const bool use_bias = p.bias != nullptr && !ln_on;      // (folded into b' otherwise)
    if (use_bias) {
#pragma unroll
        for (int j = 0; j < WB; ++j) bq[j] = *(const bf16x4*)(p.bias + ncol + j * 16);
    }
    const __amdgpu_buffer_rsrc_t crs = wt_rsrc(p.c);

    if (p.geglu == 3) {
        // GEGLU, 32-row weight blocks [16 value rows | 16 gate rows]: block j even = values, j + 1 = the gates of the same 16 outputs
        constexpr int GSEG = WB;                      // 16-byte segments of a staged output row (8 WB columns)
        constexpr int GLD = 16 * WB + 16;
        char* sG = smem + EX_BYTES + wave * (16 * GLD);
#pragma unroll
        for (int i = 0; i < XB; ++i) {
#pragma unroll
            for (int j = 0; j < WB; j += 2) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = acc[i][j][e], gt = acc[i][j + 1][e];
                    if (use_bias) { a += (float)bq[j][e]; gt += (float)bq[j + 1][e]; }
                    const float av = round_bf16(a), gv = round_bf16(gt);      // the reference rounds proj(x) to bf16 before chunk / gelu
                    o[e] = (__bf16)(av * round_bf16(gelu_erf_fast_f(gv)));
                }
                *(bf16x4*)(sG + r16 * GLD + (j >> 1) * 32 + g4 * 8) = o;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < (16 * GSEG + 63) / 64; ++it) {
                const int item = it * 64 + lane;
                const int row = item / GSEG, seg = item - row * GSEG;
                if (item < 16 * GSEG) {
                    const bf16x8 v8 = *(const bf16x8*)(sG + row * GLD + seg * 16);
                    wt_store16(crs, ((long)(m0 + (wm * XB + i) * 16 + row) * p.ldc + (ncw >> 1) + seg * 8) * 2, v8);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }

    const bool to_vt = p.vt != nullptr && ncw >= p.vt_col0;      // wave-uniform: these columns are the V block slh_attn_fwd wants transposed
    constexpr int PLD = 32 * WB + 16;                  // bytes per staged row: 16 WB bf16 + 16 (the rows of a quad write fall on distinct banks)
    constexpr int TLD = 32 * XB + 16;                  // transposed patch: one column's 16 XB rows + 16
    constexpr int PATCH = 16 * PLD + 16 * WB * TLD;    // per wave: [16 rows][PLD] row patch, then [16 WB columns][TLD] transposed patch
    static_assert(EX_BYTES + 4 * PATCH <= S * SLOT, "epilogue staging must fit the ring");
    char* sE = smem + EX_BYTES + wave * PATCH;
    char* sT = sE + 16 * PLD;
    constexpr int CW = (WB % 5 == 0) ? 80 : 64;        // LayerNorm chunk width of the producer side
    constexpr int JC = CW / 16;                        // blocks per chunk
    static_assert(WB % JC == 0, "chunks");
#pragma unroll
    for (int i = 0; i < XB; ++i) {
        bf16x4 rq[WB];
        if (p.residual) {
#pragma unroll
            for (int j = 0; j < WB; ++j) rq[j] = *(const bf16x4*)(p.residual + (long)(mrow + i * 16) * p.ld_res + ncol + j * 16);
        }
        float ln_k = 0.f, ln_s = 0.f, ln_q = 0.f;
#pragma unroll
        for (int j = 0; j < WB; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][j][e];
                if (use_bias) v += (float)bq[j][e];
                if (p.residual) v += (float)rq[j][e];
                o[e] = (__bf16)v;
            }
            if (p.ln_out) {      // statistics of the stored (rounded) values of this row's CW columns, shifted by a sample of the row
                if (j % JC == 0) { ln_k = __shfl((float)o[0], r16, 64); ln_s = 0.f; ln_q = 0.f; }
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dlt = (float)o[e] - ln_k; ln_s += dlt; ln_q += dlt * dlt; }
                if (j % JC == JC - 1) {
                    ln_s += __shfl_xor(ln_s, 16, 64); ln_q += __shfl_xor(ln_q, 16, 64);
                    ln_s += __shfl_xor(ln_s, 32, 64); ln_q += __shfl_xor(ln_q, 32, 64);
                    if (g4 == 0) {
                        const float dm = ln_s * (1.f / CW);
                        const int chunk = (ncw / CW) + j / JC;
                        *(f32x2*)(p.ln_out + ((long)chunk * p.M + mrow + i * 16) * 2) = f32x2{ln_k + dm, fmaxf(ln_q - ln_s * dm, 0.f)};
                    }
                }
            }
            if (to_vt) {
                // head-transposed: sT[column (16 WB)][row (16 XB)] - a column of the wave's tile becomes a 32 XB-byte run along the tokens
#pragma unroll
                for (int e = 0; e < 4; ++e) *(__bf16*)(sT + (j * 16 + 4 * g4 + e) * TLD + (i * 16 + r16) * 2) = o[e];
            }
            if (!to_vt || p.vt_also_c) *(bf16x4*)(sE + r16 * PLD + j * 32 + g4 * 8) = o;
        }
        if (to_vt && !p.vt_also_c) continue;
        __builtin_amdgcn_wave_barrier();       // same-wave LDS ops retire in order; only the compiler must not reorder
#pragma unroll
        for (int it = 0; it < (16 * 2 * WB + 63) / 64; ++it) {      // 16 rows x 2 WB sixteen-byte segments
            const int item = it * 64 + lane;
            const int row = item / (2 * WB), seg = item - row * (2 * WB);
            if ((16 * 2 * WB) % 64 == 0 || item < 16 * 2 * WB) {
                const bf16x8 v8 = *(const bf16x8*)(sE + row * PLD + seg * 16);
                wt_store16(crs, ((long)(m0 + (wm * XB + i) * 16 + row) * p.ldc + ncw + seg * 8) * 2, v8);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (to_vt) {
        __builtin_amdgcn_wave_barrier();
        const int Dp = (p.vt_D + 63) & ~63;
        const int mw = m0 + wm * (XB * 16);
        const int bb = mw / p.vt_tokens, tt = mw - bb * p.vt_tokens;      // the wave's 16 XB rows lie inside one sample (vt_tokens % BM == 0)
        constexpr int SEGS = 2 * XB;
#pragma unroll
        for (int it = 0; it < (16 * WB * SEGS) / 64; ++it) {
            const int item = it * 64 + lane;
            const int nl = item / SEGS, seg = item - nl * SEGS;
            const bf16x8 v8 = *(const bf16x8*)(sT + nl * TLD + seg * 16);
            const int nv = ncw + nl - p.vt_col0;
            const int hh = nv / p.vt_D, dd = nv - hh * p.vt_D;
            *(bf16x8*)(p.vt + (((long)bb * p.vt_heads + hh) * Dp + dd) * p.vt_ld + tt + seg * 8) = v8;
        }
    }
}

// ---- 256 x (32 WB) tile on EIGHT compute waves (tile code 0x7<S>8<WB>; round 6) ---------------------------------------------------
// GEGLU.proj (2048 x 10240 x 1280) as 128 x 320 tiles is 512 workgroups = two rounds of the chip: two prologues, two GEGLU epilogues and
// 56 KB staged per 64 of K for every 128 x 320 of output.  As 256 x 320 tiles it is 8 x 32 = 256 workgroups = ONE round, and a
// workgroup stages (256 + 320) rows for twice the MFMA work: 36 % fewer bytes through the L2 -> LDS stream that bounds these loops
// (DESIGN 3.5).  Eight waves = two per SIMD, each the (64 x 160) register tile of the four-wave kernel above (160 accumulator + 40 W-fragment
// + 2 x 16 X-fragment registers of the 256 a wave may use); no room for loader waves, so the waves stage the ring themselves, one LDS-DMA
// piece behind an MFMA, addressed as SGPR base + 32-bit lane offset (three lane registers for all pieces: the K walk lives in the scalar
// bases).  Ring of S half K tiles (36 KB each), S - 2 in flight across a barrier (the barrier sits in the middle of a half tile's MFMAs: see
// the loop body).
// Epilogue: the LayerNorm fold / bias vectors of the tile's columns through LDS once, then bias / residual or GEGLU (16 | 16 blocks).
template <int WB, int S>
__global__ __launch_bounds__(512) void gemm7w_kernel(const G7Args p) {
    constexpr int XB = 4;                                  // 16-row blocks per wave: waves form a 4 x 2 grid
    constexpr int BM = 256, BN = 32 * WB;
    constexpr int XBYTES = BM * 64, SLOT = (BM + BN) * 64;
    constexpr int NPIECE = (BM + BN) / 16;                 // 1 KB pieces per half tile
    constexpr int L = NPIECE / 8, REM = NPIECE % 8;        // every wave L pieces, waves < REM one more
    constexpr int NXPC = BM / 16;                          // the first NXPC pieces are X rows
    constexpr int LNB = BM * 8;                            // (mean, rstd) of the tile's rows, parked behind the ring during the K loop
    static_assert(S * SLOT + LNB <= 160 * 1024 && S >= 4, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[S * SLOT + LNB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    {
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
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K >> 5;
    const int r16 = lane & 15, g4 = lane >> 4;
    const bool ln_on = p.ln_in != nullptr;

    // ---- staging: piece q = wave + 8 k; lane -> (row = lane / 4, physical slot = lane % 4), logical slot fsl (see the kernel above)
    const int frow = lane >> 2, fps = lane & 3;
    const int fsl = fps ^ g7_key(frow >> 2);
    const unsigned vx = (unsigned)(frow * p.lda * 2 + fsl * 16);                                   // X: row pitch, 64-byte K step in the base
    const int wkey = (frow >> 1) & 7;                                                               // a piece starts at a multiple of 16 rows
    const unsigned vw0 = (unsigned)(frow * 128 + ((fsl ^ wkey) << 4));                              // W, first half of the 64-deep block
    const unsigned vw1 = (unsigned)(frow * 128 + (((4 + fsl) ^ wkey) << 4));                        // ... second half
    const bool extra = wave < REM;
    const char* base[L + 1];                               // scalar: this wave's pieces, at half tile 0
#pragma unroll
    for (int k = 0; k < L + 1; ++k) {
        int q = wave + 8 * k;
        q = q < NPIECE ? q : NPIECE - 1;
        if (q < NXPC) base[k] = (const char*)(p.a + (long)(m0 + q * 16) * p.lda);
        else {
            const int pw = q - NXPC;
            base[k] = (const char*)(p.w + ((long)((n0 >> 6) + (pw >> 2)) * (p.K >> 6)) * 4096 + (pw & 3) * (16 * 64));
        }
    }
    const unsigned lds0 = lds_addr_of(smem);
    // piece k of half tile t (PAR = t & 1, compile time) into ring slot `slot`
    auto piece = [&](const int k, const int slot, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        int q = wave + 8 * k;
        q = q < NPIECE ? q : NPIECE - 1;
        const bool is_x = (8 * k + 7 < NXPC) || (8 * k < NXPC && q < NXPC);      // (compile time for the pieces every wave shares a kind)
        const unsigned dst = lds0 + slot * SLOT + q * 1024;
        if (is_x) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vx), "s"(base[k]), "s"(dst) : "memory");
            base[k] += 64;
        } else {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(PAR ? vw1 : vw0), "s"(base[k]), "s"(dst) : "memory");
            if (PAR) base[k] += 8192;
        }
    };
    auto wait_keep = [&](auto keep_c) {
        constexpr int KEEP = decltype(keep_c)::value;
        if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * (L + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * L) : "memory");
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using T_ = std::true_type;
    using F_ = std::false_type;

    // folded LayerNorm, consumer side (as above: ordinary loads ahead of the first LDS-DMA; hipcc waits for them before the prologue's asm)
    // (kept in LDS, not in registers: the loop uses all 256 of them and hipcc would carry these eight through it in scratch)
    f32x2* sLn = (f32x2*)(smem + S * SLOT);
    if (ln_on) {
        const float nc = (float)(p.K / p.ln_in_chunks), inv_chunks = 1.f / (float)p.ln_in_chunks;
#pragma unroll
        for (int i = 0; i < XB; ++i) {
            const int m = m0 + (wm * XB + i) * 16 + r16;
            const f32x2* src = (const f32x2*)p.ln_in + m;
            f32x2 lnp[G7_LNC];
#pragma unroll
            for (int c = 0; c < G7_LNC; ++c) {
                const int ch = c == 0 ? 0 : g4 + 4 * (c - 1);
                lnp[c] = src[(long)(ch < p.ln_in_chunks ? ch : 0) * p.M];
            }
            const float m0v = lnp[0][0];
            float sm = 0.f, pq = 0.f, q = 0.f;
#pragma unroll
            for (int c = 1; c < G7_LNC; ++c) {
                const int ch = g4 + 4 * (c - 1);
                if (ch < p.ln_in_chunks) { const float dl = lnp[c][0] - m0v; sm += dl; pq += dl * dl; q += lnp[c][1]; }
            }
            sm += __shfl_xor(sm, 16, 64); pq += __shfl_xor(pq, 16, 64); q += __shfl_xor(q, 16, 64);
            sm += __shfl_xor(sm, 32, 64); pq += __shfl_xor(pq, 32, 64); q += __shfl_xor(q, 32, 64);
            const float dm = sm * inv_chunks;
            const float mean = m0v + dm;
            const float M2 = q + nc * fmaxf(pq - sm * dm, 0.f);
            const float rstd = 1.0f / sqrtf(M2 / (float)p.K + p.ln_eps);
            if (wn == 0 && g4 == 0) sLn[(wm * XB + i) * 16 + r16] = f32x2{mean, rstd};      // (the wn = 1 wave of the row block computes the same pair)
            if (p.ln_mr_out && tile_n == 0 && wn == 0 && g4 == 0) *(f32x2*)(p.ln_mr_out + (long)m * 2) = f32x2{mean, rstd};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the counted waits below count LDS-DMA pieces only
    }

    f32x4_t acc[XB][WB];
#pragma unroll
    for (int i = 0; i < XB; ++i)
#pragma unroll
        for (int j = 0; j < WB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int foff = r16 * 64 + ((g4 ^ g7_key(r16 >> 2)) << 4);
    const int fx = foff + wm * (XB * 1024), fw = foff + XBYTES + wn * (WB * 1024);
    bf16x8 xf[2][XB], wf[WB];

    // prologue: half tiles 0 .. S-2 requested, tile 0 landed
    {
        auto stage = [&](const int t, auto par_c) {
#pragma unroll
            for (int k = 0; k < L; ++k) piece(k, t, par_c);
            if (extra) piece(L, t, par_c);
        };
#pragma unroll
        for (int t = 0; t < S - 1; ++t) {
            if (t & 1) stage(t, P1{});
            else stage(t, P0{});
        }
    }
    wait_keep(std::integral_constant<int, S - 2>{});
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < XB; ++i) xf[0][i] = *(const bf16x8*)(smem + fx + i * 1024);
#pragma unroll
    for (int j = 0; j < WB; ++j) wf[j] = *(const bf16x8*)(smem + fw + j * 1024);

    int cur = 0;
    // half tile g (ring slot cur, X fragments in set SET), barrier in the MIDDLE of its MFMAs:
    //   first half = columns 0 .. WB/2-1, with this wave's pieces of half tile g+S-1 dealt out behind them (into the slot of g-1: every wave
    //   passed the previous body's barrier, i.e. finished the body before it, whose MFMAs consumed the last fragments of g-1);
    //   s_waitcnt vmcnt(KEEP pieces): my pieces of g+1 have landed - g+2 .. g+S-1 stay in flight: S-2 half tiles cross every barrier
    //   (with the barrier at the top of the body, the first version, it was S-3 = one: 0.96 -> see DESIGN 3.5);  s_barrier;
    //   second half = columns WB/2 .. WB-1, the fragments of g+1 requested behind them: X behind the first XB MFMAs, behind every column its
    //   own W fragment and the one of its first-half partner (both dead by then).
    // MORE: a next tile exists; ISSUE: half tile g+S-1 exists; IPAR: its parity.  All compile time.
    auto body = [&](auto set_c, auto more_c, auto issue_c, auto keep_c, auto ipar_c) {
        constexpr int SET = decltype(set_c)::value;
        constexpr bool MORE = decltype(more_c)::value, ISSUE = decltype(issue_c)::value;
        constexpr int H = WB / 2;
        const int nxt = cur == S - 1 ? 0 : cur + 1;
        const int prv = cur == 0 ? S - 1 : cur - 1;
        const char* nb = smem + nxt * SLOT;
        __builtin_amdgcn_sched_barrier(0);
        int m = 0;
#pragma unroll
        for (int j = 0; j < H; ++j) {
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[SET][i], acc[i][j], 0, 0, 0);
                if (ISSUE && (i & 1) == 1 && m < L) {       // a piece behind every second MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    piece(m, prv, ipar_c);
                    __builtin_amdgcn_sched_barrier(0);
                    ++m;
                }
            }
        }
        if (ISSUE) {
            if (extra) piece(L, prv, ipar_c);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) {
            wait_keep(keep_c);
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = H; j < WB; ++j) {
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[SET][i], acc[i][j], 0, 0, 0);
                if (MORE && j == H) {
                    __builtin_amdgcn_sched_barrier(0);
                    xf[1 - SET][i] = *(const bf16x8*)(nb + fx + i * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MORE) {
                __builtin_amdgcn_sched_barrier(0);
                wf[j] = *(const bf16x8*)(nb + fw + j * 1024);
                wf[j - H] = *(const bf16x8*)(nb + fw + (j - H) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    };
    using KS = std::integral_constant<int, S - 2>;
    int g = 0;
    for (; g + S < nk; g += 2) {
        body(P0{}, T_{}, T_{}, KS{}, std::integral_constant<int, (S - 1) & 1>{});
        body(P1{}, T_{}, T_{}, KS{}, std::integral_constant<int, S & 1>{});
    }
    constexpr int R = (S % 2 == 0) ? S : S - 1;                // half tiles left (nk even, nk >= S)
    auto tail = [&](auto t_c, auto&& self) {
        constexpr int T = decltype(t_c)::value;
        if constexpr (T < R) {
            constexpr bool ISSUE = (T + S - 1) < R;
            // at the barrier of body T half tile T+1 must have landed; requested beyond it: up to min(R-1, T+S-1)
            constexpr int LASTREQ = (T + S - 1) < (R - 1) ? (T + S - 1) : (R - 1);
            constexpr int KEEP = LASTREQ - (T + 1) > 0 ? LASTREQ - (T + 1) : 0;
            body(std::integral_constant<int, T & 1>{}, std::integral_constant<bool, (T < R - 1)>{}, std::integral_constant<bool, ISSUE>{},
                 std::integral_constant<int, KEEP>{}, std::integral_constant<int, (T + S - 1) & 1>{});
            self(std::integral_constant<int, T + 1>{}, self);
        }
    };
    tail(P0{}, tail);

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the tile's column vectors (LayerNorm fold s / b', or the bias) through LDS once per workgroup
    float* sCol = (float*)smem;                             // [2][BN]
    {
        const int n = n0 + tid;
        if (tid < BN) {
            sCol[tid] = ln_on ? p.ln_s[n] : (p.bias ? (float)p.bias[n] : 0.f);
            sCol[BN + tid] = ln_on ? p.ln_b[n] : 0.f;
        }
    }
    __syncthreads();
    const int mrow = m0 + wm * (XB * 16) + r16;
    const int ncw = n0 + wn * (WB * 16);
    const int cl0 = wn * (WB * 16) + 4 * g4;               // this lane's first column inside the tile (+ 16 j)
    f32x2 lnr[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) lnr[i] = ln_on ? sLn[(wm * XB + i) * 16 + r16] : f32x2{0.f, 1.f};      // (written before the first barrier of the kernel)
#pragma unroll
    for (int j = 0; j < WB; ++j) {
        const f32x4_t s4 = *(const f32x4_t*)(sCol + cl0 + j * 16), b4 = *(const f32x4_t*)(sCol + BN + cl0 + j * 16);
#pragma unroll
        for (int i = 0; i < XB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[i][j][e] = ln_on ? lnr[i][1] * (acc[i][j][e] - lnr[i][0] * s4[e]) + b4[e] : acc[i][j][e] + s4[e];
    }
    const __amdgpu_buffer_rsrc_t crs = wt_rsrc(p.c);
    constexpr int COLB = 2 * BN * 4;                        // bytes of the column vectors in front of the patches
    if (p.geglu == 3) {
        constexpr int GSEG = WB;
        constexpr int GLD = 16 * WB + 16;
        static_assert(COLB + 8 * 16 * GLD <= S * SLOT, "epilogue staging");
        char* sG = smem + COLB + wave * (16 * GLD);
#pragma unroll
        for (int i = 0; i < XB; ++i) {
#pragma unroll
            for (int j = 0; j < WB; j += 2) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = round_bf16(acc[i][j][e]), gv = round_bf16(acc[i][j + 1][e]);
                    o[e] = (__bf16)(av * round_bf16(gelu_erf_fast_f(gv)));
                }
                *(bf16x4*)(sG + r16 * GLD + (j >> 1) * 32 + g4 * 8) = o;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < (16 * GSEG + 63) / 64; ++it) {
                const int item = it * 64 + lane;
                const int row = item / GSEG, seg = item - row * GSEG;
                if (item < 16 * GSEG) {
                    const bf16x8 v8 = *(const bf16x8*)(sG + row * GLD + seg * 16);
                    wt_store16(crs, ((long)(m0 + (wm * XB + i) * 16 + row) * p.ldc + (ncw >> 1) + seg * 8) * 2, v8);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    constexpr int PLD = 32 * WB + 16;
    static_assert(COLB + 8 * 16 * PLD <= S * SLOT, "epilogue staging");
    char* sE = smem + COLB + wave * (16 * PLD);
    const int ncol = ncw + 4 * g4;
#pragma unroll
    for (int i = 0; i < XB; ++i) {
        bf16x4 rq[WB];
        if (p.residual) {
#pragma unroll
            for (int j = 0; j < WB; ++j) rq[j] = *(const bf16x4*)(p.residual + (long)(mrow + i * 16) * p.ld_res + ncol + j * 16);
        }
#pragma unroll
        for (int j = 0; j < WB; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][j][e];
                if (p.residual) v += (float)rq[j][e];
                o[e] = (__bf16)v;
            }
            *(bf16x4*)(sE + r16 * PLD + j * 32 + g4 * 8) = o;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < (16 * 2 * WB + 63) / 64; ++it) {
            const int item = it * 64 + lane;
            const int row = item / (2 * WB), seg = item - row * (2 * WB);
            if ((16 * 2 * WB) % 64 == 0 || item < 16 * 2 * WB) {
                const bf16x8 v8 = *(const bf16x8*)(sE + row * PLD + seg * 16);
                wt_store16(crs, ((long)(m0 + (wm * XB + i) * 16 + row) * p.ldc + ncw + seg * 8) * 2, v8);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// group_m of the grouped tile order (gemm5.hip): the gm that minimises the operand rows an XCD pulls through its L2
static int g7_group_m(int tiles_m, int tiles_n, int bm, int bn) {
    const int run = (tiles_m * tiles_n + 7) / 8;
    int best = 1;
    long best_cost = -1;
    for (int gm = 1; gm <= tiles_m; gm *= 2) {
        const int gn = (run + gm - 1) / gm;
        const long cost = (long)bm * gm + (long)bn * (gn < tiles_n ? gn : tiles_n);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = gm; }
    }
    return best;
}

static bool g7_shape(int tile, int& xb, int& wb, int& s) {
    s = (tile >> 8) & 15; xb = (tile >> 4) & 15; wb = tile & 15;
    // (a 128 x 320 four-wave instantiation, 0x754a, was measured and removed: it lost to the ping-pong tile in the pass and is
    // superseded by the 256 x 320 tile - profiles/r06_gemm7_insitu_ab.txt, r06_tile_256x320.txt)
    return (xb == 4 && wb == 8 && s == 6) || (xb == 4 && wb == 5 && s == 6) ||
           (xb == 8 && wb == 10 && s == 4);      // 0x748a: 256 x 320 on eight compute waves
}

// called by slh_gemm (gemm.hip) for tile codes whose bits 12-15 are 7; 1 where the tile named by d->tile can run the descriptor
extern "C" int slh_gemm7_ok(const slh_gemm_desc* d) {
    if (!d || !d->a0 || !d->w || !d->c) return 0;
    int xb, wb, s;
    if (((d->tile >> 12) & 15) != 7 || (d->tile >> 16) || !g7_shape(d->tile, xb, wb, s)) return 0;
    const int bm = 32 * xb, bn = 32 * wb;
    if (d->mode != 0 || d->a1 || d->ca1 || d->w_layout != 1) return 0;
    if (d->M <= 0 || d->M % bm || d->N <= 0 || d->N % bn || d->K % 64 || d->K / 32 < s || d->ca0 != d->K) return 0;
    if (d->lora_t || d->rowbias || d->xa_k || d->geglu_pre) return 0;
    if (d->geglu && (d->geglu != 3 || wb % 2 || d->lora_down || d->residual || d->vt_out || d->ln_out || d->ldc % 8)) return 0;
    if (d->lda0 % 8 || d->ldc % 8 || ((uintptr_t)d->c & 15) || ((uintptr_t)d->a0 & 15) || ((uintptr_t)d->w & 127)) return 0;
    if (d->residual && (d->ld_res % 4 || ((uintptr_t)d->residual & 7))) return 0;
    if (d->bias && ((uintptr_t)d->bias & 7)) return 0;
    if (d->ln_out && (((uintptr_t)d->ln_out & 7) || d->geglu)) return 0;
    if (d->ln_in) {
        if (!d->ln_s || !d->ln_b || d->bias || ((uintptr_t)d->ln_in & 7) || ((uintptr_t)d->ln_s & 15) || ((uintptr_t)d->ln_b & 15)) return 0;
        if (d->ln_in_chunks < 1 || d->ln_in_chunks > 20 || !(d->K == 64 * d->ln_in_chunks || d->K == 80 * d->ln_in_chunks)) return 0;
    } else if (d->ln_mr_out || d->ln_lora_s) {
        return 0;
    }
    if (xb == 8 && (d->lora_down || d->vt_out || d->ln_out)) return 0;      // the 256-row tile: bias / residual / ln_in / geglu = 3 only
    if (d->lora_down) {      // fused adapter, forward form: rank 4 * groups <= 12, each tile inside one column group
        if (wb != 8) return 0;                       // (instantiated on the 128 x 256 tile only)
        if (!d->lora_up || !d->lora_scale || d->lora_up_rmajor || d->lora_groups < 1 || d->lora_groups > 3 ||
            d->lora_rank != 4 * d->lora_groups || d->N % d->lora_groups || (d->N / d->lora_groups) % (16 * wb))
            return 0;
        if (((uintptr_t)d->lora_down & 15) || ((uintptr_t)d->lora_up & 7)) return 0;
        if (d->lora_t_out && (d->ld_t < d->lora_rank || d->ld_t % 4 || ((uintptr_t)d->lora_t_out & 15))) return 0;
    } else if (d->lora_t_out) {
        return 0;
    }
    if (d->vt_out) {
        if (d->geglu || d->vt_D <= 0 || d->vt_D % 64 || d->vt_col0 % (16 * wb) || d->vt_col0 >= d->N || d->vt_heads <= 0 ||
            (d->N - d->vt_col0) != d->vt_heads * d->vt_D || d->vt_tokens % bm || d->M % d->vt_tokens || d->vt_ld % 8 ||
            d->vt_ld < d->vt_tokens || ((uintptr_t)d->vt_out & 15))
            return 0;
    } else if (d->vt_also_c) {
        return 0;
    }
    return 1;
}

int slh_gemm7_launch(const slh_gemm_desc* d, slh_stream_t stream) {
    SLH_CHECK(slh_gemm7_ok(d),
              "slh_gemm: the tiles of gemm7.hip (0x7<S><XB><WB>: 0x7648 = 128 x 256, 0x7645 = 128 x 160, 0x748a = 256 x 320) run dense "
              "single-source products with packed weights, M %% (32 XB) == 0, N %% (32 WB) == 0, K >= 32 S; bias / residual / ln_out / ln_in / "
              "fused adapter (128 x 256) / vt_out / geglu = 3 only (256 x 320: bias / residual / ln_in / geglu = 3) (tile 0x%x M=%d N=%d K=%d)", d ? d->tile : 0, d ? d->M : 0, d ? d->N : 0,
              d ? d->K : 0);
    // (not part of slh_gemm7_ok: the planner asks that before it has built the adapter's fold)
    SLH_CHECK(!(d->ln_in && d->lora_down) || (d->ln_lora_s && d->ln_lora_c),
              "slh_gemm: ln_in with a fused adapter needs ln_lora_s / ln_lora_c (lora_down = A . gamma)");
    int xb, wb, s;
    g7_shape(d->tile, xb, wb, s);
    G7Args a;
    a.a = (const __bf16*)d->a0; a.w = (const __bf16*)d->w; a.bias = (const __bf16*)d->bias; a.residual = (const __bf16*)d->residual;
    a.c = (__bf16*)d->c; a.ln_out = d->ln_out; a.ln_in = d->ln_in; a.ln_s = d->ln_s; a.ln_b = d->ln_b;
    a.ln_lora_s = d->ln_lora_s; a.ln_lora_c = d->ln_lora_c; a.ln_mr_out = d->ln_mr_out;
    a.lora_down = (const __bf16*)d->lora_down; a.lora_up = (const __bf16*)d->lora_up; a.lora_scale = d->lora_scale;
    a.lora_t_out = d->lora_t_out; a.ld_t = d->ld_t; a.vt = (__bf16*)d->vt_out;
    a.lda = d->lda0; a.ldc = d->ldc; a.ld_res = d->ld_res; a.M = d->M; a.N = d->N; a.K = d->K;
    a.tiles_m = d->M / (32 * xb); a.tiles_n = d->N / (32 * wb);
    a.group_m = g7_group_m(a.tiles_m, a.tiles_n, 32 * xb, 32 * wb);
    a.ln_in_chunks = d->ln_in_chunks; a.ln_eps = d->ln_eps;
    a.lora_rank = d->lora_rank; a.lora_cols_per_group = d->lora_down ? d->N / d->lora_groups : 1;
    a.vt_col0 = d->vt_col0; a.vt_D = d->vt_D; a.vt_heads = d->vt_heads; a.vt_tokens = d->vt_tokens; a.vt_ld = d->vt_ld;
    a.vt_also_c = d->vt_also_c; a.geglu = d->geglu;
    const int grid = a.tiles_m * a.tiles_n;
    const hipStream_t st = (hipStream_t)stream;
    if (xb == 8) {
        slh_launch<gemm7w_kernel<10, 4>>(grid, 512, st, a, "gemm7w_kernel<10, 4>");
    } else if (wb == 8) {
        if (d->lora_down) slh_launch<gemm7_kernel<4, 8, 6, true>>(grid, 512, st, a, "gemm7_kernel<4, 8, 6, true>");
        else slh_launch<gemm7_kernel<4, 8, 6, false>>(grid, 512, st, a, "gemm7_kernel<4, 8, 6, false>");
    } else {
        slh_launch<gemm7_kernel<4, 5, 6, false>>(grid, 512, st, a, "gemm7_kernel<4, 5, 6, false>");
    }
    SLH_LAUNCH_CHECK("slh_gemm (four-wave tile)");
    return 0;
}
