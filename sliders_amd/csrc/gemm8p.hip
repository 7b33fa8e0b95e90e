// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution, 256 x 256 block tile, ping-pong ("8-phase") K loop.  gfx950 (CDNA4) only.
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T )          same operands, layouts and epilogue as gemm.hip (gemm_common.h)
//
// Why a second K loop: a 128 x 128 tile stages 32 KB per 64-deep K tile for 2.1 MFLOP, i.e. it needs 64 B/clk/CU of L2 -> LDS
// delivery at the MFMA peak - exactly the CU's load-path peak, of which ~36 B/clk is reached in practice - and its 32 x 64
// wave tiles read 3 KB of fragments per 2 MFMAs (LDS at 75 % of the matrix time).  256 x 256 with 128 x 64 per wave halves both
// ratios (32 B/clk/CU, 24 fragment reads per 32 MFMAs); what it loses is occupancy (one 8-wave workgroup per CU, 128 KB of LDS),
// so the latency hiding has to be built into the schedule (cdna_hip_programming.md section 5, "256^2 8-phase template"):
//
//  * 8 waves = 2 groups of 4 (one wave of each group per SIMD).  Group g owns rows g*128..+128 of the tile, wave wc of a group
//    the columns wc*64..+64: per wave MI = 4 x NI = 2 accumulator blocks of 32 x 32 (128 fp32 registers).
//  * A K tile (64 deep) is four PHASES, one per quadrant of the wave tile (2 row blocks x 1 column block x 4 k-steps = 8 MFMAs of
//    32 cycles).  Phase = [fragment reads of what the quadrant needs and the wave does not hold yet | LDS-DMA of one 16 KB unit
//    of a LATER K tile | counted vmcnt] s_barrier [lgkmcnt(0) | 8 MFMAs at raised priority] s_barrier.
//  * The groups run ONE BARRIER APART (group 1 takes an extra s_barrier before the loop, group 0 after it): while one wave of a
//    SIMD is in its MFMA half-phase its partner is in the read / DMA half-phase - a matrix-only wave beside a memory-only wave
//    overlap perfectly on a SIMD (profiles/r03_ubench_mfma_valu.txt), mixed waves do not.
//  * LDS-DMA stays in flight across barriers: every phase issues one unit (2 x global_load_lds_dwordx4 per wave) and waits
//    vmcnt(8) = "everything but the last four units has landed"; a unit is read no earlier than the phase AFTER the wait that
//    covers it (every wave has waited for its own share and passed a barrier since), and an LDS region is re-staged no earlier
//    than two phases after its last read (so the lagging group is done with it as well).  Never vmcnt(0) in the steady state.
//
// Unit order per K tile t (buffer t & 1):  U1 = X rows of the first row blocks (read in phase 1), U2 = W rows of column block 0
// (phase 1), U3 = W rows of column block 1 (phase 2), U4 = X rows of the second row blocks (phase 3).  Issue points:
// U3(t+1) in phase 1 of tile t, U4(t+1) in phase 2, U1(t+2) in phase 3, U2(t+2) in phase 4 - each 5-6 phases (>= 2500 cycles)
// ahead of its first read.
//
// Reference semantics replaced: as gemm.hip (torch.nn.functional.linear / conv2d inside diffusers-0.20.2 + lora.py:108-112).
#include "gemm_common.h"

using namespace slh_gemm_detail;

// ablation switches for scripts/build_variant.sh builds (-DSLH8P_ABL=<bits>): 1 no LDS-DMA inside the K loop, 2 no MFMA,
// 4 no fragment reads, 8 no stagger between the wave groups, 16 no priority raise around the MFMAs.  0 in the product build.
#ifndef SLH8P_ABL
#define SLH8P_ABL 0
#endif
// schedule variants: 1 = half of the waves issue the phase's LDS-DMA ahead of its fragment reads (LDS and the load path busy
// at the same time), 2 = the LDS-DMA pieces ride in the shadow of the MFMAs, 4 = linear X source (timing only, wrong results)
#ifndef SLH8P_VAR
#define SLH8P_VAR 0
#endif
#ifndef SLH8P_VM
#define SLH8P_VM ((SLH8P_VAR & 2) ? 6 : 8)   /* VAR 8 implies 2 */
#endif

namespace {

template <int MODE, bool LORA>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const GemmArgs p) {
    constexpr int MI = 4, NI = 2, NW = 8, WN = 4;
    constexpr int BM = 256, BN = 256;
    constexpr int XB = BM * 128, WB = BN * 128, BUF = XB + WB;      // one K tile: 32 KB of X rows + 32 KB of W rows
    static_assert(!LORA, "fused adapter: not in this K loop yet");
    static_assert(gemm_epilogue_lds(MI, NI, NW, WN, LORA) <= 2 * BUF, "epilogue staging must fit the operand buffers");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;          // = (wm, wn) of the epilogue
    const int lrow = lane & 31, lhi = lane >> 5;

    int tile_m, tile_n, ks_id;
    gemm_map_tile(p, tile_m, tile_n, ks_id);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int kt_begin = 0, nk = p.K / BK;
    if (p.splitk > 1) {
        kt_begin = ks_id * p.kper;          // K tiles per slice: computed once, by slh_gemm, which also makes every slice non-empty
        nk = min(nk, kt_begin + p.kper) - kt_begin;
    }

    // ---- staging geometry: every unit is 16 groups of 8 rows; wave w copies groups w and w + 8 -------------------------------
    // X rows (tile-relative) of this wave's four 8-row groups: U1 {w*8, 128 + w*8}, U4 {64 + w*8, 192 + w*8}
    // W rows:                                                  U2 {b, 128 + b},     U3 {32 + b, 160 + b},  b = (w>>2)*64 + (w&3)*8
    const int frow = lane >> 3, fslot = lane & 7;
    const int cin = p.ca0 + p.ca1;
    const int xrow0 = wave * 8;                               // + {0, 128, 64, 192}
    const int wrow0 = (wave >> 2) * 64 + (wave & 3) * 8;      // + {0, 128, 32, 160}
    constexpr int XOFF[4] = {0, 128, 64, 192};
    constexpr int WOFF[4] = {0, 128, 32, 160};


    const int wkstep = p.w_packed ? 4096 : BK;   // elements between consecutive K tiles of one W row group
    const char* wsrc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int row = wrow0 + WOFF[a] + frow;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        const __bf16* wp = p.w_packed ? p.w + ((long)(n >> 6) * (p.K >> 6)) * 4096 + ((n & 63) << 6) + (fslot << 3)
                                      : p.w + (long)n * p.ldw + ((fslot ^ ((row >> 1) & 7)) << 3);
        wsrc[a] = (const char*)(wp + (long)kt_begin * wkstep);
    }
    int wkbytes = wkstep * 2;

    // running source pointers (advanced once per K tile).  X is re-based when the implicit GEMM moves to the next filter tap or
    // the second concat source (same scheme as the ring loop of gemm.hip).  The schedule keeps issuing units for two K tiles past
    // the end of the slice: those come from the zero page (pointer parked, no advance) and land in rows nobody reads again.
    const char* xsrc[4];
    unsigned xmove = 15;                   // bit a: group a's pointer advances with K (conv: 0 for a tap outside the image - the zero page)
    const char* zero_page = (const char*)slh_zero_page;
    asm volatile("" : "+s"(zero_page));    // formed once (a GOT load), not re-materialised inside the K loop
    int i_c0 = kt_begin * BK, i_tap = 0;
    if (MODE == 1) { i_tap = i_c0 / cin; i_c0 -= i_tap * cin; }
    bool i_first = true;
    int x_left = nk, w_left = nk;          // K tiles whose X / W units have not been issued yet
    auto rebase_x = [&]() {
        const bool s1 = i_c0 >= p.ca0;
        const __bf16* base = s1 ? p.a1 : p.a0;
        const int cc = s1 ? i_c0 - p.ca0 : i_c0;
        if (MODE == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = xrow0 + XOFF[a] + frow;
                int m = m0 + row;
                m = m < p.M ? m : p.M - 1;
                xsrc[a] = (const char*)(base + (long)m * (s1 ? p.lda1 : p.lda0) + cc + (((SLH8P_VAR & 4) ? fslot : (fslot ^ ((row >> 1) & 7))) << 3));
            }
            xmove = 15;
        } else {
            const int ld = s1 ? p.lda1 : p.lda0;
            const int ky = i_tap / 3, kx = i_tap - ky * 3;
            const int sh = p.src_xform ? 1 : 0;
            const int HL = p.hs << sh, WL = p.ws << sh;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = xrow0 + XOFF[a] + frow;
                // sample / output pixel of the row: recomputed here (9 x sources times per K slice) rather than carried in 12
                // registers through the K loop
                int m = m0 + row;
                m = m < p.M ? m : p.M - 1;
                const int hw = p.ho * p.wo;
                const int xb_a = m / hw;
                const int rem = m - xb_a * hw;
                const int xoy_a = rem / p.wo, xox_a = rem - xoy_a * p.wo;
                const int iy = xoy_a * p.stride + ky - 1;
                const int ix = xox_a * p.stride + kx - 1;
                bool ok = (iy >= 0) & (iy < HL) & (ix >= 0) & (ix < WL);
                if (p.src_xform == 2) ok = ok & (((iy | ix) & 1) == 0);
                const int sy = iy >> sh, sx = ix >> sh;
                const long pix = ((long)xb_a * p.hs + sy) * p.ws + sx;
                xsrc[a] = ok ? (const char*)(base + pix * ld + cc + ((fslot ^ ((row >> 1) & 7)) << 3)) : zero_page;
                xmove = ok ? (xmove | (1u << a)) : (xmove & ~(1u << a));
            }
        }
    };
    const unsigned lds0 = lds_addr_of(smem);
    bool in_loop = false;
    auto dma = [&](const void* src, const unsigned dst) {
        if ((SLH8P_ABL & 1) && in_loop) return;
        glds16_hidden(src, dst);
    };
    // LDS byte address (wave-uniform) of this wave's 8-row group `a` of operand X / W in the buffer at byte offset `bo`
    auto x_dst = [&](const unsigned bo, const int a) { return lds0 + bo + (xrow0 + XOFF[a]) * 128; };
    auto w_dst = [&](const unsigned bo, const int a) { return lds0 + bo + XB + (wrow0 + WOFF[a]) * 128; };

    // unit issue (2 LDS-DMA instructions per wave each; unit 1 = U1 ... 4 = U4).  X units go out in the order U1(t), U4(t),
    // U1(t+1) ...; W units U2(t), U3(t), U2(t+1) ... - the running pointers advance behind the second unit of a tile.
    auto unit_begin = [&](const int u) {
        if (u == 1) {
            if (x_left <= 0) {
                if (x_left == 0) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) xsrc[a] = zero_page;
                    xmove = 0;
                }
            } else if (i_first || i_c0 == 0 || i_c0 == p.ca0) {
                rebase_x();
            }
            i_first = false;
        } else if (u == 2) {
            if (w_left == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) wsrc[a] = zero_page;
                wkbytes = 0;
            }
        }
    };
    auto unit_piece = [&](const int u, const int k, const unsigned bo) {
        if (u == 1) dma(xsrc[k], x_dst(bo, k));
        else if (u == 4) dma(xsrc[2 + k], x_dst(bo, 2 + k));
        else if (u == 2) dma(wsrc[k], w_dst(bo, k));
        else dma(wsrc[2 + k], w_dst(bo, 2 + k));
    };
    auto unit_end = [&](const int u) {
        if (u == 4) {
#pragma unroll
            for (int a = 0; a < 4; ++a) xsrc[a] += (SLH8P_ABL & 32) ? 0 : (int)((xmove >> a) & 1) << 7;     // ABL 32: every K tile re-reads the first one (L2-hit rate of the DMA path)
            i_c0 += BK;
            if (MODE == 1 && i_c0 == cin) { i_c0 = 0; ++i_tap; }
            --x_left;
        } else if (u == 3) {
#pragma unroll
            for (int a = 0; a < 4; ++a) wsrc[a] += (SLH8P_ABL & 32) ? 0 : wkbytes;
            --w_left;
        }
    };
    auto issue_unit = [&](const int u, const unsigned bo) {
        unit_begin(u);
        unit_piece(u, 0, bo);
        unit_piece(u, 1, bo);
        unit_end(u);
    };

    // ---- folded LayerNorm of the A operand (consumer side): statistics requested ahead of the first tiles ---------------------
    float ln_mean[MI], ln_rstd[MI];
    const bool ln_on = MODE == 0 && !LORA && p.ln_in != nullptr;
    {
        f32x2 ln_pairs[MI][LN_MAXC];
        if (MODE == 0 && ln_on) {
            gemm_ln_request<MI>(p, m0 + grp * 128, lrow, ln_pairs);
            gemm_ln_finish<MI>(p, m0 + grp * 128, lrow, tile_n == 0 && wc == 0 && lhi == 0, ln_pairs, ln_mean, ln_rstd);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of the compiler's own loads is pending when the DMA counting starts

    // ---- prologue: tile 0 complete + the first two units of tile 1 in flight; U1(0), U2(0) landed --------------------------------
    issue_unit(1, 0);
    issue_unit(2, 0);
    issue_unit(3, 0);
    issue_unit(4, 0);
    issue_unit(1, BUF);
    issue_unit(2, BUF);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 accl[MI];

    // per-lane fragment read addresses: row = (32-aligned block) + lrow, slot = ks*2 + lhi, swizzled with (lrow >> 1) & 7;
    // the block / column-block / buffer parts are instruction offsets (< 64 KB) or added once per K tile
    const char* xbase = smem + (grp * 128) * 128;           // this wave's X rows
    const char* wbase = smem + XB + (wc * 64) * 128;        // this wave's W rows

    bf16x8 xf[2][4], wf[2][4];      // X: the two row blocks of the current half x 4 k-steps; W: both column blocks x 4 k-steps
    auto read_x = [&](const unsigned bo, const int half) {
        if (SLH8P_ABL & 4) return;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                xf[b][ks] = *(const bf16x8*)(xbase + bo + (half * 2 + b) * 4096 + lds_off(lrow, ks * 2 + lhi));
    };
    auto read_w = [&](const unsigned bo, const int j) {
        if (SLH8P_ABL & 4) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[j][ks] = *(const bf16x8*)(wbase + bo + j * 4096 + lds_off(lrow, ks * 2 + lhi));
    };
    // the 8 MFMAs of a quadrant; with SLH8P_VAR & 2 the two LDS-DMA pieces of unit `u` go out in their shadow
    auto mfmas = [&](const int half, const int j, const int u, const unsigned ubo) {
        if (!(SLH8P_ABL & 16)) __builtin_amdgcn_s_setprio(1);
        if (SLH8P_VAR & 2) unit_begin(u);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (SLH8P_ABL & 2) asm volatile("" ::"v"(wf[j][ks]), "v"(xf[b][ks]));
                else if (SLH8P_VAR & 16) {
                    // timing experiment (wrong numbers): the same operand registers and FLOPs as two 16x16x32 MFMAs, whose
                    // accumulator traffic per FLOP is half that of 32x32x16
                    f32x16& A = acc[half * 2 + b][j];
                    f32x4 q0 = {A[ks * 4], A[ks * 4 + 1], A[ks * 4 + 2], A[ks * 4 + 3]};
                    f32x4 q1 = {A[(ks ^ 2) * 4], A[(ks ^ 2) * 4 + 1], A[(ks ^ 2) * 4 + 2], A[(ks ^ 2) * 4 + 3]};
                    q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], xf[b][ks], q0, 0, 0, 0);
                    q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], xf[b][ks], q1, 0, 0, 0);
                    A[ks * 4] = q0[0]; A[ks * 4 + 1] = q0[1]; A[ks * 4 + 2] = q0[2]; A[ks * 4 + 3] = q0[3];
                    A[(ks ^ 2) * 4] = q1[0]; A[(ks ^ 2) * 4 + 1] = q1[1]; A[(ks ^ 2) * 4 + 2] = q1[2]; A[(ks ^ 2) * 4 + 3] = q1[3];
                }
                else acc[half * 2 + b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ks], xf[b][ks], acc[half * 2 + b][j], 0, 0, 0);
                if ((SLH8P_VAR & 8) ? (wc == ((ks * 2 + b) & 3)) : ((SLH8P_VAR & 2) && b == 1 && (ks == 0 || ks == 2))) {
                    // (VAR 8: wave wc issues behind MFMA wc and wc + 4 - the four waves of the group, which run in lockstep on
                    // four SIMDs, hand the load path one piece per 32-cycle MFMA slot instead of a burst of four)
                    __builtin_amdgcn_sched_barrier(0);
                    unit_piece(u, (SLH8P_VAR & 8) ? ks >> 1 : ks >> 1, ubo);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        if (SLH8P_VAR & 2) unit_end(u);
        if (!(SLH8P_ABL & 16)) __builtin_amdgcn_s_setprio(0);
    };
    // the barrier pair of a phase: [reads / DMA issued above] -> barrier -> fragments landed -> MFMAs -> barrier
#define SLH_PHASE_MID()                              \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ \
    __builtin_amdgcn_sched_barrier(0)
#define SLH_PHASE_END()                              \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    __builtin_amdgcn_sched_barrier(0)
#define SLH_VMWAIT() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLH8P_VM) : "memory")

    __builtin_amdgcn_s_barrier();                 // every wave's share of U1(0), U2(0) has landed
    if (!(SLH8P_ABL & 8) && grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0 from here on

    in_loop = true;
    const bool dma_first = (SLH8P_VAR & 1) && wc >= 2;   // half of the waves issue their LDS-DMA ahead of the fragment reads
    // read half of a phase: the fragment reads `rd` and (unless it rides in the MFMA shadow) unit u into the buffer at ubo
    auto read_half = [&](auto rd, const int u, const unsigned ubo, const bool wait) {
        if (SLH8P_VAR & 2) {
            rd();
            __builtin_amdgcn_sched_barrier(0);
        } else if (dma_first) {
            issue_unit(u, ubo);
            __builtin_amdgcn_sched_barrier(0);
            rd();
            __builtin_amdgcn_sched_barrier(0);
        } else {
            rd();
            __builtin_amdgcn_sched_barrier(0);
            issue_unit(u, ubo);
        }
        if (wait) SLH_VMWAIT();
    };
    unsigned bo = 0;                              // byte offset of the current K tile's buffer
    for (int s = 0; s < nk; ++s) {
        const unsigned nb = bo ^ BUF;
        // phase 1: quadrant (row blocks 0-1, column block 0); U3(s+1); wait: U3(s) landed (read in phase 2)
        read_half([&]() { read_x(bo, 0); read_w(bo, 0); }, 3, nb, true);
        SLH_PHASE_MID();
        mfmas(0, 0, 3, nb);
        SLH_PHASE_END();
        // phase 2: (row blocks 0-1, column block 1); U4(s+1); wait: U4(s) landed (read in phase 3)
        read_half([&]() { read_w(bo, 1); }, 4, nb, true);
        SLH_PHASE_MID();
        mfmas(0, 1, 4, nb);
        SLH_PHASE_END();
        // phase 3: (row blocks 2-3, column block 1); U1(s+2) into the rows last read in phase 1
        read_half([&]() { read_x(bo, 1); }, 1, bo, false);
        SLH_PHASE_MID();
        mfmas(1, 1, 1, bo);
        SLH_PHASE_END();
        // phase 4: (row blocks 2-3, column block 0); U2(s+2); wait: U1(s+1), U2(s+1) landed (read in phase 1 of the next tile)
        read_half([&]() {}, 2, bo, true);
        SLH_PHASE_MID();
        mfmas(1, 0, 2, bo);
        SLH_PHASE_END();
        bo = nb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-page units of the two tiles past the end: landed before the epilogue recycles the buffers
    if (!(SLH8P_ABL & 8) && grp == 0) __builtin_amdgcn_s_barrier();        // re-align the groups
    __builtin_amdgcn_s_barrier();
#undef SLH_PHASE_MID
#undef SLH_PHASE_END
#undef SLH_VMWAIT

    gemm_epilogue<MI, NI, MODE, LORA, NW, WN>(p, smem, acc, accl, ln_mean, ln_rstd, ln_on, tile_m, tile_n, ks_id, wave, grp, wc);
}


// ---- layout B: 128 x (64*NI) block tile, waves 4 (M) x 2 (N) as in gemm.hip's 8-wave tiles, 32 x (32*NI) per wave ---------------
// The same ping-pong schedule with the phases cut along N: phase p multiplies the wave's 32 rows with its column blocks 2p, 2p+1
// (the last phase one block when NI is odd; one block per phase for NI < 5, which keeps >= 3 phases per K tile).  Group g = waves 4g .. 4g+3 = rows g*64 .. +64.  Staging units in need order:
// UX (all 128 X rows, 2 pieces per wave, read in phase 0), UW_p (the W rows of phase p's blocks for both wave columns: 64 rows
// = 1 piece per wave and block).  A unit needed in phase f is issued NP + 1 phases ahead (into the rows its buffer's previous
// tile gave up NP + 1 - 2*NP + ... >= 2 phases earlier) and waited for in phase f - 1 with vmcnt(G), G = NI + 2 = the pieces of
// one K tile per wave: exactly one K tile of LDS-DMA stays in flight across every barrier.
template <int MI, int NI, int MODE, bool LORA>
__global__ __launch_bounds__(512, 2) void gemm8pb_kernel(const GemmArgs p) {
    constexpr int NW = 8, WN = 2;
    constexpr int BM = 128 * MI, BN = 64 * NI;
    constexpr int XG = 2 * MI;                        // 8-row X groups per wave
    constexpr int XB = BM * 128, WB = BN * 128, LB = LORA ? 32 * 128 : 0, BUF = XB + WB + LB;
    constexpr int BP = (NI >= 5 && MI == 1) ? 2 : 1;  // column blocks per phase (MI = 2: one block = 8 MFMAs, and 16 fragment registers less)
    constexpr int NP = (NI + BP - 1) / BP;           // phases per K tile; re-staging a region NP + 1 phases ahead of its next read
                                                     // leaves NP - 1 >= 2 phases behind its last one (see the 256 x 256 kernel)
    constexpr int G = NI + XG + (LORA ? 1 : 0);      // LDS-DMA pieces per wave and K tile
    // LORA: the rank-r down matrix (lora_down [r][K], r <= 12, zero-padded to 32 rows) rides along as a third operand tile, as in
    // gemm.hip: its piece travels with the X unit, the two waves that share 32 rows split its k-steps (wn = 0 the even ones), their
    // MFMAs go into the last - shortest - phase, and the shared epilogue's exchange (wave ^ 1) joins the halves.
    static_assert(NI >= 3 && NI <= 5 && NP >= 3 && MI == 1, "128 x 192 ... 128 x 320 (a 256 x 320 tile, MI = 2, was built and measured: 160 accumulator registers leave no room, and one round of 256 big tiles loses to two of 512 - docs/ROUND_NOTES.md)");
    static_assert(gemm_epilogue_lds(MI, NI, NW, WN, LORA) <= 2 * BUF, "epilogue staging must fit the operand buffers");
    static_assert(2 * BUF <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int lrow = lane & 31, lhi = lane >> 5;

    int tile_m, tile_n, ks_id;
    gemm_map_tile(p, tile_m, tile_n, ks_id);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int kt_begin = 0, nk = p.K / BK;
    if (p.splitk > 1) {
        kt_begin = ks_id * p.kper;          // K tiles per slice: computed once, by slh_gemm, which also makes every slice non-empty
        nk = min(nk, kt_begin + p.kper) - kt_begin;
    }

    // ---- staging geometry ---------------------------------------------------------------------------------------------------------
    // X: 16*MI groups of 8 rows, wave w copies groups w + 8a (rows w*8 + 64a).
    // W block j (both wave columns: 64 rows = 8 groups): wave w copies group w = (wn' = w >> 2, r8 = w & 3): row wn'*32*NI + j*32 + r8*8
    const int frow = lane >> 3, fslot = lane & 7;
    const int cin = p.ca0 + p.ca1;
    const int xrow0 = wave * 8;                                      // + {0, 64}
    const int wrow0 = (wave >> 2) * (32 * NI) + (wave & 3) * 8;      // + j*32
    int xb[XG], xoy[XG], xox[XG];
    if (MODE == 1) {
#pragma unroll
        for (int a = 0; a < XG; ++a) {
            int m = m0 + xrow0 + a * 64 + frow;
            m = m < p.M ? m : p.M - 1;
            const int hw = p.ho * p.wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / p.wo;
            xb[a] = b; xoy[a] = oy; xox[a] = rem - oy * p.wo;
        }
    }
    const int wkstep = p.w_packed ? 4096 : BK;
    const char* wsrc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int row = wrow0 + j * 32 + frow;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        const __bf16* wp = p.w_packed ? p.w + ((long)(n >> 6) * (p.K >> 6)) * 4096 + ((n & 63) << 6) + (fslot << 3)
                                      : p.w + (long)n * p.ldw + ((fslot ^ ((row >> 1) & 7)) << 3);
        wsrc[j] = (const char*)(wp + (long)kt_begin * wkstep);
    }
    int wkbytes = wkstep * 2;
    const char* lsrc = nullptr;
    int ladv = 0;
    const char* xsrc[XG];
    int xadv[XG];
    const char* zero_page = (const char*)slh_zero_page;
    asm volatile("" : "+s"(zero_page));
    int i_c0 = kt_begin * BK, i_tap = 0;
    if (LORA) {
        const int row = (wave & 3) * 8 + frow;       // waves 4-7 re-issue the rows of waves 0-3 (same bytes, same place: benign)
        const bool ok = row < p.lora_rank;
        lsrc = ok ? (const char*)(p.lora_down + (long)row * p.K + kt_begin * BK + ((fslot ^ ((row >> 1) & 7)) << 3)) : zero_page;
        ladv = ok ? 128 : 0;
    }
    if (MODE == 1) { i_tap = i_c0 / cin; i_c0 -= i_tap * cin; }
    bool i_first = true;
    int x_left = nk, w_left = nk;
    auto rebase_x = [&]() {
        const bool s1 = i_c0 >= p.ca0;
        const __bf16* base = s1 ? p.a1 : p.a0;
        const int cc = s1 ? i_c0 - p.ca0 : i_c0;
        if (MODE == 0) {
#pragma unroll
            for (int a = 0; a < XG; ++a) {
                const int row = xrow0 + a * 64 + frow;
                int m = m0 + row;
                m = m < p.M ? m : p.M - 1;
                xsrc[a] = (const char*)(base + (long)m * (s1 ? p.lda1 : p.lda0) + cc + ((fslot ^ ((row >> 1) & 7)) << 3));
                xadv[a] = 128;
            }
        } else {
            const int ld = s1 ? p.lda1 : p.lda0;
            const int ky = i_tap / 3, kx = i_tap - ky * 3;
            const int sh = p.src_xform ? 1 : 0;
            const int HL = p.hs << sh, WL = p.ws << sh;
#pragma unroll
            for (int a = 0; a < XG; ++a) {
                const int row = xrow0 + a * 64 + frow;
                const int iy = xoy[a] * p.stride + ky - 1;
                const int ix = xox[a] * p.stride + kx - 1;
                bool ok = (iy >= 0) & (iy < HL) & (ix >= 0) & (ix < WL);
                if (p.src_xform == 2) ok = ok & (((iy | ix) & 1) == 0);
                const int sy = iy >> sh, sx = ix >> sh;
                const long pix = ((long)xb[a] * p.hs + sy) * p.ws + sx;
                xsrc[a] = ok ? (const char*)(base + pix * ld + cc + ((fslot ^ ((row >> 1) & 7)) << 3)) : zero_page;
                xadv[a] = ok ? 128 : 0;
            }
        }
    };
    const unsigned lds0 = lds_addr_of(smem);
    // unit u: 0 = UX, 1 + q = UW_q.  K tiles past the end of the slice come from the zero page (see the 256 x 256 kernel).
    auto issue_unit = [&](const int u, const unsigned bo) {
        if (u == 0) {
            if (x_left <= 0) {
                if (x_left == 0) {
#pragma unroll
                    for (int a = 0; a < XG; ++a) { xsrc[a] = zero_page; xadv[a] = 0; }
                    if (LORA) { lsrc = zero_page; ladv = 0; }
                }
            } else if (i_first || i_c0 == 0 || i_c0 == p.ca0) {
                rebase_x();
            }
            i_first = false;
#pragma unroll
            for (int a = 0; a < XG; ++a) {
                glds16_hidden(xsrc[a], lds0 + bo + (xrow0 + a * 64) * 128);
                xsrc[a] += xadv[a];
            }
            if (LORA) {
                glds16_hidden(lsrc, lds0 + bo + XB + WB + (wave & 3) * 1024);
                lsrc += ladv;
            }
            i_c0 += BK;
            if (MODE == 1 && i_c0 == cin) { i_c0 = 0; ++i_tap; }
            --x_left;
        } else {
            const int q = u - 1;
            if (q == 0 && w_left == 0) {
#pragma unroll
                for (int j = 0; j < NI; ++j) wsrc[j] = zero_page;
                wkbytes = 0;
            }
#pragma unroll
            for (int j = BP * q; j < BP * q + BP && j < NI; ++j) {
                glds16_hidden(wsrc[j], lds0 + bo + XB + (wrow0 + j * 32) * 128);
                wsrc[j] += wkbytes;
            }
            if (q == NP - 1) --w_left;
        }
    };

    float ln_mean[MI], ln_rstd[MI];
    // with LORA: the adapter's down-projection is folded too (ln_lora_*; NI <= 4: the 128 x 320 tile has no registers left for it)
    const bool ln_on = MODE == 0 && (!LORA || NI <= 4) && p.ln_in != nullptr;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of the compiler's own loads is pending when the counting starts
    // The chunk statistics of a folded LayerNorm are requested AHEAD of the prologue's LDS-DMA (older in the in-order vmcnt queue:
    // the counted wait below retires them with the first units) and merged behind it - requested and merged in front of it they
    // were a serial round trip at the head of a kernel that owns its CU alone.
    f32x2 ln_pairs[MI][LN_MAXC];
    if (MODE == 0 && ln_on) gemm_ln_request_hidden<MI>(p, m0 + wm * (32 * MI), lrow, ln_pairs);

    // ---- prologue: tile 0 complete + the phase-0 units of tile 1 in flight; the phase-0 units of tile 0 landed -----------------
    issue_unit(0, 0);
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_unit(1 + q, 0);
    issue_unit(0, BUF);
    issue_unit(1, BUF);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    if (MODE == 0 && ln_on) {
        gemm_ln_landed<MI>(ln_pairs);
        gemm_ln_finish<MI>(p, m0 + wm * (32 * MI), lrow, tile_n == 0 && wn == 0 && lhi == 0, ln_pairs, ln_mean, ln_rstd);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 accl[MI];
    if (LORA) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accl[0][r] = 0.f;
    }

    const char* xbase = smem + (wm * 32 * MI) * 128;
    const char* wbase = smem + XB + (wn * 32 * NI) * 128;
    bf16x8 xf[MI][4], wf[BP][4], lf[2];
    auto read_x = [&](const unsigned bo) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xf[i][ks] = *(const bf16x8*)(xbase + bo + i * 4096 + lds_off(lrow, ks * 2 + lhi));
        if (LORA) {      // this wave's k-steps of the adapter tile: ks = wn and wn + 2
#pragma unroll
            for (int h = 0; h < 2; ++h) lf[h] = *(const bf16x8*)(smem + bo + XB + WB + lds_off(lrow, (wn + 2 * h) * 2 + lhi));
        }
    };
    auto read_w = [&](const unsigned bo, const int q) {
#pragma unroll
        for (int jj = 0; jj < BP; ++jj)
            if (BP * q + jj < NI) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wf[jj][ks] = *(const bf16x8*)(wbase + bo + (BP * q + jj) * 4096 + lds_off(lrow, ks * 2 + lhi));
            }
    };
    auto mfmas = [&](const int q) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int jj = 0; jj < BP; ++jj)
                if (BP * q + jj < NI) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[i][BP * q + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[jj][ks], xf[i][ks], acc[i][BP * q + jj], 0, 0, 0);
                }
        if (LORA && q == NP - 1) {
            // xf[wn], xf[wn + 2] with a compile-time register index: both candidates are named, the scalar wn selects
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 xs = wn ? xf[0][2 * h + 1] : xf[0][2 * h];
                accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lf[h], xs, accl[0], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0 from here on

    unsigned bo = 0;
    for (int s = 0; s < nk; ++s) {
        const unsigned nb = bo ^ BUF;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            // phase q: the unit needed NP + 1 phases from now = the unit of phase (q + 1) % NP, in tile s + 1 (its buffer nb) or,
            // when q is the last phase, the phase-0 units of tile s + 2 (this buffer, whose X / block 0-1 rows phase 0 gave up)
            if (q == 0) read_x(bo);
            read_w(bo, q);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < NP) {
                issue_unit(1 + q + 1, nb);
            } else {
                issue_unit(0, bo);
                issue_unit(1, bo);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
            mfmas(q);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        bo = nb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();

    // NI = 4 (128 x 256): a wave's 128 columns lie on one side of vt_col0 (a multiple of 128), so the fused q|k|v projection can
    // write its V third head-transposed from here as well (FEAT bit 1); the odd-NI tiles keep the trimmed epilogue
    gemm_epilogue<MI, NI, MODE, LORA, NW, WN, ((NI == 4 && MODE == 0) ? 1 : 0) | ((LORA && MODE == 0 && NI <= 4) ? 16 : 0)>(p, smem, acc, accl, ln_mean, ln_rstd, ln_on, tile_m, tile_n, ks_id, wave, wm, wn);
}

}  // namespace

namespace slh_gemm_detail {

int launch_gemm8p(const GemmArgs& a, int mode, int ni_b, hipStream_t s) {
    const int grid = a.tiles_m * a.tiles_n * (a.splitk > 1 ? a.splitk : 1);
#define SLH_LAUNCH_B(NI_)                                                                                           \
    if (a.lora_down) {                                                                                              \
        if (mode == 0) slh_launch<gemm8pb_kernel<1, NI_, 0, true>>(grid, 512, s, a, "gemm8pb_kernel<1, %d, 0, true>", NI_);       \
        else slh_launch<gemm8pb_kernel<1, NI_, 1, true>>(grid, 512, s, a, "gemm8pb_kernel<1, %d, 1, true>", NI_);                 \
    } else if (mode == 0) slh_launch<gemm8pb_kernel<1, NI_, 0, false>>(grid, 512, s, a, "gemm8pb_kernel<1, %d, 0, false>", NI_);   \
    else slh_launch<gemm8pb_kernel<1, NI_, 1, false>>(grid, 512, s, a, "gemm8pb_kernel<1, %d, 1, false>", NI_)
    if (ni_b == 0) {
        if (mode == 0) slh_launch<gemm8p_kernel<0, false>>(grid, 512, s, a, "gemm8p_kernel<0, false>");
        else slh_launch<gemm8p_kernel<1, false>>(grid, 512, s, a, "gemm8p_kernel<1, false>");
    } else if (ni_b == 3) { SLH_LAUNCH_B(3); }
    else if (ni_b == 4) { SLH_LAUNCH_B(4); }
    else { SLH_LAUNCH_B(5); }
#undef SLH_LAUNCH_B
    SLH_LAUNCH_CHECK("slh_gemm (ping-pong K loop)");
    return 0;
}

}  // namespace slh_gemm_detail
