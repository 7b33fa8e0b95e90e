/*
 * sliders_hip.h - C ABI of libsliders_hip.so, the MI355X (gfx950) UNet-denoise hot path for
 * concept-slider LoRA training.
 *
 * What this boundary replaces in the reference (rohitgandikota/sliders):
 *   the model call  unet(latent_model_input, timestep, encoder_hidden_states=..., added_cond_kwargs=...)
 *   at trainscripts/textsliders/train_util.py:159-163 and :242-247 (diffusers-0.20.2
 *   UNet2DConditionModel.forward, a third-party dependency), the per-layer LoRA patch
 *   LoRAModule.forward at trainscripts/textsliders/lora.py:108-112, the CFG combine at
 *   train_util.py:166-169 / 250-253, scheduler.step at train_util.py:193 / 291, the guidance loss at
 *   trainscripts/textsliders/prompt_util.py:108-148, loss.backward()/optimizer.step() at
 *   trainscripts/textsliders/train_lora_xl.py:345-346.
 *
 * Conventions
 *   - every entry point is asynchronous on the caller-supplied hipStream_t, takes raw device pointers,
 *     never allocates, returns 0 on success and a negative code on a bad descriptor / launch failure
 *     (message via slh_last_error()).
 *   - activations are bf16, "pixel-major": a (B,C,H,W) tensor of the reference is stored as the row-major
 *     matrix [B*H*W][C] with an explicit row stride (ld, in elements).  Latents at the model boundary stay
 *     in the reference's (B,4,H,W) layout.
 *   - fp32 accumulation everywhere; bf16 is rounded once per op output.
 *   - one host thread per GPU/process; no global mutable state besides the last-error string.
 */
#ifndef SLIDERS_HIP_H
#define SLIDERS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* slh_stream_t; /* hipStream_t */

int slh_version(void);
const char* slh_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * slh_gemm: C[M][N] = epilogue( A[M][K] . W[N][K]^T )   bf16 MFMA (32x32x16), fp32 accumulate.
 * Replaces every nn.Linear / 1x1 Conv2d / 3x3 Conv2d of the diffusers UNet together
 * with the LoRA up-projection of LoRAModule.forward (lora.py:108-112): the rank-r product
 * lora_scale * T[M][r] . Bup[N][r]^T is added in the epilogue (T comes from slh_skinny).
 * mode 1 gathers A on the fly from a pixel-major image (implicit-GEMM 3x3 conv, pad 1, stride 1|2,
 * optional nearest-2x upsample read (Upsample2D) or zero-dilated read (dgrad of the stride-2 conv),
 * optional two-source channel concat (up-block skip cat)); K index = tap*Cin + c.
 * ---------------------------------------------------------------------------------------------- */
typedef struct slh_gemm_desc {
    const void* a0;          /* source 0 (bf16) */
    const void* a1;          /* source 1 or NULL: channels [ca0, ca0+ca1) */
    const void* w;           /* bf16 weights: [N][ldw] row-major (w_layout 0) or tile-packed (w_layout 1) */
    const void* bias;        /* [N] bf16 or NULL */
    const void* rowbias;     /* [batch][ld_rowbias] bf16 or NULL: per-sample per-channel add (time embedding) */
    const float* lora_t;     /* [M][ld_t] fp32 or NULL */
    const void* lora_up;     /* [N][4] bf16 */
    const float* lora_scale; /* device scalar: multiplier * alpha / rank */
    const void* residual;    /* [M][ld_res] bf16 or NULL (may alias c) */
    void* c;                 /* [M][ldc] bf16 (N/2 columns when geglu) */
    const void* lora_down;   /* [rank][K] bf16 or NULL: FUSED down-projection, T = A.lora_down^T is computed inside
                                the same kernel (rank = 4*lora_groups <= 12); excludes lora_t */
    float* lora_t_out;       /* optional [M][ld_t] fp32: the fused T, kept for the backward pass */
    int32_t lda0, lda1, ca0, ca1;
    int32_t mode;            /* 0 dense, 1 conv3x3 */
    int32_t batch, hs, ws;   /* conv: source image dims */
    int32_t src_xform;       /* conv: 0 none, 1 nearest upsample x2, 2 zero-dilate x2 */
    int32_t stride;          /* conv: 1 or 2 */
    int32_t ho, wo;          /* conv: output dims */
    int32_t ldw;
    int32_t M, N, K;
    int32_t ld_rowbias, rows_per_sample;
    int32_t ld_t, lora_groups; /* N/lora_groups columns share one rank-4 slice of T */
    int32_t ld_res, ldc;
    int32_t geglu;           /* 1: out[:, j] = a_j * gelu(g_j); W rows pre-permuted in 64-row blocks [32 a | 32 g].
                                2: backward form, for the backward-data product of the Linear BEHIND a GEGLU: the result d(ff)
                                [M][N] leaves as d(proj) [M][2N] in proj's blocked column order, computed with the forward's
                                pre-activation geglu_pre (slh_elementwise GEGLU_BWD fused; bare product only).
                                3: as 1 with W rows pre-permuted in 32-row blocks [16 a | 16 g] (any tile; no geglu_pre) */
    int32_t tile;            /* 0 auto; else (S<<16)|(WM<<12)|(stages<<8)|(MI<<4)|NI, MI,NI in {1,2}, WM in {0|2, 4}: WM*2 waves per
                                workgroup, block tile (32*MI*WM) x (64*NI); stages 0|2: double buffer, 3|4: deep LDS ring;
                                S: split-K factor (0|1 none), needs splitk_c32.
                                WM = 8: ping-pong K loops, one 8-wave workgroup per CU (csrc/gemm8p.hip): 0x8042 = 256 x 256
                                (no fused adapter), 0x801<NI> = 128 x 64*NI, NI = 3..5 (geglu 0 | 3 only, no ln_out / vt_out).
                                Bits 20 and up must be zero.  (Round 4 carried a stream-K form under bit 20 whose finishing
                                workgroups waited on flags of other workgroups; it was slower than the plain launch on every
                                shape and could hang two concurrent launches - removed: no kernel of this library waits on
                                another workgroup, profiles/r04_streamk.txt keeps the measurement.) */
    int32_t lora_rank;       /* 0 = 4.  With lora_up_rmajor: total rank 4 | 8 | 12 (T has that many columns) */
    int32_t lora_up_rmajor;  /* 1: lora_up is [rank][N] (= lora_down as stored): backward-data LoRA term */
    int32_t w_layout;        /* 0: w is [N][ldw].  1: frozen weights repacked once at load time into the order the kernel
                                streams them: [ceil(N/64)][K/64] blocks of 64 rows x 64 k (8 KB contiguous, rows past N
                                zero), 16-byte slot s of row r stored at slot s ^ ((r>>1)&7) (the LDS swizzle applied
                                in memory, so every LDS-DMA instruction reads 1 KB of consecutive addresses); ldw unused */
    int32_t reserved_;       /* 0 (ignored; rounds 2-5 read ablation bits from it in probe builds - those builds are gone) */
    float* splitk_c32;       /* split-K workspace or NULL: splitk_slabs slabs of roundup(M, 256) * roundup(N, 128) floats each (the
                                partial tiles are kept whole, in accumulator order), any contents.  With tile bits 16-19 =
                                S > 1 the K range is cut into S slices; each slice's workgroups publish their partial tiles in
                                slab number <slice> (write-through stores - no fp32 atomics, whose arrival-order sums make a
                                pass differ from run to run), take a ticket of their tile (splitk_ticket), and the slice that
                                arrives LAST adds the slabs IN SLICE ORDER and runs the ordinary epilogue - one launch, every
                                epilogue option available, bit-reproducible.  For the few-tile, long-K products (1280-channel
                                3x3 convolutions at 8x8 / 16x16: 10-40 output tiles on 256 CUs, 29 MB of weights each) this
                                is what fills the chip. */
    float* splitk_t32;       /* with lora_down and S > 1: [ceil(N / (64*NI))][2*S][M][ld_t] fp32 - per column tile, two slabs
                                per slice of the adapter's T (size it for NI = 1: ceil(N/64) * 2 * splitk_slabs * M * ld_t) */
    void* vt_out;            /* optional: the columns >= vt_col0 of the result (the V third of a fused q|k|v projection,
                                diffusers Attention.to_v) are written HEAD-TRANSPOSED for slh_attn_fwd instead of into c:
                                vt_out[((b*vt_heads + h)*Dp + d)*vt_ld + t] = C[b*vt_tokens + t][vt_col0 + h*vt_D + d],
                                Dp = 64*ceil(vt_D/64) - exactly what slh_transpose_heads would produce from c, without
                                the extra launch and the round trip of V through HBM.  Needs vt_D % 64 == 0 (no padded
                                rows), vt_col0 % 128 == 0, vt_tokens % 8 == 0, M % 8 == 0; not with geglu */
    int32_t vt_col0, vt_D, vt_heads, vt_tokens, vt_ld;
    int32_t splitk_slabs;    /* slabs splitk_c32 holds (>= the S of tile) */
    /* LayerNorm folded into the products around it (BasicTransformerBlock.norm1/2/3 of the no-grad passes: no LayerNorm
     * launch, no round trip of the normalised tensor through HBM).
     *   producer (the GEMM that writes the tensor LayerNorm reads): ln_out [N/64][M][2] fp32 (chunk-major: a wave's 32 rows
     *     are contiguous for both sides) receives (mean, M2) of every 64-column chunk of every row of the bf16 result;
     *     needs a 128-column tile, no GEGLU / vt_out.
     *   consumer (dense, single source, K = LayerNorm width <= 1280): ln_in = the producer's ln_out, ln_in_chunks = K/64 (K/80
     *     behind a producer on the 64 x 160 tile: equal-sized chunks of either width merge the same way);
     *     w must hold W * gamma, ln_s [N] fp32 its row sums, ln_b [N] fp32 = bias + W . beta (bias must be NULL):
     *     c = rstd_m * (a . w^T - mean_m * ln_s) + ln_b, the row's mean / rstd merged from the chunks in a fixed order. */
    float* ln_out;
    const float* ln_in; const float* ln_s; const float* ln_b;
    int32_t ln_in_chunks;
    float ln_eps;
    void* splitk_ticket;     /* with S > 1: one 64-bit arrival ticket per output tile ([ceil(M/64) * ceil(N/64)] is enough for
                                every tile shape), zero before the first launch, left zero by every launch */
    float* ln_mr_out;        /* with ln_in, optional: [M][2] fp32 receives the merged (mean, rstd) of every row - what
                                slh_layernorm would have left in mean_rstd for slh_layernorm_bwd (training passes) */
    void* geglu_pre;         /* geglu = 1, optional: [M][ld_pre] bf16 RECEIVES proj(x) itself (bf16-rounded, this product's column
                                order = what the same call without geglu writes to c).  geglu = 2: the same array, READ */
    int32_t ld_pre;
    int32_t vt_also_c;       /* with vt_out: 1 = the head-transposed columns are ALSO written row-major into c (training passes
                                keep V, and the backward dO, in both layouts: no separate transpose launch) */
    /* Cross-attention fused behind the query projection (diffusers Attention.to_q of attn2 + the xformers call behind it, no-grad
     * passes, head dim 64): with xa_k the product is Q = a . w^T (bf16-rounded, LayerNorm fold / bias applied first) and c receives
     *   softmax(Q_h K_h^T * xa_scale) V_h   per (sample, head h = columns [64h, 64h + 64))
     * for the xa_tk <= 96 keys of the row's sample (text tokens): every wave of the 128 x 128 ring tile (tile 0x4412, required) owns
     * 32 rows x one head, so scores, softmax and P.V stay in its registers (24 MFMAs behind the K loop; K / V^T of the workgroup's
     * two heads staged once through LDS) and neither Q nor a second launch exists.  xa_k: this layer's keys, [B][xa_tk][xa_ldk]
     * with head h in columns [64h, 64h + 64); xa_vt: this layer's first head inside a head-transposed V
     * [B][xa_vt_heads][64][xa_ldvt >= 128] (slh_transpose_heads layout, key columns >= xa_tk zero); both 16-byte aligned;
     * xa_tq = rows per sample (a multiple of 128).  Excludes residual, row bias, adapters, GEGLU, ln_out, vt_out, split-K. */
    const void* xa_k; const void* xa_vt;
    int32_t xa_tk, xa_tq, xa_ldk, xa_ldvt, xa_vt_heads;
    float xa_scale;
    /* ln_in together with a fused adapter (lora_down; ping-pong tiles 0x8013 / 0x8014 only): lora_down must hold A . gamma (bf16, what
     * slh_lora_ln_fold writes) and the down-projection is normalised like the main product,
     *   T = rstd_m * (a . (A gamma)^T - mean_m * ln_lora_s) + ln_lora_c,   ln_lora_s = row sums of the rounded A gamma, ln_lora_c = A . beta
     * (fp32 [rank] each), ahead of the up-projection - Linear(LayerNorm(x)) + LoRA(LayerNorm(x)) without the LayerNorm launch. */
    const float* ln_lora_s; const float* ln_lora_c;
    /* Optional hint: pf_bytes of frozen weights at pf_ptr (16-byte aligned) that a LATER launch will stream are touched by up to 64
     * extra workgroups of THIS launch when its grid leaves that many CUs idle (the 160-tile products of the M = 2048 level: the
     * touch rides on CUs that have nothing to do and costs no launch, no second stream, no graph edge).  A UNet pass reads every
     * matrix once, so the big ones are HBM-cold at first touch (2048 x 1280 x 5120: 56 us cold, 47 us after a touch).  Ignored
     * when the grid fills the chip or the tile is a ping-pong one. */
    const void* pf_ptr; int64_t pf_bytes;
} slh_gemm_desc;
int slh_gemm(const slh_gemm_desc* d, slh_stream_t stream);
/* (WM<<12)|(MI<<8)|(NI<<4)|mode of the kernel instantiation gemm_kernel<MI,NI,mode,..,WM> slh_gemm would launch for d
 * (used by bench.py to attribute measured time and algorithmic FLOPs to one profiled kernel name). */
int slh_gemm_variant(const slh_gemm_desc* d);
/* Name of the kernel instantiation slh_gemm would launch for d, as rocprofv3 --kernel-trace prints it without namespace and parameter
 * list (e.g. "gemm8pb_kernel<1, 5, 0, false>", "gemm5_kernel<false, 4>"), written to buf (cap >= 16 bytes, NUL-terminated).  All of
 * slh_gemm's checks and its dispatch run; the launch is replaced by a record of the selected template, so the name is exact for
 * every present and future tile.  Needs no device and launches nothing.  0, or the status slh_gemm would return for d.
 * (bench.py / scripts/make_pmc_traffic.py pair in-situ event times and PMC rows by this name.) */
int slh_gemm_kernel_name(const slh_gemm_desc* d, char* buf, int cap);
/* The 64 x 160 tile (tile code bits 12-15 = 5, e.g. 0x5425; csrc/gemm5.hip): 4 waves of 32 x 80 on the 16 x 16 x 32 MFMA, 4-slot LDS
 * ring - the M = 2048, N = 1280 products as 256 workgroups = one full round of the chip.  Dense single-source products with packed
 * weights (w_layout = 1), M % 64 == 0, N % 160 == 0; epilogue: bias, residual, ln_out - whose chunks are then 80 COLUMNS wide:
 * ln_out [N/80][M][2], and the consumer's ln_in_chunks = K / 80.  slh_gemm5_ok(d) = 1 where it can run the descriptor. */
int slh_gemm5_ok(const slh_gemm_desc* d);
/* The tiles of csrc/gemm7.hip (tile code bits 12-15 = 7: 0x7<S><XB><WB>): 128 x 256 (0x7648) and 128 x 160 (0x7645) - four loader waves
 * that issue the ring's LDS-DMA + four compute waves, one per SIMD, each a 64 x (16 WB) register tile of 16 x 16 x 32 MFMAs - and 256 x 320
 * (0x748a) on eight compute waves that stage the ring themselves (GEGLU.proj as one round of 256 workgroups); ring of S half K tiles (32
 * deep).  Dense single-source products with packed weights (w_layout = 1, w 128-byte aligned), M % (32 XB) == 0, N % (32 WB) == 0,
 * K % 64 == 0, K >= 32 S.  Epilogue: bias, residual, ln_in, geglu = 3 on all three; on the 128-row tiles also ln_out (64-column chunks;
 * 80-column chunks on 0x7645), ln_mr_out, vt_out / vt_also_c (vt_col0 % (16 WB) == 0, vt_tokens % 128 == 0) and - 0x7648 only - one fused
 * adapter of 1-3 column groups (forward form, every tile inside one group; with ln_in also its own fold: ln_lora_s / ln_lora_c) and
 * lora_t_out.  No split-K, row bias, external T, geglu 1 / 2, cross-attention.  slh_gemm7_ok(d) = 1 where d->tile can run d. */
int slh_gemm7_ok(const slh_gemm_desc* d);

/* ------------------------------------------------------------------------------------------------
 * slh_skinny: T[M][R] = A[M][K] . Wd[R][K]^T (+bias), R <= 16, same A addressing as slh_gemm.
 * LoRA down-projection lora_down(x) of lora.py:108-112 (Linear: (r,in); Conv: (r,Cin,k,k) repacked
 * [r][tap][Cin]) and the UNet's conv_out (320 -> 4).  Output fp32 [M][ldo] or bf16 NCHW (conv_out).
 * ---------------------------------------------------------------------------------------------- */
typedef struct slh_skinny_desc {
    const void* a0; const void* a1;
    const void* w;           /* [R][K] bf16 */
    const void* bias;        /* [R] bf16 or NULL */
    void* out;
    int32_t lda0, lda1, ca0, ca1;
    int32_t mode, batch, hs, ws, src_xform, stride, ho, wo;
    int32_t M, R, K;
    int32_t ldo;
    int32_t out_kind;        /* 0: fp32 [M][ldo]; 1: bf16 NCHW [batch][R][ho][wo] */
    int32_t w_kmajor;        /* 1: w is [K][4] (R must be 4): U = dY . B with B = lora_up [out][r] as stored */
} slh_skinny_desc;
int slh_skinny(const slh_skinny_desc* d, slh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * slh_gemv: y[b][n] = x'[b][:] . W[n][:] + bias[n] + addend[b][n] + lora,  b < nb <= 8 (weight-streaming).
 * TimestepEmbedding MLPs, add_embedding, and all ResnetBlock2D.time_emb_proj layers batched into one
 * launch (they only depend on emb).  in_act 1 applies SiLU to x first.
 * ---------------------------------------------------------------------------------------------- */
typedef struct slh_gemv_desc {
    const void* x;           /* [nb][ldx] bf16 */
    const void* w;           /* bf16 weights: [N][ldw] row-major (w_layout 0) or tile-packed (w_layout 1) */
    const void* bias;        /* [N] bf16 or NULL */
    const void* addend;      /* [nb][ld_add] bf16 or NULL */
    const float* lora_t;     /* [nb][ld_t] fp32 or NULL */
    const int32_t* lora_tcol;/* [N] first T column of row n (NULL: 0) */
    const void* lora_up;     /* [N][4] bf16 */
    const float* lora_scale;
    void* y;                 /* [nb][ldy] bf16 (out_f32: fp32) */
    int32_t nb, N, K, ldx, ld_add, ld_t, ldy;
    int32_t in_act, out_f32;
} slh_gemv_desc;
int slh_gemv(const slh_gemv_desc* d, slh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU) on pixel-major images; x may be a two-source channel concat.
 * slh_gn_stats leaves (sum, sum of squares) pairs per cluster of row blocks; slh_gn_apply (same descriptor, next on the
 * stream) adds a sample's cluster pairs in index order in every workgroup's prologue, writes stats[b][G] = (mean, rstd) and
 * y = act((x-mean)*rstd*gamma+beta).  No fp32 atomics anywhere: bit-reproducible.
 * diffusers ResnetBlock2D.norm1/norm2 (+nonlinearity), Transformer2DModel.norm, conv_norm_out.
 * ---------------------------------------------------------------------------------------------- */
typedef struct slh_gn_desc {
    const void* x0; const void* x1;
    const void* gamma; const void* beta; /* [C] bf16 */
    float* stats;            /* [batch][groups][2] fp32: (mean, rstd), written by slh_gn_apply (and slh_gn_fused) */
    void* y;                 /* [batch*hw][ldy] bf16 */
    int32_t ldx0, ldx1, c0, c1;
    int32_t batch, hw, groups, ldy;
    float eps;
    int32_t act;             /* 0 none, 1 SiLU */
    /* The statistics are reduced in a fixed order: every workgroup of slh_gn_stats (one row block) writes one shifted (sum, sum of
     * squares) pair per group; the LAST workgroup of a cluster of 16 row blocks to arrive (one ticket level) adds the cluster's
     * pairs in row-block order and leaves the cluster pair; slh_gn_apply adds the cluster pairs (double accumulation).
     * With R = slh_gn_row_blocks(c0+c1, hw, groups) and L = slh_gn_clusters(R): */
    float* partial;          /* [batch][R + L][groups][2] fp32 scratch, any contents; must stay untouched between the two launches */
    /* No aliasing: slh_gn_apply re-reads the statistics' pivot (the first element of every group) from x0 / x1 in every workgroup,
     * so y must be a different buffer (checked) and x must be unchanged between slh_gn_stats and slh_gn_apply. */
    uint32_t* ticket;        /* [batch][1 + L] arrival counters, ZERO before the launch (left zero by it) */
} slh_gn_desc;
int slh_gn_stats(const slh_gn_desc* d, slh_stream_t stream);
int slh_gn_apply(const slh_gn_desc* d, slh_stream_t stream);
/* workgroups per sample of slh_gn_stats / slh_gn_bwd_stats, and the clusters they form; -1 for an unsupported shape */
int slh_gn_row_blocks(int channels, int hw, int groups);
int slh_gn_clusters(int row_blocks);
/* Statistics AND normalisation in one launch; also writes stats for the backward.  partial / ticket unused; y must not alias x.
 * slh_gn_fused_ok(channels, hw, groups) = 0 where it does not apply (use the two launches above),
 *   1: tiny tensors (8x8 latents: a sample's group slab is a few KB) - one workgroup per group set keeps the slab in registers;
 *   2: cache-resident slabs (hw <= 4096, hw * channels / groups <= 81920: the 32x32 level and the 640-channel tensors of the 64x64 level of SDXL at 1024^2) -
 *      S = 1..8 sibling workgroups per (sample, group), each of which reduces the WHOLE slab in the same fixed order (identical
 *      statistics, nothing exchanged, nothing waited for) and normalises its own 1/S of it. */
int slh_gn_fused_ok(int channels, int hw, int groups);
int slh_gn_fused(const slh_gn_desc* d, slh_stream_t stream);

/* GroupNorm backward (dx only: gamma/beta are frozen).  Two launches like the forward:
 * bwd_stats reduces per (b,g) sum(dyhat) and sum(dyhat*xhat) into bstats (fixed order, see slh_gn_desc),
 * bwd_apply writes dx (+= into dx if accumulate).  dy is the gradient of the post-activation output. */
typedef struct slh_gn_bwd_desc {
    const void* x0; const void* x1;
    const void* gamma; const void* beta;
    const float* stats;      /* forward stats (mean, rstd) */
    float* bstats;           /* [batch][groups][2] fp32: (sum dxhat, sum dxhat*xhat), written by slh_gn_bwd_stats */
    const void* dy;          /* [batch*hw][lddy] bf16 */
    void* dx0; void* dx1;    /* gradient w.r.t. source 0 / 1 */
    int32_t ldx0, ldx1, c0, c1;
    int32_t batch, hw, groups, lddy, lddx0, lddx1;
    float eps;
    int32_t act;
    int32_t accumulate0, accumulate1; /* dx += instead of = */
    float* bpartial;         /* slh_gn_bwd_stats: scratch like slh_gn_desc.partial */
    uint32_t* bticket;       /* slh_gn_bwd_stats: zeroed arrival counters like slh_gn_desc.ticket */
} slh_gn_bwd_desc;
int slh_gn_bwd_stats(const slh_gn_bwd_desc* d, slh_stream_t stream);
int slh_gn_bwd_apply(const slh_gn_bwd_desc* d, slh_stream_t stream);

/* LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3), one wave per row. */
typedef struct slh_ln_desc {
    const void* x; const void* gamma; const void* beta;
    void* y;
    float* mean_rstd;        /* [M][2] fp32 or NULL (saved for backward) */
    int32_t M, C, ldx, ldy;
    float eps;
} slh_ln_desc;
int slh_layernorm(const slh_ln_desc* d, slh_stream_t stream);

typedef struct slh_ln_bwd_desc {
    const void* x; const void* gamma; const void* dy;
    const float* mean_rstd;
    void* dx;                /* dx (+)= LN-backward(dy) */
    int32_t M, C, ldx, lddy, lddx;
    int32_t accumulate;
} slh_ln_bwd_desc;
int slh_layernorm_bwd(const slh_ln_bwd_desc* d, slh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attention: O = softmax(Q K^T * scale) V per (sample, head); head dims 64 (SDXL), 40/80/160 (SD-1.x).
 * Q [B][Tq][ldq] (head h at columns h*D..), K [B][Tk][ldk], VT [B][H][Dp][ldvt] (V transposed by
 * slh_transpose_heads, Dp = 64*ceil(D/64)), O [B][Tq][ldo]; lse [B][H][Tq] fp32 (log2 domain) or NULL.
 * Replaces Attention + XFormersAttnProcessor (train_lora.py:68).
 * ---------------------------------------------------------------------------------------------- */
typedef struct slh_attn_desc {
    const void* q; const void* k; const void* vt;
    void* o;
    float* lse;
    int32_t B, H, Tq, Tk, ldq, ldk, ldvt, ldo;
    float scale;
    int32_t D;               /* head dim: 0/64 (SDXL) or 40 / 80 / 160 (SD-1.x); head h = columns [h*D, (h+1)*D);
                                VT is [B][H][64*ceil(D/64)][ldvt] with zero rows for d >= D */
    int32_t vt_batch_heads;  /* 0 = H.  Otherwise vt points into a wider [B][vt_batch_heads][..][ldvt] array (one transposed
                                V for the cross-attention of every transformer block) at this layer's first head */
    int32_t reserved_;
    /* Optional weight touch (as slh_gemm_desc.pf_*): pf_bytes bytes at pf_ptr (16-byte aligned) - the packed weights of a product a
     * few launches later - are streamed through the memory-side cache by up to 64 extra workgroups dispatched behind the launch's own.
     * A hint: taken only by the key-split form (the self-attention of the 32 x 32 level, whose workgroups all fit the chip at once, so
     * the touch runs beside them); slh_attn_fwd_carries_touch(d) = 1 where it would be.  NULL / 0 = none. */
    const void* pf_ptr; int64_t pf_bytes;
} slh_attn_desc;
int slh_attn_fwd(const slh_attn_desc* d, slh_stream_t stream);
int slh_attn_fwd_carries_touch(const slh_attn_desc* d);

/* src [B][T][ld] columns [h*64,(h+1)*64) -> dst [B][H][64][ldt] (columns >= T zero-filled up to ldt) */
typedef struct slh_transpose_desc {
    const void* src; void* dst;
    int32_t B, H, T, ld, ldt;
    int32_t D;               /* head dim (0 = 64); dst is [B][H][64*ceil(D/64)][ldt] */
} slh_transpose_desc;
int slh_transpose_heads(const slh_transpose_desc* d, slh_stream_t stream);

/* Attention backward.  dQ (and dK/dV when self-attention) given dO, Q, K, V, O, lse.
 * All operands [B][T][ld] pixel-major with heads in columns; transposed copies are built internally
 * in the caller-provided workspace.  need_dkv = 0 for cross-attention (text K/V carry no gradient). */
typedef struct slh_attn_bwd_desc {
    const void* q; const void* k; const void* v; const void* o; const void* d_o;
    const void* kt;          /* [B][H][64][ldkt] K transposed (slh_transpose_heads) */
    const void* qt;          /* [B][H][64][ldqt] Q transposed   (need_dkv only) */
    const void* dot;         /* [B][H][64][ldqt] dO transposed  (need_dkv only) */
    const float* lse;        /* [B][H][Tq] from the forward, padded by 64 floats */
    float* delta;            /* [B][H][Tq] fp32 workspace, padded by 64 floats */
    void* dq; void* dk; void* dv;
    int32_t B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddo, ldkt, ldqt, lddq, lddk, lddv;
    float scale;
    int32_t need_dkv;
    int32_t D;               /* head dim (0 = 64); kt / qt / dot use the padded [B][H][64*ceil(D/64)][ld] layout */
    int32_t pad_;
} slh_attn_bwd_desc;
int slh_attn_bwd(const slh_attn_bwd_desc* d, slh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * small ops
 * ---------------------------------------------------------------------------------------------- */
/* sinusoidal timestep embedding (Timesteps(dim, flip_sin_to_cos=True, freq_shift=0)), fp32 math -> bf16.
 * out[b][col0 + j*dim + 0..dim) for each of the n_vals values vals[b][j] (fp32). */
typedef struct slh_tembed_desc {
    const float* vals; void* out;
    int32_t nb, n_vals, dim, ldo, col0;
} slh_tembed_desc;
int slh_timestep_embed(const slh_tembed_desc* d, slh_stream_t stream);

/* conv_in: (B,Cin<=8,H,W) bf16 NCHW -> pixel-major [B*H*W][Cout], 3x3 pad 1, w [Cout][9*Cin] (tap-major) */
typedef struct slh_convin_desc {
    const void* x; const void* w; const void* bias; void* y;
    int32_t batch, cin, h, wd, cout, ldy;
} slh_convin_desc;
int slh_conv_in(const slh_convin_desc* d, slh_stream_t stream);

/* generic elementwise helpers on bf16 matrices [M][C] (8 channels per thread) */
typedef struct slh_ew_desc {
    const void* a; const void* b; void* out;
    int32_t M, C, lda, ldb, ldo;
    int32_t op;              /* see SLH_EW_* */
    int32_t iarg, iarg2;
    float alpha;
    int32_t pad_;
} slh_ew_desc;
enum {
    SLH_EW_COPY = 0,         /* out = a */
    SLH_EW_ADD = 1,          /* out = a + b */
    SLH_EW_GEGLU_FWD = 2,    /* a = GEGLU.proj output [M][2C] in 64-column blocks [32 values | 32 gates] ->
                                out[M][C] = value * gelu(gate)  (training forward keeps the pre-activation) */
    SLH_EW_GEGLU_BWD = 3,    /* a as above, b = d(out) [M][C] -> out = d(proj) [M][2C], same blocked layout */
    SLH_EW_UPSAMPLE_BWD = 4, /* a = d(upsampled) (2h x 2w image) -> out = 2x2 block sums; M = B*h*w, iarg = w, iarg2 = h*w */
    SLH_EW_COLSUM = 5        /* out (fp32 [B][ldo], +=) = per-sample column sums of a; iarg2 = rows per sample */
};
int slh_elementwise(const slh_ew_desc* d, slh_stream_t stream);

/* CFG combine + DDIM step (train_util.py:166-169 + scheduler.step, eta = 0), bf16 rounding points as the
 * reference's tensor ops.  eps [2*nb][chw] (uncond half first), x [nb][chw] -> x_prev; also used with
 * do_step = 0 to produce only the guided epsilon (predict_noise). */
typedef struct slh_cfg_ddim_desc {
    const void* eps; const void* x; void* out;
    void* out2;              /* optional second copy of the result (the CFG pair's duplicated latent input) */
    const void* eps_text;    /* optional: text-half epsilon [nb][chw] when it is not stored right after the uncond half */
    int32_t nb, chw;
    float guidance;
    float c_sqrt_beta_t, c_inv_sqrt_alpha_t, c_sqrt_alpha_prev, c_dir;
    int32_t do_step;
    int32_t v_prediction;    /* 1: the network predicts v (SD-2.x 768-v; model_util.py:126 prediction_type="v_prediction"):
                                x0 = sqrt(a_t) x - sqrt(1-a_t) v, eps = sqrt(a_t) v + sqrt(1-a_t) x, same rounding points
                                as the DDIMScheduler.step tensor ops */
    float c_sqrt_alpha_t;    /* v_prediction only: alpha_prod_t ** 0.5 (fp32) */
    int32_t pad_;
} slh_cfg_ddim_desc;
int slh_cfg_ddim(const slh_cfg_ddim_desc* d, slh_stream_t stream);

/* guidance loss (prompt_util.py:108-148): loss = mean((target - (neutral +- gs*(positive-uncond)))^2);
 * writes loss (fp32 scalar, atomically accumulated: zero it first) and d(loss)/d(target) (bf16). */
typedef struct slh_loss_desc {
    const void* target; const void* positive; const void* neutral; const void* uncond;
    float* loss; void* dtarget;
    float* dtarget_pix;      /* optional: the same gradient as fp32 pixel-major [nb*hw][nch] (backward input) */
    int32_t n; float guidance; int32_t erase;
    int32_t hw, nch;         /* n = nb * nch * hw, NCHW order */
} slh_loss_desc;
int slh_guidance_loss(const slh_loss_desc* d, slh_stream_t stream);

/* LoRA weight gradients: out (fp32, +=) = scale * sum_m Z[m][c] * V[m][r]; Z bf16 with the same addressing
 * as slh_gemm's A (mode 1: column index = tap*Cin + c), V fp32 [M][ldv].  R is 4 or 12. */
typedef struct slh_wgrad_desc {
    const void* z0; const void* z1;
    const float* v;
    float* out;              /* dense: [C][ldo]; conv: [9][C][ldo] */
    const float* scale;      /* device scalar */
    int32_t ldz0, ldz1, c0, c1;
    int32_t mode, batch, hs, ws, src_xform, stride, ho, wo;
    int32_t M, R, ldv, ldo;
    int32_t out_rmajor;      /* 1: out is [R][ldo] (down-weight layout), 0: [C][ldo] (up-weight layout) */
    int32_t vgroup_cols;     /* >0: channel c uses V columns 4*(c/vgroup_cols).. (fused q/k/v up grads) */
    float* slabs;            /* fixed-order reduction over the M splits (bit-reproducible gradients): slh_lora_wgrad_single_blocks(d)
                                slabs of 256 * R floats (contents irrelevant) ... */
    void* tickets;           /* ... and as many uint32 arrival tickets, zeroed once (the last arriver re-arms them).  Both NULL:
                                fp32 atomics (commit in arrival order).  Inside a batch: any non-NULL slabs selects the slab
                                geometry for slh_lora_wgrad_blocks; the workspace itself is the batch's */
} slh_wgrad_desc;
int slh_lora_wgrad(const slh_wgrad_desc* d, slh_stream_t stream);
int slh_lora_wgrad_single_blocks(const slh_wgrad_desc* d);

/* Batched launches: n independent problems of one kind in ONE launch (the backward of a UNet pass has ~380 rank-4
 * weight-gradient reductions and ~210 head transposes of forward activations, each far too small to fill the chip).
 * table is a DEVICE array of n descriptors of the kind's own type, prefix a DEVICE int32[n + 1] running sum of the
 * workgroups each problem needs (slh_*_blocks(descriptor), which also validates the descriptor: -1 + slh_last_error()),
 * total = prefix[n].  A workgroup finds its problem by bisection of prefix. */
typedef struct slh_batch_desc {
    const void* table; const int32_t* prefix;
    int32_t n, total;
    int32_t arg;             /* slh_lora_wgrad_batch: R (4 or 12), the same for every problem of the batch */
    int32_t pad_;
    void* slabs;             /* slh_lora_wgrad_batch: `total` slabs of 256 * R floats + ... */
    void* tickets;           /* ... `total` uint32 tickets (zeroed once): fixed-order reduction, every problem's descriptor built
                                with a non-NULL slabs; both NULL: fp32 atomics */
} slh_batch_desc;
int slh_lora_wgrad_blocks(const slh_wgrad_desc* d);
int slh_lora_wgrad_batch(const slh_batch_desc* d, slh_stream_t stream);
int slh_transpose_heads_blocks(const slh_transpose_desc* d);
int slh_transpose_heads_batch(const slh_batch_desc* d, slh_stream_t stream);

/* out[i] = idx[i] < 0 ? 0 : src[idx[i]] on 16-bit elements (the k-major copies of the adapters' up matrices that the fused
 * backward-data products stream as their third operand: rebuilt from the live parameters at the head of every backward). */
typedef struct slh_gather16_desc { const void* src; const int32_t* idx; void* out; int64_t n; } slh_gather16_desc;
int slh_gather16(const slh_gather16_desc* d, slh_stream_t stream);

/* The adapter side of a LayerNorm folded into an adapter-carrying product (slh_gemm_desc.ln_lora_*): for every item
 *   a_out[r][k] = bf16(a[r][k] * gamma[k]),  s_out[r] = sum_k float(a_out[r][k]),  c_out[r] = sum_k a[r][k] * beta[k]
 * from the LIVE adapter parameters; one launch at the head of a pass covers every folded module (items = a device array). */
typedef struct slh_lora_lnfold_item {
    const void* a; const void* gamma; const void* beta;   /* bf16 [rows][K], [K], [K] */
    void* a_out; float* s_out; float* c_out;               /* bf16 [rows][K], fp32 [rows], fp32 [rows] */
    int32_t rows, K;                                       /* rows <= 16, K % 8 == 0 */
} slh_lora_lnfold_item;
typedef struct slh_lora_lnfold_desc { const slh_lora_lnfold_item* items; int32_t n; int32_t pad_; } slh_lora_lnfold_desc;
int slh_lora_ln_fold(const slh_lora_lnfold_desc* d, slh_stream_t stream);

/* Backward-data term of a 3x3 LoRA down conv (lora.py:82-87): gx[i][c] (+)= scale * sum_{tap,r} U[o][r] *
 * A[r][tap][c] for the output pixels o with o*stride + tap - 1 = i.  U fp32 [batch*ho*wo][ldu], A = lora_down
 * as stored [4][9*cin], gx bf16 pixel-major image of hl x wl. */
typedef struct slh_lora_cdgrad_desc {
    const float* u; const void* a_down; const float* scale; void* gx;
    int32_t batch, hl, wl, ho, wo, stride, cin, ldu, ldgx, accumulate;
} slh_lora_cdgrad_desc;
int slh_lora_conv_dgrad(const slh_lora_cdgrad_desc* d, slh_stream_t stream);

/* LoRA gradients of one ResnetBlock2D.time_emb_proj for one sample: g = d(loss)/d(temb projection) [C] fp32
 * (column sums of the conv1 output gradient), t = that layer's lora_down(silu(emb)) [4], up [C][4], emb [ted]
 * bf16; accumulates d_up [C][4] and d_down [4][ted] (fp32). */
typedef struct slh_temb_lora_bwd_desc {
    const float* g; const float* t; const void* up; const void* emb; float* d_up; float* d_down;
    const float* scale;
    int32_t C, ted;
} slh_temb_lora_bwd_desc;
int slh_temb_lora_bwd(const slh_temb_lora_bwd_desc* d, slh_stream_t stream);

/* flat AdamW over the packed LoRA parameter buffer (bf16 params and moments, torch.optim.AdamW op order
 * and bf16 rounding points; train_util.py:362-363, train_lora_xl.py:346). grads fp32.
 * f32_state = 1 (train.precision: float32 - the reference's weight_dtype applied to the network, train_lora_xl.py:60-61, 84-90): param,
 * exp_avg and exp_avg_sq are fp32 arrays updated with the op order of torch.optim.AdamW on fp32 CUDA tensors (one fp32 rounding per
 * foreach op), the gradient is used unrounded, and param_lo - the bf16 copy every kernel of the UNet pass reads - receives the
 * rounded new parameters in the same launch. */
typedef struct slh_adamw_desc {
    void* param; void* exp_avg; void* exp_avg_sq; const float* grad;
    int64_t n;
    double lr, beta1, beta2, eps, weight_decay;  /* python floats of torch.optim.AdamW */
    int32_t step;            /* 1-based */
    float grad_scale;        /* multiply grads first (1/world_size after all-reduce) */
    void* param_lo;          /* f32_state only: bf16 [n], written */
    int32_t f32_state;       /* 0: bf16 state (default), 1: fp32 state + param_lo */
    int32_t pad_;
} slh_adamw_desc;
int slh_adamw(const slh_adamw_desc* d, slh_stream_t stream);

/* flat Lion over the packed LoRA parameter buffer (train.optimizer "lion": train_util.py:365-368 -> lion_pytorch==0.1.2,
 * requirements.txt:5).  One bf16 moment; op order and bf16 rounding points of that package's update_fn:
 *   p.mul_(1 - lr*wd); update = sign(exp_avg*b1 + (1-b1)*g); p.add_(update, alpha=-lr); exp_avg = exp_avg*b2 + (1-b2)*g */
typedef struct slh_lion_desc {
    void* param; void* exp_avg; const float* grad;
    int64_t n;
    double lr, beta1, beta2, weight_decay;
    float grad_scale;        /* multiply grads first (1/world_size after all-reduce) */
    int32_t pad_;
} slh_lion_desc;
int slh_lion(const slh_lion_desc* d, slh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Image sliders: AutoencoderKL encoder + posterior sample + add_noise in fp32 on the GPU.
 * Replaces trainscripts/imagesliders/train_util.py:200-235 `get_noisy_image` (vae.encode(...).latent_dist.sample()
 * * scaling_factor -> scheduler.add_noise); the VAE arithmetic itself is diffusers' AutoencoderKL.encode.
 * Activations are pixel-major fp32 [B*H*W][C]; 3x3 weights are [Cout][ky][kx][Cin] fp32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct slh_sgemm_desc {
    const void* x;          /* [M][ldx] fp32 (mode 0) or the [batch*hs*ws][ldx] source image of an implicit 3x3 GEMM (mode 1) */
    const void* w;          /* [N][ldw] fp32 */
    const void* bias;       /* [N] (or [M] if bias_per_row) fp32, may be NULL */
    const void* residual;   /* [M][ldr] fp32, may be NULL */
    void* c;                /* [M][ldc] fp32 */
    int32_t ldx, ldw, ldr, ldc;
    int32_t M, N, K;
    int32_t mode;           /* 0 dense, 1 implicit 3x3 (K = 9*cin) */
    int32_t cin, batch, hs, ws, ho, wo, stride, pad;   /* pad 1: symmetric; pad 0 with stride 2: zero pad right/bottom (Downsample2D(padding=0)) */
    int32_t bias_per_row;
    float alpha;            /* C = alpha * (X W^T) + bias + residual */
    int32_t upsample;       /* conv: 1 = the source is read through a nearest-2x upsample (Upsample2D), ho = 2*hs */
    int32_t split_bf16;     /* 0: exact fp32 products (v_mfma_f32_32x32x2_f32).  1: every operand split into two bf16 halves,
                               x.w ~= hi.hi + hi.lo + lo.hi on the bf16 matrix pipe with fp32 accumulation (16 mantissa bits per
                               operand, ~5x fewer matrix cycles); needs K % 32 == 0 (and Cin % 32 == 0), else exact.  Bits 1-3 are
                               profiling ablations (scripts/probe_sgemm.py): 2 skip the LDS staging, 4 skip the MFMAs, 8 skip the
                               global loads after the prologue */
} slh_sgemm_desc;
int slh_sgemm(const slh_sgemm_desc* d, slh_stream_t stream);

typedef struct slh_gn32_desc {
    const void* x; const void* gamma; const void* beta;   /* fp32 */
    float* stats;           /* [batch][groups][2] fp32: (mean, rstd), written by slh_gn32_stats */
    void* y;                /* [batch*hw][ldy] fp32 */
    int32_t ldx, ldy, C, batch, hw, groups;
    float eps;
    int32_t act;            /* 0 none, 1 SiLU */
    float* partial;         /* slh_gn32_stats: [batch][R + L][groups][2] fp32 scratch, R = slh_gn32_row_blocks(hw),
                               L = slh_gn_clusters(R) (see slh_gn_desc) */
    uint32_t* ticket;       /* slh_gn32_stats: [batch][1 + L] arrival counters, zero before the launch */
} slh_gn32_desc;
int slh_gn32_stats(const slh_gn32_desc* d, slh_stream_t stream);
int slh_gn32_apply(const slh_gn32_desc* d, slh_stream_t stream);
int slh_gn32_row_blocks(int hw);

typedef struct slh_softmax32_desc { float* x; int64_t ld; int32_t rows, cols; } slh_softmax32_desc;   /* in place */
int slh_softmax32(const slh_softmax32_desc* d, slh_stream_t stream);

typedef struct slh_vae_conv_desc {
    const void* x;          /* conv_in: image [batch][h*wd][3] fp32; moments: [batch*h*wd][cin] fp32 */
    const void* w; const void* bias;      /* conv_in: [cout][3][3][3]; moments: conv_out [8][3][3][cin], [8] */
    const void* qw; const void* qb;       /* moments only: quant_conv [8][8], [8] */
    void* y;                /* conv_in: [batch*h*wd][cout]; moments: [batch*h*wd][8] = mean | logvar */
    int32_t batch, h, wd, cin, cout;
    float inv_scaling;      /* post_quant only: 1 / vae.config.scaling_factor */
} slh_vae_conv_desc;
int slh_vae_conv_in(const slh_vae_conv_desc* d, slh_stream_t stream);    /* cin 3 (encoder) or 4 (decoder conv_in) */
int slh_vae_moments(const slh_vae_conv_desc* d, slh_stream_t stream);
/* decoder input: x = latents NCHW [batch][4][h*wd] (fp32, or bf16 if cin == 1) -> * inv_scaling -> post_quant_conv (qw [4][4],
 * qb [4]) -> y pixel-major [batch*h*wd][4] fp32.  eval-scripts/generate_images_sd1.py:166-168. */
int slh_vae_post_quant(const slh_vae_conv_desc* d, slh_stream_t stream);

typedef struct slh_vae_sample_desc {
    const float* moments;       /* [batch*hw][8] */
    const float* post_noise;    /* [batch][4][hw] NCHW: the draw of DiagonalGaussianDistribution.sample */
    const float* noise;         /* [batch][4][hw] NCHW: the diffusion noise of add_noise */
    float* latent_f32;          /* optional: scaling_factor * sample, NCHW */
    float* noisy_f32;           /* optional: add_noise result, NCHW fp32 */
    void* noisy_bf16;           /* optional: the same in bf16 (the UNet's input dtype) */
    int32_t batch, hw;
    float scaling, sqrt_alpha, sqrt_one_minus_alpha;
    int32_t pad_;
} slh_vae_sample_desc;
int slh_vae_sample(const slh_vae_sample_desc* d, slh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * program executor: a whole UNet forward / backward is a flat command buffer built once by the host
 * planner (sliders_amd/plan.py) and replayed with ONE call.  Record layout: {int32 opcode; int32
 * nbytes; desc bytes (8-byte aligned)}.
 * ---------------------------------------------------------------------------------------------- */
enum {
    SLH_OP_GEMM = 1, SLH_OP_SKINNY = 2, SLH_OP_GEMV = 3, SLH_OP_GN_STATS = 4, SLH_OP_GN_APPLY = 5,
    SLH_OP_LAYERNORM = 6, SLH_OP_ATTN_FWD = 7, SLH_OP_TRANSPOSE_HEADS = 8, SLH_OP_TEMBED = 9,
    SLH_OP_CONV_IN = 10, SLH_OP_ELEMENTWISE = 11, SLH_OP_CFG_DDIM = 12, SLH_OP_LOSS = 13,
    SLH_OP_WGRAD = 14, SLH_OP_ADAMW = 15, SLH_OP_GN_BWD_STATS = 16, SLH_OP_GN_BWD_APPLY = 17,
    SLH_OP_LAYERNORM_BWD = 18, SLH_OP_ATTN_BWD = 19, SLH_OP_MEMSET = 20, SLH_OP_LORA_CONV_DGRAD = 21,
    SLH_OP_TEMB_LORA_BWD = 22, SLH_OP_SGEMM = 23, SLH_OP_GN32_STATS = 24, SLH_OP_GN32_APPLY = 25, SLH_OP_SOFTMAX32 = 26,
    SLH_OP_VAE_CONV_IN = 27, SLH_OP_VAE_MOMENTS = 28, SLH_OP_VAE_SAMPLE = 29, SLH_OP_VAE_POST_QUANT = 30, SLH_OP_LION = 31,
    SLH_OP_WGRAD_BATCH = 32, SLH_OP_TRANSPOSE_BATCH = 33, SLH_OP_GATHER16 = 34, SLH_OP_GN_FUSED = 35,
    SLH_OP_LORA_LN_FOLD = 36      /* 37 was SLH_OP_PREFETCH (side-stream weight touch: measured slower, removed in round 5) */
};
/* SLH_OP_MEMSET: byte fill by a kernel of this library (not hipMemsetAsync: a captured memset node is a runtime blit whose
 * replays were observed to go wrong on the legacy default stream - see the executor's comment) */
typedef struct slh_memset_desc { void* ptr; int64_t nbytes; int32_t value; int32_t pad; } slh_memset_desc;
int slh_run_program(const void* program, int64_t nbytes, slh_stream_t stream);
/* hipGraph replay of a command buffer: capture records the launches slh_run_program would issue (on a private stream;
 * nothing executes), launch re-submits them as one graph.  The buffer must be static between replays (descriptors are baked
 * into the kernel arguments); everything the reference's loop changes per step (latents, timestep, adapter scale and
 * weights - train_util.py:220-260) is read through device pointers and may change freely.  Replaces the ~1.2k Python
 * op dispatches of one diffusers UNet forward with one submission. */
int slh_graph_capture(const void* program, int64_t nbytes, void** out_graph);
int slh_graph_launch(void* graph, slh_stream_t stream);
int slh_graph_destroy(void* graph);
/* sizeof() of every descriptor, in declaration order above, for binding self-checks */
int slh_desc_sizes(int32_t* out, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif
