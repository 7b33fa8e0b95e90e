// fp32 kernels of the image-slider path: AutoencoderKL ENCODER + posterior sample + add_noise on the GPU.  gfx950 only.
//
// Reference: trainscripts/imagesliders/train_util.py:200-235 `get_noisy_image` -
//     init_latents = vae.encode(image).latent_dist.sample(); init_latents *= vae.config.scaling_factor
//     init_latents = scheduler.add_noise(init_latents, noise, timestep)
// with the VAE in fp32 (trainscripts/imagesliders/train_lora-scale-xl.py:96).  The arithmetic lives in diffusers
// (AutoencoderKL.encode -> Encoder -> DownEncoderBlock2D / UNetMidBlock2D / Attention); here it is:
//   slh_sgemm         every 3x3 / 1x1 convolution and Linear of the encoder as ONE fp32 implicit-GEMM kernel on the
//                     exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak = the fp32 vector rate; a
//                     VALU kernel reaches about a third of it), pixel-major fp32 activations [B*H*W][C]
//   slh_gn32_*        GroupNorm(32, eps 1e-6) (+SiLU) in fp32, statistics by fp32 atomics like the bf16 kernels
//   slh_softmax32     row softmax of the 4096 x 4096 mid-block attention scores (one head of 512 channels)
//   slh_vae_conv_in   3 -> 128 channels 3x3 (K = 27 is not a GEMM)
//   slh_vae_moments   conv_out 512 -> 8 (3x3) fused with quant_conv 8 -> 8 (1x1): one wave per latent pixel
//   slh_vae_sample    mean + exp(0.5 clamp(logvar)) * n1, * scaling_factor, add_noise with n2 -> NCHW bf16 (+fp32)
// The encoder runs twice per image-slider iteration (~1.1 TFLOP per 512x512 image): far from the hot loop's cost,
// so these kernels are written for exact fp32 arithmetic first and MFMA-bound simplicity second.
#include "common.h"
#include <type_traits>
#include "../../include/sliders_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f4;

// ---------------------------------------------------------------------------------------------------------------
// C[M][N] = X[M][K] . W[N][K]^T (+ bias) (+ residual), fp32.  256 threads = 4 waves (2 x 2), tile 128 x 128, K step 16.
// LDS tiles are k-major ([16][128] floats) so that a fragment read (lane -> row lane&31, k = 2*step + lane>>5) is two
// conflict-free 128-byte rows; global loads are float4 along k, one tile row per thread, next K step prefetched into
// registers while the current one is multiplied.
// MFMA roles: A = W rows (n), B = X rows (m)  =>  a lane owns output row m = lane&31 and 4 consecutive columns n per
// accumulator quad: one 16-byte store per quad.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SBM = 128, SBN = 128, SBK = 16;

__global__ __launch_bounds__(256) void sgemm_kernel(const slh_sgemm_desc d) {
    __shared__ float sX[SBK][SBM];
    __shared__ float sW[SBK][SBN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (d.N + SBN - 1) / SBN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * SBM, n0 = tile_n * SBN;
    // loader geometry: thread -> tile row r (0..127), k quads kq and kq+2
    const int r = tid & 127, kq = tid >> 7;
    int xm = m0 + r;
    const bool xvalid = xm < d.M;
    xm = xvalid ? xm : d.M - 1;
    int wnr = n0 + r;
    const bool wvalid = wnr < d.N;
    wnr = wvalid ? wnr : d.N - 1;
    const float* X = (const float*)d.x;
    const float* W = (const float*)d.w;
    int xb = 0, xoy = 0, xox = 0;
    if (d.mode == 1) {
        const int hw = d.ho * d.wo;
        xb = xm / hw;
        const int rem = xm - xb * hw;
        xoy = rem / d.wo;
        xox = rem - xoy * d.wo;
    }
    const int cin = d.cin;
    auto load_x = [&](int k0, f4& a, f4& b) {
        const float* src;
        bool ok = xvalid;
        if (d.mode == 0) {
            src = X + (long)xm * d.ldx + k0;
        } else {
            const int tap = k0 / cin, c0 = k0 - tap * cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = xoy * d.stride + ky - d.pad, ix = xox * d.stride + kx - d.pad;
            const int sh = d.upsample ? 1 : 0;         // nearest-2x upsampled source (Upsample2D): read pixel (iy>>1, ix>>1)
            ok = ok && iy >= 0 && iy < (d.hs << sh) && ix >= 0 && ix < (d.ws << sh);
            src = X + (((long)xb * d.hs + (iy >> sh)) * d.ws + (ix >> sh)) * d.ldx + c0;
        }
        // unconditional loads from a selected address (see sgemm_bf16x3_kernel: a load behind a condition costs a vmcnt(0))
        const float* sa = ok ? src + 4 * kq : (const float*)slh_zero_page;
        const float* sb = ok ? src + 4 * (kq + 2) : (const float*)slh_zero_page;
        a = *(const f4*)sa;
        b = *(const f4*)sb;
    };
    auto load_w = [&](int k0, f4& a, f4& b) {
        const float* src = W + (long)wnr * d.ldw + k0;
        a = *(const f4*)(wvalid ? src + 4 * kq : (const float*)slh_zero_page);
        b = *(const f4*)(wvalid ? src + 4 * (kq + 2) : (const float*)slh_zero_page);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f4 xa, xb4, wa, wb;
    load_x(0, xa, xb4);
    load_w(0, wa, wb);
    const int lrow = lane & 31, lhi = lane >> 5;
    for (int k0 = 0; k0 < d.K; k0 += SBK) {
        __syncthreads();                       // previous step's fragment reads are done
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sX[4 * kq + e][r] = xa[e];
            sX[4 * (kq + 2) + e][r] = xb4[e];
            sW[4 * kq + e][r] = wa[e];
            sW[4 * (kq + 2) + e][r] = wb[e];
        }
        __syncthreads();
        if (k0 + SBK < d.K) {                  // next step's global loads fly under the MFMAs below
            load_x(k0 + SBK, xa, xb4);
            load_w(k0 + SBK, wa, wb);
        }
#pragma unroll
        for (int ks = 0; ks < SBK / 2; ++ks) {
            float xf[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[i] = sX[2 * ks + lhi][wm * 64 + i * 32 + lrow];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = sW[2 * ks + lhi][wn * 64 + j * 32 + lrow];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j][e] = C[m = m0 + wm*64 + i*32 + lrow][n = n0 + wn*64 + j*32 + (e&3) + 8*(e>>2) + 4*lhi]
    const float* bias = (const float*)d.bias;
    const float* res = (const float*)d.residual;
    float* C = (float*)d.c;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + lrow;
        if (m >= d.M) continue;
        const float rowb = (bias && d.bias_per_row) ? bias[m] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + j * 32 + q * 8 + lhi * 4;
                if (n >= d.N) continue;
                f4 v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                v *= d.alpha;
                if (bias) {
                    if (d.bias_per_row) v += rowb;
                    else v += *(const f4*)(bias + n);
                }
                if (res) v += *(const f4*)(res + (long)m * d.ldr + n);
                *(f4*)(C + (long)m * d.ldc + n) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same product with every fp32 operand split into two bf16 halves, x = hi + lo (hi = bf16(x), lo = bf16(x - hi), 16
// mantissa bits kept), and  x.w ~= hi.hi + hi.lo + lo.hi  on the bf16 matrix pipe with fp32 accumulation: three
// v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each) per 16 k - 5.3x fewer
// matrix cycles at a relative product error of ~2^-16.  (The reference's fp32 VAE convolutions run in TF32 - 10 mantissa
// bits - on its own hardware: torch.backends.cudnn.allow_tf32 defaults to True.)  slh_sgemm_desc.split_bf16 selects it.
// Tile 128 x 128 x 32; LDS holds four [128 rows][32 bf16] images (X hi/lo, W hi/lo), 64-byte rows whose 16-byte slot s is
// stored at s ^ f(row), f(row) = ((row >> 1) ^ (row >> 4)) & 3: conflict-free for the ds_write_b128 of the staging pass
// (8 consecutive rows per cycle) and for the ds_read_b128 lane groups of the fragment reads.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TBK = 32;

__device__ __forceinline__ int split_slot(int row, int slot) { return slot ^ (((row >> 1) ^ (row >> 4)) & 3); }

// TM = 128 (4 waves as 2 x 2, 64 x 64 per wave) or 64 (32 x 64 per wave: twice the workgroups for the 64 x 64-pixel layers,
// M = 4096, that would leave half the chip idle).  The LDS images are double-buffered: the global loads of step k+1 are
// requested before the MFMAs of step k and split / stored into the other buffer after them - one barrier per step.
template <int TM>
__global__ __launch_bounds__(256, 2) void sgemm_bf16x3_kernel(const slh_sgemm_desc d) {
    constexpr int MI = TM / 64;                                  // 32-row blocks per wave along M
    constexpr int XIMG = TM * 64, WIMG = SBN * 64;               // bytes of one bf16 image
    constexpr int BUF = 2 * XIMG + 2 * WIMG;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (d.N + SBN - 1) / SBN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * SBN;
    // loader geometry: 8 consecutive lanes read the 128 contiguous bytes (32 floats) one tile row contributes to a K step,
    // so a load instruction touches 8 cache lines (a lane-per-row mapping touches 64 and is bound by the address pipe);
    // a thread handles k quad k4 of rows r8 + 32*e
    constexpr int XE = TM / 32;                                  // rows per thread: X
    const int r8 = tid >> 3, k4 = tid & 7;
    const float* X = (const float*)d.x;
    const float* W = (const float*)d.w;
    const int cin = d.cin;
    bool xok[XE];
    long xoff[XE];                                               // dense: element offset of the row
    int xb[XE], xoy[XE], xox[XE];                                // conv: sample / output pixel of the row
#pragma unroll
    for (int e = 0; e < XE; ++e) {
        int xm = m0 + r8 + 32 * e;
        xok[e] = xm < d.M;
        xm = xok[e] ? xm : d.M - 1;
        xoff[e] = (long)xm * d.ldx;
        xb[e] = xoy[e] = xox[e] = 0;
        if (d.mode == 1) {
            const int hw = d.ho * d.wo;
            xb[e] = xm / hw;
            const int rem = xm - xb[e] * hw;
            xoy[e] = rem / d.wo;
            xox[e] = rem - xoy[e] * d.wo;
        }
    }
    bool wok[4];
    long woff[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int wnr = n0 + r8 + 32 * e;
        wok[e] = wnr < d.N;
        woff[e] = (long)(wok[e] ? wnr : d.N - 1) * d.ldw;
    }
    // Every load is UNCONDITIONAL, from a selected address (the zero page for rows / taps that do not exist): a load behind a
    // condition makes hipcc branch around it and wait vmcnt(0) at the join - inside the K loop that drained the three-step
    // prefetch on every step (the loop then ran at the memory latency: 7100 cycles per step pair against ~1500 of MFMA).
    const float* zero4 = (const float*)slh_zero_page;
    auto load_x = [&](int k0, f4* v) {
        if (d.mode == 0) {
#pragma unroll
            for (int e = 0; e < XE; ++e) v[e] = *(const f4*)(xok[e] ? X + xoff[e] + k0 + 4 * k4 : zero4);
        } else {
            const int tap = k0 / cin, c0 = k0 - tap * cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int sh = d.upsample ? 1 : 0;
#pragma unroll
            for (int e = 0; e < XE; ++e) {
                const int iy = xoy[e] * d.stride + ky - d.pad, ix = xox[e] * d.stride + kx - d.pad;
                const bool ok = xok[e] && iy >= 0 && iy < (d.hs << sh) && ix >= 0 && ix < (d.ws << sh);
                v[e] = *(const f4*)(ok ? X + (((long)xb[e] * d.hs + (iy >> sh)) * d.ws + (ix >> sh)) * d.ldx + c0 + 4 * k4 : zero4);
            }
        }
    };
    auto load_w = [&](int k0, f4* v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = *(const f4*)(wok[e] ? W + woff[e] + k0 + 4 * k4 : zero4);
    };
    // 4 floats -> 8 bytes of bf16 high parts and 8 bytes of bf16 remainders (half k4 & 1 of 16-byte slot k4 >> 1)
    auto split_store4 = [&](const f4& a, char* hi_img, char* lo_img, int row) {
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 xh = (__bf16)a[e];
            hi[e] = xh;
            lo[e] = (__bf16)(a[e] - (float)xh);
        }
        const int off = row * 64 + (split_slot(row, k4 >> 1) << 4) + ((k4 & 1) << 3);
        *(bf16x4*)(hi_img + off) = hi;
        *(bf16x4*)(lo_img + off) = lo;
    };
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // Global loads run THREE K steps ahead of the MFMAs in three rotating register sets (a single step of MFMAs is ~0.3 us,
    // a first-touch load from HBM / the other XCDs' writes ~2 us; with one step of prefetch the loop ran at the memory
    // latency, 200 TF/s).  Invariant at the start of step k: set k % 3 is free (step k is in LDS), sets (k+1) % 3 and
    // (k+2) % 3 hold steps k+1, k+2 in flight.
    f4 xs[3][XE], ws[3][4];
    auto store_stage = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        char* base = smem + buf * BUF;
#pragma unroll
        for (int e = 0; e < XE; ++e) split_store4(xs[SET][e], base, base + XIMG, r8 + 32 * e);
#pragma unroll
        for (int e = 0; e < 4; ++e) split_store4(ws[SET][e], base + 2 * XIMG, base + 2 * XIMG + WIMG, r8 + 32 * e);
    };
    const int lrow = lane & 31, lhi = lane >> 5;
    const int nk = d.K / TBK;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    load_x(0, xs[0]);
    load_w(0, ws[0]);
    { const int k1 = nk > 1 ? 1 : 0; load_x(k1 * TBK, xs[1]); load_w(k1 * TBK, ws[1]); }
    { const int k2 = nk > 2 ? 2 : nk - 1; load_x(k2 * TBK, xs[2]); load_w(k2 * TBK, ws[2]); }
    store_stage(I0{}, 0);
    __syncthreads();
    int cur = 0;
    auto step = [&](auto set_c, auto next_c, const int k) {
        constexpr int SET = decltype(set_c)::value;
        {   // into the set step k just vacated.  ALWAYS issued (the last steps re-request the final K tile): a conditional
            // issue makes the number of loads in flight path-dependent and hipcc then waits for all of them
            const int kn = k + 3 < nk ? k + 3 : nk - 1;
            load_x(kn * TBK, xs[SET]);
            load_w(kn * TBK, ws[SET]);
        }
        const char* base = smem + cur * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xh[MI], xl[MI], wh[2], wl[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * (32 * MI) + i * 32 + lrow;
                const int off = row * 64 + (split_slot(row, 2 * ks + lhi) << 4);
                xh[i] = *(const bf16x8*)(base + off);
                xl[i] = *(const bf16x8*)(base + XIMG + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + lrow;
                const int off = row * 64 + (split_slot(row, 2 * ks + lhi) << 4);
                wh[j] = *(const bf16x8*)(base + 2 * XIMG + off);
                wl[j] = *(const bf16x8*)(base + 2 * XIMG + WIMG + off);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (d.split_bf16 & 4) { asm volatile("" ::"v"(wl[j]), "v"(xh[i]), "v"(wh[j]), "v"(xl[i])); continue; }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[j], xh[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[j], xl[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[j], xh[i], acc[i][j], 0, 0, 0);
                }
        }
        if (k + 1 < nk && !(d.split_bf16 & 2)) store_stage(next_c, cur ^ 1);      // the other buffer: last read before the previous barrier
        __syncthreads();
        cur ^= 1;
    };
    int k = 0;
    for (; k + 2 < nk; k += 3) {          // whole triples: straight-line, the in-flight load count is the same on every path
        step(I0{}, I1{}, k);
        step(I1{}, I2{}, k + 1);
        step(I2{}, I0{}, k + 2);
    }
    if (k < nk) step(I0{}, I1{}, k);
    if (k + 1 < nk) step(I1{}, I2{}, k + 1);
    // acc[i][j][e] = C[m = m0 + wm*32*MI + i*32 + lrow][n = n0 + wn*64 + j*32 + (e&3) + 8*(e>>2) + 4*lhi]
    const float* bias = (const float*)d.bias;
    const float* res = (const float*)d.residual;
    float* C = (float*)d.c;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (32 * MI) + i * 32 + lrow;
        if (m >= d.M) continue;
        const float rowb = (bias && d.bias_per_row) ? bias[m] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + j * 32 + q * 8 + lhi * 4;
                if (n >= d.N) continue;
                f4 v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                v *= d.alpha;
                if (bias) {
                    if (d.bias_per_row) v += rowb;
                    else v += *(const f4*)(bias + n);
                }
                if (res) v += *(const f4*)(res + (long)m * d.ldr + n);
                *(f4*)(C + (long)m * d.ldc + n) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm fp32.  stats: grid (row blocks, batch); each block strides over its rows with one float4 column chunk per
// thread, reduces per group in LDS in a fixed order and publishes one pair per (block, group); the last block to arrive
// combines them in index order and writes (mean, rstd) - bit-reproducible, see common.h.  apply: y = xhat*gamma+beta (+SiLU).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn32_stats_kernel(const slh_gn32_desc d, int rows_per_block, int lpg, int row_blocks) {
    extern __shared__ __attribute__((aligned(16))) float gn_lds[];
    const int tid = threadIdx.x;
    const int nchunk = d.C / 4, cg = d.C / d.groups;      // nchunk is a power of two <= 256 (checked on the host)
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(d.hw, r0 + rows_per_block);
    const float* X = (const float*)d.x;
    const int chunk = tid % nchunk, rl = tid / nchunk, rpi = 256 / nchunk;
    const int c = chunk * 4;
    const int g0 = c / cg;                     // cg is a multiple of 4: a float4 never straddles groups
    const float k = X[(long)b * d.hw * d.ldx + g0 * cg];   // shift: the group's first element (see gn_stats_kernel, norm.hip)
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    for (int rr = r0 + rl; rr < r1; rr += rpi) {
        const f4 v = *(const f4*)(X + ((long)b * d.hw + rr) * d.ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float f = v[e] - k; s[e] += f; q[e] += f * f; }
    }
    float S, Q;
    const bool owner = gn_block_reduce<4>(gn_lds, d.C, cg, d.groups, rpi, lpg, chunk, rl, s, q, S, Q);
    const int ncl = (row_blocks + GN_CLUSTER - 1) / GN_CLUSTER;
    double Sd, Qd;
    const bool fin = gn_two_level_reduce(d.partial + (long)b * (row_blocks + ncl) * d.groups * 2, d.ticket + (long)b * (1 + ncl),
                                         row_blocks, blockIdx.x, d.groups, lpg, owner, S, Q, (int*)(gn_lds + 2 * rpi * d.C), Sd, Qd);
    const int g = tid / lpg;
    if (fin) {
        const double n = (double)d.hw * (double)cg;
        const double kg = (double)X[(long)b * d.hw * d.ldx + g * cg];
        const double m = Sd / n;
        const double var = fmax(Qd / n - m * m, 0.0);
        d.stats[((long)b * d.groups + g) * 2] = (float)(kg + m);
        d.stats[((long)b * d.groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)d.eps));
    }
}

__global__ __launch_bounds__(256) void gn32_apply_kernel(const slh_gn32_desc d) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 per thread
    const int nchunk = d.C / 4, cg = d.C / d.groups;
    const long total = (long)d.batch * d.hw * nchunk;
    if (idx >= total) return;
    const long row = idx / nchunk;
    const int c = (int)(idx - row * nchunk) * 4;
    const int b = (int)(row / d.hw);
    const int g = c / cg;
    const float mean = d.stats[((long)b * d.groups + g) * 2];
    const float rstd = d.stats[((long)b * d.groups + g) * 2 + 1];
    const f4 v = *(const f4*)((const float*)d.x + row * d.ldx + c);
    const f4 gm = *(const f4*)((const float*)d.gamma + c), bt = *(const f4*)((const float*)d.beta + c);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float y = (v[e] - mean) * rstd * gm[e] + bt[e];
        if (d.act == 1) y = y / (1.0f + expf(-y));
        o[e] = y;
    }
    *(f4*)((float*)d.y + row * d.ldy + c) = o;
}

// row softmax in place, fp32, one workgroup per row
__global__ __launch_bounds__(256) void softmax32_kernel(float* x, int rows, int cols, long ld) {
    __shared__ float red[8];
    float* row = x + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -3.0e38f;
    for (int c = tid * 4; c < cols; c += 1024) {
        const f4 v = *(const f4*)(row + c);
        mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = tid * 4; c < cols; c += 1024) {
        f4 v = *(const f4*)(row + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = expf(v[e] - mx); s += v[e]; }
        *(f4*)(row + c) = v;
    }
    s = wave_sum(s);
    __syncthreads();
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int c = tid * 4; c < cols; c += 1024) {
        f4 v = *(const f4*)(row + c);
        v *= inv;
        *(f4*)(row + c) = v;
    }
}

// conv_in: image [B][H*W][cin] fp32 (pixel-major; cin = 3: encoder input in [-1,1], cin = 4: decoder latents)
// -> [B*H*W][cout] fp32, 3x3 pad 1.  w: [cout][3(ky)][3(kx)][cin] fp32.  One thread = one pixel x 4 output channels.
__global__ __launch_bounds__(256) void vae_conv_in_kernel(const float* img, const float* w, const float* bias, float* y,
                                                          int B, int H, int Wd, int cout, int cin) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = cout / 4;
    const long total = (long)B * H * Wd * nq;
    if (idx >= total) return;
    const long pix = idx / nq;
    const int co = (int)(idx - pix * nq) * 4;
    const int b = (int)(pix / ((long)H * Wd));
    const int rem = (int)(pix - (long)b * H * Wd);
    const int oy = rem / Wd, ox = rem - oy * Wd;
    float acc[4] = {bias[co], bias[co + 1], bias[co + 2], bias[co + 3]};
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + kx - 1;
            if (ix < 0 || ix >= Wd) continue;
            const float* p = img + (((long)b * H + iy) * Wd + ix) * cin;
            const float x0 = p[0], x1 = p[1], x2 = p[2], x3 = cin > 3 ? p[3] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* wp = w + ((long)(co + e) * 9 + ky * 3 + kx) * cin;
                acc[e] += x0 * wp[0] + x1 * wp[1] + x2 * wp[2] + (cin > 3 ? x3 * wp[3] : 0.f);
            }
        }
    }
    *(f4*)(y + pix * cout + co) = f4{acc[0], acc[1], acc[2], acc[3]};
}

// conv_out (C -> 8, 3x3 pad 1) + quant_conv (8 -> 8, 1x1): one wave per latent pixel, lanes split the K = 9*C products
__global__ __launch_bounds__(256) void vae_moments_kernel(const float* x, const float* w, const float* bias, const float* qw,
                                                          const float* qb, float* out, int B, int H, int Wd, int C) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long)B * H * Wd) return;
    const int b = (int)(pix / ((long)H * Wd));
    const int rem = (int)(pix - (long)b * H * Wd);
    const int oy = rem / Wd, ox = rem - oy * Wd;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= Wd) continue;
        const float* p = x + (((long)b * H + iy) * Wd + ix) * C;
        for (int c = lane * 4; c < C; c += 256) {
            const f4 v = *(const f4*)(p + c);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const f4 wv = *(const f4*)(w + ((long)o * 9 + tap) * C + c);
                acc[o] += v[0] * wv[0] + v[1] * wv[1] + v[2] * wv[2] + v[3] * wv[3];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = wave_sum(acc[o]) + bias[o];
    if (lane < 8) {
        float q = qb[lane];
#pragma unroll
        for (int o = 0; o < 8; ++o) q += qw[lane * 8 + o] * acc[o];
        out[pix * 8 + lane] = q;
    }
}

// posterior sample + scaling + add_noise: moments [B*HW][8] (mean 0..3 | logvar 4..7) -> NCHW outputs
__global__ __launch_bounds__(256) void vae_sample_kernel(const slh_vae_sample_desc d) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)d.batch * 4 * d.hw;
    if (idx >= n) return;
    const int b = (int)(idx / (4L * d.hw));
    const int rem = (int)(idx - (long)b * 4 * d.hw);
    const int c = rem / d.hw, pix = rem - c * d.hw;
    const float* m = d.moments + ((long)b * d.hw + pix) * 8;
    const float mean = m[c];
    const float logvar = fminf(fmaxf(m[4 + c], -30.f), 20.f);
    const float stdv = expf(0.5f * logvar);
    const float z = (mean + stdv * d.post_noise[idx]) * d.scaling;
    if (d.latent_f32) d.latent_f32[idx] = z;
    const float noisy = d.sqrt_alpha * z + d.sqrt_one_minus_alpha * d.noise[idx];
    if (d.noisy_f32) d.noisy_f32[idx] = noisy;
    if (d.noisy_bf16) ((__bf16*)d.noisy_bf16)[idx] = (__bf16)noisy;
}

// decoder input: z NCHW [B][4][hw] (fp32 or bf16) -> (z * inv_scaling) -> post_quant_conv 1x1 (4 -> 4) -> pixel-major [B*hw][4]
__global__ __launch_bounds__(256) void vae_post_quant_kernel(const void* z, int z_bf16, const float* qw, const float* qb, float* y,
                                                             int B, int hw, float inv_scaling) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * hw) return;
    const int b = (int)(idx / hw), pix = (int)(idx - (long)b * hw);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long o = ((long)b * 4 + c) * hw + pix;
        v[c] = (z_bf16 ? (float)((const __bf16*)z)[o] : ((const float*)z)[o]) * inv_scaling;
    }
    f4 out;
#pragma unroll
    for (int o = 0; o < 4; ++o) out[o] = qb[o] + qw[o * 4] * v[0] + qw[o * 4 + 1] * v[1] + qw[o * 4 + 2] * v[2] + qw[o * 4 + 3] * v[3];
    *(f4*)(y + idx * 4) = out;
}

}  // namespace

extern "C" int slh_vae_post_quant(const slh_vae_conv_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->qw && d->qb && d->y, "slh_vae_post_quant: bad descriptor");
    const long n = (long)d->batch * d->h * d->wd;
    hipLaunchKernelGGL(vae_post_quant_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d->x,
                       d->cin /* 1: z is bf16 */, (const float*)d->qw, (const float*)d->qb, (float*)d->y, d->batch, d->h * d->wd,
                       d->inv_scaling);
    SLH_LAUNCH_CHECK("slh_vae_post_quant");
    return 0;
}

extern "C" int slh_sgemm(const slh_sgemm_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->w && d->c, "slh_sgemm: null pointer");
    SLH_CHECK(d->M > 0 && d->N > 0 && d->K > 0 && d->K % 16 == 0 && d->N % 4 == 0, "slh_sgemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    SLH_CHECK(d->ldx % 4 == 0 && d->ldw % 4 == 0 && d->ldc % 4 == 0 && (!d->residual || d->ldr % 4 == 0), "slh_sgemm: leading dims");
    if (d->mode == 1) {
        SLH_CHECK(d->cin > 0 && d->cin % 16 == 0 && d->K == 9 * d->cin, "slh_sgemm: conv needs Cin %% 16 == 0 and K = 9*Cin");
        SLH_CHECK((d->stride == 1 || d->stride == 2) && (d->pad == 0 || d->pad == 1), "slh_sgemm: stride / pad");
        SLH_CHECK(d->M == d->batch * d->ho * d->wo, "slh_sgemm: conv M mismatch");
    } else {
        SLH_CHECK(d->mode == 0, "slh_sgemm: bad mode");
    }
    const int tiles = ((d->M + SBM - 1) / SBM) * ((d->N + SBN - 1) / SBN);
    if ((d->split_bf16 & 1) && d->K % TBK == 0 && (d->mode == 0 || d->cin % TBK == 0)) {
        if (tiles >= 256)
            hipLaunchKernelGGL(sgemm_bf16x3_kernel<128>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, *d);
        else        // few 128-row tiles (the 64 x 64-pixel layers): 64-row tiles fill the chip
            hipLaunchKernelGGL(sgemm_bf16x3_kernel<64>, dim3(((d->M + 63) / 64) * ((d->N + SBN - 1) / SBN)), dim3(256), 0,
                               (hipStream_t)stream, *d);
    } else {
        hipLaunchKernelGGL(sgemm_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, *d);
    }
    SLH_LAUNCH_CHECK("slh_sgemm");
    return 0;
}

static int gn32_check(const slh_gn32_desc* d, const char* who) {
    SLH_CHECK(d && d->x && d->stats, "%s: null pointer", who);
    SLH_CHECK(d->groups > 0 && d->groups <= 64 && d->C % d->groups == 0 && (d->C / d->groups) % 4 == 0 && d->C <= 1024 &&
                  ((d->C / 4) & (d->C / 4 - 1)) == 0,
              "%s: C=%d groups=%d unsupported (C/4 must be a power of two <= 256, C/groups a multiple of 4)", who, d->C, d->groups);
    SLH_CHECK(d->ldx % 4 == 0, "%s: ldx", who);
    return 0;
}

constexpr int GN32_ROWS_PER_BLOCK = 64;

extern "C" int slh_gn32_row_blocks(int hw) { return hw > 0 ? (hw + GN32_ROWS_PER_BLOCK - 1) / GN32_ROWS_PER_BLOCK : -1; }

extern "C" int slh_gn32_stats(const slh_gn32_desc* d, slh_stream_t stream) {
    if (gn32_check(d, "slh_gn32_stats")) return -1;
    SLH_CHECK(d->partial && d->ticket, "slh_gn32_stats: needs the partial-sum workspace and the zeroed ticket counters");
    const int row_blocks = slh_gn32_row_blocks(d->hw);
    int lpg = 1;
    while (lpg < 16 && 2 * lpg * d->groups <= 256) lpg *= 2;
    const unsigned lds = (unsigned)((2 * 256 * 4 + 4) * sizeof(float));
    hipLaunchKernelGGL(gn32_stats_kernel, dim3(row_blocks, d->batch), dim3(256), lds, (hipStream_t)stream, *d,
                       GN32_ROWS_PER_BLOCK, lpg, row_blocks);
    SLH_LAUNCH_CHECK("slh_gn32_stats");
    return 0;
}

extern "C" int slh_gn32_apply(const slh_gn32_desc* d, slh_stream_t stream) {
    if (gn32_check(d, "slh_gn32_apply")) return -1;
    SLH_CHECK(d->y && d->gamma && d->beta && d->ldy % 4 == 0, "slh_gn32_apply: null pointer / ldy");
    const long total = (long)d->batch * d->hw * (d->C / 4);
    hipLaunchKernelGGL(gn32_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_gn32_apply");
    return 0;
}

extern "C" int slh_softmax32(const slh_softmax32_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->rows > 0 && d->cols > 0 && d->cols % 4 == 0 && d->ld % 4 == 0, "slh_softmax32: bad arguments");
    hipLaunchKernelGGL(softmax32_kernel, dim3(d->rows), dim3(256), 0, (hipStream_t)stream, d->x, d->rows, d->cols, (long)d->ld);
    SLH_LAUNCH_CHECK("slh_softmax32");
    return 0;
}

extern "C" int slh_vae_conv_in(const slh_vae_conv_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->w && d->bias && d->y && d->cout % 4 == 0 && (d->cin == 3 || d->cin == 4), "slh_vae_conv_in: bad descriptor");
    const long total = (long)d->batch * d->h * d->wd * (d->cout / 4);
    hipLaunchKernelGGL(vae_conv_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)d->x, (const float*)d->w, (const float*)d->bias, (float*)d->y, d->batch, d->h, d->wd, d->cout,
                       d->cin);
    SLH_LAUNCH_CHECK("slh_vae_conv_in");
    return 0;
}

extern "C" int slh_vae_moments(const slh_vae_conv_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->w && d->bias && d->qw && d->qb && d->y && d->cin % 4 == 0, "slh_vae_moments: bad descriptor");
    const long pix = (long)d->batch * d->h * d->wd;
    hipLaunchKernelGGL(vae_moments_kernel, dim3((unsigned)((pix + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)d->x, (const float*)d->w, (const float*)d->bias, (const float*)d->qw, (const float*)d->qb,
                       (float*)d->y, d->batch, d->h, d->wd, d->cin);
    SLH_LAUNCH_CHECK("slh_vae_moments");
    return 0;
}

extern "C" int slh_vae_sample(const slh_vae_sample_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->moments && d->post_noise && d->noise && (d->noisy_bf16 || d->noisy_f32 || d->latent_f32), "slh_vae_sample: null pointer");
    const long n = (long)d->batch * 4 * d->hw;
    hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_vae_sample");
    return 0;
}
