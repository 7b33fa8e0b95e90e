// Small HBM-bound operators around the UNet: timestep embedding, conv_in, elementwise helpers, CFG + DDIM
// step, guidance loss, flat AdamW.  gfx950 only.  bf16 rounding points follow the reference's tensor ops so
// that these kernels can be checked bit-for-bit against the oracle.
#include "common.h"
#include "../../include/sliders_hip.h"

namespace {

// ---- Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin], fp32 math -------------
__global__ void tembed_kernel(const slh_tembed_desc d) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = d.n_vals * d.dim;
    if (idx >= d.nb * per) return;
    const int b = idx / per;
    const int rem = idx - b * per;
    const int j = rem / d.dim;
    const int i = rem - j * d.dim;
    const int half = d.dim / 2;
    const int fi = i < half ? i : i - half;
    const float exponent = (-9.210340371976184f * (float)fi) / (float)half;  // -ln(10000) * i / half
    const float arg = d.vals[b * d.n_vals + j] * expf(exponent);
    const float v = i < half ? cosf(arg) : sinf(arg);
    ((__bf16*)d.out)[(long)b * d.ldo + d.col0 + rem] = (__bf16)v;
}

// ---- conv_in: NCHW (B,cin,H,W) -> pixel-major [B*H*W][cout], 3x3 pad 1 --------------------------------
// weights [cout][9*cin] are staged once per block into LDS as [k][cout] so a thread's 8 output channels are one
// 16-byte LDS read per (tap, ci); block = (256/nchunk) pixels x nchunk chunks, 8 pixel groups per block.
constexpr int CONVIN_ITERS = 8;
__global__ void conv_in_kernel(const slh_convin_desc d, int nchunk, int ppb) {
    extern __shared__ __attribute__((aligned(16))) char cin_smem[];
    __bf16* sw = (__bf16*)cin_smem;                     // [9*cin][cout]
    const int K = 9 * d.cin;
    for (int i = threadIdx.x; i < K * d.cout; i += blockDim.x) {
        const int co = i / K, k = i - co * K;
        sw[k * d.cout + co] = ((const __bf16*)d.w)[i];
    }
    __syncthreads();
    const int chunk = threadIdx.x % nchunk, pl = threadIdx.x / nchunk;
    const int hw = d.h * d.wd;
    const long npix = (long)d.batch * hw;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = d.bias ? (float)((const __bf16*)d.bias)[chunk * 8 + e] : 0.f;
    for (int it = 0; it < CONVIN_ITERS; ++it) {
        const long pix = ((long)blockIdx.x * CONVIN_ITERS + it) * ppb + pl;
        if (pix >= npix) break;
        const int b = (int)(pix / hw);
        const int rem = (int)(pix - (long)b * hw);
        const int oy = rem / d.wd, ox = rem - oy * d.wd;
        const __bf16* x = (const __bf16*)d.x + (long)b * d.cin * hw;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias[e];
        if (d.cin <= 4) {
            // the pixel's 9 x cin inputs are requested together, unconditionally (clamped coordinates, masked afterwards): as
            // 36 loads each behind the bounds test they were 36 serial round trips per pixel (hipcc waits at every join)
            __bf16 xin[9][4];
            bool tin[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
                tin[tap] = iy >= 0 && iy < d.h && ix >= 0 && ix < d.wd;
                const int cy = iy < 0 ? 0 : (iy >= d.h ? d.h - 1 : iy), cx = ix < 0 ? 0 : (ix >= d.wd ? d.wd - 1 : ix);
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) xin[tap][ci] = x[(long)(ci < d.cin ? ci : 0) * hw + cy * d.wd + cx];
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (!tin[tap]) continue;
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    if (ci >= d.cin) break;
                    const float xv = (float)xin[tap][ci];
                    const bf16x8 wv = *(const bf16x8*)(sw + (tap * d.cin + ci) * d.cout + chunk * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += xv * (float)wv[e];
                }
            }
        } else {
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
                if (iy < 0 || iy >= d.h || ix < 0 || ix >= d.wd) continue;
                for (int ci = 0; ci < d.cin; ++ci) {
                    const float xv = (float)x[(long)ci * hw + iy * d.wd + ix];
                    const bf16x8 wv = *(const bf16x8*)(sw + (tap * d.cin + ci) * d.cout + chunk * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += xv * (float)wv[e];
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
        *(bf16x8*)((__bf16*)d.y + pix * d.ldy + chunk * 8) = o;
    }
}

// ---- elementwise ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ew_kernel(const slh_ew_desc d) {
    const int nchunk = d.C / 8;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)d.M * nchunk) return;
    const int chunk = (int)(gid % nchunk);
    const long m = gid / nchunk;
    const int c = chunk * 8;
    const __bf16* A = (const __bf16*)d.a;
    const __bf16* B = (const __bf16*)d.b;
    __bf16* O = (__bf16*)d.out;
    if (d.op == SLH_EW_COPY) {
        *(bf16x8*)(O + m * d.ldo + c) = *(const bf16x8*)(A + m * d.lda + c);
    } else if (d.op == SLH_EW_ADD) {
        const bf16x8 x = *(const bf16x8*)(A + m * d.lda + c);
        const bf16x8 y = *(const bf16x8*)(B + m * d.ldb + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)x[e] + (float)y[e]);
        *(bf16x8*)(O + m * d.ldo + c) = o;
    } else if (d.op == SLH_EW_GEGLU_FWD || d.op == SLH_EW_GEGLU_BWD) {
        // a = proj output [M][2C] in the blocked layout: 64-column blocks [32 values | 32 gates]
        const int jb = c >> 5, j = c & 31;
        const __bf16* pa = A + m * d.lda + jb * 64 + j;
        const bf16x8 hv = *(const bf16x8*)pa;
        const bf16x8 gv = *(const bf16x8*)(pa + 32);
        if (d.op == SLH_EW_GEGLU_FWD) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)hv[e] * round_bf16(gelu_erf_fast_f((float)gv[e])));
            *(bf16x8*)(O + m * d.ldo + c) = o;
        } else {
            const bf16x8 dy = *(const bf16x8*)(B + m * d.ldb + c);
            bf16x8 dh, dg;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = (float)gv[e], hh = (float)hv[e], dd = (float)dy[e];
                const float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752f));
                const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
                dh[e] = (__bf16)(dd * round_bf16(g * cdf));
                dg[e] = (__bf16)(round_bf16(dd * hh) * (cdf + g * pdf));
            }
            __bf16* po = O + m * d.ldo + jb * 64 + j;
            *(bf16x8*)po = dh;
            *(bf16x8*)(po + 32) = dg;
        }
    } else if (d.op == SLH_EW_UPSAMPLE_BWD) {
        // out pixel (b, y, x) of an h x w image (w = d.iarg, hw = d.iarg2) sums the 2x2 block of a
        const int w = d.iarg, hw = d.iarg2;
        const long b = m / hw;
        const int rem = (int)(m - b * hw);
        const int y = rem / w, x = rem - y * w;
        const long base = (b * 4 * hw + (long)(2 * y) * (2 * w) + 2 * x);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long row = base + (q >> 1) * (2 * w) + (q & 1);
            const bf16x8 v = *(const bf16x8*)(A + row * d.lda + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
        *(bf16x8*)(O + m * d.ldo + c) = o;
    }
}

// per-sample column sums: out[b][c] (fp32, +=) = sum over the sample's rows of a[row][c]
__global__ __launch_bounds__(256) void colsum_kernel(const slh_ew_desc d) {
    __shared__ float red[8][33][8];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cx) * 8;
    const int hw = d.iarg2;            // rows per sample
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 512, r1 = min(hw, r0 + 512);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c < d.C) {
        for (int r = r0 + ry; r < r1; r += 8) {
            const bf16x8 v = *(const bf16x8*)((const __bf16*)d.a + ((long)b * hw + r) * d.lda + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ry][cx][e] = acc[e];
    __syncthreads();
    {
        const int ccx = threadIdx.x >> 3, e = threadIdx.x & 7;
        const int cc = (blockIdx.x * 32 + ccx) * 8 + e;
        if (cc < d.C) {
            float t = 0.f;
#pragma unroll
            for (int y = 0; y < 8; ++y) t += red[y][ccx][e];
            atomicAdd((float*)d.out + (long)b * d.ldo + cc, t);
        }
    }
}

// ---- CFG combine + DDIM step ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const slh_cfg_ddim_desc d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)d.nb * d.chw;
    if (i >= n) return;
    const __bf16* eps = (const __bf16*)d.eps;
    const float u = (float)eps[i];
    const float t = d.eps_text ? (float)((const __bf16*)d.eps_text)[i] : (float)eps[n + i];
    // train_util.py:166-169: uncond + g * (text - uncond), every tensor op rounds to bf16
    const float diff = round_bf16(t - u);
    const float scaled = round_bf16(d.guidance * diff);
    const float e = round_bf16(u + scaled);
    if (!d.do_step) { ((__bf16*)d.out)[i] = (__bf16)e; return; }
    const float x = (float)((const __bf16*)d.x)[i];
    if (d.v_prediction) {
        // DDIMScheduler.step, eta = 0, v prediction: every scalar*tensor and tensor+-tensor rounds to bf16
        const float x0 = round_bf16(round_bf16(d.c_sqrt_alpha_t * x) - round_bf16(d.c_sqrt_beta_t * e));
        const float pe = round_bf16(round_bf16(d.c_sqrt_alpha_t * e) + round_bf16(d.c_sqrt_beta_t * x));
        const float dirv = round_bf16(d.c_dir * pe);
        const __bf16 resv = (__bf16)(round_bf16(d.c_sqrt_alpha_prev * x0) + dirv);
        ((__bf16*)d.out)[i] = resv;
        if (d.out2) ((__bf16*)d.out2)[i] = resv;
        return;
    }
    // DDIMScheduler.step, eta = 0, epsilon prediction, fp32 0-dim scalars times bf16 tensors
    const float r1 = round_bf16(d.c_sqrt_beta_t * e);
    const float r2 = round_bf16(x - r1);
    const float x0 = round_bf16(r2 * d.c_inv_sqrt_alpha_t);  // torch CUDA divides by a CPU scalar as a * (1/b)
    const float dir = round_bf16(d.c_dir * e);
    const float r3 = round_bf16(d.c_sqrt_alpha_prev * x0);
    const __bf16 res = (__bf16)(r3 + dir);
    ((__bf16*)d.out)[i] = res;
    if (d.out2) ((__bf16*)d.out2)[i] = res;
}

// ---- guidance loss (prompt_util.py:108-148) --------------------------------------------------------------
// One workgroup, fixed summation order: the loss value is bit-reproducible (a grid of workgroups adding their partial
// sums with an fp32 atomic gave a different last digit from run to run).  n is 65536 per sample: a few microseconds, once
// per training iteration.
__global__ __launch_bounds__(1024) void loss_kernel(const slh_loss_desc d) {
    float sq = 0.f;
    for (int i = threadIdx.x; i < d.n; i += 1024) {
        const float tg = (float)((const __bf16*)d.target)[i];
        const float po = (float)((const __bf16*)d.positive)[i];
        const float ne = (float)((const __bf16*)d.neutral)[i];
        const float un = (float)((const __bf16*)d.uncond)[i];
        const float d1 = round_bf16(po - un);
        const float d2 = round_bf16(d.guidance * d1);
        const float y = round_bf16(d.erase ? ne - d2 : ne + d2);
        const float diff = tg - y;
        sq += diff * diff;
        const __bf16 gq = (__bf16)((2.0f / (float)d.n) * diff);
        if (d.dtarget) ((__bf16*)d.dtarget)[i] = gq;
        if (d.dtarget_pix) {
            const int per = d.nch * d.hw;
            const int b = i / per, rem = i - b * per;
            const int ch = rem / d.hw, px = rem - ch * d.hw;
            d.dtarget_pix[((long)b * d.hw + px) * d.nch + ch] = (float)gq;
        }
    }
    sq = wave_sum(sq);
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w];
        *d.loss += t / (float)d.n;
    }
}

// ---- AdamW over the flat LoRA buffer: torch.optim.AdamW single-tensor op order, bf16 state ------------
__global__ __launch_bounds__(256) void adamw_kernel(const slh_adamw_desc d, float decay, float bc2_sqrt,
                                                    float step_size, float w1, float beta2, float w2, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    __bf16* P = (__bf16*)d.param; __bf16* M1 = (__bf16*)d.exp_avg; __bf16* M2 = (__bf16*)d.exp_avg_sq;
    const float g = round_bf16(d.grad[i] * d.grad_scale);     // param.grad is bf16 in the reference
    float p = round_bf16((float)P[i] * decay);                  // param.mul_(1 - lr * wd)
    float m = (float)M1[i];
    m = round_bf16(m + w1 * (g - m));                           // exp_avg.lerp_(grad, 1 - beta1)
    float v = round_bf16((float)M2[i] * beta2);                 // exp_avg_sq.mul_(beta2)
    v = round_bf16(v + w2 * (g * g));                           // .addcmul_(grad, grad, value = 1 - beta2)
    float den = round_bf16(sqrtf(v));                           // exp_avg_sq.sqrt()
    den = round_bf16(den / bc2_sqrt);                           //   / bias_correction2_sqrt
    den = round_bf16(den + eps);                                //   .add_(eps)
    p = round_bf16(p + (-step_size) * (m / den));               // param.addcdiv_(exp_avg, denom, value = -step_size)
    P[i] = (__bf16)p; M1[i] = (__bf16)m; M2[i] = (__bf16)v;
}

// fp32 adapter state (slh_adamw_desc.f32_state): torch.optim.AdamW's foreach path on fp32 CUDA tensors - every foreach op is its own
// kernel there, so every line below is one fp32 rounding; inside lerp / addcmul / addcdiv the multiply-add is contracted to an fma
// exactly as hipcc contracts it in ATen's functors (self + weight * diff, input + value * (t1 * t2), input + value * (t1 / t2)).
__global__ __launch_bounds__(256) void adamw_f32_kernel(const slh_adamw_desc d, float decay, float bc2_sqrt,
                                                        float neg_step_size, float w1, float beta2, float w2, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    float* P = (float*)d.param; float* M1 = (float*)d.exp_avg; float* M2 = (float*)d.exp_avg_sq;
    const float g = d.grad_scale == 1.0f ? d.grad[i] : d.grad[i] * d.grad_scale;
    float p = P[i] * decay;                                     // _foreach_mul_(params, 1 - lr * wd)
    float m = M1[i];
    m = __builtin_fmaf(w1, g - m, m);                           // _foreach_lerp_(exp_avgs, grads, 1 - beta1)   (weight < 0.5 branch)
    float v = M2[i] * beta2;                                    // _foreach_mul_(exp_avg_sqs, beta2)
    v = __builtin_fmaf(w2, g * g, v);                           // _foreach_addcmul_(exp_avg_sqs, grads, grads, 1 - beta2)
    float den = __builtin_sqrtf(v);                             // _foreach_sqrt
    den = den / bc2_sqrt;                                       // _foreach_div_(., bias_correction2_sqrt)
    den = den + eps;                                            // _foreach_add_(., eps)
    p = __builtin_fmaf(neg_step_size, m / den, p);              // _foreach_addcdiv_(params, exp_avgs, denom, -step_size)
    P[i] = p; M1[i] = m; M2[i] = v;
    ((__bf16*)d.param_lo)[i] = (__bf16)p;
}

// ---- Lion over the flat LoRA buffer: lion_pytorch 0.1.2 update_fn op order, bf16 state -------------------
__global__ __launch_bounds__(256) void lion_kernel(const slh_lion_desc d, float decay, float b1, float w1, float b2,
                                                   float w2, float neg_lr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    __bf16* P = (__bf16*)d.param; __bf16* M = (__bf16*)d.exp_avg;
    const float g = round_bf16(d.grad[i] * d.grad_scale);     // param.grad is bf16 in the reference
    float p = round_bf16((float)P[i] * decay);                  // p.data.mul_(1 - lr * wd)
    const float m0 = (float)M[i];
    float u = round_bf16(m0 * b1);                              // exp_avg.clone().mul_(beta1)
    u = round_bf16(u + w1 * g);                                 //   .add(grad, alpha = 1 - beta1)
    const float sg = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);    //   .sign_()
    p = round_bf16(p + neg_lr * sg);                            // p.add_(update, alpha = -lr)
    float m = round_bf16(m0 * b2);                              // exp_avg.mul_(beta2)
    m = round_bf16(m + w2 * g);                                 //   .add_(grad, alpha = 1 - beta2)
    P[i] = (__bf16)p; M[i] = (__bf16)m;
}

}  // namespace

extern "C" int slh_timestep_embed(const slh_tembed_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->vals && d->out && d->dim % 2 == 0 && d->nb > 0 && d->n_vals > 0, "slh_timestep_embed: bad desc");
    const int total = d->nb * d->n_vals * d->dim;
    hipLaunchKernelGGL(tembed_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_timestep_embed");
    return 0;
}

extern "C" int slh_conv_in(const slh_convin_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->w && d->y, "slh_conv_in: null pointer");
    SLH_CHECK(d->cout % 8 == 0 && d->ldy % 8 == 0 && d->cin > 0 && d->cin <= 16, "slh_conv_in: bad shape");
    const int nchunk = d->cout / 8;
    SLH_CHECK(nchunk <= 256 && 9 * d->cin * d->cout * 2 <= 64 * 1024, "slh_conv_in: weights must fit 64 KB of LDS");
    const int ppb = 256 / nchunk;
    const long npix = (long)d->batch * d->h * d->wd;
    const long per_block = (long)ppb * CONVIN_ITERS;
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)((npix + per_block - 1) / per_block)), dim3(nchunk * ppb),
                       9 * d->cin * d->cout * 2, (hipStream_t)stream, *d, nchunk, ppb);
    SLH_LAUNCH_CHECK("slh_conv_in");
    return 0;
}

extern "C" int slh_elementwise(const slh_ew_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->a && d->out, "slh_elementwise: null pointer");
    SLH_CHECK(d->C % 8 == 0 && d->lda % 8 == 0 && d->ldo % 4 == 0, "slh_elementwise: alignment");
    hipStream_t s = (hipStream_t)stream;
    if (d->op == SLH_EW_COLSUM) {
        SLH_CHECK(d->iarg2 > 0 && d->M % d->iarg2 == 0, "slh_elementwise: colsum rows per sample");
        dim3 grid((d->C / 8 + 31) / 32, (d->iarg2 + 511) / 512, d->M / d->iarg2);
        hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, *d);
    } else {
        if (d->op == SLH_EW_ADD || d->op == SLH_EW_GEGLU_BWD) SLH_CHECK(d->b && d->ldb % 8 == 0, "slh_elementwise: operand b");
        if (d->op == SLH_EW_GEGLU_FWD || d->op == SLH_EW_GEGLU_BWD) SLH_CHECK(d->C % 32 == 0, "slh_elementwise: geglu C");
        if (d->op == SLH_EW_UPSAMPLE_BWD) SLH_CHECK(d->iarg > 0 && d->iarg2 > 0, "slh_elementwise: upsample dims");
        const long total = (long)d->M * (d->C / 8);
        hipLaunchKernelGGL(ew_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *d);
    }
    SLH_LAUNCH_CHECK("slh_elementwise");
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void gather16_kernel(const unsigned short* src, const int* idx, unsigned short* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[i];
    out[i] = j < 0 ? (unsigned short)0 : src[j];
}
}  // namespace

extern "C" int slh_gather16(const slh_gather16_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->src && d->idx && d->out && d->n > 0, "slh_gather16: null pointer / empty");
    hipLaunchKernelGGL(gather16_kernel, dim3((unsigned)((d->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)d->src, d->idx, (unsigned short*)d->out, (long)d->n);
    SLH_LAUNCH_CHECK("slh_gather16");
    return 0;
}

extern "C" int slh_cfg_ddim(const slh_cfg_ddim_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->eps && d->out && (d->x || !d->do_step), "slh_cfg_ddim: null pointer");
    const long n = (long)d->nb * d->chw;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_cfg_ddim");
    return 0;
}

extern "C" int slh_guidance_loss(const slh_loss_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->target && d->positive && d->neutral && d->uncond && d->loss, "slh_guidance_loss: null pointer");
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_guidance_loss");
    return 0;
}

extern "C" int slh_adamw(const slh_adamw_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->param && d->exp_avg && d->exp_avg_sq && d->grad && d->step >= 1, "slh_adamw: bad desc");
    // scalar prep in double like torch's python-side math, then cast to the fp32 opmath type
    const double bc1 = 1.0 - pow(d->beta1, (double)d->step);
    const double bc2 = 1.0 - pow(d->beta2, (double)d->step);
    const float step_size = (float)(d->lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float decay = (float)(1.0 - d->lr * d->weight_decay);
    if (d->f32_state) {
        SLH_CHECK(d->param_lo && ((uintptr_t)d->param & 3) == 0, "slh_adamw: f32_state needs param_lo (the bf16 copy the kernels read)");
        hipLaunchKernelGGL(adamw_f32_kernel, dim3((unsigned)((d->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d,
                           decay, bc2_sqrt, -step_size, (float)(1.0 - d->beta1), (float)d->beta2, (float)(1.0 - d->beta2), (float)d->eps);
        SLH_LAUNCH_CHECK("slh_adamw (fp32 state)");
        return 0;
    }
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((d->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d,
                       decay, bc2_sqrt, step_size, (float)(1.0 - d->beta1), (float)d->beta2, (float)(1.0 - d->beta2),
                       (float)d->eps);
    SLH_LAUNCH_CHECK("slh_adamw");
    return 0;
}

extern "C" int slh_lion(const slh_lion_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->param && d->exp_avg && d->grad && d->n > 0, "slh_lion: bad desc");
    // scalars are python floats in the package: computed in double, then cast to the fp32 opmath type
    hipLaunchKernelGGL(lion_kernel, dim3((unsigned)((d->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d,
                       (float)(1.0 - d->lr * d->weight_decay), (float)d->beta1, (float)(1.0 - d->beta1), (float)d->beta2,
                       (float)(1.0 - d->beta2), (float)(-d->lr));
    SLH_LAUNCH_CHECK("slh_lion");
    return 0;
}
