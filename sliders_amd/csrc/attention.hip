// Flash-style attention forward for head_dim 64 (every SDXL attention layer), bf16 MFMA 32x32x16,
// fp32 online softmax kept entirely in registers.  gfx950 only.
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

template <int NW>   // waves per workgroup: 4 (128 queries) or 2 (64 queries, used when the grid would not fill the chip)
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const slh_attn_desc p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 8192];
    char* sK = smem;           // [2][64 kv][128 B]
    char* sV = smem + 16384;   // [2][64 d ][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int frow = lane >> 3, fslot = lane & 7;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;

    const __bf16* Q = (const __bf16*)p.q;
    const __bf16* K = (const __bf16*)p.k;
    const __bf16* VT = (const __bf16*)p.vt;

    // Q fragments (B operand): lane holds Q[q][16*ks + 8*hi .. +8]
    int qrow = q0 + lrow;
    const bool qvalid = qrow < p.Tq;
    qrow = qvalid ? qrow : p.Tq - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *(const bf16x8*)(Q + ((long)b * p.Tq + qrow) * p.ldq + h * 64 + ks * 16 + lhi * 8);

    const int nt = (p.Tk + 63) / 64;
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) {
            const int row = (wave + NW * i) * 8 + frow;
            const int ks = fslot ^ ((row >> 1) & 7);
            int kv = t * 64 + row;
            kv = kv < p.Tk ? kv : p.Tk - 1;
            glds16(K + ((long)b * p.Tk + kv) * p.ldk + h * 64 + ks * 8, sK + buf * 8192 + (wave + NW * i) * 1024);
            glds16(VT + (((long)b * p.H + h) * 64 + row) * p.ldvt + t * 64 + ks * 8,
                   sV + buf * 8192 + (wave + NW * i) * 1024);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale * 1.4426950408889634f;
    // permuted key row for the A operand of S^T (swap bits 2 and 3)
    const int prow = (lrow & 3) | (((lrow >> 3) & 1) << 2) | (((lrow >> 2) & 1) << 3) | (lrow & 16);

    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char* cK = sK + (t & 1) * 8192;
        const char* cV = sV + (t & 1) * 8192;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(cK + lds_off(kt * 32 + prow, ks * 2 + lhi));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt], 0, 0, 0);
            }
        }
        // s[kt][r] = S[q = lrow][kv = t*64 + kt*32 + 16*(r>>3) + 8*lhi + (r&7)]
        if (t == nt - 1 && (p.Tk & 63)) {
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
        const float alpha = exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pb[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(s[kt][r] * c - mc);
                psum += pv;
                pb[kt][r >> 3][r & 7] = (__bf16)pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const bf16x8 vf = *(const bf16x8*)(cV + lds_off(dt * 32 + lrow, kstep * 2 + lhi));
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kstep >> 1][kstep & 1], o[dt], 0, 0, 0);
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qvalid) {
        __bf16* O = (__bf16*)p.o + ((long)b * p.Tq + qrow) * p.ldo + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__bf16)(o[dt][qd * 4 + e] * inv);
                *(bf16x4*)(O + dt * 32 + qd * 8 + lhi * 4) = v;
            }
        if (p.lse && lhi == 0) p.lse[((long)b * p.H + h) * p.Tq + qrow] = m_run * c + log2f(l_tot);
    }
}

// src [B][T][ld] columns [h*64,(h+1)*64) -> dst [B][H][64][ldt]; tokens >= T are written as zeros
__global__ __launch_bounds__(256) void transpose_heads_kernel(const slh_transpose_desc p) {
    __shared__ __bf16 tile[64][72];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const __bf16* src = (const __bf16*)p.src;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int tok = idx >> 3, ch = idx & 7;
        bf16x8 v;
        if (t0 + tok < p.T) v = *(const bf16x8*)(src + ((long)b * p.T + t0 + tok) * p.ld + h * 64 + ch * 8);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
        }
        *(bf16x8*)&tile[tok][ch * 8] = v;
    }
    __syncthreads();
    const int d = tid >> 2, tc = (tid & 3) * 16;
    __bf16* dst = (__bf16*)p.dst + (((long)b * p.H + h) * 64 + d) * p.ldt + t0 + tc;
    bf16x8 o0, o1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o0[e] = tile[tc + e][d]; o1[e] = tile[tc + 8 + e][d]; }
    *(bf16x8*)dst = o0;
    *(bf16x8*)(dst + 8) = o1;
}

}  // namespace

extern "C" int slh_attn_fwd(const slh_attn_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->q && d->k && d->vt && d->o, "slh_attn_fwd: null pointer");
    SLH_CHECK(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "slh_attn_fwd: bad shape");
    SLH_CHECK(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 64 == 0 && d->ldo % 4 == 0, "slh_attn_fwd: alignment");
    SLH_CHECK(d->ldvt >= ((d->Tk + 63) / 64) * 64, "slh_attn_fwd: VT must be padded to a multiple of 64 keys");
    const long blocks4 = (long)((d->Tq + 127) / 128) * d->H * d->B;
    if (blocks4 >= 512) {
        hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3((d->Tq + 127) / 128, d->H, d->B), dim3(256), 0, (hipStream_t)stream, *d);
    } else {
        hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((d->Tq + 63) / 64, d->H, d->B), dim3(128), 0, (hipStream_t)stream, *d);
    }
    SLH_LAUNCH_CHECK("slh_attn_fwd");
    return 0;
}

extern "C" int slh_transpose_heads(const slh_transpose_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->src && d->dst, "slh_transpose_heads: null pointer");
    SLH_CHECK(d->ld % 8 == 0 && d->ldt % 64 == 0 && d->ldt >= d->T, "slh_transpose_heads: alignment");
    dim3 grid(d->ldt / 64, d->H, d->B);
    hipLaunchKernelGGL(transpose_heads_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_transpose_heads");
    return 0;
}
