// Skinny (rank-r) products around the LoRA branch of trainscripts/textsliders/lora.py:108-112 and the
// small weight-streaming GEMVs of the time-embedding path.  gfx950 only.  These are HBM/L2-bound: every
// lane moves 16 B per load, reductions use wave shuffles, fp32 accumulation.
//
//   slh_skinny : T[M][R]   = A[M][K] . Wd[R][K]^T          (lora_down(x); also conv_out 320->4)
//   slh_gemv   : y[nb][N]  = x[nb][K] . W[N][K]^T + ...     (TimestepEmbedding, time_emb_proj batch)
//   slh_lora_wgrad : dW[C][R] += s * sum_m Z[m][c] V[m][r]  (LoRA up / down weight gradients)
#include "common.h"
#include "../../include/sliders_hip.h"

namespace {

struct AddrArgs {  // shared A-operand addressing (same meaning as slh_gemm_desc)
    const __bf16* a0; const __bf16* a1;
    int lda0, lda1, ca0, ca1;
    int mode, hs, ws, src_xform, stride, ho, wo;
};

// ------------------------------------------------------------------------------------------------
// skinny: LPR lanes per output row (16: many rows; 64: one wave per row, used when M is small so that the
// grid still covers the chip and the serial K chain per lane stays short), 256 threads per block
// ------------------------------------------------------------------------------------------------
template <int RMAX, int LPR>
__global__ __launch_bounds__(256) void skinny_kernel(AddrArgs A, const __bf16* __restrict__ w,
                                                     const __bf16* __restrict__ bias, void* out,
                                                     int M, int R, int K, int ldo, int out_kind, int kmajor) {
    const int tid = threadIdx.x;
    const int sub = tid & (LPR - 1);
    const int m = blockIdx.x * (256 / LPR) + tid / LPR;
    const bool active = m < M;
    const int mm = active ? m : M - 1;
    float acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = 0.f;
    const int cin = A.ca0 + A.ca1;

    auto fma_chunk = [&](const bf16x8 x, int k) {
        float xf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[e] = (float)x[e];
        if (kmajor) {
            // W given as [K][4] (the lora_up layout [out][r] read as a down-projection of the output gradient)
            if (RMAX == 4) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const bf16x8 wv = *(const bf16x8*)(w + (long)k * 4 + h * 8);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[r] += xf[2 * h] * (float)wv[r];
                        acc[r] += xf[2 * h + 1] * (float)wv[4 + r];
                    }
                }
            }
            return;
        }
        bf16x8 wv[RMAX];        // requested together, unconditionally (rows past R re-read row 0 and are not accumulated)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) wv[r] = *(const bf16x8*)(w + (long)(r < R ? r : 0) * K + k);
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < R) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r] += xf[e] * (float)wv[r][e];
            }
        }
    };

    // The activation chunk of step s+1 is requested before the products of step s (one register of lookahead): the loop is a
    // chain of dependent global round trips otherwise (a row's K / (8 LPR) chunks one after the other).
    if (A.mode == 0) {
        auto src_of = [&](int k) {
            const int kc = k < K ? k : 0;          // column 0 always exists (sub * 8 may lie past K when LPR * 8 > K); never accumulated
            return kc < A.ca0 ? A.a0 + (long)mm * A.lda0 + kc : A.a1 + (long)mm * A.lda1 + (kc - A.ca0);
        };
        bf16x8 cur = *(const bf16x8*)src_of(sub * 8);
        for (int k = sub * 8; k < K; k += LPR * 8) {
            const bf16x8 nxt = *(const bf16x8*)src_of(k + LPR * 8);
            fma_chunk(cur, k);
            cur = nxt;
        }
    } else {
        const int hw = A.ho * A.wo;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int oy = rem / A.wo, ox = rem - oy * A.wo;
        const int sh = A.src_xform ? 1 : 0;
        const int HL = A.hs << sh, WL = A.ws << sh;
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = oy * A.stride + ky - 1, ix = ox * A.stride + kx - 1;
            bool ok = (iy >= 0) & (iy < HL) & (ix >= 0) & (ix < WL);
            if (A.src_xform == 2) ok = ok & (((iy | ix) & 1) == 0);
            if (!ok) continue;
            const long pix = ((long)b * A.hs + (iy >> sh)) * A.ws + (ix >> sh);
            auto src_of = [&](int c) {
                const int cc = c < cin ? c : 0;
                return cc < A.ca0 ? A.a0 + pix * A.lda0 + cc : A.a1 + pix * A.lda1 + (cc - A.ca0);
            };
            bf16x8 cur = *(const bf16x8*)src_of(sub * 8);
            for (int c = sub * 8; c < cin; c += LPR * 8) {
                const bf16x8 nxt = *(const bf16x8*)src_of(c + LPR * 8);
                fma_chunk(cur, tap * cin + c);
                cur = nxt;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, 64);
    }
    if (active && sub == 0) {
        if (out_kind == 0) {
            float* o = (float*)out + (long)m * ldo;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R) o[r] = acc[r] + (bias ? (float)bias[r] : 0.f);
        } else {
            const int hw = A.ho * A.wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            __bf16* o = (__bf16*)out;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R) o[((long)b * R + r) * hw + rem] = (__bf16)(acc[r] + (bias ? (float)bias[r] : 0.f));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemv: one wave per output feature n, all nb <= 8 samples at once
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemv_kernel(const slh_gemv_desc d) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= d.N) return;
    const __bf16* x = (const __bf16*)d.x;
    const __bf16* w = (const __bf16*)d.w + (long)n * d.K;
    float acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = 0.f;
    // all loads of a k step are issued together and unconditionally (samples past nb re-read the last one and are masked): a
    // load behind `if (b < nb)` is branched around by hipcc with a vmcnt(0) at the join - ten serial round trips for K = 1280
    for (int k = lane * 8; k < d.K; k += 512) {
        const bf16x8 wv = *(const bf16x8*)(w + k);
        bf16x8 xv[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) xv[b] = *(const bf16x8*)(x + (long)(b < d.nb ? b : d.nb - 1) * d.ldx + k);
        float wf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[e] = (float)wv[e];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b < d.nb) {          // (the accumulation order of a sample is what it was: no arithmetic on the masked ones)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xe = (float)xv[b][e];
                    if (d.in_act == 1) xe = round_bf16(silu_f(xe));  // reference: nonlinearity(temb) in bf16
                    acc[b] += xe * wf[e];
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = wave_sum(acc[b]);
    if (lane == 0) {
        const float bias = d.bias ? (float)((const __bf16*)d.bias)[n] : 0.f;
        float up[4] = {0.f, 0.f, 0.f, 0.f};
        float ls = 0.f;
        int tcol = 0;
        if (d.lora_t) {
            ls = *d.lora_scale;
            tcol = d.lora_tcol ? d.lora_tcol[n] : 0;
            const bf16x4 u = *(const bf16x4*)((const __bf16*)d.lora_up + (long)n * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) up[r] = (float)u[r];
        }
        for (int b = 0; b < d.nb; ++b) {
            float v = acc[b] + bias;
            if (d.lora_t) {
                const float* t = d.lora_t + (long)b * d.ld_t + tcol;
                v += ls * (t[0] * up[0] + t[1] * up[1] + t[2] * up[2] + t[3] * up[3]);
            }
            if (d.addend) v = round_bf16(v) + (float)((const __bf16*)d.addend)[(long)b * d.ld_add + n];
            if (d.out_f32) ((float*)d.y)[(long)b * d.ldy + n] = v;
            else ((__bf16*)d.y)[(long)b * d.ldy + n] = (__bf16)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LoRA weight gradient: thread owns 8 channels (one 16-B chunk) and R accumulators per channel.
// block = 32 chunk-columns x 8 row-lanes; grid = (chunk blocks, M splits, taps)
// out index: rmajor ? r*ldo + col : col*ldo + r, col = tap*C + c (conv) or c.
// vgroup_cols > 0: V column offset 4*(c / vgroup_cols) (fused q/k/v up-projection gradients).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    AddrArgs A;
    const float* v; float* out; const float* scale;
    int M, R, ldv, ldo, rows_per_block, rmajor, vgroup_cols;
    // fixed-order reduction over the M splits (ws != null): every workgroup publishes its 32 x 8R partial sums in its own slab,
    // the split of a column block that arrives last adds the slabs in split order and does the one `out +=` of that block -
    // bit-reproducible gradients.  ws == null: fp32 atomics (commit in arrival order).
    float* ws; unsigned* ticket;
};

// slab_id: index of this workgroup's slab, slab_step: slabs between consecutive splits of one column block, ticket_id: the
// column block's arrival ticket, splits: M splits of the block
template <int R>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int bx, const int by, const int bz, const long slab_id = 0,
                                           const int slab_step = 0, const long ticket_id = 0, const int splits = 1) {
    __shared__ float red[4][32][8 * R + 1];
    __shared__ int last_flag;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int C = p.A.ca0 + p.A.ca1;
    const int c = (bx * 32 + cx) * 8;
    const bool cvalid = c < C;
    const int tap = bz;
    const int m_begin = by * p.rows_per_block;
    const int m_end = min(p.M, m_begin + p.rows_per_block);
    float acc[8][R];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[e][r] = 0.f;
    const int voff = p.vgroup_cols > 0 ? 4 * (c / p.vgroup_cols) : 0;
    const int sh = p.A.src_xform ? 1 : 0;
    const int HL = p.A.hs << sh, WL = p.A.ws << sh;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int hw = p.A.ho * p.A.wo;
    if (cvalid) {
        for (int m = m_begin + ry; m < m_end; m += 8) {
            const __bf16* src;
            if (p.A.mode == 0) {
                src = c < p.A.ca0 ? p.A.a0 + (long)m * p.A.lda0 + c : p.A.a1 + (long)m * p.A.lda1 + (c - p.A.ca0);
            } else {
                const int b = m / hw;
                const int rem = m - b * hw;
                const int oy = rem / p.A.wo, ox = rem - oy * p.A.wo;
                const int iy = oy * p.A.stride + ky - 1, ix = ox * p.A.stride + kx - 1;
                bool ok = (iy >= 0) & (iy < HL) & (ix >= 0) & (ix < WL);
                if (p.A.src_xform == 2) ok = ok & (((iy | ix) & 1) == 0);
                if (!ok) continue;
                const long pix = ((long)b * p.A.hs + (iy >> sh)) * p.A.ws + (ix >> sh);
                src = c < p.A.ca0 ? p.A.a0 + pix * p.A.lda0 + c : p.A.a1 + pix * p.A.lda1 + (c - p.A.ca0);
            }
            const bf16x8 z = *(const bf16x8*)src;
            float vv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) vv[r] = p.v[(long)m * p.ldv + voff + r];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ze = (float)z[e];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[e][r] += ze * vv[r];
            }
        }
    }
    // lanes l and l+32 of a wave hold the same chunk for two row-lanes: fold them, then 4 waves via LDS
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[e][r] += __shfl_xor(acc[e][r], 32, 64);
            if ((threadIdx.x & 32) == 0) red[threadIdx.x >> 6][cx][e * R + r] = acc[e][r];
        }
    __syncthreads();
    // 256 threads reduce 32 chunks x (8R) values over the 8 row-lanes
    const float s = *p.scale;
    if (p.ws != nullptr) {
        constexpr int PER = 32 * 8 * R;                                   // floats per slab: 4 consecutive ones per thread and pass
        const __amdgpu_buffer_rsrc_t slabs = __builtin_amdgcn_make_buffer_rsrc(p.ws + slab_id * PER - (long)by * slab_step * PER, 0,
                                                                              (int)((long)splits * slab_step * PER * 4), 0x00020000);
#pragma unroll
        for (int q = 0; q < PER / 1024; ++q) {
            const int idx = (q * 256 + threadIdx.x) * 4;
            f32x4 t4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ccx = (idx + u) / (8 * R), er = (idx + u) - ccx * (8 * R);
                float t = 0.f;
#pragma unroll
                for (int y = 0; y < 4; ++y) t += red[y][ccx][er];
                t4[u] = t;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t4), slabs, (by * slab_step * PER + idx) * 4, 0, 16 /* sc1 */);
        }
        if (!last_arriver(p.ticket + ticket_id, (unsigned)splits, &last_flag)) return;
#pragma unroll
        for (int q = 0; q < PER / 1024; ++q) {
            const int idx = (q * 256 + threadIdx.x) * 4;
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < splits; k0 += 8) {                      // 8 independent loads in flight, added in split order
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = k0 + u < splits ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 slabs, ((k0 + u) * slab_step * PER + idx) * 4, 0, 16 /* sc1 */))
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) tot += v[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ccx = (idx + u) / (8 * R), er = (idx + u) - ccx * (8 * R);
                const int e = er / R, r = er - e * R;
                const int cc = (bx * 32 + ccx) * 8 + e;
                if (cc >= C) continue;
                const long col = (p.A.mode == 0 ? 0 : (long)tap * C) + cc;
                float* o = p.rmajor ? p.out + (long)r * p.ldo + col : p.out + col * p.ldo + r;
                *o += s * tot[u];                                         // the only writer of this element in this launch
            }
        }
        return;
    }
    for (int idx = threadIdx.x; idx < 32 * 8 * R; idx += 256) {
        const int ccx = idx / (8 * R);
        const int er = idx - ccx * (8 * R);
        const int e = er / R, r = er - e * R;
        const int cc = (bx * 32 + ccx) * 8 + e;
        if (cc >= C) continue;
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < 4; ++y) t += red[y][ccx][er];
        const long col = (p.A.mode == 0 ? 0 : (long)tap * C) + cc;
        float* o = p.rmajor ? p.out + (long)r * p.ldo + col : p.out + col * p.ldo + r;
        atomicAdd(o, s * t);
    }
}

template <int R>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    // slab of workgroup (bx, by, bz): ((bz * splits + by) * gx + bx); ticket of its column block: bz * gx + bx
    wgrad_body<R>(p, blockIdx.x, blockIdx.y, blockIdx.z, ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x,
                  (int)gridDim.x, (long)blockIdx.z * gridDim.x + blockIdx.x, (int)gridDim.y);
}

// launch geometry of one weight-gradient problem (host and device agree by construction).  kind: 0 = one launch per problem,
// partial sums meet in fp32 atomics (~768 workgroups, 3 per CU); 1 = one launch per problem, slab reduction (~256 workgroups:
// every split costs a slab round trip); 2 = a problem inside a batched launch, slab reduction: the batch supplies the
// parallelism, a split is worth it only from ~512 rows on
struct WgradGeom { int gx, splits, taps, rows_per_block; };
__host__ __device__ inline WgradGeom wgrad_geom(int C, int M, int mode, int kind) {
    WgradGeom g;
    g.gx = (C / 8 + 31) / 32;
    g.taps = mode == 1 ? 9 : 1;
    // a rank-4 gradient is a reduction over M with only C/256 (x9 taps) independent column blocks: split M, at least 64 rows each
    const int blocks_xz = g.gx * g.taps;
    const int target = kind == 0 ? 768 : 256;
    int splits = kind == 2 ? (M + 511) / 512 : (target + blocks_xz - 1) / blocks_xz;
    const int max_splits = (M + 63) / 64, min_splits = kind == 2 ? 1 : (M + 1023) / 1024;
    if (splits > max_splits) splits = max_splits;
    if (splits < min_splits) splits = min_splits;
    if (splits < 1) splits = 1;
    g.splits = splits;
    g.rows_per_block = ((M + splits - 1) / splits + 7) / 8 * 8;
    return g;
}

__host__ __device__ inline void wgrad_args(WgradArgs& p, const slh_wgrad_desc& d) {
    p.A.a0 = (const __bf16*)d.z0; p.A.a1 = (const __bf16*)d.z1;
    p.A.lda0 = d.ldz0; p.A.lda1 = d.ldz1; p.A.ca0 = d.c0; p.A.ca1 = d.c1; p.A.mode = d.mode;
    p.A.hs = d.hs; p.A.ws = d.ws; p.A.src_xform = d.src_xform; p.A.stride = d.stride; p.A.ho = d.ho; p.A.wo = d.wo;
    p.v = d.v; p.out = d.out; p.scale = d.scale;
    p.M = d.M; p.R = d.R; p.ldv = d.ldv; p.ldo = d.ldo;
    p.rmajor = d.out_rmajor;
    p.vgroup_cols = d.vgroup_cols;
    p.ws = d.slabs; p.ticket = (unsigned*)d.tickets;
    p.rows_per_block = wgrad_geom(d.c0 + d.c1, d.M, d.mode, d.slabs ? 1 : 0).rows_per_block;
}

// problem of workgroup `bid`: the last index with prefix[index] <= bid (prefix[0] = 0, prefix[n] = total > bid)
__device__ __forceinline__ int batch_find(const int* prefix, int n, int bid) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= bid) lo = mid; else hi = mid;
    }
    return lo;
}

// ws / ticket: the batch's workspace (one slab per workgroup, one ticket per workgroup id - a column block uses the ticket at
// the id of its first split) or null (atomics, the legacy geometry)
template <int R>
__global__ __launch_bounds__(256) void wgrad_batch_kernel(const slh_wgrad_desc* table, const int* prefix, int n, float* ws, unsigned* ticket) {
    const int bid = blockIdx.x;
    const int pi = __builtin_amdgcn_readfirstlane(batch_find(prefix, n, bid));
    const slh_wgrad_desc d = table[pi];
    WgradArgs p;
    wgrad_args(p, d);
    p.ws = ws; p.ticket = ticket;
    const WgradGeom g = wgrad_geom(d.c0 + d.c1, d.M, d.mode, ws ? 2 : 0);
    p.rows_per_block = g.rows_per_block;
    const int local = bid - prefix[pi];
    const int bx = local % g.gx, rest = local / g.gx;
    const int by = rest % g.splits, bz = rest / g.splits;
    wgrad_body<R>(p, bx, by, bz, bid, g.gx, (long)prefix[pi] + (long)bz * g.splits * g.gx + bx, g.splits);
}

}  // namespace

static int fill_addr(AddrArgs& A, const void* a0, const void* a1, int lda0, int lda1, int ca0, int ca1, int mode,
                     int hs, int ws, int src_xform, int stride, int ho, int wo) {
    A.a0 = (const __bf16*)a0; A.a1 = (const __bf16*)a1;
    A.lda0 = lda0; A.lda1 = lda1; A.ca0 = ca0; A.ca1 = ca1; A.mode = mode;
    A.hs = hs; A.ws = ws; A.src_xform = src_xform; A.stride = stride; A.ho = ho; A.wo = wo;
    return 0;
}

extern "C" int slh_skinny(const slh_skinny_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->a0 && d->w && d->out, "slh_skinny: null pointer");
    SLH_CHECK(d->M > 0 && d->R > 0 && d->R <= 16 && d->K > 0, "slh_skinny: bad shape");
    SLH_CHECK(d->ca0 % 8 == 0 && d->ca1 % 8 == 0 && d->lda0 % 8 == 0 && d->lda1 % 8 == 0 && d->K % 8 == 0,
              "slh_skinny: 16-byte alignment");
    SLH_CHECK((d->a1 != nullptr) == (d->ca1 > 0), "slh_skinny: a1/ca1 mismatch");
    const int cin = d->ca0 + d->ca1;
    if (d->mode == 0) SLH_CHECK(cin == d->K, "slh_skinny: dense K mismatch");
    else {
        SLH_CHECK(d->mode == 1 && d->K == 9 * cin, "slh_skinny: conv K mismatch");
        SLH_CHECK(d->M == d->batch * d->ho * d->wo, "slh_skinny: conv M mismatch");
        SLH_CHECK(d->stride == 1 || d->stride == 2, "slh_skinny: stride");
    }
    if (d->out_kind == 1) SLH_CHECK(d->mode == 1, "slh_skinny: NCHW output needs conv geometry");
    if (d->w_kmajor) SLH_CHECK(d->R == 4 && d->mode == 0 && !d->bias, "slh_skinny: w_kmajor needs dense R=4");
    AddrArgs A;
    fill_addr(A, d->a0, d->a1, d->lda0, d->lda1, d->ca0, d->ca1, d->mode, d->hs, d->ws, d->src_xform, d->stride,
              d->ho, d->wo);
    hipStream_t s = (hipStream_t)stream;
    // one wave per row while that still gives < ~8 waves per SIMD of work; 16 lanes per row beyond
    const bool wide = d->M <= 16384;
#define SLH_SKINNY_LAUNCH(RM)                                                                                      \
    do {                                                                                                           \
        if (wide)                                                                                                  \
            hipLaunchKernelGGL((skinny_kernel<RM, 64>), dim3((d->M + 3) / 4), dim3(256), 0, s, A,                  \
                               (const __bf16*)d->w, (const __bf16*)d->bias, d->out, d->M, d->R, d->K, d->ldo,      \
                               d->out_kind, d->w_kmajor);                                                          \
        else                                                                                                       \
            hipLaunchKernelGGL((skinny_kernel<RM, 16>), dim3((d->M + 15) / 16), dim3(256), 0, s, A,                \
                               (const __bf16*)d->w, (const __bf16*)d->bias, d->out, d->M, d->R, d->K, d->ldo,      \
                               d->out_kind, d->w_kmajor);                                                          \
    } while (0)
    if (d->R <= 4) SLH_SKINNY_LAUNCH(4);
    else if (d->R <= 12) SLH_SKINNY_LAUNCH(12);
    else SLH_SKINNY_LAUNCH(16);
#undef SLH_SKINNY_LAUNCH
    SLH_LAUNCH_CHECK("slh_skinny");
    return 0;
}

extern "C" int slh_gemv(const slh_gemv_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->w && d->y, "slh_gemv: null pointer");
    SLH_CHECK(d->nb >= 1 && d->nb <= 8 && d->N > 0 && d->K > 0 && d->K % 8 == 0 && d->ldx % 8 == 0,
              "slh_gemv: bad shape nb=%d N=%d K=%d", d->nb, d->N, d->K);
    if (d->lora_t) SLH_CHECK(d->lora_up && d->lora_scale, "slh_gemv: lora pointers");
    hipLaunchKernelGGL(gemv_kernel, dim3((d->N + 3) / 4), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_gemv");
    return 0;
}

// ptrs: also the operand pointers (launch time); the geometry queries of plan time (slh_lora_wgrad_*blocks: also called by
// dry-run planning, which has shapes but no memory yet) check the shape fields only
static int wgrad_check(const slh_wgrad_desc* d, bool ptrs = true) {
    SLH_CHECK(d, "slh_lora_wgrad: null descriptor");
    if (ptrs) SLH_CHECK(d->z0 && d->v && d->out && d->scale, "slh_lora_wgrad: null pointer");
    SLH_CHECK(d->R == 4 || d->R == 12, "slh_lora_wgrad: R must be 4 or 12");
    SLH_CHECK(d->c0 % 8 == 0 && d->c1 % 8 == 0 && d->ldz0 % 8 == 0 && d->ldz1 % 8 == 0, "slh_lora_wgrad: alignment");
    if (ptrs) SLH_CHECK((d->z1 != nullptr) == (d->c1 > 0), "slh_lora_wgrad: z1/c1 mismatch");
    SLH_CHECK(d->M > 0 && d->c0 + d->c1 > 0, "slh_lora_wgrad: empty problem");
    if (d->mode == 1) SLH_CHECK(d->M == d->batch * d->ho * d->wo, "slh_lora_wgrad: conv M mismatch");
    return 0;
}

extern "C" int slh_lora_wgrad(const slh_wgrad_desc* d, slh_stream_t stream) {
    if (wgrad_check(d)) return -1;
    WgradArgs p;
    wgrad_args(p, *d);
    SLH_CHECK((d->slabs != nullptr) == (d->tickets != nullptr), "slh_lora_wgrad: slabs and tickets come together");
    const WgradGeom g = wgrad_geom(d->c0 + d->c1, d->M, d->mode, d->slabs ? 1 : 0);
    dim3 grid(g.gx, g.splits, g.taps);
    hipStream_t s = (hipStream_t)stream;
    if (d->R == 4) hipLaunchKernelGGL(wgrad_kernel<4>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(wgrad_kernel<12>, grid, dim3(256), 0, s, p);
    SLH_LAUNCH_CHECK("slh_lora_wgrad");
    return 0;
}

extern "C" int slh_lora_wgrad_blocks(const slh_wgrad_desc* d) {
    if (wgrad_check(d, false)) return -1;
    const WgradGeom g = wgrad_geom(d->c0 + d->c1, d->M, d->mode, d->slabs ? 2 : 0);
    return g.gx * g.splits * g.taps;
}

extern "C" int slh_lora_wgrad_single_blocks(const slh_wgrad_desc* d) {
    if (wgrad_check(d, false)) return -1;
    const WgradGeom g = wgrad_geom(d->c0 + d->c1, d->M, d->mode, 1);
    return g.gx * g.splits * g.taps;
}

extern "C" int slh_lora_wgrad_batch(const slh_batch_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->table && d->prefix && d->n > 0 && d->total > 0, "slh_lora_wgrad_batch: empty batch / null pointer");
    SLH_CHECK(d->arg == 4 || d->arg == 12, "slh_lora_wgrad_batch: arg must be the common R (4 or 12)");
    SLH_CHECK((d->slabs != nullptr) == (d->tickets != nullptr), "slh_lora_wgrad_batch: slabs and tickets come together");
    hipStream_t s = (hipStream_t)stream;
    if (d->arg == 4)
        hipLaunchKernelGGL(wgrad_batch_kernel<4>, dim3(d->total), dim3(256), 0, s, (const slh_wgrad_desc*)d->table, d->prefix, d->n,
                           (float*)d->slabs, (unsigned*)d->tickets);
    else
        hipLaunchKernelGGL(wgrad_batch_kernel<12>, dim3(d->total), dim3(256), 0, s, (const slh_wgrad_desc*)d->table, d->prefix, d->n,
                           (float*)d->slabs, (unsigned*)d->tickets);
    SLH_LAUNCH_CHECK("slh_lora_wgrad_batch");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward-data contribution of a 3x3 LoRA down conv:  gx[i][c] (+)= s * sum_{tap,r} U[o(i,tap)][r] * A[r][tap][c]
// with o*stride + tap - 1 = i.  Thread = (input pixel, 8-channel chunk).  HBM-bound on gx.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void lora_conv_dgrad_kernel(const slh_lora_cdgrad_desc d) {
    const int nchunk = d.cin / 8;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)d.batch * d.hl * d.wl * nchunk;
    if (gid >= total) return;
    const int chunk = (int)(gid % nchunk);
    const long pix = gid / nchunk;
    const int hw = d.hl * d.wl;
    const int b = (int)(pix / hw);
    const int rem = (int)(pix - (long)b * hw);
    const int iy = rem / d.wl, ix = rem - iy * d.wl;
    const int c = chunk * 8;
    const __bf16* A = (const __bf16*)d.a_down;
    const int K = 9 * d.cin;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int ty = iy + 1 - ky, tx = ix + 1 - kx;
        if (ty < 0 || tx < 0) continue;
        if (d.stride == 2 && ((ty | tx) & 1)) continue;
        const int oy = ty / d.stride, ox = tx / d.stride;
        if (oy >= d.ho || ox >= d.wo) continue;
        const f32x4 u = *(const f32x4*)(d.u + ((long)b * d.ho * d.wo + (long)oy * d.wo + ox) * d.ldu);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bf16x8 a = *(const bf16x8*)(A + (long)r * K + tap * d.cin + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += u[r] * (float)a[e];
        }
    }
    const float s = *d.scale;
    __bf16* g = (__bf16*)d.gx + pix * d.ldgx + c;
    bf16x8 o;
    if (d.accumulate) o = *(const bf16x8*)g;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(s * acc[e] + (d.accumulate ? (float)o[e] : 0.f));
    *(bf16x8*)g = o;
}

// LoRA gradients of one ResnetBlock2D.time_emb_proj (a [1 x ted] x [ted x C] linear applied to silu(emb)):
//   d_up[c][r] += s * g[c] * T[r];  U[r] = sum_c g[c] * up[c][r];  d_down[r][k] += s * U[r] * silu(emb[k])
__global__ __launch_bounds__(256) void temb_lora_bwd_kernel(const slh_temb_lora_bwd_desc d) {
    __shared__ float red[4][4];
    __shared__ float U[4];
    const int tid = threadIdx.x;
    const float s = *d.scale;
    float t[4], part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = d.t[r];
    for (int c = tid; c < d.C; c += 256) {
        const float g = d.g[c];
        const bf16x4 u = *(const bf16x4*)((const __bf16*)d.up + (long)c * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            part[r] += g * (float)u[r];
            d.d_up[(long)c * 4 + r] += s * g * t[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[r] = wave_sum(part[r]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[tid >> 6][r] = part[r];
    }
    __syncthreads();
    if (tid < 4) U[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    __syncthreads();
    for (int k = tid; k < d.ted; k += 256) {
        const float x = round_bf16(silu_f((float)((const __bf16*)d.emb)[k]));
#pragma unroll
        for (int r = 0; r < 4; ++r) d.d_down[(long)r * d.ted + k] += s * U[r] * x;
    }
}
// adapter side of a LayerNorm fold (slh_lora_ln_fold): one workgroup per (row, item); fixed-order sums (wave shuffles + LDS)
__global__ __launch_bounds__(256) void lora_ln_fold_kernel(const slh_lora_lnfold_item* items) {
    const slh_lora_lnfold_item it = items[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= it.rows) return;
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const __bf16* a = (const __bf16*)it.a + (long)r * it.K;
    __bf16* o = (__bf16*)it.a_out + (long)r * it.K;
    float s = 0.f, c = 0.f;
    for (int k = tid * 8; k < it.K; k += 256 * 8) {
        const bf16x8 av = *(const bf16x8*)(a + k);
        const bf16x8 gv = *(const bf16x8*)((const __bf16*)it.gamma + k);
        const bf16x8 bv = *(const bf16x8*)((const __bf16*)it.beta + k);
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ov[e] = (__bf16)((float)av[e] * (float)gv[e]);
            s += (float)ov[e];
            c += (float)av[e] * (float)bv[e];
        }
        *(bf16x8*)(o + k) = ov;
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = c; }
    __syncthreads();
    if (tid == 0) {
        it.s_out[r] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        it.c_out[r] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
}  // namespace

extern "C" int slh_lora_ln_fold(const slh_lora_lnfold_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->items && d->n > 0, "slh_lora_ln_fold: null pointer / empty");
    hipLaunchKernelGGL(lora_ln_fold_kernel, dim3(16, d->n), dim3(256), 0, (hipStream_t)stream, d->items);
    SLH_LAUNCH_CHECK("slh_lora_ln_fold");
    return 0;
}

extern "C" int slh_lora_conv_dgrad(const slh_lora_cdgrad_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->u && d->a_down && d->scale && d->gx, "slh_lora_conv_dgrad: null pointer");
    SLH_CHECK(d->cin % 8 == 0 && d->ldgx % 8 == 0 && d->ldu % 4 == 0 && (d->stride == 1 || d->stride == 2),
              "slh_lora_conv_dgrad: bad shape");
    const long total = (long)d->batch * d->hl * d->wl * (d->cin / 8);
    hipLaunchKernelGGL(lora_conv_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_lora_conv_dgrad");
    return 0;
}

extern "C" int slh_temb_lora_bwd(const slh_temb_lora_bwd_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->g && d->t && d->up && d->emb && d->d_up && d->d_down && d->scale, "slh_temb_lora_bwd: null pointer");
    hipLaunchKernelGGL(temb_lora_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_temb_lora_bwd");
    return 0;
}
