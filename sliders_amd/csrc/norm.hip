// GroupNorm(+SiLU) and LayerNorm, forward and backward-data, on pixel-major bf16 activations.  gfx950 only.
// HBM-bound: 16 B per lane per access, fp32 statistics, wave-shuffle / LDS reductions.
//
// Reference semantics replaced (diffusers-0.20.2, called through train_util.py:159-163 / 242-247):
//   ResnetBlock2D.norm1/norm2 + nonlinearity (GroupNorm(32, eps 1e-5) -> SiLU), Transformer2DModel.norm
//   (GroupNorm(32, eps 1e-6)), conv_norm_out + conv_act, BasicTransformerBlock.norm1/2/3 (LayerNorm eps 1e-5),
//   and their autograd backward (loss.backward(), train_lora_xl.py:345).
#include "common.h"
#include "../../include/sliders_hip.h"

namespace {

constexpr int GN_ITERS = 8;  // rows per thread per block

struct GnGeom {
    int C, nchunk, rpi, threads, rows_per_block, row_blocks, cg, lpg;
    unsigned lds_bytes;
};

static GnGeom gn_geom(int c0, int c1, int hw, int groups) {
    GnGeom g;
    g.C = c0 + c1;
    g.nchunk = g.C / 8;
    g.rpi = g.nchunk >= 256 ? 1 : 256 / g.nchunk;
    g.threads = g.nchunk * g.rpi;
    g.rows_per_block = g.rpi * GN_ITERS;
    g.row_blocks = (hw + g.rows_per_block - 1) / g.rows_per_block;
    g.cg = g.C / groups;
    g.lpg = 1;                                   // lanes that share one group in the fixed-order reductions
    while (g.lpg < 16 && 2 * g.lpg * groups <= g.threads) g.lpg *= 2;
    g.lds_bytes = (unsigned)((2 * g.threads * 8 + 4) * sizeof(float));
    return g;
}

// Forward statistics: one-shot workgroups (all of a workgroup's rows in flight at once: the launch streams at full width) whose
// pairs meet ONCE, inside clusters of GN_CLUSTER row blocks (the cluster's last arriver adds them in index order and leaves one
// pair per group); the few cluster pairs of a sample (<= ~65) are added - again in index order - by every workgroup of
// slh_gn_apply in its prologue, which also leaves (mean, rstd) in d.stats.  Round 3 combined the cluster pairs under a second
// ticket level inside the statistics launch: that serial tail (publish - ticket - dependent loads) was ~1/3 of the launch.
__device__ __forceinline__ const __bf16* gn_src(const slh_gn_desc& d, long row, int c) {
    return c < d.c0 ? (const __bf16*)d.x0 + row * d.ldx0 + c : (const __bf16*)d.x1 + row * d.ldx1 + (c - d.c0);
}

// Statistics of one (sample, group): stats[b][g] = (mean, rstd).  Two-pass-safe in one pass: the sums run over
// x - K_g with K_g = the group's first element (an actual sample of the data, so |mean - K_g| is a few standard
// deviations at most and E[(x-K)^2] - E[x-K]^2 does not cancel, whatever DC offset the activations carry).
__global__ void gn_stats_kernel(const slh_gn_desc d, int nchunk, int rpi, int rows_per_block, int cg, int lpg, int row_blocks) {
    extern __shared__ __attribute__((aligned(16))) float gn_lds[];
    const int tid = threadIdx.x;
    const int chunk = tid % nchunk, rl = tid / nchunk;
    const int c = chunk * 8;
    const int b = blockIdx.y;
    const int C = nchunk * 8;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(d.hw, r0 + rows_per_block);
    float ks[8], s[8], q[8];
    if (cg >= 8) {          // 8 consecutive channels touch at most two groups: two shift loads
        const int ga = c / cg, gb = (c + 7) / cg;
        const float ka = (float)*gn_src(d, (long)b * d.hw, ga * cg), kb = (float)*gn_src(d, (long)b * d.hw, gb * cg);
#pragma unroll
        for (int e = 0; e < 8; ++e) ks[e] = (c + e) < (ga + 1) * cg ? ka : kb;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) ks[e] = (float)*gn_src(d, (long)b * d.hw, ((c + e) / cg) * cg);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    {
        // all of the thread's rows are requested before the first one is used (HBM latency paid once, not GN_ITERS times)
        // (unconditional loads at clamped rows + a select: a load behind `if (r < r1)` makes hipcc branch around it and wait
        // vmcnt(0) at every join - the loads then complete one by one, cdna_hip_programming.md section 5 trap 4c)
        bf16x8 v[GN_ITERS];
#pragma unroll
        for (int it = 0; it < GN_ITERS; ++it) {
            const int r = min(r0 + rl + it * rpi, d.hw - 1);
            v[it] = *(const bf16x8*)gn_src(d, (long)b * d.hw + r, c);
        }
#pragma unroll
        for (int it = 0; it < GN_ITERS; ++it) {
            const float w = (r0 + rl + it * rpi) < r1 ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = ((float)v[it][e] - ks[e]) * w; s[e] += f; q[e] += f * f; }
        }
    }
    float S, Q;
    const bool owner = gn_block_reduce<8>(gn_lds, C, cg, d.groups, rpi, lpg, chunk, rl, s, q, S, Q);
    // level 1 only: the cluster's last arriver leaves the cluster's pair at part[row_blocks + cluster] (plain store: the next
    // kernel on the stream reads it)
    const int ncl = (row_blocks + GN_CLUSTER - 1) / GN_CLUSTER;
    float* part = d.partial + (long)b * (row_blocks + ncl) * d.groups * 2;
    const int g = tid / lpg;
    const int rb = blockIdx.x, cl = rb / GN_CLUSTER;
    if (owner) store_pair_sc1(part + ((long)rb * d.groups + g) * 2, S, Q);
    const int first = cl * GN_CLUSTER;
    const int members = min(GN_CLUSTER, row_blocks - first);
    if (!last_arriver(d.ticket + (long)b * (1 + ncl) + 1 + cl, (unsigned)members, (int*)(gn_lds + 2 * rpi * C))) return;
    double Sd, Qd;
    gn_combine_partials(part + (long)first * d.groups * 2, members, d.groups, lpg, Sd, Qd);
    if (g < d.groups && (tid & (lpg - 1)) == 0)
        *(f32x2*)(part + ((long)(row_blocks + cl) * d.groups + g) * 2) = f32x2{(float)Sd, (float)Qd};
}

// (mean, rstd) of every group of sample b from the `srb` cluster pairs slh_gn_stats left at `part`, in index order (double
// accumulation, `lpg` lanes per group each taking every lpg-th pair, then a fixed shuffle tree): identical in every workgroup.
// Ends with the pairs in st[0 .. groups) (mean) and st[64 ..) (rstd) behind a __syncthreads().
__device__ __forceinline__ void gn_stats_from_partials(const slh_gn_desc& d, int b, const float* part, int srb, int cg, int lpg, float* st) {
    const int tid = threadIdx.x;
    const int g = tid / lpg, j = tid & (lpg - 1);
    double a = 0.0, q = 0.0;
    if (g < d.groups) {
        constexpr int U = 16;
        for (int i0 = j; i0 < srb; i0 += U * lpg) {
            f32x2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * lpg;
                v[u] = *(const f32x2*)(part + ((long)(i < srb ? i : 0) * d.groups + g) * 2);
                if (i >= srb) v[u] = f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { a += (double)v[u][0]; q += (double)v[u][1]; }
        }
    }
    for (int o = lpg >> 1; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (g < d.groups && j == 0) {
        const double n = (double)d.hw * (double)cg;
        const double k = (double)(float)*gn_src(d, (long)b * d.hw, g * cg);
        const double m = a / n;
        const double var = fmax(q / n - m * m, 0.0);
        st[g] = (float)(k + m);
        st[64 + g] = (float)(1.0 / sqrt(var + (double)d.eps));
    }
    __syncthreads();
}

__global__ void gn_apply_kernel(const slh_gn_desc d, int nchunk, int rpi, int rows_per_block, int cg, int lpg, int row_blocks) {
    __shared__ float st[128];
    const int tid = threadIdx.x;
    const int chunk = tid % nchunk, rl = tid / nchunk;
    const int c = chunk * 8;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(d.hw, r0 + rows_per_block);
    float a[8], sft[8];
    const bf16x8 gm = *(const bf16x8*)((const __bf16*)d.gamma + c);
    const bf16x8 bt = *(const bf16x8*)((const __bf16*)d.beta + c);
    // this workgroup's rows are requested before the statistics are combined (their latency hides the combine's)
    bf16x8 v[GN_ITERS];
#pragma unroll
    for (int it = 0; it < GN_ITERS; ++it) {
        const int r = min(r0 + rl + it * rpi, d.hw - 1);       // unconditional (see gn_stats_kernel)
        v[it] = *(const bf16x8*)gn_src(d, (long)b * d.hw + r, c);
    }
    const int ncl = (row_blocks + GN_CLUSTER - 1) / GN_CLUSTER;
    gn_stats_from_partials(d, b, d.partial + ((long)b * (row_blocks + ncl) + row_blocks) * d.groups * 2, ncl, cg, lpg, st);
    if (blockIdx.x == 0 && tid < d.groups)                     // for the backward (slh_gn_bwd_*) and anyone else who reads d.stats
        *(f32x2*)(d.stats + ((long)b * d.groups + tid) * 2) = f32x2{st[tid], st[64 + tid]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / cg;
        const float mean = st[g];
        const float rstd = st[64 + g];
        a[e] = rstd * (float)gm[e];
        sft[e] = (float)bt[e] - mean * a[e];
    }
#pragma unroll
    for (int it = 0; it < GN_ITERS; ++it) {
        const int r = r0 + rl + it * rpi;
        if (r >= r1) continue;
        const long row = (long)b * d.hw + r;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[it][e] * a[e] + sft[e];
            if (d.act == 1) y = silu_f(round_bf16(y));  // reference rounds the GroupNorm output to bf16 before SiLU
            o[e] = (__bf16)y;
        }
        if constexpr (SLH_WT_MASK & 8) wt_store16(wt_rsrc(d.y), (row * d.ldy + c) * 2, o);      // write-through (common.h)
        else *(bf16x8*)((__bf16*)d.y + row * d.ldy + c) = o;
    }
}

// ---- tiny tensors: statistics + normalisation in ONE launch -----------------------------------------------------
// At 8x8 a (sample, group) slab is a few KB: one workgroup owns `gq` adjacent groups of one sample (gq * cg a
// multiple of 8 channels, so its rows are whole 16-byte chunks), keeps its slab in registers, reduces the statistics in a
// fixed order (shifted sums -> wave tree -> LDS) and writes the normalised rows - one read of x, no partials, no tickets,
// no second launch (the two-launch form costs ~9 + 6 us + a boundary on these shapes, all of it latency).
constexpr int GNF_ITEMS = 3;           // 16-byte chunks per thread.  Measured (scripts/probe_gn.py): the one-launch form wins only
                                       // while a thread holds <= 3 chunks (8x8 latents: 8-10 us against 6.5 + 6.3 + a boundary);
                                       // at 16x16 it ties and at 32x32 it LOSES 2.5x (41 us vs 8 + 7): 16-64 workgroups then carry
                                       // all of the SiLU / select arithmetic that the two-launch form spreads over the chip
constexpr int GNF_MAXG = 4;            // groups per workgroup

struct GnFusedGeom { int gq, nch, items; bool ok; };
__host__ __device__ inline GnFusedGeom gn_fused_geom(int C, int hw, int groups) {
    GnFusedGeom g;
    const int cg = C / groups;
    g.gq = 1;
    while (g.gq <= GNF_MAXG && (g.gq * cg) % 8) g.gq *= 2;
    g.nch = g.gq * cg / 8;
    g.items = (hw * g.nch + 255) / 256;
    g.ok = g.gq <= GNF_MAXG && groups % g.gq == 0 && g.items <= GNF_ITEMS && hw * g.nch >= 64;
    return g;
}

// r = idx / nch as (idx * magic) >> 32 with magic = ceil(2^32 / nch) (64-bit: it is 2^32 for nch = 1): exact for idx * nch < 2^32;
// group of channel x inside the workgroup's <= 4 groups by comparison - both replace ~30-instruction integer divisions
// that made this kernel 5x slower than its memory traffic (8 + 8 + 1 of them per 16-byte chunk)
__device__ __forceinline__ int gnf_div(int idx, unsigned long long magic) { return (int)(((unsigned long long)(unsigned)idx * magic) >> 32); }
__device__ __forceinline__ int gnf_group(int x, int cg) { return (x >= cg) + (x >= 2 * cg) + (x >= 3 * cg); }

__global__ __launch_bounds__(256) void gn_fused_kernel(const slh_gn_desc d, int cg, int gq, int nch, unsigned long long magic) {
    __shared__ float red[4][2 * GNF_MAXG];
    __shared__ float fin[2 * GNF_MAXG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int c_base = blockIdx.x * gq * cg;                 // first channel of this workgroup (a multiple of 8)
    const int total = d.hw * nch;
    // shifts: the first element of each local group (row 0 of the sample)
    float ks[GNF_MAXG];
#pragma unroll
    for (int g = 0; g < GNF_MAXG; ++g) ks[g] = g < gq ? (float)*gn_src(d, (long)b * d.hw, c_base + g * cg) : 0.f;
    bf16x8 v[GNF_ITEMS];
#pragma unroll
    for (int it = 0; it < GNF_ITEMS; ++it) {
        if (it * 256 < total) {                                 // uniform over the workgroup: no per-lane branch around the load
            const int idx = min(tid + it * 256, total - 1);
            const int r = gnf_div(idx, magic), ch = idx - r * nch;
            v[it] = *(const bf16x8*)gn_src(d, (long)b * d.hw + r, c_base + ch * 8);
        }
    }
    float s[GNF_MAXG], q[GNF_MAXG];
#pragma unroll
    for (int g = 0; g < GNF_MAXG; ++g) { s[g] = 0.f; q[g] = 0.f; }
#pragma unroll
    for (int it = 0; it < GNF_ITEMS; ++it) {
        const int idx = tid + it * 256;
        if (it * 256 < total) {
            const int ic = min(idx, total - 1);
            const int ch = ic - gnf_div(ic, magic) * nch;
            const float w = idx < total ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g = gnf_group(ch * 8 + e, cg);
#pragma unroll
                for (int gg = 0; gg < GNF_MAXG; ++gg) {
                    const float f = gg == g ? ((float)v[it][e] - ks[gg]) * w : 0.f;
                    s[gg] += f; q[gg] += f * f;
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < GNF_MAXG; ++g) { s[g] = wave_sum(s[g]); q[g] = wave_sum(q[g]); }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < GNF_MAXG; ++g) { red[wave][2 * g] = s[g]; red[wave][2 * g + 1] = q[g]; }
    }
    __syncthreads();
    if (tid < gq) {
        const double S = ((double)red[0][2 * tid] + (double)red[1][2 * tid]) + ((double)red[2][2 * tid] + (double)red[3][2 * tid]);
        const double Q = ((double)red[0][2 * tid + 1] + (double)red[1][2 * tid + 1]) + ((double)red[2][2 * tid + 1] + (double)red[3][2 * tid + 1]);
        const double n = (double)d.hw * (double)cg;
        const double m = S / n;
        const double var = fmax(Q / n - m * m, 0.0);
        const float mean = (float)((double)(float)*gn_src(d, (long)b * d.hw, c_base + tid * cg) + m);     // this group's shift + mean of the shifted data
        const float rstd = (float)(1.0 / sqrt(var + (double)d.eps));
        fin[2 * tid] = mean; fin[2 * tid + 1] = rstd;
        const int g_glob = blockIdx.x * gq + tid;
        d.stats[((long)b * d.groups + g_glob) * 2] = mean;
        d.stats[((long)b * d.groups + g_glob) * 2 + 1] = rstd;
    }
    __syncthreads();
    float mean[GNF_MAXG], rstd[GNF_MAXG];
#pragma unroll
    for (int g = 0; g < GNF_MAXG; ++g) { mean[g] = g < gq ? fin[2 * g] : 0.f; rstd[g] = g < gq ? fin[2 * g + 1] : 0.f; }
#pragma unroll
    for (int it = 0; it < GNF_ITEMS; ++it) {
        const int idx = tid + it * 256;
        if (idx >= total) continue;
        const int r = gnf_div(idx, magic), ch = idx - r * nch;
        const int c = c_base + ch * 8;
        const bf16x8 gm = *(const bf16x8*)((const __bf16*)d.gamma + c);
        const bf16x8 bt = *(const bf16x8*)((const __bf16*)d.beta + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = gnf_group(ch * 8 + e, cg);
            float mu = 0.f, rs = 0.f;
#pragma unroll
            for (int gg = 0; gg < GNF_MAXG; ++gg)
                if (gg == g) { mu = mean[gg]; rs = rstd[gg]; }
            const float a = rs * (float)gm[e];
            float y = (float)v[it][e] * a + ((float)bt[e] - mu * a);
            if (d.act == 1) y = silu_f(round_bf16(y));  // reference rounds the GroupNorm output to bf16 before SiLU
            o[e] = (__bf16)y;
        }
        *(bf16x8*)((__bf16*)d.y + ((long)b * d.hw + r) * d.ldy + c) = o;
    }
}

// ---- one launch for slabs that stay cache-resident (round 5) ---------------------------------------------------------
// The two-launch form costs a kernel boundary, a ticket tail and a partial-sum prologue per GroupNorm - 12 + 12 us in situ on
// tensors whose traffic is worth 2 us (profiles/r05_probe_gn.txt).  Here S sibling workgroups serve one (sample, group): EVERY
// sibling reduces the whole slab (hw rows x cg channels: 80-320 KB, the siblings' reads meet in L2) in the same fixed order -
// identical statistics in all of them, no partials, no tickets, nothing to wait for - and then normalises its own 1/S of the slab's
// chunks, which it reads a second time (L2 hits).  Redundant arithmetic instead of a second launch: the statistics cost 2.75 VALU
// slots per element (packed fp32), the SiLU epilogue ~15, so S = 4 adds < 20 % arithmetic; it only pays while the slab is small:
// every sibling walks the whole slab, latency-bound, so the time grows with the slab while the two-launch form spreads it over the
// chip.  Measured (scripts/probe_gn.py, B = 2, us, one-launch vs stats + apply): 16x16 x 1280: 5.3 vs 14.0; 32x32 x 640 / 1280 /
// 2560: 7.9 / 9.7 / 12.4 vs 14.8 / 16.3 / 20.3; 64x64 x 640: 17.6 vs 20.6; 64x64 x 1280 / 1920: 30.9 / 35.9 vs 27.9 / 31.7 (loses:
// GN1_MAX_ELEMS).  V = elements per access: 8 / 4 / 2 by the alignment of cg and c0.
constexpr int GN1_MAX_ELEMS = 4096 * 20;     // slab elements (hw * cg): 64x64 x 640 channels, 32x32 x 2560 (measured: scripts/probe_gn.py)
constexpr int GN1_MAX_HW = 4096;
constexpr int GN1_U = 8;                      // accesses in flight per thread and loop trip

__device__ __forceinline__ float silu_rcp_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // 1 ulp rcp: no IEEE division sequence

struct GnOneGeom { int V, nch, threads; bool ok; };
__host__ __device__ inline GnOneGeom gn_one_geom(int c0, int c1, int hw, int groups) {
    GnOneGeom g;
    const int C = c0 + c1, cg = C / groups;
    g.V = (cg % 8 == 0) ? 8 : (cg % 4 == 0 && c0 % 4 == 0) ? 4 : 2;
    g.nch = cg / g.V;
    const int total = hw * g.nch;
    g.threads = total >= 4096 ? 1024 : total >= 1024 ? 512 : 256;
    g.ok = cg % 2 == 0 && c0 % 2 == 0 && total >= 256;
    return g;
}
// size limits of the one-launch form; SLH_GN1_MAX_HW / SLH_GN1_MAX_ELEMS override them for measurements (scripts/probe_gn.py)
static bool gn_one_fits(int channels, int hw, int groups) {
    static const int max_hw = getenv("SLH_GN1_MAX_HW") ? atoi(getenv("SLH_GN1_MAX_HW")) : GN1_MAX_HW;
    static const long max_elems = getenv("SLH_GN1_MAX_ELEMS") ? atol(getenv("SLH_GN1_MAX_ELEMS")) : (long)GN1_MAX_ELEMS;
    return hw <= max_hw && (long)hw * (channels / groups) <= max_elems;
}
static int gn_one_siblings(int batch, int groups, int total_chunks, int threads) {
    int S = 1;
    while (S < 8 && 2 * S * batch * groups <= 320 && total_chunks / (2 * S) >= threads / 4) S *= 2;
    return S;
}

template <int V>
__global__ __launch_bounds__(1024) void gn_one_kernel(const slh_gn_desc d, int cg, int nch, unsigned long long magic, int S, int per) {
    typedef __attribute__((ext_vector_type(V))) __bf16 vec_t;
    typedef __attribute__((ext_vector_type(V / 2))) unsigned int raw_t;
    __shared__ float red[16][2];
    __shared__ float fin[2];
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    // Work item w = ((sample * groups + group) * S + sibling).  Workgroup n of a launch runs on XCD n % 8: handing XCD x the
    // contiguous run w in [x * nblk / 8, (x + 1) * nblk / 8) puts the siblings of a group AND the neighbouring groups (whose
    // 20-160-byte row segments share cache lines with it) behind ONE L2 - the tensor leaves HBM once, not S times (measured
    // without it, 64x64 x 1280 channels: 42.9 us for 21 MB, i.e. the 4 siblings each pulled their slab's lines from memory).
    const int nblk = gridDim.x, n = blockIdx.x;
    const int w = (nblk & 7) == 0 ? (n & 7) * (nblk >> 3) + (n >> 3) : n;
    const int gs = d.groups * S;
    const int b = w / gs, g = (w - b * gs) / S, sib = w - b * gs - g * S;
    const int c_base = g * cg;
    const int total = d.hw * nch;
    const long row0 = (long)b * d.hw;
    const __bf16 kb = *gn_src(d, row0, c_base);          // the pivot of the shifted sums (see gn_stats_kernel)
    const float k = (float)kb;
    const unsigned kraw = (unsigned)__builtin_bit_cast(unsigned short, kb) * 0x10001u;
    const f32x2 k2 = {k, k};
    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
    for (int i0 = tid; i0 < total; i0 += GN1_U * nt) {
        raw_t v[GN1_U];
#pragma unroll
        for (int u = 0; u < GN1_U; ++u) {                // unconditional loads at clamped indices: all in flight together
            const int idx = min(i0 + u * nt, total - 1);
            const int r = gnf_div(idx, magic), ch = idx - r * nch;
            v[u] = *(const raw_t*)gn_src(d, row0 + r, c_base + ch * V);
        }
#pragma unroll
        for (int u = 0; u < GN1_U; ++u) {
            const bool in = i0 + u * nt < total;
#pragma unroll
            for (int e = 0; e < V / 2; ++e) {
                const unsigned w = in ? v[u][e] : kraw;  // out of range: the pivot itself, (x - k) = 0 exactly
                const f32x2 x2 = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
                const f32x2 f2 = x2 - k2;
                s2 += f2;
                q2 = __builtin_elementwise_fma(f2, f2, q2);
            }
        }
    }
    // fixed order: (even + odd lanes' slots) -> wave shuffle tree -> waves in index order, in double
    const float s = wave_sum(s2[0] + s2[1]), q = wave_sum(q2[0] + q2[1]);
    if (lane == 0) { red[wave][0] = s; red[wave][1] = q; }
    __syncthreads();
    if (tid == 0) {
        double S_ = 0.0, Q_ = 0.0;
        for (int w = 0; w < (nt >> 6); ++w) { S_ += (double)red[w][0]; Q_ += (double)red[w][1]; }
        const double n = (double)d.hw * (double)cg;
        const double m = S_ / n;
        const double var = fmax(Q_ / n - m * m, 0.0);
        const float mean = (float)((double)k + m);
        const float rstd = (float)(1.0 / sqrt(var + (double)d.eps));
        fin[0] = mean; fin[1] = rstd;
        if (sib == 0) *(f32x2*)(d.stats + ((long)b * d.groups + g) * 2) = f32x2{mean, rstd};      // for the backward
    }
    __syncthreads();
    const float mean = fin[0], rstd = fin[1];
    const int lo = sib * per, hi = min(total, lo + per);
    for (int idx = lo + tid; idx < hi; idx += nt) {
        const int r = gnf_div(idx, magic), ch = idx - r * nch;
        const int c = c_base + ch * V;
        const vec_t xv = *(const vec_t*)gn_src(d, row0 + r, c);
        const vec_t gm = *(const vec_t*)((const __bf16*)d.gamma + c);
        const vec_t bt = *(const vec_t*)((const __bf16*)d.beta + c);
        vec_t o;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float a = rstd * (float)gm[e];
            float y = (float)xv[e] * a + ((float)bt[e] - mean * a);
            if (d.act == 1) y = silu_rcp_f(round_bf16(y));      // reference rounds the GroupNorm output to bf16 before SiLU
            o[e] = (__bf16)y;
        }
        __bf16* yp = (__bf16*)d.y + (row0 + r) * d.ldy + c;
        if constexpr (V == 8 && (SLH_WT_MASK & 8)) wt_store16(wt_rsrc(d.y), ((row0 + r) * d.ldy + c) * 2, o);
        else *(vec_t*)yp = o;
    }
}

// ---- backward -------------------------------------------------------------------------------------
__device__ __forceinline__ const __bf16* gnb_src(const slh_gn_bwd_desc& d, long row, int c) {
    return c < d.c0 ? (const __bf16*)d.x0 + row * d.ldx0 + c : (const __bf16*)d.x1 + row * d.ldx1 + (c - d.c0);
}

// dxhat[e] = dy * act'(z) * gamma, xhat[e]; z = xhat*gamma+beta
__device__ __forceinline__ void gnb_elem(const slh_gn_bwd_desc& d, const bf16x8& xv, const bf16x8& dyv,
                                         const float* mean, const float* rstd, const float* gm, const float* bt,
                                         float* dxh, float* xh) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xhat = ((float)xv[e] - mean[e]) * rstd[e];
        float g = (float)dyv[e];
        if (d.act == 1) {
            const float z = round_bf16(xhat * gm[e] + bt[e]);
            const float sg = 1.f / (1.f + __expf(-z));
            g *= sg * (1.f + z * (1.f - sg));
        }
        dxh[e] = g * gm[e];
        xh[e] = xhat;
    }
}

__global__ void gn_bwd_stats_kernel(const slh_gn_bwd_desc d, int nchunk, int rpi, int rows_per_block, int cg, int lpg,
                                    int row_blocks) {
    extern __shared__ __attribute__((aligned(16))) float gn_lds[];
    const int tid = threadIdx.x;
    const int chunk = tid % nchunk, rl = tid / nchunk;
    const int c = chunk * 8;
    const int b = blockIdx.y;
    const int C = nchunk * 8;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(d.hw, r0 + rows_per_block);
    float mean[8], rstd[8], gm[8], bt[8];
    const bf16x8 gmv = *(const bf16x8*)((const __bf16*)d.gamma + c);
    const bf16x8 btv = *(const bf16x8*)((const __bf16*)d.beta + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / cg;
        mean[e] = d.stats[((long)b * d.groups + g) * 2];
        rstd[e] = d.stats[((long)b * d.groups + g) * 2 + 1];
        gm[e] = (float)gmv[e]; bt[e] = (float)btv[e];
    }
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    for (int r = r0 + rl; r < r1; r += rpi) {
        const long row = (long)b * d.hw + r;
        const bf16x8 xv = *(const bf16x8*)gnb_src(d, row, c);
        const bf16x8 dyv = *(const bf16x8*)((const __bf16*)d.dy + row * d.lddy + c);
        float dxh[8], xh[8];
        gnb_elem(d, xv, dyv, mean, rstd, gm, bt, dxh, xh);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += dxh[e]; q[e] += dxh[e] * xh[e]; }
    }
    float S, Q;
    const bool owner = gn_block_reduce<8>(gn_lds, C, cg, d.groups, rpi, lpg, chunk, rl, s, q, S, Q);
    const int ncl = (row_blocks + GN_CLUSTER - 1) / GN_CLUSTER;
    double Sd, Qd;
    const bool fin = gn_two_level_reduce(d.bpartial + (long)b * (row_blocks + ncl) * d.groups * 2, d.bticket + (long)b * (1 + ncl),
                                         row_blocks, blockIdx.x, d.groups, lpg, owner, S, Q, (int*)(gn_lds + 2 * rpi * C), Sd, Qd);
    const int g = tid / lpg;
    if (fin) {
        d.bstats[((long)b * d.groups + g) * 2] = (float)Sd;
        d.bstats[((long)b * d.groups + g) * 2 + 1] = (float)Qd;
    }
}

__global__ void gn_bwd_apply_kernel(const slh_gn_bwd_desc d, int nchunk, int rpi, int rows_per_block, int cg) {
    const int tid = threadIdx.x;
    const int chunk = tid % nchunk, rl = tid / nchunk;
    const int c = chunk * 8;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(d.hw, r0 + rows_per_block);
    const float inv_n = 1.f / ((float)d.hw * (float)cg);
    float mean[8], rstd[8], gm[8], bt[8], m1[8], m2[8];
    const bf16x8 gmv = *(const bf16x8*)((const __bf16*)d.gamma + c);
    const bf16x8 btv = *(const bf16x8*)((const __bf16*)d.beta + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / cg;
        const long si = ((long)b * d.groups + g) * 2;
        mean[e] = d.stats[si];
        rstd[e] = d.stats[si + 1];
        gm[e] = (float)gmv[e]; bt[e] = (float)btv[e];
        m1[e] = d.bstats[si] * inv_n;
        m2[e] = d.bstats[si + 1] * inv_n;
    }
    const bool first = c < d.c0;
    __bf16* dxb = first ? (__bf16*)d.dx0 : (__bf16*)d.dx1;
    const int ldd = first ? d.lddx0 : d.lddx1;
    const int cc = first ? c : c - d.c0;
    const int accum = first ? d.accumulate0 : d.accumulate1;
    if (!dxb) return;
    for (int r = r0 + rl; r < r1; r += rpi) {
        const long row = (long)b * d.hw + r;
        const bf16x8 xv = *(const bf16x8*)gnb_src(d, row, c);
        const bf16x8 dyv = *(const bf16x8*)((const __bf16*)d.dy + row * d.lddy + c);
        float dxh[8], xh[8];
        gnb_elem(d, xv, dyv, mean, rstd, gm, bt, dxh, xh);
        __bf16* dst = dxb + row * ldd + cc;
        bf16x8 o;
        if (accum) o = *(const bf16x8*)dst;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = rstd[e] * (dxh[e] - m1[e] - xh[e] * m2[e]);
            if (accum) v += (float)o[e];
            o[e] = (__bf16)v;
        }
        *(bf16x8*)dst = o;
    }
}

// ---- LayerNorm: one wave per row, up to 3 x 8 elements per lane (C <= 1536) -------------------------
// Every load is unconditional, at a clamped column (lanes past C re-read column 0 and mask the value): a load behind
// `if (c < C)` makes hipcc branch around it and wait vmcnt(0) at the join, i.e. three serial memory round trips per row
// instead of one (norm.s before: load / vmcnt(0) / load / vmcnt(0) / load / vmcnt(0)).  gamma / beta are requested with x.
__global__ __launch_bounds__(256) void layernorm_kernel(const slh_ln_desc d) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= d.M) return;
    const __bf16* x = (const __bf16*)d.x + (long)m * d.ldx;
    bf16x8 xv[3], gv[3], bv[3];
    bool in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = (lane + 64 * j) * 8;
        in[j] = c < d.C;
        const int cc = in[j] ? c : 0;
        xv[j] = *(const bf16x8*)(x + cc);
        gv[j] = *(const bf16x8*)((const __bf16*)d.gamma + cc);
        bv[j] = *(const bf16x8*)((const __bf16*)d.beta + cc);
    }
    float v[3][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[j][e] = in[j] ? (float)xv[j][e] : 0.f; sum += v[j][e]; }
    const float mean = wave_sum(sum) / (float)d.C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = in[j] ? v[j][e] - mean : 0.f; sq += t * t; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)d.C + d.eps);
    if (d.mean_rstd && lane == 0) { d.mean_rstd[(long)m * 2] = mean; d.mean_rstd[(long)m * 2 + 1] = rstd; }
    __bf16* y = (__bf16*)d.y + (long)m * d.ldy;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = (lane + 64 * j) * 8;
        if (in[j]) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)((v[j][e] - mean) * rstd * (float)gv[j][e] + (float)bv[j][e]);
            if constexpr (SLH_WT_MASK & 8) wt_store16(wt_rsrc(d.y), ((long)m * d.ldy + c) * 2, o);
            else *(bf16x8*)(y + c) = o;
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const slh_ln_bwd_desc d) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= d.M) return;
    const __bf16* x = (const __bf16*)d.x + (long)m * d.ldx;
    const __bf16* dy = (const __bf16*)d.dy + (long)m * d.lddy;
    __bf16* dx = (__bf16*)d.dx + (long)m * d.lddx;
    bf16x8 xv[3], gv[3], dv[3], ov[3];
    bool in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {         // unconditional loads at clamped columns (see layernorm_kernel)
        const int c = (lane + 64 * j) * 8;
        in[j] = c < d.C;
        const int cc = in[j] ? c : 0;
        xv[j] = *(const bf16x8*)(x + cc);
        gv[j] = *(const bf16x8*)((const __bf16*)d.gamma + cc);
        dv[j] = *(const bf16x8*)(dy + cc);
        ov[j] = *(const bf16x8*)(dx + (d.accumulate ? cc : 0));      // read only where it is accumulated into
    }
    const float mean = d.mean_rstd[(long)m * 2], rstd = d.mean_rstd[(long)m * 2 + 1];
    float xh[3][8], dyh[3][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xh[j][e] = in[j] ? ((float)xv[j][e] - mean) * rstd : 0.f;
            dyh[j][e] = in[j] ? (float)dv[j][e] * (float)gv[j][e] : 0.f;
            s1 += dyh[j][e]; s2 += dyh[j][e] * xh[j][e];
        }
    const float m1 = wave_sum(s1) / (float)d.C, m2 = wave_sum(s2) / (float)d.C;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = (lane + 64 * j) * 8;
        if (in[j]) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = rstd * (dyh[j][e] - m1 - xh[j][e] * m2);
                if (d.accumulate) v += (float)ov[j][e];
                o[e] = (__bf16)v;
            }
            *(bf16x8*)(dx + c) = o;
        }
    }
}

}  // namespace

static int gn_check(const char* who, int c0, int c1, int groups, int ldx0, int ldx1, const void* x1) {
    const int C = c0 + c1;
    SLH_CHECK(C > 0 && C % 8 == 0 && c0 % 8 == 0 && c1 % 8 == 0, "%s: channels must be multiples of 8", who);
    SLH_CHECK(groups > 0 && groups <= 64 && C % groups == 0, "%s: bad group count", who);
    SLH_CHECK(ldx0 % 8 == 0 && ldx1 % 8 == 0, "%s: leading dims must be multiples of 8", who);
    SLH_CHECK((x1 != nullptr) == (c1 > 0), "%s: x1/c1 mismatch", who);
    SLH_CHECK(C / 8 <= 1024, "%s: too many channels", who);
    return 0;
}

extern "C" int slh_gn_row_blocks(int channels, int hw, int groups) {
    if (channels <= 0 || channels % 8 || hw <= 0 || groups <= 0 || channels % groups) return -1;
    return gn_geom(channels, 0, hw, groups).row_blocks;
}

extern "C" int slh_gn_clusters(int row_blocks) { return row_blocks > 0 ? (row_blocks + GN_CLUSTER - 1) / GN_CLUSTER : -1; }

extern "C" int slh_gn_stats(const slh_gn_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x0 && d->stats, "slh_gn_stats: null pointer");
    SLH_CHECK(d->partial && d->ticket, "slh_gn_stats: needs the partial-sum workspace and the zeroed ticket counters");
    if (gn_check("slh_gn_stats", d->c0, d->c1, d->groups, d->ldx0, d->ldx1, d->x1)) return -1;
    const GnGeom g = gn_geom(d->c0, d->c1, d->hw, d->groups);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(g.row_blocks, d->batch), dim3(g.threads), g.lds_bytes, (hipStream_t)stream, *d,
                       g.nchunk, g.rpi, g.rows_per_block, g.cg, g.lpg, g.row_blocks);
    SLH_LAUNCH_CHECK("slh_gn_stats");
    return 0;
}

extern "C" int slh_gn_fused_ok(int channels, int hw, int groups) {
    if (channels <= 0 || channels % 8 || hw <= 0 || groups <= 0 || channels % groups) return 0;
    if (gn_fused_geom(channels, hw, groups).ok) return 1;                 // tiny tensors: the register-resident kernel
    return gn_one_geom(channels, 0, hw, groups).ok && gn_one_fits(channels, hw, groups) ? 2 : 0;     // cache-resident slabs: sibling workgroups
}

extern "C" int slh_gn_fused(const slh_gn_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x0 && d->stats && d->y && d->gamma && d->beta, "slh_gn_fused: null pointer");
    if (gn_check("slh_gn_fused", d->c0, d->c1, d->groups, d->ldx0, d->ldx1, d->x1)) return -1;
    SLH_CHECK(d->ldy % 8 == 0, "slh_gn_fused: ldy");
    SLH_CHECK(wt_span_ok((long)d->batch * d->hw, d->ldy, d->c0 + d->c1), "slh_gn_fused: y reaches past 2 GiB from its base (32-bit store offsets)");
    const GnFusedGeom g = gn_fused_geom(d->c0 + d->c1, d->hw, d->groups);
    if (!g.ok) {
        const GnOneGeom o = gn_one_geom(d->c0, d->c1, d->hw, d->groups);
        SLH_CHECK(o.ok && gn_one_fits(d->c0 + d->c1, d->hw, d->groups), "slh_gn_fused: shape C=%d hw=%d is not a one-launch case (slh_gn_fused_ok)",
                  d->c0 + d->c1, d->hw);
        SLH_CHECK(d->y != d->x0 && (!d->x1 || d->y != d->x1), "slh_gn_fused: y must not alias x0 / x1 (sibling workgroups re-read the slab)");
        const int cg = (d->c0 + d->c1) / d->groups, total = d->hw * o.nch;
        const int S = gn_one_siblings(d->batch, d->groups, total, o.threads);
        const int per = (total + S - 1) / S;
        const unsigned long long magic = (0x100000000ull + o.nch - 1) / o.nch;
        const dim3 grid(d->groups * S * d->batch), block(o.threads);
        if (o.V == 8) hipLaunchKernelGGL(gn_one_kernel<8>, grid, block, 0, (hipStream_t)stream, *d, cg, o.nch, magic, S, per);
        else if (o.V == 4) hipLaunchKernelGGL(gn_one_kernel<4>, grid, block, 0, (hipStream_t)stream, *d, cg, o.nch, magic, S, per);
        else hipLaunchKernelGGL(gn_one_kernel<2>, grid, block, 0, (hipStream_t)stream, *d, cg, o.nch, magic, S, per);
        SLH_LAUNCH_CHECK("slh_gn_fused");
        return 0;
    }
    hipLaunchKernelGGL(gn_fused_kernel, dim3(d->groups / g.gq, d->batch), dim3(256), 0, (hipStream_t)stream, *d,
                       (d->c0 + d->c1) / d->groups, g.gq, g.nch, (0x100000000ull + g.nch - 1) / g.nch);       // 2^32 itself for nch = 1: 64-bit
    SLH_LAUNCH_CHECK("slh_gn_fused");
    return 0;
}

extern "C" int slh_gn_apply(const slh_gn_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x0 && d->stats && d->y && d->gamma && d->beta, "slh_gn_apply: null pointer");
    SLH_CHECK(d->partial, "slh_gn_apply: needs the partial sums slh_gn_stats left (same descriptor)");
    // every workgroup rebuilds (mean, rstd) from the partial sums AND the pivot x[b][row 0][first channel of the group], which it
    // re-reads from the input: an in-place call would let workgroups that run later see a normalised pivot - different means per
    // workgroup, silently
    SLH_CHECK(d->y != d->x0 && (!d->x1 || d->y != d->x1), "slh_gn_apply: y must not alias x0 / x1 (the statistics' pivot is re-read from x)");
    if (gn_check("slh_gn_apply", d->c0, d->c1, d->groups, d->ldx0, d->ldx1, d->x1)) return -1;
    SLH_CHECK(d->ldy % 8 == 0, "slh_gn_apply: ldy");
    SLH_CHECK(wt_span_ok((long)d->batch * d->hw, d->ldy, d->c0 + d->c1), "slh_gn_apply: y reaches past 2 GiB from its base (32-bit store offsets)");
    const GnGeom g = gn_geom(d->c0, d->c1, d->hw, d->groups);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(g.row_blocks, d->batch), dim3(g.threads), 0, (hipStream_t)stream, *d,
                       g.nchunk, g.rpi, g.rows_per_block, g.cg, g.lpg, g.row_blocks);
    SLH_LAUNCH_CHECK("slh_gn_apply");
    return 0;
}

extern "C" int slh_gn_bwd_stats(const slh_gn_bwd_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x0 && d->stats && d->bstats && d->dy, "slh_gn_bwd_stats: null pointer");
    SLH_CHECK(d->bpartial && d->bticket, "slh_gn_bwd_stats: needs the partial-sum workspace and the zeroed ticket counters");
    if (gn_check("slh_gn_bwd_stats", d->c0, d->c1, d->groups, d->ldx0, d->ldx1, d->x1)) return -1;
    const GnGeom g = gn_geom(d->c0, d->c1, d->hw, d->groups);
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(g.row_blocks, d->batch), dim3(g.threads), g.lds_bytes, (hipStream_t)stream,
                       *d, g.nchunk, g.rpi, g.rows_per_block, g.cg, g.lpg, g.row_blocks);
    SLH_LAUNCH_CHECK("slh_gn_bwd_stats");
    return 0;
}

extern "C" int slh_gn_bwd_apply(const slh_gn_bwd_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x0 && d->stats && d->bstats && d->dy, "slh_gn_bwd_apply: null pointer");
    if (gn_check("slh_gn_bwd_apply", d->c0, d->c1, d->groups, d->ldx0, d->ldx1, d->x1)) return -1;
    SLH_CHECK(d->lddx0 % 8 == 0 && d->lddx1 % 8 == 0 && d->lddy % 8 == 0, "slh_gn_bwd_apply: leading dims");
    const GnGeom g = gn_geom(d->c0, d->c1, d->hw, d->groups);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(g.row_blocks, d->batch), dim3(g.threads), 0, (hipStream_t)stream,
                       *d, g.nchunk, g.rpi, g.rows_per_block, g.cg);
    SLH_LAUNCH_CHECK("slh_gn_bwd_apply");
    return 0;
}

extern "C" int slh_layernorm(const slh_ln_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->y && d->gamma && d->beta, "slh_layernorm: null pointer");
    SLH_CHECK(d->C % 8 == 0 && d->C <= 1536 && d->ldx % 8 == 0 && d->ldy % 8 == 0, "slh_layernorm: C=%d unsupported", d->C);
    SLH_CHECK(wt_span_ok(d->M, d->ldy, d->C), "slh_layernorm: y reaches past 2 GiB from its base (32-bit store offsets)");
    hipLaunchKernelGGL(layernorm_kernel, dim3((d->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_layernorm");
    return 0;
}

extern "C" int slh_layernorm_bwd(const slh_ln_bwd_desc* d, slh_stream_t stream) {
    SLH_CHECK(d && d->x && d->dy && d->dx && d->gamma && d->mean_rstd, "slh_layernorm_bwd: null pointer");
    SLH_CHECK(d->C % 8 == 0 && d->C <= 1536 && d->ldx % 8 == 0 && d->lddy % 8 == 0 && d->lddx % 8 == 0,
              "slh_layernorm_bwd: C=%d unsupported", d->C);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((d->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *d);
    SLH_LAUNCH_CHECK("slh_layernorm_bwd");
    return 0;
}
