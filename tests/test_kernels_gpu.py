"""Per-kernel parity: every HIP entry point of include/sliders_hip.h, called through the C ABI, against a
plain PyTorch fp32 restatement of the same op on the same seeded inputs.

Tolerances (floating point, bf16 I/O, fp32 accumulate): relative L2 error of the bf16 output against the fp32
reference computed from the SAME bf16-rounded inputs must be < 6e-3 (= 1.5 bf16 ulps, i.e. output rounding
plus accumulation-order noise).  Elementwise kernels whose rounding points are restated exactly (CFG+DDIM,
loss gradient, AdamW) must be BIT-EXACT against the torch bf16 ops.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from sliders_amd import lib
from tests.util import bf, p, report, stream

pytestmark = pytest.mark.gpu
TOL = 6e-3


def _conv_ref(x_img, w, stride=1, up=False, dilate=False):
    """x_img [B,C,H,W] fp32, w [Co,Ci,3,3]"""
    if up:
        x_img = F.interpolate(x_img, scale_factor=2.0, mode="nearest")
    if dilate:
        B, C, H, W = x_img.shape
        z = torch.zeros(B, C, 2 * H, 2 * W, device=x_img.device)
        z[:, :, ::2, ::2] = x_img
        x_img = z
    return F.conv2d(x_img, w, None, stride=stride, padding=1)


def _to_pix(x_img):
    B, C, H, W = x_img.shape
    return x_img.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def _pack_conv(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("tile", [0x22, 0x21, 0x12, 0x11, 0x321, 0x312, 0x311, 0x4022, 0x4322, 0x4012, 0x4312, 0x4011,
                                  0x322, 0x422, 0x421, 0x412, 0x411, 0x4422, 0x4412, 0x4411, 0x8042, 0x8013, 0x8014, 0x8015])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 320, 320), (1024, 640, 1280), (154, 256, 2048)])
def test_gemm_dense(dev, M, N, K, tile):
    torch.manual_seed(M + N + K)
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = bf(torch.randn(N, device=dev))
    res = bf(torch.randn(M, N, device=dev))
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(w), bias=p(bias), residual=p(res), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K,
                     M=M, N=N, K=K, ld_res=N, ldc=N, rows_per_sample=M, tile=tile)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + bias.float() + res.float()
    report(f"gemm_dense M{M} N{N} K{K} tile{tile:x}", c, ref, TOL)


@pytest.mark.parametrize("tile", [0x22, 0x21, 0x12, 0x11, 0x311, 0x4022, 0x4012, 0x4011, 0x4312, 0x422, 0x412, 0x4412, 0x4322, 0x8042, 0x8013, 0x8014, 0x8015])
def test_gemm_packed_weights(dev, tile):
    """w_layout = 1: the frozen weights in the tile-packed, pre-swizzled order (sliders_amd.weights.pack_gemm_w)."""
    from sliders_amd.weights import pack_gemm_w
    for M, N, K in ((300, 320, 320), (1024, 640, 1280), (154, 200, 2048), (70, 4, 64)):
        torch.manual_seed(M + N + K)
        x = bf(torch.randn(M, K, device=dev))
        w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
        bias = bf(torch.randn(N, device=dev))
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        wp = pack_gemm_w(w)
        d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bias), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K,
                         ldc=N, rows_per_sample=M, tile=tile, w_layout=1)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"gemm_packed M{M} N{N} K{K} tile{tile:x}", c, x.float() @ w.float().t() + bias.float(), TOL)
    # 3x3 conv, stride 2, Cout not a multiple of 64
    torch.manual_seed(5)
    B, H, W, Ci, Co = 2, 12, 20, 128, 160
    img = bf(torch.randn(B, Ci, H, W, device=dev))
    w4 = bf(torch.randn(Co, Ci, 3, 3, device=dev) / math.sqrt(9 * Ci))
    ref_img = _conv_ref(img.float(), w4.float(), 2)
    Ho, Wo = ref_img.shape[2:]
    M = B * Ho * Wo
    c = torch.zeros(M, Co, device=dev, dtype=torch.bfloat16)
    wp = pack_gemm_w(_pack_conv(w4))
    d = lib.GemmDesc(a0=p(bf(_to_pix(img.float()))), w=p(wp), c=p(c), lda0=Ci, ca0=Ci, mode=1, batch=B, hs=H, ws=W,
                     stride=2, ho=Ho, wo=Wo, ldw=0, M=M, N=Co, K=9 * Ci, ldc=Co, rows_per_sample=Ho * Wo, tile=tile,
                     w_layout=1)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"conv_packed tile{tile:x}", c, _to_pix(ref_img), TOL)


@pytest.mark.parametrize("tile", [0, 0x422, 0x4412, 0x4322, 0x312, 0x8042, 0x8014, 0x8015])
def test_gemm_two_source_rowbias_lora(dev, tile):
    torch.manual_seed(1)
    B, HW, C0, C1, N = 2, 160, 128, 64, 320
    M, K = B * HW, C0 + C1
    x0 = bf(torch.randn(M, C0, device=dev))
    x1big = bf(torch.randn(M, 2 * C1, device=dev))      # second source is a column slice (ld != C)
    x1 = x1big[:, C1:]
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    rb = bf(torch.randn(B, 512, device=dev))
    for groups in (1, 2):
        T = torch.randn(M, 4 * groups, device=dev)
        up = bf(torch.randn(N, 4, device=dev))
        scale = torch.tensor([0.25], device=dev)
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        d = lib.GemmDesc(a0=p(x0), a1=x1.data_ptr(), w=p(w), rowbias=rb.data_ptr() + 2 * 64, lora_t=p(T),
                         lora_up=p(up), lora_scale=p(scale), c=p(c), lda0=C0, lda1=2 * C1, ca0=C0, ca1=C1, mode=0,
                         stride=1, ldw=K, M=M, N=N, K=K, ld_rowbias=512, rows_per_sample=HW, ld_t=4 * groups,
                         lora_groups=groups, ldc=N, tile=tile)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        xcat = torch.cat([x0.float(), x1.float()], 1)
        ref = xcat @ w.float().t() + rb.float()[:, 64:64 + N].repeat_interleave(HW, 0)
        ng = N // groups
        for g in range(groups):
            ref[:, g * ng:(g + 1) * ng] += 0.25 * T[:, 4 * g:4 * g + 4] @ up.float()[g * ng:(g + 1) * ng].t()
        report(f"gemm_2src_rowbias_lora g{groups}", c, ref, TOL)


@pytest.mark.parametrize("tile", [0, 0x22, 0x12, 0x21, 0x11, 0x4012, 0x4011, 0x4022, 0x422, 0x4412, 0x4322, 0x312, 0x8042, 0x8014])
def test_gemm_head_transposed_v_store(dev, tile):
    """vt_out: the V third of a fused q|k|v projection leaves the epilogue in slh_attn_fwd's [B][H][D][T] layout
    (what slh_transpose_heads would make of c[:, 2C:]); q and k still land in c.  With the LoRA term of to_q/k/v."""
    torch.manual_seed(11)
    B, T, heads, D, K = 2, 192, 2, 64, 192
    C = heads * D
    M, N = B * T, 3 * C
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    Tl = torch.randn(M, 12, device=dev)
    up = bf(torch.randn(N, 4, device=dev))
    scale = torch.tensor([0.5], device=dev)
    c = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
    vt = torch.full((B, heads, D, T), 7.0, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(w), lora_t=p(Tl), lora_up=p(up), lora_scale=p(scale), c=p(c), lda0=K, ca0=K, mode=0,
                     stride=1, ldw=K, M=M, N=N, K=K, ld_t=12, lora_groups=3, ldc=N, rows_per_sample=T, tile=tile,
                     vt_out=p(vt), vt_col0=2 * C, vt_D=D, vt_heads=heads, vt_tokens=T, vt_ld=T)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t()
    for g in range(3):
        ref[:, g * C:(g + 1) * C] += 0.5 * Tl[:, 4 * g:4 * g + 4] @ up.float()[g * C:(g + 1) * C].t()
    report(f"gemm_vt tile{tile:x} q|k", c[:, :2 * C], ref[:, :2 * C], TOL)
    assert (c[:, 2 * C:].float() == 7.0).all(), "the V columns must not be written to c"
    vref = ref[:, 2 * C:].reshape(B, T, heads, D).permute(0, 2, 3, 1)
    report(f"gemm_vt tile{tile:x} v^T", vt, vref, TOL)
    # training form: the V columns in BOTH layouts (vt_also_c)
    c.fill_(7.0)
    vt.fill_(7.0)
    d.vt_also_c = 1
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm_vt also_c tile{tile:x} c", c, ref, TOL)
    report(f"gemm_vt also_c tile{tile:x} v^T", vt, vref, TOL)
    assert torch.equal(c[:, 2 * C:].reshape(B, T, heads, D).permute(0, 2, 3, 1).contiguous(), vt), "the two layouts hold the same bits"
    d.vt_also_c = 0
    if tile in (0x4012, 0x8014):
        # the production form of the no-grad passes: the adapter's down matrix fused in (third operand tile), three groups
        A = bf(torch.randn(12, K, device=dev) / math.sqrt(K))
        c.fill_(7.0)
        vt.fill_(7.0)
        df = lib.GemmDesc(a0=p(x), w=p(w), lora_down=p(A), lora_up=p(up), lora_scale=p(scale), c=p(c), lda0=K, ca0=K, mode=0,
                          stride=1, ldw=K, M=M, N=N, K=K, ld_t=12, lora_groups=3, lora_rank=12, ldc=N, rows_per_sample=T, tile=tile,
                          vt_out=p(vt), vt_col0=2 * C, vt_D=D, vt_heads=heads, vt_tokens=T, vt_ld=T)
        lib.call(lib.OP_GEMM, df, stream())
        torch.cuda.synchronize()
        Tf = bf(x.float() @ A.float().t() * 0.5).float()       # scaled and rounded to bf16 ahead of the up-projection MFMA
        reff = x.float() @ w.float().t()
        for g in range(3):
            reff[:, g * C:(g + 1) * C] += Tf[:, 4 * g:4 * g + 4] @ up.float()[g * C:(g + 1) * C].t()
        report(f"gemm_vt fused adapter tile{tile:x} q|k", c[:, :2 * C], reff[:, :2 * C], TOL)
        report(f"gemm_vt fused adapter tile{tile:x} v^T", vt, reff[:, 2 * C:].reshape(B, T, heads, D).permute(0, 2, 3, 1), TOL)
        assert (c[:, 2 * C:].float() == 7.0).all()
    if tile in (0x4412, 0x22):
        # the same with K cut into slices: the last slice to arrive runs this epilogue on the slice-ordered sums
        c.fill_(7.0)
        vt.fill_(7.0)
        d.tile = tile | (3 << 16)
        slabs = torch.full((3, (M + 255) // 256 * 256, (N + 127) // 128 * 128), float("nan"), device=dev)
        tickets = torch.zeros((M // 64) * (N // 64), device=dev, dtype=torch.int64)
        d.splitk_c32, d.splitk_slabs, d.splitk_ticket = p(slabs), 3, p(tickets)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"gemm_vt splitk tile{tile:x} q|k", c[:, :2 * C], ref[:, :2 * C], TOL)
        report(f"gemm_vt splitk tile{tile:x} v^T", vt, vref, TOL)
        assert int(tickets.abs().sum()) == 0


def test_gemm_geglu(dev):
    torch.manual_seed(2)
    M, d_, K = 512, 128, 256      # proj: K -> 8*d_/... here N = 2*n_out
    n_out = 4 * d_
    N = 2 * n_out
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    b = bf(torch.randn(N, device=dev))
    from sliders_amd.weights import _geglu_perm
    wp, bp = _geglu_perm(w), _geglu_perm(b)
    for tile in (0x22, 0x12, 0x4022, 0x4312, 0x422, 0x412, 0x4412, 0x4322, 0x8042):
        c = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
        d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bp), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K,
                         ldc=n_out, geglu=1, rows_per_sample=M, tile=tile)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        proj = bf(x.float() @ w.float().t() + b.float()).float()
        ref = proj[:, :n_out] * bf(F.gelu(proj[:, n_out:])).float()
        report(f"gemm_geglu tile{tile:x}", c, ref, TOL)
    # training form: the same launch also leaves proj(x) (bf16, this product's column order) for the GEGLU backward
    c = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
    pre_f = torch.full((M, N + 8), 7.0, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bp), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K,
                     ldc=n_out, geglu=1, rows_per_sample=M, tile=0x4412, geglu_pre=p(pre_f), ld_pre=N + 8)
    lib.call(lib.OP_GEMM, d, stream())
    plain = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    d2 = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bp), c=p(plain), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K,
                      ldc=N, rows_per_sample=M, tile=0x4412)
    lib.call(lib.OP_GEMM, d2, stream())
    torch.cuda.synchronize()
    report("gemm_geglu with pre-activation output", c, ref, TOL)
    assert torch.equal(pre_f[:, :N], plain), "geglu_pre must be what the same product writes without the GEGLU epilogue"
    assert (pre_f[:, N:].float() == 7.0).all()
    # split-K with the GEGLU epilogue (the last slice of a tile to arrive runs the ordinary epilogue on the slice-ordered sums)
    for tile in (0x24412, 0x44012):
        S = tile >> 16
        c = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
        ws = torch.full((S, (M + 255) // 256 * 256, (N + 127) // 128 * 128), float("nan"), device=dev)
        tickets = torch.zeros((M // 64) * (N // 64), device=dev, dtype=torch.int64)
        d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bp), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K,
                         ldc=n_out, geglu=1, rows_per_sample=M, tile=tile, splitk_c32=p(ws), splitk_slabs=S,
                         splitk_ticket=p(tickets))
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"gemm_geglu splitk tile{tile:x}", c, ref, TOL)
        assert int(tickets.abs().sum()) == 0
    # unfused training path: blocked pre-activation + elementwise GEGLU fwd / bwd
    pre = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bp), c=p(pre), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K,
                     ldc=N, rows_per_sample=M)
    lib.call(lib.OP_GEMM, d, stream())
    out = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
    lib.call(lib.OP_ELEMENTWISE, lib.EwDesc(a=p(pre), out=p(out), M=M, C=n_out, lda=N, ldo=n_out, op=lib.EW_GEGLU_FWD), stream())
    torch.cuda.synchronize()
    report("geglu_unfused_fwd", out, ref, TOL)
    dy = bf(torch.randn(M, n_out, device=dev))
    dpre = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    lib.call(lib.OP_ELEMENTWISE, lib.EwDesc(a=p(pre), b=p(dy), out=p(dpre), M=M, C=n_out, lda=N, ldb=n_out, ldo=N,
                                            op=lib.EW_GEGLU_BWD), stream())
    torch.cuda.synchronize()
    pr = proj.clone().requires_grad_(True)
    (pr[:, :n_out] * F.gelu(pr[:, n_out:])).backward(dy.float())
    report("geglu_bwd", dpre, _perm_cols(pr.grad), 1.5e-2)


@pytest.mark.parametrize("tile", [0x8015, 0x8014, 0x8013, 0x8042, 0x4412, 0x22])
def test_gemm_geglu_16_blocks(dev, tile):
    """slh_gemm_desc.geglu = 3: GEGLU with the weight rows in 32-row blocks [16 value | 16 gate] (weights._geglu_perm16) - the
    pairing lives inside one 32 x 32 accumulator block, so tiles whose waves own an odd number of blocks can take it.  With the
    LayerNorm fold on the consumer side (ff.net.0.proj as the pass runs it) and without."""
    from sliders_amd.weights import _geglu_perm16, fold_layernorm
    torch.manual_seed(23)
    M, K, n_out = 300, 320, 448              # N = 896: 2.8 tiles of 320 columns, 3.5 of 256
    N = 2 * n_out
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    b = bf(torch.randn(N, device=dev))
    proj = bf(x.float() @ w.float().t() + b.float()).float()
    ref = proj[:, :n_out] * bf(F.gelu(proj[:, n_out:])).float()
    c = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(_geglu_perm16(w)), bias=p(_geglu_perm16(b)), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N,
                     K=K, ldc=n_out, geglu=3, rows_per_sample=M, tile=tile)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm_geglu16 tile{tile:x}", c, ref, TOL)
    # folded LayerNorm in front (statistics from slh_layernorm's chunk form are produced by a plain product here)
    gamma, beta = bf(torch.randn(K, device=dev) * 0.5 + 1.0), bf(torch.randn(K, device=dev) * 0.3)
    hc = x.float().view(M, K // 64, 64).double()
    mean_c = hc.mean(-1)
    chunks = torch.stack([mean_c, ((hc - mean_c[..., None]) ** 2).sum(-1)], -1).permute(1, 0, 2).contiguous().float()   # [chunk][M][2]
    ln = bf(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5))
    wf, sv, bp = fold_layernorm(w, b, gamma, beta)
    wf, sv, bp = _geglu_perm16(wf), _geglu_perm16(sv).contiguous(), _geglu_perm16(bp).contiguous()
    proj = bf(ln.float() @ w.float().t() + b.float()).float()
    refg = proj[:, :n_out] * bf(F.gelu(proj[:, n_out:])).float()
    c.zero_()
    d = lib.GemmDesc(a0=p(x), w=p(wf), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K, ldc=n_out, geglu=3,
                     rows_per_sample=M, tile=tile, ln_in=p(chunks), ln_in_chunks=K // 64, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm_geglu16 + ln fold tile{tile:x}", c, refg, TOL)


@pytest.mark.parametrize("tile", [0, 0x4412, 0x22, 0x11, 0x4322, 0x24412, 0x8042])
def test_gemm_geglu_backward_form(dev, tile):
    """slh_gemm_desc.geglu = 2: the backward-data product of the Linear behind a GEGLU writes d(proj) itself - bit-identical to
    the two-launch form (plain product -> bf16 d(ff) -> slh_elementwise GEGLU_BWD with the forward's pre-activation)."""
    torch.manual_seed(17)
    M, Nff, K = 300, 256, 320                    # d(ff) [M][Nff] = dY [M][K] . W2 [K][Nff]  (W2^T stored [Nff][K])
    dy = bf(torch.randn(M, K, device=dev))
    wT = bf(torch.randn(Nff, K, device=dev) / math.sqrt(K))
    pre = bf(torch.randn(M, 2 * Nff, device=dev))
    S = (tile >> 16) & 15
    kw = {}
    slabs = torch.full((max(S, 1), 512, 256), float("nan"), device=dev)          # kept alive: the descriptor only holds pointers
    tickets = torch.zeros(64, device=dev, dtype=torch.int64)
    if S > 1:
        kw = dict(splitk_c32=p(slabs), splitk_slabs=S, splitk_ticket=p(tickets))
    dff = torch.zeros(M, Nff, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(dy), w=p(wT), c=p(dff), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=Nff, K=K, ldc=Nff,
                     rows_per_sample=M, tile=tile, **kw)
    lib.call(lib.OP_GEMM, d, stream())
    two = torch.zeros(M, 2 * Nff, device=dev, dtype=torch.bfloat16)
    lib.call(lib.OP_ELEMENTWISE, lib.EwDesc(a=p(pre), b=p(dff), out=p(two), M=M, C=Nff, lda=2 * Nff, ldb=Nff, ldo=2 * Nff,
                                            op=lib.EW_GEGLU_BWD), stream())
    one = torch.full((M, 2 * Nff + 8), 7.0, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(dy), w=p(wT), c=p(one), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=Nff, K=K, ldc=2 * Nff + 8,
                     rows_per_sample=M, tile=tile, geglu=2, geglu_pre=p(pre), ld_pre=2 * Nff, **kw)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    assert torch.equal(one[:, :2 * Nff], two), float((one[:, :2 * Nff].float() - two.float()).abs().max())
    assert (one[:, 2 * Nff:].float() == 7.0).all()
    bad = lib.GemmDesc(a0=p(dy), w=p(wT), c=p(one), bias=p(dy), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=Nff, K=K,
                       ldc=2 * Nff + 8, rows_per_sample=M, geglu=2, geglu_pre=p(pre), ld_pre=2 * Nff)
    with pytest.raises(lib.SlidersHipError, match="geglu = 2"):
        lib.call(lib.OP_GEMM, bad, stream())


@pytest.mark.parametrize("C,offset", [(640, 0.0), (1280, 0.0), (320, 0.0), (1280, 8.0)])
def test_gemm_layernorm_folded(dev, C, offset):
    """BasicTransformerBlock.norm2 / norm3 folded into the products around them (slh_gemm_desc.ln_out / ln_in): the
    producer (attention out-projection + residual) leaves (mean, M2) of every 64-column chunk of its bf16 rows; the
    consumer multiplies the RAW rows with the gamma-scaled weights and normalises in the epilogue.  Against the reference's
    op sequence (LayerNorm in fp32 -> bf16 -> Linear -> bf16; GEGLU for ff1), plain and GEGLU consumers, several tiles;
    offset: rows whose mean is many standard deviations from zero (the chunk statistics are two-pass, merged with Chan's
    update - nothing cancels)."""
    from sliders_amd.weights import _geglu_perm, fold_layernorm
    torch.manual_seed(C)
    M = 300
    o = bf(torch.randn(M, C, device=dev))
    wo = bf(torch.randn(C, C, device=dev) / math.sqrt(C))
    bo = bf(torch.randn(C, device=dev) + offset)
    res = bf(torch.randn(M, C, device=dev) * 2)
    gamma, beta = bf(torch.randn(C, device=dev) * 0.5 + 1.0), bf(torch.randn(C, device=dev) * 0.3)
    h_ref = bf(o.float() @ wo.float().t() + bo.float() + res.float())
    for ptile in (0x4412, 0x422, 0x12, 0x4022, 0x8042):
        h = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
        chunks = torch.full((C // 64, M, 2), float("nan"), device=dev)      # chunk-major
        d = lib.GemmDesc(a0=p(o), w=p(wo), bias=p(bo), residual=p(res), c=p(h), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=C,
                         K=C, ld_res=C, ldc=C, rows_per_sample=M, tile=ptile, ln_out=p(chunks))
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"ln producer tile{ptile:x} C{C}", h, h_ref.float(), TOL)
        hc = h.float().view(M, C // 64, 64).double()
        mref = hc.mean(-1)
        m2ref = ((hc - mref[..., None]) ** 2).sum(-1)
        got = chunks.permute(1, 0, 2).double()
        assert float((got[..., 0] - mref).abs().max()) < 1e-5 * max(1.0, float(mref.abs().max()))
        assert float(((got[..., 1] - m2ref) / m2ref).abs().max()) < 1e-4
        # the reference's sequence on the tensor the producer actually stored
        ln = bf(F.layer_norm(h.float(), (C,), gamma.float(), beta.float(), 1e-5))
        # (a) plain consumer without bias (attn2.to_q / the fused q|k|v)
        N = 2 * C
        w = bf(torch.randn(N, C, device=dev) / math.sqrt(C))
        wf, sv, bp = fold_layernorm(w, None, gamma, beta)
        ref = ln.float() @ w.float().t()
        for tile in (0x4412, 0x22, 0x11, 0x4322, 0x8042, 0x8014, 0x8015):
            c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            mr = torch.full((M, 2), float("nan"), device=dev)
            d = lib.GemmDesc(a0=p(h), w=p(wf), c=p(c), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=N, K=C, ldc=N,
                             rows_per_sample=M, tile=tile, ln_in=p(chunks), ln_in_chunks=C // 64, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5,
                             ln_mr_out=p(mr))
            lib.call(lib.OP_GEMM, d, stream())
            torch.cuda.synchronize()
            report(f"ln consumer plain tile{tile:x} C{C} off{offset}", c, ref, TOL)
            # the rows' (mean, rstd), as slh_layernorm leaves them for the backward
            hd = h.double()
            mean_ref, var_ref = hd.mean(-1), hd.var(-1, unbiased=False)
            assert float((mr[:, 0].double() - mean_ref).abs().max()) < 1e-5 * max(1.0, float(mean_ref.abs().max()))
            assert float((mr[:, 1].double() * torch.sqrt(var_ref + 1e-5) - 1).abs().max()) < 1e-4
        # (b) GEGLU consumer with bias (ff.net.0.proj)
        n_out = 256
        wg = bf(torch.randn(2 * n_out, C, device=dev) / math.sqrt(C))
        bg = bf(torch.randn(2 * n_out, device=dev))
        wf, sv, bp = fold_layernorm(wg, bg, gamma, beta)
        wf, sv, bp = _geglu_perm(wf), _geglu_perm(sv).contiguous(), _geglu_perm(bp).contiguous()
        proj = bf(ln.float() @ wg.float().t() + bg.float()).float()
        refg = proj[:, :n_out] * bf(F.gelu(proj[:, n_out:])).float()
        for tile in (0x4412, 0x12, 0x4012, 0x8042):
            c = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
            d = lib.GemmDesc(a0=p(h), w=p(wf), c=p(c), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=2 * n_out, K=C, ldc=n_out,
                             geglu=1, rows_per_sample=M, tile=tile, ln_in=p(chunks), ln_in_chunks=C // 64, ln_s=p(sv), ln_b=p(bp),
                             ln_eps=1e-5)
            lib.call(lib.OP_GEMM, d, stream())
            torch.cuda.synchronize()
            report(f"ln consumer geglu tile{tile:x} C{C} off{offset}", c, refg, TOL)
    # both sides combined with split-K: the slice that arrives last runs the same epilogue on the slice-ordered sums
    tickets = torch.zeros(((M + 63) // 64) * ((2 * C + 63) // 64), device=dev, dtype=torch.int64)
    slabs = torch.full((3, 512, 2 * C), float("nan"), device=dev)
    h2 = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    chunks2 = torch.full((C // 64, M, 2), float("nan"), device=dev)
    d = lib.GemmDesc(a0=p(o), w=p(wo), bias=p(bo), residual=p(res), c=p(h2), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=C,
                     K=C, ld_res=C, ldc=C, rows_per_sample=M, tile=0x24412, ln_out=p(chunks2), splitk_c32=p(slabs), splitk_slabs=3,
                     splitk_ticket=p(tickets))
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"ln producer splitk C{C}", h2, h_ref.float(), TOL)
    hc = h2.float().view(M, C // 64, 64).double()
    assert float((chunks2.permute(1, 0, 2).double()[..., 0] - hc.mean(-1)).abs().max()) < 1e-5 * max(1.0, float(hc.mean(-1).abs().max()))
    ln2 = bf(F.layer_norm(h2.float(), (C,), gamma.float(), beta.float(), 1e-5))
    wf2, sv2, bp2 = fold_layernorm(w, None, gamma, beta)
    for tile in (0x24412, 0x30022):
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        d = lib.GemmDesc(a0=p(h2), w=p(wf2), c=p(c), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=N, K=C, ldc=N,
                         rows_per_sample=M, tile=tile, ln_in=p(chunks2), ln_in_chunks=C // 64, ln_s=p(sv2), ln_b=p(bp2), ln_eps=1e-5,
                         splitk_c32=p(slabs), splitk_slabs=3, splitk_ticket=p(tickets))
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"ln consumer splitk tile{tile:x} C{C} off{offset}", c, ln2.float() @ w.float().t(), TOL)
    assert int(tickets.abs().sum()) == 0
    # descriptors the kernel cannot honour are rejected before any launch
    bad = lib.GemmDesc(a0=p(o), w=p(wo), c=p(h), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=C, K=C, ldc=C, rows_per_sample=M,
                       tile=0x11, ln_out=p(chunks))
    with pytest.raises(lib.SlidersHipError, match="ln_out"):
        lib.call(lib.OP_GEMM, bad, stream())
    bad = lib.GemmDesc(a0=p(h), w=p(wf), bias=p(bo), c=p(h), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=C, K=C, ldc=C,
                       rows_per_sample=M, ln_in=p(chunks), ln_in_chunks=C // 64, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5)
    with pytest.raises(lib.SlidersHipError, match="ln_in"):
        lib.call(lib.OP_GEMM, bad, stream())


@pytest.mark.parametrize("M,N,K", [(64, 160, 64), (192, 320, 128), (256, 480, 192), (2048, 1280, 1280), (128, 160, 5120), (320, 1280, 320),
                                   (64, 160, 256), (128, 320, 384)])
@pytest.mark.parametrize("tile", [0x5425, 0x5525])
def test_gemm_tile_64x160(dev, M, N, K, tile):
    """The 64 x 160 tile (csrc/gemm5.hip, tile 0x5425: 4 waves of 32 x 80 on the 16 x 16 x 32 MFMA, 4-slot LDS ring): packed weights,
    every K-loop length class of the 4- and 5-slot rings (1, 2, 3, 4, 5 and many K tiles), plain / bias / bias + residual, against fp32; bit-equal
    between runs."""
    from sliders_amd.weights import pack_gemm_w
    torch.manual_seed(M + N + K)
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = bf(torch.randn(N, device=dev))
    res = bf(torch.randn(M, N, device=dev))
    wp = pack_gemm_w(w)
    for use_bias, use_res in ((False, False), (True, False), (True, True)):
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bias) if use_bias else 0, residual=p(res) if use_res else 0, c=p(c), lda0=K, ca0=K,
                         mode=0, stride=1, ldw=0, M=M, N=N, K=K, ld_res=N, ldc=N, rows_per_sample=M, tile=tile, w_layout=1)
        assert lib.gemm5_ok(d)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        ref = x.float() @ w.float().t() + (bias.float() if use_bias else 0) + (res.float() if use_res else 0)
        report(f"gemm 64x160 M{M} N{N} K{K} bias{int(use_bias)} res{int(use_res)}", c, ref, TOL)
        c2 = torch.zeros_like(c)
        d.c = p(c2)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        assert torch.equal(c, c2)
    # strided operands (a row slice of a wider activation, a wider output)
    xa = bf(torch.randn(M, K + 64, device=dev))
    cw = torch.zeros(M, N + 8, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(xa), w=p(wp), bias=p(bias), c=p(cw), lda0=K + 64, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K, ldc=N + 8,
                     rows_per_sample=M, tile=tile, w_layout=1)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm 64x160 strided M{M} N{N} K{K}", cw[:, :N], xa[:, :K].float() @ w.float().t() + bias.float(), TOL)
    assert float(cw[:, N:].float().abs().max()) == 0.0


@pytest.mark.parametrize("C,offset", [(320, 0.0), (640, 0.0), (1280, 0.0), (1280, 8.0)])
def test_gemm_tile_64x160_layernorm_producer(dev, C, offset):
    """ln_out on the 64 x 160 tile: (mean, M2) of every 80-COLUMN chunk of the stored rows, chunk-major [C/80][M][2]; a consumer on the
    ordinary tiles merges them with ln_in_chunks = C / 80 (equal-sized chunks of either width merge the same way) and reproduces
    Linear(LayerNorm(h)) of the reference's op sequence."""
    from sliders_amd.weights import fold_layernorm, pack_gemm_w
    torch.manual_seed(C + 5)
    M = 320
    o = bf(torch.randn(M, C, device=dev))
    wo = bf(torch.randn(C, C, device=dev) / math.sqrt(C))
    bo = bf(torch.randn(C, device=dev) + offset)
    res = bf(torch.randn(M, C, device=dev) * 2)
    gamma, beta = bf(torch.randn(C, device=dev) * 0.5 + 1.0), bf(torch.randn(C, device=dev) * 0.3)
    h = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    chunks = torch.full((C // 80, M, 2), float("nan"), device=dev)
    wop = pack_gemm_w(wo)        # (kept in a variable: a temporary would be freed before the launch)
    d = lib.GemmDesc(a0=p(o), w=p(wop), bias=p(bo), residual=p(res), c=p(h), lda0=C, ca0=C, mode=0, stride=1, ldw=0, M=M, N=C,
                     K=C, ld_res=C, ldc=C, rows_per_sample=M, tile=lib.TILE_64x160, w_layout=1, ln_out=p(chunks))
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"ln producer 64x160 C{C}", h, (o.float() @ wo.float().t() + bo.float() + res.float()), TOL)
    hc = h.float().view(M, C // 80, 80).double()
    mref = hc.mean(-1)
    m2ref = ((hc - mref[..., None]) ** 2).sum(-1)
    got = chunks.permute(1, 0, 2).double()
    assert float((got[..., 0] - mref).abs().max()) < 1e-5 * max(1.0, float(mref.abs().max()))
    assert float(((got[..., 1] - m2ref) / m2ref).abs().max()) < 1e-4
    ln = bf(F.layer_norm(h.float(), (C,), gamma.float(), beta.float(), 1e-5))
    N = 2 * C
    w = bf(torch.randn(N, C, device=dev) / math.sqrt(C))
    wf, sv, bp = fold_layernorm(w, None, gamma, beta)
    for tile in (0x4412, 0x22, 0x8014, 0x8015):
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        mr = torch.full((M, 2), float("nan"), device=dev)
        d = lib.GemmDesc(a0=p(h), w=p(wf), c=p(c), lda0=C, ca0=C, mode=0, stride=1, ldw=C, M=M, N=N, K=C, ldc=N, rows_per_sample=M, tile=tile,
                         ln_in=p(chunks), ln_in_chunks=C // 80, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5, ln_mr_out=p(mr))
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"ln consumer of 80-column chunks tile{tile:x} C{C} off{offset}", c, ln.float() @ w.float().t(), TOL)
        hd = h.double()
        assert float((mr[:, 0].double() - hd.mean(-1)).abs().max()) < 1e-5 * max(1.0, float(hd.mean(-1).abs().max()))
        assert float((mr[:, 1].double() * torch.sqrt(hd.var(-1, unbiased=False) + 1e-5) - 1).abs().max()) < 1e-4
    # the tile's own consumer side (round 6: attn2.to_q with norm2 folded in), packed weights; also over 64-column chunks
    wfp = pack_gemm_w(wf)
    hc64 = h.float().view(M, C // 64, 64).double()
    m64 = hc64.mean(-1)
    chunks64 = torch.stack([m64, ((hc64 - m64[..., None]) ** 2).sum(-1)], -1).permute(1, 0, 2).contiguous().float()
    outs = {}
    for tile, ch, nch in ((lib.TILE_64x160, chunks, C // 80), (0x5525, chunks, C // 80), (lib.TILE_64x160, chunks64, C // 64)):
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        mr = torch.full((M, 2), float("nan"), device=dev)
        d = lib.GemmDesc(a0=p(h), w=p(wfp), c=p(c), lda0=C, ca0=C, mode=0, stride=1, ldw=0, M=M, N=N, K=C, ldc=N, rows_per_sample=M, tile=tile,
                         w_layout=1, ln_in=p(ch), ln_in_chunks=nch, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5, ln_mr_out=p(mr))
        assert lib.gemm5_ok(d)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"ln consumer ON the 64x160 tile{tile:x} chunks{nch} C{C} off{offset}", c, ln.float() @ w.float().t(), TOL)
        assert float((mr[:, 0].double() - hd.mean(-1)).abs().max()) < 1e-5 * max(1.0, float(hd.mean(-1).abs().max()))
        assert float((mr[:, 1].double() * torch.sqrt(hd.var(-1, unbiased=False) + 1e-5) - 1).abs().max()) < 1e-4
        outs[(tile, nch)] = c
    assert torch.equal(outs[(lib.TILE_64x160, C // 80)], outs[(0x5525, C // 80)]), "the ring depth does not change the arithmetic"


@pytest.mark.parametrize("M,N,K", [(64, 160, 64), (192, 320, 192), (2048, 1280, 1280), (128, 160, 2560)])
def test_gemm_tile_64x160_fused_adapter(dev, M, N, K):
    """lora.py:108-112 inside the 64 x 160 tile: lora_down as 8 extra rows of the W tile, T = x . A^T by one MFMA per row block and
    k-step, the up-projection as one MFMA per accumulator block; T written out (lora_t_out) for the backward.  Against fp32 with the
    reference's rounding of the down-projection output, and against the 128 x 128 ring tile's fused form."""
    from sliders_amd.weights import pack_gemm_w
    torch.manual_seed(M + N + K + 1)
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias, res = bf(torch.randn(N, device=dev)), bf(torch.randn(M, N, device=dev))
    A = bf(torch.randn(4, K, device=dev) / math.sqrt(K))
    up = bf(torch.randn(N, 4, device=dev))
    scale = torch.tensor([0.75], device=dev)
    wp = pack_gemm_w(w)
    outs = {}
    for tile in (lib.TILE_64x160, 0x5525, 0x4412):
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        T = torch.full((M, 4), float("nan"), device=dev)
        ch = torch.full((N // 80, M, 2), float("nan"), device=dev)
        d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bias), residual=p(res), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K,
                         ld_res=N, ldc=N, rows_per_sample=M, tile=tile, w_layout=1, lora_down=p(A), lora_up=p(up), lora_scale=p(scale),
                         ld_t=4, lora_groups=1, lora_rank=4, lora_t_out=p(T))
        if tile != 0x4412:
            d.ln_out = p(ch)
            assert lib.gemm5_ok(d)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        t32 = x.float() @ A.float().t()
        ref = x.float() @ w.float().t() + bias.float() + res.float() + bf(0.75 * t32).float() @ up.float().t()
        report(f"gemm 64x160 fused adapter M{M} N{N} K{K} tile{tile:x}", c, ref, TOL)
        assert float((T - t32).abs().max()) < 2e-3 * max(1.0, float(t32.abs().max()))
        outs[tile] = c
        if tile != 0x4412:
            hc = c.float().view(M, N // 80, 80).double()
            got = ch.permute(1, 0, 2).double()
            assert float((got[..., 0] - hc.mean(-1)).abs().max()) < 1e-5 * max(1.0, float(hc.mean(-1).abs().max()))
    assert torch.equal(outs[lib.TILE_64x160], outs[0x5525]), "the ring depth does not change the arithmetic"
    assert float((outs[lib.TILE_64x160].float() - outs[0x4412].float()).abs().max()) <= 2.0 ** -6 * float(outs[0x4412].float().abs().max())
    # what the tile's adapter form does not cover is refused
    for bad in (dict(lora_groups=3, lora_rank=12, ld_t=12), dict(lora_up_rmajor=1)):
        d = lib.GemmDesc(**{**dict(a0=p(x), w=p(wp), c=p(outs[0x4412]), lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K, ldc=N,
                                   rows_per_sample=M, tile=lib.TILE_64x160, w_layout=1, lora_down=p(A), lora_up=p(up), lora_scale=p(scale),
                                   ld_t=4, lora_groups=1, lora_rank=4), **bad})
        assert not lib.gemm5_ok(d)


def test_gemm_tile_64x160_rejections(dev):
    """what the tile cannot run is refused before any launch (and slh_gemm5_ok says so to the planner)"""
    from sliders_amd.weights import pack_gemm_w
    M, N, K = 128, 320, 128
    x, w = bf(torch.randn(M, K, device=dev)), bf(torch.randn(N, K, device=dev))
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    wp = pack_gemm_w(w)
    base = dict(a0=p(x), w=p(wp), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K, ldc=N, rows_per_sample=M,
                tile=lib.TILE_64x160, w_layout=1)
    assert lib.gemm5_ok(lib.GemmDesc(**base))
    for bad in (dict(N=256), dict(M=100), dict(w_layout=0, ldw=K), dict(geglu=1), dict(rowbias=p(c), ld_rowbias=N),
                dict(ln_in=p(c), ln_in_chunks=3, ln_s=p(c), ln_b=p(c)), dict(ln_in=p(c), ln_in_chunks=2, ln_s=p(c), ln_b=p(c), bias=p(c)),
                dict(ln_mr_out=p(c)), dict(tile=lib.TILE_64x160 | 0x20000)):
        d = lib.GemmDesc(**{**base, **bad})
        if "tile" not in bad:
            assert not lib.gemm5_ok(d)
        with pytest.raises(lib.SlidersHipError, match="64 x 160"):
            lib.call(lib.OP_GEMM, d, stream())


@pytest.mark.parametrize("tile", [0x8014, 0x8013])
@pytest.mark.parametrize("offset", [0.0, 20.0])
def test_gemm_layernorm_folded_with_fused_adapter(dev, tile, offset):
    """norm1 folded into the adapter-carrying q|k|v projection (slh_gemm_desc.ln_in + lora_down + ln_lora_s / ln_lora_c): the main
    product AND the adapter's down-projection are computed from the raw rows and normalised in the epilogue, ahead of the
    up-projection; A . gamma, its row sums and A . beta come from slh_lora_ln_fold.  Against the reference's op sequence
    (LayerNorm -> bf16 -> Linear + LoRA with T rounded to bf16), with the V third written head-transposed as in production."""
    from sliders_amd.weights import fold_layernorm
    torch.manual_seed(int(offset) + tile)
    B, T, heads, D = 2, 192, 4, 64
    C = heads * D
    M, N, K = B * T, 3 * C, C
    x = bf(torch.randn(M, K, device=dev) * 1.5 + offset)
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    gamma, beta = bf(torch.randn(K, device=dev) * 0.5 + 1.0), bf(torch.randn(K, device=dev) * 0.3)
    A = bf(torch.randn(12, K, device=dev) / math.sqrt(K))
    up = bf(torch.randn(N, 4, device=dev))
    scale = torch.tensor([0.5], device=dev)
    xc = x.float().view(M, K // 64, 64)
    mean = xc.mean(-1)
    chunks = torch.stack([mean, ((xc - mean[..., None]) ** 2).sum(-1)], -1).permute(1, 0, 2).contiguous()
    wf, sv, bp = fold_layernorm(w, None, gamma, beta)
    # adapter side of the fold, by the library
    A2 = torch.zeros_like(A)
    sc = torch.full((2, 16), float("nan"), device=dev)
    items = torch.tensor([[p(A), p(gamma), p(beta), p(A2), sc.data_ptr(), sc.data_ptr() + 64, 12 | (K << 32)]], dtype=torch.int64, device=dev)
    lib.call(lib.OP_LORA_LN_FOLD, lib.LoraLnFoldDesc(items=items.data_ptr(), n=1), stream())
    torch.cuda.synchronize()
    A2_ref = bf(A.float() * gamma.float())
    assert torch.equal(A2, A2_ref)
    assert float((sc[0, :12] - A2_ref.float().sum(1)).abs().max()) < 1e-4
    assert float((sc[1, :12] - A.float() @ beta.float()).abs().max()) < 1e-4
    c = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
    vt = torch.full((B, heads, D, T), 7.0, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(wf), c=p(c), lora_down=p(A2), lora_up=p(up), lora_scale=p(scale), lda0=K, ca0=K, mode=0, stride=1,
                     ldw=K, M=M, N=N, K=K, ldc=N, rows_per_sample=T, ld_t=12, lora_groups=3, lora_rank=12, tile=tile,
                     ln_in=p(chunks), ln_in_chunks=K // 64, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5,
                     ln_lora_s=sc.data_ptr(), ln_lora_c=sc.data_ptr() + 64)
    if tile == 0x8014:
        d.vt_out, d.vt_col0, d.vt_D, d.vt_heads, d.vt_tokens, d.vt_ld = p(vt), 2 * C, D, heads, T, T
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    ln = bf(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)).float()
    Tt = bf(ln @ A.float().t() * 0.5).float()
    ref = ln @ w.float().t()
    for g in range(3):
        ref[:, g * C:(g + 1) * C] += Tt[:, 4 * g:4 * g + 4] @ up.float()[g * C:(g + 1) * C].t()
    if tile == 0x8014:
        report(f"ln + fused adapter tile{tile:x} off{offset} q|k", c[:, :2 * C], ref[:, :2 * C], TOL)
        report(f"ln + fused adapter tile{tile:x} off{offset} v^T", vt, ref[:, 2 * C:].reshape(B, T, heads, D).permute(0, 2, 3, 1), TOL)
    else:
        report(f"ln + fused adapter tile{tile:x} off{offset}", c, ref, TOL)
    # the adapter term is really there (and normalised): without it the result is off by its size
    eff = ((ref - ln @ w.float().t()).norm() / ref.norm()).item()
    assert eff > 0.05
    d.tile = 0x4012
    with pytest.raises(lib.SlidersHipError, match="ln_in with a fused adapter"):
        lib.call(lib.OP_GEMM, d, stream())


@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("Tk", [77, 64, 96, 5])
def test_gemm_fused_cross_attention(dev, fold, Tk):
    """attn2.to_q + the cross-attention behind it in one launch (slh_gemm_desc.xa_*): every wave of the 128 x 128 ring tile holds
    32 rows x one head of Q in its accumulators and runs softmax(Q K^T / sqrt(64)) V for the <= 96 text keys in registers.
    Against the reference's op sequence in fp32 (Q rounded to bf16 like the tensor the unfused path stores), and against the
    two-launch path of the library (slh_gemm + slh_attn_fwd); plain + bias, and with the LayerNorm fold on the A operand
    (norm2 of the no-grad passes).  Keys >= Tk masked; batch of 2 with different text per sample."""
    from sliders_amd.weights import fold_layernorm
    torch.manual_seed(Tk + int(fold))
    B, Tq, H, D = 2, 256, 5, 64
    C = H * D
    M, K = B * Tq, 384
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(C, K, device=dev) / math.sqrt(K))
    bias = bf(torch.randn(C, device=dev) * 0.2)
    kk = bf(torch.randn(B, Tk, C + 64, device=dev))[:, :, 32:32 + C]          # a column window of a wider matrix (ld > C, 64-byte offset)
    vv = bf(torch.randn(B, Tk, C, device=dev))
    ldt, Hall = 128, H + 3                                                     # V^T of this layer sits inside a wider head array
    vt_all = torch.zeros(B, Hall, D, ldt, device=dev, dtype=torch.bfloat16)
    vt_all[:, 2:2 + H, :, :Tk] = vv.view(B, Tk, H, D).permute(0, 2, 3, 1)
    vt = vt_all[:, 2:]
    scale = D ** -0.5
    c = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    kw = {}
    if fold:
        gamma, beta = bf(torch.randn(K, device=dev) * 0.5 + 1.0), bf(torch.randn(K, device=dev) * 0.3)
        # the producer's chunk statistics of x (here computed by the library from an identity-like product is overkill: restate)
        xc = x.float().view(M, K // 64, 64)
        mean = xc.mean(-1)
        chunks = torch.stack([mean, ((xc - mean[..., None]) ** 2).sum(-1)], -1).permute(1, 0, 2).contiguous()
        wf, sv, bp = fold_layernorm(w, None, gamma, beta)
        kw = dict(w=p(wf), ln_in=p(chunks), ln_in_chunks=K // 64, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5)
        q_ref = bf(bf(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)).float() @ w.float().t())
    else:
        kw = dict(w=p(w), bias=p(bias))
        q_ref = bf(x.float() @ w.float().t() + bias.float())
    d = lib.GemmDesc(a0=p(x), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=C, K=K, ldc=C, rows_per_sample=M, tile=0x4412,
                     xa_k=kk.data_ptr(), xa_vt=vt.data_ptr(), xa_tk=Tk, xa_tq=Tq, xa_ldk=kk.stride(1), xa_ldvt=ldt, xa_vt_heads=Hall,
                     xa_scale=scale, **kw)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    qh = q_ref.float().view(B, Tq, H, D).permute(0, 2, 1, 3)
    kh = kk.float().reshape(B, Tk, H, D).permute(0, 2, 1, 3)
    vh = vv.float().view(B, Tk, H, D).permute(0, 2, 1, 3)
    att = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh
    ref = att.permute(0, 2, 1, 3).reshape(M, C)
    report(f"fused cross-attention Tk{Tk} fold{int(fold)}", c, ref, TOL)
    # the two-launch path on the same operands
    q2 = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    d2 = lib.GemmDesc(a0=p(x), c=p(q2), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=C, K=K, ldc=C, rows_per_sample=M, tile=0x4412, **kw)
    lib.call(lib.OP_GEMM, d2, stream())
    o2 = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    da = lib.AttnDesc(q=p(q2), k=kk.data_ptr(), vt=vt.data_ptr(), o=p(o2), B=B, H=H, Tq=Tq, Tk=Tk, ldq=C, ldk=kk.stride(1), ldvt=ldt,
                      ldo=C, scale=scale, D=D, vt_batch_heads=Hall)
    lib.call(lib.OP_ATTN_FWD, da, stream())
    torch.cuda.synchronize()
    assert ((q2.float() - q_ref.float()).norm() / q_ref.float().norm()).item() < 5e-3
    rel = ((c.float() - o2.float()).norm() / o2.float().norm()).item()
    print(f"[parity] fused vs two-launch cross-attention Tk{Tk} fold{int(fold)}: rel {rel:.2e}")
    assert rel < 3e-3
    # descriptors the kernel cannot honour
    d.tile = 0x4012
    with pytest.raises(lib.SlidersHipError, match="cross-attention"):
        lib.call(lib.OP_GEMM, d, stream())
    d.tile = 0x4412
    d.residual, d.ld_res = p(q2), C
    with pytest.raises(lib.SlidersHipError, match="xa_k"):
        lib.call(lib.OP_GEMM, d, stream())


def _perm_cols(g):
    from sliders_amd.weights import _geglu_perm
    return _geglu_perm(g.t().contiguous()).t().contiguous()


@pytest.mark.parametrize("stride,xform", [(1, 0), (2, 0), (1, 1), (1, 2)])
@pytest.mark.parametrize("tile", [0x22, 0x11, 0x312, 0x311, 0x4022, 0x4312, 0x4011, 0x422, 0x411, 0x4412, 0x4322, 0x8042, 0x8013, 0x8014, 0x8015])
def test_gemm_conv3x3(dev, stride, xform, tile):
    torch.manual_seed(3 + stride + xform)
    B, H, W, Ci, Co = 2, 12, 20, 128, 192
    img = bf(torch.randn(B, Ci, H, W, device=dev))
    w4 = bf(torch.randn(Co, Ci, 3, 3, device=dev) / math.sqrt(9 * Ci))
    bias = bf(torch.randn(Co, device=dev))
    ref_img = _conv_ref(img.float(), w4.float(), stride, up=(xform == 1), dilate=(xform == 2)) + bias.float()[None, :, None, None]
    Ho, Wo = ref_img.shape[2:]
    x = bf(_to_pix(img.float()))
    wp = _pack_conv(w4)
    M = B * Ho * Wo
    c = torch.zeros(M, Co, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bias), c=p(c), lda0=Ci, ca0=Ci, mode=1, batch=B, hs=H, ws=W,
                     src_xform=xform, stride=stride, ho=Ho, wo=Wo, ldw=9 * Ci, M=M, N=Co, K=9 * Ci, ldc=Co,
                     rows_per_sample=Ho * Wo, tile=tile)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm_conv s{stride} x{xform} tile{tile:x}", c, _to_pix(ref_img), TOL)


@pytest.mark.parametrize("tile", [0, 0x422, 0x4412, 0x4322, 0x8042, 0x8015])
def test_gemm_conv_two_source_and_skinny(dev, tile):
    torch.manual_seed(5)
    B, H, W, C0, C1, Co = 2, 16, 16, 128, 64, 128
    i0 = bf(torch.randn(B, C0, H, W, device=dev))
    i1 = bf(torch.randn(B, C1, H, W, device=dev))
    w4 = bf(torch.randn(Co, C0 + C1, 3, 3, device=dev) / math.sqrt(9 * (C0 + C1)))
    ref = _to_pix(_conv_ref(torch.cat([i0, i1], 1).float(), w4.float()))
    x0, x1 = bf(_to_pix(i0.float())), bf(_to_pix(i1.float()))
    M = B * H * W
    c = torch.zeros(M, Co, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x0), a1=p(x1), w=p(_pack_conv(w4)), c=p(c), lda0=C0, lda1=C1, ca0=C0, ca1=C1, mode=1,
                     batch=B, hs=H, ws=W, stride=1, ho=H, wo=W, ldw=9 * (C0 + C1), M=M, N=Co, K=9 * (C0 + C1),
                     ldc=Co, rows_per_sample=H * W, tile=tile)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report("gemm_conv_2src", c, ref, TOL)
    # skinny: LoRA down conv (R=4), stride 2, and the NCHW conv_out form
    for stride, R, kind in ((1, 4, 0), (2, 4, 0), (1, 12, 0), (1, 4, 1)):
        wd = bf(torch.randn(R, C0 + C1, 3, 3, device=dev) / math.sqrt(9 * (C0 + C1)))
        bias = bf(torch.randn(R, device=dev))
        r_img = _conv_ref(torch.cat([i0, i1], 1).float(), wd.float(), stride) + bias.float()[None, :, None, None]
        Ho, Wo = r_img.shape[2:]
        Mo = B * Ho * Wo
        out = torch.zeros(Mo, R, device=dev) if kind == 0 else torch.zeros(B, R, Ho, Wo, device=dev, dtype=torch.bfloat16)
        sd = lib.SkinnyDesc(a0=p(x0), a1=p(x1), w=p(_pack_conv(wd)), bias=p(bias), out=p(out), lda0=C0, lda1=C1,
                            ca0=C0, ca1=C1, mode=1, batch=B, hs=H, ws=W, stride=stride, ho=Ho, wo=Wo, M=Mo, R=R,
                            K=9 * (C0 + C1), ldo=R, out_kind=kind)
        lib.call(lib.OP_SKINNY, sd, stream())
        torch.cuda.synchronize()
        report(f"skinny_conv s{stride} R{R} kind{kind}", out, _to_pix(r_img) if kind == 0 else r_img, 2e-5 if kind == 0 else TOL)
    # dense skinny
    xd = bf(torch.randn(333, 640, device=dev))
    wd = bf(torch.randn(12, 640, device=dev))
    out = torch.zeros(333, 12, device=dev)
    sd = lib.SkinnyDesc(a0=p(xd), w=p(wd), out=p(out), lda0=640, ca0=640, mode=0, stride=1, M=333, R=12, K=640, ldo=12)
    lib.call(lib.OP_SKINNY, sd, stream())
    torch.cuda.synchronize()
    report("skinny_dense", out, xd.float() @ wd.float().t(), 2e-5)


def test_gemv(dev):
    torch.manual_seed(6)
    nb, N, K = 2, 1000, 1280
    x = bf(torch.randn(nb, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    b = bf(torch.randn(N, device=dev))
    add = bf(torch.randn(nb, N, device=dev))
    T = torch.randn(nb, 16, device=dev)
    tcol = torch.randint(0, 4, (N,), device=dev, dtype=torch.int32) * 4
    up = bf(torch.randn(N, 4, device=dev))
    scale = torch.tensor([0.25], device=dev)
    y = torch.zeros(nb, N, device=dev, dtype=torch.bfloat16)
    d = lib.GemvDesc(x=p(x), w=p(w), bias=p(b), addend=p(add), lora_t=p(T), lora_tcol=p(tcol), lora_up=p(up),
                     lora_scale=p(scale), y=p(y), nb=nb, N=N, K=K, ldx=K, ld_add=N, ld_t=16, ldy=N, in_act=1, out_f32=0)
    lib.call(lib.OP_GEMV, d, stream())
    torch.cuda.synchronize()
    xs = bf(F.silu(x.float())).float()
    ref = xs @ w.float().t() + b.float()
    idx = tcol.long()[None, :, None] + torch.arange(4, device=dev)[None, None, :]
    tsel = torch.gather(T[:, None, :].expand(nb, N, 16), 2, idx.expand(nb, N, 4))
    ref = ref + 0.25 * (tsel * up.float()[None]).sum(-1)
    ref = bf(ref).float() + add.float()
    report("gemv", y, ref, TOL)


@pytest.mark.parametrize("B,C,HW", [(1, 1280, 1024), (3, 1280, 1024), (5, 640, 1024), (1, 320, 4096), (3, 1920, 256), (16, 1280, 1024)])
def test_groupnorm_one_launch_sibling_counts(dev, B, C, HW):
    """slh_gn_fused on cache-resident slabs: 8 / 4 / 2 / 1 sibling workgroups per (sample, group) by the batch size - every
    sibling reduces the whole slab in the same order, so the output must not depend on the split: compared with the two-launch
    form (same tolerance as the one-launch test above) and with float64."""
    torch.manual_seed(11)
    assert lib.gn_fused_ok(C, HW, 32) == 2
    x = bf(torch.randn(B * HW, C, device=dev) * 1.5 + 3.0)
    g, bta = bf(torch.randn(C, device=dev)), bf(torch.randn(C, device=dev))
    st1, st2 = torch.full((B, 32, 2), float("nan"), device=dev), torch.full((B, 32, 2), float("nan"), device=dev)
    y1, y2 = torch.zeros_like(x), torch.zeros_like(x)
    part, ticket = _gn_workspace(dev, B, C, HW)
    d1 = lib.GnDesc(x0=p(x), gamma=p(g), beta=p(bta), stats=p(st1), y=p(y1), ldx0=C, c0=C, batch=B, hw=HW, groups=32, ldy=C, eps=1e-6, act=1)
    d2 = lib.GnDesc(x0=p(x), gamma=p(g), beta=p(bta), stats=p(st2), y=p(y2), ldx0=C, c0=C, batch=B, hw=HW, groups=32, ldy=C, eps=1e-6, act=1,
                    partial=p(part), ticket=p(ticket))
    lib.call(lib.OP_GN_FUSED, d1, stream())
    lib.call(lib.OP_GN_STATS, d2, stream())
    lib.call(lib.OP_GN_APPLY, d2, stream())
    torch.cuda.synchronize()
    ximg = x.float().view(B, HW, C).permute(0, 2, 1)
    ref = F.silu(bf(F.group_norm(ximg.double(), 32, g.double(), bta.double(), 1e-6).float()).float()).permute(0, 2, 1).reshape(B * HW, C)
    report(f"groupnorm one-launch B{B} C{C} hw{HW}", y1, ref, TOL)
    assert torch.allclose(st1, st2, rtol=1e-5, atol=1e-6)
    assert float((y1.float() - y2.float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


def _gn_workspace(dev, B, C, HW, G=32):
    """Partial-sum scratch (deliberately filled with garbage: the kernel must not depend on its contents) + zeroed tickets."""
    prow, ntick = lib.gn_workspace(C, HW, G)
    return torch.full((B, prow, G, 2), float("nan"), device=dev), torch.zeros(B, ntick, dtype=torch.int32, device=dev)


@pytest.mark.parametrize("C0,C1,act,mean,std,HW", [(320, 0, 1, 0.5, 2.0, 300), (64, 0, 0, 0.5, 2.0, 300), (1280, 640, 1, 0.5, 2.0, 300),
                                                (2560, 0, 1, 0.5, 2.0, 300), (960, 0, 1, 0.5, 2.0, 300),
                                                # DC offset >> spread (real checkpoints' early resnet activations): a one-pass
                                                # E[x^2] - E[x]^2 in fp32 loses the variance here (bf16 spacing at 50 is 0.25, at 256 is 2)
                                                (320, 0, 1, 50.0, 1.0, 300), (640, 0, 0, 256.0, 4.0, 300), (1280, 640, 1, -50.0, 1.0, 300),
                                                # 8x8 latents: the one-launch form (slh_gn_fused) is what the planner emits
                                                (1280, 0, 1, 0.5, 2.0, 64), (2560, 0, 1, 30.0, 1.0, 64), (1280, 1280, 0, 0.5, 2.0, 64),
                                                # cache-resident slabs (32x32 / 64x64 latents): sibling workgroups, one launch (round 5):
                                                # 16- / 8- / 4-byte accesses by the alignment of a group, a group straddling the concat
                                                (1280, 0, 1, 0.5, 2.0, 1024), (1280, 640, 1, -50.0, 1.0, 1024), (640, 320, 0, 0.5, 2.0, 1024),
                                                (640, 0, 1, 256.0, 4.0, 4096), (320, 0, 1, 50.0, 1.0, 1024), (1280, 1280, 1, 0.5, 2.0, 1024)])
def test_groupnorm(dev, C0, C1, act, mean, std, HW):
    torch.manual_seed(7)
    B = 2
    C = C0 + C1
    x0 = bf(torch.randn(B * HW, C0, device=dev) * std + mean)
    x1 = bf(torch.randn(B * HW, C1, device=dev) * std + mean) if C1 else None
    g, bta = bf(torch.randn(C, device=dev)), bf(torch.randn(C, device=dev))
    stats = torch.full((B, 32, 2), float("nan"), device=dev)
    part, ticket = _gn_workspace(dev, B, C, HW)
    y = torch.zeros(B * HW, C, device=dev, dtype=torch.bfloat16)
    d = lib.GnDesc(x0=p(x0), x1=p(x1), gamma=p(g), beta=p(bta), stats=p(stats), y=p(y), ldx0=C0, ldx1=C1, c0=C0, c1=C1,
                   batch=B, hw=HW, groups=32, ldy=C, eps=1e-5, act=act, partial=p(part), ticket=p(ticket))
    lib.call(lib.OP_GN_STATS, d, stream())
    lib.call(lib.OP_GN_APPLY, d, stream())
    torch.cuda.synchronize()
    assert int(ticket.abs().sum()) == 0, "the last arriver re-arms the tickets"
    xc = torch.cat([x0, x1], 1) if C1 else x0
    ximg = xc.float().view(B, HW, C).permute(0, 2, 1)
    # the statistics themselves against fp64
    xg = ximg.double().reshape(B, 32, -1)
    mref, vref = xg.mean(-1), xg.var(-1, unbiased=False)
    assert float((stats[..., 0].double() - mref).abs().max() / mref.abs().max().clamp_min(1.0)) < 1e-6
    rref = 1.0 / torch.sqrt(vref + 1e-5)
    assert float(((stats[..., 1].double() - rref) / rref).abs().max()) < 1e-5, "rstd (cancellation-safe statistics)"
    ref = F.group_norm(ximg.double(), 32, g.double(), bta.double(), 1e-5).float()
    if act:
        ref = F.silu(bf(ref).float())
    ref = ref.permute(0, 2, 1).reshape(B * HW, C)
    report(f"groupnorm C{C0}+{C1} act{act} mean{mean}", y, ref, TOL)
    # bit-reproducible: the reduction order is fixed (no fp32 atomics)
    s0, y0 = stats.clone(), y.clone()
    for _ in range(3):
        stats.fill_(float("nan"))
        lib.call(lib.OP_GN_STATS, d, stream())
        lib.call(lib.OP_GN_APPLY, d, stream())
        torch.cuda.synchronize()
        assert torch.equal(stats, s0) and torch.equal(y, y0)
    # in-place is refused: every workgroup of slh_gn_apply re-reads the statistics' pivot from x
    if C1 == 0:
        d_alias = lib.GnDesc(x0=p(x0), gamma=p(g), beta=p(bta), stats=p(stats), y=p(x0), ldx0=C0, c0=C0, batch=B, hw=HW, groups=32,
                             ldy=C, eps=1e-5, act=act, partial=p(part), ticket=p(ticket))
        with pytest.raises(lib.SlidersHipError, match="alias"):
            lib.call(lib.OP_GN_APPLY, d_alias, stream())
    # the single-launch form for small tensors (one workgroup per group set): same statistics, same output
    if lib.gn_fused_ok(C, HW, 32):
        sf = torch.full((B, 32, 2), float("nan"), device=dev)
        yf = torch.zeros_like(y)
        df = lib.GnDesc(x0=p(x0), x1=p(x1), gamma=p(g), beta=p(bta), stats=p(sf), y=p(yf), ldx0=C0, ldx1=C1, c0=C0, c1=C1,
                        batch=B, hw=HW, groups=32, ldy=C, eps=1e-5, act=act)
        lib.call(lib.OP_GN_FUSED, df, stream())
        torch.cuda.synchronize()
        assert float((sf[..., 0].double() - mref).abs().max() / mref.abs().max().clamp_min(1.0)) < 1e-6
        assert float(((sf[..., 1].double() - rref) / rref).abs().max()) < 1e-5
        report(f"groupnorm fused C{C0}+{C1} act{act} mean{mean}", yf, ref, TOL)
        assert float((yf.float() - y.float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()), "one-launch vs two-launch form"
        y1 = yf.clone()
        lib.call(lib.OP_GN_FUSED, df, stream())
        torch.cuda.synchronize()
        assert torch.equal(yf, y1)
    else:
        with pytest.raises(lib.SlidersHipError, match="one-launch"):
            lib.call(lib.OP_GN_FUSED, d, stream())
        # the one-launch form has no in-place variant either (sibling workgroups re-read the slab)
        if C1 == 0 and lib.gn_fused_ok(C, HW, 32) == 2:
            with pytest.raises(lib.SlidersHipError, match="alias"):
                lib.call(lib.OP_GN_FUSED, lib.GnDesc(x0=p(x0), gamma=p(g), beta=p(bta), stats=p(sf), y=p(x0), ldx0=C0, c0=C0, batch=B,
                                                     hw=HW, groups=32, ldy=C, eps=1e-5, act=act), stream())
    # backward (dx only)
    dy = bf(torch.randn(B * HW, C, device=dev))
    bst = torch.full((B, 32, 2), float("nan"), device=dev)
    bpart, bticket = _gn_workspace(dev, B, C, HW)
    dx0 = torch.zeros(B * HW, C0, device=dev, dtype=torch.bfloat16)
    dx1 = torch.zeros(B * HW, max(C1, 8), device=dev, dtype=torch.bfloat16)
    bd = lib.GnBwdDesc(x0=p(x0), x1=p(x1), gamma=p(g), beta=p(bta), stats=p(stats), bstats=p(bst), dy=p(dy),
                       dx0=p(dx0), dx1=p(dx1) if C1 else 0, ldx0=C0, ldx1=C1, c0=C0, c1=C1, batch=B, hw=HW, groups=32,
                       lddy=C, lddx0=C0, lddx1=max(C1, 8), eps=1e-5, act=act, bpartial=p(bpart), bticket=p(bticket))
    lib.call(lib.OP_GN_BWD_STATS, bd, stream())
    lib.call(lib.OP_GN_BWD_APPLY, bd, stream())
    torch.cuda.synchronize()
    xi = ximg.double().clone().requires_grad_(True)
    o = F.group_norm(xi, 32, g.double(), bta.double(), 1e-5)
    if act:
        o = F.silu(o)
    o.backward(dy.double().view(B, HW, C).permute(0, 2, 1))
    gref = xi.grad.permute(0, 2, 1).reshape(B * HW, C).float()
    got = torch.cat([dx0, dx1[:, :C1]], 1) if C1 else dx0
    report(f"groupnorm_bwd C{C0}+{C1} act{act} mean{mean}", got, gref, 1.5e-2)
    b0, g0 = bst.clone(), got.clone()
    bst.fill_(float("nan"))
    lib.call(lib.OP_GN_BWD_STATS, bd, stream())
    lib.call(lib.OP_GN_BWD_APPLY, bd, stream())
    torch.cuda.synchronize()
    assert torch.equal(bst, b0) and torch.equal(torch.cat([dx0, dx1[:, :C1]], 1) if C1 else dx0, g0)


@pytest.mark.parametrize("C", [64, 320, 640, 1280])
def test_layernorm(dev, C):
    torch.manual_seed(8)
    M = 777
    x = bf(torch.randn(M, C, device=dev) * 3 + 1)
    g, b = bf(torch.randn(C, device=dev)), bf(torch.randn(C, device=dev))
    y = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    mr = torch.zeros(M, 2, device=dev)
    lib.call(lib.OP_LAYERNORM, lib.LnDesc(x=p(x), gamma=p(g), beta=p(b), y=p(y), mean_rstd=p(mr), M=M, C=C, ldx=C,
                                          ldy=C, eps=1e-5), stream())
    torch.cuda.synchronize()
    report(f"layernorm C{C}", y, F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5), TOL)
    dy = bf(torch.randn(M, C, device=dev))
    dx = bf(torch.randn(M, C, device=dev))
    dx_init = dx.clone()
    lib.call(lib.OP_LAYERNORM_BWD, lib.LnBwdDesc(x=p(x), gamma=p(g), dy=p(dy), mean_rstd=p(mr), dx=p(dx), M=M, C=C,
                                                 ldx=C, lddy=C, lddx=C, accumulate=1), stream())
    torch.cuda.synchronize()
    xi = x.float().clone().requires_grad_(True)
    F.layer_norm(xi, (C,), g.float(), b.float(), 1e-5).backward(dy.float())
    report(f"layernorm_bwd C{C}", dx, xi.grad + dx_init.float(), 1.5e-2)


# (2, 5 / 10 / 20, 1024, 1024, 64) and (2, 10, 512, 256, 64) run the key-split form (attn_fwd_ks_kernel: 64-query grids of 129..768
# workgroups, D = 64, Tk a multiple of 128), (3, 20, 1024, 1024, 64) - the frozen B = 3 pass - the plain 64-query form
@pytest.mark.parametrize("B,H,Tq,Tk,D", [(2, 5, 1024, 1024, 64), (2, 20, 1024, 1024, 64), (2, 10, 1024, 1024, 64), (2, 10, 512, 256, 64),
                                          (3, 20, 1024, 1024, 64), (1, 3, 192, 192, 64), (2, 4, 256, 77, 64),
                                          (1, 2, 100, 333, 64), (4, 8, 2048, 2048, 64), (2, 8, 1024, 1024, 40),
                                          (1, 8, 256, 77, 80), (2, 8, 64, 64, 160), (1, 4, 200, 77, 160), (1, 8, 96, 96, 8)])
def test_attention_fwd(dev, B, H, Tq, Tk, D):
    torch.manual_seed(9)
    C = H * D
    self_attn = Tq == Tk
    if self_attn:
        qkv = bf(torch.randn(B * Tq, 3 * C, device=dev))
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        ldq = ldk = ldv = 3 * C
    else:
        qb = bf(torch.randn(B * Tq, C, device=dev))
        kvb = bf(torch.randn(B * Tk, 2 * C, device=dev))
        q, k, v = qb, kvb[:, :C], kvb[:, C:]
        ldq, ldk, ldv = C, 2 * C, 2 * C
    ldt = (Tk + 63) // 64 * 64
    Dp = (D + 63) // 64 * 64
    vt = torch.full((B, H, Dp, ldt), float("nan"), device=dev, dtype=torch.bfloat16)
    lib.call(lib.OP_TRANSPOSE_HEADS, lib.TransposeDesc(src=v.data_ptr(), dst=p(vt), B=B, H=H, T=Tk, ld=ldv, ldt=ldt, D=D), stream())
    torch.cuda.synchronize()
    vref = v.reshape(B, Tk, H, D).permute(0, 2, 3, 1)
    assert torch.equal(vt[:, :, :D, :Tk], vref), "transpose_heads mismatch"
    assert (vt[..., Tk:] == 0).all() and (vt[:, :, D:] == 0).all(), "transpose_heads padding must be zero"
    o = torch.zeros(B * Tq, C, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, Tq, device=dev)
    d = lib.AttnDesc(q=q.data_ptr(), k=k.data_ptr(), vt=p(vt), o=p(o), lse=p(lse), B=B, H=H, Tq=Tq, Tk=Tk, ldq=ldq,
                     ldk=ldk, ldvt=ldt, ldo=C, scale=D ** -0.5, D=D)
    lib.call(lib.OP_ATTN_FWD, d, stream())
    torch.cuda.synchronize()
    qf = q.float().reshape(B, Tq, H, D).transpose(1, 2)
    kf = k.float().reshape(B, Tk, H, D).transpose(1, 2)
    vf = v.float().reshape(B, Tk, H, D).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) * D ** -0.5
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B * Tq, C)
    report(f"attn_fwd B{B} H{H} Tq{Tq} Tk{Tk} D{D}", o, ref, 8e-3)
    report("attn_lse", lse, torch.logsumexp(s, -1) * 1.4426950408889634, 1e-4)
    # weight touch (slh_attn_desc.pf_*): a hint - extra workgroups stream a byte range and leave; results bit-identical, any odd size
    wbuf = torch.randn(3_000_017, device=dev)
    o2 = torch.zeros_like(o)
    d.o, d.pf_ptr, d.pf_bytes = p(o2), p(wbuf), wbuf.numel() * 4
    assert lib.attn_carries_touch(d) == ((B, H, Tq, Tk, D) in ((2, 5, 1024, 1024, 64), (2, 20, 1024, 1024, 64), (2, 10, 1024, 1024, 64),
                                                              (2, 10, 512, 256, 64)))
    lib.call(lib.OP_ATTN_FWD, d, stream())
    torch.cuda.synchronize()
    assert torch.equal(o, o2)


def test_timestep_embed_and_conv_in(dev):
    torch.manual_seed(10)
    vals = torch.tensor([[999.0, 1024.0, 0.0], [20.0, 512.0, 3.0]], device=dev)
    out = torch.zeros(2, 16 + 3 * 256, device=dev, dtype=torch.bfloat16)
    lib.call(lib.OP_TEMBED, lib.TembedDesc(vals=p(vals), out=p(out), nb=2, n_vals=3, dim=256, ldo=out.shape[1], col0=16), stream())
    torch.cuda.synchronize()
    half = 128
    fr = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev) / half)
    arg = vals[:, :, None] * fr
    ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1).reshape(2, -1)
    report("timestep_embed", out[:, 16:], ref, 4e-3)
    B, H, W, Co = 2, 24, 16, 320
    x = bf(torch.randn(B, 4, H, W, device=dev))
    w4 = bf(torch.randn(Co, 4, 3, 3, device=dev) / 6)
    bias = bf(torch.randn(Co, device=dev))
    y = torch.zeros(B * H * W, Co, device=dev, dtype=torch.bfloat16)
    lib.call(lib.OP_CONV_IN, lib.ConvInDesc(x=p(x), w=p(_pack_conv(w4)), bias=p(bias), y=p(y), batch=B, cin=4, h=H, wd=W,
                                            cout=Co, ldy=Co), stream())
    torch.cuda.synchronize()
    report("conv_in", y, _to_pix(F.conv2d(x.float(), w4.float(), bias.float(), padding=1)), TOL)


def test_cfg_ddim_bit_exact(dev):
    torch.manual_seed(11)
    from oracle.ddim_oracle import DDIMScheduler
    sch = DDIMScheduler()
    sch.set_timesteps(50)
    nb, chw = 1, 4 * 32 * 32
    eps = bf(torch.randn(2 * nb, chw, device=dev))
    x = bf(torch.randn(nb, chw, device=dev))
    for t in (980, 500, 0):
        cb, ca, cp, cd = sch.step_coefficients(t)
        out = torch.zeros(nb, chw, device=dev, dtype=torch.bfloat16)
        d = lib.CfgDdimDesc(eps=p(eps), x=p(x), out=p(out), nb=nb, chw=chw, guidance=3.0, c_sqrt_beta_t=cb,
                            c_inv_sqrt_alpha_t=ca, c_sqrt_alpha_prev=cp, c_dir=cd, do_step=1)
        lib.call(lib.OP_CFG_DDIM, d, stream())
        torch.cuda.synchronize()
        e_c, x_c = eps.cpu(), x.cpu()
        u, tt = e_c.chunk(2)
        guided = u + 3 * (tt - u)                       # train_util.py:166-169 in bf16
        ref = sch.step(guided, t, x_c).prev_sample      # oracle DDIM, bf16 tensors x fp32 0-dim scalars
        assert ref.dtype == torch.bfloat16
        assert torch.equal(out.cpu(), ref), f"cfg+ddim not bit-exact at t={t}: max diff {(out.cpu().float() - ref.float()).abs().max()}"
    print("[parity] cfg_ddim: bit-exact at t=980,500,0")


def test_cfg_ddim_v_prediction_bit_exact(dev):
    """SD-2.x 768-v (pretrained_model.v_pred): the fused CFG + DDIM step with v-prediction against the oracle scheduler's
    restatement of the diffusers tensor ops (bf16 tensors x fp32 0-dim scalars), bit for bit."""
    torch.manual_seed(12)
    from oracle.ddim_oracle import DDIMScheduler
    from sliders_amd.ddim import DDIMSchedule
    sch = DDIMScheduler(prediction_type="v_prediction")
    sch.set_timesteps(50)
    prod = DDIMSchedule(prediction_type="v_prediction")
    nb, chw = 2, 4 * 32 * 32
    eps = bf(torch.randn(2 * nb, chw, device=dev))
    x = bf(torch.randn(nb, chw, device=dev))
    for t in (980, 500, 0):
        out = torch.zeros(nb, chw, device=dev, dtype=torch.bfloat16)
        d = lib.CfgDdimDesc(eps=p(eps), x=p(x), out=p(out), nb=nb, chw=chw, guidance=3.0, **prod.step_fields(t, 50))
        lib.call(lib.OP_CFG_DDIM, d, stream())
        torch.cuda.synchronize()
        u, tt = eps.cpu().chunk(2)
        ref = sch.step(u + 3 * (tt - u), t, x.cpu()).prev_sample
        assert ref.dtype == torch.bfloat16
        assert torch.equal(out.cpu(), ref), f"v-prediction cfg+ddim not bit-exact at t={t}"
    print("[parity] cfg_ddim v_prediction: bit-exact at t=980,500,0")


def test_loss_and_grad(dev):
    torch.manual_seed(12)
    n = 4 * 64 * 64
    tg, po, ne, un = [bf(torch.randn(n, device=dev)) for _ in range(4)]
    for erase in (0, 1):
        loss = torch.zeros(1, device=dev)
        dt = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        lib.call(lib.OP_LOSS, lib.LossDesc(target=p(tg), positive=p(po), neutral=p(ne), uncond=p(un), loss=p(loss),
                                           dtarget=p(dt), n=n, guidance=4.0, erase=erase), stream())
        torch.cuda.synchronize()
        t_c = tg.cpu().clone().requires_grad_(True)
        y = ne.cpu() - 4.0 * (po.cpu() - un.cpu()) if erase else ne.cpu() + 4.0 * (po.cpu() - un.cpu())
        l = F.mse_loss(t_c, y)
        l.backward()
        assert torch.equal(dt.cpu(), t_c.grad), "loss gradient not bit-exact vs torch bf16 autograd"
        assert abs(loss.item() - l.float().item()) <= 4e-3 * abs(l.float().item()) + 1e-6
    print("[parity] guidance loss: gradient bit-exact, value within bf16 rounding")


def test_adamw_bit_exact(dev):
    torch.manual_seed(13)
    n = 10007
    p0 = bf(torch.randn(n) * 0.05)
    param = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([param], lr=2e-4)
    dp = p0.clone().to(dev)
    m = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    v = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    mism = 0
    for step in range(1, 6):
        g = torch.randn(n) * 1e-3
        param.grad = bf(g)
        opt.step()
        gd = bf(g).float().to(dev)
        d = lib.AdamwDesc(param=p(dp), exp_avg=p(m), exp_avg_sq=p(v), grad=p(gd), n=n, lr=2e-4, beta1=0.9, beta2=0.999,
                          eps=1e-8, weight_decay=0.01, step=step, grad_scale=1.0)
        lib.call(lib.OP_ADAMW, d, stream())
        torch.cuda.synchronize()
        mism = (dp.cpu().view(torch.int16) != param.data.view(torch.int16)).sum().item()
        print(f"[parity] adamw step {step}: {mism}/{n} params differ from torch.optim.AdamW(bf16, CPU)")
    assert mism == 0, "AdamW not bit-exact"


def test_adamw_fp32_state_bit_exact(dev):
    """slh_adamw with f32_state (train.precision: float32 - fp32 adapter parameters and moments, train_lora_xl.py:60-61, 84-90): the
    fp32 master and both moments against torch.optim.AdamW stepping fp32 tensors ON THE DEVICE (the reference trains on the GPU:
    the default foreach path), ten steps with weight decay and a changing lr; the bf16 copy the UNet kernels read is the rounded
    master.  Exactness is reported per step; the assertion allows the last fp32 bit on a small fraction (an fma contracted
    differently inside one of torch's foreach functors would show exactly there) and nothing else."""
    torch.manual_seed(14)
    n = 10007
    p0 = torch.randn(n) * 0.05
    param = torch.nn.Parameter(p0.clone().to(dev))
    opt = torch.optim.AdamW([param], lr=2e-4, weight_decay=0.01)
    master = p0.clone().to(dev)
    lo = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    worst = 0
    for step in range(1, 11):
        lr = 2e-4 * (1.0 - 0.05 * step)
        for grp in opt.param_groups:
            grp["lr"] = lr
        g = (torch.randn(n) * 1e-3).to(dev)
        param.grad = g.clone()
        opt.step()
        d = lib.AdamwDesc(param=p(master), exp_avg=p(m), exp_avg_sq=p(v), grad=p(g), n=n, lr=lr, beta1=0.9, beta2=0.999,
                          eps=1e-8, weight_decay=0.01, step=step, grad_scale=1.0, param_lo=p(lo), f32_state=1)
        lib.call(lib.OP_ADAMW, d, stream())
        torch.cuda.synchronize()
        st = opt.state[param]
        diffs = {}
        for nm, a, b in (("param", master, param.data), ("exp_avg", m, st["exp_avg"]), ("exp_avg_sq", v, st["exp_avg_sq"])):
            ulps = (a.view(torch.int32).long() - b.view(torch.int32).long()).abs()
            diffs[nm] = (int((ulps > 0).sum()), int(ulps.max()))
        print(f"[parity] adamw fp32 state step {step}: (elements differing, worst fp32 ulp) vs torch.optim.AdamW(fp32, cuda) = {diffs}")
        worst = max(worst, max(w for _, w in diffs.values()))
        assert torch.equal(lo, master.to(torch.bfloat16)), "param_lo is the rounded master"
        assert all(cnt <= n // 50 and w <= 2 for cnt, w in diffs.values()), diffs
    d.param_lo = 0
    with pytest.raises(lib.SlidersHipError, match="param_lo"):
        lib.call(lib.OP_ADAMW, d, stream())
    print(f"[parity] adamw fp32 state: worst difference over 10 steps {worst} fp32 ulp")


def test_lion_bit_exact(dev):
    """slh_lion against the oracle's restatement of lion_pytorch 0.1.2, five steps with weight decay, including exact-zero
    updates (sign(0) = 0).  The oracle's tensor ops run ON THE GPU here: the reference trains on cuda, where
    `add(t, alpha=a)` keeps `a` in fp32 opmath; torch's CPU kernel rounds `a` to the tensor dtype (bf16) first, which
    moves ~18 % of the moments by one ulp - the kernel restates the device semantics."""
    from oracle.optim_oracle import Lion
    torch.manual_seed(15)
    n = 10007
    p0 = bf(torch.randn(n) * 0.05)
    param = torch.nn.Parameter(p0.clone().to(dev))
    opt = Lion([param], lr=1e-4, betas=(0.9, 0.99), weight_decay=0.05)
    dp = p0.clone().to(dev)
    m = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    for step in range(1, 6):
        g = torch.randn(n) * 1e-3
        g[:17] = 0.0
        param.grad = bf(g).to(dev)
        opt.step()
        gd = bf(g).float().to(dev)
        d = lib.LionDesc(param=p(dp), exp_avg=p(m), grad=p(gd), n=n, lr=1e-4, beta1=0.9, beta2=0.99, weight_decay=0.05,
                         grad_scale=1.0)
        lib.call(lib.OP_LION, d, stream())
        torch.cuda.synchronize()
        mism = (dp.view(torch.int16) != param.data.view(torch.int16)).sum().item()
        mm = (m.view(torch.int16) != opt.state[id(param)]["exp_avg"].view(torch.int16)).sum().item()
        print(f"[parity] lion step {step}: {mism}/{n} params, {mm}/{n} moments differ from the oracle (bf16 torch ops on the GPU)")
        assert mism == 0 and mm == 0, "Lion not bit-exact"


def test_lion_optimizer_class_matches_oracle(dev):
    """train_util.get_optimizer("lion") (train_util.py:365-368) hands the reference-shaped loop a torch.optim.Optimizer
    whose step is slh_lion: bit-equal to the oracle class on device tensors, two parameter tensors, three steps."""
    from oracle.optim_oracle import Lion as OracleLion
    from sliders_amd.train_util import get_optimizer
    torch.manual_seed(16)
    shapes = [(4, 320), (1280, 4)]
    ps = [torch.nn.Parameter(bf(torch.randn(*sh) * 0.05).to(dev)) for sh in shapes]
    qs = [torch.nn.Parameter(p_.detach().clone()) for p_ in ps]
    opt = get_optimizer("lion")(ps, lr=2e-4, weight_decay=0.01)
    ref = OracleLion(qs, lr=2e-4, betas=(0.9, 0.99), weight_decay=0.01)
    for _ in range(3):
        for p_, q_ in zip(ps, qs):
            g = bf(torch.randn(*p_.shape) * 1e-3).to(dev)
            p_.grad, q_.grad = g, g.clone()
        opt.step(); ref.step()
    torch.cuda.synchronize()
    for p_, q_ in zip(ps, qs):
        assert torch.equal(p_.detach().view(torch.int16), q_.detach().view(torch.int16))
    cpu = torch.nn.Parameter(torch.zeros(4, dtype=torch.bfloat16))
    cpu.grad = torch.zeros(4, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        get_optimizer("lion")([cpu]).step()                  # no CPU fallback


def test_lora_wgrad(dev):
    torch.manual_seed(14)
    M, C, R = 1500, 320, 4
    z = bf(torch.randn(M, C, device=dev))
    v = torch.randn(M, 12, device=dev)
    scale = torch.tensor([0.25], device=dev)
    out = torch.zeros(C, 4, device=dev)
    lib.call(lib.OP_WGRAD, lib.WgradDesc(z0=p(z), v=p(v), out=p(out), scale=p(scale), ldz0=C, c0=C, mode=0, stride=1,
                                         M=M, R=4, ldv=12, ldo=4, out_rmajor=0, vgroup_cols=0), stream())
    torch.cuda.synchronize()
    report("wgrad_up_layout", out, 0.25 * z.float().t() @ v[:, :4], 1e-4)
    out = torch.zeros(12, C, device=dev)
    lib.call(lib.OP_WGRAD, lib.WgradDesc(z0=p(z), v=p(v), out=p(out), scale=p(scale), ldz0=C, c0=C, mode=0, stride=1,
                                         M=M, R=12, ldv=12, ldo=C, out_rmajor=1, vgroup_cols=0), stream())
    torch.cuda.synchronize()
    report("wgrad_down_layout_R12", out, 0.25 * v.t() @ z.float(), 1e-4)
    # conv-mode: dA[r][tap][ci] = sum_m im2col(x)[m][tap,ci] * U[m][r]
    B, H, W, Ci = 2, 10, 12, 64
    img = bf(torch.randn(B, Ci, H, W, device=dev))
    U = torch.randn(B * H * W, 4, device=dev)
    out = torch.zeros(4, 9 * Ci, device=dev)
    lib.call(lib.OP_WGRAD, lib.WgradDesc(z0=p(bf(_to_pix(img.float()))), v=p(U), out=p(out), scale=p(scale), ldz0=Ci,
                                         c0=Ci, mode=1, batch=B, hs=H, ws=W, stride=1, ho=H, wo=W, M=B * H * W, R=4,
                                         ldv=4, ldo=9 * Ci, out_rmajor=1, vgroup_cols=0), stream())
    torch.cuda.synchronize()
    wd = torch.zeros(4, Ci, 3, 3, device=dev, requires_grad=True)
    yy = F.conv2d(img.float(), wd, padding=1)
    yy.backward(U.view(B, H, W, 4).permute(0, 3, 1, 2))
    report("wgrad_conv", out, 0.25 * wd.grad.permute(0, 2, 3, 1).reshape(4, -1), 1e-4)


def test_lora_wgrad_fixed_order_is_bit_reproducible(dev):
    """slh_wgrad_desc.slabs / tickets (and slh_batch_desc.slabs / tickets): the M splits of a column block publish their partial
    sums in slabs, the split that arrives last adds them in split order and does the block's one `out +=` - the same bits every
    time, whatever order the workgroups ran in; same numbers as the atomic form up to fp32 re-association; tickets left zero;
    += semantics kept."""
    torch.manual_seed(41)
    scale = torch.tensor([0.5], device=dev)
    cases = []
    for M, C, R, rmajor in ((4096, 1280, 4, 1), (1500, 320, 4, 0), (2048, 640, 12, 1), (64, 64, 4, 0)):
        z = bf(torch.randn(M, C, device=dev))
        v = torch.randn(M, 12, device=dev)
        cases.append((M, C, R, rmajor, z, v))

    def desc(case, out, slabs=None, tickets=None):
        M, C, R, rmajor, z, v = case
        return lib.WgradDesc(z0=p(z), v=p(v), out=p(out), scale=p(scale), ldz0=C, c0=C, mode=0, stride=1, M=M, R=R, ldv=12,
                             ldo=C if rmajor else R, out_rmajor=rmajor, vgroup_cols=0, slabs=p(slabs) if slabs is not None else 0,
                             tickets=p(tickets) if tickets is not None else 0)

    shape = lambda c: (c[2], c[1]) if c[3] else (c[1], c[2])
    # single launches
    for case in cases:
        probe = desc(case, torch.zeros(shape(case), device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev))
        nb = lib.wgrad_single_blocks(probe)
        outs = []
        for rep in range(3):
            out = torch.full(shape(case), 0.25, device=dev)                   # += : starts non-zero
            slabs = torch.full((nb, 256 * case[2]), float("nan"), device=dev)
            tickets = torch.zeros(nb, device=dev, dtype=torch.int32)
            for _ in range(2):                                                  # twice into the same buffer, recycling the workspace
                lib.call(lib.OP_WGRAD, desc(case, out, slabs, tickets), stream())
            torch.cuda.synchronize()
            assert int(tickets.abs().sum()) == 0
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "fixed-order weight gradients must be bit-reproducible"
        ref = 0.25 + 2 * 0.5 * (case[5][:, :case[2]].t() @ case[4].float() if case[3] else case[4].float().t() @ case[5][:, :case[2]])
        report(f"wgrad fixed order M{case[0]} C{case[1]} R{case[2]}", outs[0], ref, 1e-4)
    # one batched launch (R = 4 problems), slab geometry
    r4 = [c for c in cases if c[2] == 4]
    runs = []
    for rep in range(3):
        outs = [torch.full(shape(c), -0.5, device=dev) for c in r4]
        descs = [desc(c, o, torch.zeros(1, device=dev), torch.zeros(1, device=dev)) for c, o in zip(r4, outs)]    # non-NULL slabs: slab geometry
        bd, keep = lib.batch_table(lib.OP_WGRAD_BATCH, descs, dev, arg=4)
        slabs = torch.full((bd.total, 1024), float("nan"), device=dev)
        tickets = torch.zeros(bd.total, device=dev, dtype=torch.int32)
        bd.slabs, bd.tickets = p(slabs), p(tickets)
        lib.call(lib.OP_WGRAD_BATCH, bd, stream())
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        runs.append(outs)
    for i, c in enumerate(r4):
        assert torch.equal(runs[0][i], runs[1][i]) and torch.equal(runs[0][i], runs[2][i])
        ref = -0.5 + 0.5 * (c[5][:, :4].t() @ c[4].float() if c[3] else c[4].float().t() @ c[5][:, :4])
        report(f"wgrad batch fixed order problem {i}", runs[0][i], ref, 1e-4)


def test_batched_wgrad_transposes_and_gather(dev):
    """slh_batch_desc: n weight-gradient reductions / head transposes of different shapes in ONE launch give what the n
    single launches give (the table and the prefix sums of workgroups live in device memory; a workgroup finds its problem
    by bisection), and slh_gather16 builds the k-major, block-diagonal copy of the up matrices."""
    torch.manual_seed(15)
    scale = torch.tensor([0.5], device=dev)
    probs, keep = [], []
    for M, C, R, rmajor in ((1500, 320, 4, 0), (700, 1280, 4, 1), (64, 64, 4, 0), (2048, 640, 4, 1), (300, 2560, 4, 0)):
        z = bf(torch.randn(M, C, device=dev))
        v = torch.randn(M, 8, device=dev)
        out_b = torch.randn(R, C, device=dev) if rmajor else torch.randn(C, R, device=dev)      # += semantics: start non-zero
        out_s = out_b.clone()
        mk = lambda o: lib.WgradDesc(z0=p(z), v=p(v), out=p(o), scale=p(scale), ldz0=C, c0=C, mode=0, stride=1, M=M, R=R, ldv=8,
                                     ldo=C if rmajor else R, out_rmajor=rmajor, vgroup_cols=0)
        probs.append((mk(out_b), mk(out_s), out_b, out_s))
        keep += [z, v]
    # one conv-mode problem in the same batch
    B, H, W, Ci = 2, 10, 12, 64
    img = bf(torch.randn(B * H * W, Ci, device=dev))
    U = torch.randn(B * H * W, 4, device=dev)
    ob, os_ = torch.zeros(4, 9 * Ci, device=dev), torch.zeros(4, 9 * Ci, device=dev)
    mk = lambda o: lib.WgradDesc(z0=p(img), v=p(U), out=p(o), scale=p(scale), ldz0=Ci, c0=Ci, mode=1, batch=B, hs=H, ws=W, stride=1,
                                 ho=H, wo=W, M=B * H * W, R=4, ldv=4, ldo=9 * Ci, out_rmajor=1, vgroup_cols=0)
    probs.append((mk(ob), mk(os_), ob, os_))
    bd, tabs = lib.batch_table(lib.OP_WGRAD_BATCH, [q[0] for q in probs], dev, arg=4)
    lib.call(lib.OP_WGRAD_BATCH, bd, stream())
    for _, ds, _, _ in probs:
        lib.call(lib.OP_WGRAD, ds, stream())
    torch.cuda.synchronize()
    for i, (_, _, o_b, o_s) in enumerate(probs):
        report(f"wgrad batch problem {i}", o_b, o_s, 1e-5)
    # R = 12 batch
    z = bf(torch.randn(900, 640, device=dev)); v = torch.randn(900, 12, device=dev)
    o1, o2 = torch.zeros(12, 640, device=dev), torch.zeros(12, 640, device=dev)
    mk = lambda o: lib.WgradDesc(z0=p(z), v=p(v), out=p(o), scale=p(scale), ldz0=640, c0=640, mode=0, stride=1, M=900, R=12, ldv=12,
                                 ldo=640, out_rmajor=1, vgroup_cols=0)
    bd, tabs2 = lib.batch_table(lib.OP_WGRAD_BATCH, [mk(o1)], dev, arg=12)
    lib.call(lib.OP_WGRAD_BATCH, bd, stream())
    lib.call(lib.OP_WGRAD, mk(o2), stream())
    torch.cuda.synchronize()
    report("wgrad batch R12", o1, o2, 1e-5)
    report("wgrad batch R12 vs torch", o1, 0.5 * v.t() @ z.float(), 1e-4)
    # transposes: different head dims / token counts in one launch, bit-equal to the single launches
    tp = []
    for Bb, Hh, T, D in ((2, 10, 4096, 64), (1, 8, 77, 40), (2, 5, 1024, 160), (1, 20, 300, 64)):
        src = bf(torch.randn(Bb * T, Hh * D, device=dev))
        Dp, ldt = (D + 63) // 64 * 64, (T + 63) // 64 * 64
        d1 = torch.full((Bb, Hh, Dp, ldt), 7.0, device=dev, dtype=torch.bfloat16)
        d2 = torch.full((Bb, Hh, Dp, ldt), 7.0, device=dev, dtype=torch.bfloat16)
        mk = lambda dst: lib.TransposeDesc(src=p(src), dst=p(dst), B=Bb, H=Hh, T=T, ld=Hh * D, ldt=ldt, D=D)
        tp.append((mk(d1), mk(d2), d1, d2, src))
    bd, tabs3 = lib.batch_table(lib.OP_TRANSPOSE_BATCH, [q[0] for q in tp], dev)
    lib.call(lib.OP_TRANSPOSE_BATCH, bd, stream())
    for _, ds, _, _, _ in tp:
        lib.call(lib.OP_TRANSPOSE_HEADS, ds, stream())
    torch.cuda.synchronize()
    for i, (_, _, d1, d2, _) in enumerate(tp):
        assert torch.equal(d1, d2), f"transpose batch problem {i}"
    # a bad descriptor is refused when the table is built
    with pytest.raises(lib.SlidersHipError):
        lib.batch_table(lib.OP_TRANSPOSE_BATCH, [lib.TransposeDesc(src=p(src), dst=p(d1), B=1, H=1, T=100, ld=64, ldt=60, D=64)], dev)
    # gather16
    src = bf(torch.randn(1000, device=dev))
    idx = torch.randint(-1, 1000, (4096,), device=dev, dtype=torch.int32)
    out = torch.zeros(4096, device=dev, dtype=torch.bfloat16)
    lib.call(lib.OP_GATHER16, lib.Gather16Desc(src=p(src), idx=p(idx), out=p(out), n=4096), stream())
    torch.cuda.synchronize()
    ref = torch.where(idx >= 0, src[idx.clamp_min(0).long()], torch.zeros((), device=dev, dtype=torch.bfloat16))
    assert torch.equal(out, ref)


@pytest.mark.parametrize("tile", [0x4412, 0x422, 0x22, 0x11, 0x4322, 0x8015, 0x8014, 0x8013])
def test_gemm_fused_lora_backward_data_form(dev, tile):
    """Backward-data product of an adapted Linear with the adapter fused in (slh_gemm_desc.lora_down + lora_up_rmajor):
    dX = dY . W + s * (dY . B) . A, the rank-4..12 intermediate U = dY . B computed from the k-major copy of the up matrices
    (block-diagonal for a fused q|k|v group) as the third operand tile and written out for the down-gradient."""
    torch.manual_seed(32)
    for groups, M, Nf, Kf in ((1, 300, 320, 640), (3, 512, 3 * 128, 1280)):       # forward: y[M][Nf] = x[M][Kf] W^T
        dy = bf(torch.randn(M, Nf, device=dev))
        wT = bf(torch.randn(Kf, Nf, device=dev) / math.sqrt(Nf))                  # dgrad weights: [Kf][Nf]
        ups = [bf(torch.randn(Nf // groups, 4, device=dev)) for _ in range(groups)]
        A = bf(torch.randn(4 * groups, Kf, device=dev) / math.sqrt(Kf))           # down matrices as stored [rank][Kf]
        scale = torch.tensor([0.25], device=dev)
        up_t = torch.zeros(4 * groups, Nf, device=dev, dtype=torch.bfloat16)
        Ng = Nf // groups
        for g in range(groups):
            up_t[4 * g:4 * g + 4, g * Ng:(g + 1) * Ng] = ups[g].t()
        U_ref = torch.cat([dy.float()[:, g * Ng:(g + 1) * Ng] @ ups[g].float() for g in range(groups)], 1)
        ref = dy.float() @ wT.float().t() + 0.25 * U_ref @ A.float()
        dx = torch.zeros(M, Kf, device=dev, dtype=torch.bfloat16)
        U = torch.full((M, 4 * groups), float("nan"), device=dev)
        d = lib.GemmDesc(a0=p(dy), w=p(wT), c=p(dx), lora_down=p(up_t), lora_up=p(A), lora_scale=p(scale), lora_t_out=p(U),
                         lda0=Nf, ca0=Nf, mode=0, stride=1, ldw=Nf, M=M, N=Kf, K=Nf, ldc=Kf, rows_per_sample=M, ld_t=4 * groups,
                         lora_groups=1, lora_rank=4 * groups, lora_up_rmajor=1, tile=tile)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"fused dgrad+lora groups{groups} tile{tile:x}", dx, ref, TOL)
        report(f"fused dgrad U groups{groups} tile{tile:x}", U, U_ref, 1e-5)
        if tile in (0x4412, 0x22, 0x4322):
            # K cut into slices: U's partials are reduced by the slice that arrives last, like the product's
            S = 2
            dx.zero_()
            U.fill_(float("nan"))
            slabs = torch.full((S, (M + 255) // 256 * 256, (Kf + 127) // 128 * 128), float("nan"), device=dev)
            tslabs = torch.full(((Kf + 63) // 64 * 2 * S, M, 4 * groups), float("nan"), device=dev)
            tickets = torch.zeros(((M + 63) // 64) * ((Kf + 63) // 64), device=dev, dtype=torch.int64)
            d.tile = tile | (S << 16)
            d.splitk_c32, d.splitk_t32, d.splitk_slabs, d.splitk_ticket = p(slabs), p(tslabs), S, p(tickets)
            lib.call(lib.OP_GEMM, d, stream())
            torch.cuda.synchronize()
            report(f"fused dgrad+lora splitk groups{groups} tile{tile:x}", dx, ref, TOL)
            report(f"fused dgrad U splitk groups{groups} tile{tile:x}", U, U_ref, 1e-5)
            assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("tile", [0x22, 0x21, 0x12, 0x11, 0x311, 0x4022, 0x4322, 0x4012, 0x4011, 0x422, 0x421, 0x4412, 0x4411, 0x8015, 0x8014, 0x8013])
def test_gemm_fused_lora_down(dev, tile):
    """LoRAModule.forward fused into one launch: y = x W^T + b + s (x A^T) B^T, T = x A^T written for backward."""
    torch.manual_seed(31)
    for groups, M, N, K in ((1, 300, 320, 640), (3, 512, 3 * 128, 1280)):
        x = bf(torch.randn(M, K, device=dev))
        w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
        b = bf(torch.randn(N, device=dev))
        A = bf(torch.randn(4 * groups, K, device=dev) / math.sqrt(K))
        up = bf(torch.randn(N, 4, device=dev))
        scale = torch.tensor([0.25], device=dev)
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        Tout = torch.zeros(M, 4 * groups, device=dev)
        d = lib.GemmDesc(a0=p(x), w=p(w), bias=p(b), c=p(c), lora_down=p(A), lora_up=p(up), lora_scale=p(scale),
                         lora_t_out=p(Tout), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K, ldc=N,
                         rows_per_sample=M, ld_t=4 * groups, lora_groups=groups, lora_rank=4 * groups, tile=tile)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        T = x.float() @ A.float().t()
        ref = x.float() @ w.float().t() + b.float()
        ng = N // groups
        for g in range(groups):
            ref[:, g * ng:(g + 1) * ng] += 0.25 * T[:, 4 * g:4 * g + 4] @ up.float()[g * ng:(g + 1) * ng].t()
        report(f"gemm_fused_lora g{groups} tile{tile:x}", c, ref, TOL)
        report(f"gemm_fused_lora T g{groups} tile{tile:x}", Tout, T, 1e-5)
    # conv form (3x3, stride 2) with the down matrix in [r][tap][Cin] order
    B, H, W, Ci, Co = 2, 16, 12, 64, 128
    img = bf(torch.randn(B, Ci, H, W, device=dev))
    w4 = bf(torch.randn(Co, Ci, 3, 3, device=dev) / math.sqrt(9 * Ci))
    a4 = bf(torch.randn(4, Ci, 3, 3, device=dev) / math.sqrt(9 * Ci))
    up = bf(torch.randn(Co, 4, device=dev))
    scale = torch.tensor([0.25], device=dev)
    ref_img = _conv_ref(img.float(), w4.float(), 2)
    t_img = _conv_ref(img.float(), a4.float(), 2)
    Ho, Wo = ref_img.shape[2:]
    ref = _to_pix(ref_img) + 0.25 * _to_pix(t_img) @ up.float().t()
    M = B * Ho * Wo
    c = torch.zeros(M, Co, device=dev, dtype=torch.bfloat16)
    Tout = torch.zeros(M, 4, device=dev)
    d = lib.GemmDesc(a0=p(bf(_to_pix(img.float()))), w=p(_pack_conv(w4)), c=p(c), lora_down=p(_pack_conv(a4)), lora_up=p(up),
                     lora_scale=p(scale), lora_t_out=p(Tout), lda0=Ci, ca0=Ci, mode=1, batch=B, hs=H, ws=W, stride=2, ho=Ho,
                     wo=Wo, ldw=9 * Ci, M=M, N=Co, K=9 * Ci, ldc=Co, rows_per_sample=Ho * Wo, ld_t=4, lora_groups=1,
                     lora_rank=4, tile=tile)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"conv_fused_lora tile{tile:x}", c, ref, TOL)
    report(f"conv_fused_lora T tile{tile:x}", Tout, _to_pix(t_img), 1e-5)


@pytest.mark.parametrize("local", ["1", "0"])
def test_gemm_splitk_stress_same_slabs(dev, local):
    """Back-to-back split-K launches that recycle ONE slab workspace and ONE ticket array (as consecutive products of a pass
    do), different operands every launch, replayed as a hipGraph: every launch must reduce ITS slices' partials - a slab line
    left in an L1 / L2 by an earlier launch, or a slice's store that the last arriver reads too early, would show as a wrong
    tile.  Both read paths of the last arriver: the same-XCD shortcut (default on gfx950) and the agent-scope loads
    (SLIDERS_SPLITK_LOCAL=0), each in a fresh process state of the library's cached switch (subprocess)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import math, torch
        from sliders_amd import lib
        dev = torch.device("cuda:0")
        torch.manual_seed(5)
        M, N, K, S, L = 512, 640, 2560, 4, 72
        xs = [(torch.randn(M, K, device=dev)).bfloat16() for _ in range(3)]
        ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(3)]
        outs = [torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(L)]
        slabs = torch.full((S, 512, 640), float("nan"), device=dev)
        tickets = torch.zeros(((M + 63) // 64) * ((N + 63) // 64), device=dev, dtype=torch.int64)
        prog = lib.Program()
        for i in range(L):
            tile = (0x40412, 0x44412, 0x28015, 0x20022)[i % 4]
            d = lib.GemmDesc(a0=xs[i % 3].data_ptr(), w=ws[(i // 3) % 3].data_ptr(), c=outs[i].data_ptr(), lda0=K, ca0=K, mode=0,
                             stride=1, ldw=K, M=M, N=N, K=K, ldc=N, rows_per_sample=M, tile=tile, splitk_c32=slabs.data_ptr(),
                             splitk_slabs=S, splitk_ticket=tickets.data_ptr())
            prog.add(lib.OP_GEMM, d, f"g{i}")
        s = torch.cuda.current_stream().cuda_stream
        for rep in range(6):                     # the later replays run as one captured graph
            for o in outs:
                o.fill_(7.0)
            prog.run(s)
            torch.cuda.synchronize()
            for i in range(L):
                ref = xs[i % 3].float() @ ws[(i // 3) % 3].float().t()
                err = ((outs[i].float() - ref).norm() / ref.norm()).item()
                assert err < 4e-3, (rep, i, err)
            assert int(tickets.abs().sum()) == 0
        print("stress ok")
    """)
    env = dict(os.environ, SLIDERS_SPLITK_LOCAL=local, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("tile", [0x4412, 0x4012, 0x12])      # (only 0x4412 takes the hint; the others must ignore it)
def test_gemm_weight_touch_rides_on_idle_workgroups(dev, tile):
    """slh_gemm_desc.pf_*: a launch that leaves CUs idle streams the weights of a LATER launch through its last workgroups.  A hint:
    the product is bit-identical with and without it, whatever the byte range (odd sizes, a range the grid ignores because it
    fills the chip)."""
    torch.manual_seed(tile)
    M, N, K = 1024, 640, 1280            # 40 tiles of 128 x 128: plenty of idle CUs
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = bf(torch.randn(N, device=dev))
    later = torch.randn(3 * 1024 * 1024 + 5, device=dev)            # 12 MB + 20 bytes of "weights of a later launch"
    guard = later.clone()
    outs = []
    for pf_bytes in (0, later.numel() * 4, 4096 + 16, 48):
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        d = lib.GemmDesc(a0=p(x), w=p(w), bias=p(bias), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K, ldc=N,
                         rows_per_sample=M, tile=tile, pf_ptr=p(later) if pf_bytes else 0, pf_bytes=pf_bytes)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        outs.append(c)
    report(f"weight touch tile{tile:x}", outs[0], x.float() @ w.float().t() + bias.float(), TOL)
    assert all(torch.equal(outs[0], o) for o in outs[1:]) and torch.equal(later, guard)
    # a grid that fills the chip ignores the hint
    M2 = 8192
    x2 = bf(torch.randn(M2, K, device=dev))
    c2 = torch.zeros(M2, N, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x2), w=p(w), c=p(c2), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M2, N=N, K=K, ldc=N, rows_per_sample=M2, tile=tile,
                     pf_ptr=p(later), pf_bytes=later.numel() * 4)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"weight touch (full grid) tile{tile:x}", c2, x2.float() @ w.float().t(), TOL)
    if tile == 0x4412:
        d.pf_ptr = p(later) + 4
        d.M, d.a0, d.c = M, p(x), p(outs[0])
        with pytest.raises(lib.SlidersHipError, match="pf_ptr"):
            lib.call(lib.OP_GEMM, d, stream())


def test_gemm_rejects_reserved_tile_bits(dev):
    """Tile bits 20 and up are reserved: round 4's stream-K form (0x104412: finishing workgroups waited on flags of other
    workgroups, which can hang two concurrent launches) is gone from the library - no kernel waits on another workgroup - and
    slh_gemm refuses the code instead of silently running something else."""
    M, N, K = 256, 128, 128
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev))
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(w), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K, M=M, N=N, K=K, ldc=N, rows_per_sample=M,
                     tile=0x104412)
    with pytest.raises(lib.SlidersHipError, match="reserved"):
        lib.call(lib.OP_GEMM, d, stream())
    assert int(c.abs().sum()) == 0


@pytest.mark.parametrize("tile", [0x20412, 0x40421, 0x80422, 0x44412, 0x30011, 0x20022, 0xf0412, 0x28015, 0x38014])
def test_gemm_splitk(dev, tile):
    """Split-K (slh_gemm_desc.tile bits 16-19): every K slice publishes its partial tile in its own fp32 slab (whatever the
    workspace held), the slice of a tile that arrives last adds the slabs in slice order and runs the ordinary epilogue -
    one launch, bit-reproducible, tickets left zero.  Dense + bias + residual, 3x3 convolution, fused LoRA down/up (T reduced
    too, and written out for the backward), two-source K."""
    torch.manual_seed(tile & 0xff)
    S = (tile >> 16) & 15
    M, N, K = 300, 320, 1280
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = bf(torch.randn(N, device=dev))
    res = bf(torch.randn(M, N, device=dev))
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.full((S, 512, 384), float("nan"), device=dev)      # slabs of roundup(M, 256) x roundup(N, 128)
    tickets = torch.zeros(((M + 63) // 64) * ((N + 63) // 64), device=dev, dtype=torch.int64)
    d = lib.GemmDesc(a0=p(x), w=p(w), bias=p(bias), residual=p(res), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=K,
                     M=M, N=N, K=K, ld_res=N, ldc=N, rows_per_sample=M, tile=tile, splitk_c32=p(ws), splitk_slabs=S,
                     splitk_ticket=p(tickets))
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"splitk dense tile{tile:x}", c, x.float() @ w.float().t() + bias.float() + res.float(), TOL)
    assert int(tickets.abs().sum()) == 0, "the last slice of every tile re-arms its ticket"
    c0 = c.clone()
    for _ in range(3):
        c.zero_()
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        assert torch.equal(c, c0), "split-K must be bit-reproducible"
    d.splitk_ticket = 0
    with pytest.raises(lib.SlidersHipError, match="ticket"):
        lib.call(lib.OP_GEMM, d, stream())
    d.splitk_ticket = p(tickets)
    d.splitk_slabs = 1
    with pytest.raises(lib.SlidersHipError, match="slabs"):
        lib.call(lib.OP_GEMM, d, stream())
    # fused adapter: T is reduced across the slices as well
    A = bf(torch.randn(8, K, device=dev) / math.sqrt(K))
    up = bf(torch.randn(N, 4, device=dev))
    scale = torch.tensor([0.25], device=dev)
    ws.fill_(float("nan"))
    T32 = torch.full(((N + 63) // 64 * 2 * S, M, 8), float("nan"), device=dev)     # per column tile (sized for 64-column tiles)
    Tout = torch.full((M, 8), float("nan"), device=dev)
    d = lib.GemmDesc(a0=p(x), w=p(w), bias=p(bias), c=p(c), lora_down=p(A), lora_up=p(up), lora_scale=p(scale), lda0=K, ca0=K,
                     mode=0, stride=1, ldw=K, M=M, N=N, K=K, ldc=N, rows_per_sample=M, ld_t=8, lora_groups=2, lora_rank=8,
                     tile=tile, splitk_c32=p(ws), splitk_t32=p(T32), splitk_slabs=S, lora_t_out=p(Tout), splitk_ticket=p(tickets))
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    T = x.float() @ A.float().t()
    ref = x.float() @ w.float().t() + bias.float()
    for g in range(2):
        ref[:, g * 160:(g + 1) * 160] += 0.25 * T[:, 4 * g:4 * g + 4] @ up.float()[g * 160:(g + 1) * 160].t()
    report(f"splitk fused lora tile{tile:x}", c, ref, TOL)
    report(f"splitk fused lora T tile{tile:x}", Tout, T, 1e-5)
    # 3x3 convolution, two sources, stride 1
    B, H, W, C0, C1, Co = 2, 8, 8, 128, 64, 128
    i0, i1 = bf(torch.randn(B, C0, H, W, device=dev)), bf(torch.randn(B, C1, H, W, device=dev))
    w4 = bf(torch.randn(Co, C0 + C1, 3, 3, device=dev) / math.sqrt(9 * (C0 + C1)))
    ref = _to_pix(_conv_ref(torch.cat([i0, i1], 1).float(), w4.float()))
    x0, x1 = bf(_to_pix(i0.float())), bf(_to_pix(i1.float()))
    Mc = B * H * W
    cc = torch.zeros(Mc, Co, device=dev, dtype=torch.bfloat16)
    wsc = torch.full((S, 256, 128), float("nan"), device=dev)
    d = lib.GemmDesc(a0=p(x0), a1=p(x1), w=p(_pack_conv(w4)), c=p(cc), lda0=C0, lda1=C1, ca0=C0, ca1=C1, mode=1, batch=B, hs=H,
                     ws=W, stride=1, ho=H, wo=W, ldw=9 * (C0 + C1), M=Mc, N=Co, K=9 * (C0 + C1), ldc=Co, rows_per_sample=H * W,
                     tile=tile, splitk_c32=p(wsc), splitk_slabs=S, splitk_ticket=p(tickets))
    if (tile >> 12) & 15 == 8 and 128 * 64 * (tile & 15) > 256 * 128:
        # a 128 x 320 tile over M = N = 128 would need slabs beyond the workspace contract (roundup(M, 256) x roundup(N, 128) floats
        # per slice): refused, not overrun
        with pytest.raises(lib.SlidersHipError, match="workspace contract"):
            lib.call(lib.OP_GEMM, d, stream())
        return
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"splitk conv 2src tile{tile:x}", cc, ref, TOL)
    assert int(tickets.abs().sum()) == 0
