"""The tiles of csrc/gemm7.hip (tile codes 0x7648 = 128 x 256, 0x7645 = 128 x 160 on four compute waves; 0x748a = 256 x 320 on eight) through the C ABI
against fp32 restatements of the reference's op sequence (F.linear / LayerNorm -> Linear / lora.py:108-112 / GEGLU inside diffusers'
Attention and FeedForward, trainscripts/textsliders/train_util.py:242-247) on the same bf16-rounded inputs.  Tolerance: relative L2
< 6e-3 (bf16 output rounding + accumulation order), as tests/test_kernels_gpu.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from sliders_amd import lib
from sliders_amd.weights import _geglu_perm16, fold_layernorm, pack_gemm_w
from tests.util import bf, p, report, stream

pytestmark = pytest.mark.gpu
TOL = 6e-3
TILES = {0x7648: (128, 256, 6), 0x7645: (128, 160, 6), 0x748A: (256, 320, 4)}      # 0x748a: eight compute waves


def _chunks(x, cw):
    """producer-side chunk statistics [K / cw][M][2] = (mean, M2) of every cw-column chunk of the rows of x (fp32)"""
    M, K = x.shape
    xc = x.float().view(M, K // cw, cw).double()
    mean = xc.mean(-1)
    return torch.stack([mean, ((xc - mean[..., None]) ** 2).sum(-1)], -1).permute(1, 0, 2).contiguous().float()


@pytest.mark.parametrize("tile", sorted(TILES))
@pytest.mark.parametrize("mt,nt,K", [(1, 1, 128), (1, 1, 192), (1, 1, 256), (2, 1, 320), (1, 2, 384), (3, 3, 448), (2, 2, 1280), (16, 5, 640), (9, 7, 2560)])
def test_gemm7_dense(dev, tile, mt, nt, K):
    """plain / bias / bias + residual, every tail class of the half-tile ring (K / 32 = S, S + 1, ... and many), strided operands;
    bit-equal between runs"""
    bm, bn, S = TILES[tile]
    if K // 32 < S:
        pytest.skip("K below the ring depth of this tile")
    M, N = mt * bm, nt * bn
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
        assert lib.gemm7_ok(d)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        ref = x.float() @ w.float().t() + (bias.float() if use_bias else 0) + (res.float() if use_res else 0)
        report(f"gemm7 {tile:x} M{M} N{N} K{K} bias{int(use_bias)} res{int(use_res)}", c, ref, TOL)
        c2 = torch.zeros_like(c)
        d.c = p(c2)
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        assert torch.equal(c, c2)
    xa = bf(torch.randn(M, K + 64, device=dev))
    cw = torch.zeros(M, N + 8, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(xa), w=p(wp), bias=p(bias), c=p(cw), lda0=K + 64, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K, ldc=N + 8,
                     rows_per_sample=M, tile=tile, w_layout=1)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm7 {tile:x} strided M{M} N{N} K{K}", cw[:, :N], xa[:, :K].float() @ w.float().t() + bias.float(), TOL)
    assert float(cw[:, N:].float().abs().max()) == 0.0


@pytest.mark.parametrize("tile", [0x7648, 0x7645])      # (the 256-row tile has no producer side)
@pytest.mark.parametrize("offset", [0.0, 8.0])
def test_gemm7_layernorm_both_sides(dev, tile, offset):
    """producer side (ln_out: 64-column chunks, 80 where the wave's columns are a multiple of 80) feeding the consumer side (ln_in) of
    the same tile and of the ring tile: Linear(LayerNorm(h)) of the reference's op sequence; (mean, rstd) left for the backward"""
    bm, bn, S = TILES[tile]
    cw = 80 if (bn // 32) % 5 == 0 else 64
    C = {256: 1280, 160: 640}[bn]          # N = K = C: a multiple of the tile width and of both chunk widths
    M = 2 * bm
    torch.manual_seed(C + tile)
    o = bf(torch.randn(M, C, device=dev))
    wo = bf(torch.randn(C, C, device=dev) / math.sqrt(C))
    bo = bf(torch.randn(C, device=dev) + offset)
    res = bf(torch.randn(M, C, device=dev) * 2)
    gamma, beta = bf(torch.randn(C, device=dev) * 0.5 + 1.0), bf(torch.randn(C, device=dev) * 0.3)
    h = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
    chunks = torch.full((C // cw, M, 2), float("nan"), device=dev)
    wop = pack_gemm_w(wo)
    d = lib.GemmDesc(a0=p(o), w=p(wop), bias=p(bo), residual=p(res), c=p(h), lda0=C, ca0=C, mode=0, stride=1, ldw=0, M=M, N=C,
                     K=C, ld_res=C, ldc=C, rows_per_sample=M, tile=tile, w_layout=1, ln_out=p(chunks))
    assert lib.gemm7_ok(d)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm7 {tile:x} ln producer", h, (o.float() @ wo.float().t() + bo.float() + res.float()), TOL)
    want = _chunks(h, cw).double()
    got = chunks.double()
    assert float((got[..., 0] - want[..., 0]).abs().max()) < 1e-5 * max(1.0, float(want[..., 0].abs().max()))
    assert float(((got[..., 1] - want[..., 1]) / want[..., 1]).abs().max()) < 1e-4
    ln = bf(F.layer_norm(h.float(), (C,), gamma.float(), beta.float(), 1e-5))
    N = 2 * C
    w = bf(torch.randn(N, C, device=dev) / math.sqrt(C))
    wf, sv, bp = fold_layernorm(w, None, gamma, beta)
    wfp = pack_gemm_w(wf)
    for t2 in (tile, 0x4412):
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        mr = torch.full((M, 2), float("nan"), device=dev)
        d = lib.GemmDesc(a0=p(h), w=p(wfp), c=p(c), lda0=C, ca0=C, mode=0, stride=1, ldw=0, M=M, N=N, K=C, ldc=N, rows_per_sample=M,
                         tile=t2, w_layout=1, ln_in=p(chunks), ln_in_chunks=C // cw, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5, ln_mr_out=p(mr))
        lib.call(lib.OP_GEMM, d, stream())
        torch.cuda.synchronize()
        report(f"gemm7 {tile:x} ln consumer on tile {t2:x} off{offset}", c, ln.float() @ w.float().t(), TOL)
        hd = h.double()
        assert float((mr[:, 0].double() - hd.mean(-1)).abs().max()) < 1e-5 * max(1.0, float(hd.mean(-1).abs().max()))
        assert float((mr[:, 1].double() * torch.sqrt(hd.var(-1, unbiased=False) + 1e-5) - 1).abs().max()) < 1e-4


@pytest.mark.parametrize("groups", [1, 3])
@pytest.mark.parametrize("K", [256, 1280])
def test_gemm7_fused_adapter_and_vt(dev, groups, K):
    """lora.py:108-112 inside the 128 x 256 tile: the down matrices of 1 / 3 column groups as 16 extra rows of the W tile, T for half of
    a wave's row blocks per wave, the up-projection as one MFMA per accumulator block; T written out (lora_t_out); the V third of a
    fused q|k|v projection head-transposed (vt_out), optionally row-major as well (vt_also_c); against fp32 and the ping-pong tile"""
    B, T, heads, D = 2, 256, 4, 64
    C = heads * D
    M, N = B * T, (3 * C if groups == 3 else 2 * C)
    R = 4 * groups
    torch.manual_seed(groups + K)
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = bf(torch.randn(N, device=dev))
    A = bf(torch.randn(R, K, device=dev) / math.sqrt(K))
    up = bf(torch.randn(N, 4, device=dev))
    scale = torch.tensor([0.75], device=dev)
    wp = pack_gemm_w(w)
    t32 = x.float() @ A.float().t()
    ref = x.float() @ w.float().t() + bias.float()
    cg = N // groups
    tb = bf(0.75 * t32).float()
    for g in range(groups):
        ref[:, g * cg:(g + 1) * cg] += tb[:, 4 * g:4 * g + 4] @ up.float()[g * cg:(g + 1) * cg].t()
    outs = {}
    for tile in (0x7648, 0x8014):
        for also_c in ((0, 1) if groups == 3 else (None,)):
            c = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
            Tt = torch.full((M, R), float("nan"), device=dev)
            vt = torch.full((B, heads, D, T), 7.0, device=dev, dtype=torch.bfloat16)
            d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bias), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K, ldc=N,
                             rows_per_sample=T, tile=tile, w_layout=1, lora_down=p(A), lora_up=p(up), lora_scale=p(scale), ld_t=R,
                             lora_groups=groups, lora_rank=R, lora_t_out=p(Tt))
            if also_c is not None:
                d.vt_out, d.vt_col0, d.vt_D, d.vt_heads, d.vt_tokens, d.vt_ld, d.vt_also_c = p(vt), 2 * C, D, heads, T, T, also_c
            if tile == 0x7648:
                assert lib.gemm7_ok(d)
            lib.call(lib.OP_GEMM, d, stream())
            torch.cuda.synchronize()
            name = f"gemm7 fused adapter groups{groups} K{K} tile{tile:x} also_c{also_c}"
            if also_c is None:
                report(name, c, ref, TOL)
            else:
                report(name + " q|k", c[:, :2 * C], ref[:, :2 * C], TOL)
                report(name + " v^T", vt, ref[:, 2 * C:].reshape(B, T, heads, D).permute(0, 2, 3, 1), TOL)
                if also_c:
                    report(name + " v", c[:, 2 * C:], ref[:, 2 * C:], TOL)
                else:
                    assert float((c[:, 2 * C:].float() - 7.0).abs().max()) == 0.0
            assert float((Tt - t32).abs().max()) < 2e-3 * max(1.0, float(t32.abs().max()))
            outs[(tile, also_c)] = (c.clone(), vt.clone())
    for also_c in ((0, 1) if groups == 3 else (None,)):
        a, b = outs[(0x7648, also_c)], outs[(0x8014, also_c)]
        cols = slice(0, 2 * C) if also_c == 0 else slice(0, N)
        assert float((a[0][:, cols].float() - b[0][:, cols].float()).abs().max()) <= 2.0 ** -6 * float(b[0][:, cols].float().abs().max())


@pytest.mark.parametrize("offset", [0.0, 20.0])
def test_gemm7_layernorm_folded_with_fused_adapter(dev, offset):
    """norm1 folded into the adapter-carrying q|k|v projection (ln_in + lora_down + ln_lora_s / ln_lora_c) on the 128 x 256 tile, as the
    no-grad passes run it: main product and the adapter's down-projection from the raw rows, normalised in the epilogue ahead of the
    up-projection; V third head-transposed.  Against the reference's op sequence (LayerNorm -> bf16 -> Linear + LoRA, T rounded to bf16)."""
    torch.manual_seed(int(offset) + 7)
    B, T, heads, D = 2, 256, 4, 64
    C = heads * D
    M, N, K = B * T, 3 * C, 320
    x = bf(torch.randn(M, K, device=dev) * 1.5 + offset)
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    gamma, beta = bf(torch.randn(K, device=dev) * 0.5 + 1.0), bf(torch.randn(K, device=dev) * 0.3)
    A = bf(torch.randn(12, K, device=dev) / math.sqrt(K))
    up = bf(torch.randn(N, 4, device=dev))
    scale = torch.tensor([0.5], device=dev)
    chunks = _chunks(x, 64)
    wf, sv, bp = fold_layernorm(w, None, gamma, beta)
    wfp = pack_gemm_w(wf)
    A2 = torch.zeros_like(A)
    sc = torch.full((2, 16), float("nan"), device=dev)
    items = torch.tensor([[p(A), p(gamma), p(beta), p(A2), sc.data_ptr(), sc.data_ptr() + 64, 12 | (K << 32)]], dtype=torch.int64, device=dev)
    lib.call(lib.OP_LORA_LN_FOLD, lib.LoraLnFoldDesc(items=items.data_ptr(), n=1), stream())
    torch.cuda.synchronize()
    c = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
    vt = torch.full((B, heads, D, T), 7.0, device=dev, dtype=torch.bfloat16)
    d = lib.GemmDesc(a0=p(x), w=p(wfp), c=p(c), lora_down=p(A2), lora_up=p(up), lora_scale=p(scale), lda0=K, ca0=K, mode=0, stride=1,
                     ldw=0, w_layout=1, M=M, N=N, K=K, ldc=N, rows_per_sample=T, ld_t=12, lora_groups=3, lora_rank=12, tile=0x7648,
                     ln_in=p(chunks), ln_in_chunks=K // 64, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5,
                     ln_lora_s=sc.data_ptr(), ln_lora_c=sc.data_ptr() + 64,
                     vt_out=p(vt), vt_col0=2 * C, vt_D=D, vt_heads=heads, vt_tokens=T, vt_ld=T)
    assert lib.gemm7_ok(d)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    ln = bf(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)).float()
    Tt = bf(ln @ A.float().t() * 0.5).float()
    ref = ln @ w.float().t()
    for g in range(3):
        ref[:, g * C:(g + 1) * C] += Tt[:, 4 * g:4 * g + 4] @ up.float()[g * C:(g + 1) * C].t()
    report(f"gemm7 ln + fused adapter off{offset} q|k", c[:, :2 * C], ref[:, :2 * C], TOL)
    report(f"gemm7 ln + fused adapter off{offset} v^T", vt, ref[:, 2 * C:].reshape(B, T, heads, D).permute(0, 2, 3, 1), TOL)
    assert ((ref - ln @ w.float().t()).norm() / ref.norm()).item() > 0.05


@pytest.mark.parametrize("tile", [0x7648, 0x748A])
def test_gemm7_geglu_16_blocks(dev, tile):
    """geglu = 3 (weight rows in 32-row blocks [16 value | 16 gate]): a value block and its gate block are neighbouring 16-row blocks of
    one wave; with bias, and with the LayerNorm fold on the consumer side as the no-grad passes run ff.net.0.proj"""
    bm, bn, S = TILES[tile]
    torch.manual_seed(23 + tile)
    M, K, N = 2 * bm, 320, 3 * bn
    n_out = N // 2
    x = bf(torch.randn(M, K, device=dev))
    w = bf(torch.randn(N, K, device=dev) / math.sqrt(K))
    b = bf(torch.randn(N, device=dev))
    proj = bf(x.float() @ w.float().t() + b.float()).float()
    ref = proj[:, :n_out] * bf(F.gelu(proj[:, n_out:])).float()
    c = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
    wp, bpm = pack_gemm_w(_geglu_perm16(w)), _geglu_perm16(b).contiguous()
    d = lib.GemmDesc(a0=p(x), w=p(wp), bias=p(bpm), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=0, w_layout=1, M=M, N=N,
                     K=K, ldc=n_out, geglu=3, rows_per_sample=M, tile=tile)
    assert lib.gemm7_ok(d)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm7 geglu16 tile{tile:x}", c, ref, TOL)
    gamma, beta = bf(torch.randn(K, device=dev) * 0.5 + 1.0), bf(torch.randn(K, device=dev) * 0.3)
    chunks = _chunks(x, 64)
    ln = bf(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5))
    wf, sv, bp = fold_layernorm(w, b, gamma, beta)
    wfp, sv, bp = pack_gemm_w(_geglu_perm16(wf)), _geglu_perm16(sv).contiguous(), _geglu_perm16(bp).contiguous()
    proj = bf(ln.float() @ w.float().t() + b.float()).float()
    refg = proj[:, :n_out] * bf(F.gelu(proj[:, n_out:])).float()
    c.zero_()
    d = lib.GemmDesc(a0=p(x), w=p(wfp), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=0, w_layout=1, M=M, N=N, K=K, ldc=n_out, geglu=3,
                     rows_per_sample=M, tile=tile, ln_in=p(chunks), ln_in_chunks=K // 64, ln_s=p(sv), ln_b=p(bp), ln_eps=1e-5)
    assert lib.gemm7_ok(d)
    lib.call(lib.OP_GEMM, d, stream())
    torch.cuda.synchronize()
    report(f"gemm7 geglu16 + ln fold tile{tile:x}", c, refg, TOL)


def test_gemm7_rejections(dev):
    """what the tiles cannot run is refused before any launch (and slh_gemm7_ok says so to the planner)"""
    M, N, K = 128, 256, 256
    x, w = bf(torch.randn(M, K, device=dev)), bf(torch.randn(N, K, device=dev))
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    wp = pack_gemm_w(w)
    base = dict(a0=p(x), w=p(wp), c=p(c), lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K, ldc=N, rows_per_sample=M,
                tile=0x7648, w_layout=1)
    assert lib.gemm7_ok(lib.GemmDesc(**base))
    for bad in (dict(N=320), dict(M=64), dict(K=128), dict(w_layout=0, ldw=K), dict(geglu=1), dict(rowbias=p(c), ld_rowbias=N),
                dict(tile=0x7448), dict(tile=0x7648 | 0x20000), dict(ln_in=p(c), ln_in_chunks=3, ln_s=p(c), ln_b=p(c))):
        d = lib.GemmDesc(**{**base, **bad})
        assert not lib.gemm7_ok(d)
        with pytest.raises(lib.SlidersHipError):
            lib.call(lib.OP_GEMM, d, stream())
