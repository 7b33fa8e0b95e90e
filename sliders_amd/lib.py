"""ctypes binding of libsliders_hip.so (include/sliders_hip.h).

The product path has NO fallback: if the shared library is missing or a descriptor size disagrees
with the header the import of this module raises, and every op raises RuntimeError with the
library's slh_last_error() text on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLIDERS_HIP_LIB") or os.path.join(_HERE, "libsliders_hip.so")   # override: development builds

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


def _ptrs(*names):
    return [(n, c_vp) for n in names]


def _ints(*names):
    return [(n, c_i32) for n in names]


GemmDesc = _struct("GemmDesc", _ptrs("a0", "a1", "w", "bias", "rowbias", "lora_t", "lora_up", "lora_scale",
                                     "residual", "c", "lora_down", "lora_t_out")
                   + _ints("lda0", "lda1", "ca0", "ca1", "mode", "batch", "hs", "ws", "src_xform", "stride",
                           "ho", "wo", "ldw", "M", "N", "K", "ld_rowbias", "rows_per_sample", "ld_t",
                           "lora_groups", "ld_res", "ldc", "geglu", "tile", "lora_rank", "lora_up_rmajor", "w_layout",
                           "reserved_") + _ptrs("splitk_c32", "splitk_t32", "vt_out")
                   + _ints("vt_col0", "vt_D", "vt_heads", "vt_tokens", "vt_ld", "splitk_slabs")
                   + _ptrs("ln_out", "ln_in", "ln_s", "ln_b") + _ints("ln_in_chunks") + [("ln_eps", c_f32)]
                   + _ptrs("splitk_ticket", "ln_mr_out", "geglu_pre") + _ints("ld_pre", "vt_also_c")
                   + _ptrs("xa_k", "xa_vt") + _ints("xa_tk", "xa_tq", "xa_ldk", "xa_ldvt", "xa_vt_heads") + [("xa_scale", c_f32)]
                   + _ptrs("ln_lora_s", "ln_lora_c", "pf_ptr") + [("pf_bytes", c_i64)])
SkinnyDesc = _struct("SkinnyDesc", _ptrs("a0", "a1", "w", "bias", "out")
                     + _ints("lda0", "lda1", "ca0", "ca1", "mode", "batch", "hs", "ws", "src_xform", "stride",
                             "ho", "wo", "M", "R", "K", "ldo", "out_kind", "w_kmajor"))
GemvDesc = _struct("GemvDesc", _ptrs("x", "w", "bias", "addend", "lora_t", "lora_tcol", "lora_up", "lora_scale", "y")
                   + _ints("nb", "N", "K", "ldx", "ld_add", "ld_t", "ldy", "in_act", "out_f32"))
GnDesc = _struct("GnDesc", _ptrs("x0", "x1", "gamma", "beta", "stats", "y")
                 + _ints("ldx0", "ldx1", "c0", "c1", "batch", "hw", "groups", "ldy") + [("eps", c_f32)]
                 + _ints("act") + _ptrs("partial", "ticket"))
GnBwdDesc = _struct("GnBwdDesc", _ptrs("x0", "x1", "gamma", "beta", "stats", "bstats", "dy", "dx0", "dx1")
                    + _ints("ldx0", "ldx1", "c0", "c1", "batch", "hw", "groups", "lddy", "lddx0", "lddx1")
                    + [("eps", c_f32)] + _ints("act", "accumulate0", "accumulate1") + _ptrs("bpartial", "bticket"))
LnDesc = _struct("LnDesc", _ptrs("x", "gamma", "beta", "y", "mean_rstd") + _ints("M", "C", "ldx", "ldy")
                 + [("eps", c_f32)])
LnBwdDesc = _struct("LnBwdDesc", _ptrs("x", "gamma", "dy", "mean_rstd", "dx")
                    + _ints("M", "C", "ldx", "lddy", "lddx", "accumulate"))
AttnDesc = _struct("AttnDesc", _ptrs("q", "k", "vt", "o", "lse")
                   + _ints("B", "H", "Tq", "Tk", "ldq", "ldk", "ldvt", "ldo") + [("scale", c_f32)]
                   + _ints("D", "vt_batch_heads", "reserved_") + _ptrs("pf_ptr") + [("pf_bytes", c_i64)])
TransposeDesc = _struct("TransposeDesc", _ptrs("src", "dst") + _ints("B", "H", "T", "ld", "ldt", "D"))
AttnBwdDesc = _struct("AttnBwdDesc", _ptrs("q", "k", "v", "o", "d_o", "kt", "qt", "dot", "lse", "delta", "dq", "dk", "dv")
                      + _ints("B", "H", "Tq", "Tk", "ldq", "ldk", "ldv", "ldo", "lddo", "ldkt", "ldqt", "lddq", "lddk",
                              "lddv")
                      + [("scale", c_f32)] + _ints("need_dkv", "D", "pad_"))
TembedDesc = _struct("TembedDesc", _ptrs("vals", "out") + _ints("nb", "n_vals", "dim", "ldo", "col0"))
ConvInDesc = _struct("ConvInDesc", _ptrs("x", "w", "bias", "y") + _ints("batch", "cin", "h", "wd", "cout", "ldy"))
EwDesc = _struct("EwDesc", _ptrs("a", "b", "out") + _ints("M", "C", "lda", "ldb", "ldo", "op", "iarg", "iarg2")
                 + [("alpha", c_f32)] + _ints("pad_"))
CfgDdimDesc = _struct("CfgDdimDesc", _ptrs("eps", "x", "out", "out2", "eps_text") + _ints("nb", "chw")
                      + [("guidance", c_f32), ("c_sqrt_beta_t", c_f32), ("c_inv_sqrt_alpha_t", c_f32),
                         ("c_sqrt_alpha_prev", c_f32), ("c_dir", c_f32)] + _ints("do_step", "v_prediction")
                      + [("c_sqrt_alpha_t", c_f32)] + _ints("pad_"))
LossDesc = _struct("LossDesc", _ptrs("target", "positive", "neutral", "uncond", "loss", "dtarget", "dtarget_pix")
                   + _ints("n") + [("guidance", c_f32)] + _ints("erase", "hw", "nch"))
WgradDesc = _struct("WgradDesc", _ptrs("z0", "z1", "v", "out", "scale")
                    + _ints("ldz0", "ldz1", "c0", "c1", "mode", "batch", "hs", "ws", "src_xform", "stride", "ho",
                            "wo", "M", "R", "ldv", "ldo", "out_rmajor", "vgroup_cols") + _ptrs("slabs", "tickets"))
AdamwDesc = _struct("AdamwDesc", _ptrs("param", "exp_avg", "exp_avg_sq", "grad") + [("n", c_i64)]
                    + [(n, c_f64) for n in ("lr", "beta1", "beta2", "eps", "weight_decay")]
                    + _ints("step") + [("grad_scale", c_f32)] + _ptrs("param_lo") + _ints("f32_state", "pad_"))
LionDesc = _struct("LionDesc", _ptrs("param", "exp_avg", "grad") + [("n", c_i64)]
                   + [(n, c_f64) for n in ("lr", "beta1", "beta2", "weight_decay")] + [("grad_scale", c_f32)] + _ints("pad_"))
MemsetDesc = _struct("MemsetDesc", _ptrs("ptr") + [("nbytes", c_i64)] + _ints("value", "pad"))
LoraCdgradDesc = _struct("LoraCdgradDesc", _ptrs("u", "a_down", "scale", "gx")
                         + _ints("batch", "hl", "wl", "ho", "wo", "stride", "cin", "ldu", "ldgx", "accumulate"))
TembLoraBwdDesc = _struct("TembLoraBwdDesc", _ptrs("g", "t", "up", "emb", "d_up", "d_down", "scale") + _ints("C", "ted"))
# image sliders: fp32 AutoencoderKL encoder (csrc/vae.hip)
SgemmDesc = _struct("SgemmDesc", _ptrs("x", "w", "bias", "residual", "c")
                    + _ints("ldx", "ldw", "ldr", "ldc", "M", "N", "K", "mode", "cin", "batch", "hs", "ws", "ho", "wo", "stride",
                            "pad", "bias_per_row") + [("alpha", c_f32)] + _ints("upsample", "split_bf16"))
Gn32Desc = _struct("Gn32Desc", _ptrs("x", "gamma", "beta", "stats", "y") + _ints("ldx", "ldy", "C", "batch", "hw", "groups")
                   + [("eps", c_f32)] + _ints("act") + _ptrs("partial", "ticket"))
Softmax32Desc = _struct("Softmax32Desc", _ptrs("x") + [("ld", c_i64)] + _ints("rows", "cols"))
VaeConvDesc = _struct("VaeConvDesc", _ptrs("x", "w", "bias", "qw", "qb", "y") + _ints("batch", "h", "wd", "cin", "cout")
                      + [("inv_scaling", c_f32)])
BatchDesc = _struct("BatchDesc", _ptrs("table", "prefix") + _ints("n", "total", "arg", "pad_") + _ptrs("slabs", "tickets"))
Gather16Desc = _struct("Gather16Desc", _ptrs("src", "idx", "out") + [("n", c_i64)])
LoraLnFoldDesc = _struct("LoraLnFoldDesc", _ptrs("items") + _ints("n", "pad_"))
LORA_LNFOLD_ITEM_I64 = 7          # slh_lora_lnfold_item as int64 words: a, gamma, beta, a_out, s_out, c_out, rows | K << 32
VaeSampleDesc = _struct("VaeSampleDesc", _ptrs("moments", "post_noise", "noise", "latent_f32", "noisy_f32", "noisy_bf16")
                        + _ints("batch", "hw") + [("scaling", c_f32), ("sqrt_alpha", c_f32), ("sqrt_one_minus_alpha", c_f32)]
                        + _ints("pad_"))

# order of slh_desc_sizes()
_SIZE_ORDER = [GemmDesc, SkinnyDesc, GemvDesc, GnDesc, GnBwdDesc, LnDesc, LnBwdDesc, AttnDesc, TransposeDesc,
               AttnBwdDesc, TembedDesc, ConvInDesc, EwDesc, CfgDdimDesc, LossDesc, WgradDesc, AdamwDesc, MemsetDesc,
               LoraCdgradDesc, TembLoraBwdDesc, SgemmDesc, Gn32Desc, Softmax32Desc, VaeConvDesc, VaeSampleDesc, LionDesc,
               BatchDesc, Gather16Desc, LoraLnFoldDesc]

# opcodes (enum in sliders_hip.h)
OP_GEMM, OP_SKINNY, OP_GEMV, OP_GN_STATS, OP_GN_APPLY, OP_LAYERNORM, OP_ATTN_FWD, OP_TRANSPOSE_HEADS = range(1, 9)
OP_TEMBED, OP_CONV_IN, OP_ELEMENTWISE, OP_CFG_DDIM, OP_LOSS, OP_WGRAD, OP_ADAMW = range(9, 16)
OP_GN_BWD_STATS, OP_GN_BWD_APPLY, OP_LAYERNORM_BWD, OP_ATTN_BWD, OP_MEMSET = range(16, 21)
OP_LORA_CONV_DGRAD, OP_TEMB_LORA_BWD = 21, 22
OP_SGEMM, OP_GN32_STATS, OP_GN32_APPLY, OP_SOFTMAX32, OP_VAE_CONV_IN, OP_VAE_MOMENTS, OP_VAE_SAMPLE, OP_VAE_POST_QUANT = range(23, 31)
OP_LION = 31
OP_WGRAD_BATCH, OP_TRANSPOSE_BATCH, OP_GATHER16, OP_GN_FUSED = 32, 33, 34, 35
OP_LORA_LN_FOLD = 36

EW_COPY, EW_ADD, EW_GEGLU_FWD, EW_GEGLU_BWD, EW_UPSAMPLE_BWD, EW_COLSUM = range(6)

# C entry point per opcode, for direct (non-program) calls
_ENTRY = {
    OP_GEMM: ("slh_gemm", GemmDesc), OP_SKINNY: ("slh_skinny", SkinnyDesc), OP_GEMV: ("slh_gemv", GemvDesc),
    OP_GN_STATS: ("slh_gn_stats", GnDesc), OP_GN_APPLY: ("slh_gn_apply", GnDesc),
    OP_LAYERNORM: ("slh_layernorm", LnDesc), OP_ATTN_FWD: ("slh_attn_fwd", AttnDesc),
    OP_TRANSPOSE_HEADS: ("slh_transpose_heads", TransposeDesc), OP_TEMBED: ("slh_timestep_embed", TembedDesc),
    OP_CONV_IN: ("slh_conv_in", ConvInDesc), OP_ELEMENTWISE: ("slh_elementwise", EwDesc),
    OP_CFG_DDIM: ("slh_cfg_ddim", CfgDdimDesc), OP_LOSS: ("slh_guidance_loss", LossDesc),
    OP_WGRAD: ("slh_lora_wgrad", WgradDesc), OP_ADAMW: ("slh_adamw", AdamwDesc),
    OP_GN_BWD_STATS: ("slh_gn_bwd_stats", GnBwdDesc), OP_GN_BWD_APPLY: ("slh_gn_bwd_apply", GnBwdDesc),
    OP_LAYERNORM_BWD: ("slh_layernorm_bwd", LnBwdDesc), OP_ATTN_BWD: ("slh_attn_bwd", AttnBwdDesc),
    OP_LORA_CONV_DGRAD: ("slh_lora_conv_dgrad", LoraCdgradDesc), OP_TEMB_LORA_BWD: ("slh_temb_lora_bwd", TembLoraBwdDesc),
    OP_SGEMM: ("slh_sgemm", SgemmDesc), OP_GN32_STATS: ("slh_gn32_stats", Gn32Desc), OP_GN32_APPLY: ("slh_gn32_apply", Gn32Desc),
    OP_SOFTMAX32: ("slh_softmax32", Softmax32Desc), OP_VAE_CONV_IN: ("slh_vae_conv_in", VaeConvDesc),
    OP_VAE_MOMENTS: ("slh_vae_moments", VaeConvDesc), OP_VAE_SAMPLE: ("slh_vae_sample", VaeSampleDesc),
    OP_VAE_POST_QUANT: ("slh_vae_post_quant", VaeConvDesc), OP_LION: ("slh_lion", LionDesc),
    OP_WGRAD_BATCH: ("slh_lora_wgrad_batch", BatchDesc), OP_TRANSPOSE_BATCH: ("slh_transpose_heads_batch", BatchDesc),
    OP_GATHER16: ("slh_gather16", Gather16Desc), OP_GN_FUSED: ("slh_gn_fused", GnDesc),
    OP_LORA_LN_FOLD: ("slh_lora_ln_fold", LoraLnFoldDesc),
}

EXPORTS = ["slh_version", "slh_last_error", "slh_run_program", "slh_desc_sizes", "slh_graph_capture", "slh_graph_launch",
           "slh_graph_destroy", "slh_gemm_variant", "slh_gemm_kernel_name", "slh_gemm5_ok", "slh_gemm7_ok", "slh_attn_fwd_carries_touch", "slh_gn_row_blocks", "slh_gn_clusters", "slh_gn32_row_blocks",
           "slh_lora_wgrad_blocks", "slh_lora_wgrad_single_blocks", "slh_transpose_heads_blocks", "slh_gn_fused_ok"] + [v[0] for v in _ENTRY.values()]


class SlidersHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libsliders_hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlidersHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; run "
            f"`make -C {os.path.join(_HERE, 'csrc')}` (or __graft_entry__.build()).")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; load it first so this library binds to the SAME HIP
    # runtime instance (streams and device pointers are shared with torch).  Loading /opt/rocm's copy first
    # leaves two runtimes in the process and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.slh_last_error.restype = C.c_char_p
    lib.slh_run_program.argtypes = [c_vp, c_i64, c_vp]
    lib.slh_run_program.restype = c_i32
    lib.slh_desc_sizes.argtypes = [C.POINTER(c_i32), c_i32]
    lib.slh_graph_capture.argtypes = [c_vp, c_i64, C.POINTER(c_vp)]
    lib.slh_graph_capture.restype = c_i32
    lib.slh_graph_launch.argtypes = [c_vp, c_vp]
    lib.slh_graph_launch.restype = c_i32
    lib.slh_graph_destroy.argtypes = [c_vp]
    lib.slh_graph_destroy.restype = c_i32
    for name, desc in _ENTRY.values():
        fn = getattr(lib, name)
        fn.argtypes = [C.POINTER(desc), c_vp]
        fn.restype = c_i32
    sizes = (c_i32 * 64)()
    n = lib.slh_desc_sizes(sizes, 64)
    if n != len(_SIZE_ORDER):
        raise SlidersHipError(f"binding/library mismatch: library reports {n} descriptors, binding has {len(_SIZE_ORDER)}")
    for i, d in enumerate(_SIZE_ORDER):
        if C.sizeof(d) != sizes[i]:
            raise SlidersHipError(f"binding/library mismatch: sizeof({d.__name__}) = {C.sizeof(d)} vs {sizes[i]}")
    _lib = lib
    return lib


def last_error() -> str:
    return load().slh_last_error().decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise SlidersHipError(f"{what} failed (rc={rc}): {last_error()}")


def gn_workspace(channels: int, hw: int, groups: int):
    """(rows of the [rows][groups][2] fp32 partial-sum scratch, zeroed uint32 tickets) PER SAMPLE of slh_gn_stats /
    slh_gn_bwd_stats (include/sliders_hip.h, slh_gn_desc)."""
    l = load()
    r = l.slh_gn_row_blocks(channels, hw, groups)
    if r <= 0:
        raise SlidersHipError(f"slh_gn_row_blocks({channels}, {hw}, {groups}): unsupported shape")
    c = l.slh_gn_clusters(r)
    return r + c, 1 + c


TILE_64x160 = 0x5425      # the 64 x 160 tile of csrc/gemm5.hip (bits 12-15 = 5: family; 4 ring slots; 2 x 5 blocks of 16 x 16 per wave)


def gemm5_ok(desc) -> bool:
    """slh_gemm5_ok: the 64 x 160 tile can run this descriptor (dense, packed weights, M % 64 == 0, N % 160 == 0, bias / residual / ln_out)"""
    lib = load()
    lib.slh_gemm5_ok.argtypes = [C.POINTER(GemmDesc)]
    lib.slh_gemm5_ok.restype = c_i32
    return bool(lib.slh_gemm5_ok(C.byref(desc)))


def gn_fused_ok(channels: int, hw: int, groups: int) -> int:
    """slh_gn_fused (statistics + normalisation in one launch) applies to this shape: 0 no, 1 the register-resident kernel for
    tiny tensors, 2 sibling workgroups over a cache-resident slab"""
    return int(load().slh_gn_fused_ok(channels, hw, groups))


def gn32_workspace(hw: int):
    l = load()
    r = l.slh_gn32_row_blocks(hw)
    c = l.slh_gn_clusters(r)
    return r + c, 1 + c


_BATCH_KINDS = {OP_WGRAD_BATCH: ("slh_lora_wgrad_blocks", WgradDesc), OP_TRANSPOSE_BATCH: ("slh_transpose_heads_blocks", TransposeDesc)}


def batch_table(opcode: int, descs, device, arg: int = 0):
    """n problems of one kind -> (BatchDesc, tensors to keep alive): the descriptors packed into a device byte table and the
    int32 prefix sums of the workgroups each needs (include/sliders_hip.h, slh_batch_desc).  Every descriptor is validated by
    the library here, at plan time (slh_*_blocks); device = None builds a pointer-less descriptor for dry-run planning."""
    import torch
    fn_name, dtype = _BATCH_KINDS[opcode]
    l = load()
    fn = getattr(l, fn_name)
    fn.argtypes = [C.POINTER(dtype)]
    fn.restype = c_i32
    prefix = [0]
    for d in descs:
        assert isinstance(d, dtype)
        nb = fn(C.byref(d))         # host-side geometry only: also in dry-run planning (the arena is sized from it)
        if nb <= 0:
            raise SlidersHipError(f"{fn_name}: {last_error()}")
        prefix.append(prefix[-1] + nb)
    if device is None:
        return BatchDesc(table=0x1000, prefix=0x1000, n=len(descs), total=prefix[-1], arg=arg), ()
    raw = b"".join(bytes(d) for d in descs)
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    pre = torch.tensor(prefix, dtype=torch.int32, device=device)
    return BatchDesc(table=table.data_ptr(), prefix=pre.data_ptr(), n=len(descs), total=prefix[-1], arg=arg), (table, pre)


def wgrad_single_blocks(desc) -> int:
    """Workgroups (= slabs = tickets) of a stand-alone slh_lora_wgrad launch with the fixed-order reduction."""
    lib = load()
    lib.slh_lora_wgrad_single_blocks.argtypes = [C.POINTER(WgradDesc)]
    lib.slh_lora_wgrad_single_blocks.restype = c_i32
    nb = lib.slh_lora_wgrad_single_blocks(C.byref(desc))
    if nb <= 0:
        raise SlidersHipError(f"slh_lora_wgrad_single_blocks: {last_error()}")
    return nb


TILE_128x256 = 0x7648     # four-wave tiles of csrc/gemm7.hip (bits 12-15 = 7; ring slots | X blocks | W blocks of 16 rows per wave)
TILE_128x160 = 0x7645
TILE_256x320 = 0x748A     # eight compute waves (gemm7w_kernel): GEGLU.proj as one round of 256 workgroups


def gemm7_ok(desc) -> bool:
    """slh_gemm7_ok: the tile of csrc/gemm7.hip named by desc.tile (0x7648 / 0x7645 / 0x748a) can run this descriptor"""
    lib = load()
    lib.slh_gemm7_ok.argtypes = [C.POINTER(GemmDesc)]
    lib.slh_gemm7_ok.restype = c_i32
    return bool(lib.slh_gemm7_ok(C.byref(desc)))


def attn_carries_touch(desc) -> bool:
    """slh_attn_fwd_carries_touch: this attention launch runs the key-split form, whose idle workgroup slots can stream weights for a
    later product (slh_attn_desc.pf_*)"""
    lib = load()
    lib.slh_attn_fwd_carries_touch.argtypes = [C.POINTER(AttnDesc)]
    lib.slh_attn_fwd_carries_touch.restype = c_i32
    return bool(lib.slh_attn_fwd_carries_touch(C.byref(desc)))


def gemm_variant(desc) -> int:
    """(MI<<8)|(NI<<4)|mode of the gemm_kernel<MI,NI,mode> instantiation slh_gemm launches for desc."""
    lib = load()
    lib.slh_gemm_variant.argtypes = [C.POINTER(GemmDesc)]
    lib.slh_gemm_variant.restype = c_i32
    return lib.slh_gemm_variant(C.byref(desc))


def gemm_kernel_name(desc) -> str:
    """slh_gemm_kernel_name: the kernel instantiation slh_gemm would launch for this descriptor, spelled as rocprofv3 prints it
    (e.g. 'gemm8pb_kernel<1, 5, 0, false>').  The library runs its whole dispatch with the launch replaced by a record of the template
    it selected, so a new tile can never be mis-named here; raises (with the library's message) for a descriptor slh_gemm refuses."""
    lib = load()
    lib.slh_gemm_kernel_name.argtypes = [C.POINTER(GemmDesc), C.c_char_p, c_i32]
    lib.slh_gemm_kernel_name.restype = c_i32
    buf = C.create_string_buffer(160)
    rc = lib.slh_gemm_kernel_name(C.byref(desc), buf, 160)
    if rc != 0:
        raise SlidersHipError(f"slh_gemm_kernel_name failed ({rc}): {lib.slh_last_error().decode()}")
    return buf.value.decode()


def call(opcode: int, desc, stream: int):
    """Launch one op directly (used by the per-kernel parity tests)."""
    lib = load()
    name, _ = _ENTRY[opcode]
    check(getattr(lib, name)(C.byref(desc), c_vp(stream)), name)


_GRAPHS_ON = os.environ.get("SLIDERS_HIPGRAPH", "1") != "0"


class Program:
    """Flat command buffer replayed by slh_run_program with one C call - or, once it has been replayed GRAPH_AFTER times
    unchanged and is long enough to matter, as one hipGraph submission (slh_graph_capture / slh_graph_launch).
    SLIDERS_HIPGRAPH=0 keeps plain launches."""

    GRAPH_AFTER = 2          # eager replays before capture (first runs also load the code objects)
    GRAPH_MIN_OPS = 64

    def __init__(self):
        self._chunks = []
        self.n_ops = 0
        self._buf = None
        self.op_names = []
        self.ops = []          # (opcode, descriptor) kept for per-op timing / tuning tools
        self._graphs = None    # captured graph handle
        self._runs = 0

    def _drop_graphs(self):
        if self._graphs is not None:
            load().slh_graph_destroy(self._graphs)
        self._graphs = None
        self._runs = 0

    def __del__(self):
        try:
            if self._graphs and _lib is not None:
                self._drop_graphs()
        except Exception:
            pass

    def mark(self) -> int:
        """Number of recorded ops; truncate(mark) forgets everything recorded since (planner rollbacks)."""
        return self.n_ops

    def truncate(self, n: int):
        if n >= self.n_ops:
            return
        del self._chunks[n:], self.op_names[n:], self.ops[n:]
        self.n_ops = n
        self._buf = None
        if self._graphs:
            self._drop_graphs()

    def add(self, opcode: int, desc, name: str = ""):
        raw = bytes(desc)
        pad = (-len(raw)) % 8
        hdr = C.c_int32 * 2
        self._chunks.append(bytes(hdr(opcode, len(raw))) + raw + b"\0" * pad)
        self.n_ops += 1
        self.op_names.append(name)
        self.ops.append((opcode, desc))
        self._buf = None
        if self._graphs:
            self._drop_graphs()

    def memset(self, ptr: int, nbytes: int, value: int = 0, name: str = "memset"):
        self.add(OP_MEMSET, MemsetDesc(ptr=ptr, nbytes=nbytes, value=value, pad=0), name)

    def extend(self, other: "Program"):
        self._chunks.extend(other._chunks)
        self.ops.extend(other.ops)
        self.n_ops += other.n_ops
        self.op_names.extend(other.op_names)
        self._buf = None
        if self._graphs:
            self._drop_graphs()

    def finalize(self):
        if self._buf is None:
            data = b"".join(self._chunks)
            self._buf = C.create_string_buffer(data, len(data))
        return self._buf

    def capture(self) -> bool:
        """Capture the graph now instead of after GRAPH_AFTER plain replays (recording launches nothing).  For programs
        whose kernels have all run before: the trainer's per-step variants of a pass it has already replayed."""
        if self._graphs is None and self.n_ops >= self.GRAPH_MIN_OPS and _GRAPHS_ON:
            buf = self.finalize()
            h = c_vp()
            check(load().slh_graph_capture(C.cast(buf, c_vp), len(buf), C.byref(h)), "slh_graph_capture")
            self._graphs = h
        return self._graphs is not None

    def run(self, stream: int, graph: bool = True):
        buf = self.finalize()
        lib = load()
        if graph and self.n_ops >= self.GRAPH_MIN_OPS and _GRAPHS_ON:
            g = self._graphs
            if g is None and self._runs >= self.GRAPH_AFTER:
                self.capture()
                g = self._graphs
            if g is not None:
                check(lib.slh_graph_launch(g, c_vp(stream)), "slh_graph_launch")
                return
        self._runs += 1
        check(lib.slh_run_program(C.cast(buf, c_vp), len(buf), c_vp(stream)), "slh_run_program")
