"""CPU tests of the host side: YAML schema vs the reference's parse (golden), C-ABI exports vs the header,
static planner invariants on a virtual arena (no GPU memory), checkpoint file interchange."""
import ctypes
import json
import os
import re

import pytest
import torch

from sliders_amd import lib
from sliders_amd.arena import Arena
from sliders_amd.config import CONFIGS
from sliders_amd.config_util import load_config_from_yaml
from sliders_amd.lora_store import LoraStore
from sliders_amd.modules import build_tree
from sliders_amd.planner import BackwardPlan, UNetPlan
from sliders_amd.prompt_util import PromptEmbedsPair, PromptSettings, load_prompts_from_yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_yaml_schema_matches_reference_parse():
    gold = json.load(open(os.path.join(G, "schema.json")))
    cfg = load_config_from_yaml(os.path.join(G, "config_sample.yaml"))
    assert json.loads(cfg.model_dump_json()) == gold["config"]
    plain = load_prompts_from_yaml(os.path.join(G, "prompts_sample.yaml"))
    assert [json.loads(p.model_dump_json()) for p in plain] == gold["prompts"]
    attr = load_prompts_from_yaml(os.path.join(G, "prompts_sample.yaml"), ["male", "female"])
    assert [json.loads(p.model_dump_json()) for p in attr] == gold["prompts_attr"]
    # unknown keys (the GPT prompt files' `guidance:`) are ignored -> guidance_scale falls back to 1.0
    assert plain[1].guidance_scale == 1.0 and plain[0].guidance_scale == 4
    with pytest.raises(Exception):
        PromptSettings(positive="x")


def test_prompt_pair_loss_formula():
    st = PromptSettings(target="a", action="enhance", guidance_scale=2.0)
    pair = PromptEmbedsPair(torch.nn.MSELoss(), None, None, None, None, st)
    t, p, u, n = [torch.randn(1, 4, 8, 8) for _ in range(4)]
    got = pair.loss(target_latents=t, positive_latents=p, unconditional_latents=u, neutral_latents=n)
    assert torch.allclose(got, ((t - (n + 2.0 * (p - u))) ** 2).mean())
    pair.action = "erase"
    got = pair.loss(target_latents=t, positive_latents=p, unconditional_latents=u, neutral_latents=n)
    assert torch.allclose(got, ((t - (n - 2.0 * (p - u))) ** 2).mean())


def test_c_abi_exports_every_declared_symbol():
    """libsliders_hip.so loads without a GPU and exports exactly the entry points include/sliders_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "sliders_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(slh_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 25
    l = lib.load()
    for sym in sorted(declared):
        assert hasattr(l, sym), f"{sym} declared in sliders_hip.h but not exported"
    assert declared == set(lib.EXPORTS) | {"slh_gemm_variant"}
    assert l.slh_version() >= 1
    # descriptor sizes were cross-checked inside lib.load(); a bad descriptor is rejected with a message
    d = lib.GemmDesc()
    assert l.slh_gemm(ctypes.byref(d), None) != 0
    assert b"slh_gemm" in l.slh_last_error()


class _FakeWeights:
    def __init__(self, cfg):
        self.temb_offsets, self.resnet_paths, off = {}, [], 0
        for n, m in build_tree(cfg).named_modules():
            if m.cls == "ResnetBlock2D":
                self.resnet_paths.append(n)
                self.temb_offsets[n] = off
                off += m.out_dim
        self.temb_total = off
        self.packed = True
        self.kv_all_offset = {}
        self.kv_all_vbase = 0
        self.gemm_shape = {}
        self.ln_fold = True
        self.geglu16 = True          # WeightStore's default: GEGLU.proj also in the 16 | 16 block order

    def ptr(self, name):
        # a distinct, 256-byte aligned fake address per tensor name (the planner's weight-touch pass matches launches by pointer)
        if not hasattr(self, "_ids"):
            self._ids = {}
        return 0x10000000 + 0x100 * self._ids.setdefault(name, len(self._ids) + 1)

    def has(self, name):
        return True


@pytest.mark.parametrize("name,hw,method", [("tiny_sdxl", 16, "noxattn"), ("tiny_sd1", 16, "full"), ("sdxl", 128, "noxattn")])
def test_planner_static_invariants(name, hw, method):
    cfg = CONFIGS[name]()
    store = LoraStore(cfg, train_method=method, init="none")
    store.temb_tcol = torch.zeros(1, dtype=torch.int32)
    counts = {}
    for mode in ("off", "on", "train"):
        va, vz = Arena(1 << 50, None), Arena(1 << 40, None)
        p = UNetPlan(cfg, _FakeWeights(cfg), va, vz, 2, hw, hw, 77, store if mode != "off" else None, mode, 0x10)
        ops = list(p.prog.ops)
        counts[mode] = len(ops)
        # every allocation is 256-byte aligned and allocations never overlap
        spans = sorted((s, e) for s, e, _ in va.allocs)
        assert all(s % 256 == 0 for s, _ in spans)
        assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
        gemms = [d for o, d in ops if o == lib.OP_GEMM]
        assert all(d.K % 64 == 0 and d.N % 4 == 0 and d.ldc % 4 == 0 for d in gemms)
        assert all(d.w_layout == 1 for d in gemms), "frozen weights are streamed tile-packed"
        if mode == "off":
            assert not any(d.lora_t or d.lora_down for d in gemms), "adapters off must not launch any LoRA work"
            assert not any(o == lib.OP_SKINNY for o, d in ops[:-1])
        else:
            n_lora = sum(1 for d in gemms if d.lora_t or d.lora_down)
            assert n_lora > 0
        if mode == "train":
            bw = BackwardPlan(p, 1, 1, 0x20)
            assert len(bw.prog.ops) > len(ops)
            dg = [d for o, d in bw.prog.ops if o == lib.OP_GEMM]
            assert all(d.M == d_.M for d, d_ in zip(dg[:1], dg[:1]))
            assert all(d.K % 64 == 0 for d in dg)
            # the GEGLU backward rides in the backward-data product of the Linear behind it (geglu = 2) ...
            n_geglu = sum(1 for d in gemms if d.geglu == 1)
            assert n_geglu > 0 and all(d.geglu_pre for d in gemms if d.geglu == 1)
            assert sum(1 for d in dg if d.geglu == 2) == n_geglu and all(d.geglu_pre and d.ldc >= 2 * d.N for d in dg if d.geglu == 2)
            assert not any(o == lib.OP_ELEMENTWISE and d.op == lib.EW_GEGLU_BWD for o, d in bw.prog.ops)
            # ... and dO^T of a self-attention with 64-wide heads comes out of the product that makes dO (both layouts)
            n_dot = sum(1 for d in dg if d.vt_out)
            assert all(d.vt_also_c == 1 and d.vt_col0 == 0 for d in dg if d.vt_out)
            if name == "sdxl":
                n_self = sum(1 for n in p.prog.op_names if n.endswith("attn1.sdpa"))
                assert n_dot == n_self and not any(n.endswith(".dot") for n in bw.prog.op_names)
    assert counts["off"] <= counts["on"] <= counts["train"]
    if name == "sdxl":
        assert counts["off"] < 1300     # one launch per fused op: ~1.15k for the whole SDXL UNet


def test_tile_64x160_entries_and_fallback():
    """The tuned table's entries for the 64 x 160 tile (csrc/gemm5.hip): every launch of the SDXL 1024^2 plans that is given that tile is
    one the tile can run (slh_gemm5_ok, the library's own rule, also what tuning.tile_ok asks), its LayerNorm producers leave
    80-column chunks and their consumers are told so; a launch of the same SHAPE that needs more than the tile's epilogue offers
    falls back to the 128 x 128 ring tile, not to the untuned heuristic."""
    from sliders_amd.tuning import tile_ok, tuned_tile
    cfg = CONFIGS["sdxl"]()
    store = LoraStore(cfg, train_method="noxattn", init="none")
    store.temb_tcol = torch.zeros(1, dtype=torch.int32)
    seen = 0
    for mode, B in (("on", 2), ("train", 2), ("off", 3)):
        va, vz = Arena(1 << 50, None), Arena(1 << 40, None)
        p = UNetPlan(cfg, _FakeWeights(cfg), va, vz, B, 128, 128, 77, store if mode != "off" else None, mode, 0x10)
        gemms = [d for o, d in p.prog.ops if o == lib.OP_GEMM]
        g5 = [d for d in gemms if (d.tile >> 12) & 15 == 5]
        assert all(lib.gemm5_ok(d) and tile_ok(d, d.tile) for d in g5)
        assert all(d.M % 64 == 0 and d.N % 160 == 0 and not d.ln_in for d in g5)      # (round 6: one entry carries a fused adapter, which the tile supports)
        if mode == "off":
            assert not g5                      # M = 3072: 1.5 rounds of 64 x 160 tiles, measured slower - no entries
            continue
        assert len(g5) >= 120 and {d.tile for d in g5} == {0x5425, 0x5525}      # (the 8192 x 640 products moved on to the 128 x 160 four-wave tile in round 6)
        seen += len(g5)
        # chunk statistics: producers on the tile leave N / 80 chunks, and whoever folds that LayerNorm merges N / 80 chunks of it
        # (the four-wave tile of gemm7.hip whose waves own 80 columns - 0x7645 - leaves 80-column chunks as well)
        g7 = [d for d in gemms if (d.tile >> 12) & 15 == 7]
        assert all(lib.gemm7_ok(d) for d in g7)
        prod = {d.ln_out: d.N // 80 for d in g5 + [d for d in g7 if (d.tile & 15) % 5 == 0] if d.ln_out}
        cons = [d for d in gemms if d.ln_in in prod]
        assert prod and cons and all(d.ln_in_chunks == prod[d.ln_in] and d.K == 80 * d.ln_in_chunks for d in cons)
        assert all(d.ln_in_chunks * 64 == d.K for d in gemms if d.ln_in and d.ln_in not in prod)
    assert seen
    # same key, a feature the tile has no epilogue for (a per-sample row bias): the ring tile it replaced runs
    d = lib.GemmDesc(a0=0x1000, w=0x2000, c=0x3000, lda0=1280, ca0=1280, mode=0, stride=1, ldw=0, M=2048, N=1280, K=1280, ldc=1280,
                     rows_per_sample=1024, w_layout=1)
    assert tuned_tile(d) == 0x5425
    d.rowbias, d.ld_rowbias = 0x4000, 1280
    assert not lib.gemm5_ok(d) and tuned_tile(d) == 0x4412


@pytest.mark.parametrize("train_method", ["noxattn", "xattn", "xattn-strict", "selfattn", "full"])
@pytest.mark.parametrize("nb", [1, 2])
def test_every_recorded_tile_can_run_its_launch(train_method, nb):
    """Every GEMM a plan records - no-grad, training forward AND backward, every train_method of lora.py:126-147, one or two samples
    carrying a gradient (prompt batch_size 1 / 2) - names a tile the library accepts for that launch.  The table lookup happens before
    the planner attaches V^T / dO^T stores and the GEGLU backward, so a 64 x 160 entry could land on a launch that tile has no epilogue
    for (xattn at nb = 2 recorded 70 attn1.out dgrads on 0x5425 with vt_out; slh_gemm5_launch refuses them)."""
    from sliders_amd.tuning import tile_ok
    cfg = CONFIGS["sdxl"]()
    store = LoraStore(cfg, train_method=train_method, init="none")
    store.temb_tcol = torch.zeros(1, dtype=torch.int32)
    B = 2 * nb
    n5 = 0
    for mode in ("on", "train"):
        va, vz = Arena(1 << 50, None), Arena(1 << 40, None)
        p = UNetPlan(cfg, _FakeWeights(cfg), va, vz, B, 128, 128, 77, store, mode, 0x10)
        progs = [p.prog]
        if mode == "train":
            progs.append(BackwardPlan(p, nb, nb, 0x20).prog)
        for prog in progs:
            for (o, d), nm in zip(prog.ops, prog.op_names):
                if o != lib.OP_GEMM:
                    continue
                if (d.tile >> 12) & 15 == 5:
                    n5 += 1
                    assert lib.gemm5_ok(d), f"{nm}: tile {d.tile:#x} cannot run this launch"
                assert not d.tile or tile_ok(d, d.tile), f"{nm}: tile {d.tile:#x}"
    assert n5 > 0


@pytest.mark.parametrize("name,hw", [("sdxl", 128), ("sd2", 64), ("sd1", 64)])
def test_planner_fusions_of_the_no_grad_pass(name, hw):
    """Launch-count invariants of the adapters-on no-grad pass (the one the denoise loop replays): with 64-wide heads
    (SDXL, SD-2.x) every self-attention reads the V third of its fused q|k|v projection head-transposed straight from that
    GEMM's epilogue (vt_out) - no transpose launch per block; SD-1.x (40/80/160-wide heads, zero-padded d-tiles) keeps the
    transpose kernel.  With the batched text K/V path the plan also carries the program the loop replays from step 2 on."""
    from sliders_amd.weights import WeightStore   # noqa: F401  (kv_all path needs the real offsets only on the GPU plan)
    cfg = CONFIGS[name]()
    store = LoraStore(cfg, train_method="noxattn", init="none")
    store.temb_tcol = torch.zeros(1, dtype=torch.int32)
    va, vz = Arena(1 << 50, None), Arena(1 << 40, None)
    p = UNetPlan(cfg, _FakeWeights(cfg), va, vz, 2, hw, hw, 77, store, "on", 0x10)
    ops = list(p.prog.ops)
    names = list(p.prog.op_names)
    # weights of the big products (GEGLU.proj, ff.net.2, q|k|v of the 1280-channel level) are touched from the idle workgroup slots of
    # an earlier 160-tile launch (slh_gemm_desc.pf_*): every touch rides AHEAD of the product that streams those bytes, at most
    # TOUCH_WINDOW ops ahead
    from sliders_amd.planner import TOUCH_WINDOW
    wpos = {}
    for i, (o, d) in enumerate(ops):
        if o == lib.OP_GEMM:
            wpos.setdefault(d.w, i)
    touches = [(i, d) for i, (o, d) in enumerate(ops) if o in (lib.OP_GEMM, lib.OP_ATTN_FWD) and d.pf_ptr]
    if name == "sdxl":
        # (round 5: attn2.to_out / proj_out / ff.net.2 of the 1280-channel level run on the 64 x 160 tile - 256 workgroups, no idle slots;
        # round 6: attn1.to_out as well, its touch rides on the key-split self-attention launch in front of it (slh_attn_desc.pf_*) - so a
        # block has two carriers, attn1.sdpa and attn2.to_q, for its three big matrices: GEGLU.proj and ff.net.2 are touched, the next
        # block's q|k|v is not; measured with this assignment)
        assert len(touches) >= 120
        assert sum(1 for i, d in touches if ops[i][0] == lib.OP_ATTN_FWD) >= 55
    assert all(d.pf_bytes >= 6 << 20 and (ops[i][0] == lib.OP_ATTN_FWD and lib.attn_carries_touch(d) or (d.tile & 0xFFFFFF) == 0x4412)
               for i, d in touches)
    for j, dj in touches:       # the bytes a launch touches are the packed weights of a product at most TOUCH_WINDOW ops LATER
        later = [d for o, d in ops[j + 1:j + 1 + TOUCH_WINDOW] if o == lib.OP_GEMM and d.w == dj.pf_ptr]
        assert later and dj.pf_bytes == (later[0].N + 63) // 64 * 64 * later[0].K * 2, (j, hex(dj.pf_ptr))
    assert len({d.pf_ptr for _, d in touches}) == len(touches), "no matrix is touched twice in a pass"
    n_self = sum(1 for n in p.prog.op_names if n.endswith("attn1.sdpa"))
    n_tr = sum(1 for o, _ in ops if o == lib.OP_TRANSPOSE_HEADS)
    n_vt = sum(1 for o, d in ops if o == lib.OP_GEMM and d.vt_out)
    assert n_self > 0
    if name in ("sdxl", "sd2"):
        assert n_vt == n_self and all(d.vt_D == 64 and d.vt_col0 == 2 * d.N // 3 for o, d in ops if o == lib.OP_GEMM and d.vt_out)
        assert n_tr <= n_self          # what is left are the cross-attention V transposes (one per block here: fake weights have no batched K/V)
    else:
        assert n_vt == 0 and n_tr >= n_self
    # train-mode plans keep V row-major for the attention backward: the transposed copy comes on top (vt_also_c), and the
    # GEGLU product keeps its pre-activation (geglu_pre) instead of a separate elementwise launch
    pt = UNetPlan(cfg, _FakeWeights(cfg), Arena(1 << 50, None), Arena(1 << 40, None), 2, hw, hw, 77, store, "train", 0x10)
    assert all(d.vt_also_c == 1 for o, d in pt.prog.ops if o == lib.OP_GEMM and d.vt_out)
    assert sum(1 for o, d in pt.prog.ops if o == lib.OP_GEMM and d.vt_out) == (n_self if name in ("sdxl", "sd2") else 0)
    assert all(d.geglu_pre for o, d in pt.prog.ops if o == lib.OP_GEMM and d.geglu)
    assert sum(1 for o, d in pt.prog.ops if o == lib.OP_GEMM and d.geglu) == n_self
    if p.prog_text_cached is not None:
        assert p.prog_text_cached.n_ops == p.prog.n_ops - 2 - sum(1 for o, _ in ops if o == lib.OP_LORA_LN_FOLD)      # (+ their touches)
    # round 4: GEGLU.proj in the 16 | 16 block order in the no-grad passes (any tile), the 32 | 32 order with geglu_pre in training
    assert all(d.geglu == 3 for o, d in ops if o == lib.OP_GEMM and d.geglu)
    assert all(d.geglu == 1 for o, d in pt.prog.ops if o == lib.OP_GEMM and d.geglu)
    n_fold = sum(1 for o, d in ops if o == lib.OP_GEMM and d.ln_in and d.lora_down)
    n_head = sum(1 for o, _ in ops if o == lib.OP_LORA_LN_FOLD)
    if name == "sdxl":
        # norm1 folded into the adapter-carrying q|k|v of the 1280-channel level (60 blocks; its tuned tile is the 128 x 256
        # ping-pong tile): one slh_lora_ln_fold launch at the head of the program rebuilds A . gamma for all of them, and only the
        # 640-channel level keeps its LayerNorm launches
        assert n_fold == 60 and n_head == 1 and ops[1][0] == lib.OP_LORA_LN_FOLD and ops[1][1].n == 60
        assert all(d.ln_lora_s and d.ln_lora_c and (d.tile >> 12) & 15 == 8 and d.vt_out for o, d in ops if o == lib.OP_GEMM and d.ln_in and d.lora_down)
        assert sum(1 for o, _ in ops if o == lib.OP_LAYERNORM) == 10           # of 210 LayerNorms: norm1 of the ten 640-channel blocks
    else:
        assert n_fold == 0 and n_head == 0
    assert not any(d.ln_in and d.lora_down for o, d in pt.prog.ops if o == lib.OP_GEMM), "training passes keep the LayerNorm launch"


def test_checkpoint_file_is_reference_loadable(tmp_path):
    """save_weights writes a torch .pt OrderedDict with the reference's keys; it strict-loads into the oracle
    restatement of the reference's LoRANetwork (the notebooks' `network.load_state_dict(torch.load(path))`)."""
    from oracle.lora_oracle import LoRANetworkOracle
    from oracle.unet_oracle import build_unet
    cfg = CONFIGS["tiny_sdxl"]()
    s = LoraStore(cfg, train_method="noxattn")
    path = tmp_path / "slider_alpha1.0_rank4_noxattn_last.pt"
    torch.save(s.state_dict(torch.bfloat16), path)
    sd = torch.load(path)
    assert all(v.dtype == torch.bfloat16 for v in sd.values())
    net = build_unet("tiny_sdxl", seed=0)
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    nw.load_state_dict(sd, strict=True)
    assert list(nw.state_dict().keys()) == list(sd.keys())


def test_gemm_weight_tile_packing_layout():
    """pack_gemm_w must produce exactly the order include/sliders_hip.h documents for w_layout = 1 (the kernel
    addresses it with plain pointer arithmetic), and unpack_gemm_w must invert it."""
    import torch
    from sliders_amd.weights import pack_gemm_w, unpack_gemm_w
    n, k = 200, 320
    w = torch.arange(n * k, dtype=torch.float32).reshape(n, k)
    flat = pack_gemm_w(w)
    npad = 256
    assert flat.numel() == npad * k
    g = torch.Generator().manual_seed(0)
    for _ in range(200):
        r = int(torch.randint(0, npad, (1,), generator=g))
        c = int(torch.randint(0, k, (1,), generator=g))
        rr, slot, e = r & 63, (c & 63) >> 3, c & 7
        off = ((r >> 6) * (k // 64) + (c >> 6)) * 4096 + rr * 64 + ((slot ^ ((rr >> 1) & 7)) << 3) + e
        assert flat[off].item() == (w[r, c].item() if r < n else 0.0)
    assert torch.equal(unpack_gemm_w(flat, n, k), w)


def test_c_abi_rejects_bad_descriptors_before_any_launch():
    """Argument validation runs before the first HIP call, so it is testable without a GPU: every unsupported shape,
    alignment or flag combination returns non-zero and names the entry point (no silent fallback, no launch)."""
    one = 0x1000          # any non-null pointer: validation never dereferences device memory

    def gemm(**kw):
        base = dict(a0=one, w=one, c=one, lda0=64, ca0=64, mode=0, stride=1, ldw=64, M=64, N=64, K=64, ldc=64,
                    rows_per_sample=64)
        base.update(kw)
        return lib.GemmDesc(**base)

    bad = {
        "K multiple of 64": gemm(K=96, ca0=96, lda0=96, ldw=96),
        "N multiple of 4": gemm(N=66),
        "dense K": gemm(ca0=128, lda0=128),                            # K != ca0 + ca1
        "a1/ca1": gemm(ca1=64, K=128),                                 # second source declared but a1 null
        "bad tile": gemm(tile=0x33),
        "8-wave": gemm(tile=0x4021),
        "bad mode": gemm(mode=2),
        "conv K": gemm(mode=1, batch=1, hs=8, ws=8, ho=8, wo=8, K=64 * 8),
        "geglu": gemm(geglu=1, residual=one, ld_res=64),
        "fused lora": gemm(lora_down=one, lora_up=one, lora_scale=one, lora_groups=1, lora_rank=8, ld_t=4),
        "w_layout": gemm(w_layout=3),
    }
    for what, d in bad.items():
        with pytest.raises(lib.SlidersHipError, match="slh_gemm"):
            lib.call(lib.OP_GEMM, d, 0)
    with pytest.raises(lib.SlidersHipError, match="slh_layernorm"):
        lib.call(lib.OP_LAYERNORM, lib.LnDesc(x=one, y=one, gamma=one, beta=one, M=4, C=2048, ldx=2048, ldy=2048), 0)
    with pytest.raises(lib.SlidersHipError, match="slh_attn"):
        lib.call(lib.OP_ATTN_FWD, lib.AttnDesc(q=one, k=one, vt=one, o=one, B=1, H=1, Tq=64, Tk=64, ldq=64, ldk=64,
                                               ldvt=64, ldo=64, scale=1.0, D=200), 0)
    # a malformed command buffer is refused as a whole
    buf = ctypes.create_string_buffer(b"\x63\x00\x00\x00\x08\x00\x00\x00" + b"\0" * 8, 16)
    assert lib.load().slh_run_program(ctypes.cast(buf, ctypes.c_void_p), 16, None) != 0
    assert b"slh_run_program" in lib.load().slh_last_error()


def test_diffusers_directory_loader_and_weight_packing(tmp_path):
    """model_util.load_unet_state reads the layout the reference loads through diffusers (model_util.py:67-72,
    169-174: <dir>/unet/config.json + diffusion_pytorch_model.safetensors), and WeightStore repacks every matrix the
    GEMM streams losslessly (the tile-packed form unpacks to the fused / permuted source rows)."""
    from safetensors.torch import save_file
    from oracle.unet_oracle import build_unet
    from sliders_amd.model_util import load_unet_state
    from sliders_amd.weights import WeightStore

    net = build_unet("tiny_sdxl", seed=3)
    sd = {k: v.contiguous() for k, v in net.state_dict().items()}
    cfg0 = CONFIGS["tiny_sdxl"]()
    unet_dir = tmp_path / "model" / "unet"
    unet_dir.mkdir(parents=True)
    raw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg0.__dict__.items()}
    raw.update({"_class_name": "UNet2DConditionModel", "_diffusers_version": "0.20.2", "act_fn": "silu"})   # extra keys ignored
    (unet_dir / "config.json").write_text(json.dumps(raw))
    save_file(sd, str(unet_dir / "diffusion_pytorch_model.safetensors"))
    cfg, loaded = load_unet_state(str(tmp_path / "model"))
    assert cfg == cfg0
    assert loaded.keys() == sd.keys() and all(torch.equal(loaded[k], sd[k]) for k in sd)

    ws = WeightStore(cfg, loaded, "cpu", dtype=torch.float32)
    p = "down_blocks.1.attentions.0.transformer_blocks.0"
    qkv = torch.cat([sd[f"{p}.attn1.to_{x}.weight"] for x in "qkv"], 0)
    assert torch.equal(ws.gemm_matrix(f"{p}.attn1.qkv.w"), qkv)
    conv = sd["down_blocks.0.resnets.0.conv1.weight"]
    assert torch.equal(ws.gemm_matrix("down_blocks.0.resnets.0.conv1.w"), conv.permute(0, 2, 3, 1).reshape(conv.shape[0], -1))
    # batched text K/V: [all K rows | all V rows], each block's slice at the recorded offsets
    k_off, v_off = ws.kv_all_offset[f"{p}.attn2"]
    allkv = ws.gemm_matrix("attn2_kv_all.w")
    wk, wv = sd[f"{p}.attn2.to_k.weight"], sd[f"{p}.attn2.to_v.weight"]
    assert torch.equal(allkv[k_off:k_off + wk.shape[0]], wk) and torch.equal(allkv[v_off:v_off + wv.shape[0]], wv)
    assert allkv.shape[0] == 2 * ws.kv_all_vbase
    with pytest.raises(FileNotFoundError):
        load_unet_state(str(tmp_path / "nope"))


def test_cli_keeps_the_reference_flags():
    """train_lora.py / train_lora_xl.py flags (train_lora_xl.py:419-470) parse unchanged."""
    from sliders_amd.cli import build_parser
    a = build_parser(True).parse_args(["--config_file", "c.yaml", "--prompts_file", "p.yaml", "--alpha", "1.0", "--rank", "4",
                                       "--device", "0", "--name", "ageslider", "--attributes", "male, female"])
    got = (a.config_file, a.prompts_file, a.alpha, a.rank, a.device, a.name, a.attributes)
    assert got == ("c.yaml", "p.yaml", 1.0, 4, 0, "ageslider", "male, female")


def test_encode_prompts_xl_contract_with_tiny_clip():
    """train_util.py:77-133: per prompt the penultimate hidden states of both CLIP text encoders concatenated on the
    feature axis, and the pooled (projected) output of the SECOND encoder.  Tiny random-init CLIP models and a stub
    tokenizer stand in for the checkpoints the build image does not have."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from sliders_amd.model_util import encode_prompts_xl

    class Tok:
        model_max_length = 77

        def __call__(self, prompts, padding, max_length, truncation, return_tensors):
            g = torch.Generator().manual_seed(len(prompts))
            ids = torch.randint(1, 90, (len(prompts), max_length), generator=g)
            ids[:, -1] = 99                      # eos = highest id: CLIP pools at argmax(input_ids)
            return type("Enc", (), {"input_ids": ids})()

    torch.manual_seed(0)
    c1 = CLIPTextConfig(vocab_size=100, hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                        max_position_embeddings=77, eos_token_id=99)
    c2 = CLIPTextConfig(vocab_size=100, hidden_size=48, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                        max_position_embeddings=77, projection_dim=40, eos_token_id=99)
    e1, e2 = CLIPTextModel(c1).eval(), CLIPTextModelWithProjection(c2).eval()
    prompts = ["a person", "an old person"]
    text, pooled = encode_prompts_xl([Tok(), Tok()], [e1, e2], prompts, num_images_per_prompt=2)
    assert text.shape == (4, 77, 32 + 48) and pooled.shape == (4, 40)
    ids = Tok()(prompts, "max_length", 77, True, "pt").input_ids
    with torch.no_grad():
        h1 = e1(ids, output_hidden_states=True).hidden_states[-2]
        o2 = e2(ids, output_hidden_states=True)
    want = torch.cat([h1, o2.hidden_states[-2]], -1).repeat_interleave(2, dim=0)
    assert torch.allclose(text, want) and torch.allclose(pooled, o2.text_embeds.repeat_interleave(2, dim=0))


def test_build_pairs_caches_prompts_and_concatenates_like_the_reference():
    """cli.build_pairs: one encoder call per distinct prompt (PromptEmbedsCache, train_lora_xl.py:121-151) and ctx =
    cat([unconditional, X]).repeat_interleave(batch_size) (concat_embeddings, train_util.py:136-141)."""
    from sliders_amd.cli import build_pairs
    cfg = CONFIGS["tiny_sdxl"]()
    calls = []

    def encode(text):
        calls.append(text)
        g = torch.Generator().manual_seed(len(text))
        return torch.randn(1, 77, cfg.cross_attention_dim, generator=g), torch.randn(1, cfg.pooled_dim, generator=g)

    prompts = [PromptSettings(target="person", positive="old person", unconditional="young person", neutral="person",
                              action="enhance", guidance_scale=4, batch_size=2),
               PromptSettings(target="person", positive="smiling person", unconditional="", neutral="person",
                              action="erase", guidance_scale=2, batch_size=1)]
    pairs = build_pairs(cfg, prompts, encode, "cpu")
    assert sorted(calls) == sorted({"person", "old person", "young person", "smiling person", ""})
    (s0, p0), (s1, p1) = pairs
    assert p0.ctx_target.shape == (4, 77, cfg.cross_attention_dim) and p0.pooled_positive.shape == (4, cfg.pooled_dim)
    assert p1.ctx_target.shape == (2, 77, cfg.cross_attention_dim)
    unc, tgt = encode("young person")[0].bfloat16(), encode("person")[0].bfloat16()
    assert torch.equal(p0.ctx_target, torch.cat([unc, tgt]).repeat_interleave(2, dim=0))
    assert torch.equal(p0.ctx_uncond, torch.cat([unc, unc]).repeat_interleave(2, dim=0))
    assert (p0.guidance_scale, p0.action, p1.guidance_scale, p1.action) == (4, "enhance", 2, "erase")
    assert p0.ctx_target.dtype == torch.bfloat16


# ---- config surface: honoured or rejected, never silently ignored (train_lora_xl.py:92-107, 170-203, 394-410) ----
def _cfg(**train):
    from sliders_amd import config_util
    base = dict(prompts_file="p.yaml", pretrained_model=dict(name_or_path="x"), network=dict(type="c3lier", rank=8, alpha=2.0,
                                                                                           training_method="noxattn"),
                train=dict(precision="bfloat16", noise_scheduler="ddim", iterations=1000, lr=2e-4, optimizer="AdamW",
                           lr_scheduler="constant", max_denoising_steps=50),
                save=dict(name="s", path="./models", per_steps=500, precision="bfloat16"),
                logging=dict(use_wandb=False, verbose=False), other=dict(use_xformers=True))
    base["train"].update(train)
    return config_util.RootConfig(**base)


def test_cli_rank_and_alpha_override_only_when_given():
    from sliders_amd.cli import apply_cli_overrides, build_parser
    a = build_parser(True).parse_args(["--config_file", "c.yaml"])
    assert a.rank is None and a.alpha is None
    c = apply_cli_overrides(_cfg(), a)
    assert c.network.rank == 8 and c.network.alpha == 2.0                 # the YAML's values survive
    assert c.save.name == "s_alpha2.0_rank8_noxattn" and c.save.path == "./models/s_alpha2.0_rank8_noxattn"
    a = build_parser(True).parse_args(["--config_file", "c.yaml", "--rank", "4", "--alpha", "1", "--name", "age"])
    c = apply_cli_overrides(_cfg(), a)
    assert c.network.rank == 4 and c.network.alpha == 1.0 and c.save.name == "age_alpha1.0_rank4_noxattn"


def test_optimizer_options_and_rejections():
    from sliders_amd.cli import check_supported, optimizer_options
    assert optimizer_options(_cfg().train) == {"betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0.01, "name": "adamw"}
    o = optimizer_options(_cfg(optimizer="adamw", optimizer_args="weight_decay=0.1 betas=(0.8,0.99) eps=1e-6").train)
    assert o == {"betas": (0.8, 0.99), "eps": 1e-6, "weight_decay": 0.1, "name": "adamw"}
    assert optimizer_options(_cfg(optimizer="adam").train)["weight_decay"] == 0.0
    # lion_pytorch.Lion defaults (requirements.txt:5) and its own argument names only
    assert optimizer_options(_cfg(optimizer="Lion").train) == {"betas": (0.9, 0.99), "weight_decay": 0.0, "eps": 0.0, "name": "lion"}
    assert optimizer_options(_cfg(optimizer="lion", optimizer_args="weight_decay=0.02 betas=(0.95,0.98)").train)["betas"] == (0.95, 0.98)
    for bad in (dict(optimizer="lion", optimizer_args="eps=1e-6"), dict(optimizer="lion", optimizer_args="use_triton=True"),
                dict(optimizer="prodigy", optimizer_args="amsgrad=True"), dict(optimizer="adam8bit"), dict(optimizer="dadaptsgd"), dict(optimizer="dadaptlion", optimizer_args="eps=1e-6"),
                dict(optimizer="adam", optimizer_args="weight_decay=0.01"), dict(optimizer_args="amsgrad=True"),
                dict(precision="fp16"), dict(precision="float32", optimizer="lion"), dict(lr_scheduler="linear")):
        with pytest.raises(NotImplementedError):
            check_supported(_cfg(**bad))
    # train.precision float32 (config_util.py:75-83 -> weight_dtype, train_lora_xl.py:60-61, 84-90): fp32 adapter state with adam / adamw
    from sliders_amd.cli import adapter_state_dtype
    check_supported(_cfg(precision="float32"))
    check_supported(_cfg(precision="fp32", optimizer="adam"))
    assert adapter_state_dtype(_cfg(precision="float32"), rank=1) == torch.float32 and adapter_state_dtype(_cfg()) == torch.bfloat16
    # prodigyopt.Prodigy's arguments and defaults (requirements.txt: prodigyopt==1.0), lr = 1 is the user's business
    o = optimizer_options(_cfg(optimizer="Prodigy", optimizer_args="weight_decay=0.01 d_coef=2.0 safeguard_warmup=True").train)
    assert o["name"] == "prodigy" and o["weight_decay"] == 0.01 and o["d_coef"] == 2.0 and o["safeguard_warmup"] is True
    assert o["d0"] == 1e-6 and o["betas"] == (0.9, 0.999) and o["decouple"] is True and o["growth_rate"] == float("inf")
    # dadaptation 3.1's two classes with their own arguments and defaults (train_util.py:339-346)
    o = optimizer_options(_cfg(optimizer="DAdaptAdam", optimizer_args="decouple=True weight_decay=0.01 growth_rate=1.02").train)
    assert o["name"] == "dadaptadam" and o["decouple"] is True and o["weight_decay"] == 0.01 and o["growth_rate"] == 1.02
    assert o["d0"] == 1e-6 and o["eps"] == 1e-8 and o["use_bias_correction"] is False
    o = optimizer_options(_cfg(optimizer="dadaptlion").train)
    assert o["name"] == "dadaptlion" and o["betas"] == (0.9, 0.999) and o["weight_decay"] == 0.0
    for name in ("ddim", "ddpm", "lms", "euler_a"):      # model_util.py:230-277: all four for text sliders
        check_supported(_cfg(noise_scheduler=name))
    check_supported(_cfg(), image_slider=True)
    check_supported(_cfg(noise_scheduler="ddpm"), image_slider=True)     # same timestep grid and add_noise as ddim, no .step()
    with pytest.raises(NotImplementedError):             # image sliders: the fused noising step is DDIM-table only
        check_supported(_cfg(noise_scheduler="euler_a"), image_slider=True)
    with pytest.raises(ValueError):
        check_supported(_cfg(lr_scheduler="nope"))
    c = _cfg()
    c.pretrained_model.v2 = c.pretrained_model.v_pred = True      # SD-2.x 768-v: implemented (sd2 config + v-prediction DDIM)
    check_supported(c)
    check_supported(_cfg(lr_scheduler="cosine", optimizer="adam"))


@pytest.mark.parametrize("name", ["constant", "cosine", "cosine_with_restarts", "step"])
def test_lr_schedule_follows_the_torch_scheduler(name):
    """The host-side schedule hands slh_adamw, at iteration i, the lr torch's scheduler would have set after i calls of
    lr_scheduler.step() (train_lora_xl.py:346-347)."""
    from sliders_amd.cli import LrSchedule
    from sliders_amd.train_util import get_lr_scheduler
    lr0, n = 2e-4, 300
    sch = LrSchedule(name, lr0, n)
    prm = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.AdamW([prm], lr=lr0)
    ref = get_lr_scheduler(name, opt, max_iterations=n, lr_min=lr0 / 100)
    seen = []
    for i in range(n):
        assert sch.current() == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12)
        seen.append(sch.current())
        opt.step()
        ref.step()
        sch.step()
    if name != "constant":
        assert min(seen) < lr0
    else:
        assert all(v == lr0 for v in seen)


def test_synthetic_prompt_embeddings_are_stable_across_processes():
    """Seeded by a stable digest of the prompt text, not by Python's per-process salted hash()."""
    import subprocess
    import sys
    code = ("import sys, zlib, torch; sys.path.insert(0, %r); from sliders_amd import cli; "
            "print(zlib.crc32('a photo of a person'.encode()) & 0xFFFFFF)" % ROOT)
    outs = {subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, PYTHONHASHSEED=str(s))).strip()
            for s in (1, 2)}
    assert len(outs) == 1
    src = open(os.path.join(ROOT, "sliders_amd", "cli.py")).read()
    assert "hash(text)" not in src


def test_single_file_checkpoint_is_rejected_before_loading(tmp_path):
    from sliders_amd.cli import check_model_files
    f = tmp_path / "model.safetensors"
    f.write_bytes(b"\0")
    with pytest.raises(NotImplementedError, match="diffusers-format model directory"):
        check_model_files(str(f))
    check_model_files(str(tmp_path))      # a directory passes this gate (the loader then validates its contents)


def test_image_slider_cli_keeps_the_reference_flags():
    """trainscripts/imagesliders/train_lora-scale-xl.py:464-541: --alpha required, folder / scale lists, stylecheck."""
    from sliders_amd.cli_image import build_parser, parse_folders_scales
    ps = build_parser()
    with pytest.raises(SystemExit):
        ps.parse_args(["--config_file", "c.yaml", "--folder_main", "d/"])            # --alpha is required
    a = ps.parse_args(["--config_file", "c.yaml", "--alpha", "1", "--folder_main", "datasets/eyesize/", "--name", "eye"])
    assert (a.folders, a.scales, a.rank, a.stylecheck) == ("verylow, low, high, veryhigh", "-2, -1, 1, 2", 4, None)
    f, s = parse_folders_scales("bigsize, smallsize", "1, -1")
    assert f == ["bigsize", "smallsize"] and s == [1, -1]
    with pytest.raises(Exception):
        parse_folders_scales("a, b, c", "1, -1")
    for script in ("train_lora-scale-xl.py", "train_lora-scale.py"):
        assert os.path.isfile(os.path.join(ROOT, "trainscripts", "imagesliders", script))


def test_vae_encoder_refuses_cpu():
    from sliders_amd.vae import VaeEncoder, random_vae_state_dict
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        VaeEncoder(random_vae_state_dict((128, 128, 128, 128)), "cpu")


def test_generate_parses_the_slider_file_name_like_the_notebooks():
    """XL-sliders-inference.ipynb cell 6: rank / alpha / train_method come from substrings of the checkpoint path."""
    from sliders_amd.generate import build_parser, parse_slider_name
    assert parse_slider_name("models/age_alpha1.0_rank4_noxattn/age_alpha1.0_rank4_noxattn_last.pt") == (4, 1.0, "noxattn")
    assert parse_slider_name("eyesize_alpha2_rank8_full_500steps.pt") == (8, 2.0, "full")
    assert parse_slider_name("x_alpha1.0_rank4_xattn-strict_last.pt")[2] == "xattn-strict"
    assert parse_slider_name("unnamed.pt") == (4, 1.0, "noxattn")
    a = build_parser().parse_args(["--scales=-2,0,2", "--start_noise", "800", "--synthetic"])
    assert a.start_noise == 800 and a.ddim_steps == 50 and a.guidance_scale == 7.5


def test_host_helpers_match_the_reference_run():
    """tests/golden/host_helpers.json is the output of the reference's own train_util.py under fixed seeds
    (tests/golden/make_golden.py host_helpers): the same RNG consumption, order and results here
    (train_util.py:20-57, 136-141, 298-333, 376-419)."""
    import json
    import types
    from sliders_amd import train_util as tu
    from sliders_amd.cli import LrSchedule
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_helpers.json")))
    for seed in (0, 1, 7):
        torch.manual_seed(seed)
        assert [list(tu.get_random_resolution_in_bucket(b)) for b in (512, 1024, 512, 768, 1024)] == gold[f"bucket/{seed}"]
        torch.manual_seed(seed)
        got = [tu.get_add_time_ids(h, w, dynamic_crops=True).tolist() for h, w in ((512, 512), (1024, 1024), (768, 512))]
        assert got == gold[f"time_ids_dynamic/{seed}"]
    assert tu.get_add_time_ids(1024, 768, dynamic_crops=False).tolist() == gold["time_ids_static"]
    assert tu.get_add_time_ids(1000, 1001, dtype=torch.bfloat16).float().tolist() == gold["time_ids_bf16"]     # quirk D.8
    lat = tu.get_initial_latents(types.SimpleNamespace(init_noise_sigma=2.5), 2, 64, 96, 3, generator=torch.Generator().manual_seed(3))
    g = gold["initial_latents"]
    assert list(lat.shape) == g["shape"] and float(lat.double().sum()) == g["sum"] and float(lat.flatten()[0]) == g["first"]
    assert torch.equal(lat[:2], lat[2:4]) == g["repeat_equal"]
    a, b = torch.arange(6.0).view(1, 2, 3), 10 + torch.arange(6.0).view(1, 2, 3)
    assert tu.concat_embeddings(a, b, 2).tolist() == gold["concat_embeddings"]
    idx = (0, 1, 9, 10, 99, 100, 101, 299, 300, 500, 699, 700, 999)
    for name in ("constant", "cosine", "cosine_with_restarts", "step"):
        s = LrSchedule(name, 2e-4, 1000)
        lrs = []
        for i in range(1000):
            lrs.append(s.current())
            s.step()
        assert [lrs[i] for i in idx] == gold[f"lr/{name}"], name


def test_vae_checkpoints_with_deprecated_attention_names_are_normalised():
    """SD-1.4 / 1.5 VAE files name the mid-block attention query / key / value / proj_attn (1x1 convs in the oldest ones);
    diffusers renames them at load time, VaeEncoder / VaeDecoder read the file directly and must do the same."""
    from sliders_amd.vae import normalize_vae_key, random_vae_state_dict
    new = random_vae_state_dict(boc=(32, 64), decoder=True)
    back = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    old = {}
    for k, v in new.items():
        ko = k
        for n, o in back.items():
            if f".attentions.0.{n}." in k:
                ko = k.replace(f".attentions.0.{n}.", f".attentions.0.{o}.")
        old[ko] = v
    assert any(".query." in k for k in old) and any(".proj_attn." in k for k in old)
    assert {normalize_vae_key(k) for k in old} == set(new)
    assert all(normalize_vae_key(k) == k for k in new)
    # names that merely contain the words elsewhere stay untouched
    assert normalize_vae_key("encoder.down_blocks.0.resnets.0.conv1.weight") == "encoder.down_blocks.0.resnets.0.conv1.weight"


def test_step_sampler_ranks_agree_on_shared_draws_including_dynamic_resolution():
    """Data parallel: k and everything drawn from the shared stream (dynamic_resolution bucket, dynamic_crops seed) must be
    identical on every rank in every step - the ranks do the same amount of work and all-reduce once - while the pair index
    differs by rank (sliders_amd/cli.py:225-240 relies on this)."""
    from sliders_amd.parallel import StepSampler
    from sliders_amd.train_util import get_random_resolution_in_bucket
    seqs = []
    for rank in range(2):
        s = StepSampler(7, rank, 2, 8, 50)
        out = []
        for _ in range(20):
            k, pi = s.next()
            st = torch.random.get_rng_state()
            torch.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=s.shared).item()))
            hw = get_random_resolution_in_bucket(1024)
            torch.random.set_rng_state(st)
            out.append((k, pi, hw))
        seqs.append(out)
    assert [(k, hw) for k, _, hw in seqs[0]] == [(k, hw) for k, _, hw in seqs[1]]
    assert all(a[1] != b[1] for a, b in zip(*seqs))
    assert all(1 <= k <= 49 for k, _, _ in seqs[0])
    # the SD-1.x image-slider script draws k from 1 .. max-2 (train_lora-scale.py:186-188)
    s = StepSampler(1, 0, 1, 3, 49)
    assert max(s.next()[0] for _ in range(400)) == 48


def test_bench_pairs_counter_traffic_only_with_identical_kernel_sources(tmp_path):
    """bench.py attaches roofline.traffic (per-launch bytes from committed --pmc passes) only when the tree it runs from is the tree
    the passes were taken on - as a whole, or in every file the reported kernel is built from (its translation unit, the headers it
    includes, the public header, every tile table).  A changed file of ANOTHER kernel does not void the pairing; a changed file of
    this kernel, a changed tile table or a counter file without per-file hashes does."""
    import json
    import bench
    from sliders_amd import srchash
    k = "gemm_kernel<1, 2, 0, 4, false, 4, false>"
    here = srchash.file_hashes()
    assert set(srchash.kernel_files(k, here)) >= {"gemm.hip", "gemm_common.h", "common.h", "sliders_hip.h"} and \
        "attention.hip" not in srchash.kernel_files(k, here) and all(f in srchash.kernel_files(k, here) for f in here if f.endswith(".json"))
    prof = tmp_path / "profiles"
    prof.mkdir()

    def write(fh, combined):
        json.dump({"tree_head": "abc1234", "kernel_source_hash": combined, "file_hashes": fh,
                   "kernels": {k.replace(", ", "; "): {"fetch_bytes_per_launch": 100, "write_bytes_per_launch": 23, "launches": 1}}},
                  open(prof / "r09_pmc_traffic.json", "w"))
    # 1. identical tree
    write(dict(here), srchash.kernel_source_hash())
    assert bench.pmc_traffic_for(k, str(prof))[0] == 123
    # 2. another kernel's file changed since the passes: still this kernel's numbers
    other = dict(here)
    other["attention.hip"] = "0" * 16
    write(other, "f" * 16)
    t, src = bench.pmc_traffic_for(k, str(prof))
    assert t == 123 and "attention.hip" in src and "byte-identical" in src
    # 3. this kernel's own header changed / a tile table changed / no per-file hashes: no pairing
    for bad in ("gemm_common.h", next(f for f in here if f.endswith(".json"))):
        o2 = dict(here)
        o2[bad] = "0" * 16
        write(o2, "f" * 16)
        assert bench.pmc_traffic_for(k, str(prof))[0] is None
    write({}, "f" * 16)
    t, src = bench.pmc_traffic_for(k, str(prof))
    assert t is None and "re-run" in src



def test_dry_run_plans_the_program_the_real_plan_runs(monkeypatch):
    """UNetEngine sizes its arena by planning against _VirtualWeights (pointer-less).  The dry run must take the same branches as
    the real plan - same launches, same tiles, same arena high-water mark - for every mode: a forgotten attribute (geglu16 in
    round 4: the dry run planned GEGLU.proj on the 32 | 32 path, the real plan on the 16 | 16 path with other tile keys) makes them
    two different programs and the sizing a guess."""
    from sliders_amd.unet import _VirtualWeights
    cfg = CONFIGS["sdxl"]()
    store = LoraStore(cfg, train_method="noxattn", init="none")
    store.temb_tcol = torch.zeros(1, dtype=torch.int32)
    real = _FakeWeights(cfg)
    for mode in ("off", "on", "train"):
        plans = []
        for w in (real, _VirtualWeights(real)):
            va, vz = Arena(1 << 50, None), Arena(1 << 40, None)
            p = UNetPlan(cfg, w, va, vz, 2, 128, 128, 77, store if mode != "off" else None, mode, 0x10)
            plans.append((p, va))
        (pr, ar), (pv, av) = plans
        assert pv.prog.op_names == pr.prog.op_names, mode
        assert [d.tile for o, d in pv.prog.ops if o == lib.OP_GEMM] == [d.tile for o, d in pr.prog.ops if o == lib.OP_GEMM], mode
        assert [d.geglu for o, d in pv.prog.ops if o == lib.OP_GEMM] == [d.geglu for o, d in pr.prog.ops if o == lib.OP_GEMM], mode
        assert av.high_water >= ar.high_water, mode


def test_refused_layernorm_fold_leaves_nothing_in_the_program(monkeypatch):
    """ln_gemm tries the folded product first and falls back to LayerNorm + product when gemm() refuses.  With SLIDERS_LORA_UNFUSED
    the refusal comes AFTER the adapter's down-projection was recorded (round-4 advisor finding): the rollback must drop that
    launch too, otherwise it writes T into arena bytes that were handed out again."""
    cfg = CONFIGS["sdxl"]()
    store = LoraStore(cfg, train_method="noxattn", init="none")
    store.temb_tcol = torch.zeros(1, dtype=torch.int32)
    monkeypatch.setenv("SLIDERS_LORA_UNFUSED", "1")
    va, vz = Arena(1 << 50, None), Arena(1 << 40, None)
    p = UNetPlan(cfg, _FakeWeights(cfg), va, vz, 2, 128, 128, 77, store, "on", 0x10)
    names = p.prog.op_names
    downs = [n for n in names if n.endswith(".lora_down")]
    assert len(downs) == len(set(downs)), "a rolled-back fold must not leave its down-projection behind"
    # every recorded skinny launch writes a buffer that no LATER allocation overlaps (the stale launch wrote into re-allocated bytes)
    spans = sorted((s, e) for s, e, _ in va.allocs)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    # Program.truncate itself
    prog = lib.Program()
    prog.memset(0x1000, 16, 0, "a")
    m = prog.mark()
    prog.memset(0x2000, 16, 0, "b")
    prog.memset(0x3000, 16, 0, "c")
    prog.truncate(m)
    assert prog.n_ops == 1 and prog.op_names == ["a"] and len(prog.ops) == 1


class _FakeWeightsKvAll(_FakeWeights):
    """_FakeWeights plus the batched text K/V layout of WeightStore (weights.py: all attn2 K / V projections concatenated)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        paths = [n for n, m in build_tree(cfg).named_modules() if n.endswith(".attn2")]
        widths = {}
        for n, m in build_tree(cfg).named_modules():
            if n.endswith(".attn2.to_k"):
                widths[n[:-len(".to_k")]] = m.out_dim
        self.kv_all_vbase = sum(widths[a] for a in paths)
        row = 0
        for a in paths:
            self.kv_all_offset[a] = (row, self.kv_all_vbase + row)
            row += widths[a]
        self.gemm_shape["attn2_kv_all.w"] = (2 * self.kv_all_vbase, cfg.cross_attention_dim)


@pytest.mark.parametrize("ctx_len,fused", [(77, True), (64, False), (48, False), (96, True), (97, False)])
def test_fused_cross_attention_gate_matches_the_kernel_checks(ctx_len, fused):
    """attn2.to_q carries the cross-attention in its epilogue only where slh_gemm accepts xa_*: two 64-key V^T tiles are always
    staged, so the padded key count roundup(ctx_len, 64) must reach 128 (65..96 keys).  Shorter contexts keep the two launches
    instead of failing inside slh_gemm (round-4 advisor finding)."""
    cfg = CONFIGS["sdxl"]()
    va, vz = Arena(1 << 50, None), Arena(1 << 40, None)
    p = UNetPlan(cfg, _FakeWeightsKvAll(cfg), va, vz, 2, 128, 128, ctx_len, None, "off", 0x10)
    assert p.kv_all is not None and p.vt_all is not None
    xa = [d for o, d in p.prog.ops if o == lib.OP_GEMM and d.xa_k]
    assert bool(xa) == fused
    assert all(d.xa_ldvt >= 128 and d.xa_tk == ctx_len and d.N % 64 == 0 and (d.tile & 0xFFFFF) == 0x4412 for d in xa)
    n_cross = sum(1 for n in p.prog.op_names if n.endswith("attn2.sdpa"))
    n_blocks = sum(1 for n in p.prog.op_names if n.endswith("attn2.q"))
    assert n_cross + len(xa) == n_blocks          # every block runs its cross-attention exactly once, one way or the other


def test_fp32_adapter_state_store_layout_and_checkpoint(tmp_path):
    """train.precision float32: LoraStore keeps an fp32 master + fp32 moments next to the bf16 buffer the kernels read; the reference
    draws of the initialisation land in both; the checkpoint is written from the master in fp32 (the reference saves in train.precision,
    quirk D.7) and strict-loads back without loss; a bf16 store has no master at all."""
    cfg = CONFIGS["tiny_sdxl"]()
    torch.manual_seed(4)
    s32 = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", state_dtype=torch.float32)
    torch.manual_seed(4)
    s16 = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn")
    assert s16.master is None and s16.exp_avg.dtype == torch.bfloat16
    assert s32.master.dtype == s32.exp_avg.dtype == s32.exp_avg_sq.dtype == torch.float32 and s32.params.dtype == torch.bfloat16
    assert torch.equal(s32.params, s16.params) and torch.equal(s32.master.to(torch.bfloat16), s32.params)
    assert (s32.master != s32.params.float()).any(), "the master keeps the bits bf16 drops"
    g = torch.Generator().manual_seed(1)
    s32.master.add_(torch.randn(s32.numel, generator=g) * 1e-3)
    sd = s32.state_dict(torch.float32)
    assert all(v.dtype == torch.float32 for v in sd.values())
    torch.save(sd, tmp_path / "a.pt")
    t = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", state_dtype=torch.float32, init="none")
    t.load_state_dict(torch.load(tmp_path / "a.pt"), strict=True)
    assert torch.equal(t.master, s32.master) and torch.equal(t.params, s32.master.to(torch.bfloat16))
    with pytest.raises(NotImplementedError):
        LoraStore(cfg, state_dtype=torch.float16)


def test_bench_gpus_flag_relaunches_or_refuses():
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: the script re-launches itself as N ranks of one node over
    127.0.0.1 - exactly the driver's command line - and refuses, non-zero, when the node shows fewer GPUs; it never runs one rank under
    an `n_gpus: N` label (--gpus was parsed and never read)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.multi_gpu_relaunch(8, argv, {}, 8)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv
    # already a rank of a torch.distributed.run job, or a single-GPU run: nothing to re-launch
    assert bench.multi_gpu_relaunch(8, argv, {"WORLD_SIZE": "8"}, 8) is None
    assert bench.multi_gpu_relaunch(1, [], {}, 0) is None
    # fewer devices than ranks: loud, non-zero
    with pytest.raises(SystemExit) as ex:
        bench.multi_gpu_relaunch(8, argv, {}, 1)
    assert "--gpus 8" in str(ex.value) and "1 GPU" in str(ex.value)
    # end to end on this GPU-less container: the process exits non-zero and prints no JSON line
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={**os.environ, "WORLD_SIZE": ""})
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "n_gpus" not in r.stdout and "--gpus 2" in r.stderr


def test_gemm_kernel_name_comes_from_the_library():
    """slh_gemm_kernel_name: the kernel a descriptor runs on, named by the dispatch code itself (no device needed) - bench.py and the
    PMC pairing use it instead of rebuilding template arguments from tile codes."""
    d = lib.GemmDesc(a0=0x1000, w=0x2000, c=0x3000, lda0=1280, ca0=1280, mode=0, stride=1, ldw=0, M=2048, N=1280, K=1280, ldc=1280,
                     rows_per_sample=1024, w_layout=1)
    for tile, want in ((0x5425, "gemm5_kernel<false, 4>"), (0x5525, "gemm5_kernel<false, 5>"), (0x4412, "gemm_kernel<1, 2, 0, 4, false, 4, false>"),
                       (0x8014, "gemm8pb_kernel<1, 4, 0, false>"), (0x8042, "gemm8p_kernel<0, false>"), (0x11, "gemm_kernel<1, 1, 0, 2, false, 2, false>"),
                       (0x4322, "gemm_kernel<2, 2, 0, 3, false, 4, false>")):
        d.tile = tile
        assert lib.gemm_kernel_name(d) == want, hex(tile)
    d.tile = 0x4422          # 256 x 128 stages (256 + 128) * 128 B = 48 KB per slot: a 4-slot ring does not fit 160 KB, the launcher runs 3
    assert lib.gemm_kernel_name(d) == "gemm_kernel<2, 2, 0, 3, false, 4, false>"
    d.tile, d.K = 0x5425, 100            # a descriptor slh_gemm refuses: the error, not a name
    with pytest.raises(lib.SlidersHipError):
        lib.gemm_kernel_name(d)
