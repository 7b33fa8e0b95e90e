#!/usr/bin/env python3
"""Contract benchmark: concept-slider LoRA training throughput in UNet denoise steps/sec.

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): SDXL-base UNet, rank-4 text slider (noxattn + c3lier, alpha 1), 1024x1024
(latent 128x128), batch 1 (CFG pair B=2), bf16, DDIM-50, AdamW lr 2e-4.  One bench "step" = ONE TRAINING
ITERATION of the reference loop (train_lora_xl.py:162-356): k partial-denoise UNet passes with the adapters on
(k ~ U{1..49}, seeded, shared by all ranks), 3 frozen predictions, 1 target prediction with grad, guidance
loss, backward into the adapters, (all-reduce,) AdamW.  The metric counts UNet denoise steps
(= predict_noise calls = UNet forwards on a CFG pair): value = sum_i (k_i + 4) * N / wall.
Weights are seeded random-init with the real SDXL shapes and text embeddings are seeded randn (no checkpoints
exist offline) - throughput does not depend on the values.

"extra_configs" (N=1 default run only): the same loop on BASELINE configs[1] (SD-1.x 512x512 text slider) and configs[4] (SDXL
image slider, 512x512 pairs, VAE encode on the GPU, with "vae_roofline" for the fp32 VAE GEMM in both arithmetic modes), timed by
this process after the contract line.

Extra JSON objects: "roofline" for the dominant kernel (bf16 MFMA GEMM instantiation with the largest share
of a UNet pass; algorithmic FLOPs / HIP-event time, measured live on the launch stream) and "cpu_baseline"
(the CPU oracle = PyTorch restatement of the reference's diffusers UNet, run in fp32 on the host cores, one UNet denoise step of
the same workload, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks of this node (one process per GPU).  N > 1 outside torch.distributed.run re-launches this script under it; "
                         "default: WORLD_SIZE when launched by torch.distributed.run, else 1")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="sdxl")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--workload", default="text", choices=["text", "image"],
                    help="text: BASELINE configs[2] (the contract line); image: configs[4], SDXL image slider, 512x512 pairs, VAE encode on the GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra_configs objects (BASELINE configs[1] SD-1.x 512x512 text slider, configs[4] SDXL image slider)")
    ap.add_argument("--vae-split-bf16", action="store_true",
                    help="image workload: the VAE's fp32 operands as bf16 hi+lo halves (narrower than the reference's fp32 VAE; "
                         "the default is the exact-fp32 MFMA)")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def gemm_flops(d):
    n = d.N
    return 2.0 * d.M * n * d.K


def pmc_traffic_for(kname, profiles_dir=None, tag=""):
    """(bytes per launch, source note) of kernel `kname` from the newest committed counter passes (profiles/r*_pmc_traffic<tag>.json;
    tag "" = the SDXL 1024x1024 LoRA-on pass, "_sd1_64" / "_sdxl_64" = the passes of the other configurations), or (None, why): the
    figures are attached only when the tree this runs from is the tree the passes were taken on - as a whole, or in every file the
    kernel is built from (sliders_amd/srchash.py) - and only to the configuration the passes ran."""
    import glob
    traffic, tsrc = None, None
    profiles_dir = profiles_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    cands = sorted(glob.glob(os.path.join(profiles_dir, f"r[0-9][0-9]_pmc_traffic{tag}.json")))
    if not cands:
        return None, f"none: no counter passes of this configuration are committed (profiles/r*_pmc_traffic{tag}.json)"
    if cands:                                   # PMC counters cannot be read from inside the timed process: the
        tpath = cands[-1]                       # per-launch HBM-side bytes come from the newest committed --pmc passes
        with open(tpath) as f:
            pmj = json.load(f)
            pm = pmj["kernels"].get(kname.replace(", ", "; "))
            thead = pmj.get("tree_head")
        from sliders_amd.srchash import file_hashes, kernel_files, kernel_source_hash
        here_hash, there_hash = kernel_source_hash(), pmj.get("kernel_source_hash")
        here_files, there_files = file_hashes(), pmj.get("file_hashes") or {}
        need = kernel_files(kname, here_files)
        same_kernel = bool(there_files) and set(here_files) == set(there_files) and all(here_files.get(f) == there_files.get(f) for f in need)
        if pm and there_hash == here_hash:
            traffic = pm["fetch_bytes_per_launch"] + pm["write_bytes_per_launch"]
            tsrc = (f"profiles/{os.path.basename(tpath)} (FETCH_SIZE x2 + WRITE_SIZE per launch, LoRA-on forward pass; counter passes "
                    f"taken on tree {thead or 'unrecorded'}, kernel sources + tile tables hash {there_hash} = this tree's)")
        elif pm and same_kernel:
            # other kernels changed since the counter passes; every file THIS kernel is built from, the public header and all tile
            # tables are byte-identical to the tree the passes were taken on
            changed = sorted(f for f in here_files if here_files[f] != there_files.get(f))
            traffic = pm["fetch_bytes_per_launch"] + pm["write_bytes_per_launch"]
            tsrc = (f"profiles/{os.path.basename(tpath)} (FETCH_SIZE x2 + WRITE_SIZE per launch, LoRA-on forward pass; counter passes "
                    f"taken on tree {thead or 'unrecorded'}: {', '.join(need)} are byte-identical in this tree; changed since: "
                    f"{', '.join(changed)})")
        else:
            # a counter file of OTHER kernels is not evidence about these: say so instead of pairing the numbers silently
            tsrc = (f"none: profiles/{os.path.basename(tpath)} was taken on kernel sources / tile tables hash {there_hash or 'unrecorded'}, "
                    f"this tree hashes {here_hash}" + ("" if pm else f"; it has no row for {kname}") +
                    " - re-run scripts/measure_round.sh <round tag>")
    return traffic, tsrc


def measure_roofline(eng, plan, pmc_tag=""):
    """Time EVERY launch of one LoRA-on UNet denoise pass IN SITU: the pass is replayed op by op in program order on the
    launch stream with a HIP event pair around each launch, so every kernel sees the cache state it sees in the real pass
    (frozen weights cold in HBM, activations warm in L2/MALL).  GEMM launches are grouped by kernel instantiation under
    the name rocprofv3 prints; the instantiation with the largest share of the pass is the roofline kernel:
    achieved = sum of algorithmic FLOPs (2*M*N*K) / sum of event time.  The same replay gives the figures north_star asks
    for: whole-pass TFLOP/s, the attention path against the MFMA peak, and the convolution / GroupNorm / LayerNorm paths
    against the HBM peak (algorithmic bytes: inputs + weights read once, outputs written once)."""
    from sliders_amd import lib
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream

    def launch(opcode, d):
        if opcode in lib._ENTRY:
            lib.call(opcode, d, s)
        else:                                   # memset of the fp32 accumulator arena
            one = lib.Program()
            one.add(opcode, d)
            one.run(s)

    name = lib.gemm_kernel_name     # the instantiation slh_gemm launches for a descriptor, as rocprofv3 prints it (asked of the library)

    for _ in range(2):
        plan.prog.run(s)                        # warm-up passes (also make every input of every op valid)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        plan.prog.run(s)
    torch.cuda.synchronize()
    pass_ms = (time.time() - t0) / 3 * 1e3     # one C call per pass: what the training loop pays
    REPS = 3                                    # op-by-op replays averaged (the two largest kernel groups are ~2 % apart)
    sums = [0.0] * len(plan.prog.ops)
    for _ in range(REPS):
        evs = []
        for opcode, d in plan.prog.ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            launch(opcode, d)
            e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(evs):
            sums[i] += e0.elapsed_time(e1) / REPS
    recs = [(opcode, d, sums[i]) for i, (opcode, d) in enumerate(plan.prog.ops)]
    groups = {}
    acc = {k: dict(ms=0.0, work=0.0, n=0) for k in ("attention", "conv3x3", "groupnorm", "layernorm", "gemm_all")}
    for opcode, d, ms in recs:
        if opcode == lib.OP_GEMM:
            g = groups.setdefault(name(d), dict(ms=0.0, flops=0.0, calls=0))
            g["ms"] += ms
            g["flops"] += gemm_flops(d)
            g["calls"] += 1
            acc["gemm_all"]["ms"] += ms; acc["gemm_all"]["work"] += gemm_flops(d); acc["gemm_all"]["n"] += 1
            if d.mode == 1:
                cin = d.ca0 + d.ca1
                sh = 1 if d.src_xform == 1 else 0
                src_pix = d.batch * d.hs * d.ws
                nbytes = 2.0 * (src_pix * cin + d.M * d.N + d.N * d.K)
                acc["conv3x3"]["ms"] += ms; acc["conv3x3"]["work"] += nbytes; acc["conv3x3"]["n"] += 1
        elif opcode == lib.OP_ATTN_FWD:
            acc["attention"]["ms"] += ms; acc["attention"]["work"] += 4.0 * d.B * d.H * d.Tq * d.Tk * (d.D or 64); acc["attention"]["n"] += 1
        elif opcode in (lib.OP_GN_STATS, lib.OP_GN_APPLY, lib.OP_GN_FUSED):
            if opcode != lib.OP_GN_STATS:          # one read + one write of the tensor for the stats/apply pair (or the fused launch)
                acc["groupnorm"]["work"] += 2.0 * 2.0 * d.batch * d.hw * (d.c0 + d.c1); acc["groupnorm"]["n"] += 1
            acc["groupnorm"]["ms"] += ms
        elif opcode == lib.OP_LAYERNORM:
            acc["layernorm"]["ms"] += ms; acc["layernorm"]["work"] += 2.0 * 2.0 * d.M * d.C; acc["layernorm"]["n"] += 1
    # dominant kernel = the instantiation with the largest share of the pass; instantiations within 5 % of the largest
    # share (two of them trade places from run to run) are ranked by the algorithmic work they carry
    top = max(x["ms"] for x in groups.values())
    kname, g = max(((k, x) for k, x in groups.items() if x["ms"] >= 0.95 * top), key=lambda kv: kv[1]["flops"])
    achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
    table = {k: dict(calls_per_pass=x["calls"], ms_per_pass=round(x["ms"], 3),
                     tflops=round(x["flops"] / (x["ms"] * 1e-3) / 1e12, 1)) for k, x in sorted(groups.items())}
    traffic, tsrc = pmc_traffic_for(kname, tag=pmc_tag) if pmc_tag is not None else (None, "none: no counter passes exist for this pass kind")
    import glob
    # MFMA utilisation from the newest committed counter pass: SQ_VALU_MFMA_BUSY_CYCLES (32 per 32x32x16 MFMA, summed over
    # the SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
    def mfma_util(kernel_prefix):
        if pmc_tag is None:           # a measurement of another configuration's pass is not a measurement of this one
            return None
        c2 = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                           f"r[0-9][0-9]_pmc_mfma_busy_fwd_lora_on{pmc_tag}.csv")))
        if not c2:
            return None
        import csv
        with open(c2[-1]) as f:
            rows = list(csv.DictReader(l for l in f if "," in l))
        busy = act = 0.0
        for r in rows:
            if r["kernel"].startswith(kernel_prefix):
                busy += float(r["SQ_VALU_MFMA_BUSY_CYCLES"]); act += float(r["GRBM_GUI_ACTIVE"])
        return round(busy / (act / 8.0 * 1024.0), 4) if act > 0 else None
    pass_flops = acc["gemm_all"]["work"] + acc["attention"]["work"]
    tf = lambda a: round(a["work"] / (a["ms"] * 1e-3) / 1e12, 1) if a["ms"] > 0 else None
    gbs = lambda a: round(a["work"] / (a["ms"] * 1e-3) / 1e9, 1) if a["ms"] > 0 else None
    paths = {
        "unet_pass": {"ms": round(pass_ms, 2), "launches": len(recs), "algorithmic_tflop": round(pass_flops / 1e12, 3),
                      "tflops": round(pass_flops / (pass_ms * 1e-3) / 1e12, 1),
                      "frac_of_mfma_peak": round(pass_flops / (pass_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)},
        "all_gemm": {"ms_per_pass": round(acc["gemm_all"]["ms"], 2), "tflops": tf(acc["gemm_all"]),
                     "frac_of_mfma_peak": round((tf(acc["gemm_all"]) or 0) / MFMA_PEAK_TFLOPS, 4)},
        "attention": {"ms_per_pass": round(acc["attention"]["ms"], 2), "launches": acc["attention"]["n"], "tflops": tf(acc["attention"]),
                      "frac_of_mfma_peak": round((tf(acc["attention"]) or 0) / MFMA_PEAK_TFLOPS, 4),
                      "mfma_util_pmc": mfma_util("attn_fwd_kernel"),
                      "note": "algorithmic 4*B*H*Tq*Tk*D / event time; mfma_util_pmc = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 "
                              "SIMDs) of the attn_fwd kernels in the newest committed counter pass (profiles/r*_pmc_mfma_busy_*.csv)"},
        "conv3x3": {"ms_per_pass": round(acc["conv3x3"]["ms"], 2), "launches": acc["conv3x3"]["n"], "hbm_gbs": gbs(acc["conv3x3"]),
                    "frac_of_hbm_peak": round((gbs(acc["conv3x3"]) or 0) / HBM_PEAK_GBS, 4),
                    "note": "algorithmic bytes (source pixels + outputs + weights, bf16) / event time: these kernels are MFMA-bound"},
        "groupnorm": {"ms_per_pass": round(acc["groupnorm"]["ms"], 2), "launches": acc["groupnorm"]["n"], "hbm_gbs": gbs(acc["groupnorm"]),
                      "frac_of_hbm_peak": round((gbs(acc["groupnorm"]) or 0) / HBM_PEAK_GBS, 4)},
        "layernorm": {"ms_per_pass": round(acc["layernorm"]["ms"], 2), "launches": acc["layernorm"]["n"], "hbm_gbs": gbs(acc["layernorm"]),
                      "frac_of_hbm_peak": round((gbs(acc["layernorm"]) or 0) / HBM_PEAK_GBS, 4)},
    }
    return {
        "bound": "mfma", "kernel": kname,
        "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": tsrc,
        "mfma_util_pmc": mfma_util(kname.replace(", ", "; ")),
        "flop_per_launch": g["flops"] / g["calls"], "avg_launch_us": round(1e3 * g["ms"] / g["calls"], 2),
        "launches_per_unet_pass": g["calls"], "timing": "in situ, HIP events on the launch stream, mean of 3 op-by-op replays of one LoRA-on pass",
        "dominant_rule": "largest share of the pass; instantiations within 5 % of it are ranked by algorithmic FLOPs",
        "all_gemm_variants": table, "paths": paths,
    }


def cpu_baseline_config0(budget_s=75.0):
    """BASELINE.json configs[0] as SURVEY.md 8(d) defines its CPU line: SD-1.4 architecture, 'age'-style text slider, rank 4
    alpha 1, `noxattn`, batch 1, fp32, ONE training iteration of the reference loop (train_lora.py:155-321) with timesteps_to =
    k = 5 on the host cores through the oracle: k denoise steps with the adapters on + three frozen predictions + the target
    prediction (k + 4 = 9 UNet forwards on a CFG pair) + loss.backward() into the 150 adapters + AdamW.  512 x 512 when one
    forward fits the budget / 11, else 256 x 256 with the rate scaled by latent area (flagged in `sample`)."""
    import torch.nn.functional as F
    from oracle.ddim_oracle import DDIMScheduler
    from oracle.lora_oracle import LoRANetworkOracle
    from oracle.unet_oracle import build_unet
    from sliders_amd.config import CONFIGS
    ncpu = _usable_cpus()
    torch.set_num_threads(ncpu)
    cfg = CONFIGS["sd1"]()
    net = build_unet("sd1", seed=0)
    net.requires_grad_(False)
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    g = torch.Generator().manual_seed(3)
    for m in nw.unet_loras:                       # non-zero up matrices: the denoise chain depends on the adapters
        m.lora_up.weight.data.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.02)
    for p_ in nw.parameters():
        p_.requires_grad_(True)
    opt = torch.optim.AdamW(nw.parameters(), lr=2e-4)
    emb = {n: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for n in ("target", "positive", "neutral", "uncond")}
    k = 5

    def predict(x, which, t, gs):
        e = net(torch.cat([x] * 2), t, torch.cat([emb["uncond"], emb[which]]), None).sample
        u, c = e.chunk(2)
        return u + gs * (c - u)

    def iteration(hw):
        sch = DDIMScheduler()
        t0 = time.time()
        opt.zero_grad()
        with torch.no_grad():
            sch.set_timesteps(50)
            x = torch.randn(1, 4, hw, hw, generator=g)
            with nw:
                for t in sch.timesteps[0:k]:
                    x = sch.step(predict(x, "target", t, 3), t, x).prev_sample
            sch.set_timesteps(1000)
            t_cur = sch.timesteps[int(k * 1000 / 50)]
            pos, neu, unc = (predict(x, w, t_cur, 1) for w in ("positive", "neutral", "uncond"))
        t_fwd = time.time()
        with nw:
            tgt = predict(x, "target", t_cur, 1)
        loss = F.mse_loss(tgt, neu + 4.0 * (pos - unc))
        loss.backward()
        opt.step()
        return time.time() - t0, time.time() - t_fwd, float(loss.detach())

    with torch.no_grad():                          # probe: one forward at 256 x 256 (also creates the oneDNN primitives)
        predict(torch.randn(1, 4, 32, 32, generator=g), "target", torch.tensor(500), 1)
        t0 = time.time()
        predict(torch.randn(1, 4, 32, 32, generator=g), "target", torch.tensor(500), 1)
        f256 = time.time() - t0
    hw = 64 if 4.0 * f256 * 12 < budget_s else 32
    wall, t_train, loss = iteration(hw)
    scale = (64 * 64) / float(hw * hw)
    return {"value": round((k + 4) / (wall * scale), 4), "unit": "UNet denoise steps/s (SD-1.4 512x512 bs 1, rank-4 text slider, fp32)",
            "cores": ncpu, "kind": "port", "iterations_per_s": round(1.0 / (wall * scale), 5),
            "iteration_s": round(wall * scale, 2), "train_forward_backward_adamw_s": round(t_train * scale, 2),
            "sample": (f"one training iteration of the reference loop at k = {k} ({k + 4} forwards on a CFG pair + backward + AdamW) through the "
                       f"CPU oracle, timed at {hw * 8}x{hw * 8}" + ("" if hw == 64 else ", scaled by latent area to 512x512 (extrapolated)") +
                       f"; loss {loss:.4e}"),
            "config": "BASELINE.json configs[0] / SURVEY.md 8(d)"}


def measure_vae_roofline(vae, vae_sd, image, res_px):
    """The VAE encoder's GEMM kernel (slh_sgemm: 3x3 convolutions and the mid-block attention of AutoencoderKL in fp32,
    imagesliders/train_util.py:200-235) in BOTH arithmetic modes on the same image: exact fp32 (v_mfma_f32_32x32x2_f32,
    peak 157.3 TFLOP/s; the default) and the opt-in bf16 hi/lo split (three bf16 MFMAs per product: peak 2500 / 3 TFLOP/s)."""
    from sliders_amd import lib
    from sliders_amd.vae import VaeEncoder
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    out = {}
    for mode, enc in (("split_bf16", vae if vae.split_bf16 else None), ("exact_fp32", vae if not vae.split_bf16 else None)):
        if enc is None:
            enc = VaeEncoder(vae_sd, vae.device, vae.scaling_factor, exact_fp32=(mode == "exact_fp32"))
        enc.encode_moments(image)
        plan = enc.plan(1, res_px, res_px)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            plan.prog.run(s)
        torch.cuda.synchronize()
        enc_ms = (time.time() - t0) / 3 * 1e3
        ms = flops = 0.0
        n = 0
        evs = []
        for opcode, d in plan.prog.ops:
            if opcode == lib.OP_SGEMM:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                lib.call(opcode, d, s)
                e1.record(stream)
                evs.append((d, e0, e1))
            elif opcode in lib._ENTRY:
                lib.call(opcode, d, s)
            else:
                one = lib.Program()
                one.add(opcode, d)
                one.run(s)
        torch.cuda.synchronize()
        for d, e0, e1 in evs:
            ms += e0.elapsed_time(e1); flops += 2.0 * d.M * d.N * d.K; n += 1
        peak = 157.3 if mode == "exact_fp32" else MFMA_PEAK_TFLOPS / 3.0
        ach = flops / (ms * 1e-3) / 1e12
        out[mode] = {"bound": "mfma", "kernel": "sgemm_kernel", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                     "frac": round(ach / peak, 4), "launches": n, "sgemm_ms_per_encode": round(ms, 3),
                     "encode_ms": round(enc_ms, 3), "algorithmic_tflop_per_encode": round(flops / 1e12, 4)}
    out["note"] = ("one %dx%d image; fp32 algorithmic FLOPs 2*M*N*K of every slh_sgemm launch / HIP-event time, op-by-op replay of the "
                   "encoder's command buffer; split mode executes 3 bf16 MFMA FLOPs per algorithmic FLOP" % (res_px, res_px))
    return out


def _mem_limit_gb():
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(path).read().strip()
            if v.isdigit() and int(v) < (1 << 50):
                return int(v) / 2 ** 30
        except OSError:
            pass
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2 ** 30
    except (ValueError, OSError):
        return 64.0


def _usable_cpus():
    """Cores this process may actually run on: affinity mask clipped by the cgroup CPU quota (the GPU boxes expose 256
    hardware threads but grant 16 CPUs of quota; 256 OpenMP threads on that quota run ~500x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


class _Deadline(Exception):
    pass


def cpu_baseline(model, hw):
    """UNet denoise steps on the host cores with the CPU oracle (PyTorch restatement of the reference's
    diffusers UNet).  Bounded to ~10-30 s of CPU work: the dtype (bf16 / fp32) is chosen with a GEMM
    micro-probe (hosts without AMX run bf16 an order of magnitude slower), a 256x256 probe step estimates the
    cost, and if one full-resolution step would exceed the budget the largest resolution that fits is timed and
    the rate is extrapolated by latent area (flagged in `sample`)."""
    from oracle.unet_oracle import build_unet
    from sliders_amd.config import CONFIGS

    def log(msg):
        print(f"[cpu_baseline] {msg}", file=sys.stderr, flush=True)

    ncpu = _usable_cpus()
    torch.set_num_threads(ncpu)
    cfg = CONFIGS[model]()
    need32 = 4 * 2.6e9 / 2 ** 30 * 1.3 if model == "sdxl" else 6.0
    # fp32 whenever it fits: oneDNN bf16 convolutions fall off a cliff on hosts whose bf16 GEMM probe looks fine
    # (measured: 261 s for one 256x256 step in bf16 vs 6.4 s in fp32 on the same class of EPYC host)
    dtype = torch.float32 if _mem_limit_gb() > need32 + 8 else torch.bfloat16
    log(f"{ncpu} usable cores, mem limit {_mem_limit_gb():.0f} GB -> {dtype}")
    net = build_unet(model, device="meta")
    g = torch.Generator().manual_seed(0)
    block = ((torch.rand(1 << 22, generator=g) - 0.5) * 0.05).to(dtype)
    sd = {}
    for k, v in net.state_dict().items():   # cheap init; every weight owns its memory (values do not matter)
        n = v.numel()
        t = block.repeat((n + block.numel() - 1) // block.numel())[:n].reshape(v.shape).clone()
        if "norm" in k and k.endswith(".weight"):
            t = torch.ones_like(t)
        sd[k] = t
    net.load_state_dict(sd, assign=True)
    del sd
    net.eval()
    log("oracle built")
    B = 2

    def run(h):
        x = torch.randn(B, 4, h, h).to(dtype)
        ctx = torch.randn(B, 77, cfg.cross_attention_dim).to(dtype)
        kw = None
        if cfg.is_xl:
            kw = {"text_embeds": torch.randn(B, cfg.pooled_dim).to(dtype),
                  "time_ids": torch.tensor([[h * 8.0, h * 8.0, 0, 0, h * 8.0, h * 8.0]] * B).to(dtype)}
        with torch.no_grad():
            t0 = time.time()
            out = net(x, torch.tensor(500), ctx, kw).sample
            dt = time.time() - t0
        assert torch.isfinite(out.float()).all()
        return dt

    # ladder 64px -> 128px -> ... : go one size up only while 4x the last step (area scaling) fits what is left of
    # the ~30 s budget; a hard SIGALRM deadline bounds the whole measurement whatever the host does
    import signal

    def _alarm(signum, frame):
        raise _Deadline()

    budget, spent = 30.0, 0.0
    use_hw, dt = None, None
    old = signal.signal(signal.SIGALRM, _alarm)
    signal.alarm(90)
    try:
        h = min(8, hw)
        run(h)                      # untimed: oneDNN primitive creation, first-touch of the weights
        while True:
            d = run(h)
            spent += d
            use_hw, dt = h, d
            log(f"step at {h * 8}px: {d:.2f}s")
            if h >= hw or 4.0 * d > budget - spent:
                break
            h *= 2
    except _Deadline:
        log("deadline hit, keeping the largest size that completed")
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)
    if use_hw is None:
        return {"value": None, "unit": "steps/s", "cores": ncpu, "kind": "port",
                "sample": "CPU oracle did not finish one 64x64 step within the 90 s deadline on this host"}
    scale = (hw / use_hw) ** 2
    note = "" if use_hw == hw else (f"; timed at {use_hw * 8}x{use_hw * 8} ({dt:.1f} s) and EXTRAPOLATED x{scale:.0f} by "
                                    f"latent area to {hw * 8}x{hw * 8} (optimistic for the CPU: self-attention grows faster than area)")
    return {"value": round(1.0 / (dt * scale), 5), "unit": "steps/s", "cores": ncpu, "kind": "port",
            "sample": f"1 UNet denoise step (CFG pair B=2, {model}, {'fp32' if dtype == torch.float32 else 'bf16'} "
                      f"torch/oneDNN forward, {dt:.1f} s) with the CPU oracle = PyTorch restatement of the diffusers "
                      f"UNet{note}"}


def params_hash64(store) -> int:
    """64-bit digest of the replicated adapter state (flat parameters, both moments and - train.precision float32 - the fp32 master the
    bf16 parameters are rounded from): sum and xor-fold of the raw 16-bit words, computed on the device, exact in int64.  Equal state ->
    equal digest; used to assert that the ranks still hold bit-identical replicas after the timed region."""
    h = 0
    for i, t in enumerate((store.params, store.exp_avg, store.exp_avg_sq, getattr(store, "master", None))):
        if t is None:
            continue
        w = t.view(torch.int16).to(torch.int64) & 0xFFFF
        idx = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
        h ^= (int((w * ((idx % 65521) + 1)).sum().item()) + (i + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFFFFFFFFFF


def dp_self_check(tr, store, dev, rank, world):
    """After the timed region (N > 1, or N = 1 under torch.distributed.run): every rank contributes a digest of its adapter
    replica and the mean event time of its gradient all-reduces; rank 0 asserts that all replicas are bit-identical (the data-parallel
    contract: same all-reduced gradient, same fused optimizer step everywhere) and reports the compute / communication split."""
    import torch.distributed as dist
    ms = tr.allreduce_ms() if hasattr(tr, "allreduce_ms") else []
    mine = torch.tensor([params_hash64(store), int(1e6 * (sum(ms) / len(ms))) if ms else -1, len(ms)], dtype=torch.int64, device=dev)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    hashes = [int(v[0].item()) for v in allv]
    ar_us = [v[1].item() / 1e3 for v in allv if v[1].item() >= 0]
    same = len(set(hashes)) == 1
    if not same:
        raise RuntimeError(f"data-parallel replicas diverged: adapter-state digests per rank {[hex(h) for h in hashes]}")
    return {"rccl_ranks_seen": len(hashes), "replicas_bit_identical": same, "adapter_state_digest": hex(hashes[0]),
            "allreduce_us_mean_per_rank": [round(u, 1) for u in ar_us], "allreduces_timed_per_rank": int(allv[0][2].item()),
            "allreduce_bytes": int(store.grads.numel() * 4),
            "note": "HIP events around torch.distributed.all_reduce on the stream the backward ran on; the timed loop carries them"}


def multi_gpu_relaunch(gpus, argv, env, visible_devices):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: the command line that runs this same script as N
    ranks of one node (one process per GPU over RCCL, rendezvous on 127.0.0.1), or a SystemExit when this node cannot.  Never a
    1-rank run labelled as N.  Returns None when nothing is to be re-launched (N = 1, or already under torch.distributed.run)."""
    if gpus <= 1 or env.get("WORLD_SIZE"):
        return None
    if visible_devices < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} requested but {visible_devices} GPU(s) are visible on this node; refusing to run "
                         f"fewer ranks under that label (run `python bench.py --gpus {max(visible_devices, 1)}`)")
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    a = parse()
    if a.gpus is None:
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    cmd = multi_gpu_relaunch(a.gpus, sys.argv[1:], os.environ, torch.cuda.device_count())
    if cmd is not None:
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        print("[bench] --gpus %d: re-launching as %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or drop --gpus)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("BENCH_FORCE_PROCESS_GROUP"):      # the second form: the torchrun path on one GPU (tests/test_rccl_gpu.py)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local if torch.distributed.is_initialized() else 0)
    torch.cuda.set_device(dev)
    res = run_config(a, dev, world, rank, main_line=True)
    if rank == 0 and world == 1 and not a.no_extra and a.workload == "text" and a.model == "sdxl" and a.res == 1024:
        # the other single-GPU configurations of BASELINE.json, timed by the same process right after the contract line
        # (untimed for the contract): configs[1] and configs[4]
        extra = []
        for model, res_px, workload in (("sd1", 512, "text"), ("sdxl", 512, "image")):
            b = argparse.Namespace(**vars(a))
            b.model, b.res, b.workload, b.steps, b.warmup, b.no_cpu_baseline = model, res_px, workload, max(a.steps, 6), 1, True
            torch.cuda.empty_cache()
            try:
                extra.append(run_config(b, dev, world, rank, main_line=False))
                if workload == "image":
                    # "value" is measured with the VAE in exact-fp32 MFMA arithmetic (the reference declares its VAE fp32).  The same
                    # loop with the opt-in bf16 hi/lo split (16 mantissa bits per operand - narrower than the reference, so NOT the
                    # reported value) is quoted beside it
                    b2 = argparse.Namespace(**vars(b))
                    b2.vae_split_bf16, b2.no_roofline = True, True
                    torch.cuda.empty_cache()
                    r2 = run_config(b2, dev, world, rank, main_line=False)
                    extra[-1]["value_with_split_bf16_vae_narrower_than_reference"] = r2.get("value")
                    extra[-1]["ms_per_step_with_split_bf16_vae"] = r2.get("ms_per_step")
            except Exception as e:          # the contract line must survive whatever an extra configuration does
                extra.append({"config": {"workload": f"{model} {workload} {res_px}"}, "error": f"{type(e).__name__}: {e}"})
        res["extra_configs"] = extra
    if rank == 0:
        print(json.dumps(res), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()      # ranks > 0 wait for rank 0's (untimed) roofline replay before tearing down
        torch.distributed.destroy_process_group()


def run_config(a, dev, world, rank, main_line):
    from sliders_amd.config import CONFIGS
    from sliders_amd.lora_store import LoraStore
    from sliders_amd.random_init import random_state_dict
    from sliders_amd.trainer import PairEmbeds, SliderTrainer
    from sliders_amd.unet import UNetEngine

    cfg = CONFIGS[a.model]()
    if a.workload == "image":
        a.res = 512 if cfg.is_xl else 256           # the reference resizes every image (train_lora-scale-xl.py:220-221)
    hw = a.res // 8
    eng = UNetEngine(cfg, random_state_dict(cfg, dev, a.seed), dev)
    torch.manual_seed(a.seed)   # identical adapter init on every rank (replicated parameters)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
    # under torch.distributed.run (any N, including the N = 1 the GPU tests launch) the gradient exchange goes through RCCL
    pg = torch.distributed.group.WORLD if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
    tr = SliderTrainer(eng, store, hw, hw, batch_size=1, lr=2e-4, process_group=pg)

    # 4 prompts x 2 attributes = 8 PromptEmbedsPairs (BASELINE configs[2]); synthetic CLIP-shaped embeddings
    g = torch.Generator(device="cpu").manual_seed(1234)
    pairs = []
    for i in range(8):
        e = [torch.randn(1, 77, cfg.cross_attention_dim, generator=g).to(dev, torch.bfloat16) for _ in range(4)]
        pl = [torch.randn(1, cfg.pooled_dim, generator=g).to(dev, torch.bfloat16) for _ in range(4)] if cfg.is_xl else [None] * 4
        tgt, pos, neu, unc = e
        cat = lambda x: torch.cat([unc, x]).contiguous()
        pcat = (lambda x: torch.cat([pl[3], x]).contiguous()) if cfg.is_xl else (lambda x: None)
        pairs.append(PairEmbeds(cat(tgt), cat(pos), cat(neu), cat(unc), pcat(pl[0]), pcat(pl[1]), pcat(pl[2]),
                                pcat(pl[3]), guidance_scale=4.0, action="enhance"))
    from sliders_amd.parallel import StepSampler
    samp = StepSampler(4321, rank, world, len(pairs))   # k shared by all ranks, pair index and noise rank-local (tests/test_dp_gloo.py)

    def one_step(step_idx):
        k, pi = samp.next()
        noise = samp.noise((1, 4, hw, hw)).to(dev)
        tr.iteration(pairs[pi], k, noise)
        return k

    steps_per_iter_extra = 4            # 3 frozen + 1 target prediction on top of the k denoise passes
    if a.workload == "image":
        # BASELINE configs[4]: image slider = 2 VAE encodes (fp32, on the GPU) + 2 predictions WITH grad (+scale / -scale)
        # + 2 backward passes into the same gradient buffer + AdamW.  The reference also runs two no-grad predictions whose
        # results it never uses; they are not executed here and NOT counted: 2 UNet denoise steps per iteration.
        from sliders_amd.image_trainer import ImageSliderTrainer
        from sliders_amd.vae import VAE_SCALING, VaeEncoder, random_vae_state_dict
        vae_sd = random_vae_state_dict(device=dev, seed=a.seed)
        vae = VaeEncoder(vae_sd, dev, VAE_SCALING["sdxl" if cfg.is_xl else "sd1"], exact_fp32=not a.vae_split_bf16)
        tri = ImageSliderTrainer(eng, store, vae, hw, hw, batch_size=1, lr=2e-4, process_group=pg)
        gi = torch.Generator().manual_seed(99 + rank)
        imgs = [VaeEncoder.preprocess(torch.randint(0, 256, (a.res, a.res, 3), generator=gi, dtype=torch.uint8)).to(dev)
                for _ in range(4)]

        def one_step(step_idx):     # noqa: F811
            k, pi = samp.next()
            post = samp.noise((1, 4, hw, hw)).to(dev)
            noise = samp.noise((1, 4, hw, hw)).to(dev)
            tri.iteration(pairs[pi], k, imgs[step_idx % 2], imgs[2 + step_idx % 2], float(1 + step_idx % 2), post, noise)
            return -2                    # +4 below -> 2 steps
        tr = tri

    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(i)
    barrier()
    tr.time_allreduce = dist_on          # event pair around every gradient all-reduce (on the stream it is ordered on)
    t0 = time.time()
    unet_steps = 0
    for i in range(a.steps):
        unet_steps += one_step(a.warmup + i) + 4
    barrier()
    dt = time.time() - t0
    if dist_on:
        tdt = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(tdt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tdt.item())
    loss = float(tr.loss_low.item() if a.workload == "image" else tr.loss.item())
    dp_check = dp_self_check(tr, store, dev, rank, world) if dist_on else None
    value_no_dedup = None
    if a.workload == "text" and rank == 0 and world == 1 and main_line:
        # the headline counts the de-duplicated B=3 frozen pass as the 3 predictions the reference executes; the same loop
        # with the three CFG-pair passes actually run, for comparison (not the contract value)
        tr.dedup_frozen = False
        one_step(10_000)
        torch.cuda.synchronize()
        t1 = time.time()
        us2 = sum(one_step(10_001 + i) + 4 for i in range(max(2, a.steps // 2)))
        torch.cuda.synchronize()
        value_no_dedup = round(us2 / (time.time() - t1), 3)
        tr.dedup_frozen = True

    label = {"sdxl": "SDXL", "sd1": "SD-1.x", "sd2": "SD-2.x"}.get(a.model, a.model)
    res = {
        "metric": f"UNet denoise steps/sec ({label} rank-4 text slider)" if a.workload == "text" else
                  f"UNet denoise steps/sec ({label} rank-4 image slider, VAE encode on GPU)",
        "value": round(unet_steps * world / dt, 3), "unit": "steps/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": f"synthetic (seeded random-init weights with the real {label} shapes, randn text embeddings)",
        "config": {"workload": (f"{a.model} text slider rank=4 alpha=1 noxattn+c3lier, {a.res}x{a.res}, batch 1 "
                                f"(CFG pair), DDIM-50 partial denoise k~U{{1..49}} + 4 predictions + backward + AdamW")
                               if a.workload == "text" else
                               (f"{a.model} image slider rank=4 alpha=1 noxattn+c3lier, {a.res}x{a.res} image pair, batch 1 (CFG pair): "
                                f"2 fp32 VAE encodes ({'exact-fp32 MFMA' if not a.vae_split_bf16 else 'fp32 operands as bf16 hi+lo halves, 3 bf16 MFMAs per product, fp32 accumulation'}) "
                                f"+ add_noise on the GPU, 2 predictions with grad (+scale / -scale), 2 backward "
                                f"passes, AdamW; 2 UNet denoise steps per iteration (the reference's 2 unused no-grad predictions are "
                                f"not run and not counted)"),
                   "bench_step": "one training iteration", "unet_denoise_steps_timed": unet_steps * world,
                   "iterations_per_s": round(a.steps * world / dt, 4), "prompt_pairs": len(pairs),
                   "frozen_weights_gib": round(eng.weights.nbytes() / 2 ** 30, 2),
                   "weights_note": ("packed forward copies + transposed / flipped-tap copies for the backward-data products; GEGLU.proj is kept "
                                    "twice (32|32 row blocks for the training forward, 16|16 blocks for the no-grad passes: +3.6 GB for SDXL, "
                                    "SLIDERS_GEGLU16=0 drops the second copy and the 128 x 320 tile with it)"
                                    if getattr(eng.weights, "geglu16", False) else "GEGLU.proj kept once (SLIDERS_GEGLU16=0)"),
                   "parallelism": f"dp{world}", "final_loss": loss,
                   "steps_per_s_with_frozen_predictions_run_as_3_cfg_pairs": value_no_dedup},
    }
    if dp_check is not None:
        res["data_parallel"] = dp_check
    if rank == 0:
        print("[bench] timed region done: " + json.dumps({k: res[k] for k in ("value", "ms_per_step")}), file=sys.stderr, flush=True)
    if rank == 0 and not a.no_roofline:
        eng.set_lora(True, 1.0)
        # counter passes exist for the LoRA-on no-grad pass of each configuration (scripts/measure_round.sh); the image slider's
        # training-pass roofline gets none
        tag = None if a.workload != "text" else ("" if (a.model, hw) == ("sdxl", 128) else f"_{a.model}_{hw}")
        res["roofline"] = measure_roofline(eng, eng.plan(2, hw, hw, "on" if a.workload == "text" else "train"), pmc_tag=tag)
        if a.workload == "image":
            res["vae_roofline"] = measure_vae_roofline(vae, vae_sd, imgs[0], a.res)
    if rank == 0 and world == 1:
        if not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.model, hw)
            try:
                res["cpu_baseline_config0"] = cpu_baseline_config0()
            except Exception as ex:                          # a reported side figure must never cost the bench line
                res["cpu_baseline_config0"] = {"value": None, "error": repr(ex)[:200]}
    return res


if __name__ == "__main__":
    main()
