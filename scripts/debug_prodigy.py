"""Development aid: five training iterations of the tiny SDXL with per-program control over hipGraph replay
(DBG_NOGRAPH=on,onc,off,train,bw  DBG_AFTER=<eager runs before capture>  DBG_ALLOC=1: a torch temporary + D2H between iterations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from oracle.unet_oracle import build_unet
from sliders_amd import lib
from sliders_amd.trainer import SliderTrainer
from sliders_amd.unet import UNetEngine
from tests.test_trainer_gpu import _pair, _setup
dev = torch.device("cuda:0")
if os.environ.get("DBG_AFTER"):
    lib.Program.GRAPH_AFTER = int(os.environ["DBG_AFTER"])
if os.environ.get("DBG_SYNC"):
    _run = lib.Program.run
    mode = os.environ["DBG_SYNC"]
    def run(self, stream, graph=True):
        if mode in ("before", "both"):
            torch.cuda.synchronize()
        _run(self, stream, graph)
        if mode in ("after", "both"):
            torch.cuda.synchronize()
    lib.Program.run = run
side = torch.cuda.Stream() if os.environ.get("DBG_STREAM") else None
if side is not None:
    torch.cuda.set_stream(side)
nog = set(v for v in os.environ.get("DBG_NOGRAPH", "").split(",") if v)
for opt in os.environ.get("DBG_OPTS", "adamw,prodigy").split(","):
    cfg, store, emb, pool, noise = _setup(dev, "tiny_sdxl")
    eng = UNetEngine(cfg, build_unet("tiny_sdxl", seed=0).state_dict(), dev)
    kw = dict(lr=1.0, optimizer="prodigy", weight_decay=0.0) if opt == "prodigy" else dict(lr=2e-4)
    tr = SliderTrainer(eng, store, 16, 16, **kw)
    pair = _pair(emb, pool, dev)
    for it in range(5):
        loss = tr.iteration(pair, int(os.environ.get("DBG_K", 2 + it)), noise.to(dev)).item()
        torch.cuda.synchronize()
        if it == 0 and nog:
            B = 2 * tr.bs
            progs = {"on": eng.plan(B, tr.H, tr.W, "on").prog, "onc": eng.plan(B, tr.H, tr.W, "on").prog_text_cached,
                     "off": eng.plan(3 * tr.bs if tr.dedup_frozen else B, tr.H, tr.W, "off").prog,
                     "train": eng.plan(B, tr.H, tr.W, "train").prog, "bw": eng.plan(B, tr.H, tr.W, "train").backward.prog}
            for kname in nog:
                if progs[kname] is not None:
                    progs[kname].GRAPH_MIN_OPS = 10 ** 9
        if os.environ.get("DBG_SLEEP"):
            time.sleep(float(os.environ["DBG_SLEEP"]))
        if os.environ.get("DBG_ALLOC"):
            g = (store.grads * tr.grad_scale).to(torch.bfloat16).cpu()
        n = lambda t: f"{t.float().norm().item():.5f}"
        print(opt, it, f"loss {loss:.6f}", "|g|", n(store.grads), "den", n(tr.denoised), "pos", n(tr.e_pos), "neu", n(tr.e_neu),
              "unc", n(tr.e_unc), "tgt", n(tr.e_tgt), "|p|", n(store.params), flush=True)
