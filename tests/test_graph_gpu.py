"""hipGraph replay of a UNet command buffer (slh_graph_capture / slh_graph_launch) against plain launches of the same
buffer: the graph must be taken after Program.GRAPH_AFTER eager replays, must read the CURRENT contents of the input
buffers / adapter weights / adapter scale (everything the reference's loop changes per step - train_util.py:220-260 -
is behind device pointers), and must agree with the eager replay to within the run-to-run floor of the pass
(GroupNorm statistics use fp32 atomics: ~7e-3 rel-L2 at full size, far less on the small nets used here)."""
import pytest
import torch

from sliders_amd import lib
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.unet import UNetEngine
from tests.test_unet_gpu import make_inputs
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny_sdxl", "tiny_sd1"])
def test_graph_replay_matches_plain_launches(name, monkeypatch):
    dev = torch.device("cuda:0")
    cfg = CONFIGS[name]()
    eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="full", device=dev)
    store.params.add_(torch.randn_like(store.params.float()).to(torch.bfloat16) * 0.05)
    eng.attach_lora(store)
    x, ctx, kw = make_inputs(cfg, 2, 16)
    x, ctx = x.to(dev), ctx.to(dev)
    kw = {k: v.to(dev) for k, v in kw.items()} if kw else None

    def run(xx, t, scale):
        eng.set_lora(True, scale)
        out = eng(xx, torch.tensor(t), ctx, kw, mode="on").sample.float()
        torch.cuda.synchronize()
        return out

    cases = [(500, 1.0), (20, 2.0)]
    monkeypatch.setattr(lib, "_GRAPHS_ON", False)
    run(x, 500, 1.0)                                      # first call: plan construction, lazy state
    eager = {c: run(x, *c) for c in cases}
    again = {c: run(x, *c) for c in cases}
    noise = max(rel_err(again[c], eager[c]) for c in cases)
    p = eng.plan(2, 16, 16, "on")
    assert p.prog._graphs is None and p.prog.n_ops >= lib.Program.GRAPH_MIN_OPS
    monkeypatch.setattr(lib, "_GRAPHS_ON", True)
    p.prog._runs = 0
    for c in cases:
        run(x, *c)
        assert p.prog._graphs is None, "captured too early"
    for c in cases:
        got = run(x, *c)
        assert p.prog._graphs is not None, "the third replay of an unchanged buffer must go through a graph"
        r = rel_err(got, eager[c])
        print(f"[parity] {name} graph vs plain launches t={c[0]} scale={c[1]}: rel_l2 {r:.2e} (plain vs plain {noise:.2e})")
        assert r <= max(3.0 * noise, 1e-4)
    # the graph reads live inputs and live adapter weights
    x2 = x + 0.5
    a = run(x2, 500, 1.0)
    assert rel_err(a, eager[(500, 1.0)]) > 1e-2
    store.params.mul_(0.0)
    b = run(x2, 500, 1.0)
    eng.set_lora(False)
    off = eng(x2, torch.tensor(500), ctx, kw, mode="off").sample.float()
    assert rel_err(b, off) <= max(3.0 * noise, 1e-4), "zeroed adapter through the graph must equal the adapter-free pass"
    assert rel_err(a, off) > 1e-3
