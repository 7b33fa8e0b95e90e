"""CPU dry run of the planner: launches per op kind / name family for one pass (development aid, no GPU)."""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from sliders_amd import lib
from sliders_amd.arena import Arena
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.planner import BackwardPlan, UNetPlan
from test_host import _FakeWeights

model, hw, B, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cfg = CONFIGS[model]()
store = LoraStore(cfg, train_method="noxattn", init="none")
store.temb_tcol = torch.zeros(1, dtype=torch.int32)
p = UNetPlan(cfg, _FakeWeights(cfg), Arena(1 << 50, None), Arena(1 << 40, None), B, hw, hw, 77,
             store if mode != "off" else None, mode, 0x10)
names = {v: k for k, v in vars(lib).items() if k.startswith("OP_")}


def census(prog, title):
    c = collections.Counter()
    for (op, d), nm in zip(prog.ops, prog.op_names):
        fam = re.sub(r"\d+", "#", nm.split(".")[-1] if "." in nm else nm)
        c[(names.get(op, op), fam)] += 1
    print(f"== {title}: {len(prog.ops)} ops")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print(f"  {v:5d}  {k[0]:24s} {k[1]}")


census(p.prog, f"{model} {hw} B={B} {mode}")
if mode == "train":
    bw = BackwardPlan(p, B // 2 if B > 1 else 0, max(1, B // 2), 0x20)
    census(bw.prog, "backward")
