#!/usr/bin/env python
"""Dev tool: the supervised step of train_RLMIL.py (`supervised_step`, ABMIL, stage 1..3) at one GPU's share of config 4 - 64 raw bags x
8192 x 512 -> T = 6 sub-bags of 1024 - a few steps, for tools/trace_seq.sh (which launches are not this library's?) and a ms/step figure.
    python tools/supervised_seq.py [stage] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd.models import rlmil  # noqa: E402
from murcl_amd.optim import FlatAdam  # noqa: E402
from murcl_amd.train_RLMIL import create_model, supervised_step  # noqa: E402
from murcl_amd.utils.datasets import BagPack  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
B, RAW, K, T_, FS = 64, 8192, 10, 6, 1024
ARCH = os.environ.get("SUP_ARCH", "ABMIL")
model, fc = create_model(ARCH, 512, 2, dev, dtype=torch.bfloat16 if ARCH != "DSMIL" else torch.float32)
model.train() if stage != 2 else model.eval()
ppo = None
if stage != 1:
    ppo = rlmil.PPO(512, 512, 512, False, action_std=0.1, lr=1e-5, gamma=0.1, K_epochs=3, action_size=K)
opt = None
if stage != 2:
    opt = FlatAdam([{"params": list(model.parameters()), "lr": 1e-4}, {"params": list(fc.parameters()), "lr": 1e-4}], betas=(0.9, 0.999), weight_decay=1e-5)
g = torch.Generator(device=dev)
g.manual_seed(1)
feats = [(torch.randn((RAW, 512), generator=g, device=dev).abs() * 0.5) for _ in range(B)]
rng = np.random.default_rng(985)
clusters = []
for _ in range(B):
    lab = rng.integers(0, K, RAW)
    clusters.append([np.nonzero(lab == k)[0].tolist() for k in range(K)])
pack = BagPack.from_lists(feats, clusters, dtype=torch.bfloat16 if ARCH != "DSMIL" else None)
labels = torch.from_numpy(rng.integers(0, 2, B)).to(dev)
mem = rlmil.Memory()


def step():
    return supervised_step(ARCH, model, fc, ppo, opt, pack, labels, mem, T=T_, feat_size=FS, train_stage=stage)[0]


for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
print(f"supervised {ARCH} stage {stage}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, loss {loss.item():.5f}")

if os.environ.get("MURCL_ATEN_WHO") == "1":
    # which Python lines launch ATen kernels?  (torch.profiler: aten ops with device time, their nearest frames inside this repo)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    seen = {}
    launching = ("aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::div", "aten::sum", "aten::mean", "aten::cat",
                 "aten::sub", "aten::zero_", "aten::repeat_interleave", "aten::index_select", "aten::uniform_", "aten::_to_copy", "aten::clone")
    for e in prof.events():
        if e.name in launching:
            frames = [f for f in (e.stack or []) if "murcl_amd" in f or "tools/" in f][:3]
            key = (e.name + " " + str(e.input_shapes)[:90], tuple(frames))
            seen[key] = seen.get(key, 0) + 1
    for (name, frames), c in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(f"{c:3d} x {name:100s} " + " <- ".join(f.split('/')[-1] for f in frames))
