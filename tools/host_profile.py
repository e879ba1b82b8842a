#!/usr/bin/env python
"""Dev tool: where does the HOST spend its ~0.85 ms per headline step (25 launches)?  cProfile over N eager steps, top functions by own time."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
step = bench.make_step(model, fc, opt, crit, views, 1)
for _ in range(20):
    step()
torch.cuda.synchronize()
N = 100
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
tot = sum(v[2] for v in st.stats.values())
print(f"profiled host time per step: {tot / N * 1e3:.3f} ms (cProfile inflates it ~1.5-2x)")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:45]
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print(f"{tt / N * 1e6:8.1f} us own  {ct / N * 1e6:8.1f} us cum  {nc / N:6.1f} calls/step  {os.path.basename(fn)}:{line} {name}")
