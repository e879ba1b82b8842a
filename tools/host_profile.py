#!/usr/bin/env python
"""Dev tool: cProfile of the host side of the bench step (enqueue only; the GPU runs behind)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
step = bench.make_step(model, fc, opt, crit, views, 1)
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
