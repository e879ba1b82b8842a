#!/usr/bin/env python
"""Dev tool: is the bench step host-bound?  Times the enqueue loop (no sync) against the synchronised total.
GCF=2 prints every full (generation-2) collection of the cyclic GC - each is a 30-40 ms host stall; GCF=1 freezes the
long-lived objects first, as bench.py does."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
step = bench.make_step(model, fc, opt, crit, views, 1)
for _ in range(5): step()
torch.cuda.synchronize()
import gc
if os.environ.get('GCF')=='1':
    gc.collect(); gc.freeze()
if os.environ.get('GCF')=='2':
    gc.callbacks.append(lambda ph, info: print('gc', ph, info) if ph=='stop' and info['generation']==2 else None)
for rep in range(6):
    t0 = time.perf_counter()
    for _ in range(30): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/30:.3f} ms/step, total {1e3*(t2-t0)/30:.3f} ms/step", flush=True)
