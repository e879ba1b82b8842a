#!/usr/bin/env python
"""Dev probe: can a kernel inside a captured hipGraph be timed with HIP events (torch.cuda.Event(external=True) -> event-record nodes)?
If yes, per-kernel times could be taken in the GPU-paced region instead of the eager one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn((64 * 1024 * 1024,), device=dev)
y = torch.empty_like(x)
try:
    e0, e1 = torch.cuda.Event(enable_timing=True, external=True), torch.cuda.Event(enable_timing=True, external=True)
except TypeError as e:
    print("external events not supported by this torch:", e)
    sys.exit(0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ops.copy_flat(y, x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        e0.record()
        ops.copy_flat(y, x)
        e1.record()
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        print(f"replay {i}: {e0.elapsed_time(e1) * 1e3:.1f} us for a 268 MB -> 268 MB copy ({2 * x.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12:.2f} TB/s)")
except Exception as e:                              # noqa: BLE001
    print("timing events inside a captured graph failed:", type(e).__name__, str(e)[:300])
