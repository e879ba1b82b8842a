#!/usr/bin/env python
"""Dev: time the attention wgrad dWa = dT^T H (N1=128, N2=512) for several M-split counts (atomic bytes = splits * 256 KiB)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
M = 128 * 2048
g = torch.Generator(device=dev); g.manual_seed(1)
H = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
dT = (torch.randn((M, 128), generator=g, device=dev) * 0.01).bfloat16()
out = torch.zeros((128, 512), device=dev)
for sp in (0, 128, 96, 64, 48, 32):
    for _ in range(3): ops.gemm_tn(dT, H, splits=sp, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gemm_tn(dT, H, splits=sp, out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"splits={sp}: median {ts[7]:.1f} us min {ts[0]:.1f}", flush=True)
