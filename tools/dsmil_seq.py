#!/usr/bin/env python
"""DSMIL forward+backward at one GPU's C5 share (16 bags x 8192 x 1024 f32), a few passes and nothing after them: for
tools/trace_seq.sh (kernel sequence of the last pass)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd.models.dsmil import build_dsmil

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
B, N, d = 16, 8192, 1024
m = build_dsmil(d, 2).to(dev)
x = torch.randn((B, N, d), generator=g, device=dev).abs() * 0.5
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    m.compute_dtype = torch.bfloat16
    x = x.bfloat16()
ones = None
for _ in range(6):
    for p in m.parameters(): p.grad = None
    classes, bag, cmax = m._run(x, want_max=True)
    ones = ones or (torch.ones_like(bag), torch.ones_like(cmax))
    torch.autograd.backward((bag, cmax), ones)            # the upstream gradients of a sum loss, without the harness's own launches
torch.cuda.synchronize()
