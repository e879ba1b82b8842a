#!/usr/bin/env python
"""CLAM_SB aggregator forward+backward at the C3 shape (64 bags x 4096 x 512 bf16), a few passes and nothing after them:
for tools/trace_seq.sh (kernel sequence of the last pass)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd.models.clam import CLAM_SB

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
B, N = 64, 4096
m = CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512).to(dev)
m.compute_dtype = torch.bfloat16
x = (torch.randn((B, N, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
if len(sys.argv) > 1 and sys.argv[1] == "train":
    m.train()
else:
    m.eval()
fwd_only = len(sys.argv) > 1 and sys.argv[1] == "fwd"
inst = len(sys.argv) > 1 and sys.argv[1] == "inst"          # with the instance-level loss (clam.py:103-132), batched labels
labels = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(1)).to(dev)
_ones = {}
def ones(t):
    k = (tuple(t.shape), t.dtype)
    if k not in _ones: _ones[k] = torch.ones_like(t)
    return _ones[k]
for _ in range(6):
    if inst:
        for p in m.parameters(): p.grad = None
        M, _, _, il, _, _ = m._run(x, labels, True)
        torch.autograd.backward((M, il), (ones(M), ones(il)))      # the upstream gradients of a sum loss, without the harness's launches
        continue
    if fwd_only:
        with torch.no_grad(): m(x)
        continue
    for p in m.parameters(): p.grad = None
    out = m(x)
    torch.autograd.backward((out[0],), (ones(out[0]),))
torch.cuda.synchronize()
