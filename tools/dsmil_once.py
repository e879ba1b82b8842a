"""DSMIL forward + backward (batched internals) a few times; the last iteration is what tools/trace_seq.sh prints."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd.models.dsmil import build_dsmil
dev = torch.device("cuda:0")
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
B, N, d = 16, 8192, 1024
m = build_dsmil(d, 2).to(dev)
x = torch.randn((B, N, d), device=dev).abs() * 0.5
if bf16:
    m.compute_dtype = torch.bfloat16
    x = x.bfloat16()
for _ in range(4):
    for p in m.parameters():
        p.grad = None
    classes, bag = m._run(x)
    (bag.sum() + classes.max(1)[0].sum()).backward()
torch.cuda.synchronize()
