"""Dev: sweep the M-split count of gemm_tn for the shapes that use the 128 x 128 ring kernel."""
import os, sys, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
def t(fn, reps=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2] * 1e3
for name, M, N1, N2, dt_ in (("dWa bf16", 262144, 128, 512, torch.bfloat16), ("DSMIL dWq bf16", 131072, 128, 1024, torch.bfloat16),
                            ("DSMIL dWq f32", 131072, 128, 1024, torch.float32), ("CLAM inst f32", 1024, 512, 512, torch.float32),
                            ("head deferred f32", 768, 3072, 512, torch.float32), ("head deferred f32 hh", 768, 3072, 1024, torch.float32)):
    A = (torch.randn((M, N1), generator=g, device=dev) * 0.1).to(dt_)
    B = (torch.randn((M, N2), generator=g, device=dev) * 0.1).to(dt_)
    out = torch.zeros((N1, N2), device=dev)
    tiles = ((N1 + 127) // 128) * ((N2 + 127) // 128)
    res = []
    for sp in (0, 1, 2, 4, 8, 16, 32, 48, 64, 96, 128):
        if sp * tiles > 2048: continue
        if sp and M // sp < 64: continue
        res.append("%d:%.0f" % (sp, t(lambda: ops.gemm_tn(A, B, splits=sp, out=out))))
    print(name, "tiles", tiles, " ".join(res), flush=True)
