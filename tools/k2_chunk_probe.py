"""K2 forward / backward time per (bags, rows) shape for the current MURCL_K2_CHUNK (HIP events, median; a 512 MB copy between calls
evicts the Infinity Cache so the input arrives from HBM).  Dev tool."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
B, N = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device=dev); g.manual_seed(1)
H = (torch.randn((B, N, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
Wa = (torch.randn((128, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
ba = torch.randn((128,), generator=g, device=dev) * 0.1
wb = torch.randn((1, 128), generator=g, device=dev) * 0.3
bb = torch.zeros((1,), device=dev)
dM = torch.randn((B, 512), generator=g, device=dev)
big = torch.empty((2, 64 << 20), dtype=torch.float32, device=dev)
sc, Aw, Mp, ml = ops.abmil_pool_fwd(H, Wa, ba, wb, bb)


def t(fn, reps=15):
    ts = []
    for i in range(reps + 3):
        big[0].copy_(big[1])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


f = t(lambda: ops.abmil_pool_fwd(H, Wa, ba, wb, bb))
b = t(lambda: ops.abmil_pool_bwd(H, Wa, ba, wb, sc, ml, Mp, dM))
nb = B * N * 512 * 2
print(f"{ops.pool_chunks(B, N, 1)} fwd {f:7.1f} us {nb / f / 1e6:6.2f} TB/s   bwd {b:7.1f} us {nb * 1.25 / b / 1e6:6.2f} TB/s")
