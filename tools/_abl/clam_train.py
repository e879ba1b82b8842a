import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from murcl_amd.models.clam import CLAM_SB
dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(3)
B, N = 64, 4096
m = CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512).to(dev)
m.compute_dtype = torch.bfloat16
x = (torch.randn((B, N, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
labels = [int(v) for v in torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(1))]
def fb():
    for p in m.parameters(): p.grad = None
    M, A, s, il, ids, io = m._run(x, labels, True)
    (M.sum() + il.sum()).backward()
for mode in ("eval", "train"):
    getattr(m, mode)()
    for _ in range(5): fb()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): fb()
    torch.cuda.synchronize(); print(mode, "fwd+bwd incl. instance loss (batched internals):", round((time.perf_counter() - t) / 20 * 1e3, 3), "ms")
