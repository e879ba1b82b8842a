"""Per-shape timing of the bag-level f32 GEMMs (C = A B^T, M <= 1024 rows): the shapes of the recurrent head, the decoder and the
PPO actor-critic.  Run once per MURCL_SKINNY16 setting (the switch is read once per process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
shapes = [(128, 512, 512), (128, 3072, 512), (128, 3072, 1024), (128, 128, 1024), (128, 1024, 128), (128, 512, 3072), (128, 1024, 3072),
          (64, 2048, 512), (64, 512, 2048), (64, 1536, 512), (64, 512, 512), (320, 2048, 512), (320, 512, 2048), (320, 1536, 512), (320, 512, 1536)]
for M, N, K in shapes:
    A = torch.randn((M, K), device=dev)
    Bs = [torch.randn((N, K), device=dev) for _ in range(8)]          # rotate weights: stream from HBM / MALL like in a step
    bias = torch.randn((N,), device=dev)
    def run(i):
        ops.gemm_nt(A, Bs[i % 8], epi=ops.EPI_BIAS, bias=bias)
    for i in range(8): run(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(64): run(i)
    b.record(); torch.cuda.synchronize()
    print(f"M={M:4d} N={N:5d} K={K:5d}: {a.elapsed_time(b) / 64 * 1e3:7.1f} us per call (incl. launch gaps)", flush=True)
