"""Kernel-trace companion of skinny_shapes.py: every shape runs REPS calls between two marker launches (a torch fill); the
parser (tools/skinny_trace.sh) sums the GPU durations between markers, so the figures are kernel time without launch gaps
(memset / ReLU passes of the split form included)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
REPS = 16
shapes = [(128, 512, 512), (128, 3072, 512), (64, 3072, 512), (64, 3072, 1024), (128, 128, 1024), (64, 128, 1024), (128, 1024, 128), (128, 512, 3072), (128, 1024, 3072),
          (64, 2048, 512), (64, 512, 2048), (64, 1536, 512), (64, 512, 512), (320, 2048, 512), (320, 512, 2048), (320, 1536, 512), (320, 512, 1536),
          (32, 1024, 1024), (128, 512, 1024), (64, 512, 1024), (128, 256, 768), (128, 1024, 1024)]
mark0 = torch.zeros(1024, device=dev, dtype=torch.float64)      # start marker: FillFunctor<double>
mark1 = torch.zeros(1024, device=dev, dtype=torch.int16)        # end marker: FillFunctor<short>
COLD = os.environ.get("SK_COLD", "1") == "1"      # evict the Infinity Cache between calls: weights arrive from HBM as in a step
big = torch.empty((2, 96 << 20), dtype=torch.float32, device=dev) if COLD else None
for M, N, K in shapes:
    A = torch.randn((M, K), device=dev)
    Bs = [torch.randn((N, K), device=dev) for _ in range(8)]
    bias = torch.randn((N,), device=dev)
    for i in range(4): ops.gemm_nt(A, Bs[i % 8], epi=ops.EPI_BIAS, bias=bias)
    torch.cuda.synchronize()
    mark0.fill_(1.0)
    for i in range(REPS):
        if COLD: big[0].copy_(big[1])
        ops.gemm_nt(A, Bs[i % 8], epi=ops.EPI_BIAS, bias=bias)
    mark1.fill_(2)
    torch.cuda.synchronize()
print("SHAPES", shapes, REPS)
