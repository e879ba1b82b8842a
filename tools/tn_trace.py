"""Kernel-trace timing of the bag-level f32 weight gradients C[N1,N2] += A[M,N1]^T B[M,N2] (with the bias gradient), per shape;
parsed by tools/skinny_trace.sh-style markers (FillFunctor<double> ... FillFunctor<short>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
REPS = 16
shapes = [(128, 3072, 512), (128, 512, 512), (128, 128, 1024), (64, 3072, 1024), (320, 2048, 512), (320, 512, 2048), (320, 1536, 512), (768, 3072, 512), (768, 3072, 1024), (768, 128, 1024)]
mark0 = torch.zeros(1024, device=dev, dtype=torch.float64)
mark1 = torch.zeros(1024, device=dev, dtype=torch.int16)
for M, N1, N2 in shapes:
    A = torch.randn((M, N1), device=dev); B = torch.randn((M, N2), device=dev)
    C = torch.zeros((N1, N2), device=dev); cs = torch.zeros((N1,), device=dev)
    for i in range(3): ops.gemm_tn(A, B, out=C, colsum_into=cs)
    torch.cuda.synchronize()
    mark0.fill_(1.0)
    for i in range(REPS): ops.gemm_tn(A, B, out=C, colsum_into=cs)
    mark1.fill_(2)
    torch.cuda.synchronize()
print("SHAPES", shapes, REPS)
