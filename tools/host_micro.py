#!/usr/bin/env python
"""Dev tool: host-side cost per call of the thin wrappers (tiny tensors, GPU work negligible)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops, _lib
from murcl_amd.functional import LinearFn
dev = torch.device("cuda:0")
A = torch.randn(64, 128, device=dev); W = torch.randn(64, 128, device=dev, requires_grad=True); b = torch.randn(64, device=dev, requires_grad=True)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
print(f"torch.empty            {t(lambda: torch.empty((64, 64), device=dev)):6.2f} us")
print(f"_lib.stream()          {t(_lib.stream):6.2f} us")
print(f"torch add (eager op)   {t(lambda: A + A):6.2f} us")
print(f"ops.relu_bwd           {t(lambda: ops.relu_bwd(A, A)):6.2f} us")
print(f"ops.gemm_nt (skinny)   {t(lambda: ops.gemm_nt(A, W.detach(), epi=ops.EPI_BIAS, bias=b.detach())):6.2f} us")
print(f"ops.gemm_tn            {t(lambda: ops.gemm_tn(A, A)):6.2f} us")
print(f"LinearFn.apply (fwd)   {t(lambda: LinearFn.apply(A, W, b, True)):6.2f} us")
def fb():
    y = LinearFn.apply(A, W, b, True); y.sum().backward()
print(f"LinearFn fwd+bwd       {t(fb, 500):6.2f} us")
