#!/usr/bin/env python
"""Dev tool: the exact-f32 GEMMs of the sampler / PPO sequences (csrc/ppo_seq.hip) timed one by one, back to back (median of 50)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(128, 2048, 512, "act enc1"), (128, 512, 2048, "act enc2"), (320, 2048, 512, "epoch enc1"), (320, 512, 2048, "epoch enc2"),
          (320, 1536, 512, "epoch gi"), (320, 512, 1536, "epoch d(e2)"), (320, 2048, 512, "epoch d(e1)"), (128, 128, 512, "head proj"),
          (768, 1024, 512, "head fc"), (64, 512, 2048, "act enc2 B=32")]
for M, N, K, name in shapes:
    a = torch.randn((M, K), device=dev)
    w = torch.randn((N, K), device=dev) / K ** 0.5
    b = torch.randn((N,), device=dev)
    for _ in range(5):
        ops.gemm_nt(a, w, epi=ops.EPI_BIAS_RELU, bias=b)
    ts = []
    for _ in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_nt(a, w, epi=ops.EPI_BIAS_RELU, bias=b)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    ts.sort()
    print(f"{name:16s} [{M:4d} x {N:4d} x {K:4d}]  {ts[len(ts) // 2]:6.1f} us   {2 * M * N * K / ts[len(ts) // 2] / 1e6:7.1f} TFLOP/s")
