#!/usr/bin/env python
"""adam_multi_kernel with and without the live step counter (the hipGraph form), 6.0 M parameters in two runs as in the headline step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
sizes = (1_180_000 // 4096 * 4096 + 128, 4_850_000)
bufs = [[torch.randn((n,), device=dev) * (1e-3 if k else 1.0) for k in range(4)] for n in sizes]
for b in bufs:
    b[3].abs_()
rep = torch.zeros((2,), dtype=torch.int32, device=dev)


def run(live):
    jobs = [(b[0], b[1], b[2], b[3], 1e-4, 7) for b in bufs]
    ops.adam_multi(jobs, (0.9, 0.999), 1e-8, 0.0, zero_grad=False, replays=rep if live else None)


for live in (False, True, False, True):
    for _ in range(5):
        run(live)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            run(live)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 250)
    ts.sort()
    print(f"live={live}: median {ts[15]:.1f} us, min {ts[0]:.1f} us per launch ({sum(sizes) * 28 / ts[15] / 1e3:.0f} GB/s); replays counter {rep.tolist()}")
