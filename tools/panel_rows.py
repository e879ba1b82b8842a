#!/usr/bin/env python
"""Weight-stationary encoder forward (panel_nt_kernel, bias + ReLU, no mask) and the K2 partial pass as a function of the row count:
what a launch costs before its first row (stage 2 / 3 run them at 65,536 rows per patch step, the headline at 262,144).  Dev tool.
    python tools/panel_rows.py [--reps 40]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops  # noqa: E402


def timed(fn, reps, back_to_back=8):
    """median over reps of (HIP-event time of ``back_to_back`` launches) / back_to_back: launch gaps hidden, caches as in a chain"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(back_to_back):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / back_to_back)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    W = (torch.randn((512, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
    bias = torch.randn((512,), generator=g, device=dev) * 0.1
    Wa = (torch.randn((128, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
    ba, wb, bb = torch.randn((128,), generator=g, device=dev) * 0.1, torch.randn((1, 128), generator=g, device=dev) * 0.3, torch.zeros((1,), device=dev)
    print(f"{'rows':>8s} {'fwd us':>8s} {'GB/s':>7s} {'fwd+mask':>9s} {'k2 part':>8s} {'GB/s':>7s} {'copy us':>8s}")
    for rows in (8192, 16384, 32768, 65536, 131072, 262144):
        X = (torch.randn((rows, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
        H = torch.empty_like(X)
        bags = 128
        t_f = timed(lambda: ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, out=H), a.reps)
        t_m = timed(lambda: ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True, out=H), a.reps)
        t_k = timed(lambda: ops.abmil_pool_partials(H.view(bags, rows // bags, 512), Wa, ba, wb, bb), a.reps)
        t_c = timed(lambda: H.copy_(X), a.reps)
        print(f"{rows:8d} {t_f:8.1f} {2 * rows * 1024 / t_f / 1e3:7.0f} {t_m:9.1f} {t_k:8.1f} {rows * 1024 / t_k / 1e3:7.0f} {t_c:8.1f}", flush=True)


if __name__ == "__main__":
    main()
