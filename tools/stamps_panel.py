#!/usr/bin/env python
"""Dev tool: where a tile iteration of the panel GEMM spends its cycles (needs a -DPG_STAMPS build: tools/ab_build.py).

    MURCL_AMD_LIB=tools/_abl/lib/pg_stamps.so python tools/stamps_panel.py [fwd|mask|rank1]
"""
import ctypes
import math
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import _lib, ops  # noqa: E402

WG, IT, EV = 16, 16, 8


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    dev = torch.device("cuda:0")
    B, N = 128, 2048
    M = B * N
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    W = (torch.randn((512, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
    bias = torch.randn((512,), generator=g, device=dev) * 0.1
    H, bm, _ = ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True)
    dT = (torch.randn((M, 128), generator=g, device=dev) * 0.01).bfloat16()
    WaT = (torch.randn((512, 128), generator=g, device=dev) / math.sqrt(512)).bfloat16()
    Asc = torch.rand((M,), generator=g, device=dev)
    dM = torch.randn((B, 512), generator=g, device=dev)
    fn = {"fwd": lambda: ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True),
          "mask": lambda: ops.panel_gemm(X, W, ops.PG_MASK, bitmask=bm, colsum=True),
          "rank1": lambda: ops.panel_gemm(dT, WaT, ops.PG_RANK1_MASK, bitmask=bm, rowscale=Asc, rank1=dM, rows_per_bag=N, colsum=True)}[which]
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{which}: {a.elapsed_time(b) * 1e3:.1f} us (instrumented build)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros((WG, 2, IT, EV), dtype=np.uint32)
    rc = lib.murcl_debug_pg_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes))
    assert rc == 0, rc
    s = buf.astype(np.int64)
    ok = s[:, :, :, 0] != 0
    d = lambda i, j: ((s[..., j] - s[..., i]) & 0xffffffff)[ok]      # noqa: E731
    names = [("wait for the tile's DMA (vmcnt)", 0, 1), ("barrier", 1, 2), ("issue next tile", 2, 3), ("MFMA phase (+fused epilogue)", 3, 4),
             ("rest of the iteration", 4, 5), ("whole iteration", 0, 5)]
    for n, i, j in names:
        v = d(i, j)
        print(f"  {n:34s} median {int(np.median(v)):6d}  mean {v.mean():8.0f}  p90 {int(np.percentile(v, 90)):6d} cycles")
    # shader clock: cycles per 100 MHz tick between the first and the last recorded iteration of a wave
    cyc = ((s[:, :, -1, 0] - s[:, :, 0, 0]) & 0xffffffff).astype(np.float64)
    rt = ((s[:, :, -1, 6] - s[:, :, 0, 6]) & 0xffffffff).astype(np.float64)
    good = rt > 0
    print(f"  in-kernel clock {np.median(cyc[good] / rt[good]) * 0.1:.2f} GHz; 4 iterations take {np.median(cyc[good]) / (IT - 1):.0f} cycles "
          f"= {np.median(rt[good]) / (IT - 1) * 10:.0f} ns")
    # per-wave view of one workgroup
    for w in range(2):
        it = s[0, w]
        print(f"  wg0 wave{4 * w}: " + " | ".join(" ".join(str(int((it[k, j] - it[k, 0]) & 0xffffffff)) for j in range(1, 6)) for k in range(2, 6)))


if __name__ == "__main__":
    main()
