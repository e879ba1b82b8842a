#!/usr/bin/env python
"""Dev tool: where a pair iteration of the K2 forward spends its cycles (needs a -DK2_STAMPS build: tools/ab_build.py).

    MURCL_AMD_LIB=tools/_abl/lib/k2_stamps.so python tools/stamps_k2.py
"""
import ctypes
import math
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import _lib, ops  # noqa: E402

WG, IT, EV = 16, 16, 10


def main():
    dev = torch.device("cuda:0")
    B, N = 128, 2048
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    H = (torch.randn((B, N, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    Wa = (torch.randn((128, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
    ba = torch.randn((128,), generator=g, device=dev) * 0.1
    wb = torch.randn((1, 128), generator=g, device=dev) * 0.3
    bb = torch.zeros((1,), device=dev)
    fn = lambda: ops.abmil_pool_fwd(H, Wa, ba, wb, bb)   # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    print(f"k2 fwd (+combine): {a.elapsed_time(b) * 1e3:.1f} us (instrumented build)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros((WG, 2, IT, EV), dtype=np.uint32)
    rc = lib.murcl_debug_k2_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes))
    assert rc == 0, rc
    s = buf.astype(np.int64)
    ok = s[:, :, :, 0] != 0
    ok[:, :, 0] = False                                     # the first pair waits for the launch's first bytes
    d = lambda i, j: ((s[..., j] - s[..., i]) & 0xffffffff)[ok]      # noqa: E731
    names = [("wait for the pair's DMA (vmcnt)", 0, 1), ("barrier 1", 1, 2), ("issue next pair", 2, 3), ("score MFMAs (64)", 3, 4),
             ("tanh + quarter sums + spart", 4, 5), ("barrier 2", 5, 6), ("weights + pooling MFMAs", 6, 7), ("whole pair", 0, 7)]
    for n, i, j in names:
        v = d(i, j)
        print(f"  {n:34s} median {int(np.median(v)):6d}  mean {v.mean():8.0f}  p90 {int(np.percentile(v, 90)):6d} cycles")
    v0 = ((s[:, :, 0, 1] - s[:, :, 0, 0]) & 0xffffffff)
    print(f"  first pair's wait (launch ramp): median {int(np.median(v0))} cycles")
    n_it = ok.sum(-1).max() + 1
    cyc = ((s[:, :, n_it - 1, 0] - s[:, :, 0, 0]) & 0xffffffff).astype(np.float64)
    rt = ((s[:, :, n_it - 1, 8] - s[:, :, 0, 8]) & 0xffffffff).astype(np.float64)
    good = rt > 0
    print(f"  pairs recorded per wave {n_it}; in-kernel clock {np.median(cyc[good] / rt[good]) * 0.1:.2f} GHz; a pair takes "
          f"{np.median(cyc[good]) / (n_it - 1):.0f} cycles = {np.median(rt[good]) / (n_it - 1) * 10:.0f} ns")
    for w in range(2):
        it = s[0, w]
        print(f"  wg0 wave{2 * w}: " + " | ".join(" ".join(str(int((it[k, j] - it[k, 0]) & 0xffffffff)) for j in range(1, 8)) for k in range(2, 6)))


if __name__ == "__main__":
    main()
