#!/usr/bin/env python
"""Dev tool: where a pair iteration of the fused pooling backward (attn_pool_bwd_dwa.hip) spends its cycles
- the kernel is parked under tools/_abl since round 6; needs its library built with stamps:

    python tools/_abl/build_kd.py -DKD_STAMPS && python tools/stamps_kd.py
"""
import ctypes
import math
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_abl"))
from murcl_amd import _lib, ops  # noqa: E402,F401
import kd  # noqa: E402

WG, IT, EV = 16, 40, 12


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
    dM = torch.randn((B, 512), generator=g, device=dev)
    sc, A, Mp, ml = ops.abmil_pool_fwd(H, Wa, ba, wb, bb)
    fn = lambda: kd.pool_bwd_dwa(H, Wa, ba, wb, sc, ml, Mp, dM, dwa="new")   # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    print(f"fused pooling backward (+reduce): {a.elapsed_time(b) * 1e3:.1f} us (instrumented build)")
    lib = ctypes.CDLL(kd.LIB)
    buf = np.zeros((WG, 2, IT, EV), dtype=np.uint32)
    rc = lib.murcl_debug_kd_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes))
    assert rc == 0, rc
    s = buf.astype(np.int64)
    ok = s[:, :, :, 10] != 0                                # iterations that ran their end-of-iteration part
    ok[:, :, 0] = False
    d = lambda i, j: ((s[..., j] - s[..., i]) & 0xffffffff)[ok]      # noqa: E731
    names = [("wait for pair p+1 + top barrier", 0, 1), ("LDS-DMA burst (9 pieces)", 1, 2), ("g dot of pair p+1 + tile 0: 32 MFMAs", 2, 3),
             ("tile 1: 32 MFMAs (+ tile 0's tanh / dT)", 3, 4), ("tile 1: tanh / dT", 4, 5), ("dT readback + store", 5, 6),
             ("dWa: 64 MFMAs", 8, 9), ("whole pair", 0, 10)]
    for n, i, j in names:
        v = d(i, j)
        print(f"  {n:36s} median {int(np.median(v)):6d}  mean {v.mean():8.0f}  p90 {int(np.percentile(v, 90)):6d} cycles")
    kr = s[:, 0, IT - 1, :4]
    ph = lambda i, j: np.median((kr[:, j] - kr[:, i]) & 0xffffffff)      # noqa: E731
    print(f"  kernel phases (cycles, wave 0): prologue {ph(0, 1):.0f}, loop {ph(1, 2):.0f}, publish {ph(2, 3):.0f}")
    n_it = int(ok.sum(-1).max()) + 1
    cyc = ((s[:, :, n_it - 1, 0] - s[:, :, 0, 0]) & 0xffffffff).astype(np.float64)
    rt = ((s[:, :, n_it - 1, 11] - s[:, :, 0, 11]) & 0xffffffff).astype(np.float64)
    good = rt > 0
    print(f"  pairs recorded per wave {n_it}; in-kernel clock {np.median(cyc[good] / rt[good]) * 0.1:.2f} GHz; a pair takes "
          f"{np.median(cyc[good]) / (n_it - 1):.0f} cycles = {np.median(rt[good]) / (n_it - 1) * 10:.0f} ns")
    for w in range(2):
        it = s[0, w]
        print(f"  wg0 wave{2 * w}: " + " | ".join(" ".join(str(int((it[k, j] - it[k, 0]) & 0xffffffff)) for j in range(1, 11)) for k in range(2, 5)))


if __name__ == "__main__":
    main()
