#!/usr/bin/env python
"""Dev tool: check and time the fused-encoder prototype (tools/_abl/fused_encoder_probe.hip) at the BASELINE configs[1] shape.

    python tools/fused_probe.py            # correctness vs torch (small M), then timings at M = 262144
"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import _lib  # noqa: E402

import subprocess
PROBES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_abl", "lib", "probes.so")
if not os.path.exists(PROBES):
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_abl", "build_probes.py")])
L = ctypes.CDLL(PROBES)          # lab equipment: its own library, not part of libmurcl_amd.so
f = L.murcl_debug_fused_encoder
f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
f.restype = ctypes.c_int


def run(X, W, b, out, layers, store_all=0, nslot=4, no_mfma=0, rotate=1):
    rc = f(X.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), X.shape[0], layers, store_all, nslot, no_mfma, rotate,
           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    W = (torch.randn((3, 512, 512), generator=g, device=dev) / math.sqrt(512) * 1.4).bfloat16()
    b = torch.randn((3, 512), generator=g, device=dev) * 0.1
    # ---- correctness at a small M (ragged workgroup shares: 5 tiles)
    M = 128 * 300                       # more tiles than CUs: ragged shares, every rotation
    X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    for layers in (1, 2, 3):
        for nslot, rotate in ((3, 0), (4, 1)):
            out = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
            run(X, W, b, out, layers, store_all=1, nslot=nslot, rotate=rotate)
            h = X.float()
            for l in range(layers):
                h = torch.relu(h @ W[l].float().t() + b[l]).bfloat16().float()
                err = (out[l].float() - h).abs().max().item() / h.abs().max().item()
                assert err < 2e-2, (layers, nslot, l, err)
            print(f"layers {layers} nslot {nslot} rotate {rotate}: max rel err {err:.2e}  ok")
    # ---- timing at the C2 shape
    M = 128 * 2048
    X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    out = torch.empty((3, M, 512), dtype=torch.bfloat16, device=dev)
    gflop = 2.0 * M * 512 * 512 / 1e9           # per layer; / us / 1e3 = PFLOP/s
    for layers in (1, 3):
        for store_all, nslot, no_mfma, rotate in ((0, 4, 0, 0), (0, 4, 1, 0), (0, 4, 0, 1), (0, 4, 1, 1), (1, 4, 0, 1), (1, 3, 0, 1), (1, 4, 1, 1)):
            if True:
                if True:
                    med, mn = timed(lambda: run(X, W, b, out, layers, store_all, nslot, no_mfma, rotate))
                    print(f"M {M} layers {layers} store_all {store_all} nslot {nslot} no_mfma {no_mfma} rotate {rotate}: median {med:7.1f} us  min {mn:7.1f} us"
                          f"  = {med / layers:6.1f} us per layer, {gflop * layers / med / 1e3:6.3f} PFLOP/s", flush=True)


if __name__ == "__main__":
    main()
