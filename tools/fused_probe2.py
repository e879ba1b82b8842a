#!/usr/bin/env python
"""Dev tool: check and time the fused-encoder prototype v2 (tools/_abl/fused_encoder2_probe.hip) at the BASELINE configs[1] shape,
next to the three un-fused panel GEMM launches it would replace.

    python tools/fused_probe2.py            # correctness vs torch (small M), then timings at M = 262144
"""
import ctypes
import math
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
PROBES = os.path.join(HERE, "_abl", "lib", "probes.so")
if not os.path.exists(PROBES):
    subprocess.check_call([sys.executable, os.path.join(HERE, "_abl", "build_probes.py")])
L = ctypes.CDLL(PROBES)
f = L.murcl_debug_fused_encoder2
f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
f.restype = ctypes.c_int


def run(X, W, b, out, layers, store_all=0, nslot=4, abl=0):
    rc = f(X.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), X.shape[0], layers, store_all, nslot, abl, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


f3 = L.murcl_debug_fused_encoder3
f3.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
f3.restype = ctypes.c_int


def run3(X, W, b, out, layers, store_all=0, nslot=4, depth=6):
    rc = f3(X.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), X.shape[0], layers, store_all, nslot, depth, torch.cuda.current_stream().cuda_stream)
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
    M = 128 * 300                       # more tiles than CUs: ragged shares
    X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    for layers in (1, 2, 3):
        for nslot in (3, 4):
            out = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
            run(X, W, b, out, layers, store_all=1, nslot=nslot)
            h = X.float()
            worst = 0.0
            for l in range(layers):
                h = torch.relu(h @ W[l].float().t() + b[l]).bfloat16().float()
                err = (out[l].float() - h).abs().max().item() / h.abs().max().item()
                worst = max(worst, err)
            print(f"layers {layers} nslot {nslot}: max rel err {worst:.2e}  {'ok' if worst < 2e-2 else 'WRONG'}", flush=True)
            out2 = torch.zeros((1, M, 512), dtype=torch.bfloat16, device=dev)
            run(X, W, b, out2, layers, store_all=0, nslot=nslot)
            same = torch.equal(out2[0], out[layers - 1])
            print(f"   last layer only == stored-all last layer: {same}", flush=True)
    for layers in (1, 2, 3):
        for nslot, depth in ((4, 4), (4, 6), (3, 4)):
            out = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
            run3(X, W, b, out, layers, 1, nslot, depth)
            h = X.float()
            worst = 0.0
            for l in range(layers):
                h = torch.relu(h @ W[l].float().t() + b[l]).bfloat16().float()
                worst = max(worst, (out[l].float() - h).abs().max().item() / h.abs().max().item())
            o2 = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
            run3(X, W, b, o2, layers, 1, nslot, depth)
            print(f"v3 layers {layers} nslot {nslot} depth {depth}: max rel err {worst:.2e}  {'ok' if worst < 2e-2 else 'WRONG'}  repeatable {torch.equal(out, o2)}", flush=True)
    # repeatability (hazard bugs show as run-to-run differences)
    outs = []
    for _ in range(3):
        o = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
        run(X, W, b, o, 3, store_all=1, nslot=4)
        outs.append(o)
    print("repeatable:", all(torch.equal(outs[0], o) for o in outs[1:]), flush=True)
    M = 128 * 2048
    X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    out = torch.empty((3, M, 512), dtype=torch.bfloat16, device=dev)
    gflop = 2.0 * M * 512 * 512 / 1e9
    for layers in (1, 2, 3):
        for store_all, nslot in ((0, 4), (1, 4), (1, 3)):
            med, mn = timed(lambda: run(X, W, b, out, layers, store_all, nslot))
            print(f"M {M} layers {layers} store_all {store_all} nslot {nslot}: median {med:7.1f} us  min {mn:7.1f} us"
                  f"  = {med / layers:6.1f} us per layer, {gflop * layers / med / 1e3:6.3f} PFLOP/s", flush=True)
    for layers in (1, 2, 3):
        for store_all, nslot, depth in ((0, 4, 6), (1, 4, 6), (1, 4, 4), (1, 3, 4)):
            med, mn = timed(lambda: run3(X, W, b, out, layers, store_all, nslot, depth))
            print(f"v3 (8 waves x 16 rows) M {M} layers {layers} store_all {store_all} nslot {nslot} depth {depth}: median {med:7.1f} us  min {mn:7.1f} us"
                  f"  = {med / layers:6.1f} us per layer", flush=True)
    for abl, what in ((1, "no weight DMA"), (2, "no fragment reads"), (3, "no DMA, no reads"), (4, "no barrier / wait"), (7, "MFMA + epilogue + stores only"),
                      (8, "no stores"), (15, "MFMA + epilogue only")):
        for layers in (1, 3):
            med, mn = timed(lambda: run(X, W, b, out, layers, 1, 4, abl))
            print(f"ablation {abl:2d} ({what}): layers {layers}: median {med:7.1f} us  min {mn:7.1f} us", flush=True)
    from murcl_amd import ops
    bs = [b[i].contiguous() for i in range(3)]
    Ws = [W[i].contiguous() for i in range(3)]

    def unfused():
        h = X
        for i in range(3):
            h, _, _ = ops.panel_gemm(h, Ws[i], ops.PG_BIAS_RELU, bias=bs[i], want_bitmask=True)
        return h
    med, mn = timed(unfused)
    print(f"un-fused: 3 x panel_gemm (bias + ReLU + bit mask): median {med:7.1f} us  min {mn:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
