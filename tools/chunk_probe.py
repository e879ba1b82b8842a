#!/usr/bin/env python
"""Does walking the encoder chain in Infinity-Cache-sized row chunks beat one full-batch launch per layer?

The forward chain L1 -> L2 -> L3 -> pooling over 262144 rows writes and re-reads 268 MB tensors, larger than the 256 MiB
Infinity Cache, so every layer reads its input from HBM.  In chunks of 1/2, 1/4, 1/8 of the rows the producer's output may
still sit in the cache when the consumer starts.  Event-timed, same process, alternating."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops

torch.manual_seed(0)
dev = "cuda"
B, n, D = 128, 2048, 512
M = B * n
X = (torch.randn(M, D, device=dev) * 0.5).bfloat16()
Ws = [(torch.randn(D, D, device=dev) * 0.04).bfloat16() for _ in range(3)]
bs = [torch.zeros(D, device=dev) for _ in range(3)]
Wa = (torch.randn(128, D, device=dev) * 0.04).bfloat16()
ba = torch.zeros(128, device=dev); wb = torch.randn(128, device=dev) * 0.1; bb = torch.zeros(1, device=dev)


def chain(x, pool=True, nt=True):
    h = x
    for i in range(3):
        h, bm, _ = ops.panel_gemm(h, Ws[i], ops.PG_BIAS_RELU, bias=bs[i], want_bitmask=True, reverse=(i == 1), stream_a=(nt and i == 2))
    if pool:
        return ops.abmil_pool_fwd(h.view(-1, n, D), Wa, ba, wb, bb)
    return h


def run(chunks, pool=True, nt=True):
    rows = M // chunks
    outs = []
    for c in range(chunks):
        outs.append(chain(X[c * rows:(c + 1) * rows], pool, nt))
    return outs


def timed(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for rnd in range(2):
    for chunks in (1, 2, 4):
        for nt in (True, False):
            print(f"round {rnd} chunks={chunks:2d} nt_l3={int(nt)}: chain+pool {timed(lambda: run(chunks, True, nt)):7.1f} us   chain only {timed(lambda: run(chunks, False, nt)):7.1f} us", flush=True)
