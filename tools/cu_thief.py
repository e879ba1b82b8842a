#!/usr/bin/env python
"""Multi-GPU readiness on ONE GPU: the headline step with N "stolen" CUs.

RCCL runs one workgroup per channel for the length of a collective; the encoder-sized kernels of the step are persistent, one
workgroup per CU with 128-160 KiB of LDS and a STATIC share of the work, so a CU that a channel occupies cannot take its share and
that share runs as a second round.  Here a co-running kernel on a second stream - N workgroups x 64 KiB of LDS,
spinning for the length of the measurement (tools/_abl/cu_thief_probe.hip) - stands in for the channels, and the step is timed
with HIP events on its own stream: step time against N, per kernel where asked.

    python tools/cu_thief.py [--n 0,4,8,16,32] [--steps 20] [--lds 65536] [--threads 64] [--busy] [--arrive resident|mid]

Round 6: ``--threads`` / ``--busy`` / ``--lds 0`` give the thieves other shapes (a workgroup without LDS can sit BESIDE a persistent
tenant if registers allow; a busy one takes issue slots instead of a workgroup slot), ``--arrive mid`` makes them arrive in the middle
of every step (``--delay`` us after its start, for ``--dur`` us: a collective launched from a backward hook) instead of being resident
before the window starts.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PROBES = os.path.join(ROOT, "tools", "_abl", "lib", "probes.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", default="0,4,8,16,32")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lds", type=int, default=64 * 1024)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--busy", action="store_true", help="thieves run a dependent FMA chain instead of sleeping")
    ap.add_argument("--arrive", default="resident", choices=["resident", "mid"])
    ap.add_argument("--delay", type=float, default=480.0, help="--arrive mid: microseconds after the step's start (480: the backward pass has begun)")
    ap.add_argument("--dur", type=float, default=300.0, help="--arrive mid: how long the thieves stay, microseconds")
    ap.add_argument("--budget", default="256", help="CU budgets of the persistent kernels to try (ops.set_cu_budget), e.g. 256,248,240")
    ap.add_argument("--overlap-budget", type=int, default=0,
                    help="size only the aggregator's backward launches (pooling backward .. last input gradient) for this many CUs "
                         "(functional.set_overlap_cu_budget: what a data-parallel run does by default) - the tax of the scoped reserve")
    ap.add_argument("--kernels", action="store_true", help="per-kernel HIP-event times (adds ~2-3 us per launch)")
    a = ap.parse_args()
    if not os.path.exists(PROBES):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "_abl", "build_probes.py")])
    L = ctypes.CDLL(PROBES)
    thief2 = L.murcl_debug_cu_thief2
    thief2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]

    def thief(n, lds, us, out, stream, delay=0.0):
        return thief2(n, lds, a.threads, int(a.busy), delay, us, out, stream)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from murcl_amd import ops
    model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
    views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
    step = bench.make_step(model, fc, opt, crit, views, 1)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    sink = torch.zeros(256, dtype=torch.int32, device=dev)
    if a.overlap_budget:
        from murcl_amd import functional
        functional.set_overlap_cu_budget(a.overlap_budget)
    for budget in [int(v) for v in a.budget.split(",")]:
        got = ops.set_cu_budget(budget)
        print(f"== CU budget {got}" + (f", backward launches {a.overlap_budget}" if a.overlap_budget else ""), flush=True)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        measure(a, thief, step, side, sink, ops)


def measure(a, thief, step, side, sink, ops):
    rows = []
    for n in [int(v) for v in a.n.split(",")]:
        per_kernel = None
        for rep in range(3):                                     # three windows per N, the median window counts
            budget_us = (a.steps + 6) * 1700.0
            mid = a.arrive == "mid" and n > 0
            if n and not mid:
                rc = thief(n, a.lds, budget_us, sink.data_ptr(), side.cuda_stream)
                assert rc == 0, rc
            start = torch.cuda.Event()

            def one_step():
                if mid:                                          # the side stream waits for this step's start, sleeps --delay, then the thieves run --dur
                    start.record()
                    side.wait_event(start)
                    assert thief(n, a.lds, a.dur, sink.data_ptr(), side.cuda_stream, a.delay) == 0
                step()
                if mid:
                    torch.cuda.current_stream().wait_stream(side)      # the next step starts after the thieves have left (one arrival per step)
            for _ in range(3):                                   # the thieves are resident by now; settle
                one_step()
            if a.kernels and rep == 2:
                ops.TIMERS = ops.KernelTimers()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
            evs[0].record()
            for i in range(a.steps):
                one_step()
                evs[i + 1].record()
            torch.cuda.synchronize()
            if a.kernels and rep == 2:
                per_kernel = {k: round(v["ms_avg"] * 1e3, 1) for k, v in ops.TIMERS.summary().items() if v["ms_avg"] > 0.03}
                ops.TIMERS = None
            per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
            rows.append((n, rep, per[len(per) // 2], per[0], per[-1]))
        med = sorted(r[2] for r in rows if r[0] == n)[1]
        print(json.dumps({"stolen_workgroups": n, "lds_bytes_each": a.lds, "threads_each": a.threads, "busy": a.busy, "arrive": a.arrive, "step_ms_median": round(med, 4),
                          "windows": [round(r[2], 4) for r in rows if r[0] == n], "kernel_us": per_kernel}), flush=True)
    base = sorted(r[2] for r in rows if r[0] == 0)
    if base:
        b = base[len(base) // 2]
        for n in sorted({r[0] for r in rows if r[0]}):
            m = sorted(r[2] for r in rows if r[0] == n)[1]
            print(f"N = {n:3d}: {m:.4f} ms = +{(m - b) * 1e3:6.1f} us = +{(m / b - 1) * 100:5.1f} %  ({(m - b) * 1e3 / n:5.2f} us per stolen workgroup)")


if __name__ == "__main__":
    main()
