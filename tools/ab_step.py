#!/usr/bin/env python
"""Dev tool: A/B of a functional.py / ops.py switch on the headline step in ONE process, interleaved rounds (cdna guide rule 24: never
rank builds by timings from different boxes or processes).

    python tools/ab_step.py functional._FRAG_WEIGHTS [--rounds 6] [--steps 40]

Every round runs `steps` steps with the switch off, then on (HIP events per step on the launch stream); prints per-arm medians of
the per-step times over all rounds and the per-round medians."""
import argparse
import importlib
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("switch", help="module.attribute under murcl_amd, e.g. functional._FRAG_WEIGHTS")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    modname, attr = a.switch.rsplit(".", 1)
    mod = importlib.import_module("murcl_amd." + modname)
    dev = torch.device("cuda:0")
    model, fc, opt, crit = bench.build(torch.bfloat16, dev, 64)
    views = bench.synth_views(64, 2048, 512, torch.bfloat16, dev, 0)
    step = bench.make_step(model, fc, opt, crit, views, 1)
    times = {False: [], True: []}
    per_round = {False: [], True: []}
    for r in range(a.rounds + 1):
        for arm in (False, True):
            setattr(mod, attr, arm)
            for _ in range(8):
                step()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
            torch.cuda.synchronize()
            evs[0].record()
            for i in range(a.steps):
                step()
                evs[i + 1].record()
            torch.cuda.synchronize()
            ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)]
            if r:                                         # round 0 warms both arms up
                times[arm] += ts
                per_round[arm].append(statistics.median(ts))
    for arm in (False, True):
        print(f"{a.switch} = {arm!s:5}: median {statistics.median(times[arm]):.4f} ms/step  (per round: "
              + " ".join(f"{t:.4f}" for t in per_round[arm]) + ")")
    d = statistics.median(times[True]) - statistics.median(times[False])
    print(f"on - off = {d * 1e3:+.1f} us per step ({d / statistics.median(times[False]) * 100:+.2f} %)")


if __name__ == "__main__":
    main()
