#!/usr/bin/env python
"""Dev tool: cost of a grid-wide barrier inside a kernel (tools/_abl/grid_barrier_probe.hip) against the ~4.5 us floor of a dependent
launch: per barrier = (T(P phases with barriers) - T(P phases without)) / P, for 64..512 workgroups and 0 / 1 KiB / 16 KiB / 64 KiB
of payload exchanged per workgroup and phase; `wrong` counts payload words that arrived stale (must be 0 with barriers).

    python tools/grid_barrier.py
"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = os.path.join(ROOT, "tools", "_abl", "lib", "probes.so")


def main():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "_abl", "build_probes.py")])
    L = ctypes.CDLL(PROBES)
    fn = L.murcl_debug_grid_barrier
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4
    dev = torch.device("cuda:0")
    P = 200

    def run(grid, payload, barrier):
        buf = torch.zeros((2 * grid * max(payload, 1),), device=dev)
        cnt = torch.zeros((16 * 9,), dtype=torch.int32, device=dev)
        err = torch.zeros((2,), dtype=torch.int32, device=dev)
        ts = []
        for _ in range(7):
            cnt.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            assert fn(grid, payload, P, barrier, cnt.data_ptr(), buf.data_ptr(), err.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        e = err.tolist()
        return ts[len(ts) // 2], e[0], e[1]

    print(f"{'WGs':>5s} {'payload':>8s} {'us/phase with':>14s} {'without':>8s} {'per barrier':>12s} {'wrong':>6s} {'timeout':>8s} {'(stale words without a barrier)':>32s}")
    for grid in (64, 128, 256, 512):
        for payload in (0, 256, 4096, 16384):
            t1, bad1, to1 = run(grid, payload, 1)
            t2, bad2, to2 = run(grid, payload, 2)
            t0, bad0, _ = run(grid, payload, 0)
            print(f"{grid:5d} {payload * 4:8d} {t1 / P:14.2f} {t0 / P:8.2f} {(t1 - t0) / P:12.2f} {bad1:6d} {to1:8d} {bad0 != 0!s:>32s}"
                  f"   two-level: {(t2 - t0) / P:6.2f} us per barrier, wrong {bad2}, timeout {to2}", flush=True)


if __name__ == "__main__":
    main()
