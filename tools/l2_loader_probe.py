#!/usr/bin/env python
"""Dev tool: the fused-encoder prototype's weight loader ALONE - which L2 -> LDS rate per CU does it reach?
Prints GB/s per CU and chip-wide for a sweep of (loader waves, ring slots, barrier, rotation, table size, row stride)."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PROBES = os.path.join(HERE, "_abl", "lib", "probes.so")
if not os.path.exists(PROBES):
    subprocess.check_call([sys.executable, os.path.join(HERE, "_abl", "build_probes.py")])
L = ctypes.CDLL(PROBES)
f = L.murcl_debug_l2_loader_probe
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p]
f.restype = ctypes.c_int


def main():
    dev = torch.device("cuda:0")
    out = torch.zeros(4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    iters = 384                                    # 8 tiles x 3 layers x 16 slots: what a CU walks in the fused forward at C2
    print("table_KB stride nw nslot barrier rotate grid | us | GB/s per CU | TB/s chip", flush=True)
    for table_rows in (512, 1536, 2304):
        W = torch.randn((table_rows, 512), device=dev).bfloat16()
        for stride in (1024, 1056):
            for nw, nslot, barrier in ((4, 3, 1), (4, 4, 1), (4, 4, 0), (4, 5, 0), (4, 5, 1), (8, 4, 1), (8, 4, 0), (8, 5, 0), (16, 4, 0), (2, 4, 0)):
                for rotate in (0, 1, 5):
                    for grid in (256,):
                        def run():
                            return f(W.data_ptr(), out.data_ptr(), table_rows, stride, iters, rotate, nw, nslot, barrier, grid, st)
                        rc = run()
                        if rc:
                            continue
                        for _ in range(3):
                            run()
                        torch.cuda.synchronize()
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        for _ in range(10):
                            run()
                        b.record()
                        torch.cuda.synchronize()
                        us = a.elapsed_time(b) * 100
                        per_cu = iters * 32 * 1024 / us / 1e3
                        print(f"{table_rows:5d} {stride:5d} {nw:3d} {nslot:2d} {barrier} {rotate} {grid:4d} | {us:8.1f} | {per_cu:6.1f} | {per_cu * grid / 1e3:6.2f}", flush=True)


if __name__ == "__main__":
    main()
