#!/usr/bin/env python
"""Dev tool: workgroup -> XCD placement of a launch (tools/_abl/xcc_map_probe.hip).  The kernels that co-locate workgroups sharing
operand slabs assume blocks b and b + 8 share an XCD; this prints, per launch shape, how many workgroups sit where `b % 8` says
(relative to block 0's XCD) and the workgroups per XCD.

    python tools/xcc_map.py
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
    fn = L.murcl_debug_xcc_map
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    for grid, threads, lds, label in [(256, 512, 128 * 1024, "1 WG/CU, 512 thr, 128 KiB (grouped wgrad / panel GEMM)"),
                                      (252, 512, 128 * 1024, "252 WGs (21 splits x 12 tiles)"),
                                      (512, 256, 66 * 1024, "2 WG/CU, 256 thr, 66 KiB (K2)"),
                                      (248, 512, 128 * 1024, "248 WGs (CU budget)"),
                                      (1024, 256, 0, "1024 small WGs")]:
        for rep in range(3):
            out = torch.zeros((grid * 3,), dtype=torch.int32, device=dev)
            rc = fn(grid, threads, lds, 20.0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
            o = out.view(grid, 3).cpu()
            xcc = (o[:, 0] & 0xF).tolist()
            base = xcc[0]
            ok = sum(1 for b, x in enumerate(xcc) if (x - base) % 8 == b % 8)
            per = [xcc.count(k) for k in range(8)]
            print(f"{label:60s} rep {rep}: {ok}/{grid} where b%8 says (block 0 on XCD {base}); per XCD {per}; first 16: {xcc[:16]}")


if __name__ == "__main__":
    main()
