#!/usr/bin/env python
"""Dev tool: bytes per clock one CU's LDS delivers to waves that only read fragments - ds_read_b128 against the 8-byte forms
(plain and transposing) a TN product's operands need (tools/_abl/lds_rate_probe.hip; DESIGN section 9, the weight gradient)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = os.path.join(ROOT, "tools", "_abl", "lib", "probes.so")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "_abl", "build_probes.py")], stdout=subprocess.DEVNULL)
lib = ctypes.CDLL(PROBES)
lib.murcl_debug_lds_rate.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
cyc = torch.zeros((256, 8), dtype=torch.int64, device=dev)
sink = torch.zeros((4,), device=dev)
iters = 4096
for waves in (4, 8):
    for mode, name, bpl in ((0, "ds_read_b128", 16), (3, "ds_read_b128 (q4*256)", 16), (1, "ds_read_b64", 8), (2, "ds_read_b64_tr_b16", 8), (4, "ds_read_b64 (+32 B rows)", 8), (5, "ds_read_b64_tr (+32 B rows)", 8)):
        for _ in range(2):
            cyc.zero_()
            lib.murcl_debug_lds_rate(mode, waves, 256, iters, cyc.data_ptr(), sink.data_ptr(), None)
            torch.cuda.synchronize()
        c = cyc[:, :waves].double()
        byts = waves * iters * 16 * 64 * bpl
        print(f"{waves} waves  {name:28s} {byts / float(c.max(1).values.median()):7.1f} B/clk/CU  "
              f"({float(c.median()) / (iters * 16):5.2f} cycles per wave instruction)")
