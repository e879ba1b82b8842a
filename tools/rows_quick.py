#!/usr/bin/env python
"""Dev tool: bench.py's `rows` (CLAM-SB C3, DSMIL C5 share) on their own, twice (the first pass warms allocator and caches)."""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
for rep in range(2):
    r = bench.other_rows(dev)
    print({k: (v.get("ms"), v.get("ms_back_to_back")) for k, v in r.items()}, flush=True)
    gc.collect()
    torch.cuda.empty_cache()
