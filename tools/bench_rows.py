#!/usr/bin/env python
"""bench.py's `rows` extra alone (CLAM-SB C3 / DSMIL C5-share rows), for A/B runs through the MURCL_* switches of functional.py."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = bench.other_rows(torch.device("cuda:0"))
print(json.dumps({k: v["ms"] for k, v in out.items()}))
