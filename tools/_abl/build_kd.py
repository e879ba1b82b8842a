#!/usr/bin/env python
"""Dev tool: build the parked one-pass pooling backward (attn_pool_bwd_dwa.hip: dT AND dWa = dT^T H from one pass over H; lost
its A/B in round 5 - 130 + 10.5 us against 65 + 5 + 65 - and left the product library in round 6) into tools/_abl/lib/kd.so.
It resolves murcl_abmil_pool_workspace / murcl_cu_budget from the product library it is linked against.
tests/test_gpu_kernels.py keeps its parity test and loads this library when it exists."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "lib", "kd.so")


def main(extra=()):
    from murcl_amd import build as B
    B.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["hipcc", *B.FLAGS, *extra, "-shared", "-I", B.CSRC, os.path.join(HERE, "attn_pool_bwd_dwa.hip"), "-o", OUT,
           "-L", B.HERE, "-l:libmurcl_amd.so", "-Wl,-rpath," + B.HERE]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit("hipcc failed: %s\n%s" % (" ".join(cmd), r.stdout))
    print(OUT, os.path.getsize(OUT) >> 10, "KiB")
    return OUT


if __name__ == "__main__":
    main(sys.argv[1:])
