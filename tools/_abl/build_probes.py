#!/usr/bin/env python
"""Dev tool: build the measurement probes (LDS-DMA stream skeleton, fused-encoder prototype, L2 loader probe) into
tools/_abl/lib/probes.so - their own library, so that the product's libmurcl_amd.so carries no lab equipment."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "murcl_amd", "csrc")
OUT = os.path.join(HERE, "lib", "probes.so")
SOURCES = [f for f in sorted(os.listdir(HERE)) if f.endswith("_probe.hip")]


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-I", CSRC, "-Wno-unused-value",
           *[os.path.join(HERE, s) for s in SOURCES], "-o", OUT]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit("hipcc failed: %s\n%s" % (" ".join(cmd), r.stdout))
    print(OUT, os.path.getsize(OUT) >> 10, "KiB")


if __name__ == "__main__":
    main()
