#!/usr/bin/env python
"""Dev tool: build a VARIANT of libmurcl_amd.so with extra -D flags on some sources, for A/B runs on the GPU box.

    python tools/ab_build.py NAME [source.hip:-DFLAG[,-DFLAG2] ...]

-> tools/_abl/lib/NAME.so (git-ignored, travels with gpurun); select it with MURCL_AMD_LIB=tools/_abl/lib/NAME.so.
Sources without flags reuse the objects of the regular build (murcl_amd/build/*.o)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from murcl_amd import build as B  # noqa: E402


def main():
    name, specs = sys.argv[1], dict(a.split(":", 1) for a in sys.argv[2:])
    B.build()
    out_dir = os.path.join(ROOT, "tools", "_abl", "lib")
    tmp = os.path.join(ROOT, "murcl_amd", "build", "ab_" + name)
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(tmp, exist_ok=True)
    objs, procs = [], []
    for s in B.SOURCES:
        o = os.path.join(B.HERE, "build", s.replace(".hip", ".o"))
        if s in specs:
            o = os.path.join(tmp, s.replace(".hip", ".o"))
            cmd = ["hipcc", *B.FLAGS, *specs[s].split(","), "-c", os.path.join(B.CSRC, s), "-o", o]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit("hipcc failed: %s\n%s" % (" ".join(cmd), out))
    lib = os.path.join(out_dir, name + ".so")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit("link failed\n" + r.stdout)
    print(lib, os.path.getsize(lib) >> 10, "KiB")


if __name__ == "__main__":
    main()
