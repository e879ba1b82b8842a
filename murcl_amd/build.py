"""Build libmurcl_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmurcl_amd.so")
SOURCES = ["gemm.hip", "panel_gemm.hip", "attn_pool.hip", "attn_pool_bwd.hip", "ntxent.hip", "elementwise.hip", "subbag.hip", "dsmil.hip", "clam.hip", "ppo.hip", "kmeans.hip", "stream_probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _stale():
        return LIB
    objs, procs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:                                   # one hipcc per file, in parallel
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = ["hipcc", *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
