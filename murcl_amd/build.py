"""Build libmurcl_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmurcl_amd.so")
SOURCES = ["runtime.hip", "gemm.hip", "panel_gemm.hip", "attn_pool.hip", "attn_pool_bwd.hip", "ntxent.hip", "elementwise.hip", "gru.hip", "subbag.hip", "dsmil.hip", "clam.hip", "ppo.hip", "ppo_seq.hip", "kmeans.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"]


STAMP = LIB + ".srchash"          # travels with the .so (git-ignored like it): the hash of what it was built from


def source_hash():
    """sha256 over every file under csrc/ (names + bytes), the source list and the compiler flags."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((SOURCES, FLAGS)).encode())
    for f in sorted(os.listdir(CSRC)):
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale():
    """The library must be rebuilt when it is missing or was built from other sources (content hash, not mtimes: a
    checkout or a copy to another box resets those)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _stale():
        print(f"murcl_amd.build: {os.path.basename(LIB)} is up to date (source hash {source_hash()[:12]} matches): nothing compiled", flush=True)
        return LIB
    print(f"murcl_amd.build: compiling {len(srcs)} HIP sources for gfx950 ({'forced' if force else 'library missing or sources changed'})", flush=True)
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
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
