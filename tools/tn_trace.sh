#!/bin/bash
# usage (GPU box, repo root): tools/skinny_trace.sh <tag>  -> GPU time per call of each bag-level f32 GEMM shape (tools/tn_trace.py)
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/tn_trace.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, re, ast
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
log = open("$OUT/run.log").read()
m = re.search(r"SHAPES (\[.*\]) (\d+)", log)
shapes, reps = ast.literal_eval(m.group(1)), int(m.group(2))
segs, cur, inside = [], None, False
for s, e, name in rows:
    if "FillFunctor<double>" in name:
        cur, inside = [], True
        continue
    if "FillFunctor<short>" in name:
        if inside and cur: segs.append(cur)
        inside = False
        continue
    if inside and "copy" not in name.lower() and "elementwise" not in name.lower(): cur.append((s, e, name))
for (M, N, K), seg in zip(shapes, segs):
    busy = sum(e - s for s, e, _ in seg) / 1e3 / reps
    names = sorted({n.split("(")[0][:28] for _, _, n in seg})
    fl = 2.0 * M * N * K
    g = sorted((e - s) / 1e3 for s, e, n in seg if "gemm" in n)
    print("M=%4d N=%5d K=%5d: %6.1f us/call (GEMM kernel alone, median %5.1f)  %5.1f TFLOP/s  W %5.2f MB  %.0f launches/call  %s" % (M, N, K, busy, g[len(g) // 2], fl / busy / 1e6, N * K * 4 / 1e6, len(seg) / reps, names))
PY
