#!/bin/bash
# usage (GPU box, repo root): tools/trace_step.sh <tag>   -> gpurun_out/<tag>_seq.txt: kernel sequence of one bench step
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# one step = between consecutive adam_kernel pairs near the end
idx = [i for i, r in enumerate(rows) if r[2].startswith("ntxent")]
a, b = idx[-2], idx[-1]
seq = rows[a:b]
t0 = seq[0][0]
busy = 0
with open("$OUT/../${TAG}_seq.txt", "w") as f:
    prev_end = seq[0][0]
    for s, e, n in seq:
        busy += e - s
        f.write("%9.1f us  dur %7.1f  gap %6.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:110]))
        prev_end = e
    f.write("step span %.1f us, kernel busy %.1f us, kernels %d\n" % ((seq[-1][1] - t0) / 1e3, busy / 1e3, len(seq)))
print(open("$OUT/../${TAG}_seq.txt").read()[-300:])
PY
