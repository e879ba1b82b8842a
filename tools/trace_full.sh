#!/bin/bash
# usage (GPU box, repo root): tools/trace_full.sh <tag> <stage>  -> gpurun_out/<tag>_seq.txt: kernel sequence of one M-full step
set -u
TAG=$1; STAGE=${2:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_full.py --stage $STAGE --steps 4 > $OUT/bench_trace.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# one step = from one subbag-selection burst to the next: use the T-th-from-last .. last 'ntxent' launches (T=6 per step)
import os
idx = [i for i, r in enumerate(rows) if r[2].startswith("ntxent")]
per = int(os.environ.get("NTX_PER_STEP", "1"))      # NT-Xent launches per step: 1 (all patch steps in one launch), T with MURCL_BATCHED_HEAD=0
a, b = idx[-2 * per - 1], idx[-per - 1]             # spans exactly one step, away from the trailing subbag timing loop
seq = rows[a:b]
t0 = seq[0][0]
busy = 0
agg = {}
with open("$OUT/../${TAG}_seq.txt", "w") as f:
    prev_end = seq[0][0]
    for s, e, n in seq:
        busy += e - s
        k = n.split("(")[0][:70]
        agg[k] = (agg.get(k, (0, 0))[0] + 1, agg.get(k, (0, 0))[1] + (e - s))
        f.write("%9.1f us  dur %7.1f  gap %6.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:110]))
        prev_end = e
    f.write("step span %.1f us, kernel busy %.1f us, kernels %d\n" % ((seq[-1][1] - t0) / 1e3, busy / 1e3, len(seq)))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        f.write("%6d x %9.1f us total  %s\n" % (c, t / 1e3, k))
print(open("$OUT/../${TAG}_seq.txt").read()[-2600:])
PY
