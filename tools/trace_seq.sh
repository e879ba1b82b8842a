#!/bin/bash
# usage (GPU box, repo root): tools/trace_seq.sh <tag> <python file> [args]  -> kernel sequence (start, duration, gap) of the LAST
# `MURCL_SEQ_N` (default 80) kernels of the run + per-kernel totals; for short scripts that end with the region of interest
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, os
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(os.environ.get("MURCL_SEQ_N", "80"))
seq = rows[-n:]
t0, prev = seq[0][0], seq[0][0]
agg = {}
for s, e, name in seq:
    k = name.split("(")[0][:60]
    agg[k] = (agg.get(k, (0, 0))[0] + 1, agg.get(k, (0, 0))[1] + (e - s))
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name[:100]))
    prev = e
print("span %.1f us, busy %.1f us, kernels %d" % ((seq[-1][1] - t0) / 1e3, sum(e - s for s, e, _ in seq) / 1e3, len(seq)))
PY
