#!/bin/bash
# usage (GPU box, repo root): tools/trace_totals.sh <tag> <python file> [args]  -> per-kernel totals (calls, total us, avg us) of the whole run
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.1f us" % (tot / 1e3))
    for r in rows[:30]:
        print("%6d x %9.1f us avg  %8.1f us total %6.2f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, float(r["Percentage"]), r["Name"][:80]))
PY
tail -1 $OUT/run.log | cut -c1-200
