#!/bin/bash
# usage (GPU box): tools/trace_kbench.sh <tag> <kbench filter>   -> per-kernel stats of tools/kbench.py --only <filter>
set -u
TAG=$1; ONLY=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --only "$ONLY" --reps 20 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print("%6d x %9.1f us avg  %6.2f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:90]))
PY
