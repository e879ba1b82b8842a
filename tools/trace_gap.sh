#!/bin/bash
# usage (GPU box): tools/trace_gap.sh <tag>  - what sits between the last kernel of a step (cast_flat) and the first of the next (panel forward)?
# kernel + memory-copy + HIP API traces of a short bench run; prints the activities around that boundary of one late step
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --stat-steps 0 --no-graph > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob("$OUT/trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("K cast_flat")]
i = idx[-3]
t0 = rows[i][0]
for s, e, n in rows[max(0, i - 2): i + 4]:
    print("%9.1f us .. %9.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, n))
api = []
for f in glob.glob("$OUT/trace/*/*hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
api.sort()
print("HIP API calls (host) whose launch falls between that cast_flat's launch and the panel launch: see counts by name")
import collections
c = collections.Counter(n for s, e, n in api)
print(dict(c))
PY
