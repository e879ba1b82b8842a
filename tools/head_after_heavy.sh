#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/head_after_heavy.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
segs, cur, inside = [], None, False
for s, e, name in rows:
    if "FillFunctor<double>" in name:
        cur, inside = [], True
        continue
    if "FillFunctor<short>" in name:
        if inside: segs.append(cur)
        inside = False
        continue
    if inside: cur.append((s, e, name))
modes = ["alone", "after_stream", "after_panel", "after_3_panels"]
for i, seg in enumerate(segs):
    print(modes[i // 6], " ".join("%5.1f" % ((e - s) / 1e3) for s, e, _ in seg), " | starts", " ".join("%5.1f" % ((s - seg[0][0]) / 1e3) for s, e, _ in seg))
PY
