#!/bin/bash
# usage (GPU box, repo root): tools/pmc_bench.sh <tag>
# rocprofv3 passes over bench.py: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes
# (MI355X_MICROARCH.md: TCC slots do not fit both).  Writes gpurun_out/<tag>_* and a per-kernel traffic JSON.
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
def med(kind):
    d = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v)//2] for k, v in d.items()}
fetch, write = med("fetch"), med("write")
out = {}
for k in sorted(set(fetch) | set(write)):
    if "at::" in k or "rocclr" in k: continue
    # FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 reports exactly half of the bytes of wide (16 B/lane) coalesced reads
    out[k] = {"FETCH_SIZE_KiB": fetch.get(k), "WRITE_SIZE_KiB": write.get(k),
              "hbm_bytes_corrected": int((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024)}
json.dump(out, open("$OUT/pmc_traffic_raw.json", "w"), indent=1)
for k, v in out.items(): print(k[:70], v)
PY
