#!/bin/bash
# usage (GPU box, repo root): tools/pmc_script.sh <tag> <python file> [args]
# FETCH_SIZE and WRITE_SIZE of every kernel of a script, each counter in its own --pmc pass (MI355X_MICROARCH.md: the TCC slots do
# not fit both) -> gpurun_out/<tag>/pmc_traffic.json: per kernel the median counter values (KiB) and the corrected HBM bytes
# (gfx950 reports half of the bytes of wide coalesced reads: 2 x FETCH + WRITE).
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 "$@" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 "$@" > $OUT/write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
def med(kind):
    d = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sorted(v)[len(v)//2], len(v)) for k, v in d.items()}
fetch, write = med("fetch"), med("write")
out = {}
for k in sorted(set(fetch) | set(write)):
    if "at::" in k or "rocclr" in k: continue
    f, w = fetch.get(k, (0, 0)), write.get(k, (0, 0))
    out[k] = {"launches": f[1], "FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0], "hbm_bytes_corrected": int((2 * f[0] + w[0]) * 1024)}
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"])[:14]:
    print("%-72s %4d launches  fetch %9.0f KiB  write %9.0f KiB  -> %7.1f MB" % (k[:72], v["launches"], v["FETCH_SIZE_KiB"], v["WRITE_SIZE_KiB"], v["hbm_bytes_corrected"] / 1e6))
PY
