#!/bin/bash
# usage (GPU box, repo root): tools/pmc_lds.sh <tag>   - LDS port counters per kernel of bench.py (a PMC pass of its own):
# SQ_LDS_BANK_CONFLICT (cycles the LDS is stalled by bank conflicts), SQ_LDS_IDX_ACTIVE (cycles its index unit is busy),
# SQ_LDS_ADDR_CONFLICT, SQ_INSTS_LDS, against GRBM_GUI_ACTIVE (summed over the 8 XCDs) -> gpurun_out/<tag>_lds.txt
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --stat-steps 0 > $OUT/lds_pass.log 2>&1
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_lds.txt
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/lds/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-62s %8s %12s %10s %10s %10s" % ("kernel (median per launch)", "launches", "cycles", "LDS busy", "bank conf", "addr conf"))
rows = []
for k, c in d.items():
    med = lambda n: sorted(c[n])[len(c[n]) // 2] if c.get(n) else 0.0
    cyc = med("GRBM_GUI_ACTIVE") / 8.0
    if cyc <= 0: continue
    cu = 256.0
    rows.append((cyc, k, len(c["GRBM_GUI_ACTIVE"]), med("SQ_LDS_IDX_ACTIVE") / cu / cyc, med("SQ_LDS_BANK_CONFLICT") / cu / cyc, med("SQ_LDS_ADDR_CONFLICT") / cu / cyc))
for cyc, k, n, busy, bank, addr in sorted(rows, reverse=True)[:24]:
    print("%-62s %8d %12.0f %9.1f%% %9.1f%% %9.1f%%" % (k, n, cyc, 100 * busy, 100 * bank, 100 * addr))
PY
rm -rf $OUT/lds
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_lds.txt
