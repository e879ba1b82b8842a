#!/bin/bash
# usage (GPU box, repo root): tools/pmc_mfma.sh <tag>
# MFMA utilisation per kernel of bench.py from rocprofv3 PMC counters, in a pass of their own (counters only with
# --kernel-trace; the program directly after `--`):
#   SQ_VALU_MFMA_BUSY_CYCLES  cycles in which a SIMD's matrix pipe is busy (summed over SIMDs)
#   SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE  the time base (GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = /8)
#   SQ_INSTS_VALU_MFMA_MOPS_BF16  matrix operations issued (x 512 FLOP each on gfx950 counters, reported raw)
# Calibration: the same counters over tools/_abl/mfma_peak.hip (back-to-back v_mfma_f32_16x16x32_bf16, 2 waves per SIMD),
# whose utilisation is 100 % by construction; per-kernel utilisation is reported raw and relative to it.
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
CNT=""
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do
  grep -qw "$c" $OUT/counters_available.txt && CNT="$CNT $c"
done
echo "counters: $CNT" > $OUT/mfma_pass.log
hipcc --offload-arch=gfx950 -O3 -o $OUT/mfma_peak $GRAFT_REPO_ROOT/tools/_abl/mfma_peak.hip >> $OUT/mfma_pass.log 2>&1
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/mfma_cal -- $OUT/mfma_peak >> $OUT/mfma_pass.log 2>&1
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/mfma -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --stat-steps 0 >> $OUT/mfma_pass.log 2>&1
python3 - <<PY
import csv, glob, json, collections
def collect(kind):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in d.items()}
def util(c):
    base = c.get("GRBM_GUI_ACTIVE")
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if not base or busy is None: return None
    return busy / (base / 8.0 * 256 * 4)          # busy SIMD-cycles / (kernel cycles x 1024 SIMDs)
cal = collect("mfma_cal")
cal_u = max([util(c) or 0 for c in cal.values()] or [0])
out = {"_calibration": {"kernel": "tools/_abl/mfma_peak.hip", "raw_util_at_full_rate": cal_u, "counters": cal}}
for k, c in sorted(collect("mfma").items()):
    if "at::" in k or "rocclr" in k or "Cijk" in k: continue
    u = util(c)
    out[k] = {"counters_median_per_launch": c, "mfma_busy_raw": u, "mfma_busy_vs_calibration": (u / cal_u if (u is not None and cal_u) else None)}
json.dump(out, open("$OUT/mfma_util.json", "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"): print(k[:60], v["mfma_busy_raw"], v["mfma_busy_vs_calibration"])
print("calibration raw util", cal_u)
PY
