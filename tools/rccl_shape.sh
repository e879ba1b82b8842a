#!/bin/bash
# usage (GPU box): bash tools/rccl_shape.sh <tag>  - what RCCL's kernels look like to the dispatcher on this image (workgroup size, LDS per
# workgroup, grid, duration) in the one-rank forced-collective bench: the resource shape decides whether a channel workgroup can sit
# beside a persistent tenant (tools/cu_thief.py) -> gpurun_out/<tag>_rccl_shape.txt
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-rccl}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rccl
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && MURCL_FORCE_DIST=1 MASTER_PORT=29547 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $OUT/run.log 2>&1 )
python3 - <<PY > gpurun_out/${TAG}_rccl_shape.txt
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
print("columns:", [c for c in rows[0].keys()])
agg = {}
for r in rows:
    n = r["Kernel_Name"]
    if "murcl" in n or "nccl" in n.lower() or "rccl" in n.lower() or "Dev" in n:
        key = (n.split("(")[0][:70], r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"))
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if "nccl" in k[0].lower() or "rccl" in k[0].lower() or "Dev" in k[0]:
        print("RCCL  %-70s wg %s grid %s lds %s scratch %s vgpr %s agpr %s sgpr %s  x%d avg %.1f us" % (k + (c, t / c / 1e3)))
names = sorted({r["Kernel_Name"].split("(")[0][:60] for r in rows if "murcl" not in r["Kernel_Name"]})
print("non-library kernels:", [n for n in names if not any(s in n for s in ("panel_nt", "gemm_", "abmil_", "gru_", "adam", "cast_", "ntxent", "relu_", "colsum", "tn_", "calib", "copy_bytes", "axpby", "mean_small", "stack_", "step_draws", "subbag", "ps_", "ppo_", "dsmil", "rows_dot", "weighted", "gated", "softmax", "topk", "clam", "take_rows", "scatter", "cross_entropy", "dropout", "mul_", "transpose", "kmeans"))][:40])
PY
rm -rf $OUT/trace
cat gpurun_out/${TAG}_rccl_shape.txt
