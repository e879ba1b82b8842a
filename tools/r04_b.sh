set -u
cd $GRAFT_REPO_ROOT
T=r04_b
bash tools/ab_run.sh ${T} tn_g3 tn5 > gpurun_out/${T}_ab_tn5.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench.err
MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/tn5.so python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_tn5.json 2>> gpurun_out/${T}_bench.err
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_default2.json 2>> gpurun_out/${T}_bench.err
MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/tn5.so python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_tn52.json 2>> gpurun_out/${T}_bench.err
python tools/l2_loader_probe.py > gpurun_out/${T}_l2_loader_probe.txt 2>&1
cat gpurun_out/${T}_ab_tn5.log
python -c "
import json
for f in ('bench_default','bench_tn5','bench_default2','bench_tn52'):
    d=json.load(open('gpurun_out/${T}_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['step_stats']['median_ms'], d['kernel_ms_per_step'].get('gemm_tn_sq_grouped3<bf16>'))
"
head -60 gpurun_out/${T}_l2_loader_probe.txt
