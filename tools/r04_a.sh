set -u
cd $GRAFT_REPO_ROOT
T=r04_a
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm_tn or grouped or gate_backward or partial_bias" 2>&1 | tail -6 > gpurun_out/${T}_tests_kernels.log
python -m pytest tests/test_gpu_modules.py tests/test_gpu_step.py tests/test_gpu_rl_step.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${T}_tests_modules.log
MURCL_GROUP_WGRAD=0 python bench.py --no-cpu-baseline --breakdown > gpurun_out/${T}_bench_ungrouped.json 2> gpurun_out/${T}_bench_ungrouped.err
python bench.py --no-cpu-baseline --breakdown > gpurun_out/${T}_bench_grouped.json 2> gpurun_out/${T}_bench_grouped.err
MURCL_GROUP_WGRAD=0 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_ungrouped2.json 2>> gpurun_out/${T}_bench_ungrouped.err
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_grouped2.json 2>> gpurun_out/${T}_bench_grouped.err
bash tools/trace_step.sh ${T}_step > gpurun_out/${T}_trace_step.log 2>&1
cat gpurun_out/${T}_tests_kernels.log gpurun_out/${T}_tests_modules.log
python -c "
import json
for f in ('bench_ungrouped','bench_grouped','bench_ungrouped2','bench_grouped2'):
    d=json.load(open('gpurun_out/${T}_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline_k2']['frac'], d['step_stats']['median_ms'])
"
tail -5 gpurun_out/${T}_step_seq.txt
