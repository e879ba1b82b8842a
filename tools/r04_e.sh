set -u
cd $GRAFT_REPO_ROOT
T=r04_f
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${T}_tests_kernels.log
python -m pytest tests/test_gpu_modules.py tests/test_gpu_step.py tests/test_gpu_supervised_steps.py tests/test_gpu_eval.py tests/test_gpu_fullsize.py tests/test_gpu_kmeans.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${T}_tests_modules.log
cat gpurun_out/${T}_tests_kernels.log gpurun_out/${T}_tests_modules.log
MURCL_NTX_XCHG=0 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_ntx_old.json 2> gpurun_out/${T}_bench.err
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_ntx_new.json 2>> gpurun_out/${T}_bench.err
MURCL_NTX_XCHG=0 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_ntx_old2.json 2>> gpurun_out/${T}_bench.err
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_ntx_new2.json 2>> gpurun_out/${T}_bench.err
python -c "
import json
for f in ('bench_ntx_old','bench_ntx_new','bench_ntx_old2','bench_ntx_new2'):
    d=json.load(open('gpurun_out/${T}_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['step_stats']['median_ms'], d['kernel_ms_per_step'].get('ntxent'), d['rows']['dsmil_c5_share_fwd_bwd']['ms'] if 'rows' in d else '')
"
MURCL_SEQ_N=40 bash tools/trace_seq.sh ${T}_dsmil $GRAFT_REPO_ROOT/tools/dsmil_seq.py > gpurun_out/${T}_dsmil_seq.txt 2>&1
tail -32 gpurun_out/${T}_dsmil_seq.txt | cut -c1-130
for s in 1 2 3; do python tools/bench_full.py --stage $s --steps 30 2>&1 | tail -1; done > gpurun_out/${T}_bench_full_stages.jsonl
cat gpurun_out/${T}_bench_full_stages.jsonl
python tools/bench_full.py --stage 2 --steps 10 --cprofile > gpurun_out/${T}_stage2_cprofile.txt 2>&1
