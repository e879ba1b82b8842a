# usage (GPU box): bash tools/round_measure.sh <tag>   - the closing measurement of a round: GPU tests, bench (default + driver args +
# forced-dist), PMC traffic + MFMA passes, step / CLAM / DSMIL kernel sequences, M-full stages
set -u
cd $GRAFT_REPO_ROOT
T=${1:-r04_z}
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${T}_gpu_tests.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_args.json 2>> gpurun_out/${T}_bench.err
MURCL_FORCE_DIST=1 MASTER_PORT=29541 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_force_dist.json 2>> gpurun_out/${T}_bench.err
bash tools/pmc_bench.sh ${T}_pmc > gpurun_out/${T}_pmc.log 2>&1
bash tools/pmc_mfma.sh ${T}_mfma > gpurun_out/${T}_mfma.log 2>&1
bash tools/trace_step.sh ${T}_step > gpurun_out/${T}_trace_step.log 2>&1
MURCL_SEQ_N=40 bash tools/trace_seq.sh ${T}_dsmil $GRAFT_REPO_ROOT/tools/dsmil_seq.py > gpurun_out/${T}_dsmil_seq.txt 2>&1
MURCL_SEQ_N=36 bash tools/trace_seq.sh ${T}_clam $GRAFT_REPO_ROOT/tools/clam_seq.py train > gpurun_out/${T}_clam_train_seq.txt 2>&1
bash tools/trace_stage.sh ${T} 2 ppo_returns_kernel 2 > /dev/null 2>&1
for s in 1 2 3; do python tools/bench_full.py --stage $s --steps 30 2>&1 | tail -1; done > gpurun_out/${T}_bench_full_stages.jsonl
cat gpurun_out/${T}_gpu_tests.log
python -c "
import json
for f in ('${T}_bench','${T}_bench_driver_args','${T}_bench_force_dist'):
    d=json.loads([l for l in open('gpurun_out/%s.json'%f) if l.startswith('{')][-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_k2']['frac'], d['step_stats']['median_ms'] if d['step_stats'] else None, d.get('m_full'), d.get('comm'))
d=json.load(open('gpurun_out/${T}_bench.json'))
for k,v in d.get('rows',{}).items(): print(k, v.get('ms'))
print(d.get('cpu_baseline'))
"
cat gpurun_out/${T}_bench_full_stages.jsonl
tail -3 gpurun_out/${T}_step_seq.txt
