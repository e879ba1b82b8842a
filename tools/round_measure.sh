set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r03_zz_gpu_tests.log
python bench.py > gpurun_out/r03_zz_bench.json 2> gpurun_out/r03_zz_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_zz_bench_driver_args.json 2>> gpurun_out/r03_zz_bench.err
bash tools/pmc_bench.sh r03_zz_pmc > gpurun_out/r03_zz_pmc.log 2>&1
bash tools/pmc_mfma.sh r03_zz_mfma > gpurun_out/r03_zz_mfma.log 2>&1
bash tools/trace_step.sh r03_zz_step > gpurun_out/r03_zz_trace_step.log 2>&1
cat gpurun_out/r03_zz_gpu_tests.log
python -c "
import json
for f in ('r03_zz_bench','r03_zz_bench_driver_args'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_k2']['frac'], d['step_stats']['median_ms'], d.get('m_full'))
"
