python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_rl_step.py -m gpu -q -x 2>&1 | tail -3
for s in 1 2 3; do python tools/bench_full.py --stage $s 2>/dev/null | tail -1 | cut -c1-200; done
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --stat-steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_stats']['median_ms'], {k:v for k,v in d['kernel_ms_per_step'].items() if 'f32' in k})"
