python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_rl_step.py tests/test_gpu_step.py -m gpu -q -x 2>&1 | tail -3
for v in 0 1 0 1; do echo "MURCL_SKINNY16=$v"; for s in 2 3; do MURCL_SKINNY16=$v python tools/bench_full.py --stage $s 2>/dev/null | tail -1 | cut -c90-200; done
MURCL_SKINNY16=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --stat-steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_stats']['median_ms'], {k:v for k,v in d['kernel_ms_per_step'].items() if 'f32' in k})"; done
