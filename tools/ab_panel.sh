python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q -x -k "panel or abmil or clam" 2>&1 | tail -3
python tools/kbench.py --only panel --reps 30
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline --stat-steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_stats']['median_ms'], {k:v for k,v in d['kernel_ms_per_step'].items() if 'panel' in k})"; done
