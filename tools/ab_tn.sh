python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm_tn" 2>&1 | tail -3
for v in 0 1 0 1; do echo "MURCL_TN_SQ=$v"; MURCL_TN_SQ=$v python tools/kbench.py --only tn_512 --reps 30; MURCL_TN_SQ=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --stat-steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_stats']['median_ms'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], {k:v for k,v in d['kernel_ms_per_step'].items() if 'tn' in k})"; done
