set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or tn or small or grouped or skinny or linear" 2>&1 | tail -3
for v in 512 0 112; do
  if [ $v = 0 ]; then unset MURCL_TN_PASS; else export MURCL_TN_PASS=$v; fi
  bash tools/tn_trace.sh r04_r_tn$v > gpurun_out/r04_r_tn_pass$v.txt 2>&1; rm -rf gpurun_out/r04_r_tn$v
done
unset MURCL_TN_PASS
for v in 0 auto; do
  if [ $v = auto ]; then unset MURCL_NT_ONE_SLOT; else export MURCL_NT_ONE_SLOT=$v; fi
  for cold in 1 0; do
    SK_COLD=$cold bash tools/skinny_trace.sh r04_r_nt > gpurun_out/r04_r_nt_oneslot${v}_cold$cold.txt 2>&1; rm -rf gpurun_out/r04_r_nt
  done
done
unset MURCL_NT_ONE_SLOT
cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/bench_full.py --stage 2 2>&1 | tail -1 | cut -c1-200
MURCL_TN_PASS=512 MURCL_NT_ONE_SLOT=0 python tools/bench_full.py --stage 2 2>&1 | tail -1 | cut -c1-200
done
