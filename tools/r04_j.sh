set -u
cd $GRAFT_REPO_ROOT
T=r04_j
MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/r1nw4.so python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -x -q -k "panel or rank1 or abmil or dgrad or bias_rows" 2>&1 | tail -4
for L in default r1nw4 default r1nw4; do
  if [ "$L" = default ]; then unset MURCL_AMD_LIB; else export MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/$L.so; fi
  echo "== $L"; python tools/kbench.py --only panel_rank1 --reps 30
done
bash tools/ab_bench.sh ${T} 2 default r1nw4
