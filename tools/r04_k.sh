set -u
cd $GRAFT_REPO_ROOT
T=r04_k
MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/pgdec.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -x -q -k "panel or abmil or clam or dgrad or bias_rows or gate" 2>&1 | tail -4
for L in default pgdec default pgdec; do
  if [ "$L" = default ]; then unset MURCL_AMD_LIB; else export MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/$L.so; fi
  echo "== $L"; timeout 300 python tools/kbench.py --only panel_fwd,panel_mask --reps 30
done
timeout 600 bash tools/ab_bench.sh ${T} 3 default pgdec
