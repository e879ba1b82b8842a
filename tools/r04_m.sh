set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -x -q -k "dsmil or rows_dot or clam_softmax" 2>&1 | tail -3
for L in default rdold default rdold; do
  if [ "$L" = default ]; then unset MURCL_AMD_LIB; else export MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/$L.so; fi
  echo "== $L"; python tools/kbench.py --only rows_dot_d512,rows_dot_d1024,rows_dot_f32 --reps 30 | grep -v wsum
  echo "rows " $(python tools/bench_rows.py 2>/dev/null | tail -1)
done
