#!/bin/bash
# usage (GPU box, repo root): tools/ab_bench.sh <tag> <rounds> lib [lib ...]   - bench.py per library ("default" = the product library),
# interleaved over <rounds> rounds in one call (boxes differ by +-10 %, runs by +-1 %): prints value / ms / median / the big kernels
set -u
TAG=$1; R=$2; shift 2
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $R); do
  for L in "$@"; do
    if [ "$L" = default ]; then unset MURCL_AMD_LIB; else export MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/$L.so; fi
    python bench.py --no-cpu-baseline > gpurun_out/${TAG}_${L}_$r.json 2>> gpurun_out/${TAG}.err
    python -c "
import json
d=json.load(open('gpurun_out/${TAG}_${L}_$r.json')); k=d['kernel_ms_per_step']
print('%-10s r$r  %9.1f bags/s  %.4f ms  median %.4f | fwd %.4f dgrad %.4f rank1 %.4f wgrad %.4f k2 %.4f' % ('$L', d['value'], d['ms_per_step'], d['step_stats']['median_ms'], k.get('panel_gemm<K512,BIAS_RELU>',0), k.get('panel_gemm<K512,MASK>',0), k.get('panel_gemm<K128,RANK1_MASK>',0), k.get('gemm_tn_sq_grouped3<bf16>',0), k.get('abmil_pool_fwd<bf16>',0)))
"
  done
done
