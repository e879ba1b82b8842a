#!/bin/bash
# usage (GPU box): bash tools/kd_ab.sh <variants...>  - kbench of the fused pooling backward under each A/B library of tools/_abl/lib
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset MURCL_AMD_LIB; else export MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/$v.so; fi
  echo "== $v"; python tools/kbench.py --only k2_bwd_dwa --reps 30 2>&1 | grep k2_bwd
done
unset MURCL_AMD_LIB
