#!/bin/bash
# usage (GPU box, repo root): tools/ab_run.sh <tag> <kbench filter> [lib ...]  -> rocprofv3 kernel averages of tools/kbench.py per library
# (the default library first, then each tools/_abl/lib/<lib>.so)
set -u
TAG=$1; ONLY=$2; shift 2
for L in default "$@"; do
  if [ "$L" = default ]; then unset MURCL_AMD_LIB; else export MURCL_AMD_LIB=$GRAFT_REPO_ROOT/tools/_abl/lib/$L.so; fi
  echo "== $L"
  bash $GRAFT_REPO_ROOT/tools/trace_kbench.sh ${TAG}_$L "$ONLY"
done
