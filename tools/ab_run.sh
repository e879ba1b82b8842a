#!/bin/bash
# Dev tool (GPU box, repo root): per-variant step time (untraced) and big-kernel durations (traced) for A/B libraries
# built by tools/ab_build.sh.   usage: tools/ab_run.sh base k2brev ...
for rep in 1 2; do for v in "$@"; do
  L=$PWD/tools/_abl/libfull_$v.so
  t=$(MURCL_AMD_LIB=$L python tools/host_vs_gpu.py 2>/dev/null | tail -1 | sed 's/.*total //')
  MURCL_AMD_LIB=$L tools/trace_step.sh ab_$v > /dev/null 2>&1
  k=$(awk '$4>=40 {printf "%s ", $4}' gpurun_out/ab_${v}_seq.txt)
  echo "$v: $t | $k"
done; done
