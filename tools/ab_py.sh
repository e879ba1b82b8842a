#!/bin/bash
# usage (GPU box, repo root): tools/ab_py.sh [rounds]  - alternate bench.py of ./_ab_old (a `git archive` of the previous
# commit + the built .so) and of the working tree on the SAME box; prints ms/step per run
R=${1:-3}
for i in $(seq $R); do
  for t in _ab_old .; do
    (cd $t && python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])")
  done
done
