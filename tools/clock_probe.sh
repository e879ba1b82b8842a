#!/bin/bash
# usage (GPU box, repo root): tools/clock_probe.sh <kbench filter> [reps]  -> sclk / power samples (rocm-smi) while tools/kbench.py loops that case
ONLY=$1; REPS=${2:-3000}
python3 tools/kbench.py --only "$ONLY" --reps $REPS > /tmp/kb.log 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)|Socket Power|Average Graphics" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $PID
tail -3 /tmp/kb.log
