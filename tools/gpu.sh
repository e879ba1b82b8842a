#!/bin/bash
# usage (build container): tools/gpu.sh <timeout seconds> '<command>'  - rebuild the library if csrc/ changed (a stale .so travels
# silently otherwise), then run the command on an MI355X box through gpurun
set -e
set -o pipefail          # a failed rebuild must stop the script: `| tail` alone would hide it and gpurun would run the stale .so
cd "$(dirname "$0")/.."
python -m murcl_amd.build | tail -1
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
