#!/bin/bash
# Dev tool: build a complete alternative libmurcl_amd (all kernels) with extra -D flags for same-box A/B runs:
#   tools/ab_build.sh k2brev -DK2B_REVERSE=1   ->  tools/_abl/libfull_k2brev.so ; run with MURCL_AMD_LIB=$PWD/tools/_abl/libfull_k2brev.so
set -eu
root=$(cd "$(dirname "$0")/.." && pwd); tag=$1; shift
tmp=$(mktemp -d)
for f in gemm panel_gemm attn_pool attn_pool_bwd ntxent elementwise subbag dsmil clam ppo kmeans stream_probe; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result "$@" -c "$root/murcl_amd/csrc/$f.hip" -o "$tmp/$f.o" 2>/dev/null &
done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_abl/libfull_$tag.so" "$tmp"/*.o; rm -rf "$tmp"; echo "$root/tools/_abl/libfull_$tag.so"
