#!/bin/bash
# Dev tool: build stand-alone variants of one kernel file for same-box A/B timing and ablations.
#   tools/_abl/build.sh panel_gemm cur                      -> libabl_cur.so from the working tree
#   tools/_abl/build.sh panel_gemm w1 -DPG_WIDE=1           -> variant with extra -D flags
#   tools/_abl/build.sh gemm abl2 -DTNW_ABLATE=2            (gemm.hip: 1 = no atomics, 2 = no LDS reads/MFMAs, 3 = no DMA)
#   tools/_abl/build.sh panel_gemm ref HEAD~3               -> a git revision (4th argument)
# run.py (panel GEMM), run_tn.py (wgrad), run_k2.py / run_k2b.py (attention pooling fwd / bwd) load the .so files by name.
set -eu
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
file=$1; tag=$2; shift 2
rev=""; flags=()
for a in "$@"; do case "$a" in -D*) flags+=("$a");; *) rev=$a;; esac; done
src=$here/_$file.$tag.hip
if [ -n "$rev" ]; then git -C "$root" show "$rev:murcl_amd/csrc/$file.hip" > "$src"; else cp "$root/murcl_amd/csrc/$file.hip" "$src"; fi
sed -i "s#include \"common.h\"#include \"$root/murcl_amd/csrc/common.h\"#; s#include \"k2_common.h\"#include \"$root/murcl_amd/csrc/k2_common.h\"#" "$src"
extra=""
case "$file" in attn_pool_bwd) cp "$root/murcl_amd/csrc/attn_pool.hip" "$here/_attn_pool.dep.hip"; sed -i "s#include \"common.h\"#include \"$root/murcl_amd/csrc/common.h\"#; s#include \"k2_common.h\"#include \"$root/murcl_amd/csrc/k2_common.h\"#" "$here/_attn_pool.dep.hip"; extra="$here/_attn_pool.dep.hip";; esac
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result "${flags[@]}" -shared "$src" $extra -o "$here/libabl_${file}_$tag.so"
rm -f "$src" "$here/_attn_pool.dep.hip"
echo "$here/libabl_${file}_$tag.so"
