#!/usr/bin/env python
"""gpurun_out/<tag>/pmc_traffic_raw.json (tools/pmc_bench.sh) -> profiles/pmc_traffic.json: HBM bytes per launch under the keys
bench.py's kernel timers use, stamped with the run they came from ("_source": bench.py copies it into roofline.traffic_source).

    python tools/pmc_to_profile.py gpurun_out/r05_z_pmc/pmc_traffic_raw.json profiles/r05_z_pmc_traffic_raw.json
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# rocprofv3 kernel name (up to the argument list; a PREFIX: later template parameters - the fragment-order prologue of round 6 - follow)
# -> bench.py key
MAP = {
    "void abmil_pool_fwd_kernel<unsigned short, false": "abmil_pool_fwd<bf16>",
    "abmil_pool_combine_kernel": "abmil_pool_combine",
    "abmil_pool_decoder_kernel": "abmil_pool_decoder",
    "void abmil_pool_bwd_kernel<unsigned short, false": "abmil_pool_bwd<bf16>",
    "void gemm_tn_kernel<unsigned short, 4, false>": "gemm_tn<bf16>",
    "void panel_nt_kernel<512, 32, 8, 0, true, false": "panel_gemm<K512,BIAS_RELU>",
    "void panel_nt_kernel<512, 32, 8, 1, false, false": "panel_gemm<K512,MASK>",
    "void panel_nt_kernel<128, 64, 8, 2, false, false": "panel_gemm<K128,RANK1_MASK>",
    "adam_kernel": "adam",
    "adam_multi_kernel": "adam",
}


def _find(raw, prefix):
    hits = [k for k in raw if k.startswith(prefix)]
    return max(hits, key=lambda k: raw[k]["hbm_bytes_corrected"]) if hits else None


def main():
    raw_path, keep_as = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
    raw = json.load(open(raw_path))
    out = {}
    for name, key in MAP.items():
        hit = _find(raw, name)
        if hit:
            out[key] = raw[hit]["hbm_bytes_corrected"]
    sq = _find(raw, "void gemm_tn_sq_kernel<") or _find(raw, "gemm_tn_sq_kernel")
    if sq:
        out["gemm_tn_sq_grouped3<bf16>"] = raw[sq]["hbm_bytes_corrected"] + raw.get("tn_reduce_kernel", {}).get("hbm_bytes_corrected", 0)
    if keep_as and os.path.abspath(raw_path) != os.path.abspath(os.path.join(ROOT, keep_as)):
        shutil.copy(raw_path, os.path.join(ROOT, keep_as))
    out["_source"] = keep_as or raw_path
    out["_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes over bench.py (tools/pmc_bench.sh); "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); median over the launches of each "
                    "kernel; gemm_tn_sq_grouped3 = the grouped tile kernel (three encoder weight gradients) + its one reduce launch")
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=0)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    main()
