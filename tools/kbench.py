#!/usr/bin/env python
"""Isolated kernel timings at BASELINE configs[1] sizes (HIP events, median of reps).  Dev tool."""
import argparse
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--bags", type=int, default=128)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, N = a.bags, a.n
    M = B * N
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    X = (torch.randn((M, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    W = (torch.randn((512, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
    bias = torch.randn((512,), generator=g, device=dev) * 0.1
    Wa = (torch.randn((128, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16()
    ba = torch.randn((128,), generator=g, device=dev) * 0.1
    wb = torch.randn((1, 128), generator=g, device=dev) * 0.3
    bb = torch.zeros((1,), device=dev)
    dM = torch.randn((B, 512), generator=g, device=dev)
    dT = (torch.randn((M, 128), generator=g, device=dev) * 0.01).bfloat16()
    WaT = Wa.t().contiguous()
    Asc = torch.rand((M,), generator=g, device=dev)
    H, bm, _ = ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True)
    sc, Aw, Mp, ml = ops.abmil_pool_fwd(H.view(B, N, 512), Wa, ba, wb, bb)
    _, part0 = ops.abmil_pool_partials(H.view(B, N, 512), Wa, ba, wb, bb)
    Wd = torch.randn((512, 512), generator=g, device=dev) / math.sqrt(512)
    bd = torch.randn((512,), generator=g, device=dev) * 0.1
    cases = {
        "copy_bf16": (lambda: H.copy_(X), 2 * M * 512 * 2, 0),
        "k2_fwd": (lambda: ops.abmil_pool_fwd(H.view(B, N, 512), Wa, ba, wb, bb), M * 512 * 2, 2.0 * M * 512 * 128),
        "k2_bwd": (lambda: ops.abmil_pool_bwd(H.view(B, N, 512), Wa, ba, wb, sc, ml, Mp, dM), M * 640 * 2, 2.0 * M * 512 * 128),
        "k2_part": (lambda: ops.abmil_pool_partials(H.view(B, N, 512), Wa, ba, wb, bb), M * 512 * 2, 2.0 * M * 512 * 128),
        "k2_bwd_wantA": (lambda: ops.abmil_pool_bwd(H.view(B, N, 512), Wa, ba, wb, sc, ml, Mp, dM, want_A=True), M * 640 * 2, 2.0 * M * 512 * 128),
        "k2_decoder": (lambda: ops.abmil_pool_decoder(part0, B, N, torch.bfloat16, Wd, bd), 0, 0),
        "k2_combine": (lambda: ops.abmil_pool_combine(sc, part0, torch.bfloat16), 0, 0),
        "k2_dec_gemm": (lambda: ops.gemm_nt(Mp, Wd, epi=ops.EPI_BIAS_RELU, bias=bd), 0, 0),
        "panel_fwd": (lambda: ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True), 2 * M * 512 * 2, 2.0 * M * 512 * 512),
        "panel_fwd_nobm": (lambda: ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias), 2 * M * 512 * 2, 2.0 * M * 512 * 512),
        "panel_mask": (lambda: ops.panel_gemm(X, W, ops.PG_MASK, bitmask=bm, colsum=True), 2 * M * 512 * 2, 2.0 * M * 512 * 512),
        "panel_rank1": (lambda: ops.panel_gemm(dT, WaT, ops.PG_RANK1_MASK, bitmask=bm, rowscale=Asc, rank1=dM, rows_per_bag=N, colsum=True),
                        M * (128 + 512) * 2, 2.0 * M * 512 * 128),
        "tile_nt": (lambda: ops.gemm_nt(X, W, epi=ops.EPI_BIAS_RELU, bias=bias), 2 * M * 512 * 2, 2.0 * M * 512 * 512),
        "tn_512": (lambda: ops.gemm_tn(X, H), 2 * M * 512 * 2, 2.0 * M * 512 * 512),
        "tn_128": (lambda: ops.gemm_tn(dT, H), M * 640 * 2, 2.0 * M * 512 * 128),
    }
    # CLAM / DSMIL streaming passes at the C3 shape (64 bags x 4096 x 512 = the same 268 MB) and DSMIL's d = 1024 share
    Bc, Nc = 64, 4096
    Xc = X.view(Bc, Nc, 512)
    V1 = torch.randn((Bc, 1, 512), generator=g, device=dev)
    A1 = torch.rand((Bc, Nc, 1), generator=g, device=dev)
    wc = torch.randn((256,), generator=g, device=dev) * 0.1
    bc1 = torch.zeros((1,), device=dev)
    Xd = (torch.randn((16, 8192, 1024), generator=g, device=dev).abs() * 0.5).bfloat16()
    V2 = torch.randn((16, 2, 1024), generator=g, device=dev)
    A2 = torch.rand((16, 8192, 2), generator=g, device=dev)
    cases.update({
        "rows_dot_d512": (lambda: ops.rows_dot(Xc, V1), M * 512 * 2, 0),
        "wrowsum_d512": (lambda: ops.weighted_rowsum(Xc, A1), M * 512 * 2, 0),
        "gated_score_fwd": (lambda: ops.gated_score_fwd(X, wc, bc1), M * 512 * 2, 0),
        "rows_dot_d1024": (lambda: ops.rows_dot(Xd, V2), Xd.numel() * 2, 0),
        "wrowsum_d1024": (lambda: ops.weighted_rowsum(Xd, A2), Xd.numel() * 2, 0),
        "rows_dot_wsum_d1024": (lambda: ops.rows_dot_wsum(Xd, V2, A2), Xd.numel() * 2, 0),
    })
    # CLAM-SB's gate GEMM forms and its gate backward (C3: 262144 rows, D = 256)
    wa_, wb_ = torch.randn((256, 512), generator=g, device=dev) / math.sqrt(512), torch.randn((256, 512), generator=g, device=dev) / math.sqrt(512)
    ba_, bb_ = torch.randn((256,), generator=g, device=dev) * 0.1, torch.randn((256,), generator=g, device=dev) * 0.1
    W_il, b_il, c_il = ops.gate_interleave(wa_, ba_, wb_, bb_, wc, torch.bfloat16)
    dsa, dsb = ops.DropSeed(0.75, seed=11), ops.DropSeed(0.75, seed=12)
    U_il, s_il = ops.panel_gate_u(H, W_il, b_il, c_il, bc1)
    A_sm = ops.softmax_rows(s_il.view(Bc, Nc))
    Mp_c = ops.weighted_rowsum(H.view(Bc, Nc, 512), A_sm.view(Bc, Nc, 1)).view(Bc, 512)
    dM_c = torch.randn((Bc, 512), generator=g, device=dev)
    ds_c = torch.randn((M,), generator=g, device=dev) * 1e-3
    gbytes = 2 * M * 512 * 2
    Zr = torch.zeros_like(X)
    H2 = torch.empty_like(X)
    Rb = (torch.randn((M, 512), generator=g, device=dev) * 0.5).bfloat16()          # both signs: more bits toggle
    cases.update({
        "panel_bias": (lambda: ops.panel_gemm(H, W_il, ops.PG_BIAS, bias=b_il), gbytes, 2.0 * M * 512 * 512),
        "panel_gate_score": (lambda: ops.panel_gate_score(H, W_il, b_il, c_il, bc1), M * 512 * 2, 2.0 * M * 512 * 512),
        "panel_gate_u": (lambda: ops.panel_gate_u(H, W_il, b_il, c_il, bc1), gbytes, 2.0 * M * 512 * 512),
        "panel_gate_u_drop": (lambda: ops.panel_gate_u(H, W_il, b_il, c_il, bc1, dsa, dsb), gbytes, 2.0 * M * 512 * 512),
        "panel_fwd_drop": (lambda: ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True, drop=dsa), gbytes, 2.0 * M * 512 * 512),
        "datadep_fwd_zeros": (lambda: ops.panel_gemm(Zr, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True), gbytes, 2.0 * M * 512 * 512),
        "datadep_fwd_half_zero": (lambda: ops.panel_gemm(H, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True), gbytes, 2.0 * M * 512 * 512),
        "datadep_fwd_dense": (lambda: ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True), gbytes, 2.0 * M * 512 * 512),
        "datadep_fwd_randbits": (lambda: ops.panel_gemm(Rb, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True), gbytes, 2.0 * M * 512 * 512),
        "datadep_tn_zeros": (lambda: ops.gemm_tn(Zr, Zr), gbytes, 2.0 * M * 512 * 512),
        "datadep_tn_dense": (lambda: ops.gemm_tn(X, H), gbytes, 2.0 * M * 512 * 512),
        "datadep_k2_zeros": (lambda: ops.abmil_pool_fwd(Zr.view(B, N, 512), Wa, ba, wb, bb), M * 512 * 2, 2.0 * M * 512 * 128),
        "datadep_k2_dense": (lambda: ops.abmil_pool_fwd(H.view(B, N, 512), Wa, ba, wb, bb), M * 512 * 2, 2.0 * M * 512 * 128),
        "datadep_copy_zeros": (lambda: H2.copy_(Zr), gbytes, 0),
        "datadep_copy_dense": (lambda: H2.copy_(X), gbytes, 0),
        "pair_fc_gate_u": (lambda: (ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True, out=H), ops.panel_gate_u(H, W_il, b_il, c_il, bc1)), 2 * gbytes, 0),
        "pair_fc_biasgate": (lambda: (ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True, out=H), ops.panel_gemm(H, W_il, ops.PG_BIAS, bias=b_il)), 2 * gbytes, 0),
        "pair_fc_gatescore": (lambda: (ops.panel_gemm(X, W, ops.PG_BIAS_RELU, bias=bias, want_bitmask=True, out=H), ops.panel_gate_score(H, W_il, b_il, c_il, bc1)), 2 * gbytes, 0),
        "gate_bwd_nat": (lambda: ops.gated_score_bwd(U_il, wc, ds_c), gbytes, 0),
        "gate_bwd_il": (lambda: ops.gated_score_bwd_il(U_il, wc, ds=ds_c), gbytes, 0),
        "gate_bwd_il_onepass": (lambda: ops.gated_score_bwd_il(U_il, wc, h=H, dM=dM_c, Mp=Mp_c, A=A_sm.view(-1), rows_per_bag=Nc), 3 * M * 512 * 2, 0),
        "gate_bwd_il_onepass_drop": (lambda: ops.gated_score_bwd_il(U_il, wc, dsa, dsb, h=H, dM=dM_c, Mp=Mp_c, A=A_sm.view(-1), rows_per_bag=Nc), 3 * M * 512 * 2, 0),
    })
    Xf = torch.randn((16, 8192, 1024), generator=g, device=dev).abs() * 0.5                # DSMIL C5 share in f32
    cases.update({
        "rows_dot_f32_d1024": (lambda: ops.rows_dot(Xf, V2), Xf.numel() * 4, 0),
        "wrowsum_f32_d1024": (lambda: ops.weighted_rowsum(Xf, A2), Xf.numel() * 4, 0),
        "rows_dot_wsum_f32_d1024": (lambda: ops.rows_dot_wsum(Xf, V2, A2), Xf.numel() * 4, 0),
    })
    G3 = [torch.zeros((512, 512), device=dev) for _ in range(3)]
    cases.update({
        "tn_g3": (lambda: ops.gemm_tn_grouped([(X, H, G3[0], None, None), (Rb, X, G3[1], None, None), (H, Rb, G3[2], None, None)]),
                  3 * gbytes, 3 * 2.0 * M * 512 * 512),
    })
    for name, (fn, nbytes, flops) in cases.items():
        if a.only and not any(tok in name for tok in a.only.split(",")):
            continue
        med, mn = timeit(fn, a.reps)
        print(f"{name:16s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us   {nbytes / med / 1e6:8.1f} GB/s  {flops / med / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
