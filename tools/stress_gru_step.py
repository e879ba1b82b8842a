"""Stress: the one-launch GRU step kernels against the GEMM + gate-kernel pairs they replace, thousands of launches on changing data
(hand-counted vmcnt / LDS-DMA rings: an intermittent race shows up as a rare mismatch).  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for B, H, Kx in ((64, 512, 0), (128, 512, 512), (128, 1024, 0), (37, 96, 48), (320, 512, 0)):
    whh = torch.randn(3 * H, H, device=dev) / H ** 0.5
    bhh = torch.randn(3 * H, device=dev)
    wih = torch.randn(3 * H, max(Kx, 16), device=dev) / max(Kx, 16) ** 0.5
    bih = torch.randn(3 * H, device=dev)
    whh_t = whh.t().contiguous()
    worst = 0.0
    for it in range(int(os.environ.get("STRESS_ITERS", "600"))):
        hp = torch.randn(B, H, device=dev)
        if Kx:
            x = torch.randn(B, Kx, device=dev)
            gi = ops.gemm_nt(x, wih, epi=ops.EPI_BIAS, bias=bih)
            h1, g1, gh1 = ops.gru_step_fwd(bih, hp, whh, bhh, x=x, w_ih=wih)
        else:
            gi = torch.randn(B, 3 * H, device=dev)
            h1, g1, gh1 = ops.gru_step_fwd(gi, hp, whh, bhh)
        gh = ops.gemm_nt(hp, whh, epi=ops.EPI_BIAS, bias=bhh)
        h0, g0 = ops.gru_gates_fwd(gi, gh, hp)
        e = max((h1 - h0).abs().max().item(), (g1 - g0).abs().max().item(), (gh1 - gh).abs().max().item())
        # backward
        dgh_n, dh = torch.randn(B, 3 * H, device=dev), torch.randn(B, H, device=dev)
        dprev = torch.randn(B, H, device=dev)
        d_ref = dh.clone()
        ops.gemm_nt(dgh_n, whh_t, out=d_ref, accumulate=True)
        dgi0, dgh0 = torch.empty_like(gi), torch.empty_like(gi)
        dp0 = dprev.clone()
        ops.gru_gates_bwd_into(d_ref, g0, gh, hp, dgi0, dgh0, dp0, accumulate=True)
        d1, dp1 = dh.clone(), dprev.clone()
        dgi1, dgh1 = torch.empty_like(gi), torch.empty_like(gi)
        ops.gru_step_bwd(dgh_n, whh_t, d1, g0, gh, hp, dgi1, dgh1, dp1, accumulate=True)
        e = max(e, (d1 - d_ref).abs().max().item(), (dgi1 - dgi0).abs().max().item(), (dgh1 - dgh0).abs().max().item(),
                (dp1 - dp0).abs().max().item())
        worst = max(worst, e)
        if not (e < 2e-3):
            bad += 1
            print("MISMATCH", B, H, Kx, it, e, flush=True)
    print(f"B={B} H={H} Kx={Kx}: worst abs difference {worst:.2e}", flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
