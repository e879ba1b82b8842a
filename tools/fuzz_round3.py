#!/usr/bin/env python
"""Randomised shape sweep of the round-3 kernels against float64 / the separate-launch forms (a soak run, not part of the test
suite: tests/test_gpu_kernels.py pins the same comparisons at fixed shapes)."""
import math, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
g = torch.Generator(device=dev); g.manual_seed(7)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = time.time(); n = 0; worst = {}

def rel(a, b, floor=None):
    b = b.double(); s = b.abs().max().item() if floor is None else max(b.abs().max().item(), floor)
    return (a.double() - b).abs().max().item() / max(s, 1e-30)

def note(name, v, tol):
    worst[name] = max(worst.get(name, 0.0), v)
    assert v <= tol, (name, v, tol)

while time.time() - t0 < budget:
    n += 1
    # ---- DSMIL one-pass attention + pooling and its backward
    B = int(rng.integers(1, 6)); C = int(rng.integers(1, 3)); d = int(rng.choice([64, 320, 512, 1024]))
    N = int(rng.choice([4, 8, 12, 64, 100, 96, 256, 1000, 2048, 4100]))
    dt = [torch.float32, torch.bfloat16][int(rng.integers(0, 2))]
    X = torch.relu(torch.randn((B, N, d), generator=g, device=dev)).to(dt)
    v = torch.randn((B, C, d), generator=g, device=dev) * (4.0 / math.sqrt(d))
    S = torch.einsum("bnd,bcd->bnc", X.double(), v.double()); A64 = torch.softmax(S, 1); Z64 = torch.einsum("bnc,bnd->bcd", A64, X.double())
    one = ops.dsmil_attn_pool(X, v)
    if one is not None:
        note("attn_pool.A", rel(one[0], A64), 2e-4); note("attn_pool.Z", rel(one[1], Z64), 2e-4)
        dZ = torch.randn((B, C, d), generator=g, device=dev); dcls = torch.randn((B, N, C), generator=g, device=dev)
        dA = torch.einsum("bnd,bcd->bnc", X.double(), dZ.double()); dS = A64 * (dA - (A64 * dA).sum(1, keepdim=True))
        R64 = torch.einsum("bnc,bnd->bcd", dS, X.double()) * 0.3
        P64 = torch.einsum("bnc,bnd->bcd", A64 * dA, X.double()).abs().max().item() * 0.3
        R, dWc = ops.dsmil_attn_pool_bwd(X, dZ, one[0], one[1], dcls, 0.3)
        note("attn_pool_bwd.R", rel(R, R64, floor=0.1 * P64), 3e-4)
        note("attn_pool_bwd.dWc", rel(dWc, torch.einsum("bnc,bnd->cd", dcls.double(), X.double())), 2e-4)
    # ---- CLAM gate GEMM with scores + one-pass gate backward
    Bc = int(rng.integers(1, 5)); Nc = int(rng.choice([32, 64, 96, 512, 1056, 4096])); M = Bc * Nc
    h = torch.relu(torch.randn((M, 512), generator=g, device=dev)).bfloat16()
    wa, wb = [torch.randn((256, 512), generator=g, device=dev) * (1.5 / math.sqrt(512)) for _ in range(2)]
    ba, bb, wc = [torch.randn((256,), generator=g, device=dev) * 0.1 for _ in range(3)]
    bc = torch.randn((1,), generator=g, device=dev) * 0.1
    drop = bool(rng.integers(0, 2))
    da, db = (ops.DropSeed(0.75, seed=int(rng.integers(1, 1 << 40))), ops.DropSeed(0.75, seed=int(rng.integers(1, 1 << 40)))) if drop else (None, None)
    W_il, b_il, c_il = ops.gate_interleave(wa, ba, wb, bb, wc, torch.bfloat16)
    U_il, s = ops.panel_gate_u(h, W_il, b_il, c_il, bc, da, db)
    Ud = h.double() @ torch.cat([wa, wb], 0).bfloat16().double().t() + torch.cat([ba, bb], 0).double()
    t = torch.tanh(Ud[:, :256]) * torch.sigmoid(Ud[:, 256:])
    if drop:
        t = t * ops.dropout_mask((M, 256), torch.float32, 0.75, dev, seed=da.seed).double() * ops.dropout_mask((M, 256), torch.float32, 0.75, dev, seed=db.seed).double()
    s64 = (t * wc.double()).sum(1) + bc.double()
    note("gate_u.s", rel(s, s64), 4e-3)
    U_nat = U_il.view(M, 16, 2, 16).permute(0, 2, 1, 3).reshape(M, 512)
    note("gate_u.U", rel(U_nat, Ud), 6e-3)
    if ops.gated_bwd_il_supported(M, 256, 512, Nc):
        A = ops.softmax_rows(s.view(Bc, Nc)); Mp = ops.weighted_rowsum(h.view(Bc, Nc, 512), A.view(Bc, Nc, 1)).view(Bc, 512)
        dM = torch.randn((Bc, 512), generator=g, device=dev)
        dA = ops.rows_dot(h.view(Bc, Nc, 512), dM.view(Bc, 1, 512)).view(Bc, Nc); ds = ops.softmax_rows_bwd(A, dA).view(-1)
        ref = ops.gated_score_bwd(U_nat.contiguous(), wc, ds, da, db)
        one = ops.gated_score_bwd_il(U_il, wc, da, db, h=h, dM=dM, Mp=Mp, A=A.view(-1), rows_per_bag=Nc)
        note("gate_bwd.dU", rel(one[0].view(M, 16, 2, 16).permute(0, 2, 1, 3).reshape(M, 512).float(), ref[0].float()), 3e-2)
        note("gate_bwd.dwc", rel(one[1], ref[1]), 5e-3); note("gate_bwd.dbab", rel(one[3], ref[3]), 5e-3)
    # ---- panel forward with dropout inside
    Mx = int(rng.choice([32, 96, 1024, 8192 + 64]))
    x = torch.relu(torch.randn((Mx, 512), generator=g, device=dev)).bfloat16()
    w = (torch.randn((512, 512), generator=g, device=dev) / math.sqrt(512)).bfloat16(); b = torch.randn((512,), generator=g, device=dev) * 0.1
    dp = ops.DropSeed(0.75, seed=int(rng.integers(1, 1 << 40)))
    hh, bits, _ = ops.panel_gemm(x, w, ops.PG_BIAS_RELU, bias=b, want_bitmask=True, drop=dp)
    h2, _, _ = ops.panel_gemm(x, w, ops.PG_BIAS_RELU, bias=b, want_bitmask=True)
    bits2 = ops.dropout_relu_bitmask(h2, dp)
    assert torch.equal(bits, bits2) and torch.equal(hh != 0, h2 != 0)
    note("fc_drop.h", rel(hh, h2), 2 ** -7)
    # ---- instance branch
    k = int(rng.integers(1, 9)); n_cls = int(rng.integers(2, 5)); sub = bool(rng.integers(0, 2)); Bi = int(rng.integers(1, 6)); Ni = int(rng.choice([64, 300, 2048]))
    hi = torch.relu(torch.randn((Bi * Ni, 512), generator=g, device=dev))
    Ai = torch.softmax(torch.randn((Bi, Ni), generator=g, device=dev), 1)
    ids = ops.topk_ids(Ai, k); lab = torch.from_numpy(rng.integers(0, n_cls, Bi)).to(dev)
    Wi = torch.randn((2 * n_cls, 512), generator=g, device=dev) * 0.05; bi = torch.randn((2 * n_cls,), generator=g, device=dev) * 0.1
    loss, dl, pt = ops.clam_inst_fwd(hi, ids, lab, Wi, bi, Bi, Ni, k, n_cls, sub)
    ref_loss = torch.zeros(Bi, dtype=torch.float64)
    for bq in range(Bi):
        rows = hi[bq * Ni + ids[bq].long()].double(); lg = rows @ Wi.double().t() + bi.double()
        for c in range(n_cls):
            l2 = lg[:, 2 * c:2 * c + 2]
            if c == int(lab[bq]):
                tg = torch.tensor([1] * k + [0] * k, device=dev)
                ref_loss[bq] += torch.nn.functional.cross_entropy(l2, tg).item()
            elif sub:
                ref_loss[bq] += torch.nn.functional.cross_entropy(l2[:k], torch.zeros(k, dtype=torch.long, device=dev)).item()
    ref_loss *= (1.0 / n_cls if sub else 1.0)
    note("inst.loss", rel(loss.cpu(), ref_loss), 2e-4)
print("iterations", n, {k: float("%.2e" % v) for k, v in sorted(worst.items())})
