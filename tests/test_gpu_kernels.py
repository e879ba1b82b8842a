"""Kernel-level parity on a real MI355X: every C-ABI entry point against CPU math / the oracle.

fp32 kernels (exact-f32 MFMA) are held to 1e-4 relative (BASELINE.json north_star); bf16 kernels
are compared with the same math evaluated on the bf16-rounded inputs (tolerance stated per test).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, mil_oracle as O  # noqa: E402


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rand(seed, name, shape, scale=1.0):
    return torch.from_numpy(detrand.normal(seed, name, shape) * np.float32(scale))


def _close(got, want, rtol, atol, msg=""):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), f"{msg}: {int(bad.sum())}/{bad.numel()} off, max err {err.max():.3e} (ref max {want.abs().max():.3e})"


def _rel_fro(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).norm() / want.norm().clamp_min(1e-30)).item()


# ------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (1000, 130, 128), (64, 3072, 1024), (333, 12, 512), (4096, 512, 512)])
def test_gemm_nt_plain_and_bias_relu(dtype, M, N, K):
    from murcl_amd import ops
    dev = _dev()
    A = _rand(1, f"A{M}{K}", (M, K)).to(dtype)
    B = _rand(1, f"B{N}{K}", (N, K), 1 / math.sqrt(K)).to(dtype)
    bias = _rand(1, f"b{N}", (N,))
    ref = A.double() @ B.double().t()
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    C = ops.gemm_nt(A.to(dev), B.to(dev), out_dtype=torch.float32)
    _close(C, ref, msg="plain", **tol)
    C2, ws = ops.gemm_nt(A.to(dev), B.to(dev), epi=ops.EPI_BIAS_RELU, bias=bias.to(dev), colsum=True)
    ref2 = torch.relu(ref + bias.double())
    _close(C2.float(), ref2, msg="bias_relu", **tol)
    # per-tile column sums add up to the column sums of the (rounded) output
    _close(ws.sum(0), C2.double().sum(0).float(), rtol=1e-3, atol=1e-2 * math.sqrt(M), msg="colsum")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nt_mask_and_rank1(dtype):
    from murcl_amd import ops
    dev = _dev()
    bags, n, K, N = 3, 200, 128, 512
    M = bags * n
    A = _rand(2, "A", (M, K)).to(dtype)
    B = _rand(2, "B", (N, K), 0.1).to(dtype)
    Hm = _rand(2, "H", (M, N)).to(dtype)
    a = torch.from_numpy(detrand.uniform(2, "a", (M,)))
    dM = _rand(2, "dM", (bags, N))
    base = A.double() @ B.double().t()
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    C = ops.gemm_nt(A.to(dev), B.to(dev), epi=ops.EPI_MASK, mask=Hm.to(dev), out_dtype=torch.float32)
    _close(C, base * (Hm.double() > 0), msg="mask", **tol)
    C, ws = ops.gemm_nt(A.to(dev), B.to(dev), epi=ops.EPI_RANK1_MASK, mask=Hm.to(dev), rowscale=a.to(dev),
                        rank1=dM.to(dev), rows_per_bag=n, colsum=True)
    ref = (base + a.double()[:, None] * dM.double().repeat_interleave(n, 0)) * (Hm.double() > 0)
    _close(C.float(), ref, msg="rank1_mask", **tol)
    _close(ws.sum(0), C.double().sum(0).float(), rtol=1e-3, atol=0.3, msg="colsum")


def test_gemm_nt_accumulate():
    from murcl_amd import ops
    dev = _dev()
    A, B = _rand(3, "A", (128, 512)), _rand(3, "B", (384, 512), 0.05)
    C0 = _rand(3, "C", (128, 384))
    C = C0.to(dev).clone()
    ops.gemm_nt(A.to(dev), B.to(dev), out=C, accumulate=True)
    _close(C, C0.double() + A.double() @ B.double().t(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(64, 512, 512), (128, 3072, 512), (320, 2048, 512), (33, 40, 16), (65, 100, 272), (1, 8, 1024),
                                   (128, 128, 1024), (128, 512, 3072), (64, 512, 2048), (1024, 96, 1536), (17, 2048, 144)])
@pytest.mark.parametrize("epi", ["none", "bias", "bias_relu"])
def test_gemm_nt_bag_level_f32_forms(M, N, K, epi):
    """The LDS-staged 32 x 32-tile form of the bag-level f32 layers (M <= 1024): ragged M / N tiles, partial 256-k chunks,
    one / two / many chunks (ping-pong buffers), and the K-split atomics path of long reductions with few tiles."""
    from murcl_amd import ops
    dev = _dev()
    A = _rand(11, f"A{M}{K}", (M, K))
    B = _rand(11, f"B{N}{K}", (N, K), 1 / math.sqrt(K))
    bias = _rand(11, f"b{N}", (N,))
    ref = A.double() @ B.double().t()
    if epi == "none":
        C = ops.gemm_nt(A.to(dev), B.to(dev))
    elif epi == "bias":
        C, ref = ops.gemm_nt(A.to(dev), B.to(dev), epi=ops.EPI_BIAS, bias=bias.to(dev)), ref + bias.double()
    else:
        C, ref = ops.gemm_nt(A.to(dev), B.to(dev), epi=ops.EPI_BIAS_RELU, bias=bias.to(dev)), torch.relu(ref + bias.double())
    assert C.shape == (M, N) and C.dtype == torch.float32
    _close(C, ref, rtol=2e-5, atol=2e-5, msg=f"{epi} {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(128, 384, 512), (128, 512, 3072), (40, 72, 2048)])
def test_gemm_nt_bag_level_accumulate_and_row_views(M, N, K):
    """accumulate=True adds onto C on the single-writer AND the K-split path; operands may be row blocks of wider tensors
    (leading dimension > K) as the GRU / head code passes them."""
    from murcl_amd import ops
    dev = _dev()
    Aw = _rand(12, f"A{M}{K}", (M, K + 64)).to(dev)
    Bw = _rand(12, f"B{N}{K}", (N + 3, K), 1 / math.sqrt(K)).to(dev)
    A, B = Aw[:, :K], Bw[1:N + 1]
    C0 = _rand(12, f"C{M}{N}", (M, N))
    C = C0.to(dev).clone()
    ops.gemm_nt(A, B, out=C, accumulate=True)
    _close(C, C0.double() + A.double().cpu() @ B.double().cpu().t(), rtol=2e-5, atol=2e-5)


def test_small_f32_gemm_dispatch_fuzz():
    """The bag-level f32 dispatcher picks among the 32 x 32 tile kernel (one or two chunk buffers), the 16 x 16 K-split-in-workgroup
    kernel and their epilogues by shape: 160 seeded random (M, N, K, epilogue, accumulate) draws over the regimes
    around every switch-over point (tile counts 64 / 96 / 384, 6+ chunks, N not a multiple of 16, ragged K) against f64."""
    from murcl_amd import ops
    dev = _dev()
    rng = np.random.default_rng(20260401)
    for it in range(160):
        M = int(rng.choice([1, 7, 16, 33, 64, 100, 128, 320, 500, 768, 1024]))
        N = int(rng.choice([16, 40, 128, 136, 512, 1536, 2048, 3072]))
        K = int(rng.choice([32, 96, 256, 288, 512, 1024, 1536, 2048, 3072]))
        epi = int(rng.choice([0, 1, 2, 3]))                           # none, bias, bias + ReLU, ReLU' mask
        acc = bool(rng.integers(2)) and epi != 2
        g = torch.Generator().manual_seed(1000 + it)
        Ah, Bh = torch.randn((M, K), generator=g), torch.randn((N, K), generator=g) / math.sqrt(K)
        A, B = Ah.to(dev), Bh.to(dev)
        bias, y = torch.randn((N,), generator=g), torch.randn((M, N), generator=g)
        C0 = torch.randn((M, N), generator=g)
        ref = Ah.double() @ Bh.double().t()
        kw = {}
        if epi in (1, 2):
            kw, ref = dict(epi=ops.EPI_BIAS if epi == 1 else ops.EPI_BIAS_RELU, bias=bias.to(dev)), ref + bias.double()
        if epi == 2:
            ref = torch.relu(ref)
        if epi == 3:
            kw, ref = dict(epi=ops.EPI_MASK, mask=y.to(dev)), ref * (y > 0)
        if acc:
            ref = ref + C0.double()                                   # accumulation adds the finished epilogue onto C
        out = C0.to(dev).clone() if acc else None
        C = ops.gemm_nt(A, B, out=out, accumulate=acc, **kw)
        _close(C, ref, rtol=3e-5, atol=3e-5, msg=f"draw {it}: {M}x{N}x{K} epi {epi} acc {acc}")


def test_small_f32_weight_gradient_group_fuzz():
    """ops.gemm_tn / gemm_tn_grouped on bag-level f32 products: 60 seeded random groups of 1-4 products (M up to 512 with ragged 16-row
    tails, widths that are not multiples of the 32-wide tile, outputs given - added to - or fresh - written -, with and without the
    bias-gradient column sums) against f64; rows beyond one pass through LDS take the multi-pass form."""
    from murcl_amd import ops
    dev = _dev()
    rng = np.random.default_rng(20260402)
    for it in range(60):
        n = int(rng.integers(1, 5))
        g = torch.Generator().manual_seed(5000 + it)
        probs, want = [], []
        for k in range(n):
            M = int(rng.choice([1, 5, 16, 77, 128, 200, 320, 500, 512]))
            N1, N2 = int(rng.choice([4, 36, 64, 128, 520, 1536])), int(rng.choice([8, 32, 100, 512, 2048]))
            A, B = torch.randn((M, N1), generator=g), torch.randn((M, N2), generator=g)
            given, cs = bool(rng.integers(2)), bool(rng.integers(2))
            C0, c0 = torch.randn((N1, N2), generator=g), torch.randn((N1,), generator=g)
            out = C0.to(dev).clone() if given else None
            ci = c0.to(dev).clone() if (cs and given) else None       # column sums ride with a given output (the direct-gradient mode)
            probs.append((A.to(dev), B.to(dev), out, ci, None))
            want.append(((C0.double() if given else 0) + A.double().t() @ B.double(), (c0.double() + A.double().sum(0)) if ci is not None else None))
        if n == 1:
            A, B, out, ci, _ = probs[0]
            Cs = [ops.gemm_tn(A, B, out=out, colsum_into=ci)]
        else:
            Cs = ops.gemm_tn_grouped(probs)
        for (A, B, out, ci, _), C, (wC, wc) in zip(probs, Cs, want):
            _close(C, wC, rtol=1e-4, atol=3e-4, msg=f"draw {it}: {tuple(A.shape)} x {tuple(B.shape)}")
            if wc is not None:
                _close(ci, wc, rtol=1e-4, atol=3e-4, msg=f"draw {it}: column sums")


# ------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N1,N2", [(4096, 512, 512), (1000, 128, 512), (130, 3072, 512), (77, 16, 1024), (20000, 512, 128),
                                     (8192 + 37, 512, 512), (4100, 256, 128), (65536, 512, 512)])
def test_gemm_tn(dtype, M, N1, N2):
    from murcl_amd import ops
    dev = _dev()
    A = _rand(4, f"A{M}{N1}", (M, N1)).to(dtype)
    B = _rand(4, f"B{M}{N2}", (M, N2)).to(dtype)
    ref = A.double().t() @ B.double()
    C = ops.gemm_tn(A.to(dev), B.to(dev))
    rel = _rel_fro(C, ref)
    assert rel < (2e-6 if dtype == torch.float32 else 2e-3), rel
    s = math.sqrt(M)
    _close(C, ref, rtol=1e-4, atol=1e-4 * s if dtype == torch.float32 else 2e-2 * s, msg="tn")


@pytest.mark.parametrize("M,N,K,epi", [(16384, 128, 1024, "bias"), (4096 + 64, 200, 512, "none"), (5000, 512, 256, "relu")])
def test_gemm_nt_f32_as_three_term_bf16_split(M, N, K, epi):
    """dtype code 2 (include/murcl_amd.h): f32 operands, six bf16 MFMAs per product.  Against float64 it must be as good as
    the exact-f32 MFMA path (both ~1e-7 relative), so the 1e-4 parity contract of the f32 mode holds with it."""
    from murcl_amd import ops
    dev = _dev()
    A = _rand(8, f"A{M}{K}", (M, K))
    B = _rand(8, f"B{N}{K}", (N, K))
    bias = _rand(8, f"b{N}", (N,))
    e = {"none": ops.EPI_NONE, "bias": ops.EPI_BIAS, "relu": ops.EPI_BIAS_RELU}[epi]
    ref = A.double() @ B.double().t() + (bias.double() if epi != "none" else 0.0)
    if epi == "relu":
        ref = ref.clamp_min(0.0)
    kw = dict(epi=e, bias=bias.to(dev) if epi != "none" else None)
    exact = ops.gemm_nt(A.to(dev), B.to(dev), **kw)
    split = ops.gemm_nt(A.to(dev), B.to(dev), x3=True, **kw)
    r_exact, r_split = _rel_fro(exact, ref), _rel_fro(split, ref)
    assert r_split < 2e-6 and r_split < 3 * r_exact + 1e-8, (r_exact, r_split)
    _close(split, ref, rtol=2e-5, atol=2e-5 * math.sqrt(K), msg="x3")


@pytest.mark.parametrize("M,N1,N2,cs", [(131072, 128, 1024, False), (20000 + 17, 128, 512, True), (8192, 256, 256, True)])
def test_gemm_tn_f32_as_three_term_bf16_split(M, N1, N2, cs):
    from murcl_amd import ops
    dev = _dev()
    A = _rand(9, f"A{M}{N1}", (M, N1))
    B = _rand(9, f"B{M}{N2}", (M, N2))
    ref = A.double().t() @ B.double()
    colsum = torch.zeros((N1,), device=dev) if cs else None
    exact = ops.gemm_tn(A.to(dev), B.to(dev))
    split = ops.gemm_tn(A.to(dev), B.to(dev), x3=True, colsum_into=colsum)
    r_exact, r_split = _rel_fro(exact, ref), _rel_fro(split, ref)
    assert r_split < 2e-6 and r_split < 3 * r_exact + 1e-8, (r_exact, r_split)
    if cs:
        _close(colsum, A.double().sum(0), rtol=1e-4, atol=1e-4 * math.sqrt(M), msg="colsum")


@pytest.mark.parametrize("dtype,M,N1,N2", [(torch.float32, 128, 3072, 512), (torch.float32, 128, 512, 512), (torch.float32, 77, 132, 260),
                                           (torch.float32, 320, 10, 512), (torch.bfloat16, 4096, 128, 512), (torch.bfloat16, 8192, 512, 512),
                                           (torch.bfloat16, 1000, 136, 72)])
def test_gemm_tn_adds_the_bias_gradient_in_the_same_launch(dtype, M, N1, N2):
    """colsum_into: the column sums of A (the bias gradient of the layer) from the pass that forms the weight gradient -
    small and ragged shapes, several M splits, the wide-kernel dispatch (separate pass there) and the padded-N1 path."""
    from murcl_amd import ops
    dev = _dev()
    A = _rand(6, f"A{M}{N1}", (M, N1)).to(dtype)
    B = _rand(6, f"B{M}{N2}", (M, N2)).to(dtype)
    pre = _rand(6, f"p{N1}", (N1,))
    cs = pre.clone().to(dev)
    W0 = _rand(6, f"w{N1}{N2}", (N1, N2))
    out = W0.clone().to(dev)
    r = ops.gemm_tn(A.to(dev), B.to(dev), out=out, colsum_into=cs)
    assert r.data_ptr() == out.data_ptr()
    s = math.sqrt(M)
    f32 = dtype == torch.float32
    _close(out, W0.double() + A.double().t() @ B.double(), rtol=1e-4, atol=1e-4 * s if f32 else 2e-2 * s, msg="dW")
    _close(cs, pre.double() + A.double().sum(0), rtol=1e-4, atol=1e-4 * s if f32 else 1e-3 * s, msg="db")     # ADDED, not overwritten


@pytest.mark.parametrize("M,N1,N2", [(1, 4, 4), (15, 36, 100), (128, 1024, 128), (320, 2048, 512), (500, 60, 3072), (512, 512, 512),
                                     (513, 64, 64), (768, 3072, 512)])
def test_gemm_tn_bag_level_f32_forms(M, N1, N2):
    """The 32 x 32 single-writer form (whole reduction in LDS, M <= 512) and its boundary with the ring kernel: ragged 16-row
    tail groups, ragged column tiles, the bias gradient from the ones-fragment, accumulation onto existing gradients."""
    from murcl_amd import ops
    dev = _dev()
    A = _rand(13, f"A{M}{N1}", (M, N1))
    B = _rand(13, f"B{M}{N2}", (M, N2))
    W0, b0 = _rand(13, f"w{N1}{N2}", (N1, N2)), _rand(13, f"c{N1}", (N1,))
    out, cs = W0.clone().to(dev), b0.clone().to(dev)
    ops.gemm_tn(A.to(dev), B.to(dev), out=out, colsum_into=cs)
    s = math.sqrt(M)
    _close(out, W0.double() + A.double().t() @ B.double(), rtol=2e-5, atol=2e-5 * s, msg="dW")
    _close(cs, b0.double() + A.double().sum(0), rtol=2e-5, atol=2e-5 * s, msg="db")
    out2 = ops.gemm_tn(A.to(dev), B.to(dev))
    _close(out2, A.double().t() @ B.double(), rtol=2e-5, atol=2e-5 * s, msg="fresh")


@pytest.mark.parametrize("M", [64, 8192, 65536])
def test_panel_dgrad_partial_bias_rows_fold_into_the_weight_gradient(M):
    """panel_gemm(colsum_defer=True) leaves [R, N] partial column sums; gemm_tn(colsum_parts=...) adds them to the bias
    gradient - inside the reduce launch of the workspace path (M >= 16384), by a small launch of its own otherwise."""
    from murcl_amd import ops
    dev = _dev()
    N = 512
    dY = _rand(14, f"dY{M}", (M, N), 0.5).bfloat16().to(dev)
    Wt = _rand(14, "Wt", (N, N), 1 / math.sqrt(N)).bfloat16().to(dev)
    H = _rand(14, f"H{M}", (M, N)).to(dev)
    X = _rand(14, f"X{M}", (M, N), 0.5).bfloat16().to(dev)
    bits = ops.relu_bitmask(H)
    dZ, _, parts = ops.panel_gemm(dY, Wt, ops.PG_MASK, bitmask=bits, colsum=True, colsum_defer=True)
    dZr, _, cs_ref = ops.panel_gemm(dY, Wt, ops.PG_MASK, bitmask=bits, colsum=True)
    assert torch.equal(dZ, dZr) and parts[1] >= 1 and parts[0].numel() >= parts[1] * N
    g0, b0 = _rand(14, "g0", (N, N)), _rand(14, "b0", (N,))
    gw, gb = g0.clone().to(dev), b0.clone().to(dev)
    ops.gemm_tn(dZ, X, out=gw, colsum_into=gb, colsum_parts=parts)
    gw2 = g0.clone().to(dev)
    ops.gemm_tn(dZ, X, out=gw2)
    _close(gw, gw2.double().cpu(), rtol=1e-6, atol=1e-4 * math.sqrt(M), msg="dW unchanged by the fold")
    _close(gb, b0.double() + cs_ref.double().cpu(), rtol=1e-5, atol=1e-4 * math.sqrt(M), msg="db")


@pytest.mark.parametrize("shapes", [[(65536, 512, 512)] * 3, [(16384 + 37, 512, 512), (40000, 256, 512), (16384, 512, 256)],
                                    [(32768, 512, 512), (32768, 512, 512)], [(20000, 512, 512)] * 4,
                                    [(16384, 512, 512), (16384, 128, 512)]])
def test_grouped_weight_gradients_equal_the_single_launches(shapes):
    """murcl_gemm_tn_grouped (one round of workgroups for several products, one reduce launch) against float64 A^T B and against
    ops.gemm_tn per product: accumulation into existing gradients, the bias gradient from partial rows (one product) and from
    the column sums of A (another), ragged row counts, unequal shapes; an ineligible member sends the group down the single path."""
    from murcl_amd import ops
    dev = _dev()
    probs, refs = [], []
    for g, (M, N1, N2) in enumerate(shapes):
        A = _rand(21, f"A{g}{M}{N1}", (M, N1), 0.5).bfloat16()
        B = _rand(21, f"B{g}{M}{N2}", (M, N2)).bfloat16()
        W0, b0 = _rand(21, f"w{g}", (N1, N2)), _rand(21, f"b{g}", (N1,))
        parts_rows = _rand(21, f"p{g}", (37, N1)) if g == 0 else None
        out = W0.clone().to(dev) if g != 1 else None
        cs = b0.clone().to(dev) if g in (0, 2) else None
        parts = (parts_rows.to(dev), 37) if g == 0 else None
        probs.append((A.to(dev), B.to(dev), out, cs, parts))
        dW = A.double().t() @ B.double() + (W0.double() if g != 1 else 0.0)
        db = None
        if g == 0:
            db = b0.double() + parts_rows.double().sum(0)
        elif g == 2:
            db = b0.double() + A.double().sum(0)
        refs.append((dW, db, M))
    Cs = ops.gemm_tn_grouped(probs)
    for g, ((dW, db, M), C, pr) in enumerate(zip(refs, Cs, probs)):
        s = math.sqrt(M)
        assert _rel_fro(C, dW) < 2e-3, (g, _rel_fro(C, dW))
        _close(C, dW, rtol=1e-4, atol=2e-2 * s, msg=f"dW{g}")
        if pr[2] is not None:
            assert C.data_ptr() == pr[2].data_ptr()
        if db is not None:
            _close(pr[3], db, rtol=1e-4, atol=1e-3 * s, msg=f"db{g}")
        # the same product alone: same tiles and MFMA order inside a split, another split count -> f32 summation order only
        single = ops.gemm_tn(pr[0], pr[1])
        base = C.double().cpu() - (dW - pr[0].double().cpu().t() @ pr[1].double().cpu())
        assert _rel_fro(base, single.double().cpu()) < 1e-5


def test_grouped_weight_gradients_write_scale_and_deinterleave_in_the_reduce_launch():
    """murcl_tn_problem.flags: products written into uninitialised tensors (no zero-fill launch), a factor applied to product and
    column sums (CLAM's Dropout scale), and the 16-row a/b interleave of the gate pair undone on the way out (clam.py:69-72 backward):
    against float64 and against the plain grouped call followed by the tensor ops the flags replace."""
    from murcl_amd import ops
    dev = _dev()
    M, D, L = 32768, 256, 512
    dU = _rand(22, "dU", (M, 2 * D), 0.5).bfloat16().to(dev)
    h = _rand(22, "h", (M, L)).bfloat16().to(dev)
    dz = _rand(22, "dz", (M, L), 0.5).bfloat16().to(dev)
    x = _rand(22, "x", (M, 512)).bfloat16().to(dev)
    parts = _rand(22, "parts", (41, L)).to(dev)
    plain = ops.gemm_tn_grouped([(dU, h, None, None, None), (dz, x, None, None, None)])
    ref_ab = plain[0].view(D // 16, 2, 16, L).permute(1, 0, 2, 3).reshape(2 * D, L)
    db = torch.full((L,), float("nan"), device=dev)                       # must be overwritten, never read
    got = ops.gemm_tn_grouped([(dU, h, None, None, None, {"deinterleave": True}),
                               (dz, x, None, db, (parts, 41), {"scale": 1.0 / 0.75, "overwrite": True})], fresh=True)
    assert torch.equal(got[0], ref_ab)
    _close(got[1], (plain[1] / 0.75).cpu(), rtol=1e-6, atol=0, msg="scaled dW")
    _close(db, (parts.double().sum(0) / 0.75).cpu(), rtol=1e-5, atol=1e-5, msg="scaled db")
    s = math.sqrt(M)
    _close(got[1], dz.double().cpu().t() @ x.double().cpu() / 0.75, rtol=1e-4, atol=2e-2 * s, msg="dW vs f64")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(32, 128), (4096, 512), (96, 1024)])
def test_dropout_applied_in_place_with_the_mask_never_materialised(dtype, M, N):
    """murcl_dropout_relu_bitmask: x *= keep with the counter-based mask of murcl_dropout_mask for the same seed (bit-identical to
    multiplying by the materialised mask), and the panel layout's 1-bit mask of the surviving positive entries from the same pass."""
    from murcl_amd import ops
    dev = _dev()
    x0 = _rand(31, f"x{M}{N}", (M, N)).to(dtype).to(dev)
    drop = ops.DropSeed(0.75, seed=0x1234_5678_9ABC_DEF0 + M)
    mask = ops.dropout_mask((M, N), torch.float32, 0.75, dev, seed=drop.seed)      # the same keep pattern, scale 1/0.75 in f32
    assert torch.equal(mask != 0, ops.dropout_mask((M, N), dtype, 0.75, dev, seed=drop.seed) != 0)
    kept = float((mask != 0).float().mean())
    assert abs(kept - 0.75) < 0.05 and set(mask.unique().tolist()) <= {0.0, float(torch.tensor(1 / 0.75, dtype=torch.float32))}
    want = (x0.float() * mask).to(dtype)
    x = x0.clone()
    bits = ops.dropout_relu_bitmask(x, drop)
    assert torch.equal(x, want)
    assert torch.equal(bits, ops.relu_bitmask(want))
    x2 = x0.clone()
    assert ops.dropout_relu_bitmask(x2, drop, want_bits=False) is None and torch.equal(x2, want)


@pytest.mark.parametrize("gated", [True, False])
def test_gate_dropout_masks_generated_inside_the_score_kernels(gated):
    """gated_score_fwd / _bwd with DropSeed specs (masks never materialised) == the same kernels fed the materialised counter-based
    masks of the same seeds (f32: bit-identical keep values)."""
    from murcl_amd import ops
    dev = _dev()
    M, D = 4096, 256
    U = _rand(33, f"U{gated}", (M, 2 * D if gated else D)).to(dev)
    wc, bc = _rand(33, "wc", (D,)).to(dev), torch.zeros(1, device=dev)
    ds = _rand(33, "ds", (M,)).to(dev)
    da, db = ops.DropSeed(0.75, seed=1234567), (ops.DropSeed(0.75, seed=7654321) if gated else None)
    ka = ops.dropout_mask((M, D), torch.float32, 0.75, dev, seed=da.seed)
    kb = ops.dropout_mask((M, D), torch.float32, 0.75, dev, seed=db.seed) if gated else None
    s_seed = ops.gated_score_fwd(U, wc, bc, da, db, gated=gated)
    s_mask = ops.gated_score_fwd(U, wc, bc, ka, kb, gated=gated)
    assert torch.equal(s_seed, s_mask)
    assert not torch.equal(s_seed, ops.gated_score_fwd(U, wc, bc, gated=gated))
    outs_seed = ops.gated_score_bwd(U, wc, ds, da, db, gated=gated)
    outs_mask = ops.gated_score_bwd(U, wc, ds, ka, kb, gated=gated)
    assert torch.equal(outs_seed[0], outs_mask[0])
    for a, b in zip(outs_seed[1:], outs_mask[1:]):
        _close(a, b, rtol=1e-6, atol=1e-5)


def _gate_inputs(seed, M, D=256, L=512):
    h = torch.relu(_rand(seed, "h", (M, L))).bfloat16()
    wa, wb = _rand(seed, "wa", (D, L), 1.5 / math.sqrt(L)), _rand(seed, "wb", (D, L), 1.5 / math.sqrt(L))
    ba, bb = _rand(seed, "ba", (D,), 0.1), _rand(seed, "bb", (D,), 0.1)
    wc, bc = _rand(seed, "wc", (D,), 2.0 / math.sqrt(D)), _rand(seed, "bc", (1,), 0.1)
    return h, wa, ba, wb, bb, wc, bc


def _deinterleave(U_il, D):
    """[M, 2D] in the panel GEMM's interleaved gate order -> (a | b) natural order."""
    M = U_il.shape[0]
    return U_il.view(M, D // 16, 2, 16).permute(0, 2, 1, 3).reshape(M, 2 * D)


@pytest.mark.parametrize("D,L,d", [(256, 512, 512), (128, 512, 1024), (64, 256, 320)])
def test_clam_parameter_views_from_one_launch(D, L, d):
    """ops.clam_views: the bf16 copy of fc, attention_a / attention_b interleaved in 16-row blocks, the transpose of that, and the
    interleaved bias / attention_c vectors - one murcl_cast_batch launch of row / column BLOCK jobs - equal what cat + gather +
    cast + transpose produce; they follow the parameters (rebuilt on the next call for parameters no optimizer manages)."""
    from murcl_amd import ops
    dev = _dev()
    w1 = _rand(47, "w1", (L, d)).to(dev)
    wa, wb = _rand(47, "wa", (D, L)).to(dev), _rand(47, "wb", (D, L)).to(dev)
    ba, bb, wc = _rand(47, "ba", (D,)).to(dev), _rand(47, "bb", (D,)).to(dev), _rand(47, "wc", (1, D)).to(dev)
    for _ in range(2):
        w1c, W_il, W_ilT, b_il, c_il = ops.clam_views(w1, wa, ba, wb, bb, wc, torch.bfloat16)
        W_ref, b_ref, c_ref = ops.gate_interleave(wa, ba, wb, bb, wc, torch.bfloat16)
        assert torch.equal(w1c, w1.bfloat16()) and torch.equal(W_il, W_ref) and torch.equal(W_ilT, W_ref.t().contiguous())
        assert torch.equal(b_il, b_ref) and torch.equal(c_il, c_ref)
        with torch.no_grad():                                   # second round: changed parameters
            for p in (w1, wa, wb, ba, bb, wc):
                p.mul_(1.5).add_(0.01)


@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("M", [64, 4096, 8192 + 32])
def test_panel_gate_u_scores_and_pre_activations_from_one_gemm(M, drop):
    """murcl_panel_gemm_drop epilogue 5 (CLAM's training forward): the pre-activations it stores, de-interleaved, are the bias
    GEMM's (bit for bit: same MFMA chain, same rounding), and its scores are those of murcl_gated_score_fwd on the un-rounded
    accumulators - compared with the score pass over the bf16 U (tolerance = what that rounding is worth) and, tighter, with a
    float64 evaluation of the bf16 inputs; with seeded gate Dropout the same masks as the score kernel's."""
    from murcl_amd import ops
    dev = _dev()
    h, wa, ba, wb, bb, wc, bc = [t.to(dev) for t in _gate_inputs(41, M)]
    D = 256
    da, db = (ops.DropSeed(0.75, seed=991), ops.DropSeed(0.75, seed=17)) if drop else (None, None)
    W_il, b_il, c_il = ops.gate_interleave(wa, ba, wb, bb, wc, torch.bfloat16)
    U_il, s = ops.panel_gate_u(h, W_il, b_il, c_il, bc, da, db)
    U_ref, _, _ = ops.panel_gemm(h, torch.cat([wa, wb], 0).bfloat16(), ops.PG_BIAS, bias=torch.cat([ba, bb], 0))
    assert torch.equal(_deinterleave(U_il, D), U_ref)
    s_chain = ops.gated_score_fwd(U_ref, wc, bc, da, db)
    sc = s_chain.abs().max().item()
    assert (s - s_chain).abs().max().item() <= 2e-2 * sc
    # float64 reference on the bf16-rounded operands
    Ud = h.double() @ torch.cat([wa, wb], 0).bfloat16().double().t() + torch.cat([ba, bb], 0).double()
    t = torch.tanh(Ud[:, :D]) * torch.sigmoid(Ud[:, D:])
    if drop:
        ka = ops.dropout_mask((M, D), torch.float32, 0.75, dev, seed=da.seed).double()
        kb = ops.dropout_mask((M, D), torch.float32, 0.75, dev, seed=db.seed).double()
        t = t * ka * kb
    s64 = (t * wc.double()).sum(1) + bc.double()
    e_new, e_chain = (s.double() - s64).abs().max().item(), (s_chain.double() - s64).abs().max().item()
    assert e_new <= 3e-3 * sc and e_new <= 1.5 * e_chain + 1e-4 * sc, (e_new, e_chain)
    if not drop:
        assert torch.equal(ops.panel_gate_score(h, W_il, b_il, c_il, bc), s)       # the forward-only epilogue: the same scores


@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("B,N", [(2, 64), (3, 2048), (5, 96), (1100, 64), (1536, 40)])      # > 1024 bags: a workgroup walks several bags
def test_gate_backward_in_one_pass_over_interleaved_pre_activations(B, N, drop):
    """murcl_gated_score_bwd_il: (a) with ds given = murcl_gated_score_bwd on the de-interleaved tensors, bit for bit; (b) with ds
    derived in the pass (ds_n = A_n (h_n . dM - M . dM)) = rows_dot -> softmax_rows_bwd -> gated_score_bwd, to f32 rounding."""
    from murcl_amd import ops
    dev = _dev()
    M, D, L = B * N, 256, 512
    h, wa, ba, wb, bb, wc, bc = [t.to(dev) for t in _gate_inputs(43, M)]
    da, db = (ops.DropSeed(0.75, seed=5), ops.DropSeed(0.75, seed=6)) if drop else (None, None)
    W_il, b_il, c_il = ops.gate_interleave(wa, ba, wb, bb, wc, torch.bfloat16)
    U_il, s = ops.panel_gate_u(h, W_il, b_il, c_il, bc, da, db)
    U_nat = _deinterleave(U_il, D).contiguous()
    A = ops.softmax_rows(s.view(B, N))
    Mp = ops.weighted_rowsum(h.view(B, N, L), A.view(B, N, 1)).view(B, L)
    dM = _rand(43, "dM", (B, L)).to(dev)
    dA = ops.rows_dot(h.view(B, N, L), dM.view(B, 1, L)).view(B, N)
    ds = ops.softmax_rows_bwd(A, dA).view(-1)
    ref = ops.gated_score_bwd(U_nat, wc, ds, da, db)
    got = ops.gated_score_bwd_il(U_il, wc, da, db, ds=ds)
    assert torch.equal(_deinterleave(got[0], D), ref[0])
    for a, b in zip(got[1:], ref[1:]):
        _close(a, b.double().cpu(), rtol=1e-5, atol=1e-5 * max(1.0, b.abs().max().item()))
    if not ops.gated_bwd_il_supported(M, D, L, N):
        return
    one = ops.gated_score_bwd_il(U_il, wc, da, db, h=h, dM=dM, Mp=Mp, A=A.view(-1), rows_per_bag=N)
    scale = ref[0].float().abs().max().item()
    assert (_deinterleave(one[0], D).float() - ref[0].float()).abs().max().item() <= 2e-2 * scale      # bf16 outputs of ds values that differ in the last f32 bits
    for a, b, name in zip(one[1:], ref[1:], ("dwc", "dbc", "dbab")):
        # (dbc = sum_n ds_n is zero in exact arithmetic - a soft-max gradient sums to nothing - so it is compared on the scale of its terms)
        floor = ds.abs().sum().item() if name == "dbc" else max(1e-6, b.abs().max().item())
        _close(a, b.double().cpu(), rtol=2e-3, atol=2e-3 * floor, msg=name)


@pytest.mark.parametrize("M", [32, 4096, 8192 + 64])
def test_panel_forward_with_dropout_inside_the_epilogue(M):
    """murcl_panel_gemm_drop epilogue 0 with keep_p: relu(x W^T + b) * keep from the seed's counter-based mask - the keep pattern and the
    1-bit mask of what survives are exactly those of murcl_dropout_relu_bitmask on the un-dropped output; the values differ from that
    two-pass form by one bf16 rounding (the epilogue scales the f32 accumulator and rounds once)."""
    from murcl_amd import ops
    dev = _dev()
    x = torch.relu(_rand(45, "x", (M, 512))).bfloat16().to(dev)
    w, b = _rand(45, "w", (512, 512), 1.5 / math.sqrt(512)).bfloat16().to(dev), _rand(45, "b", (512,), 0.1).to(dev)
    drop = ops.DropSeed(0.75, seed=0xABCDEF + M)
    h, bits, _ = ops.panel_gemm(x, w, ops.PG_BIAS_RELU, bias=b, want_bitmask=True, drop=drop)
    h2, _, _ = ops.panel_gemm(x, w, ops.PG_BIAS_RELU, bias=b, want_bitmask=True)
    bits2 = ops.dropout_relu_bitmask(h2, drop)
    assert torch.equal(bits, bits2)
    assert torch.equal(h != 0, h2 != 0)
    assert (h.float() - h2.float()).abs().max().item() <= 2 ** -7 * h2.float().abs().max().item()
    # one rounding: closer to (or as close as) the exact product than the two-pass form
    mask = ops.dropout_mask((M, 512), torch.float32, 0.75, dev, seed=drop.seed).double()
    exact = torch.relu(x.double() @ w.double().t() + b.double()) * mask
    assert (h.double() - exact).abs().max().item() <= (h2.double() - exact).abs().max().item() + 1e-6


# ------------------------------------------------------------------ K6 streaming passes (reassociated DSMIL)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,d,C", [(2, 64, 512, 2), (3, 1024, 1024, 2), (2, 96, 320, 1), (16, 8192, 1024, 2)])
def test_dsmil_attention_and_pooling_in_one_pass_and_their_backward_in_one_more(B, N, d, C, dtype):
    """murcl_dsmil_attn_pool (online soft-max per wave, merged per bag) against float64 soft-max_n(X v) and A^T X, and against the
    separate launches (rows_dot -> dsmil_softmax -> weighted_rowsum); murcl_dsmil_attn_pool_bwd (R = sum A dA X - (sum A dA) Z)
    against float64 sum_n A (dA - sum A dA) X and dcls^T X.  f32: 1e-4 of the largest entry (north_star); bf16 storage: same
    accumulation, so the same bound against float64 of the rounded X."""
    from murcl_amd import ops
    dev = _dev()
    X = torch.relu(_rand(51, f"X{N}{d}", (B, N, d))).to(dtype).to(dev)
    v = (_rand(51, "v", (B, C, d)) * (6.0 / math.sqrt(d))).to(dev)
    Xd, vd = X.double(), v.double()
    S = torch.einsum("bnd,bcd->bnc", Xd, vd)
    A64 = torch.softmax(S, 1)
    Z64 = torch.einsum("bnc,bnd->bcd", A64, Xd)
    one = ops.dsmil_attn_pool(X, v * 2.0, 0.5)
    sep_A = ops.dsmil_softmax_(ops.rows_dot(X, v))
    sep_Z = ops.weighted_rowsum(X, sep_A)
    if one is None:
        assert not ops._lib.lib().murcl_dsmil_stream_plan(B, N, d, C)
        return
    A, Z = one
    _close(A, A64, rtol=1e-4, atol=1e-4 * A64.max().item(), msg="A")
    _close(Z, Z64, rtol=1e-4, atol=1e-4 * Z64.abs().max().item(), msg="Z")
    _close(A, sep_A, rtol=1e-4, atol=1e-4 * A64.max().item(), msg="A vs separate")
    _close(Z, sep_Z, rtol=1e-4, atol=1e-4 * Z64.abs().max().item(), msg="Z vs separate")
    np.testing.assert_allclose(A.sum(1).cpu().numpy(), 1.0, rtol=1e-5)
    # backward
    dZ = _rand(51, "dZ", (B, C, d)).to(dev)
    dcls = _rand(51, "dcls", (B, N, C)).to(dev)
    dA64 = torch.einsum("bnd,bcd->bnc", Xd, dZ.double())
    dS64 = A64 * (dA64 - (A64 * dA64).sum(1, keepdim=True))
    R64 = torch.einsum("bnc,bnd->bcd", dS64, Xd) * 0.25
    dWc64 = torch.einsum("bnc,bnd->cd", dcls.double(), Xd)
    R, dWc = ops.dsmil_attn_pool_bwd(X, dZ, A, Z, dcls, 0.25)
    # R is a difference of two sums of the size of P = sum A dA X: the bound is on that scale (see DESIGN)
    P64 = torch.einsum("bnc,bnd->bcd", A64 * dA64, Xd).abs().max().item() * 0.25
    _close(R, R64, rtol=1e-4, atol=1e-4 * max(R64.abs().max().item(), 0.1 * P64), msg="R")
    _close(dWc, dWc64, rtol=1e-4, atol=1e-4 * dWc64.abs().max().item(), msg="dWc")
    R2, none = ops.dsmil_attn_pool_bwd(X, dZ, A, Z, None, 0.25)
    assert none is None and torch.equal(R2, R)


# ------------------------------------------------------------------ K2 attention pool
def _k2_inputs(seed, B, N):
    H = torch.relu(_rand(seed, "H", (B, N, 512)))
    H = H * torch.from_numpy(detrand.uniform(seed, "sig", (B, 1, 512), 0.1, 1.9))
    Wa = _rand(seed, "Wa", (128, 512), 2.0 / math.sqrt(512))
    ba = _rand(seed, "ba", (128,), 0.1)
    wb = _rand(seed, "wb", (1, 128), 4.0 / math.sqrt(128))
    bb = _rand(seed, "bb", (1,), 0.1)
    return H, Wa, ba, wb, bb


@pytest.mark.parametrize("B,N", [(4, 256), (2, 300), (3, 16), (1, 1), (5, 1000), (2, 5000), (64, 2048)])
def test_abmil_pool_fwd_f32(B, N):
    """fp32 path vs oracle: attention weights and pooled vector within 1e-4 (north_star)."""
    from murcl_amd import ops
    dev = _dev()
    H, Wa, ba, wb, bb = _k2_inputs(7, B, N)
    M_ref, A_ref, s_ref = O.abmil_attn_pool(H, Wa, ba, wb, bb)
    s, A, M, ml = ops.abmil_pool_fwd(H.to(dev), Wa.to(dev), ba.to(dev), wb.to(dev), bb.to(dev))
    _close(s, s_ref, rtol=1e-4, atol=1e-5, msg="scores")
    _close(A, A_ref, rtol=1e-4, atol=1e-9, msg="A")
    _close(M, M_ref, rtol=1e-4, atol=1e-6, msg="M")
    np.testing.assert_allclose(A.sum(1).cpu().numpy(), np.full(B, 1 / math.sqrt(N)), rtol=1e-5)


@pytest.mark.parametrize("B,N", [(4, 256), (2, 300), (64, 2048)])
def test_abmil_pool_fwd_bf16(B, N):
    """bf16 storage path vs the oracle evaluated on the bf16-rounded H and Wa.
    Tolerance: scores 2e-3 abs (fast tanh + accumulation order), A 1% rel, M 1% of max."""
    from murcl_amd import ops
    dev = _dev()
    H, Wa, ba, wb, bb = _k2_inputs(8, B, N)
    Hb, Wab = H.bfloat16(), Wa.bfloat16()
    M_ref, A_ref, s_ref = O.abmil_attn_pool(Hb.float(), Wab.float(), ba, wb, bb)
    s, A, M, ml = ops.abmil_pool_fwd(Hb.to(dev), Wab.to(dev), ba.to(dev), wb.to(dev), bb.to(dev))
    _close(s, s_ref, rtol=0, atol=2e-3, msg="scores")
    _close(A, A_ref, rtol=1e-2, atol=1e-9, msg="A")
    _close(M, M_ref, rtol=1e-2, atol=1e-2 * M_ref.abs().max().item(), msg="M")


def test_abmil_pool_fwd_online_softmax_rescale():
    """Force the running-max rescale: one late row with a much larger score than everything before
    it (cdna guide rule 26: a rare data-dependent branch needs its own test)."""
    from murcl_amd import ops
    dev = _dev()
    B, N = 2, 4096
    H, Wa, ba, wb, bb = _k2_inputs(9, B, N)
    wb = wb * 6.0
    H[0, 3000] *= 8.0
    H[1, 17] *= 8.0
    M_ref, A_ref, s_ref = O.abmil_attn_pool(H, Wa, ba, wb, bb)
    s, A, M, ml = ops.abmil_pool_fwd(H.to(dev), Wa.to(dev), ba.to(dev), wb.to(dev), bb.to(dev))
    _close(A, A_ref, rtol=2e-4, atol=1e-9, msg="A")
    _close(M, M_ref, rtol=2e-4, atol=1e-6, msg="M")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N", [(4, 256), (2, 300), (16, 2048)])
def test_abmil_pool_bwd(dtype, B, N):
    from murcl_amd import ops
    dev = _dev()
    H, Wa, ba, wb, bb = _k2_inputs(10, B, N)
    H, Wa = H.to(dtype), Wa.to(dtype)
    dM = _rand(10, "dM", (B, 512))
    # oracle gradient w.r.t. the pre-tanh activations U = H Wa^T + ba
    Hf, Waf = H.float(), Wa.float()
    U = (Hf @ Waf.t() + ba).requires_grad_()
    bbr, wbr = bb.clone().requires_grad_(), wb.clone().requires_grad_()
    s = (torch.tanh(U) @ wbr.t()).squeeze(-1) + bbr
    A = torch.softmax(s, 1) / math.sqrt(N)
    Mo = torch.einsum("bn,bnl->bl", A, Hf)
    (Mo * dM).sum().backward()
    sc, A_g, M_g, ml = ops.abmil_pool_fwd(H.to(dev), Wa.to(dev), ba.to(dev), wb.to(dev), bb.to(dev))
    dT, dba, dwb, dbb = ops.abmil_pool_bwd(H.to(dev), Wa.to(dev), ba.to(dev), wb.to(dev), sc, ml, M_g, dM.to(dev))
    f32 = dtype == torch.float32
    ref = U.grad.reshape(B * N, 128)
    scale = ref.abs().max().item()
    _close(dT.float(), ref, rtol=1e-4 if f32 else 2e-2, atol=(1e-5 if f32 else 1e-2) * scale, msg="dT")
    _close(dba, ref.sum(0), rtol=1e-3 if f32 else 3e-2, atol=(1e-4 if f32 else 3e-2) * ref.sum(0).abs().max().item(), msg="dba")
    _close(dwb, wbr.grad[0], rtol=1e-3 if f32 else 3e-2, atol=(1e-4 if f32 else 3e-2) * wbr.grad.abs().max().item(), msg="dwb")
    assert abs(dbb.item()) < 1e-3 * max(1.0, wbr.grad.abs().max().item())      # softmax shift invariance


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,Lout", [(4, 256, 512), (128, 2048, 512), (2, 300, 512), (1, 1, 512), (19, 700, 64), (33, 96, 16), (3, 9000, 512)])
def test_abmil_pool_decoder_merges_the_chunk_partials_on_load(dtype, B, N, Lout):
    """murcl_abmil_pool_decoder (round 6: K2's per-bag merge inside the decoder launch, abmil.py:29-32,43) against the two launches
    it replaces - murcl_abmil_pool_combine for M / (m, l), then relu(M Wd^T + bd) in float64: M, ml to 1e-6, out to 1e-4 (exact-f32
    MFMA, only the summation order differs); bags that are not a multiple of the 16-bag tile, one to many chunks per bag."""
    from murcl_amd import ops
    dev = _dev()
    H, Wa, ba, wb, bb = _k2_inputs(21, B, N)
    wd = _rand(21, f"wd{Lout}", (Lout, 512), 1 / math.sqrt(512)).to(dev)
    bd = _rand(21, f"bd{Lout}", (Lout,), 0.1).to(dev)
    Hd, Wad = H.to(dtype).to(dev), Wa.to(dtype).to(dev)
    sc, part = ops.abmil_pool_partials(Hd, Wad, ba.to(dev), wb.to(dev), bb.to(dev))
    A0, M0, ml0 = ops.abmil_pool_combine(sc, part, dtype)
    for relu in (True, False):
        out, M, ml = ops.abmil_pool_decoder(part, B, N, dtype, wd, bd, relu=relu)
        _close(M, M0, rtol=1e-6, atol=1e-6 * M0.abs().max().item(), msg="M")
        _close(ml, ml0, rtol=1e-6, atol=0, msg="ml")
        ref = M0.double().cpu() @ wd.double().cpu().t() + bd.double().cpu()
        ref = torch.relu(ref) if relu else ref
        _close(out, ref, rtol=1e-4, atol=1e-5 * ref.abs().max().item(), msg="out")
    # the attention rows on demand = what the combine launch writes
    _close(ops.abmil_attention(sc, ml), A0, rtol=1e-6, atol=0, msg="A on demand")
    # run-to-run bit-reproducibility (single writer per element, fixed summation order)
    out2, M2, _ = ops.abmil_pool_decoder(part, B, N, dtype, wd, bd)
    out3, M3, _ = ops.abmil_pool_decoder(part, B, N, dtype, wd, bd)
    assert torch.equal(out2, out3) and torch.equal(M2, M3)


def test_abmil_pool_decoder_beyond_the_kernels_chunk_count_takes_the_two_launch_form():
    """One bag of 100 000 rows is cut into more than 512 chunks: murcl_abmil_pool_decoder returns -1 and the wrapper runs the merge
    launch + the library GEMM (both HIP): same numbers."""
    from murcl_amd import ops, _lib
    dev = _dev()
    B, N = 1, 100000
    assert ops.pool_chunks(B, N, _lib.BF16)[1] > 512
    H, Wa, ba, wb, bb = _k2_inputs(22, B, N)
    wd, bd = _rand(22, "wd", (512, 512), 1 / math.sqrt(512)).to(dev), _rand(22, "bd", (512,), 0.1).to(dev)
    sc, part = ops.abmil_pool_partials(H.bfloat16().to(dev), Wa.bfloat16().to(dev), ba.to(dev), wb.to(dev), bb.to(dev))
    _, M0, ml0 = ops.abmil_pool_combine(sc, part, torch.bfloat16)
    out, M, ml = ops.abmil_pool_decoder(part, B, N, torch.bfloat16, wd, bd)
    assert torch.equal(M, M0) and torch.equal(ml, ml0)
    ref = torch.relu(M0.double().cpu() @ wd.double().cpu().t() + bd.double().cpu())
    _close(out, ref, rtol=1e-4, atol=1e-5 * ref.abs().max().item(), msg="out")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N", [(4, 256), (2, 300), (16, 2048), (1, 1)])
def test_abmil_pool_bwd_leaves_the_attention_rows_behind(dtype, B, N):
    """``want_A``: A = softmax(s)/sqrt(N) out of the backward pass (the rank-1 input gradient's row scale) = the rows the merge launch
    writes, to 1e-6 (f32: __expf vs expf) / 1e-5 (bf16 path: the fast exp2 form); the other results unchanged bit for bit."""
    from murcl_amd import ops
    dev = _dev()
    H, Wa, ba, wb, bb = _k2_inputs(23, B, N)
    Hd, Wad, bad, wbd, bbd = H.to(dtype).to(dev), Wa.to(dtype).to(dev), ba.to(dev), wb.to(dev), bb.to(dev)
    dM = _rand(23, "dM", (B, 512)).to(dev)
    sc, A0, M, ml = ops.abmil_pool_fwd(Hd, Wad, bad, wbd, bbd)
    r0 = ops.abmil_pool_bwd(Hd, Wad, bad, wbd, sc, ml, M, dM)
    r1 = ops.abmil_pool_bwd(Hd, Wad, bad, wbd, sc, ml, M, dM, want_A=True)
    assert all(torch.equal(a, b) for a, b in zip(r0, r1[:4]))
    _close(r1[4], A0, rtol=1e-6 if dtype == torch.float32 else 1e-5, atol=1e-12, msg="A")


@pytest.mark.parametrize("B,N", [(4, 256), (2, 300), (3, 40), (1, 1), (16, 2048), (300, 96), (128, 2048)])
def test_abmil_pool_bwd_with_the_attention_weight_gradient_in_the_same_pass(B, N):
    """The PARKED one-pass pooling backward (tools/_abl/attn_pool_bwd_dwa.hip, built by tools/_abl/build_kd.py into its own library -
    it lost its A/B in round 5 and left the product library in round 6; this keeps it honest for the next attempt).
    murcl_abmil_pool_bwd_dwa (bf16): dT / dba / dwb / dbb as the two-launch form leaves them, dWa = dT^T H against
    (a) the product of the STORED (bf16) dT with H in float64 - what murcl_gemm_tn forms, to 1e-4 of the largest entry -
    and (b) the oracle's autograd of models/abmil.py:38-42 w.r.t. attention.0.weight on the bf16-rounded inputs;
    accumulation into an existing gradient; run-to-run bit-reproducibility.  Ragged N (not a multiple of 32), single-row bags,
    fewer / more items than workgroups."""
    import os
    import sys
    from murcl_amd import ops
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_abl"))
    import kd
    if not kd.available():
        pytest.skip("tools/_abl/lib/kd.so not built (python tools/_abl/build_kd.py)")
    dev = _dev()
    H, Wa, ba, wb, bb = _k2_inputs(12, B, N)
    H, Wa = H.bfloat16(), Wa.bfloat16()
    dM = _rand(12, "dM", (B, 512))
    Hd, Wad, bad, wbd, bbd, dMd = (t.to(dev) for t in (H, Wa, ba, wb, bb, dM))
    sc, A_g, M_g, ml = ops.abmil_pool_fwd(Hd, Wad, bad, wbd, bbd)
    dT0, dba0, dwb0, dbb0 = ops.abmil_pool_bwd(Hd, Wad, bad, wbd, sc, ml, M_g, dMd)
    dT, dba, dwb, dbb, dWa = kd.pool_bwd_dwa(Hd, Wad, bad, wbd, sc, ml, M_g, dMd, dwa="new")
    # the same formulas on the same inputs: the stored dT differs at most by the last bf16 bit (MFMA operand order)
    # (a single-row bag has p = 1 and g = c: ds is the rounding noise of g - c, so every comparison gets a floor of the size of
    #  that noise - fl for an entry of dT, flw for an entry of dWa)
    #  (g comes off the matrix cores with dM as a hi + lo bf16 pair: ~2^-15 of a 512-term dot product; c = dM.M is plain f32)
    fl = 2.0 ** -15 * math.sqrt(512) * (dM.abs().max() * H.float().abs().max() * wb.abs().max()).item()
    flw = fl * H.float().abs().max().item() * math.sqrt(B * N)
    sT = dT0.float().abs().max().item()
    _close(dT.float(), dT0.float().cpu(), rtol=1e-2, atol=1e-2 * sT + fl, msg="dT vs the unfused kernel")
    _close(dba, dba0.cpu(), rtol=1e-3, atol=1e-3 * dba0.abs().max().item() + fl * math.sqrt(B * N), msg="dba")
    _close(dwb, dwb0.cpu(), rtol=1e-3, atol=1e-3 * dwb0.abs().max().item() + fl * math.sqrt(B * N), msg="dwb")
    assert abs(dbb.item()) < 1e-3 * max(1.0, dwb0.abs().max().item())
    ref = dT.double().t().cpu() @ H.reshape(B * N, 512).double()               # (a) the product of what was stored
    _close(dWa, ref, rtol=1e-4, atol=1e-4 * ref.abs().max().item(), msg="dWa vs dT^T H")
    dWa_tn = ops.gemm_tn(dT0, Hd.view(B * N, 512))
    _close(dWa, dWa_tn.cpu(), rtol=2e-2, atol=5e-3 * ref.abs().max().item() + flw, msg="dWa vs murcl_gemm_tn of the unfused dT")
    # (b) oracle autograd on the rounded inputs
    Hf, War = H.float(), Wa.float().requires_grad_()
    s = (torch.tanh(Hf @ War.t() + ba) @ wb.t()).squeeze(-1) + bb
    Mo = torch.einsum("bn,bnl->bl", torch.softmax(s, 1) / math.sqrt(N), Hf)
    g_wa, = torch.autograd.grad((Mo * dM).sum(), War, retain_graph=True)
    # the pooled vector reaches Wa through the scores only: the H factor of d(Mo)/d(A) is not a Wa path, autograd gives exactly dT^T H
    _close(dWa, g_wa, rtol=3e-2, atol=2e-2 * g_wa.abs().max().item() + flw, msg="dWa vs the oracle")
    # accumulation + reproducibility
    base = _rand(12, "base", (128, 512)).to(dev)
    acc = base.clone()
    out = kd.pool_bwd_dwa(Hd, Wad, bad, wbd, sc, ml, M_g, dMd, dwa=acc)
    assert out[4] is acc
    assert torch.equal(acc, base + dWa)
    assert torch.equal(kd.pool_bwd_dwa(Hd, Wad, bad, wbd, sc, ml, M_g, dMd, dwa="new")[4], dWa)
    assert torch.equal(out[0], dT)


# ------------------------------------------------------------------ NT-Xent
@pytest.mark.parametrize("Bh", [2, 4, 9, 64, 65, 100, 128, 512, 777])
@pytest.mark.parametrize("tau", [1.0, 0.5, 0.07])
def test_ntxent(Bh, tau):
    from murcl_amd import ops
    dev = _dev()
    zi = _rand(11, f"zi{Bh}", (Bh, 128)).requires_grad_()
    zj = (zi.detach() * 0.7 + 0.3 * _rand(11, f"zj{Bh}", (Bh, 128))).requires_grad_()
    loss_ref = O.nt_xent(zi, zj, tau)
    loss_ref.backward()
    z = torch.cat([zi.detach(), zj.detach()]).to(dev)
    loss, dz, sim = ops.ntxent(z, tau)
    # loss = lse - positive logit, both O(1/tau): absolute error scales with 1/tau, not with the loss
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()) + 2e-7 / tau
    gref = torch.cat([zi.grad, zj.grad])
    # (P_ij - 1) cancels to ~1e-5 at small tau; fp32 exp carries ~1e-7 absolute error per weight on
    # BOTH sides, scaled by 1/(n tau) in the gradient.
    _close(dz, gref, rtol=1e-3, atol=2e-5 * gref.abs().max().item() + 4e-7 / (2 * Bh * tau), msg="dz")
    _close(sim, O.row_cosine(zi.detach(), zj.detach()), rtol=1e-4, atol=1e-6, msg="sim")


@pytest.mark.parametrize("T_,Bh", [(1, 4), (6, 64), (3, 9), (5, 33)])
def test_ntxent_batched_equals_one_problem_at_a_time(T_, Bh):
    """murcl_ntxent_fwd_bwd_batched: the T patch steps' NT-Xent problems (2B <= 128 rows each) in one launch - loss, gradient
    and cosines of every problem are those of the single-problem entry point (same kernel, blockIdx.y = problem)."""
    from murcl_amd import ops
    dev = _dev()
    z = _rand(21, f"z{T_}{Bh}", (T_, 2 * Bh, 128)).to(dev)
    loss, dz, sim = ops.ntxent_batched(z, 0.5)
    assert loss.shape == (T_,) and dz.shape == z.shape and sim.shape == (T_, Bh)
    for t in range(T_):
        l1, d1, s1 = ops.ntxent(z[t], 0.5)
        assert torch.equal(loss[t], l1[0]) and torch.equal(dz[t], d1) and torch.equal(sim[t], s1)
    l2, d2, s2 = ops.ntxent_batched(z, 0.5, want_grad=False)
    assert d2 is None and torch.equal(l2, loss) and torch.equal(s2, sim)


@pytest.mark.parametrize("Bh", [2, 9, 33, 64])
def test_ntxent_one_exchange_kernel_equals_the_recompute_kernel(Bh, monkeypatch):
    """murcl_ntxent_small_xchg (round 4: a workgroup forms only its own 16 rows of logits and the workgroups exchange their lse as
    {value, generation} granules) against ntxent_small_kernel (every workgroup recomputes all logits): same logits bit for bit, so
    loss / gradient / cosines agree to the last reduction-order bits - over MANY back-to-back launches on changing data (a stale
    granule of an earlier launch, or a missed one, would show as a wrong lse), single and batched, with and without a window."""
    from murcl_amd import ops
    dev = _dev()
    for it in range(40):
        z = _rand(31, f"x{Bh}.{it}", (2 * Bh, 128), 1.0 + 0.1 * it).to(dev)
        monkeypatch.setattr(ops, "_NTX_XCHG", False)
        l0, d0, s0 = ops.ntxent(z, 0.5)
        lw, dw, _ = ops.ntxent(z, 0.5, grad_lo=0, grad_hi=max(1, Bh // 2))
        monkeypatch.setattr(ops, "_NTX_XCHG", True)
        l1, d1, s1 = ops.ntxent(z, 0.5)
        lx, dx, _ = ops.ntxent(z, 0.5, grad_lo=0, grad_hi=max(1, Bh // 2))
        assert abs(l1.item() - l0.item()) <= 2e-6 * abs(l0.item()), it
        _close(d1, d0.cpu(), rtol=1e-5, atol=1e-6 * d0.abs().max().item(), msg=f"dz {it}")
        assert torch.equal(s1, s0)
        _close(dx, dw.cpu(), rtol=1e-5, atol=1e-6 * d0.abs().max().item(), msg=f"dz window {it}")
    zb = _rand(31, f"b{Bh}", (5, 2 * Bh, 128)).to(dev)
    monkeypatch.setattr(ops, "_NTX_XCHG", False)
    lb0, db0, sb0 = ops.ntxent_batched(zb, 1.0)
    monkeypatch.setattr(ops, "_NTX_XCHG", True)
    for _ in range(10):
        lb1, db1, sb1 = ops.ntxent_batched(zb, 1.0)
        _close(lb1, lb0.cpu(), rtol=2e-6, atol=0, msg="batched loss")
        _close(db1, db0.cpu(), rtol=1e-5, atol=1e-6 * db0.abs().max().item(), msg="batched dz")
        assert torch.equal(sb1, sb0)


def test_ntxent_exchange_generation_lives_on_the_device():
    """The generation number of murcl_ntxent_small_xchg is read from (and advanced in) the exchange buffer by the kernel itself, not
    passed from the host: a launch that is REPLAYED (a captured hipGraph) gets a new generation every time.  Replays on changing
    inputs must equal eager calls."""
    from murcl_amd import ops
    dev = _dev()
    Bh = 48
    z = _rand(33, "g0", (2 * Bh, 128)).to(dev)
    ops.ntxent(z, 0.5)                                   # exchange buffer, kernel attributes: before the capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.ntxent(z, 0.5)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_g, dz_g, sim_g = ops.ntxent(z, 0.5)
    for it in range(6):
        z.copy_(_rand(33, f"g{it + 1}", (2 * Bh, 128), 1.0 + 0.2 * it).to(dev))
        g.replay()
        torch.cuda.synchronize()
        l, d, s_ = loss_g.clone(), dz_g.clone(), sim_g.clone()
        l0, d0, s0 = ops.ntxent(z, 0.5)
        assert torch.equal(l, l0) and torch.equal(d, d0) and torch.equal(s_, s0), it


@pytest.mark.parametrize("Bh,step", [(32, 8), (256, 64), (200, 40)])
def test_ntxent_sharded_rows_match_global(Bh, step):
    """Rank-local gradient slices assemble to the global gradient; 256 = the global batch of 4 ranks."""
    from murcl_amd import ops
    dev = _dev()
    z = torch.cat([_rand(12, f"a{Bh}", (Bh, 128)), _rand(12, f"b{Bh}", (Bh, 128))]).to(dev)
    loss, dz, _ = ops.ntxent(z, 0.5)
    parts = torch.zeros_like(dz)
    for lo in range(0, Bh, step):
        l2, d2, _ = ops.ntxent(z, 0.5, grad_lo=lo, grad_hi=lo + step)
        rows = torch.zeros(2 * Bh, dtype=torch.bool)
        rows[lo:lo + step] = rows[Bh + lo:Bh + lo + step] = True
        assert not d2[~rows.to(dev)].any()               # other ranks' rows: exactly zero
        assert l2.item() == pytest.approx(loss.item(), rel=1e-6)
        parts += d2
    _close(parts, dz, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("world,bl", [(2, 8), (4, 64), (8, 64), (3, 40)])
def test_ntxent_gathered_layout_equals_view_major(world, bl):
    """pair_stride = bags per rank: the all-gathered [rank][view][bag] buffer gives the same loss, cosines and (row for
    row) the same gradient as the reference's cat(z_i, z_j) order, and a rank's window touches only its own rows."""
    from murcl_amd import ops
    dev = _dev()
    Bh = world * bl
    zv = torch.cat([_rand(14, f"gi{world}.{bl}", (Bh, 128)), _rand(14, f"gj{world}.{bl}", (Bh, 128))])   # view-major
    rows = torch.arange(2 * Bh)
    bag = (rows // (2 * bl)) * bl + rows % bl
    src = ((rows // bl) % 2) * Bh + bag                     # gathered row r holds view-major row src[r]
    zg = zv[src].to(dev)
    loss0, dz0, sim0 = ops.ntxent(zv.to(dev), 0.5)
    loss1, dz1, sim1 = ops.ntxent(zg, 0.5, pair_stride=bl)
    assert loss1.item() == pytest.approx(loss0.item(), rel=1e-6)
    _close(sim1, sim0.cpu(), rtol=1e-6, atol=1e-7, msg="sim")
    _close(dz1, dz0.cpu()[src], rtol=1e-5, atol=1e-9, msg="dz")
    r = world - 1
    _, dzr, _ = ops.ntxent(zg, 0.5, grad_lo=r * bl, grad_hi=(r + 1) * bl, pair_stride=bl)
    mine = slice(r * 2 * bl, (r + 1) * 2 * bl)
    _close(dzr[mine], dz1.cpu()[mine], rtol=1e-6, atol=1e-10, msg="own rows")
    dzr[mine] = 0
    assert not dzr.any()


# ------------------------------------------------------------------ small helpers
def test_cast_transpose_colsum_relu():
    from murcl_amd import ops
    dev = _dev()
    x = _rand(13, "x", (513, 130))
    xb = ops.cast(x.to(dev), torch.bfloat16)
    assert torch.equal(xb.cpu(), x.bfloat16())
    assert torch.equal(ops.cast(xb, torch.float32).cpu(), x.bfloat16().float())
    assert torch.equal(ops.transpose_cast(x.to(dev), torch.float32).cpu(), x.t().contiguous())
    assert torch.equal(ops.transpose_cast(x.to(dev), torch.bfloat16).cpu(), x.t().contiguous().bfloat16())
    _close(ops.colsum(x.to(dev)), x.double().sum(0), rtol=1e-5, atol=1e-4)
    _close(ops.colsum(xb), x.bfloat16().double().sum(0), rtol=1e-5, atol=1e-4)
    y = _rand(13, "y", (513, 130))
    assert torch.equal(ops.relu_bwd(x.to(dev), y.to(dev)).cpu(), x * (y > 0))


def test_gru_gates_fwd_bwd():
    from murcl_amd import ops
    dev = _dev()
    B, H = 6, 1024
    gi, gh = _rand(14, "gi", (B, 3 * H)).requires_grad_(), _rand(14, "gh", (B, 3 * H)).requires_grad_()
    hp = _rand(14, "hp", (B, H)).requires_grad_()
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    hn = (1 - z) * n + z * hp
    dh = _rand(14, "dh", (B, H))
    (hn * dh).sum().backward()
    hnew, gates = ops.gru_gates_fwd(gi.detach().to(dev), gh.detach().to(dev), hp.detach().to(dev))
    _close(hnew, hn, rtol=1e-5, atol=1e-6)
    dgi, dgh, dhp = ops.gru_gates_bwd(dh.to(dev), gates, gh.detach().to(dev), hp.detach().to(dev))
    _close(dgi, gi.grad, rtol=1e-4, atol=1e-6)
    _close(dgh, gh.grad, rtol=1e-4, atol=1e-6)
    _close(dhp, hp.grad, rtol=1e-4, atol=1e-6)
    # restart (no previous hidden)
    h0, _ = ops.gru_gates_fwd(gi.detach().to(dev), gh.detach().to(dev), None)
    _close(h0, (1 - z) * n, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,N,C,ld", [(16, 8192, 2, 2), (3, 5000, 3, 132), (2, 100, 1, 1), (1, 2049, 4, 4)])
def test_dsmil_argmax_first_index_of_the_maximum(B, N, C, ld):
    """dsmil_argmax (eight loads in flight per thread): the FIRST index of each column's maximum (torch.max's rule on equal values,
    dsmil.py:69), also for repeated maxima and for a column of -inf; the maxima ride along."""
    from murcl_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    s = torch.randn((B, N, ld), generator=g)
    s[:, :, :C] = (s[:, :, :C] * 4).round() / 4                  # many exact ties
    s[0, :, 0] = -float("inf")                                   # a column with no finite entry
    if N > 300:
        s[-1, 7, C - 1] = s[-1, 299, C - 1] = 100.0              # the maximum twice: the earlier row wins
    m, mx = ops.dsmil_argmax(s.view(B * N, ld).to(dev), B, N, C, want_max=True)
    want_v, want_i = s[:, :, :C].max(1)
    first = torch.stack([torch.stack([(s[b, :, c] == want_v[b, c]).nonzero()[0, 0] for c in range(C)]) for b in range(B)])
    assert torch.equal(m.cpu().long(), first)
    assert torch.equal(mx.cpu(), want_v)
    assert torch.equal(ops.dsmil_argmax(s.view(B * N, ld).to(dev), B, N, C).cpu(), m.cpu())


def test_small_f32_weight_gradients_in_one_launch():
    """ops.gemm_tn_grouped on 2-4 f32 products of a few hundred rows (a PPO epoch's weight gradients): one launch of the 32 x 32
    single-writer kernel; products and column sums are ADDED to what the outputs hold."""
    from murcl_amd import ops
    dev = _dev()
    shapes = [(320, 2048, 512, True), (320, 512, 2048, True), (320, 1536, 512, True), (256, 1536, 512, False)]
    for n in (2, 3, 4):
        probs, want = [], []
        for g, (M, N1, N2, cs) in enumerate(shapes[:n]):
            A, B = _rand(44, f"A{g}", (M, N1)), _rand(44, f"B{g}", (M, N2))
            C0, c0 = _rand(44, f"C{g}", (N1, N2)), _rand(44, f"c{g}", (N1,))
            out, ci = C0.to(dev).clone(), (c0.to(dev).clone() if cs else None)
            probs.append((A.to(dev), B.to(dev), out, ci, None))
            want.append((C0.double() + A.double().t() @ B.double(), c0.double() + A.double().sum(0) if cs else None))
        assert ops.gemm_tn_small_grouped_ok([tuple(p) + (None,) for p in probs])
        Cs = ops.gemm_tn_grouped(probs)
        for (A, B, out, ci, _), C, (wC, wc) in zip(probs, Cs, want):
            assert C.data_ptr() == out.data_ptr()
            _close(C, wC, rtol=1e-4, atol=2e-4)
            if wc is not None:
                _close(ci, wc, rtol=1e-4, atol=2e-4)
    # ragged row counts and widths that are not multiples of the tile
    A, B, A2, B2 = _rand(45, "A", (77, 36)), _rand(45, "B", (77, 100)), _rand(45, "A2", (5, 64)), _rand(45, "B2", (5, 32))
    Cs = ops.gemm_tn_grouped([(A.to(dev), B.to(dev), None, None, None), (A2.to(dev), B2.to(dev), None, None, None)])
    _close(Cs[0], A.double().t() @ B.double(), rtol=1e-4, atol=1e-4)
    _close(Cs[1], A2.double().t() @ B2.double(), rtol=1e-4, atol=1e-4)


def test_small_gemm_applies_relu_mask_in_the_epilogue():
    """murcl_gemm_nt's small f32 path with MURCL_EPI_MASK: dx = (dy W) * (y > 0) in one launch (the PPO encoder's dgrads)."""
    from murcl_amd import ops
    dev = _dev()
    for M, N, K in ((320, 512, 1536), (320, 2048, 512), (37, 96, 64)):
        dy, wt, y = _rand(46, "dy", (M, K)), _rand(46, "wt", (N, K)) / math.sqrt(K), _rand(46, "y", (M, N))
        got = ops.gemm_nt(dy.to(dev), wt.to(dev), epi=ops.EPI_MASK, mask=y.to(dev))
        _close(got, (dy.double() @ wt.double().t()) * (y > 0), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("B,H,Kx", [(64, 512, 0), (128, 512, 512), (17, 32, 0), (5, 48, 80), (33, 1024, 0), (64, 512, 1024)])
def test_gru_step_forward_in_one_launch(B, H, Kx):
    """murcl_gru_step_fwd (h W_hh^T, optionally x W_ih^T, and the gate math in one launch) against torch.nn.GRUCell's formula
    in f64; ragged row counts, hidden sizes that leave partial 256-k chunks, a longer input side."""
    from murcl_amd import ops
    dev = _dev()
    hp, whh, bhh = _rand(41, "hp", (B, H)), _rand(41, "whh", (3 * H, H)) / math.sqrt(H), _rand(41, "bhh", (3 * H,))
    bih = _rand(41, "bih", (3 * H,))
    if Kx:
        x, wih = _rand(41, "x", (B, Kx)), _rand(41, "wih", (3 * H, Kx)) / math.sqrt(Kx)
        gi = x.double() @ wih.double().t() + bih.double()
    else:
        gi = _rand(41, "gi", (B, 3 * H)).double()
    gh = hp.double() @ whh.double().t() + bhh.double()
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    hn = (1 - z) * n + z * hp.double()
    assert ops.gru_step_ok(B, H, Kx)
    if Kx:
        hnew, gates, gh_k = ops.gru_step_fwd(bih.to(dev), hp.to(dev), whh.to(dev), bhh.to(dev), x=x.to(dev), w_ih=wih.to(dev))
    else:
        hnew, gates, gh_k = ops.gru_step_fwd(gi.float().to(dev), hp.to(dev), whh.to(dev), bhh.to(dev))
    _close(hnew, hn, rtol=1e-4, atol=2e-5)
    _close(gates, torch.cat([r, z, n], 1), rtol=1e-4, atol=2e-5)
    _close(gh_k, gh, rtol=1e-4, atol=2e-5)
    if Kx:                                                 # restart: zero state, no recurrent product (gh = b_hh)
        r0, z0 = torch.sigmoid(gi[:, :H] + bhh[:H].double()), torch.sigmoid(gi[:, H:2 * H] + bhh[H:2 * H].double())
        n0 = torch.tanh(gi[:, 2 * H:] + r0 * bhh[2 * H:].double())
        h0, g0, _ = ops.gru_step_fwd(bih.to(dev), None, whh.to(dev), bhh.to(dev), x=x.to(dev), w_ih=wih.to(dev), want_gh=False)
        _close(h0, (1 - z0) * n0, rtol=1e-4, atol=2e-5)
        _close(g0, torch.cat([r0, z0, n0], 1), rtol=1e-4, atol=2e-5)
    h2, g2, gh2 = ops.gru_step_fwd(bih.to(dev) if Kx else gi.float().to(dev), hp.to(dev), whh.to(dev), bhh.to(dev),
                                   x=x.to(dev) if Kx else None, w_ih=wih.to(dev) if Kx else None, want_backward=False)
    assert g2 is None and gh2 is None and torch.equal(h2, hnew)


@pytest.mark.parametrize("B,H,first", [(64, 512, False), (64, 512, True), (17, 32, False), (33, 1024, False), (5, 48, True)])
def test_gru_step_backward_in_one_launch(B, H, first):
    """murcl_gru_step_bwd = gemm_nt(dgh_next, W_hh^T, accumulate into dh) + gru_gates_bwd_into; `first`: the step from the zero
    state (gh is the bias row, no previous hidden state, dh * z is not accumulated)."""
    from murcl_amd import ops
    dev = _dev()
    dgh_n, whh = _rand(42, "dghn", (B, 3 * H)), _rand(42, "whh", (3 * H, H)) / math.sqrt(H)
    dh0, gates = _rand(42, "dh", (B, H)), torch.rand((B, 3 * H), generator=torch.Generator().manual_seed(5)) * 0.9 + 0.05
    gh = _rand(42, "gh", (1 if first else B, 3 * H))
    hp, dprev0 = (None if first else _rand(42, "hp", (B, H))), _rand(42, "dprev", (B, H))
    d = dh0.double() + dgh_n.double() @ whh.double()
    r, z, n = gates[:, :H].double(), gates[:, H:2 * H].double(), gates[:, 2 * H:].double()
    ghn = gh[:, 2 * H:].double()
    dn = d * (1 - z) * (1 - n * n)
    dz = d * ((0 if first else hp.double()) - n) * z * (1 - z)
    dr = dn * ghn * r * (1 - r)
    dh, dprev = dh0.to(dev).clone(), dprev0.to(dev).clone()
    dgi, dgh = torch.empty((B, 3 * H), device=dev), torch.empty((B, 3 * H), device=dev)
    ops.gru_step_bwd(dgh_n.to(dev), whh.t().contiguous().to(dev), dh, gates.to(dev), gh.to(dev), None if first else hp.to(dev), dgi, dgh,
                     dprev, accumulate=not first)
    _close(dh, d, rtol=1e-4, atol=2e-5)
    _close(dgi, torch.cat([dr, dz, dn], 1), rtol=1e-4, atol=2e-5)
    _close(dgh, torch.cat([dr, dz, dn * r], 1), rtol=1e-4, atol=2e-5)
    _close(dprev, (0 if first else dprev0.double()) + d * z, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("T,B,I,H", [(6, 64, 512, 512), (3, 9, 48, 32), (1, 4, 32, 32)])
def test_gru_sequence_function_against_torch_gru(T, B, I, H):
    """functional.GRUSeqFn (one launch per time step and direction) against torch.nn.GRU on the CPU: all hidden states and
    every gradient."""
    from murcl_amd.functional import GRUSeqFn
    dev = _dev()
    torch.manual_seed(7)
    ref = torch.nn.GRU(I, H)
    x = _rand(43, "x", (T, B, I)).requires_grad_()
    dhs = _rand(43, "dhs", (T, B, H))
    out, _ = ref(x)
    (out * dhs).sum().backward()
    ps = [p.detach().to(dev).requires_grad_() for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    xd = x.detach().to(dev).requires_grad_()
    hs = GRUSeqFn.apply(xd, *ps)
    (hs * dhs.to(dev)).sum().backward()
    _close(hs, out, rtol=1e-4, atol=2e-5)
    _close(xd.grad, x.grad, rtol=1e-3, atol=5e-5)
    for p, q in zip(ps, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)):
        _close(p.grad, q.grad, rtol=1e-3, atol=2e-4)


def test_adam_matches_oracle():
    from murcl_amd import ops
    dev = _dev()
    p0, g = _rand(15, "p", (1000,)), _rand(15, "g", (1000,))
    p = p0.to(dev).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    st, ref = {}, {"w": p0.clone()}
    for step in range(1, 4):
        ops.adam_step(p, g.to(dev), m, v, 1e-3, (0.9, 0.999), 1e-8, 1e-5, step)
        ref = O.adam_step(ref, {"w": g}, st, 1e-3, weight_decay=1e-5)
    _close(p, ref["w"], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ weight-stationary panel GEMM (bf16)
def _bits(mask_bool):
    """[M,N] bool -> the kernel's 1-bit mask layout (panel_gemm.hip header): 128-byte blocks per (32-row tile,
    32-column group); 16-bit word L = 16*((n&15)>>2) + (m&15); bit (7 - idx//2) + 8*(idx&1) with
    idx = 8*((m>>4)&1) + 4*((n>>4)&1) + (n&3)."""
    M, N = mask_bool.shape
    m = torch.arange(M).view(M, 1)
    n = torch.arange(N).view(1, N)
    blk = (m // 32) * (N // 32) + n // 32
    L = 16 * ((n & 15) >> 2) + (m & 15)
    idx = 8 * ((m >> 4) & 1) + 4 * ((n >> 4) & 1) + (n & 3)
    bit = (7 - idx // 2) + 8 * (idx & 1)
    byte = (blk * 128 + L * 2 + bit // 8).reshape(-1)
    val = (mask_bool.reshape(-1).int() << (bit % 8).expand(M, N).reshape(-1)).to(torch.int32)
    out = torch.zeros(M * N // 8, dtype=torch.int32)
    out.index_add_(0, byte, val)
    return out.to(torch.uint8).view(M, N // 8)


@pytest.mark.parametrize("M", [32, 4096, 131072 + 64])
def test_panel_gemm_forward_bias_relu_and_bitmask(M):
    from murcl_amd import ops
    dev = _dev()
    A = _rand(20, f"A{M}", (M, 512)).bfloat16()
    W = _rand(20, "W", (512, 512), 1 / math.sqrt(512)).bfloat16()
    bias = _rand(20, "b", (512,), 0.5)
    assert ops.panel_supported(M, 512, 512, ops.PG_BIAS_RELU)
    C, bm, _ = ops.panel_gemm(A.to(dev), W.to(dev), ops.PG_BIAS_RELU, bias=bias.to(dev), want_bitmask=True)
    ref = torch.relu(A.double() @ W.double().t() + bias.double())
    _close(C.float(), ref, rtol=1e-2, atol=1e-2, msg="C")
    assert torch.equal(bm.cpu(), _bits(C.cpu().float() > 0)), "bit mask must describe the stored output exactly"
    # agrees with the tile kernel bit for bit? (same products, different k order -> allow bf16 rounding flips)
    C2 = ops.gemm_nt(A.to(dev), W.to(dev), epi=ops.EPI_BIAS_RELU, bias=bias.to(dev))
    assert (C.float() - C2.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("M", [64, 8192])
def test_panel_gemm_dgrad_mask_and_colsum(M):
    from murcl_amd import ops
    dev = _dev()
    dZ = _rand(21, f"dZ{M}", (M, 512)).bfloat16()
    Wt = _rand(21, "Wt", (512, 512), 1 / math.sqrt(512)).bfloat16()
    H = _rand(21, f"H{M}", (M, 512))
    bm = _bits(H > 0)
    C, _, cs = ops.panel_gemm(dZ.to(dev), Wt.to(dev), ops.PG_MASK, bitmask=bm.to(dev), colsum=True)
    ref = (dZ.double() @ Wt.double().t()) * (H.double() > 0)
    _close(C.float(), ref, rtol=1e-2, atol=1e-2, msg="C")
    _close(cs, C.double().sum(0).float(), rtol=1e-3, atol=1e-2 * math.sqrt(M), msg="colsum")


@pytest.mark.parametrize("bags,n", [(2, 64), (8, 1024)])
def test_panel_gemm_rank1_mask(bags, n):
    from murcl_amd import ops
    dev = _dev()
    M = bags * n
    dT = _rand(22, f"dT{M}", (M, 128)).bfloat16()
    WaT = _rand(22, "WaT", (512, 128), 0.1).bfloat16()
    H = _rand(22, f"H{M}", (M, 512))
    a = torch.from_numpy(detrand.uniform(22, f"a{M}", (M,)))
    dM = _rand(22, f"dM{bags}", (bags, 512))
    C, _, cs = ops.panel_gemm(dT.to(dev), WaT.to(dev), ops.PG_RANK1_MASK, bitmask=_bits(H > 0).to(dev),
                              rowscale=a.to(dev), rank1=dM.to(dev), rows_per_bag=n, colsum=True)
    ref = (dT.double() @ WaT.double().t() + a.double()[:, None] * dM.double().repeat_interleave(n, 0)) * (H.double() > 0)
    _close(C.float(), ref, rtol=1e-2, atol=1e-2, msg="C")
    _close(cs, C.double().sum(0).float(), rtol=1e-3, atol=1e-2 * math.sqrt(M), msg="colsum")


@pytest.mark.parametrize("bags,n", [(2, 64), (8, 1024), (5, 96)])
def test_panel_gemm_rank1_mask_forms_the_attention_rows_from_raw_scores(bags, n):
    """PG_RANK1_MASK with ``bias`` = the [bags,2] soft-max statistics (round 6): ``rowscale`` holds RAW scores and the row scale is
    exp(s - m) / (l sqrt n) = softmax(s)/sqrt(n) (abmil.py:40-41), formed in the epilogue - against the explicit-A form of the same
    kernel (the fast exp2 form differs from expf by ~1e-7 relative: far below the bf16 output) and float64."""
    from murcl_amd import ops
    dev = _dev()
    M = bags * n
    dT = _rand(24, f"dT{M}", (M, 128), 0.05).bfloat16()
    WaT = _rand(24, "WaT", (512, 128), 0.1).bfloat16()
    H = _rand(24, f"H{M}", (M, 512))
    sc = _rand(24, f"s{M}", (bags, n), 2.0)
    dM = _rand(24, f"dM{bags}", (bags, 512))
    m = sc.max(1).values - torch.from_numpy(detrand.uniform(24, f"sh{bags}", (bags,)))        # any reference >= ... the kernel only needs exp(s - m) / l consistent
    l = torch.exp(sc.double() - m.double()[:, None]).sum(1).float()
    ml = torch.stack([m, l], 1).contiguous()
    A = torch.softmax(sc.double(), 1) / math.sqrt(n)
    bits = _bits(H > 0).to(dev)
    C, _, cs = ops.panel_gemm(dT.to(dev), WaT.to(dev), ops.PG_RANK1_MASK, bitmask=bits, rowscale=sc.reshape(-1).to(dev),
                              bias=ml.to(dev), rank1=dM.to(dev), rows_per_bag=n, colsum=True)
    ref = (dT.double() @ WaT.double().t() + A.reshape(-1)[:, None] * dM.double().repeat_interleave(n, 0)) * (H.double() > 0)
    _close(C.float(), ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item(), msg="C")
    C2, _, _ = ops.panel_gemm(dT.to(dev), WaT.to(dev), ops.PG_RANK1_MASK, bitmask=bits, rowscale=A.float().reshape(-1).to(dev),
                              rank1=dM.to(dev), rows_per_bag=n, colsum=True)
    assert (C.float() - C2.float()).abs().max().item() <= 2 ** -7 * ref.abs().max().item()        # one bf16 ulp of the largest entry
    _close(cs, C.double().sum(0).float(), rtol=1e-3, atol=1e-2 * math.sqrt(M), msg="colsum")


def test_panel_gemm_unsupported_shapes_fall_back():
    from murcl_amd import ops
    assert not ops.panel_supported(100, 512, 512, ops.PG_BIAS_RELU)      # M % 32
    assert not ops.panel_supported(64, 384, 512, ops.PG_BIAS_RELU)       # N % 256
    assert not ops.panel_supported(64, 512, 128, ops.PG_RANK1_MASK, 48)  # bag not a whole number of tiles


@pytest.mark.parametrize("M,N1,N2", [(32, 1024, 1024), (128, 128, 1024), (300, 52, 512), (512, 3072, 512), (1000, 64, 64)])
def test_gemm_tn_with_colsum_is_one_launch_for_bag_level_gradients(M, N1, N2):
    """(dW, db) = (dy^T x, column sums of dy) of a bag-level Linear from one launch of the single-writer small-tile kernel (fresh
    outputs, no zero fill; M <= 512), two launches beyond it: float64 reference at 1e-5."""
    from murcl_amd import ops
    dev = _dev()
    A, B = _rand(41, f"A{M}{N1}", (M, N1)).to(dev), _rand(41, f"B{M}{N2}", (M, N2)).to(dev)
    C, cs = ops.gemm_tn_with_colsum(A, B)
    ref = A.double().t().cpu() @ B.double().cpu()
    _close(C, ref, rtol=1e-5, atol=1e-5 * ref.abs().max().item(), msg="C")
    _close(cs, A.double().sum(0).cpu(), rtol=1e-5, atol=1e-5 * math.sqrt(M), msg="colsum")


# ------------------------------------------------------------------ fragment-order weight views (round 6)
def _frag_perm(w):
    """The fragment order of a [R,512] matrix in plain index arithmetic (csrc/elementwise.hip frag_index): a 16-row block is 16 k-steps
    x 64 lanes x 8 elements, lane (q4, r16) of k-step kk holds elements [(kk + 16 q4) * 8, +8) of row r16."""
    R = w.shape[0]
    blk = w.reshape(R // 16, 16, 4, 16, 8)                 # [block, r16, q4, kk, e]
    return blk.permute(0, 3, 2, 1, 4).reshape(R, 512).contiguous()      # [block, kk, q4, r16, e]


def test_fragment_order_views_and_the_kernels_that_take_them():
    """``weight_views`` specs with "frag": the bf16 view equals the row-major view pushed through the fragment permutation (plain and
    transposed sources); ``panel_gemm`` (K = 512: BIAS_RELU with bit mask, MASK) and the K2 pooling passes compute BIT-IDENTICAL results
    from the fragment-order operand (same fragments in the same registers - only the prologue's loads differ); ``gemm_nt`` refuses it."""
    from murcl_amd import ops
    dev = _dev()
    W = _rand(31, "W", (512, 512), 1 / math.sqrt(512)).to(dev)
    Wa = _rand(31, "Wa", (128, 512), 2 / math.sqrt(512)).to(dev)
    Wt_src = _rand(31, "Wt", (512, 512), 1 / math.sqrt(512)).to(dev)
    bf = torch.bfloat16
    plain = ops.weight_views([(W, False, bf), (Wa, False, bf), (Wt_src, True, bf)])
    frag = ops.weight_views([(W, False, bf, "frag"), (Wa, False, bf, "frag"), (Wt_src, True, bf, "frag")])
    for p_, f_ in zip(plain, frag):
        assert ops.is_frag(f_) and not ops.is_frag(p_) and f_.shape == p_.shape
        assert torch.equal(f_.view(torch.int16), _frag_perm(p_).view(torch.int16))
    M = 4096
    X = _rand(31, "X", (M, 512)).bfloat16().to(dev)
    bias = _rand(31, "b", (512,), 0.1).to(dev)
    h0, m0, _ = ops.panel_gemm(X, plain[0], ops.PG_BIAS_RELU, bias=bias, want_bitmask=True)
    h1, m1, _ = ops.panel_gemm(X, frag[0], ops.PG_BIAS_RELU, bias=bias, want_bitmask=True)
    assert torch.equal(h0.view(torch.int16), h1.view(torch.int16)) and torch.equal(m0, m1)
    dZ = _rand(31, "dZ", (M, 512), 0.05).bfloat16().to(dev)
    z0, _, c0 = ops.panel_gemm(dZ, plain[2], ops.PG_MASK, bitmask=m0, colsum=True)
    z1, _, c1 = ops.panel_gemm(dZ, frag[2], ops.PG_MASK, bitmask=m0, colsum=True)
    assert torch.equal(z0.view(torch.int16), z1.view(torch.int16))
    _close(c1, c0, rtol=1e-5, atol=1e-5 * c0.abs().max().item(), msg="colsum")        # (its partial rows meet through float atomics)
    with pytest.raises(AssertionError):
        ops.gemm_nt(X, frag[0])
    # K2 forward / backward with Wa in fragment order
    B, N = 16, 2048
    H, _, ba, wb, bb = _k2_inputs(31, B, N)
    Hd, bad, wbd, bbd = H.bfloat16().to(dev), ba.to(dev), wb.to(dev), bb.to(dev)
    dM = _rand(31, "dM", (B, 512)).to(dev)
    r0 = ops.abmil_pool_fwd(Hd, plain[1], bad, wbd, bbd)
    r1 = ops.abmil_pool_fwd(Hd, frag[1], bad, wbd, bbd)
    assert all(torch.equal(a, b) for a, b in zip(r0, r1))
    g0 = ops.abmil_pool_bwd(Hd, plain[1], bad, wbd, r0[0], r0[3], r0[2], dM)
    g1 = ops.abmil_pool_bwd(Hd, frag[1], bad, wbd, r0[0], r0[3], r0[2], dM)
    assert torch.equal(g0[0].view(torch.int16), g1[0].view(torch.int16)) and all(torch.equal(a, b) for a, b in zip(g0[1:], g1[1:]))
    # ragged / tiny launches take the same prologue
    for B2, N2 in ((1, 1), (3, 40), (2, 300)):
        H2 = _k2_inputs(32, B2, N2)[0].bfloat16().to(dev)
        a = ops.abmil_pool_fwd(H2, plain[1], bad, wbd, bbd)
        b = ops.abmil_pool_fwd(H2, frag[1], bad, wbd, bbd)
        assert all(torch.equal(u, v) for u, v in zip(a, b))


# ------------------------------------------------------------------ cached weight views (one batched cast/transpose launch)
def test_weight_views_follow_parameter_updates():
    from murcl_amd import ops
    dev = _dev()
    w1 = _rand(40, "w1", (512, 512)).to(dev)
    w2 = _rand(40, "w2", (128, 512)).to(dev)
    w3 = _rand(40, "w3", (70, 33)).to(dev)            # ragged tiles
    specs = [(w1, False, torch.bfloat16), (w2, True, torch.bfloat16), (w3, True, torch.float32), (w3, False, torch.bfloat16)]

    def check():
        a, b, c, d = ops.weight_views(specs)
        assert torch.equal(a, w1.bfloat16()) and torch.equal(b, w2.t().contiguous().bfloat16())
        assert torch.equal(c, w3.t().contiguous()) and torch.equal(d, w3.bfloat16())
        return a.data_ptr()

    p0 = check()
    with torch.no_grad():
        w1.mul_(1.5)                                   # in-place torch op: version counter moves
        w3.add_(0.25)
    assert check() == p0                               # same persistent buffers, refreshed contents
    w2.data.copy_(w2 * -2)                             # .data copy_: also bumps the version
    check()
    # parameters owned by a FlatAdam are cached: a raw-pointer update (the Adam kernel) is announced through PARAM_EPOCH
    for w in (w1, w2, w3):
        ops.manage_param(w)
    check()
    ops.adam_step(w1.view(-1), torch.ones_like(w1).view(-1), torch.zeros_like(w1).view(-1), torch.zeros_like(w1).view(-1),
                  1e-1, (0.9, 0.999), 1e-8, 0.0, 1)
    stale = ops.weight_views(specs)[0]
    assert not torch.equal(stale, w1.bfloat16())       # cached on purpose until the epoch moves
    ops.PARAM_EPOCH += 1
    check()
    for w in (w1, w2, w3):
        ops.manage_param(w, False)


def test_relu_bitmask_matches_the_panel_layout():
    from murcl_amd import ops
    dev = _dev()
    for dtype in (torch.bfloat16, torch.float32):
        H = _rand(41, f"H{dtype}", (96, 512)).to(dtype)
        H[5, 7] = 0.0
        assert torch.equal(ops.relu_bitmask(H.to(dev)).cpu(), _bits(H.float() > 0))


@pytest.mark.parametrize("M", [64, 4096])
def test_panel_gemm_bias_only(M):
    from murcl_amd import ops
    dev = _dev()
    A = _rand(42, f"A{M}", (M, 512)).bfloat16()
    W = _rand(42, "W", (512, 512), 1 / math.sqrt(512)).bfloat16()
    bias = _rand(42, "b", (512,), 0.5)
    assert ops.panel_supported(M, 512, 512, ops.PG_BIAS)
    C, _, _ = ops.panel_gemm(A.to(dev), W.to(dev), ops.PG_BIAS, bias=bias.to(dev))
    _close(C.float(), A.double() @ W.double().t() + bias.double(), rtol=1e-2, atol=1e-2, msg="C")


@pytest.mark.parametrize("bags,n", [(2, 64), (4, 2048)])
def test_panel_gemm_rank1_mask_k512(bags, n):
    from murcl_amd import ops
    dev = _dev()
    M = bags * n
    dU = _rand(43, f"dU{M}", (M, 512)).bfloat16()
    Wt = _rand(43, "Wt", (512, 512), 1 / math.sqrt(512)).bfloat16()
    H = _rand(43, f"H{M}", (M, 512))
    a = torch.from_numpy(detrand.uniform(43, f"a{M}", (M,)))
    dM = _rand(43, f"dM{bags}", (bags, 512))
    assert ops.panel_supported(M, 512, 512, ops.PG_RANK1_MASK, n)
    C, _, cs = ops.panel_gemm(dU.to(dev), Wt.to(dev), ops.PG_RANK1_MASK, bitmask=ops.relu_bitmask(H.to(dev)),
                              rowscale=a.to(dev), rank1=dM.to(dev), rows_per_bag=n, colsum=True)
    ref = (dU.double() @ Wt.double().t() + a.double()[:, None] * dM.double().repeat_interleave(n, 0)) * (H.double() > 0)
    _close(C.float(), ref, rtol=1e-2, atol=1e-2, msg="C")
    _close(cs, C.double().sum(0).float(), rtol=1e-3, atol=1e-2 * math.sqrt(M), msg="colsum")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_mask_values_rate_and_reproducibility(dtype):
    from murcl_amd import ops
    dev = _dev()
    torch.manual_seed(123)
    ops._DROP_COUNTER = 0
    a = ops.dropout_mask((4099, 257), dtype, 0.75, dev)                 # ragged tail (not a multiple of 8)
    b = ops.dropout_mask((4099, 257), dtype, 0.75, dev)
    torch.manual_seed(123)
    ops._DROP_COUNTER = 0
    a2 = ops.dropout_mask((4099, 257), dtype, 0.75, dev)
    assert torch.equal(a, a2) and not torch.equal(a, b)                 # (seed, call counter) decide the mask
    vals = set(torch.unique(a.float()).tolist())
    assert vals == {0.0, float(torch.tensor(1 / 0.75, dtype=dtype))}
    keep = (a != 0).float()
    n = keep.numel()
    assert abs(keep.mean().item() - 0.75) < 5 * math.sqrt(0.75 * 0.25 / n)
    # no structure along rows / columns / between the two draws
    assert (keep.mean(0) - 0.75).abs().max().item() < 6 * math.sqrt(0.75 * 0.25 / 4099)
    assert (keep.mean(1) - 0.75).abs().max().item() < 6 * math.sqrt(0.75 * 0.25 / 257)
    both = (keep * (b != 0).float()).mean().item()
    assert abs(both - 0.75 ** 2) < 5 * math.sqrt(0.5625 * 0.4375 / n)
    lag = (keep[:, 1:] * keep[:, :-1]).mean().item()
    assert abs(lag - 0.75 ** 2) < 5 * math.sqrt(0.5625 * 0.4375 / n)
    assert ops.dropout_mask((3,), dtype, 1.0, dev).float().tolist() == [1.0, 1.0, 1.0]


# ------------------------------------------------------------------ CLAM instance branch, backward launch
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["distinct", "ties", "short_bag"])
def test_clam_inst_bwd_feature_gradients_and_sums_against_row_by_row_math(dtype, case):
    """``murcl_clam_inst_bwd`` (clam.py:103-132 backward: index_select of the 2k rows, the instance classifiers, mean CE): the rows'
    feature gradients are added into dz under h > 0 and (dW, db, column sums of what was added) come back.  Reference: the same sums
    row by row in f64 on the stored values.  ``ties`` / ``short_bag``: a bag whose top and bottom rows coincide - such a row is added
    twice, as index_select's backward does (round 5: the launch walks four rows at once unless it finds such a bag)."""
    from murcl_amd import ops
    dev = _dev()
    B, L, k, n_cls = 5, 512, 8, 2
    N = 12 if case == "short_bag" else 40
    R, Oc = 2 * k, 2 * n_cls
    h = torch.relu(_rand(5, "ib.h", (B * N, L))).to(dtype).to(dev)
    dz0 = _rand(5, "ib.dz", (B * N, L), 0.1).to(dtype).to(dev)
    W = _rand(5, "ib.w", (Oc, L), 0.2).to(dev)
    dl = _rand(5, "ib.dl", (B, R, Oc), 0.3).to(dev)
    up = (_rand(5, "ib.up", (B,)).abs() + 0.5).to(dev)
    rng = np.random.default_rng(11)
    ids = np.stack([rng.permutation(N)[:R] if N >= R else rng.integers(0, N, R) for _ in range(B)]).astype(np.int32)
    if case == "ties":
        ids[1, k:] = ids[1, :k]                      # a uniform soft-max: both selections pick the same rows
        ids[3, R - 1] = ids[3, 0]
    ids_t = torch.from_numpy(ids).to(dev)
    dz = dz0.clone()
    dW, db, gsum = ops.clam_inst_bwd(h, ids_t, W, dl, up, B, N, k, n_cls, dz)
    hd, Wd, dld, upd = h.double().cpu(), W.double().cpu(), dl.double().cpu(), up.double().cpu()
    want_dz = dz0.double().cpu().clone()
    want_dW, want_db, want_g = torch.zeros(Oc, L, dtype=torch.float64), torch.zeros(Oc, dtype=torch.float64), torch.zeros(L, dtype=torch.float64)
    for b in range(B):
        for r in range(R):
            row = b * N + int(ids[b, r])
            d = dld[b, r] * upd[b]
            g = (d @ Wd) * (hd[row] > 0)
            want_dz[row] = (want_dz[row] + g).to(dtype).double()          # stored in the compute dtype after every addition
            want_g += g
            want_dW += torch.outer(d, hd[row])
            want_db += d
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    _close(dz, want_dz, tol, tol * 0.1, "dz")
    _close(dW, want_dW, 1e-4, 1e-5, "dW")
    _close(db, want_db, 1e-5, 1e-6, "db")
    _close(gsum, want_g, 1e-4, 1e-5, "column sums")
    untouched = torch.ones(B * N, dtype=torch.bool)
    for b in range(B):
        untouched[b * N + torch.from_numpy(ids[b].astype(np.int64))] = False
    assert torch.equal(dz.cpu()[untouched], dz0.cpu()[untouched])


# ------------------------------------------------------------------ Adam over several flat runs in one launch
def test_adam_multi_matches_one_launch_per_run():
    """``murcl_adam_multi`` (all parameter groups of a step in one launch, train_MuRCL.py:165-171,293-295) against ``murcl_adam_step`` run
    by run: the same element-wise arithmetic (to the compiler's choice of fused multiply-adds: 2e-6 relative) - aligned runs (16-byte path), a run that starts at an odd
    element (scalar path), a run shorter than a chunk, different learning rates and step counts; g cleared when asked."""
    from murcl_amd import ops
    dev = _dev()
    sizes, lrs, steps = [3 * 4096 + 8, 4097, 130, 2 * 4096], [1e-3, 3e-4, 1e-2, 5e-5], [1, 7, 2, 1000]
    total = sum(sizes) + 1
    base = {k: _rand(9, f"am.{k}", (total,), 0.1 if k != "v" else 0.01).to(dev) for k in "pgmv"}
    base["v"] = base["v"].abs()
    offs, o = [], 1                                  # the first run starts at element 1: not 16-byte aligned
    for n in sizes:
        offs.append(o)
        o += n
    for wd, zero in ((0.0, True), (1e-2, False)):
        a = {k: t.clone() for k, t in base.items()}
        b = {k: t.clone() for k, t in base.items()}
        ops.adam_multi([(a["p"][o:o + n], a["g"][o:o + n], a["m"][o:o + n], a["v"][o:o + n], lr, st)
                        for o, n, lr, st in zip(offs, sizes, lrs, steps)], (0.9, 0.999), 1e-8, wd, zero_grad=zero)
        for o, n, lr, st in zip(offs, sizes, lrs, steps):
            ops.adam_step(b["p"][o:o + n], b["g"][o:o + n], b["m"][o:o + n], b["v"][o:o + n], lr, (0.9, 0.999), 1e-8, wd, st, zero_grad=zero)
        for k in "pmv":
            _close(a[k], b[k], 2e-6, 1e-9, k)
        assert torch.equal(a["g"], b["g"])
        assert bool((a["g"][1:] == 0).all()) == zero and torch.equal(a["p"][:1], base["p"][:1])


# ------------------------------------------------------------------ the random draws of a training step, one launch
def test_step_draws_are_valid_and_distributed_as_the_reference_draws():
    """``murcl_step_draws`` (train_MuRCL.py:235,256-258 ``torch.rand``; utils/datasets.py:265-267 lambda and ``torch.randperm``;
    models/rlmil.py:85-86 N(0,1) noise): value ranges, every row of perm a permutation, moments of the three distributions, position
    uniformity of the permutations, reproducibility from the seed and independence of the streams."""
    from murcl_amd import ops
    from murcl_amd.utils.datasets import draw_step
    dev = _dev()
    nu, nn_, V, B, alpha = 200_000, 200_000, 512, 64, 0.9
    uni, nrm, lam, perm = ops.step_draws(dev, nu, nn_, V, B, alpha, seed=1234)
    u, n, l, p = uni.cpu().double(), nrm.cpu().double(), lam.cpu().double(), perm.cpu().long()
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 4 * math.sqrt(1 / 12 / nu) and abs(float(u.var()) - 1 / 12) < 2e-3
    assert abs(float(n.mean())) < 4 / math.sqrt(nn_) and abs(float(n.var()) - 1.0) < 2e-2
    assert abs(float((n ** 4).mean()) - 3.0) < 0.15 and float(n.abs().max()) < 6.5          # kurtosis of a normal; Box-Muller's 24-bit tail
    assert float(l.min()) >= alpha and float(l.max()) <= 1.0 and abs(float(l.mean()) - (alpha + (1 - alpha) / 2)) < 1e-3
    assert torch.equal(p.sort(dim=1).values, torch.arange(B).expand(V, B))
    assert len({tuple(r.tolist()) for r in p}) == V
    # every bag is equally likely at every position: chi-square over the V draws of position 0 and of the position of bag 0
    for counts in (torch.bincount(p[:, 0], minlength=B).double(), torch.bincount((p == 0).nonzero()[:, 1], minlength=B).double()):
        chi2 = float(((counts - V / B) ** 2 / (V / B)).sum())
        assert chi2 < 63 + 5 * math.sqrt(2 * 63), chi2
    # fixed points of a uniform permutation: mean 1
    fixed = float((p == torch.arange(B)).double().sum(1).mean())
    assert abs(fixed - 1.0) < 0.25
    again = ops.step_draws(dev, nu, nn_, V, B, alpha, seed=1234)
    assert all(torch.equal(a, b) for a, b in zip((uni, nrm, lam, perm), again))
    other = ops.step_draws(dev, nu, nn_, V, B, alpha, seed=1235)
    assert not torch.equal(uni, other[0]) and not torch.equal(perm, other[3])
    assert abs(float(torch.corrcoef(torch.stack([uni[:nn_], nrm]))[0, 1])) < 0.01
    # the step-level helper: shapes, and a torch.manual_seed makes it reproducible
    torch.manual_seed(5)
    a = draw_step(dev, (2, 2, 8, 10), (5, 2, 8, 10), 12, 8, 0.9)
    assert a[0].shape == (2, 2, 8, 10) and a[1].shape == (5, 2, 8, 10) and len(a[2]) == 12
    assert a[2][3][0].shape == (8, 1) and a[2][3][1].shape == (8,) and a[2][3][1].dtype == torch.int32
    assert sorted(a[2][3][1].tolist()) == list(range(8))
    none = draw_step(dev, None, None, 3, 4, 1.0)
    assert none[0] is None and none[1] is None and bool((none[2].lam == 1.0).all())


def test_stack_lists_is_torch_stack_in_one_launch():
    """``murcl_stack_lists`` (PPO.update, rlmil.py:163-165): several lists of equally shaped tensors stacked by one launch, bit for bit;
    lists it does not cover (a non-contiguous member) fall back to ``torch.stack``."""
    from murcl_amd import ops
    dev = _dev()
    lists = [[_rand(3, f"sl.s{t}", (6, 512)).to(dev) for t in range(5)], [_rand(3, f"sl.a{t}", (6, 10)).to(dev) for t in range(5)],
             [_rand(3, f"sl.l{t}", (6,)).to(dev) for t in range(5)], [torch.arange(7, dtype=torch.int32, device=dev) + t for t in range(3)]]
    got = ops.stack_lists(lists)
    for g, l in zip(got, lists):
        assert g.dtype == l[0].dtype and torch.equal(g, torch.stack(l, 0))
    halves = _rand(3, "sl.h", (12, 512)).to(dev)
    got = ops.stack_lists([[halves[:6], halves[6:]], [halves.t()[:4], halves.t()[4:8]]])       # the second list is strided: fallback
    assert torch.equal(got[0], halves.view(2, 6, 512)) and torch.equal(got[1], torch.stack([halves.t()[:4], halves.t()[4:8]], 0))


@pytest.mark.parametrize("B,N,P", [(64, 4096, 16), (3, 1000, 16), (2, 5000, 4), (1, 7, 1)])
def test_softmax_rows_parts_equals_colsum_then_softmax(B, N, P):
    """``murcl_softmax_rows_parts`` (clam.py:144 on the gate epilogue's partial score rows): s = the column sum of the P rows, A = its
    soft-max over the N patches of each bag, against f64 (1e-6) - bags longer than the register-resident 4096 rows included."""
    from murcl_amd import ops
    dev = _dev()
    part = _rand(31, f"srp.{B}.{N}.{P}", (P, B * N), 2.0).to(dev)
    s, A = ops.softmax_rows_parts(part, B, N)
    want_s = part.double().sum(0).view(B, N)
    _close(s, want_s, 1e-6, 4e-6, "s")
    _close(A, torch.softmax(want_s, 1), 2e-5, 1e-9, "A")
    assert torch.allclose(A.sum(1).cpu(), torch.ones(B), atol=1e-5)
    # ... and the pooled rows cleared by the same launch, the pooling pass adding into them (no fill launch in between)
    X = _rand(31, f"srp.x{B}.{N}", (B, N, 64)).to(dev)
    Mz = torch.full((B, 64), 7.0, device=dev)
    s2, A2 = ops.softmax_rows_parts(part, B, N, zero=Mz)
    assert torch.equal(s2, s) and torch.equal(A2, A) and float(Mz.abs().max()) == 0.0
    got = ops.weighted_rowsum(X, A2.view(B, N, 1), into=Mz).view(B, 64)
    _close(got, torch.einsum("bn,bnd->bd", A2.double().cpu(), X.double().cpu()), 1e-5, 1e-6, "pooled rows")


def test_axpby_mean_small_and_copy_flat():
    """The bag-level odds and ends of the training step as own launches (round 6: no ATen kernel left in the stage-2 step):
    rewards = cos(t-1) - cos(t) (train_MuRCL.py:282-283), the mean of the T step losses (:291), the flat copy policy -> policy_old
    (rlmil.py:183) - against torch."""
    from murcl_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    sims = torch.randn((6, 64), generator=g, device=dev)
    r = ops.axpby(sims[:-1], sims[1:], 1.0, -1.0)
    assert torch.equal(r, sims[:-1] - sims[1:])
    np.testing.assert_allclose(ops.axpby(sims, sims, 0.25, 0.5).cpu().numpy(), (0.75 * sims).cpu().numpy(), rtol=1e-6)
    for n in (1, 6, 255, 256, 1000):
        x = torch.randn((n,), generator=g, device=dev)
        m = ops.mean_small(x)
        assert m.shape == () and abs(m.item() - x.double().mean().item()) <= 1e-6 * max(1.0, x.abs().max().item())
    for n in (3_681_291, 4, 3, 1_048_576):                                         # (the sampler's 3,681,291 parameters: a 12-byte tail)
        src = torch.randn((n,), generator=g, device=dev)
        dst = torch.zeros((n + 8,), device=dev)
        assert ops.copy_flat(dst[:n], src).data_ptr() == dst.data_ptr() and torch.equal(dst[:n], src) and (dst[n:] == 0).all()
    odd_s, odd_d = src[1:1000], torch.zeros((999,), device=dev)                     # a base that is not 16-byte aligned: the generic path
    ops.copy_flat(odd_d, odd_s)
    assert torch.equal(odd_d, odd_s)


def test_small_extent_products_without_aten_padding():
    """Classifier heads with a handful of outputs (train_RLMIL.py:316,502,709: nn.Linear(hidden, num_classes)): the input gradient
    dX = dY W (inner extent = the class count) runs as ONE small launch (murcl_gemm_nt_smallk), other awkward inner extents are
    zero-padded by ONE own launch per operand (murcl_pad_cols), the weight gradient dW = dY^T X adds into its buffer with an own
    launch - against float64 matmuls."""
    from murcl_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    for M, N, K in ((384, 1024, 2), (64, 512, 10), (7, 33, 1), (130, 96, 16), (96, 128, 40)):
        A = torch.randn((M, K), generator=g, device=dev)
        Bm = torch.randn((N, K), generator=g, device=dev)
        want = (A.double() @ Bm.double().t())
        C = ops.gemm_nt(A, Bm)
        _close(C, want, 1e-5, 1e-5 * float(want.abs().max()), f"gemm_nt {M}x{N}x{K}")
        base = torch.randn((M, N), generator=g, device=dev)
        C2 = ops.gemm_nt(A, Bm, out=base.clone(), accumulate=True)
        _close(C2, want + base.double(), 1e-5, 1e-5 * float(want.abs().max()), f"gemm_nt accumulate {M}x{N}x{K}")
    p = ops.pad_cols(torch.arange(12, device=dev, dtype=torch.float32).view(3, 4), 6)
    assert p.shape == (3, 6) and torch.equal(p[:, :4], torch.arange(12, device=dev, dtype=torch.float32).view(3, 4)) and (p[:, 4:] == 0).all()
    pb = ops.pad_cols(torch.ones((5, 3), device=dev).bfloat16(), 8)
    assert pb.dtype == torch.bfloat16 and (pb[:, :3] == 1).all() and (pb[:, 3:] == 0).all()
    for M, N1, N2 in ((384, 2, 1024), (64, 10, 512), (100, 1, 64)):
        dy = torch.randn((M, N1), generator=g, device=dev)
        x = torch.randn((M, N2), generator=g, device=dev)
        want = dy.double().t() @ x.double()
        _close(ops.gemm_tn(dy, x), want, 1e-5, 1e-5 * float(want.abs().max()), f"gemm_tn {M}x{N1}x{N2}")
        acc = torch.randn((N1, N2), generator=g, device=dev)
        cs = torch.randn((N1,), generator=g, device=dev)
        acc0, cs0 = acc.clone(), cs.clone()
        r = ops.gemm_tn(dy, x, out=acc, colsum_into=cs)
        assert r.data_ptr() == acc.data_ptr()
        _close(acc, want + acc0.double(), 1e-5, 1e-5 * float(want.abs().max()), f"gemm_tn into {M}x{N1}x{N2}")
        _close(cs, cs0.double() + dy.double().sum(0), 1e-5, 1e-4, "colsum_into")


def test_add_lists_accumulates_every_pair_in_one_launch():
    """murcl_add_lists: dst += src for a table of (src, dst) pairs of different lengths (the gradients a backward node returns, added to
    the optimizer's pre-seated gradient views: autograd's AccumulateGrad, one ATen add per parameter otherwise)."""
    from murcl_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(13)
    flat = torch.randn((9000,), generator=g, device=dev)
    views = [flat[:5], flat[5:5 + 4096].view(8, 512), flat[4101:4101 + 4097], flat[8198:8199]]
    before = [v.clone() for v in views]
    srcs = [torch.randn(v.shape, generator=g, device=dev) for v in views]
    rest = flat[8199:].clone()
    ops.add_lists(list(zip(srcs, views)) + [(None, flat[:1])])
    for v, b, s_ in zip(views, before, srcs):
        assert torch.equal(v, b + s_)
    assert torch.equal(flat[8199:], rest)
