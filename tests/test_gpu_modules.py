"""Module-level parity on MI355X: drop-in modules vs the CPU oracle on identical seeded inputs,
and vs the committed goldens that were produced by the reference itself.

fp32 path: <=1e-4 (north_star).  bf16 path: checked against the fp32 HIP path with the
tolerances written in each test.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, mil_oracle as O, params as P  # noqa: E402

T = torch.from_numpy


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _summ(g):
    g = g.detach().double().flatten().cpu()
    return np.concatenate([[g.norm().item(), g.abs().max().item()], g[:32].numpy()])


def _check_summ(got, want, rtol, key):
    np.testing.assert_allclose(got[:2], want[:2], rtol=rtol, err_msg=key)
    np.testing.assert_allclose(got[2:], want[2:], rtol=rtol, atol=rtol * want[1], err_msg=key)


def _abmil(seed, dtype=torch.float32):
    from murcl_amd.models.abmil import ABMIL
    m = ABMIL(512, L=512, D=128, dim_out=128)
    m.load_state_dict(P.to_torch(P.abmil(seed)))          # reference state-dict keys load unchanged
    m.compute_dtype = dtype
    return m.to(_dev())


def test_abmil_forward_backward_vs_reference_golden(golden):
    """G1: the reference's own outputs for the C1 shape (4 x 256 x 512)."""
    g = golden("g1_abmil")
    m = _abmil(985)
    x = T(P.bags(985, "g1.x", 4, 256, 512)).to(_dev())
    out, det = m(x)
    assert not det.requires_grad and out.shape == (4, 512)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.last_attention.cpu().numpy(), g["A"], rtol=1e-4, atol=1e-9)
    out.sum().backward()
    for k, v in m.named_parameters():
        key = "grad." + k
        if key not in g.files:
            continue
        if k == "attention.2.bias":
            assert v.grad.abs().max().item() < 1e-4 * m.attention[2].weight.grad.norm().item()
            continue
        _check_summ(_summ(v.grad), g[key], 3e-4, key)
    assert m.fc.weight.grad is None
    o1, _ = m(x[:1])
    np.testing.assert_allclose(o1.detach().cpu().numpy(), g["out_single"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,N", [(2, 300), (3, 1000), (8, 2048)])
def test_abmil_vs_oracle_full_grads(B, N):
    m = _abmil(3)
    p = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.abmil(3)).items()}
    x = T(P.bags(3, f"x{B}{N}", B, N, 512))
    w = T(detrand.normal(3, "w", (B, 512)))
    out_ref, A_ref, _, _ = O.abmil_forward(p, x)
    (out_ref * w).sum().backward()
    out, _ = m(x.to(_dev()))
    (out * w.to(_dev())).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.last_attention.cpu().numpy(), A_ref.detach().numpy(), rtol=1e-4, atol=1e-9)
    for k, v in m.named_parameters():
        if p[k].grad is None:
            continue
        ref = p[k].grad
        scale = ref.abs().max().item()
        if k == "attention.2.bias":
            continue
        np.testing.assert_allclose(v.grad.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=2e-4 * scale, err_msg=k)


def test_abmil_list_and_ragged_inputs():
    m = _abmil(4)
    dev = _dev()
    xs = [T(P.bags(4, f"r{i}", 1, n, 512)).to(dev) for i, n in enumerate((100, 257, 100))]
    out, _ = m(xs)                                            # list of [1,N_i,d]
    p = P.to_torch(P.abmil(4))
    for i, x in enumerate(xs):
        ref = O.abmil_forward(p, x.cpu())[0]
        np.testing.assert_allclose(out[i:i + 1].detach().cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    with pytest.raises(TypeError):
        m(3.0)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 8, 512))                             # CPU tensor: no fallback


def test_abmil_bf16_path_close_to_fp32_path():
    """bf16 storage of patch-level tensors: out within 2% of max, A within 3% rel, grads within 5% (norm-wise)."""
    dev = _dev()
    x = T(P.bags(5, "xb", 8, 2048, 512)).to(dev)
    w = T(detrand.normal(5, "w", (8, 512))).to(dev)
    m32, m16 = _abmil(5), _abmil(5, torch.bfloat16)
    o32, _ = m32(x)
    (o32 * w).sum().backward()
    o16, _ = m16(x)
    (o16 * w).sum().backward()
    assert (o16 - o32).abs().max().item() <= 2e-2 * o32.abs().max().item()
    np.testing.assert_allclose(m16.last_attention.cpu().numpy(), m32.last_attention.cpu().numpy(), rtol=3e-2)
    for (k, a), (_, b) in zip(m16.named_parameters(), m32.named_parameters()):
        if a.grad is None or k == "attention.2.bias":
            continue
        rel = ((a.grad - b.grad).norm() / b.grad.norm().clamp_min(1e-30)).item()
        assert rel < 5e-2, (k, rel)


def test_abmil_bf16_step_under_a_reduced_cu_budget():
    """murcl_set_cu_budget (csrc/runtime.hip): the persistent launches sized for 248 / 224 CUs (what a data-parallel run sets so
    that RCCL's channel workgroups find free CUs) compute what the 256-CU launches compute - the module output bit for bit (tile
    results do not depend on which workgroup forms them), the parameter gradients up to the order of their partial sums."""
    from murcl_amd import ops
    dev = _dev()
    x = T(P.bags(7, "xb", 16, 2048, 512)).to(dev)
    w = T(detrand.normal(7, "w", (16, 512))).to(dev)
    res = []
    try:
        for budget in (256, 248, 224):
            assert ops.set_cu_budget(budget) == budget and ops.cu_budget() == budget
            m = _abmil(7, torch.bfloat16)
            o, _ = m(x)
            (o * w).sum().backward()
            res.append((o.detach().clone(), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}))
    finally:
        assert ops.set_cu_budget(256) == 256
    assert ops.set_cu_budget(1000) == 256 and ops.set_cu_budget(250) == 248 and ops.set_cu_budget(256) == 256      # clamped, multiples of 8
    # the scoped form a data-parallel run uses by default (dist.reserve_cus_for_collectives): only the backward launches that the
    # head group's all-reduce overlaps are sized for 248 CUs, and the budget is back at 256 when backward returns
    from murcl_amd import functional
    try:
        functional.set_overlap_cu_budget(248)
        m = _abmil(7, torch.bfloat16)
        o, _ = m(x)
        assert ops.cu_budget() == 256
        (o * w).sum().backward()
        assert ops.cu_budget() == 256
        res.append((o.detach().clone(), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}))
    finally:
        functional.set_overlap_cu_budget(None)
    for o, g in res[1:]:
        assert torch.equal(o, res[0][0])
        for k, v in g.items():
            ref = res[0][1][k]
            assert (v - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-12, k


def _r16(t):
    """bf16 storage rounding, back in f32."""
    return t.bfloat16().float()


def test_bf16_forward_paths_against_the_oracle_with_the_same_storage_roundings():
    """What the bf16 path costs is its STORAGE roundings (patch features, the patch-level weight matrices, the hidden activations
    between layers), not its arithmetic: the f32 oracle fed the same rounded tensors at the same points - f32 accumulation and
    exact tanh / sigmoid otherwise - reproduces the HIP forward an order of magnitude closer than the 2e-2 the bf16-vs-f32
    comparisons allow.  ABMIL (C2's aggregator): the module output within 1e-4 of its largest entry - north_star's fp32 bound; measured
    6e-7 - attention 5e-3 relative (the K2 kernel feeds tanh outputs to the matrix cores in bf16: 2e-3 measured);
    CLAM-SB (training chain and forward-only chain): M, A and the raw scores within 2e-3 / 5e-3."""
    import torch.nn.functional as F
    dev = _dev()
    # ---- ABMIL
    B, N = 4, 2048
    p = P.to_torch(P.abmil(6))
    x = T(P.bags(6, "em.x", B, N, 512))
    h = _r16(x)
    for k in ("encoder.0", "encoder.3", "encoder.6"):
        h = _r16(torch.relu(F.linear(h, _r16(p[k + ".weight"]), p[k + ".bias"])))          # H1..H3 live in HBM as bf16
    s = F.linear(torch.tanh(F.linear(h, _r16(p["attention.0.weight"]), p["attention.0.bias"])), p["attention.2.weight"],
                 p["attention.2.bias"]).squeeze(-1)
    A_ref = torch.softmax(s, 1) / math.sqrt(N)
    M_ref = torch.einsum("bn,bnl->bl", A_ref, h)
    out_ref = torch.relu(F.linear(M_ref, p["decoder.0.weight"], p["decoder.0.bias"]))
    m16 = _abmil(6, torch.bfloat16)
    with torch.no_grad():
        out, _ = m16(x.to(dev))
    A = m16.last_attention.cpu().reshape(B, N)
    assert (out.cpu() - out_ref).abs().max().item() <= 1e-4 * out_ref.abs().max().item()
    np.testing.assert_allclose(A.numpy(), A_ref.numpy(), rtol=5e-3, atol=1e-3 * A_ref.max().item())
    # ---- CLAM-SB: forward-only chain (scores from the gate GEMM's f32 accumulators) and the chain that keeps U for a backward
    B, N = 3, 1024
    p = P.to_torch(P.clam_sb(7))
    x = T(P.bags(7, "em.x", B, N, 512))
    h = _r16(torch.relu(F.linear(_r16(x), _r16(p["attention_net.0.weight"]), p["attention_net.0.bias"])))
    a = torch.tanh(F.linear(h, _r16(p["attention_net.3.attention_a.0.weight"]), p["attention_net.3.attention_a.0.bias"]))
    g = torch.sigmoid(F.linear(h, _r16(p["attention_net.3.attention_b.0.weight"]), p["attention_net.3.attention_b.0.bias"]))
    s_ref = F.linear(a * g, p["attention_net.3.attention_c.weight"], p["attention_net.3.attention_c.bias"]).squeeze(-1)
    A_ref = torch.softmax(s_ref, 1)
    M_ref = torch.einsum("bn,bnl->bl", A_ref, h)
    m16 = _clam(7, False, torch.bfloat16)
    for grad in (False, True):
        with torch.set_grad_enabled(grad):
            M, A, s, _, _, _ = m16._run(x.to(dev), None, False)
        sc = s_ref.abs().max().item()
        assert (s.detach().cpu() - s_ref).abs().max().item() <= 2e-3 * sc, grad
        np.testing.assert_allclose(A.detach().cpu().numpy(), A_ref.numpy(), rtol=5e-3, atol=1e-3 * A_ref.max().item())
        assert (M.detach().cpu() - M_ref).abs().max().item() <= 2e-3 * M_ref.abs().max().item(), grad


def test_bf16_backward_against_the_oracle_with_the_same_storage_roundings():
    """The same statement for the C2 aggregator's gradients: autograd through the f32 oracle with the bf16 storage roundings of the HIP
    path inserted - activations H1..H3 (rounded forward, straight through backward) and the gradient tensors the backward kernels
    leave in HBM as bf16 (dZ3, dZ2, dZ1 = the gradients at the three pre-activations, and dT at the attention pre-activation) -
    against the HIP bf16 backward: every parameter gradient within 2e-3 of its largest entry (5e-2 norm-wise is what the
    bf16-vs-f32 comparison allows)."""
    import torch.nn.functional as F
    dev = _dev()
    B, N = 4, 2048
    p = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.abmil(8)).items()}
    x = T(P.bags(8, "emb.x", B, N, 512))
    w = T(detrand.normal(8, "emb.w", (B, 512)))

    def st(t):                                   # round forward, identity backward
        return t + (_r16(t) - t).detach()

    def grad_r16(t):                             # the gradient arriving at t is stored as bf16
        t.register_hook(lambda g: _r16(g))
        return t
    h = _r16(x)
    for k in ("encoder.0", "encoder.3", "encoder.6"):
        # (the bias gradients are column sums of the f32 values, taken before those are rounded for storage: the bias joins after the hook)
        z = grad_r16(F.linear(h, st(p[k + ".weight"]))) + p[k + ".bias"]
        h = st(torch.relu(z))
    t_pre = grad_r16(F.linear(h, st(p["attention.0.weight"]))) + p["attention.0.bias"]
    s = F.linear(torch.tanh(t_pre), p["attention.2.weight"], p["attention.2.bias"]).squeeze(-1)
    A = torch.softmax(s, 1) / math.sqrt(N)
    M = torch.einsum("bn,bnl->bl", A, h)
    out = torch.relu(F.linear(M, p["decoder.0.weight"], p["decoder.0.bias"]))
    (out * w).sum().backward()
    m16 = _abmil(8, torch.bfloat16)
    o16, _ = m16(x.to(dev))
    (o16 * w.to(dev)).sum().backward()
    for k, v in m16.named_parameters():
        if v.grad is None or k == "attention.2.bias":
            continue
        ref = p[k].grad
        err = (v.grad.cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-3, (k, err)


def test_full_layer_interleaved_hidden_golden(golden):
    """G9: the two views share one hidden state exactly like the reference's attribute."""
    from murcl_amd.models.rlmil import Full_layer
    g = golden("g9_full_layer")
    fc = Full_layer(512, 1024, True, 128)
    fc.load_state_dict(P.to_torch(P.full_layer(13)))
    fc = fc.to(_dev())
    with torch.no_grad():
        for t in range(3):
            for v in range(2):
                x = T(detrand.normal(13, f"g9.x.{t}.{v}", (4, 512))).to(_dev())
                z = fc(x, restart=(t == 0))
                np.testing.assert_allclose(z.cpu().numpy(), g[f"z.{t}.{v}"], rtol=1e-4, atol=2e-6)
                np.testing.assert_allclose(fc.hidden[0].cpu().numpy(), g[f"h.{t}.{v}"], rtol=1e-4, atol=2e-6)


def test_full_layer_cascade_golden(golden):
    """G21: ``Full_layer(fc_rnn=False)`` (rlmil.py:201-206,222-239; a constructor argument of the boundary, ``--fc_rnn`` defaults to
    True in both scripts) on the HIP kernels against the reference's outputs: the two restarts return None, the shared concatenation
    grows through fc_2 .. fc_5; logits, every classifier gradient entry and the input gradients at 1e-4; a sixth block raises."""
    from murcl_amd.models.rlmil import Full_layer
    g = golden("g21_full_layer_cascade")
    dev = _dev()
    fc = Full_layer(512, 1024, False, 16)
    fc.load_state_dict(P.to_torch(P.full_layer_cascade(21, 512, 16)))
    fc = fc.to(dev)
    assert sorted(fc.state_dict()) == sorted(f"fc_{k}.{n}" for k in (2, 3, 4, 5) for n in ("weight", "bias"))
    xs, loss = {}, 0.0
    for t in range(3):
        for v in range(2):
            x = T(detrand.normal(21, f"g21.x.{t}.{v}", (4, 512))).to(dev).requires_grad_()
            xs[(t, v)] = x
            z = fc(x, restart=(t == 0))
            assert (z is None) == bool(g[f"none.{t}.{v}"]) and fc.hidden.shape[1] == int(g[f"width.{t}.{v}"])
            if z is not None:
                np.testing.assert_allclose(z.detach().cpu().numpy(), g[f"z.{t}.{v}"], rtol=1e-4, atol=2e-6)
                loss = loss + (z * T(detrand.normal(21, f"g21.w.{t}.{v}", (4, 16))).to(dev)).sum()
    loss.backward()
    for k, p_ in fc.named_parameters():
        want = g["grad." + k]
        np.testing.assert_allclose(p_.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max(), err_msg=k)
    for (t, v), x in xs.items():
        want = g[f"dx.{t}.{v}"]
        got = x.grad.cpu().numpy() if x.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * max(1e-30, np.abs(want).max()), err_msg=f"dx {t} {v}")
    with pytest.raises(RuntimeError):
        fc(T(detrand.normal(21, "g21.x.3.0", (4, 512))).to(dev))


def test_full_layer_forward_views_equals_the_per_view_loop():
    """forward_views(restart=True) batches the independent views: same outputs, hidden state and gradients as
    the reference's `[fc(o, restart) for o in outputs]` loop (train_MuRCL.py:243); restart=False stays sequential."""
    from murcl_amd.models.rlmil import Full_layer
    dev = _dev()

    def run(batched):
        fc = Full_layer(512, 1024, True, 128)
        fc.load_state_dict(P.to_torch(P.full_layer(985)))
        fc = fc.to(dev)
        zs = []
        for t in range(2):
            xs = [T(detrand.normal(31, f"fv.x.{t}.{v}", (8, 512))).to(dev).requires_grad_() for v in range(2)]
            out = fc.forward_views(xs, restart=(t == 0)) if batched else [fc(x, restart=(t == 0)) for x in xs]
            zs += out
        loss = sum((z * z).sum() * (i + 1) for i, z in enumerate(zs))
        loss.backward()
        return [z.detach().cpu() for z in zs], fc.hidden.detach().cpu(), {k: v.grad.cpu() for k, v in fc.named_parameters()}

    za, ha, ga = run(True)
    zb, hb, gb = run(False)
    for a, b in zip(za, zb):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ha.numpy(), hb.numpy(), rtol=1e-5, atol=1e-6)
    for k in ga:
        np.testing.assert_allclose(ga[k].numpy(), gb[k].numpy(), rtol=2e-4, atol=1e-5 * float(gb[k].abs().max()) + 1e-7, err_msg=k)


def test_full_layer_forward_views_on_row_blocks_of_one_tensor():
    """CL.forward hands out `h.split(B)`: the batched head then consumes h itself (no concatenation) - same outputs,
    and the same gradient reaches h as through the per-view loop."""
    from murcl_amd.models.rlmil import Full_layer, _whole
    dev = _dev()
    h0 = T(detrand.normal(32, "fvb.h", (16, 512))).to(dev)

    def run(batched):
        fc = Full_layer(512, 1024, True, 128)
        fc.load_state_dict(P.to_torch(P.full_layer(985)))
        fc = fc.to(dev)
        leaf = h0.clone().requires_grad_()
        h = leaf * 1.0                                  # non-leaf, like an aggregator output
        xs = list(h.split(8, 0))
        assert (_whole(xs) is h) and _whole(xs[::-1]) is None and _whole(list(leaf.split(8, 0))) is None
        zs = fc.forward_views(xs, restart=True) if batched else [fc(x, restart=True) for x in xs]
        (sum((z * z).sum() * (i + 1) for i, z in enumerate(zs))).backward()
        return [z.detach().cpu() for z in zs], leaf.grad.cpu(), {k: v.grad.cpu() for k, v in fc.named_parameters()}

    za, da, ga = run(True)
    zb, db, gb = run(False)
    for a, b in zip(za, zb):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(da.numpy(), db.numpy(), rtol=2e-4, atol=1e-6)
    for k in ga:
        np.testing.assert_allclose(ga[k].numpy(), gb[k].numpy(), rtol=2e-4, atol=1e-5 * float(gb[k].abs().max()) + 1e-7, err_msg=k)


def test_full_layer_view_sequence_without_grad_equals_the_per_step_loop_and_the_grad_path():
    """forward_view_sequence over T = 4 patch steps x 2 views: under no_grad (frozen-aggregator stage 2: one own stacking launch, every
    hidden state written into one buffer, no saved gate tensors) == the reference's loop `[fc(x, restart=(t == 0)) for x in views]`
    step by step (train_MuRCL.py:243,272, shared hidden state included) == the autograd path, for separately allocated inputs and for
    row blocks of one tensor; the hidden state ends the same."""
    from murcl_amd.models.rlmil import Full_layer
    dev = _dev()
    fc = Full_layer(512, 1024, True, 128)
    fc.load_state_dict(P.to_torch(P.full_layer(985)))
    fc = fc.to(dev)
    Tn, B = 4, 8
    xs = [T(detrand.normal(33, f"fvs.x.{i}", (B, 512))).to(dev) for i in range(2 * Tn)]
    with torch.no_grad():
        loop = torch.cat([torch.cat(fc.forward_views(xs[2 * t:2 * t + 2], restart=(t == 0)), 0) for t in range(Tn)], 0)
        h_loop = fc.hidden.clone()
        fast = fc.forward_view_sequence(xs)
        h_fast = fc.hidden.clone()
        whole = torch.cat(xs, 0)
        fast_blocks = fc.forward_view_sequence(list(whole.split(B, 0)))
    grad = fc.forward_view_sequence([x.clone().requires_grad_() for x in xs])
    h_grad = fc.hidden.detach().clone()
    assert fast.shape == (2 * Tn * B, 128) and not fast.requires_grad and grad.requires_grad
    np.testing.assert_allclose(fast.cpu().numpy(), loop.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(fast, fast_blocks)
    np.testing.assert_allclose(fast.cpu().numpy(), grad.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(h_fast.cpu().numpy(), h_loop.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(h_fast.cpu().numpy(), h_grad.cpu().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("whole", [True, False])
def test_full_layer_view_sequence_node_equals_the_per_step_loop_with_gradients(whole):
    """forward_view_sequence under grad = ONE recurrent node over the 2T blocks (functional.GRUViewSeqFn: one input product, one input
    gradient, one weight / bias gradient launch each over all rows; blocks 0 and 1 from the zero state): outputs, the gradient that
    reaches every aggregator output and every parameter gradient equal the reference's per-step loop
    `[fc(x, restart=(t == 0)) for x in views]` (train_MuRCL.py:243,272), for row blocks of one tensor and for separate tensors; the step
    loss mean that comes out of the NT-Xent node back-propagates like `.mean()`."""
    from murcl_amd.models.rlmil import Full_layer
    from murcl_amd.utils.losses import NT_Xent
    from murcl_amd import ops
    dev = _dev()
    Tn, B = 3, 8
    x0 = T(detrand.normal(34, "fvn.x", (2 * Tn * B, 512))).to(dev)

    def run(seq):
        fc = Full_layer(512, 1024, True, 128)
        fc.load_state_dict(P.to_torch(P.full_layer(985)))
        fc = fc.to(dev)
        crit = NT_Xent(B, 1.0)
        if whole:
            leaf = x0.clone().requires_grad_()
            xs = list((leaf * 1.0).split(B, 0))
        else:
            leafs = [x0[i * B:(i + 1) * B].clone().requires_grad_() for i in range(2 * Tn)]
            xs = [l * 1.0 for l in leafs]
        if seq:
            z = fc.forward_view_sequence(xs).view(Tn, 2 * B, -1)
            loss_t, _ = crit.forward_steps(z)
            loss = crit.last_mean
            assert loss is not None and loss.requires_grad
            np.testing.assert_allclose(loss.item(), loss_t.mean().item(), rtol=1e-6)
            loss.backward(ops.unit_grad(loss))
        else:
            zs = [fc(x, restart=(i < 2)) for i, x in enumerate(xs)]
            z = torch.cat(zs, 0).view(Tn, 2 * B, -1)
            loss_t = torch.stack([crit(z[t, :B], z[t, B:]) for t in range(Tn)])
            loss_t.mean().backward()
        dx = leaf.grad if whole else torch.cat([l.grad for l in leafs], 0)
        return z.detach().cpu(), loss_t.detach().cpu(), fc.hidden.detach().cpu(), dx.cpu(), {k: v.grad.cpu() for k, v in fc.named_parameters()}

    za, la, ha, da, ga = run(True)
    zb, lb, hb, db, gb = run(False)
    np.testing.assert_allclose(za.numpy(), zb.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(la.numpy(), lb.numpy(), rtol=1e-5)
    np.testing.assert_allclose(ha.numpy(), hb.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(da.numpy(), db.numpy(), rtol=2e-4, atol=1e-5 * float(db.abs().max()))
    for k in ga:
        np.testing.assert_allclose(ga[k].numpy(), gb[k].numpy(), rtol=2e-4, atol=1e-5 * float(gb[k].abs().max()) + 1e-9, err_msg=k)


def test_flat_adam_refreshes_cached_weight_views_in_its_step():
    """Transposed f32 / compute-dtype views of optimizer-owned weights are rebuilt by FlatAdam.step (one launch for all
    of them); views of another optimizer's weights stay valid and untouched."""
    from murcl_amd import ops
    from murcl_amd.optim import FlatAdam
    dev = _dev()
    a, b = torch.nn.Linear(96, 40).to(dev), torch.nn.Linear(33, 70).to(dev)
    oa, ob = FlatAdam([{"params": list(a.parameters()), "lr": 1e-1}]), FlatAdam([{"params": list(b.parameters()), "lr": 1e-1}])
    ta, tb = ops.transposed(a.weight), ops.transposed(b.weight)
    ca = ops.weight_views([(a.weight, False, torch.bfloat16)])[0]
    assert torch.equal(ta, a.weight.t()) and torch.equal(tb, b.weight.t())
    for step in range(2):
        a.weight.grad.fill_(1.0)
        oa.mark_all_touched()                  # gradients written by hand, not by a backward pass
        b_before = b.weight.detach().clone()
        oa.step()
        assert torch.equal(ta, a.weight.detach().t()) and torch.equal(ca, a.weight.detach().bfloat16())   # refreshed in place
        assert ops.transposed(a.weight).data_ptr() == ta.data_ptr()
        assert ops.transposed(b.weight).data_ptr() == tb.data_ptr() and torch.equal(tb, b_before.t())
    b.weight.grad.fill_(-1.0)
    ob.mark_all_touched()
    ob.step()
    assert torch.equal(ops.transposed(b.weight), b.weight.detach().t()) and not torch.equal(tb, b_before.t())
    with torch.no_grad():
        a.weight.mul_(2.0)                                # an in-place torch update is picked up at the next use
    assert torch.equal(ops.transposed(a.weight), a.weight.detach().t())
    w = torch.randn(5, 7, device=dev)                     # unmanaged: a fresh transpose every time
    assert torch.equal(ops.transposed(w), w.t())


@pytest.mark.parametrize("Tn", [1, 3])
def test_pretrain_step_golden(golden, Tn):
    """G3: CL(ABMIL) + Full_layer + NT_Xent over T patch-steps: losses, rewards, gradients vs the reference."""
    from murcl_amd.models.cl import CL
    from murcl_amd.models.rlmil import Full_layer
    from murcl_amd.utils.losses import NT_Xent
    g = golden("g3_pretrain")
    dev = _dev()
    model = CL(_abmil(985), projection_dim=128, n_features=512)
    fc = Full_layer(512, 1024, True, 128)
    fc.load_state_dict(P.to_torch(P.full_layer(985)))
    fc = fc.to(dev)
    crit = NT_Xent(4, 1.0)
    losses, rewards, sim_last = [], [], None
    for t in range(Tn):
        xv = [T(P.bags(985, f"g3.x.{t}.{v}", 4, 256, 512)).to(dev) for v in range(2)]
        outs, states = model(xv)
        outs = [fc(o, restart=(t == 0)) for o in outs]
        losses.append(crit(outs[0], outs[1]))
        sim = crit.last_similarity
        if t > 0:
            rewards.append((sim_last - sim).cpu().numpy())
        sim_last = sim
    loss = sum(losses) / Tn
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"T{Tn}.loss"], rtol=1e-4)
    np.testing.assert_allclose([l.item() for l in losses], g[f"T{Tn}.losses"], rtol=1e-4)
    if Tn > 1:
        np.testing.assert_allclose(np.stack(rewards), g[f"T{Tn}.rewards"], rtol=2e-3, atol=2e-6)
    for k, v in model.named_parameters():
        key = f"T{Tn}.grad.{k}"
        if key in g.files and not k.endswith("attention.2.bias"):
            _check_summ(_summ(v.grad), g[key], 2e-3, key)
    for k, v in fc.named_parameters():
        _check_summ(_summ(v.grad), g[f"T{Tn}.grad.fc::{k}"], 2e-3, k)


def test_ntxent_module_golden(golden):
    from murcl_amd.utils.losses import NT_Xent
    g = golden("g2_ntxent")
    dev = _dev()
    for B in (2, 4, 64):
        for tau in (1.0, 0.5):
            zi = T(detrand.normal(7, f"g2.zi.{B}", (B, 128))).to(dev).requires_grad_()
            zj = T(detrand.normal(7, f"g2.zj.{B}", (B, 128))).to(dev).requires_grad_()
            loss = NT_Xent(B, tau)(zi, zj)
            loss.backward()
            np.testing.assert_allclose(loss.item(), g[f"loss.{B}.{tau}"], rtol=1e-4)
            np.testing.assert_allclose(zi.grad.cpu().numpy(), g[f"dzi.{B}.{tau}"], rtol=1e-3, atol=1e-7)
            np.testing.assert_allclose(zj.grad.cpu().numpy(), g[f"dzj.{B}.{tau}"], rtol=1e-3, atol=1e-7)


# ------------------------------------------------------------------ DSMIL (K6)
def _dsmil(seed, d=512, C=2, dtype=torch.float32):
    from murcl_amd.models.dsmil import build_dsmil
    m = build_dsmil(d, C)
    m.load_state_dict(P.to_torch(P.dsmil(seed, d, C)))
    m.compute_dtype = dtype
    return m.to(_dev())


def test_dsmil_vs_reference_golden(golden):
    """G5: classes, arg-max patch ids (bit-exact), bag embedding and gradients vs the reference."""
    g = golden("g5_dsmil")
    m = _dsmil(5)
    x = T(P.bags(5, "g5.x", 3, 200, 512)).to(_dev())
    classes, bag, det = m(x)
    assert isinstance(classes, list) and len(classes) == 3 and not det.requires_grad
    np.testing.assert_allclose(torch.stack(classes).detach().cpu().numpy(), g["classes"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(m.last_critical.cpu().numpy(), g["m_ids"])
    np.testing.assert_allclose(bag.detach().cpu().numpy(), g["bag"], rtol=1e-4, atol=1e-5)
    (bag.sum() + sum(c.max(0)[0].sum() for c in classes)).backward()
    for k, v in m.named_parameters():
        key = "grad." + k
        if key in g.files:
            _check_summ(_summ(v.grad), g[key], 1e-3, key)
    assert m.b_classifier.fcc.weight.grad is None
    c1, b1, _ = m(x[:1])                                           # single-bag path returns a tensor
    np.testing.assert_allclose(c1.detach().cpu().numpy(), g["classes_single"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b1.detach().cpu().numpy(), g["bag_single"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,N,d,C", [(2, 1000, 1024, 2), (4, 333, 512, 3)])
def test_dsmil_vs_oracle_full_grads(B, N, d, C):
    m = _dsmil(6, d, C)
    p = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.dsmil(6, d, C)).items()}
    x = T(P.bags(6, f"x{B}{N}", B, N, d))
    wb_, wc_ = T(detrand.normal(6, "wb", (B, C, d))), T(detrand.normal(6, "wc", (B, N, C)))
    c_ref, bag_ref, A_ref, m_ref = O.dsmil_forward(p, x)
    ((bag_ref * wb_).sum() + (c_ref * wc_).sum()).backward()
    classes, bag, _ = m(x.to(_dev()))
    np.testing.assert_array_equal(m.last_critical.cpu().numpy(), m_ref.numpy())
    ((bag * wb_.to(_dev())).sum() + (torch.stack(classes) * wc_.to(_dev())).sum()).backward()
    np.testing.assert_allclose(bag.detach().cpu().numpy(), bag_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    for k, v in m.named_parameters():
        if p[k].grad is None:
            continue
        ref = p[k].grad
        np.testing.assert_allclose(v.grad.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=3e-4 * ref.abs().max().item(), err_msg=k)


def test_dsmil_dropout_v_training_mode_vs_reference_golden(golden):
    """G23: ``BClassifier(dropout_v=0.25)`` while training (dsmil.py:53-59,66; no script sets it, rounds 1-5 refused it): the value
    branch's Dropout with an injected keep mask against the reference's own forward - classes, bag embedding, every stored gradient;
    then the seeded form: every training-mode call draws a new mask, eval mode equals dropout_v = 0."""
    from murcl_amd.models.dsmil import BClassifier, FCLayer, MILNet
    g = golden("g23_dsmil_dropout_v")
    dev = _dev()
    m = MILNet(FCLayer(512, 2), BClassifier(512, 2, dropout_v=0.25))
    m.load_state_dict(P.to_torch(P.dsmil(23, 512, 2)))
    m = m.to(dev).train()
    x = T(P.bags(23, "g23.x", 3, 200, 512)).to(dev)
    keep = ((detrand.uniform(23, "g23.keep", (3, 200, 512)) >= 0.25).astype(np.float32) / np.float32(0.75)).astype(np.float32)
    m.keep_mask_v = T(keep).to(dev)
    classes, bag, _ = m(x)
    np.testing.assert_allclose(torch.stack(classes).detach().cpu().numpy(), g["classes"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bag.detach().cpu().numpy(), g["bag"], rtol=1e-4, atol=1e-5)
    wb_, wc_ = T(detrand.normal(23, "g23.wb", (3, 2, 512))).to(dev), T(detrand.normal(23, "g23.wc", (3, 200, 2))).to(dev)
    ((bag * wb_).sum() + (torch.stack(classes) * wc_).sum()).backward()
    _check_grad_entries(m.named_parameters(), g, "grad.", 3e-4)
    m.keep_mask_v = None
    b1 = m(x)[1].detach().clone()
    b2 = m(x)[1].detach().clone()
    # its own draws (seed = torch's global seed + a call counter: every call a new mask), around the un-dropped value in the mean
    assert (b1 - b2).abs().max().item() > 1e-3 and (b1 - bag.detach()).abs().max().item() > 1e-3
    m.eval()
    m0 = _dsmil(23, 512, 2)
    assert torch.equal(m(x)[1], m0(x)[1])                                                     # eval mode: Dropout is the identity


def test_dsmil_bf16_and_ragged_list():
    dev = _dev()
    m32, m16 = _dsmil(7), _dsmil(7, dtype=torch.bfloat16)
    x = T(P.bags(7, "x", 4, 2048, 512)).to(dev)
    c32, b32, _ = m32(x)
    c16, b16, _ = m16(x)
    assert (b16 - b32).abs().max().item() <= 3e-2 * b32.abs().max().item()
    xs = [T(P.bags(7, f"r{i}", 1, n, 512)).to(dev) for i, n in enumerate((100, 257))]
    cl, bag, _ = m32(xs)
    assert [tuple(c.shape) for c in cl] == [(100, 2), (257, 2)] and bag.shape == (2, 2, 512)
    with pytest.raises(TypeError):
        m32(3)


# ------------------------------------------------------------------ CLAM-SB (K4/K5)
def _clam(seed, subtyping, dtype=torch.float32):
    from murcl_amd.models.clam import CLAM_SB
    m = CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=subtyping, in_dim=512)
    m.load_state_dict(P.to_torch(P.clam_sb(seed)))               # reference keys (dropout=True layout) load as is
    m.compute_dtype = dtype
    return m.to(_dev()).eval()


@pytest.mark.parametrize("subtyping", [False, True])
def test_clam_vs_reference_golden(golden, subtyping):
    """G4: pooled M, attention, raw scores, top-k ids (bit-exact), instance loss / preds / targets."""
    g = golden("g4_clam")
    tag = f"sub{int(subtyping)}"
    m = _clam(11, subtyping)
    x = T(P.bags(11, "g4.x", 3, 300, 512)).to(_dev())
    M, det = m(x)
    np.testing.assert_allclose(M.detach().cpu().numpy(), g[f"{tag}.M_batch"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.last_attention.cpu().numpy(), g[f"{tag}.A"], rtol=2e-4, atol=1e-9)
    raw = torch.cat([m.bag_forward(x[b], attention_only=True) for b in range(3)])
    np.testing.assert_allclose(raw.cpu().numpy(), g[f"{tag}.raw"], rtol=1e-4, atol=2e-5)
    for label in (0, 1):
        M2, _, res = m(x, label=[label] * 3, instance_eval=True)
        np.testing.assert_allclose(M2.detach().cpu().numpy(), g[f"{tag}.l{label}.M"], rtol=1e-4, atol=1e-5)
        for b in range(3):
            np.testing.assert_allclose(float(res[b]["instance_loss"]), g[f"{tag}.l{label}.inst_loss"][b], rtol=2e-4)
            np.testing.assert_array_equal(res[b]["inst_preds"], g[f"{tag}.l{label}.preds"][b])
            np.testing.assert_array_equal(res[b]["inst_labels"], g[f"{tag}.l{label}.targets"][b])
    from murcl_amd import ops
    ids = ops.topk_ids(m.last_attention, 8).cpu().numpy()
    np.testing.assert_array_equal(ids[:, :8], g[f"{tag}.top_p"])
    np.testing.assert_array_equal(ids[:, 8:], g[f"{tag}.top_n"])


@pytest.mark.parametrize("name", ["ce_weighted_sum", "multi_margin", "lambda_logit_gap"])
@pytest.mark.parametrize("label", [0, 1])
def test_clam_custom_instance_loss_vs_reference_golden(golden, name, label):
    """G22: ``CLAM_SB(instance_loss_fn=<callable>)`` (clam.py:64-65,118,131 call whatever loss the constructor was given; rounds 4-5
    refused anything but the default CE): the HIP kernels gather the top / bottom rows and form logits, targets and predictions, the
    caller's loss runs on each evaluated (bag, class) pair's [rows,2] logits and its own gradient goes back through the instance
    backward - per-bag instance losses at 2e-4 and every stored gradient against the reference's, three non-default losses."""
    from murcl_amd.models.clam import CLAM_SB
    g = golden("g22_clam_custom_instance_loss")
    losses = {"ce_weighted_sum": torch.nn.CrossEntropyLoss(weight=torch.tensor([0.7, 1.3]), reduction="sum"),
              "multi_margin": torch.nn.MultiMarginLoss(),
              "lambda_logit_gap": lambda lg, tg: ((lg[:, 1] - lg[:, 0]) * (1.0 - 2.0 * tg.float())).exp().mean()}
    m = CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, instance_loss_fn=losses[name], subtyping=True, in_dim=512)
    m.load_state_dict(P.to_torch(P.clam_sb(11)), strict=False)      # (a weighted CE module carries its own "instance_loss_fn.weight")
    m = m.to(_dev()).eval()
    x = T(P.bags(11, "g4.x", 3, 300, 512)).to(_dev())
    M, _, res = m(x, label=[label] * 3, instance_eval=True)
    for b in range(3):
        np.testing.assert_allclose(float(res[b]["instance_loss"].detach()), g[f"{name}.l{label}.inst_loss"][b], rtol=2e-4)
    (M.sum() + sum(r["instance_loss"] for r in res)).backward()
    _check_grad_entries(m.named_parameters(), g, f"{name}.l{label}.grad.", 3e-4, zero_keys=("attention_net.3.attention_c.bias",))


def test_clam_grads_golden(golden):
    g = golden("g4_clam")
    m = _clam(11, True)
    x = T(P.bags(11, "g4.x", 3, 300, 512)).to(_dev())
    M, _, res = m(x, label=[1, 1, 1], instance_eval=True)
    (M.sum() + sum(r["instance_loss"] for r in res)).backward()
    for k, v in m.named_parameters():
        key = "grad." + k
        if key in g.files and not k.endswith("attention_c.bias"):
            _check_summ(_summ(v.grad), g[key], 2e-3, key)


def test_clam_train_mode_dropout_masks_vs_oracle():
    """Training-mode Dropout(0.25) with injected keep masks vs the oracle's drop_mask path."""
    from murcl_amd.functional import CLAMFn
    dev = _dev()
    B, N = 2, 256
    p = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.clam_sb(12)).items()}
    x = T(P.bags(12, "x", B, N, 512))
    masks = [T((detrand.uniform(12, f"m{i}", (B, N, w)) >= 0.25).astype(np.float32)) for i, w in enumerate((512, 256, 256))]
    M_ref, A_ref, _, _ = O.clam_sb_forward(p, x, drop_mask=masks)
    w = T(detrand.normal(12, "w", (B, 512)))
    (M_ref * w).sum().backward()
    m = _clam(12, False)
    keeps = tuple((mk.reshape(B * N, -1) / 0.75).to(dev).contiguous() for mk in masks)
    M, A, s, il, ids, _ = m._run(x.to(dev), None, False, keeps)
    (M * w.to(dev)).sum().backward()
    np.testing.assert_allclose(M.detach().cpu().numpy(), M_ref.detach().numpy(), rtol=2e-4, atol=1e-5)
    for k, v in m.named_parameters():
        if p[k].grad is not None and not k.endswith("attention_c.bias"):
            ref = p[k].grad
            np.testing.assert_allclose(v.grad.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=3e-4 * ref.abs().max().item(), err_msg=k)


def test_clam_bf16_path_and_cl_wrapper():
    from murcl_amd.models.cl import CL
    dev = _dev()
    m32, m16 = _clam(13, False), _clam(13, False, torch.bfloat16)
    xs = [T(P.bags(13, f"v{v}", 4, 1024, 512)).to(dev) for v in range(2)]
    h32, _ = CL(m32, 128, 512)(xs)
    h16, _ = CL(m16, 128, 512)(xs)
    for a, b in zip(h16, h32):
        assert a.shape == (4, 512) and (a - b).abs().max().item() <= 3e-2 * b.abs().max().item()


def test_clam_fused_gate_score_forward_only_matches_the_unfused_chain_and_the_oracle(monkeypatch):
    """in forward-only calls (validation, heat-map scoring, stage 2) the gated attention score comes out of
    the gate GEMM's epilogue (murcl_panel_gemm epilogue 4) without materialising the [B*N, 2D] pre-activations: raw scores,
    soft-max, pooled vector and top-k ids against the un-fused bf16 chain and against the f32 oracle."""
    from murcl_amd import functional, ops
    dev = _dev()
    B, N = 3, 2048
    x = T(P.bags(15, "fg.x", B, N, 512)).to(dev)
    m16 = _clam(15, True, torch.bfloat16).eval()
    seen = []
    real = ops.panel_gate_score
    monkeypatch.setattr(ops, "panel_gate_score", lambda *a, **k: (seen.append(1), real(*a, **k))[1])
    with torch.no_grad():
        raw_f = m16.bag_forward(x[0], attention_only=True)                       # fused (no grad)
        M_f, A_f, s_f, _, ids_f, _ = m16._run(x, torch.tensor([1, 0, 1], device=dev), True)
    assert len(seen) == 2
    monkeypatch.setattr(functional, "_FUSED_GATE", False)
    with torch.no_grad():
        raw_u = m16.bag_forward(x[0], attention_only=True)
        M_u, A_u, s_u, _, ids_u, _ = m16._run(x, torch.tensor([1, 0, 1], device=dev), True)
    assert len(seen) == 2
    sc = s_u.abs().max().item()
    assert (s_f - s_u).abs().max().item() <= 2e-2 * sc and (raw_f - raw_u).abs().max().item() <= 2e-2 * sc
    assert (M_f - M_u).abs().max().item() <= 2e-2 * M_u.abs().max().item()
    # against the f32 oracle the fused path (f32 accumulators into tanh / sigmoid) must be no worse than the un-fused one
    # (which rounds the pre-activations to bf16 first)
    p = P.to_torch(P.clam_sb(15))
    Mo, Ao, so, _ = O.clam_sb_forward(p, x.cpu())
    e_f, e_u = (s_f.cpu() - so).abs().max().item(), (s_u.cpu() - so).abs().max().item()
    assert e_f <= 1.5 * e_u + 1e-3 * so.abs().max().item(), (e_f, e_u)
    # a call that needs gradients keeps the un-fused chain (the backward pass reads the pre-activations)
    monkeypatch.setattr(functional, "_FUSED_GATE", True)
    m16.train()
    out = m16._run(x, None, False)[0]
    out.sum().backward()
    assert len(seen) == 2 and m16.attention_net[0].weight.grad is not None


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_clam_training_chain_in_the_gemm_epilogues_matches_the_separate_passes(monkeypatch, mode):
    """Round 3: with gradients on, the bf16 CLAM-SB chain takes the gate score (and its seeded Dropouts) from the gate GEMM's
    epilogue, the first layer's Dropout from its epilogue, the pooling / soft-max / gate backward from one pass, the instance
    branch from one launch each way.  Same inputs, same dropout seeds: pooled vector, attention, instance loss, top-k ids and
    every parameter gradient against the chain of separate passes (functional switches off), to bf16 precision - and the new chain
    must be no further from the f32 path than the old one (eval mode, where both are deterministic)."""
    from murcl_amd import functional, ops
    dev = _dev()
    B, N = 4, 1024
    x = T(P.bags(21, "tc.x", B, N, 512)).to(dev)
    labels = torch.tensor([1, 0, 1, 1], device=dev)
    w = T(detrand.normal(21, "tc.w", (B, 512))).to(dev)

    def run(dtype):
        m = _clam(21, True, dtype)
        m.train(mode == "train")
        keeps = None
        if mode == "train":
            keeps = (ops.DropSeed(0.75, seed=101), ops.DropSeed(0.75, seed=202), ops.DropSeed(0.75, seed=303))
        M, A, s, il, ids, _ = m._run(x if dtype == torch.float32 else x.bfloat16(), labels, True, keeps)
        ((M * w).sum() + il.sum()).backward()
        return M.detach(), A.detach(), il.detach(), ids, {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}

    new = run(torch.bfloat16)
    for name in ("_GATE_U", "_FUSED_FC_DROP", "_FUSED_INST"):
        monkeypatch.setattr(functional, name, False)
    old = run(torch.bfloat16)
    ref = run(torch.float32) if mode == "eval" else None

    def rel(a, b):
        return (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-30)
    assert rel(new[0], old[0]) <= 2e-2 and rel(new[1], old[1]) <= 5e-2 and rel(new[2], old[2]) <= 2e-2
    assert set(new[4]) == set(old[4])
    for k in new[4]:
        if not k.endswith("attention_c.bias"):          # (its gradient is a sum that cancels to rounding noise)
            assert rel(new[4][k], old[4][k]) <= 6e-2, (k, rel(new[4][k], old[4][k]))
    if ref is not None:
        assert torch.equal(new[3], ref[3]) or (new[3] != ref[3]).float().mean().item() <= 0.1     # ids: near-ties may flip in bf16
        assert rel(new[0], ref[0]) <= 1.5 * rel(old[0], ref[0]) + 2e-3
        for k in new[4]:
            if not k.endswith("attention_c.bias"):
                assert rel(new[4][k], ref[4][k]) <= 1.5 * rel(old[4][k], ref[4][k]) + 1e-2, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dsmil_reassociated_attention_equals_the_literal_order(monkeypatch, dtype):
    """Round 3: K6 without the [B*N,128] queries (logits = X . Wq^T q_max / sqrt(128), one streaming pass each for attention +
    pooling and for their backward) against the reference's literal order (queries by one GEMM over all patches): classes, bag
    vectors, arg-max ids and every parameter gradient.  f32: 1e-4 of the largest entry (north_star); bf16 storage: 2e-2."""
    from murcl_amd import functional
    dev = _dev()
    B, N, d = 3, 2048, 1024
    x = T(P.bags(22, "dr.x", B, N, d)).to(dev).to(dtype)

    def run():
        m = _dsmil(22, d=d, dtype=dtype)
        classes, bag = m._run(x)
        (bag.sum() + 0.5 * (bag * bag).sum() + classes.max(1)[0].sum()).backward()
        return classes.detach(), bag.detach(), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}

    outs = {}
    for name, re, one in (("onepass", True, True), ("reassoc", True, False), ("literal", False, False)):
        monkeypatch.setattr(functional, "_DSMIL_REASSOC", re)
        monkeypatch.setattr(functional, "_DSMIL_ONEPASS", one)
        outs[name] = run()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    lit = outs["literal"]
    for name in ("onepass", "reassoc"):
        got = outs[name]
        assert torch.equal(got[0], lit[0])                                   # the instance scores are computed identically
        assert (got[1] - lit[1]).abs().max().item() <= tol * lit[1].abs().max().item()
        assert set(got[2]) == set(lit[2])
        for k in got[2]:
            scale = max(lit[2][k].abs().max().item(), 1e-6 * max(v.abs().max().item() for v in lit[2].values()))
            assert (got[2][k] - lit[2][k]).abs().max().item() <= tol * scale, (name, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("also_dense", [False, True])
def test_dsmil_max_instance_scores_come_out_of_the_argmax_launch(dtype, also_dense, monkeypatch):
    """Round 4: ``_run(want_max=True)`` returns the max-instance class scores (train_RLMIL.py:516, ``torch.max(outputs_ins, 0)``) as its
    own differentiable output: equal to ``classes.max(1)[0]`` bit for bit, and a loss on it gives the parameter gradients of the dense
    route (ATen max -> scatter into a [B,N,C] gradient -> the streaming backward with dcls) - alone and together with a dense term."""
    dev = _dev()
    B, N, d = 3, 1024, 1024
    x = T(P.bags(23, "dm.x", B, N, d)).to(dev).to(dtype)
    w = T(detrand.normal(23, "dm.w", (B, 2))).to(dev)

    def run(route):
        m = _dsmil(23, d=d, dtype=dtype)
        if route == "max":
            classes, bag, cmax = m._run(x, want_max=True)
        else:
            classes, bag = m._run(x)
            cmax = classes.max(1)[0]
        loss = bag.sum() + (cmax * w).sum()
        if also_dense:
            loss = loss + 0.01 * (classes * classes).sum()
        loss.backward()
        return cmax.detach(), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}

    cm_a, g_a = run("max")
    cm_b, g_b = run("dense")
    assert torch.equal(cm_a, cm_b)
    assert set(g_a) == set(g_b)
    tol = 1e-4 if dtype == torch.float32 else 2e-3
    for k in g_a:
        scale = max(g_b[k].abs().max().item(), 1e-6 * max(v.abs().max().item() for v in g_b.values()))
        assert (g_a[k] - g_b[k]).abs().max().item() <= tol * scale, k


# ------------------------------------------------------------------ PPO (K10/K11)
def test_ppo_act_evaluate_update_vs_reference_golden(golden):
    """G8: act (injected eps) -> actions/logp/hidden; evaluate; one update (K_epochs=1, Adam lr 1e-3) -> parameters."""
    from murcl_amd.models.rlmil import PPO, Memory
    g = golden("g8_ppo")
    dev = _dev()
    seed, B, S_, H, K, Tm, std = 21, 6, 512, 512, 10, 3, 0.5
    ppo = PPO(512, S_, H, False, action_std=std, lr=1e-3, gamma=0.1, K_epochs=1, action_size=K)
    sd = P.to_torch(P.actor_critic(seed, S_, H, K))
    ppo.policy.load_state_dict(sd)
    ppo.policy_old.load_state_dict(sd)
    mem = Memory()
    for t in range(Tm):
        st = T(detrand.normal(seed, f"g8.s{t}", (B, S_))).to(dev)
        eps = T(detrand.normal(seed, f"g8.e{t}", (B, K))).to(dev)
        a = ppo.select_action(st, mem, restart_batch=(t == 0), eps=eps)
        np.testing.assert_allclose(a.cpu().numpy(), g[f"act.{t}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(mem.logprobs[-1].cpu().numpy(), g[f"logp.{t}"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(mem.hidden[-1][0].cpu().numpy(), g[f"hidden.{t}"], rtol=1e-3, atol=1e-5)
        mem.rewards.append((T(detrand.normal(seed, f"g8.r{t}", (1, B))) * 0.1).to(dev))
    with torch.no_grad():
        lp, v, ent = ppo.policy.evaluate(torch.stack(mem.states, 0), torch.stack(mem.actions, 0))
    np.testing.assert_allclose(lp.cpu().numpy(), g["eval.logp"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(v.cpu().numpy(), g["eval.value"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), g["eval.entropy"], rtol=1e-6)
    ppo.update(mem)
    for k, v in ppo.policy.state_dict().items():
        np.testing.assert_allclose(_summ(v), g["post." + k], rtol=5e-4, atol=5e-6, err_msg=k)
    for a, b in zip(ppo.policy.parameters(), ppo.policy_old.parameters()):
        assert torch.equal(a, b)


def test_ppo_two_views_in_one_policy_step_match_per_view_calls():
    """PPO.select_actions (both views of a patch step as the rows of ONE policy step) fills each view's Memory with what two
    select_action calls would: actions, log-probs, hidden states, stored states - every row depends on its own row only."""
    from murcl_amd.models.rlmil import PPO, Memory
    dev = _dev()
    seed, B, S_, H, K = 23, 64, 512, 512, 10
    ppo = PPO(512, S_, H, False, action_std=0.5, action_size=K)
    sd = P.to_torch(P.actor_critic(seed, S_, H, K))
    ppo.policy_old.load_state_dict(sd)
    one, two = [Memory(), Memory()], [Memory(), Memory()]
    for t in range(3):
        st = T(detrand.normal(seed, f"v.s{t}", (2 * B, S_))).to(dev)
        eps = T(detrand.normal(seed, f"v.e{t}", (2, B, K))).to(dev)
        states = list(st.split(B, 0))
        acts = ppo.select_actions(states, two, restart_batch=(t == 0), eps=[eps[0], eps[1]])
        for v in range(2):
            a = ppo.select_action(states[v], one[v], restart_batch=(t == 0), eps=eps[v])
            np.testing.assert_allclose(acts[v].cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(two[v].logprobs[-1].cpu().numpy(), one[v].logprobs[-1].cpu().numpy(), rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(two[v].hidden[-1].cpu().numpy(), one[v].hidden[-1].cpu().numpy(), rtol=1e-4, atol=1e-5)
            assert two[v].hidden[-1].shape == one[v].hidden[-1].shape and len(two[v].hidden) == len(one[v].hidden)
            assert torch.equal(two[v].states[-1], one[v].states[-1]) and torch.equal(two[v].actions[-1], acts[v])


def test_ppo_returns_and_loss_vs_oracle():
    from murcl_amd import ops
    dev = _dev()
    Tn, B = 5, 64
    rw = [T(detrand.normal(22, f"r{t}", (1, B)) * 0.1) for t in range(Tn)]
    R_ref = O.ppo_returns(rw, 0.1)
    R = ops.ppo_returns(torch.cat(rw, 0).to(dev), 0.1)
    np.testing.assert_allclose(R.cpu().numpy(), R_ref.numpy(), rtol=1e-4, atol=1e-5)
    lp = T(detrand.normal(22, "lp", (Tn * B,)) * 0.3).requires_grad_()
    olp = T(detrand.normal(22, "olp", (Tn * B,)) * 0.3)
    val = T(detrand.normal(22, "v", (Tn * B,))).requires_grad_()
    ret = R_ref.reshape(-1)
    ratio = torch.exp(lp - olp)
    adv = ret - val.detach()
    ref = (-torch.min(ratio * adv, ratio.clamp(0.8, 1.2) * adv) + 0.5 * torch.nn.functional.mse_loss(val, ret) - 0.01 * 3.0).mean()
    ref.backward()
    loss, dlp, dv = ops.ppo_loss(lp.detach().to(dev), olp.to(dev), val.detach().to(dev), ret.to(dev), 0.2, 3.0)
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
    np.testing.assert_allclose(dlp.cpu().numpy(), lp.grad.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(dv.cpu().numpy(), val.grad.numpy(), rtol=1e-4, atol=1e-8)


def test_modules_accept_non_contiguous_inputs():
    """inputs may arrive as strided views; the modules make them contiguous internally - same results."""
    from murcl_amd.utils.losses import NT_Xent
    dev = _dev()
    m = _abmil(985)
    x = T(P.bags(985, "nc.x", 2, 256, 512)).to(dev)
    xt = x.transpose(1, 2).contiguous().transpose(1, 2)              # same values, strides (N*d, 1, N)
    assert not xt.is_contiguous()
    a, b = m(x)[0], m(xt)[0]
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, atol=1e-6)   # (split-K float atomics: order varies)
    z = T(detrand.normal(7, "nc.z", (128, 16))).to(dev).t()          # [16,128] view of a [128,16] buffer
    zi, zj = z[:8], z[8:]
    assert not zi.is_contiguous()
    crit = NT_Xent(8, 0.5)
    l1 = crit(zi, zj)
    l2 = crit(zi.contiguous(), zj.contiguous())
    assert l1.item() == pytest.approx(l2.item(), rel=1e-6)


def test_forward_only_fast_paths_equal_the_autograd_paths():
    """Under torch.no_grad() (frozen-encoder stage 2, validation) Full_layer and NT_Xent call the kernels directly and
    NT_Xent skips the gradient half of its kernel: same numbers, same hidden-state bookkeeping."""
    from murcl_amd.models.rlmil import Full_layer
    from murcl_amd.utils.losses import NT_Xent
    dev = _dev()

    def run(no_grad):
        fc = Full_layer(512, 1024, True, 128)
        fc.load_state_dict(P.to_torch(P.full_layer(985)))
        fc = fc.to(dev)
        crit = NT_Xent(8, 0.5)
        outs = []
        with torch.set_grad_enabled(not no_grad):
            for t in range(3):
                xs = [T(detrand.normal(33, f"ff.x.{t}.{v}", (8, 512))).to(dev) for v in range(2)]
                z = fc.forward_views(xs, restart=(t == 0))
                loss = crit(z[0], z[1])
                outs += [z[0].detach().cpu(), z[1].detach().cpu(), loss.detach().cpu().reshape(1), crit.last_similarity.cpu(),
                         fc.hidden.detach().cpu().reshape(-1)]
                assert loss.requires_grad != no_grad
        return outs

    for a, b in zip(run(True), run(False)):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_clam_plain_attention_net_gate_false(golden, dtype):
    """CLAM_SB(gate=False): Attn_Net (clam.py:18-34,80-81) on the gate kernels with the sigmoid branch off - raw scores,
    soft-max, pooled vector, top-k ids, instance loss and gradients vs the reference golden G14 (f32), and the bf16 path
    against the f32 one."""
    from murcl_amd.models.clam import CLAM_SB
    g = golden("g14_clam_plain")
    dev = _dev()
    pk = P.clam_sb_plain(11)
    m = CLAM_SB(gate=False, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512)
    assert sorted(m.state_dict()) == sorted(pk)                       # the reference's key names
    m.load_state_dict(P.to_torch(pk))
    m = m.to(dev).eval()
    m.compute_dtype = dtype
    x = T(P.bags(11, "g4.x", 3, 300, 512)).to(dev)
    M, A, s, il, ids, _ = m._run(x, [1, 1, 1], True)
    if dtype == torch.float32:
        np.testing.assert_allclose(s.cpu().numpy(), g["raw"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(A.cpu().numpy(), g["A"], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(M.detach().cpu().numpy(), g["M_batch"], rtol=1e-4, atol=1e-5)
        assert np.array_equal(ids[:, :8].cpu().numpy(), g["top_p"]) and np.array_equal(ids[:, 8:].cpu().numpy(), g["top_n"])
        np.testing.assert_allclose(il.detach().cpu().numpy(), g["inst_loss"], rtol=2e-4)
        (M.sum() + il.sum()).backward()
        for k, v in m.named_parameters():
            key = "grad." + k
            if key in g.files and not k.endswith("module.3.bias"):
                got = _summ(v.grad)
                np.testing.assert_allclose(got, g[key], rtol=2e-3, atol=3e-4 * g[key][1], err_msg=k)
        out = m(x[0:1].squeeze(0).unsqueeze(0), attention_only=True)[0]
        assert out.shape == (1, 300)
    else:
        assert (s.cpu() - T(g["raw"])).abs().max().item() < 3e-2 * float(np.abs(g["raw"]).max()) + 3e-2
        assert (M.detach().cpu() - T(g["M_batch"])).abs().max().item() < 3e-2 * float(np.abs(g["M_batch"]).max())
        (M.sum() + il.sum()).backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


# ---------------------------------------------------------------- ABMIL outside the launch scripts' default shape (G16)
def _abmil_general(case, dtype=torch.float32):
    from murcl_amd.models.abmil import ABMIL
    from oracle.recipes import G16, g16_inputs
    k = G16["cases"][case]
    pd, x, masks = g16_inputs(case)
    m = ABMIL(k["d"], L=k["L"], D=k["D"], dim_out=2, dropout=0.25 if masks is not None else 0.0)
    m.load_state_dict(P.to_torch(pd))
    m.compute_dtype = dtype
    m = m.to(_dev()).train()
    if masks is not None:
        m.keep_masks = tuple(T(mk).reshape(-1, k["L"]).to(_dev()) for mk in masks)
    return m, T(x).to(_dev())


@pytest.mark.parametrize("case", ["dropout", "small", "small_dropout"])
def test_abmil_dropout_and_other_L_D_vs_reference_golden(golden, case):
    """VERDICT r2 missing 5: ``--dropout`` > 0 in training mode and ``--L`` / ``--D`` other than 512 / 128 (abmil.py:8-33,
    train_RLMIL.py:91-97) run on the kernels and match the reference module (G16: out, attention, every parameter gradient)."""
    g = golden("g16_abmil_general")
    m, x = _abmil_general(case)
    out, det = m(x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"{case}.out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.last_attention.cpu().numpy(), g[f"{case}.A"], rtol=1e-4, atol=1e-9)
    out.sum().backward()
    n = 0
    for k, v in m.named_parameters():
        key = f"{case}.grad.{k}"
        if key not in g.files:
            continue
        if k == "attention.2.bias":
            assert v.grad.abs().max().item() < 1e-4 * m.attention[2].weight.grad.norm().item()
            continue
        _check_summ(_summ(v.grad), g[key], 5e-4, key)
        n += 1
    assert n >= 11 and m.fc.weight.grad is None


def test_abmil_dropout_bf16_fast_path_close_to_fp32_and_seeded_masks_equal_materialised_ones():
    """The bf16 weight-stationary path with Dropout: (a) injected masks - close to the f32 kernels; (b) masks generated inside
    the passes from DropSeeds == the same masks materialised and injected (outputs and gradients bit for bit: same kernels,
    same keep pattern), so training-mode dropout back-propagates through exactly the mask the forward applied."""
    from murcl_amd import ops
    m32, x = _abmil_general("dropout")
    m16, _ = _abmil_general("dropout", torch.bfloat16)
    o32, _ = m32(x)
    o16, _ = m16(x)
    assert (o16 - o32).abs().max().item() <= 3e-2 * o32.abs().max().item()
    o32.sum().backward(), o16.sum().backward()
    for (k, a), (_, b) in zip(m32.named_parameters(), m16.named_parameters()):
        if a.grad is not None and k != "attention.2.bias":
            assert ((a.grad - b.grad).norm() / a.grad.norm()).item() < 6e-2, k
    # (b) seeds vs materialised masks, bf16 fast path and f32 general path
    for dtype in (torch.bfloat16, torch.float32):
        seeds = (ops.DropSeed(0.75, seed=1234567), ops.DropSeed(0.75, seed=7654321))
        ma, _ = _abmil_general("dropout", dtype)
        mb, _ = _abmil_general("dropout", dtype)
        ma.keep_masks = seeds
        mb.keep_masks = tuple(ops.dropout_mask((x.shape[0] * x.shape[1], 512), dtype, 0.75, x.device, seed=s.seed) for s in seeds)
        frac = (mb.keep_masks[0] != 0).float().mean().item()
        assert abs(frac - 0.75) < 5e-3
        oa, _ = ma(x)
        ob, _ = mb(x)
        # f32: the same multipliers -> the same numbers; bf16: the materialised mask holds 1/0.75 ROUNDED to bf16 (1.3359) while
        # the seeded pass multiplies by the f32 value and rounds once, so the two agree to bf16 precision only
        tol = 1e-6 if dtype == torch.float32 else 1e-2
        assert (oa - ob).abs().max().item() <= tol * ob.abs().max().item()
        oa.sum().backward(), ob.sum().backward()
        for (k, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()):
            if a.grad is not None and k != "attention.2.bias":
                assert ((a.grad - b.grad).norm() / b.grad.norm()).item() < (1e-5 if dtype == torch.float32 else 5e-2), k


def test_abmil_training_mode_dropout_draws_fresh_masks_and_eval_is_deterministic():
    m, x = _abmil_general("dropout")
    m.keep_masks = None                                   # the module's own draws
    torch.manual_seed(11)
    a, _ = m(x)
    b, _ = m(x)
    assert not torch.equal(a, b)                           # two calls, two masks
    m.eval()
    c, _ = m(x)
    d, _ = m(x)
    assert torch.equal(c, d)
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_dropout_scale_is_the_realised_keep_probability():
    """survivors are scaled by 1 / (the keep probability the kernels realise), e.g. 230/256 for keep 0.9."""
    from murcl_amd import ops
    k = ops.dropout_mask((1 << 16, 64), torch.float32, 0.9, _dev(), seed=99)
    kept = k[k != 0]
    assert torch.all(kept == kept[0]) and abs(kept[0].item() - 256.0 / 230.0) < 1e-6
    assert abs(k.mean().item() - 1.0) < 2e-3                # unbiased: E[keep multiplier] = 1
    k75 = ops.dropout_mask((1 << 12, 64), torch.float32, 0.75, _dev(), seed=5)
    assert abs(k75[k75 != 0][0].item() - 1.0 / 0.75) < 1e-6


# ------------------------------------------------------------------ round 5: the constructor branches VERDICT r4 listed
def _check_grad_entries(named, gold, prefix, rtol, zero_keys=()):
    """Gradients the golden stores in full: every entry within rtol of the LARGEST entry (north_star's 1e-4 at rtol = 1e-4);
    34-number fingerprints as _check_summ."""
    n = 0
    for k, v in named:
        key = prefix + k
        if key not in gold.files:
            continue
        want = gold[key]
        if k in zero_keys:                                    # exact zeros in exact arithmetic (soft-max shift invariance)
            n += 1
            continue
        if want.shape == tuple(v.grad.shape):
            np.testing.assert_allclose(v.grad.cpu().numpy(), want, rtol=rtol, atol=rtol * np.abs(want).max(), err_msg=key)
        else:
            _check_summ(_summ(v.grad), want, max(rtol, 3e-4), key)
        n += 1
    assert n > 0


def test_abmil_every_gradient_entry_vs_reference_golden(golden):
    """G20: the reference's gradients of ALL ABMIL parameters at the C1 shape, stored in full - every entry of every gradient of
    the f32 path within 1e-4 of the tensor's largest entry (north_star's tolerance; G1 pins fingerprints only)."""
    g = golden("g20_abmil_full_grads")
    m = _abmil(985)
    out, _ = m(T(P.bags(985, "g1.x", 4, 256, 512)).to(_dev()))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    out.sum().backward()
    _check_grad_entries(m.named_parameters(), g, "grad.", 1e-4, zero_keys=("attention.2.bias",))
    assert m.attention[2].bias.grad.abs().max().item() < 1e-4 * m.attention[2].weight.grad.abs().max().item()


def test_abmil_three_attention_heads_vs_reference_golden(golden):
    """G19: ABMIL(K=3) (abmil.py:8,23-27,38-44) - out [B*K, L] bag-major, attention [B, K, N], the x.shape[0] == 1 branch and
    the parameter gradients of a weighted sum of the outputs."""
    from murcl_amd.models.abmil import ABMIL
    g = golden("g19_abmil_heads")
    m = ABMIL(512, L=512, D=128, K=3, dim_out=2)
    m.load_state_dict(P.to_torch(P.abmil(19, K=3, dim_out=2)))
    m = m.to(_dev())
    x = T(P.bags(19, "g19.x", 3, 200, 512)).to(_dev())
    out, det = m(x)
    assert out.shape == (9, 512) and not det.requires_grad
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.last_attention.cpu().numpy(), g["A"], rtol=1e-4, atol=1e-9)
    (out * T(detrand.normal(19, "g19.w", (9, 512))).to(_dev())).sum().backward()
    _check_grad_entries(m.named_parameters(), g, "grad.", 2e-4, zero_keys=("attention.2.bias",))
    np.testing.assert_allclose(m(x[:1])[0].detach().cpu().numpy(), g["out_single"], rtol=1e-4, atol=1e-5)
    # bf16 storage path: K single-head passes through the one-pass pooling kernels
    m16 = ABMIL(512, L=512, D=128, K=3, dim_out=2)
    m16.load_state_dict(P.to_torch(P.abmil(19, K=3, dim_out=2)))
    m16.compute_dtype = torch.bfloat16
    o16, _ = m16.to(_dev())(x)
    assert (o16 - out).abs().max().item() <= 3e-2 * out.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_clam_size_big_vs_reference_golden(golden, dtype):
    """G18: CLAM_SB(size_arg="big") - a 384-wide attention net (clam.py:66-67; --size_arg big of both entry scripts).  f32 at the
    golden's 1e-4; bf16 (the fused gate kernels are built for D = 256: this width takes the generic chain) against the reference
    at the bf16 tolerances of the D = 256 tests."""
    from murcl_amd.models.clam import CLAM_SB
    g = golden("g18_clam_big")
    m = CLAM_SB(gate=True, size_arg="big", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512)
    m.load_state_dict(P.to_torch(P.clam_sb(18, size=(512, 384))))
    m.compute_dtype = dtype
    m = m.to(_dev()).eval()
    x = T(P.bags(18, "g18.x", 3, 300, 512)).to(_dev())
    f32 = dtype == torch.float32
    M, _ = m(x)
    sM = np.abs(g["M_batch"]).max()
    np.testing.assert_allclose(M.detach().cpu().numpy(), g["M_batch"], rtol=1e-4 if f32 else 3e-2, atol=(1e-5 if f32 else 3e-2 * sM))
    np.testing.assert_allclose(m.last_attention.cpu().numpy(), g["A"], rtol=2e-4 if f32 else 5e-2, atol=1e-9 if f32 else 1e-5)
    raw = torch.cat([m.bag_forward(x[b], attention_only=True) for b in range(3)])
    np.testing.assert_allclose(raw.cpu().numpy(), g["raw"], rtol=1e-4 if f32 else 0, atol=2e-5 if f32 else 3e-2)
    if not f32:
        return
    from murcl_amd import ops
    m(x)                                                       # (the single-bag calls above left the last bag's attention behind)
    ids = ops.topk_ids(m.last_attention, 8).cpu().numpy()
    np.testing.assert_array_equal(ids[:, :8], g["top_p"])
    np.testing.assert_array_equal(ids[:, 8:], g["top_n"])
    for label in (0, 1):
        _, _, res = m(x, label=[label] * 3, instance_eval=True)
        for b in range(3):
            np.testing.assert_allclose(float(res[b]["instance_loss"].detach()), g[f"l{label}.inst_loss"][b], rtol=2e-4)
            np.testing.assert_array_equal(res[b]["inst_preds"], g[f"l{label}.preds"][b])
            np.testing.assert_array_equal(res[b]["inst_labels"], g[f"l{label}.targets"][b])
    M2, _, res = m(x, label=[1, 1, 1], instance_eval=True)
    (M2.sum() + sum(r["instance_loss"] for r in res)).backward()
    _check_grad_entries(m.named_parameters(), g, "grad.", 3e-4, zero_keys=("attention_net.3.attention_c.bias",))


@pytest.mark.parametrize("mode", ["clam", "dsmil", "abmil"])
def test_step_loss_nodes_equal_the_tensor_op_composition(mode):
    """functional.StepLossFn / StepCEMeanFn (the supervised step's loss of T patch steps as one node: train_RLMIL.py:336 CLAM-SB,
    :527-529 DSMIL, :727 ABMIL) against the same formulas written with torch ops on the same logits: the step loss, the per-step
    losses, the confidences and the gradients that reach the logits and the second term."""
    import torch.nn.functional as F
    from murcl_amd import ops
    from murcl_amd.functional import StepCEMeanFn, StepLossFn
    dev = _dev()
    Tn, B, C = 6, 8, 2
    lg0 = T(detrand.normal(35, f"sl.{mode}.lg", (Tn * B, C))).to(dev)
    ex0 = T(detrand.normal(35, f"sl.{mode}.ex", (Tn * B, C) if mode == "dsmil" else (Tn * B,))).to(dev)
    labels = T(np.asarray(detrand.permutation(35, "sl.lab", B)) % C).to(dev).long()
    lab_all = labels.repeat(Tn)
    bw = 0.7

    def ref():
        lg, ex = lg0.clone().requires_grad_(), ex0.clone().requires_grad_()
        ce = torch.stack([F.cross_entropy(lg[t * B:(t + 1) * B], labels) for t in range(Tn)])
        if mode == "clam":
            lt = bw * ce + (1 - bw) * ex.view(Tn, B).mean(1)
        elif mode == "dsmil":
            lt = 0.5 * ce + 0.5 * torch.stack([F.cross_entropy(ex[t * B:(t + 1) * B], labels) for t in range(Tn)])
        else:
            lt = ce
        total = lt.sum() / Tn
        total.backward()
        conf = torch.softmax(lg.detach(), 1).gather(1, lab_all.view(-1, 1)).view(-1)
        return total.detach(), lt.detach(), conf, lg.grad, (ex.grad if mode != "abmil" else None)

    lg, ex = lg0.clone().requires_grad_(), ex0.clone().requires_grad_()
    if mode == "clam":
        total, lt, conf = StepLossFn.apply(lg, lab_all, B, bw, ex, 1 - bw, False)
    elif mode == "dsmil":
        total, lt, conf = StepLossFn.apply(lg, lab_all, B, 0.5, ex, 0.5, True)
    else:
        total, lt, conf = StepCEMeanFn.apply(lg, lab_all, B)
    assert total.shape == () and not lt.requires_grad and not conf.requires_grad
    total.backward(ops.unit_grad(total))
    rt, rl, rc, rg, rx = ref()
    np.testing.assert_allclose(total.item(), rt.item(), rtol=1e-5)
    np.testing.assert_allclose(lt.cpu().numpy(), rl.cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(conf.cpu().numpy(), rc.cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), rg.cpu().numpy(), rtol=1e-4, atol=1e-7)
    if mode != "abmil":
        np.testing.assert_allclose(ex.grad.cpu().numpy(), rx.cpu().numpy(), rtol=1e-4, atol=1e-8)
    # a non-unit upstream gradient takes the general path
    lg2, ex2 = lg0.clone().requires_grad_(), ex0.clone().requires_grad_()
    if mode == "abmil":
        (StepCEMeanFn.apply(lg2, lab_all, B)[0] * 3.0).backward()
    else:
        (StepLossFn.apply(lg2, lab_all, B, bw if mode == "clam" else 0.5, ex2, (1 - bw) if mode == "clam" else 0.5, mode == "dsmil")[0] * 3.0).backward()
        np.testing.assert_allclose(ex2.grad.cpu().numpy(), 3.0 * rx.cpu().numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(lg2.grad.cpu().numpy(), 3.0 * rg.cpu().numpy(), rtol=1e-4, atol=1e-7)
