"""Host-side mirror of the reference interface: constructors, state-dict keys/shapes, error behaviour."""
import pytest
import torch

from oracle import params as P


def test_abmil_state_dict_keys_match_reference_names():
    from murcl_amd.models.abmil import ABMIL
    m = ABMIL(512, L=512, D=128, dim_out=128)
    want = P.abmil(1)
    sd = m.state_dict()
    assert sorted(sd) == sorted(want)
    assert all(tuple(sd[k].shape) == want[k].shape for k in sd)
    assert sum(p.numel() for p in m.parameters()) == 1182081          # SURVEY.md section 2.3 [probe]
    assert (m.L, m.D, m.K) == (512, 128, 1)


def test_full_layer_state_dict_and_memory():
    from murcl_amd.models.rlmil import Full_layer, Memory
    f = Full_layer(512, 1024, True, 128)
    want = P.full_layer(1)
    assert sorted(f.state_dict()) == sorted(want)
    assert sum(p.numel() for p in f.parameters()) == 4855936
    mem = Memory()
    mem.actions.append(1), mem.hidden.append(2)
    mem.clear_memory()
    assert all(getattr(mem, k) == [] for k in Memory.FIELDS)
    g = Full_layer(512, 1024, False, 10)
    assert sorted(k.split(".")[0] for k in g.state_dict()) == sorted(["fc_2", "fc_3", "fc_4", "fc_5"] * 2)


def test_cl_wrapper_contract():
    from murcl_amd.models.abmil import ABMIL
    from murcl_amd.models.cl import CL
    c = CL(ABMIL(512), projection_dim=128, n_features=512)
    assert c.projection_dim == 128 and c.n_features == 512
    assert all(k.startswith("encoder.") for k in c.state_dict())
    with pytest.raises(AssertionError):
        c(torch.zeros(1, 2, 512))


def test_abmil_type_error_and_guards():
    from murcl_amd.models.abmil import ABMIL
    with pytest.raises(TypeError):
        ABMIL(512)("not a tensor")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ABMIL(512, K=2)._bags(torch.zeros(1, 4, 512))                 # K != 1 heads are built (round 5) - on the GPU only
    assert ABMIL(512, K=3).attention[2].weight.shape == (3, 128)      # abmil.py:23-27
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ABMIL(512, L=256)._bags(torch.zeros(1, 4, 512))               # other L / D are built (general path) - on the GPU only


def test_shard_range():
    from murcl_amd.dist import shard_range
    assert [shard_range(r, 4, 16) for r in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]


def test_get_metrics_and_score_match_the_reference(golden):
    """utils/general.get_metrics / get_score (8(f) rank 2) against values computed by the reference (g10_eval)."""
    import numpy as np
    import torch
    from oracle import detrand
    from murcl_amd.train_RLMIL import BestPick, get_metrics, get_score
    g = golden("g10_eval")
    for name, n, C in (("bin", 14, 2), ("tri", 18, 3)):
        out = torch.from_numpy(detrand.normal(72, f"g10.m.{name}", (n, C)).astype(np.float32))
        tgt = torch.from_numpy(np.arange(n) % C)
        m = get_metrics(out, tgt)
        np.testing.assert_allclose(m, g[f"metrics.{name}"], rtol=1e-6)
        assert get_score(*m) == __import__("pytest").approx(float(g[f"score.{name}"]), rel=1e-6)
    pick = BestPick("loss")
    assert pick.update(1, 0.9, 0.5, 0.5, 0.5, 0.5, 0.5) and not pick.update(2, 1.1, 0.9, 0.9, 0.9, 0.9, 0.9) and pick.update(3, 0.7, 0, 0, 0, 0, 0)
    assert pick.epoch == 3
    pick = BestPick("score")
    assert pick.update(1, 0.9, 0.5, 0.5, 0.5, 0.5, 0.5) and pick.update(2, 1.1, 0.9, 0.9, 0.9, 0.9, 0.9) and pick.epoch == 2
    try:
        BestPick("f2")
        assert False
    except ValueError:
        pass


def test_draw_mixups_are_valid_draws():
    """datasets.draw_mixups: every view gets lambda in [alpha, 1] per bag and a permutation of the bags (the step-wide form
    of the reference's per-view torch.rand + torch.randperm, datasets.py:265-267); views differ from one another."""
    from murcl_amd.utils.datasets import draw_mixups
    torch.manual_seed(3)
    B, alpha, n = 64, 0.9, 12
    draws = draw_mixups(n, B, alpha, torch.device("cpu"))
    assert len(draws) == n
    for lam, perm in draws:
        assert lam.shape == (B, 1) and lam.dtype == torch.float32 and lam.is_contiguous()
        assert float(lam.min()) >= alpha and float(lam.max()) <= 1.0
        assert perm.shape == (B,) and perm.dtype == torch.int32 and perm.is_contiguous()
        assert sorted(perm.tolist()) == list(range(B))
    assert len({tuple(p.tolist()) for _, p in draws}) == n
    # mean of lambda ~ alpha + (1 - alpha) / 2 over n * B draws
    m = torch.cat([l for l, _ in draws]).mean().item()
    assert abs(m - (alpha + (1 - alpha) / 2)) < 0.01
    lam1, perm1 = draw_mixups(1, 2, 1.0, torch.device("cpu"))[0]
    assert torch.all(lam1 == 1.0) and sorted(perm1.tolist()) == [0, 1]


def test_stack_rows_is_a_view_when_the_blocks_are_consecutive():
    """ops.stack_rows (the two views' states / noise / hidden rows as one policy step): consecutive row blocks of one buffer come back as a
    view of it, anything else as a copy with the same contents."""
    from murcl_amd import ops
    x = torch.arange(48.).view(12, 4)
    a, b = x[:6], x[6:]
    y = ops.stack_rows([a, b])
    assert y.data_ptr() == x.data_ptr() and torch.equal(y, x)
    halves = x.split(6, 0)                                     # CL.forward hands the views out like this
    assert ops.stack_rows([h.detach() for h in halves]).data_ptr() == x.data_ptr()
    z = ops.stack_rows([b, a])                                 # wrong order: a copy
    assert z.data_ptr() != x.data_ptr() and torch.equal(z, torch.cat([b, a]))
    n = torch.arange(120.).view(5, 2, 3, 4)                    # one noise draw [T-1, views, B, K]
    assert ops.stack_rows([n[2, 0], n[2, 1]]).data_ptr() == n[2].data_ptr()
    w = ops.stack_rows([x[:6], torch.zeros(6, 4)])             # different buffers: a copy
    assert w.shape == (12, 4) and torch.equal(w[:6], x[:6])
    s = ops.stack_rows([x[:6, :2], x[6:, :2]])                 # strided blocks: a copy
    assert torch.equal(s, x[:, :2])


def test_clam_sb_takes_any_callable_instance_loss():
    """clam.py:64-65,118,131 call the loss they were given: None means the default nn.CrossEntropyLoss() (which runs inside the fused
    instance branch); any other callable is kept and handed the logits / targets of every evaluated class (round 6: G22 pins the
    numbers on the GPU); a non-callable raises."""
    import pytest
    from torch import nn
    from murcl_amd.models.clam import CLAM_SB, _is_default_ce
    assert isinstance(CLAM_SB().instance_loss_fn, nn.CrossEntropyLoss) and _is_default_ce(CLAM_SB().instance_loss_fn)
    assert _is_default_ce(CLAM_SB(instance_loss_fn=nn.CrossEntropyLoss()).instance_loss_fn)
    for other in (nn.CrossEntropyLoss(reduction="sum"), nn.CrossEntropyLoss(label_smoothing=0.1), nn.MultiMarginLoss(),
                  nn.CrossEntropyLoss(weight=torch.tensor([1.0, 2.0])), lambda a, b: (a.sum() + b.sum())):
        m = CLAM_SB(instance_loss_fn=other)
        assert m.instance_loss_fn is other and not _is_default_ce(other)
    with pytest.raises(TypeError):
        CLAM_SB(instance_loss_fn=3)
    m = CLAM_SB(size_arg="big", dropout=True)                     # clam.py:66-67: a 384-wide attention net (index 3 with the Dropout)
    assert m.attention_net[3].attention_a[0].weight.shape == (384, 512) and m.attention_net[3].attention_c.weight.shape == (1, 384)


def test_cu_reserve_follows_the_rccl_channel_cap():
    """dist.budget_for_channels (round 6): the CU budget of the launches a collective overlaps leaves RCCL one CU per channel it MAY
    run - a user-set NCCL_MAX_NCHANNELS above the reserve, or no cap at all, lowers the budget (with a note that is printed) instead
    of silently re-creating the second-round cliff; a fitting cap or a switched-off reserve (256) changes nothing."""
    from murcl_amd.dist import budget_for_channels as f
    assert f("248", 8) == (248, None)
    assert f(248, 4) == (248, None)
    assert f(256, 64) == (256, None)                    # reserve switched off on purpose
    b, note = f("248", 16)
    assert b == 240 and "NCCL_MAX_NCHANNELS=16" in note and "240" in note
    b, note = f("248", None)
    assert b == 224 and "not set" in note
    assert f(240, 12) == (240, None) and f(200, 56)[0] == 200 and f(248, 1000)[0] == 64
    assert f("251", 8) == (248, None)                   # multiples of 8: one step per XCD


def test_shared_seed_stream_is_the_same_on_every_rank_and_independent_of_the_global_generator():
    """train_MuRCL._shared_seed (--global_mixup): a function of --seed and the number of steps taken only - two 'ranks' with different
    torch global seeds (main() offsets them per rank) draw the same sequence; another --seed gives another."""
    import types
    from murcl_amd.train_MuRCL import _shared_seed
    r0, r1, other = (types.SimpleNamespace(seed=s) for s in (985, 985, 986))
    torch.manual_seed(1)
    a = [_shared_seed(r0) for _ in range(5)]
    torch.manual_seed(2)
    b = [_shared_seed(r1) for _ in range(5)]
    assert a == b and len(set(a)) == 5 and all(0 <= v < 2 ** 63 for v in a)
    assert [_shared_seed(other) for _ in range(5)] != a
