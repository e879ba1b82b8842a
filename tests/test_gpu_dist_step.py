"""The multi-GPU branch of the MuRCL batch step, executed:

* one rank on RCCL ("nccl" backend, group initialised before any other GPU call in a child process): the
  all-gathered NT-Xent + flat gradient all-reduce + PPO collectives path gives the single-process step's numbers;
* two ranks sharing this box's one GPU over gloo (RCCL refuses two ranks per device): B/2 bags per rank with the real
  kernels == one process with B bags - same global loss, and after the update the ranks hold ONE policy (stage 2) /
  ONE model (stages 1, 3), bit-identical across ranks.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = os.path.join(HERE, "_dist_child.py")


def _spawn(mode, rank, world, port, workdir):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.Popen([sys.executable, CHILD, mode, str(rank), str(world), str(port), str(workdir)], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _wait(procs, timeout=600):
    for p in procs:
        out, _ = p.communicate(timeout=timeout)
        assert p.returncode == 0, out[-4000:]


def _moved(a, b):
    return (a - b).abs().max().item()


def _same_update(a, w, pre, what):
    """``a`` and ``w`` are the same optimizer update of ``pre`` up to kernel-order rounding.  Adam's steps are
    lr * sign-like, so the rare element whose gradient sits at the rounding level may land a full step apart: allow a
    0.1 % fraction of such outliers (at least one element), everything else within 5 % of the largest move, and the
    mean deviation within 1 % of the mean move."""
    mv = _moved(w, pre)
    if mv == 0.0:
        assert torch.equal(a, pre), what                  # parameters no gradient reaches (ABMIL.fc)
        return
    d = (a - w).abs()
    outliers = int((d > 0.05 * mv).sum())
    assert outliers <= max(1, int(1e-3 * d.numel())), f"{what}: {outliers} of {d.numel()} elements differ"
    assert d.mean().item() <= 1e-2 * (w - pre).abs().mean().item() + 1e-12, what


def test_single_rank_rccl_path_equals_single_process_step(tmp_path):
    _wait([_spawn("nccl1", 0, 1, 29640 + os.getpid() % 300, tmp_path)])
    res = torch.load(tmp_path / "nccl1_0.pt")
    from oracle import params as P
    pre = {"model": {"encoder." + k: v for k, v in P.to_torch(P.abmil(41)).items()}, "fc": P.to_torch(P.full_layer(41)),
           "policy": P.to_torch(P.actor_critic(41, 512, 512, 6))}
    for stage in (1, 2, 3):
        a, b = res[f"s{stage}.plain"], res[f"s{stage}.dist"]
        np.testing.assert_allclose(b["losses"].numpy(), a["losses"].numpy(), rtol=2e-6)
        np.testing.assert_allclose(b["rewards"].numpy(), a["rewards"].numpy(), rtol=1e-3, atol=1e-6)
        for part in ("model", "fc") + (("policy",) if stage > 1 else ()):
            for k in a[part]:
                _same_update(b[part][k], a[part][k], pre[part][k], f"s{stage} {part}.{k}")


def test_two_ranks_equal_one_process_and_stay_identical(tmp_path):
    port = 29950 + os.getpid() % 300
    _wait([_spawn("gloo2", r, 2, port, tmp_path) for r in range(2)])
    _wait([_spawn("single", 0, 1, 0, tmp_path)])
    r0, r1, one = (torch.load(tmp_path / f) for f in ("gloo2_0.pt", "gloo2_1.pt", "single_0.pt"))
    from oracle import params as P
    pre = {"model": {"encoder." + k: v for k, v in P.to_torch(P.abmil(41)).items()}, "fc": P.to_torch(P.full_layer(41)),
           "policy": P.to_torch(P.actor_critic(41, 512, 512, 6))}
    for stage in (1, 2, 3):
        a, b, w = r0[f"s{stage}"], r1[f"s{stage}"], one[f"s{stage}"]
        # every rank evaluates the GLOBAL contrastive loss (all-gathered embeddings)
        np.testing.assert_allclose(a["losses"].numpy(), w["losses"].numpy(), rtol=1e-5)
        assert torch.equal(a["losses"], b["losses"])
        np.testing.assert_allclose(torch.cat([a["rewards"], b["rewards"]], 1).numpy(), w["rewards"].numpy(), rtol=2e-3, atol=2e-6)
        trained = ("policy",) if stage == 2 else ("model", "fc")
        for part in trained:
            for k in a[part]:
                assert torch.equal(a[part][k], b[part][k]), f"stage {stage}: ranks diverged on {part}.{k}"
                _same_update(a[part][k], w[part][k], pre[part][k], f"stage {stage} {part}.{k}")


def test_two_ranks_with_batch_global_mixup_equal_one_process(tmp_path):
    """--global_mixup (round 6; utils/datasets.py:263-271 permutes over the WHOLE batch, which the default rank-local form changes at
    W > 1): two ranks that each hold all B raw bags, train on their half (``pretrain_step(local=...)``) with mix-up partners drawn over
    the whole batch - at least one partner sits on the other rank - and all-gather the sampler's actions == ONE process with the same
    global draws: same global loss at every patch step, same rewards, the same update (stages 1-3), ranks bit-identical."""
    port = 30250 + os.getpid() % 300
    _wait([_spawn("gloo2g", r, 2, port, tmp_path) for r in range(2)])
    _wait([_spawn("singleg", 0, 1, 0, tmp_path)])
    r0, r1, one = (torch.load(tmp_path / f) for f in ("gloo2g_0.pt", "gloo2g_1.pt", "singleg_0.pt"))
    from oracle import params as P
    pre = {"model": {"encoder." + k: v for k, v in P.to_torch(P.abmil(41)).items()}, "fc": P.to_torch(P.full_layer(41)),
           "policy": P.to_torch(P.actor_critic(41, 512, 512, 6))}
    for stage in (1, 2, 3):
        a, b, w = r0[f"s{stage}"], r1[f"s{stage}"], one[f"s{stage}"]
        np.testing.assert_allclose(a["losses"].numpy(), w["losses"].numpy(), rtol=1e-5)
        assert torch.equal(a["losses"], b["losses"])
        np.testing.assert_allclose(torch.cat([a["rewards"], b["rewards"]], 1).numpy(), w["rewards"].numpy(), rtol=2e-3, atol=2e-6)
        trained = ("policy",) if stage == 2 else ("model", "fc")
        for part in trained:
            for k in a[part]:
                assert torch.equal(a[part][k], b[part][k]), f"stage {stage}: ranks diverged on {part}.{k}"
                _same_update(a[part][k], w[part][k], pre[part][k], f"stage {stage} {part}.{k}")


def test_entry_script_on_two_ranks_with_global_mixup_keeps_the_ranks_identical(tmp_path):
    """``train_MuRCL.main --global_mixup --dist_backend gloo`` on two ranks sharing this GPU, stages 1-3 with the script's OWN random
    draws: the shared stream (window positions, mix-up draws, epoch order) needs no collective, the sampler's noise is per rank and its
    actions are all-gathered - every stage trains (parameters move), stays finite, and the two ranks end each stage with bit-identical
    parameters; rank 0 reports the whole split resident."""
    port = 31250 + os.getpid() % 300
    procs = [_spawn("cli2g", r, 2, port, tmp_path) for r in range(2)]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out[-4000:]
        outs.append(out)
    assert "whole split on every rank, --global_mixup): 12 slides" in outs[0] and "whole split" not in outs[1]
    r0, r1 = (torch.load(tmp_path / f"cli2g_{r}.pt") for r in range(2))
    for stage, parts in ((1, ("model", "fc")), (2, ("policy",)), (3, ("model", "fc"))):
        a, b = r0[f"s{stage}"], r1[f"s{stage}"]
        for part in parts:
            for k in a[part]:
                assert torch.isfinite(a[part][k]).all(), (stage, part, k)
                assert torch.equal(a[part][k], b[part][k]), f"stage {stage}: ranks diverged on {part}.{k}"
    assert not torch.equal(r0["s1"]["model"]["encoder.encoder.0.weight"], r0["s3"]["model"]["encoder.encoder.0.weight"])
