"""BASELINE config 4's body on MI355X: the MuRCL batch step with the PPO sub-bag sampler in the loop
(train_MuRCL.py:233-304, stages 2 and 3) against
  * tests/golden/g12_rl_step.npz - ONE batch through the reference's own ``train()`` with every draw injected, and
  * oracle/step_oracle.pretrain_step_rl (pinned to the same golden on the CPU side) at another size.
Compared: the sampler's actions and log-probs per patch step, the patch ids they select (bit-exact), per-step losses,
rewards, and the parameters after the update (policy at stage 2; aggregator + head at stage 3)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, mil_oracle as O, params as P, step_oracle as SO  # noqa: E402
from oracle.recipes import G12, g12_inputs, window_margin  # noqa: E402

T = torch.from_numpy


def _summ(g):
    g = g.detach().double().flatten().cpu()
    return np.concatenate([[g.norm().item(), g.abs().max().item()], g[:32].numpy()])


def _close_summ(got, want, rtol, msg=""):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * want[1], err_msg=msg)


def _build(tmp_path, stage, Tn, seed, K, fs, B, ppo_lr, lr, K_epochs=3, dtype="f32"):
    """Stage-k objects through the product's own create_model: the previous stage's checkpoint is written from the
    deterministic parameter sets and picked up by the default ``../stage_{k-1}/model_best.pth.tar`` rule."""
    from murcl_amd.train_MuRCL import build_parser, create_model, get_optimizer
    prev = tmp_path / f"stage_{stage - 1}"
    prev.mkdir(parents=True, exist_ok=True)
    pol = P.to_torch(P.actor_critic(seed, 512, 512, K))
    torch.save({"epoch": 1, "model_state_dict": {"encoder." + k: v for k, v in P.to_torch(P.abmil(seed)).items()},
                "fc": P.to_torch(P.full_layer(seed)), "optimizer": None, "ppo_optimizer": None, "policy": pol},
               prev / "model_best.pth.tar")
    args = build_parser().parse_args(["--arch", "ABMIL", "--dtype", dtype, "--train_stage", str(stage), "--T", str(Tn),
                                      "--feat_size", str(fs), "--batch_size", str(B), "--num_clusters", str(K),
                                      "--ppo_lr", str(ppo_lr), "--backbone_lr", str(lr), "--fc_lr", str(lr),
                                      "--K_epochs", str(K_epochs), "--save_dir", str(tmp_path / f"stage_{stage}")])
    dev = torch.device("cuda:0")
    model, fc, ppo = create_model(args, 512, dev)
    if stage == 2:                       # the reference starts stage 2 from a FRESH sampler (:117-122): seat the test policy
        ppo.policy.load_state_dict(pol)
        ppo.policy_old.load_state_dict(pol)
    return args, model, fc, ppo, get_optimizer(args, model, fc), dev


def _run(args, model, fc, ppo, opt, dev, feats, cls, inj, B):
    from murcl_amd.models import rlmil
    from murcl_amd.train_MuRCL import pretrain_step
    from murcl_amd.utils.datasets import BagPack, select_indices
    from murcl_amd.utils.losses import NT_Xent
    pack = BagPack.from_lists([T(f).to(dev) for f in feats], cls)
    dinj = {"actions": [[T(np.asarray(a)).to(dev) for a in inj["actions"][0]]],
            "draws": [[(T(np.asarray(l)).to(dev), T(np.asarray(p)).to(dev)) for l, p in row] for row in inj["draws"]],
            "eps": [[T(np.asarray(e)).to(dev) for e in row] for row in inj["eps"]], "trace": []}
    mems = [rlmil.Memory(), rlmil.Memory()]
    loss, losses, rewards = pretrain_step(args, model, fc, ppo, NT_Xent(B, 1.0), opt, pack, mems, injected=dinj)
    assert all(len(getattr(m, f)) == 0 for m in mems for f in rlmil.Memory.FIELDS)                 # :300-302
    acts = [step for step in dinj["trace"] if isinstance(step, list)]
    logp = [step for step in dinj["trace"] if isinstance(step, dict)][0]["logprobs"]
    ids = [[select_indices(pack, a, args.feat_size)[0].cpu().numpy() for a in step] for step in acts]
    return loss, losses, rewards, acts, logp, ids


@pytest.mark.parametrize("Tn", [3, 6])
@pytest.mark.parametrize("stage", [2, 3])
def test_rl_in_the_loop_step_vs_reference_train_loop(golden, tmp_path, Tn, stage):
    g, c = golden("g12_rl_step"), G12
    tag = f"T{Tn}.s{stage}"
    seed, B, K, fs = c["seed"], c["B"], c["K"], c["fs"]
    Ns, feats, cls, inj = g12_inputs(Tn)
    args, model, fc, ppo, opt, dev = _build(tmp_path, stage, Tn, seed, K, fs, B, c["ppo_lr"], c["lr"], c["K_epochs"])
    pre_pol = {k: v.detach().clone() for k, v in ppo.policy.state_dict().items()}
    pre_m = {k: v.detach().clone() for k, v in model.encoder.state_dict().items()}
    pre_f = {k: v.detach().clone() for k, v in fc.state_dict().items()}
    loss, losses, rewards, acts, logp, ids = _run(args, model, fc, ppo, opt, dev, feats, cls, inj, B)
    np.testing.assert_allclose([l.item() for l in losses], g[f"{tag}.losses"], rtol=1e-4)
    np.testing.assert_allclose(torch.cat(rewards).cpu().numpy(), g[f"{tag}.rewards"], rtol=5e-3, atol=3e-6)
    for v in range(2):
        got = torch.stack([acts[t][v] for t in range(1, Tn)]).cpu().numpy()
        np.testing.assert_allclose(got, g[f"{tag}.actions.{v}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(logp[v].cpu().numpy(), g[f"{tag}.logp.{v}"], rtol=1e-4, atol=1e-4)
        for t in range(1, Tn):
            assert np.array_equal(ids[t][v], g[f"{tag}.ids.{t}.{v}"]), f"patch ids differ at step {t} view {v}"
    for k, v in ppo.policy.state_dict().items():
        want = g[f"{tag}.policy_delta.{k}"]
        if stage == 3:
            assert torch.equal(v, pre_pol[k]) and want[0] == 0.0                     # the joint stage only samples (:293-298)
        else:
            _close_summ(_summ(v - pre_pol[k]), want, 3e-2, k)
    if stage == 2:
        assert all(torch.equal(a, b) for a, b in zip(ppo.policy.parameters(), ppo.policy_old.parameters()))   # rlmil.py:183
        assert all(torch.equal(v, pre_m[k]) for k, v in model.encoder.state_dict().items())                     # encoder frozen
    else:
        for name, sd, pre in (("model", model.encoder.state_dict(), pre_m), ("fc", fc.state_dict(), pre_f)):
            for k, v in sd.items():
                want = g[f"{tag}.{name}_delta.{k}"]
                if want[0] == 0.0:      # ABMIL.fc is never applied (abmil.py:33): torch's Adam skips it - no decay, no step
                    assert torch.equal(v, pre[k]), k
                else:
                    _close_summ(_summ(v - pre[k]), want, 3e-2, f"{name}.{k}")


@pytest.mark.parametrize("stage", [2, 3])
def test_two_optimizer_steps_vs_reference_train_loop(golden, tmp_path, stage):
    """G17: TWO consecutive batches through the reference's own train(); the second batch's losses, rewards, actions and patch
    ids depend on the first update (policy at stage 2, aggregator + head at stage 3), and the fingerprints are taken after both."""
    g, c, Tn = golden("g17_rl_two_steps"), G12, 3
    tag = f"s{stage}"
    seed, B, K, fs = c["seed"], c["B"], c["K"], c["fs"]
    args, model, fc, ppo, opt, dev = _build(tmp_path, stage, Tn, seed, K, fs, B, c["ppo_lr"], c["lr"], c["K_epochs"])
    pre = {"policy": {k: v.detach().clone() for k, v in ppo.policy.state_dict().items()},
           "model": {k: v.detach().clone() for k, v in model.encoder.state_dict().items()},
           "fc": {k: v.detach().clone() for k, v in fc.state_dict().items()}}
    for b in range(2):
        Ns, feats, cls, inj = g12_inputs(Tn, b)
        loss, losses, rewards, acts, logp, ids = _run(args, model, fc, ppo, opt, dev, feats, cls, inj, B)
        np.testing.assert_allclose([l.item() for l in losses], g[f"{tag}.losses"][b], rtol=1e-4 if b == 0 else 3e-4, err_msg=f"batch {b}")
        np.testing.assert_allclose(torch.cat(rewards).cpu().numpy(), g[f"{tag}.rewards"][b], rtol=1e-2, atol=5e-6)
        for v in range(2):
            got = torch.stack([acts[t][v] for t in range(1, Tn)]).cpu().numpy()
            np.testing.assert_allclose(got, g[f"{tag}.actions.{b}.{v}"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(logp[v].cpu().numpy(), g[f"{tag}.logp.{b}.{v}"], rtol=1e-4, atol=1e-4)
            for t in range(1, Tn):
                assert np.array_equal(ids[t][v], g[f"{tag}.ids.{b}.{t}.{v}"]), f"patch ids differ: batch {b}, step {t}, view {v}"
    now = {"policy": ppo.policy.state_dict(), "model": model.encoder.state_dict(), "fc": fc.state_dict()}
    for name in ("policy", "model", "fc"):
        for k, v in now[name].items():
            key = f"{tag}.{name}_delta.{k}"
            if key not in g.files or g[key][0] == 0.0:
                assert torch.equal(v, pre[name][k]), key
                continue
            _close_summ(_summ(v - pre[name][k]), g[key], 3e-2, key)
            # absolute values: two sign-like Adam steps of 1e-3 each; an element whose gradient is ~0 may move differently by a
            # few per cent of a step (2e-5 seen), so the values are held to 1e-3 of the tensor's maximum
            _close_summ(_summ(v), g[f"{tag}.{name}.{k}"], 1e-3, key)


@pytest.mark.parametrize("stage", [2, 3])
def test_rl_in_the_loop_step_vs_oracle_other_shape(tmp_path, stage):
    """Same comparison against the oracle composition at another size (B = 6 ragged bags, K = 6 clusters, T = 4)."""
    seed, B, K, fs, Tn = 31, 6, 6, 96, 4
    Ns = [500 + 41 * b for b in range(B)]
    feats = [P.bags(seed, f"f{b}", 1, Ns[b], 512)[0] for b in range(B)]
    cls = [P.cluster_lists(seed, f"c{b}", Ns[b], K) for b in range(B)]
    inj = {"actions": [[detrand.uniform(seed, f"a{v}", (B, K)).astype(np.float32) for v in range(2)]],
           "draws": [[(detrand.uniform(seed, f"l{t}{v}", (B, 1), 0.9, 1.0).astype(np.float32), detrand.permutation(seed, f"p{t}{v}", B))
                      for v in range(2)] for t in range(Tn)],
           "eps": [[detrand.normal(seed, f"e{t}{v}", (B, K)).astype(np.float32) for v in range(2)] for t in range(Tn - 1)]}
    mp, fp = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.abmil(seed)).items()}, \
        {k: v.clone().requires_grad_() for k, v in P.to_torch(P.full_layer(seed)).items()}
    pp = P.to_torch(P.actor_critic(seed, 512, 512, K))
    r = SO.pretrain_step_rl(mp, fp, pp, feats, cls, inj, T=Tn, feat_size=fs, stage=stage, K_epochs=2, ppo_lr=1e-5)
    margin = min(window_margin(Ns[b], cls[b], r["actions"][t][v][b].numpy(), fs) for t in range(1, Tn) for v in range(2) for b in range(B))
    args, model, fc, ppo, opt, dev = _build(tmp_path, stage, Tn, seed, K, fs, B, 1e-5, 1e-3, K_epochs=2)
    pre_pol = {k: v.detach().clone() for k, v in ppo.policy.state_dict().items()}
    loss, losses, rewards, acts, logp, ids = _run(args, model, fc, ppo, opt, dev, feats, cls, inj, B)
    np.testing.assert_allclose([l.item() for l in losses], [l.item() for l in r["losses"]], rtol=1e-4)
    np.testing.assert_allclose(torch.cat(rewards).cpu().numpy(), torch.stack(r["rewards"]).numpy(), rtol=5e-3, atol=3e-6)
    for t in range(1, Tn):
        for v in range(2):
            np.testing.assert_allclose(acts[t][v].cpu().numpy(), r["actions"][t][v].numpy(), rtol=1e-4, atol=1e-5)
            if margin > 1e-3:                                              # (ids are only comparable away from floor() boundaries)
                want = np.array([i + [-1] * (fs - len(i)) for i in r["ids"][t][v]], dtype=np.int32)
                assert np.array_equal(ids[t][v], want)
    assert margin > 1e-3, f"pick another seed: window margin {margin}"
    if stage == 2:
        for k, v in ppo.policy.state_dict().items():
            want = _summ(r["policy"][k] - pp[k])
            _close_summ(_summ(v.cpu() - pre_pol[k].cpu()), want, 3e-2, k)


def test_stage3_one_deferred_aggregator_backward_equals_the_per_step_backwards(tmp_path, monkeypatch):
    """Stage 3 (bf16): the T sequential aggregator passes write into one EncoderSession and share ONE backward over all T * 2B
    bags; same forward (losses, rewards, actions bit for bit) and the same update as T separate backward passes."""
    from murcl_amd import functional, train_MuRCL as TM
    seed, B, K, fs, Tn = 41, 4, 10, 128, 4
    Ns = [700 + 53 * b for b in range(B)]
    feats = [P.bags(seed, f"f{b}", 1, Ns[b], 512)[0] for b in range(B)]
    cls = [P.cluster_lists(seed, f"c{b}", Ns[b], K) for b in range(B)]
    inj = {"actions": [[detrand.uniform(seed, f"a{v}", (B, K)).astype(np.float32) for v in range(2)]],
           "draws": [[(detrand.uniform(seed, f"l{t}{v}", (B, 1), 0.9, 1.0).astype(np.float32), detrand.permutation(seed, f"p{t}{v}", B))
                      for v in range(2)] for t in range(Tn)],
           "eps": [[detrand.normal(seed, f"e{t}{v}", (B, K)).astype(np.float32) for v in range(2)] for t in range(Tn - 1)]}
    made = []
    real_init = functional.EncoderSession.__init__
    monkeypatch.setattr(functional.EncoderSession, "__init__", lambda self, *a, **k: (made.append(a), real_init(self, *a, **k))[1])

    def run(deferred):
        monkeypatch.setattr(TM, "_DEFERRED_ENCODER", deferred)
        args, model, fc, ppo, opt, dev = _build(tmp_path / f"d{int(deferred)}", 3, Tn, seed, K, fs, B, 1e-5, 1e-3, dtype="bf16")
        p0 = [p.detach().clone() for p in list(model.parameters()) + list(fc.parameters())]
        loss, losses, rewards, acts, logp, ids = _run(args, model, fc, ppo, opt, dev, feats, cls, inj, B)
        assert model.encoder.session is None
        p1 = [p.detach().clone() for p in list(model.parameters()) + list(fc.parameters())]
        return [l.item() for l in losses], torch.cat(rewards).cpu().numpy(), [torch.stack(a).cpu().numpy() for a in acts], p0, p1

    n0 = len(made)
    a = run(True)
    assert len(made) == n0 + 1 and made[-1][0] == Tn and made[-1][1] == 2 * B          # one session of T steps x 2B bags
    b = run(False)
    assert len(made) == n0 + 1
    np.testing.assert_allclose(a[0], b[0], rtol=1e-6)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-7)
    for t, (x, y) in enumerate(zip(a[2], b[2])):
        np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-6, err_msg=f"actions of patch step {t}")
    moved = 0
    for p0, pa, pb in zip(a[3], a[4], b[4]):
        step = (pb - p0).norm().item()
        if step == 0.0:
            assert torch.equal(pa, p0)
            continue
        moved += 1
        assert ((pa - pb).norm().item() / step) < 3e-2
    assert moved >= 10


def test_flat_adam_skips_parameters_without_gradient_like_torch():
    """torch.optim.Adam skips ``grad is None`` parameters (no weight decay, no moments, no step count); a
    parameter first used at a later step starts its bias correction then."""
    from murcl_amd.functional import LinearFn
    from murcl_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mine = [torch.nn.Parameter(torch.randn(8, 16, device=dev)), torch.nn.Parameter(torch.randn(8, device=dev)),
            torch.nn.Parameter(torch.randn(4, 16, device=dev)), torch.nn.Parameter(torch.randn(4, device=dev))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt = FlatAdam([{"params": mine, "lr": 1e-2}], weight_decay=1e-2)
    topt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-2)
    x = torch.randn(5, 16, device=dev)
    for step in range(4):
        use_second = step >= 2                                     # the second Linear joins at step 2
        opt.zero_grad()
        y = LinearFn.apply(x, mine[0], mine[1], False).sum()
        if use_second:
            y = y + LinearFn.apply(x, mine[2], mine[3], False).pow(2).sum()
        y.backward()
        opt.step()
        topt.zero_grad(set_to_none=True)
        yr = torch.nn.functional.linear(x, ref[0], ref[1]).sum()
        if use_second:
            yr = yr + torch.nn.functional.linear(x, ref[2], ref[3]).pow(2).sum()
        yr.backward()
        topt.step()
        for a, b in zip(mine, ref):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
        if not use_second:
            assert mine[2].grad.abs().max().item() == 0.0


@pytest.mark.parametrize("momentum,nesterov", [(0.0, False), (0.9, False), (0.9, True)])
def test_flat_sgd_equals_torch_sgd(momentum, nesterov):
    """--optimizer SGD (train_MuRCL.py:158-163): momentum buffer seeded by the first gradient, optional Nesterov, L2 decay."""
    from murcl_amd.functional import LinearFn
    from murcl_amd.optim import FlatSGD
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    mine = [torch.nn.Parameter(torch.randn(8, 16, device=dev)), torch.nn.Parameter(torch.randn(8, device=dev))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt = FlatSGD([{"params": mine, "lr": 1e-2}], momentum=momentum, nesterov=nesterov, weight_decay=1e-3)
    topt = torch.optim.SGD(ref, lr=1e-2, momentum=momentum, nesterov=nesterov, weight_decay=1e-3)
    x = torch.randn(5, 16, device=dev)
    for _ in range(4):
        opt.zero_grad()
        LinearFn.apply(x, mine[0], mine[1], False).pow(2).sum().backward()
        opt.step()
        topt.zero_grad()
        torch.nn.functional.linear(x, ref[0], ref[1]).pow(2).sum().backward()
        topt.step()
        for a, b in zip(mine, ref):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    sd = opt.state_dict()
    opt2 = FlatSGD([{"params": [torch.nn.Parameter(p.detach().clone()) for p in mine], "lr": 1e-2}], momentum=momentum,
                   nesterov=nesterov, weight_decay=1e-3)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 4 and torch.equal(opt2.groups[0]["buf"], opt.groups[0]["buf"])


def test_ppo_returns_sharded_statistics_equal_the_single_kernel():
    """The data-parallel split of the return normalisation (raw + all-reduced sums + finish) == the one-kernel form."""
    from murcl_amd import ops
    dev = torch.device("cuda:0")
    Tn, B = 5, 64
    rw = T(detrand.normal(23, "r", (Tn, B)).astype(np.float32) * 0.01).to(dev)
    whole = ops.ppo_returns(rw, 0.1)
    parts, stats = [], torch.zeros(2, dtype=torch.float64, device=dev)
    for lo in (0, 16, 48):                                            # three unequal "ranks"
        hi = {0: 16, 16: 48, 48: 64}[lo]
        r, s = ops.ppo_returns_raw(rw[:, lo:hi].contiguous(), 0.1)
        parts.append(r)
        stats += s                                                    # the all-reduce
    got = torch.cat([ops.ppo_returns_finish(r, stats, Tn * B) for r in parts], 1)
    np.testing.assert_allclose(got.cpu().numpy(), whole.cpu().numpy(), rtol=2e-5, atol=2e-6)
    ref = O.ppo_returns([rw[t:t + 1].cpu() for t in range(Tn)], 0.1)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


def test_native_ppo_sequences_equal_the_per_launch_autograd_path():
    """csrc/ppo_seq.hip (one C call per sampling step / per K_epoch) against the same math driven launch by launch through
    autograd (ActorCritic.evaluate + PPOLossFn, the round-1 path, still used for shapes the sequences do not cover):
    actions, log-probs, hidden states, and every parameter gradient of one epoch."""
    from murcl_amd.models.rlmil import PPO, ActorCritic, Memory, _HipPolicyKernels
    dev = torch.device("cuda:0")
    seed, B, S_, H, K, Tm = 57, 12, 512, 512, 10, 4
    sd = P.to_torch(P.actor_critic(seed, S_, H, K))

    def rollout(native):
        ppo = PPO(512, S_, H, False, action_std=0.5, lr=1e-4, gamma=0.1, K_epochs=1, action_size=K)
        ppo.policy.load_state_dict(sd)
        ppo.policy_old.load_state_dict(sd)
        if not native:
            for m in (ppo.policy, ppo.policy_old):
                m._native_ok = lambda S: False
        mem = Memory()
        for t in range(Tm):
            st = T(detrand.normal(seed, f"s{t}", (B, S_))).to(dev)
            eps = T(detrand.normal(seed, f"e{t}", (B, K))).to(dev)
            ppo.select_action(st, mem, restart_batch=(t == 0), eps=eps)
            mem.rewards.append((T(detrand.normal(seed, f"r{t}", (1, B))) * 0.01).to(dev))
        from murcl_amd import ops
        rewards = torch.cat(mem.rewards, 0)
        returns = ops.ppo_returns(rewards, 0.1)
        _HipPolicyKernels.epoch_grads(ppo, torch.stack(mem.states, 0), torch.stack(mem.actions, 0), torch.stack(mem.logprobs, 0),
                                      returns, rewards.numel())
        grads = {k: p.grad.detach().clone() for k, p in ppo.policy.named_parameters()}
        return (torch.stack(mem.actions, 0), torch.stack(mem.logprobs, 0), torch.cat(mem.hidden[1:], 0)), grads

    (a_n, lp_n, h_n), g_n = rollout(True)
    (a_r, lp_r, h_r), g_r = rollout(False)
    np.testing.assert_allclose(a_n.cpu().numpy(), a_r.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp_n.cpu().numpy(), lp_r.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(h_n.cpu().numpy(), h_r.cpu().numpy(), rtol=1e-4, atol=1e-6)
    for k in g_r:
        ref = g_r[k]
        assert ref.abs().max().item() > 0, k
        np.testing.assert_allclose(g_n[k].cpu().numpy(), ref.cpu().numpy(), rtol=2e-3, atol=2e-4 * ref.abs().max().item(), err_msg=k)
