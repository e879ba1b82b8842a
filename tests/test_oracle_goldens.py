"""Pin the CPU oracle (oracle/) against outputs of the reference itself.

The goldens in tests/golden/ were produced by oracle/gen_goldens.py, which imports and
runs the reference checkout in the build container.  Nothing here reads the reference.
"""
import numpy as np
import pytest
import torch

from oracle import detrand, mil_oracle as O, params as P, select_oracle as S

T = torch.from_numpy
TOL = dict(rtol=2e-5, atol=2e-6)


def _summ(g):
    """Fingerprint of a tensor: L2 norm, max |.|, first 32 values."""
    g = g.detach().double().flatten()
    return np.concatenate([[g.norm().item(), g.abs().max().item()], g[:32].numpy()])


def _leaf(d):
    return {k: v.clone().requires_grad_() for k, v in P.to_torch(d).items()}


def _check_grads(gold, prefix, named, rtol=2e-4):
    n = 0
    for k, v in named.items():
        key = prefix + k
        if key in gold.files:
            assert v.grad is not None, key
            got, want = _summ(v.grad), gold[key]
            if k.endswith("attention.2.bias") or k.endswith("attention_c.bias") or k.endswith("module.3.bias"):
                # d softmax / d(uniform shift) == 0 exactly: both sides hold only rounding noise
                ref_scale = _summ(named[k[:-4] + "weight"].grad)[0]
                assert got[0] < 1e-4 * ref_scale and want[0] < 1e-4 * ref_scale, key
                n += 1
                continue
            np.testing.assert_allclose(got[:2], want[:2], rtol=rtol, err_msg=key)
            np.testing.assert_allclose(got[2:], want[2:], rtol=rtol, atol=rtol * want[1], err_msg=key)
            n += 1
    assert n > 0


def test_g1_abmil(golden):
    g = golden("g1_abmil")
    p = _leaf(P.abmil(985))
    x = T(P.bags(985, "g1.x", 4, 256, 512))
    out, A, s, M = O.abmil_forward(p, x)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    np.testing.assert_allclose(A.detach().numpy(), g["A"], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(O.abmil_forward(p, x[:1])[0].detach().numpy(), g["out_single"], **TOL)
    out.sum().backward()
    _check_grads(g, "grad.", p)
    assert p["fc.weight"].grad is None          # abmil.py:33 is never applied


@pytest.mark.parametrize("B", [2, 4, 64])
@pytest.mark.parametrize("tau", [1.0, 0.5])
def test_g2_ntxent(golden, B, tau):
    g = golden("g2_ntxent")
    zi = T(detrand.normal(7, f"g2.zi.{B}", (B, 128))).requires_grad_()
    zj = T(detrand.normal(7, f"g2.zj.{B}", (B, 128))).requires_grad_()
    loss = O.nt_xent(zi, zj, tau)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"loss.{B}.{tau}"], rtol=1e-5)
    np.testing.assert_allclose(zi.grad.numpy(), g[f"dzi.{B}.{tau}"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(zj.grad.numpy(), g[f"dzj.{B}.{tau}"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("Tn", [1, 3])
def test_g3_pretrain_step(golden, Tn):
    g = golden("g3_pretrain")
    mp, fp = _leaf(P.abmil(985)), _leaf(P.full_layer(985))
    views = [[T(P.bags(985, f"g3.x.{t}.{v}", 4, 256, 512)) for v in range(2)] for t in range(Tn)]
    loss, losses, rewards, _ = O.pretrain_step(mp, fp, views, 1.0)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"T{Tn}.loss"], rtol=1e-5)
    np.testing.assert_allclose([l.item() for l in losses], g[f"T{Tn}.losses"], rtol=1e-5)
    if Tn > 1:
        np.testing.assert_allclose(torch.stack(rewards).numpy(), g[f"T{Tn}.rewards"], rtol=1e-3, atol=2e-7)
    _check_grads(g, f"T{Tn}.grad.encoder.", mp)
    _check_grads(g, f"T{Tn}.grad.fc::", fp)


@pytest.mark.parametrize("subtyping", [False, True])
def test_g4_clam(golden, subtyping):
    g = golden("g4_clam")
    tag = f"sub{int(subtyping)}"
    p = P.to_torch(P.clam_sb(11))
    x = T(P.bags(11, "g4.x", 3, 300, 512))
    M, A, s, h = O.clam_sb_forward(p, x)
    np.testing.assert_allclose(s.numpy(), g[f"{tag}.raw"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(A.numpy(), g[f"{tag}.A"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(M.numpy(), g[f"{tag}.M_batch"], **TOL)
    np.testing.assert_array_equal(torch.topk(A, 8)[1].numpy(), g[f"{tag}.top_p"])
    np.testing.assert_array_equal(torch.topk(-A, 8)[1].numpy(), g[f"{tag}.top_n"])
    for label in (0, 1):
        np.testing.assert_allclose(M.numpy(), g[f"{tag}.l{label}.M"], **TOL)
        for b in range(3):
            loss, preds, tg, ids = O.clam_instance_eval(p, A[b], h[b], label, 2, 8, subtyping)
            np.testing.assert_allclose(loss.item(), g[f"{tag}.l{label}.inst_loss"][b], rtol=1e-5)
            np.testing.assert_array_equal(preds.numpy(), g[f"{tag}.l{label}.preds"][b])
            np.testing.assert_array_equal(tg.numpy(), g[f"{tag}.l{label}.targets"][b])


def test_g4_clam_grads(golden):
    g = golden("g4_clam")
    p = _leaf(P.clam_sb(11))
    x = T(P.bags(11, "g4.x", 3, 300, 512))
    M, A, s, h = O.clam_sb_forward(p, x)
    tot = M.sum()
    for b in range(3):
        tot = tot + O.clam_instance_eval(p, A[b], h[b], 1, 2, 8, True)[0]
    tot.backward()
    _check_grads(g, "grad.", p)


def test_g5_dsmil(golden):
    g = golden("g5_dsmil")
    p = _leaf(P.dsmil(5))
    x = T(P.bags(5, "g5.x", 3, 200, 512))
    c, bag, A, m = O.dsmil_forward(p, x)
    np.testing.assert_allclose(c.detach().numpy(), g["classes"], **TOL)
    np.testing.assert_array_equal(m.numpy(), g["m_ids"])
    np.testing.assert_allclose(bag.detach().numpy(), g["bag"], **TOL)
    c1, b1, _, _ = O.dsmil_forward(p, x[:1])
    np.testing.assert_allclose(c1[0].detach().numpy(), g["classes_single"], **TOL)
    np.testing.assert_allclose(b1.detach().numpy(), g["bag_single"], **TOL)
    (bag.sum() + c.max(1)[0].sum()).backward()
    _check_grads(g, "grad.", p)
    assert p["b_classifier.fcc.weight"].grad is None      # dsmil.py:62,80 unused


def _g6_clusters(g, name):
    flat, sizes = g[f"{name}.cluster_flat"], g[f"{name}.cluster_sizes"]
    return flat, sizes


@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e"])
def test_g6_get_feats_indices(golden, name):
    g = golden("g6_get_feats")
    N, fs, act = int(g[f"{name}.N"]), int(g[f"{name}.fs"]), g[f"{name}.act"]
    want = g[f"{name}.ids_plus1"]
    cls = _g6_rebuild(name)
    assert len(cls) == want.shape[0]
    feats = []
    for _ in cls:
        f = np.zeros((N, 4), np.float32)
        f[:, 0] = np.arange(N) + 1
        f[:, 1] = 7.0
        feats.append(f)
    out, kept = S.get_feats(feats, cls, act, fs)
    np.testing.assert_array_equal(out[:, :, 0].astype(np.int64), want)
    for b, ids in enumerate(kept):                       # padded tail is exactly zero
        assert (out[b, len(ids):] == 0).all()
        assert (out[b, :len(ids), 1] == 7.0).all()


def _g6_rebuild(name):
    """Same cluster constructions as oracle/gen_goldens.py:g6_get_feats."""
    if name == "a":
        return [P.cluster_lists(3, "g6.a", 4000, 10), P.cluster_lists(3, "g6.a2", 4000, 10)]
    if name == "b":
        ids = list(range(30))
        return [[ids[:10], ids[10:25], ids[25:]]]
    if name == "c":
        c = P.cluster_lists(4, "g6.c", 1000, 5) + [[]]
        return [c, c, c]
    sizes = (2, 6, 10, 14, 18, 14) if name == "d" else (6,) * 10
    cl, start = [], 0
    for n in sizes:
        cl.append(list(range(start, start + n)))
        start += n
    return [cl, cl] if name == "d" else [cl]


def test_g6_negative_slice_quirk():
    """n=(10,15,5), feat_size 40, a=0.5 -> ids [8,9,22,23,24,29]."""
    ids = list(range(30))
    got = S.select_indices(30, [ids[:10], ids[10:25], ids[25:]], [0.5, 0.5, 0.5], 40)
    assert got == [8, 9, 22, 23, 24, 29]


def test_g7_mixup(golden):
    g = golden("g7_mixup")
    x = detrand.normal(9, "g7.x", (5, 16, 8))
    lam = (np.float32(0.9) + g["u"] * np.float32(1 - 0.9)).astype(np.float32)
    np.testing.assert_array_equal(lam, g["lam"])
    np.testing.assert_array_equal(g["perm"], detrand.permutation(9, "g7.perm", 5))
    np.testing.assert_allclose(S.mixup(x, lam, g["perm"]), g["out"], rtol=1e-6, atol=1e-7)


def test_g8_ppo(golden):
    g = golden("g8_ppo")
    seed, B, S_, H, K, Tm, std = 21, 6, 512, 512, 10, 3, 0.5
    p = _leaf(P.actor_critic(seed, S_, H, K))
    hidden = torch.zeros(B, H)
    states, actions, logps, rewards = [], [], [], []
    with torch.no_grad():
        for t in range(Tm):
            st = T(detrand.normal(seed, f"g8.s{t}", (B, S_)))
            eps = T(detrand.normal(seed, f"g8.e{t}", (B, K)))
            a, lp, hidden = O.ppo_act(p, st, hidden, eps, std)
            np.testing.assert_allclose(a.numpy(), g[f"act.{t}"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(lp.numpy(), g[f"logp.{t}"], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(hidden.numpy(), g[f"hidden.{t}"], rtol=1e-4, atol=1e-6)
            states.append(st), actions.append(a), logps.append(lp)
            rewards.append(T(detrand.normal(seed, f"g8.r{t}", (1, B)) * 0.1))
        lp, v, ent = O.ppo_evaluate(p, torch.stack(states), torch.stack(actions), std)
    np.testing.assert_allclose(lp.numpy(), g["eval.logp"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(v.numpy(), g["eval.value"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ent.numpy(), g["eval.entropy"], rtol=1e-6)
    # one PPO.update with K_epochs=1, Adam lr 1e-3 (rlmil.py:152-184)
    R = O.ppo_returns(rewards, 0.1)
    loss = O.ppo_loss(p, torch.stack(states), torch.stack(actions), torch.stack(logps), R, std)
    loss.backward()
    newp = O.adam_step({k: v.detach() for k, v in p.items()}, {k: v.grad for k, v in p.items()}, {}, 1e-3)
    for k, v in newp.items():
        np.testing.assert_allclose(_summ(v), g["post." + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_g9_full_layer_interleaved_hidden(golden):
    g = golden("g9_full_layer")
    p = P.to_torch(P.full_layer(13))
    hidden = None
    for t in range(3):
        for v in range(2):
            x = T(detrand.normal(13, f"g9.x.{t}.{v}", (4, 512)))
            z, hidden = O.full_layer_step(p, x, None if t == 0 else hidden)
            np.testing.assert_allclose(z.numpy(), g[f"z.{t}.{v}"], rtol=1e-4, atol=2e-6)
            np.testing.assert_allclose(hidden.numpy(), g[f"h.{t}.{v}"], rtol=1e-4, atol=2e-6)


def test_g21_full_layer_cascade(golden):
    """G21: Full_layer(fc_rnn=False) - the cascaded classifiers over the shared, growing concatenation (rlmil.py:201-206,222-239):
    None for the two restarts, then fc_2 .. fc_5; logits, classifier gradients (in full) and input gradients of the reference."""
    g = golden("g21_full_layer_cascade")
    p = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.full_layer_cascade(21, 512, 16)).items()}
    hidden, xs, loss = None, {}, 0.0
    for t in range(3):
        for v in range(2):
            x = T(detrand.normal(21, f"g21.x.{t}.{v}", (4, 512))).requires_grad_()
            xs[(t, v)] = x
            z, hidden = O.full_layer_cascade_step(p, x, None if t == 0 else hidden, 512)
            assert (z is None) == bool(g[f"none.{t}.{v}"]) and hidden.shape[1] == int(g[f"width.{t}.{v}"])
            if z is not None:
                np.testing.assert_allclose(z.detach().numpy(), g[f"z.{t}.{v}"], rtol=1e-4, atol=2e-6)
                loss = loss + (z * T(detrand.normal(21, f"g21.w.{t}.{v}", (4, 16)))).sum()
    loss.backward()
    for k, v in p.items():
        np.testing.assert_allclose(v.grad.numpy(), g["grad." + k], rtol=1e-4, atol=1e-5 * np.abs(g["grad." + k]).max(), err_msg=k)
    for (t, v), x in xs.items():
        want = g[f"dx.{t}.{v}"]
        got = x.grad.numpy() if x.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * max(1e-30, np.abs(want).max()), err_msg=f"dx {t} {v}")
    with pytest.raises(RuntimeError):
        O.full_layer_cascade_step(p, T(detrand.normal(21, "g21.x.3.0", (4, 512))), hidden, 512)      # a sixth block: the reference exits


# ------------------------------------------------------------------ G12: BASELINE config 4's body (PPO sampler in the loop)
def _fp(x):
    return _summ(x if torch.is_tensor(x) else torch.as_tensor(x))


def _close_summ(got, want, rtol, scale=None, msg=""):
    """Fingerprints (norm, max, first 32 values): norm/max relative, values absolute to the tensor's max."""
    scale = want[1] if scale is None else scale
    np.testing.assert_allclose(got[:2], want[:2], rtol=rtol, atol=rtol * scale, err_msg=msg)
    np.testing.assert_allclose(got[2:], want[2:], rtol=rtol, atol=rtol * scale, err_msg=msg)


@pytest.mark.parametrize("Tn", [3, 6])
@pytest.mark.parametrize("stage", [2, 3])
def test_g12_rl_step_oracle_vs_reference_train_loop(golden, Tn, stage):
    """oracle/step_oracle.pretrain_step_rl == one batch of the reference's own train() at stage 2 / 3 (same injected draws):
    sampler actions + log-probs, selected patch ids (bit-exact), per-step losses, rewards, updated policy / model."""
    from oracle import step_oracle as SO
    from oracle.recipes import G12, g12_inputs
    g, c = golden("g12_rl_step"), G12
    tag = f"T{Tn}.s{stage}"
    Ns, feats, cls, inj = g12_inputs(Tn)
    mp, fp, pp = _leaf(P.abmil(c["seed"])), _leaf(P.full_layer(c["seed"])), P.to_torch(P.actor_critic(c["seed"], 512, 512, c["K"]))
    r = SO.pretrain_step_rl(mp, fp, pp, feats, cls, inj, T=Tn, feat_size=c["fs"], stage=stage, action_std=c["std"],
                            gamma=c["gamma"], K_epochs=c["K_epochs"], ppo_lr=c["ppo_lr"])
    np.testing.assert_allclose([l.item() for l in r["losses"]], g[f"{tag}.losses"], rtol=2e-6)
    np.testing.assert_allclose(torch.stack(r["rewards"]).numpy(), g[f"{tag}.rewards"], rtol=2e-3, atol=2e-7)
    for v in range(2):
        acts = torch.stack([r["actions"][t][v] for t in range(1, Tn)]).numpy()
        np.testing.assert_allclose(acts, g[f"{tag}.actions.{v}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(torch.stack(r["memories"][v]["logprobs"]).numpy(), g[f"{tag}.logp.{v}"], rtol=1e-5, atol=1e-5)
        for t in range(1, Tn):
            ids = np.array([i + [-1] * (c["fs"] - len(i)) for i in r["ids"][t][v]], dtype=np.int32)
            assert np.array_equal(ids, g[f"{tag}.ids.{t}.{v}"]), (t, v)
    pre = P.to_torch(P.actor_critic(c["seed"], 512, 512, c["K"]))
    for k, v in r["policy"].items():
        want = g[f"{tag}.policy_delta.{k}"]
        if stage == 3:
            assert want[0] == 0.0 and torch.equal(v, pre[k])            # the joint stage never updates the sampler (:293-298)
        else:
            _close_summ(_fp(v.detach() - pre[k]), want, 2e-2, msg=k)
            _close_summ(_fp(v.detach()), g[f"{tag}.policy.{k}"], 1e-4, msg=k)
    if stage == 3:
        r["loss"].backward()
        for name, params, base, lr in (("model", mp, P.to_torch(P.abmil(c["seed"])), c["lr"]), ("fc", fp, P.to_torch(P.full_layer(c["seed"])), c["lr"])):
            used = {k: v for k, v in params.items() if v.grad is not None}
            new = O.adam_step({k: v.detach() for k, v in used.items()}, {k: v.grad for k, v in used.items()}, {}, lr, weight_decay=c["wd"])
            for k in params:
                want = g[f"{tag}.{name}_delta.{k}"]
                if k not in used:                                       # e.g. ABMIL.fc: never applied (abmil.py:33) -> Adam skips it
                    assert want[0] == 0.0, k
                    continue
                _close_summ(_fp(new[k] - base[k]), want, 3e-2, msg=k)


@pytest.mark.parametrize("stage", [2, 3])
def test_g17_two_optimizer_steps_oracle_vs_reference_train_loop(golden, stage):
    """Two consecutive batches through the reference's own train() (G17): the second batch runs on the parameters the first
    update left behind, so a wrong update direction or Adam state shows in its losses and actions, not only in fingerprints."""
    from oracle import step_oracle as SO
    from oracle.recipes import G12, g12_inputs
    g, c, Tn = golden("g17_rl_two_steps"), G12, 3
    tag = f"s{stage}"
    mp, fp = P.to_torch(P.abmil(c["seed"])), P.to_torch(P.full_layer(c["seed"]))
    pp = P.to_torch(P.actor_critic(c["seed"], 512, 512, c["K"]))
    pp0, mp0, fp0 = dict(pp), dict(mp), dict(fp)
    ppo_state, adam = {}, {"model": {}, "fc": {}}
    for b in range(2):
        Ns, feats, cls, inj = g12_inputs(Tn, b)
        mpl, fpl = ({k: v.clone().requires_grad_() for k, v in d.items()} for d in (mp, fp))
        r = SO.pretrain_step_rl(mpl, fpl, pp, feats, cls, inj, T=Tn, feat_size=c["fs"], stage=stage, action_std=c["std"],
                                gamma=c["gamma"], K_epochs=c["K_epochs"], ppo_lr=c["ppo_lr"], ppo_state=ppo_state)
        np.testing.assert_allclose([l.item() for l in r["losses"]], g[f"{tag}.losses"][b], rtol=2e-5 if b else 2e-6, err_msg=f"batch {b}")
        np.testing.assert_allclose(torch.stack(r["rewards"]).numpy(), g[f"{tag}.rewards"][b], rtol=2e-3, atol=2e-7)
        for v in range(2):
            acts = torch.stack([r["actions"][t][v] for t in range(1, Tn)]).numpy()
            # (batch 1 runs on parameters that Adam's sign-like first steps produced: last-bit differences in those gradients
            #  show up at the 1e-5 level)
            np.testing.assert_allclose(acts, g[f"{tag}.actions.{b}.{v}"], rtol=1e-4 if b else 1e-5, atol=1e-5 if b else 1e-6)
            np.testing.assert_allclose(torch.stack(r["memories"][v]["logprobs"]).numpy(), g[f"{tag}.logp.{b}.{v}"], rtol=1e-4 if b else 1e-5,
                                       atol=1e-4 if b else 1e-5)
            for t in range(1, Tn):
                ids = np.array([i + [-1] * (c["fs"] - len(i)) for i in r["ids"][t][v]], dtype=np.int32)
                assert np.array_equal(ids, g[f"{tag}.ids.{b}.{t}.{v}"]), (b, t, v)
        pp = r["policy"]
        if stage == 3:
            r["loss"].backward()
            for name, live in (("model", mpl), ("fc", fpl)):
                used = {k: v for k, v in live.items() if v.grad is not None}
                new = O.adam_step({k: v.detach() for k, v in used.items()}, {k: v.grad for k, v in used.items()}, adam[name], c["lr"],
                                  weight_decay=c["wd"])
                (mp if name == "model" else fp).update({k: new[k] for k in used})
    for name, now, base in (("policy", pp, pp0), ("model", mp, mp0), ("fc", fp, fp0)):
        for k, v in now.items():
            key = f"{tag}.{name}_delta.{k}"
            if key not in g.files or g[key][0] == 0.0:
                assert torch.equal(v.detach(), base[k]), key
                continue
            _close_summ(_fp(v.detach() - base[k]), g[key], 3e-2, msg=key)
            _close_summ(_fp(v.detach()), g[f"{tag}.{name}.{k}"], 1e-4, msg=key)


def test_g14_clam_plain_attention_net(golden):
    """CLAM_SB(gate=False): the oracle's Attn_Net branch vs the reference (scores, soft-max, pooled vector, gradients)."""
    g = golden("g14_clam_plain")
    p = _leaf(P.clam_sb_plain(11))
    x = T(P.bags(11, "g4.x", 3, 300, 512))
    M, A, s, h = O.clam_sb_forward(p, x)
    np.testing.assert_allclose(s.detach().numpy(), g["raw"], **TOL)
    np.testing.assert_allclose(A.detach().numpy(), g["A"], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(M.detach().numpy(), g["M_batch"], **TOL)
    tot = 0
    for b in range(3):
        loss, *_ = O.clam_instance_eval(p, A[b], h[b], 1, 2, 8, True)
        tot = tot + M[b].sum() + loss
    tot.backward()
    _check_grads(g, "grad.", p)


# ------------------------------------------------------------------------------------------------ G15 (row a21)
def _g15_batch(sl, u, eps, Tn):
    """Per-slide draws of oracle/recipes.py -> the [T][B,K] / [T-1][B,K] batch form."""
    return [np.stack([u[s][t] for s in sl]) for t in range(Tn)], [np.stack([eps[s][t] for s in sl]) for t in range(Tn - 1)]


@pytest.mark.parametrize("arch", ["ABMIL", "CLAM_SB", "DSMIL"])
@pytest.mark.parametrize("stage", [1, 2, 3])
def test_g15_supervised_step_oracle_vs_reference_train_bodies(golden, arch, stage):
    """oracle/step_oracle.supervised_step_rl == the reference's own train_ABMIL / train_CLAM / train_DSMIL bodies
    (train_RLMIL.py:715-781, 323-392, 508-590): per-step losses, confidence rewards, sampler actions / log-probs / selected
    patch ids and post-update parameters after TWO optimizer steps at batch size 1 (and 4 for ABMIL), and - batched over four
    slides at frozen parameters - the mean of the reference's four batch-size-1 bodies."""
    from oracle import step_oracle as SO
    from oracle.recipes import G15, G15_RUNS, g15_inputs, g15_params
    g, c = golden("g15_supervised_steps"), G15
    Tn, fs = c["T"], c["fs"]
    Ns, feats, cls, labels, u, eps = g15_inputs()
    mp0, fp0, pp0 = (P.to_torch(d) for d in g15_params(arch))
    kw = dict(T=Tn, feat_size=fs, stage=stage, bag_weight=c["bag_weight"], k_sample=c["k_sample"], action_std=c["std"],
              gamma=c["gamma"], K_epochs=c["K_epochs"], ppo_lr=c["ppo_lr"], lr=c["lr"], fc_lr=c["fc_lr"], wd=c["wd"])
    for run, (B, steps, lr_on) in G15_RUNS.items():
        tag = f"{arch}.s{stage}.{run}"
        if f"{tag}.losses" not in g.files:
            assert B > 1 and arch != "ABMIL"
            continue
        if not lr_on:
            # ONE batched step over the four slides at the initial parameters == the mean of the four B = 1 bodies
            sl = list(range(steps))
            ub, eb = _g15_batch(sl, u, eps, Tn)
            r = SO.supervised_step_rl(arch, mp0, fp0, pp0, [feats[s] for s in sl], [cls[s] for s in sl], labels[sl], ub, eb, **kw)
            np.testing.assert_allclose([l.item() for l in r["losses"]], g[f"{tag}.losses"].mean(0), rtol=2e-5)
            np.testing.assert_allclose(r["rewards"].numpy(), g[f"{tag}.rewards"][:, :, 0].T, rtol=1e-3, atol=1e-6)
            if stage != 1:
                acts = torch.stack(r["actions"][1:]).numpy()                               # [T-1, 4, K]
                np.testing.assert_allclose(acts, g[f"{tag}.actions"][:, :, 0].transpose(1, 0, 2), rtol=1e-5, atol=1e-6)
                for t in range(1, Tn):
                    ids = np.array([i + [-1] * (fs - len(i)) for i in r["ids"][t]], dtype=np.int32)
                    want = np.concatenate([g[f"{tag}.ids.{s}.{t}"] for s in sl])
                    assert np.array_equal(ids, want), (tag, t)
            continue
        mp, fp, pp, st = mp0, fp0, pp0, {}
        for it in range(steps):
            sl = list(range(it * B, (it + 1) * B))
            ub, eb = _g15_batch(sl, u, eps, Tn)
            r = SO.supervised_step_rl(arch, mp, fp, pp, [feats[s] for s in sl], [cls[s] for s in sl], labels[sl], ub, eb,
                                      adam_state=st, **kw)
            mp, fp, pp = r["model"], r["fc"], r["policy"]
            np.testing.assert_allclose([l.item() for l in r["losses"]], g[f"{tag}.losses"][it], rtol=1e-4 if it else 2e-5,
                                       err_msg=f"{tag} step {it}")
            np.testing.assert_allclose(r["rewards"].numpy(), g[f"{tag}.rewards"][it], rtol=2e-3, atol=2e-6)
            if stage != 1:
                np.testing.assert_allclose(torch.stack(r["actions"][1:]).numpy(), g[f"{tag}.actions"][it], rtol=1e-4, atol=1e-5)
                np.testing.assert_allclose(r["logp"].numpy(), g[f"{tag}.logp"][it], rtol=1e-4, atol=1e-4)
                for t in range(1, Tn):
                    ids = np.array([i + [-1] * (fs - len(i)) for i in r["ids"][t]], dtype=np.int32)
                    assert np.array_equal(ids, g[f"{tag}.ids.{it}.{t}"]), (tag, it, t)
        for name, new, base in (("model", mp, mp0), ("fc", fp, fp0), ("policy", pp, pp0)):
            for k, v in new.items():
                key = f"{tag}.{name}_delta.{k}"
                if key not in g.files:
                    assert torch.equal(v, base[k]), (tag, name, k)         # frozen at this stage
                    continue
                want = g[key]
                if want[0] == 0.0:                                           # never applied by the reference: Adam skips it
                    assert torch.equal(v, base[k]), key
                    continue
                _close_summ(_fp(v - base[k]), want, 3e-2, msg=key)
                _close_summ(_fp(v), g[f"{tag}.{name}.{k}"], 1e-4, msg=key)


@pytest.mark.parametrize("case", ["dropout", "small", "small_dropout"])
def test_g16_abmil_outside_the_default_shape(golden, case):
    """The oracle's ABMIL with injected Dropout masks and with L / D other than 512 / 128 == the reference module in train mode."""
    from oracle.recipes import g16_inputs
    g = golden("g16_abmil_general")
    pd, x, masks = g16_inputs(case)
    p = _leaf(pd)
    out, A, s, M = O.abmil_forward(p, T(x), None if masks is None else [T(k) for k in masks])
    np.testing.assert_allclose(out.detach().numpy(), g[f"{case}.out"], **TOL)
    np.testing.assert_allclose(A.detach().numpy(), g[f"{case}.A"], rtol=2e-5, atol=1e-8)
    out.sum().backward()
    _check_grads(g, f"{case}.grad.", p)


# ------------------------------------------------------------------------------------------------ G18 / G19 / G20 (round 5)
def _check_grad_entries(gold, prefix, named, rtol=2e-4):
    """Gradients stored in full (any shape but the 34-number fingerprint) are compared entry by entry at rtol of the largest
    entry; fingerprints as in _check_grads."""
    n = 0
    for k, v in named.items():
        key = prefix + k
        if key not in gold.files:
            continue
        assert v.grad is not None, key
        want = gold[key]
        if k.endswith("attention.2.bias") or k.endswith("attention_c.bias"):
            scale = named[k[:-4] + "weight"].grad.abs().max().item()          # exact zeros in exact arithmetic: rounding noise on both sides
            assert v.grad.abs().max().item() < 1e-4 * scale and np.abs(want).max() < 1e-4 * scale, key
        elif want.shape == tuple(v.grad.shape):
            np.testing.assert_allclose(v.grad.numpy(), want, rtol=rtol, atol=rtol * np.abs(want).max(), err_msg=key)
        else:
            got = _summ(v.grad)
            np.testing.assert_allclose(got[:2], want[:2], rtol=rtol, err_msg=key)
            np.testing.assert_allclose(got[2:], want[2:], rtol=rtol, atol=rtol * want[1], err_msg=key)
        n += 1
    assert n > 0


def test_g18_clam_sb_size_big(golden):
    """CLAM_SB(size_arg="big") (clam.py:66-67): the oracle with a 384-wide attention net vs the reference."""
    g = golden("g18_clam_big")
    p = _leaf(P.clam_sb(18, size=(512, 384)))
    x = T(P.bags(18, "g18.x", 3, 300, 512))
    M, A, s, h = O.clam_sb_forward(p, x)
    np.testing.assert_allclose(s.detach().numpy(), g["raw"], **TOL)
    np.testing.assert_allclose(A.detach().numpy(), g["A"], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(M.detach().numpy(), g["M_batch"], **TOL)
    for label in (0, 1):
        for b in range(3):
            loss, preds, tgts, *_ = O.clam_instance_eval(p, A[b], h[b], label, 2, 8, True)
            np.testing.assert_allclose(float(loss), g[f"l{label}.inst_loss"][b], rtol=2e-5)
            np.testing.assert_array_equal(np.asarray(preds), g[f"l{label}.preds"][b])
            np.testing.assert_array_equal(np.asarray(tgts), g[f"l{label}.targets"][b])
    tot = 0
    for b in range(3):
        loss, *_ = O.clam_instance_eval(p, A[b], h[b], 1, 2, 8, True)
        tot = tot + M[b].sum() + loss
    tot.backward()
    _check_grad_entries(g, "grad.", p)


def test_g19_abmil_with_three_attention_heads(golden):
    """ABMIL(K=3) (abmil.py:8,23-27,38-44): out [B*K, L] bag-major, A [B, K, N], gradients of a weighted sum."""
    g = golden("g19_abmil_heads")
    p = _leaf(P.abmil(19, K=3, dim_out=2))
    x = T(P.bags(19, "g19.x", 3, 200, 512))
    out, A = O.abmil_forward_heads(p, x)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    np.testing.assert_allclose(A.detach().numpy(), g["A"], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(O.abmil_forward_heads(p, x[:1])[0].detach().numpy(), g["out_single"], **TOL)
    (out * T(detrand.normal(19, "g19.w", (9, 512)))).sum().backward()
    _check_grad_entries(g, "grad.", p)


def test_g20_abmil_every_gradient_entry(golden):
    """G1 with the parameter gradients stored in full: every entry of every gradient within 2e-5 of the largest."""
    g = golden("g20_abmil_full_grads")
    p = _leaf(P.abmil(985))
    out, *_ = O.abmil_forward(p, T(P.bags(985, "g1.x", 4, 256, 512)))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    out.sum().backward()
    _check_grad_entries(g, "grad.", p, rtol=2e-5)


@pytest.mark.parametrize("name", ["ce_weighted_sum", "multi_margin", "lambda_logit_gap"])
@pytest.mark.parametrize("label", [0, 1])
def test_g22_clam_custom_instance_loss(golden, name, label):
    """G22: CLAM_SB(instance_loss_fn=<callable>) - the reference calls whatever loss it was constructed with on every evaluated
    class (clam.py:64-65,118,131): per-bag instance losses and the gradients of bag + instance objective for three non-default
    losses (the same callables the generator handed to the reference: oracle/gen_goldens.instance_losses is re-stated here)."""
    g = golden("g22_clam_custom_instance_loss")
    fn = _instance_losses()[name]
    p = _leaf(P.clam_sb(11))
    x = T(P.bags(11, "g4.x", 3, 300, 512))
    M, A, s, h = O.clam_sb_forward(p, x)
    tot = M.sum()
    for b in range(3):
        loss = O.clam_instance_eval(p, A[b], h[b], label, 2, 8, True, loss_fn=fn)[0]
        np.testing.assert_allclose(loss.item(), g[f"{name}.l{label}.inst_loss"][b], rtol=1e-5)
        tot = tot + loss
    tot.backward()
    _check_grad_entries(g, f"{name}.l{label}.grad.", p)


def _instance_losses():
    return {"ce_weighted_sum": torch.nn.CrossEntropyLoss(weight=torch.tensor([0.7, 1.3]), reduction="sum"),
            "multi_margin": torch.nn.MultiMarginLoss(),
            "lambda_logit_gap": lambda lg, tg: ((lg[:, 1] - lg[:, 0]) * (1.0 - 2.0 * tg.float())).exp().mean()}


def _dsmil_keep_mask(seed, B, N, d, p_drop):
    return T(((detrand.uniform(seed, "g23.keep", (B, N, d)) >= p_drop).astype(np.float32) / np.float32(1.0 - p_drop)).astype(np.float32))


def test_g23_dsmil_dropout_v(golden):
    """G23: BClassifier(dropout_v = 0.25) in training mode with an injected keep mask (dsmil.py:53-59,66): the value branch sees the
    dropped features, scores and attention the un-dropped ones - outputs and gradients of the reference's own forward."""
    g = golden("g23_dsmil_dropout_v")
    p = _leaf(P.dsmil(23, 512, 2))
    x = T(P.bags(23, "g23.x", 3, 200, 512))
    c, bag, A, m = O.dsmil_forward(p, x, keep_v=_dsmil_keep_mask(23, 3, 200, 512, 0.25))
    np.testing.assert_allclose(c.detach().numpy(), g["classes"], **TOL)
    np.testing.assert_allclose(bag.detach().numpy(), g["bag"], rtol=1e-4, atol=1e-5)
    wb_, wc_ = T(detrand.normal(23, "g23.wb", (3, 2, 512))), T(detrand.normal(23, "g23.wc", (3, 200, 2)))
    ((bag * wb_).sum() + (c * wc_).sum()).backward()
    _check_grad_entries(g, "grad.", p)
    # and it is not the un-dropped forward
    assert (O.dsmil_forward(p, x)[1] - bag).abs().max().item() > 1e-3

