"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE ONLY.  Imports /root/reference (never shipped, never read at test
time), drives it with oracle/detrand.py inputs and oracle/params.py weights loaded via
``load_state_dict``, and stores *outputs only* (the inputs are regenerated from seeds).
Two shims: ``Tensor.cuda``/``Module.cuda`` become the identity,
because dsmil.py and rlmil.py hard-code ``.cuda()``.

    python -m oracle.gen_goldens        # from the repo root
"""
import os
import sys
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
from models import abmil as r_abmil, clam as r_clam, dsmil as r_dsmil, cl as r_cl, rlmil as r_rlmil  # noqa: E402
from utils import losses as r_losses, datasets as r_datasets  # noqa: E402
sys.path.pop(0)

from oracle import detrand, params as P  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
T = torch.from_numpy


def _summ(g):
    """Fingerprint of a tensor: L2 norm, max |.|, first 32 values."""
    g = g.detach().double().flatten()
    return np.concatenate([[g.norm().item(), g.abs().max().item()], g[:32].numpy()])


def g1_abmil():
    seed, B, N, d = 985, 4, 256, 512
    m = r_abmil.ABMIL(d, L=512, D=128, dim_out=128)
    m.load_state_dict(P.to_torch(P.abmil(seed)))
    x = T(P.bags(seed, "g1.x", B, N, d))
    out, det = m(x)
    out.sum().backward()
    # attention weights: re-run the reference's own sub-modules per bag
    A = []
    with torch.no_grad():
        for b in range(B):
            H = m.encoder(x[b])
            a = torch.softmax(m.attention(H).t(), dim=1)
            A.append(a / np.sqrt(a.shape[-1]))
    res = {"out": out.detach().numpy(), "A": torch.cat(A).numpy()}
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = _summ(v.grad)
    # single-bag path (x.shape[0]==1 branch, abmil.py:57-58)
    o1, _ = m(x[:1])
    res["out_single"] = o1.detach().numpy()
    np.savez(os.path.join(OUT, "g1_abmil.npz"), **res)


def g2_ntxent():
    res = {}
    for B in (2, 4, 64):
        for tau in (1.0, 0.5):
            zi = T(detrand.normal(7, f"g2.zi.{B}", (B, 128))).requires_grad_()
            zj = T(detrand.normal(7, f"g2.zj.{B}", (B, 128))).requires_grad_()
            loss = r_losses.NT_Xent(B, tau)(zi, zj)
            loss.backward()
            res[f"loss.{B}.{tau}"] = loss.detach().numpy()
            res[f"dzi.{B}.{tau}"] = zi.grad.numpy()
            res[f"dzj.{B}.{tau}"] = zj.grad.numpy()
    np.savez(os.path.join(OUT, "g2_ntxent.npz"), **res)


def g3_pretrain():
    """C1-shaped pre-train step: CL(ABMIL) + Full_layer + NT_Xent, T=1 and T=3, sub-bags injected."""
    seed, B, N, d = 985, 4, 256, 512
    res = {}
    for Tn in (1, 3):
        enc = r_abmil.ABMIL(d, L=512, D=128, dim_out=128)
        enc.load_state_dict(P.to_torch(P.abmil(seed)))
        model = r_cl.CL(enc, projection_dim=128, n_features=512)
        fc = r_rlmil.Full_layer(512, 1024, True, 128)
        fc.load_state_dict(P.to_torch(P.full_layer(seed)))
        crit = r_losses.NT_Xent(B, 1.0)
        losses, rewards, sim_last = [], [], None
        for t in range(Tn):                                  # mirrors train_MuRCL.py:242-288
            xv = [T(P.bags(seed, f"g3.x.{t}.{v}", B, N, d)) for v in range(2)]
            outs, states = model(xv)
            outs = [fc(o, restart=(t == 0)) for o in outs]
            losses.append(crit(outs[0], outs[1]))
            sim = torch.cosine_similarity(outs[0], outs[1]).view(1, -1)
            if t > 0:
                rewards.append((sim_last - sim).detach().numpy())
            sim_last = sim
        loss = sum(losses) / Tn
        loss.backward()
        res[f"T{Tn}.loss"] = loss.detach().numpy()
        res[f"T{Tn}.losses"] = np.array([l.item() for l in losses])
        if rewards:
            res[f"T{Tn}.rewards"] = np.concatenate(rewards, 0)
        for k, v in list(model.named_parameters()) + [("fc::" + k, v) for k, v in fc.named_parameters()]:
            if v.grad is not None:
                res[f"T{Tn}.grad.{k}"] = _summ(v.grad)
    np.savez(os.path.join(OUT, "g3_pretrain.npz"), **res)


def _topk_margin(A, k):
    s = torch.sort(A.flatten(), descending=True)[0]
    return min((s[k - 1] - s[k]).item(), (s[-k - 1] - s[-k]).item()) / s[0].item()


def g4_clam():
    seed, B, N, d = 11, 3, 300, 512
    res = {}
    x = T(P.bags(seed, "g4.x", B, N, d))
    for subtyping in (False, True):
        m = r_clam.CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2,
                           subtyping=subtyping, in_dim=d)
        m.load_state_dict(P.to_torch(P.clam_sb(seed)))
        m.eval()
        tag = f"sub{int(subtyping)}"
        for label in (0, 1):
            xx = x.clone().requires_grad_()
            M, A_or, rd = None, None, None
            Ms, losses, preds, tgts = [], [], [], []
            for b in range(B):
                Mb, rdb = m.bag_forward(xx[b], label=torch.tensor([label]), instance_eval=True)
                Ms.append(Mb)
                losses.append(rdb["instance_loss"])
                preds.append(rdb["inst_preds"])
                tgts.append(rdb["inst_labels"])
            res[f"{tag}.l{label}.M"] = torch.cat(Ms).detach().numpy()
            res[f"{tag}.l{label}.inst_loss"] = np.array([float(l) for l in losses])
            res[f"{tag}.l{label}.preds"] = np.stack(preds)
            res[f"{tag}.l{label}.targets"] = np.stack(tgts)
        with torch.no_grad():
            raw = torch.cat([m.bag_forward(x[b], attention_only=True) for b in range(B)])   # [B,N]
            A = torch.softmax(raw, 1)
            for b in range(B):
                assert _topk_margin(A[b], 8) > 1e-4, "golden top-k margin too small"
            res[f"{tag}.raw"] = raw.numpy()
            res[f"{tag}.A"] = A.numpy()
            res[f"{tag}.top_p"] = torch.topk(A, 8)[1].numpy()
            res[f"{tag}.top_n"] = torch.topk(-A, 8)[1].numpy()
            # batch forward path (clam.py:183-211)
            Mb, _ = m(x)
            res[f"{tag}.M_batch"] = Mb.numpy()
    # grads of a combined objective in eval mode (bag + instance loss), label 1, subtyping
    m = r_clam.CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=d)
    m.load_state_dict(P.to_torch(P.clam_sb(seed)))
    m.eval()
    tot = 0
    for b in range(B):
        Mb, rdb = m.bag_forward(x[b], label=torch.tensor([1]), instance_eval=True)
        tot = tot + Mb.sum() + rdb["instance_loss"]
    tot.backward()
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = _summ(v.grad)
    np.savez(os.path.join(OUT, "g4_clam.npz"), **res)


def g14_clam_plain():
    """CLAM_SB(gate=False): the plain Attn_Net (clam.py:18-34,80-81) - pooled M, soft-max A, raw scores, the instance
    branch for label 1 and the gradients of a combined objective, eval mode."""
    seed, B, N, d = 11, 3, 300, 512
    x = T(P.bags(seed, "g4.x", B, N, d))
    m = r_clam.CLAM_SB(gate=False, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=d)
    m.load_state_dict(P.to_torch(P.clam_sb_plain(seed)))
    m.eval()
    res = {}
    with torch.no_grad():
        raw = torch.cat([m.bag_forward(x[b], attention_only=True) for b in range(B)])
        A = torch.softmax(raw, 1)
        for b in range(B):
            assert _topk_margin(A[b], 8) > 1e-4, "golden top-k margin too small"
        res["raw"], res["A"] = raw.numpy(), A.numpy()
        res["top_p"], res["top_n"] = torch.topk(A, 8)[1].numpy(), torch.topk(-A, 8)[1].numpy()
        res["M_batch"] = m(x)[0].numpy()
    tot, losses = 0, []
    for b in range(B):
        Mb, rdb = m.bag_forward(x[b], label=torch.tensor([1]), instance_eval=True)
        tot = tot + Mb.sum() + rdb["instance_loss"]
        losses.append(float(rdb["instance_loss"]))
    tot.backward()
    res["inst_loss"] = np.array(losses)
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = _summ(v.grad)
    np.savez(os.path.join(OUT, "g14_clam_plain.npz"), **res)


def g5_dsmil():
    seed, B, N, d, C = 5, 3, 200, 512, 2
    m = r_dsmil.build_dsmil(d, C)
    m.load_state_dict(P.to_torch(P.dsmil(seed, d, C)))
    x = T(P.bags(seed, "g5.x", B, N, d))
    res = {}
    classes, bag, _ = m(x)                                   # batch path -> lists
    res["classes"] = torch.stack(classes).detach().numpy()
    res["bag"] = bag.detach().numpy()
    c = torch.stack(classes)
    srt = torch.sort(c, 1, descending=True)
    assert ((srt[0][:, 0] - srt[0][:, 1]).abs() > 1e-5).all(), "arg-max margin too small"
    res["m_ids"] = srt[1][:, 0, :].numpy()
    c1, b1, _ = m(x[:1])                                     # single-bag path
    res["classes_single"] = c1.detach().numpy()
    res["bag_single"] = b1.detach().numpy()
    (bag.sum() + sum(cc.max(0)[0].sum() for cc in classes)).backward()
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = _summ(v.grad)
    np.savez(os.path.join(OUT, "g5_dsmil.npz"), **res)


def _ref_get_feats(feats, clusters, actions, feat_size):
    fl = [T(f).unsqueeze(0) for f in feats]
    out = r_datasets.get_feats(fl, clusters, T(np.asarray(actions, np.float32)), feat_size)
    return out.numpy()


def g6_get_feats():
    """Index selection cases.  Feature value of patch i is i+1 in column 0, so the
    selected ids can be read back from the reference's output tensor exactly."""
    res = {}
    cases = {}
    # (a) N >> feat_size, random clusters / actions
    N, K, fs = 4000, 10, 256
    cl = P.cluster_lists(3, "g6.a", N, K)
    cases["a"] = (N, [cl, P.cluster_lists(3, "g6.a2", N, K)], detrand.uniform(3, "g6.a.act", (2, K)), fs)
    # (b) N < feat_size: negative-slice quirk
    sizes = (10, 15, 5)
    ids = np.arange(30)
    clb = [ids[:10].tolist(), ids[10:25].tolist(), ids[25:].tolist()]
    cases["b"] = (30, [clb], np.full((1, 3), 0.5, np.float32), 40)
    # (c) action edges 0 and 1 (+ tiny/empty clusters)
    N, K, fs = 1000, 6, 128
    clc = P.cluster_lists(4, "g6.c", N, K - 1) + [[]]
    cases["c"] = (N, [clc, clc, clc],
                  np.array([[0] * K, [1] * K, [0, 1, 0.999999, 1e-7, 0.5, 0.3]], np.float32), fs)
    # (d) half-way rounding: n_j * ratio exactly .5 -> round-half-even; N=64, fs=16 -> ratio .25
    cld, start = [], 0
    for n in (2, 6, 10, 14, 18, 14):
        cld.append(list(range(start, start + n)))
        start += n
    cases["d"] = (64, [cld, cld], detrand.uniform(5, "g6.d.act", (2, 6)), 16)
    # (e) total > feat_size after rounding -> truncation branch (:304-305)
    cle, start = [], 0
    for n in (6, 6, 6, 6, 6, 6, 6, 6, 6, 6):
        cle.append(list(range(start, start + n)))
        start += n
    cases["e"] = (60, [cle], detrand.uniform(6, "g6.e.act", (1, 10)), 36)   # ratio .6 -> 3.6 -> 4 each = 40 > 36
    for name, (N, cls, act, fs) in cases.items():
        feats = []
        for _ in cls:
            f = np.zeros((N, 4), np.float32)
            f[:, 0] = np.arange(N) + 1
            f[:, 1] = 7.0
            feats.append(f)
        out = _ref_get_feats(feats, cls, act, fs)
        res[f"{name}.ids_plus1"] = out[:, :, 0].astype(np.int64)      # 0 == padded row
        res[f"{name}.N"] = np.array(N)
        res[f"{name}.fs"] = np.array(fs)
        res[f"{name}.act"] = np.asarray(act, np.float32)
        res[f"{name}.cluster_flat"] = np.array([np.concatenate([np.asarray(c, np.int64) for c in cl] or [[]]) for cl in cls][0])
        res[f"{name}.cluster_sizes"] = np.array([[len(c) for c in cl] for cl in cls])
    np.savez(os.path.join(OUT, "g6_get_feats.npz"), **res)


def g7_mixup():
    B, N, d = 5, 16, 8
    x = T(detrand.normal(9, "g7.x", (B, N, d)))
    u = T(detrand.uniform(9, "g7.u", (B, 1)))
    perm = T(detrand.permutation(9, "g7.perm", B))
    with mock.patch("torch.rand", lambda *a, **k: u), mock.patch("torch.randperm", lambda *a, **k: perm):
        out, lam, idx = r_datasets.mixup(x, 0.9)
    np.savez(os.path.join(OUT, "g7_mixup.npz"), out=out.numpy(), lam=lam.numpy(), perm=idx.numpy(), u=u.numpy())


def g8_ppo():
    seed, B, S, H, K, Tm = 21, 6, 512, 512, 10, 3
    std = 0.5
    res = {}
    ppo = r_rlmil.PPO(512, S, H, False, action_std=std, lr=1e-3, gamma=0.1, K_epochs=1, action_size=K)
    sd = P.to_torch(P.actor_critic(seed, S, H, K))
    ppo.policy.load_state_dict(sd)
    ppo.policy_old.load_state_dict(sd)
    mem = r_rlmil.Memory()
    MVN = torch.distributions.multivariate_normal.MultivariateNormal
    for t in range(Tm):
        st = T(detrand.normal(seed, f"g8.s{t}", (B, S)))
        eps = T(detrand.normal(seed, f"g8.e{t}", (B, K)))
        with mock.patch.object(MVN, "sample", lambda self, *a, **k: self.loc + std * eps):
            a = ppo.select_action(st, mem, restart_batch=(t == 0))
        res[f"act.{t}"] = a.numpy()
        res[f"logp.{t}"] = mem.logprobs[-1].detach().numpy()
        res[f"hidden.{t}"] = mem.hidden[-1].detach().numpy()[0]
        mem.rewards.append(T(detrand.normal(seed, f"g8.r{t}", (1, B)) * 0.1))
    with torch.no_grad():
        lp, v, ent = ppo.policy.evaluate(torch.stack(mem.states, 0), torch.stack(mem.actions, 0))
    res["eval.logp"], res["eval.value"], res["eval.entropy"] = lp.numpy(), v.numpy(), ent.numpy()
    ppo.update(mem)
    for k, v in ppo.policy.state_dict().items():
        res["post." + k] = _summ(v)
    np.savez(os.path.join(OUT, "g8_ppo.npz"), **res)


def g9_full_layer():
    seed, B = 13, 4
    fc = r_rlmil.Full_layer(512, 1024, True, 128)
    fc.load_state_dict(P.to_torch(P.full_layer(seed)))
    res = {}
    with torch.no_grad():
        for t in range(3):
            for v in range(2):
                x = T(detrand.normal(seed, f"g9.x.{t}.{v}", (B, 512)))
                res[f"z.{t}.{v}"] = fc(x, restart=(t == 0)).numpy()
                res[f"h.{t}.{v}"] = fc.hidden[0].numpy()
    np.savez(os.path.join(OUT, "g9_full_layer.npz"), **res)


def _ref_train_rlmil():
    """The reference's train_RLMIL module (needs a stand-in for the absent tensorboard import, SURVEY 8(c) shim 2)."""
    import types
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    sys.modules.setdefault("tensorboard", types.ModuleType("tensorboard"))
    sys.path.insert(0, REF)
    try:
        import train_RLMIL as r_train
        from utils import general as r_general
    finally:
        sys.path.pop(0)
    return r_train, r_general


G10_SPLIT = dict(S=7, K=4, fs=64, T=3, d=512, C=2)


def g10_inputs(seed=71):
    """The small validation split of G10 (shared with tests/test_gpu_eval.py through this function's recipe)."""
    c = G10_SPLIT
    Ns = [260 - 23 * b for b in range(c["S"])]
    feats = [P.bags(seed, f"g10.f{b}", 1, Ns[b], c["d"])[0] for b in range(c["S"])]
    cls = [P.cluster_lists(seed, f"g10.c{b}", Ns[b], c["K"]) for b in range(c["S"])]
    labels = np.array([0, 1, 1, 0, 1, 0, 0], dtype=np.int64)
    acts = [detrand.uniform(seed, f"g10.a{t}", (c["S"], c["K"])).astype(np.float32) for t in range(c["T"])]
    return Ns, feats, cls, labels, acts


def g10_eval():
    """8(f) rank 2: the reference's own validation bodies (train_RLMIL.py test_ABMIL / test_CLAM / test_DSMIL) and
    utils/general.get_metrics / get_score, with the random sub-bag actions injected."""
    r_train, r_general = _ref_train_rlmil()
    res = {}
    # ---- metrics alone: binary and 3-class
    for name, n, C in (("bin", 14, 2), ("tri", 18, 3)):
        out = T(detrand.normal(72, f"g10.m.{name}", (n, C)).astype(np.float32))
        tgt = torch.from_numpy(np.arange(n) % C)
        m = r_general.get_metrics(out, tgt)
        res[f"metrics.{name}"] = np.array(m, dtype=np.float64)
        res[f"score.{name}"] = np.float64(r_general.get_score(*m))
    # ---- whole-split evaluation
    c = G10_SPLIT
    Ns, feats, cls, labels, acts = g10_inputs()
    test_set = [(T(f), cl, torch.tensor(int(y)), f"case{i}") for i, (f, cl, y) in enumerate(zip(feats, cls, labels))]
    import argparse
    for arch in ("ABMIL", "CLAM_SB", "DSMIL"):
        if arch == "ABMIL":
            model = r_abmil.ABMIL(c["d"], L=512, D=128, dim_out=c["C"])
            model.load_state_dict(P.to_torch(P.abmil(73, dim_out=c["C"])))
        elif arch == "CLAM_SB":
            model = r_clam.CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=c["C"], subtyping=True, in_dim=c["d"])
            model.load_state_dict(P.to_torch(P.clam_sb(73)))
        else:
            model = r_dsmil.build_dsmil(c["d"], c["C"])
            model.load_state_dict(P.to_torch(P.dsmil(73)))
        fc = r_rlmil.Full_layer(512, 1024, True, c["C"])
        fc.load_state_dict(P.to_torch(P.full_layer(73, 512, 1024, c["C"])))
        args = argparse.Namespace(T=c["T"], device="cpu", num_clusters=c["K"], feat_size=c["fs"], train_stage=1, bag_weight=0.7)
        draws = iter([T(a) for a in acts])
        with mock.patch.object(torch, "rand", lambda *a, **k: next(draws)):
            fn = {"ABMIL": r_train.test_ABMIL, "CLAM_SB": r_train.test_CLAM, "DSMIL": r_train.test_DSMIL}[arch]   # its TEST map
            out = fn(args, test_set, model, fc, None, r_rlmil.Memory(), torch.nn.CrossEntropyLoss())
        loss, acc, auc, prec, rec, f1, outputs, labs, case_ids = out
        res[f"{arch}.loss"] = np.float64(loss)
        res[f"{arch}.metrics"] = np.array([acc, auc, prec, rec, f1], dtype=np.float64)
        res[f"{arch}.outputs"] = outputs.detach().numpy()
        assert labs.tolist() == labels.tolist() and case_ids[2] == "case2"
    np.savez(os.path.join(OUT, "g10_eval.npz"), **res)


def _ref_train_murcl():
    """The reference's train_MuRCL module (stand-in for the absent tensorboard import, SURVEY 8(c) shim 2)."""
    import types
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    sys.modules.setdefault("tensorboard", types.ModuleType("tensorboard"))
    sys.path.insert(0, REF)
    try:
        import train_MuRCL as r_tm
    finally:
        sys.path.pop(0)
    return r_tm


from oracle.recipes import G12, g12_inputs, window_margin as _window_margin  # noqa: E402


def g12_rl_step():
    """BASELINE config 4's body: ONE batch through the reference's own ``train()`` (train_MuRCL.py:189-343) at
    train_stage 2 and 3, T = 3 and 6, with torch.rand / torch.randperm / MultivariateNormal.sample replaced by the
    injected draws.  Stored: the sampler's actions and log-probs per patch step, the patch ids those actions select,
    per-step losses, rewards, and fingerprints of the policy (stage 2) / model + head (stage 3) after the update."""
    import argparse
    import tempfile
    from oracle import select_oracle as S
    r_tm = _ref_train_murcl()
    c = G12
    seed, B, K, fs, std = c["seed"], c["B"], c["K"], c["fs"], c["std"]
    MVN = torch.distributions.multivariate_normal.MultivariateNormal
    res = {}
    for Tn in (3, 6):
        Ns, feats, cls, inj = g12_inputs(Tn)
        for stage in (2, 3):
            tag = f"T{Tn}.s{stage}"
            enc = r_abmil.ABMIL(c["d"], L=512, D=128, dim_out=128)
            enc.load_state_dict(P.to_torch(P.abmil(seed)))
            model = torch.nn.DataParallel(r_cl.CL(enc, projection_dim=128, n_features=512))      # train_MuRCL.py:145 (no GPU: pass-through)
            fc = r_rlmil.Full_layer(512, 1024, True, 128)
            fc.load_state_dict(P.to_torch(P.full_layer(seed)))
            ppo = r_rlmil.PPO(c["d"], 512, 512, False, action_std=std, lr=c["ppo_lr"], gamma=c["gamma"], K_epochs=c["K_epochs"],
                              action_size=K)
            sd = P.to_torch(P.actor_critic(seed, 512, 512, K))
            ppo.policy.load_state_dict(sd)
            ppo.policy_old.load_state_dict(sd)
            optimizer = None if stage == 2 else torch.optim.Adam(
                [{"params": model.parameters(), "lr": c["lr"]}, {"params": fc.parameters(), "lr": c["lr"]}],
                betas=(0.9, 0.999), weight_decay=c["wd"])                                          # train_MuRCL.py:165
            args = argparse.Namespace(T=Tn, device="cpu", num_clusters=K, feat_size=fs, train_stage=stage, batch_size=B,
                                      epochs=1, num_data=B, eval_step=1, alpha=c["alpha"], patience=None, warmup=0)

            class _Set:
                def shuffle(self):
                    pass

                def __len__(self):
                    return B

                def __getitem__(self, i):
                    return T(feats[i]), cls[i], 0, f"case{i}"

            act_draws = iter([T(a) for a in inj["actions"][0]])
            u_draws = iter([T(inj["u"][t][v]) for t in range(Tn) for v in range(2)])
            perm_draws = iter([T(inj["perm"][t][v]) for t in range(Tn) for v in range(2)])
            eps_draws = iter([T(inj["eps"][t][v]) for t in range(Tn - 1) for v in range(2)])

            def fake_rand(*a, **k):
                size = tuple(k["size"]) if "size" in k else (tuple(a[0]) if isinstance(a[0], (tuple, list)) else tuple(a))
                return next(act_draws) if size == (B, K) else next(u_draws)

            snaps, step_losses = [], []
            inner = r_losses.NT_Xent(B, 1.0)

            class _Crit(torch.nn.Module):
                def forward(self, zi, zj):
                    l = inner(zi, zj)
                    step_losses.append(l.item())
                    return l

            def snap_clear(self):
                snaps.append({f: [x.detach().clone() for x in getattr(self, f)] for f in ("actions", "states", "logprobs", "rewards")})
                for f in ("actions", "states", "logprobs", "rewards", "is_terminals", "hidden"):
                    del getattr(self, f)[:]

            real_to = torch.Tensor.to

            def to_ignoring_ordinals(self, *a, **k):              # `s.to(0)` (train_MuRCL.py:262,265): device 0 == here
                return self if (a and isinstance(a[0], int)) else real_to(self, *a, **k)

            with tempfile.TemporaryDirectory() as tmp, \
                    mock.patch("torch.rand", fake_rand), mock.patch("torch.randperm", lambda *a, **k: next(perm_draws)), \
                    mock.patch.object(MVN, "sample", lambda self, *a, **k: self.loc + std * next(eps_draws)), \
                    mock.patch.object(r_rlmil.Memory, "clear_memory", snap_clear), \
                    mock.patch.object(torch.Tensor, "to", to_ignoring_ordinals):
                r_tm.train(args, _Set(), model, fc, ppo, _Crit(), optimizer, None, None, tmp)
            assert len(snaps) == 2 and len(step_losses) == Tn
            for it in (act_draws, u_draws, perm_draws, eps_draws):
                assert next(it, None) is None, "an injected draw was not consumed"
            res[f"{tag}.losses"] = np.array(step_losses)
            res[f"{tag}.rewards"] = torch.cat(snaps[0]["rewards"], 0).numpy()                     # [T-1, B]
            assert all(torch.equal(a, b) for a, b in zip(snaps[0]["rewards"], snaps[1]["rewards"]))
            margin = 1.0
            for v in range(2):
                res[f"{tag}.actions.{v}"] = torch.stack(snaps[v]["actions"], 0).numpy()            # [T-1, B, K]
                res[f"{tag}.logp.{v}"] = torch.stack(snaps[v]["logprobs"], 0).numpy()
                res[f"{tag}.states.{v}"] = np.stack([_summ(s_) for s_ in snaps[v]["states"]])
                for t in range(Tn - 1):
                    a_t = snaps[v]["actions"][t].numpy()
                    _, ids = S.get_feats(feats, cls, a_t, fs)
                    res[f"{tag}.ids.{t + 1}.{v}"] = np.array([i + [-1] * (fs - len(i)) for i in ids], dtype=np.int32)
                    margin = min(margin, min(_window_margin(Ns[b], cls[b], a_t[b], fs) for b in range(B)))
            assert margin > 2e-3, f"golden window margin too small ({margin}): pick another seed"
            res[f"{tag}.window_margin"] = np.float64(margin)
            pre = P.to_torch(P.actor_critic(seed, 512, 512, K))
            for k, v in ppo.policy.state_dict().items():
                res[f"{tag}.policy.{k}"] = _summ(v)
                res[f"{tag}.policy_delta.{k}"] = _summ(v - pre[k])
            if stage == 3:
                pre_m, pre_f = P.to_torch(P.abmil(seed)), P.to_torch(P.full_layer(seed))
                for k, v in model.module.encoder.state_dict().items():
                    res[f"{tag}.model.{k}"] = _summ(v)
                    res[f"{tag}.model_delta.{k}"] = _summ(v - pre_m[k])
                for k, v in fc.state_dict().items():
                    res[f"{tag}.fc.{k}"] = _summ(v)
                    res[f"{tag}.fc_delta.{k}"] = _summ(v - pre_f[k])
    np.savez(os.path.join(OUT, "g12_rl_step.npz"), **res)



def g15_supervised_steps():
    """Row a21 / BASELINE config 5's path: batches through the reference's own ``train_ABMIL`` / ``train_CLAM`` /
    ``train_DSMIL`` (train_RLMIL.py:715-781, 323-392, 508-590) at train_stage 1, 2 and 3, T = 3, with torch.rand and
    MultivariateNormal.sample replaced by the injected draws and Dropout switched off (its Philox stream cannot be matched).
    Runs (oracle/recipes.py G15_RUNS): two consecutive optimizer steps at the scripts' own batch size 1; four batch-size-1
    bodies at frozen parameters (= the per-slide terms of the product's batched step); for ABMIL - the one body the
    reference can batch (CLAM's result dict and DSMIL's instance scores are lists for B > 1, :335,516) - two steps of four.
    ``args.train_model_prime`` (:719) is read by train_ABMIL but defined by no parser: True here (the t = 0 term trains,
    as in the two other bodies).
    Stored per (arch, stage, run): per-step losses [steps,T], rewards [steps,T-1,B], the last patch step's logits, at
    stages 2 / 3 the sampler's actions / log-probs / selected patch ids, and the post-update fingerprints of the aggregator
    + head (stages 1, 3) or the sampler (stage 2)."""
    import argparse
    from oracle import select_oracle as S
    from oracle.recipes import G15, G15_RUNS, g15_inputs, g15_params, window_margin
    r_train, _ = _ref_train_rlmil()
    c = G15
    K, fs, Tn, C, std = c["K"], c["fs"], c["T"], c["C"], c["std"]
    Ns, feats, cls, labels, u, eps = g15_inputs()
    MVN = torch.distributions.multivariate_normal.MultivariateNormal
    res, margin = {}, 1.0
    for arch in ("ABMIL", "CLAM_SB", "DSMIL"):
        mp, fp, pp = (P.to_torch(d) for d in g15_params(arch))
        for stage in (1, 2, 3):
            for run, (B, steps, lr_on) in G15_RUNS.items():
                if B > 1 and arch != "ABMIL":
                    continue
                tag = f"{arch}.s{stage}.{run}"
                if arch == "ABMIL":
                    model = r_abmil.ABMIL(c["d"], L=512, D=128, dim_out=C, dropout=0.0)
                elif arch == "CLAM_SB":
                    model = r_clam.CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=c["k_sample"], n_classes=C,
                                           subtyping=True, in_dim=c["d"])                         # train_RLMIL.py:102-110
                else:
                    model = r_dsmil.build_dsmil(c["d"], C)
                model.load_state_dict(mp)
                model = torch.nn.DataParallel(model)                                             # :234 (no GPU: pass-through)
                fc = r_rlmil.Full_layer(512, 1024, True, C)
                fc.load_state_dict(fp)
                ppo = None
                if stage != 1:
                    ppo = r_rlmil.PPO(c["d"], 512, 512, False, action_std=std, lr=c["ppo_lr"] if lr_on else 0.0, gamma=c["gamma"],
                                      K_epochs=c["K_epochs"], action_size=K)                     # :206-211
                    ppo.policy.load_state_dict(pp)
                    ppo.policy_old.load_state_dict(pp)
                optimizer = None if stage == 2 else torch.optim.Adam(
                    [{"params": model.parameters(), "lr": c["lr"] if lr_on else 0.0},
                     {"params": fc.parameters(), "lr": c["fc_lr"] if lr_on else 0.0}], betas=(0.9, 0.999), weight_decay=c["wd"])   # :257-267
                n_data = B * steps
                args = argparse.Namespace(T=Tn, device="cpu", num_clusters=K, feat_size=fs, train_stage=stage, batch_size=B,
                                          num_data=n_data, bag_weight=c["bag_weight"], epochs=1, eval_step=steps, warmup=0,
                                          train_model_prime=True)

                class _Set:
                    def shuffle(self):
                        pass

                    def __len__(self):
                        return n_data

                    def __getitem__(self, i):
                        return T(feats[i]), cls[i], torch.tensor(int(labels[i])), f"case{i}"

                # draws in consumption order: per optimizer step, torch.rand at t = 0 (and at every t at stage 1), the
                # sampler's noise at t >= 1 otherwise
                rand_q, eps_q = [], []
                for st in range(steps):
                    sl = range(st * B, (st + 1) * B)
                    for t in range(Tn if stage == 1 else 1):
                        rand_q.append(T(np.stack([u[s][t] for s in sl])))
                    if stage != 1:
                        for t in range(Tn - 1):
                            eps_q.append(T(np.stack([eps[s][t] for s in sl])))
                rand_it, eps_it = iter(rand_q), iter(eps_q)
                meters, snaps = [], []

                class _Meter:                                        # stands in for utils.general.AverageMeter: records
                    def __init__(self):
                        self.vals, self.avg = [], 0.0
                        meters.append(self)

                    def update(self, val, n=1):
                        self.vals.append(float(val))
                        self.avg = float(np.mean(self.vals))

                def snap_clear(self):
                    snaps.append({f: [x.detach().clone() for x in getattr(self, f)] for f in ("actions", "states", "logprobs", "rewards")})
                    for f in ("actions", "states", "logprobs", "rewards", "is_terminals", "hidden"):
                        del getattr(self, f)[:]

                real_to = torch.Tensor.to

                def to_ignoring_ordinals(self, *a, **k):              # `states.to(0)` (:352,354): device 0 == here
                    return self if (a and isinstance(a[0], int)) else real_to(self, *a, **k)

                fn = {"ABMIL": r_train.train_ABMIL, "CLAM_SB": r_train.train_CLAM, "DSMIL": r_train.train_DSMIL}[arch]
                with mock.patch("torch.rand", lambda *a, **k: next(rand_it)), \
                        mock.patch.object(MVN, "sample", lambda self, *a, **k: self.loc + std * next(eps_it)), \
                        mock.patch("torch.nn.functional.dropout", lambda x, *a, **k: x), \
                        mock.patch.object(r_train, "AverageMeter", _Meter), \
                        mock.patch.object(r_rlmil.Memory, "clear_memory", snap_clear), \
                        mock.patch.object(torch.Tensor, "to", to_ignoring_ordinals):
                    out = fn(args, 0, _Set(), model, fc, ppo, r_rlmil.Memory(), torch.nn.CrossEntropyLoss(), optimizer, None)
                assert next(rand_it, None) is None and next(eps_it, None) is None, "an injected draw was not consumed"
                assert len(snaps) == steps and len(meters) == 3 * Tn - 1
                res[f"{tag}.losses"] = np.array([m.vals for m in meters[:Tn]]).T                  # [steps, T]
                res[f"{tag}.rewards"] = np.stack([torch.cat(sn["rewards"], 0).numpy() for sn in snaps])   # [steps, T-1, B]
                res[f"{tag}.last_loss_avg"] = np.float64(out[0])
                if stage != 1:
                    res[f"{tag}.actions"] = np.stack([torch.stack(sn["actions"], 0).numpy() for sn in snaps])   # [steps,T-1,B,K]
                    res[f"{tag}.logp"] = np.stack([torch.stack(sn["logprobs"], 0).numpy() for sn in snaps])
                    res[f"{tag}.states"] = np.stack([np.stack([_summ(x) for x in sn["states"]]) for sn in snaps])
                    for st, sn in enumerate(snaps):
                        sl = list(range(st * B, (st + 1) * B))
                        for t in range(Tn - 1):
                            a_t = sn["actions"][t].numpy()
                            _, ids = S.get_feats([feats[s] for s in sl], [cls[s] for s in sl], a_t, fs)
                            res[f"{tag}.ids.{st}.{t + 1}"] = np.array([i + [-1] * (fs - len(i)) for i in ids], dtype=np.int32)
                            margin = min(margin, min(window_margin(Ns[s], cls[s], a_t[b], fs) for b, s in enumerate(sl)))
                if lr_on:
                    if stage == 2:
                        for k, v in ppo.policy.state_dict().items():
                            res[f"{tag}.policy.{k}"] = _summ(v)
                            res[f"{tag}.policy_delta.{k}"] = _summ(v - pp[k])
                        assert all(torch.equal(v, mp[k]) for k, v in model.module.state_dict().items())   # aggregator frozen
                    else:
                        for k, v in model.module.state_dict().items():
                            res[f"{tag}.model.{k}"] = _summ(v)
                            res[f"{tag}.model_delta.{k}"] = _summ(v - mp[k])
                        for k, v in fc.state_dict().items():
                            res[f"{tag}.fc.{k}"] = _summ(v)
                            res[f"{tag}.fc_delta.{k}"] = _summ(v - fp[k])
                else:
                    assert all(torch.equal(v, mp[k]) for k, v in model.module.state_dict().items())
                    assert ppo is None or all(torch.equal(v, pp[k]) for k, v in ppo.policy.state_dict().items())
    assert margin > 1e-3, f"golden window margin too small ({margin}): pick another seed"
    res["window_margin"] = np.float64(margin)
    np.savez(os.path.join(OUT, "g15_supervised_steps.npz"), **res)


def g16_abmil_general():
    """The reference's ABMIL (models/abmil.py:8-45) outside the launch scripts' defaults: training-mode Dropout(0.25) after
    encoder layers 1 and 2 (torch.nn.functional.dropout replaced by the injected keep masks, in call order), and L / D other
    than 512 / 128 (train_RLMIL.py:91-97 passes --L / --D).  Outputs, attention weights and parameter-gradient fingerprints
    for loss = out.sum()."""
    from oracle.recipes import G16, g16_inputs
    res = {}
    for case, k in G16["cases"].items():
        p, x, masks = g16_inputs(case)
        m = r_abmil.ABMIL(k["d"], L=k["L"], D=k["D"], dim_out=2, dropout=0.25 if masks is not None else 0.0)
        m.load_state_dict(P.to_torch(p))
        m.train()
        calls = []

        def fake_dropout(inp, p_=0.5, training=True, inplace=False):
            if masks is None or not training or p_ == 0.0:
                return inp
            b = len(calls) // 2                                  # bag b: the batch loop runs the encoder once per bag (abmil.py:47-51)
            mk = T(masks[len(calls) % 2][b])
            calls.append(1)
            return inp * mk

        with mock.patch("torch.nn.functional.dropout", fake_dropout):
            out, _ = m(T(x))
            out.sum().backward()
        assert masks is None or len(calls) == 2 * x.shape[0]
        res[f"{case}.out"] = out.detach().numpy()
        with torch.no_grad():
            A = []
            for b in range(x.shape[0]):
                h = T(x[b])
                for i, lin in enumerate((m.encoder[0], m.encoder[3], m.encoder[6])):
                    h = torch.relu(lin(h))
                    if masks is not None and i < 2:
                        h = h * T(masks[i][b])
                a = torch.softmax(m.attention(h).t(), dim=1)
                A.append(a / np.sqrt(a.shape[-1]))
        res[f"{case}.A"] = torch.cat(A).numpy()
        for name, v in m.named_parameters():
            if v.grad is not None:
                res[f"{case}.grad.{name}"] = _summ(v.grad)
    np.savez(os.path.join(OUT, "g16_abmil_general.npz"), **res)


def _grad_entry(g):
    """Small gradients in full (float32), the 512 x 512-sized ones as fingerprints (G20 holds a full set for ABMIL)."""
    return g.numpy().astype(np.float32) if g.numel() <= 70000 else _summ(g)


def g18_clam_big():
    """CLAM_SB(size_arg="big") - the attention net's hidden width 384 instead of 256 (clam.py:66-67; --size_arg big in both entry
    scripts): raw scores, soft-max, top-k ids, pooled M, the instance branch for both labels and the parameter gradients (small tensors in full) of
    the combined objective, eval mode."""
    seed, B, N, d = 18, 3, 300, 512
    x = T(P.bags(seed, "g18.x", B, N, d))
    m = r_clam.CLAM_SB(gate=True, size_arg="big", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=d)
    m.load_state_dict(P.to_torch(P.clam_sb(seed, size=(512, 384))))
    m.eval()
    res = {}
    with torch.no_grad():
        raw = torch.cat([m.bag_forward(x[b], attention_only=True) for b in range(B)])
        A = torch.softmax(raw, 1)
        for b in range(B):
            assert _topk_margin(A[b], 8) > 1e-4, "golden top-k margin too small"
        res["raw"], res["A"] = raw.numpy(), A.numpy()
        res["top_p"], res["top_n"] = torch.topk(A, 8)[1].numpy(), torch.topk(-A, 8)[1].numpy()
        res["M_batch"] = m(x)[0].numpy()
    for label in (0, 1):
        losses, preds, tgts = [], [], []
        with torch.no_grad():
            for b in range(B):
                _, rdb = m.bag_forward(x[b], label=torch.tensor([label]), instance_eval=True)
                losses.append(float(rdb["instance_loss"]))
                preds.append(rdb["inst_preds"])
                tgts.append(rdb["inst_labels"])
        res[f"l{label}.inst_loss"], res[f"l{label}.preds"], res[f"l{label}.targets"] = np.array(losses), np.stack(preds), np.stack(tgts)
    tot = 0
    for b in range(B):
        Mb, rdb = m.bag_forward(x[b], label=torch.tensor([1]), instance_eval=True)
        tot = tot + Mb.sum() + rdb["instance_loss"]
    tot.backward()
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = _grad_entry(v.grad)
    np.savez_compressed(os.path.join(OUT, "g18_clam_big.npz"), **res)


def g19_abmil_heads():
    """ABMIL(K=3): three attention heads (abmil.py:8,23-27,38-44) - out [B*K, L] (the batch loop concatenates the [K, L] blocks),
    the attention weights [B, K, N] and the parameter gradients (small tensors in full) of a weighted sum of the outputs."""
    seed, B, N, d, K = 19, 3, 200, 512, 3
    m = r_abmil.ABMIL(d, L=512, D=128, K=K, dim_out=2)
    m.load_state_dict(P.to_torch(P.abmil(seed, K=K, dim_out=2)))
    x = T(P.bags(seed, "g19.x", B, N, d))
    out, _ = m(x)
    w = T(detrand.normal(seed, "g19.w", (B * K, 512)))
    (out * w).sum().backward()
    res = {"out": out.detach().numpy()}
    with torch.no_grad():
        A = []
        for b in range(B):
            a = torch.softmax(m.attention(m.encoder(x[b])).t(), dim=1)
            A.append(a / np.sqrt(a.shape[-1]))
        res["A"] = torch.stack(A).numpy()
        res["out_single"] = m(x[:1])[0].numpy()                       # the x.shape[0] == 1 branch: [K, L]
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = _grad_entry(v.grad)
    np.savez_compressed(os.path.join(OUT, "g19_abmil_heads.npz"), **res)


def g20_abmil_full_grads():
    """G1 again with every parameter gradient stored IN FULL (the 34-number fingerprints of g1_abmil.npz pin norm, maximum
    and the first values only): the reference's ABMIL at the C1 shape, loss = out.sum()."""
    seed, B, N, d = 985, 4, 256, 512
    m = r_abmil.ABMIL(d, L=512, D=128, dim_out=128)
    m.load_state_dict(P.to_torch(P.abmil(seed)))
    out, _ = m(T(P.bags(seed, "g1.x", B, N, d)))
    out.sum().backward()
    res = {"out": out.detach().numpy()}
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = v.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g20_abmil_full_grads.npz"), **res)


def g21_full_layer_cascade():
    """Full_layer(fc_rnn=False), the cascaded head (rlmil.py:201-206,222-239), driven the way the training loops drive the head - both
    views through ONE module, restart at patch step 0 (train_MuRCL.py:243,272): view 0 and view 1 of step 0 each restart (None, None),
    then the shared concatenation grows by one block per call: fc_2 (step 1 view 0), fc_3 (step 1 view 1), fc_4, fc_5.  Outputs,
    the gradients of all four classifiers (in full) and of the inputs for a weighted sum of the logits."""
    seed, B, F_, C = 21, 4, 512, 16
    fc = r_rlmil.Full_layer(F_, 1024, False, C)
    fc.load_state_dict(P.to_torch(P.full_layer_cascade(seed, F_, C)))
    res, xs, loss = {}, {}, 0.0
    for t in range(3):
        for v in range(2):
            x = T(detrand.normal(seed, f"g21.x.{t}.{v}", (B, F_))).requires_grad_()
            xs[(t, v)] = x
            z = fc(x, restart=(t == 0))
            res[f"none.{t}.{v}"] = np.array(z is None)
            res[f"width.{t}.{v}"] = np.array(fc.hidden.shape[1])
            if z is not None:
                res[f"z.{t}.{v}"] = z.detach().numpy()
                loss = loss + (z * T(detrand.normal(seed, f"g21.w.{t}.{v}", (B, C)))).sum()
    loss.backward()
    for k, p_ in fc.named_parameters():
        res["grad." + k] = p_.grad.numpy().astype(np.float32)
    for (t, v), x in xs.items():
        res[f"dx.{t}.{v}"] = (x.grad if x.grad is not None else torch.zeros_like(x)).numpy()
    np.savez_compressed(os.path.join(OUT, "g21_full_layer_cascade.npz"), **res)


def instance_losses():
    """The non-default instance losses of G22 (shared with the tests: the SAME callables go to the reference and to murcl_amd)."""
    return {"ce_weighted_sum": torch.nn.CrossEntropyLoss(weight=torch.tensor([0.7, 1.3]), reduction="sum"),
            "multi_margin": torch.nn.MultiMarginLoss(),
            "lambda_logit_gap": lambda lg, tg: ((lg[:, 1] - lg[:, 0]) * (1.0 - 2.0 * tg.float())).exp().mean()}


def g22_clam_custom_instance_loss():
    """CLAM_SB(instance_loss_fn=<callable>) - clam.py:64-65,118,131 call whatever loss the constructor was given: G4's inputs through
    three non-default losses (weighted / summed CE, multi-margin, a plain lambda), subtyping on (both inst_eval and inst_eval_out
    run), labels 0 and 1: per-bag instance losses and the parameter gradients (small ones in full) of bag + instance objective."""
    seed, B, N, d = 11, 3, 300, 512
    x = T(P.bags(seed, "g4.x", B, N, d))
    res = {}
    for name, fn in instance_losses().items():
        for label in (0, 1):
            m = r_clam.CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, instance_loss_fn=fn,
                               subtyping=True, in_dim=d)
            m.load_state_dict(P.to_torch(P.clam_sb(seed)), strict=False)      # (a weighted CE module adds its own "instance_loss_fn.weight" key)
            m.eval()
            tot, losses = 0, []
            for b in range(B):
                Mb, rdb = m.bag_forward(x[b], label=torch.tensor([label]), instance_eval=True)
                losses.append(float(rdb["instance_loss"]))
                tot = tot + Mb.sum() + rdb["instance_loss"]
            tot.backward()
            res[f"{name}.l{label}.inst_loss"] = np.array(losses)
            for k, v in m.named_parameters():
                if v.grad is not None:
                    res[f"{name}.l{label}.grad.{k}"] = _grad_entry(v.grad)
    np.savez_compressed(os.path.join(OUT, "g22_clam_custom_instance_loss.npz"), **res)


def dsmil_keep_mask(seed, B, N, d, p_drop):
    """The injected keep multiplier of G23 (0 where dropped, 1/(1-p) where kept), as a [B,N,d] float32 array (shared with the tests)."""
    return ((detrand.uniform(seed, "g23.keep", (B, N, d)) >= p_drop).astype(np.float32) / np.float32(1.0 - p_drop)).astype(np.float32)


def g23_dsmil_dropout_v():
    """BClassifier(dropout_v=0.25) in TRAINING mode (dsmil.py:53-59,66: ``v = Sequential(Dropout(dropout_v), Linear)``; build_dsmil never
    sets it).  torch's dropout draw cannot be reproduced elsewhere, so the Dropout module in front of v's Linear is swapped for one that
    multiplies by an injected keep mask - everything else is the reference's own forward: instance scores, critical instances,
    attention from the UN-dropped features, bag = A^T V with V from the dropped ones; plus the gradients of a weighted objective."""
    seed, B, N, d, C, p_drop = 23, 3, 200, 512, 2, 0.25
    fcl = r_dsmil.FCLayer(d, C)
    bcl = r_dsmil.BClassifier(input_size=d, output_class=C, dropout_v=p_drop)
    m = r_dsmil.MILNet(fcl, bcl)
    m.load_state_dict(P.to_torch(P.dsmil(seed, d, C)))
    m.train()
    keep = T(dsmil_keep_mask(seed, B, N, d, p_drop))

    class Injected(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.b = 0

        def forward(self, feats):                               # called once per bag, in batch order (dsmil.py:85-88)
            out = feats * keep[self.b]
            self.b += 1
            return out
    bcl.v[0] = Injected()
    x = T(P.bags(seed, "g23.x", B, N, d))
    classes, bag, _ = m(x)
    res = {"classes": torch.stack(classes).detach().numpy(), "bag": bag.detach().numpy()}
    wb_, wc_ = T(detrand.normal(seed, "g23.wb", (B, C, d))), T(detrand.normal(seed, "g23.wc", (B, N, C)))
    ((bag * wb_).sum() + (torch.stack(classes) * wc_).sum()).backward()
    for k, v in m.named_parameters():
        if v.grad is not None:
            res["grad." + k] = _grad_entry(v.grad)
    np.savez_compressed(os.path.join(OUT, "g23_dsmil_dropout_v.npz"), **res)


def g17_rl_two_steps():
    """G12 with a second optimizer step (Adam's first step is sign-like, a second one makes the comparison bite):
    TWO consecutive batches through the reference's own ``train()`` (train_MuRCL.py:189-343) at train_stage 2 and 3, T = 3,
    every draw injected.  Stored per stage: per-step losses, rewards, sampler actions / log-probs and selected patch ids of
    BOTH batches, and the parameter fingerprints after the second update."""
    import argparse
    import tempfile
    from oracle import select_oracle as S
    r_tm = _ref_train_murcl()
    c = G12
    seed, B, K, fs, std, Tn = c["seed"], c["B"], c["K"], c["fs"], c["std"], 3
    MVN = torch.distributions.multivariate_normal.MultivariateNormal
    batches = [g12_inputs(Tn, b) for b in range(2)]
    feats = batches[0][1] + batches[1][1]
    cls = batches[0][2] + batches[1][2]
    res, margin = {}, 1.0
    for stage in (2, 3):
        tag = f"s{stage}"
        enc = r_abmil.ABMIL(c["d"], L=512, D=128, dim_out=128)
        enc.load_state_dict(P.to_torch(P.abmil(seed)))
        model = torch.nn.DataParallel(r_cl.CL(enc, projection_dim=128, n_features=512))
        fc = r_rlmil.Full_layer(512, 1024, True, 128)
        fc.load_state_dict(P.to_torch(P.full_layer(seed)))
        ppo = r_rlmil.PPO(c["d"], 512, 512, False, action_std=std, lr=c["ppo_lr"], gamma=c["gamma"], K_epochs=c["K_epochs"], action_size=K)
        sd = P.to_torch(P.actor_critic(seed, 512, 512, K))
        ppo.policy.load_state_dict(sd)
        ppo.policy_old.load_state_dict(sd)
        optimizer = None if stage == 2 else torch.optim.Adam(
            [{"params": model.parameters(), "lr": c["lr"]}, {"params": fc.parameters(), "lr": c["lr"]}],
            betas=(0.9, 0.999), weight_decay=c["wd"])
        args = argparse.Namespace(T=Tn, device="cpu", num_clusters=K, feat_size=fs, train_stage=stage, batch_size=B,
                                  epochs=1, num_data=2 * B, eval_step=2, alpha=c["alpha"], patience=None, warmup=0)

        class _Set:
            def shuffle(self):
                pass

            def __len__(self):
                return 2 * B

            def __getitem__(self, i):
                return T(feats[i]), cls[i], 0, f"case{i}"

        injs = [b[3] for b in batches]
        act_draws = iter([T(a) for inj in injs for a in inj["actions"][0]])
        u_draws = iter([T(inj["u"][t][v]) for inj in injs for t in range(Tn) for v in range(2)])
        perm_draws = iter([T(inj["perm"][t][v]) for inj in injs for t in range(Tn) for v in range(2)])
        eps_draws = iter([T(inj["eps"][t][v]) for inj in injs for t in range(Tn - 1) for v in range(2)])

        def fake_rand(*a, **k):
            size = tuple(k["size"]) if "size" in k else (tuple(a[0]) if isinstance(a[0], (tuple, list)) else tuple(a))
            return next(act_draws) if size == (B, K) else next(u_draws)

        snaps, step_losses = [], []
        inner = r_losses.NT_Xent(B, 1.0)

        class _Crit(torch.nn.Module):
            def forward(self, zi, zj):
                l = inner(zi, zj)
                step_losses.append(l.item())
                return l

        def snap_clear(self):
            snaps.append({f: [x.detach().clone() for x in getattr(self, f)] for f in ("actions", "states", "logprobs", "rewards")})
            for f in ("actions", "states", "logprobs", "rewards", "is_terminals", "hidden"):
                del getattr(self, f)[:]

        real_to = torch.Tensor.to

        def to_ignoring_ordinals(self, *a, **k):
            return self if (a and isinstance(a[0], int)) else real_to(self, *a, **k)

        with tempfile.TemporaryDirectory() as tmp, \
                mock.patch("torch.rand", fake_rand), mock.patch("torch.randperm", lambda *a, **k: next(perm_draws)), \
                mock.patch.object(MVN, "sample", lambda self, *a, **k: self.loc + std * next(eps_draws)), \
                mock.patch.object(r_rlmil.Memory, "clear_memory", snap_clear), \
                mock.patch.object(torch.Tensor, "to", to_ignoring_ordinals):
            r_tm.train(args, _Set(), model, fc, ppo, _Crit(), optimizer, None, None, tmp)
        assert len(snaps) == 4 and len(step_losses) == 2 * Tn
        for it in (act_draws, u_draws, perm_draws, eps_draws):
            assert next(it, None) is None, "an injected draw was not consumed"
        res[f"{tag}.losses"] = np.array(step_losses).reshape(2, Tn)
        res[f"{tag}.rewards"] = np.stack([torch.cat(snaps[2 * b]["rewards"], 0).numpy() for b in range(2)])       # [2, T-1, B]
        for b in range(2):
            Ns_b, feats_b, cls_b = batches[b][0], batches[b][1], batches[b][2]
            for v in range(2):
                sn = snaps[2 * b + v]
                res[f"{tag}.actions.{b}.{v}"] = torch.stack(sn["actions"], 0).numpy()
                res[f"{tag}.logp.{b}.{v}"] = torch.stack(sn["logprobs"], 0).numpy()
                for t in range(Tn - 1):
                    a_t = sn["actions"][t].numpy()
                    _, ids = S.get_feats(feats_b, cls_b, a_t, fs)
                    res[f"{tag}.ids.{b}.{t + 1}.{v}"] = np.array([i + [-1] * (fs - len(i)) for i in ids], dtype=np.int32)
                    margin = min(margin, min(_window_margin(Ns_b[x], cls_b[x], a_t[x], fs) for x in range(B)))
        pre = P.to_torch(P.actor_critic(seed, 512, 512, K))
        for k, v in ppo.policy.state_dict().items():
            res[f"{tag}.policy.{k}"] = _summ(v)
            res[f"{tag}.policy_delta.{k}"] = _summ(v - pre[k])
        if stage == 3:
            pre_m, pre_f = P.to_torch(P.abmil(seed)), P.to_torch(P.full_layer(seed))
            for k, v in model.module.encoder.state_dict().items():
                res[f"{tag}.model.{k}"] = _summ(v)
                res[f"{tag}.model_delta.{k}"] = _summ(v - pre_m[k])
            for k, v in fc.state_dict().items():
                res[f"{tag}.fc.{k}"] = _summ(v)
                res[f"{tag}.fc_delta.{k}"] = _summ(v - pre_f[k])
    assert margin > 1e-3, f"golden window margin too small ({margin})"
    res["window_margin"] = np.float64(margin)
    np.savez(os.path.join(OUT, "g17_rl_two_steps.npz"), **res)


def _reference_parser(mod, globals_needed):
    """The ArgumentParser that the reference's ``main()`` builds (it is local to main): run main() with parse_args
    replaced by a hook that keeps the parser and stops."""
    import argparse
    for k, v in globals_needed.items():
        setattr(mod, k, v)
    box = {}

    class _Stop(Exception):
        pass

    def grab(self, *a, **k):
        box["p"] = self
        raise _Stop

    with mock.patch.object(argparse.ArgumentParser, "parse_args", grab):
        try:
            mod.main()
        except _Stop:
            pass
    return box["p"]


def g13_cli_flags():
    """Drop-in CLI: every flag of the reference's two entry scripts - option strings, dest, type, default,
    choices, action - and the namespaces its parsers produce for the invocations of runs/*.sh."""
    import json
    from oracle.recipes import run_script_argvs
    r_tm = _ref_train_murcl()
    r_tr, _ = _ref_train_rlmil()
    parsers = {"train_MuRCL": _reference_parser(r_tm, {"MODELS": ["ABMIL", "CLAM_SB"]}),
               "train_RLMIL": _reference_parser(r_tr, {"MODELS": ["ABMIL", "CLAM_SB", "DSMIL"], "LOSSES": ["CrossEntropyLoss"]})}
    man = {"flags": {}, "runs": {}}
    for name, p in parsers.items():
        rows = []
        for a in p._actions:
            if a.dest == "help":
                continue
            rows.append({"options": list(a.option_strings), "dest": a.dest, "type": getattr(a.type, "__name__", None),
                         "default": a.default, "choices": list(a.choices) if a.choices is not None else None,
                         "action": type(a).__name__, "nargs": a.nargs})
        man["flags"][name] = rows
    for script, (entry, argvs) in run_script_argvs().items():
        man["runs"][script] = {"entry": entry, "namespaces": [vars(parsers[entry].parse_args(av)) for av in argvs]}
    with open(os.path.join(OUT, "g13_cli_flags.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)


def _exchange_checkpoints(mods):
    """Real files across the boundary: the product's writer -> the reference's modules (strict), the reference's state
    dicts -> the product's loader, and the reference's fine-tune key loop (train_RLMIL.py:121-130) vs the product's."""
    import tempfile
    from murcl_amd.models import abmil as p_abmil, cl as p_cl, rlmil as p_rlmil
    from murcl_amd.utils import checkpoint as C
    torch.manual_seed(4)
    ours = (p_cl.CL(p_abmil.ABMIL(512, L=512, D=128, dim_out=128), 128, 512), p_rlmil.Full_layer(512, 1024, True, 128),
            p_rlmil.ActorCritic(512, 512, 512, False, 0.5, 10))

    class _P:
        policy = ours[2]
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        path = C.save_checkpoint(C.make_state(3, ours[0], ours[1], ppo=_P), False, tmp)
        ck = torch.load(path, map_location="cpu")
        theirs = (mods["CL(ABMIL)"], r_rlmil.Full_layer(512, 1024, True, 128), mods["ActorCritic"])
        theirs[0].load_state_dict(ck["model_state_dict"])            # strict
        theirs[1].load_state_dict(ck["fc"])
        theirs[2].load_state_dict(ck["policy"])
        res["reference_loaded_product_file_strict"] = bool(
            torch.equal(theirs[0].encoder.attention[0].weight, ours[0].encoder.attention[0].weight)
            and torch.equal(theirs[2].gru.weight_hh_l0, ours[2].gru.weight_hh_l0))
        sd = dict(ck["model_state_dict"])
        for k in list(sd.keys()):                                    # the reference's loop, verbatim semantics
            if k.startswith("encoder") and not k.startswith("encoder.fc") and not k.startswith("encoder.classifiers"):
                sd[k[len("encoder."):]] = sd[k]
            del sd[k]
        mine = C.strip_pretrained_encoder(ck["model_state_dict"])
        res["finetune_strip_identical"] = list(sd) == list(mine) and all(torch.equal(sd[k], mine[k]) for k in sd)
        torch.manual_seed(5)
        fresh = (r_cl.CL(r_abmil.ABMIL(512, L=512, D=128, dim_out=128), 128, 512), r_rlmil.Full_layer(512, 1024, True, 128),
                 r_rlmil.ActorCritic(512, 512, 512, False, 0.5, 10))
        torch.save({"epoch": 1, "model_state_dict": fresh[0].state_dict(), "fc": fresh[1].state_dict(), "optimizer": None,
                    "ppo_optimizer": None, "policy": fresh[2].state_dict()}, os.path.join(tmp, "theirs.pth.tar"))
        m, h = p_cl.CL(p_abmil.ABMIL(512, L=512, D=128, dim_out=128), 128, 512), p_rlmil.Full_layer(512, 1024, True, 128)

        class _Q:
            policy, policy_old = p_rlmil.ActorCritic(512, 512, 512, False, 0.5, 10), p_rlmil.ActorCritic(512, 512, 512, False, 0.5, 10)
        C.load_stage(m, h, _Q, os.path.join(tmp, "theirs.pth.tar"))
        res["product_loaded_reference_file_strict"] = bool(
            torch.equal(m.encoder.decoder[0].bias, fresh[0].encoder.decoder[0].bias)
            and torch.equal(_Q.policy_old.critic[0].weight, fresh[2].critic[0].weight))
    return res


def g11_manifest():
    """8(f) rank 3: names and shapes of every state-dict entry of the reference modules, and its checkpoint dict keys."""
    import json
    mods = {
        "ABMIL": r_abmil.ABMIL(512, L=512, D=128, dim_out=2),
        "CLAM_SB": r_clam.CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512),
        "DSMIL": r_dsmil.build_dsmil(512, 2),
        "CL(ABMIL)": r_cl.CL(r_abmil.ABMIL(512, L=512, D=128, dim_out=128), projection_dim=128, n_features=512),
        "Full_layer": r_rlmil.Full_layer(512, 1024, True, 2),
        "ActorCritic": r_rlmil.ActorCritic(512, 512, 512, False, 0.5, 10),
    }
    man = {k: [[n, list(v.shape)] for n, v in m.state_dict().items()] for k, m in mods.items()}
    man["checkpoint_keys"] = ["epoch", "model_state_dict", "fc", "optimizer", "ppo_optimizer", "policy"]   # train_MuRCL.py:322-329
    man["exchange"] = _exchange_checkpoints(mods)
    with open(os.path.join(OUT, "g11_state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for fn in (g1_abmil, g2_ntxent, g3_pretrain, g4_clam, g5_dsmil, g6_get_feats, g7_mixup, g8_ppo, g9_full_layer, g10_eval, g11_manifest, g12_rl_step, g13_cli_flags, g14_clam_plain, g15_supervised_steps, g16_abmil_general, g17_rl_two_steps,
               g18_clam_big, g19_abmil_heads, g20_abmil_full_grads, g21_full_layer_cascade, g22_clam_custom_instance_loss, g23_dsmil_dropout_v):
        if not only or fn.__name__ in only:
            fn()
            print("wrote", fn.__name__)
