"""The full MuRCL hot step (train_MuRCL.py:233-304) on MI355X vs the oracle composition
get_feats -> mixup -> CL(ABMIL) -> Full_layer -> NT-Xent over T patch-steps, with every random draw injected."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, mil_oracle as O, params as P, select_oracle as S  # noqa: E402

T = torch.from_numpy


def _args(**kw):
    from murcl_amd.train_MuRCL import build_parser
    a = build_parser().parse_args(["--arch", "ABMIL"])             # (the reference's default arch is CLAM_SB)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_stage1_step_matches_oracle_and_updates_parameters():
    from murcl_amd.train_MuRCL import create_model, get_optimizer, pretrain_step
    from murcl_amd.utils.datasets import BagPack
    from murcl_amd.utils.losses import NT_Xent
    from murcl_amd.models import rlmil
    dev = torch.device("cuda:0")
    B, N, K, fs, Tn = 4, 700, 10, 256, 3
    args = _args(T=Tn, feat_size=fs, batch_size=B, dtype="f32", train_stage=1, num_clusters=K, backbone_lr=1e-3, fc_lr=1e-3)
    model, fc, ppo = create_model(args, 512, dev)
    model.encoder.load_state_dict(P.to_torch(P.abmil(985)))
    fc.load_state_dict(P.to_torch(P.full_layer(985)))
    opt = get_optimizer(args, model, fc)
    feats_np = [P.bags(51, f"f{b}", 1, N + 13 * b, 512)[0] for b in range(B)]
    cls = [P.cluster_lists(51, f"c{b}", N + 13 * b, K) for b in range(B)]
    pack = BagPack.from_lists([T(f).to(dev) for f in feats_np], cls)
    inj = {"actions": [[T(detrand.uniform(51, f"a{t}{v}", (B, K))) for v in range(2)] for t in range(Tn)],
           "draws": [[(T(detrand.uniform(51, f"l{t}{v}", (B, 1), 0.9, 1.0)).to(dev), T(detrand.permutation(51, f"p{t}{v}", B)).to(dev))
                      for v in range(2)] for t in range(Tn)]}
    w_before = model.encoder.encoder[0].weight.detach().clone()
    loss, losses, rewards = pretrain_step(args, model, fc, ppo, NT_Xent(B, 1.0), opt, pack, [rlmil.Memory(), rlmil.Memory()], injected=inj)
    # oracle
    views = []
    for t in range(Tn):
        vt = []
        for v in range(2):
            sub, _ = S.get_feats(feats_np, cls, inj["actions"][t][v].numpy(), fs)
            lam, perm = inj["draws"][t][v]
            vt.append(T(S.mixup(sub, lam.cpu().numpy(), perm.cpu().numpy())))
        views.append(vt)
    mp = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.abmil(985)).items()}
    fp = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.full_layer(985)).items()}
    ref, ref_losses, ref_rewards, _ = O.pretrain_step(mp, fp, views, 1.0)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-4)
    np.testing.assert_allclose([l.item() for l in losses], [l.item() for l in ref_losses], rtol=1e-4)
    np.testing.assert_allclose(torch.cat(rewards).cpu().numpy(), torch.stack(ref_rewards).numpy(), rtol=5e-3, atol=5e-6)
    # one Adam step happened and matches the oracle's Adam on the oracle's gradient
    ref.backward()
    want = O.adam_step({"w": mp["encoder.0.weight"].detach()}, {"w": mp["encoder.0.weight"].grad}, {}, 1e-3, weight_decay=1e-5)["w"]
    got = model.encoder.encoder[0].weight.detach().cpu()
    assert not torch.equal(got, w_before.cpu())
    assert ((got - want).norm() / (want - w_before.cpu()).norm()).item() < 5e-2


def test_stage3_step_runs_with_ppo_sampler(tmp_path):
    """Stage 2/3 wiring: PPO.select_action feeds the sub-bag sampler, rewards reach both memories, PPO.update runs."""
    from murcl_amd.train_MuRCL import create_model, get_optimizer, pretrain_step
    from murcl_amd.utils.datasets import BagPack
    from murcl_amd.utils.losses import NT_Xent
    from murcl_amd.models import rlmil
    dev = torch.device("cuda:0")
    B, N, K, fs = 4, 600, 10, 128
    a1 = _args(T=3, feat_size=fs, batch_size=B, dtype="f32", train_stage=1, num_clusters=K, save_dir=str(tmp_path / "stage_1"))
    model, fc, _ = create_model(a1, 512, dev)
    (tmp_path / "stage_1").mkdir()
    torch.save({"model_state_dict": model.state_dict(), "fc": fc.state_dict()}, tmp_path / "stage_1" / "model_best.pth.tar")
    feats = [T(P.bags(52, f"f{b}", 1, N, 512)[0]).to(dev) for b in range(B)]
    pack = BagPack.from_lists(feats, [P.cluster_lists(52, f"c{b}", N, K) for b in range(B)])
    for stage in (2, 3):
        a = _args(T=3, feat_size=fs, batch_size=B, dtype="f32", train_stage=stage, num_clusters=K, K_epochs=2,
                  save_dir=str(tmp_path / f"stage_{stage}"), ppo_lr=1e-3)
        if stage == 3:
            (tmp_path / "stage_2").mkdir()
            torch.save({"model_state_dict": model.state_dict(), "fc": fc.state_dict(), "policy": ppo.policy.state_dict()},
                       tmp_path / "stage_2" / "model_best.pth.tar")
        m, f, ppo = create_model(a, 512, dev)
        opt = get_optimizer(a, m, f)
        pol_before = ppo.policy.actor[0].weight.detach().clone()
        loss, losses, rewards = pretrain_step(a, m, f, ppo, NT_Xent(B, 1.0), opt, pack, [rlmil.Memory(), rlmil.Memory()])
        assert torch.isfinite(loss) and len(losses) == 3 and len(rewards) == 2 and rewards[0].shape == (1, B)
        if stage == 2:
            assert opt is None and not torch.equal(ppo.policy.actor[0].weight, pol_before)      # PPO.update moved the policy
            assert all(torch.equal(x, y) for x, y in zip(ppo.policy.parameters(), ppo.policy_old.parameters()))


@pytest.mark.parametrize("arch", ["ABMIL", "CLAM_SB", "DSMIL"])
def test_supervised_rlmil_step_first_loss_matches_oracle_and_trains(arch):
    """a21: the supervised step body for each aggregator; t=0 loss vs the oracle, then a few steps reduce the loss."""
    from murcl_amd.train_RLMIL import create_model, supervised_step
    from murcl_amd.optim import FlatAdam
    from murcl_amd.models import rlmil
    from murcl_amd.utils.datasets import BagPack
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    B, N, K, fs, C = 4, 500, 10, 128, 2
    model, fc = create_model(arch, 512, C, dev)
    model.eval() if arch == "CLAM_SB" else None                    # dropout off so the oracle comparison is exact
    pk = P.abmil(61, dim_out=C) if arch == "ABMIL" else {"CLAM_SB": P.clam_sb, "DSMIL": P.dsmil}[arch](61)
    model.load_state_dict(P.to_torch(pk))
    fcp = P.full_layer(61, 512, 1024, C)
    fc.load_state_dict(P.to_torch(fcp))
    feats_np = [P.bags(61, f"f{b}", 1, N, 512)[0] for b in range(B)]
    cls = [P.cluster_lists(61, f"c{b}", N, K) for b in range(B)]
    pack = BagPack.from_lists([T(f).to(dev) for f in feats_np], cls)
    labels = torch.tensor([0, 1, 1, 0], device=dev)
    acts = [T(detrand.uniform(61, f"a{t}", (B, K))) for t in range(2)]
    opt = FlatAdam([{"params": list(model.parameters()) + list(fc.parameters()), "lr": 1e-3}])
    loss0, losses, rewards = supervised_step(arch, model, fc, None, opt, pack, labels, rlmil.Memory(), T=2, feat_size=fs, actions=acts)
    # oracle for the t = 0 term
    sub, _ = S.get_feats(feats_np, cls, acts[0].numpy(), fs)
    p, fp, x, y = P.to_torch(pk), P.to_torch(fcp), T(sub), labels.cpu()
    if arch == "ABMIL":
        out = O.abmil_forward(p, x)[0]
        ref = F.cross_entropy(O.full_layer_step(fp, out, None)[0], y)
    elif arch == "CLAM_SB":
        M, A, s, h = O.clam_sb_forward(p, x)
        inst = torch.stack([O.clam_instance_eval(p, A[b], h[b], int(y[b]), C, 8, True)[0] for b in range(B)]).mean()
        ref = 0.7 * F.cross_entropy(O.full_layer_step(fp, M, None)[0], y) + 0.3 * inst
    else:
        c, bag, _, _ = O.dsmil_forward(p, x)
        ref = 0.5 * F.cross_entropy(O.full_layer_step(fp, bag.mean(1), None)[0], y) + 0.5 * F.cross_entropy(c.max(1)[0], y)
    np.testing.assert_allclose(losses[0].item(), ref.item(), rtol=2e-4)
    assert rewards[0].shape == (1, B)
    w0 = fc.fc.weight.detach().clone()
    for _ in range(4):
        l, _, _ = supervised_step(arch, model, fc, None, opt, pack, labels, rlmil.Memory(), T=2, feat_size=fs, actions=acts)
    assert torch.isfinite(l) and not torch.equal(fc.fc.weight, w0)
    if arch != "ABMIL":          # (the ABMIL test weights carry a x60 decoder gain: Adam at 1e-3 overshoots there)
        assert l.item() < loss0.item()


@pytest.mark.parametrize("resident", [True, False])
def test_train_script_runs_from_the_resident_store_and_from_per_step_uploads(tmp_path, resident, capsys):
    """train_MuRCL.main on synthetic slides: the HBM-resident split (default) and the reference-style per-step upload
    both train, write the reference's checkpoint keys, and the resident run reports its store."""
    from murcl_amd import train_MuRCL
    save = tmp_path / ("res" if resident else "stream") / "stage_1"
    argv = ["--synthetic", "6,300", "--batch_size", "3", "--feat_size", "64", "--T", "2", "--epochs", "2", "--data_repeat", "2",
            "--num_clusters", "4", "--dtype", "f32", "--arch", "ABMIL", "--device", "0", "--scheduler", "CosineAnnealingLR",
            "--patience", "10", "--save_dir", str(save)] + ([] if resident else ["--no_resident"])
    train_MuRCL.main(argv)
    out = capsys.readouterr().out
    assert ("resident slide store: 6 slides" in out) == resident
    assert out.count("Loss: ") == 2 and "Epoch:" in out
    for f in ("args.yaml", "losses.csv", "results.csv", "checkpoint.pth.tar"):                 # train_MuRCL.py:196-199,330,375
        assert (save / f).exists(), f
    ck = torch.load(save / "model_best.pth.tar", map_location="cpu")
    assert {"epoch", "model_state_dict", "fc", "optimizer", "ppo_optimizer", "policy"} <= set(ck)
    assert ck["optimizer"]["kind"] == "FlatAdam" and ck["optimizer"]["step_count"] >= 4        # Adam state is saved (:326)
    assert "encoder.encoder.0.weight" in ck["model_state_dict"] and "rnn.weight_ih_l0" in ck["fc"]
    assert all(torch.isfinite(v).all() for v in ck["model_state_dict"].values())


def test_stage1_all_patch_steps_at_once_equals_the_step_by_step_loop():
    """Stage 1 runs the aggregator once over the sub-bags of all T patch steps; with the same random seed it draws the same
    windows / mix-up partners as the reference-order loop and lands on the same losses, rewards and updated weights."""
    from murcl_amd.train_MuRCL import create_model, get_optimizer, pretrain_step
    from murcl_amd.utils.datasets import BagPack
    from murcl_amd.utils.losses import NT_Xent
    from murcl_amd.models import rlmil
    dev = torch.device("cuda:0")
    B, N, K, fs, Tn = 4, 600, 10, 128, 4
    feats = [T(P.bags(53, f"f{b}", 1, N + 31 * b, 512)[0]).to(dev) for b in range(B)]
    pack = BagPack.from_lists(feats, [P.cluster_lists(53, f"c{b}", N + 31 * b, K) for b in range(B)])

    def run(stepwise):
        args = _args(T=Tn, feat_size=fs, batch_size=B, dtype="f32", train_stage=1, num_clusters=K, backbone_lr=1e-3, fc_lr=1e-3)
        args.no_batched_stage1 = stepwise
        model, fc, ppo = create_model(args, 512, dev)
        model.encoder.load_state_dict(P.to_torch(P.abmil(985)))
        fc.load_state_dict(P.to_torch(P.full_layer(985)))
        opt = get_optimizer(args, model, fc)
        torch.manual_seed(77)
        loss, losses, rewards = pretrain_step(args, model, fc, ppo, NT_Xent(B, 1.0), opt, pack, [rlmil.Memory(), rlmil.Memory()])
        return loss.item(), [l.item() for l in losses], torch.cat(rewards).cpu(), model.encoder.encoder[3].weight.detach().cpu(), \
            fc.rnn.weight_hh_l0.detach().cpu()

    a, b = run(False), run(True)
    assert a[0] == pytest.approx(b[0], rel=1e-5)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5)
    np.testing.assert_allclose(a[2].numpy(), b[2].numpy(), rtol=1e-3, atol=1e-6)
    for x, y in zip(a[3:], b[3:]):
        # Adam's first step moves every weight by ~lr * sign(g): compare the moves
        assert ((x - y).norm() / (1e-3 * x.numel() ** 0.5)).item() < 2e-2


@pytest.mark.parametrize("arch", ["ABMIL", "CLAM_SB", "DSMIL"])
def test_supervised_stage1_all_patch_steps_at_once_equals_the_loop(arch):
    from murcl_amd.train_RLMIL import create_model, supervised_step
    from murcl_amd.optim import FlatAdam
    from murcl_amd.models import rlmil
    from murcl_amd.utils.datasets import BagPack
    dev = torch.device("cuda:0")
    B, N, K, fs, C, Tn = 4, 400, 10, 64, 2, 3
    pack = BagPack.from_lists([T(P.bags(62, f"f{b}", 1, N + 9 * b, 512)[0]).to(dev) for b in range(B)],
                              [P.cluster_lists(62, f"c{b}", N + 9 * b, K) for b in range(B)])
    labels = torch.tensor([1, 0, 1, 0], device=dev)
    acts = [T(detrand.uniform(62, f"a{t}", (B, K))) for t in range(Tn)]

    def run(at_once):
        model, fc = create_model(arch, 512, C, dev)
        model.eval() if arch == "CLAM_SB" else None                 # dropout draws differ between one call and three
        pk = P.abmil(62, dim_out=C) if arch == "ABMIL" else {"CLAM_SB": P.clam_sb, "DSMIL": P.dsmil}[arch](62)
        model.load_state_dict(P.to_torch(pk))
        fc.load_state_dict(P.to_torch(P.full_layer(62, 512, 1024, C)))
        opt = FlatAdam([{"params": list(model.parameters()) + list(fc.parameters()), "lr": 1e-3}])
        loss, losses, rewards = supervised_step(arch, model, fc, None, opt, pack, labels, rlmil.Memory(), T=Tn, feat_size=fs,
                                                actions=acts, batch_patch_steps=at_once)
        return loss.item(), [l.item() for l in losses], torch.cat(rewards).cpu().numpy(), fc.fc.weight.detach().cpu()

    a, b = run(True), run(False)
    assert a[0] == pytest.approx(b[0], rel=2e-5)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-5)
    np.testing.assert_allclose(a[2], b[2], rtol=2e-3, atol=2e-6)
    assert ((a[3] - b[3]).norm() / (1e-3 * a[3].numel() ** 0.5)).item() < 2e-2


def test_bf16_training_tracks_f32_over_a_step_sequence():
    """per-kernel bf16 tolerances say little about training.  40 Adam steps of the contrastive view-pair step
    (CL(ABMIL) + Full_layer + NT-Xent) on the same bags, same initial weights, once with f32 patch tensors and once with
    bf16 storage: the loss trajectories must stay together (every step within 2 %, final loss within 1 %) and both must
    actually train."""
    from murcl_amd.models.abmil import ABMIL
    from murcl_amd.models.cl import CL
    from murcl_amd.models.rlmil import Full_layer
    from murcl_amd.optim import FlatAdam
    from murcl_amd.utils.losses import NT_Xent
    dev = torch.device("cuda:0")
    B, N = 16, 512
    xs = [T(P.bags(97, f"v{v}", B, N, 512)).to(dev) for v in range(2)]

    def run(dtype):
        enc = ABMIL(512, L=512, D=128, dim_out=128)
        enc.load_state_dict(P.to_torch(P.abmil(985)))
        enc.compute_dtype = dtype
        model = CL(enc, 128, 512).to(dev)
        fc = Full_layer(512, 1024, True, 128)
        fc.load_state_dict(P.to_torch(P.full_layer(985)))
        fc = fc.to(dev)
        opt = FlatAdam([{"params": list(model.parameters()), "lr": 1e-4}, {"params": list(fc.parameters()), "lr": 5e-5}],
                       weight_decay=1e-5)
        crit = NT_Xent(B, 1.0)
        views = [x.to(dtype) for x in xs]
        out = []
        for _ in range(40):
            opt.zero_grad()
            outs, _ = model(views)
            z = fc.forward_views(outs, restart=True)
            loss = crit(z[0], z[1])
            loss.backward()
            opt.step()
            out.append(loss.item())
        return np.array(out)

    f32, bf16 = run(torch.float32), run(torch.bfloat16)
    assert f32[-1] < 0.9 * f32[0] and bf16[-1] < 0.9 * bf16[0], (f32[0], f32[-1], bf16[0], bf16[-1])      # both learn
    np.testing.assert_allclose(bf16, f32, rtol=2e-2)
    assert abs(bf16[-1] - f32[-1]) <= 1e-2 * f32[-1]


@pytest.mark.parametrize("arch", ["CLAM_SB", "ABMIL"])
def test_pretrain_script_three_stages_like_runs_pretrain_sh(tmp_path, arch, capsys):
    """runs/pretrain.sh drives train_MuRCL.py through stages 1, 2 (sampler only, ``--ppo_epochs``) and 3 (joint, halved
    learning rates) with CLAM_SB, a cosine schedule and ``--patience``; each stage picks up ``../stage_{k-1}/model_best.pth.tar``.
    Same sequence here on synthetic slides (the reference's default arch included)."""
    from murcl_amd import train_MuRCL
    base = ["--synthetic", "8,320", "--num_clusters", "4", "--feat_size", "64", "--T", "3", "--batch_size", "4", "--data_repeat", "1",
            "--arch", arch, "--device", "0", "--scheduler", "CosineAnnealingLR", "--patience", "10", "--exist_ok", "--dtype", "f32",
            "--base_save_dir", str(tmp_path), "--dataset", "Synth"]
    for stage, lrs in ((1, ("0.0001", "0.00005")), (2, ("0.0001", "0.00005")), (3, ("0.00005", "0.00001"))):
        train_MuRCL.main(base + ["--train_stage", str(stage), "--epochs", "2", "--ppo_epochs", "2", "--backbone_lr", lrs[0], "--fc_lr", lrs[1]])
    out = capsys.readouterr().out
    assert out.count("Loss: ") == 6
    runs = sorted(p for p in tmp_path.rglob("stage_*") if p.is_dir())
    assert [p.name for p in runs] == ["stage_1", "stage_2", "stage_3"] and len({p.parent for p in runs}) == 1   # one run directory
    assert f"Synth_np_64/MuRCL/T3_pd128_as0.5_pg0.1_tau1.0_alpha0.9/{arch}" in str(runs[0])
    ck = [torch.load(p / "model_best.pth.tar", map_location="cpu") for p in runs]
    assert ck[0]["policy"] is None and ck[1]["policy"] is not None and ck[2]["policy"] is not None
    key = next(iter(ck[0]["model_state_dict"]))
    assert torch.equal(ck[0]["model_state_dict"][key], ck[1]["model_state_dict"][key])         # stage 2 trains the sampler only
    assert not torch.equal(ck[1]["model_state_dict"][key], ck[2]["model_state_dict"][key])     # stage 3 trains the aggregator again
    assert all(torch.equal(ck[1]["policy"][k], ck[2]["policy"][k]) for k in ck[1]["policy"])   # ... and only samples with the policy
    assert all(torch.isfinite(v).all() for c in ck for v in c["model_state_dict"].values())


def _run_pretrain(tmp, sync, extra=()):
    import os
    from murcl_amd import train_MuRCL
    argv = ["--synthetic", "8,320", "--num_clusters", "4", "--feat_size", "64", "--T", "2", "--batch_size", "4", "--data_repeat", "2",
            "--arch", "ABMIL", "--device", "0", "--exist_ok", "--dtype", "f32", "--base_save_dir", str(tmp), "--dataset", "Synth",
            "--train_stage", "1", "--epochs", "5", *extra]
    old = os.environ.get("MURCL_SYNC_EPOCH_END")
    os.environ["MURCL_SYNC_EPOCH_END"] = "1" if sync else "0"
    try:
        train_MuRCL.main(argv)
    finally:
        if old is None:
            os.environ.pop("MURCL_SYNC_EPOCH_END", None)
        else:
            os.environ["MURCL_SYNC_EPOCH_END"] = old
    run = next(p for p in tmp.rglob("stage_1") if p.is_dir())
    return run, torch.load(run / "checkpoint.pth.tar", map_location="cpu"), torch.load(run / "model_best.pth.tar", map_location="cpu")


def test_epoch_boundaries_without_a_queue_drain_write_the_same_run(tmp_path, capsys):
    """EpochSnapshots (state and loss captured in stream order, read back behind the next epoch's steps) against the synchronous
    boundary of the reference loop: same epochs logged, same csv rows, the same parameters in checkpoint.pth.tar / model_best
    (up to the order of float atomics between two runs of the same step sequence)."""
    import csv
    d_run, d_last, d_best = _run_pretrain(tmp_path / "deferred", sync=False)
    s_run, s_last, s_best = _run_pretrain(tmp_path / "sync", sync=True)
    assert capsys.readouterr().out.count("Loss: ") == 10
    assert d_last["epoch"] == s_last["epoch"] == 5 and d_best["epoch"] == s_best["epoch"]
    rows_d = list(csv.reader(open(d_run / "losses.csv")))
    rows_s = list(csv.reader(open(s_run / "losses.csv")))
    assert len(rows_d) == len(rows_s) == 6 and [r[0] for r in rows_d] == [r[0] for r in rows_s]
    for a, b in zip(rows_d[1:], rows_s[1:]):
        assert abs(float(a[1]) - float(b[1])) <= 1e-4 * max(1.0, abs(float(b[1])))
    for key in ("model_state_dict", "fc"):
        for k, v in s_last[key].items():
            torch.testing.assert_close(d_last[key][k], v, rtol=1e-3, atol=1e-4, msg=lambda m: f"{key}.{k}: {m}")
    for g_d, g_s in zip(d_last["optimizer"]["groups"], s_last["optimizer"]["groups"]):
        assert g_d["steps"] == g_s["steps"] and g_d["lr"] == g_s["lr"]
        torch.testing.assert_close(g_d["m"], g_s["m"], rtol=1e-2, atol=1e-5)
    # epoch k's snapshot was taken at the end of epoch k, not when it was read back: it differs from the final state
    first = next(iter(d_last["model_state_dict"]))
    assert d_best["epoch"] == 5 or not torch.equal(d_best["model_state_dict"][first], d_last["model_state_dict"][first])


def test_early_stop_with_deferred_boundaries_discards_the_extra_steps(tmp_path, capsys):
    """--patience (runs/pretrain.sh): the stop decision arrives a few steps into the next epoch; nothing of that epoch is
    logged or saved.  A learning rate of 0 makes the loss constant, so both loops stop after `patience` epochs."""
    import csv
    extra = ("--patience", "2", "--backbone_lr", "0", "--fc_lr", "0", "--epochs", "6")
    d_run, d_last, _ = _run_pretrain(tmp_path / "deferred", sync=False, extra=extra)
    s_run, s_last, _ = _run_pretrain(tmp_path / "sync", sync=True, extra=extra)
    assert d_last["epoch"] == s_last["epoch"] == 2
    assert len(list(csv.reader(open(d_run / "losses.csv")))) == len(list(csv.reader(open(s_run / "losses.csv")))) == 3
    assert capsys.readouterr().out.count("Loss: ") == 4


@pytest.mark.parametrize("G", [1, 5])
def test_hipgraph_replayed_steps_are_the_next_training_steps(G):
    """bench.py's graph region (round 6): G steps captured ONCE with the optimizer in live-graph mode - Adam's step counts advance on
    the device (murcl_adam_multi_live) - and replayed: after the same number of steps the parameters, Adam moments and host-side step
    counts equal those of the step-by-step eager loop up to the run-to-run noise of the loop itself (the attention weight gradient's
    split sums meet through float atomics; measured here as eager vs eager), while a replay with FROZEN step counts (no live mode: the
    captured bias correction again and again) lands clearly outside it."""
    import bench
    dev = torch.device("cuda:0")

    def run(mode):
        model, fc, opt, crit = bench.build(torch.bfloat16, dev, 8)
        views = bench.synth_views(8, 512, 512, torch.bfloat16, dev, 0)
        step = bench.make_step(model, fc, opt, crit, views, 1)
        for _ in range(2):
            step()
        total = 10
        if mode != "eager":
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()                                            # (step 3: per-stream state exists before the capture)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            if mode == "live":
                opt.live_graph(True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(G):
                    step()
            for _ in range(total // G):
                g.replay()
            torch.cuda.synchronize()
            if mode == "live":
                opt.after_replays(total)
                step()                                            # and the eager loop carries on from the right step count
        else:
            for _ in range(total + 2):
                step()
        torch.cuda.synchronize()
        return ([p.detach().float().clone() for grp in opt.groups for p in (grp["p"], grp["m"], grp["v"])], opt.step_count,
                sorted(opt._pstep.values()))

    def dist(x, y, which):
        """largest difference of the parameters (which = 0) / first moments (1) / second moments (2), relative to the tensor's largest entry"""
        return max(((a - b).abs().max() / a.abs().max().clamp_min(1e-30)).item() for a, b in zip(x[which::3], y[which::3]))
    (pe, ce, se), (pe2, _, _), (pg, cg, sg) = run("eager"), run("eager"), run("live")
    assert ce == cg == 14 and se == sg
    for which in range(3):
        noise = dist(pe, pe2, which)
        assert dist(pe, pg, which) <= max(4 * noise, 1e-6), (which, dist(pe, pg, which), noise)
    if G == 1:
        pf = run("frozen")[0]                                     # 10 replays of step 4's bias correction: the wrong steps
        # (measured: noise p / m / v = 6e-4 / 7e-3 / 9e-3 of the largest entry after 14 steps - Adam amplifies the last bits of the
        #  atomically summed attention weight gradient; live replays 6e-4 / 9e-3 / 1.3e-2; frozen replays 2e-3 / 0.11 / 0.08)
        assert dist(pe, pf, 0) > 2 * max(dist(pe, pe2, 0), 1e-6) and dist(pe, pf, 1) > 5 * max(dist(pe, pe2, 1), 1e-6), \
            ([dist(pe, pf, w) for w in range(3)], [dist(pe, pe2, w) for w in range(3)])
