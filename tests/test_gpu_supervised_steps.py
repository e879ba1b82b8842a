"""Row a21 / BASELINE config 5's path on MI355X: ``murcl_amd.train_RLMIL.supervised_step`` against
tests/golden/g15_supervised_steps.npz - batches through the reference's own ``train_ABMIL`` / ``train_CLAM`` / ``train_DSMIL``
(train_RLMIL.py:715-781, 323-392, 508-590) at train_stage 1, 2 and 3 with every draw injected and Dropout off.
Compared: per-patch-step losses (incl. CLAM's bag_weight mix and DSMIL's max-instance term), the confidence rewards, the
sampler's actions / log-probs, the patch ids they select (bit-exact), the last step's logits, and the parameters after TWO
optimizer steps (aggregator + head at stages 1 / 3, the sampler at stage 2) - at the reference scripts' batch size 1, at
batch size 4 for ABMIL (the one body the reference can batch), and the product's batched form of CLAM / DSMIL against the
mean of the reference's four batch-size-1 bodies at frozen parameters."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import params as P  # noqa: E402
from oracle.recipes import G15, G15_RUNS, g15_inputs, g15_params  # noqa: E402

T = torch.from_numpy


def _summ(g):
    g = g.detach().double().flatten().cpu()
    return np.concatenate([[g.norm().item(), g.abs().max().item()], g[:32].numpy()])


def _close_summ(got, want, rtol, msg=""):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * want[1], err_msg=msg)


def _build(arch, stage, lr_on=True):
    from murcl_amd.models import rlmil
    from murcl_amd.optim import FlatAdam
    from murcl_amd.train_RLMIL import create_model
    c, dev = G15, torch.device("cuda:0")
    mp, fp, pp = (P.to_torch(d) for d in g15_params(arch))
    model, fc = create_model(arch, c["d"], c["C"], dev, k_sample=c["k_sample"])
    model.load_state_dict(mp)
    fc.load_state_dict(fp)
    # the golden was generated with Dropout off (its Philox stream cannot be matched); stage 2 is eval mode anyway (:299-301)
    model.eval() if (arch == "CLAM_SB" or stage == 2) else model.train()
    ppo = None
    if stage != 1:
        ppo = rlmil.PPO(c["d"], 512, 512, False, action_std=c["std"], lr=c["ppo_lr"] if lr_on else 0.0, gamma=c["gamma"],
                        K_epochs=c["K_epochs"], action_size=c["K"])
        ppo.policy.load_state_dict(pp)
        ppo.policy_old.load_state_dict(pp)
    opt = None
    if stage != 2:                                                                       # train_RLMIL.py:257-267
        opt = FlatAdam([{"params": list(model.parameters()), "lr": c["lr"] if lr_on else 0.0},
                        {"params": list(fc.parameters()), "lr": c["fc_lr"] if lr_on else 0.0}], betas=(0.9, 0.999), weight_decay=c["wd"])
    return model, fc, ppo, opt, dev


def _step(arch, stage, model, fc, ppo, opt, dev, sl, batch_patch_steps=True):
    from murcl_amd.models import rlmil
    from murcl_amd.train_RLMIL import supervised_step
    from murcl_amd.utils.datasets import BagPack, select_indices
    c = G15
    Ns, feats, cls, labels, u, eps = g15_inputs()
    Tn = c["T"]
    pack = BagPack.from_lists([T(feats[s]).to(dev) for s in sl], [cls[s] for s in sl])
    acts = [T(np.stack([u[s][t] for s in sl])).to(dev) for t in range(Tn if stage == 1 else 1)]
    noise = [T(np.stack([eps[s][t] for s in sl])).to(dev) for t in range(Tn - 1)]
    trace, mem = [], rlmil.Memory()
    loss, losses, rewards, logits = supervised_step(arch, model, fc, ppo, opt, pack, T(labels[sl]).to(dev), mem, T=Tn,
                                                    feat_size=c["fs"], train_stage=stage, bag_weight=c["bag_weight"], actions=acts,
                                                    eps=noise, trace=trace, return_logits=True, batch_patch_steps=batch_patch_steps)
    assert all(len(getattr(mem, f)) == 0 for f in rlmil.Memory.FIELDS)                       # :381
    a = [x for x in trace if torch.is_tensor(x)]
    assert len(a) == Tn
    lp = [x for x in trace if isinstance(x, dict)]
    ids = [select_indices(pack, x, c["fs"])[0].cpu().numpy() for x in a]
    return [l.item() for l in losses], torch.cat(rewards).cpu().numpy(), a, (lp[0]["logprobs"] if lp else None), ids, logits


@pytest.mark.parametrize("arch", ["ABMIL", "CLAM_SB", "DSMIL"])
@pytest.mark.parametrize("stage", [1, 2, 3])
@pytest.mark.parametrize("run", ["b1x2", "b4x2"])
def test_supervised_step_two_optimizer_steps_vs_reference_bodies(golden, arch, stage, run):
    g, c = golden("g15_supervised_steps"), G15
    B, steps, _ = G15_RUNS[run]
    tag = f"{arch}.s{stage}.{run}"
    if f"{tag}.losses" not in g.files:
        pytest.skip("the reference's CLAM / DSMIL bodies cannot run at batch size > 1 (train_RLMIL.py:335,516)")
    model, fc, ppo, opt, dev = _build(arch, stage)
    pre = {"model": {k: v.detach().clone() for k, v in model.state_dict().items()},
           "fc": {k: v.detach().clone() for k, v in fc.state_dict().items()},
           "policy": {} if ppo is None else {k: v.detach().clone() for k, v in ppo.policy.state_dict().items()}}
    for it in range(steps):
        sl = list(range(it * B, (it + 1) * B))
        losses, rewards, acts, logp, ids, logits = _step(arch, stage, model, fc, ppo, opt, dev, sl)
        np.testing.assert_allclose(losses, g[f"{tag}.losses"][it], rtol=1e-4 if it == 0 else 5e-4, err_msg=f"{tag} step {it}")
        np.testing.assert_allclose(rewards, g[f"{tag}.rewards"][it], rtol=5e-3, atol=5e-6)
        if stage != 1:
            np.testing.assert_allclose(torch.stack(acts[1:]).cpu().numpy(), g[f"{tag}.actions"][it], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(logp.cpu().numpy(), g[f"{tag}.logp"][it], rtol=1e-4, atol=1e-4)
            for t in range(1, c["T"]):
                assert np.array_equal(ids[t], g[f"{tag}.ids.{it}.{t}"]), f"{tag}: patch ids differ at step {it}, patch step {t}"
    now = {"model": model.state_dict(), "fc": fc.state_dict(), "policy": {} if ppo is None else ppo.policy.state_dict()}
    for name in ("model", "fc", "policy"):
        for k, v in now[name].items():
            key = f"{tag}.{name}_delta.{k}"
            if key not in g.files or g[key][0] == 0.0:
                # frozen at this stage, or never applied by the reference (ABMIL.fc, CLAM's classifiers, DSMIL's fcc): Adam skips it
                assert torch.equal(v, pre[name][k]), key
                continue
            _close_summ(_summ(v - pre[name][k]), g[key], 3e-2, key)
            _close_summ(_summ(v), g[f"{tag}.{name}.{k}"], 1e-4, key)
    if stage == 2:
        assert all(torch.equal(a, b) for a, b in zip(ppo.policy.parameters(), ppo.policy_old.parameters()))   # rlmil.py:183


@pytest.mark.parametrize("arch", ["ABMIL", "CLAM_SB", "DSMIL"])
@pytest.mark.parametrize("stage", [1, 2, 3])
@pytest.mark.parametrize("at_once", [True, False])
def test_batched_supervised_step_is_the_mean_of_the_reference_bodies(golden, arch, stage, at_once):
    """ONE product step over four slides (the batched form the reference's CLAM / DSMIL bodies do not have) at frozen parameters
    == the four batch-size-1 reference bodies: loss_t = their mean, rewards / actions / patch ids = theirs side by side."""
    g, c = golden("g15_supervised_steps"), G15
    if stage != 1 and not at_once:
        pytest.skip("patch steps are only batched when no step depends on the sampler")
    tag = f"{arch}.s{stage}.b1lr0"
    n = G15_RUNS["b1lr0"][1]
    model, fc, ppo, opt, dev = _build(arch, stage, lr_on=False)
    losses, rewards, acts, logp, ids, logits = _step(arch, stage, model, fc, ppo, opt, dev, list(range(n)), batch_patch_steps=at_once)
    np.testing.assert_allclose(losses, g[f"{tag}.losses"].mean(0), rtol=1e-4)
    np.testing.assert_allclose(rewards, g[f"{tag}.rewards"][:, :, 0].T, rtol=5e-3, atol=5e-6)
    if stage != 1:
        np.testing.assert_allclose(torch.stack(acts[1:]).cpu().numpy(), g[f"{tag}.actions"][:, :, 0].transpose(1, 0, 2),
                                   rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(logp.cpu().numpy(), g[f"{tag}.logp"][:, :, 0].T, rtol=1e-4, atol=1e-4)
        for t in range(1, c["T"]):
            want = np.concatenate([g[f"{tag}.ids.{s}.{t}"] for s in range(n)])
            assert np.array_equal(ids[t], want), f"{tag}: patch ids differ at patch step {t}"


def test_stage2_supervised_epochs_run_the_aggregator_in_eval_mode(monkeypatch):
    """``fit`` at train_stage 2 must score the sampler with Dropout off (train_RLMIL.py:299-301, 484-486, 691-693)
    and train with it on at stages 1 / 3: the mode every ``supervised_step`` of an epoch sees is recorded."""
    from murcl_amd import train_RLMIL as TR
    from murcl_amd.utils.datasets import DeviceSlideStore
    dev = torch.device("cuda:0")
    sets = [TR._SyntheticLabelled(n, 200, 512, 4, s) for n, s in ((4, 1), (2, 2), (2, 3))]
    stores = tuple(DeviceSlideStore.from_dataset(s, dev, dtype=torch.float32) for s in sets)
    seen, real = [], TR.supervised_step

    def spy(arch, model, fc, *a, **k):
        seen.append((model.training, fc.training))
        return real(arch, model, fc, *a, **k)

    monkeypatch.setattr(TR, "supervised_step", spy)
    from murcl_amd.models import rlmil
    from murcl_amd.optim import FlatAdam
    for stage in (1, 2):
        model, fc = TR.create_model("CLAM_SB", 512, 2, dev)
        ppo = rlmil.PPO(512, 512, 512, False, action_std=0.5, lr=1e-5, gamma=0.1, K_epochs=1, action_size=4) if stage == 2 else None
        opt = FlatAdam([{"params": list(model.parameters()) + list(fc.parameters()), "lr": 1e-4}]) if stage == 1 else None
        seen.clear()
        TR.fit("CLAM_SB", model, fc, ppo, opt, stores, 2, 2, T=2, feat_size=32, train_stage=stage, log=lambda *_: None)
        assert len(seen) == 4 and all(m == ((False, False) if stage == 2 else (True, True)) for m in seen), (stage, seen)
