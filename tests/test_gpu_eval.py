"""SURVEY 8(f) rank 2 on MI355X: the validation / test path of train_RLMIL.py against goldens produced by the
reference's own test_ABMIL / test_CLAM / test_DSMIL and utils/general.get_metrics (oracle/gen_goldens.py: g10_eval)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, params as P  # noqa: E402

T = torch.from_numpy
SPLIT = dict(S=7, K=4, fs=64, T=3, d=512, C=2)             # = oracle/gen_goldens.py G10_SPLIT


def _inputs(seed=71):
    c = SPLIT
    Ns = [260 - 23 * b for b in range(c["S"])]
    feats = [P.bags(seed, f"g10.f{b}", 1, Ns[b], c["d"])[0] for b in range(c["S"])]
    cls = [P.cluster_lists(seed, f"g10.c{b}", Ns[b], c["K"]) for b in range(c["S"])]
    labels = np.array([0, 1, 1, 0, 1, 0, 0], dtype=np.int64)
    acts = [detrand.uniform(seed, f"g10.a{t}", (c["S"], c["K"])).astype(np.float32) for t in range(c["T"])]
    return Ns, feats, cls, labels, acts


class _Set:
    def __init__(self, feats, cls, labels):
        self.f, self.c, self.y = feats, cls, labels

    def __len__(self):
        return len(self.f)

    def __getitem__(self, i):
        return T(self.f[i]), self.c[i], int(self.y[i]), f"case{i}"


@pytest.mark.parametrize("arch", ["ABMIL", "CLAM_SB", "DSMIL"])
def test_whole_split_evaluation_matches_the_reference(golden, arch):
    from murcl_amd.models import rlmil
    from murcl_amd.train_RLMIL import create_model, evaluate_split, predictions_frame
    from murcl_amd.utils.datasets import DeviceSlideStore
    g = golden("g10_eval")
    dev = torch.device("cuda:0")
    c = SPLIT
    Ns, feats, cls, labels, acts = _inputs()
    model, fc = create_model(arch, c["d"], c["C"], dev)
    pk = P.abmil(73, dim_out=c["C"]) if arch == "ABMIL" else {"CLAM_SB": P.clam_sb, "DSMIL": P.dsmil}[arch](73)
    model.load_state_dict(P.to_torch(pk))
    fc.load_state_dict(P.to_torch(P.full_layer(73, 512, 1024, c["C"])))
    model.train(), fc.train()                                    # evaluate_split must switch to eval itself (CLAM dropout)
    store = DeviceSlideStore.from_dataset(_Set(feats, cls, labels), dev)
    y = T(store.labels).to(dev)
    loss, acc, auc, prec, rec, f1, logits, labs = evaluate_split(arch, model, fc, None, rlmil.Memory(), store.pack(range(c["S"])), y,
                                                                  T=c["T"], feat_size=c["fs"], actions=[T(a) for a in acts])
    assert model.training and fc.training                        # restored
    np.testing.assert_allclose(logits.cpu().numpy(), g[f"{arch}.outputs"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(loss, float(g[f"{arch}.loss"]), rtol=2e-4)
    np.testing.assert_allclose([acc, auc, prec, rec, f1], g[f"{arch}.metrics"], rtol=1e-6, atol=1e-7)
    df = predictions_frame(logits, labs, store.case_ids)
    assert list(df.columns) == ["label", "pred", "correct", "prob0", "prob1"] and df.index.name == "case_id"
    assert df.loc["case2", "label"] == 1 and abs(df.loc["case2", "prob0"] + df.loc["case2", "prob1"] - 1) < 1e-6
    assert df["correct"].mean() == pytest.approx(acc)


def test_fit_selects_the_best_epoch_and_trains(tmp_path):
    """The epoch loop on resident splits: training moves the weights, the best validation epoch is kept with the
    reference's checkpoint keys, predictions cover the test split."""
    from murcl_amd.optim import FlatAdam
    from murcl_amd.train_RLMIL import create_model, fit
    from murcl_amd.utils.datasets import DeviceSlideStore
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)

    def split(n, tag):
        Ns = [200 + 7 * i for i in range(n)]
        ys = np.arange(n) % 2
        # class signal: class-1 slides carry a shifted first feature block
        f = [P.bags(81, f"{tag}{i}", 1, Ns[i], 512)[0] + (0.8 * ys[i]) * (np.arange(512) < 64) for i in range(n)]
        cl = [P.cluster_lists(81, f"{tag}c{i}", Ns[i], 4) for i in range(n)]
        return DeviceSlideStore.from_dataset(_Set([x.astype(np.float32) for x in f], cl, ys), dev)

    stores = (split(16, "tr"), split(6, "va"), split(6, "te"))
    model, fc = create_model("ABMIL", 512, 2, dev)
    opt = FlatAdam([{"params": list(model.parameters()) + list(fc.parameters()), "lr": 2e-4}])
    w0 = model.encoder[0].weight.detach().clone()
    lines = []
    best, final, frame = fit("ABMIL", model, fc, None, opt, stores, epochs=3, batch_size=8, T=2, feat_size=64, rng=rng, log=lines.append)
    assert len(lines) == 3 and not torch.equal(model.encoder[0].weight, w0)
    assert {"epoch", "model_state_dict", "fc", "optimizer", "ppo_optimizer", "policy"} <= set(best) and 1 <= best["epoch"] <= 3
    assert final[0] == best["epoch"] and len(frame) == 6 and frame.index[0] == "case0"


def test_rlmil_script_end_to_end_with_finetune_from_a_pretraining_checkpoint(tmp_path, capsys):
    """train_RLMIL.main: synthetic labelled splits in HBM, encoder from a MuRCL pre-training checkpoint (prefix strip),
    epochs with validation-based selection, model_best.pth.tar + pred.csv written."""
    import pandas as pd
    from murcl_amd import train_RLMIL
    from murcl_amd.models import abmil, cl, rlmil
    from murcl_amd.utils import checkpoint as C
    torch.manual_seed(2)
    pre = cl.CL(abmil.ABMIL(512, L=512, D=128, dim_out=128), 128, 512)
    C.save_checkpoint(C.make_state(1, pre, rlmil.Full_layer(512, 1024, True, 128)), True, str(tmp_path / "pre"))
    save = tmp_path / "ft" / "stage_1"
    final = train_RLMIL.main(["--arch", "ABMIL", "--synthetic", "12,4,6,300", "--num_clusters", "4", "--feat_size", "64", "--T", "2",
                              "--epochs", "2", "--batch_size", "4", "--train_method", "finetune",
                              "--checkpoint_pretrained", str(tmp_path / "pre" / "model_best.pth.tar"), "--save_dir", str(save),
                              "--device", "0", "--save_model", "--scheduler", "CosineAnnealingLR"])
    out = capsys.readouterr().out
    assert "msg_model missing_keys: ['fc.weight', 'fc.bias']" in out and "epoch 2:" in out
    for f in ("args.yaml", "losses.csv", "accs.csv", "aucs.csv", "results.csv", "final_res.csv"):      # train_RLMIL.py:868-878,1033,1056
        assert (save / f).exists(), f
    df = pd.read_csv(save / "pred.csv", index_col="case_id")
    assert len(df) == 6 and list(df.columns) == ["label", "pred", "correct", "prob0", "prob1"]
    ck = torch.load(save / "model_best.pth.tar", map_location="cpu")
    assert tuple(ck) == C.CHECKPOINT_KEYS and ck["epoch"] == final[0]
    assert not torch.equal(ck["model_state_dict"]["encoder.0.weight"], pre.encoder.encoder[0].weight)   # trained on from the loaded weights


@pytest.mark.parametrize("N,dtype", [(23457, torch.float32), (40000, torch.bfloat16), (37, torch.float32)])
def test_whole_slide_attention_scores_for_heatmaps(N, dtype):
    """8(f) rank 4: CLAM_SB.bag_forward(attention_only=True) on every patch of a slide (no sub-sampling, N not a
    multiple of any tile), the call scripts/create_heatmaps.py:159-162 makes - raw scores [1,N] vs the oracle."""
    from oracle import mil_oracle as O
    from murcl_amd.models.clam import CLAM_SB
    dev = torch.device("cuda:0")
    m = CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512)
    pk = P.clam_sb(91)
    m.load_state_dict(P.to_torch(pk))
    m = m.to(dev).eval()
    m.compute_dtype = dtype
    x = P.bags(91, f"slide{N}", 1, N, 512)[0]
    with torch.no_grad():
        s = m.bag_forward(T(x).to(dev), attention_only=True)
        s3 = m(T(x).unsqueeze(0).to(dev), attention_only=True)[0]           # the forward() spelling
    assert s.shape == (1, N) and torch.allclose(s, s3, rtol=1e-5, atol=1e-6)   # (N <= 128 rows take the split-K atomics path)
    want = O.clam_sb_forward(P.to_torch(pk), T(x).unsqueeze(0))[2]
    if dtype == torch.float32:
        np.testing.assert_allclose(s.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4 * float(want.abs().max()))
    else:
        err = (s.cpu() - want).abs().max().item()
        assert err < 3e-2 * float(want.abs().max()) + 3e-2
        # what a heat-map needs: the ranking of the hottest patches survives bf16
        top = set(want[0].topk(50).indices.tolist())
        assert len(top & set(s[0].cpu().topk(200).indices.tolist())) >= 45


def test_scratch_three_stage_pipeline_runs_end_to_end(tmp_path, capsys):
    """``--train_method scratch`` (the parser default) through stages 1 -> 2 -> 3 with the default
    ``../stage_{k-1}/model_best.pth.tar`` hand-off: stage 2 starts from a FRESH sampler (train_RLMIL.py:199-214; the
    stage-1 checkpoint holds no policy), stage 3 loads the stage-2 policy."""
    from murcl_amd import train_RLMIL
    common = ["--arch", "ABMIL", "--synthetic", "8,4,4,200", "--num_clusters", "4", "--feat_size", "64", "--T", "3",
              "--batch_size", "4", "--device", "0", "--save_model", "--exist_ok", "--train_method", "scratch"]
    for stage in (1, 2, 3):
        train_RLMIL.main(common + ["--train_stage", str(stage), "--epochs", "1", "--ppo_epochs", "1",
                                   "--save_dir", str(tmp_path / "run" / f"stage_{stage}")])
        ck = torch.load(tmp_path / "run" / f"stage_{stage}" / "model_best.pth.tar", map_location="cpu")
        assert (ck["policy"] is None) == (stage == 1)
        assert (ck["optimizer"] is None) == (stage == 2) and (ck["ppo_optimizer"] is None) == (stage == 1)
    s2 = torch.load(tmp_path / "run" / "stage_2" / "model_best.pth.tar", map_location="cpu")
    s1 = torch.load(tmp_path / "run" / "stage_1" / "model_best.pth.tar", map_location="cpu")
    assert all(torch.equal(s1["model_state_dict"][k], s2["model_state_dict"][k]) for k in s1["model_state_dict"])   # stage 2 trains the sampler only
