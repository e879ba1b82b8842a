"""state-dict / checkpoint interchange with the reference.

CPU-only (modules are plain parameter holders until ``forward``).  The manifest fixture was written by
oracle/gen_goldens.py:g11_manifest from the reference's own modules; that generator also hands real checkpoint files
across (product file -> reference modules with strict loading, and back) and records the outcome in the manifest."""
import json
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _manifest():
    with open(os.path.join(HERE, "golden", "g11_state_dict_manifest.json")) as f:
        return json.load(f)


def _product():
    from murcl_amd.models import abmil, cl, clam, dsmil, rlmil
    return {
        "ABMIL": abmil.ABMIL(512, L=512, D=128, dim_out=2),
        "CLAM_SB": clam.CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512),
        "DSMIL": dsmil.build_dsmil(512, 2),
        "CL(ABMIL)": cl.CL(abmil.ABMIL(512, L=512, D=128, dim_out=128), projection_dim=128, n_features=512),
        "Full_layer": rlmil.Full_layer(512, 1024, True, 2),
        "ActorCritic": rlmil.ActorCritic(512, 512, 512, False, 0.5, 10),
    }


@pytest.mark.parametrize("name", ["ABMIL", "CLAM_SB", "DSMIL", "CL(ABMIL)", "Full_layer", "ActorCritic"])
def test_state_dicts_have_the_reference_names_shapes_and_order(name):
    want = _manifest()[name]
    got = [[k, list(v.shape)] for k, v in _product()[name].state_dict().items()]
    assert got == want


def test_checkpoint_round_trip_and_policies(tmp_path):
    from murcl_amd.models import abmil, cl, rlmil
    from murcl_amd.utils import checkpoint as C
    assert list(C.CHECKPOINT_KEYS) == _manifest()["checkpoint_keys"]
    torch.manual_seed(3)
    pre = cl.CL(abmil.ABMIL(512, L=512, D=128, dim_out=128), projection_dim=128, n_features=512)
    head = rlmil.Full_layer(512, 1024, True, 128)
    pol = rlmil.ActorCritic(512, 512, 512, False, 0.5, 10)

    class _P:
        policy, policy_old = pol, rlmil.ActorCritic(512, 512, 512, False, 0.5, 10)
    stage1 = tmp_path / "run" / "stage_1"
    C.save_checkpoint(C.make_state(7, pre, head, ppo=_P), True, str(stage1))
    assert (stage1 / "checkpoint.pth.tar").exists() and (stage1 / "model_best.pth.tar").exists()
    ck = torch.load(stage1 / "model_best.pth.tar", map_location="cpu")
    assert tuple(ck) == C.CHECKPOINT_KEYS and ck["epoch"] == 7
    # fine-tune: encoder.* without the prefix, encoder.fc* dropped -> the new 2-class fc is the only missing part
    clf = abmil.ABMIL(512, L=512, D=128, dim_out=2)
    fc_before = clf.fc.weight.detach().clone()
    missing = C.load_pretrained(clf, str(stage1 / "model_best.pth.tar"), "finetune")
    assert missing == ["fc.weight", "fc.bias"] and torch.equal(clf.fc.weight, fc_before)
    assert torch.equal(clf.encoder[3].weight, pre.encoder.encoder[3].weight) and all(p.requires_grad for p in clf.parameters())
    # linear evaluation: same weights, backbone frozen
    lin = abmil.ABMIL(512, L=512, D=128, dim_out=2)
    C.load_pretrained(lin, ck, "linear")
    assert sorted(n for n, p in lin.named_parameters() if p.requires_grad) == ["fc.bias", "fc.weight"]
    with pytest.raises(ValueError):
        C.load_pretrained(lin, ck, "scratch")
    # stage k continues from ../stage_{k-1}/model_best.pth.tar
    assert C.stage_checkpoint_path(str(tmp_path / "run" / "stage_2"), 2) == str(stage1 / "model_best.pth.tar")
    m2 = cl.CL(abmil.ABMIL(512, L=512, D=128, dim_out=128), projection_dim=128, n_features=512)
    h2 = rlmil.Full_layer(512, 1024, True, 128)

    class _Q:
        policy, policy_old = rlmil.ActorCritic(512, 512, 512, False, 0.5, 10), rlmil.ActorCritic(512, 512, 512, False, 0.5, 10)
    assert C.load_stage(m2, h2, _Q, C.stage_checkpoint_path(str(tmp_path / "run" / "stage_2"), 2)) == 7
    assert torch.equal(h2.rnn.weight_hh_l0, head.rnn.weight_hh_l0)
    assert torch.equal(_Q.policy.gru.weight_ih_l0, pol.gru.weight_ih_l0) and torch.equal(_Q.policy_old.actor[0].bias, pol.actor[0].bias)
    no_pol = dict(ck, policy=None)
    with pytest.raises(KeyError):
        C.load_stage(m2, h2, _Q, no_pol)
    C.load_stage(m2, h2, _Q, no_pol, policy_ckpt=ck)            # stage 2: policy from the pre-training checkpoint


def test_generator_exchanged_real_files_with_the_reference():
    assert _manifest()["exchange"] == {"reference_loaded_product_file_strict": True, "product_loaded_reference_file_strict": True,
                                       "finetune_strip_identical": True}


def test_checkpoint_writer_writes_in_order_off_thread_and_reports_failures(tmp_path):
    """utils.checkpoint.CheckpointWriter: the per-epoch files of the training loop, written by a background thread - same files
    as save_checkpoint, in submission order, complete after close(); a failure surfaces in the training thread."""
    import torch
    from murcl_amd.utils import checkpoint as C
    w = C.CheckpointWriter()
    d = tmp_path / "run"
    for epoch in range(1, 6):
        state = {"epoch": epoch, "model_state_dict": {"w": torch.full((256, 256), float(epoch))}, "fc": {}, "optimizer": None,
                 "ppo_optimizer": None, "policy": None}
        w.submit(state, epoch in (2, 4), str(d))
    w.close()
    last, best = torch.load(d / "checkpoint.pth.tar"), torch.load(d / "model_best.pth.tar")
    assert last["epoch"] == 5 and float(last["model_state_dict"]["w"][0, 0]) == 5.0
    assert best["epoch"] == 4 and set(last) == set(C.CHECKPOINT_KEYS)
    w.close()                                                          # idempotent
    bad = C.CheckpointWriter()
    blocker = tmp_path / "file_not_dir"
    blocker.write_text("x")
    bad.submit({"epoch": 1}, False, str(blocker / "sub"))               # makedirs under a regular file fails in the thread
    with pytest.raises(RuntimeError, match="writing a checkpoint failed"):
        bad.close()
