"""Input recipes shared by oracle/gen_goldens.py (which runs the reference) and the tests (which must not):
deterministic inputs regenerated from seeds, so goldens store outputs only.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch

from . import detrand, params as P

T = torch.from_numpy

G12 = dict(seed=77, B=4, K=10, fs=128, d=512, std=0.5, gamma=0.1, K_epochs=3, ppo_lr=1e-5, lr=1e-3, wd=1e-5, alpha=0.9)


def g12_inputs(Tn):
    """Inputs of the G12 step (shared with the tests through this recipe): ragged bags, cluster lists, every draw."""
    c = G12
    seed, B, K = c["seed"], c["B"], c["K"]
    Ns = [640 + 37 * b for b in range(B)]
    feats = [P.bags(seed, f"g12.f{b}", 1, Ns[b], c["d"])[0] for b in range(B)]
    cls = [P.cluster_lists(seed, f"g12.c{b}", Ns[b], K) for b in range(B)]
    inj = {"actions": [[detrand.uniform(seed, f"g12.a{v}", (B, K)).astype(np.float32) for v in range(2)]],
           "u": [[detrand.uniform(seed, f"g12.u{t}{v}", (B, 1)).astype(np.float32) for v in range(2)] for t in range(Tn)],
           "perm": [[detrand.permutation(seed, f"g12.p{t}{v}", B) for v in range(2)] for t in range(Tn)],
           "eps": [[detrand.normal(seed, f"g12.e{t}{v}", (B, K)).astype(np.float32) for v in range(2)] for t in range(Tn - 1)]}
    # lambda exactly as mixup forms it (datasets.py:265): alpha + U * (1 - alpha) in float32 tensor arithmetic
    inj["draws"] = [[((c["alpha"] + T(inj["u"][t][v]) * (1 - c["alpha"])).numpy(), inj["perm"][t][v]) for v in range(2)]
                    for t in range(Tn)]
    return Ns, feats, cls, inj


def window_margin(n_patches, clusters, actions, feat_size):
    """Distance of a * (n_j - size_j) from the nearest integer boundary of floor() (datasets.py:290), over all clusters
    whose action is strictly inside (0,1): the GPU's actions differ from the CPU's in the last bits, so goldens whose ids
    must match bit-for-bit are generated away from such boundaries."""
    from oracle import select_oracle as S
    _, _, size = S.window_bounds(n_patches, [len(c) for c in clusters], actions, feat_size)
    span = (np.array([len(c) for c in clusters]) - size).astype(np.float32)
    x = np.asarray(actions, np.float32) * span
    inside = (np.asarray(actions) > 0) & (np.asarray(actions) < 1) & (span != 0)
    fr = x - np.floor(x)
    return float(np.min(np.where(inside, np.minimum(fr, 1 - fr), 1.0)))




# ------------------------------------------------------------------ launch-script argument vectors (runs/*.sh)
def _murcl_argv(stage, backbone_lr, fc_lr):
    return ["--dataset", "Camelyon16", "--data_csv", "path/to/data_csv.csv", "--data_split_json", "path/to/data_split_json.json",
            "--feat_size", "1024", "--preload", "--train_stage", str(stage), "--T", "6", "--scheduler", "CosineAnnealingLR",
            "--batch_size", "128", "--epochs", "100", "--backbone_lr", backbone_lr, "--fc_lr", fc_lr, "--patience", "10",
            "--arch", "CLAM_SB", "--device", "3", "--exist_ok"]


def _rlmil_argv(method, stage, backbone_lr, fc_lr):
    pre = [] if method == "scratch" else ["--checkpoint_pretrained", "path/to/pretrained/checkpoint/stage_3/model.best.tar"]
    return ["--dataset", "Camelyon16", "--data_csv", "path/to/data_csv.csv", "--data_split_json", "path/to/data_split_json.json",
            "--train_data", "train", "--feat_size", "1024", "--preload", "--train_method", method, "--train_stage", str(stage),
            *pre, "--T", "6", "--scheduler", "CosineAnnealingLR", "--batch_size", "1", "--epochs", "40",
            "--backbone_lr", backbone_lr, "--fc_lr", fc_lr, "--arch", "CLAM_SB", "--device", "3", "--save_model", "--exist_ok"]


def run_script_argvs():
    """The invocations the reference's launch scripts make: ``{script: (entry, [argv per invocation])}`` - stages 1 and 2
    with the first learning rates, stage 3 with the halved ones (runs/pretrain.sh:4-39, scratch.sh, finetune.sh, linear.sh)."""
    out = {"pretrain.sh": ("train_MuRCL", [_murcl_argv(1, "0.0001", "0.00005"), _murcl_argv(2, "0.0001", "0.00005"),
                                           _murcl_argv(3, "0.00005", "0.00001")])}
    for m in ("scratch", "finetune", "linear"):
        out[f"{m}.sh"] = ("train_RLMIL", [_rlmil_argv(m, 1, "0.0001", "0.00005"), _rlmil_argv(m, 2, "0.0001", "0.00005"),
                                          _rlmil_argv(m, 3, "0.00005", "0.00001")])
    return out
