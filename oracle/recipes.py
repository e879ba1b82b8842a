"""Input recipes shared by oracle/gen_goldens.py (which runs the reference) and the tests (which must not):
deterministic inputs regenerated from seeds, so goldens store outputs only.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch

from . import detrand, params as P

T = torch.from_numpy

G12 = dict(seed=77, B=4, K=10, fs=128, d=512, std=0.5, gamma=0.1, K_epochs=3, ppo_lr=1e-5, lr=1e-3, wd=1e-5, alpha=0.9)


def g12_inputs(Tn, batch=0):
    """Inputs of the G12 step (shared with the tests through this recipe): ragged bags, cluster lists, every draw.
    ``batch`` > 0: the inputs of a further, different batch (G17 runs two consecutive optimizer steps)."""
    c = G12
    seed, B, K = c["seed"], c["B"], c["K"]
    g = "g12." if batch == 0 else f"g12.b{batch}."
    Ns = [640 + 37 * b + 19 * batch for b in range(B)]
    feats = [P.bags(seed, f"{g}f{b}", 1, Ns[b], c["d"])[0] for b in range(B)]
    cls = [P.cluster_lists(seed, f"{g}c{b}", Ns[b], K) for b in range(B)]
    inj = {"actions": [[detrand.uniform(seed, f"{g}a{v}", (B, K)).astype(np.float32) for v in range(2)]],
           "u": [[detrand.uniform(seed, f"{g}u{t}{v}", (B, 1)).astype(np.float32) for v in range(2)] for t in range(Tn)],
           "perm": [[detrand.permutation(seed, f"{g}p{t}{v}", B) for v in range(2)] for t in range(Tn)],
           "eps": [[detrand.normal(seed, f"{g}e{t}{v}", (B, K)).astype(np.float32) for v in range(2)] for t in range(Tn - 1)]}
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



# ------------------------------------------------------------------ G15: the supervised step bodies of train_RLMIL.py
G15 = dict(seed=86, S=8, K=10, fs=64, d=512, C=2, T=3, std=0.5, gamma=0.1, K_epochs=3, ppo_lr=1e-5, lr=1e-4, fc_lr=1e-4,
           wd=1e-5, bag_weight=0.7, k_sample=8, labels=[0, 1, 1, 0, 1, 0, 0, 1])
# run name -> (bags per optimizer step, optimizer steps, learning rates on?)
#   b1x2:  the reference scripts' own --batch_size 1, two consecutive optimizer steps (slides 0, 1)
#   b1lr0: four batch-size-1 bodies at FROZEN parameters (lr = 0) = the per-slide terms of one batched step over slides 0..3
#   b4x2:  ABMIL only (the one body the reference can batch): two optimizer steps of four slides
G15_RUNS = {"b1x2": (1, 2, True), "b1lr0": (1, 4, False), "b4x2": (4, 2, True)}


def g15_inputs():
    """Slides, labels and the per-SLIDE draws of G15: u[s][t] [K] uniform window positions (every patch step at stage 1,
    t = 0 only at stages 2 / 3) and eps[s][t-1] [K] ~ N(0,1), the sampler's noise at stages 2 / 3."""
    c = G15
    seed, S, K, T_ = c["seed"], c["S"], c["K"], c["T"]
    Ns = [300 + 29 * s for s in range(S)]
    feats = [P.bags(seed, f"g15.f{s}", 1, Ns[s], c["d"])[0] for s in range(S)]
    cls = [P.cluster_lists(seed, f"g15.c{s}", Ns[s], K) for s in range(S)]
    u = [[detrand.uniform(seed, f"g15.u{s}.{t}", (K,)).astype(np.float32) for t in range(T_)] for s in range(S)]
    eps = [[detrand.normal(seed, f"g15.e{s}.{t}", (K,)).astype(np.float32) for t in range(T_ - 1)] for s in range(S)]
    return Ns, feats, cls, np.array(c["labels"], dtype=np.int64), u, eps


def g15_params(arch):
    """(aggregator, head, sampler) parameter dicts of G15."""
    c = G15
    seed = c["seed"]
    mp = P.abmil(seed, dim_out=c["C"]) if arch == "ABMIL" else {"CLAM_SB": P.clam_sb, "DSMIL": P.dsmil}[arch](seed)
    return mp, P.full_layer(seed, 512, 1024, c["C"]), P.actor_critic(seed, 512, 512, c["K"])



# ------------------------------------------------------------------ G16: ABMIL outside the launch scripts' default shape
G16 = dict(seed=29, B=3, N=160, keep=0.75,
           cases={"dropout": dict(d=512, L=512, D=128), "small": dict(d=320, L=256, D=64), "small_dropout": dict(d=320, L=256, D=64)})


def g16_inputs(case):
    """(parameter dict, bags [B,N,d], the two keep-multiplier masks [B,N,L] or None) of a G16 case."""
    c, k = G16, G16["cases"][case]
    p = P.abmil(c["seed"], dim_in=k["d"], L=k["L"], D=k["D"], dim_out=2)
    x = P.bags(c["seed"], f"g16.{case}.x", c["B"], c["N"], k["d"])
    masks = None
    if "dropout" in case:
        masks = [(detrand.uniform(c["seed"], f"g16.{case}.k{i}", (c["B"], c["N"], k["L"])) < c["keep"]).astype(np.float32) / c["keep"]
                 for i in range(2)]
    return p, x, masks




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
