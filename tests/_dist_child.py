"""Child process of tests/test_gpu_dist_step.py (not a test module).

    python tests/_dist_child.py <mode> <rank> <world> <port> <workdir>

mode "nccl1": ONE rank, backend nccl (= RCCL), initialised before any other GPU call; runs the MuRCL batch step for
stages 1-3 twice from the same state and draws - the single-process path and the multi-GPU path (all-gathered NT-Xent,
flat gradient all-reduce, PPO collectives) - and stores both results.
mode "gloo2": rank ``rank`` of ``world`` processes that share cuda:0, backend gloo (RCCL refuses two ranks on one
device; gloo collectives are staged through the host by murcl_amd.dist) - the real kernels on B/world bags per rank.
mode "single": no process group, all B bags: the reference result for "gloo2".
modes "gloo2g" / "singleg" (round 6, --global_mixup): the same with mix-up permutations over the WHOLE batch - every rank holds all B
raw bags (the replicated store), trains on its half (``pretrain_step(local=...)``) and all-gathers the sampler's actions; the
single-process run with the same global permutations is its reference.
mode "cli2g": rank ``rank`` of ``world`` processes running the ENTRY SCRIPT itself (``train_MuRCL.main``, stages 1-3 in turn) with
``--global_mixup --dist_backend gloo`` and its own random draws; stores the parameters each rank holds after each stage.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED, B, K, FS, TN = 41, 4, 6, 96, 3


def scenario(lo, hi, global_mix=False):
    """Bags [lo, hi) of the fixed B-bag batch with their draws; mix-up partners stay inside each half (rank-local).
    ``global_mix``: ALL B bags with actions / draws for all of them (only the sampler's noise is cut to [lo, hi)) and permutations
    over the whole batch."""
    from oracle import detrand, params as P
    Ns = [420 + 29 * b for b in range(B)]
    blo, bhi = (0, B) if global_mix else (lo, hi)
    feats = [P.bags(SEED, f"f{b}", 1, Ns[b], 512)[0] for b in range(blo, bhi)]
    cls = [P.cluster_lists(SEED, f"c{b}", Ns[b], K) for b in range(blo, bhi)]
    half = B // 2
    perms = [[np.concatenate([detrand.permutation(SEED, f"p{t}{v}{h}", half) + h * half for h in range(2)]) for v in range(2)]
             for t in range(TN)]
    if global_mix:
        perms = [[detrand.permutation(SEED, f"gp{t}{v}", B) for v in range(2)] for t in range(TN)]
        assert any((np.asarray(perms[t][v])[:half] >= half).any() for t in range(TN) for v in range(2)), "no cross-rank partner drawn"
        full = lambda a: np.ascontiguousarray(a)                      # noqa: E731
        cut_ = lambda a: np.ascontiguousarray(a[lo:hi])               # noqa: E731
        inj = {"actions": [[full(detrand.uniform(SEED, f"a{t}{v}", (B, K)).astype(np.float32)) for v in range(2)] for t in range(TN)],
               "draws": [[(full(detrand.uniform(SEED, f"l{t}{v}", (B, 1), 0.9, 1.0).astype(np.float32)), full(np.asarray(perms[t][v])))
                          for v in range(2)] for t in range(TN)],
               "eps": [[cut_(detrand.normal(SEED, f"e{t}{v}", (B, K)).astype(np.float32)) for v in range(2)] for t in range(TN - 1)]}
        return feats, cls, inj

    def cut(a):
        return np.ascontiguousarray(a[lo:hi])
    inj = {"actions": [[cut(detrand.uniform(SEED, f"a{t}{v}", (B, K)).astype(np.float32)) for v in range(2)] for t in range(TN)],
           "draws": [[(cut(detrand.uniform(SEED, f"l{t}{v}", (B, 1), 0.9, 1.0).astype(np.float32)), cut(perms[t][v]) - lo)
                      for v in range(2)] for t in range(TN)],
           "eps": [[cut(detrand.normal(SEED, f"e{t}{v}", (B, K)).astype(np.float32)) for v in range(2)] for t in range(TN - 1)]}
    return feats, cls, inj


def write_prev_stage(workdir, stage):
    from oracle import params as P
    prev = os.path.join(workdir, f"stage_{stage - 1}")
    os.makedirs(prev, exist_ok=True)
    path = os.path.join(prev, "model_best.pth.tar")
    if not os.path.exists(path):
        torch.save({"epoch": 1, "model_state_dict": {"encoder." + k: v for k, v in P.to_torch(P.abmil(SEED)).items()},
                    "fc": P.to_torch(P.full_layer(SEED)), "optimizer": None, "ppo_optimizer": None,
                    "policy": P.to_torch(P.actor_critic(SEED, 512, 512, K))}, path)


def run_step(workdir, stage, lo, hi, dist_path, global_mix=False):
    from oracle import params as P
    from murcl_amd.models import rlmil
    from murcl_amd.train_MuRCL import build_parser, create_model, get_optimizer, pretrain_step
    from murcl_amd.utils.datasets import BagPack
    from murcl_amd.utils.losses import NT_Xent
    T = torch.from_numpy
    dev = torch.device("cuda:0")
    args = build_parser().parse_args(["--arch", "ABMIL", "--dtype", "f32", "--train_stage", str(stage), "--T", str(TN),
                                      "--feat_size", str(FS), "--batch_size", str(hi - lo), "--num_clusters", str(K),
                                      "--ppo_lr", "1e-5", "--backbone_lr", "1e-3", "--fc_lr", "1e-3", "--K_epochs", "2",
                                      "--save_dir", os.path.join(workdir, f"stage_{stage}")])
    if stage == 1:
        model, fc, ppo = create_model(args, 512, dev)
        model.encoder.load_state_dict(P.to_torch(P.abmil(SEED)))
        fc.load_state_dict(P.to_torch(P.full_layer(SEED)))
    else:
        model, fc, ppo = create_model(args, 512, dev)
        pol = P.to_torch(P.actor_critic(SEED, 512, 512, K))
        ppo.policy.load_state_dict(pol)
        ppo.policy_old.load_state_dict(pol)
        ppo.data_parallel = True if dist_path else False
    opt = get_optimizer(args, model, fc)
    feats, cls, inj = scenario(lo, hi, global_mix)
    pack = BagPack.from_lists([T(f).to(dev) for f in feats], cls)
    dinj = {"actions": [[T(a).to(dev) for a in row] for row in inj["actions"]],
            "draws": [[(T(l).to(dev), T(p).to(dev)) for l, p in row] for row in inj["draws"]],
            "eps": [[T(e).to(dev) for e in row] for row in inj["eps"]]}
    loss, losses, rewards = pretrain_step(args, model, fc, ppo, NT_Xent(hi - lo, 1.0), opt, pack,
                                          [rlmil.Memory(), rlmil.Memory()], world=2 if dist_path else 1, injected=dinj,
                                          local=(lo, hi - lo) if (global_mix and dist_path) else None)
    assert not any(r.requires_grad for r in rewards)
    out = {"losses": torch.stack(losses).cpu(), "rewards": torch.cat(rewards).cpu(),
           "model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
           "fc": {k: v.detach().cpu().clone() for k, v in fc.state_dict().items()}}
    if ppo is not None:
        out["policy"] = {k: v.detach().cpu().clone() for k, v in ppo.policy.state_dict().items()}
    return out


def run_cli(rank, world, port, workdir):
    from murcl_amd import train_MuRCL
    kept, inner = {}, train_MuRCL.train

    def train_and_keep(args, train_set, model, fc, ppo, *rest):
        out = inner(args, train_set, model, fc, ppo, *rest)
        kept[f"s{args.train_stage}"] = {"model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                                        "fc": {k: v.detach().cpu().clone() for k, v in fc.state_dict().items()},
                                        "policy": None if ppo is None else {k: v.detach().cpu().clone() for k, v in ppo.policy.state_dict().items()}}
        return out
    train_MuRCL.train = train_and_keep
    for stage in (1, 2, 3):
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(port) + stage), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        train_MuRCL.main(["--synthetic", "12,360", "--num_clusters", "4", "--feat_size", "64", "--T", "3", "--batch_size", "3",
                          "--data_repeat", "2", "--arch", "ABMIL", "--device", "0", "--exist_ok", "--dtype", "f32", "--epochs", "2",
                          "--ppo_epochs", "2", "--train_stage", str(stage), "--global_mixup", "--dist_backend", "gloo",
                          "--save_dir", os.path.join(workdir, "cli", f"stage_{stage}")])
    torch.save(kept, os.path.join(workdir, f"cli2g_{rank}.pt"))


def main():
    mode, rank, world, port, workdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    if mode == "cli2g":
        return run_cli(rank, world, port, workdir)
    if mode not in ("single", "singleg"):
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
        if mode == "nccl1":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))   # before any other GPU call
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for stage in (1, 2, 3):
        if stage > 1:
            write_prev_stage(workdir, stage) if rank == 0 else None
            if mode in ("gloo2", "gloo2g"):
                import torch.distributed as dist
                dist.barrier()
        if mode == "nccl1":
            res[f"s{stage}.plain"] = run_step(workdir, stage, 0, B, False)
            res[f"s{stage}.dist"] = run_step(workdir, stage, 0, B, True)
        elif mode in ("gloo2", "gloo2g"):
            per = B // world
            res[f"s{stage}"] = run_step(workdir, stage, rank * per, (rank + 1) * per, True, global_mix=(mode == "gloo2g"))
        else:
            res[f"s{stage}"] = run_step(workdir, stage, 0, B, False, global_mix=(mode == "singleg"))
    torch.save(res, os.path.join(workdir, f"{mode}_{rank}.pt"))
    if mode not in ("single", "singleg"):
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
