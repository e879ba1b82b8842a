#!/usr/bin/env python
"""MuRCL pre-training entry on MI355X (reference: train_MuRCL.py).

Keeps the reference's flags, three-stage schedule (1: contrastive warm-up with random sub-bags, 2: PPO only,
3: joint), model construction (``create_model``), optimizer / scheduler choices and checkpoint dictionary keys;
the per-batch hot step (train_MuRCL.py:233-304) is ``pretrain_step`` below and runs entirely on the HIP kernels:
device-side sub-bag selection + fused gather/mix-up, one batched aggregator call for both views, the recurrent
head, single-launch NT-Xent, PPO act/update.  One process per GPU: launch with
``python -m torch.distributed.run --nproc-per-node N -m murcl_amd.train_MuRCL ...`` (instead of DataParallel).

Data: ``--data_csv`` in the reference's WSIWithCluster format (csv + npz ``img_features`` + json cluster lists),
or ``--synthetic B,N`` for random bags.
"""
import gc
import argparse
import json
import math
import os
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from murcl_amd import dist as mdist, ops
from murcl_amd.models import abmil, cl, clam, rlmil
from murcl_amd.optim import FlatAdam
from murcl_amd.utils.datasets import BagPack, DeviceSlideStore, subbag_views
from murcl_amd.utils.losses import NT_Xent


# ------------------------------------------------------------------------------------------------ model / optim
def create_model(args, dim_patch, device):
    """train_MuRCL.py:70-151 (DataParallel replaced by one process per GPU)."""
    if args.arch == "ABMIL":
        enc = abmil.ABMIL(dim_in=dim_patch, L=args.model_dim, D=args.D, dim_out=args.projection_dim, dropout=args.dropout)
        n_features = enc.fc.in_features
    elif args.arch == "CLAM_SB":
        enc = clam.CLAM_SB(gate=True, size_arg=args.size_arg, dropout=True, k_sample=args.k_sample,
                           n_classes=args.projection_dim, subtyping=True, in_dim=dim_patch)
        n_features = enc.classifiers.in_features
    else:
        raise NotImplementedError(f"args.arch error, {args.arch}. ")
    enc.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = cl.CL(enc, projection_dim=args.projection_dim, n_features=n_features)
    fc = rlmil.Full_layer(args.feature_num, args.fc_hidden_dim, args.fc_rnn, args.projection_dim)
    ppo = None
    if args.train_stage in (2, 3):
        if args.checkpoint is None:
            args.checkpoint = str(Path(args.save_dir).parent / f"stage_{args.train_stage - 1}" / "model_best.pth.tar")
        ckpt = torch.load(args.checkpoint, map_location="cpu")
        model.load_state_dict(ckpt["model_state_dict"])
        fc.load_state_dict(ckpt["fc"])
        ppo = rlmil.PPO(dim_patch, args.model_dim, args.policy_hidden_dim, args.policy_conv, action_std=args.action_std,
                        lr=args.ppo_lr, gamma=args.ppo_gamma, K_epochs=args.K_epochs, action_size=args.num_clusters)
        if args.train_stage == 3:
            ppo.policy.load_state_dict(ckpt["policy"])
            ppo.policy_old.load_state_dict(ckpt["policy"])
    elif args.train_stage != 1:
        raise ValueError
    return model.to(device), fc.to(device), ppo


def get_optimizer(args, model, fc):
    """train_MuRCL.py:154-171 (Adam only; one flat buffer per parameter group)."""
    if args.train_stage == 2:
        args.epochs = args.ppo_epochs
        return None
    if args.optimizer != "Adam":
        raise NotImplementedError("murcl_amd ships the fused Adam only")
    return FlatAdam([{"params": list(model.parameters()), "lr": args.backbone_lr},
                     {"params": list(fc.parameters()), "lr": args.fc_lr}],
                    betas=(args.beta1, args.beta2), weight_decay=args.wdecay)


def cosine_lr(base, epoch, epochs, warmup, eta_min=1e-6):
    """CosineAnnealingLR(T_max=epochs-warmup, eta_min=1e-6) stepped after epoch >= warmup (train_MuRCL.py:181,312-313)."""
    t = max(0, epoch + 1 - warmup)
    return eta_min + (base - eta_min) * (1 + math.cos(math.pi * t / max(1, epochs - warmup))) / 2


# ------------------------------------------------------------------------------------------------ the hot step
def pretrain_step(args, model, fc, ppo, criterion, optimizer, pack, memory_list, world=1, injected=None):
    """One optimizer step on a batch of raw bags (train_MuRCL.py:233-304).

    ``pack``: BagPack of this rank's bags.  ``injected`` (tests): dict with 'actions' [T][2][B,K] and
    'draws' [T][2](lambda [B,1], perm [B]) replacing the random draws.  Returns (loss, losses[T], rewards[T-1])."""
    B, K, dev = pack.B, pack.K, pack.feats.device
    dt_ = model.encoder.compute_dtype
    train_enc = args.train_stage != 2
    if (args.train_stage == 1 or injected is not None) and args.T > 1 and train_enc \
            and not getattr(args, "no_batched_stage1", False):
        return _pretrain_step_all_patch_steps_at_once(args, model, fc, criterion, optimizer, pack, injected, world)
    losses, rewards, sim_last, states = [], [], None, None
    for t in range(args.T):
        if injected is not None:
            acts = [a.to(dev) for a in injected["actions"][t]]
        elif t == 0 or args.train_stage == 1:
            acts = [torch.rand((B, K), device=dev) for _ in range(2)]                        # :235,256-258
        else:
            acts = [ppo.select_action(s, m, restart_batch=(t == 1)) for s, m in zip(states, memory_list)]   # :259-265
        views, _ = subbag_views(pack, acts, args.feat_size, alpha=args.alpha, out_dtype=dt_,
                                draws=None if injected is None else injected["draws"][t])   # :237-239,266-269
        with torch.set_grad_enabled(train_enc):
            outputs, states = model(views)                                                   # :242,271
            outputs = fc.forward_views(outputs, restart=(t == 0))                            # :243,272
            if world > 1:
                loss, sim = mdist.gathered_nt_xent(outputs[0], outputs[1], args.temperature)
            else:
                loss = criterion(outputs[0], outputs[1])                                     # :249,277
                sim = criterion.last_similarity
        losses.append(loss)
        if t > 0:
            reward = (sim_last - sim).view(1, -1)                                            # :282-283
            rewards.append(reward)
            for m in memory_list:
                m.rewards.append(reward)
        sim_last = sim
    loss = sum(losses) / args.T                                                              # :291
    if train_enc:
        optimizer.zero_grad()
        loss.backward(ops.unit_grad(loss))
        if world > 1:
            mdist.all_reduce_grads(optimizer.flat_grads())
        optimizer.step()                                                                     # :293-295
    else:
        for m in memory_list:
            ppo.update(m)                                                                    # :297-298
    for m in memory_list:
        m.clear_memory()
    return loss.detach(), [l.detach() for l in losses], rewards


def _pretrain_step_all_patch_steps_at_once(args, model, fc, criterion, optimizer, pack, injected=None, world=1):
    """Stage 1 draws every patch step's window positions at random (train_MuRCL.py:235,256-258): no step depends on the
    aggregator states of the step before, so the sub-bags of all T steps are built into ONE buffer and the aggregator
    runs ONCE over 2*T*B bags - each weight-stationary / wgrad kernel is launched once at full size instead of T times
    at a fraction of it (a launch costs ~15 us before its first tile).  Only the recurrent head and the T NT-Xent
    launches stay sequential.  The random draws are made in the reference's order (per step: two action tensors, then
    lambda and the permutation of each view), so the sampled sub-bags are the ones the step-by-step loop would build."""
    B, K, dev, T_ = pack.B, pack.K, pack.feats.device, args.T
    acts, draws = [], []
    for t in range(T_):
        if injected is not None:
            acts += [a.to(dev) for a in injected["actions"][t]]
            draws += list(injected["draws"][t])
        else:
            acts += [torch.rand((B, K), device=dev) for _ in range(2)]                       # :235,256-258
            for _ in range(2):                                                               # mixup's draws (datasets.py:265-267)
                lam = args.alpha + torch.rand(size=(B, 1), device=dev) * (1 - args.alpha)
                draws.append((lam, torch.randperm(B, device=dev)))
    views, _ = subbag_views(pack, acts, args.feat_size, alpha=args.alpha, out_dtype=model.encoder.compute_dtype, draws=draws)
    outputs, _ = model(views)                                                                # 2*T*B bags, one batch
    losses, rewards, sim_last = [], [], None
    for t in range(T_):
        z = fc.forward_views(outputs[2 * t:2 * t + 2], restart=(t == 0))                      # :243,272
        if world > 1:
            loss_t, sim = mdist.gathered_nt_xent(z[0], z[1], args.temperature)               # global denominator (dist.py)
            losses.append(loss_t)
        else:
            losses.append(criterion(z[0], z[1]))                                             # :249,277
            sim = criterion.last_similarity
        if t > 0:
            rewards.append((sim_last - sim).view(1, -1))                                     # :282-283
        sim_last = sim
    loss = sum(losses) / T_                                                                  # :291
    optimizer.zero_grad()
    loss.backward(ops.unit_grad(loss))
    if world > 1:
        mdist.all_reduce_grads(optimizer.flat_grads())
    optimizer.step()                                                                         # :293-295
    return loss.detach(), [l.detach() for l in losses], rewards


# ------------------------------------------------------------------------------------------------ data
class SyntheticWSI:
    """Random slides in the WSIWithCluster output format: (feat [N,d] f32, clusters: K ascending id lists)."""

    def __init__(self, n_slides, n_patches, dim, num_clusters, seed=985):
        self.n, self.N, self.d, self.num_clusters, self.patch_dim = n_slides, n_patches, dim, num_clusters, dim
        self.rng = np.random.default_rng(seed)
        self.order = np.arange(n_slides)

    def __len__(self):
        return self.n

    def shuffle(self):
        self.rng.shuffle(self.order)

    def __getitem__(self, i):
        r = np.random.default_rng(1000 + int(self.order[i % self.n]))
        feat = (np.abs(r.standard_normal((self.N, self.d), dtype=np.float32)) * 0.5 * r.uniform(0.1, 1.9, (1, self.d))).astype(np.float32)
        lab = r.integers(0, self.num_clusters, self.N)
        return torch.from_numpy(feat), [np.nonzero(lab == k)[0].tolist() for k in range(self.num_clusters)], 0, str(i)


class WSIWithCluster:
    """Reader for the reference's dataset layout (utils/datasets.py:115-165): csv columns case_id,
    features_filepath, label, clusters_filepath, clusters_json_filepath; num_clusters from the csv name suffix."""

    def __init__(self, data_csv, indices=None, shuffle=False):
        import pandas as pd
        df = pd.read_csv(data_csv)
        if indices is not None:
            df = df[df["case_id"].isin(set(indices))]
        self.rows = df.reset_index(drop=True)
        self.num_clusters = int(Path(data_csv).stem.split("_")[-1])
        self.order = np.arange(len(self.rows))
        self.patch_dim = np.load(self.rows.loc[0, "features_filepath"])["img_features"].shape[-1]
        if shuffle:
            self.shuffle()

    def __len__(self):
        return len(self.rows)

    def shuffle(self):
        np.random.shuffle(self.order)

    def __getitem__(self, i):
        r = self.rows.loc[self.order[i % len(self.rows)]]
        feat = torch.from_numpy(np.load(r["features_filepath"])["img_features"].astype(np.float32))
        with open(r["clusters_json_filepath"]) as f:
            clusters = json.load(f)
        return feat, clusters, int(r["label"]), r["case_id"]


# ------------------------------------------------------------------------------------------------ driver
def train(args, train_set, model, fc, ppo, criterion, optimizer, device, rank, world):
    (model.eval(), fc.eval()) if args.train_stage == 2 else (model.train(), fc.train())
    memory_list = [rlmil.Memory(), rlmil.Memory()]
    best = float("inf")
    base_lrs = [g["lr"] for g in optimizer.param_groups] if optimizer else []
    store = None
    if not args.no_resident:
        # this rank's slides, uploaded once and kept in HBM for the whole run (SURVEY 8(e),(f)): a batch is an index list
        store = DeviceSlideStore.from_dataset(train_set, device, dtype=model.encoder.compute_dtype,
                                              indices=range(rank, len(train_set), world))
        if rank == 0:
            print(f"resident slide store: {len(store)} slides, {store.bytes() / 2 ** 30:.2f} GiB on {device}", flush=True)
    gc.collect()
    gc.freeze()          # models/optimizer state are long-lived: keep full collections (tens of ms) out of the step loop
    for epoch in range(args.epochs):
        last = float("nan")
        if store is not None:
            order = np.random.permutation(len(store))
            draws = len(store) * args.data_repeat
            for s in range(0, draws - args.batch_size + 1, args.batch_size):
                pack = store.pack(order[np.arange(s, s + args.batch_size) % len(store)])
                loss, _, _ = pretrain_step(args, model, fc, ppo, criterion, optimizer, pack, memory_list, world)
                last = loss.item()
        else:
            train_set.shuffle()
            feats, clusters = [], []
            for data_idx in range(rank, len(train_set) * args.data_repeat, world):       # bags shard by WSI across ranks
                feat, cluster, *_ = train_set[data_idx % len(train_set)]
                feats.append(feat.to(device, non_blocking=True))
                clusters.append(cluster)
                if len(feats) == args.batch_size:
                    pack = BagPack.from_lists(feats, clusters)
                    loss, _, _ = pretrain_step(args, model, fc, ppo, criterion, optimizer, pack, memory_list, world)
                    last = loss.item()
                    feats, clusters = [], []
        if optimizer is not None and epoch >= args.warmup:
            for g, b in zip(optimizer.param_groups, base_lrs):
                g["lr"] = cosine_lr(b, epoch, args.epochs, args.warmup)
        if rank == 0:
            is_best = last < best
            best = min(best, last)
            state = {"epoch": epoch + 1, "model_state_dict": model.state_dict(), "fc": fc.state_dict(),
                     "optimizer": None, "ppo_optimizer": None, "policy": ppo.policy.state_dict() if ppo else None}
            os.makedirs(args.save_dir, exist_ok=True)
            torch.save(state, os.path.join(args.save_dir, "checkpoint.pth.tar"))          # utils/general.py:207-211
            if is_best:
                torch.save(state, os.path.join(args.save_dir, "model_best.pth.tar"))
            print(f"epoch {epoch + 1}: loss {last:.4f} best {best:.4f}", flush=True)


def build_parser():
    p = argparse.ArgumentParser()       # flag names and defaults: train_MuRCL.py:386-477
    p.add_argument("--data_csv", type=str, default=None)
    p.add_argument("--data_split_json", type=str, default=None)
    p.add_argument("--synthetic", type=str, default=None, help="n_slides,n_patches (random bags instead of --data_csv)")
    p.add_argument("--feat_size", default=1024, type=int)
    p.add_argument("--T", default=6, type=int)
    p.add_argument("--no_batched_stage1", action="store_true",
                   help="stage 1: run the aggregator once per patch step like the reference instead of once per optimizer step")
    p.add_argument("--no_resident", action="store_true",
                   help="re-read and upload every slide on every step like the reference, instead of keeping the split in HBM")
    p.add_argument("--train_stage", default=1, type=int)
    p.add_argument("--checkpoint", default=None, type=str)
    p.add_argument("--optimizer", default="Adam", type=str)
    p.add_argument("--epochs", default=100, type=int)
    p.add_argument("--ppo_epochs", default=30, type=int)
    p.add_argument("--batch_size", default=128, type=int)
    p.add_argument("--backbone_lr", default=1e-4, type=float)
    p.add_argument("--fc_lr", default=5e-5, type=float)
    p.add_argument("--beta1", default=0.9, type=float)
    p.add_argument("--beta2", default=0.999, type=float)
    p.add_argument("--wdecay", default=1e-5, type=float)
    p.add_argument("--warmup", default=0, type=float)
    p.add_argument("--temperature", default=1.0, type=float)
    p.add_argument("--alpha", default=0.9, type=float)
    p.add_argument("--projection_dim", default=128, type=int)
    p.add_argument("--arch", default="ABMIL", type=str, choices=["ABMIL", "CLAM_SB"])
    p.add_argument("--model_dim", default=512, type=int)
    p.add_argument("--policy_hidden_dim", default=512, type=int)
    p.add_argument("--policy_conv", action="store_true", default=False)
    p.add_argument("--action_std", default=0.5, type=float)
    p.add_argument("--ppo_lr", default=1e-5, type=float)
    p.add_argument("--ppo_gamma", default=0.1, type=float)
    p.add_argument("--K_epochs", default=3, type=int)
    p.add_argument("--feature_num", default=512, type=int)
    p.add_argument("--fc_hidden_dim", default=1024, type=int)
    p.add_argument("--fc_rnn", action="store_true", default=True)
    p.add_argument("--D", default=128, type=int)
    p.add_argument("--dropout", default=0.0, type=float)
    p.add_argument("--size_arg", default="small", type=str)
    p.add_argument("--k_sample", default=8, type=int)
    p.add_argument("--data_repeat", default=10, type=int)
    p.add_argument("--num_clusters", default=10, type=int)
    p.add_argument("--save_dir", default="./results/murcl_amd/stage_1", type=str)
    p.add_argument("--seed", default=985, type=int)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if args.synthetic:
        n, N = (int(v) for v in args.synthetic.split(","))
        train_set = SyntheticWSI(n, N, 512, args.num_clusters, args.seed)
    else:
        idx = json.load(open(args.data_split_json))["train"] if args.data_split_json else None
        train_set = WSIWithCluster(args.data_csv, idx, shuffle=True)
        args.num_clusters = train_set.num_clusters
    model, fc, ppo = create_model(args, train_set.patch_dim, device)
    optimizer = get_optimizer(args, model, fc)
    criterion = NT_Xent(args.batch_size, args.temperature)
    train(args, train_set, model, fc, ppo, criterion, optimizer, device, rank, world)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
