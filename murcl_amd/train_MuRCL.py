#!/usr/bin/env python
"""MuRCL pre-training entry on MI355X (reference: train_MuRCL.py).

Keeps the reference's flags, three-stage schedule (1: contrastive warm-up with random sub-bags, 2: PPO only,
3: joint), model construction (``create_model``), optimizer / scheduler choices and checkpoint dictionary keys;
the per-batch hot step (train_MuRCL.py:233-304) is ``pretrain_step`` below and runs entirely on the HIP kernels:
device-side sub-bag selection + fused gather/mix-up, one batched aggregator call for both views, the recurrent
head, single-launch NT-Xent, PPO act/update.  One process per GPU: launch with
``python -m torch.distributed.run --nproc-per-node N -m murcl_amd.train_MuRCL --device 0,1,..,N-1 ...`` (instead of
DataParallel; a ``--device`` list shorter than N makes rank r take cuda:r).

Data: ``--data_csv`` in the reference's WSIWithCluster format (csv + npz ``img_features`` + json cluster lists),
or ``--synthetic B,N`` for random bags.
"""
import gc
import argparse
import json
import os
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from murcl_amd import dist as mdist, functional, ops
from murcl_amd.models import abmil, cl, clam, rlmil
from murcl_amd.optim import FlatAdam, FlatSGD, make_scheduler
from murcl_amd.utils import general as G
from murcl_amd.utils import checkpoint as C
from murcl_amd.utils.datasets import BagPack, DeviceSlideStore, draw_mixups, draw_step, subbag_views
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
        assert Path(args.checkpoint).exists(), f"{args.checkpoint} is not exist!"
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
    """train_MuRCL.py:154-171: Adam or SGD over two parameter groups (one flat buffer each); none in stage 2, which
    trains the sampler only and runs for ``--ppo_epochs``."""
    if args.train_stage == 2:
        args.epochs = args.ppo_epochs
        return None
    groups = [{"params": list(model.parameters()), "lr": args.backbone_lr},
              {"params": list(fc.parameters()), "lr": args.fc_lr}]
    if args.optimizer == "SGD":
        return FlatSGD(groups, momentum=args.momentum, nesterov=args.nesterov, weight_decay=args.wdecay)
    if args.optimizer == "Adam":
        return FlatAdam(groups, betas=(args.beta1, args.beta2), weight_decay=args.wdecay)
    raise NotImplementedError(args.optimizer)


def get_scheduler(args, optimizer):
    """train_MuRCL.py:174-186."""
    return make_scheduler(optimizer, args.scheduler, args.epochs, args.warmup)


# Which of two BUILT forms a sequential step takes (test hooks flipped through monkeypatch - tests/test_gpu_step.py - not environment
# switches: the per-step forms are what other shapes / stages run anyway, so both stay tested):
_DEFERRED_ENCODER = True       # one aggregator backward for all T patch steps of a sequential step (functional.EncoderSession)
_BATCHED_HEAD = True           # the recurrent head over all patch steps at once (Full_layer.forward_view_sequence)


# ------------------------------------------------------------------------------------------------ the hot step
def _shared_seed(args):
    """The next 63-bit seed of the run's SHARED random stream (``--global_mixup``): a CPU generator that every rank seeds with
    ``--seed`` and advances once per optimizer step, so that draws which must agree across ranks - the window positions and mix-up
    draws of the GLOBAL batch - come out identical everywhere without a collective."""
    gen = getattr(args, "_shared_gen", None)
    if gen is None:
        gen = args._shared_gen = torch.Generator().manual_seed(int(getattr(args, "seed", 985)) + 104729)
    return int(torch.empty((), dtype=torch.int64).random_(generator=gen))


def _gather_actions(acts, world):
    """[2] x [B_local, K] sampler actions of this rank -> [2] x [world * B_local, K] in rank order (one all-gather of both views):
    under batch-global mix-up a bag's partner may sit on another rank, and its sub-bag is cut by THAT bag's actions."""
    both = torch.stack([a.detach().float() for a in acts], 0).contiguous()            # [2, B, K]
    out = torch.empty((world,) + tuple(both.shape), dtype=torch.float32, device=both.device)
    mdist.all_gather_rows(out.view(world * both.shape[0] * both.shape[1], -1), both.view(both.shape[0] * both.shape[1], -1))
    g = out.permute(1, 0, 2, 3).reshape(2, world * both.shape[1], both.shape[2])        # [2, world * B, K]
    return [g[0].contiguous(), g[1].contiguous()]


def pretrain_step(args, model, fc, ppo, criterion, optimizer, pack, memory_list, world=1, injected=None, local=None):
    """One optimizer step on a batch of raw bags (train_MuRCL.py:233-304).

    ``pack``: BagPack of this rank's bags.  ``local`` = (lo, n) (``--global_mixup``, world > 1): ``pack`` is the GLOBAL batch - every
    rank keeps the whole cohort resident - and this rank trains on its bags [lo, lo + n) of it; window positions and mix-up draws are
    made for the global batch from the run's shared random stream (identical on every rank), the sampler's actions of stages 2 / 3
    are all-gathered, and a bag's mix-up partner is any bag of the global batch exactly as utils/datasets.py:263-271 permutes it.  ``injected`` (tests): dict replacing the random draws - 'actions'
    [T][2][B,K] (stage 1: every patch step; stages 2/3: only entry 0 is read, the later window positions come from the
    PPO sampler), 'draws' [T][2](lambda [B,1], perm [B]), and for stages 2/3 'eps' [T-1][2][B,K] ~ N(0,1), the sampler's
    Gaussian noise (rlmil.py:85-86); an optional 'trace' list receives the action tensors of every patch step.
    Returns (loss, losses[T], rewards[T-1])."""
    gB, K, dev = pack.B, pack.K, pack.feats.device            # gB: bags the draws / actions are made for (the global batch under ``local``)
    B = gB if local is None else int(local[1])                # bags this rank trains on
    dt_ = model.encoder.compute_dtype
    train_enc = args.train_stage != 2
    if args.train_stage == 1 and args.T > 1 and not getattr(args, "no_batched_stage1", False):
        return _pretrain_step_all_patch_steps_at_once(args, model, fc, criterion, optimizer, pack, injected, world, local)
    losses, rewards, sim_last, states, loss_vec, loss_mean = [], [], None, None, None, None
    late_head, agg_outs, agg_whole = _BATCHED_HEAD and getattr(fc, "fc_rnn", False), [], []
    # stage 3: the T aggregator passes stay sequential (the sampler needs step t's states for step t+1's windows) but share ONE
    # backward over all T * 2B bags (functional.EncoderSession); ABMIL's default shape in bf16 only
    enc, session = model.encoder, None
    d_feat = pack.feats.shape[1]
    if (_DEFERRED_ENCODER and train_enc and args.T > 1 and isinstance(enc, abmil.ABMIL) and enc.K == 1 and not (enc.training and enc.dropout > 0)
            and functional.abmil_fast_path(2 * B * args.feat_size, args.feat_size, d_feat, enc.L, enc.D, dt_)):
        session = enc.session = functional.EncoderSession(args.T, 2 * B, args.feat_size, d_feat, enc.L, dt_, dev)
    if injected is None:
        # every random number of the step in four launches (uniform window positions, mix-up draws, the sampler's Gaussian
        # noise) instead of ~10 tiny launches per view and patch step; none of them depends on anything computed in the step
        rl = args.train_stage != 1
        if local is None:
            acts_u, noise, mix = draw_step(dev, (1 if rl else args.T, 2, B, K),                  # :235,256-258
                                           (args.T - 1, 2, B, K) if rl and args.T > 1 else None,  # rlmil.py:85-86
                                           2 * args.T, B, args.alpha)                             # datasets.py:265-267
        else:           # global batch: positions + mix-up draws from the shared stream, the sampler's noise (this rank's rows) from its own
            acts_u, _, mix = draw_step(dev, (1 if rl else args.T, 2, gB, K), None, 2 * args.T, gB, args.alpha, seed=_shared_seed(args))
            noise = draw_step(dev, None, (args.T - 1, 2, B, K), 0, 0, args.alpha)[1] if rl and args.T > 1 else None
    for t in range(args.T):
        if t == 0 or args.train_stage == 1:
            acts = [a.to(dev) for a in injected["actions"][t]] if injected is not None else acts_u[t]      # [2,B,K]: one launch each
        else:
            eps = [noise[t - 1, 0], noise[t - 1, 1]] if injected is None else [e.to(dev) for e in injected["eps"][t - 1]]
            acts = ppo.select_actions(states, memory_list, restart_batch=(t == 1), eps=eps)  # :259-265, both views in one policy step
            if local is not None:
                acts = _gather_actions(acts, world)                                          # the partners' windows: their ranks' actions
        if injected is not None and injected.get("trace") is not None:
            injected["trace"].append([a.detach().clone() for a in acts])
        views, _ = subbag_views(pack, acts, args.feat_size, alpha=args.alpha, out_dtype=dt_,
                                draws=mix[2 * t:2 * t + 2] if injected is None else injected["draws"][t],
                                out=None if session is None else session.views(t), local=local)           # :237-239,266-269
        with torch.set_grad_enabled(train_enc):
            outputs, states = model(views)                                                   # :242,271
            if late_head:
                # the sampler picks the next windows from the aggregator's states; what the head and the loss produce (the
                # loss itself, the rewards) is needed after the last patch step only: they run once, below
                agg_outs += list(outputs)
                agg_whole.append(getattr(model, "last_whole", None))
                continue
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
    if late_head:
        with torch.set_grad_enabled(train_enc):
            # (a training session keeps every step's aggregator output in one buffer: the head reads that buffer, no concatenation)
            x_whole = functional.session_whole(session, agg_whole) if session is not None and all(h is not None for h in agg_whole) else None
            z_all = fc.forward_view_sequence(agg_outs, whole=x_whole).view(args.T, 2, B, -1)   # :243,272 for every step at once
            if world == 1 and 2 * B <= 128:
                loss_t, sims = criterion.forward_steps(z_all.view(args.T, 2 * B, -1))        # :249,277: one launch
                loss_vec, losses, loss_mean = loss_t, list(loss_t.unbind(0)), criterion.last_mean
                rewards = list(ops.axpby(sims[:-1], sims[1:], 1.0, -1.0).unsqueeze(1).unbind(0))   # :282-283, all steps in one launch
                for m in memory_list:
                    m.rewards.extend(rewards)
            for t in range(args.T if not losses else 0):
                if world > 1:
                    loss, sim = mdist.gathered_nt_xent(z_all[t, 0], z_all[t, 1], args.temperature)
                else:
                    loss = criterion.forward_stacked(z_all[t].reshape(2 * B, -1))            # :249,277
                    sim = criterion.last_similarity
                losses.append(loss)
                if t > 0:
                    reward = (sim_last - sim).view(1, -1)                                    # :282-283
                    rewards.append(reward)
                    for m in memory_list:
                        m.rewards.append(reward)
                sim_last = sim
    # :291 - the mean of the [T] loss vector is one launch (and one in the backward); sum(list) / T was T + 1 (and T + 1 back)
    if loss_vec is not None and loss_mean is not None:
        loss = loss_mean                                      # (the mean came out of the NT-Xent node: no mean node in the graph)
    elif loss_vec is not None and not loss_vec.requires_grad and loss_vec.is_cuda:
        loss = ops.mean_small(loss_vec)                       # (stage 2: nothing differentiates it)
    else:
        loss = loss_vec.mean() if loss_vec is not None else sum(losses) / args.T
    enc.session = None
    if hasattr(model, "last_whole"):
        model.last_whole = None                               # (do not keep the last patch step's graph alive past the step)
    if train_enc:
        optimizer.zero_grad()
        with functional.deferred_wgrads():          # the head's T weight gradients per parameter as one product each
            loss.backward(ops.unit_grad(loss))
        # the session's ONE aggregator backward runs when autograd has reached the node of every patch step: a step whose
        # aggregator output never reached the loss would silently drop every encoder gradient of the step
        assert session is None or session.pending == 0, \
            f"EncoderSession: {session.pending} of {session.t} patch steps were not reached by backward - encoder gradients not computed"
        if world > 1:
            mdist.all_reduce_grads(optimizer.flat_grads())
        optimizer.step()                                                                     # :293-295
    else:
        for m in memory_list:
            ppo.update(m)                                                                    # :297-298
    if injected is not None and injected.get("trace") is not None and memory_list[0].logprobs:
        injected["trace"].append({"logprobs": [torch.stack(m.logprobs, 0) for m in memory_list]})
    for m in memory_list:
        m.clear_memory()
    return loss.detach(), [l.detach() for l in losses], rewards


def _pretrain_step_all_patch_steps_at_once(args, model, fc, criterion, optimizer, pack, injected=None, world=1, local=None):
    """Stage 1 draws every patch step's window positions at random (train_MuRCL.py:235,256-258): no step depends on the
    aggregator states of the step before, so the sub-bags of all T steps are built into ONE buffer and the aggregator
    runs ONCE over 2*T*B bags - each weight-stationary / wgrad kernel is launched once at full size instead of T times
    at a fraction of it (a launch costs ~15 us before its first tile).  Only the recurrent head and the T NT-Xent
    launches stay sequential.  The random draws of the whole step are made up front in a few launches
    (``datasets.draw_mixups``); injected draws keep the reference's per-step order."""
    gB, K, dev, T_ = pack.B, pack.K, pack.feats.device, args.T
    B = gB if local is None else int(local[1])
    acts, draws = [], []
    if injected is not None:
        for t in range(T_):
            acts += [a.to(dev) for a in injected["actions"][t]]
            draws += list(injected["draws"][t])
    else:
        acts, _, draws = draw_step(dev, (2 * T_, gB, K), None, 2 * T_, gB, args.alpha,       # :235,256-258; datasets.py:265-267
                                   seed=None if local is None else _shared_seed(args))
    views, _ = subbag_views(pack, acts, args.feat_size, alpha=args.alpha, out_dtype=model.encoder.compute_dtype, draws=draws, local=local)
    outputs, _ = model(views)                                                                # 2*T*B bags, one batch
    losses, rewards, sim_last = [], [], None
    z_all = fc.forward_view_sequence(outputs).view(T_, 2, B, -1) if _BATCHED_HEAD and fc.fc_rnn else None   # :243,272, all steps
    if z_all is not None and world == 1 and 2 * B <= 128:
        loss_t, sims = criterion.forward_steps(z_all.view(T_, 2 * B, -1))                     # :249,277 for all steps: one launch
        loss = criterion.last_mean if criterion.last_mean is not None else loss_t.mean()      # :291 (an output of the NT-Xent node)
        optimizer.zero_grad()
        with functional.deferred_wgrads():
            loss.backward(ops.unit_grad(loss))
        optimizer.step()                                                                      # :293-295
        return loss.detach(), list(loss_t.detach().unbind(0)), list(ops.axpby(sims[:-1], sims[1:], 1.0, -1.0).unsqueeze(1).unbind(0))   # :282-283
    for t in range(T_):
        z = (z_all[t, 0], z_all[t, 1]) if z_all is not None else fc.forward_views(outputs[2 * t:2 * t + 2], restart=(t == 0))
        if world > 1:
            loss_t, sim = mdist.gathered_nt_xent(z[0], z[1], args.temperature)               # global denominator (dist.py)
            losses.append(loss_t)
        else:
            losses.append(criterion.forward_stacked(z_all[t].reshape(2 * B, -1)) if z_all is not None
                          else criterion(z[0], z[1]))                                        # :249,277
            sim = criterion.last_similarity
        if t > 0:
            rewards.append((sim_last - sim).view(1, -1))                                     # :282-283
        sim_last = sim
    loss = sum(losses) / T_                                                                  # :291
    optimizer.zero_grad()
    with functional.deferred_wgrads():              # the head's T weight gradients per parameter as one product each
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
def shard_slides(n_slides, rank, world):
    """Slides of rank ``rank``: ``rank, rank + world, ...`` trimmed so that EVERY rank owns the same number
    (``n_slides // world``) - ranks then run the same number of steps per epoch and their collectives pair up."""
    per = n_slides // world
    if per == 0:
        raise ValueError(f"{n_slides} slides cannot be sharded over {world} ranks")
    return list(range(rank, per * world, world))


def train(args, train_set, model, fc, ppo, criterion, optimizer, scheduler, device, rank, world, tb_writer=None):
    """Epoch loop of train_MuRCL.py:189-343: per-epoch shuffle, batches of ``--batch_size`` bags (per rank), the hot step,
    scheduler after ``--warmup`` epochs, best-of-run by the epoch average of the last patch step's loss (:315-321),
    checkpoint every epoch (:322-330), csv logs, early stop."""
    save_dir = Path(args.save_dir)
    (model.eval(), fc.eval()) if args.train_stage == 2 else (model.train(), fc.train())
    memory_list = [rlmil.Memory(), rlmil.Memory()]
    best = G.Best("min")
    losses_csv = results_csv = None
    if rank == 0:
        losses_csv = G.CsvLog(save_dir / "losses.csv", ["epoch", "train", "best_epoch", "best_train"])
        results_csv = G.CsvLog(save_dir / "results.csv", ["epoch", "final_epoch", "final_loss"])
    early_stop = G.EarlyStop(args.patience) if args.patience is not None else None
    mine = shard_slides(len(train_set), rank, world)
    store = None
    # --global_mixup (world > 1): every rank keeps the WHOLE split resident (C4's 512 slides are 4.3 GB in bf16 of 288), the epoch
    # order is one shared permutation, step s takes the global batch order[s * world * B : (s + 1) * world * B] and this rank trains
    # on its B bags of it - so that a bag's mix-up partner is any bag of the global batch, as the reference's one-process
    # DataParallel step permutes it (utils/datasets.py:263-271, train_MuRCL.py:145); default: rank-local partners
    global_mix = bool(getattr(args, "global_mixup", False)) and world > 1
    if global_mix and args.no_resident:
        raise ValueError("--global_mixup needs the resident slide store (drop --no_resident)")
    if global_mix:
        store = DeviceSlideStore.from_dataset(train_set, device, dtype=model.encoder.compute_dtype)
        if rank == 0:
            print(f"resident slide store (whole split on every rank, --global_mixup): {len(store)} slides, "
                  f"{store.bytes() / 2 ** 30:.2f} GiB on {device}", flush=True)
    elif not args.no_resident:
        # this rank's slides, uploaded once and kept in HBM for the whole run (SURVEY 8(e),(f)): a batch is an index list.
        # (--preload asks the reference to keep the split in host memory; the resident store subsumes it.)
        store = DeviceSlideStore.from_dataset(train_set, device, dtype=model.encoder.compute_dtype, indices=mine)
        if rank == 0:
            print(f"resident slide store: {len(store)} slides, {store.bytes() / 2 ** 30:.2f} GiB on {device}", flush=True)
    steps_per_epoch = len(mine) * args.data_repeat // args.batch_size          # identical on every rank by construction
    writer = C.CheckpointWriter() if rank == 0 else None                      # the per-epoch files are written off this thread
    # Epoch boundaries without a queue drain (utils.checkpoint.EpochSnapshots): state and loss are captured in stream order and
    # reach the host behind the next epoch's first steps; the bookkeeping below then runs a few steps late.  Early stopping
    # (--patience, as runs/pretrain.sh uses it) is decided from the same late loss: the steps of the following epoch that were
    # already taken are discarded - nothing of that epoch is logged or saved and the modules go back to the snapshot - so the
    # files of a run are those of the synchronous loop.  Several ranks would have to agree on the step at which they stop:
    # with --patience they keep the synchronous boundary.
    deferred = device.type == "cuda" and os.environ.get("MURCL_SYNC_EPOCH_END") != "1" and (early_stop is None or world == 1)
    snaps = C.EpochSnapshots(device) if (deferred and rank == 0) else None
    stopped = []                                                              # [state of the epoch after which training stops]

    def finish_epoch(ep, train_loss, state):                                  # rank 0: :315-330 for epoch `ep` (1-based)
        if stopped:
            return                                                            # a later epoch of a run that has already stopped
        if tb_writer is not None:
            tb_writer.add_scalar("train/1.train_loss", train_loss, ep - 1)
        is_best = best.compare(train_loss, ep, inplace=True)
        writer.submit(state, is_best, str(save_dir))                          # :322-330
        losses_csv.write_row([ep, train_loss, best.epoch, best.best])
        results_csv.write_row([ep, best.epoch, best.best])
        print(f"Loss: {train_loss:.4f}, Best: {best.best:.4f}, Epoch: {best.epoch:2}\n", flush=True)
        if early_stop is not None:
            early_stop.update(best.best)
            if early_stop.is_stop():
                stopped.append(state)

    def drain_snapshots(block=False):
        if snaps is not None:
            for done in snaps.poll(block):
                finish_epoch(*done)
        return bool(stopped)
    gc.collect()
    gc.freeze()          # models/optimizer state are long-lived: keep full collections (tens of ms) out of the step loop
    for epoch in range(args.epochs):
        last_step = []                                                        # loss of patch step T-1, one entry per batch
        if rank == 0 and optimizer is not None:
            print(f"Training Stage: {args.train_stage}, lr: " + ", ".join(f"group[{k}]: {g['lr']}" for k, g in enumerate(optimizer.param_groups)), flush=True)
        if global_mix:
            order = np.random.default_rng(int(args.seed) * 7919 + epoch).permutation(len(store))       # the same on every rank
            gbs = args.batch_size * world
            for it in range(steps_per_epoch):
                pack = store.pack(order[np.arange(it * gbs, (it + 1) * gbs) % len(store)])             # the GLOBAL batch
                _, ls, _ = pretrain_step(args, model, fc, ppo, criterion, optimizer, pack, memory_list, world,
                                         local=(rank * args.batch_size, args.batch_size))
                last_step.append(ls[-1])
                if drain_snapshots():
                    break
        elif store is not None:
            order = np.random.permutation(len(store))
            for it in range(steps_per_epoch):
                s = it * args.batch_size
                pack = store.pack(order[np.arange(s, s + args.batch_size) % len(store)])
                _, ls, _ = pretrain_step(args, model, fc, ppo, criterion, optimizer, pack, memory_list, world)
                last_step.append(ls[-1])
                if drain_snapshots():
                    break
        else:
            train_set.shuffle()
            feats, clusters = [], []
            for it in range(steps_per_epoch * args.batch_size):                # bags shard by WSI across ranks
                feat, cluster, *_ = train_set[mine[it % len(mine)]]
                feats.append(feat.to(device, non_blocking=True))
                clusters.append(cluster)
                if len(feats) == args.batch_size:
                    pack = BagPack.from_lists(feats, clusters)
                    _, ls, _ = pretrain_step(args, model, fc, ppo, criterion, optimizer, pack, memory_list, world)
                    last_step.append(ls[-1])
                    feats, clusters = [], []
                    if drain_snapshots():
                        break
        if stopped or drain_snapshots():
            break                                   # decided from an earlier epoch's loss: this epoch's steps are discarded
        if scheduler is not None and epoch >= args.warmup:
            scheduler.step()                                                   # :312-313
        if deferred and last_step:
            if snaps is not None:                                              # losses[-1].avg (:315), read back later
                snaps.capture(epoch + 1, torch.stack(last_step).mean(), model, fc, optimizer, ppo)
            continue                                # (other ranks: nothing needs the loss on the host without early stopping)
        train_loss = torch.stack(last_step).mean().item() if last_step else float("nan")   # losses[-1].avg (:315)
        if rank == 0:
            finish_epoch(epoch + 1, train_loss, C.make_state(epoch + 1, model, fc, optimizer, ppo))
            if stopped:
                break
        else:
            best.compare(train_loss, epoch + 1, inplace=True)                  # every rank sees the same (global) loss
            if early_stop is not None:
                early_stop.update(best.best)
                if early_stop.is_stop():
                    break
    drain_snapshots(block=True)
    if stopped and deferred:                        # the modules as they were at the end of the last epoch that counts
        st = stopped[0]
        model.load_state_dict(st["model_state_dict"])
        fc.load_state_dict(st["fc"])
        if ppo is not None and st.get("policy") is not None:
            ppo.policy.load_state_dict(st["policy"])
            ppo.policy_old.load_state_dict(st["policy"])
    if writer is not None:
        writer.close()                                                         # every file is complete when train() returns
    if tb_writer is not None:
        tb_writer.close()


def build_parser():
    """Every flag of the reference's parser with its type, default, choices and action (train_MuRCL.py:386-477; pinned by
    tests/golden/g13_cli_flags.json), plus the murcl_amd extras at the end."""
    p = argparse.ArgumentParser()
    # Data
    p.add_argument("--dataset", type=str, default="Camelyon16", help="dataset name (only names the result directory)")
    p.add_argument("--data_csv", type=str, default="", help="the .csv filepath used")
    p.add_argument("--data_split_json", type=str, default="/path/to/data_split.json")
    p.add_argument("--preload", action="store_true", default=False,
                   help="preload the patch features (murcl_amd keeps the split resident in HBM either way)")
    p.add_argument("--data_repeat", type=int, default=10)
    p.add_argument("--feat_size", default=1024, type=int)
    # Train
    p.add_argument("--train_stage", default=1, type=int, help="1: warm-up, 2: learn to select patches with RL, 3: joint")
    p.add_argument("--T", default=6, type=int)
    p.add_argument("--optimizer", type=str, default="Adam", choices=["Adam", "SGD"])
    p.add_argument("--scheduler", type=str, default=None, choices=[None, "StepLR", "CosineAnnealingLR"])
    p.add_argument("--batch_size", type=int, default=128, help="bags per step (per GPU when launched with several ranks)")
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--ppo_epochs", type=int, default=30)
    p.add_argument("--backbone_lr", default=1e-4, type=float)
    p.add_argument("--fc_lr", default=1e-4, type=float)
    p.add_argument("--temperature", type=float, default=1.0)
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--nesterov", action="store_true", default=True)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--beta2", type=float, default=0.999)
    p.add_argument("--warmup", default=0, type=float)
    p.add_argument("--wdecay", default=1e-5, type=float)
    p.add_argument("--patience", type=int, default=None)
    # Architecture
    p.add_argument("--checkpoint", default=None, type=str)
    p.add_argument("--arch", default="CLAM_SB", type=str, choices=["ABMIL", "CLAM_SB"])
    p.add_argument("--alpha", type=float, default=0.9)
    p.add_argument("--projection_dim", type=int, default=128)
    p.add_argument("--model_dim", type=int, default=512)
    p.add_argument("--policy_hidden_dim", type=int, default=512)
    p.add_argument("--policy_conv", action="store_true", default=False)
    p.add_argument("--action_std", type=float, default=0.5)
    p.add_argument("--ppo_lr", type=float, default=0.00001)
    p.add_argument("--ppo_gamma", type=float, default=0.1)
    p.add_argument("--K_epochs", type=int, default=3)
    p.add_argument("--feature_num", type=int, default=512)
    p.add_argument("--fc_hidden_dim", type=int, default=1024)
    p.add_argument("--fc_rnn", action="store_true", default=True)
    p.add_argument("--D", type=int, default=128)
    p.add_argument("--dropout", type=float, default=0.0)
    p.add_argument("--size_arg", type=str, default="small", choices=["small", "big"])
    p.add_argument("--k_sample", type=int, default=8)
    p.add_argument("--use_tensorboard", action="store_true", default=False)
    # Save
    p.add_argument("--base_save_dir", type=str, default="./results")
    p.add_argument("--save_dir", type=str, default=None)
    p.add_argument("--save_dir_flag", type=str, default=None)
    p.add_argument("--exist_ok", action="store_true", default=False)
    # Global
    p.add_argument("--device", default="3", help="cuda device, i.e. 0 or 0,1,2,3 (one process per GPU takes its LOCAL_RANK-th entry)")
    p.add_argument("--seed", type=int, default=985)
    # murcl_amd extras
    x = p.add_argument_group("murcl_amd")
    x.add_argument("--synthetic", type=str, default=None, help="n_slides,n_patches (random bags instead of --data_csv)")
    x.add_argument("--num_clusters", default=10, type=int, help="clusters per slide for --synthetic (else read from the csv name)")
    x.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="storage type of patch-level tensors")
    x.add_argument("--no_batched_stage1", action="store_true",
                   help="stage 1: run the aggregator once per patch step like the reference instead of once per optimizer step")
    x.add_argument("--no_resident", action="store_true",
                   help="re-read and upload every slide on every step like the reference, instead of keeping the split in HBM")
    x.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"],
                   help="several ranks: nccl (= RCCL over xGMI, one GPU per rank) or gloo (debugging: ranks may share a GPU, "
                        "collectives are staged through host memory)")
    x.add_argument("--global_mixup", action="store_true",
                   help="several ranks: mix-up partners from the whole GLOBAL batch, as the reference's one-process step permutes it "
                        "(every rank then keeps the whole split resident; default: partners from the rank's own bags)")
    return p


def run(args):
    """train_MuRCL.py:346-383."""
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    G.init_seeds(args.seed)
    if rank == 0:
        G.prepare_run_dir(args, "MuRCL")
    device = G.pick_device(args.device, local)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from . import dist as mdist
        mdist.cap_rccl_channels(world)                     # RCCL's channel workgroups must fit the CUs the step leaves free
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:                                                  # gloo: several ranks on ONE device (tests); collectives staged through the host
            dist.init_process_group(args.dist_backend)
        box = [args.save_dir]
        dist.broadcast_object_list(box, src=0)                 # rank 0 resolved (and possibly incremented) the directory
        args.save_dir = box[0]
    if args.synthetic:
        n, N = (int(v) for v in args.synthetic.split(","))
        train_set = SyntheticWSI(n, N, 512, args.num_clusters, args.seed)
    else:
        idx = json.load(open(args.data_split_json))["train"]
        train_set = WSIWithCluster(args.data_csv, idx, shuffle=True)
        args.num_clusters = train_set.num_clusters
    args.num_data = len(train_set) * args.data_repeat
    args.eval_step = int(args.num_data / args.batch_size)
    model, fc, ppo = create_model(args, train_set.patch_dim, device)
    criterion = NT_Xent(args.batch_size, args.temperature)
    optimizer = get_optimizer(args, model, fc)
    scheduler = get_scheduler(args, optimizer)
    if world > 1:
        # parameters (and the dataset order) come from the common seed above, so the replicas start identical; the SAMPLING
        # streams - window positions, mix-up draws, the sampler's noise - must differ per rank, or every rank would draw the
        # same numbers for its bags where the reference (one process, B_global bags) draws independent ones
        torch.manual_seed(args.seed + 7919 * (rank + 1))
    tb_writer = None
    if rank == 0:
        G.dump_args(args, args.save_dir)
        tb_writer = G.tensorboard_writer(args.save_dir, args.use_tensorboard)
    train(args, train_set, model, fc, ppo, criterion, optimizer, scheduler, device, rank, world, tb_writer)
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
