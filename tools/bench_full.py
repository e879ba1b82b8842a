#!/usr/bin/env python
"""M-full: the whole MuRCL hot step (train_MuRCL.py:233-304, stage 1) on synthetic raw bags:
T patch-steps x 2 views of device-side sub-bag selection + gather/mix-up, aggregator, head, NT-Xent, one backward,
Adam.  Dev/measurement tool (the headline metric is bench.py)."""
import argparse, os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
from murcl_amd.models import rlmil
from murcl_amd.train_MuRCL import build_parser, create_model, get_optimizer, pretrain_step
from murcl_amd.utils.datasets import BagPack
from murcl_amd.utils.losses import NT_Xent

ap = argparse.ArgumentParser()
ap.add_argument("--bags", type=int, default=64)
ap.add_argument("--raw", type=int, default=8192)
ap.add_argument("--feat_size", type=int, default=1024)
ap.add_argument("--T", type=int, default=6)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--stage", type=int, default=1, help="1: MIL + contrastive head; 2: PPO sampler trained, encoder frozen; 3: both")
ap.add_argument("--force-dist", action="store_true", help="single-rank RCCL group, the multi-GPU code path (world=2 semantics)")
ap.add_argument("--cprofile", action="store_true", help="print the host-side cProfile of the timed steps")
a = ap.parse_args()
dev = torch.device("cuda:0")
WORLD = 1
if a.force_dist:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    WORLD = 2
args = build_parser().parse_args(["--arch", os.environ.get("BF_ARCH", "ABMIL"), "--fc_lr", "5e-5"])
args.T, args.feat_size, args.batch_size, args.dtype, args.train_stage, args.num_clusters = a.T, a.feat_size, a.bags, a.dtype, 1, 10
torch.manual_seed(985)
model, fc, ppo = create_model(args, 512, dev)          # stage-1 construction (no checkpoint needed) ...
opt = get_optimizer(args, model, fc)
if a.stage in (2, 3):                                  # ... then switch the step to the requested stage with a fresh sampler
    args.train_stage = a.stage
    ppo = rlmil.PPO(512, args.model_dim, args.policy_hidden_dim, args.policy_conv, action_std=args.action_std,
                    lr=args.ppo_lr, gamma=args.ppo_gamma, K_epochs=args.K_epochs, action_size=args.num_clusters)
g = torch.Generator(device=dev); g.manual_seed(1)
feats = [(torch.randn((a.raw, 512), generator=g, device=dev).abs() * 0.5) for _ in range(a.bags)]
rng = np.random.default_rng(985)
clusters = []
for _ in range(a.bags):
    lab = rng.integers(0, 10, a.raw)
    clusters.append([np.nonzero(lab == k)[0].tolist() for k in range(10)])
pack = BagPack.from_lists(feats, clusters, dtype=torch.bfloat16 if a.dtype == "bf16" else None)
crit = NT_Xent(a.bags, 1.0)
mem = [rlmil.Memory(), rlmil.Memory()]
for _ in range(3):
    pretrain_step(args, model, fc, ppo, crit, opt, pack, mem, WORLD)
torch.cuda.synchronize()
ops.TIMERS = None
if a.cprofile:
    import cProfile, pstats
    prof = cProfile.Profile(); prof.enable()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss, _, _ = pretrain_step(args, model, fc, ppo, crit, opt, pack, mem, WORLD)
host = (time.perf_counter() - t0) / a.steps
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
if a.cprofile:
    prof.disable()
    pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
# time the sub-bag builder alone
from murcl_amd.utils.datasets import subbag_views
acts = [torch.rand((a.bags, 10), device=dev) for _ in range(2)]
for _ in range(3): subbag_views(pack, acts, a.feat_size, alpha=0.9, out_dtype=model.encoder.compute_dtype)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(20): subbag_views(pack, acts, a.feat_size, alpha=0.9, out_dtype=model.encoder.compute_dtype)
torch.cuda.synchronize(); sb = (time.perf_counter() - t1) / 20
out_bytes = 2 * a.bags * a.feat_size * 512 * (2 if a.dtype == "bf16" else 4)
print(json.dumps({"workload": f"M-full stage-{a.stage} step: {a.bags} raw bags x {a.raw} x 512 -> T={a.T} x 2 views of {a.feat_size}", "dtype": a.dtype,
                  "ms_per_step": round(dt * 1e3, 3), "host_enqueue_ms": round(host * 1e3, 3), "bags_per_s": round(a.bags / dt, 1), "loss": round(loss.item(), 5),
                  "subbag_two_views_ms": round(sb * 1e3, 4), "subbag_GBps_out+2in": round(3 * out_bytes / sb / 1e9, 1)}))
