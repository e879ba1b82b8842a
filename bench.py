#!/usr/bin/env python
"""MuRCL hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

Workload (BASELINE.json configs[1]): ABMIL + NT-Xent pre-train step on one view pair,
64 bags x 2048 patches x 512-d per GPU, bf16 patch-level tensors with f32 accumulation.
A step = CL(ABMIL) forward on both views (128 bag-forwards) -> Full_layer GRU step + projection
-> NT-Xent -> full backward -> Adam on model + head.  Inputs are resident in HBM before the
timed region.  With N > 1 GPUs each rank owns 64 bags (weak scaling); embeddings are
all-gathered for the global contrastive denominator and gradients all-reduced (RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"hbm_GBps": 8000.0, "mfma_bf16_TFLOPs": 2500.0, "mfma_f32_TFLOPs": 157.3}   # MI355X_MICROARCH.md


def build(dtype, device, bags):
    from murcl_amd.models.abmil import ABMIL
    from murcl_amd.models.cl import CL
    from murcl_amd.models.rlmil import Full_layer
    from murcl_amd.optim import FlatAdam
    from murcl_amd.utils.losses import NT_Xent
    torch.manual_seed(985)                                   # reference seed (train_MuRCL.py:473)
    enc = ABMIL(512, L=512, D=128, dim_out=128)
    enc.compute_dtype = dtype
    model = CL(enc, projection_dim=128, n_features=512).to(device)
    fc = Full_layer(512, 1024, True, 128).to(device)
    opt = FlatAdam([{"params": list(model.parameters()), "lr": 1e-4},
                    {"params": list(fc.parameters()), "lr": 5e-5}], betas=(0.9, 0.999), weight_decay=1e-5)
    crit = NT_Xent(bags, 1.0)
    return model, fc, opt, crit


def synth_views(bags, n, d, dtype, device, rank):
    """Two views of `bags` synthetic slides: |N(0,1)|*0.5 with a per-slide feature signature."""
    g = torch.Generator(device=device)
    g.manual_seed(985 + rank)
    sig = torch.rand((bags, 1, d), generator=g, device=device) * 1.8 + 0.1
    both = torch.empty((2 * bags, n, d), dtype=dtype, device=device)     # the two views back to back
    for v in range(2):
        x = torch.randn((bags, n, d), generator=g, device=device).abs_().mul_(0.5).mul_(sig)
        both[v * bags:(v + 1) * bags].copy_(x)
    return [both[:bags], both[bags:]]


def make_step(model, fc, opt, crit, views, world):
    from murcl_amd import dist as mdist, functional, ops
    # milestones (encoder gradients reduced layer by layer under the remaining backward): opt-in - three more collectives
    # cost 85 us per step on one rank (1.83 vs 1.75 ms under MURCL_FORCE_DIST=1), more than the ~60 us of exposed
    # all-reduce they can hide; to be re-measured on a real multi-GPU node
    reducer = mdist.OverlappedGradReduce(opt, early_groups=(1,), milestones=os.environ.get("MURCL_MILESTONES") == "1") \
        if world > 1 else None

    def step():
        opt.zero_grad()
        outs, _ = model(views)
        if reducer is not None:
            reducer.arm(outs)              # head-gradient all-reduce starts when backward reaches the aggregator
        z = fc.forward_views(outs, restart=True)
        if world > 1:
            loss, _ = mdist.gathered_nt_xent(z[0], z[1], 1.0)
        else:
            loss = crit(z[0], z[1])
        with functional.deferred_wgrads():      # as train_MuRCL.pretrain_step: the head's weight gradients as one launch
            loss.backward(ops.unit_grad(loss))
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss
    return step


_ONES = {}


def _ones_like(t):
    """A cached tensor of ones with t's shape / dtype / device (the upstream gradient of a .sum() loss)."""
    k = (tuple(t.shape), t.dtype, t.device)
    if k not in _ONES:
        _ONES[k] = torch.ones_like(t)
    return _ONES[k]


def cpu_baseline(bags, n, d, budget_s=14.0):
    """The CPU oracle (a port of the reference step) timed on this host beside the GPU number: with all
    cores (the fastest of a few probed thread-pool sizes) and with one thread - the reference's own setting
    (train_MuRCL.py:484) - both on the FULL headline input (64 bags x 2 views); ~25 s of CPU work in all."""
    import os as _os
    from oracle import mil_oracle as O, params as P        # the checker, timed here as the CPU baseline - nowhere else

    def _cpu_step_timer(sample, n, d, threads, budget_s, max_steps=50, warm_sample=None):
        """Time the CPU oracle's step (oracle/mil_oracle.py: the reference's ABMIL + Full_layer + NT-Xent view-pair step with
        backward and Adam, fp32) on ``sample`` bags per view with ``threads`` torch threads: one untimed step (on
        ``warm_sample`` bags when given: thread pool / allocator warm-up without paying a full step), then timed steps until
        ``budget_s`` is used (at least one)."""
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        mp = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.abmil(985)).items()}
        fp = {k: v.clone().requires_grad_() for k, v in P.to_torch(P.full_layer(985)).items()}
        full_xs = [torch.randn(sample, n, d).abs_() * 0.5 for _ in range(2)]
        xs = full_xs if warm_sample is None else [x[:warm_sample] for x in full_xs]
        st_m, st_f = {}, {}

        def one():
            nonlocal xs
            for p in list(mp.values()) + list(fp.values()):
                p.grad = None
            loss, *_ = O.pretrain_step(mp, fp, [xs], 1.0)
            loss.backward()
            gm = {k: v.grad for k, v in mp.items() if v.grad is not None}
            gf = {k: v.grad for k, v in fp.items() if v.grad is not None}
            with torch.no_grad():
                newm = O.adam_step({k: mp[k].detach() for k in gm}, gm, st_m, 1e-4, weight_decay=1e-5)
                newf = O.adam_step({k: fp[k].detach() for k in gf}, gf, st_f, 5e-5, weight_decay=1e-5)
                for k, v in newm.items():
                    mp[k].copy_(v)
                for k, v in newf.items():
                    fp[k].copy_(v)
        t0 = time.time()
        one()                                                   # untimed: allocator / thread-pool warm-up
        first = time.time() - t0
        xs = full_xs
        t0, k = time.time(), 0
        while k < 1 or (time.time() - t0 + first < budget_s and k < max_steps):
            one()
            k += 1
        return (time.time() - t0) / k, k

    cores = _os.cpu_count() or torch.get_num_threads()
    # torch's CPU kernels stop scaling (and regress) long before 256 SMT threads on this kind of host: probe a few pool
    # sizes on a small sample and run the full input with the fastest ("all cores" = the best the host does)
    probe = {}
    for t in sorted({min(cores, c) for c in (8, 16, 32, 64)}):        # (128 and 256 threads measured 5.7 and 0.25 bags/s here)
        probe[t] = _cpu_step_timer(8, n, d, t, 0.0, max_steps=1)[0]
    best = min(probe, key=probe.get)
    full = probe[best] * bags / 8 <= budget_s              # projected full-input step fits the budget
    sample = bags if full else 16
    dt_all, k_all = _cpu_step_timer(sample, n, d, best, budget_s, warm_sample=8)
    # one thread, the reference's own setting, on the SAME full input: exactly one timed step (~11 s) after a small warm-up
    one_sample = bags
    dt_one, k_one = _cpu_step_timer(one_sample, n, d, 1, 0.0, max_steps=1, warm_sample=2)
    torch.set_num_threads(min(cores, 64))
    return dict(value=sample / dt_all, unit="bags/s", cores=best, kind="port",
                sample=f"{k_all} timed step(s) of the {'full' if full else 'same'} step on {sample} bags x {n} x {d} per view (fp32 "
                       f"torch CPU, {best} threads = the fastest of the probed pool sizes "
                       f"{ {t: round(8 / v, 2) for t, v in probe.items()} } bags/s on 8 bags; host has {cores} logical CPUs), "
                       f"{dt_all * 1e3:.0f} ms/step",
                single_thread=dict(value=one_sample / dt_one, unit="bags/s", cores=1,
                                   sample=f"{k_one} timed step(s) on {one_sample} bags x {n} x {d} per view (threads=1, the "
                                          f"reference's torch.set_num_threads(1)), {dt_one * 1e3:.0f} ms/step"))


def m_full(device, dtype, bags=64, raw=8192, steps=30):
    """SURVEY 8(d) "M-full": the reference's whole stage-1 step (train_MuRCL.py:233-304) - T = 6 patch steps x 2 views of
    1024 patches drawn from raw bags of 8192 by the cluster-window sampler + mix-up, aggregator, head, NT-Xent, one
    backward, Adam - reported beside the headline (outside its timed region; tools/bench_full.py has stages 2 and 3)."""
    import numpy as np
    from murcl_amd.models import rlmil
    from murcl_amd.train_MuRCL import build_parser, create_model, get_optimizer, pretrain_step
    from murcl_amd.utils.datasets import BagPack
    from murcl_amd.utils.losses import NT_Xent
    args = build_parser().parse_args(["--arch", "ABMIL", "--fc_lr", "5e-5"])
    args.T, args.feat_size, args.batch_size, args.train_stage, args.num_clusters = 6, 1024, bags, 1, 10
    args.dtype = "bf16" if dtype == torch.bfloat16 else "f32"
    torch.manual_seed(985)
    model, fc, ppo = create_model(args, 512, device)
    opt = get_optimizer(args, model, fc)
    g = torch.Generator(device=device)
    g.manual_seed(1)
    feats = [torch.randn((raw, 512), generator=g, device=device).abs_().mul_(0.5) for _ in range(bags)]
    rng = np.random.default_rng(985)
    clusters = []
    for _ in range(bags):
        lab = rng.integers(0, 10, raw)
        clusters.append([np.nonzero(lab == k)[0].tolist() for k in range(10)])
    pack = BagPack.from_lists(feats, clusters, dtype=dtype)
    crit, mem = NT_Xent(bags, 1.0), [rlmil.Memory(), rlmil.Memory()]

    def run():
        for _ in range(5):
            pretrain_step(args, model, fc, ppo, crit, opt, pack, mem)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pretrain_step(args, model, fc, ppo, crit, opt, pack, mem)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    dt = run()
    out = dict(workload=f"MuRCL stage-1 step: {bags} raw bags x {raw} x 512 -> T=6 x 2 views of 1024 (sampler + mix-up included)",
               ms_per_step=round(dt * 1e3, 3), bags_per_s=round(bags / dt, 1), steps=steps)
    # BASELINE configs[3]'s body on one GPU's 64 bags: the PPO sampler in the loop - stage 2 (encoder frozen, two
    # PPO.update per step) and stage 3 (joint: sampler picks the windows, encoder + head train)
    for stage in (2, 3):
        args.train_stage = stage
        ppo = rlmil.PPO(512, args.model_dim, args.policy_hidden_dim, args.policy_conv, action_std=args.action_std, lr=args.ppo_lr,
                        gamma=args.ppo_gamma, K_epochs=args.K_epochs, action_size=args.num_clusters)
        dts = run()
        out[f"stage{stage}"] = dict(ms_per_step=round(dts * 1e3, 3), bags_per_s=round(bags / dts, 1))
    return out


def headline_f32(device, B, N, D, steps=8):
    """The headline step on the PARITY path (f32 everywhere, exact-f32 MFMA: the kernels held to 1e-4 against the reference's outputs)
    - the throughput line of the path that carries the parity claim (only `--dtype f32` by hand measured it)."""
    model, fc, opt, crit = build(torch.float32, device, B)
    views = synth_views(B, N, D, torch.float32, device, 0)
    step = make_step(model, fc, opt, crit, views, 1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        loss = step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    flops = 2 * B * 9.41e9 * (N / 2048.0)
    return dict(workload=f"the headline step with f32 patch tensors (parity path): {B} bags x {N} x {D} per view", ms_per_step=round(ms, 4),
                bags_per_s=round(B / ms * 1e3, 1), steps=steps, loss=round(float(loss.item()), 6),
                frac_of_f32_mfma_peak=round(flops / (ms * 1e-3) / 1e12 / PEAK["mfma_f32_TFLOPs"], 4),
                note="exact-f32 v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 matrix rate: this path exists for parity (<= 1e-4 vs the "
                     "reference), the bf16 storage path above for throughput")


def _timed_ms(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def _timed_ms_back_to_back(fn, reps=10, warm=2):
    """``fn`` ``reps`` times between ONE pair of events (no synchronisation in between): the steady state of a loop, where the host runs
    ahead of the GPU - ``_timed_ms`` times single passes from an idle GPU and includes the host's way to the first launch."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(3):                              # median of three windows: one allocator growth inside a window would own its mean
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
    return sorted(ts)[1]


def _kernel_table(fn, passes=3):
    """Per-kernel HIP-event times of ``fn`` (the spans ops.py defines: the kernels that carry algorithmic bytes / FLOPs), averaged
    over ``passes`` runs - an untimed extra pass, like the headline's breakdown.  -> {key: {us, GBps, frac_of_8TBps | TFLOPs}}"""
    from murcl_amd import ops
    ops.TIMERS = ops.KernelTimers()
    try:
        for _ in range(passes):
            fn()
        torch.cuda.synchronize()
        summ = ops.TIMERS.summary()
    finally:
        ops.TIMERS = None
    out = {}
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms_total"]):
        if v["ms_total"] / passes < 0.02:
            continue
        us = v["ms_avg"] * 1e3
        e = {"launches_per_pass": v["calls"] // passes, "us": round(us, 1)}
        if v["bytes"]:
            gbps = v["bytes"] / v["calls"] / (v["ms_avg"] * 1e-3) / 1e9
            e.update(GBps=round(gbps, 1), frac_of_8TBps=round(gbps / PEAK["hbm_GBps"], 4))
        if v["flops"]:
            e["TFLOPs"] = round(v["flops"] / v["calls"] / (v["ms_avg"] * 1e-3) / 1e12, 1)
        out[k] = e
    return out


def other_rows(device):
    """The other aggregators of SURVEY section 8 at their BASELINE shapes, beside the headline (outside its timed
    region): BASELINE configs[2] CLAM-SB + instance loss 64 x 4096 x 512 (bf16 storage) and one GPU's share of configs[4]
    DSMIL 16 x 8192 x 1024 (fp32 like the reference), forward + backward through the batched internals the training
    steps use (train_RLMIL.supervised_step), median of 10 HIP-event timings.  Two accountings per row: `survey_8d` = the
    ALGORITHMIC work of SURVEY section 8(d) (FLOPs of the K4 / K6 chain forward + backward against the dense bf16 MFMA peak; X
    once forward and twice backward against 8 TB/s), and `chain_traffic_*` = the bytes the un-fused operator chain as built
    moves (14 passes over [B*N,512] for CLAM: 5 forward + 9 backward; for the reassociated DSMIL chain the two accountings coincide: 3 passes over X) - the second says how well the passes stream, not how close
    the row is to its algorithmic floor."""
    from murcl_amd import ops
    from murcl_amd.models.clam import CLAM_SB
    from murcl_amd.models.dsmil import build_dsmil
    g = torch.Generator(device=device)
    g.manual_seed(3)
    out = {}
    B, N = 64, 4096
    m = CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512).to(device)
    m.compute_dtype = torch.bfloat16
    m.eval()
    # as under the training scripts' optimizers, the parameters are "managed": their compute-dtype / transposed / interleaved views are
    # cached between optimizer steps instead of being rebuilt by a launch on every call (nothing updates them inside these rows)
    for p_ in m.parameters():
        ops.manage_param(p_)
    x = (torch.randn((B, N, 512), generator=g, device=device).abs() * 0.5).bfloat16()
    labels = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(1)).to(device)    # a device tensor, as in training

    def clam_fb(inst):
        for p in m.parameters():
            p.grad = None
        M, _, _, il, _, _ = m._run(x, labels if inst else None, inst)
        # d(sum)/d(output) = ones, handed to autograd directly: the row times the operator, not a harness's sum / add / fill launches
        outs = (M, il) if inst else (M,)
        torch.autograd.backward(outs, [_ones_like(o) for o in outs])
    for name, inst in (("clam_sb_c3_fwd_bwd_instance_loss", True), ("clam_sb_c3_fwd_bwd_aggregator", False),
                       ("clam_sb_c3_fwd_bwd_aggregator_training_mode", False)):
        # the last row: Dropout(0.25) behind fc and on both gate branches live, as the training scripts run the aggregator
        m.train(name.endswith("training_mode"))
        ms = _timed_ms(lambda: clam_fb(inst))
        ms_b2b = _timed_ms_back_to_back(lambda: clam_fb(inst))
        nbytes = 14 * B * N * 512 * 2          # passes over [B*N,512] bf16 tensors in the chain: forward 5 (x, h w/r/r, U w), backward 9
        # the two 512 x 512 projections dominate - fc forward + weight gradient (no dX), gate forward + dgrad + weight
        # gradient = 5 x 2*N*512^2 FLOP per bag (10.7 GFLOP at N = 4096); bytes: X once forward, twice backward
        flops = B * 5 * 2.0 * N * 512 * 512
        xbytes = 3 * B * N * 512 * 2
        out[name] = dict(workload=f"CLAM_SB {B} bags x {N} x 512 bf16", ms=round(ms, 4), ms_back_to_back=round(ms_b2b, 4),
                         bags_per_s=round(B / ms * 1e3, 1),
                         survey_8d=dict(bound="mfma", GFLOP_fwd_bwd=round(flops / 1e9, 1),
                                        frac_of_bf16_mfma_peak=round(flops / (ms * 1e-3) / (PEAK["mfma_bf16_TFLOPs"] * 1e12), 4),
                                        x_bytes_GB=round(xbytes / 1e9, 3), frac_of_8TBps_on_x_bytes=round(xbytes / ms / 1e6 / 8000, 4)),
                         chain_traffic_GB=round(nbytes / 1e9, 3), chain_traffic_frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 4))
        if name.endswith("training_mode"):
            out[name]["kernels"] = _kernel_table(lambda: clam_fb(False))
    del m, x
    B, N, d = 16, 8192, 1024
    md = build_dsmil(d, 2).to(device)
    for p_ in md.parameters():
        ops.manage_param(p_)
    xd = torch.randn((B, N, d), generator=g, device=device).abs() * 0.5

    def dsmil_fb():
        for p in md.parameters():
            p.grad = None
        classes, bag, cmax = md._run(xd, want_max=True)
        torch.autograd.backward((bag, cmax), (_ones_like(bag), _ones_like(cmax)))     # bag term + max-instance term (train_RLMIL.py:516-529)
    ms = _timed_ms(dsmil_fb)
    ms_b2b = _timed_ms_back_to_back(dsmil_fb)
    # Round 3: K6 reassociated (functional.DSMILFn) - the attention logits are X . (Wq^T q_max), so no GEMM over all patches is
    # left; the row is three streaming passes over X (instance scores; attention + pooling with an online soft-max; the whole
    # backward of both, dWc included) and ~30 launches on [B*C]-row tensors.  Its floor is three reads of X at the HBM roof.
    nbytes = 3 * B * N * d * 4
    floor_ms = nbytes / (PEAK["hbm_GBps"] * 1e9) * 1e3
    out["dsmil_c5_share_fwd_bwd"] = dict(workload=f"DSMIL {B} bags x {N} x {d} f32 (one GPU's share of 128 bags)", ms=round(ms, 4),
                                         ms_back_to_back=round(ms_b2b, 4), bags_per_s=round(B / ms * 1e3, 1),
                                         survey_8d=dict(bound="hbm", x_bytes_GB=round(nbytes / 1e9, 3),
                                                        frac_of_8TBps_on_x_bytes=round(nbytes / ms / 1e6 / 8000, 4),
                                                        note="reassociated K6: X twice forward (the critical instance must be known "
                                                             "before the attention pass), once backward"),
                                         chain_traffic_GB=round(nbytes / 1e9, 3),
                                         chain_traffic_frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 4),
                                         kernelwise_floor_ms=round(floor_ms, 4), frac_of_kernelwise_floor=round(floor_ms / ms, 4),
                                         floor_note="3 HBM passes over X at 8 TB/s (no GEMM over all patches remains)",
                                         kernels=_kernel_table(dsmil_fb))
    md.compute_dtype = torch.bfloat16                               # patch features stored in bf16, f32 accumulation
    xd = xd.bfloat16()
    ms = _timed_ms(dsmil_fb)
    nbytes = 3 * B * N * d * 2
    out["dsmil_c5_share_fwd_bwd_bf16"] = dict(workload=f"DSMIL {B} bags x {N} x {d} bf16 storage", ms=round(ms, 4),
                                              bags_per_s=round(B / ms * 1e3, 1), chain_traffic_GB=round(nbytes / 1e9, 3),
                                              chain_traffic_frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 4))
    return out


def box_calibration(device):
    """What THIS box delivers, measured in this process before the timed region (the same tree read 45.2 k and
    42.7 k bags/s on two boxes of the pool and nothing in the line could tell a slow box from a regression): a streaming copy of one
    [B*N,512] bf16 activation tensor (268 MB in, 268 MB out = the traffic of one encoder launch) and a register-only bf16 MFMA loop on
    non-trivial operands, both own kernels behind the C-ABI (csrc/runtime.hip), HIP-event medians; plus the clocks sysfs shows."""
    from murcl_amd import _lib
    L = _lib.lib()
    nbytes = 128 * 2048 * 512 * 2
    src = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=device).normal_()
    dst = torch.empty_like(src)
    copy_ms = _timed_ms(lambda: _lib.check(L.murcl_calib_copy(src.data_ptr(), dst.data_ptr(), nbytes, _lib.stream()), "calib_copy"), reps=15)
    del src, dst
    buf = torch.empty((256 * 512,), dtype=torch.float32, device=device)
    iters = 16000
    mfma_ms = _timed_ms(lambda: _lib.check(L.murcl_calib_mfma_bf16(buf.data_ptr(), iters, _lib.stream()), "calib_mfma"), reps=7)
    flops = 256.0 * 8 * iters * 4 * 16384
    out = dict(copy_GBps=round(2 * nbytes / copy_ms / 1e6, 1), copy_ms=round(copy_ms, 4), copy_bytes_moved=2 * nbytes,
               mfma_bf16_TFLOPs=round(flops / mfma_ms / 1e9, 1), mfma_ms=round(mfma_ms, 4),
               mfma_clock_MHz_if_4096_flop_per_clk_per_cu=round(flops / (mfma_ms * 1e-3) / (256 * 4096.0) / 1e6, 1),
               note="own kernels (murcl_calib_copy: 16-byte streaming copy of one activation tensor; murcl_calib_mfma_bf16: 8 waves per CU of "
                    "back-to-back v_mfma_f32_16x16x32_bf16 on register operands), measured in this process before the timed region")
    clocks = {}
    try:
        import glob
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            for name in ("pp_dpm_sclk", "pp_dpm_mclk"):
                fn = os.path.join(card, name)
                if os.path.exists(fn):
                    with open(fn) as f:
                        cur = [ln.split(":")[1].strip().rstrip("*").strip() for ln in f if ln.strip().endswith("*")]
                    if cur:
                        clocks.setdefault(os.path.basename(os.path.dirname(card)), {})[name] = cur[0]
            if clocks:
                break
    except Exception:                                                          # noqa: BLE001
        pass
    out["sysfs_clocks_idle"] = clocks or None
    return out


def graph_timed_region(step, opt, steps, barrier):
    """The timed region as hipGraph replays (round 6): G consecutive steps are captured ONCE - with the optimizer in its live-graph mode,
    where Adam's step counts advance on the device (murcl_adam_multi_live), so every replayed step is the NEXT training step, not the
    captured one again - and the region is steps / G replays bracketed like the eager one.  No host work sits between the launches of a
    step: one host stall cannot land in a 20-step window (it cost the driver's round-5 line 2-5 %, and an 8.9 ms one 25 % in round 4).
    G = the largest of (10, 5, 4, 2, 1) that divides ``steps`` (a replay costs ~10-16 us to start: amortised over G steps).
    -> dict(ms_per_step, elapsed_s, steps_per_graph, replays, per-replay statistics)."""
    G = next(g for g in (10, 5, 4, 2, 1) if steps % g == 0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):                                   # lazily allocated per-stream state (NT-Xent exchange buffer) exists before capture
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    opt.live_graph(True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            for _ in range(G):
                step()
    except Exception:
        opt.live_graph(False)
        raise
    n_rep = steps // G
    done = 0
    for _ in range(max(2, 20 // G)):                         # untimed replays (real steps) right up to the barrier
        g.replay()
        done += G
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(n_rep):
        g.replay()
        evs[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    done += steps
    opt.after_replays(done)
    per = sorted(evs[i].elapsed_time(evs[i + 1]) / G for i in range(n_rep))
    return dict(ms_per_step=elapsed / steps * 1e3, elapsed_s=elapsed, steps_per_graph=G, replays=n_rep,
                per_replay_ms_per_step=dict(median=round(per[len(per) // 2], 4), min=round(per[0], 4), max=round(per[-1], 4)),
                note="G steps captured once, replayed steps / G times; Adam's step counts live on the device (murcl_adam_multi_live): every "
                     "replayed step is the next optimizer step")


def launch_argv(gpus, argv, port=None):
    """The command line ``python bench.py --gpus N ...`` turns itself into when no launcher started it: the driver's own form
    (one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1)."""
    port = port or int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py"), *argv]


def count_gpus(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs of this node WITHOUT touching the HIP / HSA runtime: KFD topology nodes with a non-zero ``simd_count`` (CPU nodes report
    0).  ``torch.cuda.device_count()`` falls through to ``hipGetDeviceCount`` on ROCm builds without amdsmi, which initialises
    the runtime in the parent that is about to start its ranks as children - so it is only the fallback for a node whose sysfs
    is not readable."""
    n, seen = 0, False
    try:
        for node in sorted(os.listdir(topology)):
            try:
                with open(os.path.join(topology, node, "properties")) as f:
                    props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            except OSError:
                continue
            seen = True
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        pass
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if seen and vis:
        n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
    return n if seen else torch.cuda.device_count()


def comm_diagnostics(model, fc, opt, crit, views, world, device, reps=20):
    """The `comm` block of the bench line (world > 1, or MURCL_FORCE_DIST=1 on one rank): what the collectives of the step cost
    by themselves, and how much of them the step leaves exposed - so that a scaling curve can be READ, not only recorded.
    Every rank runs it (the passes contain collectives); all times are HIP-event medians on the launch stream, taken outside the
    timed region.

    * `z_all_gather_us`: the one data-path exchange: all_gather_into_tensor of [2 B_local, 128] f32 per rank.
    * `grad_all_reduce_us`: the SUM all-reduce of each optimizer group's flat gradient buffer, alone on an idle GPU.
    * `exposed_us_per_step`: median step with the collectives minus median step without them on the same rank (same kernels, the
      local NT-Xent instead of the gathered one): the part of the communication the backward pass does not cover.
    * `rccl`: the RCCL knobs in the environment (channel / CU budget) - the persistent kernels of the step are one workgroup per CU
      with 128-130 KiB of LDS, so every CU RCCL's channels occupy delays a workgroup by a whole round."""
    import torch.distributed as dist
    from murcl_amd import dist as mdist

    def ev_median(fn, n=reps):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dist.barrier()
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        return round(ts[len(ts) // 2], 1)

    B = views[0].shape[0]
    nranks = dist.get_world_size()
    z_local = torch.randn((2 * B, 128), device=device)
    zg = torch.empty((nranks * 2 * B, 128), device=device)
    from murcl_amd import ops as _ops
    from murcl_amd import functional as _fnl
    out = {"ranks": nranks, "backend": dist.get_backend(), "cu_budget": _ops.cu_budget(),
           "cu_budget_overlapped_backward": _fnl._OVERLAP_BUDGET,
           "z_all_gather_us": ev_median(lambda: mdist.all_gather_rows(zg, z_local)),
           "z_all_gather_bytes_per_rank": z_local.numel() * 4}
    ar = {}
    for gi, g in enumerate(opt.flat_grads()):
        buf = torch.zeros_like(g)
        us = ev_median(lambda: mdist.all_reduce_sum(buf))
        ar[f"group{gi}"] = {"bytes": buf.numel() * 4, "us": us,
                            "busbw_GBps": round(2 * (nranks - 1) / nranks * buf.numel() * 4 / (us * 1e-6) / 1e9, 1) if nranks > 1 else None}
    out["grad_all_reduce_us"] = ar

    def step_median(step, n=40):
        for _ in range(5):
            step()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        dist.barrier()
        torch.cuda.synchronize()
        evs[0].record()
        for i in range(n):
            step()
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        return per[n // 2] * 1e3
    with_comm = step_median(make_step(model, fc, opt, crit, views, 2))          # the multi-rank branch (also with one rank)
    # (under MURCL_MILESTONES=1 the multi-rank step leaves its reducer bound as the aggregator's gradient milestone: a local step
    #  built behind it would still launch the reducer's collectives and never finish() them)
    from murcl_amd import functional as _fn
    _fn.set_grad_milestone(None)
    local = step_median(make_step(model, fc, opt, crit, views, 1))
    out["step_us_with_collectives"] = round(with_comm, 1)
    out["step_us_local_only"] = round(local, 1)
    out["exposed_us_per_step"] = round(with_comm - local, 1)
    out["overlap"] = "head-group all-reduce launched from an autograd hook under the aggregator backward (dist.OverlappedGradReduce); " \
                     "model group + z all-gather are on the critical path"
    knobs = ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_NCHANNELS_PER_PEER", "RCCL_ENABLE_INTRANET", "NCCL_ALGO", "NCCL_PROTO",
             "NCCL_BUFFSIZE", "HSA_ENABLE_IPC_MODE_LEGACY", "RCCL_MSCCL_ENABLE", "MURCL_MILESTONES")
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                                          # noqa: BLE001
        ver = None
    out["rccl"] = {"version": ver, "env": {k: os.environ[k] for k in knobs if k in os.environ},
                   "channels_note": "RCCL does not export its channel / CU count through torch; NCCL_DEBUG=INFO prints it at init "
                                    "(\"N coll channels\"); cap it with NCCL_MAX_NCHANNELS if exposed_us_per_step grows with N"}
    return out


def self_launch(gpus, argv):
    """``python bench.py --gpus N`` without a launcher (the reference needs none: one process, nn.DataParallel,
    train_MuRCL.py:145): this parent - which has made NO GPU call, so nothing is re-executed over an initialised device -
    starts the N ranks as CHILD processes, lets rank 0's JSON line through on its stdout and returns the children's exit code."""
    import subprocess
    have = count_gpus()                               # from sysfs: the parent makes no HIP / HSA call at all
    if have < gpus:
        print(f"bench.py: --gpus {gpus} but this node has {have} GPU(s)", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    return subprocess.run(launch_argv(gpus, argv), env=env, cwd=ROOT).returncode


SETTLE_STEPS = 30          # untimed steps (warm-up + breakdown pass + extra) that precede the timed region at the least


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--stat-steps", type=int, default=100, help="steps of the per-step HIP-event statistics pass (0 = off)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--bags", type=int, default=64)
    ap.add_argument("--patches", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action=argparse.BooleanOptionalAction, default=True,
                    help="after the timed region, capture one step as a hipGraph and report ms_per_step_graph (single GPU)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    args = ap.parse_args()

    if (args.gpus > 1 or os.environ.get("MURCL_BENCH_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    force_dist = os.environ.get("MURCL_FORCE_DIST") == "1"          # dev: run the RCCL code path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        from murcl_amd import dist as _mdist
        _mdist.cap_rccl_channels(world)                   # RCCL's channel workgroups must fit the CUs the step leaves free
        dist.init_process_group("nccl", device_id=device)

    from murcl_amd import ops
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B, N, D = args.bags, args.patches, 512
    model, fc, opt, crit = build(dtype, device, B)
    views = synth_views(B, N, D, dtype, device, rank)
    step = make_step(model, fc, opt, crit, views, 2 if force_dist else world)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step()                                                   # first call: allocator growth, lazy kernel-attribute set-up
    barrier()
    # Everything alive now (modules, torch, the extension) is long-lived: move it out of the cyclic collector's young
    # generations so that a full collection (30-40 ms of host stall once every few dozen steps, measured with
    # tools/host_vs_gpu.py) does not land inside the timed steps.  The collector stays enabled.  This happens BEFORE the
    # warm-up: the tens of milliseconds it takes leave the GPU idle, and the first ~10 steps after an idle period run
    # 14 % slower (1.82 vs 1.595 ms per step in 10-step regions after a cold start or a 2 s pause, tools/_dbg_bench.py) -
    # with the driver's --warmup 5 those steps used to fall into the timed region.
    gc.collect()
    gc.freeze()
    box = None
    if rank == 0:
        try:
            box = box_calibration(device)
        except Exception as e:                                                 # noqa: BLE001  (a diagnostic must not take the line with it)
            box = {"error": f"{type(e).__name__}: {e}"[:300]}
    for _ in range(max(0, args.warmup - 1)):
        step()

    # -- breakdown pass (untimed): find the kernel that dominates the step
    ops.TIMERS = ops.KernelTimers()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    breakdown = ops.TIMERS.summary()
    ops.TIMERS = None
    dominant = max(breakdown, key=lambda k: breakdown[k]["ms_total"])
    k2_key = f"abmil_pool_fwd<{args.dtype}>"

    # -- timed region: exactly K steps; only the dominant kernel and K2 carry HIP events, on every third launch
    k2row_key = f"row:k2_fwd<{args.dtype}>"             # the K2 row of the step: ONE launch since round 6 (its per-bag merge is inside the decoder launch)
    ops.TIMERS = ops.KernelTimers(only={dominant, k2row_key}, every=3, pool=6 * (args.steps + 40))
    # settle: untimed steps (in the timed region's configuration) back to back right up to the barrier that opens it, so
    # that at least SETTLE_STEPS steps precede the timed ones whatever --warmup says and the few milliseconds of host work
    # above (reading the breakdown events) are not the last thing the GPU saw
    warmup_extra = max(10, SETTLE_STEPS - args.warmup - 2)
    for _ in range(warmup_extra):
        step()
    ops.TIMERS.reset()
    diag = os.environ.get("MURCL_BENCH_DIAG") == "1"          # dev: one HIP event per timed step (where does a short region lose time?)
    devs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if diag else None
    barrier()
    t0 = time.perf_counter()
    host_t = [t0]
    if diag:
        devs[0].record()
    for i in range(args.steps):
        loss = step()
        if diag:
            devs[i + 1].record()
        host_t.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    if diag and rank == 0:
        per = [devs[i].elapsed_time(devs[i + 1]) for i in range(args.steps)]
        print("DIAG gpu ms per timed step:", [round(p, 3) for p in per], file=sys.stderr)
        print("DIAG host ms per timed step:", [round((host_t[i + 1] - host_t[i]) * 1e3, 2) for i in range(args.steps)], file=sys.stderr)
        print("DIAG elapsed", round(elapsed * 1e3, 3), "sum gpu", round(sum(per), 3), file=sys.stderr)
    host_ms = [(host_t[i + 1] - host_t[i]) * 1e3 for i in range(args.steps)]
    live = ops.TIMERS.summary()
    ops.TIMERS = None
    # -- statistics pass (outside the timed region): per-step HIP-event durations on the launch stream, no kernel timers,
    #    every rank takes part (the steps contain collectives).  median / p10 / p90 go out beside the headline value.
    stats = None
    if args.stat_steps > 0:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.stat_steps + 1)]
        barrier()
        evs[0].record()
        for i in range(args.stat_steps):
            step()
            evs[i + 1].record()
        barrier()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.stat_steps))
        q = lambda f: per[min(len(per) - 1, int(f * len(per)))]  # noqa: E731
        stats = dict(steps=args.stat_steps, median_ms=round(q(0.5), 4), p10_ms=round(q(0.1), 4), p90_ms=round(q(0.9), 4),
                     min_ms=round(per[0], 4), max_ms=round(per[-1], 4), timing="HIP events on the launch stream, one per step")
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    comm = None
    if world > 1 or force_dist:
        try:
            comm = comm_diagnostics(model, fc, opt, crit, views, world, device)
        except Exception as e:                                                 # noqa: BLE001  (diagnostics must not take the line with them)
            comm = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank != 0:
        dist.destroy_process_group()
        return

    ms_step = elapsed / args.steps * 1e3
    value = B * world / (elapsed / args.steps)
    eager = dict(value=round(value, 2), ms_per_step=round(ms_step, 4),
                 note="K steps enqueued step by step from Python between the same barriers (the per-kernel HIP events of `roofline` were taken in this region)")
    timing_mode = "eager: K steps enqueued from Python (the host runs ~0.6 ms per step ahead of the GPU)"
    graph = None
    if world == 1 and not force_dist and args.graph:
        # a SECOND region of exactly K steps, as hipGraph replays with live optimizer state (round 6: every replayed step is a real
        # training step, Adam's step counts advance on the device).  It is GPU-paced: the host launches K / G graphs and waits.  The
        # eager region is host-paced whenever the host is slower than the GPU - measured on this pool: a loaded host (four tenants per
        # box) enqueued a step in 1.56 ms instead of 0.8 and the eager region read 1.958 ms per step while the GPU needed 1.45
        # (gpurun_out/r06_zz_bench_driver_args2.json), and in a 20-step window the first step alone runs from an empty queue (+ 1-2 %).
        # With a healthy host the two regions agree to 0.5 % (profiles/r06_zz_bench*.json).  `value` is therefore the graph region
        # when it exists ("make value robust to one host stall ... the graph replay once its arguments are live"),
        # and the eager region is reported beside it; --no-graph, more than one rank, or a capture failure leave `value` eager.
        try:
            graph = graph_timed_region(step, opt, args.steps, barrier)
            graph["ms_per_step_graph"] = round(graph["ms_per_step"], 4)
            graph["value_graph"] = round(B * world / (graph["elapsed_s"] / args.steps), 2)
            value, ms_step = B * world / (graph["elapsed_s"] / args.steps), graph["ms_per_step"]
            timing_mode = ("hipgraph: the K steps as replays of captured graphs of G steps with LIVE optimizer state, between the same "
                           "barriers (GPU-paced; the eager region of the same K steps - `eager` - is host-paced when the host is loaded)")
        except Exception as e:                                                 # noqa: BLE001  (a diagnostic must not take the line with it)
            graph = {"error": f"{type(e).__name__}: {e}"[:300]}

    # SURVEY 8(d) classifies the kernels: the encoder-sized GEMMs (K1: forward, input gradients, weight gradients) are MFMA-bound
    # (a fused encoder moves X once: intensity ~1.7 kFLOP/B), K2 and the streaming passes are HBM-bound.  The layer-wise bytes of a
    # GEMM launch (A in + C out) are what THIS build moves per layer, not what the algorithm needs: they are reported beside the
    # flops fraction, never instead of it.
    def bound_of(key):
        return "mfma" if key.startswith(("panel_gemm", "gemm_tn", "gemm_nt")) else "hbm"

    def roof(key, extra_keys=()):
        """Roofline of one kernel from its live HIP-event time in the timed region (``extra_keys``: further launches that belong to
        the same kernel row - their time is added, their internal bytes are not)."""
        r = live[key]
        sec_avg = r["ms_avg"] * 1e-3 + sum(live[k]["ms_avg"] * 1e-3 for k in extra_keys if k in live)
        mfma_peak = PEAK["mfma_bf16_TFLOPs"] if "bf16" in key or "K512" in key or "K128" in key else PEAK["mfma_f32_TFLOPs"]
        by, fl = r["bytes"] / r["calls"], r["flops"] / r["calls"]
        frac_bytes = by / sec_avg / 1e9 / PEAK["hbm_GBps"]
        frac_flops = fl / sec_avg / 1e12 / mfma_peak
        bound = bound_of(key)
        if bound == "hbm":
            ach, peak, unit, frac = by / sec_avg / 1e9, PEAK["hbm_GBps"], "GB/s", frac_bytes
        else:
            ach, peak, unit, frac = fl / sec_avg / 1e12, mfma_peak, "TFLOP/s", frac_flops
        traffic, src = _pmc_traffic(key)
        if key.startswith("row:k2_fwd"):
            traffic, src = _pmc_traffic(f"abmil_pool_fwd<{args.dtype}>")
        box_copy = (box or {}).get("copy_GBps")
        box_mfma = (box or {}).get("mfma_bf16_TFLOPs")
        out = dict(kernel=key, bound=bound, achieved=round(ach, 2), peak=peak, unit=unit, frac=round(frac, 4),
                   frac_flops=round(frac_flops, 4), frac_layer_bytes=round(frac_bytes, 4),
                   frac_of_box_copy=round(by / sec_avg / 1e9 / box_copy, 4) if box_copy else None,
                   frac_of_box_mfma=round(fl / sec_avg / 1e12 / box_mfma, 4) if (box_mfma and mfma_peak == PEAK["mfma_bf16_TFLOPs"]) else None,
                   bound_source="SURVEY.md 8(d): K1 (encoder GEMMs) MFMA-bound, K2 / streaming passes HBM-bound",
                   traffic=traffic, traffic_source=src, launches=r["calls"], avg_launch_ms=round(sec_avg * 1e3, 4),
                   algorithmic_bytes_per_launch=int(by), algorithmic_flops_per_launch=int(fl))
        if extra_keys:
            out["launches_of_the_row"] = [key] + [k for k in extra_keys if k in live]
            out["avg_ms_each"] = [round(live[k]["ms_avg"], 4) for k in out["launches_of_the_row"]]
        if key.startswith("gemm_tn_sq"):
            out["note"] = ("the three encoder weight gradients of the step as ONE grouped launch of gemm_tn_sq_kernel (256x256 tiles, 12 "
                           "(layer, tile) pairs x 21 row splits = one workgroup per CU, partial sums to a workspace) + ONE tn_reduce_kernel "
                           "(adds them to the gradients, bias-gradient rows ride along): avg_launch_ms and traffic cover BOTH launches plus "
                           "~2-3 us of event records; rocprofv3 lists them separately (profiles/*_kernel_stats.csv)")
        return out

    def step_roof():
        """The whole step against SURVEY 8(d): 2B bag-forwards x (3.493 GFLOP forward, ~9.41 GFLOP forward + backward at N = 2048)
        on the dense bf16 MFMA peak, and the HBM bytes it moves (PMC bytes per launch of the committed passes x launches per step)
        against the fused minimum (X once)."""
        flops = 2 * B * 9.41e9 * (N / 2048.0)
        fused_min = 2 * B * N * D * (2 if args.dtype == "bf16" else 4)
        calls = {k: v["calls"] // 2 for k, v in breakdown.items()}          # the breakdown pass ran two steps
        moved, covered, src = 0, [], None
        for k, c in calls.items():
            t, s_ = _pmc_traffic(k if not k.startswith("gemm_tn_sq_grouped") else "gemm_tn_sq_grouped3<bf16>")
            if t:
                moved += t * c
                covered.append(k)
                src = src or s_
        out = dict(flops=flops, mfma_peak_TFLOPs=PEAK["mfma_bf16_TFLOPs"] if args.dtype == "bf16" else PEAK["mfma_f32_TFLOPs"],
                   frac_of_mfma=round(flops / (ms_step * 1e-3) / 1e12 / (PEAK["mfma_bf16_TFLOPs"] if args.dtype == "bf16" else PEAK["mfma_f32_TFLOPs"]), 4),
                   fused_min_bytes=fused_min, hbm_bytes_pmc=int(moved) if moved else None,
                   ratio_vs_fused_min=round(moved / fused_min, 2) if moved else None,
                   avg_hbm_GBps=round(moved / (ms_step * 1e-3) / 1e9, 1) if moved else None,
                   pmc_kernels=sorted(covered), pmc_source=src,
                   note="hbm_bytes_pmc sums the encoder-sized kernels only (the bag-level head moves < 3 % of the step's bytes)")
        return out

    out = {
        "metric": "WSI-bags/sec pretrain step (ABMIL+NT-Xent) at N=2048,d=512",
        "value": round(value, 2), "unit": "bags/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "warmup_extra": warmup_extra + 2,
        "warmup_note": f"W={args.warmup} warm-up steps as asked, then {warmup_extra + 2} more untimed steps (2 of them the per-kernel breakdown "
                       f"pass, the rest right up to the barrier that opens the timed region) so that >= {SETTLE_STEPS} steps precede the timed ones: "
                       "the first ~10 steps after an idle GPU run 14 % slower",
        "ms_per_step": round(ms_step, 4), "timing_mode": timing_mode, "eager": eager,
        # eager slower than the GPU-paced region by more than 3 %: the host, not the GPU, paced the eager region - its per-kernel HIP
        # events (`roofline`, `roofline_k2`) then include the GPU's wait for late launches and read low
        "eager_host_paced": (bool(eager["ms_per_step"] > 1.03 * graph["ms_per_step"]) if graph and "ms_per_step" in graph else None),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "precision_note": ("bf16 storage of patch-level tensors, f32 accumulation and f32 bag-level math: checked against the f32 kernels to ~2e-2 "
                           "(outputs <= 2 % of max, attention <= 3-5 % rel, gradients <= 5 % norm-wise, 40-step loss trajectories within 2 %: "
                           "tests/test_gpu_modules.py, test_gpu_step.py); the f32 kernels are the ones held to 1e-4 against the reference") if args.dtype == "bf16" else
                          "f32 parity path: <= 1e-4 against outputs of the reference (tests/golden)",
        "config": {"workload": f"ABMIL+Full_layer+NT-Xent view-pair pretrain step (fwd+bwd+Adam), {B} bags x {N} x {D} "
                               f"per GPU, {args.dtype} patch tensors / f32 accumulate (BASELINE configs[1])",
                   "bags_per_gpu": B, "patches": N, "feat_dim": D, "global_bags": B * world,
                   "sharding": "bags by WSI; all-gather of z + grad all-reduce" if world > 1 else "single GPU"},
        "box": box,
        "roofline": dict(roof(dominant), step=step_roof()),
        "roofline_k2": dict(roof(k2row_key), kernel=k2_key, launches_of_the_row=[k2_key],
                            avg_ms_each_untimed_pass=[round(breakdown[k]["ms_avg"], 4) for k in (k2_key,) if k in breakdown],
                            merge_launch={"kernel": "abmil_pool_decoder", "replaces": ["abmil_pool_combine", "gemm_nt<f32,f32,BIAS_RELU> (decoder)"],
                                          "avg_ms_untimed_pass": round(breakdown["abmil_pool_decoder"]["ms_avg"], 4) if "abmil_pool_decoder" in breakdown else None},
                            note="the K2 row is ONE launch (scores + soft-max partials + pooled partials in one pass over H); its per-bag merge of "
                                 "<= 8 chunk partials happens while the decoder product loads its A operand (merge_launch: one launch where "
                                 "rounds 1-5 ran a combine launch and a GEMM), the normalised attention rows come out of the backward pass; "
                                 "avg_launch_ms = one HIP-event pair around the launch inside the timed region (~2.5 us of record cost included)"),
        "comm": comm,
        "graph": graph,
        "ms_per_step_graph": graph.get("ms_per_step_graph") if graph else None,
        "ms_per_step_eager": eager["ms_per_step"],
        "loss": round(float(loss.item()), 6),
        "step_stats": stats,
        "timed_region_host_enqueue_ms": {"median": round(sorted(host_ms)[len(host_ms) // 2], 3), "max": round(max(host_ms), 3),
                                         "drain_after_last_enqueue": round((elapsed - (host_t[-1] - t0)) * 1e3, 3)},
        "bags_per_s_at_median": round(B * world / (stats["median_ms"] * 1e-3), 1) if stats else None,
        "kernels_vs_box": {k: dict(us=round(v["ms_avg"] * 1e3, 1), layer_GBps=round(v["bytes"] / v["calls"] / (v["ms_avg"] * 1e-3) / 1e9, 1),
                                   frac_of_box_copy=round(v["bytes"] / v["calls"] / (v["ms_avg"] * 1e-3) / 1e9 / box["copy_GBps"], 3),
                                   TFLOPs=round(v["flops"] / v["calls"] / (v["ms_avg"] * 1e-3) / 1e12, 1),
                                   frac_of_box_mfma=round(v["flops"] / v["calls"] / (v["ms_avg"] * 1e-3) / 1e12 / box["mfma_bf16_TFLOPs"], 3))
                           for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms_total"])
                           if v["bytes"] and v["ms_avg"] > 0.03 and box and box.get("copy_GBps")} or None,
        "kernels_vs_box_note": "every encoder-sized launch of the step (untimed breakdown pass, one HIP-event pair each): its layer-wise bytes and FLOPs "
                               "against what a plain copy / a register-only MFMA loop reach on THIS box (`box`), not against the data-sheet peaks",
        "kernel_ms_per_step_note": "untimed 2-step pass with EVERY launch bracketed by HIP events (~2-3 us each): sums above ms_per_step",
        "kernel_ms_per_step": {k: round(v["ms_total"] / 2, 4) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms_total"])},
    }
    if world == 1 and not args.no_cpu_baseline:
        del model, fc, opt, views, step                    # free the headline step's activations before the second workload
        gc.collect()
        torch.cuda.empty_cache()
        # the extras run after the headline has been measured: a failure in one of them is recorded, it must not take the
        # contract's JSON line with it
        def extra(name, fn):
            try:
                out[name] = fn()
            except Exception as e:                                   # noqa: BLE001
                out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                print(f"bench.py: extra '{name}' failed: {e!r}", file=sys.stderr, flush=True)
            gc.collect()
            torch.cuda.empty_cache()
        extra("abmil_c2_f32", lambda: headline_f32(device, B, N, D))
        extra("m_full", lambda: m_full(device, dtype))
        extra("rows", lambda: other_rows(device))
        extra("cpu_baseline", lambda: cpu_baseline(B, N, D))
    if args.breakdown:
        for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms_total"]):
            print(f"{k:44s} calls/step {v['calls'] // 2:3d}  ms/step {v['ms_total'] / 2:8.4f}", file=sys.stderr)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)          # RCCL's banner sits in the C stdio buffer: push it out BEFORE the result line
    print(json.dumps(out), flush=True)


def _pmc_traffic(key):
    """(HBM bytes per launch, the run they were measured in) from a committed rocprofv3 --pmc pass (profiles/pmc_traffic.json:
    separate --pmc passes over this script, tools/pmc_bench.sh), else (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d.get(key), d.get("_source")
    except (OSError, ValueError):
        return None, None


if __name__ == "__main__":
    main()
