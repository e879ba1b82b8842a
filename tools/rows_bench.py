#!/usr/bin/env python
"""Per-row measurements for SURVEY section 8 beyond the headline step: module forward / forward+backward times at
the BASELINE config shapes, with algorithmic HBM bytes and the fraction of the 8 TB/s roof.  Writes JSON lines."""
import json, math, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import ops
from murcl_amd.models.clam import CLAM_SB
from murcl_amd.models.dsmil import build_dsmil
from murcl_amd.models.abmil import ABMIL
from murcl_amd.models import rlmil
from murcl_amd.utils.datasets import BagPack, subbag_views
from murcl_amd.utils.losses import NT_Xent

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)

def timed(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]

def report(row, what, ms, nbytes, units, unit_name):
    print(json.dumps({"row": row, "what": what, "ms": round(ms, 4), "algorithmic_GB": round(nbytes / 1e9, 3),
                      "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4),
                      unit_name + "_per_s": round(units / ms * 1e3, 1)}), flush=True)

only = sys.argv[1] if len(sys.argv) > 1 else ""
# ---- C3: CLAM-SB + instance loss, 64 bags x 4096 x 512 (bf16 storage)
if "clam" in only or not only:
    B, N = 64, 4096
    m = CLAM_SB(gate=True, size_arg="small", dropout=True, k_sample=8, n_classes=2, subtyping=True, in_dim=512).to(dev)
    m.compute_dtype = torch.bfloat16
    x = (torch.randn((B, N, 512), generator=g, device=dev).abs() * 0.5).bfloat16()
    labels = [int(v) for v in torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(1))]
    m.eval()
    def fwd():
        with torch.no_grad(): m(x, label=labels, instance_eval=True)
    def fb():
        for p in m.parameters(): p.grad = None
        M, _, res = m(x, label=labels, instance_eval=True)
        (M.sum() + sum(r["instance_loss"] for r in res)).backward()
    es = 2
    # passes over [B*N,512]-sized tensors: fwd: x r, h w, h r (gate), U w, U r (score), h r (pool) = 6; bwd adds ~9
    report("a4-a8 CLAM_SB (K4/K5) C3", "forward+instance eval (eval mode)", timed(fwd), 6 * B * N * 512 * es, B, "bags")
    report("a4-a8 CLAM_SB (K4/K5) C3", "forward+backward", timed(fb), 15 * B * N * 512 * es, B, "bags")
    def fb_agg():                      # the aggregator alone, as the contrastive pre-training uses it (no labels)
        for p in m.parameters(): p.grad = None
        out = m(x)
        out[0].sum().backward()
    report("a4-a8 CLAM_SB (K4/K5) C3", "forward+backward, aggregator only (no instance eval)", timed(fb_agg), 15 * B * N * 512 * es, B, "bags")
    m.train()                          # Dropout(0.25) behind fc and on both gate branches: what the training scripts run
    report("a4-a8 CLAM_SB (K4/K5) C3", "forward+backward, aggregator only, training mode (dropout)", timed(fb_agg), 15 * B * N * 512 * es, B, "bags")
    m.eval()
# ---- C5 share of one GPU: DSMIL 16 bags x 8192 x 1024 f32
if "dsmil" in only or not only:
    B, N, d = 16, 8192, 1024
    m = build_dsmil(d, 2).to(dev)
    x = torch.randn((B, N, d), generator=g, device=dev).abs() * 0.5
    def fwd():
        with torch.no_grad(): m(x)
    def fb():                          # bag term + max-instance term (train_RLMIL.py:516-529), batched as the training step has it
        for p in m.parameters(): p.grad = None
        classes, bag, cmax = m._run(x, want_max=True)
        (bag.sum() + cmax.sum()).backward()
    report("a9-a11 DSMIL (K6) C5/8", "forward (2 passes over X)", timed(fwd), 2 * B * N * d * 4, B, "bags")
    report("a9-a11 DSMIL (K6) C5/8", "forward+backward (4 passes over X)", timed(fb), 4 * B * N * d * 4, B, "bags")
# ---- C4 per-GPU share: sub-bag builder, 64 raw bags x 8192 x 512 -> 2 views x 1024
if "subbag" in only or not only:
    B, R, fs = 64, 8192, 1024
    feats = [(torch.randn((R, 512), generator=g, device=dev).abs() * 0.5) for _ in range(B)]
    rng = np.random.default_rng(5)
    cl = []
    for _ in range(B):
        lab = rng.integers(0, 10, R); cl.append([np.nonzero(lab == k)[0].tolist() for k in range(10)])
    for dt_, name in ((torch.bfloat16, "bf16"), (None, "f32")):
        pack = BagPack.from_lists(feats, cl, dtype=dt_)
        acts = [torch.rand((B, 10), device=dev) for _ in range(2)]
        es = 2 if dt_ is not None else 4
        ms = timed(lambda: subbag_views(pack, acts, fs, alpha=0.9))
        report("a18-a19 get_feats+mixup (K12/K13) C4/8", f"2 views select+gather+mixup {name}", ms, 3 * 2 * B * fs * 512 * es, 2 * B, "subbags")
# ---- NT-Xent at the C4 global size (2B = 1024) and the C2 size
if "ntxent" in only or not only:
    for n in (128, 1024):
        z = torch.randn((n, 128), generator=g, device=dev)
        ms = timed(lambda: ops.ntxent(z, 1.0))
        print(json.dumps({"row": "a13 NT_Xent (K8/K9)", "what": f"fwd+bwd+cosine n={n}", "ms": round(ms, 4),
                          "GFLOPs": round(6.0 * n * n * 128 / ms / 1e6, 1)}), flush=True)
# ---- PPO act + update (C4 share: 64 bags, T=6 -> 5 stored steps, K_epochs 3)
if "ppo" in only or not only:
    B, K = 64, 10
    ppo = rlmil.PPO(512, 512, 512, False, action_std=0.5, lr=1e-5, gamma=0.1, K_epochs=3, action_size=K)
    mem = rlmil.Memory()
    st = torch.randn((B, 512), generator=g, device=dev)
    ms_act = timed(lambda: (mem.clear_memory(), ppo.select_action(st, mem, restart_batch=True)))
    def upd():
        mem.clear_memory()
        for t in range(5):
            ppo.select_action(st, mem, restart_batch=(t == 0))
            mem.rewards.append(torch.randn((1, B), device=dev) * 0.1)
        ppo.update(mem)
    ms_upd = timed(upd, reps=5)
    print(json.dumps({"row": "a15-a17 PPO (K10/K11) C4/8", "what": "select_action (one step, 64 bags)", "ms": round(ms_act, 4)}), flush=True)
    print(json.dumps({"row": "a15-a17 PPO (K10/K11) C4/8", "what": "5 acts + update (K_epochs=3, 320 rows)", "ms": round(ms_upd, 4)}), flush=True)
# ---- Full_layer step
if "gru" in only or not only:
    fc = rlmil.Full_layer(512, 1024, True, 128).to(dev)
    x = torch.randn((128, 512), generator=g, device=dev)
    def f():
        with torch.no_grad(): fc(x, restart=True); fc(x, restart=False)
    print(json.dumps({"row": "a14 Full_layer (K7)", "what": "2 GRU steps + projection, 128 rows", "ms": round(timed(f), 4)}), flush=True)
# ---- f1: forming a C4 batch (64 raw slides x 8192 x 512): per-step upload as the reference does vs the HBM-resident store
if "store" in only or not only:
    from murcl_amd.utils.datasets import DeviceSlideStore
    S_, B, N, K = 128, 64, 8192, 10
    rng = np.random.default_rng(985)
    host = [torch.from_numpy((np.abs(rng.standard_normal((N, 512), dtype=np.float32)) * 0.5)) for _ in range(S_)]
    cls = []
    for _ in range(S_):
        lab = rng.integers(0, K, N)
        cls.append([np.nonzero(lab == k)[0].tolist() for k in range(K)])
    class _DS:
        def __len__(self): return S_
        def __getitem__(self, i): return host[i], cls[i], 0, str(i)
    t0 = time.perf_counter(); store = DeviceSlideStore.from_dataset(_DS(), dev, dtype=torch.bfloat16); torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    def wall(fn, reps=5):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    pick = rng.permutation(S_)[:B]
    ms_ref = wall(lambda: BagPack.from_lists([host[i].to(dev, non_blocking=True) for i in pick], [cls[i] for i in pick], dtype=torch.bfloat16))
    ms_store = wall(lambda: store.pack(pick), reps=50)
    print(json.dumps({"row": "f1 device-resident split (8(f) rank 1)", "what": f"form one batch of {B} slides x {N} x 512",
                      "per_step_upload_ms": round(ms_ref, 2), "resident_store_pack_ms": round(ms_store, 4),
                      "store_GiB": round(store.bytes() / 2 ** 30, 2), "one_time_upload_s": round(build_s, 2)}), flush=True)
# ---- f4: the clustering pre-step (features_clustering.py): one slide of 20 000 patches x 512, 10 clusters
if "kmeans" in only or not only:
    from murcl_amd.utils.clustering import kmeans, lloyd
    N, d, K = 20000, 512, 10
    rng = np.random.default_rng(985)
    cent = rng.standard_normal((K, d)).astype(np.float32)
    X = (cent[rng.integers(0, K, N)] * 0.6 + np.abs(rng.standard_normal((N, d), dtype=np.float32)) * 0.5)
    Xd = torch.from_numpy(X).to(dev)
    init = Xd[torch.linspace(0, N - 1, K).long()].clone()
    lloyd(Xd, init, max_iter=3); kmeans(Xd, K, seed=1, n_init=1); torch.cuda.synchronize()      # warm: first-use module loads
    t0 = time.perf_counter(); _, _, inertia, it = lloyd(Xd, init, max_iter=300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); _, _, best = kmeans(Xd, K, seed=985, n_init=3); torch.cuda.synchronize(); dt3 = time.perf_counter() - t0
    from sklearn.cluster import KMeans
    t0 = time.perf_counter(); ref = KMeans(n_clusters=K, random_state=985).fit(X); dts = time.perf_counter() - t0
    print(json.dumps({"row": "f4 k-means pre-step (8(f) rank 4)", "what": f"one slide {N} x {d}, K={K}",
                      "lloyd_ms_per_iteration": round(dt / (it + 1) * 1e3, 4), "iterations": it,
                      "GBps_of_X": round(N * d * 4 / (dt / (it + 1)) / 1e9, 1),
                      "kmeans++_3_starts_ms": round(dt3 * 1e3, 1), "inertia": round(best, 1),
                      "sklearn_host_ms": round(dts * 1e3, 1), "sklearn_inertia": round(float(ref.inertia_), 1)}), flush=True)
# ---- a21: the supervised RL-MIL stage-1 step (train_RLMIL.py step bodies), 64 raw bags x 8192 -> T = 6 sub-bags of 1024
if "supervised" in only or not only:
    from murcl_amd.optim import FlatAdam
    from murcl_amd.train_RLMIL import create_model, supervised_step
    B, N, K = 64, 8192, 10
    rng = np.random.default_rng(985)
    feats = [(torch.randn((N, 512), generator=g, device=dev).abs() * 0.5) for _ in range(B)]
    cls = []
    for _ in range(B):
        lab = rng.integers(0, K, N)
        cls.append([np.nonzero(lab == k)[0].tolist() for k in range(K)])
    pack = BagPack.from_lists(feats, cls, dtype=torch.bfloat16)
    labels = torch.from_numpy(rng.integers(0, 2, B)).to(dev)
    for arch in ("ABMIL", "CLAM_SB", "DSMIL"):
        model, fc = create_model(arch, 512, 2, dev, dtype=torch.bfloat16)
        opt = FlatAdam([{"params": list(model.parameters()) + list(fc.parameters()), "lr": 1e-4}])
        mem = rlmil.Memory()
        def step(at_once=True):
            supervised_step(arch, model, fc, None, opt, pack, labels, mem, T=6, feat_size=1024, batch_patch_steps=at_once)
        def wall(fn, reps=10):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
        ms, ms_loop = wall(step), wall(lambda: step(False))
        print(json.dumps({"row": "a21 supervised RL-MIL stage-1 step", "what": f"{arch}: {B} raw bags x {N} -> T=6 x 1024, bf16",
                          "ms_per_step": round(ms, 3), "bags_per_s": round(B / ms * 1e3, 1),
                          "ms_per_step_one_aggregator_pass_per_patch_step": round(ms_loop, 3)}), flush=True)
