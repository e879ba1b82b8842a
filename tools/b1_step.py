"""The supervised RL-MIL step at the batch size the reference's launch scripts use (runs/finetune.sh, scratch.sh, linear.sh:
--batch_size 1): one slide, T = 6 patch steps of 1024 patches.  Wall time per step, host enqueue time, launches per step."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd.models import rlmil
from murcl_amd.optim import FlatAdam
from murcl_amd.train_RLMIL import create_model, supervised_step
from murcl_amd.utils.datasets import BagPack
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
rng = np.random.default_rng(5)
B, R = int(os.environ.get("B1_BAGS", "1")), 8192
feats = [(torch.randn((R, 512), generator=g, device=dev).abs() * 0.5) for _ in range(B)]
cl = []
for _ in range(B):
    lab = rng.integers(0, 10, R)
    cl.append([np.nonzero(lab == k)[0].tolist() for k in range(10)])
pack = BagPack.from_lists(feats, cl, dtype=torch.bfloat16)
labels = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(1)).to(dev)
for stage in (1, 3):
    for arch in ("ABMIL", "CLAM_SB", "DSMIL"):
        model, fc = create_model(arch, 512, 2, dev, dtype=torch.bfloat16)
        opt = FlatAdam([{"params": list(model.parameters()) + list(fc.parameters()), "lr": 1e-4}])
        ppo = rlmil.PPO(512, 512, 512, False, action_size=10) if stage == 3 else None
        mem = rlmil.Memory()
        def step():
            supervised_step(arch, model, fc, ppo, opt, pack, labels, mem, T=6, feat_size=1024, train_stage=stage)
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(json.dumps({"arch": arch, "stage": stage, "bags": B, "ms_per_step": round((t2 - t0) / n * 1e3, 3),
                          "host_enqueue_ms": round((t1 - t0) / n * 1e3, 3)}), flush=True)
if os.environ.get("B1_PROFILE") == "1":
    import cProfile, pstats
    parch, pstage = os.environ.get("B1_ARCH", "ABMIL"), int(os.environ.get("B1_STAGE", "1"))
    model, fc = create_model(parch, 512, 2, dev, dtype=torch.bfloat16)
    opt = FlatAdam([{"params": list(model.parameters()) + list(fc.parameters()), "lr": 1e-4}])
    ppo = rlmil.PPO(512, 512, 512, False, action_size=10) if pstage == 3 else None
    mem = rlmil.Memory()
    def step():
        supervised_step(parch, model, fc, ppo, opt, pack, labels, mem, T=6, feat_size=1024, train_stage=pstage)
    for _ in range(5): step()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
