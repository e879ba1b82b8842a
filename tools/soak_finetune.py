"""Soak: train_RLMIL.main (supervised RL-MIL, stage 1 -> 2 -> 3) on synthetic labelled slides; per-epoch wall time."""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import train_RLMIL
tmp = tempfile.mkdtemp(prefix="soakft_")
arch = os.environ.get("SOAK_ARCH", "ABMIL")
epochs = int(os.environ.get("SOAK_EPOCHS", "6"))
base = ["--synthetic", "256,64,64,8192", "--num_clusters", "10", "--feat_size", "1024", "--T", "6", "--batch_size", "64", "--arch", arch,
        "--device", "0", "--exist_ok", "--base_save_dir", tmp, "--dataset", "Soak", "--train_method", "scratch", "--save_model"]
for stage in (1, 2, 3):
    t0 = time.time()
    train_RLMIL.main(base + ["--train_stage", str(stage), "--epochs", str(epochs)])
    torch.cuda.synchronize()
    print(f"SOAK-FT {arch} stage {stage}: {epochs} epochs of 4 train steps + valid(64) + test(64); total {time.time() - t0:.1f} s "
          "(run with two SOAK_EPOCHS values and difference them: the synthetic split takes ~10 s to generate)", flush=True)
