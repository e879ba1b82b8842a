"""Soak: the three-stage pre-training script at the C4 per-GPU share (64 bags x 8192 raw patches, T = 6, feat_size 1024, bf16) for a
few hundred steps per stage; prints the loss per epoch (the script's own output), peak memory and steps/s per stage."""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd import train_MuRCL
tmp = tempfile.mkdtemp(prefix="soak_")
slides, epochs = int(os.environ.get("SOAK_SLIDES", "256")), int(os.environ.get("SOAK_EPOCHS", "6"))
base = ["--synthetic", f"{slides},8192", "--num_clusters", "10", "--feat_size", "1024", "--T", "6", "--batch_size", "64", "--data_repeat", "4",
        "--arch", "ABMIL", "--device", "0", "--scheduler", "CosineAnnealingLR", "--exist_ok", "--base_save_dir", tmp, "--dataset", "Soak",
        "--model_dim", "512", "--fc_lr", "0.00005"]
import builtins
from murcl_amd.utils import checkpoint as _C
if os.environ.get("SOAK_NO_CKPT") == "1":                 # floor: the epoch loop without any checkpoint work
    _C.CheckpointWriter.submit = lambda self, *a, **k: None
    _C.make_state = lambda *a, **k: None
_print, stamps = builtins.print, []
def tprint(*a, **k):                                   # time-stamp the script's per-epoch lines
    if a and isinstance(a[0], str) and a[0].startswith("Loss:"):
        stamps.append(time.time())
    _print(*a, **k)
builtins.print = tprint
for stage in (1, 2, 3):
    torch.cuda.reset_peak_memory_stats()
    stamps.clear()
    train_MuRCL.main(base + ["--train_stage", str(stage), "--epochs", str(epochs), "--ppo_epochs", str(epochs)])
    torch.cuda.synchronize()
    per = slides * 4 // 64
    ep = [(stamps[i + 1] - stamps[i]) / per * 1e3 for i in range(len(stamps) - 1)]
    _print(f"SOAK stage {stage}: {per} steps per epoch, ms/step per epoch (incl. the per-epoch checkpoint): " + " ".join(f"{x:.2f}" for x in ep)
           + f"; peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
