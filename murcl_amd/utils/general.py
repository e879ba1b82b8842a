"""Host-side bookkeeping of the two entry scripts (reference: utils/general.py, train_MuRCL.py:18-55,346-383,
train_RLMIL.py:20-58,1005-1057): where a run's files go, its csv logs, early stopping, device choice.

None of this is on the step's critical path; it exists so that the reference's launch scripts (runs/*.sh) drive
murcl_amd unchanged - same flags, same result directories (``.../stage_k`` with ``../stage_{k-1}/model_best.pth.tar`` as
the next stage's default input), same files (``args.yaml``, ``losses.csv``, ``results.csv``, ``checkpoint.pth.tar``,
``model_best.pth.tar``).
"""
import csv
import glob
import os
import random
import re
from pathlib import Path

import numpy as np
import torch


def init_seeds(seed=0):
    """utils/general.py:17-28 (the cudnn switches have no MIOpen-free counterpart here: no library kernels on the path)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def run_dir_name(args, script):
    """The directory the reference derives from the hyper-parameters when ``--save_dir`` is not given
    (train_MuRCL.py:18-55 for ``script='MuRCL'``, train_RLMIL.py:20-58 for ``script='RLMIL'``)."""
    parts = [f"{args.dataset}_np_{args.feat_size}", script]
    if script == "MuRCL":
        parts.append("_".join([f"T{args.T}", f"pd{args.projection_dim}", f"as{args.action_std}", f"pg{args.ppo_gamma}",
                               f"tau{args.temperature}", f"alpha{args.alpha}"]))
        arch = {"ABMIL": [f"L{args.model_dim}", f"D{args.D}", f"dpt{args.dropout}"],
                "CLAM_SB": [f"size_{args.size_arg}", f"ks_{args.k_sample}"]}
    else:
        parts.append("_".join([f"T{args.T}", f"as{args.action_std}", f"pg{args.ppo_gamma}", f"phd{args.policy_hidden_dim}",
                               f"fhd{args.fc_hidden_dim}"]))
        arch = {"ABMIL": [f"L{args.L}", f"D{args.D}", f"dpt{args.dropout}"], "DSMIL": ["default"],
                "CLAM_SB": [f"size_{args.size_arg}", f"ks_{args.k_sample}", f"bw_{args.bag_weight}"]}
    if args.arch not in arch:
        raise ValueError(args.arch)
    parts += [args.arch, "_".join(arch[args.arch])]
    if script == "RLMIL":
        parts.append(args.train_method)
    parts.append("exp" if args.save_dir_flag is None else f"exp_{args.save_dir_flag}")
    parts += [f"seed{args.seed}", f"stage_{args.train_stage}"]
    return str(Path(args.base_save_dir).joinpath(*parts))


def next_free_path(path, exist_ok=True, sep=""):
    """``path`` itself when it is free or may be reused, else ``path{sep}N`` with the next unused N >= 2
    (utils/general.py:42-53)."""
    path = Path(path)
    if not path.exists() or exist_ok:
        return str(path)
    taken = []
    for d in glob.glob(f"{path}{sep}*"):
        m = re.search(rf"{re.escape(path.stem)}{re.escape(sep)}(\d+)", d)
        if m:
            taken.append(int(m.group(1)))
    return f"{path}{sep}{max(taken) + 1 if taken else 2}"


def prepare_run_dir(args, script):
    """run() of both scripts: resolve ``args.save_dir``, avoid clobbering unless ``--exist_ok``, create it."""
    args.save_dir = run_dir_name(args, script) if args.save_dir is None else str(Path(args.base_save_dir) / args.save_dir)
    args.save_dir = next_free_path(args.save_dir, exist_ok=args.exist_ok, sep="_")
    Path(args.save_dir).mkdir(parents=True, exist_ok=True)
    return args.save_dir


def dump_args(args, save_dir):
    import yaml
    plain = {k: (str(v) if isinstance(v, (torch.device, Path)) else v) for k, v in vars(args).items()}
    with open(Path(save_dir) / "args.yaml", "w") as fp:
        yaml.dump(plain, fp, sort_keys=False)


def pick_device(device_flag, local_rank=0):
    """``--device`` (train_MuRCL.py:356-360: a CUDA_VISIBLE_DEVICES string such as '3' or '0,1,2,3', or 'cpu').

    The reference hands the list to DataParallel; murcl_amd runs one process per GPU, so process ``local_rank`` takes the
    ``local_rank``-th listed device.  A listed ordinal this box does not have falls back to ``cuda:local_rank`` with a
    warning (the reference silently drops to the CPU there); 'cpu' is refused - the kernels have no CPU path."""
    flag = str(device_flag).strip()
    if flag == "cpu":
        raise RuntimeError("--device cpu: murcl_amd runs on MI355X only (no CPU path for its kernels)")
    ids = [int(v) for v in flag.split(",") if v.strip() != ""] or [local_rank]
    local_world = max(int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))), local_rank + 1)
    if len(ids) >= local_world:
        want = ids[local_rank]
    else:
        # fewer listed devices than local ranks (e.g. the inherited default --device "3" under a launcher of N ranks):
        # wrapping around would seat several ranks on one GPU, which RCCL refuses - every rank takes its own ordinal
        if local_rank == 0:
            print(f"--device {flag}: lists {len(ids)} device(s) for {local_world} local ranks; rank r uses cuda:r", flush=True)
        want = local_rank
    n = torch.cuda.device_count()
    if want >= n:
        print(f"--device {flag}: cuda:{want} does not exist here ({n} device(s)); using cuda:{local_rank}", flush=True)
        want = local_rank
    return torch.device("cuda", want)


class CsvLog:
    """A csv file written row by row (utils/general.py:88-105): recreated at construction, header first."""

    def __init__(self, filename, header=None):
        self.filename = str(filename)
        if os.path.exists(self.filename):
            os.remove(self.filename)
        if header is not None:
            self.write_row(header)

    def write_row(self, row):
        with open(self.filename, "a+", newline="") as fp:
            csv.writer(fp).writerow(row)


class Best:
    """Running best of a metric and the epoch it occurred in (utils/general.py:128-158)."""

    def __init__(self, order="max"):
        if order not in ("max", "min"):
            raise ValueError(order)
        self.order, self.best, self.epoch = order, float("-inf") if order == "max" else float("inf"), 0

    def compare(self, val, epoch=None, inplace=False):
        better = val > self.best if self.order == "max" else val < self.best
        if better and inplace:
            self.best = val
            if epoch is not None:
                self.epoch = epoch
        return better


class EarlyStop:
    """Stop when the tracked best has not changed for ``patience`` consecutive epochs (utils/general.py:71-85)."""

    def __init__(self, patience=5):
        self.patience, self.base, self.same = patience, (), 0

    def update(self, variable):
        if variable == self.base:
            self.same += 1
        else:
            self.same, self.base = 1, variable

    def is_stop(self):
        return self.same >= self.patience


class Mean:
    """Weighted running mean (utils/general.py:108-125)."""

    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, val, n=1):
        self.sum += float(val) * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(1, self.count)


def tensorboard_writer(save_dir, enabled):
    """``--use_tensorboard``: a SummaryWriter when the tensorboard package exists in this environment, else None with a note."""
    if not enabled:
        return None
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(save_dir)
    except Exception as e:                                    # the package is optional
        print(f"--use_tensorboard: tensorboard is not available here ({type(e).__name__}); scalars go to the csv logs only", flush=True)
        return None
