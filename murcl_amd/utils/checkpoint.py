"""Checkpoint interchange with the reference.

The modules keep the reference's state-dict names and shapes, so a file written by either side loads on the other with
``strict=True``.  This module adds the reference's loading *policies* (train_RLMIL.py:118-193, train_MuRCL.py:104-141) and
its writer (utils/general.py:207-211):

* ``strip_pretrained_encoder``: a MuRCL pre-training checkpoint stores ``CL(encoder)``; fine-tuning / linear evaluation
  keep the ``encoder.*`` entries without the prefix and drop ``encoder.fc*`` / ``encoder.classifiers*`` (the supervised
  head is trained from scratch);
* ``load_pretrained``: that state into an aggregator with ``strict=False``; ``train_method='linear'`` freezes everything
  except ``fc*`` / ``classifiers*`` / ``instance_classifiers*``;
* ``load_stage``: stage 2 / 3 continue from ``../stage_{k-1}/model_best.pth.tar`` (aggregator + head, and the policy);
* ``save_checkpoint``: ``checkpoint.pth.tar`` plus a copy as ``model_best.pth.tar`` when it is the best so far.
"""
import os
import queue
import shutil
import threading
from pathlib import Path

import torch

CHECKPOINT_KEYS = ("epoch", "model_state_dict", "fc", "optimizer", "ppo_optimizer", "policy")   # train_MuRCL.py:322-329


def _read(ckpt):
    return torch.load(ckpt, map_location="cpu") if isinstance(ckpt, (str, os.PathLike)) else ckpt


def strip_pretrained_encoder(model_state_dict):
    out = {}
    for k, v in model_state_dict.items():
        if k.startswith("encoder") and not k.startswith("encoder.fc") and not k.startswith("encoder.classifiers"):
            out[k[len("encoder."):]] = v
    return out


def load_pretrained(model, ckpt, train_method="finetune"):
    """-> missing keys (the parts that start from their initialisation)."""
    if train_method not in ("finetune", "linear"):
        raise ValueError(train_method)
    msg = model.load_state_dict(strip_pretrained_encoder(_read(ckpt)["model_state_dict"]), strict=False)
    if train_method == "linear":
        freeze_backbone(model)
    return list(msg.missing_keys)


def freeze_backbone(model):
    """Linear evaluation: only the classification heads stay trainable (train_RLMIL.py:137-143)."""
    kept = []
    for n, p in model.named_parameters():
        if n.startswith("fc") or n.startswith("classifiers") or n.startswith("instance_classifiers"):
            kept.append(n)
        else:
            p.requires_grad = False
    return kept


def stage_checkpoint_path(save_dir, train_stage):
    return str(Path(save_dir).parent / f"stage_{train_stage - 1}" / "model_best.pth.tar")


def load_stage(model, fc, ppo, ckpt, policy_ckpt=None, load_policy=True):
    """Aggregator + head from ``ckpt``; the policy (both copies) from ``policy_ckpt`` (default: the same file).
    ``load_policy=False``: keep the freshly initialised sampler - stage 2 of training from scratch, whose stage-1
    checkpoint has no policy yet (train_RLMIL.py:199-214, train_MuRCL.py:104-122)."""
    ck = _read(ckpt)
    model.load_state_dict(ck["model_state_dict"])
    fc.load_state_dict(ck["fc"])
    if ppo is not None and load_policy:
        pol = _read(policy_ckpt)["policy"] if policy_ckpt is not None else ck.get("policy")
        if pol is None:
            raise KeyError("checkpoint holds no 'policy' (stage 2 takes it from the pre-training checkpoint)")
        ppo.policy.load_state_dict(pol)
        ppo.policy_old.load_state_dict(pol)
    return ck.get("epoch")


def make_state(epoch, model, fc, optimizer=None, ppo=None):
    """The reference's checkpoint dictionary; tensors on the host so that either side can ``torch.load`` it anywhere."""
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}  # noqa: E731
    ppo_opt = getattr(ppo, "optimizer", None) if ppo is not None else None
    return {"epoch": epoch, "model_state_dict": cpu(model.state_dict()), "fc": cpu(fc.state_dict()),
            "optimizer": optimizer.state_dict() if optimizer is not None else None,           # train_MuRCL.py:326-327
            "ppo_optimizer": ppo_opt.state_dict() if ppo_opt is not None else None,
            "policy": cpu(ppo.policy.state_dict()) if ppo is not None else None}


def save_checkpoint(state, is_best, checkpoint, filename="checkpoint.pth.tar"):
    os.makedirs(checkpoint, exist_ok=True)
    path = os.path.join(checkpoint, filename)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(checkpoint, "model_best.pth.tar"))
    return path


class CheckpointWriter:
    """``save_checkpoint`` off the training thread.

    The reference writes ``checkpoint.pth.tar`` (+ the ``model_best`` copy) at the end of EVERY epoch (train_MuRCL.py:322-330), and
    its epochs are short - a few hundred slides x ``--data_repeat`` / ``--batch_size`` = 16 to 60 steps.  With the step at 5-7 ms the
    110 MiB file (aggregator + head + sampler + both Adam states) costs 50-85 ms of ``torch.save`` + copy per epoch: 1 - 4 ms per
    step, a fifth to a third of the training time.  Here the training thread only snapshots the state to host memory (``make_state``:
    4-9 ms of device->host copies); one background thread serialises and writes the files in submission order.  ``submit`` waits for
    the previous write to finish (at most one snapshot is pending), ``close`` joins; a failure in the thread is raised by the next
    ``submit`` / ``close``.  Files, names and contents are those of ``save_checkpoint``."""

    def __init__(self):
        self._q = queue.Queue(maxsize=1)
        self._err = None
        self._t = threading.Thread(target=self._run, name="murcl-checkpoint-writer", daemon=True)
        self._t.start()

    def _run(self):
        while True:
            job = self._q.get()
            try:
                if job is None:
                    return
                if self._err is None:
                    save_checkpoint(*job)
            except BaseException as e:                           # noqa: BLE001 - handed to the training thread
                self._err = e
            finally:
                self._q.task_done()

    def _raise(self):
        if self._err is not None:
            err, self._err = self._err, None
            raise RuntimeError("writing a checkpoint failed") from err

    def submit(self, state, is_best, checkpoint, filename="checkpoint.pth.tar"):
        self._q.join()                                           # the previous file is complete before the next snapshot queues
        self._raise()
        self._q.put((state, is_best, checkpoint, filename))

    def close(self):
        if self._t.is_alive():
            self._q.join()
            self._q.put(None)
            self._t.join()
        self._raise()
