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


def make_state(epoch, model, fc, optimizer=None, ppo=None, on_device=False):
    """The reference's checkpoint dictionary; tensors on the host so that either side can ``torch.load`` it anywhere.
    ``on_device``: device clones made in stream order instead (no host synchronisation; see ``EpochSnapshots``)."""
    if on_device:
        take = lambda sd: {k: v.detach().clone() for k, v in sd.items()}  # noqa: E731
    else:
        take = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}  # noqa: E731

    def opt_state(o):
        if o is None:
            return None
        try:
            return o.state_dict(on_device=on_device)
        except TypeError:                                  # a torch.optim optimizer: its state_dict holds the LIVE state tensors
            sd = o.state_dict()
            return _map_tensors(sd, lambda t: t.detach().clone()) if on_device else sd
    ppo_opt = getattr(ppo, "optimizer", None) if ppo is not None else None
    return {"epoch": epoch, "model_state_dict": take(model.state_dict()), "fc": take(fc.state_dict()),
            "optimizer": opt_state(optimizer),                                                # train_MuRCL.py:326-327
            "ppo_optimizer": opt_state(ppo_opt),
            "policy": take(ppo.policy.state_dict()) if ppo is not None else None}


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


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class EpochSnapshots:
    """The end-of-epoch state and loss of a training loop WITHOUT draining the GPU queue.

    The reference reads the epoch's mean loss back and saves the whole state at every epoch boundary (train_MuRCL.py:315-330).
    Done naively that is a queue drain (``.item()``), 4-9 ms of device->host copies during which the GPU idles, and a refill:
    6-11 ms per epoch, 0.5 ms per step at 16-step epochs.  ``capture`` instead, in stream order and without touching the host:
    clones the state on the device (111 MB at HBM speed: ~50 us), records an event, and lets a side stream copy the clones and
    the loss scalar into pinned host memory behind that event, while the training stream goes straight on with the next epoch.
    ``poll`` hands back the snapshots whose copies have completed, oldest first: (epoch, loss value, host state); the caller does
    the bookkeeping (best model, csv, ``CheckpointWriter.submit``) then - typically a few steps into the next epoch."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(self.device)
        self._pending = []

    def capture(self, epoch, loss_dev, model, fc, optimizer=None, ppo=None):
        dev_state = make_state(epoch, model, fc, optimizer, ppo, on_device=True)
        loss_dev = loss_dev.detach().reshape(1).float().clone()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))

        def to_host(t):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            return h
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            host_state = _map_tensors(dev_state, to_host)
            loss_host = to_host(loss_dev)
            done = torch.cuda.Event()
            done.record(self.side)
        # the device clones stay referenced until their copies have completed (they were allocated on the training stream)
        self._pending.append((done, epoch, loss_host, host_state, dev_state, loss_dev))

    def poll(self, block=False):
        out = []
        while self._pending:
            done = self._pending[0][0]
            if block:
                done.synchronize()
            elif not done.query():
                break
            _, epoch, loss_host, host_state, _, _ = self._pending.pop(0)
            out.append((epoch, float(loss_host[0]), host_state))
        return out
