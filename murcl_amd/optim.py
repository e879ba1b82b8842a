"""Flat-buffer optimizers on the HIP kernels (torch.optim.Adam / torch.optim.SGD semantics; train_MuRCL.py:154-171).

Parameters of each group are re-seated as views of one contiguous f32 buffer, and so are their
``.grad``s: ``zero_grad`` is one memset, ``step`` one kernel launch per group, and a data-parallel
gradient all-reduce is one collective over the flat gradient buffer (no per-tensor buckets).

Because every ``.grad`` exists (zeroed) before backward starts, the backward kernels add weight and bias gradients
straight into the flat buffer (``functional.set_direct_grad``; pass ``direct_grad=False`` to keep autograd's own
AccumulateGrad path, e.g. when per-parameter hooks are registered).  ``step`` also clears the gradient buffer in its
own pass (``fused_zero``), so the ``zero_grad`` that follows costs nothing; gradients are therefore zero AFTER ``step``.

Parameters that no backward pass reached since the last step are skipped like torch's optimizers skip ``grad is None``
(no weight decay, no moment / momentum update, no step count): writers announce themselves through
``functional._touch`` / an autograd post-accumulate hook; ``step`` launches the kernel over the contiguous runs of
touched parameters (one launch per group when everything was touched, or when only a tail such as ABMIL's unused
``fc`` was not).
"""
import math

import torch

from . import ops


class _FlatOptimizer:
    STATE = ()                                   # names of the per-element state buffers

    def __init__(self, param_groups, weight_decay=0.0, direct_grad=True, fused_zero=True):
        from . import functional
        functional.set_direct_grad(direct_grad)
        self.weight_decay = weight_decay
        self.fused_zero, self._maybe_dirty = fused_zero, False      # the flat grads start as zeros
        self.groups = []
        self.step_count = 0
        self._owned, self._ids = set(), set()
        self._pstep = {}                                      # id(param) -> number of steps it took part in (torch: state['step'])
        self._fn = functional
        for g in param_groups:
            params = [p for p in g["params"] if p.requires_grad]
            n = sum(p.numel() for p in params)
            dev = params[0].device
            flat_p = torch.empty(n, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
            off, segs = 0, []
            for p in params:
                k = p.numel()
                flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + k].view_as(p.data)
                p.grad = flat_g[off:off + k].view_as(p.data)
                ops.manage_param(p)                           # every raw update of p goes through step() below
                self._owned.add(p.data_ptr())
                self._ids.add(id(p))
                p.register_post_accumulate_grad_hook(functional._touch)     # gradients that arrive through autograd
                segs.append((id(p), off, off + k))
                off += k
            grp = dict(params=params, lr=g["lr"], initial_lr=g["lr"], p=flat_p, g=flat_g, segs=segs)
            for name in self.STATE:
                grp[name] = torch.zeros_like(flat_p)
            self.groups.append(grp)

    @property
    def param_groups(self):            # lr schedulers poke group['lr']
        return self.groups

    def flat_grads(self):
        return [g["g"] for g in self.groups]

    def zero_grad(self):
        if self._maybe_dirty:                      # step() already cleared the buffers in its own pass
            for g in self.groups:
                g["g"].zero_()
        self._maybe_dirty = True                   # a backward pass follows

    def mark_all_touched(self):
        """Treat every parameter as having a gradient (callers that fill the flat gradient buffers themselves)."""
        self._fn._TOUCHED |= self._ids

    def _runs(self, g):
        """[lo, hi, step] over contiguous parameters that received a gradient and share a step count."""
        touched, pstep, runs = self._fn._TOUCHED, self._pstep, []
        for pid, lo, hi in g["segs"]:
            if pid not in touched:
                continue
            n = pstep[pid] = pstep.get(pid, 0) + 1
            if runs and runs[-1][1] == lo and runs[-1][2] == n:
                runs[-1][1] = hi
            else:
                runs.append([lo, hi, n])
        return runs

    def _launch(self, g, lo, hi, n):
        raise NotImplementedError

    def step(self):
        for g in self.groups:
            for lo, hi, n in self._runs(g):
                self._launch(g, lo, hi, n)
        self._finish_step()

    def _finish_step(self):
        self.step_count += 1
        self._fn._TOUCHED -= self._ids
        pending = getattr(self, "_pending_tick", None)   # a captured Adam step left its live step counter to the next launch
        ticked = ops.refresh_views(self._owned, tick=pending)   # cached compute-dtype / transposed weight views: one launch
        if pending is not None:
            if not ticked:
                ops.replay_tick(pending)           # (no view to refresh: a one-thread launch advances the counter)
            self._pending_tick = None
        if self.fused_zero:
            self._maybe_dirty = False

    # -- checkpointing (the reference stores optimizer.state_dict() under 'optimizer' / 'ppo_optimizer', train_MuRCL.py:326-327)
    def state_dict(self, on_device=False):
        """``on_device``: the state tensors as device clones made in stream order (no host synchronisation) - for
        ``utils.checkpoint.EpochSnapshots``, which moves them to the host on a side stream."""
        take = (lambda t: t.detach().clone()) if on_device else (lambda t: t.detach().cpu())
        return {"kind": type(self).__name__, "step_count": self.step_count,
                "groups": [dict(lr=g["lr"], initial_lr=g["initial_lr"], steps=[self._pstep.get(pid, 0) for pid, _, _ in g["segs"]],
                                **{name: take(g[name]) for name in self.STATE}) for g in self.groups]}

    def load_state_dict(self, sd):
        if sd.get("kind") != type(self).__name__ or len(sd["groups"]) != len(self.groups):
            raise ValueError("optimizer state does not match this optimizer")
        self.step_count = int(sd["step_count"])
        for g, s in zip(self.groups, sd["groups"]):
            g["lr"], g["initial_lr"] = s["lr"], s["initial_lr"]
            for (pid, _, _), n in zip(g["segs"], s["steps"]):
                self._pstep[pid] = int(n)
            for name in self.STATE:
                g[name].copy_(s[name])


class FlatAdam(_FlatOptimizer):
    STATE = ("m", "v")

    def __init__(self, param_groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, direct_grad=True, fused_zero=True):
        self.betas, self.eps = betas, eps
        self.replays = None                                     # device int32[2] while a captured step is being replayed (live_graph)
        super().__init__(param_groups, weight_decay, direct_grad, fused_zero)

    def live_graph(self, on=True):
        """Capture mode for a step that will be REPLAYED as a hipGraph: ``step()`` launches murcl_adam_multi_live, whose step counts
        advance on the device with every replay (a replay is the next optimizer step, not the captured one again).  Call with
        ``on=True`` before capturing (the counter is zeroed), replay, then ``after_replays(k)`` to bring the host-side counts up to
        date and leave the mode."""
        if on:
            dev = self.groups[0]["p"].device
            self.replays = torch.zeros(2, dtype=torch.int32, device=dev)
            self._live_captured, self._live_ids = 0, []
        else:
            self.replays = None

    def after_replays(self, steps):
        """``steps`` optimizer steps have run as replays of the captured step(s) (replays x captured steps per graph): bring the
        host-side counts up to date and leave capture mode.  The capture itself executed nothing but was counted on the host."""
        extra = steps - self._live_captured
        for pid in set(self._live_ids):
            self._pstep[pid] += extra
        self.step_count += extra
        self.replays = None

    def _launch(self, g, lo, hi, n):
        ops.adam_step(g["p"][lo:hi], g["g"][lo:hi], g["m"][lo:hi], g["v"][lo:hi], g["lr"], self.betas, self.eps,
                      self.weight_decay, n, zero_grad=self.fused_zero)

    def step(self):
        """All groups' runs in ONE launch when there are few of them (the usual step: one run per group); the general walk otherwise."""
        live = 0
        if self.replays is not None:
            # every captured launch carries the step count of the FIRST captured step: the device counter advances once per launch
            self._live_ids += [pid for g in self.groups for pid, _, _ in g["segs"] if pid in self._fn._TOUCHED]
            live, self._live_captured = self._live_captured, self._live_captured + 1
        runs = [(g, lo, hi, n - live) for g in self.groups for lo, hi, n in self._runs(g)]
        if not 0 < len(runs) <= ops.ADAM_MAX_JOBS:
            if self.replays is not None:
                raise RuntimeError("FlatAdam.live_graph: the captured step must fit one adam_multi launch")
            for g, lo, hi, n in runs:
                self._launch(g, lo, hi, n)
        else:
            ops.adam_multi([(g["p"][lo:hi], g["g"][lo:hi], g["m"][lo:hi], g["v"][lo:hi], g["lr"], n) for g, lo, hi, n in runs],
                           self.betas, self.eps, self.weight_decay, zero_grad=self.fused_zero, replays=self.replays, tick=False)
            self._pending_tick = self.replays       # advanced by the view refresh that follows (_finish_step)
        self._finish_step()


class FlatSGD(_FlatOptimizer):
    """torch.optim.SGD(params, momentum, nesterov, weight_decay), dampening 0 (train_MuRCL.py:158-163)."""
    STATE = ("buf",)

    def __init__(self, param_groups, momentum=0.0, nesterov=False, weight_decay=0.0, direct_grad=True, fused_zero=True):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        self.momentum, self.nesterov = momentum, nesterov
        super().__init__(param_groups, weight_decay, direct_grad, fused_zero)

    def _launch(self, g, lo, hi, n):
        ops.sgd_step(g["p"][lo:hi], g["g"][lo:hi], g["buf"][lo:hi], g["lr"], self.momentum, self.nesterov,
                     self.weight_decay, first=(n == 1), zero_grad=self.fused_zero)


class LRSchedule:
    """The two schedulers the reference's scripts offer (train_MuRCL.py:174-186), in closed form over the number of
    ``step()`` calls: StepLR(step_size=7, gamma=0.1) and CosineAnnealingLR(T_max, eta_min=1e-6)."""

    def __init__(self, optimizer, name, T_max=None, eta_min=1e-6, step_size=7, gamma=0.1):
        if name not in ("StepLR", "CosineAnnealingLR"):
            raise ValueError(name)
        self.opt, self.name, self.T_max, self.eta_min, self.step_size, self.gamma, self.k = \
            optimizer, name, T_max, eta_min, step_size, gamma, 0

    def lr_at(self, base, k):
        if self.name == "StepLR":
            return base * self.gamma ** (k // self.step_size)
        return self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * k / max(1, self.T_max))) / 2

    def step(self):
        self.k += 1
        for g in self.opt.param_groups:
            g["lr"] = self.lr_at(g["initial_lr"], self.k)


def make_scheduler(optimizer, name, epochs, warmup=0):
    """get_scheduler (train_MuRCL.py:174-186): None for no optimizer / no name."""
    if optimizer is None or name is None:
        return None
    return LRSchedule(optimizer, name, T_max=epochs - warmup)
