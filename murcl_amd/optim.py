"""Flat-buffer Adam on the HIP kernel (torch.optim.Adam semantics; train_MuRCL.py:154-171).

Parameters of each group are re-seated as views of one contiguous f32 buffer, and so are their
``.grad``s: ``zero_grad`` is one memset, ``step`` one kernel launch per group, and a data-parallel
gradient all-reduce is one collective over the flat gradient buffer (no per-tensor buckets).

Because every ``.grad`` exists (zeroed) before backward starts, the backward kernels add weight and bias gradients
straight into the flat buffer (``functional.set_direct_grad``; pass ``direct_grad=False`` to keep autograd's own
AccumulateGrad path, e.g. when per-parameter hooks are registered).  ``step`` also clears the gradient buffer in its
own pass (``fused_zero``), so the ``zero_grad`` that follows costs nothing; gradients are therefore zero AFTER ``step``.
"""
import torch

from . import ops


class FlatAdam:
    def __init__(self, param_groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, direct_grad=True, fused_zero=True):
        from . import functional
        functional.set_direct_grad(direct_grad)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.fused_zero, self._maybe_dirty = fused_zero, False      # the flat grads start as zeros
        self.groups = []
        self.step_count = 0
        self._owned = set()
        for g in param_groups:
            params = [p for p in g["params"] if p.requires_grad]
            n = sum(p.numel() for p in params)
            dev = params[0].device
            flat_p = torch.empty(n, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
            off = 0
            for p in params:
                k = p.numel()
                flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + k].view_as(p.data)
                p.grad = flat_g[off:off + k].view_as(p.data)
                ops.manage_param(p)                           # every raw update of p goes through step() below
                self._owned.add(p.data_ptr())
                off += k
            self.groups.append(dict(params=params, lr=g["lr"], p=flat_p, g=flat_g,
                                    m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p)))

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

    def step(self):
        self.step_count += 1
        for g in self.groups:
            ops.adam_step(g["p"], g["g"], g["m"], g["v"], g["lr"], self.betas, self.eps, self.weight_decay,
                          self.step_count, zero_grad=self.fused_zero)
        ops.refresh_views(self._owned)             # cached compute-dtype / transposed weight views: one launch
        if self.fused_zero:
            self._maybe_dirty = False
