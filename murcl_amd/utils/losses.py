"""Drop-in ``NT_Xent`` (reference: utils/losses.py).

One HIP launch computes the loss, its gradient and the positive-pair cosines; no
[2B,2B,128] broadcast temp and no boolean-mask gather as in losses.py:27-36.
``set_shard(lo, hi)`` restricts the gradient to bags [lo,hi) when the batch given to
``forward`` is the all-gathered global batch (one process per GPU, see murcl_amd/dist.py).
"""
import torch
from torch import nn

from ..functional import NTXentFn
from .views import whole


class NT_Xent(nn.Module):
    def __init__(self, batch_size, temperature):
        super().__init__()
        self.batch_size = batch_size
        self.temperature = temperature
        self._shard = None
        self.last_similarity = None          # cos(z_i[b], z_j[b]) of the latest call (K9)
        self.last_mean = None                # forward_steps under grad: the mean of the T step losses, differentiable

    def set_shard(self, lo, hi):
        self._shard = (int(lo), int(hi))

    def forward_stacked(self, z):
        """``forward(z[:B], z[B:])`` for a contiguous [2B,P] float32 tensor that already holds view 0's rows, then view 1's
        (e.g. one patch step's block of ``Full_layer.forward_view_sequence``): nothing is concatenated."""
        return self._run(z, None, z.shape[0] // 2)

    def forward_steps(self, z):
        """T independent batches at once: z [T,2B,P] float32 (view 0's rows, then view 1's, per step; 2B <= 128)
        -> (losses [T], cosines [T,B]); ``last_similarity`` is the last step's."""
        from ..functional import NTXentSeqFn
        if not torch.is_grad_enabled() or not z.requires_grad:
            from .. import ops
            loss, _, sim = ops.ntxent_batched(z, float(self.temperature), want_grad=False)
            self.last_mean = None
        else:
            loss, sim, self.last_mean = NTXentSeqFn.apply(z, float(self.temperature))     # last_mean: loss.mean() as an output of the node
        self.last_similarity = sim[-1]
        return loss, sim

    def forward(self, z_i, z_j):
        zi, zj = z_i, z_j
        base = whole([z_i, z_j]) if z_i.shape == z_j.shape and z_i.dtype == torch.float32 else None
        if base is not None:
            zi, zj = base, None          # the two views are the halves of one tensor (Full_layer.forward_views): no cat
        return self._run(zi, zj, z_i.shape[0])

    def _run(self, zi, zj, B):
        lo, hi = self._shard if self._shard is not None else (0, B)
        if not torch.is_grad_enabled() or not zi.requires_grad:
            # nobody will differentiate this loss (frozen-encoder stage 2, validation): skip the gradient half of the kernel
            from .. import ops
            z = zi if zj is None else torch.cat([zi, zj], 0)
            loss, _, sim = ops.ntxent(z, float(self.temperature), want_grad=False, grad_lo=lo, grad_hi=hi)
            self.last_similarity = sim
            return loss[0]
        loss, sim = NTXentFn.apply(zi, zj, float(self.temperature), lo, hi)
        self.last_similarity = sim
        return loss
