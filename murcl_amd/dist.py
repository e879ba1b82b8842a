"""One-process-per-GPU data parallelism for the contrastive step.

Bags shard by WSI: rank r owns ``B_local`` bags and everything per-bag (sub-bag gather, MIL
aggregator, recurrent head) stays local.  The only exchange in the data path is an all-gather
of the projected embeddings z ([2*B_local,128] f32 per rank, 64 KiB at 64 bags) so that every
rank sees the global NT-Xent denominator; each rank then back-propagates only into its own
rows (the kernel's grad_lo/grad_hi window), which is already the exact global gradient for local
bags - no second collective on the activation side.  Parameter gradients are summed with one
all-reduce over the optimizer's flat gradient buffer.  Backend "nccl" is RCCL over xGMI on ROCm;
the same code runs on gloo for the CPU tests (with an injected CPU kernel).
"""
import torch
import torch.distributed as dist


def shard_range(rank, world, b_local):
    return rank * b_local, (rank + 1) * b_local


class _GatheredNTXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_i, z_j, temperature, group, kernel):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        bl = z_i.shape[0]
        local = torch.cat([z_i, z_j], 0).contiguous()                  # [2*bl, P]
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
        # global layout expected by NT_Xent: all view-0 rows (rank-major), then all view-1 rows
        zg = torch.cat([p[:bl] for p in parts] + [p[bl:] for p in parts], 0)
        lo, hi = shard_range(rank, world, bl)
        loss, dz, sim = kernel(zg, temperature, grad_lo=lo, grad_hi=hi)
        Bg = bl * world
        ctx.save_for_backward(dz[lo:hi], dz[Bg + lo:Bg + hi])
        ctx.mark_non_differentiable(sim)
        return loss.reshape(()), sim[lo:hi]

    @staticmethod
    def backward(ctx, dloss, _dsim):
        dzi, dzj = ctx.saved_tensors
        return dzi * dloss, dzj * dloss, None, None, None


def gathered_nt_xent(z_i, z_j, temperature, group=None, kernel=None):
    """Global NT-Xent over all ranks' bags; returns (loss identical on every rank, local cosines)."""
    if kernel is None:
        from . import ops
        kernel = lambda z, t, grad_lo, grad_hi: ops.ntxent(z, t, True, grad_lo, grad_hi)  # noqa: E731
    return _GatheredNTXent.apply(z_i, z_j, float(temperature), group, kernel)


def all_reduce_grads(flat_grads, group=None):
    """Sum the flat gradient buffers over ranks (one collective per optimizer group)."""
    for g in flat_grads:
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)


class OverlappedGradReduce:
    """All-reduce the optimizer's flat gradient buffers with the early groups overlapped with backward.

    In the pre-train step the recurrent head's gradients (19.4 MB, group 1) are final as soon as the backward
    pass reaches the aggregator outputs, long before the encoder gradients (4.7 MB, group 0) exist.  ``arm()``
    registers hooks on those aggregator outputs; when the last one fires, the head group's all-reduce is
    launched asynchronously (RCCL runs on its own stream, ordered after the kernels already queued) and overlaps
    with the aggregator backward; ``finish()`` reduces the remaining groups and waits for everything.
    """

    def __init__(self, optimizer, early_groups=(1,), group=None):
        self.opt, self.early, self.group = optimizer, tuple(early_groups), group
        self._pending, self._works = 0, []

    def arm(self, aggregator_outputs):
        # outputs handed out as row blocks of one tensor (CL.forward) may be consumed through that tensor directly
        # (Full_layer.forward_views), so the hook goes on the common base when there is one
        targets = {}
        for t in aggregator_outputs:
            b = getattr(t, "_base", None)
            b = b if b is not None and b.requires_grad and b.grad_fn is not None else t
            targets[id(b)] = b
        self._works, self._pending = [], len(targets)
        for t in targets.values():
            t.register_hook(self._fired)

    def _fired(self, grad):
        self._pending -= 1
        if self._pending == 0:
            flats = self.opt.flat_grads()
            for gi in self.early:
                self._works.append(dist.all_reduce(flats[gi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return grad

    def finish(self):
        flats = self.opt.flat_grads()
        for gi in range(len(flats)):
            if gi not in self.early or not self._works:
                self._works.append(dist.all_reduce(flats[gi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self._works:
            w.wait()
        self._works = []
