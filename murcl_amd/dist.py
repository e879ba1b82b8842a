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
