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


def _gloo_on_device(t, group):
    """True when the group's backend is gloo but the tensor lives in HBM (debug / single-GPU multi-process tests): the
    collective is then staged through host memory.  The production backend is "nccl" (RCCL), which takes device buffers."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_rows(out, local, group=None):
    """``out`` [world * n, ...] <- every rank's ``local`` [n, ...] in rank order (one collective)."""
    if _gloo_on_device(local, group):
        parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, local.cpu(), group=group)
        out.copy_(torch.cat(parts, 0))
    else:
        dist.all_gather_into_tensor(out, local, group=group)
    return out


def all_reduce_sum(t, group=None, async_op=False):
    """In-place SUM all-reduce; returns the work handle when ``async_op``."""
    if _gloo_on_device(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class _Done:
    def wait(self):
        return True


def _whole(z_i, z_j):
    """The tensor whose two halves z_i and z_j are (Full_layer.forward_views hands out ``z.split(B)``), or None."""
    from .utils.views import whole
    return whole([z_i, z_j]) if z_i.shape == z_j.shape else None


class _GatheredNTXent(torch.autograd.Function):
    """z_j None: z_i already is the stacked [2*bl, P] batch (view 0 rows, then view 1 rows) - nothing is concatenated on
    the way in and ONE gradient tensor goes back."""

    @staticmethod
    def forward(ctx, z_i, z_j, temperature, group, kernel):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        local = (z_i if z_j is None else torch.cat([z_i, z_j], 0)).contiguous()
        bl = local.shape[0] // 2
        # one collective into one buffer, used as it arrives: [rank][view][bag] (the kernel's pair_stride layout)
        zg = torch.empty((world * 2 * bl, local.shape[1]), dtype=local.dtype, device=local.device)
        all_gather_rows(zg, local, group)
        lo, hi = shard_range(rank, world, bl)
        loss, dz, sim = kernel(zg, temperature, grad_lo=lo, grad_hi=hi, pair_stride=bl)
        r0 = rank * 2 * bl
        ctx.save_for_backward(dz[r0:r0 + 2 * bl])
        ctx.bl, ctx.joint = bl, z_j is None
        sim = sim[lo:hi]                                   # this rank's bags (the tensor that is returned must be the marked one)
        ctx.mark_non_differentiable(sim)
        ctx.set_materialize_grads(False)
        return loss.reshape(()), sim

    @staticmethod
    def backward(ctx, dloss, _dsim):
        (dz,) = ctx.saved_tensors
        if dloss is None:
            return None, None, None, None, None
        from . import ops
        if not ops.is_unit_grad(dloss):
            dz = dz * dloss
        if ctx.joint:
            return dz, None, None, None, None
        return dz[:ctx.bl], dz[ctx.bl:], None, None, None


def gathered_nt_xent(z_i, z_j, temperature, group=None, kernel=None):
    """Global NT-Xent over all ranks' bags; returns (loss identical on every rank, local cosines)."""
    if kernel is None:
        from . import ops
        kernel = lambda z, t, grad_lo, grad_hi, pair_stride: ops.ntxent(z, t, True, grad_lo, grad_hi, pair_stride)  # noqa: E731
    whole = _whole(z_i, z_j)
    if whole is not None:
        return _GatheredNTXent.apply(whole, None, float(temperature), group, kernel)
    return _GatheredNTXent.apply(z_i, z_j, float(temperature), group, kernel)


def all_reduce_grads(flat_grads, group=None):
    """Sum the flat gradient buffers over ranks (one collective per optimizer group)."""
    for g in flat_grads:
        all_reduce_sum(g, group)


def cap_rccl_channels(world):
    """Call BEFORE ``init_process_group("nccl")`` when ``world`` > 1: RCCL runs one workgroup per channel for the length of a
    collective and its defaults may exceed the CUs ``reserve_cus_for_collectives`` leaves free (256 - ``MURCL_CU_BUDGET``, 8 by
    default) - channels beyond the reserve would push the persistent kernels into a second round again.  Sets ``NCCL_MAX_NCHANNELS``
    to the reserve unless the environment already says otherwise (8 channels carry the 19.4 MB head all-reduce at the xGMI ring's
    ~150 GB/s just as well; raise both knobs together if ``comm.grad_all_reduce_us.*.busbw_GBps`` sits far below that).
    -> the value in force (str) or None."""
    import os
    if world < 2:
        return None
    reserve = max(1, 256 - int(os.environ.get("MURCL_CU_BUDGET", "248")))
    os.environ.setdefault("NCCL_MAX_NCHANNELS", str(reserve))
    return os.environ["NCCL_MAX_NCHANNELS"]


def budget_for_channels(budget, max_channels):
    """The CU budget that leaves RCCL's channel workgroups their CUs: ``budget`` (what MURCL_CU_BUDGET asks for) lowered until the
    reserve 256 - budget covers ``max_channels`` (NCCL_MAX_NCHANNELS; None = not set: RCCL picks its own count, 32 is assumed), in steps
    of 8 (one per XCD), never below 64.  -> (budget, note or None); the note says what was overridden and why - callers print it,
    because a reserve smaller than the channel count puts every overlapped launch into a second round (+ 50 % step time on one GPU,
    profiles/r05_f_cu_thief.txt) without any other symptom."""
    budget = max(64, min(256, int(budget))) & ~7
    if budget >= 256:
        return 256, None                                      # the reserve was switched off on purpose
    channels = 32 if max_channels is None else max(1, int(max_channels))
    need = min(192, (channels + 7) // 8 * 8)
    if 256 - budget >= need:
        return budget, None
    fitted = 256 - need
    why = ("NCCL_MAX_NCHANNELS is not set (dist.cap_rccl_channels was not called before init_process_group): RCCL may run up to 32 "
           "channel workgroups") if max_channels is None else f"NCCL_MAX_NCHANNELS={channels}"
    return fitted, (f"murcl_amd.dist: {why}, more than the {256 - budget} CUs MURCL_CU_BUDGET={budget} leaves free - the overlapped launches "
                    f"are sized for {fitted} CUs instead (set NCCL_MAX_NCHANNELS <= {256 - budget} or lower MURCL_CU_BUDGET to silence this)")


def reserve_cus_for_collectives(group=None):
    """Collectives that overlap the backward pass run RCCL's channel workgroups beside the persistent kernels of the step.  Those
    kernels are ONE round of workgroups with a static share each (one per CU): with even 4 CUs held by somebody else a launch runs
    a second round and the step goes from 1.41 to 2.15 ms (a co-running "CU thief" on one GPU, tools/cu_thief.py,
    profiles/r05_f_cu_thief.txt); sized for 248 CUs the same thieves cost 2-4 %.  So with more than one rank on the nccl backend the
    launches a collective can overlap are sized for ``MURCL_CU_BUDGET`` CUs (default 248 = 8 left to RCCL, ``cap_rccl_channels``;
    256 switches the reserve off).  ``MURCL_CU_BUDGET_SCOPE``: "backward" (default) - only the aggregator's backward launches that
    the head group's all-reduce runs beside (functional.set_overlap_cu_budget; the forward pass and the grouped weight gradients keep
    the full chip: ~2 % instead of ~9 % of the step) - or "all" (every persistent launch, murcl_set_cu_budget).
    -> the budget in force for the overlapped launches, None when nothing was changed."""
    import os
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2 or dist.get_backend(group) != "nccl":
        return None
    from . import functional, ops
    # (round 6) the environment may have said NCCL_MAX_NCHANNELS before cap_rccl_channels could (it only sets a default), or nobody
    # called it: the reserve follows the channel cap that is actually in force, and says so loudly
    cap = os.environ.get("NCCL_MAX_NCHANNELS")
    budget, note = budget_for_channels(os.environ.get("MURCL_CU_BUDGET", "248"), int(cap) if cap and cap.isdigit() else None)
    if note and dist.get_rank(group) == 0:
        import warnings
        warnings.warn(note, RuntimeWarning, stacklevel=2)
    if os.environ.get("MURCL_CU_BUDGET_SCOPE", "backward") == "all":
        return ops.set_cu_budget(budget)
    functional.set_overlap_cu_budget(budget)
    return budget


class OverlappedGradReduce:
    """All-reduce the optimizer's flat gradient buffers in pieces, each as soon as it is final, overlapped with backward.

    In the pre-train step the recurrent head's gradients (19.4 MB, group 1) are final as soon as the backward
    pass reaches the aggregator outputs, long before the encoder gradients (4.7 MB, group 0) exist.  ``arm()``
    registers hooks on those aggregator outputs; when the last one fires, the head group's all-reduce is
    launched asynchronously (RCCL runs on its own stream, ordered after the kernels already queued) and overlaps
    with the aggregator backward.  With ``milestones=True`` the aggregator's backward additionally announces the
    parameters whose gradients its kernels have just completed (``functional.set_grad_milestone``: attention + decoder,
    then encoder layer 3, then layer 2), so only the first encoder layer (1 MB) is left for ``finish()``, which reduces
    whatever has not been submitted yet and waits for everything.  Milestones assume the aggregator runs once per step.
    """

    def __init__(self, optimizer, early_groups=(1,), group=None, milestones=False):
        self.opt, self.early, self.group = optimizer, tuple(early_groups), group
        self.cu_budget = reserve_cus_for_collectives(group)
        self._pending, self._works, self._covered = 0, [], {}
        self._where = {}
        if milestones and hasattr(optimizer, "groups"):
            for gi, g in enumerate(optimizer.groups):
                off = 0
                for p in g["params"]:
                    self._where[id(p)] = (gi, off, off + p.numel())
                    off += p.numel()
            from . import functional
            functional.set_grad_milestone(self.milestone)

    def _submit(self, gi, lo, hi):
        flat = self.opt.flat_grads()[gi]
        self._works.append(all_reduce_sum(flat[lo:hi], self.group, async_op=True))
        self._covered.setdefault(gi, []).append((lo, hi))

    def arm(self, aggregator_outputs):
        # outputs handed out as row blocks of one tensor (CL.forward) may be consumed through that tensor directly
        # (Full_layer.forward_views), so the hook goes on the common base when there is one
        targets = {}
        for t in aggregator_outputs:
            b = getattr(t, "_base", None)
            b = b if b is not None and b.requires_grad and b.grad_fn is not None else t
            targets[id(b)] = b
        self._works, self._covered, self._pending = [], {}, len(targets)
        for t in targets.values():
            t.register_hook(self._fired)

    def _fired(self, grad):
        self._pending -= 1
        if self._pending == 0:
            from . import functional
            functional.flush_deferred()               # weight gradients queued for one launch at the end: the head's are due now
            for gi in self.early:
                self._submit(gi, 0, self.opt.flat_grads()[gi].numel())
        return grad

    def milestone(self, params):
        from . import functional
        functional.flush_deferred()                   # queued weight gradients of the announced layers must exist before their reduce
        spans = sorted(self._where[id(p)] for p in params if id(p) in self._where)
        merged = []
        for gi, lo, hi in spans:
            if gi in self.early:
                continue                              # the whole group is (or will be) reduced by the hook
            if merged and merged[-1][0] == gi and merged[-1][2] == lo:
                merged[-1][2] = hi
            else:
                merged.append([gi, lo, hi])
        for gi, lo, hi in merged:
            self._submit(gi, lo, hi)

    def finish(self):
        for gi, flat in enumerate(self.opt.flat_grads()):
            pos = 0
            for lo, hi in sorted(self._covered.get(gi, [])):
                if lo > pos:
                    self._submit(gi, pos, lo)
                pos = max(pos, hi)
            if pos < flat.numel():
                self._submit(gi, pos, flat.numel())
        for w in self._works:
            w.wait()
        self._works, self._covered = [], {}
