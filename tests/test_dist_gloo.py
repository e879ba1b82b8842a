"""world_size-2 gloo test of the sharded contrastive step.

The HIP NT-Xent kernel is GPU-only, so the CPU test injects the oracle with the same
(z, tau, grad_lo, grad_hi, pair_stride) -> (loss, dz, sim) contract; what is under test is the distributed
glue: all-gather layout, shard windows, local gradient slices, flat gradient all-reduce.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_kernel(z, tau, grad_lo, grad_hi, pair_stride):
    """Oracle with the HIP kernel's contract: rows are [rank][view][bag] blocks of ``pair_stride``, bag ids global."""
    from oracle import mil_oracle as O
    n = z.shape[0]
    Bh, ps = n // 2, pair_stride
    rows = torch.arange(n)
    bag = (rows // (2 * ps)) * ps + rows % ps
    to_view_major = torch.argsort(((rows // ps) % 2) * Bh + bag)          # view-major position -> gathered row
    with torch.enable_grad():                    # Function.forward runs with grad mode off
        zz = z.detach()[to_view_major].clone().requires_grad_()
        loss = O.nt_xent(zz[:Bh], zz[Bh:], tau)
        loss.backward()
    dz = torch.zeros_like(z)
    dz[to_view_major] = zz.grad
    dz[(bag < grad_lo) | (bag >= grad_hi)] = 0
    return loss.detach().reshape(1), dz, O.row_cosine(zz[:Bh].detach(), zz[Bh:].detach())


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from murcl_amd import dist as mdist
    from oracle import detrand
    bl, P_ = 3, 128
    torch.manual_seed(0)
    W = torch.from_numpy(detrand.normal(1, "W", (P_, 16)))              # shared "model": z = x W^T
    W.requires_grad_()
    x_i = torch.from_numpy(detrand.normal(2, f"xi{rank}", (bl, 16)))
    x_j = torch.from_numpy(detrand.normal(2, f"xj{rank}", (bl, 16)))
    loss, sim = mdist.gathered_nt_xent(x_i @ W.t(), x_j @ W.t(), 0.5, kernel=_oracle_kernel)
    loss.backward()
    flat = W.grad.reshape(-1).clone()
    # the same step with both views coming out of ONE projection (Full_layer.forward_views): the halves are recognised,
    # nothing is concatenated and one gradient tensor flows back - identical numbers
    W.grad = None
    z = torch.cat([x_i, x_j]) @ W.t()
    zi, zj = z.split(bl, 0)
    assert mdist._whole(zi, zj) is z and mdist._whole(zj, zi) is None
    loss2, sim2 = mdist.gathered_nt_xent(zi, zj, 0.5, kernel=_oracle_kernel)
    loss2.backward()
    assert abs(loss2.item() - loss.item()) < 1e-6 and torch.allclose(W.grad.reshape(-1), flat, rtol=1e-5, atol=1e-8)   # (the oracle kernel is f32)
    assert torch.allclose(sim2, sim, rtol=1e-6)
    mdist.all_reduce_grads([flat])
    out[rank] = (loss.item(), flat.detach().numpy().copy(), sim.detach().numpy().copy())
    dist.destroy_process_group()


def test_sharded_ntxent_equals_global_single_process():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    # single-process global reference: same bags, rank-major order
    from oracle import detrand, mil_oracle as O
    W = torch.from_numpy(detrand.normal(1, "W", (128, 16))).requires_grad_()
    xi = torch.cat([torch.from_numpy(detrand.normal(2, f"xi{r}", (3, 16))) for r in range(world)])
    xj = torch.cat([torch.from_numpy(detrand.normal(2, f"xj{r}", (3, 16))) for r in range(world)])
    loss = O.nt_xent(xi @ W.t(), xj @ W.t(), 0.5)
    loss.backward()
    sim = O.row_cosine((xi @ W.t()).detach(), (xj @ W.t()).detach()).numpy()
    for r in range(world):
        l, g, s = res[r]
        assert l == pytest.approx(loss.item(), rel=1e-6)
        np.testing.assert_allclose(g, W.grad.reshape(-1).numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(s, sim[3 * r:3 * r + 3], rtol=1e-5)


def _worker_overlap(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from murcl_amd import dist as mdist

    class FakeOpt:                      # FlatAdam's flat-gradient interface without the HIP optimizer step
        def __init__(self, groups):
            self.flat = [torch.zeros(sum(p.numel() for p in g)) for g in groups]
            for g, f in zip(groups, self.flat):
                off = 0
                for p in g:
                    p.grad = f[off:off + p.numel()].view_as(p)
                    off += p.numel()

        def flat_grads(self):
            return self.flat

    torch.manual_seed(rank)
    enc = torch.nn.Linear(8, 8)
    head = torch.nn.Linear(8, 4)
    opt = FakeOpt([list(enc.parameters()), list(head.parameters())])
    opt.groups = [{"params": list(enc.parameters())}, {"params": list(head.parameters())}]
    red = mdist.OverlappedGradReduce(opt, early_groups=(1,), milestones=True)
    x = torch.randn(5, 8)
    h = enc(x)
    outs = list(h.split([2, 3], 0))
    red.arm(outs)
    loss = sum(head(o).pow(2).sum() for o in outs)
    loss.backward()
    # what the aggregator's backward announces while it runs: here the bias is "final" first; finish() covers the weight
    red.milestone((enc.bias, head.weight))
    assert red._covered == {1: [(0, 36)], 0: [(64, 72)]}
    red.finish()
    from murcl_amd import functional
    functional.set_grad_milestone(None)
    out[rank] = [f.clone().numpy() for f in opt.flat]
    dist.destroy_process_group()


def test_overlapped_grad_reduce_sums_every_group_once():
    world = 2
    port = 31500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_overlap, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    # reference: sum over ranks of the single-process gradients
    want = None
    for rank in range(world):
        torch.manual_seed(rank)
        enc, head = torch.nn.Linear(8, 8), torch.nn.Linear(8, 4)
        x = torch.randn(5, 8)
        h = enc(x)
        sum(head(o).pow(2).sum() for o in h.split([2, 3], 0)).backward()
        g = [torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).numpy(),
             torch.cat([p.grad.reshape(-1) for p in head.parameters()]).numpy()]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    for rank in range(world):
        for got, w in zip(res[rank], want):
            np.testing.assert_allclose(got, w, rtol=1e-5, atol=1e-6)
