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


# ------------------------------------------------------------------ PPO.update on N ranks == one sampler
class _OraclePolicyKernels:
    """CPU stand-in for models/rlmil._HipPolicyKernels (same five methods, oracle arithmetic): what is under test is
    PPO.update's data-parallel glue - the 16-byte return-statistics all-reduce, the 1/n_total scaling, the per-epoch
    gradient all-reduce, identical optimizer steps on every rank."""

    def __init__(self, lr):
        self.lr, self.adam, self.grads = lr, {}, None

    @staticmethod
    def _raw(rewards, gamma):
        out, run = [], torch.zeros_like(rewards[0])
        for t in range(rewards.shape[0] - 1, -1, -1):
            run = rewards[t] + gamma * run
            out.insert(0, run)
        return torch.stack(out, 0)

    def returns(self, rewards, gamma):
        R = self._raw(rewards, gamma)
        return (R - R.mean()) / (R.std() + 1e-5)

    def returns_raw(self, rewards, gamma):
        R = self._raw(rewards, gamma)
        return R, torch.stack([R.double().sum(), (R.double() ** 2).sum()])

    @staticmethod
    def returns_finish(ret, stats, n_total):
        mean = stats[0] / n_total
        var = (stats[1] - stats[0] * mean) / (n_total - 1)
        return ((ret.double() - mean) / (var.sqrt() + 1e-5)).float()

    def epoch_grads(self, ppo, states, actions, old_logp, returns, n_total):
        from oracle import mil_oracle as O
        p = {k: v.detach().clone().requires_grad_() for k, v in ppo.policy.state_dict().items()}
        logp, value, ent = O.ppo_evaluate(p, states, actions, ppo.policy.action_std)
        ratio = torch.exp(logp - old_logp)
        adv = returns - value.detach()
        per_row = -torch.min(ratio * adv, ratio.clamp(1 - ppo.eps_clip, 1 + ppo.eps_clip) * adv) \
            + 0.5 * (value - returns) ** 2 - 0.01 * ent
        (per_row.sum() / n_total).backward()                         # this rank's share of the global mean
        self.names = list(p)
        self.grads = [torch.cat([p[k].grad.reshape(-1) for k in self.names])]

    def flat_grads(self, ppo):
        return self.grads

    def step(self, ppo):
        from oracle import mil_oracle as O
        sd = ppo.policy.state_dict()
        grads, off = {}, 0
        for k in self.names:
            n = sd[k].numel()
            grads[k] = self.grads[0][off:off + n].view_as(sd[k])
            off += n
        new = O.adam_step({k: sd[k].detach().clone() for k in self.names}, grads, self.adam, self.lr)
        ppo.policy.load_state_dict(new)

    @staticmethod
    def sync_old(ppo):
        ppo.policy_old.load_state_dict(ppo.policy.state_dict())


def _ppo_rollout(seed, bags, Tm, S_, K):
    from oracle import detrand
    return ([torch.from_numpy(detrand.normal(seed, f"s{t}", (bags, S_))) for t in range(Tm)],
            [torch.from_numpy(detrand.normal(seed, f"e{t}", (bags, K))) for t in range(Tm)],
            [torch.from_numpy(detrand.normal(seed, f"r{t}", (1, bags)) * 0.01) for t in range(Tm)])


def _ppo_run(lo, hi, group_on):
    """PPO.update over bags [lo, hi) of a fixed 8-bag rollout, through the product's PPO class (CPU parameters)."""
    from murcl_amd.models.rlmil import PPO, Memory
    from oracle import mil_oracle as O, params as P
    S_, H, K, Tm = 32, 16, 4, 3
    ppo = PPO(S_, S_, H, False, action_std=0.5, lr=1e-3, gamma=0.1, K_epochs=2, action_size=K)
    sd = {k: v for k, v in P.to_torch(P.actor_critic(5, S_, H, K)).items()}
    ppo.policy.load_state_dict(sd)
    ppo.policy_old.load_state_dict(sd)
    ppo._k = _OraclePolicyKernels(1e-3)
    ppo.data_parallel = group_on
    states, eps, rewards = _ppo_rollout(9, 8, Tm, S_, K)
    mem = Memory()
    hid = torch.zeros(hi - lo, H)
    for t in range(Tm):                                              # the rollout itself is per-bag: rows lo..hi of the global one
        a, lp, hid = O.ppo_act(sd, states[t][lo:hi], hid, eps[t][lo:hi], 0.5)
        mem.states.append(states[t][lo:hi]), mem.actions.append(a), mem.logprobs.append(lp)
        mem.rewards.append(rewards[t][:, lo:hi])
    ppo.update(mem)
    assert all(torch.equal(a, b) for a, b in zip(ppo.policy.parameters(), ppo.policy_old.parameters()))
    return {k: v.detach().clone() for k, v in ppo.policy.state_dict().items()}


def _worker_ppo(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = 8 // world
    sd = _ppo_run(rank * per, (rank + 1) * per, None)
    out[rank] = {k: v.numpy() for k, v in sd.items()}
    dist.destroy_process_group()


def test_ppo_update_on_two_ranks_trains_one_sampler():
    """2 ranks x 4 bags end with the SAME policy (bit-identical across ranks), equal to 1 rank x 8 bags: returns are
    normalised with the global mean / std (rlmil.py:162) and the policy gradient is the global mean's (rlmil.py:180)."""
    world = 2
    port = 33500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_ppo, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    sys.path.insert(0, ROOT)
    single = _ppo_run(0, 8, False)
    from oracle import params as P
    pre = P.to_torch(P.actor_critic(5, 32, 16, 4))
    for k in single:
        assert np.array_equal(res[0][k], res[1][k]), f"ranks diverged on {k}"
        moved = (single[k] - pre[k]).abs().max().item()
        assert moved > 0
        np.testing.assert_allclose(res[0][k], single[k].numpy(), rtol=0, atol=2e-2 * moved, err_msg=k)
