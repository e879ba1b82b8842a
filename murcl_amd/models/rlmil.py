"""Drop-in ``Memory``, ``ActorCritic``, ``PPO`` and recurrent head ``Full_layer`` (reference: models/rlmil.py).

``ActorCritic`` (state MLP 512->2048->H, GRU(H,H), sigmoid actor, critic; keys ``state_encoder.{0,2}``, ``gru.*_l0``,
``actor.0``, ``critic.0``) and ``PPO`` (``select_action`` / ``update`` with Adam, clipped surrogate, K_epochs) keep the
reference API; ``policy_conv=True`` is never enabled by the scripts and is not built.

``Full_layer`` (models/rlmil.py:187-239):

``fc_rnn=True`` (the only mode the training scripts use): one GRU time step per call with the
hidden state carried in ``self.hidden`` across calls - including the reference's behaviour
that the two views share that single attribute - followed by the class/projection Linear.
State-dict keys: ``rnn.{weight,bias}_{ih,hh}_l0``, ``fc.{weight,bias}``.
"""
import torch
from torch import nn

import math

from ..functional import GRUSeqFn, GRUStepFn, LinearFn, PolicyHeadFn, PPOLossFn
from ..utils.views import whole as _whole


class Memory:
    """Rollout storage of the PPO sampler (rlmil.py:7-22)."""

    FIELDS = ("actions", "states", "logprobs", "rewards", "is_terminals", "hidden")

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, [])

    def clear_memory(self):
        for f in self.FIELDS:
            del getattr(self, f)[:]


class Full_layer(nn.Module):
    def __init__(self, feature_num, hidden_state_dim=1024, fc_rnn=True, class_num=1000):
        super().__init__()
        self.class_num, self.feature_num = class_num, feature_num
        self.hidden_state_dim = hidden_state_dim
        self.hidden = None
        self.fc_rnn = fc_rnn
        if fc_rnn:
            self.rnn = nn.GRU(feature_num, hidden_state_dim)      # parameter holder; math in GRUStepFn
            self.fc = nn.Linear(hidden_state_dim, class_num)
        else:
            for k in (2, 3, 4, 5):                                # cascaded variant (rlmil.py:203-206)
                setattr(self, f"fc_{k}", nn.Linear(feature_num * k, class_num))

    def _step_no_grad(self, x, h_prev):
        """One GRU step + classifier with the kernels called directly (forward-only callers: frozen-encoder stage 2,
        validation): no autograd nodes to build."""
        from .. import ops
        r = self.rnn
        x = x.contiguous()
        h_prev = None if h_prev is None else h_prev.contiguous()
        if x.dtype == torch.float32 and ops.gru_step_ok(x.shape[0], self.hidden_state_dim, x.shape[1]):
            # both products of the cell (from the zero state: the input product) and the gate math in one launch
            h = ops.gru_step_fwd(r.bias_ih_l0.detach(), h_prev, r.weight_hh_l0.detach(), r.bias_hh_l0.detach(), x=x,
                                 w_ih=r.weight_ih_l0.detach(), want_backward=False)[0]
        else:
            gi = ops.gemm_nt(x, r.weight_ih_l0, epi=ops.EPI_BIAS, bias=r.bias_ih_l0)
            gh = r.bias_hh_l0.detach().view(1, -1) if h_prev is None else \
                ops.gemm_nt(h_prev, r.weight_hh_l0, epi=ops.EPI_BIAS, bias=r.bias_hh_l0)
            h = ops.gru_gates_fwd(gi, gh, h_prev)[0]
        return h, ops.gemm_nt(h, self.fc.weight, epi=ops.EPI_BIAS, bias=self.fc.bias)

    def forward(self, x, restart=False):
        if self.fc_rnn:
            h_prev = None if restart else self.hidden[0]
            if not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
                h, z = self._step_no_grad(x, h_prev)
                self.hidden = h.unsqueeze(0)
                return z
            r = self.rnn
            h = GRUStepFn.apply(x, h_prev, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
            self.hidden = h.unsqueeze(0)                          # [1,B,H] like nn.GRU's h_n
            return LinearFn.apply(h, self.fc.weight, self.fc.bias, False)
        self.hidden = x if restart else torch.cat([self.hidden, x], 1)
        k = self.hidden.size(1) // self.feature_num
        if k == 1:
            return None
        if k not in (2, 3, 4, 5) or self.hidden.size(1) != k * self.feature_num:
            raise RuntimeError(f"Full_layer cascade: unexpected width {tuple(self.hidden.size())}")
        head = getattr(self, f"fc_{k}")
        return LinearFn.apply(self.hidden, head.weight, head.bias, False)


    def forward_sequence(self, x):
        """x [T,B,F]: ``[self(x[t], restart=(t == 0)) for t in range(T)]`` as ONE recurrent node and one classifier product over
        the T*B rows -> logits [T*B, class_num]; ``self.hidden`` ends as after the loop.  (The supervised step computes all T
        sub-bags of a stage-1 step before the head runs, so the head needs no per-step launches from Python.)"""
        if not self.fc_rnn:
            return torch.cat([self(x[t], restart=(t == 0)) for t in range(x.shape[0])], 0)
        from ..functional import GRUSeqFn
        r = self.rnn
        hs = GRUSeqFn.apply(x.float(), r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
        self.hidden = hs[-1].unsqueeze(0)
        return LinearFn.apply(hs.reshape(x.shape[0] * x.shape[1], -1), self.fc.weight, self.fc.bias, False)

    def forward_view_sequence(self, xs, whole=None):
        """The head over ALL patch steps of a contrastive step: ``xs`` = the aggregator outputs [x_t0, x_t1 for t in range(T)]
        (2T tensors [B,F]) -> z [2T*B, class_num] in the same order, as if
        ``forward_views([x_t0, x_t1], restart=(t == 0))`` had been called step by step (train_MuRCL.py:243,272).  With the one
        shared hidden state of the reference that loop is ONE chain: both views of step 0 start from zero, then
        h(0,1) -> h(1,0) -> h(1,1) -> h(2,0) -> ...; so x_01, x_10, x_11, ... run as one recurrent node from a zero state and
        x_00 as a single step beside it, and the classifier is one product over the 2T*B rows."""
        if not self.fc_rnn:
            raise RuntimeError("forward_view_sequence needs the recurrent head (fc_rnn=True)")
        from ..functional import GRUSeqFn
        xs = list(xs)
        n2, B = len(xs), xs[0].shape[0]
        x = whole if whole is not None else (_whole(xs) if n2 > 1 else xs[0])        # ``whole``: torch.cat(xs, 0), already in one tensor
        assert whole is None or (whole.dim() == 2 and whole.shape[0] == n2 * B)
        r = self.rnn
        if (not torch.is_grad_enabled() and xs[0].is_cuda and xs[0].dtype == torch.float32 and xs[0].dim() == 2 and n2 > 1
                and len({tuple(t.shape) for t in xs}) == 1):
            from .. import ops
            H = r.weight_hh_l0.shape[1]
            if ops.gru_step_ok(B, H) and n2 >= 3:
                # nobody differentiates this pass (frozen-aggregator stage 2, validation): the launches of GRUViewSeqFn.forward without
                # the tensors its backward pass would need - one stacking launch, one input product over all rows, every hidden state
                # written straight into the rows of ONE buffer
                if x is None:
                    x = ops.stack_lists([xs])[0].view(n2 * B, -1)
                wih, whh, bih, bhh = (t.detach() for t in (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0))
                h_all = torch.empty((n2 * B, H), dtype=torch.float32, device=x.device)
                gi = ops.gemm_nt(x, wih, epi=ops.EPI_BIAS, bias=bih)
                ops.gru_gates_fwd(gi[:2 * B], bhh.view(1, -1), None, hnew=h_all[:2 * B])          # blocks 0 and 1: from the zero state
                for k in range(2, n2):
                    ops.gru_step_fwd(gi[k * B:(k + 1) * B], h_all[(k - 1) * B:k * B], whh, bhh, hnew=h_all[k * B:(k + 1) * B],
                                     want_backward=False)
                self.hidden = h_all[-B:].unsqueeze(0)
                return ops.gemm_nt(h_all, self.fc.weight.detach(), epi=ops.EPI_BIAS, bias=self.fc.bias.detach())
        if x is None:
            x = torch.cat(xs, 0)
        x = x.float()
        if n2 >= 3 and x.is_cuda and x.dim() == 2:
            from .. import ops
            from ..functional import GRUViewSeqFn
            if ops.gru_step_ok(B, r.weight_hh_l0.shape[1]):
                h_all = GRUViewSeqFn.apply(x, B, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)      # one node, one buffer
                self.hidden = h_all[-B:].unsqueeze(0)
                return LinearFn.apply(h_all, self.fc.weight, self.fc.bias, False)
        h00 = GRUStepFn.apply(x[:B], None, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
        hs = GRUSeqFn.apply(x[B:].view(n2 - 1, B, -1), r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
        self.hidden = hs[-1].unsqueeze(0)
        h_all = torch.cat([h00, hs.reshape((n2 - 1) * B, -1)], 0)
        return LinearFn.apply(h_all, self.fc.weight, self.fc.bias, False)

    def forward_views(self, xs, restart=False):
        """``[self(x, restart) for x in xs]`` - the per-view loop of the training scripts (train_MuRCL.py:243,272).

        With ``restart=True`` every view starts from a zero hidden state, so the views are independent rows of one
        batch: they run through ONE GRU step / classifier GEMM (row-wise identical math, half the launches) and
        ``self.hidden`` ends as the last view's state, exactly as after the sequential loop."""
        xs = list(xs)
        if not (self.fc_rnn and restart and len(xs) > 1 and len({tuple(x.shape) for x in xs}) == 1):
            return [self(x, restart) for x in xs]
        n = xs[0].shape[0]
        r = self.rnn
        if not torch.is_grad_enabled() and xs[0].is_cuda and xs[0].dtype == torch.float32 and xs[0].dim() == 2:
            x = _whole(xs)
            h, z = self._step_no_grad(torch.cat(xs, 0) if x is None else x, None)
            self.hidden = h[-n:].unsqueeze(0)
            return list(z.split(n, 0))
        x = _whole(xs)
        h = GRUStepFn.apply(torch.cat(xs, 0) if x is None else x, None, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
        self.hidden = h[-n:].unsqueeze(0)
        return list(LinearFn.apply(h, self.fc.weight, self.fc.bias, False).split(n, 0))


class ActorCritic(nn.Module):
    def __init__(self, feature_dim, state_dim, hidden_state_dim=1024, policy_conv=False, action_std=0.1, action_size=2):
        super().__init__()
        if policy_conv:
            raise NotImplementedError("policy_conv=True is never enabled by the training scripts (train_MuRCL.py:445)")
        self.state_encoder = nn.Sequential(nn.Linear(state_dim, 2048), nn.ReLU(), nn.Linear(2048, hidden_state_dim), nn.ReLU())
        self.gru = nn.GRU(hidden_state_dim, hidden_state_dim, batch_first=False)        # parameter holder
        self.actor = nn.Sequential(nn.Linear(hidden_state_dim, action_size), nn.Sigmoid())
        self.critic = nn.Sequential(nn.Linear(hidden_state_dim, 1))
        self.action_std = float(action_std)       # used as the Cholesky factor: a std, not a variance (rlmil.py:84-85)
        self.action_size = action_size
        self.hidden_state_dim, self.policy_conv, self.feature_dim = hidden_state_dim, policy_conv, feature_dim
        self.feature_ratio = int(math.sqrt(state_dim / feature_dim))

    def forward(self):
        raise NotImplementedError

    def _trunk(self, state, hidden):
        e = LinearFn.apply(state, self.state_encoder[0].weight, self.state_encoder[0].bias, True)
        e = LinearFn.apply(e, self.state_encoder[2].weight, self.state_encoder[2].bias, True)
        g = self.gru
        return GRUStepFn.apply(e, hidden, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)

    def _trunk_no_grad(self, state, hidden):
        """``_trunk`` for the sampler's forward-only steps: the same kernels called directly - ``act`` runs under
        ``torch.no_grad()``, so three autograd-node constructions per call (ten calls per training step) buy nothing."""
        from .. import ops
        enc, g = self.state_encoder, self.gru
        e = ops.gemm_nt(state, enc[0].weight, epi=ops.EPI_BIAS_RELU, bias=enc[0].bias)
        e = ops.gemm_nt(e, enc[2].weight, epi=ops.EPI_BIAS_RELU, bias=enc[2].bias)
        gi = ops.gemm_nt(e, g.weight_ih_l0, epi=ops.EPI_BIAS, bias=g.bias_ih_l0)
        gh = ops.gemm_nt(hidden.contiguous(), g.weight_hh_l0, epi=ops.EPI_BIAS, bias=g.bias_hh_l0)
        return ops.gru_gates_fwd(gi, gh, hidden.contiguous())[0]

    def _native_ok(self, S):
        """Shapes the native launch sequences (csrc/ppo_seq.hip) cover: the reference's own (S = 512, H = 512, K = 10)."""
        return (self.state_encoder[0].out_features == 2048 and S % 32 == 0 and self.hidden_state_dim % 32 == 0
                and self.action_size <= 16 and self.state_encoder[0].weight.is_cuda)

    def _plist(self):
        """The 12 Parameter objects in ``parameters()`` order, collected once: the objects keep their identity for the life of
        the module (optimizers re-seat ``.data`` / ``.grad``, load_state_dict copies in place), and walking the module tree for
        them cost 0.9 ms per sampler-in-the-loop step (368 ``parameters()`` calls)."""
        pl = self.__dict__.get("_plist_cache")
        if pl is None:
            pl = list(self.parameters())
            object.__setattr__(self, "_plist_cache", pl)
        return pl

    def pointer_table(self, grads=False):
        """ctypes table of this module's 12 parameter (or gradient) pointers, rebuilt when a tensor was re-seated."""
        from .. import ops
        ts = [p.grad if grads else p for p in self._plist()]
        key = (ts[0].data_ptr(), ts[-1].data_ptr())
        slot = "_gtab" if grads else "_ptab"
        cur = getattr(self, slot, None)
        if cur is None or cur[0] != key:
            cur = (key, ops.pointer_table(ts))
            object.__setattr__(self, slot, cur)
        return cur[1]

    def act(self, state_ini, memory, restart_batch=False, training=False, eps=None):
        """One policy step (rlmil.py:66-97).  ``eps`` ~ N(0,1) [B,K] may be injected (parity tests)."""
        from .. import ops
        with torch.no_grad():
            if restart_batch:
                del memory.hidden[:]
                memory.hidden.append(torch.zeros(1, state_ini.size(0), self.hidden_state_dim, device=state_ini.device))
            state = state_ini.flatten(1)
            if training and state.is_cuda and self._native_ok(state.shape[1]):
                # the whole step - encoder, GRU cell, actor head, sampling, log-prob - as one native launch sequence
                if eps is None:
                    eps = torch.randn((state.shape[0], self.action_size), device=state.device)
                h, action, logp = ops.ppo_act(self.pointer_table(), state.shape[1], self.hidden_state_dim, self.action_size,
                                              state.float(), None if restart_batch else memory.hidden[-1][0], eps, self.action_std)
                memory.hidden.append(h.unsqueeze(0))
                memory.states.append(state_ini)
                memory.actions.append(action)
                memory.logprobs.append(logp)
                return action
            h = self._trunk_no_grad(state.float().contiguous(), memory.hidden[-1][0])
            memory.hidden.append(h.unsqueeze(0))
            z = ops.gemm_nt(h, self.actor[0].weight, epi=ops.EPI_BIAS, bias=self.actor[0].bias)
            if eps is None:
                eps = torch.randn((z.shape[0], self.action_size), device=z.device)
            mu, action, logp = ops.policy_head_fwd(z, self.action_std, eps=eps)
            if training:
                memory.states.append(state_ini)
                memory.actions.append(action)
                memory.logprobs.append(logp)
            else:
                action = mu
        return action.detach()

    def act_views(self, states, memories, restart_batch=False, eps=None):
        """``act(training=True)`` for several independent rollouts (the two views of a MuRCL step, train_MuRCL.py:259-265) as ONE
        policy step over their stacked rows: every row of the step - encoder, GRU cell, head, sample, log-prob - depends on
        its own state / hidden row only, so the views' calls are the same arithmetic with half the launches.  Each memory
        receives its own rows (views of the step's outputs), exactly what per-view calls append."""
        from .. import ops
        flat = [s.flatten(1) for s in states]
        if not (len(states) > 1 and all(s.is_cuda for s in flat) and self._native_ok(flat[0].shape[1])
                and len({tuple(s.shape) for s in flat}) == 1):
            return [self.act(s, m, restart_batch, True, e)
                    for s, m, e in zip(states, memories, eps if eps is not None else [None] * len(states))]
        with torch.no_grad():
            n = flat[0].shape[0]
            if restart_batch:
                h0 = _zero_hidden(n, self.hidden_state_dim, flat[0].device)      # read-only: the first step starts from "no hidden state"
                for m in memories:
                    del m.hidden[:]
                    m.hidden.append(h0)
            state = ops.stack_rows([s.float() for s in flat])
            noise = (torch.randn((n * len(flat), self.action_size), device=state.device) if eps is None
                     else ops.stack_rows([e.float() for e in eps]))
            hidden = None if restart_batch else ops.stack_rows([m.hidden[-1][0] for m in memories])
            h, action, logp = ops.ppo_act(self.pointer_table(), state.shape[1], self.hidden_state_dim, self.action_size, state,
                                          hidden, noise, self.action_std)
            out = []
            for i, (s, m) in enumerate(zip(states, memories)):
                rows = slice(i * n, (i + 1) * n)
                m.hidden.append(h[rows].unsqueeze(0))
                m.states.append(s)
                m.actions.append(action[rows])
                m.logprobs.append(logp[rows])
                out.append(action[rows])
        return out

    def evaluate(self, state, action):
        """states [T,B,S], actions [T,B,K] -> (logp, value, entropy) each [T,B] (rlmil.py:99-127)."""
        T_, B = state.shape[0], state.shape[1]
        enc = self.state_encoder                                # encoder over all T*B rows at once (rlmil.py:103-109)
        e = LinearFn.apply(state.reshape(T_ * B, -1).float().contiguous(), enc[0].weight, enc[0].bias, True)
        e = LinearFn.apply(e, enc[2].weight, enc[2].bias, True)
        g = self.gru                                            # GRU over the rollout from a zero hidden state
        hs = GRUSeqFn.apply(e.view(T_, B, -1), g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        hs = hs.reshape(T_ * B, -1)
        z = LinearFn.apply(hs, self.actor[0].weight, self.actor[0].bias, False)
        logp = PolicyHeadFn.apply(z, action.reshape(T_ * B, -1).float().contiguous(), self.action_std)
        value = LinearFn.apply(hs, self.critic[0].weight, self.critic[0].bias, False)
        k = self.action_size
        ent = 0.5 * k * (1.0 + math.log(2 * math.pi)) + k * math.log(self.action_std)
        return logp.view(T_, B), value.view(T_, B), torch.full((T_, B), ent, device=state.device)


_ZERO_HIDDEN = {}


def _zero_hidden(n, H, device):
    """A shared read-only [1,n,H] zero state (rlmil.py:69-71 allocates one per rollout): no fill launch per restart."""
    key = (n, H, str(device))
    z = _ZERO_HIDDEN.get(key)
    if z is None:
        if len(_ZERO_HIDDEN) > 8:
            _ZERO_HIDDEN.clear()
        z = _ZERO_HIDDEN[key] = torch.zeros(1, n, H, device=device)
    return z


class PPO:
    def __init__(self, feature_dim, state_dim, hidden_state_dim, policy_conv, action_std=0.1, lr=0.0003,
                 betas=(0.9, 0.999), gamma=0.7, K_epochs=1, eps_clip=0.2, action_size=2):
        from ..optim import FlatAdam
        self.lr, self.betas, self.gamma, self.eps_clip, self.K_epochs = lr, betas, gamma, eps_clip, K_epochs
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.policy = ActorCritic(feature_dim, state_dim, hidden_state_dim, policy_conv, action_std, action_size).to(dev)
        self.policy_old = ActorCritic(feature_dim, state_dim, hidden_state_dim, policy_conv, action_std, action_size).to(dev)
        self.policy_old.load_state_dict(self.policy.state_dict())
        self.optimizer = FlatAdam([{"params": list(self.policy.parameters()), "lr": lr}], betas=betas) if dev.type == "cuda" else None
        self._old_flat = None
        self._k = _HipPolicyKernels            # device side of update()
        self.data_parallel = None              # None: collectives when a process group with > 1 ranks exists; True: whenever
        if self.optimizer is not None:         # one is initialised (single-rank runs of the multi-GPU code path); False: never
            # policy_old's parameters as views of one buffer with the layout of the optimizer's flat parameter buffer: the
            # sync after every update (rlmil.py:183) is one copy instead of a load_state_dict over twelve tensors
            src = self.optimizer.groups[0]
            flat, off = torch.empty_like(src["p"]), 0
            for p_new, p_old in zip(src["params"], self.policy_old.parameters()):
                assert p_new.shape == p_old.shape
                k = p_old.numel()
                flat[off:off + k].copy_(p_old.data.reshape(-1))
                p_old.data = flat[off:off + k].view_as(p_old.data)
                off += k
            self._old_flat = flat

    def select_action(self, state, memory, restart_batch=False, training=True, eps=None):
        return self.policy_old.act(state, memory, restart_batch, training, eps)

    def select_actions(self, states, memories, restart_batch=False, eps=None):
        """``select_action`` for the views of one patch step at once (``ActorCritic.act_views``)."""
        return self.policy_old.act_views(states, memories, restart_batch, eps)

    def update(self, memory, group=None):
        """PPO.update (rlmil.py:152-184).  Data-parallel (one process per GPU, rollout rows sharded by bag, SURVEY.md
        8(e)): the reference normalises the returns over the WHOLE batch and takes the loss mean over all its rows, so
        under an initialised process group the (sum, sum of squares) of the returns are all-reduced (one 16-byte
        collective) and, per K_epoch, the flat policy gradient (14.7 MB) - every rank then applies the identical Adam
        step and the N ranks keep ONE sampler, bit-identical across ranks."""
        import torch.distributed as dist
        from .. import dist as mdist
        k = self._k
        world = 1
        if dist.is_available() and dist.is_initialized() and self.data_parallel is not False:
            world = dist.get_world_size(group)
        collectives = world > 1 or (self.data_parallel is True and dist.is_available() and dist.is_initialized())
        from .. import ops
        rewards = ops.stack_rows([r.reshape(1, -1) for r in memory.rewards])           # [T,B] (rlmil.py:156-160); a view when the step
        #                                                                                computed its T rewards as one tensor
        n_total = rewards.numel() * world                                              # equal shards: B_local bags per rank
        if collectives:
            returns, stats = k.returns_raw(rewards, self.gamma)
            mdist.all_reduce_sum(stats, group)
            returns = k.returns_finish(returns, stats, n_total)                        # rlmil.py:162 over all ranks' returns
        else:
            returns = k.returns(rewards, self.gamma)
        if memory.states[0].is_cuda:           # rlmil.py:163-165: the three stacks as one launch
            old_states, old_actions, old_logprobs = ops.stack_lists([[t.detach() for t in memory.states], [t.detach() for t in memory.actions],
                                                                     [t.detach() for t in memory.logprobs]])
        else:
            old_states = torch.stack(memory.states, 0).detach()
            old_actions = torch.stack(memory.actions, 0).detach()
            old_logprobs = torch.stack(memory.logprobs, 0).detach()
        for _ in range(self.K_epochs):
            k.epoch_grads(self, old_states, old_actions, old_logprobs, returns, n_total)   # rlmil.py:169-180
            if collectives:
                for g in k.flat_grads(self):                                           # gradients carry 1/n_total: SUM = global mean
                    mdist.all_reduce_sum(g, group)
            k.step(self)                                                               # rlmil.py:181
        k.sync_old(self)                                                               # rlmil.py:183


class _HipPolicyKernels:
    """The device side of ``PPO.update``: every piece is a sequence of C-ABI launches (tests on a CPU-only box inject an
    object with the same five methods built on the oracle, to exercise the collective glue under gloo)."""

    @staticmethod
    def returns(rewards, gamma):
        from .. import ops
        return ops.ppo_returns(rewards, gamma)

    @staticmethod
    def returns_raw(rewards, gamma):
        from .. import ops
        return ops.ppo_returns_raw(rewards, gamma)

    @staticmethod
    def returns_finish(ret, stats, n_total):
        from .. import ops
        return ops.ppo_returns_finish(ret, stats, n_total)

    @staticmethod
    def epoch_grads(ppo, states, actions, old_logp, returns, n_total):
        from .. import ops
        pol = ppo.policy
        entropy = 0.5 * pol.action_size * (1.0 + math.log(2 * math.pi)) + pol.action_size * math.log(pol.action_std)
        S = states[0].flatten(1).shape[1]
        if pol._native_ok(S) and all(p.grad is not None and p.grad.is_contiguous() for p in pol._plist()):
            # evaluate() forward + loss + backward as ONE native launch sequence adding into the flat gradient buffer
            ppo.optimizer.zero_grad()
            g, enc = pol.gru, pol.state_encoder
            wt = None
            if all(ops.is_managed(w) for w in (g.weight_ih_l0, g.weight_hh_l0, enc[2].weight)):
                # the dgrad operands W^T: the optimizer-managed views, rebuilt by ONE launch after each Adam step
                wt = ops.weight_views(tuple((w, True, torch.float32) for w in (g.weight_ih_l0, g.weight_hh_l0, enc[2].weight)))
            ops.ppo_epoch(pol.pointer_table(), pol.pointer_table(grads=True), S, pol.hidden_state_dim, pol.action_size,
                          states.flatten(2), actions, old_logp, returns, n_total, pol.action_std, ppo.eps_clip, entropy, wt=wt)
            ppo.optimizer.mark_all_touched()
            return
        logp, value, _ = pol.evaluate(states, actions)
        loss = PPOLossFn.apply(logp.reshape(-1), old_logp.reshape(-1), value.reshape(-1), returns.reshape(-1),
                               ppo.eps_clip, entropy, n_total)
        ppo.optimizer.zero_grad()
        loss.backward()

    @staticmethod
    def flat_grads(ppo):
        return ppo.optimizer.flat_grads()

    @staticmethod
    def step(ppo):
        ppo.optimizer.step()

    @staticmethod
    def sync_old(ppo):
        if ppo._old_flat is not None and ppo._old_flat.data_ptr() == ppo.policy_old._plist()[0].data_ptr():
            from .. import ops
            ops.copy_flat(ppo._old_flat, ppo.optimizer.groups[0]["p"])
        else:                                                      # someone re-seated policy_old's tensors (e.g. .to()): generic path
            ppo.policy_old.load_state_dict(ppo.policy.state_dict())
