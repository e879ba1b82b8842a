"""CPU oracle: fp32 restatement of the MuRCL MIL aggregators, NT-Xent and recurrent heads.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Arithmetic backend: plain PyTorch
CPU fp32 tensor ops - the same ATen arithmetic the reference itself runs on - but
restated batch-wise (no per-bag Python loop) and as pure functions of explicit
parameter dicts keyed by the reference's state-dict names.  Gradients come from
autograd over these functions.

All ``file:line`` citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F


def _lin(x, p, name):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


# ----------------------------------------------------------------------------- ABMIL
def abmil_forward(p, x, drop_masks=None):
    """ABMIL.bag_forward over a batch (models/abmil.py:35-45, batch loop :47-51).

    x [B,N,d] -> out [B,L], A [B,N] (post-softmax, already divided by sqrt(N) as in
    abmil.py:40-41), s [B,N] raw scores, M [B,L] pooled vector before the decoder.
    Dropout (abmil.py:15,18) is p=0 in every launch script (train_MuRCL.py:456): ``drop_masks`` = None is the
    identity; (k1, k2) keep multipliers [B,N,L] (0 or 1/keep) inject the training-mode masks after layers 1 and 2.
    ``self.fc`` (abmil.py:33) is never applied by the reference.  Any L / D (abmil.py:8).
    """
    h = torch.relu(_lin(x, p, "encoder.0"))          # abmil.py:13-14
    if drop_masks is not None:
        h = h * drop_masks[0]                        # abmil.py:15
    h = torch.relu(_lin(h, p, "encoder.3"))          # abmil.py:16-17
    if drop_masks is not None:
        h = h * drop_masks[1]                        # abmil.py:18
    h = torch.relu(_lin(h, p, "encoder.6"))          # abmil.py:19-20
    t = torch.tanh(_lin(h, p, "attention.0"))        # abmil.py:24-25
    s = _lin(t, p, "attention.2").squeeze(-1)        # abmil.py:26 (K=1)
    A = torch.softmax(s, dim=1) / math.sqrt(s.shape[1])   # abmil.py:40-41
    M = torch.einsum("bn,bnl->bl", A, h)             # abmil.py:42
    out = torch.relu(_lin(M, p, "decoder.0"))        # abmil.py:29-32,44
    return out, A, s, M


def abmil_forward_heads(p, x):
    """ABMIL with K > 1 attention heads (models/abmil.py:23-27,38-44): ``attention.2`` has K rows, the soft-max runs over the
    patches of every head, ``torch.mm(A, H)`` is [K, L] per bag and the batch loop concatenates them: out [B*K, L] (bag-major,
    head-minor), A [B, K, N]."""
    h = torch.relu(_lin(x, p, "encoder.0"))
    h = torch.relu(_lin(h, p, "encoder.3"))
    h = torch.relu(_lin(h, p, "encoder.6"))
    s = _lin(torch.tanh(_lin(h, p, "attention.0")), p, "attention.2").transpose(1, 2)        # abmil.py:38-39: [B, K, N]
    A = torch.softmax(s, dim=2) / math.sqrt(s.shape[2])                                      # abmil.py:40-41
    M = torch.einsum("bkn,bnl->bkl", A, h)                                                   # abmil.py:42
    out = torch.relu(_lin(M, p, "decoder.0"))                                                # abmil.py:44
    return out.reshape(-1, out.shape[-1]), A


def abmil_attn_pool(h, wa, ba, wb, bb):
    """The K2 kernel in isolation: scores, softmax/sqrt(N), pooling (abmil.py:38-42)."""
    s = F.linear(torch.tanh(F.linear(h, wa, ba)), wb, bb).squeeze(-1)
    A = torch.softmax(s, dim=1) / math.sqrt(s.shape[1])
    return torch.einsum("bn,bnl->bl", A, h), A, s


# ----------------------------------------------------------------------------- CLAM-SB
def clam_sb_forward(p, x, drop_mask=None):
    """CLAM_SB.bag_forward without instance eval (models/clam.py:134-144,170).

    x [B,N,d] -> M [B,512], A [B,N] softmax weights, s [B,N] raw scores (what
    ``attention_only=True`` returns, clam.py:141-142), h [B,N,512].
    ``drop_mask`` (three tensors or None) injects the Dropout(0.25) masks of
    clam.py:71-72,47-48 for train-mode parity; None == eval mode.
    """
    h = torch.relu(_lin(x, p, "attention_net.0"))                  # clam.py:69
    if drop_mask is not None:
        h = h * drop_mask[0] / 0.75
    if "attention_net.3.module.0.weight" in p:                     # gate=False: Attn_Net (clam.py:18-34), dropout layout
        a = torch.tanh(_lin(h, p, "attention_net.3.module.0"))
        if drop_mask is not None:
            a = a * drop_mask[1] / 0.75
        s = _lin(a, p, "attention_net.3.module.3").squeeze(-1)
        A = torch.softmax(s, dim=1)
        return torch.einsum("bn,bnl->bl", A, h), A, s, h
    a = torch.tanh(_lin(h, p, "attention_net.3.attention_a.0"))    # clam.py:40-41,56
    g = torch.sigmoid(_lin(h, p, "attention_net.3.attention_b.0"))  # clam.py:44-45,57
    if drop_mask is not None:
        a = a * drop_mask[1] / 0.75
        g = g * drop_mask[2] / 0.75
    s = _lin(a * g, p, "attention_net.3.attention_c").squeeze(-1)  # clam.py:58-59
    A = torch.softmax(s, dim=1)                                    # clam.py:144
    M = torch.einsum("bn,bnl->bl", A, h)                           # clam.py:170
    return M, A, s, h


def clam_instance_eval(p, A, h, label, n_classes, k_sample, subtyping, loss_fn=None):
    """CLAM_SB.inst_eval / inst_eval_out for ONE bag (clam.py:103-132,146-168).

    A [N] post-softmax, h [N,512], label int.  Returns (loss, preds int64, targets
    int64, ids) where ids lists the selected patch indices in reference order (top-k
    of A, then top-k of -A for the in-class branch).  Ties in top-k are broken by
    ATen on the reference side; goldens are generated with margins.
    """
    total = torch.zeros((), dtype=h.dtype)
    preds, targets, ids_all = [], [], []
    for i in range(n_classes):
        w = p[f"instance_classifiers.{i}.weight"]
        b = p[f"instance_classifiers.{i}.bias"]
        if i == int(label):                                   # clam.py:152-155
            top_p = torch.topk(A, k_sample)[1]                # clam.py:107
            top_n = torch.topk(-A, k_sample)[1]               # clam.py:109
            ids = torch.cat([top_p, top_n])
            tgt = torch.cat([torch.ones(k_sample), torch.zeros(k_sample)]).long()
        elif subtyping:                                       # clam.py:159-162
            ids = torch.topk(A, k_sample)[1]                  # clam.py:126
            tgt = torch.zeros(k_sample).long()
        else:
            continue
        logits = F.linear(h[ids], w, b)                       # clam.py:116,129
        total = total + (F.cross_entropy(logits, tgt) if loss_fn is None else loss_fn(logits, tgt))   # clam.py:118,131 (instance_loss_fn)
        preds.append(logits.argmax(1))
        targets.append(tgt)
        ids_all.append(ids)
    if subtyping:
        total = total / n_classes                             # clam.py:167-168
    return total, torch.cat(preds), torch.cat(targets), torch.cat(ids_all)


# ----------------------------------------------------------------------------- DSMIL
def dsmil_forward(p, x, keep_v=None):
    """MILNet.forward for a batch (models/dsmil.py:9-16,64-81,104-113).

    x [B,N,d] -> classes [B,N,C], bag [B,C,d], A [B,N,C], m [B,C] critical-instance ids.
    ``keep_v`` [B,N,d] (0 or 1/keep): the value branch's nn.Dropout(dropout_v) in training mode (dsmil.py:55-58,66) with its mask
    injected; None = dropout_v 0 / eval mode.
    """
    c = _lin(x, p, "i_classifier.fc.0")                         # dsmil.py:15
    V = _lin(x if keep_v is None else x * keep_v, p, "b_classifier.v.1")      # dsmil.py:66 (Dropout, then Linear)
    Q = _lin(x, p, "b_classifier.q")                            # dsmil.py:67
    m = c.argmax(dim=1)                                         # dsmil.py:71-73 (sort desc, row 0)
    m_feats = torch.gather(x, 1, m.unsqueeze(-1).expand(-1, -1, x.shape[-1]))
    q_max = _lin(m_feats, p, "b_classifier.q")                  # dsmil.py:74
    A = torch.einsum("bnq,bcq->bnc", Q, q_max)                  # dsmil.py:76
    A = torch.softmax(A / math.sqrt(Q.shape[-1]), dim=1)        # dsmil.py:77
    bag = torch.einsum("bnc,bnd->bcd", A, V)                    # dsmil.py:78
    return c, bag, A, m


# ----------------------------------------------------------------------------- NT-Xent
def nt_xent(z_i, z_j, temperature):
    """NT_Xent.forward (utils/losses.py:24-41) in closed form.

    loss = mean_i [ logsumexp_{j != i} S_ij - S_{i,pos(i)} ], S = cos(z_i,z_j)/tau,
    pos(i) = i +/- B.  Cosine follows ATen: each norm clamped at eps=1e-8.
    """
    z = torch.cat([z_i, z_j], 0)
    n2 = z.shape[0]
    b = n2 // 2
    zn = z / z.norm(dim=1, keepdim=True).clamp_min(1e-8)
    S = zn @ zn.t() / temperature
    pos = torch.cat([torch.diagonal(S, b), torch.diagonal(S, -b)])
    Sm = S.masked_fill(torch.eye(n2, dtype=torch.bool), float("-inf"))
    return (torch.logsumexp(Sm, dim=1) - pos).sum() / n2


def row_cosine(a, b):
    """torch.cosine_similarity(outputs[0], outputs[1]) (train_MuRCL.py:253,282)."""
    an = a.norm(dim=1).clamp_min(1e-8)
    bn = b.norm(dim=1).clamp_min(1e-8)
    return (a * b).sum(1) / (an * bn)


# ----------------------------------------------------------------------------- GRU heads
def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One nn.GRU time-step, PyTorch gate order (r, z, n)."""
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    H = h.shape[-1]
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1 - z) * n + z * h


def full_layer_step(p, x, hidden):
    """Full_layer.forward, fc_rnn=True (models/rlmil.py:208-220).

    hidden None == restart=True (zeros).  Returns (fc(h'), h').  The caller carries
    ``hidden`` exactly like the reference's shared ``self.hidden`` attribute, i.e. the
    two views interleave on one state (hard parts).
    """
    if hidden is None:
        hidden = torch.zeros(x.shape[0], p["rnn.weight_hh_l0"].shape[1], dtype=x.dtype)
    h = gru_cell(x, hidden, p["rnn.weight_ih_l0"], p["rnn.weight_hh_l0"],
                 p["rnn.bias_ih_l0"], p["rnn.bias_hh_l0"])
    return _lin(h, p, "fc"), h


def full_layer_cascade_step(p, x, hidden, feature_num):
    """Full_layer.forward, fc_rnn=False (models/rlmil.py:222-239): ``hidden`` is the running concatenation of the inputs (None ==
    restart=True); the classifier of the current width k * feature_num (k = 2..5) is applied, a single block gives None.
    Returns (logits or None, hidden').  As with the recurrent head the caller carries ONE ``hidden`` for both views."""
    hidden = x if hidden is None else torch.cat([hidden, x], 1)
    k = hidden.shape[1] // feature_num
    if k == 1:
        return None, hidden
    if k not in (2, 3, 4, 5):
        raise RuntimeError("cascade wider than fc_5 (the reference prints the size and exits)")
    return _lin(hidden, p, f"fc_{k}"), hidden


# ----------------------------------------------------------------------------- PPO
LOG_2PI = math.log(2.0 * math.pi)


def _policy_trunk(p, state, hidden):
    e = torch.relu(_lin(state, p, "state_encoder.0"))           # rlmil.py:41-42
    e = torch.relu(_lin(e, p, "state_encoder.2"))               # rlmil.py:43-44
    h = gru_cell(e, hidden, p["gru.weight_ih_l0"], p["gru.weight_hh_l0"],
                 p["gru.bias_ih_l0"], p["gru.bias_hh_l0"])      # rlmil.py:47,78
    return h


def gaussian_logprob(a, mu, std):
    """MultivariateNormal(mu, scale_tril=diag(std)).log_prob(a) (rlmil.py:84-90):
    ``action_var`` is used as the Cholesky factor, so it is a std, not a variance."""
    k = a.shape[-1]
    return (-0.5 * ((a - mu) / std) ** 2).sum(-1) - k * math.log(std) - 0.5 * k * LOG_2PI


def gaussian_entropy(k, std):
    return 0.5 * k * (1.0 + LOG_2PI) + k * math.log(std)


def ppo_act(p, state, hidden, eps, action_std):
    """ActorCritic.act with training=True (rlmil.py:66-97); eps ~ N(0,1) injected.

    Returns (action [B,K] clamped to [0,1], logprob [B], new hidden [B,H])."""
    h = _policy_trunk(p, state.flatten(1), hidden)
    mu = torch.sigmoid(_lin(h, p, "actor.0"))                   # rlmil.py:49-51,82
    a = (mu + action_std * eps).clamp(0.0, 1.0)                 # rlmil.py:86-89
    return a, gaussian_logprob(a, mu, action_std), h


def ppo_evaluate(p, states, actions, action_std):
    """ActorCritic.evaluate (rlmil.py:99-127): states [T,B,S], actions [T,B,K]."""
    T, B = states.shape[:2]
    h = torch.zeros(B, p["gru.weight_hh_l0"].shape[1], dtype=states.dtype)
    hs = []
    for t in range(T):                                          # GRU over seq from zero hidden
        h = _policy_trunk(p, states[t], h)
        hs.append(h)
    hs = torch.stack(hs, 0)
    mu = torch.sigmoid(_lin(hs, p, "actor.0"))
    logp = gaussian_logprob(actions, mu, action_std)
    value = _lin(hs, p, "critic.0").squeeze(-1)
    ent = torch.full_like(logp, gaussian_entropy(actions.shape[-1], action_std))
    return logp, value, ent


def ppo_returns(rewards, gamma):
    """Discounted, normalised returns (rlmil.py:153-162). rewards: list of [1,B]."""
    out, run = [], 0
    for r in reversed(rewards):
        run = r + gamma * run
        out.insert(0, run)
    R = torch.cat(out, 0)
    return (R - R.mean()) / (R.std() + 1e-5)


def ppo_loss(p, states, actions, old_logp, returns, action_std, eps_clip=0.2):
    """Clipped surrogate of PPO.update (rlmil.py:169-181), mean over [T,B]."""
    logp, value, ent = ppo_evaluate(p, states, actions, action_std)
    ratio = torch.exp(logp - old_logp)
    adv = returns - value.detach()
    s1 = ratio * adv
    s2 = ratio.clamp(1 - eps_clip, 1 + eps_clip) * adv
    loss = -torch.min(s1, s2) + 0.5 * F.mse_loss(value, returns) - 0.01 * ent
    return loss.mean()


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam single step (L2 weight decay folded into the gradient)."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k in params:
        g = grads[k]
        if weight_decay != 0.0:
            g = g + weight_decay * params[k]
        m = state.setdefault("m." + k, torch.zeros_like(g))
        v = state.setdefault("v." + k, torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        params[k] = params[k] - (lr / (1 - b1 ** t)) * m / denom
    return params


# ----------------------------------------------------------------------------- pre-train step
def pretrain_step(model_p, fc_p, views_per_t, temperature, arch="ABMIL"):
    """The loss part of train_MuRCL.py:233-291 for already-built sub-bags.

    views_per_t: list over T patch-steps of [x_view0, x_view1], each [B,feat_size,d]
    (the output of get_feats + mixup).  Returns (loss, losses[T], rewards[T-1][B],
    states) with loss = sum_t loss_t / T (train_MuRCL.py:291).  The shared
    Full_layer hidden state is threaded view0 -> view1 -> view0 ... exactly as the
    reference's single ``fc.hidden`` attribute is (rlmil.py:210-218).
    """
    fwd = (lambda x: abmil_forward(model_p, x)[0]) if arch == "ABMIL" else \
          (lambda x: clam_sb_forward(model_p, x)[0])
    hidden, losses, rewards, states = None, [], [], []
    sim_last = None
    for t, (x0, x1) in enumerate(views_per_t):
        o0, o1 = fwd(x0), fwd(x1)                      # cl.py:14 (two encoder calls)
        states.append((o0.detach(), o1.detach()))
        if t == 0:
            z0, hidden = full_layer_step(fc_p, o0, None)       # restart=True, view 0
            z1, hidden = full_layer_step(fc_p, o1, None)       # restart=True, view 1 (overwrites)
        else:
            z0, hidden = full_layer_step(fc_p, o0, hidden)
            z1, hidden = full_layer_step(fc_p, o1, hidden)
        losses.append(nt_xent(z0, z1, temperature))            # train_MuRCL.py:249,277
        sim = row_cosine(z0, z1)
        if t > 0:
            rewards.append((sim_last - sim).detach())          # train_MuRCL.py:283
        sim_last = sim
    return sum(losses) / len(losses), losses, rewards, states
