"""CPU oracle: the whole MuRCL hot step with the PPO sub-bag sampler in the loop (BASELINE config 4's body).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Composition of the pieces in mil_oracle.py / select_oracle.py in
the order of the reference's batch body (train_MuRCL.py:233-304), every random draw injected:

    t = 0:  random window positions (:235) -> get_feats -> mixup -> CL(aggregator) -> Full_layer(restart) -> NT-Xent
    t >= 1: ppo.select_action(states of step t-1, memory_v, restart_batch = (t == 1)) per view (:259-265) -> get_feats
            -> mixup -> aggregator -> Full_layer -> NT-Xent; reward = cos_{t-1} - cos_t appended to BOTH memories (:282-288)
    stage 2: aggregator/head under no_grad, then ppo.update(memory_0); ppo.update(memory_1) (:297-298) - one policy, one
             Adam state, policy_old re-synchronised after each update (rlmil.py:152-184)
    stage 3: loss = sum_t loss_t / T -> backward -> Adam on (model, fc) (:291-295); the policy only samples.

Pinned by tests/golden/g12_rl_step.npz, which oracle/gen_goldens.py produces by running the reference's own
``train()`` loop for one batch.
"""
import numpy as np
import torch

from . import mil_oracle as O, select_oracle as S


def ppo_update(policy_p, adam_state, memory, gamma, K_epochs, action_std, lr, eps_clip=0.2, betas=(0.9, 0.999),
               returns_stats=None):
    """PPO.update (rlmil.py:152-184) on a parameter dict; returns the new dict (Adam state updated in place).

    ``returns_stats`` = (mean, std) overrides the normalisation statistics (the data-parallel form normalises with the
    statistics of ALL ranks' returns)."""
    out, run = [], 0
    for r in reversed(memory["rewards"]):                       # rlmil.py:153-158
        run = r + gamma * run
        out.insert(0, run)
    R = torch.cat(out, 0)
    if returns_stats is None:
        R = (R - R.mean()) / (R.std() + 1e-5)                   # rlmil.py:162
    else:
        R = (R - returns_stats[0]) / (returns_stats[1] + 1e-5)
    states = torch.stack(memory["states"], 0)
    actions = torch.stack(memory["actions"], 0)
    old_logp = torch.stack(memory["logprobs"], 0)
    for _ in range(K_epochs):
        p = {k: v.detach().clone().requires_grad_() for k, v in policy_p.items()}
        loss = O.ppo_loss(p, states, actions, old_logp, R, action_std, eps_clip)
        loss.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
        policy_p = O.adam_step({k: v.detach() for k, v in p.items()}, grads, adam_state, lr, betas=betas)
    return policy_p


def pretrain_step_rl(model_p, fc_p, policy_p, feats, clusters, inj, *, T, feat_size, stage, temperature=1.0,
                     action_std=0.5, gamma=0.1, K_epochs=3, ppo_lr=1e-5, arch="ABMIL", ppo_state=None):
    """One batch of train_MuRCL.py:233-304 at train_stage 2 or 3.

    feats: list of B [N_i,d] float32 arrays; clusters: B lists of K ascending id lists; ``inj``: 'actions'[0][v] [B,K],
    'draws'[t][v] = (lambda [B,1], perm [B]), 'eps'[t-1][v] [B,K].
    Returns dict(loss, losses[T], rewards[T-1] each [B], actions[T][2], ids[T][2] (selected patch ids per bag),
    policy (post-update parameter dict; unchanged at stage 3), memories)."""
    B = len(feats)
    fwd = (lambda x: O.abmil_forward(model_p, x)[0]) if arch == "ABMIL" else (lambda x: O.clam_sb_forward(model_p, x)[0])
    H = policy_p["gru.weight_hh_l0"].shape[1]
    mem = [dict(states=[], actions=[], logprobs=[], rewards=[]) for _ in range(2)]
    pol_hidden = [None, None]
    hidden, losses, rewards, sim_last, states = None, [], [], None, None
    actions_all, ids_all = [], []
    grad_on = stage != 2
    for t in range(T):
        if t == 0:
            acts = [torch.as_tensor(np.asarray(a, np.float32)) for a in inj["actions"][0]]
        else:
            acts = []
            for v in range(2):
                if t == 1:                                                       # restart_batch=True (:259-262)
                    pol_hidden[v] = torch.zeros(B, H)
                with torch.no_grad():
                    a, logp, pol_hidden[v] = O.ppo_act(policy_p, states[v], pol_hidden[v],
                                                       torch.as_tensor(np.asarray(inj["eps"][t - 1][v], np.float32)), action_std)
                mem[v]["states"].append(states[v])
                mem[v]["actions"].append(a)
                mem[v]["logprobs"].append(logp)
                acts.append(a)
        actions_all.append([a.clone() for a in acts])
        xs, ids_t = [], []
        for v in range(2):
            sub, ids = S.get_feats(feats, clusters, acts[v].numpy(), feat_size)      # :237-238,266-267
            lam, perm = inj["draws"][t][v]
            xs.append(torch.from_numpy(S.mixup(sub, np.asarray(lam, np.float32), np.asarray(perm))))   # :239,268
            ids_t.append(ids)
        ids_all.append(ids_t)
        with torch.set_grad_enabled(grad_on):
            o0, o1 = fwd(xs[0]), fwd(xs[1])                                      # :242,271
            states = (o0.detach(), o1.detach())
            z0, hidden = O.full_layer_step(fc_p, o0, None if t == 0 else hidden)  # view 0 (restart at t = 0)
            z1, hidden = O.full_layer_step(fc_p, o1, None if t == 0 else hidden)  # view 1: the SHARED fc.hidden (rlmil.py:210-218)
            losses.append(O.nt_xent(z0, z1, temperature))                        # :249,277
            sim = O.row_cosine(z0, z1).detach()
        if t > 0:
            reward = sim_last - sim                                              # :283
            rewards.append(reward)
            for m in mem:
                m["rewards"].append(reward.view(1, -1))
        sim_last = sim
    loss = sum(losses) / T                                                       # :291
    res = dict(loss=loss, losses=losses, rewards=rewards, actions=actions_all, ids=ids_all, memories=mem, policy=policy_p)
    if stage == 2:
        st = ppo_state if ppo_state is not None else {}          # the sampler's ONE Adam state lives across batches (rlmil.py:141)
        for m in mem:                                                            # :297-298
            policy_p = ppo_update(policy_p, st, m, gamma, K_epochs, action_std, ppo_lr)
        res["policy"] = policy_p
    return res


def supervised_step_rl(arch, model_p, fc_p, policy_p, feats, clusters, labels, u, eps, *, T, feat_size, stage,
                       bag_weight=0.7, k_sample=8, action_std=0.5, gamma=0.1, K_epochs=3, ppo_lr=1e-5, lr=1e-4, fc_lr=1e-4,
                       wd=1e-5, adam_state=None):
    """One batch of the supervised bodies train_ABMIL / train_CLAM / train_DSMIL (train_RLMIL.py:715-781, 323-392, 508-590),
    batched over the B slides of ``feats`` (the reference's CLAM / DSMIL bodies only run at B = 1; every term below is a
    mean over bags, so B = 1 reproduces them and B > 1 is their per-bag average).

    u: [T][B,K] uniform window positions (t >= 1 used at stage 1 only, :347-348); eps: [T-1][B,K] sampler noise (stages 2, 3).
    adam_state: dict carried between calls ({'model': {}, 'fc': {}, 'ppo': {}}).
    Returns dict(loss, losses[T], rewards [T-1,B], actions [T][B,K], logp [T-1,B], ids[T], logits (last step), model, fc,
    policy = post-update parameter dicts)."""
    import torch.nn.functional as F
    B = len(feats)
    y = torch.as_tensor(np.asarray(labels)).long()
    C = fc_p["fc.weight"].shape[0]
    grad_on = stage != 2
    mp = {k: v.detach().clone().requires_grad_(grad_on) for k, v in model_p.items()}
    fp = {k: v.detach().clone().requires_grad_(grad_on) for k, v in fc_p.items()}
    mem = dict(states=[], actions=[], logprobs=[], rewards=[])
    hidden = pol_hidden = states = conf_last = None
    losses, rewards, actions_all, ids_all = [], [], [], []
    for t in range(T):
        if t == 0 or stage == 1:
            a = torch.as_tensor(np.asarray(u[t], np.float32))                            # :331,347-348
        else:
            if t == 1:                                                                   # restart_batch=True (:351-352)
                pol_hidden = torch.zeros(B, policy_p["gru.weight_hh_l0"].shape[1])
            with torch.no_grad():
                a, logp, pol_hidden = O.ppo_act(policy_p, states, pol_hidden, torch.as_tensor(np.asarray(eps[t - 1], np.float32)),
                                                action_std)
            mem["states"].append(states)
            mem["actions"].append(a)
            mem["logprobs"].append(logp)
        actions_all.append(a.clone())
        sub, ids = S.get_feats(feats, clusters, a.numpy(), feat_size)                    # :332,355
        ids_all.append(ids)
        x = torch.from_numpy(sub)
        with torch.set_grad_enabled(grad_on):
            if arch == "ABMIL":
                out = O.abmil_forward(mp, x)[0]
                states = out.detach()
                logits, hidden = O.full_layer_step(fp, out, None if t == 0 else hidden)
                loss = F.cross_entropy(logits, y)                                        # :727,750
            elif arch == "CLAM_SB":
                M, A, _, h = O.clam_sb_forward(mp, x)
                states = M.detach()
                inst = torch.stack([O.clam_instance_eval(mp, A[b], h[b], int(y[b]), C, k_sample, True)[0] for b in range(B)]).mean()
                logits, hidden = O.full_layer_step(fp, M, None if t == 0 else hidden)
                loss = bag_weight * F.cross_entropy(logits, y) + (1 - bag_weight) * inst  # :336,364
            else:
                c, bag, _, _ = O.dsmil_forward(mp, x)
                states = bag.detach().mean(1)                                             # :516,549
                logits, hidden = O.full_layer_step(fp, bag.mean(1), None if t == 0 else hidden)   # :518-519
                loss = 0.5 * F.cross_entropy(logits, y) + 0.5 * F.cross_entropy(c.max(1)[0], y)   # :527-529
        losses.append(loss)
        conf = torch.softmax(logits.detach(), 1).gather(1, y.view(-1, 1)).view(1, -1)     # :345,367
        if t > 0:
            rewards.append(conf - conf_last)                                              # :368
            mem["rewards"].append(rewards[-1])
        conf_last = conf
    total = sum(losses) / T                                                               # :374
    adam_state = adam_state if adam_state is not None else {}
    res = dict(loss=total.detach(), losses=[l.detach() for l in losses], rewards=torch.cat(rewards, 0), actions=actions_all,
               ids=ids_all, logits=logits.detach(), logp=torch.stack(mem["logprobs"], 0) if mem["logprobs"] else None,
               model=model_p, fc=fc_p, policy=policy_p)
    if stage != 2:
        total.backward()                                                                  # :375-378
        for name, p, rate in (("model", mp, lr), ("fc", fp, fc_lr)):
            live = {k: v for k, v in p.items() if v.grad is not None}                     # torch's Adam skips grad-less params
            new = O.adam_step({k: v.detach() for k, v in live.items()}, {k: v.grad for k, v in live.items()},
                              adam_state.setdefault(name, {}), rate, weight_decay=wd)
            res[name] = {k: new.get(k, v.detach()) for k, v in p.items()}
    else:
        res["policy"] = ppo_update(policy_p, adam_state.setdefault("ppo", {}), mem, gamma, K_epochs, action_std, ppo_lr)   # :380
    return res
