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
                     action_std=0.5, gamma=0.1, K_epochs=3, ppo_lr=1e-5, arch="ABMIL"):
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
        st = {}
        for m in mem:                                                            # :297-298
            policy_p = ppo_update(policy_p, st, m, gamma, K_epochs, action_std, ppo_lr)
        res["policy"] = policy_p
    return res
