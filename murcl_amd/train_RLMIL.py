"""Supervised RL-MIL step bodies on MI355X (reference: train_RLMIL.py:323-392 CLAM, 508-590 DSMIL, 715-781 ABMIL).

``supervised_step`` mirrors the reference's per-batch body for the three aggregators: random (t = 0) then
PPO-chosen sub-bags, aggregator + recurrent classifier head, cross-entropy (plus CLAM's instance loss or DSMIL's
max-instance term), reward = increase of the true-class soft-max confidence, loss averaged over T, Adam or
PPO.update.  Unlike the reference (whose CLAM / DSMIL bodies only run at batch_size 1) the
bodies here are batched over bags.  Model construction follows train_RLMIL.py:90-116.
"""
import torch

from murcl_amd.functional import CrossEntropyFn
from murcl_amd.models import abmil, clam, dsmil, rlmil
from murcl_amd.utils.datasets import subbag_views


def create_model(arch, dim_patch, num_classes, device, model_dim=512, D=128, size_arg="small", k_sample=8,
                 fc_hidden_dim=1024, dtype=torch.float32):
    if arch == "ABMIL":
        model = abmil.ABMIL(dim_in=dim_patch, L=model_dim, D=D, dim_out=num_classes)
        feat = model_dim
    elif arch == "CLAM_SB":
        model = clam.CLAM_SB(gate=True, size_arg=size_arg, dropout=True, k_sample=k_sample, n_classes=num_classes,
                             subtyping=True, in_dim=dim_patch)
        feat = 512
    elif arch == "DSMIL":
        model = dsmil.build_dsmil(dim_patch, num_classes)
        feat = dim_patch
    else:
        raise NotImplementedError(arch)
    model.compute_dtype = dtype
    fc = rlmil.Full_layer(feat, fc_hidden_dim, True, num_classes)
    return model.to(device), fc.to(device)


def _confidence(logits, labels):
    """soft-max confidence of the true class (train_RLMIL.py:345,537,735)."""
    return torch.softmax(logits.detach(), 1).gather(1, labels.view(-1, 1)).view(1, -1)


def supervised_step(arch, model, fc, ppo, optimizer, pack, labels, memory, T=6, feat_size=1024, train_stage=1,
                    bag_weight=0.7, actions=None):
    """One step on a BagPack with int64 labels [B].  Returns (loss, losses[T], rewards[T-1])."""
    B, K, dev = pack.B, pack.K, pack.feats.device
    train_enc = train_stage != 2
    losses, rewards, conf_last, states = [], [], None, None
    for t in range(T):
        if actions is not None:
            act = actions[t].to(dev)
        elif t == 0 or train_stage == 1:
            act = torch.rand((B, K), device=dev)
        else:
            act = ppo.select_action(states, memory, restart_batch=(t == 1))
        (feats,), _ = subbag_views(pack, [act], feat_size, out_dtype=model.compute_dtype)
        with torch.set_grad_enabled(train_enc):
            if arch == "ABMIL":
                out, states = model(feats)
                logits = fc(out, restart=(t == 0))
                loss = CrossEntropyFn.apply(logits, labels)                                    # :727
            elif arch == "CLAM_SB":
                out, states, res = model(feats, label=labels, instance_eval=True)
                logits = fc(out, restart=(t == 0))
                inst = torch.stack([r["instance_loss"] for r in res]).mean()
                loss = bag_weight * CrossEntropyFn.apply(logits, labels) + (1 - bag_weight) * inst   # :336
            else:
                classes, bag, bag_det = model(feats)
                states = bag_det.mean(1)                                                       # :515
                cls = torch.stack(classes) if isinstance(classes, list) else classes.unsqueeze(0)
                logits = fc(bag.mean(1), restart=(t == 0))                                     # :517-518
                loss = 0.5 * CrossEntropyFn.apply(logits, labels) + 0.5 * CrossEntropyFn.apply(cls.max(1)[0], labels)   # :527-529
        losses.append(loss)
        conf = _confidence(logits, labels)
        if t > 0:
            rewards.append(conf - conf_last)                                                   # :369-371,569-571
            memory.rewards.append(rewards[-1])
        conf_last = conf
    loss = sum(losses) / T
    if train_enc:
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    else:
        ppo.update(memory)
    memory.clear_memory()
    return loss.detach(), [l.detach() for l in losses], rewards
