"""Supervised RL-MIL step bodies on MI355X (reference: train_RLMIL.py:323-392 CLAM, 508-590 DSMIL, 715-781 ABMIL).

``supervised_step`` mirrors the reference's per-batch body for the three aggregators: random (t = 0) then
PPO-chosen sub-bags, aggregator + recurrent classifier head, cross-entropy (plus CLAM's instance loss or DSMIL's
max-instance term), reward = increase of the true-class soft-max confidence, loss averaged over T, Adam or
PPO.update.  Unlike the reference (whose CLAM / DSMIL bodies only run at batch_size 1) the
bodies here are batched over bags.  Model construction follows train_RLMIL.py:90-116.
"""
import os

import torch

from murcl_amd import functional, ops
from murcl_amd.functional import CrossEntropyFn, GroupedCrossEntropyFn, StepCEMeanFn, StepLossFn
from murcl_amd.models import abmil, clam, dsmil, rlmil
from murcl_amd.utils.datasets import draw_step, subbag_views
from murcl_amd.utils.views import as_one


def create_model(arch, dim_patch, num_classes, device, model_dim=512, D=128, size_arg="small", k_sample=8,
                 fc_hidden_dim=1024, dtype=torch.float32, dropout=0.0, fc_rnn=True):
    if arch == "ABMIL":
        model = abmil.ABMIL(dim_in=dim_patch, L=model_dim, D=D, dim_out=num_classes, dropout=dropout)     # train_RLMIL.py:90-96
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
    fc = rlmil.Full_layer(feat, fc_hidden_dim, fc_rnn, num_classes)                                       # :115
    return model.to(device), fc.to(device)


_BATCHED_HEAD = True           # test hook (monkeypatch), not an environment switch: the recurrent head over all patch steps at once


def _confidence(logits, labels):
    """soft-max confidence of the true class (train_RLMIL.py:345,537,735): the fourth output of the cross-entropy launch."""
    from . import ops
    lg = logits.detach().float().contiguous()
    return ops.cross_entropy(lg, labels.to(torch.int64).contiguous(), lg.shape[0], want_conf=True)[3].view(1, -1)


def _aggregate(arch, model, feats, labels):
    """Aggregator over a batch of sub-bags -> (head input [B',F], states [B',S], extra): ``extra`` is what the arch's loss
    needs besides the head logits - CLAM: instance loss per bag [B']; DSMIL: max-instance class scores [B',C]."""
    if arch == "ABMIL":
        out, states = model(feats)
        return out, states, None
    # CLAM / DSMIL: the batched internals of the modules - same kernels as ``model(feats, ...)``, without unpacking the
    # batch into the reference API's per-bag result dicts / lists (and, for CLAM, without the device->host copy of the
    # instance predictions that nothing here reads): 384 bags per call made that 11 of 14.7 ms
    if arch == "CLAM_SB":
        M, _, _, inst_loss, _, _ = model._run(feats if feats.dim() == 3 else feats.unsqueeze(0), labels, True)
        return M, M.detach(), inst_loss
    _, bag, cmax = model._run(feats if feats.dim() == 3 else feats.unsqueeze(0), want_max=True)
    return bag.mean(1), bag.detach().mean(1), cmax                                         # :515-518,527 (cmax = classes.max(1)[0])


def _head_loss(arch, fc, head_in, extra, labels, t, bag_weight):
    logits = fc(head_in, restart=(t == 0))
    ce = CrossEntropyFn.apply(logits, labels)
    if arch == "ABMIL":
        return ce, logits                                                                  # :727
    if arch == "CLAM_SB":
        return bag_weight * ce + (1 - bag_weight) * extra.mean(), logits                   # :336
    return 0.5 * ce + 0.5 * CrossEntropyFn.apply(extra, labels), logits                    # :527-529


def _forward_loss(arch, model, fc, feats, labels, t, bag_weight):
    """Aggregator + recurrent head + the arch's loss for one patch step -> (loss, logits, states)."""
    head_in, states, extra = _aggregate(arch, model, feats, labels)
    loss, logits = _head_loss(arch, fc, head_in, extra, labels, t, bag_weight)
    return loss, logits, states


def _next_action(t, train_stage, ppo, states, memory, B, K, dev, actions, eps=None):
    if actions is not None and t < len(actions):
        return actions[t].to(dev)
    if t == 0 or train_stage == 1:
        return draw_step(dev, (B, K), None, 0, 0, 0.0)[0] if torch.device(dev).type == "cuda" else torch.rand((B, K), device=dev)
    return ppo.select_action(states, memory, restart_batch=(t == 1), eps=None if eps is None else eps[t - 1].to(dev))


def _head_all_steps(arch, fc, head_in_all, extra_all, labels, T, B, bag_weight, memory):
    """The recurrent head, the losses and the rewards of all T patch steps at once: head_in_all [T*B,F] (step-major) -> the T GRU
    steps as one recurrent node, one classifier product and ONE grouped cross-entropy over the T*B rows (loss_t = mean over the B
    rows of step t, exactly the per-step values).  At the reference scripts' --batch_size 1 the step is bound by the number of
    launches issued from Python, and the per-step head was two thirds of them.
    -> (sum_t loss_t / T, [loss_t] detached, rewards [T-1] of [1,B] (also appended to ``memory``), logits of the last step)."""
    lab_all = ops.stack_lists([[labels.contiguous()] * T])[0].view(-1) if labels.is_cuda else labels.repeat(T)     # labels.repeat(T), own launch
    logits_all = fc.forward_sequence(head_in_all.view(T, B, -1))
    if arch == "ABMIL":                                                                    # :727: the step loss is the mean of the T CE values
        total, loss_t, conf = StepCEMeanFn.apply(logits_all, lab_all, B)
        rewards = list(ops.axpby(conf[B:], conf[:-B], 1.0, -1.0).view(T - 1, 1, B).unbind(0)) if T > 1 else []   # :369-371, one launch
        memory.rewards.extend(rewards)
        return total, list(loss_t.unbind(0)), rewards, logits_all[-B:]
    if logits_all.is_cuda and arch in ("CLAM_SB", "DSMIL") and extra_all is not None:
        # :336 / :527-529: the two-term step losses, their mean over the T steps and its gradient in one node (functional.StepLossFn)
        if arch == "CLAM_SB":
            total, loss_t, conf = StepLossFn.apply(logits_all, lab_all, B, bag_weight, extra_all, 1.0 - bag_weight, False)
        else:
            total, loss_t, conf = StepLossFn.apply(logits_all, lab_all, B, 0.5, extra_all, 0.5, True)
        rewards = list(ops.axpby(conf[B:], conf[:-B], 1.0, -1.0).view(T - 1, 1, B).unbind(0)) if T > 1 else []   # :369-371,569-571
        memory.rewards.extend(rewards)
        return total, list(loss_t.unbind(0)), rewards, logits_all[-B:]
    ce, conf = GroupedCrossEntropyFn.apply(logits_all, lab_all, B, True)                # conf: soft-max confidence of the true class (:345,537,735)
    if arch == "ABMIL":
        loss_t = ce                                                                        # :727
    elif arch == "CLAM_SB":
        loss_t = bag_weight * ce + (1 - bag_weight) * extra_all.view(T, B).mean(1)         # :336
    else:
        loss_t = 0.5 * ce + 0.5 * GroupedCrossEntropyFn.apply(extra_all, lab_all, B)       # :527-529
    conf = conf.view(T, 1, B)
    rewards = list((conf[1:] - conf[:-1]).unbind(0))                                       # :369-371,569-571
    memory.rewards.extend(rewards)
    return loss_t.sum() / T, list(loss_t.detach().unbind(0)), rewards, logits_all[-B:]


def supervised_step(arch, model, fc, ppo, optimizer, pack, labels, memory, T=6, feat_size=1024, train_stage=1,
                    bag_weight=0.7, actions=None, return_logits=False, batch_patch_steps=True, eps=None, trace=None):
    """One step on a BagPack with int64 labels [B].  Returns (loss, losses[T], rewards[T-1]) (+ the last patch step's
    logits with ``return_logits``).
    Tests inject the draws: ``actions`` [<=T][B,K] replaces the uniform window positions (and, given for all T steps, the
    sampler), ``eps`` [T-1][B,K] the sampler's Gaussian noise; ``trace`` (a list) receives every step's action tensor and,
    at stages 2 / 3, ``{'logprobs': [T-1,B]}``."""
    B, K, dev = pack.B, pack.K, pack.feats.device
    train_enc = train_stage != 2
    losses, rewards, conf_last, states, loss_total = [], [], None, None, None
    # stage 1 (and injected actions): no patch step depends on the states of the one before, so the sub-bags of all T
    # steps go through the aggregator as ONE batch of T*B bags (cf. train_MuRCL._pretrain_step_all_patch_steps_at_once);
    # the recurrent head and the losses stay per step
    at_once = None
    all_given = actions is not None and len(actions) >= T
    if (train_stage == 1 or all_given) and train_enc and T > 1 and batch_patch_steps:
        if actions is None:
            acts = draw_step(dev, (T, B, K), None, 0, 0, 0.0)[0]             # all T uniform draws in one launch (:345,539,735)
        else:
            acts = [_next_action(t, 1, None, None, memory, B, K, dev, actions) for t in range(T)]
        if trace is not None:
            trace.extend(a.detach().clone() for a in acts)
        views, _ = subbag_views(pack, acts, feat_size, out_dtype=model.compute_dtype)
        at_once = _aggregate(arch, model, as_one(views), labels.repeat(T) if arch == "CLAM_SB" else None)     # (only CLAM's instance branch reads them)
    batched_head = getattr(fc, "fc_rnn", False) and _BATCHED_HEAD
    if at_once is not None and batched_head:
        loss_total, losses, rewards, logits = _head_all_steps(arch, fc, at_once[0], at_once[2], labels, T, B, bag_weight, memory)
    head_ins, extras = [], []
    for t in range(T if not losses else 0):
        if at_once is not None:
            sl = slice(t * B, (t + 1) * B)
            loss, logits = _head_loss(arch, fc, at_once[0][sl], None if at_once[2] is None else at_once[2][sl], labels, t, bag_weight)
            losses.append(loss)
            conf = _confidence(logits, labels)
            if t > 0:
                rewards.append(conf - conf_last)
                memory.rewards.append(rewards[-1])
            conf_last = conf
            continue
        act = _next_action(t, train_stage, ppo, states, memory, B, K, dev, actions, eps)
        if trace is not None:
            trace.append(act.detach().clone())
        (feats,), _ = subbag_views(pack, [act], feat_size, out_dtype=model.compute_dtype)
        with torch.set_grad_enabled(train_enc):
            if batched_head:
                # the sampler picks step t+1's windows from the AGGREGATOR's states; the head's outputs are only needed for the
                # loss and the rewards, so the head waits until all T aggregator outputs exist and then runs once (below)
                head_in, states, extra = _aggregate(arch, model, feats, labels)
                head_ins.append(head_in)
                extras.append(extra)
                continue
            loss, logits, states = _forward_loss(arch, model, fc, feats, labels, t, bag_weight)
        losses.append(loss)
        conf = _confidence(logits, labels)
        if t > 0:
            rewards.append(conf - conf_last)                                                   # :369-371,569-571
            memory.rewards.append(rewards[-1])
        conf_last = conf
    if head_ins:
        with torch.set_grad_enabled(train_enc):
            loss_total, losses, rewards, logits = _head_all_steps(arch, fc, torch.cat(head_ins, 0),
                                                                  None if extras[0] is None else torch.cat(extras, 0),
                                                                  labels, T, B, bag_weight, memory)
    loss = loss_total if loss_total is not None else sum(losses) / T
    if train_enc:
        optimizer.zero_grad()
        with functional.deferred_wgrads():          # the head's T weight gradients per parameter as one product each
            loss.backward(ops.unit_grad(loss))      # (a persistent ones tensor: no fill launch to seed the backward pass)
        optimizer.step()
    else:
        ppo.update(memory)
    if trace is not None and memory.logprobs:
        trace.append({"logprobs": torch.stack(memory.logprobs, 0)})
    memory.clear_memory()
    out = (loss.detach(), [l.detach() for l in losses], rewards)
    return out + (logits.detach(),) if return_logits else out


# ------------------------------------------------------------------------------------ validation / test (8(f) rank 2)
def get_metrics(outputs, targets):
    """(acc, auc, precision, recall, f1) exactly as utils/general.py:174-200: arg-max accuracy, soft-max AUC (one-vs-rest
    for more than two classes), binary / macro precision-recall-F1 from scikit-learn."""
    from sklearn.metrics import precision_recall_fscore_support, roc_auc_score
    with torch.no_grad():
        assert outputs.shape[0] == targets.shape[0]
        multi = outputs.shape[1] > 2
        preds = outputs.argmax(1)
        acc = (preds.eq(targets).sum() / targets.shape[0]).item()
        t = targets.cpu().numpy().astype(int).reshape(-1)
        probs = torch.softmax(outputs.float(), 1).cpu().numpy()
        auc = roc_auc_score(t, probs, multi_class="ovr") if multi else roc_auc_score(t, probs[:, 1])
        precision, recall, f1, _ = precision_recall_fscore_support(t, preds.cpu().numpy(), average="macro" if multi else "binary")
    return acc, auc, precision, recall, f1


def get_score(acc, auc, precision, recall, f1_score):
    """Model-selection score (utils/general.py:203-204)."""
    return 0.3 * acc + 0.3 * auc + 0.1 * precision + 0.1 * recall + 0.2 * f1_score


def evaluate_split(arch, model, fc, ppo, memory, pack, labels, T=6, feat_size=1024, train_stage=1, bag_weight=0.7,
                   actions=None):
    """The reference's test_ABMIL / test_CLAM / test_DSMIL (train_RLMIL.py:410-472,607-679,799-854): the WHOLE split as
    one batch, T forward-only patch steps in eval mode, metrics on the last step's logits.
    Returns (loss of the last patch step, acc, auc, precision, recall, f1, logits [S,C], labels)."""
    B, K, dev = pack.B, pack.K, pack.feats.device
    was = (model.training, fc.training)
    model.eval(), fc.eval()
    conf_last = states = None
    with torch.no_grad():
        for t in range(T):
            act = _next_action(t, train_stage, ppo, states, memory, B, K, dev, actions)
            (feats,), _ = subbag_views(pack, [act], feat_size, out_dtype=model.compute_dtype)
            loss, logits, states = _forward_loss(arch, model, fc, feats, labels, t, bag_weight)
            conf = _confidence(logits, labels)
            if t > 0:
                memory.rewards.append(conf - conf_last)
            conf_last = conf
        memory.clear_memory()
    model.train(was[0]), fc.train(was[1])
    return (loss.item(), *get_metrics(logits, labels), logits, labels)


def predictions_frame(logits, labels, case_ids):
    """pred.csv of the reference's ``test`` (train_RLMIL.py:984-1002): label, pred, correct, prob0.. indexed by case_id."""
    import pandas as pd
    prob = torch.softmax(logits.float(), 1).cpu()
    pred = prob.argmax(1)
    lab = labels.cpu()
    df = pd.DataFrame({"label": lab.tolist(), "pred": pred.tolist(), "correct": (lab == pred).tolist(),
                       **{f"prob{j}": prob[:, j].tolist() for j in range(prob.shape[1])}}, index=list(case_ids))
    df.index.rename("case_id", inplace=True)
    return df


class BestPick:
    """Best-epoch selection on the validation split (train_RLMIL.py:900-915): 'acc' | 'auc' | 'score' maximise, 'loss' minimises."""

    def __init__(self, method="score"):
        if method not in ("acc", "loss", "auc", "score"):
            raise ValueError("picked_method error. ")
        self.method, self.best, self.epoch = method, None, 0

    def update(self, epoch, loss, acc, auc, precision, recall, f1):
        v = {"acc": acc, "loss": -loss, "auc": auc, "score": get_score(acc, auc, precision, recall, f1)}[self.method]
        if self.best is None or v > self.best:
            self.best, self.epoch = v, epoch
            return True
        return False


def fit(arch, model, fc, ppo, optimizer, stores, epochs, batch_size, T=6, feat_size=1024, train_stage=1, bag_weight=0.7,
        picked_method="score", rng=None, log=print, scheduler=None, warmup=0, patience=None, save_dir=None, save_model=False,
        tb_writer=None):
    """Epoch loop of the reference's ``train`` (train_RLMIL.py:856-975) on HBM-resident splits.
    ``stores`` = (train, valid, test) DeviceSlideStore with labels.  Returns (best state dict, final test tuple, pred frame).
    With ``save_dir``: the reference's csv logs (losses / accs / aucs / results) and, with ``save_model``, the best
    checkpoint as soon as it improves (:938-941)."""
    import copy
    import numpy as np
    from murcl_amd.models import rlmil as _rl
    from murcl_amd.utils import checkpoint as C, general as G
    rng = rng or np.random.default_rng(985)
    train, valid, test = stores
    dev = train.feats.device
    lab = [torch.from_numpy(s.labels).to(dev) for s in stores]
    memory, pick, best, final, frame = _rl.Memory(), BestPick(picked_method), None, None, None
    logs = None
    if save_dir is not None:
        hdr = ["epoch", "train", "valid", "test", "best_train", "best_valid", "best_test"]
        logs = {k: G.CsvLog(os.path.join(save_dir, f"{k}.csv"), hdr) for k in ("losses", "accs", "aucs")}
        logs["results"] = G.CsvLog(os.path.join(save_dir, "results.csv"),
                                   ["epoch", "final_epoch", "final_loss", "final_acc", "final_auc", "final_precision",
                                    "final_recall", "final_f1_score"])
    bests = {k: [G.Best("min" if k == "losses" else "max") for _ in range(3)] for k in ("losses", "accs", "aucs")}
    early_stop = G.EarlyStop(patience) if patience is not None else None
    for epoch in range(epochs):
        # train_RLMIL.py:299-304,484-489,691-696: the PPO-only stage scores the sampler with the aggregator and the head in
        # eval mode (CLAM's Dropout(0.25) off); evaluate_split below restores whatever mode it finds
        (model.eval(), fc.eval()) if train_stage == 2 else (model.train(), fc.train())
        order, tl, outs, ys = rng.permutation(len(train)), [], [], []
        for s in range(0, len(order) - batch_size + 1, batch_size):
            sel = order[s:s + batch_size]
            loss, _, _, logits = supervised_step(arch, model, fc, ppo, optimizer, train.pack(sel), lab[0][torch.from_numpy(sel).to(dev)],
                                                 memory, T, feat_size, train_stage, bag_weight, return_logits=True)
            tl.append(loss)
            outs.append(logits)
            ys.append(lab[0][torch.from_numpy(sel).to(dev)])
        if scheduler is not None and epoch >= warmup:
            scheduler.step()                                                            # :400-401,598-599,789-790
        tr_loss = torch.stack(tl).mean().item()
        tr = (tr_loss, *get_metrics(torch.cat(outs), torch.cat(ys)))
        v = evaluate_split(arch, model, fc, ppo, memory, valid.pack(range(len(valid))), lab[1], T, feat_size, train_stage, bag_weight)
        te = evaluate_split(arch, model, fc, ppo, memory, test.pack(range(len(test))), lab[2], T, feat_size, train_stage, bag_weight)
        if tb_writer is not None:
            tb_writer.add_scalar("train/1.train_loss", tr_loss, epoch)
            tb_writer.add_scalar("test/2.test_loss", v[0], epoch)
        if pick.update(epoch + 1, *v[:6]):
            final = (epoch + 1, *te[:6])
            frame = predictions_frame(te[6], te[7], test.case_ids)
            best = copy.deepcopy(C.make_state(epoch + 1, model, fc, optimizer, ppo))         # keys of :930-937
            if save_model and save_dir is not None:
                C.save_checkpoint(best, True, save_dir)
        for k, col in (("losses", 0), ("accs", 1), ("aucs", 2)):
            vals = (tr[col], v[col], te[col])
            for b_, x in zip(bests[k], vals):
                b_.compare(x, epoch + 1, inplace=True)
            if logs is not None:
                logs[k].write_row([epoch + 1, *vals, *[(b_.best, b_.epoch) for b_ in bests[k]]])
        if logs is not None:
            logs["results"].write_row([epoch + 1, final[0], *te[:6]])
        log(f"epoch {epoch + 1}: train loss {tr_loss:.4f} acc {tr[1]:.4f} | valid loss {v[0]:.4f} acc {v[1]:.4f} auc {v[2]:.4f} | "
            f"test loss {te[0]:.4f} acc {te[1]:.4f} auc {te[2]:.4f} | final epoch {final[0]}")
        if early_stop is not None:
            early_stop.update((bests["losses"][1].best, bests["accs"][1].best, bests["aucs"][1].best))    # :971-974
            if early_stop.is_stop():
                break
    if tb_writer is not None:
        tb_writer.close()
    return best, final, frame


# ------------------------------------------------------------------------------------------------------ script entry
class _SyntheticLabelled:
    """Random labelled slides in the WSIWithCluster item format; class-1 slides carry a shifted feature block."""

    def __init__(self, n, n_patches, dim, num_clusters, seed):
        import numpy as np
        self.n, self.N, self.d, self.K, self.seed, self.np = n, n_patches, dim, num_clusters, seed, np
        self.patch_dim, self.num_clusters = dim, num_clusters

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        np = self.np
        r = np.random.default_rng(self.seed * 100003 + i)
        y = i % 2
        feat = np.abs(r.standard_normal((self.N, self.d), dtype=np.float32)) * 0.5
        feat[:, : self.d // 8] += 0.5 * y
        lab = r.integers(0, self.K, self.N)
        return torch.from_numpy(feat), [np.nonzero(lab == k)[0].tolist() for k in range(self.K)], y, f"case{self.seed}_{i}"


def build_parser():
    """Every flag of the reference's parser with its type, default, choices and action (train_RLMIL.py:1060-1153; pinned by
    tests/golden/g13_cli_flags.json), plus the murcl_amd extras at the end."""
    import argparse
    p = argparse.ArgumentParser()
    # Data
    p.add_argument("--dataset", type=str, default="Camelyon16", help="dataset name (only names the result directory)")
    p.add_argument("--data_csv", type=str, default="")
    p.add_argument("--data_split_json", type=str, default="/path/to/data_split.json")
    p.add_argument("--train_data", type=str, default="train", choices=["train", "train_sub_per10"])
    p.add_argument("--preload", action="store_true", default=False,
                   help="preload the patch features (murcl_amd keeps the splits resident in HBM either way)")
    p.add_argument("--feat_size", default=1024, type=int)
    # Train
    p.add_argument("--train_method", type=str, default="scratch", choices=["scratch", "finetune", "linear"])
    p.add_argument("--train_stage", default=1, type=int)
    p.add_argument("--T", default=6, type=int)
    p.add_argument("--checkpoint_stage", default=None, type=str)
    p.add_argument("--checkpoint_pretrained", type=str, default=None)
    p.add_argument("--optimizer", type=str, default="Adam", choices=["Adam", "SGD"])
    p.add_argument("--scheduler", type=str, default=None, choices=[None, "StepLR", "CosineAnnealingLR"])
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--epochs", type=int, default=40)
    p.add_argument("--ppo_epochs", type=int, default=10)
    p.add_argument("--backbone_lr", default=1e-4, type=float)
    p.add_argument("--fc_lr", default=1e-4, type=float)
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--nesterov", action="store_true", default=True)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--beta2", type=float, default=0.999)
    p.add_argument("--warmup", default=0, type=float)
    p.add_argument("--wdecay", default=1e-5, type=float)
    p.add_argument("--picked_method", type=str, default="score")
    p.add_argument("--patience", type=int, default=None)
    # Architecture
    p.add_argument("--arch", default="CLAM_SB", type=str, choices=["ABMIL", "CLAM_SB", "DSMIL"])
    p.add_argument("--num_classes", type=int, default=2)
    p.add_argument("--model_dim", type=int, default=512)
    p.add_argument("--policy_hidden_dim", type=int, default=512)
    p.add_argument("--policy_conv", action="store_true", default=False)
    p.add_argument("--action_std", type=float, default=0.5)
    p.add_argument("--ppo_lr", type=float, default=0.00001)
    p.add_argument("--ppo_gamma", type=float, default=0.1)
    p.add_argument("--K_epochs", type=int, default=3)
    p.add_argument("--feature_num", type=int, default=512)
    p.add_argument("--fc_hidden_dim", type=int, default=1024)
    p.add_argument("--fc_rnn", action="store_true", default=True)
    p.add_argument("--load_fc", action="store_true", default=False, help="accepted; the reference never reads it either")
    p.add_argument("--L", type=int, default=512)
    p.add_argument("--D", type=int, default=128)
    p.add_argument("--dropout", type=float, default=0.0)
    p.add_argument("--size_arg", type=str, default="small", choices=["small", "big"])
    p.add_argument("--k_sample", type=int, default=8)
    p.add_argument("--bag_weight", type=float, default=0.7)
    p.add_argument("--loss", default="CrossEntropyLoss", type=str, choices=["CrossEntropyLoss"])
    p.add_argument("--use_tensorboard", action="store_true", default=False)
    # Save
    p.add_argument("--base_save_dir", type=str, default="./results")
    p.add_argument("--save_dir", type=str, default=None)
    p.add_argument("--save_dir_flag", type=str, default=None)
    p.add_argument("--exist_ok", action="store_true", default=False)
    p.add_argument("--save_model", action="store_true", default=False)
    # Global
    p.add_argument("--device", default="2", help="cuda device, i.e. 0 or 0,1,2,3 (one process per GPU takes its LOCAL_RANK-th entry)")
    p.add_argument("--seed", type=int, default=985)
    # murcl_amd extras
    x = p.add_argument_group("murcl_amd")
    x.add_argument("--synthetic", type=str, default=None, help="n_train,n_valid,n_test,n_patches (random labelled slides)")
    x.add_argument("--num_clusters", default=10, type=int, help="clusters per slide for --synthetic (else read from the csv name)")
    x.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="storage type of patch-level tensors")
    return p


def get_optimizer(args, model, fc):
    """train_RLMIL.py:255-272 (frozen parameters of the linear protocol stay out of the flat buffers)."""
    from murcl_amd.optim import FlatAdam, FlatSGD
    if args.train_stage == 2:
        args.epochs = args.ppo_epochs
        return None
    groups = [{"params": [p for p in model.parameters() if p.requires_grad], "lr": args.backbone_lr},
              {"params": list(fc.parameters()), "lr": args.fc_lr}]
    groups = [g for g in groups if g["params"]]
    if args.optimizer == "SGD":
        return FlatSGD(groups, momentum=args.momentum, nesterov=args.nesterov, weight_decay=args.wdecay)
    if args.optimizer == "Adam":
        return FlatAdam(groups, betas=(args.beta1, args.beta2), weight_decay=args.wdecay)
    raise NotImplementedError(args.optimizer)


def run(args):
    """train_RLMIL.py:1005-1057."""
    import json
    import numpy as np
    import pandas as pd
    from murcl_amd.optim import make_scheduler
    from murcl_amd.utils import checkpoint as C, general as G
    from murcl_amd.utils.datasets import DeviceSlideStore
    G.init_seeds(args.seed)
    G.prepare_run_dir(args, "RLMIL")
    dev = G.pick_device(args.device, int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dt_ = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.synthetic:
        ntr, nva, nte, N = (int(v) for v in args.synthetic.split(","))
        sets = [_SyntheticLabelled(n, N, 512, args.num_clusters, s) for n, s in ((ntr, 1), (nva, 2), (nte, 3))]
    else:
        from murcl_amd.train_MuRCL import WSIWithCluster
        split = json.load(open(args.data_split_json))
        sets = [WSIWithCluster(args.data_csv, split[k]) for k in (args.train_data, "valid", "test")]
        args.num_clusters = sets[0].num_clusters
    stores = tuple(DeviceSlideStore.from_dataset(s, dev, dtype=dt_) for s in sets)
    dim_patch = stores[0].patch_dim
    # train_RLMIL.py:88-116: ABMIL takes --L, the head's input width follows the aggregator's output
    model, fc = create_model(args.arch, dim_patch, args.num_classes, dev, model_dim=args.L, D=args.D, size_arg=args.size_arg,
                             k_sample=args.k_sample, fc_hidden_dim=args.fc_hidden_dim, dtype=dt_, dropout=args.dropout,
                             fc_rnn=args.fc_rnn)
    args.feature_num = fc.feature_num
    ppo = None
    if args.train_stage in (2, 3):
        if args.policy_conv:
            raise NotImplementedError("--policy_conv is never enabled by the reference's launch scripts and is not built")
        ppo = rlmil.PPO(dim_patch, args.model_dim, args.policy_hidden_dim, False, action_std=args.action_std, lr=args.ppo_lr,
                        gamma=args.ppo_gamma, K_epochs=args.K_epochs, action_size=args.num_clusters)
        if args.checkpoint_stage is None:
            args.checkpoint_stage = C.stage_checkpoint_path(args.save_dir, args.train_stage)
        assert os.path.exists(args.checkpoint_stage), f"{args.checkpoint_stage} is not exist!"
        if args.train_stage == 2:
            # scratch: a fresh sampler (:199-214); finetune / linear: the pre-trained one (:150-163)
            pretrained = args.train_method in ("finetune", "linear")
            if pretrained:
                assert args.checkpoint_pretrained is not None and os.path.exists(args.checkpoint_pretrained), \
                    f"{args.checkpoint_pretrained} is not exists!"
            C.load_stage(model, fc, ppo, args.checkpoint_stage, policy_ckpt=args.checkpoint_pretrained if pretrained else None,
                         load_policy=pretrained)
        else:
            C.load_stage(model, fc, ppo, args.checkpoint_stage)
            if args.train_method == "linear":
                C.freeze_backbone(model)
    elif args.train_stage != 1:
        raise ValueError(args.train_stage)
    elif args.train_method in ("finetune", "linear"):
        assert args.checkpoint_pretrained is not None and os.path.exists(args.checkpoint_pretrained), \
            f"{args.checkpoint_pretrained} is not exists!"
        print("msg_model missing_keys:", C.load_pretrained(model, args.checkpoint_pretrained, args.train_method))
    optimizer = get_optimizer(args, model, fc)
    scheduler = make_scheduler(optimizer, args.scheduler, args.epochs, args.warmup)
    G.dump_args(args, args.save_dir)
    tb_writer = G.tensorboard_writer(args.save_dir, args.use_tensorboard)
    best, final, frame = fit(args.arch, model, fc, ppo, optimizer, stores, args.epochs, args.batch_size, args.T, args.feat_size,
                             args.train_stage, args.bag_weight, args.picked_method, np.random.default_rng(args.seed),
                             scheduler=scheduler, warmup=args.warmup, patience=args.patience, save_dir=args.save_dir,
                             save_model=args.save_model, tb_writer=tb_writer)
    frame.to_csv(os.path.join(args.save_dir, "pred.csv"))                               # :1051-1056
    res = pd.DataFrame(columns=["loss", "acc", "auc", "precision", "recall", "f1_score"])
    res.loc[f"seed{args.seed}"] = [float(v) for v in final[1:]]
    res.to_csv(os.path.join(args.save_dir, "final_res.csv"))
    print("final (epoch, loss, acc, auc, precision, recall, f1):", tuple(round(float(v), 4) for v in final))
    return final


def main(argv=None):
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
