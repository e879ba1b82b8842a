"""Drop-in DSMIL (reference: models/dsmil.py): ``FCLayer``, ``BClassifier``, ``MILNet``, ``build_dsmil``.

State-dict keys match the reference (``i_classifier.fc.0.*``, ``b_classifier.q.*``,
``b_classifier.v.1.*``, ``b_classifier.fcc.*`` - the Conv1d is built and never applied, dsmil.py:62,80),
``MILNet.forward`` returns ``(classes, prediction_bag, prediction_bag.detach())`` with ``classes`` a
tensor ``[N,C]`` for a single ``[1,N,d]`` bag and a list of ``[N,C]`` for a batch, as the reference does.
All bags of a batch run through the kernels at once (murcl_amd.functional.DSMILFn).
"""
import torch
from torch import nn

from ..functional import DSMILFn


class FCLayer(nn.Module):
    def __init__(self, in_size, out_size=1):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(in_size, out_size))


class BClassifier(nn.Module):
    def __init__(self, input_size, output_class, dropout_v=0.0):
        super().__init__()
        if not 0.0 <= dropout_v < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {dropout_v}")
        # dropout_v > 0 (no reference script sets it, dsmil.py:118): Dropout on the value branch's input while training - built for
        # the constructor's contract (round 6), on the explicit chain of DSMILFn instead of its one-pass kernels
        self.dropout_v = float(dropout_v)
        self.q = nn.Linear(input_size, 128)
        self.v = nn.Sequential(nn.Dropout(dropout_v), nn.Linear(input_size, input_size))
        self.fcc = nn.Conv1d(output_class, output_class, kernel_size=input_size)     # unused, kept for checkpoints


class MILNet(nn.Module):
    def __init__(self, i_classifier, b_classifier):
        super().__init__()
        self.i_classifier = i_classifier
        self.b_classifier = b_classifier
        self.compute_dtype = torch.float32
        self.last_critical = None          # m [B,C]: arg-max patch per class of the latest call
        self.keep_mask_v = None            # tests: a [B,N,d] keep multiplier (0 or 1/keep) replacing the value branch's dropout draw

    def _run(self, x, want_max=False):
        """-> (classes [B,N,C], bag [B,C,d]); with ``want_max`` also the max-instance class scores [B,C] = classes.max(1)[0]
        (train_RLMIL.py:516) as a differentiable output of the same launches (what the training step uses)."""
        from .. import ops
        if x.dtype != self.compute_dtype:
            x = ops.cast(x.float().contiguous(), self.compute_dtype)
        fc, b = self.i_classifier.fc[0], self.b_classifier
        keep_v = self.keep_mask_v
        if keep_v is None and self.training and getattr(b, "dropout_v", 0.0) > 0.0:
            keep_v = ops.DropSeed(1.0 - b.dropout_v)                          # nn.Dropout(dropout_v) in front of v's Linear (dsmil.py:55-58)
        classes, bag, m, cmax = DSMILFn.apply(x.contiguous(), fc.weight, fc.bias, b.q.weight, b.q.bias, b.v[1].weight, b.v[1].bias,
                                              bool(want_max), keep_v)
        self.last_critical = m
        return (classes, bag, cmax) if want_max else (classes, bag)

    def forward(self, x):
        if isinstance(x, torch.Tensor) and x.dim() == 3 and x.shape[0] == 1:
            classes, bag = self._run(x)
            return classes[0], bag, bag.detach()
        if isinstance(x, torch.Tensor) and x.dim() == 3:
            classes, bag = self._run(x)
            return [classes[i] for i in range(x.shape[0])], bag, bag.detach()
        if isinstance(x, list):
            bags = [b if b.dim() == 3 else b.unsqueeze(0) for b in x]
            for b in bags:
                assert b.dim() == 3 and b.shape[0] == 1, f"feats.shape: {tuple(b.shape)}"      # dsmil.py:12
            if len({b.shape[1] for b in bags}) == 1:
                classes, bag = self._run(torch.cat(bags, 0))
                return [classes[i] for i in range(len(bags))], bag, bag.detach()
            outs = [self._run(b) for b in bags]                                                # ragged bags
            bag = torch.cat([o[1] for o in outs], 0)
            return [o[0][0] for o in outs], bag, bag.detach()
        raise TypeError


def build_dsmil(dim_feat, num_classes):
    i_classifier = FCLayer(in_size=dim_feat, out_size=num_classes)
    b_classifier = BClassifier(input_size=dim_feat, output_class=num_classes)
    net = MILNet(i_classifier, b_classifier)
    return net.cuda() if torch.cuda.is_available() else net
