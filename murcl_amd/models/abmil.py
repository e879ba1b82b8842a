"""Drop-in ``ABMIL`` (reference: models/abmil.py).

Same constructor, ``forward`` contract ``(out [B,L], out.detach())``, TypeError on foreign
inputs and state-dict keys (``encoder.{0,3,6}``, ``attention.{0,2}``, ``decoder.0``, ``fc``) as
the reference, so its checkpoints load unchanged.  The containers only *hold* parameters: a bag
batch runs as three MFMA GEMMs with fused bias+ReLU, one streaming attention-pool kernel
(scores, online soft-max, /sqrt(N), weighted sum in a single pass over H) and the decoder GEMM,
for all bags of a batch at once instead of the reference's per-bag Python loop (abmil.py:47-51).
"""
import torch
from torch import nn

from ..functional import ABMILFn, ABMILStepFn, attention_rows


def _stack(spec):
    return nn.Sequential(*[m for m in spec])


class ABMIL(nn.Module):
    def __init__(self, dim_in, L=512, D=128, K=1, dim_out=2, dropout=0.):
        super().__init__()
        self.L, self.D, self.K = L, D, K
        self.dropout = float(dropout)
        enc, width = [], dim_in
        for depth in range(3):                       # indices 0,3,6 are the Linears (abmil.py:12-21)
            enc += [nn.Linear(width, L), nn.ReLU()] + ([nn.Dropout(dropout)] if depth < 2 else [])
            width = L
        self.encoder = _stack(enc)
        self.attention = _stack([nn.Linear(L, D), nn.Tanh(), nn.Linear(D, K)])
        self.decoder = _stack([nn.Linear(L, L), nn.ReLU()])
        self.fc = nn.Linear(L, dim_out)              # built, never applied (abmil.py:33)
        self.compute_dtype = torch.float32           # torch.bfloat16 = throughput path
        self._att = None                             # most recent call: (att, stats) of ABMILFn per head (see last_attention)
        self.keep_masks = None                       # tests: (k1, k2) keep-multiplier tensors [B*N, L] replacing the dropout draws
        self.session = None                          # functional.EncoderSession of the training step in progress (deferred backward)

    # -- kernels -------------------------------------------------------------------------
    def _bags(self, x):
        """x [B,N,d] -> [B,L] through the fused HIP path."""
        # L = 512, D = 128 (every launch script's values) run the one-pass K2 pooling kernel and, in bf16, the weight-stationary
        # encoder; other --L / --D (train_RLMIL.py:91-97) and --dropout > 0 while training take the general path of ABMILFn
        drops = None
        if self.keep_masks is not None:                              # injected keep multipliers (parity tests)
            drops = self.keep_masks
        elif self.training and self.dropout > 0.0:                   # nn.Dropout(p) after encoder layers 1 and 2 (abmil.py:15,18)
            from .. import ops
            drops = (ops.DropSeed(1.0 - self.dropout), ops.DropSeed(1.0 - self.dropout))
        if x.dtype != self.compute_dtype:
            from .. import ops
            x = ops.cast(x.float().contiguous(), self.compute_dtype) if x.dtype != torch.float32 else \
                ops.cast(x.contiguous(), self.compute_dtype)
        e, a, d = self.encoder, self.attention, self.decoder
        if self.K != 1:
            return self._bags_heads(x.contiguous(), drops)
        if self.session is not None and drops is None and torch.is_grad_enabled():
            # a sequential training step keeps all its patch steps' activations in one set of buffers and runs ONE backward
            out, att, stats = ABMILStepFn.apply(x.contiguous(), e[0].weight, e[0].bias, e[3].weight, e[3].bias, e[6].weight, e[6].bias,
                                                a[0].weight, a[0].bias, a[2].weight, a[2].bias, d[0].weight, d[0].bias, self.session)
            self._att = [(att, stats)]
            return out
        out, att, stats = ABMILFn.apply(x.contiguous(), e[0].weight, e[0].bias, e[3].weight, e[3].bias, e[6].weight, e[6].bias,
                                        a[0].weight, a[0].bias, a[2].weight, a[2].bias, d[0].weight, d[0].bias, drops,
                                        torch.is_grad_enabled())
        self._att = [(att, stats)]
        return out

    @property
    def last_attention(self):
        """A [B,N] ([B,K,N] for K > 1 heads) of the most recent call, detached: softmax_N(scores)/sqrt(N) (abmil.py:38-41).  The
        training step never reads it, so the one-pass pooling kernel does not form it in the forward pass: it is computed from the
        call's raw scores and soft-max statistics on first access (one small launch) and kept.  Like the buffers of a sequential
        training step it is only valid until the next call."""
        if self._att is None:
            return None
        if not torch.is_tensor(self._att):
            rows = [attention_rows(att, stats) for att, stats in self._att]
            self._att = rows[0] if self.K == 1 else torch.stack(rows, 1)
        return self._att

    @last_attention.setter
    def last_attention(self, value):
        self._att = value

    def _bags_heads(self, x, drops):
        """K > 1 attention heads (abmil.py:8,23-27,38-44): ``attention.2`` has K rows, the soft-max runs over the patches of every
        head, ``torch.mm(A, H)`` is [K, L] per bag and the batch loop concatenates the blocks -> [B*K, L], bag-major.  No reference
        script sets K, so this branch is built for the contract, not for speed: head k is the single-head operator with row k of
        ``attention.2`` (the heads share everything else; autograd adds their parameter gradients), i.e. K encoder passes instead
        of one.  Seeded Dropout masks are the same in every pass (one draw per call, as the reference's single encoder pass)."""
        e, a, d = self.encoder, self.attention, self.decoder
        outs, atts = [], []
        for k in range(self.K):
            out, att, stats = ABMILFn.apply(x, e[0].weight, e[0].bias, e[3].weight, e[3].bias, e[6].weight, e[6].bias, a[0].weight,
                                            a[0].bias, a[2].weight[k:k + 1], a[2].bias[k:k + 1], d[0].weight, d[0].bias, drops,
                                            torch.is_grad_enabled())
            outs.append(out)
            atts.append((att, stats))
        self._att = atts                                               # -> [B, K, N] on access
        return torch.stack(outs, 1).reshape(x.shape[0] * self.K, -1)

    def bag_forward(self, bag):
        return self._bags(bag.unsqueeze(0))

    def batch_forward(self, batch):
        if isinstance(batch, torch.Tensor):
            return self._bags(batch)
        bags = [b.squeeze(0) if b.dim() == 3 else b for b in batch]
        if len({b.shape[0] for b in bags}) == 1:
            return self._bags(torch.stack(bags, 0))
        return torch.cat([self._bags(b.unsqueeze(0)) for b in bags], 0)      # ragged bags

    def forward(self, x):
        if isinstance(x, list):
            outputs = self.batch_forward(x)
        elif isinstance(x, torch.Tensor):
            outputs = self._bags(x if x.dim() == 3 else x.unsqueeze(0))
        else:
            raise TypeError
        return outputs, outputs.detach()
