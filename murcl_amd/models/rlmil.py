"""Drop-in recurrent head ``Full_layer`` (reference: models/rlmil.py:187-239).

``fc_rnn=True`` (the only mode the training scripts use): one GRU time step per call with the
hidden state carried in ``self.hidden`` across calls - including the reference's behaviour
that the two views share that single attribute - followed by the class/projection Linear.
State-dict keys: ``rnn.{weight,bias}_{ih,hh}_l0``, ``fc.{weight,bias}``.
"""
import torch
from torch import nn

from ..functional import GRUStepFn, LinearFn


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

    def forward(self, x, restart=False):
        if self.fc_rnn:
            h_prev = None if restart else self.hidden[0]
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
