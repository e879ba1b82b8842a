"""Drop-in ``CL`` wrapper (reference: models/cl.py).

The reference runs the encoder once per view; here both views go through the kernels as one
2B-bag batch when their shapes agree, and are split again for the caller.
"""
import torch
from torch import nn

from ..utils.views import as_one


class CL(nn.Module):
    def __init__(self, encoder, projection_dim, n_features):
        super().__init__()
        self.encoder = encoder
        self.projection_dim = projection_dim      # stored, unused - as in the reference (cl.py:5-10)
        self.n_features = n_features
        self.last_whole = None                    # the latest call's views as one tensor [V*B, F] (batched path), else None

    def forward(self, x_views):
        assert isinstance(x_views, list), "CL expects a list of views"
        same = all(isinstance(v, torch.Tensor) and v.shape == x_views[0].shape for v in x_views)
        self.last_whole = None
        if same and x_views[0].dim() == 3:
            h = self.encoder(as_one(x_views))[0]
            h_views = list(h.split(x_views[0].shape[0], 0))
            self.last_whole = h                      # the views' rows as ONE tensor (a consumer of all of them needs no concatenation)
        else:
            h_views = [self.encoder(v)[0] for v in x_views]
        return h_views, [h.detach() for h in h_views]
