"""Drop-in ``CL`` wrapper (reference: models/cl.py).

The reference runs the encoder once per view; here both views go through the kernels as one
2B-bag batch when their shapes agree, and are split again for the caller.
"""
import torch
from torch import nn


def _as_one_batch(views):
    """[V x (B,N,d)] -> (V*B,N,d).  Zero-copy when the views are back-to-back slices of one buffer (which
    is how the sub-bag builder lays them out); otherwise one concatenation."""
    v0 = views[0]
    n = v0.numel()
    adjacent = all(v.is_contiguous() and not v.requires_grad
                   and v.untyped_storage().data_ptr() == v0.untyped_storage().data_ptr()
                   and v.storage_offset() == v0.storage_offset() + i * n for i, v in enumerate(views))
    if adjacent:
        return torch.as_strided(v0, (len(views) * v0.shape[0],) + tuple(v0.shape[1:]), v0.stride(), v0.storage_offset())
    return torch.cat(views, 0)


class CL(nn.Module):
    def __init__(self, encoder, projection_dim, n_features):
        super().__init__()
        self.encoder = encoder
        self.projection_dim = projection_dim      # stored, unused - as in the reference (cl.py:5-10)
        self.n_features = n_features

    def forward(self, x_views):
        assert isinstance(x_views, list), "CL expects a list of views"
        same = all(isinstance(v, torch.Tensor) and v.shape == x_views[0].shape for v in x_views)
        if same and x_views[0].dim() == 3:
            h = self.encoder(_as_one_batch(x_views))[0]
            h_views = list(h.split(x_views[0].shape[0], 0))
        else:
            h_views = [self.encoder(v)[0] for v in x_views]
        return h_views, [h.detach() for h in h_views]
