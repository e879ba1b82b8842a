"""Recognising tensors that are row blocks of one tensor, so that batched code can skip a concatenation (and its backward).

Two situations occur on the hot path:

* ``whole(xs)`` - autograd level: ``xs`` are the consecutive row blocks an ``x.split(B)`` handed out (``CL.forward`` for the
  aggregator output, ``Full_layer.forward_views`` for the projections).  Returns ``x`` itself when consuming it directly is
  equivalent to concatenating the blocks: same order, full cover, contiguous, and a non-leaf (or a tensor that needs no
  gradient) so that nobody expects ``.grad`` on the blocks' own leaf.
* ``as_one(views)`` - storage level, no autograd involved: equally shaped contiguous tensors that sit back to back in one
  buffer (the sub-bag builder writes all views into one allocation) are re-viewed as a single batch.
"""
import torch


def whole(xs):
    base = getattr(xs[0], "_base", None)
    if base is None or base.dim() != 2 or not base.is_contiguous() or base.shape[0] != sum(x.shape[0] for x in xs):
        return None
    if base.requires_grad and base.grad_fn is None:
        return None
    ptr, es = base.data_ptr(), base.element_size()
    for x in xs:
        if (getattr(x, "_base", None) is not base or x.requires_grad != base.requires_grad or x.dtype != base.dtype
                or x.dim() != 2 or not x.is_contiguous() or x.data_ptr() != ptr):
            return None
        ptr += x.numel() * es
    return base


def adjacent(views):
    v0 = views[0]
    n = v0.numel()
    return all(isinstance(v, torch.Tensor) and v.shape == v0.shape and v.is_contiguous() and not v.requires_grad
               and v.untyped_storage().data_ptr() == v0.untyped_storage().data_ptr()
               and v.storage_offset() == v0.storage_offset() + i * n for i, v in enumerate(views))


def as_one(views):
    """[V x (B, ...)] -> (V*B, ...): a re-view when the views are adjacent slices of one buffer, else one concatenation."""
    if adjacent(views):
        v0 = views[0]
        return torch.as_strided(v0, (len(views) * v0.shape[0],) + tuple(v0.shape[1:]), v0.stride(), v0.storage_offset())
    return torch.cat(list(views), 0)
