"""Drop-in ``get_feats`` / ``mixup`` (reference: utils/datasets.py:263-308) on HIP kernels.

``get_feats`` keeps the reference signature (list of ``[1,N_i,d]`` tensors, nested Python cluster
lists, action tensor).  Internally the batch is packed once into a device-resident ``BagPack``
(features back to back, cluster lists in CSR form) and every call is two launches: an integer
selection kernel (bit-exact with the reference's float32 window arithmetic, no ``.item()`` syncs)
and a row gather.  ``subbag_views`` is the fused form the training loop uses: selection, gather and
mix-up for all views straight into one ``[V*B, feat_size, d]`` buffer.
"""
import numpy as np
import torch

from .. import _lib
from .._lib import check, dt, ptr, stream


class BagPack:
    """A batch of raw bags resident in HBM: rows of all bags back to back + CSR cluster lists."""

    def __init__(self, feats, row_off, n_patches, cluster_ids, cluster_off, num_clusters):
        self.feats, self.row_off, self.n_patches = feats, row_off, n_patches
        self.cluster_ids, self.cluster_off, self.K = cluster_ids, cluster_off, num_clusters
        self.B = int(n_patches.numel())
        self.n_host = n_patches.cpu().numpy().astype(np.int64)
        self._ratio = {}

    @classmethod
    def from_lists(cls, feat_list, clusters_list, dtype=None):
        dev = feat_list[0].device
        if dev.type != "cuda":
            raise RuntimeError("murcl_amd sub-bag kernels run on the GPU only (no CPU fallback)")
        rows = [f.reshape(-1, f.shape[-1]) for f in feat_list]
        feats = torch.cat(rows, 0)
        if dtype is not None and feats.dtype != dtype:
            feats = feats.to(dtype)
        n = np.array([r.shape[0] for r in rows], dtype=np.int64)
        row_off = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64)
        K = len(clusters_list[0])
        ids, off, base = [], np.zeros((len(rows), K + 1), dtype=np.int32), 0
        for b, cl in enumerate(clusters_list):
            if len(cl) != K:
                raise ValueError("every bag must have the same number of clusters")
            for j, c in enumerate(cl):
                off[b, j] = base
                ids.append(np.asarray(c, dtype=np.int32))
                base += len(c)
            off[b, K] = base
        ids = np.concatenate(ids) if ids else np.zeros(0, np.int32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        return cls(feats.contiguous(), t(row_off), t(n.astype(np.int32)), t(ids), t(off), K)

    def ratio(self, feat_size):
        """float32(feat_size / N_b): a double-precision quotient narrowed to f32 (datasets.py:285-287)."""
        if feat_size not in self._ratio:
            r = np.array([np.float32(feat_size / int(nb)) for nb in self.n_host], dtype=np.float32)
            self._ratio[feat_size] = torch.from_numpy(r).to(self.feats.device)
        return self._ratio[feat_size]


def select_indices(pack, action_sequence, feat_size):
    """-> idx [B, feat_size] int32 (ascending patch ids, -1 padded), count [B] int32."""
    a = action_sequence.to(torch.float32).contiguous()
    assert a.shape == (pack.B, pack.K), f"actions {tuple(a.shape)} vs {(pack.B, pack.K)}"
    dev = pack.feats.device
    idx = torch.empty((pack.B, feat_size), dtype=torch.int32, device=dev)
    cnt = torch.empty((pack.B,), dtype=torch.int32, device=dev)
    check(_lib.lib().murcl_subbag_select(ptr(pack.cluster_ids), ptr(pack.cluster_off), ptr(pack.n_patches),
                                         ptr(pack.ratio(feat_size)), ptr(a), pack.B, pack.K, feat_size,
                                         int(pack.n_host.max()), ptr(idx), ptr(cnt), stream()), "subbag_select")
    return idx, cnt


def _gather(pack, idx, feat_size, out, lam=None, perm=None):
    d = pack.feats.shape[1]
    check(_lib.lib().murcl_subbag_gather_mix(ptr(pack.feats), ptr(pack.row_off), ptr(idx), ptr(lam), ptr(perm),
                                             ptr(out), pack.B, feat_size, d, dt(pack.feats), dt(out), stream()),
          "subbag_gather_mix")
    return out


_PACK_CACHE = {}


def _pack_for(feat_list, clusters_list):
    key = (tuple(f.data_ptr() for f in feat_list), tuple(id(c) for c in clusters_list))
    p = _PACK_CACHE.get(key)
    if p is None:
        _PACK_CACHE.clear()                      # one live batch at a time, like the reference's feat_list
        p = _PACK_CACHE[key] = BagPack.from_lists(feat_list, clusters_list)
    return p


def get_feats(feat_list, clusters_list, action_sequence, feat_size=1024):
    """Construct the WSI-Fset: [B, feat_size, d] (zero padded), same contract as the reference."""
    pack = feat_list if isinstance(feat_list, BagPack) else _pack_for(feat_list, clusters_list)
    idx, _ = select_indices(pack, action_sequence, feat_size)
    out = torch.empty((pack.B, feat_size, pack.feats.shape[1]), dtype=pack.feats.dtype, device=pack.feats.device)
    return _gather(pack, idx, feat_size, out)


def mixup(inputs, alpha):
    """Mix-up a batch tensor -> (outputs, lambda_ [B,1], rand_idx [B]); same draws as the reference
    (one torch.rand, one torch.randperm on the input's device)."""
    if not inputs.is_cuda:
        raise RuntimeError("murcl_amd mixup runs on the GPU only (no CPU fallback)")
    B = inputs.shape[0]
    lambda_ = alpha + torch.rand(size=(B, 1), device=inputs.device) * (1 - alpha)
    rand_idx = torch.randperm(B, device=inputs.device)
    return mixup_with(inputs, lambda_, rand_idx), lambda_, rand_idx


def mixup_with(inputs, lambda_, rand_idx):
    x = inputs.contiguous()
    out = torch.empty_like(x)
    lam = lambda_.reshape(-1).to(torch.float32).contiguous()
    perm = rand_idx.to(torch.int32).contiguous()
    check(_lib.lib().murcl_mixup(ptr(x), ptr(lam), ptr(perm), ptr(out), x.shape[0], x[0].numel(), dt(x), stream()),
          "mixup")
    return out


def subbag_views(pack, action_sequences, feat_size, alpha=None, out_dtype=None, draws=None):
    """Fused K12+K13 for V views: returns (views: list of [B,feat_size,d] slices of ONE buffer, draws).

    ``draws`` = list of (lambda_ [B,1], rand_idx [B]) per view (generated like ``mixup`` when None and
    alpha is given; alpha None disables mix-up)."""
    V, dev = len(action_sequences), pack.feats.device
    d = pack.feats.shape[1]
    buf = torch.empty((V * pack.B, feat_size, d), dtype=out_dtype or pack.feats.dtype, device=dev)
    used = []
    for v, a in enumerate(action_sequences):
        idx, _ = select_indices(pack, a, feat_size)
        lam = perm = None
        if alpha is not None or draws is not None:
            if draws is not None:
                lambda_, rand_idx = draws[v]
            else:
                lambda_ = alpha + torch.rand(size=(pack.B, 1), device=dev) * (1 - alpha)
                rand_idx = torch.randperm(pack.B, device=dev)
            used.append((lambda_, rand_idx))
            lam = lambda_.reshape(-1).to(torch.float32).contiguous()
            perm = rand_idx.to(torch.int32).contiguous()
        _gather(pack, idx, feat_size, buf[v * pack.B:(v + 1) * pack.B], lam, perm)
    return [buf[v * pack.B:(v + 1) * pack.B] for v in range(V)], used
