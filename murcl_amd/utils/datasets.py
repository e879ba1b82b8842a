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

    def __init__(self, feats, row_off, n_patches, cluster_ids, cluster_off, num_clusters, n_host=None):
        self.feats, self.row_off, self.n_patches = feats, row_off, n_patches
        self.cluster_ids, self.cluster_off, self.K = cluster_ids, cluster_off, num_clusters
        self.B = int(n_patches.numel())
        self.n_host = (n_patches.cpu().numpy() if n_host is None else np.asarray(n_host)).astype(np.int64)
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


class DeviceSlideStore:
    """A whole dataset split resident in HBM (reference reader: utils/datasets.py:115-165).

    The reference re-reads each slide's ``img_features`` npz and ships the raw bag host -> device on every step
    (train_MuRCL.py:224-227).  Here every slide of this rank's shard is uploaded ONCE into one arena - feature rows back
    to back in the compute dtype, cluster id lists in CSR form - and a step's batch is ``store.pack(slide_indices)``: a
    ``BagPack`` whose per-bag row / cluster offsets point into the arena.  No feature byte moves when a batch is formed;
    the selection and gather kernels (K12/K13) read the arena directly.  C4's 512 slides x 8192 x 512 are 4.3 GB in
    bf16, a 10 000-slide cohort at 20 000 patches about 205 GB: it fits one MI355X (288 GB)."""

    def __init__(self, feats, row_off, n_patches, cluster_ids, cluster_off, num_clusters, labels=None, case_ids=None):
        self.feats, self.K = feats, num_clusters
        self.row_off_h, self.n_h = np.asarray(row_off, np.int64), np.asarray(n_patches, np.int64)
        self.cluster_off_h = np.asarray(cluster_off, np.int32)
        dev = feats.device
        self.row_off = torch.from_numpy(self.row_off_h).to(dev)
        self.n_patches = torch.from_numpy(self.n_h.astype(np.int32)).to(dev)
        self.cluster_ids = cluster_ids
        self.cluster_off = torch.from_numpy(self.cluster_off_h).to(dev)
        self.labels = None if labels is None else np.asarray(labels, np.int64)
        self.case_ids = case_ids
        self.patch_dim = feats.shape[1]

    def __len__(self):
        return len(self.n_h)

    @classmethod
    def from_dataset(cls, dataset, device, dtype=torch.float32, indices=None, chunk_rows=1 << 20):
        """``dataset[i] -> (feat [N_i,d] or [1,N_i,d], K ascending id lists, label, case_id)`` (``WSIWithCluster`` /
        ``SyntheticWSI`` order).  ``indices``: the slides of this rank's shard (default: all)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DeviceSlideStore lives in GPU memory (no CPU fallback)")
        indices = list(range(len(dataset))) if indices is None else list(indices)
        items = [dataset[i] for i in indices]
        n = np.array([int(np.prod(it[0].shape[:-1])) for it in items], dtype=np.int64)
        d = int(items[0][0].shape[-1])
        K = len(items[0][1])
        row_off = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64)
        feats = torch.empty((int(n.sum()), d), dtype=dtype, device=device)
        for it, o, k in zip(items, row_off, n):                       # one upload per slide, converted on the device
            f = torch.as_tensor(it[0]).reshape(-1, d)
            for r in range(0, int(k), chunk_rows):
                feats[o + r:o + min(r + chunk_rows, int(k))].copy_(f[r:r + chunk_rows].to(device, non_blocking=True))
        ids, off, base = [], np.zeros((len(items), K + 1), dtype=np.int64), 0
        for b, it in enumerate(items):
            if len(it[1]) != K:
                raise ValueError("every slide must have the same number of clusters")
            for j, c in enumerate(it[1]):
                off[b, j] = base
                ids.append(np.asarray(c, dtype=np.int32))
                base += len(c)
            off[b, K] = base
        if base >= 2 ** 31:
            raise ValueError("cluster id lists exceed the int32 CSR offsets of the selection kernel; shard the split")
        ids = np.concatenate(ids) if ids else np.zeros(0, np.int32)
        return cls(feats, row_off, n, torch.from_numpy(np.ascontiguousarray(ids)).to(device), off.astype(np.int32), K,
                   labels=[int(it[2]) for it in items], case_ids=[it[3] for it in items])

    def pack(self, slide_indices):
        """BagPack of the given slides (any order, repeats allowed) - index arithmetic only, no feature copy."""
        sel = np.asarray(slide_indices, dtype=np.int64)
        t = torch.from_numpy(sel).to(self.feats.device)
        return BagPack(self.feats, self.row_off.index_select(0, t), self.n_patches.index_select(0, t), self.cluster_ids,
                       self.cluster_off.index_select(0, t), self.K, n_host=self.n_h[sel])

    def bytes(self):
        return self.feats.numel() * self.feats.element_size() + self.cluster_ids.numel() * 4


def select_indices(pack, action_sequence, feat_size):
    """-> idx [B, feat_size] int32 (ascending patch ids, -1 padded), count [B] int32.  action_sequence [B,K]; or [V,B,K] for V
    sub-bags of the same bags in one launch -> idx [V,B,feat_size], count [V,B]."""
    a = action_sequence.to(torch.float32).contiguous()
    V = a.shape[0] if a.dim() == 3 else 1
    assert tuple(a.shape[-2:]) == (pack.B, pack.K), f"actions {tuple(a.shape)} vs {(pack.B, pack.K)}"
    dev = pack.feats.device
    idx = torch.empty((V * pack.B, feat_size), dtype=torch.int32, device=dev)
    cnt = torch.empty((V * pack.B,), dtype=torch.int32, device=dev)
    check(_lib.lib().murcl_subbag_select(ptr(pack.cluster_ids), ptr(pack.cluster_off), ptr(pack.n_patches),
                                         ptr(pack.ratio(feat_size)), ptr(a), V, pack.B, pack.K, feat_size,
                                         int(pack.n_host.max()), ptr(idx), ptr(cnt), stream()), "subbag_select")
    if a.dim() == 3:
        return idx.view(V, pack.B, feat_size), cnt.view(V, pack.B)
    return idx, cnt


def _gather(pack, idx, feat_size, out, lam=None, perm=None, views=1, local=None):
    """idx [views*B, feat_size], lam / perm [views*B] (perm: partner bag 0..B-1 inside the view), out [views*B, feat_size, d].
    ``local`` = (lo, n): only the bags [lo, lo + n) of every view are written - out [views*n, feat_size, d] - while idx / lam / perm
    cover all B bags (a sharded step whose mix-up partners come from the whole global batch)."""
    d = pack.feats.shape[1]
    lo, n = (0, pack.B) if local is None else local
    check(_lib.lib().murcl_subbag_gather_mix_rows(ptr(pack.feats), ptr(pack.row_off), ptr(idx), ptr(lam), ptr(perm),
                                                  ptr(out), views, pack.B, feat_size, d, int(lo), int(n), dt(pack.feats), dt(out),
                                                  stream()), "subbag_gather_mix")
    return out


_PACK_CACHE = []          # [(feat tensors, their versions, cluster lists, pack)] - at most one entry


def _pack_for(feat_list, clusters_list):
    """The BagPack of the reference-style (feat_list, clusters_list) pair, reused while a step calls ``get_feats`` on the
    SAME objects (T x 2 times per batch).  The entry holds strong references to the keyed tensors and lists and is matched
    by identity (+ tensor versions): addresses or ids recycled by the allocator for the next batch cannot alias it."""
    if _PACK_CACHE:
        feats, vers, cls, pack = _PACK_CACHE[0]
        if (len(feats) == len(feat_list) and len(cls) == len(clusters_list)
                and all(a is b for a, b in zip(feats, feat_list)) and all(a is b for a, b in zip(cls, clusters_list))
                and vers == [f._version for f in feat_list]):
            return pack
        _PACK_CACHE.clear()                      # one live batch at a time, like the reference's feat_list
    pack = BagPack.from_lists(feat_list, clusters_list)
    _PACK_CACHE.append((list(feat_list), [f._version for f in feat_list], list(clusters_list), pack))
    return pack


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


def draw_mixups(n_views, B, alpha, device):
    """``n_views`` independent mix-up draws (datasets.py:265-267: lambda ~ alpha + U(0,1)(1-alpha) per bag, a uniform random
    permutation of the bags) for ALL the views of a step - one launch on the GPU (``draw_step``); elsewhere one uniform tensor,
    lambda by two elementwise ops, the permutations as the row-wise argsort of i.i.d. uniforms (uniform over permutations; an exact
    tie, ~B^2/2^25, still yields a valid permutation).  View by view the same draws are ``torch.rand`` + ``torch.randperm``: ten tiny
    launches each, 120 per sampler-in-the-loop step at T = 6.  Returns [(lambda [B,1] f32, perm [B] int32)] * n_views."""
    return draw_step(device, None, None, n_views, B, alpha)[2]


def draw_step(device, uni_shape, nrm_shape, n_views, B, alpha, seed=None):
    """Every random draw of one training step: (u ~ U[0,1) of ``uni_shape`` or None - the window positions, train_MuRCL.py:235,256-258;
    n ~ N(0,1) of ``nrm_shape`` or None - the sampler's noise, rlmil.py:85-86; the ``draw_mixups`` of ``n_views`` views).  On the GPU
    ONE launch (``ops.step_draws``: counter-based generator seeded from torch's global seed) instead of ~12 (two ``torch.rand``, two
    elementwise ops, an argsort - radix sort, arange, copies - and ``torch.randn``); elsewhere the same distributions from torch ops."""
    import math
    nu = 0 if uni_shape is None else math.prod(uni_shape)
    nn_ = 0 if nrm_shape is None else math.prod(nrm_shape)
    if torch.device(device).type == "cuda":
        from .. import ops
        if B <= ops.DRAWS_MAX_B:
            uni, nrm, lam, perm = ops.step_draws(torch.device(device), nu, nn_, n_views, B, alpha, seed=seed)
            return (None if uni_shape is None else uni.view(uni_shape), None if nrm_shape is None else nrm.view(nrm_shape),
                    MixDraws(lam.unsqueeze(-1), perm))
    if seed is not None:
        raise NotImplementedError("a seeded draw_step needs the counter-based launch (GPU, B <= ops.DRAWS_MAX_B)")
    u = torch.rand((2, n_views, B), device=device)
    lam = u[0].mul(1 - alpha).add_(alpha).unsqueeze(-1)
    perm = u[1].argsort(dim=1).to(torch.int32)
    return (None if uni_shape is None else torch.rand(uni_shape, device=device),
            None if nrm_shape is None else torch.randn(nrm_shape, device=device), MixDraws(lam, perm))


class MixDraws(list):
    """[(lambda [B,1], perm [B])] per view, as slices of whole tensors ``lam`` [n,B,1] f32 / ``perm`` [n,B] int32 that
    ``subbag_views`` hands to ONE gather launch; a slice of it is again a MixDraws over the corresponding rows."""

    def __init__(self, lam, perm):
        super().__init__((lam[v], perm[v]) for v in range(lam.shape[0]))
        self.lam, self.perm = lam, perm

    def __getitem__(self, i):
        if isinstance(i, slice):
            return MixDraws(self.lam[i], self.perm[i])
        return super().__getitem__(i)


def subbag_views(pack, action_sequences, feat_size, alpha=None, out_dtype=None, draws=None, out=None, local=None):
    """Fused K12+K13 for V views: returns (views: list of [B,feat_size,d] slices of ONE buffer, draws).

    ``draws`` = list of (lambda_ [B,1], rand_idx [B]) per view (generated like ``mixup`` when None and
    alpha is given; alpha None disables mix-up).
    ``local`` = (lo, n) (round 6, batch-global mix-up under data parallelism): ``pack`` / actions / draws describe the GLOBAL batch
    of B bags - the partner ``rand_idx[b]`` is any of them, as datasets.py:267-269 permutes over the whole batch - and only the views
    of this rank's bags [lo, lo + n) are built: -> V views of [n, feat_size, d]."""
    dev, B = pack.feats.device, pack.B
    lo_n = (0, B) if local is None else (int(local[0]), int(local[1]))
    nB = lo_n[1]
    d = pack.feats.shape[1]
    # all V views through ONE selection launch and ONE gather (+ mix-up) launch: the views differ only in their rows of
    # actions / lambda / perm (12 views per stage-1 step: 24 launches before)
    if torch.is_tensor(action_sequences):
        acts, V = action_sequences, action_sequences.shape[0]
    else:
        V = len(action_sequences)
        if V == 1:
            acts = action_sequences[0].unsqueeze(0)
        elif all(a.is_cuda and a.dtype == torch.float32 for a in action_sequences):
            from .. import ops                       # the sampler's per-view rows are slices of one step output: a view, no copy
            acts = ops.stack_rows([a.reshape(1, -1) for a in action_sequences])
        else:
            acts = torch.stack([a.to(torch.float32) for a in action_sequences], 0)
    if out is None:
        buf = torch.empty((V * nB, feat_size, d), dtype=out_dtype or pack.feats.dtype, device=dev)
    else:                                   # a caller-owned [V*B, feat_size, d] block (functional.EncoderSession keeps all patch steps' views)
        buf = out
        assert buf.is_contiguous() and tuple(buf.shape) == (V * nB, feat_size, d) and buf.dtype == (out_dtype or pack.feats.dtype)
    idx, _ = select_indices(pack, acts.reshape(V, B, pack.K), feat_size)
    used, lam, perm = [], None, None
    if alpha is not None or draws is not None:
        if draws is None:
            draws = draw_mixups(V, B, alpha, dev)
        if isinstance(draws, MixDraws) and draws.lam.shape[0] == V and draws.lam.is_contiguous() and draws.perm.is_contiguous():
            lam, perm = draws.lam.reshape(-1), draws.perm.reshape(-1)
        else:                                                       # separate tensors per view (injected draws of the parity tests)
            lam = torch.stack([dr[0].reshape(-1).to(torch.float32) for dr in draws[:V]], 0).reshape(-1)
            perm = torch.stack([dr[1].to(torch.int32) for dr in draws[:V]], 0).reshape(-1)
        used = [(draws[v][0], draws[v][1]) for v in range(V)]
        lam, perm = lam.to(torch.float32).contiguous(), perm.to(torch.int32).contiguous()
    _gather(pack, idx.view(V * B, feat_size), feat_size, buf, lam, perm, views=V, local=local)
    return [buf[v * nB:(v + 1) * nB] for v in range(V)], used
