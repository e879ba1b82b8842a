"""GPU k-means for the clustering pre-step (reference: wsi_processing/features_clustering.py).

``clustering(feats, num_clusters)`` / ``save_to_json(indices, num_clusters)`` keep the reference's signatures and file
formats (``features_cluster_indices`` ``[N,1]`` in an npz; a json list of ``num_clusters`` ascending patch-id lists - the
input of ``WSIWithCluster``).  The reference runs scikit-learn's KMeans(random_state=985) on the host, about a second
per slide; here Lloyd's iterations run on the HIP kernel (one pass over the slide's features per iteration, no float
atomics, so a run is reproducible bit for bit).  scikit-learn's random stream cannot be reproduced, so the partitions are
not the reference's label for label; what is pinned (tests) is that from the SAME initial centres the iterations are
scikit-learn's Lloyd iterations (same labels, centres and inertia), and that k-means++ seeding + restarts reach the
same inertia range.
"""
import json

import numpy as np
import torch

from .. import _lib, ops
from .._lib import check, ptr, stream


def _step(X, centers, labels, counts, stats, mind2, ws, update):
    N, d = X.shape
    check(_lib.lib().murcl_kmeans_step(ptr(X), N, d, centers.shape[0], ptr(centers), ptr(labels), ptr(counts), ptr(stats),
                                       ptr(mind2), int(update), ptr(ws), stream()), "kmeans_step")


def _relocate_empty(X, centers, labels, mind2, n):
    """scikit-learn's ``_relocate_empty_clusters_dense``: every cluster that received no row is moved onto one of the rows
    farthest from their own centres (largest first), and that row is taken out of the mean of the cluster it was in."""
    empty = np.nonzero(n == 0)[0]
    dist = mind2.cpu().numpy()
    far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]
    n = n.astype(np.float64).copy()
    for e, i in zip(empty, far):
        o = int(labels[i].item())
        x = X[int(i)]
        centers[o] = (centers[o] * n[o] - x) / max(n[o] - 1.0, 1.0)
        n[o] -= 1
        centers[e] = x
        n[e] = 1


def lloyd(X, centers, max_iter=300, tol=1e-4):
    """Lloyd's algorithm from the given centres, with scikit-learn's conventions (what the reference's KMeans call runs):
    stop when no label changes, or when the squared centre shift falls to ``tol * mean feature variance``, or after
    ``max_iter`` iterations; empty clusters are relocated; final labels / inertia are those of the final centres.
    X [N,d] f32 cuda (d in 256/512/1024), centers [K,d] (K <= 16).
    -> (labels int32 [N], centers [K,d], inertia float, iterations)."""
    if not X.is_cuda:
        raise RuntimeError("murcl_amd k-means runs on the GPU only (no CPU fallback)")
    X = X.float().contiguous()
    N, d = X.shape
    centers = centers.to(X.device, torch.float32).contiguous().clone()
    K = centers.shape[0]
    labels = torch.full((N,), -1, dtype=torch.int32, device=X.device)
    counts = torch.empty((K,), dtype=torch.int32, device=X.device)
    mind2 = torch.empty((N,), dtype=torch.float32, device=X.device)
    stats = torch.zeros((3 + K,), dtype=torch.float32, device=X.device)
    ws = torch.empty((_lib.lib().murcl_kmeans_workspace_bytes(N, d, K) + 3) // 4, dtype=torch.float32, device=X.device)
    thresh = tol * float(X.var(0, unbiased=False).mean())
    it = 0
    while it < max_iter:
        prev = centers.clone()
        _step(X, centers, labels, counts, stats, mind2, ws, True)
        it += 1
        h = stats.cpu().numpy()                       # one small copy per iteration steers the loop (an iteration is ~30 us)
        shift = float(h[0])
        if (h[3:] == 0).any():
            _relocate_empty(X, centers, labels, mind2, h[3:])
            shift = float(((centers - prev) ** 2).sum().item())
        if h[2] == 0 or shift <= thresh:              # no label moved (strict convergence), or the centres stopped moving
            break
    _step(X, centers, labels, counts, stats, mind2, ws, False)       # labels and inertia against the final centres
    return labels, centers, float(stats[1].item()), it


def kmeans_plusplus(X, K, generator):
    """Greedy k-means++ seeding as scikit-learn runs it (``_kmeans_plusplus``: 2 + log K candidate rows per step, drawn
    with probability ~ squared distance to the nearest chosen centre; the candidate that lowers the potential most wins)."""
    N, d = X.shape
    trials = 2 + int(np.log(K))
    if d % 32:                                                # the GEMM's reduction runs in 128-byte slabs: zero columns change no distance
        X = torch.nn.functional.pad(X, (0, 32 - d % 32)).contiguous()

    def cross(rows):
        """rows [r, d] . X^T -> [r, N] on the repo's own f32 GEMM (exact-f32 MFMA), not a library matmul."""
        return ops.gemm_nt(rows.contiguous(), X)

    xx = (X * X).sum(1)
    first = int(torch.randint(N, (1,), generator=generator, device=X.device).item())
    centers = [X[first, :d]]
    d2 = (xx - 2 * cross(X[first:first + 1]).view(-1) + xx[first]).clamp_min_(0)
    for _ in range(1, K):
        pot = d2.sum()
        if float(pot) <= 0:                                   # fewer distinct rows than clusters
            centers.append(X[int(torch.randint(N, (1,), generator=generator, device=X.device).item()), :d])   # (X may carry zero pad columns)
            continue
        cand = torch.multinomial(d2 / pot, trials, replacement=True, generator=generator)
        dc = (xx[None, :] - 2 * cross(X[cand]) + xx[cand][:, None]).clamp_min_(0)         # [trials, N]
        dc = torch.minimum(dc, d2[None, :])
        best = int(dc.sum(1).argmin().item())
        centers.append(X[cand[best], :d])
        d2 = dc[best]
    return torch.stack(centers, 0)


def kmeans(X, num_clusters, seed=985, n_init=3, max_iter=300, tol=1e-4):
    """Best of ``n_init`` k-means++ starts by inertia.  -> (labels int32 [N], centers, inertia)."""
    g = torch.Generator(device=X.device)
    g.manual_seed(seed)
    X = X.float().contiguous()
    best = None
    for _ in range(n_init):
        out = lloyd(X, kmeans_plusplus(X, num_clusters, g), max_iter, tol)
        if best is None or out[2] < best[2]:
            best = out
    return best[0], best[1], best[2]


def clustering(feats, num_clusters, filepath=None, device="cuda", seed=985):
    """features_clustering.py:10-16: -> ``features_cluster_indices`` int array [N,1]; saved to ``filepath`` (npz) if given."""
    X = torch.as_tensor(np.asarray(feats, dtype=np.float32)) if not isinstance(feats, torch.Tensor) else feats
    labels, _, _ = kmeans(X.to(device), num_clusters, seed=seed)
    idx = labels.cpu().numpy().astype(np.int64).reshape(-1, 1)
    if filepath is not None:
        np.savez(file=filepath, features_cluster_indices=idx)
    return idx


def save_to_json(features_cluster_indices, num_clusters, filepath=None):
    """features_clustering.py:19-25: ``num_clusters`` ascending patch-id lists (the ``clusters_json_filepath`` format)."""
    lab = np.asarray(features_cluster_indices).reshape(-1)
    lists = [np.nonzero(lab == k)[0].tolist() for k in range(num_clusters)]
    if filepath is not None:
        with open(filepath, "w") as f:
            json.dump(lists, f)
    return lists
