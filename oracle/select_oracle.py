"""CPU oracle: sub-bag construction (integer patch selection) and mix-up.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy float32 scalars and Python
integers; pure-Python loops (inputs are small in tests).  Citations relative to
/root/reference.
"""
import numpy as np


def window_bounds(n_patches, cluster_sizes, actions, feat_size):
    """Per-cluster slice bounds (utils/datasets.py:284-291).

    The reference computes, in float32 tensor arithmetic:
        size_j = round_half_even(float32(n_j) * float32(feat_size / N))   (:286-287)
        l_j    = floor(a_j * float32(n_j - size_j))                       (:290)
        r_j    = l_j + size_j                                             (:291)
    (``feat_size / N`` is a Python float64 quotient, narrowed to float32 when it
    multiplies the integer tensor).  Returns int arrays (l, r, size).
    """
    ratio = np.float32(feat_size / n_patches)
    n = np.asarray(cluster_sizes, dtype=np.int64)
    size = np.rint(n.astype(np.float32) * ratio).astype(np.int32)
    span = (n - size).astype(np.float32)
    left = np.floor(np.asarray(actions, dtype=np.float32) * span).astype(np.int32)
    return left, left + size, size


def select_indices(n_patches, clusters, actions, feat_size):
    """Sorted patch ids of one sub-bag before padding/truncation (datasets.py:292-296).

    ``clusters`` is a list of ascending id lists.  Python slice semantics are kept,
    including negative ``l`` (which happens whenever N < feat_size so size_j > n_j).
    """
    left, right, _ = window_bounds(n_patches, [len(c) for c in clusters], actions, feat_size)
    picked = []
    for c, l, r in zip(clusters, left.tolist(), right.tolist()):
        picked.extend(c[l:r])
    return sorted(picked)


def get_feats(feat_list, clusters_list, action_sequence, feat_size=1024):
    """utils/datasets.py:274-308 on numpy arrays.

    feat_list: list of [N_i,d] float arrays; returns ([B,feat_size,d], list of index lists
    after truncation to feat_size).
    """
    out, kept = [], []
    for feat, clusters, act in zip(feat_list, clusters_list, action_sequence):
        ids = select_indices(feat.shape[0], clusters, act, feat_size)[:feat_size]   # :304-305
        bag = np.zeros((feat_size, feat.shape[1]), dtype=feat.dtype)                # :300-303 zero pad
        bag[:len(ids)] = feat[ids]
        out.append(bag)
        kept.append(ids)
    return np.stack(out), kept


def mixup(x, lam, perm):
    """utils/datasets.py:263-271 with the random draws injected.

    out_i = lam_i * x_i + (1 - lam_i) * x_{perm[i]}, evaluated in the array's dtype like
    the reference's two products and one sum (:268-270)."""
    lam = np.asarray(lam, dtype=x.dtype).reshape(-1, 1, 1)
    return lam * x + (1 - lam) * x[np.asarray(perm)]
