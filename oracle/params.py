"""Deterministic parameter sets keyed by the reference's state-dict names.

TEST INFRASTRUCTURE ONLY.  Shapes follow SURVEY.md section 8(a)/(b) (probed from the
reference's constructors).  Values come from oracle/detrand.py so goldens, CPU tests
and GPU tests all see identical bits without shipping weights.
"""
import math

import numpy as np

from . import detrand


def _linear(seed, name, out_f, in_f, d, gain=1.0, zero_bias=False):
    bound = gain / math.sqrt(in_f)
    d[name + ".weight"] = detrand.uniform(seed, name + ".weight", (out_f, in_f), -bound, bound)
    d[name + ".bias"] = (np.zeros(out_f, np.float32) if zero_bias else
                         detrand.uniform(seed, name + ".bias", (out_f,), -bound, bound))


def abmil(seed, dim_in=512, L=512, D=128, K=1, dim_out=128):
    """models/abmil.py:8-33 parameter names/shapes."""
    d = {}
    _linear(seed, "encoder.0", L, dim_in, d, gain=1.7)
    _linear(seed, "encoder.3", L, L, d, gain=1.7)
    _linear(seed, "encoder.6", L, L, d, gain=1.7)
    _linear(seed, "attention.0", D, L, d, gain=2.0)
    _linear(seed, "attention.2", K, D, d, gain=4.0)
    _linear(seed, "decoder.0", L, L, d, gain=60.0)   # undoes the 1/sqrt(N) of abmil.py:41 so heads see O(1) inputs
    _linear(seed, "fc", dim_out, L, d)
    return d


def clam_sb(seed, in_dim=512, size=(512, 256), n_classes=2):
    """models/clam.py:64-86 ('small': [in_dim,512,256])."""
    d = {}
    _linear(seed, "attention_net.0", size[0], in_dim, d, gain=1.7)
    _linear(seed, "attention_net.3.attention_a.0", size[1], size[0], d, gain=2.0)
    _linear(seed, "attention_net.3.attention_b.0", size[1], size[0], d, gain=2.0)
    _linear(seed, "attention_net.3.attention_c", 1, size[1], d, gain=6.0)
    _linear(seed, "classifiers", n_classes, size[0], d)
    for i in range(n_classes):
        _linear(seed, f"instance_classifiers.{i}", 2, size[0], d)
    return d


def clam_sb_plain(seed, in_dim=512, size=(512, 256), n_classes=2):
    """CLAM_SB(gate=False, dropout=True): Attn_Net keys (clam.py:18-34,80-81)."""
    d = {}
    _linear(seed, "attention_net.0", size[0], in_dim, d, gain=1.7)
    _linear(seed, "attention_net.3.module.0", size[1], size[0], d, gain=2.0)
    _linear(seed, "attention_net.3.module.3", 1, size[1], d, gain=6.0)
    _linear(seed, "classifiers", n_classes, size[0], d)
    for i in range(n_classes):
        _linear(seed, f"instance_classifiers.{i}", 2, size[0], d)
    return d


def dsmil(seed, dim_feat=512, num_classes=2):
    """models/dsmil.py:9,55-62,116-119."""
    d = {}
    _linear(seed, "i_classifier.fc.0", num_classes, dim_feat, d, gain=2.0)
    _linear(seed, "b_classifier.q", 128, dim_feat, d, gain=2.0)
    _linear(seed, "b_classifier.v.1", dim_feat, dim_feat, d)
    bound = 1.0 / math.sqrt(num_classes * dim_feat)
    d["b_classifier.fcc.weight"] = detrand.uniform(seed, "fcc.w", (num_classes, num_classes, dim_feat), -bound, bound)
    d["b_classifier.fcc.bias"] = detrand.uniform(seed, "fcc.b", (num_classes,), -bound, bound)
    return d


def _gru(seed, prefix, in_f, hid, d):
    bound = 1.0 / math.sqrt(hid)
    d[prefix + ".weight_ih_l0"] = detrand.uniform(seed, prefix + ".wih", (3 * hid, in_f), -3 * bound, 3 * bound)
    d[prefix + ".weight_hh_l0"] = detrand.uniform(seed, prefix + ".whh", (3 * hid, hid), -bound, bound)
    d[prefix + ".bias_ih_l0"] = detrand.uniform(seed, prefix + ".bih", (3 * hid,), -bound, bound)
    d[prefix + ".bias_hh_l0"] = detrand.uniform(seed, prefix + ".bhh", (3 * hid,), -bound, bound)


def full_layer(seed, feature_num=512, hidden=1024, class_num=128):
    """models/rlmil.py:198-200."""
    d = {}
    _gru(seed, "rnn", feature_num, hidden, d)
    _linear(seed, "fc", class_num, hidden, d)
    return d


def full_layer_cascade(seed, feature_num=512, class_num=16):
    """models/rlmil.py:201-206 (fc_rnn=False): the cascaded classifiers fc_2 .. fc_5."""
    d = {}
    for k in (2, 3, 4, 5):
        _linear(seed, f"fc_{k}", class_num, feature_num * k, d)
    return d


def actor_critic(seed, state_dim=512, hidden=512, action_size=10):
    """models/rlmil.py:40-54."""
    d = {}
    _linear(seed, "state_encoder.0", 2048, state_dim, d, gain=1.7)
    _linear(seed, "state_encoder.2", hidden, 2048, d, gain=1.7)
    _gru(seed, "gru", hidden, hidden, d)
    _linear(seed, "actor.0", action_size, hidden, d)
    _linear(seed, "critic.0", 1, hidden, d)
    return d


def to_torch(d):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def bags(seed, stream, B, N, d, nonneg=True):
    """Synthetic patch features: |N(0,1)|*0.5 (post-ReLU ResNet-like) or N(0,1)
    (SURVEY.md section 8(d))."""
    x = detrand.normal(seed, stream, (B, N, d))
    # per-slide feature signature so bags are distinguishable (i.i.d. bags would pool to
    # near-identical embeddings and make contrastive gradients vanish)
    sig = detrand.uniform(seed, stream + "/sig", (B, 1, d), 0.1, 1.9)
    x = x * sig
    return (np.abs(x) * 0.5).astype(np.float32) if nonneg else x.astype(np.float32)


def cluster_lists(seed, stream, n_patches, num_clusters):
    """Ascending patch-id lists per cluster (wsi_processing/features_clustering.py:19-25)."""
    lab = detrand.integers(seed, stream, (n_patches,), 0, num_clusters)
    return [np.nonzero(lab == k)[0].tolist() for k in range(num_clusters)]
