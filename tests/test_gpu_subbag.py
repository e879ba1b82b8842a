"""K12/K13 on MI355X: patch selection bit-exact vs the reference goldens (G6) and the oracle; mix-up vs G7."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, params as P, select_oracle as S  # noqa: E402
from tests.test_oracle_goldens import _g6_rebuild  # noqa: E402

T = torch.from_numpy


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _marker_feats(N, d=8):
    f = np.zeros((N, d), np.float32)
    f[:, 0] = np.arange(N) + 1
    f[:, 1] = 7.0
    return f


@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e"])
def test_get_feats_indices_bit_exact_vs_reference_golden(golden, name):
    from murcl_amd.utils.datasets import get_feats
    g = golden("g6_get_feats")
    N, fs, act = int(g[f"{name}.N"]), int(g[f"{name}.fs"]), g[f"{name}.act"]
    cls = _g6_rebuild(name)
    dev = _dev()
    feats = [T(_marker_feats(N)).unsqueeze(0).to(dev) for _ in cls]
    out = get_feats(feats, cls, T(act).to(dev), fs)
    assert out.shape == (len(cls), fs, 8)
    np.testing.assert_array_equal(out[:, :, 0].cpu().numpy().astype(np.int64), g[f"{name}.ids_plus1"])
    pad = out[:, :, 0] == 0
    assert (out[:, :, 1][pad] == 0).all() and (out[:, :, 1][~pad] == 7.0).all()


@pytest.mark.parametrize("B,N,K,fs", [(8, 8192, 10, 1024), (3, 300, 4, 512), (5, 20000, 10, 1024), (2, 64, 3, 16)])
def test_select_matches_oracle_random(B, N, K, fs):
    """Random bags incl. ragged lengths and N < feat_size; ids must be identical (integers: no tolerance)."""
    from murcl_amd.utils.datasets import BagPack, select_indices
    dev = _dev()
    Ns = [N - 37 * b for b in range(B)]
    cls = [P.cluster_lists(31, f"c{b}", Ns[b], K) for b in range(B)]
    feats = [T(_marker_feats(n)).to(dev) for n in Ns]
    act = detrand.uniform(31, "act", (B, K))
    act[0, 0], act[0, 1] = 0.0, 1.0
    pack = BagPack.from_lists(feats, cls)
    idx, cnt = select_indices(pack, T(act).to(dev), fs)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for b in range(B):
        want = S.select_indices(Ns[b], cls[b], act[b], fs)[:fs]
        assert cnt[b] == len(want)
        np.testing.assert_array_equal(idx[b, :len(want)], np.asarray(want, dtype=np.int64))
        assert (idx[b, len(want):] == -1).all()


def test_mixup_golden(golden):
    from murcl_amd.utils.datasets import mixup_with, mixup
    g = golden("g7_mixup")
    dev = _dev()
    x = T(detrand.normal(9, "g7.x", (5, 16, 8))).to(dev)
    out = mixup_with(x, T(g["lam"]).to(dev), T(g["perm"]).to(dev))
    np.testing.assert_array_equal(out.cpu().numpy(), g["out"])            # same two products + one sum: bit-exact
    o2, lam, perm = mixup(x, 0.9)
    assert lam.shape == (5, 1) and (lam >= 0.9).all() and (lam <= 1).all()
    assert sorted(perm.cpu().tolist()) == list(range(5))
    np.testing.assert_array_equal(o2.cpu().numpy(), S.mixup(x.cpu().numpy(), lam.cpu().numpy(), perm.cpu().numpy()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_views_equal_get_feats_then_mixup(dtype):
    from murcl_amd.utils.datasets import BagPack, subbag_views
    dev = _dev()
    B, N, K, fs, d = 6, 3000, 10, 256, 512
    feats_np = [P.bags(41, f"f{b}", 1, N - 100 * b, d)[0] for b in range(B)]
    cls = [P.cluster_lists(41, f"c{b}", N - 100 * b, K) for b in range(B)]
    pack = BagPack.from_lists([T(f).to(dev) for f in feats_np], cls)
    acts = [detrand.uniform(41, f"a{v}", (B, K)) for v in range(2)]
    draws = [(T(detrand.uniform(41, f"l{v}", (B, 1), 0.9, 1.0)).to(dev), T(detrand.permutation(41, f"p{v}", B)).to(dev)) for v in range(2)]
    views, used = subbag_views(pack, [T(a).to(dev) for a in acts], fs, draws=draws, out_dtype=dtype)
    assert views[1].data_ptr() == views[0].data_ptr() + views[0].numel() * views[0].element_size()   # back to back
    for v in range(2):
        sub, _ = S.get_feats(feats_np, cls, acts[v], fs)
        want = S.mixup(sub, draws[v][0].cpu().numpy(), draws[v][1].cpu().numpy())
        got = views[v].float().cpu().numpy()
        if dtype == torch.float32:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=8e-3, atol=1e-6)


class _ListSet:
    """dataset[i] -> (feat, clusters, label, case_id), the WSIWithCluster item format."""

    def __init__(self, feats, cls):
        self.feats, self.cls = feats, cls

    def __len__(self):
        return len(self.feats)

    def __getitem__(self, i):
        return T(self.feats[i]).unsqueeze(0), self.cls[i], i % 2, f"case{i}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_device_slide_store_packs_equal_per_step_packs(dtype):
    """a batch drawn from the HBM-resident split selects the same patch ids (bit-exact) and gathers
    the same rows as a BagPack built from those slides' host tensors - any order, repeats, ragged slide lengths, N < feat_size."""
    from murcl_amd.utils.datasets import BagPack, DeviceSlideStore, select_indices, subbag_views
    dev = _dev()
    S_, K, fs, d = 9, 10, 256, 64
    Ns = [3000 - 311 * b for b in range(S_ - 1)] + [100]                   # last slide shorter than feat_size
    feats_np = [P.bags(51, f"f{b}", 1, Ns[b], d)[0] for b in range(S_)]
    cls = [P.cluster_lists(51, f"c{b}", Ns[b], K) for b in range(S_)]
    store = DeviceSlideStore.from_dataset(_ListSet(feats_np, cls), dev, dtype=dtype, chunk_rows=1000)
    assert len(store) == S_ and store.K == K and store.patch_dim == d and store.feats.shape[0] == sum(Ns)
    assert store.labels.tolist() == [i % 2 for i in range(S_)] and store.case_ids[3] == "case3"
    assert store.bytes() == sum(Ns) * d * (2 if dtype == torch.bfloat16 else 4) + sum(Ns) * 4
    for pick in ([8, 0, 3, 3, 7], list(range(S_)), [5]):
        ref = BagPack.from_lists([T(feats_np[i]).to(dev) for i in pick], [cls[i] for i in pick], dtype=dtype)
        got = store.pack(pick)
        assert got.feats.data_ptr() == store.feats.data_ptr()               # the arena itself: nothing was copied
        B = len(pick)
        acts = [T(detrand.uniform(51, f"a{v}{B}", (B, K))).to(dev) for v in range(2)]
        for a in acts:
            i0, c0 = select_indices(ref, a, fs)
            i1, c1 = select_indices(got, a, fs)
            assert torch.equal(i0, i1) and torch.equal(c0, c1)
            for b, s in enumerate(pick):                                    # and against the oracle
                want = S.select_indices(Ns[s], cls[s], a[b].cpu().numpy(), fs)[:fs]
                assert c1[b].item() == len(want) and i1[b, :len(want)].cpu().tolist() == list(want)
        draws = [(T(detrand.uniform(51, f"l{v}{B}", (B, 1), 0.9, 1.0)).to(dev), T(detrand.permutation(51, f"p{v}{B}", B)).to(dev))
                 for v in range(2)]
        v0, _ = subbag_views(ref, acts, fs, draws=draws)
        v1, _ = subbag_views(got, acts, fs, draws=draws)
        for x, y in zip(v0, v1):
            assert torch.equal(x, y)


def test_device_slide_store_refuses_cpu():
    from murcl_amd.utils.datasets import DeviceSlideStore
    with pytest.raises(RuntimeError):
        DeviceSlideStore.from_dataset(_ListSet([np.zeros((4, 8), np.float32)], [[[0, 1], [2, 3]]]), "cpu")


def test_get_feats_with_empty_and_singleton_clusters():
    """A cluster may own no patch at all (k-means left it empty for this slide) or a single one: the window arithmetic
    (size rounds to 0, slices of empty lists) must agree with the reference semantics bit for bit."""
    from murcl_amd.utils.datasets import BagPack, select_indices
    dev = _dev()
    N, K, fs = 500, 6, 64
    base = P.cluster_lists(33, "c", N, K)
    merged = sorted(base[1] + base[2])
    cases = [
        [base[0], [], merged, base[3], base[4], base[5]],                # an empty cluster in the middle
        [[], [], sorted(sum(base, [])), [], [], []],                     # everything in one cluster
        [[0], [1], [2], sorted(set(range(N)) - {0, 1, 2, 499}), [499], []],   # singletons at both ends
    ]
    feats = [T(_marker_feats(N)).to(dev) for _ in cases]
    act = detrand.uniform(33, "act", (len(cases), K))
    act[1, 2], act[2, 0], act[2, 4] = 1.0, 0.0, 1.0
    pack = BagPack.from_lists(feats, cases)
    idx, cnt = select_indices(pack, T(act).to(dev), fs)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for b, cl in enumerate(cases):
        want = S.select_indices(N, cl, act[b], fs)[:fs]
        assert cnt[b] == len(want), (b, cnt[b], len(want))
        np.testing.assert_array_equal(idx[b, :len(want)], np.asarray(want, dtype=np.int64))
        assert (idx[b, len(want):] == -1).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_subbag_views_of_a_shard_with_global_partners_equal_the_rows_of_the_global_call(dtype):
    """``subbag_views(local=(lo, n))`` (round 6, --global_mixup): actions, lambdas and permutations describe the GLOBAL batch - the
    partner of a rank's bag may be any of its bags (utils/datasets.py:267-269) - and only the rank's bags are built: for every
    shard of a 12-bag batch the V views are bit-for-bit the corresponding rows of the one global call, into caller-owned buffers too."""
    from murcl_amd.utils.datasets import BagPack, subbag_views
    dev = _dev()
    B, K, fs, V = 12, 5, 128, 3
    Ns = [700 + 41 * b for b in range(B)]
    cls = [P.cluster_lists(61, f"c{b}", Ns[b], K) for b in range(B)]
    feats = [T(P.bags(61, f"f{b}", 1, Ns[b], 64)[0]).to(dev) for b in range(B)]
    pack = BagPack.from_lists(feats, cls, dtype=dtype)
    acts = T(detrand.uniform(61, "act", (V, B, K)).astype(np.float32)).to(dev)
    draws = [(T(detrand.uniform(61, f"l{v}", (B, 1), 0.5, 1.0).astype(np.float32)).to(dev),
              T(np.asarray(detrand.permutation(61, f"p{v}", B)).astype(np.int32)).to(dev)) for v in range(V)]
    whole, _ = subbag_views(pack, acts, fs, alpha=0.9, draws=draws)
    for lo, n in ((0, 4), (4, 4), (8, 4), (0, 12), (5, 1), (9, 3)):
        part, _ = subbag_views(pack, acts, fs, alpha=0.9, draws=draws, local=(lo, n))
        assert len(part) == V and all(tuple(p.shape) == (n, fs, 64) for p in part)
        for v in range(V):
            assert torch.equal(part[v], whole[v][lo:lo + n]), (lo, n, v)
    buf = torch.empty((V * 4, fs, 64), dtype=dtype, device=dev)
    part, _ = subbag_views(pack, acts, fs, alpha=0.9, draws=draws, out=buf, local=(4, 4))
    assert part[0].data_ptr() == buf.data_ptr() and torch.equal(part[2], whole[2][4:8])


def test_seeded_step_draws_repeat_and_unseeded_ones_do_not():
    """``draw_step(seed=s)`` (the shared stream of --global_mixup: every rank calls it with the same seed and needs no collective):
    the same seed gives the same window positions, lambdas and permutations bit for bit, another seed or no seed gives others; the
    permutations are permutations and lambda lies in [alpha, 1)."""
    from murcl_amd.utils.datasets import draw_step
    dev = _dev()
    a = draw_step(dev, (2, 2, 12, 5), None, 4, 12, 0.9, seed=1234567)
    b = draw_step(dev, (2, 2, 12, 5), None, 4, 12, 0.9, seed=1234567)
    c = draw_step(dev, (2, 2, 12, 5), None, 4, 12, 0.9, seed=1234568)
    d = draw_step(dev, (2, 2, 12, 5), None, 4, 12, 0.9)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2].lam, b[2].lam) and torch.equal(a[2].perm, b[2].perm) and a[1] is None
    assert not torch.equal(a[0], c[0]) and not torch.equal(a[0], d[0]) and not torch.equal(a[2].lam, c[2].lam)
    assert (a[2].perm.sort(1)[0] == torch.arange(12, device=dev, dtype=torch.int32)).all()
    assert (a[2].lam >= 0.9).all() and (a[2].lam < 1.0).all()
    only_noise = draw_step(dev, None, (2, 2, 3, 5), 0, 0, 0.9)                       # a rank's own sampler noise under --global_mixup
    assert only_noise[0] is None and tuple(only_noise[1].shape) == (2, 2, 3, 5) and torch.isfinite(only_noise[1]).all()
