import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detrand, params as P, select_oracle as S
from murcl_amd.utils.datasets import BagPack, subbag_views, mixup_with, get_feats
T = torch.from_numpy; dev = torch.device("cuda:0")
B, N, K, fs, d = 6, 3000, 10, 256, 512
feats_np = [P.bags(41, f"f{b}", 1, N - 100 * b, d)[0] for b in range(B)]
cls = [P.cluster_lists(41, f"c{b}", N - 100 * b, K) for b in range(B)]
pack = BagPack.from_lists([T(f).to(dev) for f in feats_np], cls)
acts = [detrand.uniform(41, f"a{v}", (B, K)) for v in range(2)]
draws = [(T(detrand.uniform(41, f"l{v}", (B, 1), 0.9, 1.0)).to(dev), T(detrand.permutation(41, f"p{v}", B)).to(dev)) for v in range(2)]
views, _ = subbag_views(pack, [T(a).to(dev) for a in acts], fs, draws=draws)
sub, _ = S.get_feats(feats_np, cls, acts[0], fs)
want = S.mixup(sub, draws[0][0].cpu().numpy(), draws[0][1].cpu().numpy())
got = views[0].cpu().numpy()
bad = got != want
print("mismatch", bad.sum(), "of", bad.size, "per bag", bad.reshape(B, -1).sum(1))
sub_g = get_feats(pack, None, T(acts[0]).to(dev), fs)
print("gather exact:", np.array_equal(sub_g.cpu().numpy(), sub))
two = mixup_with(sub_g, draws[0][0], draws[0][1]).cpu().numpy()
print("standalone mixup == numpy:", np.array_equal(two, want), " fused == standalone:", np.array_equal(two, got))
lam = draws[0][0].cpu().numpy().reshape(-1); print("lam", lam, "1-lam f32", (np.float32(1) - lam))
