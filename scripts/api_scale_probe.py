"""GPU box: wall time of the public functions next to the headline path at 1M cells (not a test)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import make_adata
import spatialcore_amd.spatial as sp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1)
L = np.sqrt(N) * 10
coords = rng.uniform(0, L, (N, 2))
G = 32
X = rng.poisson(1.0, (N, G)).astype(np.float32)
labels = rng.integers(0, 12, N)
ad = make_adata(coords, X, labels)
def timed(name, f):
    t = time.time(); r = f(); print(f"{name}: {time.time() - t:.2f} s", flush=True); return r
timed("build_spatial_weights k=6", lambda: sp.build_spatial_weights(ad, n_neighbors=6))
timed("morans_i 32 genes, k=15, P=199", lambda: sp.morans_i(ad, n_neighbors=15, n_permutations=199))
timed("local_morans_i 8 genes, P=99", lambda: sp.local_morans_i(ad, genes=list(ad.var_names[:8]), n_permutations=99))
pairs = [(ad.var_names[0], ad.var_names[1]), (ad.var_names[2], ad.var_names[3])]
timed("lees_l 2 pairs, P=199", lambda: sp.lees_l(ad, pairs, n_permutations=199))
timed("lees_l_local 1 pair, P=99", lambda: sp.lees_l_local(ad, pairs[:1], n_permutations=99))
col = [c for c in ad.obs.columns][0]
timed("compute_neighborhood_profile k=30", lambda: sp.compute_neighborhood_profile(ad, col, k=30))
