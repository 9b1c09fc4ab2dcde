"""GPU box: wall time of the public functions next to the headline path at 1M cells (not a test).
Run under `rocprofv3 --kernel-trace --stats` to see which kernels carry each call."""
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
G = 128
X = rng.poisson(1.0, (N, G)).astype(np.float32)
labels = rng.integers(0, 12, N)
ad = make_adata(coords, X, labels)
ad.obs["dom_a"] = np.where(coords[:, 0] < L * 0.2, "A1", np.where(coords[:, 0] < L * 0.3, "A2", None))
ad.obs["dom_b"] = np.where(coords[:, 0] > L * 0.8, "B1", np.where(coords[:, 1] > L * 0.9, "B2", None))
def timed(name, f):
    t = time.time(); r = f(); print(f"TIMING {name}: {time.time() - t:.2f} s", flush=True); return r
names = list(ad.var_names)
timed("build_spatial_weights k=6", lambda: sp.build_spatial_weights(ad, n_neighbors=6))
timed("morans_i 128 genes, k=15, P=199", lambda: sp.morans_i(ad, n_neighbors=15, n_permutations=199))
timed("local_morans_i 100 genes, P=99", lambda: sp.local_morans_i(ad, genes=names[:100], n_permutations=99))
pairs = [(names[2 * i], names[2 * i + 1]) for i in range(10)]
timed("lees_l 10 pairs, P=199", lambda: sp.lees_l(ad, pairs, n_permutations=199))
timed("lees_l_local 2 pairs, cell p-values, P=99", lambda: sp.lees_l_local(ad, pairs[:2], n_permutations=99, compute_cell_pvalues=True))
timed("compute_neighborhood_profile k=30", lambda: sp.compute_neighborhood_profile(ad, "cell_type", k=30))
timed("compute_neighborhood_profile radius=30", lambda: sp.compute_neighborhood_profile(ad, "cell_type", method="radius", radius=30.0))
timed("neighborhood_enrichment k=30, P=1000", lambda: sp.neighborhood_enrichment(ad, "cell_type", k=30, n_permutations=1000))
timed("calculate_domain_distances minimum", lambda: sp.calculate_domain_distances(ad, "dom_a", "dom_b"))
timed("calculate_domain_distances mean", lambda: sp.calculate_domain_distances(ad, "dom_a", "dom_b", distance_metric="mean"))
