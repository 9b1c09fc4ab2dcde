"""BASELINE configs[2] shape through the public API: 1M cells, radius graph r = 30 um, Lee's L for an Gx x Gy grid of
gene pairs with 199 numpy-exact permutations per pair (the reference draws a fresh block per pair).
usage: python scripts/lee_scale_probe.py GX GY [P]   (100 100 = the full config; 10 100 = a tenth of it)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
import pandas as pd
from spatialcore_amd import SimpleAnnData, _lib
from spatialcore_amd.spatial import lees_l
import logging
logging.getLogger("spatialcore_amd").setLevel(logging.WARNING)
GX, GY = int(sys.argv[1]), int(sys.argv[2])
P = int(sys.argv[3]) if len(sys.argv) > 3 else 199
SHARED = len(sys.argv) > 4 and sys.argv[4] == "shared"   # extension: one block of permutations for all pairs
N = 1_000_000
rng = np.random.default_rng(42)
coords = rng.uniform(0, np.sqrt(N) * 10, (N, 2))
G = GX + GY
lam = np.exp(rng.uniform(np.log(0.05), np.log(5.0), G))
X = rng.poisson(lam, (N, G)).astype(np.float32)
X[:, ::2] += (2.0 * (1 + np.sin(coords[:, :1] / 900.0))).astype(np.float32)
ad = SimpleAnnData(X, obs=pd.DataFrame(index=pd.RangeIndex(N).astype(str)), var_names=[f"g{i}" for i in range(G)],
                   obsm={"spatial": coords})
pairs = [(f"g{a}", f"g{GX + b}") for a in range(GX) for b in range(GY)]
ctx = _lib.default_context(0)
# warm-up: one full sub-job, so that the generator's scratch (~30 GB at this size; hipMalloc took 0.3-1.2 s of a 2.3-s sample on
# different boxes) is allocated before the clock starts -- the sample is extrapolated to 100 x its size, the allocation happens once
t0 = time.perf_counter()
lees_l(ad, pairs[:12], n_permutations=P, radius=30.0, shared_permutations=SHARED)
warm = time.perf_counter() - t0
ctx.reset_timers()
t0 = time.perf_counter()
res = lees_l(ad, pairs, n_permutations=P, seed=0, radius=30.0, shared_permutations=SHARED)
wall = time.perf_counter() - t0
lee_ms, lee_launches = ctx.kernel_time(_lib.K_LEE_PERM)
scan_ms, _ = ctx.kernel_time(_lib.K_PERM_SCAN)
L = np.array([r["L"] for r in res]); p = np.array([r["p_value"] for r in res])
n_graph, nnz = ctx.graph_shape()
print(json.dumps({"workload": f"{N} cells, radius 30 um graph ({nnz / N:.1f} neighbours per cell), {GX} x {GY} = {len(pairs)} pairs, " +
                              (f"{P} numpy-exact permutations SHARED by all pairs (extension)" if SHARED else
                               f"{P} numpy-exact permutations per pair ({len(pairs) * P} permutations of {N} in total)"),
                  "wall_s": wall, "pairs_per_s": len(pairs) / wall, "ms_per_pair": wall / len(pairs) * 1e3,
                  "generator_chain_ms": scan_ms, "lee_row_kernel_ms": lee_ms, "lee_row_kernel_launches": lee_launches,
                  "extrapolated_100x100_s": wall / len(pairs) * 1e4, "warm_up_call_s": warm,
                  "permgen_stats": ctx.permgen_stats(), "L_min_max": [float(L.min()), float(L.max())],
                  "p_min": float(p.min()), "device_mem_GiB": ctx.device_mem() / 2**30}))
