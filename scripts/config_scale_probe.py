"""At-size runs of the remaining BASELINE shapes through the public API (GPU box; not a test):
  local   local_morans_i, 1M cells x 100 genes x 999 permutations (N1) + lees_l_local, 2 pairs with per-cell p-values (N2)
  enrich  neighborhood_enrichment, 1M cells, k = 30, ~20 cell types, 10 000 label permutations (BASELINE configs[4])
usage: python scripts/config_scale_probe.py local|enrich [n_perm] [numpy|philox]"""
import json, logging, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_adata
from spatialcore_amd import _lib
import spatialcore_amd.spatial as sp
logging.getLogger("spatialcore_amd").setLevel(logging.WARNING)
what = sys.argv[1]
N = 1_000_000
rng = np.random.default_rng(1)
coords = rng.uniform(0, np.sqrt(N) * 10, (N, 2))
ctx = _lib.default_context(0)
out = {"what": what, "cells": N}
if what == "local":
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 999
    G = 100
    X = rng.poisson(1.0, (N, G)).astype(np.float32)
    ad = make_adata(coords, X)
    names = list(ad.var_names)
    sp.local_morans_i(ad, genes=names[:16], n_permutations=3)                      # warm-up (allocations)
    t = time.perf_counter(); sp.local_morans_i(ad, genes=names, n_neighbors=6, n_permutations=P); out["local_morans_i_s"] = time.perf_counter() - t
    out["local_morans_i"] = f"{G} genes, k=6, {P} permutations, all six obsm outputs"
    t = time.perf_counter(); sp.lees_l_local(ad, [(names[0], names[1]), (names[2], names[3])], n_permutations=P, compute_cell_pvalues=True)
    out["lees_l_local_s"] = time.perf_counter() - t
    out["lees_l_local"] = f"2 pairs, k=6, {P} permutations for the global p + {P} for the per-cell p-values each"
else:
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    w = rng.dirichlet(np.full(20, 0.7))
    labels = rng.choice(20, N, p=w)
    ad = make_adata(coords, np.zeros((N, 1), dtype=np.float32), labels)
    sp.neighborhood_enrichment(ad, "cell_type", k=30, n_permutations=8)           # warm-up
    source = sys.argv[3] if len(sys.argv) > 3 else "numpy"
    sp.neighborhood_enrichment(ad, "cell_type", k=30, n_permutations=600, seed=1, rng=source)  # warm-up at the batch size (allocations)
    t = time.perf_counter(); sp.neighborhood_enrichment(ad, "cell_type", k=30, n_permutations=P, seed=0, rng=source); out["neighborhood_enrichment_s"] = time.perf_counter() - t
    r = ad.uns["neighborhood_enrichment"]
    out["neighborhood_enrichment"] = (f"k=30, {len(r['celltypes'])} cell types, {P} label permutations, rng={source} "
                                      + ("(numpy-exact: one sequential stream, one GPU)" if source == "numpy"
                                         else "(counter-based: Philox4x32-10 + Lemire, shardable over GPUs)"))
    out["zscore_diag_mean"] = float(np.nanmean(np.diag(r["zscore"])))
    out["permutations_per_s"] = P / out["neighborhood_enrichment_s"]
out["permgen_stats"] = ctx.permgen_stats()
out["device_mem_GiB"] = ctx.device_mem() / 2**30
print(json.dumps(out), flush=True)
