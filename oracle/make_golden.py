"""Generate tests/golden/*.npz -- run ONLY in the build container (needs /root/reference).

Two kinds of vectors are produced:

1. ``rng_kat.npz``  -- known answers of numpy's own ``default_rng(seed).permutation`` stream
   (numpy is the un-vendored dependency that owns this algorithm; installed here: see the
   ``numpy_version`` field).
2. ``ref_*.npz``    -- inputs and outputs of the REFERENCE's in-repo functions
   (``build_spatial_weights``, ``lees_l``, ``lees_l_local``, ``local_morans_i``, BH / Bonferroni,
   quadrants, ``compute_neighborhood_profile``), obtained by importing
   /root/reference/src/spatialcore/spatial/{autocorrelation,neighborhoods}.py with inert stand-ins
   for the absent ``anndata`` and ``squidpy`` modules.  Only data
   (arrays) is written -- no reference source travels.

The reference's global ``morans_i`` cannot be run (its arithmetic is inside squidpy, absent), so
no golden exists for it: that function's parity is "unpinned" (see oracle/oracle.py header).

Usage:  python oracle/make_golden.py
"""

from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import pandas as pd
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF_SRC = "/root/reference/src"

sys.path.insert(0, ROOT)
from spatialcore_amd._adata import SimpleAnnData  # noqa: E402


def import_reference():
    """Register stand-ins, then import the two reference modules from their own files."""
    ad = types.ModuleType("anndata")
    ad.AnnData = SimpleAnnData
    sys.modules["anndata"] = ad
    sq = types.ModuleType("squidpy")
    sq.gr = types.SimpleNamespace()
    sys.modules["squidpy"] = sq
    # bare package objects so that spatialcore/__init__ and core/__init__ (-> core/utils) are skipped
    for name, sub in (("spatialcore", ""), ("spatialcore.core", "core"), ("spatialcore.spatial", "spatial")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF_SRC, "spatialcore", sub)]
        sys.modules[name] = pkg
    ac = importlib.import_module("spatialcore.spatial.autocorrelation")
    nb = importlib.import_module("spatialcore.spatial.neighborhoods")
    global DS
    DS = importlib.import_module("spatialcore.spatial.distance")
    return ac, nb


def synth(n, n_genes, seed, dtype=np.float64, sparse_x=True):
    """Tie-free coordinates + half smooth / half Poisson-noise genes (SURVEY 8(d) recipe)."""
    rng = np.random.default_rng(seed)
    L = np.sqrt(n) * 10.0
    coords = rng.uniform(0, L, (n, 2))
    X = np.empty((n, n_genes), dtype=np.float64)
    for g in range(n_genes):
        lam = np.exp(rng.uniform(np.log(0.05), np.log(5.0)))
        if g % 2 == 0:
            wl = rng.uniform(L / 8, L / 2, 2)
            ph = rng.uniform(0, 2 * np.pi, 2)
            field = 1.0 + 0.9 * np.sin(2 * np.pi * coords[:, 0] / wl[0] + ph[0]) * np.cos(
                2 * np.pi * coords[:, 1] / wl[1] + ph[1])
            X[:, g] = rng.poisson(lam * field)
        else:
            X[:, g] = rng.poisson(lam, n)
    X = X.astype(dtype)
    return coords, (sparse.csr_matrix(X) if sparse_x else X)


def make_adata(coords, X, labels=None):
    names = [f"g{i}" for i in range(X.shape[1])]
    obs = pd.DataFrame(index=pd.RangeIndex(X.shape[0]).astype(str))
    if labels is not None:
        obs["cell_type"] = labels
    return SimpleAnnData(X, obs=obs, var_names=names, obsm={"spatial": coords})


def rng_kats():
    out = {"numpy_version": np.array(np.__version__)}
    cases = [(0, 10, 3), (0, 1000, 3), (1, 7, 5), (42, 257, 4), (123456789, 4099, 2), (7, 2, 6),
             (5, 1, 2), (2**40 + 3, 65537, 2), (0, 100000, 2)]
    for ci, (seed, n, reps) in enumerate(cases):
        rng = np.random.default_rng(seed)
        perms = np.stack([rng.permutation(n) for _ in range(reps)])
        st = rng.bit_generator.state
        out[f"case{ci}_seed"] = np.array(seed, dtype=np.uint64)
        out[f"case{ci}_n"] = np.array(n)
        # the largest case is stored as a checksum + head/tail only (keeps the fixture small)
        if n > 10000:
            w = np.arange(1, n + 1, dtype=np.uint64)
            out[f"case{ci}_checksum"] = np.array([(p.astype(np.uint64) * w).sum() for p in perms], dtype=np.uint64)
            out[f"case{ci}_head"] = perms[:, :16].astype(np.int32)
            out[f"case{ci}_tail"] = perms[:, -16:].astype(np.int32)
        else:
            out[f"case{ci}_perms"] = perms.astype(np.int32)
        m = 0xFFFFFFFFFFFFFFFF
        s, inc = st["state"]["state"], st["state"]["inc"]
        out[f"case{ci}_final_state"] = np.array(
            [s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]], dtype=np.uint64)
    out["n_cases"] = np.array(len(cases))
    # a generator that starts mid-word (has_uint32 = 1) and permutes values, not indices
    rng = np.random.default_rng(99)
    rng.integers(0, 2**32, dtype=np.uint32)  # consumes one 32-bit half -> buffered half pending
    st = rng.bit_generator.state
    m = 0xFFFFFFFFFFFFFFFF
    s, inc = st["state"]["state"], st["state"]["inc"]
    out["mid_state"] = np.array([s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]],
                                dtype=np.uint64)
    vals = np.linspace(0, 1, 37)
    out["mid_vals"] = vals
    out["mid_perm_vals"] = np.stack([rng.permutation(vals) for _ in range(3)])
    # raw 32-bit stream
    bg = np.random.PCG64(2024)
    g = np.random.Generator(bg)
    st = bg.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    out["raw_state"] = np.array([s >> 64, s & m, inc >> 64, inc & m, 0, 0], dtype=np.uint64)
    out["raw_u32"] = g.integers(0, 2**32, size=33, dtype=np.uint32, endpoint=False)
    np.savez_compressed(os.path.join(OUT, "rng_kat.npz"), **out)
    print("rng_kat.npz written")


def csr_fields(prefix, W, out):
    W = W.tocsr()
    out[prefix + "_indptr"] = W.indptr.astype(np.int64)
    out[prefix + "_indices"] = W.indices.astype(np.int32)
    out[prefix + "_data"] = W.data
    out[prefix + "_sorted"] = np.array(bool(W.has_sorted_indices))


def ref_weights(ac):
    out = {}
    for ci, (n, k, seed, inc_self) in enumerate([(500, 6, 1, False), (1500, 15, 2, False), (300, 4, 3, True), (64, 1, 4, False)]):
        coords, X = synth(n, 2, seed)
        W = ac.build_spatial_weights(make_adata(coords, X), n_neighbors=k, include_self=inc_self)
        out[f"c{ci}_coords"] = coords
        out[f"c{ci}_k"] = np.array(k)
        out[f"c{ci}_include_self"] = np.array(inc_self)
        W.sort_indices()  # canonical order for storage; the unsorted flag is stored below
        csr_fields(f"c{ci}_W", W, out)
        out[f"c{ci}_dtype"] = np.array(str(W.dtype))
    out["n_cases"] = np.array(4)
    np.savez_compressed(os.path.join(OUT, "ref_weights.npz"), **out)
    print("ref_weights.npz written")


def ref_lee(ac):
    out = {}
    cases = [
        dict(n=800, genes=6, k=6, P=19, seed=0, dtype=np.float64, pairs=[(0, 1), (2, 4), (1, 0), (0, 2)]),
        dict(n=1200, genes=5, k=15, P=9, seed=11, dtype=np.float32, pairs=[(0, 2), (3, 4)]),
        dict(n=400, genes=4, k=6, P=0, seed=5, dtype=np.float64, pairs=[(0, 2)]),
    ]
    for ci, c in enumerate(cases):
        coords, X = synth(c["n"], c["genes"], 100 + ci, dtype=c["dtype"])
        if ci == 0:  # a zero-variance gene in the middle of the list consumes no random numbers
            X = X.tolil(); X[:, 3] = 2.0; X = X.tocsr()
            c["pairs"] = [(0, 1), (3, 2), (2, 4), (1, 0), (0, 2)]
        adata = make_adata(coords, X)
        names = [(f"g{a}", f"g{b}") for a, b in c["pairs"]]
        res = ac.lees_l(adata, gene_pairs=names, n_neighbors=c["k"], n_permutations=c["P"], seed=c["seed"])
        out[f"c{ci}_coords"] = coords
        out[f"c{ci}_X"] = X.toarray()
        out[f"c{ci}_pairs"] = np.array(c["pairs"])
        out[f"c{ci}_k"] = np.array(c["k"]); out[f"c{ci}_P"] = np.array(c["P"]); out[f"c{ci}_seed"] = np.array(c["seed"])
        out[f"c{ci}_L"] = np.array([r["L"] for r in res], dtype=np.float64)
        out[f"c{ci}_p"] = np.array([r["p_value"] for r in res], dtype=np.float64)
        # single-pair call returns a dict (AC:1160-1163)
        one = ac.lees_l(adata, gene_pairs=names[0], n_neighbors=c["k"], n_permutations=c["P"], seed=c["seed"])
        assert isinstance(one, dict)
        out[f"c{ci}_single_L"] = np.array(one["L"]); out[f"c{ci}_single_p"] = np.array(one["p_value"])
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "ref_lees_l.npz"), **out)
    print("ref_lees_l.npz written")


def ref_lee_local(ac):
    out = {}
    n, k, P, seed = 600, 6, 9, 3
    coords, X = synth(n, 4, 200, dtype=np.float64)
    adata = make_adata(coords, X)
    pairs = [("g0", "g2"), ("g1", "g3")]
    ac.lees_l_local(adata, gene_pairs=pairs, n_neighbors=k, n_permutations=P,
                    compute_cell_pvalues=True, significance_filter=True, alpha=0.2, seed=seed)
    out["coords"] = coords; out["X"] = X.toarray()
    out["pairs"] = np.array([(0, 2), (1, 3)]); out["k"] = np.array(k); out["P"] = np.array(P); out["seed"] = np.array(seed)
    out["alpha"] = np.array(0.2)
    for gi, (a, b) in enumerate(pairs):
        key = f"{a}_{b}"
        out[f"p{gi}_L_local"] = np.asarray(adata.obs[f"{key}_lees_l"].values)
        out[f"p{gi}_quadrant"] = np.asarray(adata.obs[f"{key}_quadrant"].astype(str).values, dtype="U2")
        out[f"p{gi}_pvalue"] = np.asarray(adata.obs[f"{key}_pvalue"].values)
        prm = adata.uns[f"{key}_lees_l_params"]
        out[f"p{gi}_global_L"] = np.array(prm["global_L"]); out[f"p{gi}_global_p"] = np.array(prm["global_pvalue"])
        out[f"p{gi}_quadrant_counts"] = np.array([prm["quadrant_counts"][q] for q in ["NS", "HH", "LL", "HL", "LH"]])
    np.savez_compressed(os.path.join(OUT, "ref_lees_l_local.npz"), **out)
    print("ref_lees_l_local.npz written")


def ref_local_moran(ac):
    out = {}
    cases = [dict(n=400, genes=5, k=6, P=9, seed=0, batch=2, fdr="fdr_bh", alpha=0.3),
             dict(n=300, genes=3, k=4, P=0, seed=1, batch=100, fdr="bonferroni", alpha=0.05),
             dict(n=350, genes=4, k=6, P=19, seed=2, batch=3, fdr="none", alpha=0.1)]
    for ci, c in enumerate(cases):
        coords, X = synth(c["n"], c["genes"], 300 + ci, dtype=np.float32)
        if ci == 0:
            X = X.tolil(); X[:, 1] = 3.0; X = X.tocsr()   # zero-variance gene
        adata = make_adata(coords, X)
        ac.local_morans_i(adata, genes=[f"g{i}" for i in range(c["genes"])], n_neighbors=c["k"],
                          n_permutations=c["P"], fdr_correction=c["fdr"], alpha=c["alpha"], seed=c["seed"],
                          batch_size=c["batch"])
        out[f"c{ci}_coords"] = coords; out[f"c{ci}_X"] = X.toarray()
        for f in ("k", "P", "seed", "batch", "alpha"):
            out[f"c{ci}_{f}"] = np.array(c[f])
        out[f"c{ci}_fdr"] = np.array(c["fdr"])
        for f in ("I", "z", "lag", "p", "p_adj", "quadrant"):
            out[f"c{ci}_{f}"] = adata.obsm[f"local_morans_{f}"]
        out[f"c{ci}_zero_variance_genes"] = np.array(adata.uns["local_morans_params"]["zero_variance_genes"])
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "ref_local_morans.npz"), **out)
    print("ref_local_morans.npz written")


def ref_fdr_quadrants(ac):
    rng = np.random.default_rng(7)
    p = np.concatenate([rng.uniform(0, 1, 40), [0.0, 1.0, 0.05, 0.05], rng.uniform(0, 0.01, 6)])
    z = rng.normal(size=(50, 3)); lag = rng.normal(size=(50, 3)); z[0, 0] = 0.0; lag[1, 1] = 0.0
    pq = rng.uniform(0, 0.2, size=(50, 3))
    out = dict(p=p, bh=ac._fdr_correction_bh(p), bonf=ac._fdr_correction_bonferroni(p),
               z=z, lag=lag, pq=pq, quad_sig=ac._classify_quadrants(z, lag, pq, 0.05),
               quad_nosig=ac._classify_quadrants(z, lag, None, 0.05))
    np.savez_compressed(os.path.join(OUT, "ref_fdr_quadrants.npz"), **out)
    print("ref_fdr_quadrants.npz written")


def ref_profile(nb):
    out = {}
    n = 700
    coords, X = synth(n, 2, 400)
    rng = np.random.default_rng(8)
    labels = rng.choice(["T", "B", "Mac", "Epi", "Fib"], size=n, p=[0.3, 0.1, 0.2, 0.25, 0.15])
    out["coords"] = coords; out["labels"] = labels
    for name, kw in (("knn", dict(method="knn", k=15)), ("knn_raw", dict(method="knn", k=5, normalize=False)),
                     ("radius", dict(method="radius", radius=25.0)),
                     ("radius_raw", dict(method="radius", radius=18.0, normalize=False))):
        adata = make_adata(coords, X, labels)
        try:
            nb.compute_neighborhood_profile(adata, celltype_column="cell_type", **kw)
            out[f"{name}_profile"] = adata.obsm["neighborhood_profile"]
            out[f"{name}_celltypes"] = np.array(adata.uns["neighborhood_profile_celltypes"])
            out[f"{name}_error"] = np.array("")
        except ValueError as e:   # empty neighbourhoods are an error in the reference (NB:253-260)
            out[f"{name}_error"] = np.array(str(e))
        for k_, v in kw.items():
            out[f"{name}_{k_}"] = np.array(v)
    np.savez_compressed(os.path.join(OUT, "ref_profile.npz"), **out)
    print("ref_profile.npz written")


def domain_fixture():
    """Cells on a tie-free plane with blob-shaped 'domains' in two label columns (NaN = no domain)."""
    rng = np.random.default_rng(77)
    n = 1500
    coords = rng.uniform(0, 1000, (n, 2))
    centres_a = {"B_1": (200, 200), "B_2": (700, 300), "B_3": (450, 800)}
    centres_b = {"T_1": (800, 800), "T_2": (150, 650)}
    dom_a = np.full(n, None, dtype=object)
    dom_b = np.full(n, None, dtype=object)
    for name, (cx, cy) in centres_a.items():
        dom_a[np.hypot(coords[:, 0] - cx, coords[:, 1] - cy) < 90] = name
    for name, (cx, cy) in centres_b.items():
        dom_b[np.hypot(coords[:, 0] - cx, coords[:, 1] - cy) < 120] = name
    return coords, dom_a, dom_b


def ref_distance():
    coords, dom_a, dom_b = domain_fixture()
    out = {"coords": coords, "dom_a": np.array([x or "" for x in dom_a]), "dom_b": np.array([x or "" for x in dom_b])}
    cases = [("min_both", dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="minimum", output_mode="both")),
             ("min_matrix", dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="minimum", output_mode="matrix")),
             ("mean_both", dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="mean", output_mode="both")),
             ("centroid_both", dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="centroid", output_mode="both")),
             ("self_min", dict(source_domain_column="dom_a", target_domain_column="dom_a", distance_metric="minimum", output_mode="both")),
             ("self_centroid", dict(source_domain_column="dom_a", target_domain_column="dom_a", distance_metric="centroid", output_mode="both")),
             ("subset_min", dict(source_domain_column="dom_a", target_domain_column="dom_b", source_domain_subset=["B_1", "B_3"],
                                 target_domain_subset=["T_2"], distance_metric="minimum", output_mode="both"))]
    for name, kw in cases:
        obs = pd.DataFrame({"dom_a": dom_a, "dom_b": dom_b}, index=pd.RangeIndex(len(coords)).astype(str))
        adata = SimpleAnnData(np.zeros((len(coords), 1)), obs=obs, var_names=["g0"], obsm={"spatial": coords})
        DS.calculate_domain_distances(adata, **kw)
        if kw["output_mode"] in ("cell", "both"):
            out[f"{name}_dist"] = adata.obs["distance_to_target"].values.astype(np.float64)
            out[f"{name}_nearest"] = np.array([x if isinstance(x, str) else "" for x in adata.obs["nearest_target_domain"].values])
        dd = adata.uns["domain_distances"]
        m = DS.get_distance_matrix(adata)
        out[f"{name}_matrix"] = m.values.astype(np.float64)
        out[f"{name}_rows"] = np.array(list(m.index)); out[f"{name}_cols"] = np.array(list(m.columns))
        out[f"{name}_summary"] = np.array([dd["summary_statistics"][k] for k in ("min_distance", "max_distance", "mean_distance", "median_distance")])
    np.savez_compressed(os.path.join(OUT, "ref_distance.npz"), **out)
    print("ref_distance.npz written")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng_kats()
    ac, nb = import_reference()
    ref_weights(ac)
    ref_lee(ac)
    ref_lee_local(ac)
    ref_local_moran(ac)
    ref_fdr_quadrants(ac)
    ref_profile(nb)
    ref_distance()


if __name__ == "__main__":
    main()
