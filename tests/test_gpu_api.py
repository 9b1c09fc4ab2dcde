"""GPU: the AnnData-level drop-in functions vs the reference's golden outputs and the oracle."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden, make_adata, synth

pytestmark = pytest.mark.gpu


def test_build_spatial_weights_golden():
    from spatialcore_amd.spatial import build_spatial_weights

    g = load_golden("ref_weights.npz")
    for ci in range(int(g["n_cases"])):
        coords = g[f"c{ci}_coords"]
        ad = make_adata(coords, np.zeros((coords.shape[0], 1)))
        W = build_spatial_weights(ad, n_neighbors=int(g[f"c{ci}_k"]), include_self=bool(g[f"c{ci}_include_self"]))
        assert W.dtype == np.float32 and W.format == "csr" and W.has_sorted_indices
        np.testing.assert_array_equal(W.indptr, g[f"c{ci}_W_indptr"])
        np.testing.assert_array_equal(W.indices, g[f"c{ci}_W_indices"])
        np.testing.assert_array_equal(W.data, g[f"c{ci}_W_data"])
    with pytest.raises(ValueError, match="not found"):
        build_spatial_weights(ad, spatial_key="nope")


def test_morans_i_table_matches_oracle(oracle):
    from spatialcore_amd.spatial import morans_i

    coords, X = synth(10000, 50, 1, dtype=np.float32)           # BASELINE configs[0]
    ad = make_adata(coords, X)
    genes = [f"g{i}" for i in (3, 0, 17, 42, 8)]
    out = morans_i(ad, genes=genes, n_neighbors=6, n_permutations=199, seed=0)
    assert out is ad
    df = ad.uns["morans_i"]
    assert list(df.columns) == ["gene", "I", "expected_I", "z_score", "p_value"]
    assert list(df["gene"]) == genes
    tab = oracle.morans_i_reference_table(coords, X, [3, 0, 17, 42, 8], 6, 199, seed=0)
    np.testing.assert_allclose(df["I"].values, tab["I"], rtol=1e-9)
    np.testing.assert_allclose(df["z_score"].values, tab["z_score"], rtol=1e-9)
    np.testing.assert_allclose(df["expected_I"].values, -1 / 9999, rtol=1e-15)
    # p-values are (count+1)/(P+1); the integer-count genes of a kNN graph are decided on the exact integer lattice
    # (DESIGN.md "Ties"): identical to the oracle's, exact ties included
    assert tab["lattice"].all()
    np.testing.assert_array_equal(df["p_value"].values, tab["p_value"])
    # side effects of the reference: graph left in obsp, provenance appended
    conn = ad.obsp["spatial_connectivities"]
    assert conn.shape == (10000, 10000) and conn.nnz == 60000 and conn.dtype == np.float64
    np.testing.assert_array_equal(conn.indices.reshape(-1, 6), oracle.knn_bruteforce(coords, 6))
    assert ad.uns["spatialcore_metadata"]["operations"][-1]["function"] == "morans_i"
    # use_existing_graph re-uses obsp; P = 0 falls back to the analytic p-value
    morans_i(ad, genes=genes, n_neighbors=6, n_permutations=0, use_existing_graph=True, key_added="m0")
    np.testing.assert_allclose(ad.uns["m0"]["I"].values, tab["I"], rtol=1e-9)
    t0 = oracle.morans_i_reference_table(coords, X, [3, 0, 17, 42, 8], 6, 0, seed=0)
    np.testing.assert_allclose(ad.uns["m0"]["p_value"].values, t0["p_value"], rtol=1e-6, atol=1e-300)


def test_morans_i_table_matches_oracle_on_log_normalised_values(oracle):
    """The usual input of Moran's I -- size-factor normalised, log1p'ed float32 values: no integer, no lattice gene --
    through the public function: the centred float32-source kernel.  I and z at 1e-9; p-values equal to the oracle's
    except where a permutation's statistic is within rounding noise of the observed one (none here, asserted)."""
    from spatialcore_amd.spatial import morans_i

    coords, X = synth(10000, 50, 1, normalize=True)             # BASELINE configs[0], log-normalised
    ad = make_adata(coords, X)
    cols = [3, 0, 17, 42, 8, 49, 21]
    genes = [f"g{i}" for i in cols]
    morans_i(ad, genes=genes, n_neighbors=6, n_permutations=199, seed=0)
    df = ad.uns["morans_i"]
    tab = oracle.morans_i_reference_table(coords, X, cols, 6, 199, seed=0)
    assert not tab["lattice"].any()
    np.testing.assert_allclose(df["I"].values, tab["I"], rtol=1e-9)
    np.testing.assert_allclose(df["z_score"].values, tab["z_score"], rtol=1e-9)
    near = (np.abs(tab["sims"] - tab["I"]) <= 1e-11 * np.abs(tab["I"])).sum(axis=0)
    assert (near == 0).all()                                    # no rounding-level near-tie in this data: p must be EQUAL
    np.testing.assert_array_equal(df["p_value"].values, tab["p_value"])


def test_morans_i_gene_batches_share_the_permutation_table():
    """Genes scored in device batches (a matrix too wide for HBM) see the same permutations: the table of the first
    batch stays resident.  Batched == unbatched, bit for bit, dense and sparse, 130 permutations (two chunks)."""
    from spatialcore_amd.spatial import morans_i

    coords, X = synth(70001, 23, 6, dtype=np.float32)
    genes = [f"g{i}" for i in np.random.default_rng(2).permutation(23)]
    whole = make_adata(coords, X)
    morans_i(whole, genes=genes, n_neighbors=6, n_permutations=130, seed=9)
    for M in (X, X.toarray()):
        part = make_adata(coords, M)
        morans_i(part, genes=genes, n_neighbors=6, n_permutations=130, seed=9, gene_batch=7)
        for col in ("I", "expected_I", "z_score", "p_value"):
            np.testing.assert_array_equal(part.uns["morans_i"][col].values, whole.uns["morans_i"][col].values, err_msg=col)
    with pytest.raises(ValueError, match="gene_batch must be >= 1"):
        morans_i(whole, genes=genes, n_permutations=2, gene_batch=0)
    # a count >= 256 (uint16 source) and a fractional gene (float32 source) in OTHER batches than the rest: the source
    # width is a per-batch decision, a gene's I / p-value is not (r02 advisor finding: the uint8 kernel used to round
    # differently from the wider ones, so p-values depended on the batch mates)
    Xm = X.toarray()
    Xm[5, 3] = 300.0
    Xm[:, 20] += np.float32(0.5)
    whole = make_adata(coords, Xm)
    order = [f"g{i}" for i in range(23)]
    morans_i(whole, genes=order, n_neighbors=6, n_permutations=130, seed=9)                # one batch: float32 source
    part = make_adata(coords, Xm)
    morans_i(part, genes=order, n_neighbors=6, n_permutations=130, seed=9, gene_batch=6)   # uint16, uint8, uint8, float32
    for col in ("I", "z_score", "p_value"):
        np.testing.assert_array_equal(part.uns["morans_i"][col].values, whole.uns["morans_i"][col].values, err_msg=col)


def test_morans_i_errors_and_copy():
    from spatialcore_amd.spatial import morans_i

    coords, X = synth(500, 3, 2)
    ad = make_adata(coords, X)
    with pytest.raises(ValueError, match="n_neighbors must be >= 1"):
        morans_i(ad, n_neighbors=0)
    with pytest.raises(ValueError, match="n_permutations must be >= 0"):
        morans_i(ad, n_permutations=-1)
    with pytest.raises(ValueError, match="Genes not found"):
        morans_i(ad, genes=["nope"])
    with pytest.raises(ValueError, match="not found"):
        morans_i(ad, spatial_key="xy")
    out = morans_i(ad, genes="g1", n_permutations=5, copy=True)
    assert out is not ad and "morans_i" in out.uns and "morans_i" not in ad.uns
    assert len(out.uns["morans_i"]) == 1


def test_lees_l_golden():
    from spatialcore_amd.spatial import lees_l

    g = load_golden("ref_lees_l.npz")
    for ci in range(int(g["n_cases"])):
        X = g[f"c{ci}_X"]
        ad = make_adata(g[f"c{ci}_coords"], X)
        pairs = [(f"g{a}", f"g{b}") for a, b in g[f"c{ci}_pairs"]]
        res = lees_l(ad, gene_pairs=pairs, n_neighbors=int(g[f"c{ci}_k"]), n_permutations=int(g[f"c{ci}_P"]),
                     seed=int(g[f"c{ci}_seed"]))
        assert isinstance(res, list) and [(r["gene_x"], r["gene_y"]) for r in res] == pairs
        if X.dtype == np.float64:
            np.testing.assert_allclose([r["L"] for r in res], g[f"c{ci}_L"], rtol=1e-9, atol=1e-9)
        else:   # float32 matrix: the reference's own float32 arithmetic, reproduced bit for bit
            np.testing.assert_array_equal([r["L"] for r in res], g[f"c{ci}_L"])
        tol = 1e-9 if X.dtype == np.float64 else 0.0
        np.testing.assert_array_equal([r["p_value"] for r in res], g[f"c{ci}_p"])
        one = lees_l(ad, gene_pairs=pairs[0], n_neighbors=int(g[f"c{ci}_k"]), n_permutations=int(g[f"c{ci}_P"]),
                     seed=int(g[f"c{ci}_seed"]))
        assert isinstance(one, dict)
        assert one["L"] == pytest.approx(float(g[f"c{ci}_single_L"]), rel=tol, abs=tol)
        assert one["p_value"] == float(g[f"c{ci}_single_p"])


def test_lees_l_constant_gene_at_awkward_cell_count(oracle):
    """n = 501 cells (501 * fl(1/501) != 1): a gene constant at 1.0 must be zero-variance -> L = 0, p = 1, NO draws
    from the shared stream (AC:1129-1140), so the later pair's p-value is what it is without that pair."""
    from spatialcore_amd.spatial import lees_l, morans_i

    coords, X = synth(501, 4, 8, dtype=np.float64, sparse_x=False)
    X[:, 1] = 1.0
    ad = make_adata(coords, X)
    res = lees_l(ad, gene_pairs=[("g0", "g1"), ("g2", "g3")], n_neighbors=6, n_permutations=99, seed=3)
    assert res[0]["L"] == 0.0 and res[0]["p_value"] == 1.0
    alone = lees_l(ad, gene_pairs=[("g2", "g3")], n_neighbors=6, n_permutations=99, seed=3)
    assert res[1] == alone[0]
    want = oracle.lees_l(coords, X, [(0, 1), (2, 3)], 6, 99, 3)
    assert want[0]["L"] == 0.0 and want[0]["p_value"] == 1.0
    assert res[1]["p_value"] == want[1]["p_value"] and res[1]["L"] == pytest.approx(want[1]["L"], rel=1e-9)
    morans_i(ad, genes=["g0", "g1"], n_neighbors=6, n_permutations=9)
    assert np.isnan(ad.uns["morans_i"]["I"].values[1])


def test_radius_keyword_extension_for_moran_and_lee(oracle):
    """`radius=` (extension for BASELINE configs[2]; the reference's Moran / Lee functions only take n_neighbors):
    closed-ball radius graph, self excluded, row-normalised -- against the oracle on the same graph."""
    from scipy.sparse import csr_matrix
    from spatialcore_amd.spatial import lees_l, morans_i

    n, G, P, r = 3000, 6, 29, 10.0
    coords, X = synth(n, G, 12, dtype=np.float64, sparse_x=False)
    indptr, indices = oracle.radius_neighbors(coords, r)
    assert (np.diff(indptr) > 0).mean() > 0.9 and (np.diff(indptr) == 0).any()     # some isolated cells: empty rows stay
    conn = csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    ad = make_adata(coords, X)
    morans_i(ad, genes=[f"g{i}" for i in range(G)], n_permutations=P, seed=2, radius=r)
    tab = oracle.morans_i_reference_table(coords, X, list(range(G)), 6, P, seed=2, graph=conn)
    np.testing.assert_allclose(ad.uns["morans_i"]["I"].values, tab["I"], rtol=1e-9)
    np.testing.assert_allclose(ad.uns["morans_i"]["z_score"].values, tab["z_score"], rtol=1e-9)
    ties = (np.abs(tab["sims"] - tab["I"]) <= 1e-11 * np.abs(tab["I"])).sum(axis=0)
    assert (np.abs(ad.uns["morans_i"]["p_value"].values - tab["p_value"]) <= ties / (P + 1) + 1e-15).all()
    assert ad.obsp["spatial_connectivities"].nnz == indices.size
    assert ad.uns["spatial_neighbors"]["params"]["radius"] == r
    # Lee: float32 row-normalised weights like build_spatial_weights, the reference's core loop on that W
    deg = np.diff(indptr)
    w = np.repeat(np.where(deg > 0, np.float32(1.0) / np.maximum(deg, 1).astype(np.float32), 0).astype(np.float32), deg)
    W = csr_matrix((w, indices, indptr), shape=(n, n))
    rng = np.random.default_rng(5)
    want = []
    for a, b in [(0, 1), (2, 3)]:
        zx = (X[:, a] - X[:, a].mean()) / X[:, a].std()
        zy = (X[:, b] - X[:, b].mean()) / X[:, b].std()
        _, L, _, p, _ = oracle.lees_l_core(zx, zy, W.astype(np.float64), 19, rng)
        want.append((L, p))
    got = lees_l(ad, [("g0", "g1"), ("g2", "g3")], n_permutations=19, seed=5, radius=r)
    for gq, (L, p) in zip(got, want):
        assert gq["L"] == pytest.approx(L, rel=1e-9) and gq["p_value"] == p
    with pytest.raises(ValueError, match="radius must be > 0"):
        morans_i(ad, n_permutations=2, radius=0.0)


def test_neighborhood_profile_golden():
    from spatialcore_amd.spatial import compute_neighborhood_profile

    g = load_golden("ref_profile.npz")
    X = np.zeros((g["coords"].shape[0], 1))
    for name in ("knn", "knn_raw", "radius", "radius_raw"):
        ad = make_adata(g["coords"], X, labels=g["labels"])
        kw = dict(method=str(g[f"{name}_method"]))
        if kw["method"] == "knn":
            kw["k"] = int(g[f"{name}_k"])
        else:
            kw["radius"] = float(g[f"{name}_radius"])
        if f"{name}_normalize" in g:
            kw["normalize"] = bool(g[f"{name}_normalize"])
        err = str(g[f"{name}_error"])
        if err:
            with pytest.raises(ValueError) as ei:
                compute_neighborhood_profile(ad, "cell_type", **kw)
            assert str(ei.value) == err
            continue
        compute_neighborhood_profile(ad, "cell_type", **kw)
        prof = ad.obsm["neighborhood_profile"]
        assert prof.dtype == np.float32
        np.testing.assert_array_equal(prof, g[f"{name}_profile"])
        assert ad.uns["neighborhood_profile_celltypes"] == list(g[f"{name}_celltypes"])
    ad = make_adata(g["coords"], X, labels=g["labels"])
    with pytest.raises(ValueError, match="'radius' must be provided"):
        compute_neighborhood_profile(ad, "cell_type", method="radius")
    with pytest.raises(ValueError, match="k must be <"):
        compute_neighborhood_profile(ad, "cell_type", k=10**6)
    with pytest.raises(ValueError, match="empty neighborhood profiles"):
        compute_neighborhood_profile(ad, "cell_type", method="radius", radius=1e-3)


def test_local_morans_i_golden():
    """Reference golden (float32 outputs of the reference's own local_morans_i): every array bit for
    bit -- z, lag, I, per-cell p, adjusted p, quadrants.  (The float32 mean / sd are reproduced with
    numpy's pairwise summation order, see k_np_colstats.)"""
    from spatialcore_amd.spatial import local_morans_i

    g = load_golden("ref_local_morans.npz")
    for ci in range(int(g["n_cases"])):
        X = g[f"c{ci}_X"]
        for as_sparse in (False, True):
            from scipy import sparse as sp
            ad = make_adata(g[f"c{ci}_coords"], sp.csr_matrix(X) if as_sparse else X)
            local_morans_i(ad, genes=[f"g{i}" for i in range(X.shape[1])], n_neighbors=int(g[f"c{ci}_k"]),
                           n_permutations=int(g[f"c{ci}_P"]), fdr_correction=str(g[f"c{ci}_fdr"]),
                           alpha=float(g[f"c{ci}_alpha"]), seed=int(g[f"c{ci}_seed"]), batch_size=int(g[f"c{ci}_batch"]))
            for f in ("z", "lag", "I", "p", "p_adj", "quadrant"):
                got = ad.obsm[f"local_morans_{f}"]
                assert got.dtype == g[f"c{ci}_{f}"].dtype
                np.testing.assert_array_equal(got, g[f"c{ci}_{f}"], err_msg=f"case {ci} field {f} sparse={as_sparse}")
            assert list(ad.uns["local_morans_params"]["zero_variance_genes"]) == list(g[f"c{ci}_zero_variance_genes"])
            assert ad.uns["spatialcore_metadata"]["operations"][-1]["function"] == "local_morans_i"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_local_morans_i_vs_oracle_larger(oracle, dtype):
    from spatialcore_amd.spatial import local_morans_i

    coords, X = synth(3000, 7, 9, dtype=dtype)
    ad = make_adata(coords, X)
    local_morans_i(ad, n_neighbors=6, n_permutations=29, seed=3, batch_size=4, fdr_correction="fdr_bh", alpha=0.4)
    r = oracle.local_morans_i(coords, X, np.arange(7), 6, 29, 3, fdr="fdr_bh", alpha=0.4, batch_size=4)
    for f in ("z", "lag", "I", "p", "p_adj", "quadrant"):
        np.testing.assert_array_equal(ad.obsm[f"local_morans_{f}"], r[f], err_msg=f)
    # a gene list that is neither sorted nor free of repeats, every correction method, and no permutations at all
    # (800 permutations: the count histogram no longer fits the workgroup-private LDS copy)
    for fdr, P in (("bonferroni", 19), ("none", 19), ("fdr_bh", 0), ("fdr_bh", 800)):
        genes = ["g3", "g1", "g3", "g0", "g6"]
        ad2 = make_adata(coords, X)
        local_morans_i(ad2, genes=genes, n_neighbors=6, n_permutations=P, seed=4, batch_size=3, fdr_correction=fdr, alpha=0.3)
        r2 = oracle.local_morans_i(coords, X, np.array([3, 1, 3, 0, 6]), 6, P, 4, fdr=fdr, alpha=0.3, batch_size=3)
        for f in ("z", "lag", "I", "p", "p_adj", "quadrant"):
            np.testing.assert_array_equal(ad2.obsm[f"local_morans_{f}"], r2[f], err_msg=f"{fdr} P={P} {f}")
    with pytest.raises(ValueError, match="Invalid fdr_correction"):
        local_morans_i(ad, fdr_correction="holm")


def test_lees_l_local_golden():
    from spatialcore_amd.spatial import lees_l_local

    g = load_golden("ref_lees_l_local.npz")
    ad = make_adata(g["coords"], g["X"])
    pairs = [(f"g{a}", f"g{b}") for a, b in g["pairs"]]
    lees_l_local(ad, gene_pairs=pairs, n_neighbors=int(g["k"]), n_permutations=int(g["P"]),
                 compute_cell_pvalues=True, significance_filter=True, alpha=float(g["alpha"]), seed=int(g["seed"]))
    for gi, (a, b) in enumerate(pairs):
        key = f"{a}_{b}"
        np.testing.assert_allclose(ad.obs[f"{key}_lees_l"].values, g[f"p{gi}_L_local"], rtol=1e-6, atol=1e-7)
        assert ad.obs[f"{key}_lees_l"].dtype == np.float32
        np.testing.assert_array_equal(ad.obs[f"{key}_pvalue"].values, g[f"p{gi}_pvalue"])
        np.testing.assert_array_equal(ad.obs[f"{key}_quadrant"].astype(str).values, g[f"p{gi}_quadrant"])
        prm = ad.uns[f"{key}_lees_l_params"]
        assert prm["global_L"] == pytest.approx(float(g[f"p{gi}_global_L"]), rel=1e-9)
        assert prm["global_pvalue"] == float(g[f"p{gi}_global_p"])
        assert [prm["quadrant_counts"][q] for q in ["NS", "HH", "LL", "HL", "LH"]] == list(g[f"p{gi}_quadrant_counts"])
    with pytest.raises(ValueError, match="requires compute_cell_pvalues"):
        lees_l_local(ad, gene_pairs=pairs, significance_filter=True)
    with pytest.raises(ValueError, match="Must provide either"):
        lees_l_local(ad)


def _domain_adata(g):
    import pandas as pd
    from spatialcore_amd import SimpleAnnData

    coords = g["coords"]
    obs = pd.DataFrame({"dom_a": [x if x else None for x in g["dom_a"]], "dom_b": [x if x else None for x in g["dom_b"]]},
                       index=pd.RangeIndex(len(coords)).astype(str))
    return SimpleAnnData(np.zeros((len(coords), 1)), obs=obs, var_names=["g0"], obsm={"spatial": coords})


def test_calculate_domain_distances_golden():
    """Goldens from the reference's own calculate_domain_distances / get_distance_matrix."""
    from spatialcore_amd.spatial import calculate_domain_distances, get_distance_matrix

    g = load_golden("ref_distance.npz")
    cases = {
        "min_both": dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="minimum", output_mode="both"),
        "min_matrix": dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="minimum", output_mode="matrix"),
        "mean_both": dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="mean", output_mode="both"),
        "centroid_both": dict(source_domain_column="dom_a", target_domain_column="dom_b", distance_metric="centroid", output_mode="both"),
        "self_min": dict(source_domain_column="dom_a", target_domain_column="dom_a", distance_metric="minimum", output_mode="both"),
        "self_centroid": dict(source_domain_column="dom_a", target_domain_column="dom_a", distance_metric="centroid", output_mode="both"),
        "subset_min": dict(source_domain_column="dom_a", target_domain_column="dom_b", source_domain_subset=["B_1", "B_3"],
                           target_domain_subset=["T_2"], distance_metric="minimum", output_mode="both"),
    }
    for name, kw in cases.items():
        ad = _domain_adata(g)
        calculate_domain_distances(ad, **kw)
        m = get_distance_matrix(ad)
        assert list(m.index) == list(g[f"{name}_rows"]) and list(m.columns) == list(g[f"{name}_cols"]), name
        exact = kw["distance_metric"] == "minimum"
        if exact:   # min of sqrt(fl(dx^2)+fl(dy^2)): same floats as cKDTree / cdist
            np.testing.assert_array_equal(m.values.astype(float), g[f"{name}_matrix"], err_msg=name)
        else:       # mean: summation order; centroid: identical numpy means
            np.testing.assert_allclose(m.values.astype(float), g[f"{name}_matrix"], rtol=1e-12, err_msg=name)
        np.testing.assert_allclose([ad.uns["domain_distances"]["summary_statistics"][k] for k in
                                    ("min_distance", "max_distance", "mean_distance", "median_distance")],
                                   g[f"{name}_summary"], rtol=1e-12)
        if kw["output_mode"] in ("cell", "both"):
            d = ad.obs["distance_to_target"].values.astype(float)
            if kw["distance_metric"] == "centroid":
                np.testing.assert_allclose(d, g[f"{name}_dist"], rtol=1e-14, equal_nan=True, err_msg=name)
            else:
                np.testing.assert_array_equal(d, g[f"{name}_dist"], err_msg=name)
            near = np.array([x if isinstance(x, str) else "" for x in ad.obs["nearest_target_domain"].values])
            np.testing.assert_array_equal(near, g[f"{name}_nearest"], err_msg=name)
        assert ad.uns["spatialcore_metadata"]["operations"][-1]["function"] == "calculate_domain_distances"
    ad = _domain_adata(g)
    with pytest.raises(ValueError, match="Invalid distance_metric"):
        calculate_domain_distances(ad, "dom_a", "dom_b", distance_metric="median")
    with pytest.raises(ValueError, match="Target column 'nope' not found"):
        calculate_domain_distances(ad, "dom_a", "nope")
    with pytest.raises(ValueError, match="No valid source domains"):
        calculate_domain_distances(ad, "dom_a", "dom_b", source_domain_subset=["zzz"])
    with pytest.raises(KeyError):
        get_distance_matrix(ad)


def test_neighborhood_enrichment_extension(oracle):
    """N4 is an extension (nothing to match in the reference): checked against the oracle's restatement
    of the definition, exact integer counts, numpy-exact label permutations continued across batches."""
    from spatialcore_amd.spatial import neighborhood_enrichment

    rng = np.random.default_rng(21)
    n = 3000
    coords = rng.uniform(0, 550, (n, 2))
    # spatially structured labels: left half mostly A/B, right half mostly C/D
    left = coords[:, 0] < 275
    labels = np.where(left, rng.choice(["A", "B", "E"], n, p=[.5, .4, .1]), rng.choice(["C", "D", "E"], n, p=[.5, .4, .1]))
    X = np.zeros((n, 1))
    for kw in (dict(method="knn", k=8), dict(method="radius", radius=22.0)):
        ad = make_adata(coords, X, labels=labels)
        neighborhood_enrichment(ad, "cell_type", n_permutations=37, seed=5, perm_batch=16, **kw)
        res = ad.uns["neighborhood_enrichment"]
        cats = sorted(set(labels.tolist()))
        assert res["celltypes"] == cats
        codes = np.array([cats.index(v) for v in labels.tolist()])
        if kw["method"] == "knn":
            nbr = oracle.knn_bruteforce(coords, 8)
            indptr, indices = np.arange(0, n * 8 + 1, 8), nbr.reshape(-1)
        else:
            indptr, indices = oracle.radius_neighbors(coords, 22.0)
        perms, _ = oracle.perm_table(5, n, 37)
        want = oracle.enrichment_counts(indptr, indices, codes, len(cats), perms)
        np.testing.assert_array_equal(res["count"], want[-1])
        null = want[:-1].astype(float)
        np.testing.assert_allclose(res["mean"], null.mean(axis=0), rtol=1e-12)
        np.testing.assert_allclose(res["std"], null.std(axis=0), rtol=1e-9, atol=1e-9)
        np.testing.assert_array_equal(res["p_value"], ((want[:-1] >= want[-1]).sum(axis=0) + 1) / 38)
        assert res["count"].sum() == indices.size
        ia, ic = cats.index("A"), cats.index("C")
        assert res["zscore"][ia, ia] > 3 and res["zscore"][ia, ic] < -3     # same-side types attract
    with pytest.raises(ValueError, match="Invalid method"):
        neighborhood_enrichment(ad, "cell_type", method="grid")
    # rng="philox": the counter-based source (sc_perm_generate_counter), permutation p a pure function of (seed, p);
    # against the oracle's restatement of that definition (numpy Philox4x32-10 pinned by Random123's known answers)
    ad = make_adata(coords, X, labels=labels)
    neighborhood_enrichment(ad, "cell_type", method="knn", k=8, n_permutations=21, seed=77, perm_batch=8, rng="philox")
    res = ad.uns["neighborhood_enrichment"]
    nbr = oracle.knn_bruteforce(coords, 8)
    perms = np.stack([oracle.counter_permutation(77, n, p) for p in range(21)])
    want = oracle.enrichment_counts(np.arange(0, n * 8 + 1, 8), nbr.reshape(-1), codes, len(cats), perms)
    np.testing.assert_array_equal(res["count"], want[-1])
    np.testing.assert_array_equal(res["p_value"], ((want[:-1] >= want[-1]).sum(axis=0) + 1) / 22)
    np.testing.assert_allclose(res["mean"], want[:-1].astype(float).mean(axis=0), rtol=1e-12)
    np.testing.assert_allclose(res["std"], want[:-1].astype(float).std(axis=0), rtol=1e-9, atol=1e-9)
    with pytest.raises(ValueError, match="rng must be"):
        neighborhood_enrichment(ad, "cell_type", rng="mt19937")


def test_counter_permutations_on_the_device_and_lee_shared_philox(oracle):
    """The device's counter-based table equals the host definition (both long-permutation swap forms), and
    lees_l(shared_permutations=True, rng="philox") scores exactly those rows."""
    from spatialcore_amd import _lib
    from spatialcore_amd.spatial import lees_l

    ctx = _lib.default_context(0)
    for n, P, p0 in [(1, 3, 0), (2, 5, 1), (1000, 9, 4), (70001, 6, 2**33 + 1)]:
        got = ctx.generate_permutations_counter(123, n, P, p_first=p0, fetch=True)
        np.testing.assert_array_equal(got, _lib.perm_counter_host(123, n, P, p_first=p0))
    # ... and the ORACLE's restatement of the definition directly (numpy Philox4x32-10 pinned by Random123's known answers,
    # python-integer Lemire): the workgroup swap form (n >= 65536) and the wavefront form, rows far apart in p
    for n, p0 in [(70001, 0), (70001, 2**33 + 5), (3001, 7)]:
        got = ctx.generate_permutations_counter(123, n, 2, p_first=p0, fetch=True)
        for r in range(2):
            np.testing.assert_array_equal(got[r], oracle.counter_permutation(123, n, p0 + r), err_msg=f"n={n} p={p0 + r}")
    n, G, P = 4000, 6, 19
    coords, X = synth(n, G, 9, dtype=np.float64, sparse_x=False)
    ad = make_adata(coords, X)
    pairs = [("g0", "g3"), ("g1", "g4"), ("g2", "g5"), ("g0", "g5")]
    res = lees_l(ad, pairs, n_neighbors=6, n_permutations=P, seed=31, shared_permutations=True, rng="philox")
    perms = _lib.perm_counter_host(31, n, P)
    W = oracle.reference_weights(coords, 6).astype(np.float64)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    for r, (a, b) in zip(res, [(0, 3), (1, 4), (2, 5), (0, 5)]):
        L = float(Z[:, a] @ (W @ Z[:, b]))
        Lp = np.array([float(Z[:, a] @ (W @ Z[perms[p], b])) for p in range(P)])
        assert r["L"] == pytest.approx(L, rel=1e-9)
        assert r["p_value"] == ((np.abs(Lp) >= abs(L)).sum() + 1) / (P + 1)
    with pytest.raises(ValueError, match="rng must be"):
        lees_l(ad, pairs, n_permutations=3, rng="philox")           # per-pair permutations follow numpy's stream
