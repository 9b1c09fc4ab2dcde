"""CPU: pin the oracle (oracle/oracle.py + oracle_c.c) against the committed golden vectors.

rng_kat.npz comes from numpy itself; ref_*.npz from the reference's own in-repo functions
(oracle/make_golden.py).  Integer results must be bit-exact.
"""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

from conftest import load_golden


def test_rng_kat_c_and_python_model(oracle):
    kat = load_golden("rng_kat.npz")
    for ci in range(int(kat["n_cases"])):
        seed, n = int(kat[f"case{ci}_seed"]), int(kat[f"case{ci}_n"])
        key = f"case{ci}_perms"
        reps = kat[key].shape[0] if key in kat else kat[f"case{ci}_head"].shape[0]
        perms, words = oracle.perm_table(seed, n, reps)
        if key in kat:
            np.testing.assert_array_equal(perms, kat[key])
        else:
            w = np.arange(1, n + 1, dtype=np.uint64)
            chk = np.array([(p.astype(np.uint64) * w).sum() for p in perms], dtype=np.uint64)
            np.testing.assert_array_equal(chk, kat[f"case{ci}_checksum"])
            np.testing.assert_array_equal(perms[:, :16], kat[f"case{ci}_head"])
            np.testing.assert_array_equal(perms[:, -16:], kat[f"case{ci}_tail"])
        np.testing.assert_array_equal(words, kat[f"case{ci}_final_state"])
        if n <= 1000:  # pure-Python model, small cases
            m = oracle.PCG64Model.from_generator(np.random.default_rng(seed))
            for r in range(reps):
                np.testing.assert_array_equal(m.permutation(n), perms[r])


def test_rng_mid_word_and_value_permutation(oracle):
    kat = load_golden("rng_kat.npz")
    assert int(kat["mid_state"][4]) == 1  # starts with a buffered 32-bit half
    vals = kat["mid_vals"]
    perms, _ = oracle.perm_table(kat["mid_state"], vals.size, 3)
    # rng.permutation(values) == values[rng.permutation(n)] on the same stream
    np.testing.assert_array_equal(vals[perms], kat["mid_perm_vals"])


def test_rng_raw_stream(oracle):
    kat = load_golden("rng_kat.npz")
    np.testing.assert_array_equal(oracle.raw_uint32(kat["raw_state"], 33), kat["raw_u32"])


def test_live_numpy_agrees(oracle):
    """Same check against the numpy installed wherever the test runs."""
    for seed, n, reps in [(0, 5000, 3), (31337, 77, 9)]:
        rng = np.random.default_rng(seed)
        want = np.stack([rng.permutation(n) for _ in range(reps)])
        got, words = oracle.perm_table(seed, n, reps)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(words, oracle.rng_state_words(rng))


def test_reference_weights(oracle):
    g = load_golden("ref_weights.npz")
    for ci in range(int(g["n_cases"])):
        W = oracle.reference_weights(g[f"c{ci}_coords"], int(g[f"c{ci}_k"]), bool(g[f"c{ci}_include_self"]))
        assert str(W.dtype) == str(g[f"c{ci}_dtype"]) == "float32"
        np.testing.assert_array_equal(W.indptr, g[f"c{ci}_W_indptr"])
        np.testing.assert_array_equal(W.indices, g[f"c{ci}_W_indices"])   # bit-exact neighbours
        np.testing.assert_array_equal(W.data, g[f"c{ci}_W_data"])


def test_knn_bruteforce_matches_trees(oracle):
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 100, (3000, 2))
    a = oracle.knn_bruteforce(xy, 15)
    np.testing.assert_array_equal(a, oracle.knn_tree(xy, 15))
    from sklearn.neighbors import NearestNeighbors

    for algo in ("ball_tree", "kd_tree"):
        nn = NearestNeighbors(n_neighbors=15, algorithm=algo).fit(xy)
        np.testing.assert_array_equal(a, nn.kneighbors()[1])


def test_lees_l_golden(oracle):
    g = load_golden("ref_lees_l.npz")
    for ci in range(int(g["n_cases"])):
        X = g[f"c{ci}_X"]
        res = oracle.lees_l(g[f"c{ci}_coords"], X, [tuple(p) for p in g[f"c{ci}_pairs"]],
                            int(g[f"c{ci}_k"]), int(g[f"c{ci}_P"]), int(g[f"c{ci}_seed"]))
        L = np.array([r["L"] for r in res])
        p = np.array([r["p_value"] for r in res])
        tol = 1e-12 if X.dtype == np.float64 else 1e-5
        np.testing.assert_allclose(L, g[f"c{ci}_L"], rtol=tol, atol=tol)
        np.testing.assert_array_equal(p, g[f"c{ci}_p"])
        assert res[0]["L"] == pytest.approx(float(g[f"c{ci}_single_L"]), rel=tol, abs=tol)


def test_lee_literal_c_equals_numpy(oracle):
    """The C literal loop (shuffle z_y, redo W@z) and the gather form agree with numpy's loop."""
    import ctypes

    g = load_golden("ref_lees_l.npz")
    coords, X = g["c0_coords"], g["c0_X"]
    W = oracle.reference_weights(coords, 6)
    x, y = X[:, 0], X[:, 1]
    zx, zy = (x - x.mean()) / x.std(), (y - y.mean()) / y.std()
    rng = np.random.default_rng(0)
    _, L, _, p, L_perm = oracle.lees_l_core(zx, zy, W, 19, rng)
    words = oracle.rng_state_words(np.random.default_rng(0))
    out = np.empty(19)
    Wd = csr_matrix(W, dtype=np.float64)
    oracle.clib().orc_lee_perm_literal(
        words.ctypes.data_as(ctypes.c_void_p), Wd.indptr.astype(np.int64).ctypes.data_as(ctypes.c_void_p),
        Wd.indices.astype(np.int32).ctypes.data_as(ctypes.c_void_p), Wd.data.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(zx.size), zx.ctypes.data_as(ctypes.c_void_p), zy.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(19), out.ctypes.data_as(ctypes.c_void_p))
    np.testing.assert_allclose(out, L_perm, rtol=1e-10, atol=1e-10)
    np.testing.assert_array_equal(words, oracle.rng_state_words(rng))
    # gather form: sum_j (W^T zx)[j] * zy[perm[j]]
    perms, _ = oracle.perm_table(0, zx.size, 19)
    u = Wd.T @ zx
    np.testing.assert_allclose((u[None, :] * zy[perms]).sum(axis=1), L_perm, rtol=1e-10, atol=1e-10)
    assert L == pytest.approx(float(g["c0_L"][0]), rel=1e-12)


def test_local_morans_golden(oracle):
    g = load_golden("ref_local_morans.npz")
    for ci in range(int(g["n_cases"])):
        X = g[f"c{ci}_X"]
        r = oracle.local_morans_i(g[f"c{ci}_coords"], X, np.arange(X.shape[1]), int(g[f"c{ci}_k"]),
                                  int(g[f"c{ci}_P"]), int(g[f"c{ci}_seed"]), fdr=str(g[f"c{ci}_fdr"]),
                                  alpha=float(g[f"c{ci}_alpha"]), batch_size=int(g[f"c{ci}_batch"]))
        for f in ("I", "z", "lag", "p", "p_adj", "quadrant"):
            np.testing.assert_array_equal(r[f], g[f"c{ci}_{f}"], err_msg=f"case {ci} field {f}")


def test_fdr_quadrants_golden(oracle):
    g = load_golden("ref_fdr_quadrants.npz")
    np.testing.assert_array_equal(oracle.fdr_bh(g["p"]), g["bh"])
    np.testing.assert_array_equal(oracle.bonferroni(g["p"]), g["bonf"])
    np.testing.assert_array_equal(oracle.quadrants(g["z"], g["lag"], g["pq"], 0.05), g["quad_sig"])
    np.testing.assert_array_equal(oracle.quadrants(g["z"], g["lag"]), g["quad_nosig"])


def test_profile_golden(oracle):
    g = load_golden("ref_profile.npz")
    for name in ("knn", "knn_raw", "radius", "radius_raw"):
        kw = dict(method=str(g[f"{name}_method"]))
        if kw["method"] == "knn":
            kw["k"] = int(g[f"{name}_k"])
        else:
            kw["radius"] = float(g[f"{name}_radius"])
        kw["normalize"] = bool(g[f"{name}_normalize"]) if f"{name}_normalize" in g else True
        err = str(g[f"{name}_error"])
        if err:
            with pytest.raises(ValueError):
                oracle.neighborhood_profile(g["coords"], g["labels"], **kw)
            continue
        prof, cats = oracle.neighborhood_profile(g["coords"], g["labels"], **kw)
        assert cats == list(g[f"{name}_celltypes"])
        np.testing.assert_array_equal(prof, g[f"{name}_profile"])
        assert prof.dtype == np.float32


def test_moran_restatement_cross_checks(oracle):
    """Global Moran's I is 'parity unpinned' (squidpy absent).  Cross-checks available here:
    (i) literal row-permuted form == gather form; (ii) hand-computed rook lattice;
    (iii) N * I_global == sum of the reference's local I on the same graph, rescaled."""
    from conftest import synth

    coords, X = synth(600, 4, 5)
    tab_l = oracle.morans_i_reference_table(coords, X, [0, 1, 2, 3], 6, 7, 3, literal=True)
    tab_g = oracle.morans_i_reference_table(coords, X, [0, 1, 2, 3], 6, 7, 3, literal=False)
    np.testing.assert_allclose(tab_l["sims"], tab_g["sims"], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(tab_l["p_value"], tab_g["p_value"])
    assert ((tab_g["p_value"] >= 1 / 8) & (tab_g["p_value"] <= 1)).all()
    # (ii) 1-D chain 0-1-2-3 with values 1,2,3,4, binary symmetric neighbours, row-normalised
    rows = [0, 1, 1, 2, 2, 3]; cols = [1, 0, 2, 1, 3, 2]
    gmat = oracle.row_normalize_l1(csr_matrix((np.ones(6), (rows, cols)), shape=(4, 4)))
    x = np.array([[1.0, 2.0, 3.0, 4.0]])
    z = x[0] - 2.5
    lag = np.array([z[1], (z[0] + z[2]) / 2, (z[1] + z[3]) / 2, z[2]])
    want = 4 / 4 * (z * lag).sum() / (z * z).sum()
    assert oracle.morans_i_scores(gmat, x)[0] == pytest.approx(want, rel=1e-14)
    s0, s1, s2 = oracle.graph_moments(gmat)
    assert s0 == pytest.approx(4.0)
    # s1 = 1/2 sum (w_ij + w_ji)^2 : pairs (0,1),(1,0): (1+.5)^2 each; (1,2),(2,1): 1; (2,3),(3,2): 1.5^2
    assert s1 == pytest.approx(0.5 * (2 * 2.25 + 2 * 1.0 + 2 * 2.25))
    # (iii) tie to the importable local statistic: with population-std z, sum_i z_i*lag_i / N == I
    gl = load_golden("ref_local_morans.npz")
    Xl = gl["c2_X"].astype(np.float64)
    W = oracle.reference_weights(gl["c2_coords"], int(gl["c2_k"]))
    I_glob = oracle.morans_i_scores(csr_matrix(W, dtype=np.float64), Xl.T.copy())
    local_sum = gl["c2_I"].astype(np.float64).sum(axis=0) / Xl.shape[0]
    np.testing.assert_allclose(I_glob, local_sum, rtol=2e-4, atol=2e-6)  # float32 reference arrays


def test_counter_based_permutation_source_philox_kat_and_host_generator(oracle):
    """The counter-based source (an EXTENSION for paths without reference seed semantics): the oracle's numpy
    Philox4x32-10 reproduces Random123's published known answers, and the library's host generator
    (sc_perm_counter_host, the same code the device kernel compiles) equals the oracle's restatement of the whole
    definition (Philox -> Lemire -> Fisher-Yates) -- including that permutation p does not depend on where a range starts."""
    from spatialcore_amd import _lib

    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = oracle.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want
    for seed, n, p0, cnt in [(0, 10, 0, 4), (12345678901234567, 257, 3, 3), (7, 1, 0, 2), (7, 2, 9, 5), (2**63 + 5, 1000, 2**33, 2)]:
        got = _lib.perm_counter_host(seed, n, cnt, p_first=p0)
        for k in range(cnt):
            np.testing.assert_array_equal(got[k], oracle.counter_permutation(seed, n, p0 + k))
            assert sorted(got[k].tolist()) == list(range(n))
    a = _lib.perm_counter_host(3, 500, 6, p_first=0)
    b = _lib.perm_counter_host(3, 500, 2, p_first=4)
    np.testing.assert_array_equal(a[4:], b)                   # a pure function of (seed, p)
    assert not (a[0] == a[1]).all() and not (a[0] == _lib.perm_counter_host(4, 500, 1)[0]).all()
