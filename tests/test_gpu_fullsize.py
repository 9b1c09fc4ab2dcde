"""GPU: BASELINE.json's full problem size (1M cells, k=15) through size-independent properties.

The oracle cannot run the whole workload in seconds, so these tests use invariants of the domain:
permutation-table validity + spot rows against the host generator, identity / inverse permutation,
affine invariance of Moran's I, agreement of a slice with the oracle, kNN spot checks against a tree.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1_000_000
K = 15


@pytest.fixture(scope="module")
def big():
    from spatialcore_amd import _lib

    rng = np.random.default_rng(42)
    coords = rng.uniform(0, np.sqrt(N) * 10.0, (N, 2))
    G = 24
    lam = np.exp(rng.uniform(np.log(0.05), np.log(5.0), G))
    X = rng.poisson(lam, (N, G)).astype(np.float32)
    X[:, ::2] += (2.0 * (1 + np.sin(coords[:, :1] / 900.0))).astype(np.float32)   # spatially smooth genes
    yield _lib.default_context(0), coords, X   # (one context per process: see tests/test_gpu_kernels.py::ctx)


def test_knn_1m_spot_check_against_tree(big, oracle):
    from scipy.spatial import cKDTree

    ctx, coords, _ = big
    idx = ctx.knn(coords, K)
    assert idx.shape == (N, K) and idx.min() >= 0 and idx.max() < N
    assert (idx != np.arange(N)[:, None]).all()                       # self excluded
    rows = np.random.default_rng(0).choice(N, 3000, replace=False)
    _, nb = cKDTree(coords).query(coords[rows], k=K + 1)
    np.testing.assert_array_equal(idx[rows], nb[:, 1:])               # bit-exact on the sampled rows


def test_permutation_table_1m(big, oracle):
    from spatialcore_amd import _lib

    ctx, _, _ = big
    P = 40
    w = _lib.rng_state_words(np.random.default_rng(0))
    before = ctx.permgen_stats()
    perms = ctx.generate_permutations(w, N, P, fetch=True)
    par, seq, fallbacks, prepared, chained = (a - b for a, b in zip(ctx.permgen_stats(), before))
    assert (par, seq, fallbacks) == (1, 0, 0) and prepared > 3 * chained > 0       # block-parallel scan, verified
    # every row is a permutation of 0..N-1 (checksum of checksums: sum and sum of squares + sort of one row)
    s1 = perms.astype(np.int64).sum(axis=1)
    assert (s1 == N * (N - 1) // 2).all()
    assert (np.sort(perms[P - 1]) == np.arange(N)).all()
    # rows 0 and P-1 and the final generator state against the host generator (numpy-exact, tested on CPU)
    wh = _lib.rng_state_words(np.random.default_rng(0))
    host = _lib.perm_numpy_host(wh, N, P)
    np.testing.assert_array_equal(perms[0], host[0])
    np.testing.assert_array_equal(perms[P - 1], host[P - 1])
    np.testing.assert_array_equal(w, wh)


def test_moran_1m_invariants(big, oracle):
    from spatialcore_amd import _lib

    ctx, coords, X = big
    G = X.shape[1]
    ctx.knn(coords, K, fetch=False)
    ctx.graph_from_knn(1.0 / K)
    ctx.set_expression(X, np.arange(G))
    P = 24
    w = _lib.rng_state_words(np.random.default_rng(3))
    out = ctx.moran_seeded(w, P)
    assert np.isfinite(out["I"]).all() and (np.abs(out["I"]) < 1.0).all()
    assert (out["I"][::2] > 0.05).all()                               # the smooth genes are autocorrelated
    np.testing.assert_array_equal(out["count_ge"], (out["sims"] >= out["I"]).sum(axis=0))
    # permutation null of the i.i.d. genes (odd columns): centred on E[I] = -1/(N-1) with sd ~ sqrt(var_norm).
    # (For smooth genes the row-permutation null is wider: its spread follows var(lag), not var_norm.)
    s0, s1, s2 = ctx.graph_moments()
    var_norm = (N * N * s1 - N * s2 + 3 * s0 * s0) / ((N - 1.0) * (N + 1.0) * s0 * s0) - (1.0 / (N - 1)) ** 2
    zs = (out["sims"][:, 1::2] - (-1.0 / (N - 1))) / np.sqrt(var_norm)
    assert abs(zs.mean()) < 0.3 and 0.7 < zs.std() < 1.3
    # identity permutation reproduces the observed statistic; a table row and its use are consistent
    ident = np.arange(N, dtype=np.int32)[None, :]
    ctx.set_permutations(ident)
    same = ctx.moran(1)
    np.testing.assert_allclose(same["sims"][0], same["I"], rtol=1e-12)
    np.testing.assert_allclose(same["I"], out["I"], rtol=1e-13)
    # affine invariance: I(a x + b) == I(x)
    ctx.set_expression(3.0 * X.astype(np.float64) + 7.0, np.arange(G))
    aff = ctx.moran(1)
    np.testing.assert_allclose(aff["I"], out["I"], rtol=1e-9)
    # a slice against the oracle: 2 genes, observed I and one permutation's statistic
    ctx.set_expression(X, np.arange(G))
    perm1 = np.random.default_rng(5).permutation(N).astype(np.int32)[None, :]
    ctx.set_permutations(perm1)
    got = ctx.moran(1)
    nbr = ctx.knn(coords, K)
    from scipy.sparse import csr_matrix
    g = csr_matrix((np.full(N * K, 1.0 / K), nbr.reshape(-1), np.arange(0, N * K + 1, K)), shape=(N, N))
    g.sort_indices()
    vals = np.ascontiguousarray(X[:, :2].T, dtype=np.float64)
    np.testing.assert_allclose(got["I"][:2], oracle.morans_i_scores(g, vals), rtol=1e-9)
    np.testing.assert_allclose(got["sims"][0, :2], oracle.morans_i_sims_gather(g, vals, perm1)[0], rtol=1e-9)


def test_config1_at_size_bench_path_p1000(big, oracle):
    """BASELINE configs[1] at size, the path bench.py times: 1M cells, 64 genes, k = 15, P = 1000 through
    sc_moran_seeded (block-parallel generator, chunk schedule 32+40 / 128 x 7 / 64, inverse-only tables, CU-masked
    scoring stream).  Every statistic bit-equal to the two-step path fed with the HOST generator's table; an oracle
    slice (2 genes: observed I + 3 permutation rows) ties both to the CPU restatement."""
    from scipy.sparse import csr_matrix
    from spatialcore_amd import _lib

    ctx, coords, _ = big
    G, P = 64, 1000
    rng = np.random.default_rng(7)
    lam = np.exp(rng.uniform(np.log(0.05), np.log(5.0), G))
    X = rng.poisson(lam, (N, G)).astype(np.float32)
    X[:, ::2] += (2.0 * (1 + np.sin(coords[:, 1:2] / 700.0))).astype(np.float32)
    nbr = ctx.knn(coords, K)
    ctx.graph_from_knn(1.0 / K)
    ctx.set_expression(X, np.arange(G))
    w = _lib.rng_state_words(np.random.default_rng(0))
    before = ctx.permgen_stats()
    one = ctx.moran_seeded(w, P)
    par, seq, fallbacks = (a - b for a, b in zip(ctx.permgen_stats()[:3], before[:3]))
    assert (par, seq, fallbacks) == (1, 0, 0) and ctx.moran_source_bits() == 32      # fractional values: float32 source
    wh = _lib.rng_state_words(np.random.default_rng(0))
    table = _lib.perm_numpy_host(wh, N, P)
    np.testing.assert_array_equal(w, wh)                                # generator state after 1000 x 1M steps
    ctx.set_permutations(table)
    two = ctx.moran(P)
    for key in ("I", "sims", "count_ge", "sim_sum", "sim_sumsq"):
        np.testing.assert_array_equal(one[key], two[key], err_msg=key)
    np.testing.assert_array_equal(one["count_ge"], (one["sims"] >= one["I"]).sum(axis=0))
    g = csr_matrix((np.full(N * K, 1.0 / K), nbr.reshape(-1), np.arange(0, N * K + 1, K)), shape=(N, N))
    g.sort_indices()
    vals = np.ascontiguousarray(X[:, [0, 63]].T, dtype=np.float64)
    rows = [0, 517, P - 1]
    np.testing.assert_allclose(one["I"][[0, 63]], oracle.morans_i_scores(g, vals), rtol=1e-9)
    np.testing.assert_allclose(one["sims"][rows][:, [0, 63]], oracle.morans_i_sims_gather(g, vals, table[rows]),
                               rtol=1e-9, atol=1e-13)


def test_pipeline_is_exact_next_to_other_kernels_at_bench_size(big):
    """Regression for the r02 finding (profiles/r02_gpu_sharing_raw_stream_corruption.txt): the raw-stream kernel's
    64-bit variable shifts came out wrong for whole wavefronts whenever kernels of other hardware queues ran beside it
    (here: the kNN / graph build enqueued right before, the lag and moments kernels, the scoring kernel) -- at bench
    size every repetition of the r01 pipeline drew dozens of non-numpy permutations.  With the 32-bit funnel-shift
    form each repetition must equal the two-step path fed with the HOST generator's table in every bit, state included."""
    from spatialcore_amd import _lib

    ctx, coords, _ = big
    G, P = 160, 1000
    X = np.random.default_rng(11).poisson(1.0, (N, G)).astype(np.float32)
    ctx.knn(coords, K, fetch=False)
    ctx.graph_from_knn(1.0 / K)
    ctx.set_expression(X, np.arange(G))
    wh = _lib.rng_state_words(np.random.default_rng(0))
    ctx.set_permutations(_lib.perm_numpy_host(wh, N, P))
    ref = ctx.moran(P)
    for rep in range(3):
        ctx.knn(coords, K, fetch=False)             # asynchronous: still running when the generator starts
        ctx.graph_from_knn(1.0 / K)
        w = _lib.rng_state_words(np.random.default_rng(0))
        out = ctx.moran_seeded(w, P)
        assert ctx.moran_source_bits() == 8                       # small Poisson counts: uint8 source
        np.testing.assert_array_equal(w, wh, err_msg=f"generator state, repetition {rep}")
        bad = np.flatnonzero((out["sims"] != ref["sims"]).any(axis=1))
        assert bad.size == 0, f"repetition {rep}: {bad.size} permutations differ, first {bad[:5].tolist()}"
        np.testing.assert_array_equal(out["count_ge"], ref["count_ge"])


def test_config3_shape_radius_graph_and_lee_pairs(big, oracle):
    """BASELINE configs[2] shape: 1M cells, radius graph r = 30 um, Lee's L for 100 x 100 gene pairs
    (observed statistic; the reference has no radius option for Lee, so this is the kernel-level path)."""
    from scipy.spatial import cKDTree
    from scipy.sparse import csr_matrix

    ctx, coords, _ = big
    indptr, indices = ctx.radius_graph(coords, 30.0)
    deg = np.diff(indptr)
    assert indptr[0] == 0 and indptr[-1] == indices.size and 20 < deg.mean() < 36          # ~ pi * 30^2 / 100
    rows = np.random.default_rng(1).choice(N, 2000, replace=False)
    tree = cKDTree(coords)
    for i in rows[:300]:
        want = np.array(sorted(j for j in tree.query_ball_point(coords[i], 30.0) if j != i), dtype=np.int32)
        np.testing.assert_array_equal(indices[indptr[i]:indptr[i + 1]], want)
    assert (deg > 0).all()
    w = 1.0 / np.repeat(deg, deg)
    ctx.set_graph_csr(indptr, indices, w, N)
    rng = np.random.default_rng(2)
    G = 20                                                   # 10 x 10 distinct genes -> 100 ordered pairs
    X = rng.poisson(rng.uniform(0.2, 3.0, G), (N, G)).astype(np.float32)
    ctx.set_expression(X, np.arange(G))
    px, py = np.meshgrid(np.arange(10), np.arange(10, 20), indexing="ij")
    out = ctx.lee(px.ravel(), py.ravel(), None, 0)
    L = out["L"].reshape(10, 10)
    # definition on a sample of pairs: L = sum_i zx_i * (W zy)_i with population-std z-scores
    W = csr_matrix((w, indices, indptr), shape=(N, N))
    Z = (X.astype(np.float64) - X.mean(axis=0, dtype=np.float64)) / X.astype(np.float64).std(axis=0)
    for a, b in [(0, 10), (3, 17), (9, 19), (5, 12)]:
        want = float(Z[:, a] @ (W @ Z[:, b]))
        assert L[a, b - 10] == pytest.approx(want, rel=1e-9, abs=1e-6)
    # independent Poisson genes: L / N is a correlation-like quantity near 0
    assert np.abs(L / N).max() < 0.01


def test_config5_shape_k30_composition(big):
    """BASELINE configs[4] shape: 1M cells, k = 30 neighbourhood composition (the reference's
    compute_neighborhood_profile counting step; label-permutation enrichment is not in the reference)."""
    from scipy.spatial import cKDTree

    ctx, coords, _ = big
    T = 20
    labels = np.random.default_rng(3).choice(T, N, p=np.random.default_rng(4).dirichlet(np.ones(T))).astype(np.int32)
    nbr = ctx.knn(coords, 30)
    ctx.graph_from_knn(1.0)
    cnt = ctx.profile_counts(labels, T)
    assert cnt.dtype == np.float32 and (cnt.sum(axis=1) == 30).all()
    rows = np.random.default_rng(5).choice(N, 2000, replace=False)
    _, nb = cKDTree(coords).query(coords[rows], k=31)
    np.testing.assert_array_equal(nbr[rows], nb[:, 1:])
    want = np.stack([np.bincount(labels[nb[r, 1:]], minlength=T) for r in range(len(rows))]).astype(np.float32)
    np.testing.assert_array_equal(cnt[rows], want)


def test_config4_shape_5m_cells_indexing():
    """BASELINE configs[3] per-GPU shape reduced in genes/permutations: 5M cells -- exercises 64-bit
    offsets in tiles, permutation table, generator scratch and the pipelined scoring path."""
    from scipy.spatial import cKDTree

    from spatialcore_amd import _lib

    n = 5_000_000
    rng = np.random.default_rng(11)
    coords = rng.uniform(0, np.sqrt(n) * 10.0, (n, 2))
    G, P = 20, 12
    X = rng.poisson(1.5, (n, G)).astype(np.float32)
    with _lib.Context(0) as ctx:
        idx = ctx.knn(coords, K)
        rows = rng.choice(n, 1000, replace=False)
        _, nb = cKDTree(coords).query(coords[rows], k=K + 1)
        np.testing.assert_array_equal(idx[rows], nb[:, 1:])
        ctx.graph_from_knn(1.0 / K)
        ctx.set_expression(X, np.arange(G))
        w = _lib.rng_state_words(np.random.default_rng(0))
        out = ctx.moran_seeded(w, P)
        par, seq, fallbacks, prepared, chained = ctx.permgen_stats()
        assert (par, seq, fallbacks) == (1, 0, 0) and prepared > 10 * chained > 0   # block-parallel scan, verified
        wh = _lib.rng_state_words(np.random.default_rng(0))
        host = _lib.perm_numpy_host(wh, n, P)                     # numpy-exact reference for the whole table
        np.testing.assert_array_equal(w, wh)
        np.testing.assert_array_equal(out["count_ge"], (out["sims"] >= out["I"]).sum(axis=0))
        # the resident table equals the host table: score the host table explicitly and compare
        ctx.set_permutations(host)
        again = ctx.moran(P)
        np.testing.assert_array_equal(again["sims"], out["sims"])
        np.testing.assert_array_equal(again["I"], out["I"])
        assert np.abs(out["I"]).max() < 0.01                      # i.i.d. genes
        del host
        # the bench's chunk schedule (P > 384) at 5M cells: the pipeline against the host generator's state after
        # 400 x 5M Fisher-Yates steps and against the two-step path on the device-resident table (bit-equal)
        P2 = 400
        w = _lib.rng_state_words(np.random.default_rng(3))
        one = ctx.moran_seeded(w, P2)
        assert ctx.permgen_stats()[2] == 0 and ctx.moran_source_bits() == 8
        wh = _lib.rng_state_words(np.random.default_rng(3))
        last = _lib.perm_numpy_host(wh, n, P2)[P2 - 1].copy()
        np.testing.assert_array_equal(w, wh)
        two = ctx.moran(P2)                                       # the table the pipeline left (inverse rows), re-scored
        for key in ("sims", "count_ge", "sim_sum"):
            np.testing.assert_array_equal(one[key], two[key], err_msg=key)
        ctx.set_permutations(last[None, :])                       # the last row, scored from the host's table
        np.testing.assert_array_equal(ctx.moran(1)["sims"][0], one["sims"][P2 - 1])
        assert ctx.device_mem() < 120 * 2**30


def _radius_graph_1m(ctx, coords):
    """BASELINE configs[2] graph: closed-ball radius graph r = 30 um at 1M cells (~28 neighbours), float32-valued
    1 / degree weights as the in-repo Lee path uses; returns the scipy matrix of the same values."""
    from scipy.sparse import csr_matrix

    indptr, indices = ctx.radius_graph(coords, 30.0)
    deg = np.diff(indptr)
    assert (deg > 0).all() and 20 < deg.mean() < 36
    w = np.repeat((np.float32(1.0) / deg.astype(np.float32)).astype(np.float64), deg)
    ctx.set_graph_csr(indptr, indices, w, N)
    return csr_matrix((w, indices, indptr), shape=(N, N))


def _zscores(X):
    X = X.astype(np.float64)
    return (X - X.mean(axis=0)) / X.std(axis=0)


def test_config2_at_size_shared_permutation_grid_100x100(big, oracle):
    """BASELINE configs[2] AT SIZE through the code it runs: 1M cells, radius graph r = 30 um, Lee's L for 100 x 100
    gene pairs x 199 permutations with `shared_permutations` semantics (sc_lee_shared: k_lee_observed_mfma +
    k_lee_shared_mfma).  One pair goes through the oracle's literal restatement of the reference's core loop
    (oracle.lees_l_core: numpy's own rng.permutation + scipy mat-vec per permutation) over all 199 permutations; six
    more pairs through the definition on three rows of the HOST generator's table; the grid's counts are those of its
    own L_perm; the generator ends in numpy's state."""
    from spatialcore_amd import _lib

    ctx, coords, _ = big
    W = _radius_graph_1m(ctx, coords)
    rng = np.random.default_rng(21)
    G, P, seed = 200, 199, 5
    X = rng.poisson(rng.uniform(0.2, 3.0, G), (N, G)).astype(np.float32)
    smooth = (2.0 * (1 + np.sin(coords[:, 0] / 900.0))).astype(np.float32)
    X[:, 0:30] += smooth[:, None]                                # genes 0..29 and 100..129 share a spatially smooth part:
    X[:, 100:130] += smooth[:, None]                             # truly associated pairs among the independent ones
    ctx.set_expression(X, np.arange(G))
    gx, gy = np.arange(100), np.arange(100, 200)
    w = _lib.rng_state_words(np.random.default_rng(seed))
    out = ctx.lee_shared(w, gx, gy, P, return_perms=True)
    wh = _lib.rng_state_words(np.random.default_rng(seed))
    host = _lib.perm_numpy_host(wh, N, P)
    np.testing.assert_array_equal(w, wh)                         # generator state after 199 x 1M Fisher-Yates steps
    L, Lp, cnt = out["L"], out["L_perm"], out["count_abs_ge"]
    assert L.shape == (100, 100) and Lp.shape == (P, 100, 100)
    np.testing.assert_array_equal(cnt, (np.abs(Lp) >= np.abs(L)[None]).sum(axis=0))
    assert (cnt[:30, :30] == 0).all() and (L[:30, :30] > 1e4).all()   # the associated pairs: no permutation reaches L
    assert 0.3 < (cnt[40:, 40:] / P).mean() < 0.7                # independent pairs: |L_perm| >= |L| about half the time
    Z = _zscores(X[:, [0, 3, 17, 42, 99, 100, 103, 117, 160, 199]])
    zcol = {g: Z[:, i] for i, g in enumerate([0, 3, 17, 42, 99, 100, 103, 117, 160, 199])}
    # (1) the reference's own loop, restated literally, for one pair over all permutations
    _, L_ref, _, p_ref, Lp_ref = oracle.lees_l_core(zcol[3], zcol[103], W, P, np.random.default_rng(seed))
    assert L[3, 3] == pytest.approx(L_ref, rel=1e-9)
    np.testing.assert_allclose(Lp[:, 3, 3], Lp_ref, rtol=1e-9, atol=1e-7)
    assert (cnt[3, 3] + 1) / (P + 1) == p_ref
    # (2) the definition on the host generator's rows for other pairs: L_perm = sum_i zx_i (W zy[perm])_i
    for a, b in [(0, 100), (17, 117), (42, 160), (99, 199), (0, 199), (42, 103)]:
        assert L[a, b - 100] == pytest.approx(float(zcol[a] @ (W @ zcol[b])), rel=1e-9, abs=1e-6)
        for p in (0, 57, P - 1):
            want = float(zcol[a] @ (W @ zcol[b][host[p]]))
            assert Lp[p, a, b - 100] == pytest.approx(want, rel=1e-9, abs=1e-6)


def test_config2_at_size_per_pair_permutations_lee_seeded(big, oracle):
    """The reference's semantics at size (AC:1109-1148: a fresh block of P permutations per pair from ONE stream):
    sc_lee_seeded for 24 pairs x 199 permutations at 1M cells on the radius graph -- k_lee_observed_mfma + pipelined
    k_lee_rows -- against (a) the per-pair path (sc_perm_generate of all 24 x 199 rows + sc_lee) on the same stream:
    same counts, same statistics to summation order, same generator state; (b) the oracle's literal loop for the first
    pair (it owns the first block of the stream)."""
    from spatialcore_amd import _lib

    ctx, coords, _ = big
    W = _radius_graph_1m(ctx, coords)
    rng = np.random.default_rng(22)
    G, P, seed = 12, 199, 9
    X = rng.poisson(rng.uniform(0.2, 3.0, G), (N, G)).astype(np.float32)
    X[:, 7] = 2.0                                                # a zero-variance gene: its pairs draw nothing
    ctx.set_expression(X, np.arange(G))
    pairs = np.array([(a, b) for a in range(4) for b in range(4, 10)])       # 24 pairs, 4 of them with gene 7
    live = (pairs != 7).all(axis=1)
    w = _lib.rng_state_words(np.random.default_rng(seed))
    before = ctx.permgen_stats()
    out = ctx.lee_seeded(w, pairs[:, 0], pairs[:, 1], P, return_perms=True)
    assert ctx.permgen_stats()[2] == before[2]                   # no verification fallback
    w2 = _lib.rng_state_words(np.random.default_rng(seed))
    ctx.generate_permutations(w2, N, int(live.sum()) * P)
    np.testing.assert_array_equal(w, w2)
    off = np.where(live, np.cumsum(live) - 1, -1) * P
    off[~live] = -1
    ref = ctx.lee(pairs[:, 0], pairs[:, 1], off, P, return_perms=True)
    np.testing.assert_allclose(out["L"], ref["L"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(out["L_perm"][live], ref["L_perm"][live], rtol=1e-9, atol=1e-7)
    np.testing.assert_array_equal(out["count_abs_ge"][live], ref["count_abs_ge"][live])
    np.testing.assert_array_equal(out["count_abs_ge"][~live], P)
    Z = _zscores(X[:, [0, 4]])
    _, L_ref, _, p_ref, Lp_ref = oracle.lees_l_core(Z[:, 0], Z[:, 1], W, P, np.random.default_rng(seed))
    assert out["L"][0] == pytest.approx(L_ref, rel=1e-9)
    np.testing.assert_allclose(out["L_perm"][0], Lp_ref, rtol=1e-9, atol=1e-7)
    assert (out["count_abs_ge"][0] + 1) / (P + 1) == p_ref


def test_config4_at_size_enrichment_k30(big, oracle):
    """BASELINE configs[4] AT SIZE through k_enrich: 1M cells, k = 30 neighbour graph, 20 cell types, 600 label
    permutations through the public neighborhood_enrichment (two generator batches), and the raw T x T count tables
    of the observed labels and of three sampled permutations against the oracle's restatement on the host generator's
    rows (exact integers)."""
    from scipy.spatial import cKDTree

    from conftest import make_adata
    from spatialcore_amd import _lib
    from spatialcore_amd.spatial import neighborhood_enrichment

    ctx, coords, _ = big
    T, P, k, seed = 20, 600, 30, 13
    codes = np.random.default_rng(3).choice(T, N, p=np.random.default_rng(4).dirichlet(np.ones(T))).astype(np.int32)
    nbr = ctx.knn(coords, k)
    ctx.graph_from_knn(1.0)
    rows = np.random.default_rng(5).choice(N, 1000, replace=False)
    _, nb = cKDTree(coords).query(coords[rows], k=k + 1)
    np.testing.assert_array_equal(nbr[rows], nb[:, 1:])
    w = _lib.rng_state_words(np.random.default_rng(seed))
    ctx.generate_permutations(w, N, 512)
    cnt = ctx.enrichment_counts(codes, T, 512)
    wh = _lib.rng_state_words(np.random.default_rng(seed))
    host = _lib.perm_numpy_host(wh, N, 512)
    np.testing.assert_array_equal(w, wh)
    assert cnt.shape == (513, T, T) and (cnt.sum(axis=(1, 2)) == N * k).all()
    indptr = np.arange(0, N * k + 1, k, dtype=np.int64)
    picks = [0, 255, 511]
    want = oracle.enrichment_counts(indptr, np.sort(nbr, axis=1).reshape(-1), codes, T, host[picks])
    np.testing.assert_array_equal(cnt[picks], want[:3])
    np.testing.assert_array_equal(cnt[512], want[3])             # observed labels
    # the public function (its own graph build, 512 + 88 permutations from one stream)
    labels = np.array([f"type{c:02d}" for c in range(T)])[codes]
    ad = make_adata(coords, np.zeros((N, 1), dtype=np.float32), labels=labels)
    neighborhood_enrichment(ad, "cell_type", k=k, n_permutations=P, seed=seed)
    res = ad.uns["neighborhood_enrichment"]
    np.testing.assert_array_equal(res["count"], want[3])
    assert res["p_value"].shape == (T, T) and res["p_value"].min() >= 1 / (P + 1) and res["p_value"].max() <= 1.0
    first = (cnt[:512] >= cnt[512]).sum(axis=0)                  # the first batch's share of the exceedance counts
    ge = np.rint(res["p_value"] * (P + 1) - 1).astype(np.int64)
    assert (ge >= first).all() and (ge <= first + (P - 512)).all()
    with np.errstate(invalid="ignore"):
        assert np.nanmax(np.abs(res["zscore"])) < 8              # labels are independent of position: no enrichment


def test_config3_one_ranks_full_share_5m_cells_250_genes_p1000(oracle):
    """BASELINE configs[3] as ONE of the 8 ranks runs it (r04): 5M cells, its 250 of the 2000 genes,
    all 1000 numpy-exact permutations, in one sc_moran_seeded call (124-133 GB of the 288).  Checked: block-parallel
    generator without fallback; final generator state == the host generator's after 1000 x 5M Fisher-Yates steps; the
    last permutation re-scored from the HOST generator's row bit-equal to the pipeline's; observed I and the last
    permutation's statistic of two genes against the oracle (CSR sweep / gather form on the device's own -- elsewhere
    tree-checked -- neighbour lists); counts self-consistent; memory bound."""
    from scipy.sparse import csr_matrix

    from spatialcore_amd import _lib

    n, G, P = 5_000_000, 250, 1000
    rng = np.random.default_rng(21)
    coords = rng.uniform(0, np.sqrt(n) * 10.0, (n, 2))
    X = rng.integers(0, 4, (n, G), dtype=np.uint8).astype(np.float32)      # counts 0 .. 3 (uint8 source, lattice genes)
    with _lib.Context(0) as ctx:
        nbr = ctx.knn(coords, K)
        ctx.graph_from_knn(1.0 / K)
        ctx.set_expression(X, np.arange(G))
        w = _lib.rng_state_words(np.random.default_rng(8))
        out = ctx.moran_seeded(w, P)
        par, seq, fallbacks, prepared, chained = ctx.permgen_stats()
        assert (par, seq, fallbacks) == (1, 0, 0) and prepared > 5 * chained > 0
        assert ctx.moran_source_bits() == 8 and ctx.moran_lag_bits() == 16 and ctx.moran_row_groups() == 2
        assert ctx.device_mem() < 150 * 2**30
        wh = _lib.rng_state_words(np.random.default_rng(8))
        last = None
        for _ in range(10):                                        # the host generator in ten pieces of 100 x 5M (2 GB each)
            last = _lib.perm_numpy_host(wh, n, P // 10)[-1].copy()
        np.testing.assert_array_equal(w, wh)                      # state after P numpy permutations of 5M
        np.testing.assert_array_equal(out["count_ge"], (out["sims"] >= out["I"]).sum(axis=0))
        ctx.set_permutations(last[None, :])
        np.testing.assert_array_equal(ctx.moran(1)["sims"][0], out["sims"][P - 1])
        # two genes against the oracle: I (literal CSR sweep) and the last permutation's statistic (gather form)
        cols = [0, 249]
        g = csr_matrix((np.full(n * K, 1.0 / K), np.sort(nbr, axis=1).reshape(-1), np.arange(0, n * K + 1, K)), shape=(n, n))
        vals = np.ascontiguousarray(X[:, cols].T, dtype=np.float64)
        np.testing.assert_allclose(out["I"][cols], oracle.morans_i_scores(g, vals), rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(out["sims"][P - 1, cols], oracle.morans_i_sims_gather(g, vals, last[None, :])[0], rtol=1e-9, atol=1e-13)
        assert np.abs(out["I"]).max() < 0.01                      # i.i.d. genes


def test_config4_at_size_philox_enrichment_2048_permutations(big, oracle):
    """BASELINE configs[4]'s 8-GPU form at size (r04): 1M cells, k = 30, 20 types, the COUNTER-BASED
    label permutations (rng="philox": permutation p a pure function of (seed, p), so ranks take disjoint ranges of p and
    add integer sums).  2048 permutations through the public function == the sums of two disjoint halves (what two ranks
    would all-reduce), and the T x T tables of two single permutations against the oracle's restatement of the definition
    (numpy Philox4x32-10 pinned by Random123's known answers, python-integer Lemire) -- the workgroup swap form at 1M."""
    from conftest import make_adata
    from spatialcore_amd.spatial import neighborhood_enrichment

    ctx, coords, _ = big
    T, P, k, seed = 20, 2048, 30, 99
    codes = np.random.default_rng(3).choice(T, N, p=np.random.default_rng(4).dirichlet(np.ones(T))).astype(np.int32)
    nbr = ctx.knn(coords, k)
    ctx.graph_from_knn(1.0)
    obs, sums = ctx.enrichment_counter(codes, T, seed, 0, P)
    oa, sa = ctx.enrichment_counter(codes, T, seed, 0, P // 2)
    ob, sb = ctx.enrichment_counter(codes, T, seed, P // 2, P // 2)
    np.testing.assert_array_equal(obs, oa)
    np.testing.assert_array_equal(obs, ob)
    np.testing.assert_array_equal(sums, sa + sb)                 # two ranks' integer sums == one rank's
    assert obs.sum() == N * k
    indptr = np.arange(0, N * k + 1, k, dtype=np.int64)
    cols = np.sort(nbr, axis=1).reshape(-1)
    for p in (0, 1777):
        _, one = ctx.enrichment_counter(codes, T, seed, p, 1)
        perm = oracle.counter_permutation(seed, N, p)
        want = oracle.enrichment_counts(indptr, cols, codes, T, perm[None, :])
        np.testing.assert_array_equal(obs, want[1])
        np.testing.assert_array_equal(obs + one[0], want[0])      # sums[0] = count - observed of that one permutation
        np.testing.assert_array_equal(one[2], (want[0] >= want[1]).astype(np.int64))
    labels = np.array([f"type{c:02d}" for c in range(T)])[codes]
    ad = make_adata(coords, np.zeros((N, 1), dtype=np.float32), labels=labels)
    neighborhood_enrichment(ad, "cell_type", k=k, n_permutations=P, seed=seed, rng="philox")
    res = ad.uns["neighborhood_enrichment"]
    np.testing.assert_array_equal(res["count"], obs)
    np.testing.assert_array_equal(res["p_value"], (sums[2] + 1) / (P + 1))
    np.testing.assert_allclose(res["mean"], obs + sums[0] / P, rtol=1e-12)
    assert ad.uns["spatialcore_metadata"]["operations"][-1]["parameters"]["permgen_form"] == "counter-based (philox)"
