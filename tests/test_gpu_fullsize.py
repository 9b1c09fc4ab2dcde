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
    from spatialcore_amd._lib import Context

    rng = np.random.default_rng(42)
    coords = rng.uniform(0, np.sqrt(N) * 10.0, (N, 2))
    G = 24
    lam = np.exp(rng.uniform(np.log(0.05), np.log(5.0), G))
    X = rng.poisson(lam, (N, G)).astype(np.float32)
    X[:, ::2] += (2.0 * (1 + np.sin(coords[:, :1] / 900.0))).astype(np.float32)   # spatially smooth genes
    with Context(0) as ctx:
        yield ctx, coords, X


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
