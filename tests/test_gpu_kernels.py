"""GPU parity tests (run with -m gpu on an MI355X): every C-ABI entry point vs the CPU oracle.

Integer / index results are bit-exact; fp64 statistics are compared at 1e-9 relative (the contract
from BASELINE.json is 1e-6) -- the only differences are summation order.
"""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

from conftest import load_golden, synth

pytestmark = pytest.mark.gpu


def assert_counts_match(count, tab):
    """#{p : sims >= I} against the oracle's table.  Lattice genes (integer counts on a graph with equal weights,
    DESIGN.md "Ties"): EXACTLY the oracle's count, which is decided on the integers T_p >= T_obs -- exact ties
    included.  Other genes: equal except where a permutation's statistic is within rounding noise of the observed
    one (relative gap <= 1e-11), where any floating-point implementation decides by its summation order."""
    lat, want = tab["lattice"], tab["count_ge"]
    np.testing.assert_array_equal(count[lat], want[lat])
    with np.errstate(invalid="ignore", divide="ignore"):
        near = (np.abs(tab["sims"] - tab["I"]) <= 1e-11 * np.abs(tab["I"])).sum(axis=0)
    assert (np.abs(count - want)[~lat] <= near[~lat]).all(), (count, want, near)


@pytest.fixture(scope="module")
def ctx():
    # the process-wide context the public functions use: ONE context's streams (11) stay below the 16 hardware queues the
    # library asks for; two live contexts' streams would share queues, and the generator then (rightly) refuses its
    # block-parallel form, which some tests below insist on
    from spatialcore_amd import _lib

    c = _lib.default_context(0)
    yield c
    c.set_permgen_mode(0)


def test_library_reports_native_path():
    from spatialcore_amd import _lib

    assert _lib.load_library().sc_version() >= 100
    assert _lib.device_count() >= 1


@pytest.mark.parametrize("n,k", [(2000, 6), (5000, 15), (3000, 30), (400, 1), (700, 64), (50, 49), (3000, 33), (2500, 100),
                                 (1500, 257), (300, 299)])   # k > 32: the global-memory heap form, any k < n
def test_knn_bit_exact(ctx, oracle, n, k):
    rng = np.random.default_rng(n + k)
    xy = rng.uniform(0, np.sqrt(n) * 10, (n, 2))
    idx, rd = ctx.knn(xy, k, return_distance=True)
    want = oracle.knn_bruteforce(xy, k)
    np.testing.assert_array_equal(idx, want)
    d = xy[:, None, :] - xy[want]
    np.testing.assert_array_equal(rd, d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])
    # the same search without waiting, the lists fetched afterwards on the side stream (what morans_i does beside its upload)
    assert ctx.knn(xy, k, fetch=False) is None
    ctx.graph_from_knn(1.0 / k)
    idx2, rd2 = ctx.knn_fetch()
    np.testing.assert_array_equal(idx2, want)
    np.testing.assert_array_equal(rd2, rd)


def test_knn_include_self_and_clustered(ctx, oracle):
    rng = np.random.default_rng(5)
    # strongly non-uniform density + an isolated far-away point: the ring search must still be exact
    xy = np.concatenate([rng.normal(0, 1, (3000, 2)), rng.normal(50, 0.01, (500, 2)), [[1e4, -1e4]]])
    np.testing.assert_array_equal(ctx.knn(xy, 10), oracle.knn_bruteforce(xy, 10))
    np.testing.assert_array_equal(ctx.knn(xy, 7, include_self=True), oracle.knn_bruteforce(xy, 7, include_self=True))


def test_knn_ties_lowest_index(ctx, oracle):
    # integer lattice: exact distance ties; rule = (distance, index) ascending
    g = np.stack(np.meshgrid(np.arange(30.0), np.arange(30.0)), -1).reshape(-1, 2)
    np.testing.assert_array_equal(ctx.knn(g, 8), oracle.knn_bruteforce(g, 8))
    # duplicate coordinates
    d = np.concatenate([g[:100], g[:100]])
    np.testing.assert_array_equal(ctx.knn(d, 5), oracle.knn_bruteforce(d, 5))


def test_knn_degenerate_shapes(ctx, oracle):
    rng = np.random.default_rng(2)
    line = np.stack([rng.uniform(0, 100, 500), np.zeros(500)], 1)   # zero height
    np.testing.assert_array_equal(ctx.knn(line, 4), oracle.knn_bruteforce(line, 4))
    with pytest.raises(ValueError):
        ctx.knn(line[:5], 5)
    with pytest.raises(ValueError):
        ctx.knn(np.array([[0.0, np.nan], [1.0, 1.0]]), 1)


@pytest.mark.parametrize("seed,n,P", [(0, 10, 7), (1, 2, 9), (2, 1, 3), (3, 3, 40), (42, 257, 33), (7, 1000, 64),
                                      (11, 4099, 21), (5, 65537, 9), (123456789, 200000, 12), (9, 1048577, 3)])
def test_device_permutation_stream_is_numpy_exact(ctx, oracle, seed, n, P):
    """sc_perm_generate (parallel device generator) == numpy's default_rng(seed).permutation(n) x P,
    including the generator state it leaves behind."""
    from spatialcore_amd._lib import rng_state_words

    words = rng_state_words(np.random.default_rng(seed))
    got = ctx.generate_permutations(words, n, P, fetch=True)
    want, wwords = oracle.perm_table(seed, n, P)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(words, wwords)
    # continue the same stream: a second block must pick up exactly where the first stopped
    got2 = ctx.generate_permutations(words, n, 2, fetch=True)
    want2, wwords2 = oracle.perm_table(wwords, n, 2)
    np.testing.assert_array_equal(got2, want2)
    np.testing.assert_array_equal(words, wwords2)


@pytest.mark.parametrize("seed,n,P", [(4, 131072, 5), (8, 140001, 300), (15, 300007, 140), (16, 2500000, 4)])
def test_block_parallel_scan_modes_agree_with_numpy(ctx, oracle, seed, n, P):
    """n >= 131072 runs the block-parallel rejection scan (prepared gap-transfer tables + chain + on-device
    verification).  Mode 0 (automatic), 1 (sequential scan) and 2 (injected fault -> verification must catch it
    and the call falls back) all return numpy's table and final state, over several pipeline chunks."""
    from spatialcore_amd._lib import rng_state_words, perm_numpy_host

    wh = rng_state_words(np.random.default_rng(seed))
    want = perm_numpy_host(wh, n, P)
    rows = sorted(set([0, 1, P // 2, P - 2, P - 1]))
    try:
        for mode in (0, 1, 2):
            ctx.set_permgen_mode(mode)
            before = ctx.permgen_stats()
            w = rng_state_words(np.random.default_rng(seed))
            got = ctx.generate_permutations(w, n, P, fetch=True)
            par, seq, fb, prepared, chained = (a - b for a, b in zip(ctx.permgen_stats(), before))
            if mode == 0:
                had_prepared = prepared > 0      # n = 131072: every block holds a band crossing, nothing to corrupt
                assert chained > 0, ctx.permgen_note()
            # mode 0: the block-parallel form ran and passed its verification; 2: it was caught and redone
            want_stats = {0: (1, 0, 0), 1: (0, 1, 0), 2: (0, 1, 1) if had_prepared else (1, 0, 0)}[mode]
            assert (par, seq, fb) == want_stats, (mode, par, seq, fb, prepared, chained)
            for r in rows:
                np.testing.assert_array_equal(got[r], want[r], err_msg=f"mode {mode} row {r}")
            assert (got.astype(np.int64).sum(axis=1) == n * (n - 1) // 2).all()
            np.testing.assert_array_equal(w, wh, err_msg=f"mode {mode} final state")
    finally:
        ctx.set_permgen_mode(0)
    with pytest.raises(ValueError):
        ctx.set_permgen_mode(3)


def test_device_permutation_stream_golden_and_midword(ctx):
    """numpy's own known answers (tests/golden/rng_kat.npz), incl. a generator that starts with a
    buffered 32-bit half."""
    kat = load_golden("rng_kat.npz")
    for ci in range(int(kat["n_cases"])):
        key = f"case{ci}_perms"
        if key not in kat:
            continue
        n, reps = int(kat[f"case{ci}_n"]), kat[key].shape[0]
        words = None
        from spatialcore_amd._lib import rng_state_words
        words = rng_state_words(np.random.default_rng(int(kat[f"case{ci}_seed"])))
        np.testing.assert_array_equal(ctx.generate_permutations(words, n, reps, fetch=True), kat[key])
        np.testing.assert_array_equal(words, kat[f"case{ci}_final_state"])
    words = kat["mid_state"].copy()
    assert int(words[4]) == 1
    perms = ctx.generate_permutations(words, kat["mid_vals"].size, 3, fetch=True)
    np.testing.assert_array_equal(kat["mid_vals"][perms], kat["mid_perm_vals"])


def test_host_permutation_stream(oracle):
    from spatialcore_amd import _lib

    w = _lib.rng_state_words(np.random.default_rng(4))
    got = _lib.perm_numpy_host(w, 5000, 4)
    want, ww = oracle.perm_table(4, 5000, 4)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(w, ww)


def test_reference_weights_golden(ctx):
    """build_spatial_weights goldens from the reference itself: indices bit-exact."""
    g = load_golden("ref_weights.npz")
    for ci in range(int(g["n_cases"])):
        k, inc = int(g[f"c{ci}_k"]), bool(g[f"c{ci}_include_self"])
        ctx.knn(g[f"c{ci}_coords"], k + 1 if inc else k, include_self=inc, fetch=False)
        w32 = np.float32(1.0) / np.float32(k + 1 if inc else k)
        ctx.graph_from_knn(float(w32))
        indptr, indices, data = ctx.get_graph()
        np.testing.assert_array_equal(indptr, g[f"c{ci}_W_indptr"])
        np.testing.assert_array_equal(indices, g[f"c{ci}_W_indices"])
        np.testing.assert_array_equal(data.astype(np.float32), g[f"c{ci}_W_data"])


@pytest.mark.parametrize("radius", [12.0, 30.0, 75.0])   # ~11, ~70 (heap-sorted rows), ~440 neighbours
def test_radius_graph(ctx, oracle, radius):
    rng = np.random.default_rng(9)
    xy = rng.uniform(0, 400, (4000, 2))
    indptr, indices = ctx.radius_graph(xy, radius)
    wp, wi = oracle.radius_neighbors(xy, radius)
    np.testing.assert_array_equal(indptr, wp)
    np.testing.assert_array_equal(indices, wi)


def test_graph_moments(ctx, oracle):
    coords, _ = synth(3000, 1, 3)
    g = oracle.row_normalize_l1(oracle.squidpy_connectivities(coords, 6))
    ctx.set_graph_csr(g.indptr, g.indices, g.data, g.shape[0])
    np.testing.assert_allclose(ctx.graph_moments(), oracle.graph_moments(g), rtol=1e-12)
    # general weighted, partly symmetric graph
    rng = np.random.default_rng(0)
    A = csr_matrix(g)
    A.data = rng.uniform(0.1, 2.0, A.nnz)
    A = (A + A.T.multiply(0.3)).tocsr()
    A.sort_indices()
    ctx.set_graph_csr(A.indptr, A.indices, A.data, A.shape[0])
    np.testing.assert_allclose(ctx.graph_moments(), oracle.graph_moments(A), rtol=1e-12)


def test_graph_rejects_bad_input(ctx):
    with pytest.raises(ValueError):
        ctx.set_graph_csr([0, 1, 2], [0, 5], [1.0, 1.0], 2)       # column out of range
    with pytest.raises(ValueError):
        ctx.set_graph_csr([0, 2, 2], [1, 0], [1.0, 1.0], 2)       # unsorted row


@pytest.mark.parametrize("n,G,k,P,dtype,sparse_x,normalize", [
    (3000, 5, 6, 19, np.float64, True, False),
    (10000, 50, 6, 199, np.float32, True, False),      # BASELINE configs[0]
    (2500, 33, 15, 40, np.float32, False, False),
    (777, 17, 4, 33, np.float64, False, False),        # ragged: n, G, P not multiples of anything
    (10000, 50, 6, 199, np.float32, True, True),       # configs[0] on log-normalised values: no lattice gene,
    (777, 37, 4, 33, np.float32, False, True),         # ... the centred float32-source kernel over whole permutation sets
])
def test_moran_vs_oracle(ctx, oracle, n, G, k, P, dtype, sparse_x, normalize):
    coords, X = synth(n, G, 17, dtype=dtype, sparse_x=sparse_x, normalize=normalize)
    tab = oracle.morans_i_reference_table(coords, X, list(range(G)), k, P, seed=0)
    assert tab["lattice"].any() != normalize
    ctx.knn(coords, k, fetch=False)
    ctx.graph_from_knn(1.0 / k)
    ctx.set_expression(X, np.arange(G))
    from spatialcore_amd._lib import rng_state_words

    words = rng_state_words(np.random.default_rng(0))
    perms = ctx.generate_permutations(words, n, P, fetch=True)
    np.testing.assert_array_equal(perms, tab["perms"])                 # bit-exact numpy stream
    assert (words == oracle.perm_table(0, n, P)[1]).all()
    out = ctx.moran(P)
    assert ctx.moran_source_bits() == (32 if normalize else 8)
    np.testing.assert_allclose(out["I"], tab["I"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["sims"], tab["sims"], rtol=1e-9, atol=1e-13)
    assert_counts_match(out["count_ge"], tab)
    np.testing.assert_array_equal(out["count_ge"], (out["sims"] >= out["I"]).sum(axis=0))   # self-consistent
    np.testing.assert_allclose(out["sim_sum"], tab["sims"].sum(axis=0), rtol=1e-9, atol=1e-12)
    s0, s1, s2 = ctx.graph_moments()
    np.testing.assert_allclose((s0, s1, s2), oracle.graph_moments(tab["graph"]), rtol=1e-12)


@pytest.mark.parametrize("n,G,P", [(5000, 20, 300), (1234, 3, 129), (800, 40, 1)])
def test_moran_seeded_pipeline_equals_two_step(ctx, oracle, n, G, P):
    """The fused generator/scoring pipeline (several 128-permutation chunks on two streams) gives
    exactly what sc_perm_generate + sc_moran give, and leaves the same generator state."""
    from spatialcore_amd._lib import rng_state_words

    coords, X = synth(n, G, 31, dtype=np.float32)
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, np.arange(G))
    w1 = rng_state_words(np.random.default_rng(7))
    ctx.generate_permutations(w1, n, P)
    two = ctx.moran(P)
    w2 = rng_state_words(np.random.default_rng(7))
    one = ctx.moran_seeded(w2, P)
    np.testing.assert_array_equal(w1, w2)
    for key in ("I", "sims", "count_ge", "sim_sum", "sim_sumsq"):
        np.testing.assert_array_equal(one[key], two[key], err_msg=key)
    tab = oracle.morans_i_reference_table(coords, X, list(range(G)), 6, P, seed=7)
    np.testing.assert_allclose(one["sims"], tab["sims"], rtol=1e-9, atol=1e-13)
    assert_counts_match(one["count_ge"], tab)


@pytest.mark.parametrize("n,P", [(3000, 150), (140001, 300)])
def test_moran_seeded_in_two_halves_equals_one_call(ctx, oracle, n, P):
    """sc_moran_seeded_begin needs only n_cells and the generator state: the generator runs while the graph is built and
    the expression uploaded; sc_moran_seeded_finish then gives exactly what the one-call form gives (statistics, counts,
    final generator state), for the sequential and the block-parallel scan.  A begun job can be dropped."""
    from spatialcore_amd._lib import SpatialCoreHipError, rng_state_words

    G = 21
    coords, X = synth(n, G, 12, dtype=np.float32, sparse_x=False)
    ctx.knn(coords, 8, fetch=False)
    ctx.graph_from_knn(1.0 / 8)
    ctx.set_expression(X, np.arange(G))
    w1 = rng_state_words(np.random.default_rng(5))
    one = ctx.moran_seeded(w1, P)
    from spatialcore_amd._lib import Context

    with Context(0) as fresh:                                 # a context that has never seen a graph or an expression
        w2 = rng_state_words(np.random.default_rng(5))
        fresh.moran_seeded_begin(w2, n, P)                    # ... and only now the graph and the expression
        fresh.knn(coords, 8, fetch=False)
        fresh.graph_from_knn(1.0 / 8)
        fresh.set_expression(X, np.arange(G))
        two = fresh.moran_seeded_finish(w2)
    np.testing.assert_array_equal(w1, w2)
    for key in ("I", "sims", "count_ge", "sim_sum", "sim_sumsq"):
        np.testing.assert_array_equal(one[key], two[key], err_msg=key)
    w2 = rng_state_words(np.random.default_rng(5))
    ctx.moran_seeded_begin(w2, n, P)
    ctx.moran_seeded_finish(w2)
    again = ctx.moran(P)                                      # the table the job left is the resident table
    np.testing.assert_array_equal(again["sims"], one["sims"])
    with pytest.raises(SpatialCoreHipError, match="no job begun"):
        ctx.moran_seeded_finish(w2)
    w3 = rng_state_words(np.random.default_rng(6))
    ctx.moran_seeded_begin(w3, n, P)
    ctx.moran_seeded_abort()                                  # dropped: nothing returned, state untouched
    np.testing.assert_array_equal(w3, rng_state_words(np.random.default_rng(6)))
    ctx.moran_seeded_begin(w3, n, P)
    w4 = rng_state_words(np.random.default_rng(5))
    ctx.generate_permutations(w4, n, 3)                       # replaces the table: the begun job is dropped first
    np.testing.assert_array_equal(ctx.moran(3)["sims"], one["sims"][:3])
    ctx.moran_seeded_begin(w3, n + 1, 5)                      # begun for another cell count than the expression
    with pytest.raises(ValueError, match="begun for"):
        ctx.moran_seeded_finish(w3)


@pytest.mark.parametrize("n,G,seed,normalize", [(5000, 70, 0, False), (4096, 33, 9, False), (5000, 70, 0, True)])
def test_moran_seeded_bench_schedule_p1000_vs_oracle(ctx, oracle, n, G, seed, normalize):
    """The exact schedule bench.py runs: P = 1000 > 3 chunks switches sc_moran_seeded to the short-first /
    128 x 7 / short-last chunk bounds, and G = 70 / 33 leaves an odd 16-gene tile count for the 32-gene kernel.
    Every permutation's statistic, the counts and the final generator state against the oracle."""
    from spatialcore_amd._lib import rng_state_words

    P, k = 1000, 15
    coords, X = synth(n, G, 101 + seed, dtype=np.float32, normalize=normalize)
    tab = oracle.morans_i_reference_table(coords, X, list(range(G)), k, P, seed=seed)
    assert tab["lattice"].all() != normalize and tab["lattice"].any() != normalize
    ctx.knn(coords, k, fetch=False)
    ctx.graph_from_knn(1.0 / k)
    ctx.set_expression(X, np.arange(G))
    w = rng_state_words(np.random.default_rng(seed))
    before = ctx.permgen_stats()
    out = ctx.moran_seeded(w, P)
    assert ctx.moran_source_bits() == (32 if normalize else 8)          # small counts: uint8 source, 128 genes per row;
                                                                        # log-normalised values: float32 source, centred
    assert ctx.permgen_stats()[2] == before[2]                         # no silent verification fallback
    np.testing.assert_array_equal(w, oracle.perm_table(seed, n, P)[1])  # generator state after P permutations
    np.testing.assert_allclose(out["I"], tab["I"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["sims"], tab["sims"], rtol=1e-9, atol=1e-13)
    assert_counts_match(out["count_ge"], tab)
    np.testing.assert_array_equal(out["count_ge"], (out["sims"] >= out["I"]).sum(axis=0))
    np.testing.assert_allclose(out["sim_sum"], tab["sims"].sum(axis=0), rtol=1e-9, atol=1e-11)


def test_moran_seeded_bench_schedule_block_parallel_generator(ctx, oracle):
    """Same schedule with n >= 131072: the block-parallel exact scan, the CU-masked scoring stream and the
    inverse-only swap tables all take part (as at 1M cells).  Bit-equal to the two-step path fed with the ORACLE
    generator's table; three genes against the oracle over all 1000 permutations."""
    from spatialcore_amd._lib import perm_numpy_host, rng_state_words

    n, G, P, k = 140001, 33, 1000, 15
    coords, X = synth(n, G, 77, dtype=np.float32, sparse_x=False)
    ctx.knn(coords, k, fetch=False)
    ctx.graph_from_knn(1.0 / k)
    ctx.set_expression(X, np.arange(G))
    w = rng_state_words(np.random.default_rng(4))
    before = ctx.permgen_stats()
    one = ctx.moran_seeded(w, P)
    par, seq, fallbacks = (a - b for a, b in zip(ctx.permgen_stats()[:3], before[:3]))
    assert (par, seq, fallbacks) == (1, 0, 0) and ctx.moran_source_bits() == 8
    # the oracle's own generator (scalar C model of numpy's stream) supplies the table of the two-step run
    cols = [0, 17, 32]
    conn = csr_matrix((np.ones(n * k), oracle.knn_tree(coords, k).reshape(-1), np.arange(0, n * k + 1, k)), shape=(n, n))
    tab = oracle.morans_i_reference_table(coords, X, cols, k, P, seed=4, graph=conn)
    ctx.set_permutations(tab["perms"])
    two = ctx.moran(P)
    for key in ("I", "sims", "count_ge", "sim_sum", "sim_sumsq"):
        np.testing.assert_array_equal(one[key], two[key], err_msg=key)
    wh = rng_state_words(np.random.default_rng(4))
    last = perm_numpy_host(wh, n, P)[P - 1]                  # host generator: last row + state after P permutations
    np.testing.assert_array_equal(last, tab["perms"][P - 1])
    np.testing.assert_array_equal(w, wh)
    np.testing.assert_allclose(one["I"][cols], tab["I"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(one["sims"][:, cols], tab["sims"], rtol=1e-9, atol=1e-13)
    assert_counts_match(one["count_ge"][cols], tab)


@pytest.mark.parametrize("P", [431, 999])
def test_moran_pipeline_odd_chunks_with_two_permutations_per_swap_workgroup(ctx, oracle, monkeypatch, P):
    """Beside the scoring kernel the pipeline's middle chunks are swapped two permutations per workgroup (r04).  P = 431 makes
    the schedule 32, 103, 128, 96, 48, 24 and P = 999 makes it 32, 31, 128 x 6, ...: chunks with an ODD number of permutations, whose last
    workgroup has one idle half.  Bit-equal to the one-permutation form (SC_SWAP_PW=1) and to the two-step path on the host
    generator's table, generator state included."""
    from spatialcore_amd._lib import perm_numpy_host, rng_state_words

    n, G, k = 140001, 9, 15
    coords, X = synth(n, G, 21, dtype=np.float32, sparse_x=False)
    ctx.knn(coords, k, fetch=False)
    ctx.graph_from_knn(1.0 / k)
    ctx.set_expression(X, np.arange(G))
    w2 = rng_state_words(np.random.default_rng(9))
    two_per_wg = ctx.moran_seeded(w2, P)
    monkeypatch.setenv("SC_SWAP_PW", "1")
    w1 = rng_state_words(np.random.default_rng(9))
    one_per_wg = ctx.moran_seeded(w1, P)
    monkeypatch.delenv("SC_SWAP_PW")
    wh = rng_state_words(np.random.default_rng(9))
    ctx.set_permutations(perm_numpy_host(wh, n, P))
    two_step = ctx.moran(P)
    np.testing.assert_array_equal(w2, wh)
    np.testing.assert_array_equal(w1, wh)
    for key in ("I", "sims", "count_ge"):
        np.testing.assert_array_equal(two_per_wg[key], one_per_wg[key], err_msg=key)
        np.testing.assert_array_equal(two_per_wg[key], two_step[key], err_msg=key)


@pytest.mark.parametrize("n,P,cell_p", [(3000, 40, True), (70001, 150, True), (140001, 300, True), (140001, 99, False)])
def test_lee_local_seeded_equals_the_three_calls(ctx, oracle, n, P, cell_p):
    """sc_lee_local_seeded (the pair body of lees_l_local as one pipeline behind the generator) == sc_perm_generate +
    sc_lee + sc_lee_local on the same rows: every output bit for bit, and the generator state it leaves."""
    from spatialcore_amd._lib import rng_state_words

    coords, X = synth(n, 5, 3, sparse_x=False)
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, np.arange(5))
    Pl = P if cell_p else 0
    w1 = rng_state_words(np.random.default_rng(5))
    for sx, sy in ((0, 1), (3, 2)):          # two pairs continue the one stream
        w0 = w1.copy()
        ctx.generate_permutations(w0, n, P + Pl)
        g = ctx.lee([sx], [sy], [0], P)
        loc = ctx.lee_local(n, sx, sy, Pl, P if cell_p else 0)
        got = ctx.lee_local_seeded(w1, n, sx, sy, P, Pl)
        np.testing.assert_array_equal(w1, w0)
        assert got["L"] == float(g["L"][0]) and got["count_abs_ge"] == int(g["count_abs_ge"][0])
        for key in ("zx", "lag", "L_local"):
            np.testing.assert_array_equal(got[key], loc[key], err_msg=key)
        if cell_p:
            np.testing.assert_array_equal(got["count"], loc["count"])
        else:
            assert got["count"] is None
    Xz = X.copy(); Xz[:, 4] = 2.0
    ctx.set_expression(Xz, np.arange(5))
    with pytest.raises(ValueError):
        ctx.lee_local_seeded(rng_state_words(np.random.default_rng(5)), n, 0, 4, P, Pl)


def test_whole_round_swap_kernel_returns_the_same_tables(ctx, oracle, monkeypatch):
    """SC_SWAP_FULL_ROUNDS=1 (opt-in, a negative result of r04: DESIGN.md 4.3) applies the Fisher-Yates transpositions in
    whole rounds of 1024 steps, hazards resolved through an LDS table instead of ending the round.  Forward tables equal
    numpy's; the inverse-only pipeline scores bit-equal to the default kernel (its rounds are cut at every shared slot)."""
    from spatialcore_amd._lib import rng_state_words, perm_numpy_host

    for seed, n, P in ((5, 65537, 9), (21, 300007, 6), (9, 1048577, 3)):
        want = perm_numpy_host(rng_state_words(np.random.default_rng(seed)), n, P)
        monkeypatch.setenv("SC_SWAP_FULL_ROUNDS", "1")
        got = ctx.generate_permutations(rng_state_words(np.random.default_rng(seed)), n, P, fetch=True)
        monkeypatch.delenv("SC_SWAP_FULL_ROUNDS")
        np.testing.assert_array_equal(got, want)
    n, G, P = 70001, 6, 150
    coords, X = synth(n, G, 2, dtype=np.float32, sparse_x=False)
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, np.arange(G))
    ref = ctx.moran_seeded(rng_state_words(np.random.default_rng(11)), P)
    monkeypatch.setenv("SC_SWAP_FULL_ROUNDS", "1")
    got = ctx.moran_seeded(rng_state_words(np.random.default_rng(11)), P)
    monkeypatch.delenv("SC_SWAP_FULL_ROUNDS")
    for key in ("I", "sims", "count_ge"):
        np.testing.assert_array_equal(got[key], ref[key], err_msg=key)


def test_moran_seeded_inverse_only_tables(ctx, oracle):
    """n >= 65536 with a float32 matrix: the pipeline never builds the permutation table, only its inverse (the same
    Fisher-Yates transpositions applied in ascending order).  Scores must equal the two-step path bit for bit, and
    calls that use the resident table afterwards (sc_moran, a fetch through sc_lee) see the table numpy returns."""
    from spatialcore_amd._lib import rng_state_words, perm_numpy_host

    n, G, P = 70001, 6, 150
    coords, X = synth(n, G, 2, dtype=np.float32, sparse_x=False)
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, np.arange(G))
    w1 = rng_state_words(np.random.default_rng(11))
    ctx.generate_permutations(w1, n, P)
    two = ctx.moran(P)
    w2 = rng_state_words(np.random.default_rng(11))
    one = ctx.moran_seeded(w2, P)
    assert ctx.moran_source_bits() == 8
    np.testing.assert_array_equal(w1, w2)
    for key in ("I", "sims", "count_ge", "sim_sum", "sim_sumsq"):
        np.testing.assert_array_equal(one[key], two[key], err_msg=key)
    again = ctx.moran(P)                       # the resident table: forward rows are materialised on demand
    np.testing.assert_array_equal(again["sims"], two["sims"])
    # NaN in a float32 matrix: no narrow copy holds it, the kernel gathers the fp64 rows of Z (inverse rows all the same)
    Xn = X.copy(); Xn[5, 1] = np.nan
    ctx.set_expression(Xn, np.arange(G))
    w3 = rng_state_words(np.random.default_rng(11))
    nan_run = ctx.moran_seeded(w3, P)
    assert ctx.moran_source_bits() == 64 and np.isnan(nan_run["I"][1])
    keep = [0, 2, 3, 4, 5]
    np.testing.assert_array_equal(nan_run["sims"][:, keep], two["sims"][:, keep])     # the other genes: not a bit changes
    np.testing.assert_array_equal(nan_run["count_ge"][keep], two["count_ge"][keep])
    want = perm_numpy_host(rng_state_words(np.random.default_rng(11)), n, P)
    ctx.set_expression(X, np.arange(G))
    ctx.moran_seeded(rng_state_words(np.random.default_rng(11)), P)
    lee = ctx.lee(np.array([0]), np.array([1]), np.array([0], dtype=np.int64), P, return_perms=True)
    ctx.set_permutations(want)
    lee2 = ctx.lee(np.array([0]), np.array([1]), np.array([0], dtype=np.int64), P, return_perms=True)
    np.testing.assert_array_equal(lee["L_perm"], lee2["L_perm"])


@pytest.mark.parametrize("G", [5, 32, 47, 70, 131, 260])
def test_moran_source_widths_agree(ctx, oracle, G):
    """The permutation kernels gather the narrowest EXACT copy of the raw values: uint8 (128 genes per 128-byte row) for
    integer counts < 256, uint16 (64 genes) for counts < 65536, else float32 (32 genes) when every value is one, else
    the fp64 tiles (16).  Every width rebuilds the same operands and adds the same products in the same order, and the
    integer-count genes of this kNN graph are scored on the exact integer lattice: ALL widths return bit-identical
    statistics and counts.  A kernel for a narrower type is refused when the values do not fit."""
    from spatialcore_amd._lib import rng_state_words

    n, k, P = 3000, 6, 37
    coords, X = synth(n, G, 5, dtype=np.float32, sparse_x=False)
    tab = oracle.morans_i_reference_table(coords, X, list(range(G)), k, P, seed=3)
    ctx.knn(coords, k, fetch=False)
    ctx.graph_from_knn(1.0 / k)
    out = {}
    # r04: 4-bit slots (a count < 16 one nibble, a count < 256 two: x = lo + 16 hi) are used when they take fewer rows
    n_wide = int((np.asarray(X).max(axis=0) >= 16).sum())
    nib_expected = -(-(G + n_wide) // 256) < -(-G // 128)
    try:
        for bits in (4, 8, 16, 32, 64):
            ctx.set_moran_source_bits(bits)
            ctx.set_expression(X, np.arange(G))
            w = rng_state_words(np.random.default_rng(3))
            out[bits] = ctx.moran_seeded(w, P)
            assert ctx.moran_source_bits() == (bits if bits > 4 or nib_expected else 8), (bits, G, n_wide)
        ctx.set_moran_source_bits(8)
        for Xd, want in ((X + np.float32(250.0), 16),              # counts beyond 255: no uint8 copy
                         (X + np.float32(0.5), 32),                # fractional float32 values: no uint16 copy
                         (np.where(X == 0, -1, X).astype(np.float32), 32),   # negative
                         (X * np.float32(70000.0), 32),                    # beyond 65535 (still float32-exact)
                         (X.astype(np.float64) + 1e-9, 64)):       # float64 values that are not float32
            ctx.set_expression(Xd, np.arange(G))
            ctx.moran_seeded(rng_state_words(np.random.default_rng(3)), 3)
            assert ctx.moran_source_bits() == want
        for top, want in ((255.0, 8), (65535.0, 16)):              # the largest counts that still fit
            big = X.copy(); big[7, G - 1] = top
            ctx.set_moran_source_bits(8)
            ctx.set_expression(big, np.arange(G))
            edge = ctx.moran_seeded(rng_state_words(np.random.default_rng(3)), 3)
            assert ctx.moran_source_bits() == want
            ctx.set_moran_source_bits(64)
            ctx.set_expression(big, np.arange(G))
            edge64 = ctx.moran_seeded(rng_state_words(np.random.default_rng(3)), 3)
            for key in ("I", "sims", "count_ge"):
                np.testing.assert_array_equal(edge[key], edge64[key], err_msg=key)
        dense = np.full_like(X, 255.0); dense[::3] = 254.0; dense[:, 1::2] = X[:, 1::2]   # mean / sd at its worst
        ctx.set_moran_source_bits(8)
        ctx.set_expression(dense, np.arange(G))
        hard = ctx.moran_seeded(rng_state_words(np.random.default_rng(3)), 5)
        assert ctx.moran_source_bits() == 8
        ctx.set_moran_source_bits(64)
        ctx.set_expression(dense, np.arange(G))
        hard64 = ctx.moran_seeded(rng_state_words(np.random.default_rng(3)), 5)
        for key in ("I", "sims", "count_ge"):
            np.testing.assert_array_equal(hard[key], hard64[key], err_msg=key)
    finally:
        ctx.set_moran_source_bits(8)      # the default
    assert tab["lattice"].all()
    for bits in (4, 16, 32, 64):
        for key in ("I", "sims", "count_ge", "sim_sum", "sim_sumsq"):
            np.testing.assert_array_equal(out[8][key], out[bits][key], err_msg=f"{key} uint8 vs {bits}")   # bit-identical
    for bits in (8, 16):
        np.testing.assert_allclose(out[bits]["sims"], tab["sims"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(out[bits]["I"], tab["I"], rtol=1e-9, atol=1e-14)
        assert_counts_match(out[bits]["count_ge"], tab)
    with pytest.raises(ValueError):
        ctx.set_moran_source_bits(12)


@pytest.mark.parametrize("n,G,P,wide_every", [(3000, 300, 37, 0), (5000, 523, 130, 7), (2049, 131, 9, 3), (70001, 260, 150, 5)])
def test_moran_four_bit_source_equals_uint8_and_the_oracle(ctx, oracle, n, G, P, wide_every):
    """r04, the 4-bit source (opt-in, sc_ctx_set_moran_source_bits(4)) on its own: counts below 16 as one nibble slot, genes with larger counts (every
    `wide_every`-th, up to 255) as two, in every combination of group fill -- statistics, counts and sums EQUAL to the
    uint8 source's (both are exact integer arithmetic) and the oracle's; sc_moran on a resident table and the pipelined
    sc_moran_seeded (chunks shorter than 24 permutations included: the nibble form has no per-wavefront fallback)."""
    from spatialcore_amd._lib import rng_state_words

    k = 6
    rng = np.random.default_rng(n + G)
    coords = rng.uniform(0, np.sqrt(n) * 10, (n, 2))
    X = rng.poisson(1.2, (n, G)).astype(np.float32)
    X = np.minimum(X, 15)
    if wide_every:
        X[:, ::wide_every] = rng.integers(0, 256, (n, X[:, ::wide_every].shape[1]))
        X[0, 0] = 255.0
    n_wide = int((X.max(axis=0) >= 16).sum())
    assert -(-(G + n_wide) // 256) < -(-G // 128)
    ctx.knn(coords, k, fetch=False)
    ctx.graph_from_knn(1.0 / k)
    runs = {}
    try:
        for bits in (4, 8):
            ctx.set_moran_source_bits(bits)
            ctx.set_expression(X, np.arange(G))
            runs[bits] = ctx.moran_seeded(rng_state_words(np.random.default_rng(5)), P)
            assert ctx.moran_source_bits() == bits and ctx.moran_lag_bits() == 16
            if bits == 4:
                assert ctx.moran_row_groups() == -(-(G + n_wide) // 256)
                two_step = ctx.moran(P)                     # the resident (inverse) table, scored again by sc_moran
                for key in ("I", "sims", "count_ge"):
                    np.testing.assert_array_equal(two_step[key], runs[4][key], err_msg=key)
    finally:
        ctx.set_moran_source_bits(8)
    for key in ("I", "sims", "count_ge", "sim_sum", "sim_sumsq"):
        np.testing.assert_array_equal(runs[4][key], runs[8][key], err_msg=key)
    cols = [0, 1, G // 2, G - 1] + ([wide_every, 2 * wide_every] if wide_every else [])
    tab = oracle.morans_i_reference_table(coords, X[:, cols], list(range(len(cols))), k, P, seed=5)
    np.testing.assert_allclose(runs[4]["I"][cols], tab["I"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(runs[4]["sims"][:, cols], tab["sims"], rtol=1e-9, atol=1e-13)
    assert tab["lattice"].all()
    np.testing.assert_array_equal(runs[4]["count_ge"][cols], tab["count_ge"])


@pytest.mark.parametrize("graph", ["knn", "radius"])
def test_moran_gene_results_do_not_depend_on_coloaded_genes(ctx, oracle, graph):
    """A gene's statistics, counts and hence p-value are a function of the gene, the graph and the permutations
    alone -- not of the genes it shares a device batch (or a rank's shard) with, although those decide which source
    width the kernel gathers: one count >= 256 elsewhere in the batch switches it to uint16, a fractional gene to
    float32, a non-float32 value to the fp64 rows.  kNN graph: the integer genes are lattice genes (exact integer
    statistics).  Radius graph (unequal weights): ordinary arithmetic, identical operands and summation order in
    every width.  Also: uploaded index rows that are not permutations take the index-row kernel."""
    from spatialcore_amd._lib import rng_state_words

    n, G, P = 6000, 37, 45
    coords, X = synth(n, G, 8, dtype=np.float64, sparse_x=False)
    if graph == "knn":
        ctx.knn(coords, 7, fetch=False)
        ctx.graph_from_knn(1.0 / 7)
        conn = oracle.squidpy_connectivities(coords, 7)
    else:
        indptr, indices = oracle.radius_neighbors(coords, 14.0)
        deg = np.diff(indptr)
        ctx.set_graph_csr(indptr, indices, np.repeat(1.0 / np.maximum(deg, 1), deg), n)
        conn = csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))
    extra = {"alone": None,
             "with a count of 300": np.where(np.arange(n) == 11, 300.0, X[:, 0]),
             "with a fractional gene": X[:, 1] + 0.25,
             "with a float64-only gene": X[:, 2] + 1e-9}
    want_bits = {"alone": 8 if graph == "knn" else 16, "with a count of 300": 16, "with a fractional gene": 32,
                 "with a float64-only gene": 64}
    runs = {}
    for name, col in extra.items():
        Xb = X if col is None else np.column_stack([X, col])
        ctx.set_expression(Xb, np.arange(Xb.shape[1]))
        runs[name] = ctx.moran_seeded(rng_state_words(np.random.default_rng(21)), P)
        assert ctx.moran_source_bits() == want_bits[name], name
    for name, out in runs.items():
        for key in ("I", "count_ge", "sim_sum", "sim_sumsq"):
            np.testing.assert_array_equal(out[key][:G], runs["alone"][key], err_msg=f"{key} {name}")
        np.testing.assert_array_equal(out["sims"][:, :G], runs["alone"]["sims"], err_msg=name)
    tab = oracle.morans_i_reference_table(coords, X, list(range(G)), 7, P, seed=21, graph=conn)
    assert tab["lattice"].all() == (graph == "knn") and tab["lattice"].any() == (graph == "knn")
    np.testing.assert_allclose(runs["alone"]["I"], tab["I"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(runs["alone"]["sims"], tab["sims"], rtol=1e-9, atol=1e-13)
    assert_counts_match(runs["alone"]["count_ge"], tab)
    # index rows that are NOT permutations (an extension of sc_perm_set): no inverse exists, no lattice shortcut
    idx = np.random.default_rng(3).integers(0, n, (5, n)).astype(np.int32)
    ctx.set_expression(X, np.arange(G))
    ctx.set_permutations(idx)
    free = ctx.moran(5)
    z, lag, scale = oracle.moran_operands(tab["graph"], oracle.dense_genes(X))
    want = np.stack([scale * (z * lag[:, idx[p]]).sum(axis=1) for p in range(5)])
    np.testing.assert_allclose(free["sims"], want, rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(free["count_ge"], (free["sims"] >= free["I"]).sum(axis=0))


def test_moran_uniform_weights_unequal_degrees_take_the_ordinary_arithmetic(ctx, oracle):
    """r03 advisor finding: the integer-lattice form sum_i z_i lag[pi(i)] = w (T - mean sum S) needs equal weights AND
    equal row degrees (otherwise -w mean sum_i z_i deg[pi(i)] is left over and depends on the permutation).  A binary
    adjacency of a radius graph (every stored weight 1.0, degrees 2 .. 30, some empty rows) with integer counts must
    therefore take the ordinary arithmetic: I, every permutation's statistic and the counts against the oracle's CSR sweep /
    gather form on that very graph; a regular graph with the same weights still is a lattice graph."""
    n, G, P = 5000, 9, 60
    coords, X = synth(n, G, 44, dtype=np.float64, sparse_x=False)
    indptr, indices = oracle.radius_neighbors(coords, 16.0)
    deg = np.diff(indptr)
    assert deg.min() < deg.max()
    g = csr_matrix((np.ones(indices.size), indices, indptr), shape=(n, n))      # NOT row-normalised: all weights equal
    vals = oracle.dense_genes(X)
    assert not oracle.lattice_genes(g, vals).any()
    perms, _ = oracle.perm_table(3, n, P)
    want_I = oracle.morans_i_scores(g, vals)
    want_sims = oracle.morans_i_sims_gather(g, vals, perms)
    want_lit = np.stack([oracle.morans_i_scores(g, vals, perms[p]) for p in range(3)])      # the literal row-permuted sweep
    np.testing.assert_allclose(want_sims[:3], want_lit, rtol=1e-10, atol=1e-14)
    ctx.set_graph_csr(indptr, indices, np.ones(indices.size), n)
    ctx.set_expression(X, np.arange(G))
    ctx.set_permutations(perms)
    out = ctx.moran(P)
    assert ctx.moran_source_bits() == 16                  # integer counts, but no lattice gene: the centred uint16 kernel
    np.testing.assert_allclose(out["I"], want_I, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["sims"], want_sims, rtol=1e-9, atol=1e-13)
    count, lat = oracle.morans_count_ge(g, vals, perms, want_sims, want_I)
    assert not lat.any()
    assert_counts_match(out["count_ge"], {"lattice": lat, "count_ge": count, "sims": want_sims, "I": want_I})
    # the same weights on a regular graph (kNN, every row k entries of weight 1.0): lattice genes, exact counts
    k = 7
    nbr = np.sort(oracle.knn_tree(coords, k), axis=1)
    gk = csr_matrix((np.ones(n * k), nbr.reshape(-1), np.arange(0, n * k + 1, k)), shape=(n, n))
    assert oracle.lattice_genes(gk, vals).all()
    ctx.set_graph_csr(gk.indptr, gk.indices, gk.data, n)
    ctx.set_expression(X, np.arange(G))
    ctx.set_permutations(perms)
    outk = ctx.moran(P)
    assert ctx.moran_source_bits() == 8
    simsk = oracle.morans_i_sims_gather(gk, vals, perms)
    Ik = oracle.morans_i_scores(gk, vals)
    np.testing.assert_allclose(outk["sims"], simsk, rtol=1e-9, atol=1e-13)
    countk, latk = oracle.morans_count_ge(gk, vals, perms, simsk, Ik)
    np.testing.assert_array_equal(outk["count_ge"], countk)


def test_moran_gene_subset_and_uploaded_perms(ctx, oracle):
    coords, X = synth(4000, 12, 23, dtype=np.float32)
    cols = [7, 2, 11]
    tab = oracle.morans_i_reference_table(coords, X, cols, 6, 9, seed=5)
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, cols)
    ctx.set_permutations(tab["perms"])
    out = ctx.moran(9)
    np.testing.assert_allclose(out["I"], tab["I"], rtol=1e-9)
    np.testing.assert_allclose(out["sims"], tab["sims"], rtol=1e-9, atol=1e-13)
    with pytest.raises(ValueError):
        ctx.set_permutations(np.full((2, 4000), 4000, dtype=np.int32))   # index out of range


def test_moran_zero_variance_gene_is_nan(ctx, oracle):
    coords, X = synth(1000, 3, 4, sparse_x=False)
    X[:, 1] = 2.0
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, [0, 1, 2])
    ctx.set_permutations(oracle.perm_table(0, 1000, 5)[0])
    out = ctx.moran(5)
    assert np.isnan(out["I"][1]) and out["count_ge"][1] == 0
    assert np.isfinite(out["I"][[0, 2]]).all()


@pytest.mark.parametrize("n", [501, 561, 927, 1000])
def test_constant_gene_is_zero_variance_for_any_cell_count(ctx, oracle, n):
    """mean = sum / n, not sum * (1/n): for ~10-15 % of the cell counts (49, 98, 103, 107, 501, 561, 927, ...)
    c * n * fl(1/n) != c, which would leave a constant gene with z = +-1e-16 and a finite I."""
    assert (float(n) * (1.0 / n) != 1.0) == (n != 1000)                 # three counts that used to fail + one that did not
    coords, X = synth(n, 3, 4, sparse_x=False)
    X[:, 1] = 1.0
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, [0, 1, 2])
    mean, var = ctx.expr_stats()
    assert mean[1] == 1.0 and var[1] == 0.0
    ctx.set_permutations(oracle.perm_table(0, n, 5)[0])
    out = ctx.moran(5)
    assert np.isnan(out["I"][1]) and out["count_ge"][1] == 0
    assert np.isfinite(out["I"][[0, 2]]).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_numpy_order_column_sums_large(ctx, dtype):
    """local_morans_i standardises with float means / sds that must be numpy's own (scipy sparse .mean(axis=0):
    first stored entry + PAIRWISE sum of the rest).  The device evaluates that summation tree in parallel
    (compaction, leaves, ordered combine); every z must equal the host's bit for bit, for column populations
    around every branch of the recursion (0, 1, 2, 7, 8, 9, 128, 129, 130, 1000, half, all)."""
    from scipy import sparse

    n = 150_001
    rng = np.random.default_rng(12)
    pops = [0, 1, 2, 7, 8, 9, 128, 129, 130, 137, 1000, 4097, n // 2, n - 1, n]
    X = np.zeros((n, len(pops)), dtype=dtype)
    for g, cnt in enumerate(pops):
        rows = rng.choice(n, cnt, replace=False)
        X[rows, g] = (rng.poisson(3.0, cnt) + 1) * (1.0 if g % 2 else 0.37)     # counts and non-integers
    coords = rng.uniform(0, 4000.0, (n, 2))
    Xs = sparse.csc_matrix(X)
    mean = np.asarray(Xs.mean(axis=0)).ravel()                                 # the reference's expressions (AC:79-107)
    sq_mean = np.asarray(Xs.power(2).mean(axis=0)).ravel()
    means = mean.astype(np.float32)
    stds = np.sqrt(sq_mean - mean ** 2).astype(np.float32)
    zero = stds == 0
    stds[zero] = 1.0
    want = (X.astype(np.float32) - means) / stds
    want[:, zero] = 0.0
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(1.0 / 6)
    ctx.set_expression(X, np.arange(len(pops)))
    out = ctx.local_moran(n, 0)
    np.testing.assert_array_equal(out["zero_var"], zero)
    got = out["z"].copy(); got[:, zero] = 0.0
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("uniform", [True, False])
def test_local_moran_code_rows_equal_float_rows(ctx, monkeypatch, uniform):
    """Count data (every value an integer below 32) travels through the per-cell permutation counts as uint8 code rows
    with z looked up per (gene, value) (k_lm_gather_u8 / k_lm_count_u8; with equal weights also w * z from a table).  The
    counts must be those of the float-row form (SC_LM_FLOAT_ROWS), cell for cell: 150 genes = two 128-gene groups with a
    ragged second one, 37 permutations = a ragged last quad, on a row-normalised kNN graph and on a graph with unequal
    weights.  A value of 32 or a non-integer sends the call to the float rows by itself (same counts again)."""
    from scipy import sparse
    from spatialcore_amd._lib import rng_state_words

    n, G, P, k = 5000, 150, 37, 6
    rng = np.random.default_rng(5)
    coords = rng.uniform(0, 700.0, (n, 2))
    X = rng.poisson(rng.uniform(0.05, 4.0, G), (n, G)).astype(np.float32)
    X[X > 31] = 31
    X[:, 7] = 0                      # zero variance
    idx = ctx.knn(coords, k)
    if uniform:
        ctx.graph_from_knn(1.0 / k)
    else:
        w = rng.uniform(0.1, 1.0, (n, k))
        W = sparse.csr_matrix((w.ravel(), idx.ravel().astype(np.int32), np.arange(0, n * k + 1, k)), shape=(n, n))
        W.sort_indices()
        ctx.set_graph_csr(W.indptr, W.indices, W.data, n)

    def run(Xm):
        ctx.set_expression(Xm, np.arange(G))
        ctx.generate_permutations(rng_state_words(np.random.default_rng(9)), n, P)
        return ctx.local_moran(n, P)

    got = run(X)
    monkeypatch.setenv("SC_LM_FLOAT_ROWS", "1")
    want = run(X)
    monkeypatch.delenv("SC_LM_FLOAT_ROWS")
    for f in ("z", "lag", "I", "count"):
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    assert got["count"].max() <= P and got["count"].sum() > 0
    for bad in (32.0, 0.5):
        Xb = X.copy(); Xb[11, 3] = bad
        a = run(Xb)
        monkeypatch.setenv("SC_LM_FLOAT_ROWS", "1")
        b = run(Xb)
        monkeypatch.delenv("SC_LM_FLOAT_ROWS")
        np.testing.assert_array_equal(a["count"], b["count"])


@pytest.mark.parametrize("n,P,counts", [(3000, 41, True), (140_000, 150, True), (140_000, 37, False)])
def test_local_moran_seeded_equals_generate_then_count(ctx, n, P, counts):
    """sc_local_moran_seeded draws its permutations inside the call, chunk by chunk beside the per-cell counts (block-parallel
    generator from 131072 cells on): outputs, counts and the advanced generator state must be those of
    sc_perm_generate followed by sc_local_moran -- for count data (uint8 code rows) and for non-integer data (float
    rows)."""
    from spatialcore_amd._lib import rng_state_words

    G, k = 21, 6
    rng = np.random.default_rng(n + P)
    coords = rng.uniform(0, np.sqrt(n) * 10, (n, 2))
    X = rng.poisson(rng.uniform(0.1, 3.0, G), (n, G)).astype(np.float32)
    if not counts:
        X *= np.float32(0.37)
    ctx.knn(coords, k, fetch=False)
    ctx.graph_from_knn(1.0 / k)
    ctx.set_expression(X, np.arange(G))
    fallbacks = ctx.permgen_stats()[2]
    w1 = rng_state_words(np.random.default_rng(77))
    ctx.generate_permutations(w1, n, P)
    want = ctx.local_moran(n, P)
    w2 = rng_state_words(np.random.default_rng(77))
    got = ctx.local_moran_seeded(w2, n, P)
    np.testing.assert_array_equal(w1, w2)
    for f in ("z", "lag", "I", "count"):
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    assert ctx.permgen_stats()[2] == fallbacks    # no verification fallback


def test_lee_vs_reference_golden(ctx, oracle):
    from spatialcore_amd._lib import rng_state_words

    g = load_golden("ref_lees_l.npz")
    for ci in range(int(g["n_cases"])):
        coords, X = g[f"c{ci}_coords"], g[f"c{ci}_X"]
        k, P, seed = int(g[f"c{ci}_k"]), int(g[f"c{ci}_P"]), int(g[f"c{ci}_seed"])
        pairs = g[f"c{ci}_pairs"]
        n = X.shape[0]
        ctx.knn(coords, k, fetch=False)
        ctx.graph_from_knn(float(np.float32(1.0) / np.float32(k)))
        ctx.set_expression(X, np.arange(X.shape[1]))
        _, var = ctx.expr_stats()
        # one stream for all pairs; degenerate pairs draw nothing (AC:1109-1148)
        off, nxt = [], 0
        for a, b in pairs:
            if var[a] > 0 and var[b] > 0:
                off.append(nxt)
                nxt += P
            else:
                off.append(-1)
        if P > 0 and nxt > 0:
            ctx.generate_permutations(rng_state_words(np.random.default_rng(seed)), n, nxt)
        out = ctx.lee(pairs[:, 0], pairs[:, 1], off, P)
        tol = 1e-9 if X.dtype == np.float64 else 2e-5
        np.testing.assert_allclose(out["L"], g[f"c{ci}_L"], rtol=tol, atol=tol)
        p = (out["count_abs_ge"] + 1) / (P + 1) if P > 0 else np.ones(len(pairs))
        p = np.where(np.array(off) < 0, 1.0, p)
        np.testing.assert_array_equal(p, g[f"c{ci}_p"])


def test_lee_seeded_batch_vs_pairwise_path_and_golden(ctx, oracle):
    """sc_lee_seeded (one call: MFMA observed statistics + pipelined per-pair permutation blocks) against the
    reference's golden output and against the per-pair path (sc_perm_generate + sc_lee) on the same stream."""
    from spatialcore_amd._lib import rng_state_words

    g = load_golden("ref_lees_l.npz")
    for ci in range(int(g["n_cases"])):
        coords, X = g[f"c{ci}_coords"], g[f"c{ci}_X"]
        k, P, seed = int(g[f"c{ci}_k"]), int(g[f"c{ci}_P"]), int(g[f"c{ci}_seed"])
        pairs = g[f"c{ci}_pairs"]
        n = X.shape[0]
        ctx.knn(coords, k, fetch=False)
        ctx.graph_from_knn(float(np.float32(1.0) / np.float32(k)))
        ctx.set_expression(X, np.arange(X.shape[1]))
        _, var = ctx.expr_stats()
        live = np.array([var[a] > 0 and var[b] > 0 for a, b in pairs])
        w = rng_state_words(np.random.default_rng(seed))
        out = ctx.lee_seeded(w, pairs[:, 0], pairs[:, 1], P, return_perms=True)
        tol = 1e-9 if X.dtype == np.float64 else 2e-5
        np.testing.assert_allclose(out["L"], g[f"c{ci}_L"], rtol=tol, atol=tol)
        p = np.where(live, (out["count_abs_ge"] + 1) / (P + 1), 1.0) if P > 0 else np.ones(len(pairs))
        np.testing.assert_array_equal(p, g[f"c{ci}_p"])
        # the per-pair path on the same stream: same permutation statistics to summation order, same state afterwards
        w2 = rng_state_words(np.random.default_rng(seed))
        off = np.where(live, np.cumsum(live) - 1, -1) * P
        off[~live] = -1
        if P > 0 and live.any():
            ctx.generate_permutations(w2, n, int(live.sum()) * P)
            np.testing.assert_array_equal(w, w2)
        ref = ctx.lee(pairs[:, 0], pairs[:, 1], off, P, return_perms=True)
        np.testing.assert_allclose(out["L"], ref["L"], rtol=1e-12, atol=1e-12)
        if P > 0:
            np.testing.assert_allclose(out["L_perm"], ref["L_perm"], rtol=1e-9, atol=1e-10)
            np.testing.assert_array_equal(out["count_abs_ge"][live], ref["count_abs_ge"][live])
            np.testing.assert_array_equal(out["count_abs_ge"][~live], P)


def test_lee_seeded_dense_pair_grid_mfma(ctx, oracle):
    """A 37 x 21 grid of pairs over 58 genes (several 16-gene tiles per side, ragged last tiles): the fp64 MFMA
    contraction of the observed statistics against plain dot products of the oracle's z-scores and lags, P = 0."""
    n, G = 6000, 58
    coords, X = synth(n, G, 21, dtype=np.float64, sparse_x=False)
    X[:, 5] = 3.0                                              # one zero-variance gene on the x side
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(float(np.float32(1.0) / np.float32(6)))
    ctx.set_expression(X, np.arange(G))
    px, py = np.meshgrid(np.arange(37), np.arange(37, 58), indexing="ij")
    out = ctx.lee_seeded(None, px.ravel(), py.ravel(), 0)
    W = oracle.reference_weights(coords, 6).astype(np.float64)
    sd = X.std(axis=0)
    Z = np.where(sd > 0, (X - X.mean(axis=0)) / np.where(sd > 0, sd, 1), 0.0)
    want = Z[:, :37].T @ (W @ Z[:, 37:])
    want[5, :] = 0.0
    np.testing.assert_allclose(out["L"].reshape(37, 21), want, rtol=1e-10, atol=1e-9)
    assert (out["count_abs_ge"] == 0).all()


@pytest.mark.parametrize("n", [1200, 8192, 8200, 70001])
def test_lee_observed_float32_is_numpys_own_number(ctx, oracle, n):
    """For a float32 matrix the reference's L is numpy / scipy float32 arithmetic (AC:1118-1146, 307-315).  The device
    evaluates the same summation tree with the same roundings (chunks of 8192, pairwise inside): bit-for-bit the
    numbers numpy itself produces here for the restated formula, across the chunk boundary cases."""
    coords, X = synth(n, 6, 40 + n % 7, dtype=np.float32, sparse_x=False)
    X[:, 4] = 2.0                                                   # zero float32 std
    W = oracle.reference_weights(coords, 6)                         # float32 CSR, as build_spatial_weights returns
    pairs = np.array([[0, 1], [2, 3], [1, 0], [4, 2], [5, 5]])
    want = []
    for a, b in pairs:
        x, y = X[:, a], X[:, b]
        if x.std() == 0 or y.std() == 0:
            want.append(0.0)
            continue
        zx, zy = (x - x.mean()) / x.std(), (y - y.mean()) / y.std()
        want.append(float((zx * (W @ zy)).sum()))
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(float(np.float32(1.0) / np.float32(6)))
    ctx.set_expression(X, np.arange(6))
    got = ctx.lee_observed_f32(pairs[:, 0], pairs[:, 1])
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got.astype(np.float64), np.array(want))
    ctx.set_expression(X.astype(np.float64), np.arange(6))
    with pytest.raises(Exception, match="not float32"):
        ctx.lee_observed_f32(pairs[:, 0], pairs[:, 1])


def test_lee_shared_permutation_grid_mfma(ctx, oracle):
    """EXTENSION: the genes_x x genes_y grid under one shared block of permutations (fp64 MFMA with a row-gathered
    operand) against the reference's core loop restated with that one table: L, every L_perm, the counts."""
    from spatialcore_amd._lib import rng_state_words

    n, G, P = 9000, 45, 21
    coords, X = synth(n, G, 33, dtype=np.float64, sparse_x=False)
    X[:, 3] = 1.0                                                # zero variance on the x side
    ctx.knn(coords, 6, fetch=False)
    ctx.graph_from_knn(float(np.float32(1.0) / np.float32(6)))
    ctx.set_expression(X, np.arange(G))
    gx, gy = np.array([0, 3, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 40]), np.arange(23, 44)
    w = rng_state_words(np.random.default_rng(8))
    out = ctx.lee_shared(w, gx, gy, P, return_perms=True)
    perms, wfin = oracle.perm_table(8, n, P)
    np.testing.assert_array_equal(w, wfin)
    W = oracle.reference_weights(coords, 6).astype(np.float64)
    sd = X.std(axis=0)
    Z = np.where(sd > 0, (X - X.mean(axis=0)) / np.where(sd > 0, sd, 1), 0.0)
    U = W.T @ Z[:, gx]
    L = Z[:, gx].T @ (W @ Z[:, gy])
    np.testing.assert_allclose(out["L"], L, rtol=1e-10, atol=1e-9)
    Lp = np.stack([U.T @ Z[perms[p]][:, gy] for p in range(P)])
    np.testing.assert_allclose(out["L_perm"], Lp, rtol=1e-9, atol=1e-8)
    want = (np.abs(Lp) >= np.abs(L)[None]).sum(axis=0)
    np.testing.assert_array_equal(out["count_abs_ge"], want)
    assert (out["count_abs_ge"][1] == P).all() and (out["L"][1] == 0).all()      # the constant gene: p = 1


def test_profile_counts_golden(ctx):
    g = load_golden("ref_profile.npz")
    coords, labels = g["coords"], g["labels"]
    cats = sorted(set(labels.tolist()))
    code = np.array([cats.index(v) for v in labels.tolist()], dtype=np.int32)
    ctx.knn(coords, 15, fetch=False)
    ctx.graph_from_knn(1.0)
    cnt = ctx.profile_counts(code, len(cats))
    np.testing.assert_array_equal(cnt / cnt.sum(axis=1, keepdims=True), g["knn_profile"])
    indptr, indices = ctx.radius_graph(coords, 25.0)
    ctx.set_graph_csr(indptr, indices, np.ones(indices.size), coords.shape[0])
    cnt = ctx.profile_counts(code, len(cats))
    np.testing.assert_array_equal((cnt / cnt.sum(axis=1, keepdims=True)).astype(np.float32), g["radius_profile"])


def test_nearest_and_pairwise_vs_scipy(ctx):
    from scipy.spatial import cKDTree
    from scipy.spatial.distance import cdist

    rng = np.random.default_rng(12)
    targets = rng.normal([300, 300], 40, (5000, 2))
    queries = np.concatenate([rng.uniform(0, 1000, (3000, 2)),            # mostly far outside the target box
                              rng.normal([300, 300], 40, (500, 2)),       # inside it
                              [[-5e4, 7e4], [300.0, 300.0]]])
    d, idx = ctx.nearest(targets, queries)
    wd, wi = cKDTree(targets).query(queries, k=1)
    np.testing.assert_array_equal(idx, wi)
    np.testing.assert_array_equal(d, wd)
    a, b = rng.uniform(0, 100, (1300, 2)), rng.uniform(50, 400, (2111, 2))
    mean, mn = ctx.pairwise(a, b)
    pw = cdist(a, b)
    assert mn == pw.min()
    assert mean == pytest.approx(pw.mean(), rel=1e-13)
    one_mean, one_min = ctx.pairwise(a[:1], b[:1])
    assert one_mean == one_min == pw[0, 0]
