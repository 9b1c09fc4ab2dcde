"""CPU property tests (hypothesis) of the host-side logic in spatialcore_amd."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import load_golden


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 60), st.integers(1, 400), st.integers(0, 2**32 - 1))
def test_count_level_bh_equals_sort_based_bh(n_perm, n_cells, seed):
    """The O(P) count-level Benjamini-Hochberg tables used by local_morans_i give, bit for bit, what the
    reference's sort-based BH (AC:132-164; restated in the oracle, pinned by the reference's golden below) gives on
    p = float32((c + 1) / (P + 1)), stored into a float32 column as the reference does (AC:912-914)."""
    import oracle as orc
    from spatialcore_amd.spatial.autocorrelation import _padj_tables

    rng = np.random.default_rng(seed)
    counts = rng.integers(0, n_perm + 1, n_cells).astype(np.int32)
    if seed % 3 == 0:
        counts[:] = rng.integers(0, n_perm + 1)          # all tied
    p32 = ((counts + 1) / (n_perm + 1)).astype(np.float32)
    want = np.ones(n_cells, dtype=np.float32)
    want[:] = orc.fdr_bh(p32)                           # float64 result assigned into a float32 column
    hist = np.bincount(counts, minlength=n_perm + 1)[None, :]
    got = _padj_tables(hist, n_cells, n_perm, "fdr_bh")[0][counts]
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, want)


def test_oracle_bh_matches_reference_golden_values():
    import oracle as orc

    g = load_golden("ref_fdr_quadrants.npz")
    np.testing.assert_array_equal(orc.fdr_bh(g["p"]), g["bh"])
    np.testing.assert_array_equal(orc.bonferroni(g["p"]), g["bonf"])


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 16))
def test_shard_bounds_partition(n_items, world):
    from spatialcore_amd.parallel import shard_bounds

    cover = np.zeros(n_items, dtype=int)
    for r in range(world):
        a, b = shard_bounds(n_items, world, r)
        assert 0 <= a <= b <= n_items
        cover[a:b] += 1
    assert (cover == 1).all()


@settings(max_examples=100, deadline=None)
@given(st.lists(st.integers(0, 9), min_size=1, max_size=30))
def test_unique_columns_roundtrip(picks):
    from spatialcore_amd import SimpleAnnData
    from spatialcore_amd.spatial.autocorrelation import _unique_columns

    ad = SimpleAnnData(np.zeros((3, 10)), var_names=[f"g{i}" for i in range(10)])
    names = [f"g{i}" for i in picks]
    cols, where = _unique_columns(ad, names)
    assert len(set(cols.tolist())) == len(cols)                     # distinct columns go to the device
    assert [f"g{cols[w]}" for w in where] == names                   # every request maps back to its gene


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 2**31 - 1), st.integers(1, 300), st.integers(1, 4))
def test_host_generator_is_a_permutation_and_matches_numpy(seed, n, reps):
    from spatialcore_amd import _lib

    w = _lib.rng_state_words(np.random.default_rng(seed))
    got = _lib.perm_numpy_host(w, n, reps)
    rng = np.random.default_rng(seed)
    for r in range(reps):
        np.testing.assert_array_equal(got[r], rng.permutation(n))
    np.testing.assert_array_equal(w, _lib.rng_state_words(rng))


def test_quadrant_rules():
    from spatialcore_amd.spatial.autocorrelation import QUADRANT_LABELS, _classify_quadrants

    z = np.array([1.0, -1.0, 1.0, -1.0, 0.0, 2.0])
    lag = np.array([1.0, -1.0, -1.0, 1.0, 3.0, 0.0])
    q = _classify_quadrants(z, lag)
    assert [QUADRANT_LABELS[v] for v in q] == ["HH", "LL", "HL", "LH", "NS", "NS"]
    q = _classify_quadrants(z, lag, np.array([0.01, 0.05, 0.2, 0.049, 0.0, 0.0]), alpha=0.05)
    assert [QUADRANT_LABELS[v] for v in q] == ["HH", "NS", "NS", "LH", "NS", "NS"]
    assert q.dtype == np.int8


def test_padj_tables_match_per_gene_corrections():
    """The per-(gene, count level) lookup tables handed to the device finalisation of local_morans_i give exactly
    what the reference's per-gene FDR functions give cell by cell (BH, Bonferroni, none; oracle restatements)."""
    import oracle as orc
    from spatialcore_amd.spatial.autocorrelation import _padj_tables

    rng = np.random.default_rng(5)
    n, P, G = 5000, 19, 6
    counts = rng.integers(0, P + 1, (n, G))
    counts[:, 2] = P                       # a degenerate gene: every cell at the top level
    counts[:, 3] = rng.integers(0, 3, n)   # only a few levels populated
    hist = np.stack([np.bincount(counts[:, g], minlength=P + 1) for g in range(G)])
    p = ((counts + 1) / (P + 1)).astype(np.float32)
    correct = {"fdr_bh": orc.fdr_bh, "bonferroni": orc.bonferroni, "none": lambda v: v.copy()}
    for method, fn in correct.items():
        tab = _padj_tables(hist, n, P, method)
        assert tab.dtype == np.float32 and tab.shape == (G, P + 1)
        for g in range(G):
            want = np.ones(n, dtype=np.float32)
            want[:] = fn(p[:, g])
            np.testing.assert_array_equal(tab[g][counts[:, g]], want, err_msg=f"{method} gene {g}")


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**31 - 1), st.integers(30, 120), st.integers(2, 6))
def test_lattice_count_is_the_exact_arithmetic_count(seed, n, k):
    """Integer counts on a kNN graph (all weights 1/k): the oracle's lattice count #{T_p >= T_obs}, T = sum_i x_i S[perm_i],
    equals the count of `sims_p >= I` evaluated in EXACT rational arithmetic on the reference's definition
    (z = x - mean, lag = W z, sum_i z_i lag[perm_i]); the float comparison agrees wherever no permutation ties."""
    from fractions import Fraction

    import oracle as orc
    from scipy.sparse import csr_matrix

    rng = np.random.default_rng(seed)
    coords = rng.uniform(0, 100, (n, 2))
    # two levels only for gene 0: many exact ties; gene 1 Poisson; gene 2 not a lattice gene (fractional values)
    vals = np.stack([rng.integers(0, 2, n), rng.poisson(1.5, n), rng.poisson(1.5, n) + 0.5]).astype(np.float64)
    idx = orc.knn_tree(coords, k)
    g = orc.row_normalize_l1(csr_matrix((np.ones(n * k), idx.reshape(-1), np.arange(0, n * k + 1, k)), shape=(n, n)))
    P = 40
    perms, _ = orc.perm_table(seed % 1000, n, P)
    count, lat = orc.morans_count_ge(g, vals, perms)
    assert lat.tolist() == [True, True, False]
    sims = orc.morans_i_sims_gather(g, vals, perms)
    score = orc.morans_i_scores(g, vals)
    for gi in (0, 1):
        x = [Fraction(int(v)) for v in vals[gi]]
        mean = sum(x) / n
        z = [v - mean for v in x]
        lag = [sum(z[j] for j in idx[i]) / k for i in range(n)]
        obs = sum(z[i] * lag[i] for i in range(n))
        exact = [sum(z[i] * lag[perms[p][i]] for i in range(n)) for p in range(P)]
        want = sum(1 for v in exact if v >= obs) if any(v != x[0] for v in x) else 0
        assert count[gi] == want
        ties = sum(1 for v in exact if v == obs)
        assert abs(int((sims[:, gi] >= score[gi]).sum()) - want) <= ties
    assert count[2] == (sims[:, 2] >= score[2]).sum()
    # a graph with unequal weights has no lattice genes
    g2 = g.copy(); g2.data[0] *= 0.5
    assert not orc.lattice_genes(g2, vals).any()
    # ... nor has one with equal weights but unequal degrees (r03 advisor finding): the identity behind the lattice form
    # loses the term -w mean sum_i z_i deg[perm_i], which depends on the permutation -- shown here in exact arithmetic
    keep = np.ones(n * k, dtype=bool); keep[: k - 1] = False                 # row 0 keeps one neighbour
    indptr3 = np.concatenate([[0], np.cumsum(np.where(np.arange(n) == 0, 1, k))])
    g3 = csr_matrix((np.ones(int(keep.sum())), idx.reshape(-1)[keep], indptr3), shape=(n, n))
    assert not orc.lattice_genes(g3, vals).any()
    x = [Fraction(int(v)) for v in vals[1]]
    mean = sum(x) / n
    z = [v - mean for v in x]
    rows = [g3.indices[g3.indptr[i]:g3.indptr[i + 1]] for i in range(n)]
    S = [sum(x[j] for j in rows[i]) for i in range(n)]
    lag = [sum(z[j] for j in rows[i]) for i in range(n)]
    diffs = set()
    for p in range(6):
        exact = sum(z[i] * lag[perms[p][i]] for i in range(n))
        T = sum(x[i] * S[perms[p][i]] for i in range(n))
        diffs.add(exact - (T - mean * sum(S)))
    assert len(diffs) > 1                                                     # not a constant offset: counts would differ


def test_swap_rounds_rule_equals_the_sequential_shuffle():
    """The rule of k_apply_swaps_full (whole rounds of T steps, hazards resolved through the smallest / largest step per
    partner slot, cut at the first middle step) on the CPU: both directions, small n and T, i.e. conflicts in every round."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(__file__), "..", "scripts", "swap_rounds_sim.py")
    spec = importlib.util.spec_from_file_location("swap_rounds_sim", path)
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    assert sim.check(seed=11, trials=200) is None


def test_fixed_point_rounds_restatement_is_exact():
    """scripts/fixed_point_rounds_sim.py restates block_fixed_point (sc_permgen.hip) on the CPU to count its rounds; the
    entering counts it converges to must be those of a plain sequential scan of the block, for a block in which a
    permutation ends and for an ordinary one, in all three variants it compares."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(__file__), "..", "scripts", "fixed_point_rounds_sim.py")
    spec = importlib.util.spec_from_file_location("fixed_point_rounds_sim", path)
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    rng = np.random.default_rng(17)
    for rem in (int(rng.integers(500, 12288)), int(rng.integers(200_000, 900_000))):
        u = rng.integers(0, 1 << 32, size=(sim.TH, sim.D), dtype=np.int64)
        want = sim.truth(u, rem)
        for kw in ({}, {"mode": "newton"}, {"mode": "tail", "tail_i": 1024}):
            rounds, cnt = sim.run(u, rem, **kw)
            assert rounds > 0 and np.array_equal(cnt, want), (rem, kw)
