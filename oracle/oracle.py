"""CPU oracle for the spatial-statistics hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``spatialcore_amd/`` does.  It restates, in NumPy/SciPy (+ the scalar C
helpers of ``oracle_c.c``), what the reference computes on the path of SURVEY.md section 8(a).
``AC`` = /root/reference/src/spatialcore/spatial/autocorrelation.py,
``NB`` = /root/reference/src/spatialcore/spatial/neighborhoods.py.

Pinning status
--------------
* In-repo arithmetic (``build_spatial_weights``, ``lees_l``, ``lees_l_local``,
  ``local_morans_i``, BH / Bonferroni, quadrants, ``compute_neighborhood_profile``): pinned by
  ``tests/golden/ref_*.npz``, produced by running the reference's own functions in the build
  container (``oracle/make_golden.py``; anndata / squidpy replaced by inert stand-ins).
* Permutation stream: pinned by ``tests/golden/rng_kat.npz`` (produced by numpy itself).
* Global ``morans_i`` (AC:421-648) delegates its arithmetic to squidpy -> scanpy, an un-vendored,
  un-pinned dependency (pyproject.toml:39) that is absent here: **parity unpinned**.  The
  restatement below follows the published algorithm of squidpy ``gr.spatial_autocorr`` /
  ``_score_helper`` / ``_p_value_calc`` / ``_analytic_pval`` / ``_g_moments`` and scanpy
  ``metrics.morans_i``; it is cross-checked against the importable ``local_morans_i``
  (sum of local I / N == global I on the same graph) and hand-computed lattices.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy import sparse, stats
from scipy.sparse import csr_matrix

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force: bool = False) -> str:
    """Compile oracle_c.c -> liboracle.so (gcc)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle_c.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def clib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = os.environ.get("SC_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # (SC_ORACLE_LIB: the sanitizer build)
        if not os.path.exists(so):
            build_c()
        _LIB = ctypes.CDLL(so)
    return _LIB


def set_threads(n: int) -> int:
    """OpenMP threads of the two Moran kernels (genes in parallel, as scanpy's prange); returns the count in effect."""
    return int(clib().orc_set_threads(ctypes.c_int(int(n))))


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# =============================================================================================
# A4 -- permutation source (numpy Generator(PCG64).permutation), AC:839,879 / 1109,324 / 1367,1404
# =============================================================================================

_PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645
_M128 = (1 << 128) - 1


class PCG64Model:
    """Pure-Python model of numpy's PCG64 + Generator.permutation.  Small n only."""

    def __init__(self, state: int, inc: int, has_uint32: int = 0, uinteger: int = 0):
        self.state, self.inc, self.has_uint32, self.uinteger = state, inc, has_uint32, uinteger

    @classmethod
    def from_generator(cls, rng: np.random.Generator) -> "PCG64Model":
        st = rng.bit_generator.state
        assert st["bit_generator"] == "PCG64"
        return cls(st["state"]["state"], st["state"]["inc"], st["has_uint32"], st["uinteger"])

    def next64(self) -> int:
        self.state = (self.state * _PCG_MULT + self.inc) & _M128
        hi, lo = self.state >> 64, self.state & 0xFFFFFFFFFFFFFFFF
        x, r = hi ^ lo, hi >> 58
        return ((x >> r) | (x << ((64 - r) & 63))) & 0xFFFFFFFFFFFFFFFF

    def next32(self) -> int:
        if self.has_uint32:
            self.has_uint32 = 0
            return self.uinteger
        v = self.next64()
        self.has_uint32, self.uinteger = 1, v >> 32
        return v & 0xFFFFFFFF

    def interval(self, mx: int) -> int:
        if mx == 0:
            return 0
        mask = (1 << mx.bit_length()) - 1
        while True:
            v = (self.next32() if mx <= 0xFFFFFFFF else self.next64()) & mask
            if v <= mx:
                return v

    def permutation(self, n: int) -> np.ndarray:
        a = list(range(n))
        for i in range(n - 1, 0, -1):
            j = self.interval(i)
            a[i], a[j] = a[j], a[i]
        return np.array(a, dtype=np.int64)


def rng_state_words(rng: np.random.Generator) -> np.ndarray:
    """{state, inc, has_uint32, uinteger} of a PCG64 Generator as 6 uint64 words."""
    st = rng.bit_generator.state
    if st["bit_generator"] != "PCG64":
        raise ValueError("only PCG64 generators are modelled")
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = 0xFFFFFFFFFFFFFFFF
    return np.array([s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]],
                    dtype=np.uint64)


def set_rng_state(rng: np.random.Generator, words: np.ndarray) -> None:
    w = [int(x) for x in words]
    rng.bit_generator.state = {
        "bit_generator": "PCG64",
        "state": {"state": (w[0] << 64) | w[1], "inc": (w[2] << 64) | w[3]},
        "has_uint32": w[4],
        "uinteger": w[5],
    }


def perm_table(seed_or_words, n: int, n_perm: int) -> Tuple[np.ndarray, np.ndarray]:
    """``n_perm`` consecutive ``rng.permutation(n)`` results, (P, n) int32, + final state words."""
    if isinstance(seed_or_words, (int, np.integer)):
        words = rng_state_words(np.random.default_rng(int(seed_or_words)))
    else:
        words = np.array(seed_or_words, dtype=np.uint64).copy()
    out = np.empty((n_perm, n), dtype=np.int32)
    rc = clib().orc_perm_numpy(_p(words), ctypes.c_int64(n), ctypes.c_int64(n_perm), _p(out))
    if rc:
        raise ValueError("orc_perm_numpy: bad arguments")
    return out, words


def raw_uint32(words: np.ndarray, count: int) -> np.ndarray:
    words = np.array(words, dtype=np.uint64).copy()
    out = np.empty(count, dtype=np.uint32)
    clib().orc_raw_uint32(_p(words), ctypes.c_int64(count), _p(out))
    return out


def philox4x32_10(counter: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123):
    counter (..., 4) uint32 -> (..., 4) uint32.  Pinned by Random123's published known answers (tests)."""
    c = np.array(counter, dtype=np.uint64).reshape(-1, 4).T.copy()
    k0, k1 = np.uint64(key[0] & 0xFFFFFFFF), np.uint64(key[1] & 0xFFFFFFFF)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32
        c = np.stack([n0, p1 & m32, n2, p0 & m32])
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c.T.astype(np.uint32).reshape(np.shape(counter))


def counter_permutation(seed: int, n: int, p: int) -> np.ndarray:
    """EXTENSION (no reference semantics; defined in include/spatialcore_hip.h, sc_perm_generate_counter): permutation
    p of the counter-based source -- Fisher-Yates as numpy runs it (i = n-1 .. 1: swap a[i], a[j]) with
    j = Lemire-bounded(u, i + 1), u = first two words of Philox4x32-10(key = seed words, counter = (i, retry, p lo, p hi))."""
    a = np.arange(n, dtype=np.int32)
    if n < 2:
        return a
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    i = np.arange(n - 1, 0, -1, dtype=np.uint64)
    ctr = np.stack([i, np.zeros_like(i), np.full_like(i, p & 0xFFFFFFFF), np.full_like(i, (p >> 32) & 0xFFFFFFFF)], axis=1)
    out = philox4x32_10(ctr.astype(np.uint32), key)
    J = np.empty(n - 1, dtype=np.int64)
    for s in range(n - 1):                                   # python integers: the 128-bit product of Lemire's method
        rng_ = int(i[s]) + 1
        u, r = (int(out[s, 1]) << 32) | int(out[s, 0]), 0
        while True:
            m = u * rng_
            low = m & 0xFFFFFFFFFFFFFFFF
            if low >= rng_ or low >= ((1 << 64) - rng_) % rng_:
                break
            r += 1
            o = philox4x32_10(np.array([[int(i[s]), r, p & 0xFFFFFFFF, (p >> 32) & 0xFFFFFFFF]], dtype=np.uint32), key)[0]
            u = (int(o[1]) << 32) | int(o[0])
        J[s] = m >> 64
    for s in range(n - 1):
        ii, j = n - 1 - s, J[s]
        a[ii], a[j] = a[j], a[ii]
    return a


# =============================================================================================
# A1 / A2 -- neighbour search
# =============================================================================================


def knn_bruteforce(coords: np.ndarray, k: int, include_self: bool = False) -> np.ndarray:
    """Exact kNN ordered by (squared distance, index); equals sklearn ball_tree/kd_tree and scipy
    cKDTree on tie-free input.  AC:393-401, NB:213-228."""
    xy = np.ascontiguousarray(coords, dtype=np.float64)
    n = xy.shape[0]
    idx = np.empty((n, k), dtype=np.int32)
    rc = clib().orc_knn_bruteforce(_p(xy), ctypes.c_int64(n), ctypes.c_int(k),
                                   ctypes.c_int(int(include_self)), _p(idx), None)
    if rc:
        raise ValueError("k out of range")
    return idx


def knn_tree(coords: np.ndarray, k: int) -> np.ndarray:
    """Larger-n oracle through scipy's cKDTree (k+1 query, self dropped by index as NB:223-228)."""
    from scipy.spatial import cKDTree

    coords = np.ascontiguousarray(coords, dtype=np.float64)
    n = coords.shape[0]
    _, nbr = cKDTree(coords).query(coords, k=k + 1)
    out = np.empty((n, k), dtype=np.int32)
    rows = np.arange(n)
    is_self = nbr == rows[:, None]
    # drop the self hit (or the last column when self is not among the k+1, e.g. duplicates)
    drop = np.where(is_self.any(axis=1), is_self.argmax(axis=1), k)
    keep = np.ones_like(nbr, dtype=bool)
    keep[rows, drop] = False
    out[:] = nbr[keep].reshape(n, k)
    return out


def radius_neighbors(coords: np.ndarray, radius: float) -> Tuple[np.ndarray, np.ndarray]:
    """Closed ball d <= r, self removed, ascending index per row.  NB:241-244 (cKDTree
    query_ball_point; the reference consumes the lists as unordered sets)."""
    from scipy.spatial import cKDTree

    coords = np.ascontiguousarray(coords, dtype=np.float64)
    lists = cKDTree(coords).query_ball_point(coords, r=radius)
    indptr = np.zeros(len(lists) + 1, dtype=np.int64)
    cols: List[np.ndarray] = []
    for i, nb in enumerate(lists):
        a = np.sort(np.array([j for j in nb if j != i], dtype=np.int32))
        cols.append(a)
        indptr[i + 1] = indptr[i] + a.size
    indices = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int32)
    return indptr, indices.astype(np.int32)


# =============================================================================================
# A3 -- weights
# =============================================================================================


def reference_weights(coords: np.ndarray, k: int, include_self: bool = False) -> csr_matrix:
    """Row-normalised float32 CSR exactly as ``build_spatial_weights`` assembles it (AC:398-413):
    binary float32 COO -> CSR -> multiply by float32 1/rowsum -> CSR."""
    n = coords.shape[0]
    if include_self:
        nbr = knn_bruteforce(coords, k + 1, include_self=True)
    else:
        nbr = knn_bruteforce(coords, k, include_self=False)
    m = nbr.shape[1]
    rows = np.repeat(np.arange(n), m)
    data = np.ones(rows.size, dtype=np.float32)
    W = csr_matrix((data, (rows, nbr.reshape(-1))), shape=(n, n))
    rs = np.array(W.sum(axis=1)).flatten()
    rs[rs == 0] = 1
    W = W.multiply(1.0 / rs[:, np.newaxis])
    return W.tocsr()


def squidpy_connectivities(coords: np.ndarray, k: int) -> csr_matrix:
    """[upstream squidpy _build_connectivity, generic coords] binary float64 CSR, k per row, no
    self, not symmetrised.  Used by the reference at AC:565-570."""
    n = coords.shape[0]
    nbr = knn_bruteforce(coords, k, include_self=False)
    rows = np.repeat(np.arange(n), k)
    return csr_matrix((np.ones(rows.size, dtype=np.float64), (rows, nbr.reshape(-1))), shape=(n, n))


def row_normalize_l1(g: csr_matrix) -> csr_matrix:
    """[upstream] sklearn.preprocessing.normalize(g, norm='l1', axis=1): data / sum|row|."""
    g = csr_matrix(g, dtype=np.float64, copy=True)
    g.sort_indices()
    rs = np.add.reduceat(np.abs(g.data), g.indptr[:-1]) if g.nnz else np.zeros(g.shape[0])
    rs = np.asarray(rs, dtype=np.float64)
    counts = np.diff(g.indptr)
    rs[counts == 0] = 0.0
    scale = np.where(rs == 0, 1.0, rs)
    g.data = g.data / np.repeat(scale, counts)
    return g


def graph_moments(g: csr_matrix) -> Tuple[float, float, float]:
    """[upstream squidpy _g_moments] s0 = sum w; s1 = sum (w + w^T)^2 / 2; s2 = sum_i (row_i + col_i)^2."""
    g = csr_matrix(g, dtype=np.float64)
    s0 = float(g.sum())
    t = g.transpose() + g
    s1 = float(t.multiply(t).sum() / 2.0)
    s2 = float(((np.asarray(g.sum(1)).ravel() + np.asarray(g.sum(0)).ravel()) ** 2).sum())
    return s0, s1, s2


# =============================================================================================
# A5 / A6 / A7 -- global Moran's I as squidpy/scanpy compute it for AC:576-583
# =============================================================================================


def _csr_parts(g: csr_matrix):
    g = csr_matrix(g)
    return (np.ascontiguousarray(g.indptr, dtype=np.int64),
            np.ascontiguousarray(g.indices, dtype=np.int32),
            np.ascontiguousarray(g.data, dtype=np.float64))


def dense_genes(X, cols: Optional[Sequence[int]] = None) -> np.ndarray:
    """(G, N) float64 gene-major dense values (scanpy casts to float64)."""
    if cols is not None:
        X = X.tocsc()[:, list(cols)] if sparse.issparse(X) else np.asarray(X)[:, list(cols)]
    A = X.toarray() if sparse.issparse(X) else np.asarray(X)
    return np.ascontiguousarray(A.T, dtype=np.float64)


def morans_i_scores(g: csr_matrix, vals: np.ndarray, idx: Optional[np.ndarray] = None) -> np.ndarray:
    """[upstream scanpy _morans_i_mtx] vals (G, N); literal CSR sweep, optionally on g[idx, :]."""
    indptr, indices, data = _csr_parts(g)
    n = g.shape[0]
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    out = np.empty(vals.shape[0], dtype=np.float64)
    ip = None
    if idx is not None:
        idx32 = np.ascontiguousarray(idx, dtype=np.int32)
        ip = _p(idx32)
    clib().orc_moran_rowperm(_p(indptr), _p(indices), _p(data), ctypes.c_int64(n), _p(vals),
                             ctypes.c_int64(vals.shape[0]), ip, _p(out))
    return out


def morans_i_sims_literal(g: csr_matrix, vals: np.ndarray, n_perms: int, seed: int) -> np.ndarray:
    """[upstream squidpy _score_helper, n_jobs=1 => one chunk, ix=0]: rng = default_rng(seed + 0);
    per permutation idx = rng.permutation(N); score(g[idx, :], vals)."""
    n = g.shape[0]
    perms, _ = perm_table(int(seed), n, n_perms)
    sims = np.empty((n_perms, vals.shape[0]), dtype=np.float64)
    for p in range(n_perms):
        sims[p] = morans_i_scores(g, vals, perms[p])
    return sims


def moran_operands(g: csr_matrix, vals: np.ndarray):
    """z = x - mean, lag = g @ z (row-sequential fp64), scale = N / W / sum z^2 (per gene)."""
    indptr, indices, data = _csr_parts(g)
    n = g.shape[0]
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    z = vals - vals.mean(axis=1, keepdims=True)
    lag = np.empty_like(z)
    for gi in range(z.shape[0]):
        clib().orc_csr_lag(_p(indptr), _p(indices), _p(data), ctypes.c_int64(n), _p(z[gi]), _p(lag[gi]))
    W = float(np.sum(data))
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = n / W / (z * z).sum(axis=1)
    return z, lag, scale


def morans_i_sims_gather(g: csr_matrix, vals: np.ndarray, perms: np.ndarray) -> np.ndarray:
    """Gather form: sims[p, g] = scale_g * sum_i z_g[i] * lag_g[perm_p[i]]."""
    z, lag, scale = moran_operands(g, vals)
    perms = np.ascontiguousarray(perms, dtype=np.int32)
    sims = np.empty((perms.shape[0], z.shape[0]), dtype=np.float64)
    clib().orc_gather_dot(_p(z), _p(lag), ctypes.c_int64(z.shape[1]), ctypes.c_int64(z.shape[0]),
                          _p(perms), ctypes.c_int64(perms.shape[0]), _p(scale), _p(sims))
    return sims


def lattice_genes(g: csr_matrix, vals: np.ndarray) -> np.ndarray:
    """Genes whose permutation statistic lives on an integer lattice: every value an integer count in [0, 65535] and
    every stored weight of the graph equal AND every row of the same degree d (a kNN graph after l1 row normalisation:
    w = 1/k, d = k).  With S = A x (A = 0/1 adjacency) sum_i z[i] lag[perm[i]] = w (T_perm - mean * sum S) in exact
    arithmetic, T_perm = sum_i x[i] S[perm[i]] an integer: a permutation can TIE the observed value exactly.  (With
    unequal degrees the term -w mean sum_i (x[i] - mean) deg[perm[i]] is left over and depends on the permutation: such
    graphs -- a binary adjacency of a radius graph, say -- take the ordinary arithmetic.)  The reference's float loop (scanpy's numba kernel
    behind AC:576-583) decides such ties by the rounding noise of its summation order -- nothing reproducible; the
    exact-arithmetic outcome (a tie counts as >=, as `sims >= I` reads) is what this oracle pins."""
    vals = np.asarray(vals)
    g = csr_matrix(g)
    data = g.data
    deg = np.diff(g.indptr)
    uniform = data.size > 0 and bool((data == data[0]).all()) and data[0] > 0 and int(deg.min()) == int(deg.max())
    ok = np.zeros(vals.shape[0], dtype=bool)
    if uniform:
        with np.errstate(invalid="ignore"):
            ok = ((vals >= 0) & (vals <= 65535) & (vals == np.floor(vals))).all(axis=1)
    return ok


def morans_count_ge(g: csr_matrix, vals: np.ndarray, perms: np.ndarray, sims: Optional[np.ndarray] = None,
                    score: Optional[np.ndarray] = None):
    """#{p : sims_p >= I} per gene.  Lattice genes (see lattice_genes): decided on the exact integers T_p >= T_obs.
    Other genes: the float comparison of the gather form.  Returns (count, is_lattice)."""
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    perms = np.ascontiguousarray(perms, dtype=np.int32)
    lat = lattice_genes(g, vals)
    if sims is None:
        sims = morans_i_sims_gather(g, vals, perms)
    if score is None:
        score = morans_i_scores(g, vals)
    with np.errstate(invalid="ignore"):
        count = (sims >= score).sum(axis=0).astype(np.int64)
    if lat.any():
        n = vals.shape[1]
        A = csr_matrix(g).copy()
        A.data = np.ones_like(A.data)
        x = np.ascontiguousarray(vals[lat].astype(np.int64))
        S = np.ascontiguousarray(np.rint(A @ x.T.astype(np.float64)).T.astype(np.int64))
        T = np.empty((perms.shape[0], x.shape[0]), dtype=np.int64)
        T0 = np.empty((1, x.shape[0]), dtype=np.int64)
        lib = clib()
        lib.orc_lattice_T(_p(x), _p(S), ctypes.c_int64(n), ctypes.c_int64(x.shape[0]), _p(perms),
                          ctypes.c_int64(perms.shape[0]), _p(T))
        lib.orc_lattice_T(_p(x), _p(S), ctypes.c_int64(n), ctypes.c_int64(x.shape[0]), None, ctypes.c_int64(1), _p(T0))
        exact = (T >= T0).sum(axis=0).astype(np.int64)
        exact[vals[lat].var(axis=1) == 0] = 0          # zero-variance gene: I is NaN, nothing counts
        count[lat] = exact
    return count, lat


def analytic_pval(score: np.ndarray, g: csr_matrix) -> Tuple[np.ndarray, float]:
    """[upstream squidpy _analytic_pval, mode='moran', two_tailed=False]."""
    s0, s1, s2 = graph_moments(g)
    n = g.shape[0]
    v_num = n * n * s1 - n * s2 + 3 * s0 * s0
    v_den = (n - 1) * (n + 1) * s0 * s0
    var_norm = v_num / v_den - (1.0 / (n - 1)) ** 2
    z_norm = (score - (-1.0 / (n - 1))) / var_norm ** 0.5
    p = np.empty(score.shape)
    pos = z_norm > 0
    p[pos] = 1 - stats.norm.cdf(z_norm[pos])
    p[~pos] = stats.norm.cdf(z_norm[~pos])
    return p, float(var_norm)


def pvalues_squidpy(score: np.ndarray, sims: Optional[np.ndarray], g: csr_matrix,
                    count_ge: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """[upstream squidpy _p_value_calc].  count_ge: #{sims >= score} decided elsewhere (morans_count_ge: exact for
    lattice genes) instead of by the float comparison."""
    p_norm, var_norm = analytic_pval(score, g)
    res = {"pval_norm": p_norm, "var_norm": np.full(score.shape, var_norm)}
    if sims is None:
        return res
    P = sims.shape[0]
    large = (sims >= score).sum(axis=0) if count_ge is None else np.array(count_ge, dtype=np.int64)
    flip = (P - large) < large
    large[flip] = P - large[flip]
    res["pval_sim"] = (large + 1) / (P + 1)
    e = sims.sum(axis=0) / P
    se = sims.std(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        zs = (score - e) / se
    pz = np.empty(zs.shape)
    pos = zs > 0
    pz[pos] = 1 - stats.norm.cdf(zs[pos])
    pz[~pos] = stats.norm.cdf(zs[~pos])
    res["pval_z_sim"] = pz
    res["var_sim"] = np.var(sims, axis=0)
    return res


def morans_i_reference_table(coords, X, gene_cols, k, n_permutations, seed,
                             graph: Optional[csr_matrix] = None, literal: bool = False):
    """What AC:565-625 stores in ``uns[key_added]``: per gene (input order)
    I, expected_I, z_score, p_value.  Returns dict of arrays (+ 'sims' and 'perms')."""
    n = X.shape[0]
    conn = squidpy_connectivities(coords, k) if graph is None else csr_matrix(graph, dtype=np.float64)
    g = row_normalize_l1(conn)
    vals = dense_genes(X, gene_cols)
    score = morans_i_scores(g, vals)
    sims = perms = count_ge = lattice = None
    if n_permutations > 0:
        perms, _ = perm_table(int(seed), n, n_permutations)
        sims = morans_i_sims_literal(g, vals, n_permutations, seed) if literal \
            else morans_i_sims_gather(g, vals, perms)
        count_ge, lattice = morans_count_ge(g, vals, perms, sims, score)
    pv = pvalues_squidpy(score, sims, g, count_ge)
    expected = -1.0 / (n - 1)
    var_norm = pv["var_norm"]
    z_score = np.where(var_norm > 0, (score - expected) / np.sqrt(np.where(var_norm > 0, var_norm, 1)), 0.0)
    p_value = pv["pval_sim"] if n_permutations > 0 else pv["pval_norm"]
    return {"I": score, "expected_I": np.full(score.shape, expected), "z_score": z_score,
            "p_value": np.asarray(p_value, dtype=np.float64), "sims": sims, "perms": perms,
            "count_ge": count_ge, "lattice": lattice, "var_norm": var_norm, "graph": g, "connectivities": conn, **{k_: v for k_, v in pv.items()}}


# =============================================================================================
# A8 -- Lee's L (AC:273-334, 1113-1155)
# =============================================================================================


def lees_l_core(z_x, z_y, W, n_permutations: int, rng: np.random.Generator):
    """Literal numpy restatement of ``_compute_lees_l_core`` (AC:307-332)."""
    lag = np.asarray(W @ z_y).ravel()
    L_local = z_x * lag
    L = float(L_local.sum())
    p = 1.0
    L_perm = np.zeros(n_permutations)
    if n_permutations > 0:
        for i in range(n_permutations):
            zp = rng.permutation(z_y)
            L_perm[i] = (z_x * np.asarray(W @ zp).ravel()).sum()
        p = float((np.sum(np.abs(L_perm) >= np.abs(L)) + 1) / (n_permutations + 1))
    return L_local, L, lag, p, L_perm


def lees_l(coords, X, pairs: Sequence[Tuple[int, int]], k: int, n_permutations: int, seed: int):
    """``lees_l`` for column-index pairs (AC:1096-1155): one rng for all pairs, population-std
    standardisation, zero-variance pairs consume no random numbers."""
    W = reference_weights(coords, k)
    Xc = X.tocsc() if sparse.issparse(X) else np.asarray(X)
    rng = np.random.default_rng(seed)
    out = []
    for ix, iy in pairs:
        col = (lambda j: np.asarray(Xc[:, j].toarray()).ravel()) if sparse.issparse(Xc) \
            else (lambda j: np.asarray(Xc[:, j]).ravel())
        x, y = col(ix), col(iy)
        sx, sy = x.std(), y.std()
        if sx == 0 or sy == 0:
            out.append({"L": 0.0, "p_value": 1.0, "L_perm": np.zeros(0)})
            continue
        zx, zy = (x - x.mean()) / sx, (y - y.mean()) / sy
        _, L, _, p, L_perm = lees_l_core(zx, zy, W, n_permutations, rng)
        out.append({"L": L, "p_value": p, "L_perm": L_perm})
    return out


# =============================================================================================
# A7 -- multiple testing + quadrants (AC:132-183, 219-265)
# =============================================================================================


def fdr_bh(p: np.ndarray) -> np.ndarray:
    n = p.size
    if n == 0:
        return p.copy()
    order = np.argsort(p)
    adj = p[order] * n / np.arange(1, n + 1)
    adj = np.minimum.accumulate(adj[::-1])[::-1]
    out = np.empty(n)
    out[order] = adj
    return np.clip(out, 0, 1)


def bonferroni(p: np.ndarray) -> np.ndarray:
    return np.clip(p * p.size, 0, 1) if p.size else p.copy()


def quadrants(z, lag, p=None, alpha=0.05) -> np.ndarray:
    q = np.zeros(np.shape(z), dtype=np.int8)
    q[(z > 0) & (lag > 0)] = 1
    q[(z < 0) & (lag < 0)] = 2
    q[(z > 0) & (lag < 0)] = 3
    q[(z < 0) & (lag > 0)] = 4
    if p is not None:
        q[p >= alpha] = 0
    return q


# =============================================================================================
# N1 -- Local Moran's I (AC:804-934), float32 arithmetic as the reference
# =============================================================================================


def local_morans_i(coords, X, gene_cols, k, n_permutations, seed, fdr="fdr_bh", alpha=0.05,
                   batch_size=100):
    n = X.shape[0]
    W = reference_weights(coords, k)
    Xs = sparse.csc_matrix(X) if not sparse.issparse(X) else X.tocsc()
    gene_cols = np.asarray(gene_cols)
    Xg = Xs[:, gene_cols]
    mean = np.asarray(Xg.mean(axis=0)).ravel()
    sq_mean = np.asarray(Xg.power(2).mean(axis=0)).ravel()
    means = mean.astype(np.float32)
    stds = np.sqrt(sq_mean - mean ** 2).astype(np.float32)
    zero = stds == 0
    stds[zero] = 1.0
    G = gene_cols.size
    I = np.zeros((n, G), np.float32)
    Z = np.zeros((n, G), np.float32)
    LAG = np.zeros((n, G), np.float32)
    Pv = np.ones((n, G), np.float32)
    rng = np.random.default_rng(seed)
    for b0 in range(0, G, batch_size):
        b1 = min(b0 + batch_size, G)
        Xb = Xs[:, gene_cols[b0:b1]].toarray().astype(np.float32)
        Zb = (Xb - means[b0:b1]) / stds[b0:b1]
        Z[:, b0:b1] = Zb
        lagb = W @ Zb
        LAG[:, b0:b1] = lagb
        I[:, b0:b1] = Zb * lagb
        if n_permutations > 0:
            cnt = np.zeros((n, b1 - b0), dtype=np.int64)
            absI = np.abs(I[:, b0:b1])
            for _ in range(n_permutations):
                pi = rng.permutation(n)
                Zs = Zb[pi, :]
                cnt += np.abs(Zs * (W @ Zs)) >= absI
            # (extreme + 1) / (P + 1) evaluated in float64, stored float32 (AC:894-896)
            Pv[:, b0:b1] = ((cnt + 1) / (n_permutations + 1)).astype(np.float32)
    if zero.any():
        I[:, zero] = 0.0
        Z[:, zero] = 0.0
        LAG[:, zero] = 0.0
        Pv[:, zero] = 1.0
    if n_permutations > 0:
        Padj = np.ones_like(Pv)
        for gi in range(G):
            col = Pv[:, gi]
            Padj[:, gi] = {"fdr_bh": fdr_bh, "bonferroni": bonferroni, "none": np.copy}[fdr](col)
        Q = quadrants(Z, LAG, Padj, alpha)
    else:
        Padj = Pv
        Q = quadrants(Z, LAG, None, alpha)
    return {"I": I, "z": Z, "lag": LAG, "p": Pv, "p_adj": Padj, "quadrant": Q,
            "zero_variance": zero}


# =============================================================================================
# A9 -- neighbourhood composition (NB:196-264)
# =============================================================================================


def neighborhood_profile(coords, labels, method="knn", k=15, radius=None, normalize=True):
    labels = np.asarray(labels)
    cats = sorted(set(labels.tolist()))
    code = np.array([cats.index(v) for v in labels.tolist()], dtype=np.int64)
    n = labels.size
    prof = np.zeros((n, len(cats)), dtype=np.float32)
    if method == "knn":
        nbr = knn_bruteforce(coords, k)
        for j in range(k):
            np.add.at(prof, (np.arange(n), code[nbr[:, j]]), 1)
    else:
        indptr, indices = radius_neighbors(coords, radius)
        rows = np.repeat(np.arange(n), np.diff(indptr))
        np.add.at(prof, (rows, code[indices]), 1)
    rs = prof.sum(axis=1)
    if (rs == 0).any():
        raise ValueError(f"{int((rs == 0).sum())} cells have empty neighborhood profiles.")
    if normalize:
        prof = prof / rs[:, None]
    return prof, cats


# =============================================================================================
# N4 -- label-permutation enrichment.  NOT in the reference: this restates the definition given
# in spatialcore_amd.spatial.neighborhoods.neighborhood_enrichment ("parity unpinned" by nature).
# =============================================================================================


def enrichment_counts(indptr, indices, codes, n_types, perms):
    """counts[p, a, b] over edges i -> j with labels codes[perm_p]; last slice = observed."""
    codes = np.asarray(codes)
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    out = np.zeros((len(perms) + 1, n_types, n_types), dtype=np.int64)
    for p in range(len(perms) + 1):
        lab = codes if p == len(perms) else codes[perms[p]]
        np.add.at(out[p], (lab[rows], lab[indices]), 1)
    return out
