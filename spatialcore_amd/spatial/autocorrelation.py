"""Spatial autocorrelation on MI355X: Moran's I and Lee's L with permutation significance.

Drop-in mirror of the reference's ``spatialcore.spatial.autocorrelation`` public functions
(same names, keyword arguments, defaults, outputs and error messages;
reference src/spatialcore/spatial/autocorrelation.py, abbreviated ``AC`` below).  All arithmetic on
the path runs in hand-written HIP kernels behind the C ABI of ``include/spatialcore_hip.h``; there is
no CPU fallback -- without the shared library or a gfx950 device the functions raise.

Extra keywords (keyword-only, defaults preserve reference behaviour): ``device`` = GPU ordinal; ``radius`` on
``morans_i`` / ``lees_l`` (closed-ball radius graph instead of kNN).

Limit of the device path that the reference does not have: 2-D coordinates only (``ValueError`` otherwise; BASELINE's
north_star is 2-D).  ``n_neighbors`` is unlimited (k <= 32: register top-k; above: per-cell heaps in device memory), and
so is ``n_cells`` up to 2**31 - 1 (from 2**25 cells the permutation kernel gathers fp64 rows with 64-bit addresses).
"""

from __future__ import annotations

import time
from typing import List, Optional, Tuple, Union

import numpy as np
import pandas as pd
from scipy import sparse, stats
from scipy.sparse import csr_matrix

from spatialcore_amd import _lib
from spatialcore_amd._logging import get_logger
from spatialcore_amd._metadata import update_metadata

logger = get_logger("spatial.autocorrelation")

# Quadrant encoding (AC:57-58): 0=NS, 1=HH, 2=LL, 3=HL, 4=LH
QUADRANT_LABELS = {0: "NS", 1: "HH", 2: "LL", 3: "HL", 4: "LH"}


# =============================================================================================
# helpers
# =============================================================================================


def _require_spatial(adata, spatial_key: str) -> np.ndarray:
    if spatial_key not in adata.obsm:
        raise ValueError(f"adata.obsm['{spatial_key}'] not found. Spatial coordinates are required.")
    coords = np.asarray(adata.obsm[spatial_key])
    if coords.ndim != 2 or coords.shape[1] < 2:
        raise ValueError(f"adata.obsm['{spatial_key}'] must have shape (n_cells, 2), got {coords.shape}")
    if coords.shape[1] > 2:
        raise ValueError("only 2-D coordinates are supported by the MI355X path "
                         f"(adata.obsm['{spatial_key}'] has {coords.shape[1]} columns)")
    return np.ascontiguousarray(coords, dtype=np.float64)


def _check_counts(n_neighbors: int, n_permutations: int) -> None:
    if n_neighbors < 1:
        raise ValueError(f"n_neighbors must be >= 1, got {n_neighbors}")
    if n_permutations < 0:
        raise ValueError(f"n_permutations must be >= 0, got {n_permutations}")


def _resolve_genes(adata, genes, warn_suffix: str) -> List[str]:
    if genes is None:
        names = list(adata.var_names)
        logger.warning(f"No genes specified, analyzing all {len(names)} genes. {warn_suffix}")
    elif isinstance(genes, str):
        names = [genes]
    else:
        names = list(genes)
    missing = set(names) - set(adata.var_names)
    if missing:
        raise ValueError(f"Genes not found in adata.var_names: {list(missing)[:10]}")
    return names


def _expression(adata, layer: Optional[str]):
    return adata.layers[layer] if layer is not None else adata.X


def _unique_columns(adata, names: List[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Column ids of the distinct genes (first-seen order) and, per requested name, its slot."""
    slot, cols, where = {}, [], []
    for g in names:
        if g not in slot:
            slot[g] = len(cols)
            cols.append(int(adata.var_names.get_loc(g)))
        where.append(slot[g])
    return np.asarray(cols, dtype=np.int32), np.asarray(where, dtype=np.int64)


def _knn_weights_f32(ctx, coords: np.ndarray, n_neighbors: int, include_self: bool = False) -> float:
    """kNN on the device + the reference's float32 row-normalised weights (AC:393-413) as the active
    graph.  Returns the (float32-valued) weight."""
    k = n_neighbors + 1 if include_self else n_neighbors
    ctx.knn(coords, k, include_self=include_self, fetch=False)
    w = float(np.float32(1.0) / np.float32(k))
    ctx.graph_from_knn(w)
    return w


def _radius_weights(ctx, coords: np.ndarray, radius: float, float32_weights: bool):
    """EXTENSION (BASELINE configs[2]; the reference's Moran / Lee functions only take ``n_neighbors``, AC:342-347):
    the closed-ball radius graph of ``compute_neighborhood_profile`` (NB:241-251: d <= radius, self excluded) as the
    active graph, row-normalised like the kNN weights (1 / degree; float32-valued for the in-repo float32 paths,
    fp64 for the squidpy-style Moran path).  Cells without a neighbour keep an empty row (lag 0).
    Returns (indptr, indices, data) as uploaded."""
    if radius is None or not radius > 0:
        raise ValueError(f"radius must be > 0, got {radius}")
    indptr, indices = ctx.radius_graph(coords, float(radius))
    deg = np.diff(indptr)
    with np.errstate(divide="ignore"):
        w = (np.float32(1.0) / deg.astype(np.float32)).astype(np.float64) if float32_weights else 1.0 / deg
    data = np.repeat(w, deg)
    ctx.set_graph_csr(indptr, indices, data, coords.shape[0])
    return indptr, indices, data


# =============================================================================================
# FDR / quadrants (AC:132-265) -- O(n) bookkeeping on per-cell outputs
# =============================================================================================


def _classify_quadrants(z_values, lag_values, p_values=None, alpha: float = 0.05) -> np.ndarray:
    """LISA quadrants as int8 (AC:219-265): 1=HH, 2=LL, 3=HL, 4=LH, 0=NS / not significant."""
    q = np.zeros(np.shape(z_values), dtype=np.int8)
    q[(z_values > 0) & (lag_values > 0)] = 1
    q[(z_values < 0) & (lag_values < 0)] = 2
    q[(z_values > 0) & (lag_values < 0)] = 3
    q[(z_values < 0) & (lag_values > 0)] = 4
    if p_values is not None:
        q[p_values >= alpha] = 0
    return q


# =============================================================================================
# build_spatial_weights (AC:342-413)
# =============================================================================================


def build_spatial_weights(adata, n_neighbors: int = 6, spatial_key: str = "spatial",
                          include_self: bool = False, *, device: int = 0) -> csr_matrix:
    """Row-normalised sparse kNN weights, float32 CSR, each row sums to 1 (AC:342-413).

    Neighbours come from the exact GPU kNN (ordered by squared distance, then index: identical to
    the reference's ball tree on tie-free coordinates); columns are ascending within a row, as the
    reference's COO->CSR conversion leaves them.
    """
    coords = _require_spatial(adata, spatial_key)
    n_cells = coords.shape[0]
    logger.debug(f"Building spatial weights: {n_cells:,} cells, k={n_neighbors}")
    ctx = _lib.default_context(device)
    _knn_weights_f32(ctx, coords, n_neighbors, include_self)
    indptr, indices, data = ctx.get_graph()
    W = csr_matrix((data.astype(np.float32), indices, indptr), shape=(n_cells, n_cells))
    logger.debug(f"Spatial weights: nnz={W.nnz:,}")
    return W


# =============================================================================================
# Global Moran's I (AC:421-648)
# =============================================================================================


def _knn_device_graph(ctx, coords, n_neighbors: int):
    """Device half of squidpy's neighbour step: exact kNN, and the row-normalised graph (squidpy's
    ``transformation=True``) as the active device graph.  Returns the neighbour indices and squared distances."""
    idx, rd = ctx.knn(coords, n_neighbors, return_distance=True)
    ctx.graph_from_knn(1.0 / n_neighbors)
    return idx, rd


def _knn_device_graph_async(ctx, coords, n_neighbors: int) -> None:
    """The same without waiting and without fetching: search and graph are enqueued, the neighbour lists stay on the
    device for ``ctx.knn_fetch()``."""
    ctx.knn(coords, n_neighbors, fetch=False)
    ctx.graph_from_knn(1.0 / n_neighbors)


def _record_squidpy_neighbors(adata, idx, rd, n_neighbors: int) -> None:
    """Host half: what ``sq.gr.spatial_neighbors(adata, n_neighs=k, coord_type='generic')`` leaves behind
    (AC:565-570) [upstream squidpy]: binary float64 connectivities + euclidean distances in ``adata.obsp`` and a
    ``uns['spatial_neighbors']`` record."""
    n = idx.shape[0]
    indptr = np.arange(0, n * n_neighbors + 1, n_neighbors, dtype=np.int64)
    conn = csr_matrix((np.ones(idx.size, dtype=np.float64), idx.reshape(-1), indptr), shape=(n, n))
    dist = csr_matrix((np.sqrt(rd).reshape(-1), idx.reshape(-1), indptr.copy()), shape=(n, n))
    adata.obsp["spatial_connectivities"] = conn
    adata.obsp["spatial_distances"] = dist
    adata.uns["spatial_neighbors"] = {
        "connectivities_key": "spatial_connectivities",
        "distances_key": "spatial_distances",
        "params": {"n_neighbors": n_neighbors, "coord_type": "generic", "radius": None, "transform": None},
    }


def _squidpy_neighbors(ctx, adata, coords, n_neighbors: int, spatial_key: str):
    idx, rd = _knn_device_graph(ctx, coords, n_neighbors)
    _record_squidpy_neighbors(adata, idx, rd, n_neighbors)


def _beside(device_call, host_call):
    """Run one blocking library call (ctypes releases the GIL) beside host-side assembly that does not touch the
    device context; the host work always completes, then either side's exception propagates (the device call's first:
    it is the later step of the sequential order)."""
    import threading

    box = {}

    def run():
        try:
            box["value"] = device_call()
        except BaseException as exc:   # re-raised in the caller's thread
            box["error"] = exc

    worker = threading.Thread(target=run, name="spatialcore-upload")
    worker.start()
    try:
        host_call()
    finally:
        worker.join()
    if "error" in box:
        raise box["error"]
    return box.get("value")


def _upload_existing_graph(ctx, g) -> None:
    """Row-normalise (l1, as squidpy's transformation=True does) and upload a user graph."""
    g = csr_matrix(g, dtype=np.float64, copy=True)
    g.sum_duplicates()
    g.sort_indices()
    rs = np.asarray(abs(g).sum(axis=1)).ravel()
    rs[rs == 0] = 1.0
    g.data = g.data / np.repeat(rs, np.diff(g.indptr))
    ctx.set_graph_csr(g.indptr, g.indices, g.data, g.shape[0])


def _norm_sf_cdf(z: np.ndarray) -> np.ndarray:
    """[upstream squidpy] one-sided normal p: 1 - cdf(z) for z > 0, cdf(z) otherwise."""
    p = np.empty(z.shape)
    pos = z > 0
    p[pos] = 1 - stats.norm.cdf(z[pos])
    p[~pos] = stats.norm.cdf(z[~pos])
    return p


def _moran_gene_batch(n_cells: int, requested: Optional[int]) -> int:
    """Genes per device batch of ``morans_i``: X, Z, Lag tiles + the narrow copy = ~28 bytes per (cell, gene)."""
    if requested is not None:
        if requested < 1:
            raise ValueError(f"gene_batch must be >= 1, got {requested}")
        return int(requested)
    fit = int((128 << 30) // (28 * max(n_cells, 1)))
    return max(64, fit // 64 * 64)


def _moran_resident(ctx, n_cells: int, n_permutations: int, seed: int, reuse_table: bool = False, begun=None) -> dict:
    """Global Moran's I on operands already resident on the device (graph + expression tiles):
    numpy-exact permutation table -> lag / permutation kernels -> p-value assembly as squidpy's
    ``_p_value_calc`` / ``_analytic_pval`` do [upstream].  Shared by ``morans_i`` and ``bench.py``.
    ``reuse_table``: score against the permutation table an earlier call with the same seed left on the device
    (gene batches of one ``morans_i`` call share the table, as squidpy's permutations are shared by all genes).
    ``begun``: the generator state words of a job already started with ``ctx.moran_seeded_begin``."""
    if n_permutations > 0 and reuse_table:
        out = ctx.moran(n_permutations, return_sims=False)
    elif n_permutations > 0 and begun is not None:
        # the graph moments (transpose + reverse-edge search on the side stream) are collected HERE, a host wait of a few
        # milliseconds in front of the scoring's set-up.  r04 measured the set-up without it (it starts the moments itself):
        # the scoring began ~4 ms earlier and the step was 4 ms LONGER (161.6 / 162.5 vs 158.0 / 157.2 ms, same box) -- like
        # every other attempt to pull the scoring's full-chip prelude kernels forward into the generator's first units.
        ctx.graph_moments()
        out = ctx.moran_seeded_finish(begun, return_sims=False)
    elif n_permutations > 0:
        # squidpy: default_rng(seed + chunk index), one chunk when n_jobs=1.  Table generation and
        # scoring are pipelined on the device (sc_moran_seeded).
        words = _lib.rng_state_words(np.random.default_rng(seed))
        out = ctx.moran_seeded(words, n_permutations, return_sims=False)
    else:
        out = ctx.moran(0, return_sims=False)
    score = out["I"]
    s0, s1, s2 = ctx.graph_moments()
    n = float(n_cells)
    expected_I = -1 / (n_cells - 1)
    var_norm = (n * n * s1 - n * s2 + 3 * s0 * s0) / ((n - 1) * (n + 1) * s0 * s0) - (1.0 / (n - 1)) ** 2
    with np.errstate(invalid="ignore", divide="ignore"):
        p_norm = _norm_sf_cdf((score - expected_I) / np.sqrt(var_norm))
    res = {"I": score, "expected_I": expected_I, "var_norm": var_norm, "pval_norm": p_norm, "p_value": p_norm}
    if n_permutations > 0:
        large = out["count_ge"].copy()
        flip = (n_permutations - large) < large
        large[flip] = n_permutations - large[flip]
        res["count_ge"] = out["count_ge"]
        res["pval_sim"] = res["p_value"] = (large + 1) / (n_permutations + 1)
        mean_sim = out["sim_sum"] / n_permutations
        res["var_sim"] = np.maximum(out["sim_sumsq"] / n_permutations - mean_sim ** 2, 0.0)
        with np.errstate(invalid="ignore", divide="ignore"):
            res["pval_z_sim"] = _norm_sf_cdf((score - mean_sim) / np.sqrt(res["var_sim"]))
    return res


def morans_i(
    adata,
    genes: Optional[Union[str, List[str]]] = None,
    layer: Optional[str] = None,
    spatial_key: str = "spatial",
    n_neighbors: int = 6,
    n_permutations: int = 10,
    seed: int = 0,
    key_added: str = "morans_i",
    copy: bool = False,
    use_existing_graph: bool = False,
    *,
    device: int = 0,
    radius: Optional[float] = None,
    gene_batch: Optional[int] = None,
):
    """Global Moran's I with permutation p-values (AC:421-648).

    Same contract as the reference: results go to ``adata.uns[key_added]`` as a DataFrame with
    columns ``gene, I, expected_I, z_score, p_value`` in input gene order; the kNN graph is left in
    ``adata.obsp['spatial_connectivities'|'spatial_distances']``; one provenance entry is appended.

    The reference delegates the arithmetic to squidpy (``spatial_neighbors`` +
    ``spatial_autocorr(mode='moran', n_perms=P, n_jobs=1, seed=seed)``, AC:565-583).  Here it runs on
    the GPU: exact kNN, row-normalised lag, and ``P`` permutations drawn from the numpy-exact
    ``default_rng(seed).permutation(n)`` stream, each scored as
    ``sum_i z_i * lag[perm[i]]`` (identical to scoring the row-permuted graph).

    Extensions (keyword-only, defaults keep the reference's behaviour): ``radius`` -- use the closed-ball radius graph
    (row-normalised) instead of the kNN graph, ``n_neighbors`` is then ignored; ``gene_batch`` -- genes resident on
    the device at a time (default: as many as fit ~128 GB of tiles; results do not depend on it).
    """
    start_time = time.time()
    coords = _require_spatial(adata, spatial_key)
    _check_counts(n_neighbors, n_permutations)
    adata = adata.copy() if copy else adata
    gene_names = _resolve_genes(adata, genes, "This may be slow for large datasets.")
    n_cells, n_genes = adata.n_obs, len(gene_names)
    logger.info(f"Computing Global Moran's I: {n_cells:,} cells, {n_genes} genes, "
                f"k={n_neighbors}, permutations={n_permutations}")
    # The reference calls sq.gr.spatial_autocorr without `genes=` (AC:576-583); squidpy then keeps only
    # adata.var['highly_variable'] genes when that column exists [upstream], and the reference's result
    # loop fails for every other requested gene (AC:617-621).  Same inputs, same failure.
    var = getattr(adata, "var", None)
    if var is not None and "highly_variable" in getattr(var, "columns", []):
        hv = np.asarray(var["highly_variable"].loc[gene_names].values, dtype=bool)
        if not hv.all():
            gene_name = gene_names[int(np.flatnonzero(~hv)[0])]
            raise RuntimeError(f"Gene '{gene_name}' was passed to squidpy but not found in results. "
                               "This indicates an internal error in squidpy or data corruption.")

    ctx = _lib.default_context(device)
    # The permutation generator needs nothing but n_cells and the seed: it is started FIRST and runs on its own
    # streams while the graph is built and the matrix crosses PCIe (squidpy: default_rng(seed + chunk index), one chunk
    # when n_jobs=1); the scoring below joins it chunk by chunk.
    begun = None
    if n_permutations > 0 and len(gene_names) > 0:
        begun = _lib.rng_state_words(np.random.default_rng(seed))
        ctx.moran_seeded_begin(begun, n_cells, n_permutations)
    try:
        return _morans_i_on_device(adata, ctx, coords, gene_names, layer, n_neighbors, n_permutations, seed, key_added,
                                   use_existing_graph, radius, gene_batch, begun, start_time)
    except BaseException:
        if begun is not None:
            ctx.moran_seeded_abort()
        raise


def _morans_i_on_device(adata, ctx, coords, gene_names, layer, n_neighbors, n_permutations, seed, key_added,
                        use_existing_graph, radius, gene_batch, begun, start_time):
    n_cells, n_genes = adata.n_obs, len(gene_names)
    knn_found = None
    if use_existing_graph and "spatial_connectivities" in adata.obsp:
        logger.info("Using existing spatial connectivity graph (use_existing_graph=True)")
        _upload_existing_graph(ctx, adata.obsp["spatial_connectivities"])
    elif radius is not None:
        logger.debug(f"Building spatial radius graph (r={radius})")
        indptr, indices, data = _radius_weights(ctx, coords, radius, float32_weights=False)
        adata.obsp["spatial_connectivities"] = csr_matrix((np.ones(indices.size), indices, indptr), shape=(n_cells, n_cells))
        adata.uns["spatial_neighbors"] = {"connectivities_key": "spatial_connectivities", "distances_key": None,
                                          "params": {"n_neighbors": None, "coord_type": "generic", "radius": float(radius),
                                                     "transform": None}}
    else:
        logger.debug(f"Building spatial neighbors graph (k={n_neighbors})")
        _knn_device_graph_async(ctx, coords, n_neighbors)   # enqueued; the lists are fetched beside the upload below
        knn_found = True

    cols, where = _unique_columns(adata, gene_names)
    # Genes are scored in batches that fit the device (four tile sets of n_cells x 8 bytes per gene, ~1/2 of the HBM
    # left to them); every batch after the first re-uses the permutation table the first one left on the device --
    # squidpy's permutations are shared by all genes too, so the result does not depend on the batching.
    X = _expression(adata, layer)
    per_batch = _moran_gene_batch(n_cells, gene_batch)
    score, p_all = np.empty(cols.size), np.empty(cols.size)
    if sparse.issparse(X) and cols.size > per_batch:
        X = X.tocsc()
    for b0 in range(0, cols.size, per_batch):
        part = cols[b0:b0 + per_batch]
        if sparse.issparse(X) and cols.size > per_batch:
            upload = lambda: ctx.set_expression(X[:, part], np.arange(part.size, dtype=np.int32))
        else:
            upload = lambda: ctx.set_expression(X, part)
        if knn_found is not None:
            # the first batch's upload (PCIe, host -> device) runs beside the fetch of the neighbour lists (device -> host,
            # on the library's copy stream) and the host-side assembly of squidpy's obsp / uns side effects
            _beside(upload, lambda: _record_squidpy_neighbors(adata, *ctx.knn_fetch(), n_neighbors))
            knn_found = None
        else:
            upload()
        res = _moran_resident(ctx, n_cells, n_permutations, seed, reuse_table=b0 > 0, begun=begun if b0 == 0 else None)
        score[b0:b0 + per_batch], p_all[b0:b0 + per_batch] = res["I"], res["p_value"]
    if knn_found is not None:   # (no gene batch ran)
        _record_squidpy_neighbors(adata, *ctx.knn_fetch(), n_neighbors)
    var_norm, expected_I = res["var_norm"], res["expected_I"]

    results = []
    for gene_name, u in zip(gene_names, where):
        I_value = float(score[u])
        z_score = float((I_value - expected_I) / np.sqrt(var_norm)) if var_norm > 0 else 0.0
        results.append({"gene": gene_name, "I": I_value, "expected_I": expected_I,
                        "z_score": z_score, "p_value": float(p_all[u])})
    adata.uns[key_added] = pd.DataFrame(results)

    elapsed = time.time() - start_time
    logger.info(f"Global Moran's I completed in {elapsed:.1f}s")
    update_metadata(
        adata,
        function_name="morans_i",
        parameters={
            "genes": gene_names[:10] if len(gene_names) > 10 else gene_names,
            "n_genes": n_genes,
            "n_neighbors": n_neighbors,
            "n_permutations": n_permutations,
            "use_existing_graph": use_existing_graph,
            "seed": seed,
            "backend": "hip_gfx950",
            "permgen_form": ctx.permgen_form(n_cells) if n_permutations > 0 else None,
        },
        outputs={"uns": key_added},
    )
    return adata


# =============================================================================================
# Global Lee's L (AC:991-1163)
# =============================================================================================


def _normalize_pairs(gene_pairs):
    single = False
    if isinstance(gene_pairs, tuple) and len(gene_pairs) == 2 and isinstance(gene_pairs[0], str):
        gene_pairs = [gene_pairs]
        single = True
    return list(gene_pairs), single


def lees_l(
    adata,
    gene_pairs: Union[Tuple[str, str], List[Tuple[str, str]]],
    layer: Optional[str] = None,
    spatial_key: str = "spatial",
    n_neighbors: int = 6,
    n_permutations: int = 199,
    seed: int = 0,
    *,
    device: int = 0,
    radius: Optional[float] = None,
    shared_permutations: bool = False,
    rng: str = "numpy",
) -> Union[dict, List[dict]]:
    """Global Lee's L bivariate spatial association with permutation p-values (AC:991-1163).

    Returns a dict ``{gene_x, gene_y, L, p_value}`` for a single pair, else a list of dicts.
    ``L = sum_i z_x[i] * (W z_y)[i]`` on population-std z-scores; the permutation loop shuffles
    ``z_y`` with the numpy-exact stream (one generator for all pairs, pairs with a zero-variance
    gene draw nothing) and is evaluated on the GPU as ``sum_j (W^T z_x)[j] * z_y[perm[j]]``.
    The whole pair loop is ONE device call (``sc_lee_seeded``): all observed statistics as a dense
    contraction on the fp64 matrix cores, permutation blocks generated and scored in a pipeline.
    Permutation statistics are fp64; for a float32 matrix the reported ``L`` is the reference's own float32 result
    (numpy's pairwise float32 sums and scipy's float32 mat-vec reproduced on the device).

    Extensions (keyword-only, defaults keep the reference's behaviour): ``radius`` -- radius graph instead of kNN;
    ``shared_permutations=True`` -- ONE block of ``n_permutations`` permutations (``default_rng(seed)``) is shared by
    all pairs instead of a fresh block per pair.  Every pair's null is still "y shuffled against x", but the p-values
    are no longer the reference's for the same seed (pairs after the first see other permutations there); in exchange
    a 100 x 100 screen at 1M cells costs seconds instead of minutes: the permutation statistics of the whole
    (distinct x genes) x (distinct y genes) grid become dense fp64 contractions on the matrix cores.
    ``rng="philox"`` (with ``shared_permutations=True`` only: that mode has no reference seed semantics to keep) draws
    the shared block from the counter-based source (``sc_perm_generate_counter``) instead of the numpy stream.
    """
    start_time = time.time()
    coords = _require_spatial(adata, spatial_key)
    _check_counts(n_neighbors, n_permutations)
    gene_pairs, single_pair = _normalize_pairs(gene_pairs)
    all_genes = set(g for pair in gene_pairs for g in pair)
    missing = all_genes - set(adata.var_names)
    if missing:
        raise ValueError(f"Genes not found in adata.var_names: {list(missing)}")
    if rng not in ("numpy", "philox") or (rng == "philox" and not shared_permutations):
        raise ValueError("rng must be 'numpy', or 'philox' together with shared_permutations=True "
                         "(per-pair permutations follow the reference's numpy stream)")
    n_cells, n_pairs = adata.n_obs, len(gene_pairs)
    logger.info(f"Computing Global Lee's L: {n_cells:,} cells, {n_pairs} pair(s), "
                f"k={n_neighbors}, permutations={n_permutations}")

    ctx = _lib.default_context(device)
    if radius is not None:   # extension: radius graph instead of kNN (see _radius_weights)
        _radius_weights(ctx, coords, radius, float32_weights=True)
    else:
        _knn_weights_f32(ctx, coords, n_neighbors)
    flat = [g for pair in gene_pairs for g in pair]
    cols, where = _unique_columns(adata, flat)
    ctx.set_expression(_expression(adata, layer), cols)
    _, var = ctx.expr_stats()
    pair_slots = where.reshape(-1, 2)
    degenerate = ~((var[pair_slots[:, 0]] > 0) & (var[pair_slots[:, 1]] > 0))
    # one device call for the whole pair loop: observed L of every pair on the fp64 matrix cores, then a fresh block of
    # P numpy-exact permutations per live pair, in pair order, from ONE stream (AC:1109-1148)
    words = _lib.rng_state_words(np.random.default_rng(seed))
    if shared_permutations:
        xs, xi = np.unique(pair_slots[:, 0], return_inverse=True)
        ys, yi = np.unique(pair_slots[:, 1], return_inverse=True)
        if rng == "philox" and n_permutations > 0:
            ctx.generate_permutations_counter(seed, n_cells, n_permutations)
            grid = ctx.lee_shared(None, xs, ys, n_permutations)
        else:
            grid = ctx.lee_shared(words, xs, ys, n_permutations)
        L, cnt = grid["L"][xi, yi], grid["count_abs_ge"][xi, yi]
    else:
        out = ctx.lee_seeded(words, pair_slots[:, 0], pair_slots[:, 1], n_permutations)
        L, cnt = out["L"], out["count_abs_ge"]
    X_in = _expression(adata, layer)
    if getattr(X_in, "dtype", None) == np.float32:
        # the reference computes a float32 matrix in float32 (AC:1118-1146); hand back ITS number: same roundings,
        # same (numpy pairwise) summation order -- an fp64 L differs from it by the float32 noise, ~1e-5 relative
        L = ctx.lee_observed_f32(pair_slots[:, 0], pair_slots[:, 1]).astype(np.float64)

    results = []
    for q, (gene_x, gene_y) in enumerate(gene_pairs):
        if degenerate[q]:
            logger.warning(f"Gene pair ({gene_x}, {gene_y}) has zero variance gene - setting L to 0")
            results.append({"gene_x": gene_x, "gene_y": gene_y, "L": 0.0, "p_value": 1.0})
            continue
        p_value = float((cnt[q] + 1) / (n_permutations + 1)) if n_permutations > 0 else 1.0
        results.append({"gene_x": gene_x, "gene_y": gene_y, "L": float(L[q]), "p_value": p_value})

    elapsed = time.time() - start_time
    logger.info(f"Global Lee's L completed in {elapsed:.1f}s")
    return results[0] if single_pair else results


# =============================================================================================
# Local Moran's I (AC:656-983)
# =============================================================================================


def _padj_tables(hist: np.ndarray, n_cells: int, n_permutations: int, method: str) -> np.ndarray:
    """Adjusted p-value of every permutation-count level, per gene: ``tab[g, c]`` is what the reference's FDR step
    (AC:132-183 applied per gene at AC:912-920) gives a cell of gene g whose count is c.  ``hist[g, c]`` = number of
    cells at that level.  Benjamini-Hochberg without sorting N values: p takes at most P + 1 levels, and after the
    reference's cumulative minimum every cell of a level gets ``min over higher-or-equal levels of p * n / (last rank of
    that level)`` -- evaluated with the reference's own dtypes (float32 levels, float32 product, float64 quotient,
    float32 store); ``tests/test_cpu_properties.py`` proves it equal, bit for bit, to the sort-based form."""
    levels = ((np.arange(n_permutations + 1) + 1) / (n_permutations + 1)).astype(np.float32)
    if method == "none":
        return np.tile(levels, (hist.shape[0], 1))
    if method == "bonferroni":
        return np.tile(np.clip(levels * n_cells, 0, 1).astype(np.float32), (hist.shape[0], 1))
    last_rank = np.cumsum(hist, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = (levels * np.float32(n_cells)).astype(np.float64)[None, :] / last_rank
    a[hist == 0] = np.inf
    a = np.minimum.accumulate(a[:, ::-1], axis=1)[:, ::-1]
    return np.clip(a, 0, 1).astype(np.float32)


def local_morans_i(
    adata,
    genes: Optional[Union[str, List[str]]] = None,
    layer: Optional[str] = None,
    spatial_key: str = "spatial",
    n_neighbors: int = 6,
    n_permutations: int = 10,
    fdr_correction: str = "fdr_bh",
    alpha: float = 0.05,
    seed: int = 0,
    batch_size: int = 100,
    key_added: str = "local_morans",
    copy: bool = False,
    *,
    device: int = 0,
):
    """Local Moran's I (LISA) with per-cell permutation p-values, FDR and quadrants (AC:656-983).

    Outputs as the reference: ``obsm[{key}_I|_z|_lag|_p|_p_adj]`` float32 ``(n_cells, n_genes)``,
    ``obsm[{key}_quadrant]`` int8, ``uns[{key}_params]``.  Arithmetic follows the reference's float32
    path; genes are processed in batches of ``batch_size`` and every batch draws its own
    ``n_permutations`` permutations from ONE numpy-exact stream (AC:839-879).  The per-cell counts
    are accumulated on the GPU instead of materialising the reference's ``(P, N, B)`` tensor.
    """
    start_time = time.time()
    coords = _require_spatial(adata, spatial_key)
    _check_counts(n_neighbors, n_permutations)
    if fdr_correction not in ["bonferroni", "fdr_bh", "none"]:
        raise ValueError(f"Invalid fdr_correction: '{fdr_correction}'. Must be 'bonferroni', 'fdr_bh', or 'none'.")
    adata = adata.copy() if copy else adata
    gene_names = _resolve_genes(adata, genes, "This may be slow and memory-intensive.")
    n_cells, n_genes = adata.n_obs, len(gene_names)
    gene_indices = np.array([adata.var_names.get_loc(g) for g in gene_names])
    logger.info(f"Computing Local Moran's I: {n_cells:,} cells, {n_genes} genes, "
                f"k={n_neighbors}, permutations={n_permutations}")

    ctx = _lib.default_context(device)
    _knn_weights_f32(ctx, coords, n_neighbors)
    X = _expression(adata, layer)
    if sparse.issparse(X) and n_genes > batch_size:
        X = X.tocsc()        # one conversion; every batch then ships only its own columns (AC:813-816 does the same)

    words = _lib.rng_state_words(np.random.default_rng(seed))
    n_batches = (n_genes + batch_size - 1) // batch_size
    logger.info(f"Processing {n_genes} genes in {n_batches} batches")
    levels = ((np.arange(n_permutations + 1) + 1) / (n_permutations + 1)).astype(np.float32)
    single = n_batches == 1

    def alloc(dtype, fill=None):
        if single:
            return None                      # the batch's own arrays become the outputs
        a = np.empty((n_cells, n_genes), dtype=dtype)
        if fill is not None:
            a.fill(fill)
        return a

    local_I, z_values, lag_values = alloc(np.float32), alloc(np.float32), alloc(np.float32)
    p_values = alloc(np.float32) if n_permutations > 0 else None
    p_adj = alloc(np.float32) if n_permutations > 0 else None
    quadrants = alloc(np.int8)
    zero_var_mask = np.zeros(n_genes, dtype=bool)

    def put(dst, src, b0, b1, inv):
        if inv is not None:
            src = src[:, inv]
        if single:
            return np.ascontiguousarray(src)
        dst[:, b0:b1] = src
        return dst

    for batch_idx in range(n_batches):
        b0, b1 = batch_idx * batch_size, min((batch_idx + 1) * batch_size, n_genes)
        logger.debug(f"Processing batch {batch_idx + 1}/{n_batches}")
        cols, inv = np.unique(gene_indices[b0:b1], return_inverse=True)
        if inv.size == cols.size and np.array_equal(inv, np.arange(cols.size)):
            inv = None                       # the usual case: distinct genes in ascending column order
        if sparse.issparse(X) and n_batches > 1:
            ctx.set_expression(X[:, cols], np.arange(cols.size, dtype=np.int32))
        else:
            ctx.set_expression(X, cols.astype(np.int32))
        if n_permutations > 0:   # continues the one stream; generator and per-cell counts run as one pipeline
            r = ctx.local_moran_seeded(words, n_cells, n_permutations, fetch_counts=False)
        else:
            r = ctx.local_moran(n_cells, 0, fetch_counts=False)
        zero = r["zero_var"]
        # per-cell p, adjusted p and quadrants on the device: lookup tables per (gene, permutation count) built here
        # with the reference's expressions (AC:894-896, 912-920); zero-variance genes get p = p_adj = 1, quadrant 0
        if n_permutations > 0:
            hist = ctx.local_moran_hist(n_permutations)
            hist[zero] = 0
            hist[zero, n_permutations] = n_cells
            p_tab = np.tile(levels, (cols.size, 1))
            p_tab[zero] = 1.0
            padj_tab = _padj_tables(hist, n_cells, n_permutations, fdr_correction)
            padj_tab[zero] = 1.0
            pb, ab, qb = ctx.local_moran_classify(n_cells, p_tab, padj_tab, zero, alpha)
            p_values = put(p_values, pb, b0, b1, inv)
            p_adj = put(p_adj, ab, b0, b1, inv)
        else:
            _, _, qb = ctx.local_moran_classify(n_cells, None, None, zero, alpha)
        quadrants = put(quadrants, qb, b0, b1, inv)
        for name in ("z", "lag", "I"):
            if zero.any():
                r[name][:, zero] = 0.0
        z_values = put(z_values, r["z"], b0, b1, inv)
        lag_values = put(lag_values, r["lag"], b0, b1, inv)
        local_I = put(local_I, r["I"], b0, b1, inv)
        zero_var_mask[b0:b1] = zero if inv is None else zero[inv]

    zero_variance_genes = [gene_names[i] for i in np.where(zero_var_mask)[0]]
    if zero_var_mask.any():
        logger.warning(f"{int(zero_var_mask.sum())} genes have zero variance and will be skipped: "
                       f"{zero_variance_genes[:5]}")
    if n_permutations > 0:
        logger.debug(f"Applied {fdr_correction} correction; LISA quadrants with significance filtering")
    else:
        logger.warning("n_permutations=0: Quadrants classified by z/lag signs only, "
                       "without significance filtering. Consider n_permutations>=99 for p-values.")
        p_values = np.ones((n_cells, n_genes), dtype=np.float32)
        p_adj = p_values

    adata.obsm[f"{key_added}_I"] = local_I
    adata.obsm[f"{key_added}_z"] = z_values
    adata.obsm[f"{key_added}_lag"] = lag_values
    adata.obsm[f"{key_added}_p"] = p_values
    adata.obsm[f"{key_added}_p_adj"] = p_adj
    adata.obsm[f"{key_added}_quadrant"] = quadrants

    elapsed = time.time() - start_time
    adata.uns[f"{key_added}_params"] = {
        "genes": gene_names,
        "n_neighbors": n_neighbors,
        "n_permutations": n_permutations,
        "fdr_correction": fdr_correction,
        "alpha": alpha,
        "n_cells": n_cells,
        "n_genes": n_genes,
        "seed": seed,
        "computation_time_seconds": elapsed,
        "zero_variance_genes": zero_variance_genes,
    }
    n_significant = (quadrants != 0).sum(axis=0)
    logger.info(f"Local Moran's I completed in {elapsed:.1f}s. "
                f"Significant cells per gene: min={n_significant.min()}, max={n_significant.max()}")
    update_metadata(
        adata,
        function_name="local_morans_i",
        parameters={
            "genes": gene_names[:10] if len(gene_names) > 10 else gene_names,
            "n_genes": n_genes,
            "n_neighbors": n_neighbors,
            "n_permutations": n_permutations,
            "fdr_correction": fdr_correction,
            "alpha": alpha,
            "seed": seed,
            "permgen_form": ctx.permgen_form(n_cells) if n_permutations > 0 else None,
        },
        outputs={
            "obsm_I": f"{key_added}_I",
            "obsm_z": f"{key_added}_z",
            "obsm_lag": f"{key_added}_lag",
            "obsm_p": f"{key_added}_p",
            "obsm_p_adj": f"{key_added}_p_adj",
            "obsm_quadrant": f"{key_added}_quadrant",
            "uns_params": f"{key_added}_params",
        },
    )
    return adata


# =============================================================================================
# Local Lee's L (AC:1171-1479)
# =============================================================================================


def lees_l_local(
    adata,
    gene_pairs: Optional[Union[Tuple[str, str], List[Tuple[str, str]]]] = None,
    genes: Optional[List[str]] = None,
    layer: Optional[str] = None,
    spatial_key: str = "spatial",
    n_neighbors: int = 6,
    n_permutations: int = 199,
    compute_cell_pvalues: bool = False,
    significance_filter: bool = False,
    alpha: float = 0.05,
    seed: int = 0,
    copy: bool = False,
    *,
    device: int = 0,
):
    """Local Lee's L per gene pair (AC:1171-1479): ``obs[{x}_{y}_lees_l]`` float32,
    ``obs[{x}_{y}_quadrant]`` Categorical [NS, HH, LL, HL, LH], ``obs[{x}_{y}_pvalue]`` float32,
    ``uns[{x}_{y}_lees_l_params]``.  One numpy-exact stream serves all pairs: P permutations for the
    global p-value, then (``compute_cell_pvalues``) P more for the per-cell p-values."""
    from itertools import combinations

    start_time = time.time()
    if gene_pairs is None and genes is None:
        raise ValueError("Must provide either 'gene_pairs' or 'genes' parameter. "
                         "Example: gene_pairs=('CD8A', 'GZMB') or genes=['CD8A', 'GZMB', 'FOXP3']")
    coords = _require_spatial(adata, spatial_key)
    _check_counts(n_neighbors, n_permutations)
    if significance_filter and not compute_cell_pvalues:
        raise ValueError("significance_filter=True requires compute_cell_pvalues=True")
    if genes is not None:
        n_all = len(genes) * (len(genes) - 1) // 2
        logger.warning(f"All-pairs mode: {len(genes)} genes = {n_all} pairs. "
                       "This may take a very long time for large gene sets. "
                       "Consider using explicit gene_pairs for better performance.")
        gene_pairs = list(combinations(genes, 2))
    else:
        gene_pairs, _ = _normalize_pairs(gene_pairs)
    all_genes = list(dict.fromkeys(g for pair in gene_pairs for g in pair))
    missing = set(all_genes) - set(adata.var_names)
    if missing:
        raise ValueError(f"Genes not found in adata.var_names: {list(missing)}")
    adata = adata.copy() if copy else adata
    n_cells, n_pairs = adata.n_obs, len(gene_pairs)
    logger.info(f"Computing Local Lee's L: {n_cells:,} cells, {n_pairs} pair(s), "
                f"k={n_neighbors}, permutations={n_permutations}")

    ctx = _lib.default_context(device)
    _knn_weights_f32(ctx, coords, n_neighbors)
    cols, _ = _unique_columns(adata, all_genes)
    slot = {g: i for i, g in enumerate(all_genes)}
    ctx.set_expression(_expression(adata, layer), cols)
    _, var = ctx.expr_stats()
    zero_var_genes = {g for g in all_genes if not var[slot[g]] > 0}
    if zero_var_genes:
        logger.warning(f"Genes with zero variance: {zero_var_genes}")

    words = _lib.rng_state_words(np.random.default_rng(seed))
    categories = ["NS", "HH", "LL", "HL", "LH"]
    for pair_idx, (gene_x, gene_y) in enumerate(gene_pairs):
        logger.debug(f"Processing pair {pair_idx + 1}/{n_pairs}: {gene_x} vs {gene_y}")
        key = f"{gene_x}_{gene_y}"
        if gene_x in zero_var_genes or gene_y in zero_var_genes:
            adata.obs[f"{key}_lees_l"] = np.zeros(n_cells, dtype=np.float32)
            adata.obs[f"{key}_quadrant"] = pd.Categorical(["NS"] * n_cells, categories=categories)
            adata.uns[f"{key}_lees_l_params"] = {"gene_x": gene_x, "gene_y": gene_y, "global_L": 0.0,
                                                 "global_pvalue": 1.0, "n_neighbors": n_neighbors,
                                                 "zero_variance": True}
            continue
        sx, sy = slot[gene_x], slot[gene_y]
        cell_p = compute_cell_pvalues and n_permutations > 0
        global_pvalue = 1.0
        if n_permutations > 0:
            # global block first, then the per-cell block, from the same stream (AC:1394-1408): one pipeline on the device,
            # the permuted sums and the per-cell counts taken chunk by chunk behind the generator
            loc = ctx.lee_local_seeded(words, n_cells, sx, sy, n_permutations, n_permutations if cell_p else 0)
            global_pvalue = float((loc["count_abs_ge"] + 1) / (n_permutations + 1))
            L_global = loc["L"]
        else:
            g = ctx.lee([sx], [sy], None, 0)
            if compute_cell_pvalues:
                logger.warning("compute_cell_pvalues=True but n_permutations=0; p-values will be 1.0")
            L_global = float(g["L"][0])
            loc = ctx.lee_local(n_cells, sx, sy, 0, 0)
        p_values = np.ones(n_cells, dtype=np.float32)
        if cell_p:
            p_values = ((loc["count"] + 1) / (n_permutations + 1)).astype(np.float32)
        quadrants = _classify_quadrants(loc["zx"], loc["lag"], p_values if significance_filter else None, alpha)
        adata.obs[f"{key}_lees_l"] = loc["L_local"].astype(np.float32)
        # (the same Categorical as pd.Categorical(labels, categories=...) of AC:1429, built from the codes: 0.7 instead of
        # 260 ms per pair at 10^6 cells)
        adata.obs[f"{key}_quadrant"] = pd.Categorical.from_codes(quadrants, categories=categories)
        adata.obs[f"{key}_pvalue"] = p_values.astype(np.float32)
        cnt = np.bincount(quadrants, minlength=5)
        adata.uns[f"{key}_lees_l_params"] = {
            "gene_x": gene_x,
            "gene_y": gene_y,
            "global_L": L_global,
            "global_pvalue": global_pvalue,
            "n_neighbors": n_neighbors,
            "n_permutations": n_permutations,
            "compute_cell_pvalues": compute_cell_pvalues,
            "significance_filter": significance_filter,
            "alpha": alpha,
            "quadrant_counts": {lab: int(cnt[i]) for i, lab in enumerate(categories)},
        }

    elapsed = time.time() - start_time
    logger.info(f"Local Lee's L completed in {elapsed:.1f}s for {n_pairs} pair(s)")
    pair_keys = [f"{gx}_{gy}" for gx, gy in gene_pairs]
    update_metadata(
        adata,
        function_name="lees_l_local",
        parameters={
            "gene_pairs": [(gx, gy) for gx, gy in gene_pairs[:10]],
            "n_pairs": n_pairs,
            "n_neighbors": n_neighbors,
            "n_permutations": n_permutations,
            "compute_cell_pvalues": compute_cell_pvalues,
            "significance_filter": significance_filter,
            "alpha": alpha,
            "seed": seed,
            "permgen_form": ctx.permgen_form(adata.n_obs) if n_permutations > 0 else None,
        },
        outputs={"obs_keys": [f"{k}_lees_l" for k in pair_keys[:5]],
                 "uns_keys": [f"{k}_lees_l_params" for k in pair_keys[:5]]},
    )
    return adata
