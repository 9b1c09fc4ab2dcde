"""Domain-to-domain distances on MI355X.

Same public contract as the reference's ``calculate_domain_distances`` / ``get_distance_matrix``
(reference src/spatialcore/spatial/distance.py:46-449 and 452-495, ``DS`` below): keywords, defaults,
``adata.obs[output_distance_column | output_nearest_column]``, ``adata.uns['domain_distances']``,
error messages and the provenance entry.

How it is computed here (nothing of the reference's per-pair Python loops survives):

* Domain labels become integer codes once (``_DomainCodes``); every later step is array arithmetic on
  codes.
* The S x T result table is a NumPy array that three device calls fill:
  ``sc_nearest_2d`` (grid ring walk, one launch for all source cells) for per-cell nearest targets,
  ``sc_pair_table_2d`` (LDS-tiled all-pairs, every (source, target) block in one launch) for the
  ``mean`` metric, and ``sc_nearest_excluding_2d`` for "nearest centroid that is not my own domain".
  Per-pair minima are segmented reductions over code pairs (``np.minimum.at``).
* Semantics kept from the reference, including the odd one: with ``minimum`` + per-cell output an
  entry (s, t) is the smallest *nearest-target distance among the cells of s whose nearest target
  lies in t* (DS:246-257), and only pairs no cell points at fall back to the true minimum (DS:258-265).
"""

from __future__ import annotations

from typing import List, Optional

import numpy as np
import pandas as pd

from spatialcore_amd import _lib
from spatialcore_amd._logging import get_logger
from spatialcore_amd._metadata import update_metadata

logger = get_logger("spatial.distance")

_METRICS = ("minimum", "centroid", "mean")
_MODES = ("cell", "matrix", "both")


class _DomainCodes:
    """One obs column reduced to what the geometry needs: the ordered domain names (first appearance,
    nulls dropped, optional subset filter -- DS:182-189), a code per cell (-1 = not in any selected
    domain), and the cells of all selected domains grouped by code."""

    def __init__(self, column: pd.Series, subset: Optional[List[str]]):
        names = column.dropna().unique().tolist()
        if subset:
            names = [d for d in names if d in subset]
        self.names = names
        lookup = pd.Index(names)
        codes = lookup.get_indexer(column.values) if len(names) else np.full(len(column), -1)
        codes = np.asarray(codes, dtype=np.int64)
        codes[pd.isna(column.values)] = -1
        self.codes = codes
        self.cells = np.flatnonzero(codes >= 0)                       # ascending cell order
        order = np.argsort(codes[self.cells], kind="stable")
        self.grouped = self.cells[order]                              # the same cells, sorted by code
        self.offsets = np.concatenate([[0], np.cumsum(np.bincount(codes[self.cells], minlength=len(names)))])

    def __len__(self) -> int:
        return len(self.names)


def _coordinates(adata) -> np.ndarray:
    xy = np.asarray(adata.obsm["spatial"])
    if xy.ndim != 2 or xy.shape[1] != 2:
        raise ValueError("only 2-D coordinates are supported by the MI355X path "
                         f"(adata.obsm['spatial'] has shape {xy.shape})")
    return np.ascontiguousarray(xy, dtype=np.float64)


def _annotate_nearest(ctx, adata, xy, src: _DomainCodes, tgt: _DomainCodes, dist_col: str, near_col: str):
    """Per source cell: distance to, and domain of, the nearest target cell (DS:219-238, 356-373).
    Returns (distance, target code) aligned with ``src.cells``."""
    if src.cells.size == 0 or tgt.cells.size == 0:
        return None
    dist, hit = ctx.nearest(xy[tgt.cells], xy[src.cells])
    hit_code = tgt.codes[tgt.cells[hit]]
    adata.obs.iloc[src.cells, adata.obs.columns.get_loc(dist_col)] = dist
    adata.obs.iloc[src.cells, adata.obs.columns.get_loc(near_col)] = np.asarray(tgt.names, dtype=object)[hit_code]
    return dist, hit_code


def _true_minima(ctx, xy, src: _DomainCodes, tgt: _DomainCodes, table: np.ndarray, wanted: np.ndarray) -> None:
    """table[s, t] = min over cells of s, cells of t of their distance, for the pairs flagged in
    ``wanted`` (the reference's cdist(...).min(), DS:260-265, 387-398): one nearest-target launch per
    target domain that has a flagged pair, reduced per source code."""
    for t in np.flatnonzero(wanted.any(axis=0)):
        rows = np.flatnonzero(wanted[:, t])
        cells = src.cells[np.isin(src.codes[src.cells], rows)]
        members = tgt.grouped[tgt.offsets[t]:tgt.offsets[t + 1]]
        if cells.size == 0 or members.size == 0:
            continue
        dist, _ = ctx.nearest(xy[members], xy[cells])
        best = np.full(len(src), np.inf)
        np.minimum.at(best, src.codes[cells], dist)
        table[rows, t] = best[rows]


def _centroids(xy: np.ndarray, dom: _DomainCodes) -> np.ndarray:
    """Mean coordinate of every selected domain, accumulated in cell order as ``coords.mean(axis=0)`` does."""
    c = dom.codes[dom.cells]
    n = np.bincount(c, minlength=len(dom)).astype(np.float64)
    return np.stack([np.bincount(c, weights=xy[dom.cells, a], minlength=len(dom)) / n for a in (0, 1)], axis=1)


def calculate_domain_distances(
    adata,
    source_domain_column: str,
    target_domain_column: str,
    source_domain_subset: Optional[List[str]] = None,
    target_domain_subset: Optional[List[str]] = None,
    distance_metric: str = "minimum",
    output_mode: str = "both",
    output_distance_column: str = "distance_to_target",
    output_nearest_column: str = "nearest_target_domain",
    copy: bool = False,
    *,
    device: int = 0,
):
    """Spatial distances from source domains to target domains (DS:46-449)."""
    if "spatial" not in adata.obsm:
        raise ValueError(f"adata.obsm['spatial'] not found. Available keys: {list(adata.obsm.keys())}")
    for role, col in (("Source", source_domain_column), ("Target", target_domain_column)):
        if col not in adata.obs.columns:
            raise ValueError(f"{role} column '{col}' not found in adata.obs. "
                             f"Available columns: {list(adata.obs.columns)}")
    if distance_metric not in _METRICS:
        raise ValueError(f"Invalid distance_metric: '{distance_metric}'. Must be 'minimum', 'centroid', or 'mean'.")
    if output_mode not in _MODES:
        raise ValueError(f"Invalid output_mode: '{output_mode}'. Must be 'cell', 'matrix', or 'both'.")
    if copy:
        adata = adata.copy()
    logger.info(f"Calculating domain distances: {source_domain_column} → {target_domain_column} "
                f"(metric={distance_metric}, mode={output_mode})")

    src = _DomainCodes(adata.obs[source_domain_column], source_domain_subset)
    tgt = _DomainCodes(adata.obs[target_domain_column], target_domain_subset)
    if not len(src):
        raise ValueError(f"No valid source domains found in '{source_domain_column}'")
    if not len(tgt):
        raise ValueError(f"No valid target domains found in '{target_domain_column}'")
    logger.debug(f"Source domains ({len(src)}): {src.names[:5]}...")
    logger.debug(f"Target domains ({len(tgt)}): {tgt.names[:5]}...")

    xy = _coordinates(adata)
    ctx = _lib.default_context(device)
    per_cell = output_mode in ("cell", "both")
    if per_cell:
        adata.obs[output_distance_column] = np.nan
        adata.obs[output_nearest_column] = None

    S, T = len(src), len(tgt)
    table = np.full((S, T), np.nan)
    # a domain paired with itself (same column, same label) is 0 by definition (DS:243-245 and siblings)
    own = np.zeros((S, T), dtype=bool)
    if source_domain_column == target_domain_column:
        own = np.asarray(src.names, dtype=object)[:, None] == np.asarray(tgt.names, dtype=object)[None, :]
    # the target code a source domain must not be matched with (-1: none)
    own_target = np.where(own.any(axis=1), own.argmax(axis=1), -1)

    if distance_metric == "minimum":
        missing = ~own
        if per_cell:
            found = _annotate_nearest(ctx, adata, xy, src, tgt, output_distance_column, output_nearest_column)
            if found is not None:
                dist, hit_code = found
                pointed = np.full((S, T), np.inf)
                np.minimum.at(pointed, (src.codes[src.cells], hit_code), dist)
                got = np.isfinite(pointed) & ~own
                table[got] = pointed[got]
                missing &= ~got
        _true_minima(ctx, xy, src, tgt, table, missing)
    elif distance_metric == "mean":
        total, _ = ctx.pair_table(xy[src.grouped], src.offsets, xy[tgt.grouped], tgt.offsets)
        pairs = np.outer(np.diff(src.offsets), np.diff(tgt.offsets)).astype(np.float64)
        np.divide(total, pairs, out=table, where=pairs > 0)
        if per_cell:
            logger.debug("Using minimum distance for per-cell annotation with mean metric")
            _annotate_nearest(ctx, adata, xy, src, tgt, output_distance_column, output_nearest_column)
    else:  # centroid
        cs, ct = _centroids(xy, src), _centroids(xy, tgt)
        table[:] = np.linalg.norm(cs[:, None, :] - ct[None, :, :], axis=-1)
        if per_cell and src.cells.size:
            # nearest target centroid per source cell, its own domain's centroid skipped (DS:305-326)
            dist, hit = ctx.nearest_excluding(ct, np.arange(T), xy[src.cells], own_target[src.codes[src.cells]])
            names = np.asarray(tgt.names + [None], dtype=object)
            adata.obs.iloc[src.cells, adata.obs.columns.get_loc(output_distance_column)] = dist
            adata.obs.iloc[src.cells, adata.obs.columns.get_loc(output_nearest_column)] = names[hit]
    table[own] = 0.0

    known = table[~np.isnan(table)]
    stat = (lambda f: float(f(known))) if known.size else (lambda f: None)
    summary = {"min_distance": stat(np.min), "max_distance": stat(np.max),
               "mean_distance": stat(np.mean), "median_distance": stat(np.median)}
    logger.info(f"Distance statistics: min={summary['min_distance']:.1f}, "
                f"max={summary['max_distance']:.1f}, mean={summary['mean_distance']:.1f}")

    outputs = {"summary_statistics": summary}
    if per_cell:
        outputs["obs_distance"] = output_distance_column
        outputs["obs_nearest"] = output_nearest_column
    if output_mode in ("matrix", "both"):
        frame = pd.DataFrame(table, index=src.names, columns=tgt.names)
        adata.uns["domain_distances"] = {
            "source_domain_column": source_domain_column,
            "target_domain_column": target_domain_column,
            "distance_metric": distance_metric,
            "source_domains": src.names,
            "target_domains": tgt.names,
            "summary_statistics": summary,
            "distance_matrix": frame.to_dict(orient="index"),
        }
        outputs["uns"] = "domain_distances"
    update_metadata(
        adata,
        function_name="calculate_domain_distances",
        parameters={
            "source_domain_column": source_domain_column,
            "target_domain_column": target_domain_column,
            "source_domain_subset": source_domain_subset,
            "target_domain_subset": target_domain_subset,
            "distance_metric": distance_metric,
            "output_mode": output_mode,
        },
        outputs=outputs,
    )
    return adata


def get_distance_matrix(adata, key: str = "domain_distances") -> pd.DataFrame:
    """Distance matrix (source rows x target columns) from ``adata.uns[key]`` (DS:452-495)."""
    if key not in adata.uns:
        raise KeyError(f"'{key}' not found in adata.uns. "
                       "Run calculate_domain_distances() with output_mode='matrix' or 'both' first.")
    stored = adata.uns[key]
    if "distance_matrix" not in stored:
        raise KeyError(f"'distance_matrix' not found in adata.uns['{key}']")
    return pd.DataFrame.from_dict(stored["distance_matrix"], orient="index")
