"""Domain-to-domain distances on MI355X.

Drop-in mirror of the reference's ``calculate_domain_distances`` / ``get_distance_matrix``
(reference src/spatialcore/spatial/distance.py:46-449, 452-495, ``DS`` below): same keywords,
defaults, outputs (``adata.obs[output_distance_column|output_nearest_column]``,
``adata.uns['domain_distances']``), errors and provenance entry.  The geometry runs on the GPU:
nearest-target search on the bin grid (``sc_nearest_2d``, replaces ``cKDTree.query(k=1)``) and an
LDS-tiled brute-force kernel over all pairs (``sc_pairwise_2d``, replaces ``cdist(...).mean()/min()``).
The domain bookkeeping (label lists, result tables) is the reference's pandas logic.
"""

from __future__ import annotations

from typing import List, Optional

import numpy as np
import pandas as pd

from spatialcore_amd import _lib
from spatialcore_amd._logging import get_logger
from spatialcore_amd._metadata import update_metadata

logger = get_logger("spatial.distance")


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
    if source_domain_column not in adata.obs.columns:
        raise ValueError(f"Source column '{source_domain_column}' not found in adata.obs. "
                         f"Available columns: {list(adata.obs.columns)}")
    if target_domain_column not in adata.obs.columns:
        raise ValueError(f"Target column '{target_domain_column}' not found in adata.obs. "
                         f"Available columns: {list(adata.obs.columns)}")
    if distance_metric not in ["minimum", "centroid", "mean"]:
        raise ValueError(f"Invalid distance_metric: '{distance_metric}'. Must be 'minimum', 'centroid', or 'mean'.")
    if output_mode not in ["cell", "matrix", "both"]:
        raise ValueError(f"Invalid output_mode: '{output_mode}'. Must be 'cell', 'matrix', or 'both'.")

    adata = adata.copy() if copy else adata
    logger.info(f"Calculating domain distances: {source_domain_column} → {target_domain_column} "
                f"(metric={distance_metric}, mode={output_mode})")

    source_domains = adata.obs[source_domain_column].dropna().unique().tolist()
    target_domains = adata.obs[target_domain_column].dropna().unique().tolist()
    if source_domain_subset:
        source_domains = [d for d in source_domains if d in source_domain_subset]
    if target_domain_subset:
        target_domains = [d for d in target_domains if d in target_domain_subset]
    if not source_domains:
        raise ValueError(f"No valid source domains found in '{source_domain_column}'")
    if not target_domains:
        raise ValueError(f"No valid target domains found in '{target_domain_column}'")

    distance_matrix = pd.DataFrame(index=source_domains, columns=target_domains, dtype=float)
    spatial = np.ascontiguousarray(np.asarray(adata.obsm["spatial"])[:, :2], dtype=np.float64)
    ctx = _lib.default_context(device)
    same_column = source_domain_column == target_domain_column
    src_labels = adata.obs[source_domain_column].values
    tgt_labels = adata.obs[target_domain_column].values

    if output_mode in ["cell", "both"]:
        adata.obs[output_distance_column] = np.nan
        adata.obs[output_nearest_column] = None

    def per_cell_minimum():
        """Nearest target cell of every source cell (DS:219-238, 356-373)."""
        target_indices = np.where(adata.obs[target_domain_column].isin(target_domains).values)[0]
        source_indices = np.where(adata.obs[source_domain_column].isin(source_domains).values)[0]
        if len(source_indices) == 0 or len(target_indices) == 0:
            return None
        distances, nearest_idx = ctx.nearest(spatial[target_indices], spatial[source_indices])
        target_domains_arr = tgt_labels[target_indices]
        nearest_domains = target_domains_arr[nearest_idx]
        adata.obs.iloc[source_indices, adata.obs.columns.get_loc(output_distance_column)] = distances
        adata.obs.iloc[source_indices, adata.obs.columns.get_loc(output_nearest_column)] = nearest_domains
        return source_indices, target_indices, distances, nearest_domains, target_domains_arr

    if distance_metric == "minimum" and output_mode in ["cell", "both"]:
        res = per_cell_minimum()
        if res is not None:
            source_indices, target_indices, distances, nearest_domains, target_domains_arr = res
            source_domains_arr = src_labels[source_indices]
            source_coords, target_coords = spatial[source_indices], spatial[target_indices]
            for src in source_domains:
                src_mask = source_domains_arr == src
                if not src_mask.any():
                    continue
                for tgt in target_domains:
                    if src == tgt and same_column:
                        distance_matrix.loc[src, tgt] = 0.0
                        continue
                    hit = nearest_domains[src_mask] == tgt
                    if hit.any():
                        distance_matrix.loc[src, tgt] = distances[src_mask][hit].min()
                    else:
                        tgt_cell_mask = target_domains_arr == tgt
                        if tgt_cell_mask.any():
                            distance_matrix.loc[src, tgt] = ctx.pairwise(source_coords[src_mask],
                                                                         target_coords[tgt_cell_mask])[1]

    elif distance_metric == "centroid":
        source_centroids = {}
        target_centroids = {}
        for src in source_domains:
            coords = spatial[(adata.obs[source_domain_column] == src).values]
            if len(coords) > 0:
                source_centroids[src] = coords.mean(axis=0)
        for tgt in target_domains:
            coords = spatial[(adata.obs[target_domain_column] == tgt).values]
            if len(coords) > 0:
                target_centroids[tgt] = coords.mean(axis=0)
        for src in source_domains:
            if src not in source_centroids:
                continue
            for tgt in target_domains:
                if src == tgt and same_column:
                    distance_matrix.loc[src, tgt] = 0.0
                    continue
                if tgt not in target_centroids:
                    continue
                distance_matrix.loc[src, tgt] = np.linalg.norm(source_centroids[src] - target_centroids[tgt])
        if output_mode in ["cell", "both"] and target_centroids:
            # per-cell: nearest target centroid (excluding the cell's own domain when the columns coincide),
            # DS:305-326 -- the reference's per-row Python loop, here one nearest query per excluded label
            src_sel = np.where(adata.obs[source_domain_column].isin(source_domains).values)[0]
            names = list(target_centroids.keys())
            cents = np.array([target_centroids[t] for t in names], dtype=np.float64)
            dist_col = adata.obs.columns.get_loc(output_distance_column)
            near_col = adata.obs.columns.get_loc(output_nearest_column)
            groups = {None: src_sel}
            if same_column:
                lab = src_labels[src_sel]
                groups = {t: src_sel[lab == t] for t in pd.unique(lab)}
            for own, cells in groups.items():
                keep = [i for i, t in enumerate(names) if not (same_column and t == own)]
                if len(cells) == 0:
                    continue
                if not keep:
                    adata.obs.iloc[cells, dist_col] = np.inf
                    continue
                d, idx = ctx.nearest(cents[keep], spatial[cells])
                adata.obs.iloc[cells, dist_col] = d
                adata.obs.iloc[cells, near_col] = np.array(names, dtype=object)[np.array(keep)[idx]]

    elif distance_metric == "mean":
        for src in source_domains:
            src_coords = spatial[(adata.obs[source_domain_column] == src).values]
            if len(src_coords) == 0:
                continue
            for tgt in target_domains:
                if src == tgt and same_column:
                    distance_matrix.loc[src, tgt] = 0.0
                    continue
                tgt_coords = spatial[(adata.obs[target_domain_column] == tgt).values]
                if len(tgt_coords) == 0:
                    continue
                distance_matrix.loc[src, tgt] = ctx.pairwise(src_coords, tgt_coords)[0]
        if output_mode in ["cell", "both"]:
            logger.debug("Using minimum distance for per-cell annotation with mean metric")
            per_cell_minimum()

    else:  # minimum, matrix only
        for src in source_domains:
            src_coords = spatial[(adata.obs[source_domain_column] == src).values]
            if len(src_coords) == 0:
                continue
            for tgt in target_domains:
                if src == tgt and same_column:
                    distance_matrix.loc[src, tgt] = 0.0
                    continue
                tgt_coords = spatial[(adata.obs[target_domain_column] == tgt).values]
                if len(tgt_coords) == 0:
                    continue
                distance_matrix.loc[src, tgt] = ctx.nearest(tgt_coords, src_coords)[0].min()

    valid = distance_matrix.values[~np.isnan(distance_matrix.values)]
    summary = {
        "min_distance": float(valid.min()) if len(valid) > 0 else None,
        "max_distance": float(valid.max()) if len(valid) > 0 else None,
        "mean_distance": float(valid.mean()) if len(valid) > 0 else None,
        "median_distance": float(np.median(valid)) if len(valid) > 0 else None,
    }
    logger.info(f"Distance statistics: min={summary['min_distance']:.1f}, "
                f"max={summary['max_distance']:.1f}, mean={summary['mean_distance']:.1f}")

    if output_mode in ["matrix", "both"]:
        adata.uns["domain_distances"] = {
            "source_domain_column": source_domain_column,
            "target_domain_column": target_domain_column,
            "distance_metric": distance_metric,
            "source_domains": source_domains,
            "target_domains": target_domains,
            "summary_statistics": summary,
            "distance_matrix": distance_matrix.to_dict(orient="index"),
        }

    outputs = {"summary_statistics": summary}
    if output_mode in ["cell", "both"]:
        outputs["obs_distance"] = output_distance_column
        outputs["obs_nearest"] = output_nearest_column
    if output_mode in ["matrix", "both"]:
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
    data = adata.uns[key]
    if "distance_matrix" not in data:
        raise KeyError(f"'distance_matrix' not found in adata.uns['{key}']")
    return pd.DataFrame(data["distance_matrix"]).T
