"""MI355X-native spatial statistics: the hot path of ``spatialcore.spatial``
(reference src/spatialcore/spatial/__init__.py:11-52) behind the same function names."""

from spatialcore_amd.spatial.autocorrelation import (
    build_spatial_weights,
    lees_l,
    lees_l_local,
    local_morans_i,
    morans_i,
)
from spatialcore_amd.spatial.distance import calculate_domain_distances, get_distance_matrix
from spatialcore_amd.spatial.neighborhoods import compute_neighborhood_profile, neighborhood_enrichment

__all__ = [
    "morans_i",
    "local_morans_i",
    "lees_l",
    "lees_l_local",
    "build_spatial_weights",
    "compute_neighborhood_profile",
    "neighborhood_enrichment",  # extension: not in the reference
    "calculate_domain_distances",
    "get_distance_matrix",
]
