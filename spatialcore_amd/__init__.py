"""spatialcore_amd -- MI355X-native drop-in for the spatialcore.spatial hot path."""
from spatialcore_amd._adata import SimpleAnnData

__all__ = ["SimpleAnnData"]
__version__ = "0.1.0"
