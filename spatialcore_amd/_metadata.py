"""Provenance side effect of the hot path: every AnnData-returning function appends one entry to
``adata.uns["spatialcore_metadata"]["operations"]`` with the same keys the reference writes
(reference src/spatialcore/core/metadata.py:49-77, 94-117, 132-149), so tools that read that log
keep working after the switch."""

from datetime import datetime
from pathlib import Path
from typing import Any, Dict, Optional


def _plain(params: Dict[str, Any]) -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for key, value in params.items():
        if value is None or isinstance(value, (str, int, float, bool)):
            out[key] = value
        elif isinstance(value, (list, tuple)):
            out[key] = list(value)
        elif isinstance(value, dict):
            out[key] = _plain(value)
        elif isinstance(value, Path):
            out[key] = str(value)
        else:
            out[key] = type(value).__name__
    return out


def update_metadata(adata, function_name: str, parameters: Dict[str, Any],
                    outputs: Optional[Dict[str, Any]] = None) -> None:
    meta = adata.uns.get("spatialcore_metadata")
    if meta is None:
        meta = {"created": datetime.now().isoformat(), "operations": []}
        adata.uns["spatialcore_metadata"] = meta
    ops = meta.get("operations")
    if ops is None:
        ops = []
    elif not isinstance(ops, list):
        ops = list(ops)  # h5ad round trips turn the list into an array
    meta["operations"] = ops
    entry = {"timestamp": datetime.now().isoformat(), "function": function_name,
             "parameters": _plain(parameters)}
    if outputs:
        entry["outputs"] = outputs
    ops.append(entry)
