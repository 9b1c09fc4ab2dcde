"""Gene sharding across the GPUs of one node.

The path shards embarrassingly: rank r owns a contiguous slice of the gene list, builds the same
graph and the same permutation table (same seed) on its own GPU, and there is no data-path
collective.  One all-gather of the small per-gene result table (I, z, p) at the end makes the full
table available on every rank -- over RCCL/xGMI when the process group's backend is "nccl"
(which is RCCL on ROCm), over gloo in the CPU tests.

torch is imported lazily and only here: it provides the process group (launch contract of
``torch.distributed.run``), nothing on the compute path.
"""

from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split: the first ``n_items % world`` ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def world_info() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the process group if initialised, else from the env."""
    d = _dist()
    if d is not None:
        return d.get_rank(), d.get_world_size(), int(os.environ.get("LOCAL_RANK", d.get_rank()))
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def all_gather_rows(local: np.ndarray, n_total: int) -> np.ndarray:
    """All-gather a (rows_local, F) float64 block whose row ranges follow ``shard_bounds``."""
    d = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if d is None or d.get_world_size() == 1:
        return local
    import torch

    world, rank = d.get_world_size(), d.get_rank()
    width = local.shape[1]
    longest = max(b - a for a, b in (shard_bounds(n_total, world, r) for r in range(world)))
    pad = np.zeros((longest, width), dtype=np.float64)
    pad[: local.shape[0]] = local
    on_gpu = d.get_backend() == "nccl"
    mine = torch.from_numpy(pad)
    if on_gpu:
        mine = mine.cuda()
    parts = [torch.empty_like(mine) for _ in range(world)]
    d.all_gather(parts, mine)  # the single collective of the path
    out = np.empty((n_total, width), dtype=np.float64)
    for r, part in enumerate(parts):
        a, b = shard_bounds(n_total, world, r)
        out[a:b] = part.cpu().numpy()[: b - a]
    return out


def morans_i_sharded(adata, genes: Optional[Sequence[str]] = None, key_added: str = "morans_i",
                     compute: Optional[Callable] = None, **kwargs):
    """``morans_i`` with the gene list sharded over the ranks of the current process group.

    Every rank ends with the complete ``adata.uns[key_added]`` table (input gene order), identical
    to an unsharded call: the permutation table depends only on ``seed`` and ``n_cells``, so each
    gene sees the same permutations whichever rank computes it.  ``compute`` defaults to the HIP
    ``morans_i`` on GPU ``LOCAL_RANK``; tests inject a CPU checker to exercise the shard/merge logic.
    """
    rank, world, local_rank = world_info()
    names: List[str] = list(adata.var_names) if genes is None else ([genes] if isinstance(genes, str) else list(genes))
    lo, hi = shard_bounds(len(names), world, rank)
    if compute is None:
        from spatialcore_amd.spatial.autocorrelation import morans_i

        def compute(ad, gene_list, **kw):
            return morans_i(ad, genes=gene_list, key_added="_shard", device=local_rank, **kw).uns.pop("_shard")

    mine = names[lo:hi]
    cols = ["I", "expected_I", "z_score", "p_value"]
    if mine:
        df = compute(adata, mine, **kwargs)
        local = df[cols].to_numpy(dtype=np.float64)
    else:
        local = np.zeros((0, len(cols)))
    full = all_gather_rows(local, len(names))
    table = pd.DataFrame(full, columns=cols)
    table.insert(0, "gene", names)
    adata.uns[key_added] = table
    return adata
