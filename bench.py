"""Headline benchmark: genes/sec of global Moran's I (1000 permutations, 1M cells, k=15) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full pass of the hot path over one batch of synthetic input that is already
resident (coordinates on the host as the API takes them, expression tiles in HBM): exact kNN build,
row-normalised graph, numpy-exact permutation table for `seed`, lag, the permutation kernel for all
genes, p-value assembly, and (N > 1) one RCCL all-gather of the per-gene results.  Nothing is cached
between steps.  Workload at every N: BASELINE.json configs[1] per GPU (1M cells, 500 genes, k=15,
P=1000) -- genes shard across ranks with no data-path collective, so scaling is "weak"
(total genes = 500 * N).

torch is used only as plumbing for the multi-process launch contract (process group, barrier,
all-gather over RCCL); the product itself (spatialcore_amd) does not import it.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def synth_inputs(n_cells: int, n_genes: int, seed: int, gene_offset: int = 0):
    """uniform tie-free coordinates (one cell per ~100 um^2); half the genes spatially
    smooth (sin/cos field x Poisson), half i.i.d. Poisson, lambda ~ LogUniform(0.05, 5); float32."""
    rng = np.random.default_rng(seed)
    L = np.sqrt(n_cells) * 10.0
    coords = rng.uniform(0, L, (n_cells, 2))
    X = np.empty((n_cells, n_genes), dtype=np.float32)
    grng = np.random.default_rng([seed, 1 + gene_offset])
    for g in range(n_genes):
        lam = np.exp(grng.uniform(np.log(0.05), np.log(5.0)))
        if g % 2 == 0:
            wl = grng.uniform(L / 8, L / 2, 2)
            ph = grng.uniform(0, 2 * np.pi, 2)
            field = 1.0 + 0.9 * np.sin(2 * np.pi * coords[:, 0] / wl[0] + ph[0]) * np.cos(
                2 * np.pi * coords[:, 1] / wl[1] + ph[1])
            X[:, g] = grng.poisson(lam * field)
        else:
            X[:, g] = grng.poisson(lam, n_cells)
    return coords, X


def cpu_baseline(coords, X, k: int, n_perm_full: int, seed: int, budget_s: float = 20.0):
    """The oracle's C port of the reference-faithful form (permute the graph rows and redo the CSR
    sweep, as squidpy -> scanpy do for AC:576-583), one thread, on a bounded sample of the same
    workload: 8 genes x as many permutations as fit the budget; scaled linearly to P permutations."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc  # test infrastructure: CPU baseline leg only

    orc.build_c()
    n = coords.shape[0]
    genes = min(8, X.shape[1])
    t0 = time.perf_counter()
    g = orc.row_normalize_l1(orc.csr_matrix((np.ones(n * k), orc.knn_tree(coords, k).reshape(-1),
                                             np.arange(0, n * k + 1, k)), shape=(n, n)))
    t_graph = time.perf_counter() - t0
    vals = np.ascontiguousarray(X[:, :genes].T, dtype=np.float64)
    t0 = time.perf_counter()
    orc.morans_i_scores(g, vals)
    t_obs = time.perf_counter() - t0
    n_perm = int(max(2, min(n_perm_full, budget_s / max(t_obs, 1e-3))))
    t0 = time.perf_counter()
    perms, _ = orc.perm_table(seed, n, n_perm)
    for p in range(n_perm):
        orc.morans_i_scores(g, vals, perms[p])
    t_perm = time.perf_counter() - t0
    per_gene_s = (t_obs + t_perm * n_perm_full / n_perm) / genes
    return {"value": 1.0 / per_gene_s, "unit": "genes/s", "cores": 1, "kind": "port",
            "sample": f"{genes} genes x {n_perm} of {n_perm_full} permutations at {n} cells, k={k}, "
                      f"row-permuted CSR sweep (scalar C port of the squidpy/scanpy form), scaled linearly; "
                      f"kNN graph build by cKDTree took {t_graph:.1f}s and is not included",
            "host_cpus": os.cpu_count()}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=1_000_000)
    ap.add_argument("--genes", type=int, default=500, help="genes per GPU")
    ap.add_argument("--perms", type=int, default=1000)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--source-bits", type=int, default=32, choices=(32, 64),
                    help="narrowest exact copy of the expression values the permutation kernel may gather "
                         "(32: float32 raw values, 64: the general fp64 kernel)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="testing only: all ranks share GPU 0 and talk over gloo (exercises the N > 1 code path "
                         "on a 1-GPU box; never a measurement)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    import torch.distributed as dist

    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = "cpu" if rehearse else "cuda"

    from spatialcore_amd import _lib
    from spatialcore_amd.spatial.autocorrelation import _moran_resident

    n, G, P, k = args.cells, args.genes, args.perms, args.k
    coords, X = synth_inputs(n, G, seed=42, gene_offset=rank * G)
    ctx = _lib.Context(local_rank)
    ctx.set_moran_source_bits(args.source_bits)
    ctx.set_expression(X, np.arange(G))   # inputs resident in HBM before the timed region

    gathered = None

    def step():
        nonlocal gathered
        ctx.knn(coords, k, fetch=False)
        ctx.graph_from_knn(1.0 / k)
        res = _moran_resident(ctx, n, P, args.seed)
        if world > 1:
            mine = torch.from_numpy(np.stack([res["I"], res["p_value"]])).to(dev)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)           # the single RCCL collective: per-gene I and p
            gathered = torch.stack(parts).cpu().numpy()
        return res

    def fence():
        if world > 1:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.reset_timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    perm_ms, perm_launches = ctx.kernel_time(_lib.K_MORAN_PERM)
    lag_ms, _ = ctx.kernel_time(_lib.K_LAG)
    scan_ms, _ = ctx.kernel_time(_lib.K_PERM_SCAN)
    swap_ms, _ = ctx.kernel_time(_lib.K_PERM_SWAP)
    knn_ms, _ = ctx.kernel_time(_lib.K_KNN)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = G * world * args.steps / elapsed
        # algorithmic bytes (SURVEY.md 8(d) streaming model, the contract figure): 16 B per (permutation, gene,
        # cell) [z + gathered lag, fp64] + 4 B per (permutation, cell) -> per step tiles16 x (P*16*n*16 + P*n*4),
        # divided by the launches of a step (a launch = one 32-gene tile pair x one chunk of permutations).
        # The float32-source kernel moves LESS than this model (4-B gathered operand, streamed lag shared through
        # the caches), so `achieved` can exceed the HBM peak; `traffic` (PMC) is what the launch really fetched
        # and `hbm_frac_measured` prices that against the peak.
        tiles = (G + 15) // 16
        launches_per_step = max(perm_launches // max(args.steps, 1), 1)
        alg_bytes = tiles * (P * 16 * n * 16.0 + P * n * 4.0) / launches_per_step
        avg_ms = perm_ms / max(perm_launches, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if perm_launches else 0.0
        traffic = None
        # the synthetic matrix is float32 like an AnnData X: the library gathers the raw float32 values, 32 genes
        # per row (--source-bits 64 forces the general fp64 kernel)
        source_bits = ctx.moran_source_bits()
        kernel_name = {32: "k_moran_perm32", 64: "k_moran_perm"}[source_bits]
        tpath = os.path.join(ROOT, "profiles", f"{kernel_name}_pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("cells") == n and tj.get("perms") == P and tj.get("kernel") == kernel_name:
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": "genes/sec Moran's I (1000 perms, 1M cells, k=15)",
            "value": value,
            "unit": "genes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL on one GPU over gloo -- not a measurement)",
            "config": {"workload": f"{n} cells (uniform 2-D), {G} genes per GPU, k={k} kNN, "
                                   f"{P} numpy-exact permutations, seed={args.seed}"
                                   + (" (BASELINE configs[1])" if (n, G, P, k) == (1_000_000, 500, 1000, 15) else " (non-default size)"),
                       "expression_source": {32: "float32", 64: "float64"}[source_bits],
                       "cells": n, "genes_per_gpu": G, "genes_total": G * world, "k": k, "perms": P,
                       "parallelism": f"gene-shard x{world}, one all-gather of (I, p)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "hbm_frac_measured": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and perm_launches else None,
                         "kernel": kernel_name, "avg_launch_ms": avg_ms, "launches": perm_launches,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "breakdown_ms_per_step": {"perm_scan_overlapped": scan_ms / args.steps, "perm_swaps": swap_ms / args.steps,
                                      "moran_perm_kernel": perm_ms / args.steps,
                                      "lag_kernel": lag_ms / args.steps, "knn_kernel": knn_ms / args.steps},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(coords, X, k, P, args.seed)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
