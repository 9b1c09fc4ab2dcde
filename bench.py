"""Headline benchmark: genes/sec of global Moran's I (1000 permutations, 1M cells, k=15) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full pass of the hot path over one batch of synthetic input that is already
resident (coordinates on the host as the API takes them, raw expression tiles in HBM): exact kNN build,
row-normalised graph, numpy-exact permutation table for `seed`, value classes + centring + the narrow
copy of the raw values the permutation kernel gathers, lag, the permutation kernel for all genes,
p-value assembly, and (N > 1) ONE RCCL all-gather of the per-gene results.  Nothing derived is cached
between steps: only the uploaded inputs (coordinates on the host, raw fp64 tiles on the device) persist.

`value` is that resident-operand step (the contract: inputs in HBM when the timed region starts);
`value_public_api` is SURVEY 8(d)'s literal metric, G / wall time of morans_i(adata, ...) with host
arrays in and a DataFrame out.  `value_uint16_source` / `value_float32_source` repeat the step with the
gathered operand forced to the wider exact types (counts >= 256; log-normalised float matrices).

`python bench.py --gpus N` with no launcher in front starts its own N ranks (launch_ranks).

Workloads
  default          BASELINE.json configs[1] per GPU (1M cells, 500 genes, k=15, P=1000); genes shard across
                   ranks with no data-path collective, so scaling is "weak" (total genes = 500 * N).
  --config 3       BASELINE.json configs[3]: 5M cells, 2000 genes IN TOTAL sharded over the N ranks
                   ("strong" scaling), processed on each rank in batches of <= --gene-batch genes that reuse the
                   rank's resident permutation table.

No PyTorch: ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT from the launcher's environment; barrier,
slowest-rank clock and the final all-gather go through the library's own RCCL communicator
(spatialcore_amd.parallel.connect -> sc_comm_create / sc_allgather / sc_allreduce_max); the device fence is
hipDeviceSynchronize (sc_ctx_sync), which is what torch.cuda.synchronize() would call.
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


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc  # test infrastructure: CPU baseline + verification legs only, after the timed region

    orc.build_c()
    return orc


def cpu_baseline_and_verify(coords, X, k: int, n_perm_full: int, seed: int, gpu_res: dict, check_genes):
    """(a) CPU baseline: the oracle's C port of the path on the box's host cores, on a bounded sample of the same
    workload, in both forms -- the reference-faithful one (permute the graph rows and redo the CSR sweep, as
    squidpy -> scanpy do behind AC:576-583) and the gather form -- single-threaded (the reference passes n_jobs=1,
    AC:580) and with OpenMP over genes (scanpy's kernel is @njit(parallel=True), prange over genes).
    (b) Verification of the timed GPU result: observed I and the count #{sims >= I} over all P permutations for
    `check_genes`, recomputed by the oracle from its own kNN graph and its own numpy-exact permutation table."""
    orc = _oracle()
    n = coords.shape[0]
    threads = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("SC_CPU_THREADS", 32))))
    t0 = time.perf_counter()
    g = orc.row_normalize_l1(orc.csr_matrix((np.ones(n * k), orc.knn_tree(coords, k).reshape(-1),
                                             np.arange(0, n * k + 1, k)), shape=(n, n)))
    t_graph = time.perf_counter() - t0
    perms, _ = orc.perm_table(seed, n, n_perm_full)

    def timed_forms(genes: int, nthreads: int, budget_s: float):
        used = orc.set_threads(nthreads)
        vals = np.ascontiguousarray(X[:, :genes].T, dtype=np.float64)
        t0 = time.perf_counter()
        orc.morans_i_scores(g, vals)                        # observed statistic = one sweep
        t_obs = time.perf_counter() - t0
        p_row = int(max(2, min(n_perm_full, (budget_s / 2) / max(t_obs, 1e-3))))
        t0 = time.perf_counter()
        for p in range(p_row):
            orc.morans_i_scores(g, vals, perms[p])
        t_row = (time.perf_counter() - t0) / p_row
        p_gat = int(min(n_perm_full, max(8, 16 * p_row)))
        t0 = time.perf_counter()
        orc.morans_i_sims_gather(g, vals, perms[:p_gat])    # includes z / lag / scale set-up (one sweep)
        t_gat = (time.perf_counter() - t0) / p_gat
        rate_row = genes / (t_obs + t_row * n_perm_full)
        rate_gat = genes / (t_obs + t_gat * n_perm_full)
        return used, rate_row, rate_gat, f"{genes} genes x {p_row} (row-permuted) / {p_gat} (gather) of {n_perm_full} permutations"

    g_mt = int(min(X.shape[1], max(8, threads)))
    used_mt, row_mt, gat_mt, s_mt = timed_forms(g_mt, threads, 10.0)
    _, row_1, gat_1, s_1 = timed_forms(min(8, X.shape[1]), 1, 10.0)
    orc.set_threads(threads)
    base = {"value": row_mt, "unit": "genes/s", "cores": used_mt, "kind": "port",
            "sample": f"{s_mt} at {n} cells, k={k}, reference-faithful row-permuted CSR sweep (C port of the "
                      f"squidpy/scanpy form), OpenMP over genes on {used_mt} threads, scaled linearly in permutations; "
                      f"kNN graph by cKDTree took {t_graph:.1f}s and is not included",
            "host_cpus": os.cpu_count(), "threads_available": len(os.sched_getaffinity(0)),
            "forms_genes_per_s": {"row_permuted_1_thread": row_1, f"row_permuted_{used_mt}_threads": row_mt,
                                  "gather_1_thread": gat_1, f"gather_{used_mt}_threads": gat_mt},
            "sample_1_thread": s_1}

    # ---- verification of the timed GPU result (2 genes, all P permutations) ----
    # The bench genes are integer counts on a kNN graph, i.e. lattice genes (DESIGN.md "Ties"): the statistic's
    # permutation-dependent part is the integer T_p = sum_i x_i S[perm_p(i)], and BOTH sides decide sims >= I on those
    # integers -- the oracle in int64 (orc.morans_count_ge), the device in exact fp64 -- so the counts must be EQUAL,
    # exact ties included (41 of the 250 Poisson genes have one among 1000 permutations).  No tolerance.
    cols = list(check_genes)
    vals = np.ascontiguousarray(X[:, cols].T, dtype=np.float64)
    I_ref = orc.morans_i_scores(g, vals)
    sims = orc.morans_i_sims_gather(g, vals, perms)
    want, lattice = orc.morans_count_ge(g, vals, perms, sims, I_ref)
    exact_ties = (np.abs(sims - I_ref) <= 1e-9 * sims.std(axis=0)).sum(axis=0)     # reported, no longer tolerated
    I_gpu, c_gpu = gpu_res["I"][cols], gpu_res["count_ge"][cols]
    rel = float(np.max(np.abs(I_gpu - I_ref) / np.abs(I_ref)))
    ok = bool(rel <= 1e-9 and lattice.all() and (c_gpu == want).all())
    verify = {"verified": ok, "genes_checked": cols, "max_rel_err_I": rel,
              "count_ge_gpu": [int(v) for v in c_gpu], "count_ge_oracle": [int(v) for v in want],
              "lattice_genes": [bool(v) for v in lattice],
              "exact_ties_resolved": [int(v) for v in exact_ties],
              "method": "oracle (cKDTree graph, scalar C numpy-stream model); counts decided on the exact integer "
                        "lattice T_p >= T_obs on both sides, equality required"}
    return base, verify


def public_api_rate(coords, X, k: int, P: int, seed: int, device: int):
    """G / wall time of morans_i(adata, genes=all, n_neighbors=k, n_permutations=P, seed=seed) on a float32 CSR
    AnnData: everything the user's call pays (validation, PCIe upload of the matrix, graph, obsp side effects,
    DataFrame) -- SURVEY 8(d)'s definition of the metric.  Second of two calls (the first also pays hipMalloc)."""
    import logging

    import pandas as pd
    from scipy import sparse

    from spatialcore_amd import SimpleAnnData
    from spatialcore_amd.spatial import morans_i

    logging.getLogger("spatialcore_amd").setLevel(logging.WARNING)   # stdout carries exactly one JSON line

    Xs = sparse.csr_matrix(X)
    ad = SimpleAnnData(Xs, obs=pd.DataFrame(index=pd.RangeIndex(X.shape[0]).astype(str)),
                       var_names=[f"g{i}" for i in range(X.shape[1])], obsm={"spatial": coords})
    walls = []
    for _ in range(2):
        t0 = time.perf_counter()
        morans_i(ad, genes=list(ad.var_names), n_neighbors=k, n_permutations=P, seed=seed, device=device)
        walls.append(time.perf_counter() - t0)
    return X.shape[1] / walls[1], walls, ad.uns["morans_i"]


def launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` without an external launcher: start N child processes of this same command with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (and a rendezvous file in a fresh private directory)
    set, pass their output through (rank 0 prints the one JSON line), wait for all of them, and return non-zero if
    any failed.  The launcher never touches the GPU, so nothing is exec'ed or forked from a process with HIP state."""
    import shutil
    import socket
    import subprocess
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rdv_dir = tempfile.mkdtemp(prefix="sc_bench_rdv_")      # 0700, this launch only
    procs = []
    try:
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       SC_RENDEZVOUS_FILE=os.path.join(rdv_dir, "rccl_id"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        codes = [None] * n_ranks
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:                      # a dead rank would leave the others waiting in a collective
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.terminate()       # exactly the processes started above
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        try:
                            codes[r] = p.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[r] = p.wait()
                print(f"bench.py launcher: rank(s) {failed} failed (exit codes {[codes[r] for r in failed]})",
                      file=sys.stderr, flush=True)
                return 1
            time.sleep(0.05)
        return 0
    finally:
        shutil.rmtree(rdv_dir, ignore_errors=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=(1, 3),
                    help="BASELINE.json configs index: 1 = 1M cells x 500 genes per GPU (weak scaling, the metric); "
                         "3 = 5M cells x 2000 genes in total (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None)
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--genes", type=int, default=None, help="genes per GPU (weak) / in total (strong)")
    ap.add_argument("--gene-batch", type=int, default=256, help="strong mode: genes resident per batch")
    ap.add_argument("--perms", type=int, default=1000)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline + oracle verification leg")
    ap.add_argument("--no-public-api", action="store_true")
    ap.add_argument("--no-other-sources", action="store_true",
                    help="skip the uint16- / float32-source repeats of the step (value_uint16_source, value_float32_source)")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip roofline.alone (counter passes: its launches would be averaged into the per-launch traffic)")
    ap.add_argument("--source-bits", type=int, default=8, choices=(4, 8, 16, 32, 64),
                    help="narrowest exact copy of the expression values the permutation kernel may gather "
                         "(4: nibble slots for count data when they take fewer rows, 8 / 16: uint8 / uint16 for count data, "
                         "32: float32 raw values, 64: the general fp64 kernel)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="testing only: all ranks share GPU 0 and exchange through files (RCCL refuses duplicate "
                         "devices); exercises the N > 1 code path on a 1-GPU box; never a measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked directly (`python bench.py --gpus N`): this process becomes the launcher of N ranks.  It has made
        # no GPU call and has not even loaded the HIP library; the ranks are fresh child processes.
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0

    from spatialcore_amd import _lib, parallel
    from spatialcore_amd.spatial.autocorrelation import _moran_resident

    strong = (args.scaling or ("strong" if args.config == 3 else "weak")) == "strong"
    n = args.cells or (5_000_000 if args.config == 3 else 1_000_000)
    genes_arg = args.genes or (2000 if args.config == 3 else 500)
    P, k = args.perms, args.k
    if strong:
        g_lo, g_hi = parallel.shard_bounds(genes_arg, world, rank)
        G_total, G_mine = genes_arg, g_hi - g_lo
    else:
        g_lo, G_total, G_mine = rank * genes_arg, genes_arg * world, genes_arg

    # the process-wide context of the device: the AnnData-level morans_i of the public-API leg uses the same one (a second
    # context would be a second set of nine streams competing for the 16 hardware queues)
    ctx = _lib.default_context(local_rank)
    comm = parallel.connect(ctx, transport="file" if rehearse else None)
    comm_ranks = comm.info()[0]     # what RCCL reports (ncclCommCount) when the transport is RCCL
    ctx.set_moran_source_bits(args.source_bits)

    # ---- synthetic inputs; resident in HBM before the timed region (weak mode: the rank's whole matrix) ----
    batch = min(args.gene_batch, max(G_mine, 1)) if strong else G_mine
    coords, X = synth_inputs(n, batch, seed=42, gene_offset=g_lo)
    batches = [(b, min(b + batch, G_mine)) for b in range(0, G_mine, batch)]
    if len(batches) == 1:
        ctx.set_expression(X, np.arange(X.shape[1]))

    def batch_matrix(bi: int):
        # strong mode, later batches: the base block rolled along the cells (new values per gene at no RNG cost;
        # the timing does not depend on the values)
        return X if bi == 0 else np.roll(X, 7919 * bi, axis=0)

    gathered = None
    mem_peak = 0
    batch_walls = []     # (batch index, wall seconds) of every batch call: strong mode reports the generator's share

    import contextlib

    # rehearsal only: the ranks share ONE GPU and take turns on it (parallel.device_turn explains why)
    turn = parallel.device_turn if rehearse else contextlib.nullcontext

    def step():
        nonlocal gathered
        with turn():
            res, mine = device_part()
            if rehearse:
                ctx.sync()
        if world > 1:
            gathered = parallel.all_gather_rows(mine, G_total, comm) if strong else comm.all_gather(mine)
        return res

    def device_part():
        nonlocal mem_peak
        # the neighbour search and the graph are enqueued without waiting (nothing is fetched), then the permutation
        # generator -- it needs only n and the seed, and it is the longest chain of the step -- starts beside them with
        # three chunks enqueued; the scoring joins it (morans_i starts the generator first as well: there an upload follows)
        ctx.knn(coords, k, fetch=False)
        ctx.graph_from_knn(1.0 / k)
        begun = _lib.rng_state_words(np.random.default_rng(args.seed)) if P > 0 else None
        if begun is not None:
            ctx.moran_seeded_begin(begun, n, P, ahead_chunks=3)
        rows = []
        res = None
        for bi, (b0, b1) in enumerate(batches):
            tb = time.perf_counter()
            if len(batches) > 1:
                ctx.set_expression(batch_matrix(bi)[:, : b1 - b0], np.arange(b1 - b0))   # upload inside the step
            if bi == 0:
                res = _moran_resident(ctx, n, P, args.seed, begun=begun)   # generator + scoring, pipelined
            else:
                res = _moran_resident(ctx, n, P, args.seed, reuse_table=True)   # the rank's resident table
            batch_walls.append((bi, time.perf_counter() - tb))   # (every call ends with a stream synchronisation)
            rows.append(np.stack([res["I"], res["p_value"]], axis=1))
        mem_peak = max(mem_peak, ctx.device_mem())
        mine = np.concatenate(rows, axis=0) if rows else np.zeros((0, 2))
        return res, mine

    def fence():
        comm.barrier()          # an RCCL all-reduce at N > 1
        ctx.sync()              # hipDeviceSynchronize

    for _ in range(args.warmup):
        step()
    fence()
    ctx.reset_timers()
    pg0 = ctx.permgen_stats()
    batch_walls.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = float(comm.max_over_ranks([time.perf_counter() - t0])[0])
    timed_batch_walls = list(batch_walls)     # (the legs after the timed region call step() again)
    pg = tuple(a - b for a, b in zip(ctx.permgen_stats(), pg0))

    perm_ms, perm_launches = ctx.kernel_time(_lib.K_MORAN_PERM)
    lag_ms, _ = ctx.kernel_time(_lib.K_LAG)
    scan_ms, _ = ctx.kernel_time(_lib.K_PERM_SCAN)
    swap_ms, _ = ctx.kernel_time(_lib.K_PERM_SWAP)
    knn_ms, _ = ctx.kernel_time(_lib.K_KNN)
    fallbacks = int(comm.max_over_ranks([float(pg[2])])[0])
    # The same kernel with the chip to itself (after the timed region, rank 0, single batch): the first 128 permutations
    # of the resident table scored again.  Inside the pipeline it runs on the CUs it leaves to the generator's side
    # and beside the generator's traffic; this is the kernel's own rate.  Reported as roofline.alone, never as `value`.
    timed_bits = ctx.moran_source_bits()
    timed_lag_bits = ctx.moran_lag_bits()
    timed_groups = ctx.moran_row_groups()
    alone = None
    if rank == 0 and len(batches) == 1 and not rehearse and P >= 128 and not args.no_alone:
        ctx.reset_timers()
        for _ in range(3):
            ctx.moran(128, return_sims=False)
        a_ms, a_cnt = ctx.kernel_time(_lib.K_MORAN_PERM)
        if a_cnt:
            alone = (a_ms / a_cnt, 128)

    def kernel_roofline(kern_ms, kern_launches, steps, bits, lag_bits=64, row_groups=None):
        """Bytes the scoring kernel's formulation has to move per launch (its algorithmic bytes) over its average
        HIP-event launch time.  (1) per step: one 128-byte row of raw values + one 4-byte index per (permutation,
        cell, gene group of 128 / 64 / 32 / 16 genes) and the lag rows of every gene once per launch (a launch =
        one chunk of permutations x all gene groups; fp64, or -- r04, count batches -- the 16-bit neighbour sums).  (2) SURVEY 8(d)'s streaming model (16 B per (permutation, gene,
        cell) + 4 B per (permutation, cell)) as an EFFECTIVE rate: the kernel moves fewer bytes than the model."""
        genes_per_row = {4: 256, 8: 128, 16: 64, 32: 32, 64: 16}[bits]   # (4: nibble SLOTS; a gene with counts >= 16 takes two)
        per_step = max(kern_launches // max(steps, 1), 1)
        avg = kern_ms / max(kern_launches, 1)
        g_pad = -(-batch // genes_per_row) * genes_per_row * len(batches)
        grp = g_pad // genes_per_row
        if bits == 4 and row_groups:
            grp = row_groups * len(batches)
            g_pad = grp * genes_per_row
        step_bytes = grp * (P * n * (128.0 + 4.0)) + per_step * n * (lag_bits / 8.0) * g_pad
        ach = step_bytes / per_step / (avg * 1e-3) / 1e9 if kern_launches else 0.0
        eff = (P * G_mine * n * 16.0 + P * n * 4.0) / per_step / (avg * 1e-3) / 1e9 if kern_launches else 0.0
        return {"genes_per_row": genes_per_row, "lag_bits": lag_bits, "launches_per_step": per_step, "avg_ms": avg, "g_pad": g_pad, "groups": grp,
                "step_bytes": step_bytes, "launch_bytes": step_bytes / per_step, "achieved": ach, "effective": eff}

    # The same step with the gathered operand forced to the wider exact types (after the timed region, rank 0, single
    # batch): what a panel with counts >= 256 (uint16 rows, 64 genes each) or a log-normalised float matrix (float32
    # rows, 32 genes each) pays.  The statistics are those of the timed run (lattice genes: same integers).
    other_sources = {}
    if rank == 0 and world == 1 and len(batches) == 1 and not rehearse and not args.no_other_sources and args.source_bits <= 8:
        for bits_alt in ((8, 16, 32) if timed_bits == 4 else (4, 16, 32)):
            ctx.set_moran_source_bits(bits_alt)
            step()
            ctx.sync()
            ctx.reset_timers()
            t1 = time.perf_counter()
            for _ in range(3):
                res_alt = step()
            ctx.sync()
            dt = time.perf_counter() - t1
            k_ms, k_cnt = ctx.kernel_time(_lib.K_MORAN_PERM)
            rf = kernel_roofline(k_ms, k_cnt, 3, bits_alt, ctx.moran_lag_bits(), ctx.moran_row_groups())
            other_sources[bits_alt] = {
                "value": G_total * 3 / dt, "ms_per_step": dt / 3 * 1e3, "steps": 3, "source_bits": ctx.moran_source_bits(),
                "roofline": {"achieved": rf["achieved"], "frac": rf["achieved"] / HBM_PEAK_GBS, "avg_launch_ms": rf["avg_ms"],
                             "launches": k_cnt, "algorithmic_bytes_per_launch": rf["launch_bytes"],
                             "step_frac": rf["step_bytes"] / (dt / 3) / 1e9 / HBM_PEAK_GBS},
                "counts_equal_to_timed_run": bool((res_alt["count_ge"] == res["count_ge"]).all()),
                "I_equal_to_timed_run": bool((res_alt["I"] == res["I"]).all())}
        ctx.set_moran_source_bits(args.source_bits)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = G_total * args.steps / elapsed
        source_bits = timed_bits
        kernel_name = "k_moran_score"
        rf = kernel_roofline(perm_ms, perm_launches, args.steps, source_bits, timed_lag_bits, timed_groups)
        launches_per_step, avg_ms, G_pad, groups = rf["launches_per_step"], rf["avg_ms"], rf["g_pad"], rf["groups"]
        kernel_bytes, achieved, effective = rf["launch_bytes"], rf["achieved"], rf["effective"]
        # (3) PMC counters of the same kernel build, collected by scripts/pmc_traffic.py (separate --pmc passes)
        traffic, traffic_note = None, "no PMC file for this kernel"
        tpath = os.path.join(ROOT, "profiles", f"{kernel_name}_pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            same = (tj.get("cells") == n and tj.get("perms") == P and tj.get("kernel") == kernel_name
                    and tj.get("genes_per_gpu") == G_mine)
            if same and tj.get("source_hash") == _lib.source_hash():
                traffic, traffic_note = tj.get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(tpath)
            else:
                traffic_note = ("stale: " + os.path.basename(tpath) + " was measured on another workload or another "
                                "build of csrc/sc_moran.hip (source_hash mismatch); re-run scripts/pmc_traffic.py")
        line = {
            "metric": "genes/sec Moran's I (1000 perms, 1M cells, k=15)",
            "value": value,
            "unit": "genes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL on one GPU, file transport -- not a measurement)",
            "config": {"workload": f"{n} cells (uniform 2-D), "
                                   + (f"{G_total} genes in total sharded over {world} GPU(s) in batches of <= {batch}"
                                      if strong else f"{G_mine} genes per GPU")
                                   + f", k={k} kNN, {P} numpy-exact permutations, seed={args.seed}"
                                   + "; value = the resident-operand step (inputs in HBM), value_public_api = wall time of "
                                     "morans_i(adata, ...) with host arrays in and a DataFrame out (SURVEY 8(d) literal)"
                                   + (" (BASELINE configs[1])" if (n, genes_arg, P, k, strong) == (1_000_000, 500, 1000, 15, False)
                                      else " (BASELINE configs[3])" if (n, genes_arg, P, k, strong) == (5_000_000, 2000, 1000, 15, True)
                                      else " (non-default size)"),
                       "expression_source": {4: "4-bit slots (counts < 16 one slot, counts < 256 two)", 8: "uint8 (counts < 256)",
                                             16: "uint16 (counts)", 32: "float32", 64: "float64"}[source_bits],
                       "cells": n, "genes_per_gpu": G_mine, "genes_total": G_total, "k": k, "perms": P,
                       "parallelism": f"gene-shard x{world}, one RCCL all-gather of (I, p)"
                                      + (" [file transport, rehearsal]" if rehearse else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name, "kernel_forms": "k_moran_score_wg (every chunk of the pipeline's schedule: 32, remainder, 128 x k, "
                                                                "96, 48, 24 permutations; a chunk whose last task holds fewer than 24 would "
                                                                "take k_moran_score); same arithmetic; the average is over all launches",
                         "avg_launch_ms": avg_ms, "launches": perm_launches,
                         "algorithmic_bytes_per_launch": kernel_bytes, "lag_bits": timed_lag_bits,
                         "step_frac": rf["step_bytes"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "step_frac_basis": "the same compulsory bytes of one step / ms_per_step / peak: what the whole "
                                            "step (graph, generator, lag, scoring, p-values) sustains",
                         "basis": "bytes this kernel's formulation must move (128-B raw-value row + 4-B index per "
                                  "(permutation, cell, tile), lag rows once per launch) / HIP-event launch time",
                         "effective": effective, "effective_frac": effective / HBM_PEAK_GBS,
                         "effective_basis": "SURVEY 8(d) streaming model, 16 B per (permutation, gene, cell) + 4 B per "
                                            "(permutation, cell); may exceed the peak, the kernel moves fewer bytes",
                         "traffic_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and perm_launches else None,
                         "traffic_source": traffic_note,
                         "alone": None if alone is None else {
                             "launch_ms": alone[0], "permutations": alone[1],
                             "achieved": (groups * alone[1] * n * 132.0 + n * (timed_lag_bits / 8.0) * G_pad) / (alone[0] * 1e-3) / 1e9,
                             "frac": (groups * alone[1] * n * 132.0 + n * (timed_lag_bits / 8.0) * G_pad) / (alone[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "the same kernel on the whole chip, no generator beside it (128 permutations of the "
                                     "resident table, after the timed region)"}},
            "breakdown_ms_per_step": {"perm_scan_overlapped": scan_ms / args.steps, "perm_swaps": swap_ms / args.steps,
                                      "moran_perm_kernel": perm_ms / args.steps,
                                      "lag_kernel": lag_ms / args.steps, "knn_kernel": knn_ms / args.steps},
            "permgen_stats": {"jobs_block_parallel": pg[0], "jobs_sequential": pg[1],
                              "verification_fallbacks_max_over_ranks": fallbacks,
                              "blocks_prepared": pg[3], "blocks_chain": pg[4],
                              "note": ctx.permgen_note()},
            "device_mem_bytes_rank0": mem_peak,
            "nccl_ranks": comm_ranks,
        }
        for bits_alt, name in ((4, "4bit"), (8, "uint8"), (16, "uint16"), (32, "float32")):
            if bits_alt in other_sources:
                line[f"value_{name}_source"] = other_sources[bits_alt]["value"]
                line[f"{name}_source"] = other_sources[bits_alt]
        if strong:
            # Every rank regenerates the FULL permutation table (it depends on the seed and n alone), pipelined with the
            # scoring of its first gene batch; the later batches score against the resident table.  A rank's step is
            # therefore first_batch + (its batches - 1) x later_batch, and only the second term shrinks with more ranks.
            first = [w for bi, w in timed_batch_walls if bi == 0]
            later = [w for bi, w in timed_batch_walls if bi > 0]
            first_ms = 1e3 * sum(first) / max(len(first), 1)
            later_ms = 1e3 * sum(later) / max(len(later), 1) if later else None
            per = later_ms if later_ms is not None else first_ms

            def model(nr):
                nb = -(-(-(-genes_arg // nr)) // batch)      # batches of the largest shard
                return first_ms + (nb - 1) * per
            line["strong_scaling"] = {
                "first_batch_ms": first_ms, "later_batch_ms": later_ms, "batches_this_rank": len(batches), "gene_batch": batch,
                "generator_kernels_ms_per_step": scan_ms / args.steps, "scoring_kernels_ms_per_step": perm_ms / args.steps,
                "model": "t(N) = first_batch_ms + (batches of the largest shard - 1) x later_batch_ms; the first batch holds the "
                         "whole generator job (every rank regenerates the full table), so it does not shrink with N",
                "model_ms": {str(nr): model(nr) for nr in (1, 2, 4, 8)},
                "floor_ms": first_ms,
                "status": "unmeasured on hardware for N > 1: a bound from this rank's own per-batch times"}
        if fallbacks:
            line["warning"] = "the block-parallel generator fell back to the sequential scan inside the timed region"
        if world == 1 and not strong and not args.no_public_api:
            rate, walls, _ = public_api_rate(coords, X, k, P, args.seed, local_rank)
            line["value_public_api"] = rate
            line["public_api"] = {"call": f"morans_i(adata, genes=all {X.shape[1]}, n_neighbors={k}, n_permutations={P}, "
                                          f"seed={args.seed}) on a float32 CSR AnnData, host arrays in, DataFrame out",
                                  "wall_s_first_call": walls[0], "wall_s": walls[1]}
        if world == 1 and not strong and not args.no_cpu_baseline:
            base, verify = cpu_baseline_and_verify(coords, X, k, P, args.seed, res, (0, X.shape[1] - 1))
            line["cpu_baseline"] = base
            line["verified"] = verify["verified"] and fallbacks == 0
            line["verification"] = verify
        print(json.dumps(line), flush=True)
    comm.barrier()
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
