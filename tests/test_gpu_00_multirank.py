"""GPU: the N > 1 path with the HIP compute.  This file sorts first on purpose: its child processes are started
before the pytest process itself has touched the GPU.

A 1-GPU box cannot run RCCL between two ranks (RCCL refuses two ranks on one device), so the two-rank runs share
GPU 0 and exchange their result blocks through the product's rehearsal transport (parallel.FileComm); the RCCL
communicator itself (dlopen, ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclAllReduce) is exercised with a
world of one rank.  The 8-GPU run over xGMI is the driver's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
from conftest import synth, make_adata
from spatialcore_amd.parallel import device_turn, morans_i_sharded, world_info
from spatialcore_amd.spatial import morans_i
assert 'torch' not in sys.modules
def hip_shard(adata, gene_list, **kw):
    # the two ranks share ONE GPU here: they take turns on it (see parallel.device_turn); the HIP compute is the product's
    with device_turn():
        out = morans_i(adata, genes=gene_list, key_added="_shard", device=0, **kw).uns.pop("_shard")
        _lib.default_context(0).sync()
    return out
from spatialcore_amd import _lib
rank, world, local_rank = world_info()
assert world == 2 and local_rank == 0
coords, X = synth(30000, 45, 3, dtype=np.float32)
genes = [f"g{{i}}" for i in np.random.default_rng(1).permutation(45)]
ad = make_adata(coords, X)
morans_i_sharded(ad, genes=genes, compute=hip_shard, n_neighbors=15, n_permutations=130, seed=4)    # 23 / 22 genes
whole = make_adata(coords, X)
with device_turn():
    morans_i(whole, genes=genes, n_neighbors=15, n_permutations=130, seed=4)
    _lib.default_context(0).sync()
a, b = ad.uns["morans_i"], whole.uns["morans_i"]
assert list(a["gene"]) == genes == list(b["gene"])
for col in ("I", "expected_I", "z_score", "p_value"):
    np.testing.assert_array_equal(a[col].values, b[col].values, err_msg=col)       # sharded == unsharded, bit for bit
assert 'torch' not in sys.modules
open(os.path.join(os.environ["SC_TEST_OUT"], f"ok_{{rank}}"), "w").write("ok")
"""


def _launch_two(cmd_of_rank, tmp_path, extra_env=None):
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_PORT="29571",
                   SC_TEST_OUT=str(tmp_path), SC_RENDEZVOUS_FILE=str(tmp_path / "rdv"), SC_COMM_TRANSPORT="file",
                   SC_COMM_TIMEOUT_S="240", **(extra_env or {}))
        procs.append(subprocess.Popen(cmd_of_rank(rank), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return outs


def test_two_ranks_hip_compute_sharded_equals_unsharded(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    _launch_two(lambda rank: [sys.executable, str(script)], tmp_path)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_bench_two_rank_rehearsal_prints_one_valid_line(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--cells", "150000",
           "--genes", "40", "--perms", "400", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    outs = _launch_two(lambda rank: cmd, tmp_path)
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1].splitlines() if ln.startswith("{")]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["genes_total"] == 80 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["permgen_stats"]["verification_fallbacks_max_over_ranks"] == 0
    assert 0 < line["roofline"]["frac"] < 1.0
    # strong-scaling shape (configs[3] logic at toy size): 70 genes in total, batches of 16 reuse the resident table
    cmd2 = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--config", "3",
            "--cells", "140000", "--genes", "70", "--gene-batch", "16", "--perms", "200", "--steps", "1",
            "--warmup", "0"]
    outs = _launch_two(lambda rank: cmd2, tmp_path)
    line = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][0])
    assert line["scaling"] == "strong" and line["config"]["genes_total"] == 70 and line["config"]["genes_per_gpu"] == 35


_ENRICH_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
from conftest import make_adata
from spatialcore_amd import _lib
from spatialcore_amd.parallel import connect, device_turn, world_info
from spatialcore_amd.spatial import neighborhood_enrichment
rank, world, _ = world_info()
rng = np.random.default_rng(4)
n, T, P = 60000, 7, 45
coords = rng.uniform(0, 2500, (n, 2))
labels = np.array([f"t{{c}}" for c in rng.integers(0, T, n)])
class TurnComm:
    # The two ranks share ONE GPU and take turns on it; the collective must run OUTSIDE a rank's turn, or the rank that
    # holds the GPU waits for the other, which waits for the GPU.  (One process per GPU -- the deployment -- has no turns.)
    def __init__(self, inner, turn):
        self.inner, self.turn, self.world, self.rank = inner, turn, inner.world, inner.rank
    def sum_over_ranks_i64(self, values):
        _lib.default_context(0).sync()
        self.turn.__exit__(None, None, None)
        try:
            return self.inner.sum_over_ranks_i64(values)
        finally:
            self.turn.__enter__()
inner = connect(None, transport="file")
ad = make_adata(coords, np.zeros((n, 1)), labels=labels)
turn = device_turn()
with turn:
    neighborhood_enrichment(ad, "cell_type", k=10, n_permutations=P, seed=8, perm_batch=16, rng="philox",
                            comm=TurnComm(inner, turn))
    _lib.default_context(0).sync()
comm = inner
solo = make_adata(coords, np.zeros((n, 1)), labels=labels)
with device_turn():
    neighborhood_enrichment(solo, "cell_type", k=10, n_permutations=P, seed=8, perm_batch=512, rng="philox")
    _lib.default_context(0).sync()
a, b = ad.uns["neighborhood_enrichment"], solo.uns["neighborhood_enrichment"]
for key in ("count", "mean", "std", "zscore", "p_value"):
    np.testing.assert_array_equal(a[key], b[key], err_msg=key)        # permutation shards merge to the one-rank result
try:
    neighborhood_enrichment(ad, "cell_type", k=10, n_permutations=4, comm=comm)
    raise SystemExit("the numpy stream must refuse to be sharded")
except ValueError:
    pass
comm.close()
open(os.path.join(os.environ["SC_TEST_OUT"], f"ok_{{rank}}"), "w").write("ok")
"""


def test_two_ranks_shard_the_permutations_of_the_enrichment_philox(tmp_path):
    """BASELINE configs[4]'s multi-GPU form (SURVEY 8(e) "alternative"): with the counter-based source rank r scores
    permutations shard_bounds(P, 2, r) and ONE integer all-reduce merges the sums -- the table equals a one-rank run bit
    for bit (two ranks with the HIP compute on one GPU, file transport: RCCL refuses duplicate devices)."""
    script = tmp_path / "enrich_worker.py"
    script.write_text(_ENRICH_WORKER.format(root=ROOT))
    _launch_two(lambda rank: [sys.executable, str(script)], tmp_path)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_bench_launches_its_own_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2 ...` exactly as the driver invokes the bench for N = 1, no torchrun in front: the
    parent starts the two ranks itself (fresh processes; it never touches the GPU) and relays rank 0's one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SC_RENDEZVOUS_FILE")}
    env["SC_COMM_TIMEOUT_S"] = "240"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--cells", "150000",
           "--genes", "40", "--perms", "300", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["nccl_ranks"] == 2 and line["config"]["genes_total"] == 80
    assert line["value"] > 0 and line["permgen_stats"]["verification_fallbacks_max_over_ranks"] == 0


def test_rccl_communicator_world_of_one():
    """dlopen(librccl) + ncclGetUniqueId + ncclCommInitRank + ncclAllGather + ncclAllReduce on the real device."""
    from spatialcore_amd._lib import Context, RcclComm

    with Context(0) as ctx:
        comm = RcclComm(ctx, RcclComm.unique_id(), 1, 0)
        block = np.arange(12, dtype=np.float64).reshape(3, 4) * 0.5
        np.testing.assert_array_equal(comm.all_gather(block), block[None])
        np.testing.assert_array_equal(comm.max_over_ranks([1.5, -2.0]), [1.5, -2.0])
        np.testing.assert_array_equal(comm.sum_over_ranks_i64([[2**60 + 1, -5]]), [[2**60 + 1, -5]])
        assert comm.info() == (1, 0, 0)                       # ncclCommCount / ncclCommUserRank / ncclCommCuDevice
        comm.barrier()
        comm.close()
    with pytest.raises(ValueError):
        RcclComm(ctx, b"short", 1, 0)


_SERIAL_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from spatialcore_amd import _lib
n, P = 1_000_000, 150                     # block-parallel generator first; a 128-permutation chunk is 22 launch units
ctx = _lib.Context(0)
rng = np.random.default_rng(5)
coords = rng.uniform(0, 10000, (n, 2))
X = rng.poisson(1.0, (n, 20)).astype(np.float32)
ctx.knn(coords, 6, fetch=False); ctx.graph_from_knn(1.0 / 6)
ctx.set_expression(X, np.arange(20))
walls, outs = [], []
for rep in range(2):
    w = _lib.rng_state_words(np.random.default_rng(9))
    t0 = time.perf_counter()
    outs.append(ctx.moran_seeded(w, P))
    walls.append(time.perf_counter() - t0)
    wh = _lib.rng_state_words(np.random.default_rng(9))
    _lib.perm_numpy_host(wh, n, P)
    np.testing.assert_array_equal(w, wh)                     # generator state after P numpy permutations
par, seq, fallbacks = ctx.permgen_stats()[:3]
for key in ("I", "sims", "count_ge"):
    np.testing.assert_array_equal(outs[0][key], outs[1][key], err_msg=key)
print("NOTE", ctx.permgen_note(), flush=True)
print("STATS", par, seq, fallbacks, walls[0], walls[1], flush=True)
"""


def test_generator_without_concurrent_streams_uses_the_sequential_scan(tmp_path):
    """The block-parallel generator orders its launches through words in device memory, which needs its streams to run
    concurrently.  With two hardware queues for a dozen streams they cannot: the context's one-off probe (r03) notices
    within milliseconds -- or, should the probe's streams happen to overlap, the first job's waits give up after 1 s --,
    the sequential scan returns the numpy-exact result, and the reason is reported (sc_ctx_permgen_note)."""
    script = tmp_path / "serial_worker.py"
    script.write_text(_SERIAL_WORKER.format(root=ROOT))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2")
    run = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    stats = [ln for ln in run.stdout.splitlines() if ln.startswith("STATS")][-1].split()
    par, seq, fallbacks, first, second = int(stats[1]), int(stats[2]), int(stats[3]), float(stats[4]), float(stats[5])
    note = [ln for ln in run.stdout.splitlines() if ln.startswith("NOTE")][-1]
    if (par, seq, fallbacks) == (2, 0, 0):
        pytest.skip("the runtime ran the streams concurrently on two hardware queues: nothing to fall back from")
    assert (par, seq) == (0, 2) and fallbacks in (0, 1), stats    # 0: caught by the probe; 1: by a give-up inside the first job
    assert "sequential scan" in note and "GPU_MAX_HW_QUEUES=2" in note or fallbacks == 1, note
    assert first < 6.0 and second < 5.0, stats      # no 10-s stall any more


_INIT_WORKER = r"""
import os, sys, logging, io
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import spatialcore_amd
from spatialcore_amd import _lib
buf = io.StringIO()
logging.getLogger("spatialcore_amd").addHandler(logging.StreamHandler(buf))
rep = spatialcore_amd.init()
print("REPORT", rep["hw_queues_requested"], int(rep["streams_concurrent"]), rep["generator"].replace("\n", " "), flush=True)
from conftest import make_adata, synth
from spatialcore_amd.spatial import morans_i
coords, X = synth(140001, 6, 3, dtype=np.float32)
ad = make_adata(coords, X)
morans_i(ad, n_neighbors=6, n_permutations=40, seed=2)
op = ad.uns["spatialcore_metadata"]["operations"][-1]
print("FORM", op["parameters"]["permgen_form"].replace("\n", " "), flush=True)
print("PVAL", " ".join(repr(float(v)) for v in ad.uns["morans_i"]["p_value"].values), flush=True)
print("WARNED", int("sequential scan" in buf.getvalue()), flush=True)
"""


def _run_init_worker(tmp_path, **env_extra):
    script = tmp_path / "init_worker.py"
    script.write_text(_INIT_WORKER.format(root=ROOT))
    run = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    return {ln.split()[0]: ln.split(None, 1)[1] for ln in run.stdout.splitlines() if ln.split() and ln.split()[0] in ("REPORT", "FORM", "PVAL", "WARNED")}


def test_init_reports_the_generator_form_and_every_call_records_it(tmp_path):
    """r04: the fast path must not depend silently on who initialised HIP first.
    spatialcore_amd.init() measures whether the generator's streams overlap and says which form it will take;
    morans_i records the form in its provenance entry.  Same p-values either way; a process that cannot overlap
    its streams (two hardware queues) is told so: report, warning on the logger, metadata."""
    fast = _run_init_worker(tmp_path)
    q, ok, gen = fast["REPORT"].split(None, 2)
    assert int(q) == 24 and int(ok) == 1 and gen == "block-parallel", fast
    assert fast["FORM"] == "block-parallel" and fast["WARNED"] == "0", fast
    slow = _run_init_worker(tmp_path, GPU_MAX_HW_QUEUES="2")
    q, ok, gen = slow["REPORT"].split(None, 2)
    if int(ok) == 1:
        pytest.skip("the runtime ran the streams concurrently on two hardware queues: nothing to report")
    assert int(q) == 2 and gen.startswith("sequential: ") and "GPU_MAX_HW_QUEUES=2" in gen, slow
    assert slow["FORM"].startswith("sequential: ") and slow["WARNED"] == "1", slow
    assert slow["PVAL"] == fast["PVAL"]                      # identical results


def test_three_live_contexts_agree_and_say_which_form_they_took():
    """Three contexts in one process have ~33 streams for the 24 hardware queues the library asks for: the third one's
    generator streams may share queues.  Whatever each context's probe finds, the tables are numpy's, and a context that
    took the sequential scan says so (note + form) instead of being silently slower."""
    from spatialcore_amd import _lib

    n, P = 140001, 6
    want = _lib.perm_numpy_host(_lib.rng_state_words(np.random.default_rng(12)), n, P)
    ctxs = [_lib.Context(0) for _ in range(3)]
    try:
        forms = []
        for c in ctxs:
            c.knn(np.random.default_rng(1).uniform(0, 100, (500, 2)), 4, fetch=False)     # (its other streams exist too)
            ok, queues = c.probe_streams()
            got = c.generate_permutations(_lib.rng_state_words(np.random.default_rng(12)), n, P, fetch=True)
            np.testing.assert_array_equal(got, want)
            par, seq = c.permgen_stats()[:2]
            form = c.permgen_form(n)
            forms.append(form)
            if ok:
                assert (par, seq) == (1, 0) and form == "block-parallel", (form, par, seq)
            else:
                assert (par, seq) == (0, 1) and form.startswith("sequential: ") and c.permgen_note(), (form, par, seq)
        assert forms[0] == "block-parallel", forms      # the first context always has its queues to itself
    finally:
        for c in ctxs:
            c.close()
