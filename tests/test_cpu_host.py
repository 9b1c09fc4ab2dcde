"""CPU-only tests: C-ABI surface, host logic, no-fallback guarantee, shard/merge over gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT, load_golden, make_adata, synth


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "spatialcore_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


def test_cabi_library_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/*.h declares."""
    import ctypes

    from spatialcore_amd import _lib

    lib = _lib.load_library()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert set(declared) == set(_lib.SYMBOLS) | {"sc_last_error"}
    assert lib.sc_version() >= 100


def test_nm_shows_no_undeclared_sc_exports():
    from spatialcore_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (sc_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == _declared_symbols()


def test_no_cpu_fallback_without_gpu():
    from spatialcore_amd import _lib
    from spatialcore_amd.spatial import morans_i

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.SpatialCoreHipError):
        _lib.Context(0)
    coords, X = synth(200, 2, 0)
    with pytest.raises(_lib.SpatialCoreHipError):
        morans_i(make_adata(coords, X), n_permutations=3)


def test_release_default_contexts_without_any_context():
    """Giving the cached per-device contexts back is a no-op when none was created (and needs no GPU)."""
    from spatialcore_amd import _lib

    _lib.release_default_contexts()
    assert _lib._default_ctx == {}


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "spatialcore_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|oracle_c", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_host_permutation_generator_matches_numpy():
    from spatialcore_amd import _lib

    kat = load_golden("rng_kat.npz")
    for ci in range(int(kat["n_cases"])):
        key = f"case{ci}_perms"
        if key not in kat:
            continue
        n, reps = int(kat[f"case{ci}_n"]), kat[key].shape[0]
        w = _lib.rng_state_words(np.random.default_rng(int(kat[f"case{ci}_seed"])))
        np.testing.assert_array_equal(_lib.perm_numpy_host(w, n, reps), kat[key])
        np.testing.assert_array_equal(w, kat[f"case{ci}_final_state"])
    rng = np.random.default_rng(77)
    w = _lib.rng_state_words(rng)
    _lib.perm_numpy_host(w, 1000, 3)
    [rng.permutation(1000) for _ in range(3)]
    rng2 = np.random.default_rng(0)
    _lib.set_rng_state(rng2, w)
    np.testing.assert_array_equal(rng2.permutation(50), rng.permutation(50))
    with pytest.raises(ValueError):
        _lib.rng_state_words(np.random.Generator(np.random.MT19937(1)))


def test_quadrants_match_reference_golden(oracle):
    from spatialcore_amd.spatial import autocorrelation as ac

    g = load_golden("ref_fdr_quadrants.npz")
    np.testing.assert_array_equal(ac._classify_quadrants(g["z"], g["lag"], g["pq"], 0.05), g["quad_sig"])
    np.testing.assert_array_equal(ac._classify_quadrants(g["z"], g["lag"]), g["quad_nosig"])
    np.testing.assert_array_equal(oracle.quadrants(g["z"], g["lag"], g["pq"], 0.05), g["quad_sig"])


def test_simple_anndata_and_metadata():
    from spatialcore_amd import SimpleAnnData
    from spatialcore_amd._metadata import update_metadata

    coords, X = synth(50, 4, 1)
    ad = make_adata(coords, X)
    assert ad.n_obs == 50 and ad.n_vars == 4 and list(ad.var_names) == ["g0", "g1", "g2", "g3"]
    sub = ad[:, ["g2", "g0"]]
    np.testing.assert_array_equal(sub.X.toarray(), X.toarray()[:, [2, 0]])
    cp = ad.copy()
    cp.uns["x"] = 1
    cp.obsm["spatial"][0, 0] = -1
    assert "x" not in ad.uns and ad.obsm["spatial"][0, 0] != -1
    update_metadata(ad, "f", {"a": 1, "b": [1, 2], "c": np.zeros(2), "d": {"e": None}}, {"uns": "k"})
    update_metadata(ad, "g", {})
    ops = ad.uns["spatialcore_metadata"]["operations"]
    assert [o["function"] for o in ops] == ["f", "g"] and ops[0]["parameters"]["c"] == "ndarray"
    assert ops[0]["outputs"] == {"uns": "k"} and "outputs" not in ops[1]
    with pytest.raises(ValueError):
        SimpleAnnData(np.zeros(3))


def test_argument_validation_happens_before_the_gpu_is_touched():
    from spatialcore_amd.spatial import compute_neighborhood_profile, lees_l, morans_i

    coords, X = synth(100, 3, 2)
    ad = make_adata(coords, X, labels=np.array(["a", "b"] * 50))
    with pytest.raises(ValueError, match="n_neighbors must be >= 1"):
        morans_i(ad, n_neighbors=0)
    with pytest.raises(ValueError, match="n_permutations must be >= 0"):
        lees_l(ad, ("g0", "g1"), n_permutations=-2)
    with pytest.raises(ValueError, match="Genes not found"):
        lees_l(ad, [("g0", "zz")])
    with pytest.raises(ValueError, match="Spatial coordinates are required"):
        morans_i(ad, spatial_key="missing")
    with pytest.raises(ValueError, match="Invalid method"):
        compute_neighborhood_profile(ad, "cell_type", method="grid")
    with pytest.raises(ValueError, match="Column 'nope' not found"):
        compute_neighborhood_profile(ad, "nope")
    with pytest.raises(ValueError, match="radius must be > 0"):
        compute_neighborhood_profile(ad, "cell_type", method="radius", radius=0)
    ad.obs.loc[ad.obs.index[3], "cell_type"] = None
    with pytest.raises(ValueError, match="missing labels"):
        compute_neighborhood_profile(ad, "cell_type")


def test_morans_i_highly_variable_quirk_is_mirrored():
    """AC:576-583 + AC:617-621: with a var['highly_variable'] column the reference fails for non-HVG genes."""
    import pandas as pd

    from spatialcore_amd import SimpleAnnData
    from spatialcore_amd.spatial import morans_i

    coords, X = synth(60, 3, 1)
    var = pd.DataFrame({"highly_variable": [True, False, True]}, index=["g0", "g1", "g2"])
    ad = SimpleAnnData(X, var_names=["g0", "g1", "g2"], var=var, obsm={"spatial": coords})
    with pytest.raises(RuntimeError, match="Gene 'g1' was passed to squidpy but not found in results"):
        morans_i(ad, genes=["g0", "g1"], n_permutations=2)
    assert list(ad[:, ["g2", "g0"]].var["highly_variable"]) == [True, True]


def test_shard_bounds_cover_everything_once():
    from spatialcore_amd.parallel import shard_bounds

    for n in (0, 1, 7, 500, 2000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(5, 2, 2)


_WORKER_COMMON = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, pandas as pd
import oracle as orc
from conftest import synth, make_adata
from spatialcore_amd.parallel import morans_i_sharded, world_info
coords, X = synth(600, 7, 3)
ad = make_adata(coords, X)
calls = []
def cpu_checker(adata, gene_list, n_neighbors=6, n_permutations=10, seed=0):
    # the oracle stands in for the per-shard compute so that the shard/merge logic runs on CPU
    cols = [int(g[1:]) for g in gene_list]
    calls.append(cols)
    t = orc.morans_i_reference_table(coords, X, cols, n_neighbors, n_permutations, seed)
    return pd.DataFrame({{"gene": gene_list, "I": t["I"], "expected_I": t["expected_I"], "z_score": t["z_score"], "p_value": t["p_value"]}})
genes = [f"g{{i}}" for i in (5, 0, 3, 6, 1, 2, 4)]
def check(rank):
    full = orc.morans_i_reference_table(coords, X, [5, 0, 3, 6, 1, 2, 4], 6, 9, 4)
    df = ad.uns["morans_i"]
    assert list(df["gene"]) == genes
    np.testing.assert_array_equal(df["I"].values, full["I"])
    np.testing.assert_array_equal(df["p_value"].values, full["p_value"])
    np.testing.assert_array_equal(df["z_score"].values, full["z_score"])
    assert len(calls) == 1 and len(calls[0]) == (4 if rank == 0 else 3)
    open(os.path.join(os.environ["SC_TEST_OUT"], f"ok_{{rank}}"), "w").write("ok")
"""

_GLOO_WORKER = _WORKER_COMMON + r"""
import torch, torch.distributed as dist
dist.init_process_group("gloo")
class GlooComm:
    # torch lives in this TEST only: the product's collective is RCCL through the C ABI (or FileComm for rehearsal)
    world, rank = dist.get_world_size(), dist.get_rank()
    def all_gather(self, block):
        mine = torch.from_numpy(np.ascontiguousarray(block, dtype=np.float64))
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        return torch.stack(parts).numpy()
    def close(self): pass
rank, world, _ = world_info()
assert (rank, world) == (dist.get_rank(), 2)
morans_i_sharded(ad, genes=genes, compute=cpu_checker, comm=GlooComm(), n_neighbors=6, n_permutations=9, seed=4)
check(rank)
dist.barrier(); dist.destroy_process_group()
"""

_FILE_WORKER = _WORKER_COMMON + r"""
rank, world, _ = world_info()
# no comm given + a CPU compute: morans_i_sharded opens the file transport itself (and closes it)
morans_i_sharded(ad, genes=genes, compute=cpu_checker, n_neighbors=6, n_permutations=9, seed=4)
check(rank)
from spatialcore_amd.parallel import connect
c = connect(None, transport="file")
assert c.max_over_ranks([float(rank), 5.0 - rank]).tolist() == [1.0, 5.0]
got = c.all_gather(np.full((2, 3), float(rank)))
assert got.shape == (2, 2, 3) and (got[0] == 0).all() and (got[1] == 1).all()
big = np.array([[2**40 + 3 + rank, 7], [rank, 2**62 // 4]], dtype=np.int64)     # integers beyond float64's 2^53 survive
np.testing.assert_array_equal(c.sum_over_ranks_i64(big), np.array([[2**41 + 7, 14], [1, 2**61]], dtype=np.int64))
assert c.info()[:2] == (2, rank)
c.barrier(); c.close()
"""


def test_gene_sharding_world_size_2_gloo(tmp_path, oracle):
    """N > 1 path on CPU: two gloo ranks shard 7 genes 4/3, all-gather, and both hold the table an
    unsharded run gives."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", SC_TEST_OUT=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_gene_sharding_world_size_2_file_transport_without_torch(tmp_path, oracle):
    """The same shard/merge through the product's own rehearsal transport (FileComm), launched as two plain
    processes with RANK / WORLD_SIZE in the environment: nothing on this path imports torch."""
    script = tmp_path / "worker.py"
    script.write_text(_FILE_WORKER.format(root=ROOT) + "\nassert 'torch' not in sys.modules\n")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", OMP_NUM_THREADS="1",
                   SC_TEST_OUT=str(tmp_path), SC_RENDEZVOUS_FILE=str(tmp_path / "rdv"), SC_COMM_TRANSPORT="file", SC_COMM_TIMEOUT_S="60")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()
    assert not list(tmp_path.glob("rdv*"))                      # the transport cleans up after itself


def test_rendezvous_name_is_shared_by_siblings_and_overridable(monkeypatch):
    from spatialcore_amd import parallel

    monkeypatch.delenv("SC_RENDEZVOUS_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29999")
    a = parallel.rendezvous_file(0)
    assert a == parallel.rendezvous_file(0) != parallel.rendezvous_file(1)
    assert f"_{os.getppid()}_" in a and a.endswith("_29999.0")
    monkeypatch.setenv("SC_RENDEZVOUS_FILE", "/tmp/x/y")
    assert parallel.rendezvous_file(3) == "/tmp/x/y.3"
    monkeypatch.setenv("WORLD_SIZE", "1")
    solo = parallel.connect()
    assert solo.world == 1 and solo.all_gather(np.ones(3)).shape == (1, 3)


def test_rendezvous_ignores_stale_and_foreign_files(tmp_path, monkeypatch):
    """A file left at the rendezvous path by an earlier (crashed) launch, or put there by someone else, carries
    another launch nonce (or none): a reader keeps polling until THIS launch's rank 0 has published."""
    import threading
    import time

    from spatialcore_amd import parallel

    monkeypatch.setenv("MASTER_PORT", "29998")
    path = str(tmp_path / "rdv.0")
    nonce = parallel._launch_nonce()
    assert len(nonce) == 16 and nonce == parallel._launch_nonce()
    monkeypatch.setenv("MASTER_PORT", "29997")
    assert parallel._launch_nonce() != nonce                    # another launch, another nonce
    monkeypatch.setenv("MASTER_PORT", "29998")
    with open(path, "wb") as f:
        f.write(b"\x01" * 128)                                  # pre-nonce format / foreign file
    with pytest.raises(TimeoutError):
        parallel._await_id(path, nonce, 128, 0.2)
    with open(path, "wb") as f:
        f.write(b"\x02" * 16 + b"\x03" * 128)                   # right size, someone else's nonce
    with pytest.raises(TimeoutError):
        parallel._await_id(path, nonce, 128, 0.2)

    def late_rank0():
        time.sleep(0.3)
        parallel._publish(path, nonce + b"\x07" * 128)

    t = threading.Thread(target=late_rank0)
    t.start()
    assert parallel._await_id(path, nonce, 128, 10.0) == b"\x07" * 128
    t.join()


def test_bench_launches_its_own_ranks_and_reports_a_failed_rank():
    """`python bench.py --gpus 2` with no launcher in front: the parent (which never touches the GPU) starts the two
    ranks itself and exits non-zero when one fails -- here both do, there is no GPU (the success path runs in
    tests/test_gpu_00_multirank.py)."""
    from spatialcore_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the -m gpu test of the same launcher")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "1000", "--genes", "4",
                          "--perms", "10", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env,
                         timeout=300)
    assert res.returncode == 1
    assert "bench.py launcher: rank(s)" in res.stderr and "no ROCm-capable device" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
