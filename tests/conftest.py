"""pytest configuration: markers, paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Every GPU test gets a wall-clock limit (pytest-timeout, when installed): a test that waits for something that
    never comes fails with a traceback after 6 minutes instead of holding the GPU box until it is killed."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(360))


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build libspatialcore_hip.so (hipcc cross-compiles without a GPU) and liboracle.so if they are
    missing or stale, so that the suite does not depend on a prior manual build."""
    import subprocess

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "spatialcore_amd", "csrc"), "-j4"],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc  # oracle/oracle.py (test infrastructure)

    orc.build_c()
    return orc


def synth(n, n_genes, seed, dtype=np.float64, sparse_x=True, normalize=False):
    """Tie-free coordinates + half smooth / half Poisson genes (same recipe as oracle/make_golden.py).
    normalize=True: the usual input of Moran's I instead of raw counts -- per-cell size factors + log1p
    (scanpy's normalize_total + log1p), float32: no value is an integer, no gene sits on the integer lattice, so the
    ordinary (centred, rounded) arithmetic of the float32-source kernels is what runs."""
    from scipy import sparse

    rng = np.random.default_rng(seed)
    L = np.sqrt(n) * 10.0
    coords = rng.uniform(0, L, (n, 2))
    X = np.empty((n, n_genes), dtype=np.float64)
    for g in range(n_genes):
        lam = np.exp(rng.uniform(np.log(0.05), np.log(5.0)))
        if g % 2 == 0:
            wl = rng.uniform(L / 8, L / 2, 2)
            ph = rng.uniform(0, 2 * np.pi, 2)
            field = 1.0 + 0.9 * np.sin(2 * np.pi * coords[:, 0] / wl[0] + ph[0]) * np.cos(
                2 * np.pi * coords[:, 1] / wl[1] + ph[1])
            X[:, g] = rng.poisson(lam * field)
        else:
            X[:, g] = rng.poisson(lam, n)
    if normalize:
        depth = X.sum(axis=1, keepdims=True) + rng.uniform(0.5, 1.5, (n, 1))   # (no empty cell, no two equal depths)
        X = np.log1p(X / depth * np.median(depth)).astype(np.float32)
        dtype = np.float32
    X = X.astype(dtype)
    return coords, (sparse.csr_matrix(X) if sparse_x else X)


def make_adata(coords, X, labels=None):
    import pandas as pd

    from spatialcore_amd import SimpleAnnData

    obs = pd.DataFrame(index=pd.RangeIndex(X.shape[0]).astype(str))
    if labels is not None:
        obs["cell_type"] = labels
    return SimpleAnnData(X, obs=obs, var_names=[f"g{i}" for i in range(X.shape[1])],
                         obsm={"spatial": coords})
