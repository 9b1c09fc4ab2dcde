"""Where does the public morans_i call spend its wall time at configs[1] size? (GPU box; not a test)"""
import cProfile, logging, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
import pandas as pd
from scipy import sparse
sys.argv = ["bench.py"]
import bench
from spatialcore_amd import SimpleAnnData, _lib
from spatialcore_amd.spatial import morans_i
logging.getLogger("spatialcore_amd").setLevel(logging.WARNING)
n, G = 1_000_000, 500
coords, X = bench.synth_inputs(n, G, seed=42)
Xs = sparse.csr_matrix(X)
print("nnz", Xs.nnz, "density", Xs.nnz / (n * G), flush=True)
for name, M in (("csr", Xs), ("dense", X)):
    ad = SimpleAnnData(M, obs=pd.DataFrame(index=pd.RangeIndex(n).astype(str)), var_names=[f"g{i}" for i in range(G)], obsm={"spatial": coords})
    morans_i(ad, genes=list(ad.var_names), n_neighbors=15, n_permutations=1000, seed=0)
    t = time.perf_counter(); morans_i(ad, genes=list(ad.var_names), n_neighbors=15, n_permutations=1000, seed=0); print(name, "wall", time.perf_counter() - t, flush=True)
    pr = cProfile.Profile(); pr.enable()
    morans_i(ad, genes=list(ad.var_names), n_neighbors=15, n_permutations=1000, seed=0)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(14)
