"""Debug probe: is sc_moran_seeded bit-reproducible when two processes share one GPU?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from conftest import synth
from spatialcore_amd import _lib
tag = sys.argv[1]
n, G, P = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
coords, X = synth(n, G, 3, dtype=np.float32)
ctx = _lib.Context(0)
outs = []
for rep in range(3):
    ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1 / 15)
    ctx.set_expression(X, np.arange(G))
    w = _lib.rng_state_words(np.random.default_rng(4))
    outs.append(ctx.moran_seeded(w, P))
w = _lib.rng_state_words(np.random.default_rng(4))
ctx.generate_permutations(w, n, P)
two = ctx.moran(P)
np.savez(f"gpurun_out/conc_{tag}.npz", sims=np.stack([o["sims"] for o in outs]), I=np.stack([o["I"] for o in outs]),
         cnt=np.stack([o["count_ge"] for o in outs]), two_sims=two["sims"], two_cnt=two["count_ge"])
print(tag, "done", [int((o["sims"] != two["sims"]).sum()) for o in outs], flush=True)
