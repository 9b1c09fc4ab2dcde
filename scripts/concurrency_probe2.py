"""Debug probe: which stage of the device permutation generator breaks when another process shares the GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from conftest import synth
from spatialcore_amd import _lib
role = sys.argv[1]
ctx = _lib.Context(0)
if role == "hammer":
    coords, X = synth(200000, 64, 3, dtype=np.float32, sparse_x=False)
    ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1 / 15); ctx.set_expression(X, np.arange(64))
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        ctx.moran_seeded(_lib.rng_state_words(np.random.default_rng(1)), 128, return_sims=False)
    print("hammer done", flush=True)
else:
    for n, P in [(30000, 130), (30000, 130), (70001, 100), (140001, 60), (30000, 130)]:
        for mode in (0, 1):
            ctx.set_permgen_mode(mode)
            w = _lib.rng_state_words(np.random.default_rng(4))
            got = ctx.generate_permutations(w, n, P, fetch=True)
            wh = _lib.rng_state_words(np.random.default_rng(4))
            want = _lib.perm_numpy_host(wh, n, P)
            bad = np.flatnonzero((got != want).any(axis=1))
            msg = f"{role} n={n} P={P} mode={mode}: wrong rows {bad.size} state_ok={bool((w == wh).all())}"
            if bad.size:
                r = bad[0]
                d = np.flatnonzero(got[r] != want[r])
                isperm = bool((np.sort(got[r]) == np.arange(n)).all())
                msg += f" first row {r} (rows {bad[:8].tolist()}) differing positions {d.size} first {d[:6].tolist()} last {d[-3:].tolist()} is_perm={isperm}"
            print(msg, flush=True)
