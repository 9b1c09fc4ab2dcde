"""Correctness sweep of the device generator over permutation lengths and seeds: table rows and final generator state
against the host implementation of numpy's stream; no verification fallback allowed (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialcore_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
sizes = [65536, 65537, 100000, 131071, 131072, 131073, 150001, 262144, 333333, 524287, 524288, 777777, 1048576, 1500000, 2097153, 3000000, 4999999]
bad = 0
for n in sizes:
    for rep in range(2):
        seed = int(rng.integers(0, 2**31))
        P = int(max(3, min(400, 4e8 // n)))
        if rep: P = int(rng.integers(2, P + 1))
        wh = _lib.rng_state_words(np.random.default_rng(seed))
        want = _lib.perm_numpy_host(wh, n, P)
        before = ctx.permgen_stats()
        w = _lib.rng_state_words(np.random.default_rng(seed))
        t0 = time.perf_counter()
        got = ctx.generate_permutations(w, n, P, fetch=True)
        dt = time.perf_counter() - t0
        par, seq, fb = (a - b for a, b in zip(ctx.permgen_stats()[:3], before[:3]))
        ok = bool((w == wh).all()) and bool((got == want).all()) and (par, seq, fb) == ((1, 0, 0) if n >= 131072 else (0, 1, 0))
        bad += not ok
        print(f"n={n} P={P} seed={seed}: {'ok' if ok else 'MISMATCH'} (jobs {par}/{seq}, fallbacks {fb}) {dt * 1e3:.0f} ms", flush=True)
print("sweep:", "all ok" if bad == 0 else f"{bad} FAILED")
