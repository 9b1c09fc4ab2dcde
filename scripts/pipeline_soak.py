"""Soak: is sc_moran_seeded (generator pipelined with scoring on prioritised / CU-masked streams) reproducible at the
bench size?  Each repetition is compared with the two-step path fed with the HOST generator's table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialcore_amd import _lib
N, G, P, reps = 1_000_000, int(sys.argv[1]), 1000, int(sys.argv[2])
rng = np.random.default_rng(42)
coords = rng.uniform(0, np.sqrt(N) * 10, (N, 2))
X = rng.poisson(1.0, (N, G)).astype(np.float32)
ctx = _lib.Context(0)
ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1.0 / 15)
ctx.set_expression(X, np.arange(G))
wh = _lib.rng_state_words(np.random.default_rng(0))
table = _lib.perm_numpy_host(wh, N, P)
ctx.set_permutations(table)
ref = ctx.moran(P)
print("reference (host table, two-step) done", flush=True)
T, D = 1024, 16
for rep in range(reps):
    ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1.0 / 15)
    w = _lib.rng_state_words(np.random.default_rng(0))
    out = ctx.moran_seeded(w, P)
    badp = np.flatnonzero((out["sims"] != ref["sims"]).any(axis=1))
    print(f"rep {rep}: state_ok={bool((w == wh).all())} permutations with differing sims: {badp.size} (first {badp[:5].tolist()}) "
          f"count_ge differs for {int((out['count_ge'] != ref['count_ge']).sum())} genes; permgen {ctx.permgen_stats()} {ctx.permgen_note()}", flush=True)
    if badp.size:
        st = ctx.debug_copy(5, 0, 8, np.uint64)
        nb = int(st[1])
        bg = np.random.PCG64(0)
        bad_total = 0
        for b0 in range(0, nb, 4096):
            b1 = min(nb, b0 + 4096)
            raw = ctx.debug_copy(1, b0 * T * D * 4, (b1 - b0) * T * D, np.uint32).reshape(b1 - b0, D // 4, T, 4)
            lin = raw.transpose(0, 2, 1, 3).reshape(-1)
            rr = bg.random_raw((b1 - b0) * T * D // 2)
            want = np.stack([rr & np.uint64(0xffffffff), rr >> np.uint64(32)], 1).reshape(-1).astype(np.uint32)
            wrong = np.flatnonzero(lin != want)
            if wrong.size:
                bad_total += wrong.size
                k = wrong[0]
                print(f"   raw stream wrong: {wrong.size} draws in blocks [{b0},{b1}); first at block {b0 + k // (T * D)} tau {(k % (T * D)) // D} s {k % D}", flush=True)
        print(f"   raw stream wrong draws in total: {bad_total} of {nb * T * D}", flush=True)
