"""GPU box: soak test of the block-parallel generator against the host generator (numpy-exact, tested on CPU).
usage: python scripts/permgen_soak.py [rounds]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from spatialcore_amd import _lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = _lib.Context(0)
sizes = [131072, 131073, 200001, 262143, 262144, 262145, 524287, 524288, 524289, 777777, 1048575, 1048576,
         1048577, 1500000, 2097151, 2097152, 2097153, 3000001, 4194305]
rng = np.random.default_rng(2026)
bad = 0
t0 = time.time()
for rnd in range(rounds):
    for n in sizes:
        seed = int(rng.integers(0, 2**31))
        P = int(rng.integers(1, 9))
        w = _lib.rng_state_words(np.random.default_rng(seed))
        wh = w.copy()
        # burn a random number of 32-bit draws first so that has_uint32 / stream offsets vary
        g = np.random.default_rng(seed)
        for _ in range(int(rng.integers(0, 4))):
            g.integers(0, 2**32, dtype=np.uint32)
        w = _lib.rng_state_words(g); wh = w.copy()
        before = ctx.permgen_stats()
        got = ctx.generate_permutations(w, n, P, fetch=True)
        got2 = ctx.generate_permutations(w, n, 2, fetch=True)       # continue the stream
        st = tuple(a - b for a, b in zip(ctx.permgen_stats(), before))
        want = _lib.perm_numpy_host(wh, n, P)
        want2 = _lib.perm_numpy_host(wh, n, 2)
        ok = np.array_equal(got, want) and np.array_equal(got2, want2) and np.array_equal(w, wh) and st[2] == 0 and st[0] == 2
        bad += 0 if ok else 1
        print(f"n={n} P={P} seed={seed} h={int(wh[4])} stats={st} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{'ALL OK' if bad == 0 else str(bad) + ' FAILURES'} in {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
