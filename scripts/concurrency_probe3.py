"""Debug probe: localise the generator failure under GPU sharing (J vs swaps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from conftest import synth
from spatialcore_amd import _lib
role = sys.argv[1]
ctx = _lib.Context(0)

def j_of_row(final):
    """Fisher-Yates targets j_i (i = n-1 .. 1) that turn the identity into `final`."""
    n = final.size
    a = np.arange(n); pos = np.arange(n)
    J = np.empty(n - 1, dtype=np.int64)
    for s, i in enumerate(range(n - 1, 0, -1)):
        v = final[i]; j = pos[v]; J[s] = j
        w = a[i]; a[i], a[j] = v, w; pos[v], pos[w] = i, j
    return J

if role == "hammer":
    coords, X = synth(int(sys.argv[3]), 64, 3, dtype=np.float32, sparse_x=False)
    n = coords.shape[0]
    ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1 / 15); ctx.set_expression(X, np.arange(64))
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        ctx.moran_seeded(_lib.rng_state_words(np.random.default_rng(1)), 128, return_sims=False)
    print("hammer done", flush=True)
else:
    reps = int(sys.argv[2])
    for rep in range(reps):
        for n, P in [(30000, 130), (70001, 100)]:
            w = _lib.rng_state_words(np.random.default_rng(4))
            got = ctx.generate_permutations(w, n, P, fetch=True)
            wh = _lib.rng_state_words(np.random.default_rng(4))
            want = _lib.perm_numpy_host(wh, n, P)
            bad = np.flatnonzero((got != want).any(axis=1))
            print(f"{role} rep {rep} n={n} P={P}: wrong rows {bad.size} state_ok={bool((w == wh).all())}", flush=True)
            if bad.size:
                M = n - 1
                st = ctx.debug_copy(5, 0, 8, np.uint64)
                print("   scan state", st.tolist(), "expected steps", P * M)
                for r in bad[:3]:
                    Jw = j_of_row(want[r])
                    Jd = ctx.debug_copy(0, int(r) * M * 4, M, np.int32).astype(np.int64)
                    dj = np.flatnonzero(Jw != Jd)
                    # apply the DEVICE J on the host: does the swap kernel's output equal that?
                    a = np.arange(n)
                    for s, i in enumerate(range(n - 1, 0, -1)):
                        j = min(int(Jd[s]), i); a[i], a[j] = a[j], a[i]
                    print(f"   row {r}: J differs at {dj.size} steps (first {dj[:5].tolist()}, last {dj[-3:].tolist()}); "
                          f"swap(deviceJ)==device row: {bool((a == got[r]).all())}; first J diffs dev {Jd[dj[:4]].tolist()} want {Jw[dj[:4]].tolist()}", flush=True)
                    if dj.size:
                        s0 = dj[0]
                        # is the device J a shifted copy of the expected one?
                        for sh in range(-4, 5):
                            if sh and 0 <= s0 + sh and s0 + 40 + abs(sh) < M and (Jd[s0:s0 + 40] == Jw[s0 + sh:s0 + 40 + sh]).all():
                                print("   device J[s] == expected J[s + %d] from the first difference on" % sh)
                break
