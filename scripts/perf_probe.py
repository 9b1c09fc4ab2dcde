"""Ad-hoc scale probe of the hot kernels (GPU box).  Not part of the product or the tests."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from spatialcore_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
P = int(sys.argv[3]) if len(sys.argv) > 3 else 64
K = 15
rng = np.random.default_rng(42)
L = np.sqrt(N) * 10
coords = rng.uniform(0, L, (N, 2))
X = rng.poisson(1.0, (N, G)).astype(np.float32)
ctx = _lib.Context(0)
t = time.time(); ctx.knn(coords, K, fetch=False); ctx.sync(); print(f"knn N={N} k={K}: {time.time()-t:.3f}s (incl. upload, sort)", flush=True)
t = time.time(); ctx.knn(coords, K, fetch=False); ctx.sync(); print(f"knn again: {time.time()-t:.3f}s  kernel {ctx.kernel_time(_lib.K_KNN)}", flush=True)
ctx.graph_from_knn(1.0 / K)
t = time.time(); ctx.set_expression(X, np.arange(G)); print(f"expr upload: {time.time()-t:.3f}s", flush=True)
for rep in range(2):
    w = _lib.rng_state_words(np.random.default_rng(0))
    ctx.reset_timers()
    t = time.time(); ctx.generate_permutations(w, N, P); dt = time.time() - t
    print(f"device numpy-exact perms ({P} x {N}): wall {dt*1e3:.1f} ms, event {ctx.kernel_time(_lib.K_PERMGEN)[0]:.1f} ms", flush=True)
for rep in range(3):
    ctx.reset_timers()
    t = time.time(); out = ctx.moran(P, return_sims=False); dt = time.time() - t
    ms, cnt = ctx.kernel_time(_lib.K_MORAN_PERM)
    lag_ms, _ = ctx.kernel_time(_lib.K_LAG)
    tiles = (G + 15) // 16
    alg = P * tiles * 16 * N * 16.0
    print(f"moran N={N} G={G} P={P}: wall {dt*1e3:.1f} ms; perm kernel {ms:.2f} ms over {cnt} launches "
          f"-> {alg/ms/1e9:.2f} TB/s algorithmic; lag {lag_ms:.2f} ms; "
          f"extrapolated P=1000 G=500: {ms/ (P*tiles) * 1000*32/1e3:.2f} s", flush=True)
print("mem GiB", ctx.device_mem() / 2**30)
