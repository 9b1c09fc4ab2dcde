"""A/B of the scoring-kernel variants (SC_MORAN_VARIANT=0|1|2) on one box: launch time + a checksum of the sims."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialcore_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
P = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rng = np.random.default_rng(42)
coords = rng.uniform(0, np.sqrt(N) * 10, (N, 2))
X = rng.poisson(1.0, (N, G)).astype(np.float32)
ctx = _lib.Context(0)
ctx.set_moran_source_bits(int(os.environ.get("SC_BITS", 16)))
ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1.0 / 15)
ctx.set_expression(X, np.arange(G))
w = _lib.rng_state_words(np.random.default_rng(0))
ctx.generate_permutations(w, N, P)
out = ctx.moran(P)
for rep in range(3):
    ctx.reset_timers()
    out = ctx.moran(P)
    ms, cnt = ctx.kernel_time(_lib.K_MORAN_PERM)
    print(f"variant {os.environ.get('SC_MORAN_VARIANT', 'default')} N={N} G={G} P={P}: {ms / cnt:.3f} ms per launch ({cnt} launches), "
          f"source bits {ctx.moran_source_bits()}, gathered rows {P * N * 128 * 1e-9 / (ms / cnt * 1e-3) / 1e3:.2f} TB/s; sims sha {hashlib.sha1(out['sims'].tobytes()).hexdigest()[:12]}", flush=True)
