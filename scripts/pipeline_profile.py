"""Development: the chain's clock counters (a -DPHI_PROFILE build, scripts/build_variant.sh) INSIDE the Moran pipeline at bench
size -- where the chain workgroup's time goes while the scoring kernel runs beside it.
usage: SC_LIB=spatialcore_amd/libvar_phiprofile.so python scripts/pipeline_profile.py [genes] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialcore_amd import _lib
if os.environ.get("SC_LIB"):
    _lib.LIB_PATH = os.environ["SC_LIB"]
N, G, P = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 500, 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(42)
coords = rng.uniform(0, np.sqrt(N) * 10, (N, 2))
X = rng.poisson(1.0, (N, G)).astype(np.float32)
ctx = _lib.Context(0)
ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1.0 / 15)
ctx.set_expression(X, np.arange(G))
for rep in range(reps + 1):
    if rep == 1:
        ctx.debug_copy(100, 1, 32, np.uint64)   # reset the counters after the warm-up repetition
    w = _lib.rng_state_words(np.random.default_rng(0))
    ctx.sync(); t0 = time.perf_counter()
    ctx.moran_seeded(w, P, return_sims=False)
    ctx.sync(); dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt * 1e3:.1f} ms, permgen {ctx.permgen_stats()}", flush=True)
prof = ctx.debug_copy(100, 0, 32, np.uint64)
jobs = reps * P
names = ["> 98304", "49152 .. 98304", "24576 .. 49152", "12288 .. 24576", "<= 12288"]
tot = 0
for k, nm in enumerate(names):
    cnt, clk, rounds = int(prof[4 * k]), int(prof[4 * k + 1]), int(prof[4 * k + 2])
    tot += clk
    if cnt:
        print(f"computed blocks with {nm} steps left: {cnt / jobs:.2f} per permutation, {clk / cnt:.0f} clocks and {rounds / cnt:.1f} rounds each, {clk / jobs:.0f} clocks per permutation")
print(f"units that waited > 20 us for their preparation: {int(prof[30])} of ~{jobs // 6} ({int(prof[19]) / jobs:.0f} clocks per permutation of the wait); longest wait {int(prof[31]) / 2.1e3:.0f} us")
print(f"computed blocks in all {tot / jobs:.0f} clocks per permutation; waiting for a unit's preparation {int(prof[20]) / jobs:.0f}; "
      f"slow paths {int(prof[22]) / jobs:.2f} per permutation x {int(prof[21]) / max(int(prof[22]), 1):.0f} clocks; exposed table loads {int(prof[23]) / jobs:.2f} per permutation")
