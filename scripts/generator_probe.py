"""The device generator alone (no scoring): time of sc_perm_generate at bench size, and its block statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialcore_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = _lib.Context(0)
for rep in range(3):
    w = _lib.rng_state_words(np.random.default_rng(0))
    ctx.sync(); t0 = time.perf_counter()
    ctx.generate_permutations(w, N, P)
    ctx.sync(); dt = time.perf_counter() - t0
    print(f"SC_TAIL_REM={os.environ.get('SC_TAIL_REM', 'default')}: {P} x {N}: {dt * 1e3:.1f} ms, stats {ctx.permgen_stats()}, state {w[:2]}", flush=True)
if os.environ.get("SC_PHI_PROFILE"):   # library built with EXTRA=-DPHI_PROFILE: clocks of the chain workgroup (thread 0) in computed blocks
    st = ctx.debug_copy(5, 0, 8, np.uint64)
    fp, hard = int(st[6]) & 0xffffffff, int(st[6]) >> 32
    rounds, trounds, ntail = int(st[7]) & 0xfffff, (int(st[7]) >> 20) & 0xfffff, int(st[7]) >> 40
    n_easy, n_hard = int(st[4]), int(st[5])
    print(f"computed blocks: {n_hard} ({ntail} entered with <= one block of steps left), {64 * hard / max(n_hard, 1):.0f} clocks each, of which "
          f"{64 * fp / max(n_hard, 1):.0f} in the fixed point; rounds: {trounds / max(ntail, 1):.1f} per tail block, "
          f"{rounds / max(n_hard - ntail, 1):.1f} per other block")
