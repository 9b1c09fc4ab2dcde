"""GPU box: time sc_perm_generate alone (for rocprofv3 --kernel-trace --stats).
usage: permgen_probe.py [n] [n_perm] [mode]   mode 0 = automatic (block-parallel scan), 1 = sequential scan"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from spatialcore_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 100
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = _lib.Context(0)
ctx.set_permgen_mode(MODE)
for rep in range(2):
    w = _lib.rng_state_words(np.random.default_rng(0))
    ctx.reset_timers()
    t = time.time(); ctx.generate_permutations(w, N, P); dt = (time.time() - t) * 1e3
    print(f"mode {MODE} perms {P} x {N}: {dt:.1f} ms  scan {ctx.kernel_time(_lib.K_PERM_SCAN)[0]:.1f} ms  swaps {ctx.kernel_time(_lib.K_PERM_SWAP)[0]:.1f} ms  state {w[:2]} stats {ctx.permgen_stats()}", flush=True)
