"""Development: time sc_perm_generate with another build of the library.
usage: python scripts/permgen_variant.py n n_perm mode [lib.so]"""
import sys, time, re
import numpy as np
sys.path.insert(0, ".")
from spatialcore_amd import _lib
if len(sys.argv) > 4: _lib.LIB_PATH = sys.argv[4]
N, P, MODE = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.Context(0); ctx.set_permgen_mode(MODE)
for rep in range(2):
    w = _lib.rng_state_words(np.random.default_rng(0)); ctx.reset_timers()
    t = time.time(); ctx.generate_permutations(w, N, P); dt = (time.time() - t) * 1e3
    print(f"{_lib.LIB_PATH} mode {MODE} perms {P} x {N}: {dt:.1f} ms  scan {ctx.kernel_time(_lib.K_PERM_SCAN)[0]:.1f} ms stats {ctx.permgen_stats()}", flush=True)
