"""The device generator alone (no scoring): time of sc_perm_generate at bench size, and its block statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialcore_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if os.environ.get("SC_LIB"):   # another build of the library (e.g. spatialcore_amd/libvar_phiprofile.so)
    _lib.LIB_PATH = os.environ["SC_LIB"]
ctx = _lib.Context(0)
for rep in range(3):
    w = _lib.rng_state_words(np.random.default_rng(0))
    ctx.sync(); t0 = time.perf_counter()
    ctx.generate_permutations(w, N, P)
    ctx.sync(); dt = time.perf_counter() - t0
    print(f"{P} x {N}: {dt * 1e3:.1f} ms, stats {ctx.permgen_stats()}, state {w[:2]}", flush=True)
st = ctx.debug_copy(5, 0, 8, np.uint64)
try:
    fc = ctx.debug_copy(8, 0, 16, np.uint64)   # the fresh-table control block: [seq | done, posts (b, S, t) x 2, diagnostics x 9]
    dg = [int(v) for v in fc[7:16]]
    print(f"fresh tables (last job): posts by the chain {int(fc[0]) & 0xffffffff}; helper rounds {dg[0]}, blocks attempted {dg[1]}, post -> start of work "
          f"{dg[2] / max(dg[0], 1) / 100:.1f} us; clean {dg[3]}, not clean {dg[4]}; post -> ready {dg[5] / max(dg[3], 1) / 100:.1f} us; chain lookups tried {dg[6]} "
          f"at {dg[7] / max(dg[6], 1) / 100:.1f} us after the post, of which a table for that block existed {dg[8]}")
except Exception as exc:
    print("fresh tables: n/a", exc)
if os.environ.get("SC_PHI_PROFILE"):   # library built with EXTRA=-DPHI_PROFILE
    prof = ctx.debug_copy(100, 0, 32, np.uint64)
    names = ["> 98304 steps left (band changes, window misses)", "49152 .. 98304", "24576 .. 49152", "12288 .. 24576", "<= 12288 (the permutation ends inside)"]
    jobs = 3 * P
    for k, nm in enumerate(names):
        cnt, clk, rounds = int(prof[4 * k]), int(prof[4 * k + 1]), int(prof[4 * k + 2])
        if cnt:
            print(f"computed blocks with {nm}: {cnt / jobs:.2f} per permutation, {clk / cnt:.0f} clocks and {rounds / cnt:.1f} rounds each, {clk / jobs:.0f} clocks per permutation")
    easy, hard = 64 * (int(st[6]) & 0xffffffff), 64 * (int(st[6]) >> 32)
    print(f"chain clocks of the last job: lookup phases (segments + slow paths) {easy / 1e6:.1f}M = {easy / max(int(st[4]), 1):.0f} per block resolved by lookup; "
          f"computed blocks {hard / 1e6:.1f}M = {hard / max(int(st[5]), 1):.0f} each; slow paths {int(st[7]) & 0xffffffff}; blocks by fresh table {int(st[7]) >> 32}")
    print(f"all {jobs} permutations: waiting for a unit's preparation {int(prof[20]) / jobs:.0f} clocks per permutation; slow paths (a segment's window "
          f"missed) {int(prof[22]) / jobs:.2f} per permutation, {int(prof[21]) / max(int(prof[22]), 1):.0f} clocks each; exposed table loads {int(prof[23]) / jobs:.2f} per permutation")
    nb = max(int(prof[25]), 1)
    print(f"fixed point of the computed blocks: first evaluation (guess + 16 draws) {int(prof[24]) / nb:.0f} clocks of thread 0; wavefronts that recompute "
          f"after round 1 / 2 / 3 / later: {int(prof[26]) / nb:.1f} / {int(prof[27]) / nb:.1f} / {int(prof[28]) / nb:.1f} / {int(prof[29]) / nb:.1f} of 16 per block")
    nr = max(int(prof[11]), 1)
    print(f"rounds of the computed blocks: {nr / jobs:.1f} per permutation; wavefront 0 (which seldom recomputes) works {int(prof[3]) / nr:.0f} clocks a round "
          f"and waits {int(prof[7]) / nr:.0f} at the round's barrier for the wavefronts that do")
else:
    print(f"last job: {int(st[4])} blocks by lookup in {int(st[6])} segment lookups, {int(st[7]) & 0xffffffff} segments block by block (window missed), {int(st[7]) >> 32} blocks by fresh table; {int(st[5])} computed")
