"""Debug probe: after a wrong generator job under GPU sharing, check the scan arrays for self-consistency on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from spatialcore_amd import _lib
T, D = 1024, 16
BLK = T * D
ctx = _lib.Context(0)

def mask_of(i):
    m = i.astype(np.uint64)
    for s in (1, 2, 4, 8, 16):
        m |= m >> np.uint64(s)
    return m

def analyse(n, P, tag):
    M = n - 1
    st = ctx.debug_copy(5, 0, 8, np.uint64)
    nb = int(st[1])
    raw = ctx.debug_copy(1, 0, nb * BLK, np.uint32).reshape(nb, D // 4, T, 4)          # [block][g][tau][4]
    u = raw.transpose(0, 2, 1, 3).reshape(nb, T, D).astype(np.uint64)                    # [block][tau][draw]
    bits = ctx.debug_copy(2, 0, nb * T, np.uint32).reshape(nb, T)
    enter = ctx.debug_copy(3, 0, nb * T, np.uint32).reshape(nb, T).astype(np.int64)
    sblk = ctx.debug_copy(4, 0, nb + 1, np.uint64).astype(np.int64)
    # raw stream itself against numpy
    bg = np.random.PCG64(4); rr = bg.random_raw(nb * BLK // 2)
    want_raw = np.stack([rr & np.uint64(0xffffffff), rr >> np.uint64(32)], 1).reshape(-1)
    lin = u.reshape(-1)
    print(f"[{tag}] raw stream wrong entries: {int((lin != want_raw).sum())} of {lin.size}")
    wr = np.flatnonzero(lin != want_raw)
    if wr.size:
        # linear draw index r -> (block, tau, s); 64-bit output m = r // 2
        blk, tau, sd = wr // BLK, (wr % BLK) // D, wr % D
        print("      wrong draws: blocks", np.unique(blk)[:20].tolist(), "n blocks", np.unique(blk).size)
        print("      tau range", int(tau.min()), int(tau.max()), "distinct tau", np.unique(tau).size, "draw-in-thread histogram", np.bincount(sd, minlength=D).tolist())
        for k in wr[:6]:
            got, exp = int(lin[k]), int(want_raw[k])
            where = np.flatnonzero(want_raw == np.uint64(got))
            print(f"      r={k} (block {k // BLK}, tau {(k % BLK) // D}, s {k % D}): got {got:#010x} want {exp:#010x}; got-value occurs at expected r={where[:3].tolist()}")
        # contiguous runs
        runs = np.split(wr, np.flatnonzero(np.diff(wr) != 1) + 1)
        print("      runs:", [(int(r[0]), len(r)) for r in runs[:12]], "n runs", len(runs))
    pc = np.zeros((nb, T), dtype=np.int64)
    for s in range(D):
        pc += (bits >> s) & 1
    excl = np.cumsum(pc, axis=1) - pc
    bad_enter = np.argwhere(enter != excl)
    print(f"[{tag}] enter != prefix(popcount(bits)): {len(bad_enter)} threads, first {bad_enter[:3].tolist()}")
    tot = pc.sum(axis=1)
    bad_s = np.flatnonzero(sblk[1:nb] != sblk[:nb - 1] + tot[:nb - 1])
    print(f"[{tag}] sblk chain breaks at blocks {bad_s[:5].tolist()} (of {nb}); sblk[0..3]={sblk[:4].tolist()}")
    # re-simulate every thread from its own entering state
    total = P * M
    S = sblk[:nb, None] + enter
    exp_bits = np.zeros((nb, T), dtype=np.uint32)
    Scur = S.copy()
    for s in range(D):
        i = M - (Scur % M)
        v = u[:, :, s] & mask_of(i)
        acc = (v <= i.astype(np.uint64)) & (Scur < total)
        exp_bits |= acc.astype(np.uint32) << s
        Scur = Scur + acc
    wrong = np.argwhere(exp_bits != bits)
    print(f"[{tag}] threads whose bits are not the result for their own entering state: {len(wrong)}, first {wrong[:5].tolist()}")
    if len(wrong):
        b, t = wrong[0]
        print(f"      block {b} thread {t}: dev bits {bits[b, t]:#x} exp {exp_bits[b, t]:#x} enter {enter[b, t]} sblk {sblk[b]}")
    # expand on the host and compare with the device J
    J = ctx.debug_copy(0, 0, total, np.int32)
    expJ = np.full(total, -1, dtype=np.int64)
    Scur = S.copy()
    for s in range(D):
        i = M - (Scur % M)
        v = u[:, :, s] & mask_of(i)
        acc = ((bits >> s) & 1).astype(bool) & (Scur < total)
        expJ[Scur[acc]] = v[acc].astype(np.int64)
        Scur = Scur + acc
    dj = np.flatnonzero(expJ != J)
    print(f"[{tag}] J != expand(device arrays): {dj.size} steps, first {dj[:5].tolist()}; unwritten steps {int((expJ < 0).sum())}")

role = sys.argv[1]
if role == "hammer":
    from conftest import synth
    coords, X = synth(int(sys.argv[3]), 64, 3, dtype=np.float32, sparse_x=False)
    ctx.knn(coords, 15, fetch=False); ctx.graph_from_knn(1 / 15); ctx.set_expression(X, np.arange(64))
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        ctx.moran_seeded(_lib.rng_state_words(np.random.default_rng(1)), 128, return_sims=False)
    print("hammer done", flush=True)
else:
    n, P = 30000, 130
    wh = _lib.rng_state_words(np.random.default_rng(4))
    want = _lib.perm_numpy_host(wh, n, P)
    done = 0
    for rep in range(int(sys.argv[2])):
        w = _lib.rng_state_words(np.random.default_rng(4))
        got = ctx.generate_permutations(w, n, P, fetch=True)
        bad = np.flatnonzero((got != want).any(axis=1))
        print(f"{role} rep {rep}: wrong rows {bad.size} {bad[:6].tolist()} state_ok={bool((w == wh).all())}", flush=True)
        if bad.size or rep == 0:
            analyse(n, P, "BAD" if bad.size else "good")
            done += bad.size > 0
        if done >= 2:
            break
