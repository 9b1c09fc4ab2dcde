"""Rounds of the in-block fixed point (block_fixed_point, sc_permgen.hip) restated on the CPU, for blocks in which a
permutation of 10^6 cells ends: where the exact prefix stands after each round, and what two accelerations would buy.

    python scripts/fixed_point_rounds_sim.py trace [seed]           the front (first thread with a wrong entering count) by round
    python scripts/fixed_point_rounds_sim.py compare LO HI [N]      mean rounds for blocks entered with LO <= steps left < HI:
                                                                    plain / Newton-predicted entering counts / the last TAIL_I
                                                                    steps solved exactly once the front reaches them

r04 findings: the plain iteration needs ~15 rounds for such a block (the kernel's counter: 16.1), half of
them for the last ~20 threads (thresholds of a few hundred: the front moves 2-3 threads a round); predicting the entering
counts with the linearised sensitivity (an affine scan) saves 2 rounds; solving the last 512 / 1024 / 2048 steps exactly
saves 5 / 7 / 8.5 -- built (-DPHI_TAIL=true), and slower on the device: the solver is one wavefront."""
import os
import sys
import numpy as np

D, TH = 16, 1024
M, TOP = 999_999, (1 << 20) - 1


def mask_of(i):
    return (1 << int(i).bit_length()) - 1


def mask_of_v(i):
    return (1 << (np.floor(np.log2(i)).astype(np.int64) + 1)) - 1


def expected_steps(rem, q):
    i, acc = float(rem), 0.0
    for _ in range(64):
        if q <= 0:
            break
        ii = int(i)
        if ii == 0:
            i, ii = float(M), M
        m = mask_of(ii)
        top, lo = m + 1.0, (m >> 1) + 1
        need = top * np.log((i + 1.0) / lo)
        if need <= q:
            q -= need
            acc += i - lo + 1.0
            i = lo - 1.0
        else:
            inew = (i + 1.0) * np.exp(-q / top) - 1.0
            acc += i - inew
            q = 0
    return int(acc + 0.5)


def scan_threads(u, c_in, rem_block):
    c = c_in.copy()
    rem = np.full(TH, rem_block, dtype=np.int64)
    wrap = c >= rem
    c = np.where(wrap, (c - rem) % M, c)
    rem = np.where(wrap, M, rem)
    i = rem - c
    i0 = i.copy()
    mask = mask_of_v(i)
    mask0 = mask.copy()
    fast = i0 > (mask >> 1) + D
    cnt = np.zeros(TH, dtype=np.int64)
    gap = np.full(TH, 1 << 40, dtype=np.int64)
    lam = np.zeros(TH)
    for s in range(D):
        v = u[:, s] & mask
        lam += 1.0 / (mask + 1.0)
        d = i - v
        acc = d >= 0
        gap = np.minimum(gap, np.where(acc, d, -d - 1))
        cnt += acc
        i = i - acc
        w = i == 0
        half = mask >> 1
        mask = np.where(w, TOP, np.where(i <= half, half, mask))
        i = np.where(w, M, i)
    return [cnt, fast, np.where(fast, gap, 0), i0, mask0, lam]


def still_valid(res, c_used, c_new):
    cnt, fast, gap, i0, mask0, _ = res
    delta = c_new - c_used
    i0n = i0 - delta
    ok = fast & (i0n <= M) & (i0n <= mask0) & (i0n > (mask0 >> 1) + D) & (np.abs(delta) <= gap)
    return (delta == 0) | ok


def truth(u, rem_block):
    i, mask, cnt = rem_block, mask_of(rem_block), np.zeros(TH, dtype=np.int64)
    for q, x in enumerate(u.reshape(-1)):
        if (int(x) & mask) <= i:
            cnt[q // D] += 1
            i -= 1
            if i == 0:
                i, mask = M, TOP
            elif i <= (mask >> 1):
                mask >>= 1
    return cnt


def run(u, rem_block, mode="plain", tail_i=1024, trace=None):
    guess = np.array([expected_steps(rem_block, t * D) for t in range(TH)], dtype=np.int64)
    res = scan_threads(u, guess, rem_block)
    c_used, c_lin = guess.copy(), guess.copy()
    pinned = np.zeros(TH, bool)
    tail_open = mode == "tail"
    rounds = 0
    while True:
        rounds += 1
        cnt = res[0]
        E = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        valid = still_valid(res, c_used, E) | (pinned & (c_used == E))
        if trace is not None:
            wrong = np.nonzero(E != trace)[0]
            print("round %2d: front at thread %4d, %4d wrong entering counts, largest error %d" %
                  (rounds, wrong[0] if wrong.size else TH, wrong.size, np.abs(E - trace).max()))
        if valid.all():
            return rounds, cnt
        target, stale = E, ~valid
        if mode == "newton":
            lam = 1.0 - np.exp(-res[5])
            a, N = 0.0, np.zeros(TH)
            for t in range(TH):
                N[t] = a
                a = a + cnt[t] - lam[t] * (a - c_lin[t])
            target = np.maximum(0, np.floor(N + 0.5)).astype(np.int64)
            stale = ~still_valid(res, c_used, target)
            if not stale.any():
                target, stale = E, ~valid
        if tail_open:
            s = int(np.nonzero(stale)[0][0])
            if E[s] < rem_block and rem_block - E[s] <= tail_i:
                i = rem_block - E[s]
                mask, cc, wrapped = mask_of(i), E[s], False
                for t in range(s, min(TH, s + tail_i // 4 + 32)):
                    k = 0
                    for x in u[t]:
                        if (int(x) & mask) <= i:
                            k += 1
                            i -= 1
                            if i == 0:
                                i, mask, wrapped = M, TOP, True
                            elif i <= (mask >> 1):
                                mask >>= 1
                    res[0][t], res[1][t], res[2][t], c_used[t], pinned[t], stale[t] = k, False, 0, cc, True, False
                    cc += k
                    if wrapped:
                        break
                tail_open = False
        new = scan_threads(u, target, rem_block)
        for k in range(6):
            res[k] = np.where(stale, new[k], res[k])
        c_used = np.where(stale, target, c_used)
        c_lin = target.copy()
        if rounds > 300:
            return -1, cnt


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "compare"
    if what == "trace":
        rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
        rem = int(rng.integers(2000, 12288))
        u = rng.integers(0, 1 << 32, size=(TH, D), dtype=np.int64)
        t = truth(u, rem)
        print("block entered with %d steps left; the permutation ends in thread %d" % (rem, int(np.searchsorted(np.cumsum(t), rem))))
        run(u, rem, trace=np.concatenate([[0], np.cumsum(t)[:-1]]))
    else:
        lo, hi = int(sys.argv[2]), int(sys.argv[3])
        n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
        rng = np.random.default_rng(0)
        out = {"plain": [], "newton": [], "tail 512": [], "tail 1024": [], "tail 2048": []}
        for _ in range(n):
            rem = int(rng.integers(lo, hi))
            u = rng.integers(0, 1 << 32, size=(TH, D), dtype=np.int64)
            ref = run(u, rem)
            out["plain"].append(ref[0])
            for name, kw in (("newton", dict(mode="newton")), ("tail 512", dict(mode="tail", tail_i=512)),
                             ("tail 1024", dict(mode="tail", tail_i=1024)), ("tail 2048", dict(mode="tail", tail_i=2048))):
                got = run(u, rem, **kw)
                assert np.array_equal(got[1], ref[1]), name
                out[name].append(got[0])
        print("blocks entered with %d .. %d steps left, %d of them: " % (lo, hi, n) +
              "; ".join("%s %.1f rounds" % (k, np.mean(v)) for k, v in out.items()))
