"""The rule of k_apply_swaps_full (sc_permgen.hip), restated on the CPU: whole rounds of T Fisher-Yates steps, hazards
resolved through the smallest / largest step index per partner slot, a round cut only at the first MIDDLE step of a slot.

    python scripts/swap_rounds_sim.py check [seed]     rounds form == sequential form, both directions, small n / T (many conflicts)
    python scripts/swap_rounds_sim.py rounds [n]       rounds per permutation of n cells for T = 512 / 1024 / 2048

tests/test_cpu_properties.py runs `check` with a few hundred cases."""
import sys
import numpy as np

NONE = -1


def sequential(n, J, asc):
    """J[M - i] in [0, i] is step i's partner; descending: the shuffle, ascending: its inverse table."""
    A = np.arange(n)
    M = n - 1
    for i in (range(1, n) if asc else range(n - 1, 0, -1)):
        j = J[M - i]
        A[i], A[j] = A[j], A[i]
    return A


def by_rounds(n, J, asc, T):
    M = n - 1
    A = np.full(n, -12345)          # ascending: nothing but slot 0 is initialised (an own slot's value is its index)
    if asc:
        A[0] = 0
    else:
        A = np.arange(n)
    i_cur = 1 if asc else n - 1
    rounds = 0
    while (i_cur <= n - 1) if asc else (i_cur >= 1):
        rounds += 1
        nvalid = min(n - i_cur if asc else i_cur, T)
        I = [(i_cur + l if asc else i_cur - l) for l in range(nvalid)]
        Jv = [min(J[M - i], i) for i in I]
        a_i = [A[i] for i in I]
        a_j = [A[j] for j in Jv]
        keys = {}
        for l, j in enumerate(Jv):
            mn, mx = keys.get(j, (10 ** 9, -1))
            keys[j] = (min(mn, l), max(mx, l))
        count = nvalid
        for l, j in enumerate(Jv):
            mn, mx = keys[j]
            if mn < l < mx:
                count = min(count, l)

        def last_lt(x, k):
            if x not in keys:
                return NONE
            mn, mx = keys[x]
            return mx if mx < k else (mn if mn < k else NONE)

        def last_any(x):
            return last_lt(x, count)

        val = [None] * count
        ptr = [NONE] * count
        p2s = [NONE] * count
        for l in range(count):
            i, j = I[l], Jv[l]
            if not asc:
                p1 = last_lt(i, l)
                p2s[l] = p1 if j == i else last_lt(j, l)
                if p1 == NONE:
                    val[l] = a_i[l]
                else:
                    ptr[l] = p1
            elif j == i:
                val[l] = i
            else:
                a_p, t = last_lt(j, l), j - i_cur
                own = 0 <= t < l
                if a_p == NONE and not own:
                    val[l] = a_j[l]
                elif a_p != NONE and (not own or a_p >= t):
                    val[l] = i_cur + a_p
                else:
                    ptr[l] = t
        while True:
            pending, upd = False, []
            for l in range(count):
                if ptr[l] != NONE:
                    q = ptr[l]
                    if ptr[q] == NONE:
                        upd.append((l, val[q]))
                    else:
                        pending = True
            for l, v in upd:
                val[l], ptr[l] = v, NONE
            if not pending:
                break
        for l in range(count):
            i, j = I[l], Jv[l]
            if not asc:
                A[i] = a_j[l] if p2s[l] == NONE else val[p2s[l]]
                if j != i and j <= i_cur - count and l == last_any(j):
                    A[j] = val[l]
            else:
                b = last_any(i)
                if not (b != NONE and b > l):
                    A[i] = val[l]
                if j != i and l == last_any(j):
                    A[j] = i
        i_cur += count if asc else -count
    return A, rounds


def check(seed=0, trials=300):
    rng = np.random.default_rng(seed)
    for _ in range(trials):
        n = int(rng.integers(2, 400))
        T = int(rng.choice([4, 8, 16, 64, 256]))
        J = np.array([rng.integers(0, i + 1) for i in range(n - 1, 0, -1)])
        for asc in (False, True):
            got, _ = by_rounds(n, J, asc, T)
            if not np.array_equal(sequential(n, J, asc), got):
                return "mismatch: n=%d T=%d ascending=%s" % (n, T, asc)
    return None


def rounds_per_permutation(n):
    rng = np.random.default_rng(0)
    out = {}
    for T in (512, 1024, 2048):
        i_cur, cut_first, cut_middle = n - 1, 0, 0
        i2 = n - 1
        while i_cur >= 1:          # this kernel: cut at the first middle step
            nv = min(T, i_cur)
            I = i_cur - np.arange(nv)
            Jr = (rng.random(nv) * (I + 1)).astype(np.int64)
            order = np.lexsort((np.arange(nv), Jr))
            Js = Jr[order]
            prev = np.r_[False, Js[1:] == Js[:-1]]
            nxt = np.r_[Js[1:] == Js[:-1], False]
            mid = order[prev & nxt]
            i_cur -= min(nv, int(mid.min())) if mid.size else nv
            cut_middle += 1
        while i2 >= 1:             # the r03 kernel: cut at the first step that shares a slot with an earlier one
            nv = min(T, i2)
            I = i2 - np.arange(nv)
            Jr = (rng.random(nv) * (I + 1)).astype(np.int64)
            order = np.lexsort((np.arange(nv), Jr))
            Js = Jr[order]
            dup = order[np.r_[False, Js[1:] == Js[:-1]]]
            hit = np.nonzero((Jr > i2 - nv) & (Jr != I))[0]       # a partner that is a later step's own slot ends the round there
            first = nv
            if dup.size:
                first = min(first, int(dup.min()))
            if hit.size:
                first = min(first, int((i2 - Jr[hit]).min()))
            i2 -= max(first, 1)
            cut_first += 1
        out[T] = (cut_first, cut_middle)
    return out


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        bad = check(int(sys.argv[2]) if len(sys.argv) > 2 else 0, 3000)
        print(bad or "ok")
        sys.exit(1 if bad else 0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    for T, (a, b) in rounds_per_permutation(n).items():
        print("T = %4d: %5d rounds cut at the first shared slot, %5d cut at the first middle step" % (T, a, b))
