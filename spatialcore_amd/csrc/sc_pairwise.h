// numpy's pairwise float summation, shared by the float32-faithful paths (local Moran, Lee).  gfx950 only.
#pragma once

#include <stdint.h>

// numpy's pairwise recursion over `len` elements: a leaf is <= 128 elements, above that the split is len / 2
// rounded down to a multiple of 8.  Visits the leaves in order; at a leaf `leaf(start, len)` supplies the value.
template <typename T, typename Leaf>
__device__ T pw_walk(uint32_t n, Leaf leaf)
{
    uint32_t f_len[34], f_start[34];
    T f_left[34];
    unsigned char f_state[34];
    int sp = 0;
    T ret = (T)0;
    f_len[0] = n; f_start[0] = 0; f_state[0] = 0; sp = 1;
    while (sp > 0) {
        const int k = sp - 1;
        const uint32_t len = f_len[k];
        uint32_t n2 = len / 2; n2 -= n2 % 8;
        if (f_state[k] == 0) {
            if (len <= 128) { ret = leaf(f_start[k], len); --sp; continue; }
            f_state[k] = 1;
            f_len[sp] = n2; f_start[sp] = f_start[k]; f_state[sp] = 0; ++sp;
        } else if (f_state[k] == 1) {
            f_left[k] = ret;
            f_state[k] = 2;
            f_len[sp] = len - n2; f_start[sp] = f_start[k] + n2; f_state[sp] = 0; ++sp;
        } else {
            ret = f_left[k] + ret;
            --sp;
        }
    }
    return ret;
}


// one leaf of that recursion: numpy's unrolled block sum of val(0) .. val(len - 1), len <= 128 (8 strided
// accumulators, pairwise combine, then the tail), every operation rounded in T
template <typename T, typename Val>
__device__ __forceinline__ T pw_block(uint32_t len, Val val)
{
    if (len < 8) {
        T res = (T)(-0.0);
        for (uint32_t k = 0; k < len; ++k) res += val(k);
        return res;
    }
    T r0 = val(0), r1 = val(1), r2 = val(2), r3 = val(3), r4 = val(4), r5 = val(5), r6 = val(6), r7 = val(7);
    uint32_t k = 8;
    for (; k < len - (len % 8); k += 8) {
        r0 += val(k); r1 += val(k + 1); r2 += val(k + 2); r3 += val(k + 3);
        r4 += val(k + 4); r5 += val(k + 5); r6 += val(k + 6); r7 += val(k + 7);
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; k < len; ++k) res += val(k);
    return res;
}
