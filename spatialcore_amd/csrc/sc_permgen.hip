// A4 on the device: the numpy-exact permutation table, generated in parallel.  gfx950 only.
//
// numpy's stream (see sc_perm.hip header) is sequential: every Fisher-Yates step consumes a
// data-dependent number of 32-bit draws (masked rejection), so the position of every later draw
// depends on all earlier rejections.  Exactness therefore needs the TRUE position of every step;
// two offset rejection scans over the same draws never re-synchronise (their time lag is
// conserved), so speculative chunking cannot be made exact.  The generator below is exact by
// construction and still parallel:
//
//  A0  raw stream   PCG64 is an LCG, so output m is a pure function of m (jump-ahead in O(log m)
//                   128-bit multiplies).  The whole raw 32-bit stream is produced in one massively
//                   parallel kernel (each lane strides by 64 outputs with the constant A^64, C_64).
//  A1  rejection    A 1024-thread workgroup resolves the raw stream in blocks of 16384 draws.  Each
//                   thread simulates its 16 consecutive draws sequentially (exact semantics) from a
//                   guessed number of accepts in front of it; a workgroup prefix sum of the accept
//                   counts gives new entering counts; this repeats until no entering count changes.
//                   A thread whose entering count is right produces the right count, so the correct
//                   prefix grows every round and the fixed point IS the sequential result (typically
//                   2-3 rounds, the guess being the expected acceptance rate).  Output: J[step] =
//                   the accepted value j of every Fisher-Yates step, and the exact stream position.
//  B   swaps        permutations are independent given J: one wavefront per permutation applies
//                   `swap(a[i], a[j_i])` for 64 consecutive steps at a time; the longest prefix of
//                   the 64 steps that touches pairwise-distinct array slots is applied in parallel
//                   (those swaps commute), the rest is retried, so the result equals the sequential
//                   shuffle bit for bit.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sc_ctx.h"

typedef unsigned __int128 u128;

#define PCG_MULT_HI 0x2360ed051fc65da4ULL
#define PCG_MULT_LO 0x4385df649fccf645ULL

struct Affine {  // x -> mult * x + plus  (mod 2^128)
    u128 mult, plus;
};

__host__ __device__ static inline u128 pcg_mult() { return ((u128)PCG_MULT_HI << 64) | PCG_MULT_LO; }

// the LCG step composed `delta` times
__host__ __device__ static inline Affine lcg_pow(u128 inc, uint64_t delta)
{
    u128 acc_m = 1, acc_p = 0, cur_m = pcg_mult(), cur_p = inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_m *= cur_m;
            acc_p = acc_p * cur_m + cur_p;
        }
        cur_p = (cur_m + 1) * cur_p;
        cur_m *= cur_m;
        delta >>= 1;
    }
    Affine a;
    a.mult = acc_m;
    a.plus = acc_p;
    return a;
}

__host__ __device__ static inline uint64_t xsl_rr(u128 s)
{
    uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    uint64_t x = hi ^ lo;
    unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64 - r) & 63));
}

// The same rotation from 32-bit funnel shifts only (v_alignbit_b32) -- the form the DEVICE uses.
// r02 finding: with the plain form above, the compiler emits v_lshlrev_b64 / v_lshrrev_b64 with a per-lane shift
// amount, and k_raw_stream then wrote WRONG outputs for whole wavefronts (the left-shifted half of the rotation)
// whenever kernels of other hardware queues ran on the chip at the same time -- never when it ran alone:
// ~50 wavefronts per 1M x 1000 job inside the Moran pipeline, i.e. every r01 pipeline run at bench size drew some
// non-numpy permutations; also when a second process used the GPU.  Same job, same box, A/B by kernel variant
// (scripts/pipeline_soak.py, 3 repetitions each): 64-bit shifts 38k-139k wrong draws per job; with an added
// s_waitcnt after every store 1.5-2.1M; this form 0, and every statistic bit-equal to the host generator's.
// Evidence and decoding of the wrong words: profiles/r02_gpu_sharing_raw_stream_corruption.txt.
__device__ static inline uint64_t xsl_rr32(u128 s)
{
    const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    const uint64_t x = hi ^ lo;
    const uint32_t r = (uint32_t)(hi >> 58);
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    if (r & 32) { const uint32_t t = xl; xl = xh; xh = t; }          // rotate by 32: swap the halves
    const uint32_t k = r & 31;
    const uint32_t ol = __builtin_amdgcn_alignbit(xh, xl, k);         // ({xh, xl} >> k) low word
    const uint32_t oh = __builtin_amdgcn_alignbit(xl, xh, k);
    return ((uint64_t)oh << 32) | ol;
}

// ------------------------------------------------------------------------------------------------
// A0: the raw 32-bit stream, stored in the layout the scan reads.
//
// Stream draw r (r = 0: low half of 64-bit output 0, r = 1: its high half, ...) lives at
//   phys(r) = block(r) * SCAN_BLOCK + g * (4 * SCAN_THREADS) + tau * 4 + (r & 3),
//   tau = (r % SCAN_BLOCK) / SCAN_D, g = ((r % SCAN_D) / 4)
// i.e. inside every SCAN_BLOCK-draw block, scan thread tau's draws [D tau, D tau + D) are stored as D/4
// groups of 4, group g at block + g*4*SCAN_THREADS + 4*tau: the scan's g-th 16-byte load is contiguous
// across the workgroup's threads.  One generator thread produces one such 16-byte group (2 consecutive 64-bit
// outputs) per block for RAW_BLOCKS consecutive blocks, stepping its LCG state by the constant
// jump A^16384 between blocks.
// ------------------------------------------------------------------------------------------------

#ifndef SCAN_THREADS
#define SCAN_THREADS 1024
#endif
#ifndef SCAN_D
#define SCAN_D 16  // draws per thread and round (r01 sweep at 1M cells, sequential / block-parallel scan of 300
                   // permutations: 8 -> 145 / 83 ms, 12 -> 119 / 72, 16 -> 107 / 58, 20 -> 103 / 59, 24 -> 114 / 63, 32 -> 154 / 84)
#endif
#define SCAN_BLOCK (SCAN_THREADS * SCAN_D)
#define SCAN_GROUPS (SCAN_D / 4)
#if SCAN_D <= 32
typedef uint32_t bits_t;
#else
typedef uint64_t bits_t;
#endif
#define RAW_BLOCKS 8

__global__ __launch_bounds__(256) void k_raw_stream(uint64_t st_hi, uint64_t st_lo, uint64_t inc_hi,
                                                    uint64_t inc_lo, uint64_t n_blocks, uint64_t jm_hi,
                                                    uint64_t jm_lo, uint64_t jp_hi, uint64_t jp_lo,
                                                    uint32_t *__restrict__ raw)
{
    const u128 state0 = ((u128)st_hi << 64) | st_lo, inc = ((u128)inc_hi << 64) | inc_lo;
    const u128 jm = ((u128)jm_hi << 64) | jm_lo, jp = ((u128)jp_hi << 64) | jp_lo;  // LCG^16384
    const u128 mult = pcg_mult();
    // thread = (block group, g, tau): consecutive threads write consecutive 16-byte groups
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tau = (uint32_t)(t % SCAN_THREADS), g = (uint32_t)((t / SCAN_THREADS) % SCAN_GROUPS);
    uint64_t b = (t / (SCAN_THREADS * SCAN_GROUPS)) * RAW_BLOCKS;
    if (b >= n_blocks) return;
    // first draw of the group: r = b*SCAN_BLOCK + SCAN_D*tau + 4*g  ->  64-bit output m = r / 2
    const uint64_t m = b * (SCAN_BLOCK / 2) + (uint64_t)(SCAN_D / 2) * tau + 2ull * g;
    const Affine j = lcg_pow(inc, m + 1);  // output m is made from the state after m + 1 steps
    u128 s = j.mult * state0 + j.plus;
    for (int k = 0; k < RAW_BLOCKS && b < n_blocks; ++k, ++b) {
        const uint64_t o0 = xsl_rr32(s);
        const uint64_t o1 = xsl_rr32(s * mult + inc);
        uint4 v;
        v.x = (uint32_t)o0; v.y = (uint32_t)(o0 >> 32); v.z = (uint32_t)o1; v.w = (uint32_t)(o1 >> 32);
        *reinterpret_cast<uint4 *>(raw + b * SCAN_BLOCK + (uint64_t)g * (4 * SCAN_THREADS) + 4ull * tau) = v;
        s = jm * s + jp;
    }
}

// ------------------------------------------------------------------------------------------------
// A1: rejection scan by one workgroup
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t mask_of(uint32_t i) { return 0xffffffffu >> __clz((int)i); }  // i >= 1

// What one thread knows about its SCAN_D draws for a given entering count.
struct ScanRes {
    uint32_t c_used;  // entering count (accepted steps of this block in front of the thread) it was computed for
    uint32_t cnt;     // accepted draws
    bits_t bits;      // accept mask, bit s = draw s accepted
    uint32_t gap;     // fast path: the entering count may move by up to +-gap without flipping any decision
                      //   (min over draws of: threshold - value if accepted, value - threshold - 1 if rejected)
    uint32_t i0;      // threshold of the first draw
    uint32_t mask;    // fast path: the one mask used
    uint32_t fast;    // computed on the fast path
    uint32_t end;     // 1 + local index of the draw that completed the job's last step (0: none)
};

// Sequential pass of one thread over its draws, entering with c accepted steps in front of it.
__device__ __forceinline__ void scan_thread(const uint32_t (&u)[SCAN_D], uint32_t c_in,
                                            uint32_t rem_block, uint32_t M, uint32_t top_mask, uint32_t limit,
                                            ScanRes &r)
{
    uint32_t c = c_in, rem = rem_block;
    if (c >= rem) { c = (c - rem) % M; rem = M; }
    const uint32_t i0 = rem - c;
    uint32_t mask = mask_of(i0);
    r.c_used = c_in; r.i0 = i0; r.mask = mask; r.end = 0;
    // fast path: neither a mask change, nor the end of a permutation, nor the end of the job can
    // happen within SCAN_D accepts
    const bool fast = i0 > (mask >> 1) + SCAN_D && c_in + SCAN_D < limit;
#ifndef SCAN_GAP_FORM
#define SCAN_GAP_FORM 2      // 1: (shift, xor, min) per draw; 2: two unsigned mins (A/B builds)
#endif
#ifndef SCAN_GENERAL_GAP
#define SCAN_GENERAL_GAP 0   // 1: the general loop tracks the gap for its fast lanes (first step of r04; A/B builds)
#endif
#ifdef SCAN_R03_PATHS
    if (fast) {
        uint32_t thr = i0, gap = 0xffffffffu;
        bits_t bits = 0;
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) {
            const uint32_t v = u[s] & mask;
            const int32_t d = (int32_t)(thr - v);        // both < 2^31; accepted iff d >= 0
            const uint32_t acc = (uint32_t)(~d) >> 31;
            gap = min(gap, (uint32_t)(d ^ (d >> 31)));   // d if accepted, -d - 1 if rejected
            bits |= (bits_t)acc << s;
            thr -= acc;
        }
        r.cnt = i0 - thr; r.bits = bits; r.gap = gap; r.fast = 1;
    }
    if (!__any(!fast)) return;  // a wavefront-uniform branch: the straight-line code below must not be merged into every pass
    if (fast) return;
    // general path, branch-free: the band / permutation bookkeeping is evaluated for every draw (it is the
    // identity unless the draw was accepted), so a wavefront with a single such thread pays ~13 plain ALU
    // operations per draw instead of a divergent branch tree
    uint32_t i = i0, off = c_in, end = 0;
    bits_t bits = 0;
#pragma unroll
    for (int s = 0; s < SCAN_D; ++s) {
        const uint32_t v = u[s] & mask;
        const uint32_t acc = ((off < limit) & (v <= i)) ? 1u : 0u;
        bits |= (bits_t)acc << s;
        off += acc;
        i -= acc;
        end = (acc & (off == limit ? 1u : 0u)) ? (uint32_t)s + 1 : end;
        const bool wrap = i == 0;                 // the permutation is complete: the next one starts at M
        const uint32_t half = mask >> 1;
        mask = wrap ? top_mask : (i <= half ? half : mask);
        i = wrap ? M : i;
    }
    const uint32_t cnt = off - c_in;
    r.end = end;
    r.cnt = cnt; r.bits = bits; r.gap = 0; r.fast = 0;
#else
    // r04: ONE pass per wavefront (r03 ran the fast loop for its fast lanes and then the general loop for the others: the
    // wavefront that holds a band change -- the one every round of a computed block waits for -- paid both, ~310
    // instructions).  All lanes fast: the fast loop.  Otherwise every lane takes the general loop, which also tracks the
    // gap, so a fast lane leaves it with exactly what the fast loop would have given it; and the job's end is looked for
    // only by wavefronts that can reach it.
    if (!__any(!fast)) {
        uint32_t thr = i0;
        bits_t bits = 0;
#if SCAN_GAP_FORM == 1
        uint32_t gap = 0xffffffffu;
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) {
            const uint32_t v = u[s] & mask;
            const int32_t d = (int32_t)(thr - v);        // both < 2^31; accepted iff d >= 0
            const uint32_t acc = (uint32_t)(~d) >> 31;
            gap = min(gap, (uint32_t)(d ^ (d >> 31)));   // d if accepted, -d - 1 if rejected
            bits |= (bits_t)acc << s;
            thr -= acc;
        }
#else
        // the slack of an accepted draw is d, of a rejected one -d - 1 = ~d: as UNSIGNED numbers the other one of the pair is
        // >= 2^31 and never the minimum -- two mins on values the loop has anyway, instead of (shift, xor, min)
        uint32_t gacc = 0xffffffffu, grej = 0xffffffffu;
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) {
            const uint32_t v = u[s] & mask;
            const uint32_t d = thr - v;                  // both < 2^31; accepted iff d < 2^31
            const uint32_t nd = ~d;
            const uint32_t acc = nd >> 31;
            gacc = min(gacc, d);
            grej = min(grej, nd);
            bits |= (bits_t)acc << s;
            thr -= acc;
        }
        const uint32_t gap = min(gacc, grej);
#endif
        r.cnt = i0 - thr; r.bits = bits; r.gap = gap; r.fast = 1;
        return;
    }
    // general path, branch-free: the band / permutation bookkeeping is evaluated for every draw (it is the identity
    // unless the draw was accepted) instead of a divergent branch tree
    // (r04, second step: no gap in the general loop.  A wavefront comes here because one of its lanes sits at a band edge or
    // a permutation's end; whatever moves its entering counts moves that edge, and the wavefront is re-evaluated as a whole
    // anyway -- a validity range for its fast lanes bought nothing in the rounds counter, and costs 3 of 12 operations a draw.)
    uint32_t i = i0, gap = 0xffffffffu;
    bits_t bits = 0;
    if (!__any(!(c_in + SCAN_D < limit))) {   // (wavefront-uniform) the job does not end inside these draws
#ifndef SCAN_WRAP_ALWAYS
#define SCAN_WRAP_ALWAYS 0   // (A/B builds)
#endif
        if (!SCAN_WRAP_ALWAYS && !__any(i0 <= SCAN_D)) {           // (wavefront-uniform) nor does a permutation: mask changes only
#pragma unroll
            for (int s = 0; s < SCAN_D; ++s) {
                const uint32_t v = u[s] & mask;
                const int32_t d = (int32_t)(i - v);          // accepted iff d >= 0
                const uint32_t acc = (uint32_t)(~d) >> 31;
#if SCAN_GENERAL_GAP
                gap = min(gap, (uint32_t)(d ^ (d >> 31)));
#endif
                bits |= (bits_t)acc << s;
                i -= acc;
                const uint32_t half = mask >> 1;
                mask = i <= half ? half : mask;
            }
        } else {
#pragma unroll
            for (int s = 0; s < SCAN_D; ++s) {
                const uint32_t v = u[s] & mask;
                const int32_t d = (int32_t)(i - v);
                const uint32_t acc = (uint32_t)(~d) >> 31;
#if SCAN_GENERAL_GAP
                gap = min(gap, (uint32_t)(d ^ (d >> 31)));
#endif
                bits |= (bits_t)acc << s;
                i -= acc;
                const bool wrap = i == 0;                 // the permutation is complete: the next one starts at M
                const uint32_t half = mask >> 1;
                mask = wrap ? top_mask : (i <= half ? half : mask);
                i = wrap ? M : i;
            }
        }
        r.cnt = (uint32_t)__popcll((unsigned long long)bits);
#if SCAN_GENERAL_GAP
        r.bits = bits; r.gap = fast ? gap : 0u; r.fast = fast ? 1u : 0u;
#else
        (void)gap;
        r.bits = bits; r.gap = 0u; r.fast = 0u;
#endif
        return;
    }
    uint32_t off = c_in, end = 0;
#pragma unroll
    for (int s = 0; s < SCAN_D; ++s) {
        const uint32_t v = u[s] & mask;
        const uint32_t acc = ((off < limit) & (v <= i)) ? 1u : 0u;
        bits |= (bits_t)acc << s;
        off += acc;
        i -= acc;
        end = (acc & (off == limit ? 1u : 0u)) ? (uint32_t)s + 1 : end;
        const bool wrap = i == 0;
        const uint32_t half = mask >> 1;
        mask = wrap ? top_mask : (i <= half ? half : mask);
        i = wrap ? M : i;
    }
    r.end = end;
    r.cnt = off - c_in; r.bits = bits; r.gap = 0; r.fast = 0;
#endif
}

// Is the cached result still the exact result for entering count c_new?  On the fast path every
// threshold moves by -(c_new - c_used); no decision flips while the move stays inside the gaps.
// (32-bit arithmetic: entering counts are at most SCAN_BLOCK, thresholds below 2^31.)
__device__ __forceinline__ bool scan_still_valid(const ScanRes &r, uint32_t c_new, uint32_t M, uint32_t limit)
{
    const int32_t delta = (int32_t)c_new - (int32_t)r.c_used;
    const int32_t i0n = (int32_t)r.i0 - delta;  // new first threshold (same permutation, same band required)
    const uint32_t mag = (uint32_t)(delta < 0 ? -delta : delta);
    const bool moved_ok = r.fast && i0n <= (int32_t)M && i0n <= (int32_t)r.mask && i0n > (int32_t)((r.mask >> 1) + SCAN_D) &&
                          c_new + SCAN_D < limit && mag <= r.gap;
    return delta == 0 || moved_ok;
}

// thread tau's 32 draws of the block at `base` (tiled layout, see k_raw_stream): 8 coalesced loads
__device__ __forceinline__ void scan_load(const uint32_t *__restrict__ raw, uint64_t base, uint32_t tau,
                                          uint32_t (&u)[SCAN_D])
{
    const uint4 *src = reinterpret_cast<const uint4 *>(raw + base) + tau;
#pragma unroll
    for (int q = 0; q < SCAN_D / 4; ++q) {
        const uint4 v = src[q * SCAN_THREADS];
        u[4 * q] = v.x; u[4 * q + 1] = v.y; u[4 * q + 2] = v.z; u[4 * q + 3] = v.w;
    }
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS round trips)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return x;
}

// Expected number of accepted steps after q draws, starting with `rem` steps left in the permutation (mean
// field, closed form per mask band: in a band with top = mask + 1 the threshold decays like exp(-q / top)).
// Only the first guess of the in-block fixed point; follows the acceptance rate through band changes and
// permutation ends, where a constant rate is off by thousands of steps.
__device__ __forceinline__ uint32_t expected_steps(uint32_t rem, float q, uint32_t M)
{
    float i = (float)rem, acc = 0.f;
    for (int guard = 0; guard < 64 && q > 0.f; ++guard) {
        uint32_t ii = (uint32_t)i;
        if (ii == 0) { i = (float)M; ii = M; }
        const uint32_t m = mask_of(ii);
        const float top = (float)m + 1.f, lo = (float)((m >> 1) + 1);
        const float need = top * __logf((i + 1.f) / lo);  // draws to leave the band
        if (need <= q) { q -= need; acc += i - lo + 1.f; i = lo - 1.f; }
        else { const float inew = (i + 1.f) * __expf(-q / top) - 1.f; acc += i - inew; q = 0.f; }
    }
    return (uint32_t)(acc + 0.5f);
}

#ifdef PHI_PROFILE
__device__ unsigned long long g_phi_prof[32 + 2 * 128];   // development: see k_chain, block_fixed_point (read by sc_permgen_profile);
                                                          // [32 ..): log of (unit | first unit of the launch << 32, clocks waited) of long waits
#endif

__device__ __forceinline__ uint32_t select64(uint64_t x, uint32_t r)  // position of the set bit of rank r < popc(x)
{
    uint32_t pos = 0;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        const uint32_t c = (uint32_t)__popcll(sc_shr64(x, pos) & ((1ull << sh) - 1ull));   // (sh is a literal: a constant mask)
        if (r >= c) { r -= c; pos += sh; }
    }
    return pos;
}

struct BlockShared {
    uint32_t wsum[2][SCAN_THREADS / 64];   // per wavefront: accept count | (recomputed something last round) << 31; by round parity
};

// r04: the END of a permutation, solved draw by draw (k_chain's computed blocks).
// The fixed point below grows its exact prefix band by band, and through the last few hundred steps of a permutation --
// thresholds of a few hundred, where every accepted draw before a thread changes what the thread does -- by two or three
// THREADS a round: 8 of the ~16 rounds of the block in which a permutation ends (CPU restatement of the rounds: the
// front's thread by round reads 924, 926, 937, 939, 940, 943, 945 of 1024; scripts/fixed_point_rounds_sim.py).  Once the
// exact prefix has reached TAIL_I steps before the permutation's end, wavefront 0 takes over from there with one LANE per
// draw: 64 draws at a time, accept set = fixed point of "v <= i - (accepted lanes below)" by ballot (2-3 iterations, ~50
// clocks each), cut at the draw that halves the mask or completes the permutation.  It runs to the end of the thread in
// which the permutation ends; those threads take their accept bits and entering counts from LDS and are never stale
// again; the threads behind start a permutation with i = M, where a round settles them.
#ifndef TAIL_I
#define TAIL_I 1024                    // steps before the permutation's end at which the lanes take over
#endif
#define TAIL_THREADS (TAIL_I / 4 + 32) // threads' worth of staged draws: the last TAIL_I steps take ~1.37 TAIL_I draws (sd ~ sqrt)
struct TailShared {
    uint32_t fst[2][SCAN_THREADS / 64];            // per wavefront: (first stale thread << 16 | min(steps left there, 0xffff)), or ~0; by round parity
    uint32_t u[TAIL_THREADS * SCAN_D];             // the staged draws
    uint8_t acc[TAIL_THREADS * SCAN_D];            // their accept decisions
    uint32_t cin[TAIL_THREADS];                    // entering count of each solved thread, relative to the front's
    uint32_t nsolved;                              // threads solved
    uint32_t open;                                 // this block may still call the lanes (kept here: k_chain has no register to spare)
};

// wavefront 0: draws tu[0 .. nq) entered with i steps left in the permutation (mask = mask_of(i)); returns the number of
// draws decided (a multiple of SCAN_D: through the thread in which the permutation ends, or all nq)
__device__ __forceinline__ uint32_t tail_solve(TailShared &ts, uint32_t nq, uint32_t i, uint32_t M, uint32_t top_mask)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t mask = mask_of(i), c = 0, q = 0;
    bool wrapped = false;
    while (q < nq) {
        uint32_t len = nq - q < 64u ? nq - q : 64u;
        if (wrapped) {                       // finish the thread in which the permutation ended, then stop
            const uint32_t restd = (SCAN_D - (q & (SCAN_D - 1))) & (SCAN_D - 1);
            if (restd == 0) break;
            len = restd;
        }
        const bool in = lane < len;
        const uint32_t v = in ? (ts.u[q + lane] & mask) : 0xffffffffu;
        unsigned long long A = __ballot(in && v <= i);          // every threshold at its upper bound
        for (int it = 0; it < 66; ++it) {                        // (the exact prefix grows by a lane an iteration at least)
            const uint32_t pre = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(A >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)A, 0u));
            const unsigned long long A2 = __ballot(in && pre <= i && v <= i - pre);
            if (A2 == A) break;
            A = A2;
        }
        // the accepts that change the mask (or complete the permutation): i falls to mask >> 1 after i - (mask >> 1) of them
        const uint32_t half = mask >> 1, nb = i - half;
        uint32_t T = (uint32_t)__popcll(A);
        if (T >= nb) {
            len = select64(A, nb - 1u) + 1u;
            A &= sc_low_mask64(len);
            T = nb;
        }
        if (lane < len) {
            ts.acc[q + lane] = (uint8_t)((A >> lane) & 1ull);
            if (((q + lane) & (SCAN_D - 1)) == 0)
                ts.cin[(q + lane) / SCAN_D] = c + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(A >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)A, 0u));
        }
        c += T; i -= T; q += len;
        if (i == 0) { i = M; mask = top_mask; wrapped = true; }
        else if (i <= half) mask = half;
    }
    return q;
}

// inclusive prefix sum inside each row of 16 lanes
__device__ __forceinline__ uint32_t row16_inclusive_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    return x;
}

// The exact result of ONE block of SCAN_BLOCK draws entered with S_block completed steps, by the whole
// workgroup: every thread ends with its accept mask (r.bits), its entering count (excl = accepted steps of
// the block in front of it) and the block's accept count.  Fixed point on the entering counts: a thread
// recomputes only when its cached result is not provably the result for its new entering count; a thread
// with the right entering count produces the right count, so the correct prefix grows every round.
// One barrier per round: the wavefronts publish their counts together with "one of my threads recomputed in the
// previous round"; a round that learns that nobody did has just rebuilt the entering counts of the previous round,
// for which every cached result was valid: the result.  (r02: the first form paid two barriers and ~150 instructions
// of bookkeeping per wavefront and round -- 16 wavefronts on one CU make a round throughput-bound, ~2.5 us; measured
// 3.6 rounds for an ordinary computed block, 13 for the block in which a permutation ends.)
// Returns 1 if the iteration cap was hit (cannot happen: the prefix grows by at least one thread a round).
template <bool TAIL = false>
__device__ __forceinline__ int block_fixed_point(const uint32_t (&u)[SCAN_D], uint64_t S_block,
                                                 uint32_t rem_block, uint32_t M, uint32_t top_mask, uint64_t total_steps, BlockShared &sh,
                                                 uint32_t &parity, ScanRes &r, uint32_t &excl, uint32_t &total_cnt,
                                                 int *rounds_out = nullptr, TailShared *ts = nullptr)
{
    // rem_block = M - S_block % M, the steps left in the current permutation (callers carry it along: a
    // 64-bit modulo per block by every wavefront costs more than a fifth of the block)
    constexpr int NW = SCAN_THREADS / 64;
    static_assert(NW <= 16, "the wavefront counts are combined inside one row of 16 lanes");
    const uint32_t tau = threadIdx.x, lane = tau & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tau >> 6));
    const uint64_t left = total_steps - S_block;
    const uint32_t limit = left > 0xffffffffULL ? 0xffffffffu : (uint32_t)left;
    // first guess of the entering count: the expected count (any guess converges; a good one saves rounds)
#ifdef PHI_PROFILE
    const long long pf_a = clock64();
#endif
    scan_thread(u, expected_steps(rem_block, (float)(tau * SCAN_D), M), rem_block, M, top_mask, limit, r);
#ifdef PHI_PROFILE
    if (rounds_out && tau == 0) { atomicAdd(&g_phi_prof[24], (unsigned long long)(clock64() - pf_a)); atomicAdd(&g_phi_prof[25], 1ull); }
#endif
    excl = 0; total_cnt = 0;
    uint32_t recomputed = 1u;
#ifdef PHI_PROFILE
    long long pf_r = 0;
#endif
    // TAIL: only where the permutation that ends is not the job's last (its end is the job's `limit`, which the lanes do not know)
    if (TAIL) {
        if (tau == 0) ts->open = (left > (uint64_t)rem_block && rem_block <= SCAN_BLOCK) ? 1u : 0u;
        if (lane == 0) ts->fst[parity][wave] = 0xffffffffu;   // (no front before the first round)
    }
    uint32_t incl = 0;
    for (int iter = 0;; ++iter) {
        if (recomputed) incl = wave_inclusive_scan(r.cnt);   // (wavefront-uniform: a wavefront that re-evaluated nothing keeps its sums)
        if (lane == 63) sh.wsum[parity][wave] = incl | (recomputed << 31);
#ifdef PHI_PROFILE
        const long long pf_b0 = clock64();
        if (rounds_out && tau == 0 && iter > 0) atomicAdd(&g_phi_prof[3], (unsigned long long)(pf_b0 - pf_r));   // wavefront 0's own work of a round
#endif
        __syncthreads();
#ifdef PHI_PROFILE
        pf_r = clock64();
        if (rounds_out && tau == 0) { atomicAdd(&g_phi_prof[7], (unsigned long long)(pf_r - pf_b0)); atomicAdd(&g_phi_prof[11], 1ull); }   // its wait at the barrier
#endif
        const uint32_t mine = lane < NW ? sh.wsum[parity][lane] : 0u;
        const bool tail_open = TAIL && ts->open != 0u;
        const uint32_t fst = (tail_open && lane < NW) ? ts->fst[parity][lane] : 0xffffffffu;
        parity ^= 1u;   // the other buffer is rewritten only after the next barrier, i.e. after everybody has read this one
        const bool anybody = __any((int)(mine >> 31));
        const uint32_t run = row16_inclusive_scan(mine & 0x7fffffffu);
        total_cnt = (uint32_t)__builtin_amdgcn_readlane((int)run, NW - 1);
        const uint32_t before = wave ? (uint32_t)__builtin_amdgcn_readlane((int)run, wave - 1) : 0u;
        excl = before + incl - r.cnt;
        if (!anybody) {           // the counts are those of the previous round, in which every cached result was valid
            if (rounds_out) *rounds_out = iter + 1;
            return 0;
        }
        bool pinned_now = false;
        if (tail_open) {
            // The front: the first stale thread of the previous round.  Nobody in front of it was stale, so its entering
            // count was exact then and still is (and it has been recomputed with it since).
            const unsigned long long has = __ballot(fst != 0xffffffffu);
            if (has) {
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)fst, (int)__builtin_ctzll(has));
                const uint32_t s = f >> 16, i_front = f & 0xffffu;
                if (i_front <= TAIL_I) {   // (uniform) -- steps left in the permutation the BLOCK was entered in
                    const uint32_t nthr = SCAN_THREADS - s < TAIL_THREADS ? SCAN_THREADS - s : TAIL_THREADS;
                    if (tau >= s && tau < s + nthr) {
#pragma unroll
                        for (int k = 0; k < SCAN_D; ++k) ts->u[(tau - s) * SCAN_D + k] = u[k];
                    }
                    __syncthreads();
                    if (wave == 0) {
                        const uint32_t nq = tail_solve(*ts, nthr * SCAN_D, i_front, M, top_mask);
                        if (lane == 0) { ts->nsolved = nq / SCAN_D; ts->open = 0u; }   // (everybody has read `open` in front of the barrier above)
                    }
                    __syncthreads();
                    const uint32_t ns = ts->nsolved;
                    if (tau >= s && tau < s + ns) {
                        bits_t b = 0;
#pragma unroll
                        for (int k = 0; k < SCAN_D; ++k) b |= (bits_t)ts->acc[(tau - s) * SCAN_D + k] << k;
                        r.bits = b; r.cnt = (uint32_t)__popcll((unsigned long long)b); r.gap = 0; r.fast = 0; r.end = 0;
                        r.c_used = (rem_block - i_front) + ts->cin[tau - s];   // exact: valid from the next round on, and for good
                        pinned_now = true;
                    }
                }
            }
        }
        const bool stale = !pinned_now && !scan_still_valid(r, excl, M, limit);
        recomputed = (__any(stale) || __any(pinned_now)) ? 1u : 0u;
        if (tail_open) {   // my wavefront's first stale thread | steps left there, for the next round (that buffer's readers are done)
            const unsigned long long sm = __ballot(stale);
            uint32_t my_front = 0xffffffffu;
            if (sm) {
                const int fl = (int)__builtin_ctzll(sm);
                const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)excl, fl);
                // steps left at that thread in the permutation the block was entered in (beyond its end: not our business)
                const uint32_t il = ce < rem_block ? rem_block - ce : 0xffffu;
                my_front = (((uint32_t)wave * 64u + (uint32_t)fl) << 16) | (il < 0xffffu ? il : 0xffffu);
            }
            if (lane == 0) ts->fst[parity][wave] = my_front;
        }
#ifdef PHI_PROFILE
        if (rounds_out && lane == 0) {   // wavefronts that recompute, by round (1, 2, 3, later)
            if (recomputed) atomicAdd(&g_phi_prof[26 + (iter < 3 ? iter : 3)], 1ull);
        }
#endif
        if (recomputed) {
            if (stale) scan_thread(u, excl, rem_block, M, top_mask, limit, r);
        }
        if (iter > SCAN_THREADS + 8) return 1;
    }
}

// steps left in the current permutation after t more steps
__device__ __forceinline__ uint32_t rem_advance(uint32_t rem, uint32_t t, uint32_t M)
{
    if (t >= rem) { t = (t - rem) % M; rem = M; }
    return rem - t;
}

// Per processed block the scan leaves: sblk[b] = steps completed before the block, and per thread
// acc_bits[b*SCAN_THREADS + tau], enter[b*SCAN_THREADS + tau] (accepted steps of the block in front of the thread).
// k_expand turns these into J with the whole chip; one CU cannot store 4 bytes per step fast enough.
//
// st[0] = steps completed so far, st[1] = next block to process, st[2] = sticky failure flags,
// st[3] = number of raw draws consumed when the job's last step completed.
// A launch processes WHOLE blocks while fewer than S_target steps are complete (the last block may
// run past the target; only the end of the job, total_steps, stops mid-block).
// This is the sequential form of the scan: every block is entered with the exact state its predecessor
// left.  It is the whole generator for short permutations and the fallback of the block-parallel form below.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(const uint32_t *__restrict__ raw, uint64_t n_blocks,
                                                       uint32_t n, uint64_t S_target, uint64_t total_steps,
                                                       bits_t *__restrict__ acc_bits,
                                                       uint32_t *__restrict__ enter,
                                                       unsigned long long *__restrict__ sblk,
                                                       unsigned long long *__restrict__ st)
{
    __shared__ BlockShared sh;
    static_assert(SCAN_THREADS % 64 == 0 && SCAN_D % 4 == 0 && SCAN_D <= 64, "scan geometry");
    const uint32_t tau = threadIdx.x;
    const uint32_t M = n - 1;
    const uint32_t top_mask = mask_of(M);
    uint64_t S_block = st[0];
    uint64_t b = st[1];
    uint32_t rem_block = M - (uint32_t)(S_block % M);
    uint32_t parity = 0;
    int failed = 0;
    uint64_t endpos = 0;

    uint32_t un[SCAN_D];  // the next block's draws, loaded while the current block is processed
    if (b < n_blocks) scan_load(raw, b * SCAN_BLOCK, tau, un);
    for (; b < n_blocks && S_block < S_target; ++b) {
        uint32_t u[SCAN_D];
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) u[s] = un[s];
        if (b + 1 < n_blocks) scan_load(raw, (b + 1) * SCAN_BLOCK, tau, un);
        ScanRes r;
        uint32_t excl, total_cnt;
        if (block_fixed_point(u, S_block, rem_block, M, top_mask, total_steps, sh, parity, r, excl, total_cnt) > 0) {
            failed = 1;
            break;
        }
        acc_bits[b * SCAN_THREADS + tau] = r.bits;
        enter[b * SCAN_THREADS + tau] = excl;
        if (tau == 0) sblk[b] = S_block;
        if (r.end) endpos = b * SCAN_BLOCK + (uint64_t)tau * SCAN_D + r.end;
        S_block += total_cnt;
        rem_block = rem_advance(rem_block, total_cnt, M);
    }
    if (endpos) st[3] = endpos;  // exactly one thread of one launch sees the job's last step
    if (tau == 0) {
        st[0] = S_block;
        st[1] = b;
        if (failed) st[2] = st[2] | 1ull;
    }
}

// J[step] for every accepted draw of blocks [st_prev_block, st[1]) -- the whole chip, one thread per
// scan thread.  blk0 = first block of this range (read from st_range[0]), end = st_range[1].
__global__ __launch_bounds__(256) void k_expand(const uint32_t *__restrict__ raw,
                                                const bits_t *__restrict__ acc_bits,
                                                const uint32_t *__restrict__ enter,
                                                const unsigned long long *__restrict__ sblk,
                                                const unsigned long long *__restrict__ range, uint32_t n,
                                                uint64_t total_steps, int32_t *__restrict__ J)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tau = (uint32_t)(t % SCAN_THREADS);
    const uint32_t M = n - 1;
    const uint32_t top_mask = mask_of(M);
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) / SCAN_THREADS;
    for (uint64_t b = range[0] + t / SCAN_THREADS; b < range[1]; b += stride) {
        const bits_t bits = acc_bits[b * SCAN_THREADS + tau];
        if (!bits) continue;
        uint64_t S = sblk[b] + enter[b * SCAN_THREADS + tau];
        uint32_t i = M - (uint32_t)(S % M);
        uint32_t mask = mask_of(i);
        const uint4 *src = reinterpret_cast<const uint4 *>(raw + b * SCAN_BLOCK) + tau;
#pragma unroll
        for (int q = 0; q < SCAN_D / 4; ++q) {
            if (!((uint32_t)(bits >> (4 * q)) & 0xfu)) continue;
            const uint4 v4 = src[q * SCAN_THREADS];
            const uint32_t vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((uint32_t)(bits >> (4 * q + e)) & 1u) {
                    if (S < total_steps) J[S] = (int32_t)(vv[e] & mask);
                    ++S; --i;
                    if (i == 0) { i = M; mask = top_mask; }
                    else if (i <= (mask >> 1)) mask >>= 1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A1 block-parallel: the same exact scan with the per-block work spread over the chip
//
// The only thing block b needs from its predecessors is ONE number, the state S_b (completed steps) it
// is entered with.  S_b is known in advance up to a random-walk error (sigma ~ 0.5 sqrt(draws) since the
// last exactly known state), so the chip prepares every block of a unit in parallel for a WINDOW of
// entry states around a guess G_b (k_phi_events + k_phi_tbuild), a single workgroup then chains the exact states through
// the prepared blocks (k_chain), and the chip finally recomputes every block from its now known exact
// entry state and checks S_b + count_b == S_{b+1} (k_block_exact): the result is exact by induction or
// a failure flag is raised (then the caller reruns the sequential form).
//
// Preparation of block b ("gap transfer").  Let the BASE trajectory enter with G_b and a second one with
// G_b + g (gap g, |g| <= w).  As long as both stay inside one permutation and one mask band, a draw with
// masked value v at a position where the base threshold is t is decided differently only when
//   g > 0 (second is ahead, its threshold is t - g):  base accepts, second rejects  <=>  g > t - v       (>= 0)
//   g < 0 (second is behind, threshold t + |g|):      base rejects, second accepts  <=>  |g| > v - t - 1 (>= 0)
// and each such event shrinks |g| by one.  The map entry gap -> exit gap is therefore monotone with unit
// steps, and it is represented per side by the set of increments d-1 -> d that survive: start with w set
// bits, and for every event of slack s (in draw order) clear the set bit of rank s if it exists.  The exit
// gap of entry gap d is the number of set bits among the first d.  Blocks in which some trajectory of the
// window crosses a mask band, a permutation end or the job end ("hard" blocks, about a fifth at n = 1M),
// or that hold too many events, are not prepared; the chain workgroup computes them itself from the exact
// entry state, exactly like the sequential scan.
// ------------------------------------------------------------------------------------------------

#define PHI_W 16384               // window bits per side
#define PHI_WORDS (PHI_W / 64)
#ifndef PHI_MAX_EV
#define PHI_MAX_EV 2048           // events per side a prepared block may hold
#endif
#ifndef PHI_WINDOW
#define PHI_WINDOW 2.25           // window half-width in units of sqrt(draws since the reference state) (= 4.5 sigma)
#endif
#ifndef PHI_UNIT
#define PHI_UNIT 512              // blocks per launch unit (the chain pays ~0.17 ms between launches; with 32-draw
                                  // threads: 80 -> 842, 112 -> 900, 160 -> 1017, 224 -> 1024, 320 -> 1020-1033 genes/s in
                                  // the pipeline; with 16-draw threads: 384 -> 1068, 448 -> 1067, 512 -> 1071)
#endif
#ifndef PHI_AHEAD_MAX
#define PHI_AHEAD_MAX 3
#endif
                                  // units prepared ahead of the chain (their guesses use a state ahead + 1 units old):
                                  // 1 when the generator has the chip to itself, 3 next to the scoring kernel, whose
                                  // workgroups hold the CUs for milliseconds (wider windows, ~25 % more computed blocks)
#ifndef PHI_RING
#define PHI_RING 4096
#endif
                                // table ring slots: EIGHT units.  Units are cut at chunk ends, so a short unit shifts the ring
                                  // positions of its successors, and unit v + 5 can then land on slots of unit v.  The chain is
                                  // done with unit v by then (k_gate), but k_seg_fill(v) -- which runs behind the chain on stream
                                  // v % 4 -- need not be: with four units of slots, a fill starved of compute units for a
                                  // millisecond read descriptors that unit v + 5's preparation had overwritten (seen as a
                                  // verification fallback when the scoring kernel left 64 or 32 CUs).  With eight, the first
                                  // unit on ANOTHER stream that can reach v's slots is v + 9, whose gate (chain done with unit
                                  // >= v + 5) implies publish(v + 4), which sits behind fill(v) in stream v % 4.
#define PHI_STREAMS 4             // preparation streams (units rotate over them)
#define PHI_MIN_N (1 << 17)       // below this every block holds a band crossing: sequential form

static_assert((PHI_AHEAD_MAX + 1) * PHI_UNIT <= PHI_RING, "a unit's ring slots are reused only after the chain consumed them");

struct PhiDesc {
    unsigned long long G;  // guessed entry state of the block
    uint32_t cnt;          // accepts of the base trajectory
    uint32_t i_in;         // steps left in G's permutation (M - G % M)
    uint16_t w_pos;        // entry states G + d, 0 <= d <= w_pos, are covered (trajectories ahead of the base)
    uint16_t w_neg;        // entry states G - d, 0 <= d <= w_neg, are covered (trajectories behind the base)
    uint16_t n_pos, n_neg; // events per side
    uint32_t prepared;     // 0: the chain computes this block itself
    uint32_t w;            // the window the block was prepared for (w_pos / w_neg are smaller next to a band edge)
};

// Expected state after dq more draws from state S (mean-field, closed form per mask band).  Only a guess:
// exactness never depends on it.
__device__ static unsigned long long phi_expect(unsigned long long S, double dq, uint32_t M, double dpp,
                                                unsigned long long total)
{
    int phase = 0;
    for (int guard = 0; guard < 256 && dq > 0.0 && S < total; ++guard) {
        const uint32_t done = (uint32_t)(S % M);
        if (done == 0 && phase == 0) {  // at a permutation boundary: skip whole permutations
            const double k = floor(dq / dpp);
            if (k >= 1.0) { S += (unsigned long long)k * M; dq -= k * dpp; }
            phase = 1;
            continue;
        }
        const uint32_t i = M - done, m = mask_of(i), lo = (m >> 1) + 1;  // band: i in [lo, m]
        const double top = (double)m + 1.0;
        const double need = top * log(((double)i + 1.0) / (double)lo);  // draws to leave the band
        if (need <= dq) { dq -= need; S += (unsigned long long)(i - lo + 1); }
        else { const double inew = ((double)i + 1.0) * exp(-dq / top) - 1.0; S += (unsigned long long)((double)i - inew + 0.5); dq = 0.0; }
    }
    return S < total ? S : total;
}

// One wavefront builds the surviving-increment bitset of one side (lane l holds bits [256 l, 256 l + 256)).
__device__ __forceinline__ void phi_tbuild(const uint16_t *ev, uint32_t nev, uint32_t w, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63;
    uint64_t w0, w1, w2, w3;
    {
        const uint32_t base = 256 * lane;
#define PHI_INIT(k) (w > base + 64 * (k) ? sc_low_mask64(w - base - 64 * (k) < 64u ? w - base - 64 * (k) : 64u) : 0ull)
        w0 = PHI_INIT(0); w1 = PHI_INIT(1); w2 = PHI_INIT(2); w3 = PHI_INIT(3);
#undef PHI_INIT
    }
    uint32_t cnt = (uint32_t)(__popcll(w0) + __popcll(w1) + __popcll(w2) + __popcll(w3));
    uint32_t pre = wave_inclusive_scan(cnt);
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
    for (uint32_t e0 = 0; e0 < nev; e0 += 64) {
        const uint32_t mine = e0 + lane < nev ? ev[e0 + lane] : 0xffffu;  // 64 events per (coalesced) load
        const uint32_t nb = nev - e0 < 64 ? nev - e0 : 64;
        for (uint32_t j = 0; j < nb; ++j) {
            const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)j);
            if (s >= total) continue;  // no trajectory of the window has a gap above s any more
            const bool own = (pre - cnt <= s) && (s < pre);
            const uint32_t L = (uint32_t)__builtin_ctzll(__ballot(own));
            if (lane == L) {
                uint32_t r = s - (pre - cnt);
                const uint32_t c0 = (uint32_t)__popcll(w0), c1 = (uint32_t)__popcll(w1), c2 = (uint32_t)__popcll(w2);
                if (r < c0) w0 &= ~sc_bit64(select64(w0, r));
                else if (r < c0 + c1) w1 &= ~sc_bit64(select64(w1, r - c0));
                else if (r < c0 + c1 + c2) w2 &= ~sc_bit64(select64(w2, r - c0 - c1));
                else w3 &= ~sc_bit64(select64(w3, r - c0 - c1 - c2));
                cnt -= 1;
            }
            pre -= (lane >= L) ? 1u : 0u;
            total -= 1;
        }
    }
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out + 4 * lane);
    o[0] = make_ulonglong2(w0, w1);
    o[1] = make_ulonglong2(w2, w3);
}

// exit gap of entry gap idx (<= w) on one side: set bits among the first idx (wave 0 only, all lanes)
__device__ __forceinline__ uint32_t phi_lookup(const unsigned long long *tb, uint32_t idx)
{
    const uint32_t lane = threadIdx.x & 63, base = 256 * lane;
    uint32_t t = 0;
    if (idx > base) {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(tb + 4 * lane);
        const ulonglong2 a = src[0], b = src[1];
        const uint64_t wd[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = base + 64 * k;
            if (idx > lo) t += (uint32_t)__popcll(wd[k] & sc_low_mask64(idx - lo < 64u ? idx - lo : 64u));
        }
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(t), 63);
}

// Prepare blocks [b0, b1), part 1: one workgroup per block finds the base trajectory from the guess G_b and
// writes the events of both sides (slacks, in draw order).  The guess comes from the exact state at ref_block
// (ahead + 1 units back), the window from the distance to it.
__global__ __launch_bounds__(SCAN_THREADS) void k_phi_events(const uint32_t *__restrict__ raw, uint32_t n,
                                                             uint64_t total_steps, double dpp, uint64_t b0,
                                                             uint64_t b1, uint64_t ref_block,
                                                             const unsigned long long *__restrict__ sblk,
                                                             PhiDesc *__restrict__ desc,
                                                             uint16_t *__restrict__ events, uint32_t *__restrict__ seglist)
{
    __shared__ BlockShared sh;
    __shared__ unsigned long long shG;
    __shared__ uint32_t shw, shi;
    __shared__ uint32_t wpk[SCAN_THREADS / 64];
    const uint64_t b = b0 + blockIdx.x;
    if (b >= b1) return;
    const uint32_t tau = threadIdx.x, lane = tau & 63, wave = tau >> 6;
    const uint32_t M = n - 1, top_mask = mask_of(M);
    const uint64_t slot = b % PHI_RING;
    if (blockIdx.x == 0 && tau == 0) seglist[0] = 0;   // the unit's list of multi-block segments (filled by k_phi_tbuild)
    if (tau == 0) {
        const double dq = (double)(b - ref_block) * (double)SCAN_BLOCK;
        shG = phi_expect(sblk[ref_block], dq, M, dpp, total_steps);
        // ~4.5 sigma of the random walk since the reference state (sigma = 0.49 sqrt(draws), measured); an entry
        // state outside the window only costs the chain one computed block
        const double wd = PHI_WINDOW * sqrt(dq) + 64.0;
        shw = wd < (double)(PHI_W - 1) ? (uint32_t)wd : (uint32_t)(PHI_W - 1);
        shi = M - (uint32_t)(shG % M);
    }
    uint32_t u[SCAN_D];
    scan_load(raw, b * SCAN_BLOCK, tau, u);
    __syncthreads();
    const uint64_t G = shG;
    const uint32_t w = shw;
    const uint32_t i_in = shi;
    bool easy = G + (uint64_t)SCAN_BLOCK + w + 1 < total_steps;
    if (i_in <= SCAN_BLOCK / 2) easy = false;  // the permutation ends inside the block (acceptance >= 1/2): the chain
                                               // computes it anyway, no need to solve it here first
    ScanRes r;
    uint32_t excl = 0, total_cnt = 0, parity = 0;
    uint32_t mask = 0, w_pos = 0, w_neg = 0;
    if (easy) {  // uniform
        if (block_fixed_point(u, G, i_in, M, top_mask, total_steps, sh, parity, r, excl, total_cnt) > 0) easy = false;
        mask = mask_of(i_in);
        const uint32_t cap = mask < M ? mask : M, low = (mask >> 1) + 1;  // the band is [low, mask], capped by M
        // The base must stay in its band and permutation.  A trajectory that enters d ahead of it stays at
        // thresholds >= i_out - d, one that enters d behind at thresholds <= i_in + d: each side is covered as
        // far as its trajectories cannot leave the band either.
        if (i_in < total_cnt + low) easy = false;
        else {
            const uint32_t i_out = i_in - total_cnt;
            w_pos = w < i_out - low ? w : i_out - low;
            w_neg = w < cap - i_in ? w : cap - i_in;
        }
    }
    uint32_t totP = 0, totN = 0, offP = 0, offN = 0;
    if (easy) {
        uint32_t thr = i_in - excl, np = 0, nn = 0;
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) {
            const int32_t d = (int32_t)(thr - (u[s] & mask));
            if (d >= 0) { np += ((uint32_t)d < w_pos) ? 1u : 0u; --thr; }
            else nn += ((uint32_t)(-d - 1) < w_neg) ? 1u : 0u;
        }
        const uint32_t pk = np | (nn << 16);  // both totals <= SCAN_BLOCK <= 65535: no carry between the fields
        const uint32_t incl = wave_inclusive_scan(pk);
        if (lane == 63) wpk[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < SCAN_THREADS / 64; ++k) {
            const uint32_t t = wpk[k];
            before += (k < (int)wave) ? t : 0u;
            all += t;
        }
        const uint32_t ex = before + incl - pk;
        offP = ex & 0xffffu; offN = ex >> 16;
        totP = all & 0xffffu; totN = all >> 16;
        if (totP > PHI_MAX_EV || totN > PHI_MAX_EV) easy = false;
    }
    PhiDesc d;
    d.G = G; d.cnt = total_cnt; d.i_in = i_in; d.w = w;
    d.w_pos = (uint16_t)w_pos; d.w_neg = (uint16_t)w_neg; d.n_pos = (uint16_t)totP; d.n_neg = (uint16_t)totN;
    d.prepared = easy ? 1u : 0u;
    if (tau == 0) desc[slot] = d;
    if (!easy) return;
    uint16_t *evP = events + (slot * 2 + 0) * PHI_MAX_EV, *evN = events + (slot * 2 + 1) * PHI_MAX_EV;
    uint32_t thr = i_in - excl;
#pragma unroll
    for (int s = 0; s < SCAN_D; ++s) {
        const int32_t dd = (int32_t)(thr - (u[s] & mask));
        if (dd >= 0) { if ((uint32_t)dd < w_pos) evP[offP++] = (uint16_t)dd; --thr; }
        else if ((uint32_t)(-dd - 1) < w_neg) evN[offN++] = (uint16_t)(-dd - 1);
    }
}

#ifndef PHI_SEG_MAX
#define PHI_SEG_MAX 16        // blocks per segment at most (segments are cut at multiples of this inside a unit)
#endif
#define PHI_COMPOSE_WGS (PHI_UNIT / 2)   // workgroups of k_phi_compose: one per multi-block segment, the others leave at once
#ifndef PHI_TAIL_PLUS
#define PHI_TAIL_PLUS
#endif
#ifndef PHI_TAIL
#define PHI_TAIL false        // r04 NEGATIVE RESULT, kept as a development build (-DPHI_TAIL=true): the end of a permutation inside a
                              // computed block by lanes (tail_solve, see BlockShared) -- rounds 16.1 -> 10.2, clocks 67.9 k -> 76.2 k
#endif
#define PHI_TAIL_LDS (0 PHI_TAIL_PLUS)   // (the preprocessor cannot test `true`: build the variant with -DPHI_TAIL=true -DPHI_TAIL_PLUS=+1)
#define PHI_NS 6              // segments whose tables the chain stages in LDS at once (a run of prepared blocks)
#define PHI_STAGE_PIECES 64   // 16-byte pieces per side the chain stages: entry gaps up to 8192 (beyond: global memory)

struct PhiSeg {               // one per ring slot, written by k_phi_compose
    unsigned long long G;     // guessed entry state of the segment's first block
    int32_t exit0;            // exit state of the segment for entry state G, relative to G
    uint32_t i_in;            // steps left in G's permutation
    uint16_t vpos, vneg;      // entry states G - vneg .. G + vpos are covered
    uint8_t kind;             // 0: the chain computes this block itself, 1: first block of a segment, 2: inside one
    uint8_t len;              // kind 1: blocks in the segment
    uint8_t own;              // kind 1: the segment's table is the block's own (tbits), else the composed one (ctbits)
    uint8_t bad;              // kind 1: the composition left the windows even for the base trajectory (never seen): no lookup
};
static_assert(sizeof(PhiSeg) == 24, "PhiSeg layout");

__device__ __forceinline__ bool phi_full(const PhiDesc &d) { return d.prepared && d.w_pos == d.w && d.w_neg == d.w; }

// set bits among the first nbit (1 .. 128) bits of a 16-byte piece (32-bit masks only, see xsl_rr32)
__device__ __forceinline__ uint32_t phi_piece_rank(const ulonglong2 a, uint32_t nbit)
{
    const uint32_t wd[4] = {(uint32_t)a.x, (uint32_t)(a.x >> 32), (uint32_t)a.y, (uint32_t)(a.y >> 32)};
    uint32_t T = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t lo = 32u * j;
        const uint32_t m = nbit >= lo + 32u ? 0xffffffffu : (nbit > lo ? ((1u << ((nbit - lo) & 31u)) - 1u) : 0u);
        T += (uint32_t)__popc(wd[j] & m);
    }
    return T;
}

// Which segment a prepared block belongs to (r04, see "SEGMENTS" below): a function of the descriptors of the block, its
// predecessor and its successors alone, so every block classifies ITSELF (thread 0 of its k_phi_tbuild workgroup) and the
// first block of a segment of more than one block enters the unit's list for k_phi_compose.
__device__ __forceinline__ void phi_classify(uint64_t b0, uint64_t b1, uint64_t b, const PhiDesc *__restrict__ desc,
                                             PhiSeg *__restrict__ seg, uint32_t *__restrict__ seglist)
{
    const uint32_t r = (uint32_t)(b - b0), nb = (uint32_t)(b1 - b0);
    const uint64_t slot = b % PHI_RING;
    const PhiDesc cur = desc[slot];
    PhiSeg s;
    s.G = cur.G; s.exit0 = (int32_t)cur.cnt; s.i_in = cur.i_in; s.vpos = cur.w_pos; s.vneg = cur.w_neg;
    s.kind = 0; s.len = 0; s.own = 1; s.bad = 0;
    if (cur.prepared) {
        bool start = r == 0 || (r % PHI_SEG_MAX) == 0 || !phi_full(cur);
        if (!start) start = !phi_full(desc[(b - 1) % PHI_RING]);
        if (!start) s.kind = 2;
        else {
            uint32_t len = 1;
            if (phi_full(cur))
                while (r + len < nb && ((r + len) % PHI_SEG_MAX) != 0 && phi_full(desc[(b + len) % PHI_RING])) ++len;
            s.kind = 1; s.len = (uint8_t)len;
            if (len > 1) {   // its table is composed by k_phi_compose (which completes this descriptor); until then: unusable
                s.bad = 1;
                seglist[1 + atomicAdd(seglist, 1u)] = r;
            }
        }
    }
    seg[slot] = s;
}

// Prepare blocks [b0, b1), part 2: two wavefronts per block turn the event lists into the gap-transfer tables.
__global__ __launch_bounds__(128) void k_phi_tbuild(uint64_t b0, uint64_t b1, const PhiDesc *__restrict__ desc,
                                                    const uint16_t *__restrict__ events,
                                                    unsigned long long *__restrict__ tbits, PhiSeg *__restrict__ seg,
                                                    uint32_t *__restrict__ seglist)
{
    const uint64_t b = b0 + blockIdx.x;
    if (b >= b1) return;
    const uint64_t slot = b % PHI_RING;
    if (threadIdx.x == 64) phi_classify(b0, b1, b, desc, seg, seglist);   // (the second wavefront's first lane; descriptors only)
    const PhiDesc d = desc[slot];
    if (!d.prepared) return;
    const uint32_t side = threadIdx.x >> 6;
    phi_tbuild(events + (slot * 2 + side) * PHI_MAX_EV, side ? d.n_neg : d.n_pos, side ? d.w_neg : d.w_pos,
               tbits + (slot * 2 + side) * PHI_WORDS);
}

// ------------------------------------------------------------------------------------------------
// Hand-over words between the chain workgroup and the preparation launches (r02).
//
// r01 ordered "preparation of unit u -> chain of unit u -> preparation of unit u + ahead + 1" with events: one chain
// launch per unit, a barrier packet in front of it and a marker behind it -- 0.21 ms of idle chain stream per unit
// (37 of 232 ms per bench step).  Now ONE chain launch runs a whole chunk of permutations and both directions are words
// in device memory:  flags[1 + u % 16] = u + 1 once unit u is prepared (k_publish, behind the unit's preparation
// launches in their stream), flags[0] = number of units the chain has completed (k_chain, after each unit; the
// preparation of unit u starts behind k_gate, one wavefront that waits for flags[0] >= u - ahead).
// Every wait gives up after 1 s or when a failure flag is up (e.g. when the streams do not run concurrently: a
// profiler that serialises kernels, fewer hardware queues than streams) and raises flag 8 / 16: the caller then
// reruns the job with the sequential scan, as after a failed verification.
// ------------------------------------------------------------------------------------------------
#define PHI_FLAG_SLOTS 16
#define PHI_WAIT_TICKS 100000000ll    // 1 s of the 100 MHz wall clock; a wait is for ONE launch unit (512 blocks, ~1 ms of work at any n)

// 0: the word arrived; 1: gave up waiting (the caller raises its flag); 2: abandoned, a failure flag is up already
__device__ __forceinline__ int phi_wait_at_least(const uint32_t *flag, uint32_t want, const unsigned long long *st)
{
    const long long t0 = wall_clock64();
    for (uint32_t spins = 0;; ++spins) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return 0;
        if (__hip_atomic_load(st + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return 2;
        if (wall_clock64() - t0 > PHI_WAIT_TICKS || spins > (1u << 28)) return 1;
        __builtin_amdgcn_s_sleep(16);
    }
}

__global__ void k_publish(uint32_t *flags, uint32_t slot, uint32_t value)
{
    if (threadIdx.x == 0) __hip_atomic_store(flags + slot, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_gate(const uint32_t *flags, uint32_t chain_units_needed, unsigned long long *st)
{
    if (threadIdx.x == 0 && phi_wait_at_least(flags, chain_units_needed, st) == 1) atomicOr(st + 2, 16ull);
}

// Can the generator's streams run concurrently?  The hand-over words need the chain's stream and the four preparation
// streams on different hardware queues (GPU_MAX_HW_QUEUES; a profiler that serialises kernels breaks it too).  Probed
// ONCE per context, before the first block-parallel job, instead of finding out through a one-second give-up inside a
// job: in five rounds each stream in turn hosts a setter kernel that is enqueued LAST, behind waiters on the other four;
// two streams that share a queue deadlock in the round where the waiter of the pair sits in front of the setter, and
// that waiter gives up after 20 ms.
__global__ void k_probe_wait(const uint32_t *flag, uint32_t want, uint32_t *timed_out)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (wall_clock64() - t0 > 2000000ll) { atomicOr(timed_out, 1u); return; }   // 20 ms
        __builtin_amdgcn_s_sleep(8);
    }
}

static int permgen_probe_streams(sc_ctx *c, hipStream_t chain_stream)
{
    if (c->pg_probed) return SC_OK;
    c->pg_probed = true;
    std::vector<hipStream_t> ss;
    ss.push_back(chain_stream);
    for (hipStream_t sp : c->stream_pg)
        if (sp && sp != chain_stream) ss.push_back(sp);
    SC_TRY(c->perm_flag.ensure(sizeof(unsigned long long), &c->mem));
    uint32_t *words = c->perm_flag.as<uint32_t>();
    SC_HIP(hipDeviceSynchronize());
    SC_HIP(hipMemset(words, 0, 2 * sizeof(uint32_t)));
    // a stream's hardware queue is created at its first launch, which takes milliseconds: warm every stream up first, or
    // the waiters of round one give up before the setter's queue exists (seen with a second context in one process)
    for (hipStream_t sp : ss) hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, sp, words, 0u, 0u);
    for (hipStream_t sp : ss) SC_HIP(hipStreamSynchronize(sp));
    for (size_t setter = 0; setter < ss.size(); ++setter) {
        for (size_t k = 0; k < ss.size(); ++k)
            if (k != setter) hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, ss[k], words, (uint32_t)(setter + 1), words + 1);
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ss[setter], words, 0u, (uint32_t)(setter + 1));
        for (hipStream_t sp : ss) SC_HIP(hipStreamSynchronize(sp));
    }
    uint32_t host[2] = {0, 0};
    SC_HIP(hipMemcpy(host, words, sizeof(host), hipMemcpyDeviceToHost));
    if (host[1]) {
        c->pg_streams_serial = true;
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        char buf[320];
        snprintf(buf, sizeof(buf), "the HIP streams of this process do not run concurrently (GPU_MAX_HW_QUEUES=%s; the library "
                 "asks for 24 when it is loaded BEFORE the HIP runtime initialises, or a profiler serialises kernels): the "
                 "permutation generator uses its sequential scan (same results, about half the speed)", q ? q : "unset");
        c->pg_note = buf;
    }
    return SC_OK;
}

// ---- r04: the generator's form, asked for instead of discovered (sc_init / spatialcore_amd.init) ----
// Probe the context's generator streams NOW (the first block-parallel job would do it otherwise) and report whether they
// run concurrently, together with the hardware-queue request the runtime saw when it initialised.
extern "C" int sc_ctx_probe_streams(sc_ctx *c, int *concurrent, int *hw_queues_requested)
{
    SC_REQUIRE(c && concurrent, SC_ERR_INVALID, "sc_ctx_probe_streams: null pointer");
    SC_HIP(hipSetDevice(c->device));
    if (!c->stream2) SC_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    for (hipStream_t &sp : c->stream_pg)
        if (!sp) SC_HIP(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    SC_TRY(permgen_probe_streams(c, c->stream2));
    *concurrent = c->pg_streams_serial ? 0 : 1;
    if (hw_queues_requested) {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        *hw_queues_requested = q ? atoi(q) : 0;   // 0: unset (the runtime's default of 4)
    }
    return SC_OK;
}

// Which scan a permutation job of length n takes on this context right now, in words (for provenance records).
extern "C" int sc_ctx_permgen_form(sc_ctx *c, int64_t n, const char **form)
{
    SC_REQUIRE(c && form, SC_ERR_INVALID, "sc_ctx_permgen_form: null pointer");
    if (n < PHI_MIN_N) c->pg_form = "sequential (permutations shorter than 131072: every block holds a band change)";
    else if (c->pg_mode == 1) c->pg_form = "sequential (sc_ctx_set_permgen_mode 1)";
    else if (c->pg_streams_serial) c->pg_form = "sequential: " + c->pg_note;
    else if (!c->pg_note.empty()) c->pg_form = "block-parallel; " + c->pg_note;
    else c->pg_form = "block-parallel";
    *form = c->pg_form.c_str();
    return SC_OK;
}

#ifdef PHI_PROFILE
int sc_permgen_profile(unsigned long long *out32, int reset)
{
    SC_HIP(hipDeviceSynchronize());
    SC_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_phi_prof), sizeof(unsigned long long) * 32));
    if (getenv("SC_PHI_WAIT_LOG")) {   // development: the long waits of the chain, one line each
        unsigned long long lg[2 * 128];
        SC_HIP(hipMemcpyFromSymbol(lg, HIP_SYMBOL(g_phi_prof), sizeof(lg), sizeof(unsigned long long) * 32));
        const unsigned long long cnt = out32[30] < 128 ? out32[30] : 128;
        for (unsigned long long k = 0; k < cnt; ++k)
            fprintf(stderr, "wait: unit %llu (launch began at unit %llu) %.0f us\n", lg[2 * k] & 0xffffffffull, lg[2 * k] >> 32, (double)lg[2 * k + 1] / 2100.0);
    }
    if (reset) {
        unsigned long long z[32 + 2 * 128] = {};
        SC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phi_prof), z, sizeof(z)));
    }
    return SC_OK;
}
#else
int sc_permgen_profile(unsigned long long *out32, int) { memset(out32, 0, sizeof(unsigned long long) * 32); return SC_OK; }
#endif

// ------------------------------------------------------------------------------------------------
// r04: SEGMENTS -- the gap-transfer tables of consecutive prepared blocks composed into one.
//
// r03's clock profile of the chain: 22 % of its time were the table lookups of the prepared blocks (815 clocks per
// block, 75 blocks per permutation of 1M cells), although the tables of a run of prepared blocks are all known before
// the chain gets there.  The map entry gap -> exit gap of one block is monotone with unit steps, and so is a
// composition of such maps (with the constant shifts G_j + cnt_j - G_{j+1} between the blocks' guesses in between):
// the composition is again a bitset of surviving increments.  k_phi_compose builds it on the chip, one workgroup per
// SEGMENT (<= PHI_SEG_MAX consecutive prepared blocks whose windows are not narrowed by a band edge; a block with a
// narrowed window is a segment of its own and keeps its own table), by pushing every entry state of the window through
// the segment's tables.  The chain then pays ONE lookup per segment (every thread evaluates it redundantly from LDS: no
// hand-over between wavefronts), k_seg_fill -- one wavefront per segment, behind the chain's "unit done" word --
// fills in the entry states of the blocks inside the segments from the per-block tables, and k_block_exact verifies all
// of it exactly as before: the composed table of a segment is right or its last block's exit state does not meet the
// chain's.
// ------------------------------------------------------------------------------------------------
// Prepare blocks [b0, b1), part 3: segments.  One workgroup per block; the workgroup of a segment's first block
// composes the segment's table, the others only classify their block.
__device__ __forceinline__ void phi_compose_block(uint64_t b0, uint32_t r, const PhiDesc *__restrict__ desc,
                                                  const unsigned long long *__restrict__ tbits, PhiSeg *__restrict__ seg,
                                                  unsigned long long *__restrict__ ctbits)
{
    static_assert(SCAN_THREADS == 1024 && PHI_W == 16384, "thread t of a side owns entry gaps 32 t .. 32 t + 32");
    __shared__ ulonglong2 tl[2 * 128];      // the current block's tables: [side][128 pieces]
    __shared__ uint32_t tpre[2 * 128];      // set bits in front of each piece inside its wavefront's 64 pieces
    __shared__ uint32_t wtot[4];            // set bits of pieces 0 .. 63 / 64 .. 127 of each side
    __shared__ uint32_t Uw[2 * PHI_W / 32]; // the block's increments on the signed gap axis: bit PHI_W + d = F(d + 1) - F(d)
    __shared__ int32_t shLoT, shHiT, shE0;  // first thread of each side that dropped out; exit state of the base
    const uint64_t b = b0 + r;
    const uint32_t tau = threadIdx.x;
    const uint64_t slot = b % PHI_RING;
    const PhiDesc cur = desc[slot];
    PhiSeg s = seg[slot];                   // kind 1, len > 1 (phi_classify)
    const uint32_t len = s.len;
    __syncthreads();                        // (the shared cells below are reused from the workgroup's previous segment)

    // ---- compose: every entry state of the window through the segment's tables ----
    // Thread (side, t) owns the 33 entry gaps 32 t .. 32 t + 32 of its side, held ASCENDING on the signed state axis
    // (negative side: st[k] belongs to the gap -(32 t + 32 - k)).  The images of neighbouring states differ by 0 or 1
    // (monotone, unit steps), so a block maps the thread's states with ONE rank lookup (its lowest state) and one bit of
    // the block's increment array U per further state:  F(a + 1) - F(a) = U[a - G_j],  U = the negative side's bits
    // reversed, then the positive side's.  A thread whose states are not all inside a block's window drops out; the
    // segment then covers the gaps below that thread (the window's rim, 4.5 sigma out: nothing is lost).
    const uint32_t side = tau >> 9, t = tau & 511u;
    int32_t st[33];
#pragma unroll
    for (int k = 0; k <= 32; ++k) st[k] = side ? -(int32_t)(32u * t + 32u - k) : (int32_t)(32u * t + k);
    bool ok = true;
    if (tau == 0) { shLoT = 512; shHiT = 512; }
    const ulonglong2 *tb2 = reinterpret_cast<const ulonglong2 *>(tbits);
    // the tables and the descriptor of block j + 1 are on their way (registers) while block j is applied: a step is
    // then its ~360 instructions per thread, not those plus two dependent trips to memory (inside the Moran pipeline,
    // next to 6 TB/s of scoring traffic, such a trip takes several microseconds)
    ulonglong2 vnext = make_ulonglong2(0ull, 0ull);
    unsigned long long nG = cur.G;      // (only the three fields a step needs travel ahead: the whole descriptor spilled)
    uint32_t ncnt = cur.cnt, nw = cur.w;
    if (tau < 256) vnext = tb2[((slot * 2 + (tau >> 7)) * PHI_WORDS) / 2 + (tau & 127u)];
    for (uint32_t j = 0; j < len; ++j) {
        const unsigned long long djG = nG;
        const uint32_t djcnt = ncnt, djw = nw;
        const ulonglong2 v = vnext;
        __syncthreads();     // the previous block's lookups are done (and shLoT / shHiT are set)
        if (tau < 256) {     // piece (tau & 127) of side (tau >> 7); a wavefront's 64 pieces are half a side
            const uint32_t piece = tau & 127u;
            const uint32_t ones = (uint32_t)(__popcll(v.x) + __popcll(v.y));
            const uint32_t upto = wave_inclusive_scan(ones);
            tl[tau] = v;
            tpre[tau] = upto - ones;
            if ((tau & 63u) == 63u) wtot[tau >> 6] = upto;
            const uint32_t wd[4] = {(uint32_t)v.x, (uint32_t)(v.x >> 32), (uint32_t)v.y, (uint32_t)(v.y >> 32)};
            if (tau < 128) {   // positive side: bit i of the side is U position PHI_W + i
#pragma unroll
                for (int m = 0; m < 4; ++m) Uw[PHI_W / 32 + 4 * piece + m] = wd[m];
            } else {           // negative side: bit i is U position PHI_W - 1 - i
#pragma unroll
                for (int m = 0; m < 4; ++m) Uw[PHI_W / 32 - 4 * piece - 1 - m] = __brev(wd[m]);
            }
        }
        if (j + 1 < len) {
            const uint64_t sn = (b + j + 1) % PHI_RING;
            nG = desc[sn].G; ncnt = desc[sn].cnt; nw = desc[sn].w;
            if (tau < 256) vnext = tb2[((sn * 2 + (tau >> 7)) * PHI_WORDS) / 2 + (tau & 127u)];
        }
        __syncthreads();
        const int32_t rel = (int32_t)(int64_t)(djG - cur.G);   // this block's guess, relative to the first one's
        const int32_t wj = (int32_t)djw;
        if (ok && (st[0] - rel < -wj || st[32] - rel > wj)) {   // (also: gaps beyond the first block's own window)
            ok = false;
            atomicMin(side ? &shLoT : &shHiT, (int32_t)t);
        }
        if (ok) {
            const int32_t d0 = st[0] - rel;
            const bool neg = d0 < 0;
            const uint32_t idx = (uint32_t)(neg ? -d0 : d0);
            uint32_t T = 0;
            if (idx) {
                const uint32_t piece = (idx - 1u) >> 7, nbit = idx - 128u * piece;
                const uint32_t row = (neg ? 128u : 0u) + piece;
                T = tpre[row] + (piece >= 64u ? wtot[neg ? 2 : 0] : 0u) + phi_piece_rank(tl[row], nbit);
            }
            int32_t run = rel + (int32_t)djcnt + (neg ? -(int32_t)T : (int32_t)T);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int32_t inc = st[k + 1] - st[k];                       // 0 or 1
                const uint32_t q = (uint32_t)(PHI_W + st[k] - rel);          // U position of the step st[k] -> st[k] + 1
                const uint32_t bit = (Uw[q >> 5] >> (q & 31u)) & 1u;
                st[k] = run;
                run += inc & (int32_t)bit;
            }
            st[32] = run;
        }
    }
    if (t == 0 && side == 0) shE0 = ok ? st[0] : (int32_t)0x80000000;
    __syncthreads();
    // surviving increments: positive side bit p = exit(p + 1) - exit(p); negative side bit p = exit(-p) - exit(-p - 1)
    uint32_t word = 0;
    if (ok) {
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (st[k + 1] != st[k]) word |= 1u << (side ? 31 - k : k);
    }
    reinterpret_cast<uint32_t *>(ctbits + (slot * 2 + side) * PHI_WORDS)[t] = word;
    if (tau == 0) {
        const int32_t hiT = shHiT, loT = shLoT, e0 = shE0;
        s.own = 0;
        s.bad = 0;
        if (e0 == (int32_t)0x80000000 || hiT == 0 || loT == 0) { s.bad = 1; s.vpos = 0; s.vneg = 0; }
        else {
            const uint32_t vp = 32u * (uint32_t)hiT, vn = 32u * (uint32_t)loT;
            s.exit0 = e0;
            s.vpos = (uint16_t)(vp < cur.w ? vp : cur.w);
            s.vneg = (uint16_t)(vn < cur.w ? vn : cur.w);
        }
        seg[slot] = s;
    }
}

// (r04 measured and dropped: the unit published by the LAST workgroup of this kernel -- a device-scope release per
// workgroup, i.e. a write-back of the XCD's L2 512 times per unit: generator alone 119 -> 144 ms, bench step 167 -> 214 ms.
// The kernel boundary in front of k_publish does that once.  Also measured: one 1024-thread workgroup per BLOCK, all but
// the ~35 that start a multi-block segment leaving at once -- inside the Moran pipeline those 512 heavy workgroups queued
// for the 96 free CUs, the chain waited 34-40 k clocks per permutation for its units (3.4 k with the chip to itself).)
__global__ __launch_bounds__(SCAN_THREADS) void k_phi_compose(uint64_t b0, const PhiDesc *__restrict__ desc,
                                                             const unsigned long long *__restrict__ tbits,
                                                             PhiSeg *__restrict__ seg,
                                                             unsigned long long *__restrict__ ctbits,
                                                             const uint32_t *__restrict__ seglist)
{
    const uint32_t count = seglist[0];
    // (one segment per workgroup -- a segment has at least two blocks, so PHI_UNIT / 2 workgroups cover any unit; a loop
    // over segments here made hipcc spill 25 registers of the unrolled state arrays)
    if (blockIdx.x < count) phi_compose_block(b0, seglist[1 + blockIdx.x], desc, tbits, seg, ctbits);
}

// Entry states of the blocks inside the segments of blocks [b0, b1) that the chain resolved by ONE lookup (segmode 1):
// one wavefront per segment walks the per-block tables from the segment's entry state.  Runs behind the chain's
// "unit done" word; k_block_exact then verifies every block (the last one's exit state must meet the chain's).
__global__ __launch_bounds__(64) void k_seg_fill(uint64_t b0, uint64_t b1, const PhiDesc *__restrict__ desc,
                                                 const PhiSeg *__restrict__ seg,
                                                 const unsigned long long *__restrict__ tbits,
                                                 const uint8_t *__restrict__ segmode,
                                                 unsigned long long *__restrict__ sblk, uint8_t *__restrict__ hardmask,
                                                 unsigned long long *__restrict__ st)
{
    const uint64_t b = b0 + blockIdx.x;
    if (b >= b1 || segmode[b] != 1) return;
    const uint32_t len = seg[b % PHI_RING].len;
    const uint32_t lane = threadIdx.x;
    unsigned long long S = sblk[b];
    for (uint32_t j = 0; j < len; ++j) {
        const uint64_t slot = (b + j) % PHI_RING;
        const PhiDesc d = desc[slot];
        const int64_t g = (int64_t)S - (int64_t)d.G;
        const bool neg = g < 0;
        const uint64_t idx = (uint64_t)(neg ? -g : g);
        if (!d.prepared || idx > (neg ? d.w_neg : d.w_pos)) {   // the composed table said this could not happen
            if (lane == 0) atomicOr(st + 2, 32ull);
            return;
        }
        const uint32_t T = idx ? phi_lookup(tbits + (slot * 2 + (neg ? 1 : 0)) * PHI_WORDS, (uint32_t)idx) : 0u;
        if (lane == 0) { sblk[b + j] = S; hardmask[b + j] = 0; }
        S = d.G + d.cnt + (neg ? -(long long)T : (long long)T);
    }
}

// ------------------------------------------------------------------------------------------------
// r04: FRESH TABLES for the end of a permutation -- a second tier of preparation.
//
// Of the ~10 blocks per permutation (of 1M cells) the chain computes itself, 7.4 hold a band change or the permutation's
// end and cannot be tabulated; the other 2.6 are blocks of the permutation's last bands whose FIRST-tier preparation
// failed only because its window was too wide for them: a guess made 2-4 launch units ahead is +-13 k states uncertain,
// which is a third of the band [32768, 65535] (the narrowed windows miss, or the events exceed the table limit).  A guess
// made from the exact state at the START of the permutation is +-2 k states uncertain when it reaches those blocks.
// k_fresh runs beside the chain (a handful of workgroups on a stream of their own, for the life of the chain launch): the
// chain posts the exact state of every block boundary at which a new permutation has begun; each helper takes one of the
// clean-looking blocks among the last FR_TAIL of that permutation, solves it from the fresh guess, and turns its events
// into the gap-transfer table of a +-2047 window by walking every entry gap through the event list (with so narrow a
// window that is cheaper than the rank-select build of k_phi_tbuild, and it is done before the chain arrives).  The chain
// consults the fresh table when it meets a block it would otherwise compute (or whose first-tier window missed) in the
// permutation's last FR_REM_MAX steps; what it takes from it is verified by k_block_exact like any prepared block.
// ------------------------------------------------------------------------------------------------
#ifndef PHI_FRESH_TABLES
#define PHI_FRESH_TABLES 0     // r04 NEGATIVE RESULT, kept as a development build (-DPHI_FRESH_TABLES=1; scripts/build_variant.sh): see DESIGN.md 4.3
#endif
#define FR_RING 128            // fresh descriptors / tables (the tails of consecutive permutations are ~85 blocks apart)
#define FR_W 2047              // entry gaps covered per side
#define FR_WORDS 32            // 64-bit words of a side's table
#ifndef FR_HELPERS
#define FR_HELPERS 3
#ifndef FR_ROUNDS
#define FR_ROUNDS 2             // blocks per helper and post
#endif
#endif
#define FR_TAIL 12             // candidates: the blocks before the expected end of the permutation
#define FR_MAX_EV 2048         // events per side
#define FR_REM_MAX 150000u     // the chain asks for a fresh table when at most this many steps of the permutation are left
#define FR_POST_REM 120000u    // ... and posts the reference state for the NEXT permutation's tables when this many are left
#ifndef FR_WINDOW
#define FR_WINDOW 1.75         // window half-width in units of sqrt(draws since the reference state) (3.6 sigma)
#endif

struct FreshDesc {
    unsigned long long G;      // guessed entry state
    uint32_t cnt, i_in;        // accepts of the base trajectory, steps left in G's permutation
    uint16_t w_pos, w_neg;     // covered entry gaps per side
    uint32_t pad_;
    unsigned long long ready;  // block index + 1 once the table is complete (release / acquire)
};
struct FreshCtl {
    uint32_t seq, done;        // posts so far (chain), id of the last chain launch that has ended
    unsigned long long post_b[2], post_S[2];   // [seq & 1]: a block boundary at which a permutation has just begun + its exact state
    unsigned long long post_t[2];              // ... and the wall clock (100 MHz) of the post (diagnostics)
    unsigned long long diag[9];                // diagnostics (scripts/generator_probe.py)
};
static_assert(sizeof(FreshDesc) == 32 && sizeof(FreshCtl) == 128, "fresh-table records");

// Mean-field walk inside ONE permutation, in single precision with the fast intrinsics (a helper has ~5 us for all its
// guesses; steps < 2^24 are exact in float, and a guess needs +-a few states): the steps left after q more draws from
// `rem` steps left (0 when the permutation is over), and the draws until it is over.
__device__ static uint32_t phi_rem_after(uint32_t rem, float q)
{
    float i = (float)rem;
    for (int guard = 0; guard < 40 && q > 0.f && i >= 1.f; ++guard) {
        const uint32_t m = mask_of((uint32_t)i);
        const float top = (float)m + 1.f, lo = (float)((m >> 1) + 1);
        const float need = top * __logf((i + 1.f) / lo);   // draws to leave the band
        if (need <= q) { q -= need; i = lo - 1.f; }
        else { i = (i + 1.f) * __expf(-q / top) - 1.f; q = 0.f; }
    }
    return q > 0.f || i < 1.f ? 0u : (uint32_t)(i + 0.5f);
}
__device__ static float phi_draws_to_finish(uint32_t rem)
{
    float q = 0.f, i = (float)rem;
    for (int guard = 0; guard < 40 && i >= 1.f; ++guard) {
        const uint32_t m = mask_of((uint32_t)i);
        const float lo = (float)((m >> 1) + 1);
        q += ((float)m + 1.f) * __logf((i + 1.f) / lo);
        i = lo - 1.f;
    }
    return q;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_fresh(const uint32_t *__restrict__ raw, uint64_t n_blocks, uint32_t n,
                                                        uint64_t total_steps, double dpp, FreshCtl *__restrict__ ctl,
                                                        FreshDesc *__restrict__ fdesc,
                                                        unsigned long long *__restrict__ ftbits,
                                                        const unsigned long long *__restrict__ st, uint32_t launch_id)
{
    __shared__ BlockShared sh;
    __shared__ uint32_t shCmd, shSeq, shW, shIin;
    __shared__ unsigned long long shB, shG, shBref, shSref, shE, shGs[FR_TAIL + 1];
    __shared__ uint32_t wpk[SCAN_THREADS / 64];
    __shared__ __align__(16) uint16_t evs[2][FR_MAX_EV];
    __shared__ __align__(8) uint8_t tab[2][(FR_W + 1) / 8];
    const uint32_t tau = threadIdx.x, lane = tau & 63, wave = tau >> 6;
    const uint32_t M = n - 1, top_mask = mask_of(M);
    uint32_t last = 0;
    for (;;) {
        if (tau == 0) {   // wait for a post newer than the last one served, the end of the chain launch, a failure or 1 s
            uint32_t cmd = 0, seq = 0;
            const long long t0 = wall_clock64();
            for (;;) {
                seq = __hip_atomic_load(&ctl->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (seq != last) break;
                if (__hip_atomic_load(&ctl->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= launch_id ||
                    __hip_atomic_load(st + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull ||
                    wall_clock64() - t0 > PHI_WAIT_TICKS) { cmd = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            unsigned long long b_ref = 0, S_ref = total_steps;
            if (!cmd) {
                b_ref = ctl->post_b[seq & 1u]; S_ref = ctl->post_S[seq & 1u];
                if (__hip_atomic_load(&ctl->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) S_ref = total_steps;   // (overwritten meanwhile: next round)
            }
            shCmd = cmd; shSeq = seq; shBref = b_ref; shSref = S_ref;
        }
        __syncthreads();
        if (shCmd) return;
        {   // the guesses of the last FR_TAIL + 1 blocks of this permutation, one lane each (a guess is a dozen logarithms)
            const unsigned long long b_ref = shBref, S_ref = shSref;
            if (tau <= FR_TAIL && S_ref < total_steps) {
                const uint32_t rem_ref = M - (uint32_t)(S_ref % M);
                // the post is from the END of a permutation (rem_ref steps left); the tables are for the end of the NEXT one
                const float q_end = phi_draws_to_finish(rem_ref), q_perm = phi_draws_to_finish(M);
                const unsigned long long e = b_ref + (unsigned long long)((q_end + q_perm) / (float)SCAN_BLOCK);
                // cell k: block e + 1 - k; its guess = the next permutation's end minus the steps expected to be left there
                const unsigned long long bk = e + 1 - tau;
                unsigned long long Gk = ~0ull;
                const float q = (float)(bk - b_ref) * (float)SCAN_BLOCK - q_end;   // draws into the next permutation
                if (e + 1 >= b_ref + tau && q > 0.f) {
                    const uint32_t left = phi_rem_after(M, q);
                    Gk = S_ref + rem_ref + M - left;     // (left == 0: at or beyond that permutation's end)
                }
                shGs[tau] = Gk;
                if (tau == 0) shE = e;
            }
        }
        __syncthreads();
        for (uint32_t round = 0; round < FR_ROUNDS; ++round) {   // this helper's blocks of the post, latest first
        if (tau == 0) {
            unsigned long long bsel = ~0ull, Gsel = 0;
            uint32_t wsel = 0, isel = 0;
            const unsigned long long b_ref = shBref, S_ref = shSref;
            if (S_ref < total_steps) {
                // the clean-looking blocks among the last FR_TAIL of this permutation; this helper takes the
                // (blockIdx.x)-th from the END (the late ones are never prepared by the first tier)
                const uint32_t rem_ref = M - (uint32_t)(S_ref % M);
                const unsigned long long perm_end = S_ref + rem_ref + M, e = shE;   // (the end of the NEXT permutation)
                uint32_t taken = 0;
                for (uint32_t k = 0; k < FR_TAIL; ++k) {
                    if (e < b_ref + k) break;
                    const unsigned long long b = e - k;
                    if (b + 1 >= n_blocks) continue;
                    const unsigned long long G = shGs[k + 1], Gnext = shGs[k];   // blocks e - k and e - k + 1
                    if (G == ~0ull || G >= perm_end || Gnext >= perm_end || Gnext + SCAN_BLOCK >= total_steps) continue;
                    const double dq = (double)(b - b_ref) * (double)SCAN_BLOCK;
                    const uint32_t i_in = (uint32_t)(perm_end - G), i_out = (uint32_t)(perm_end - Gnext);
                    const double wd = FR_WINDOW * sqrt(dq) + 32.0;
                    const uint32_t w = wd < (double)FR_W ? (uint32_t)wd : (uint32_t)FR_W;
                    const uint32_t m = mask_of(i_in), low = (m >> 1) + 1;
                    if (i_out < low + w / 4 || i_in > FR_REM_MAX + SCAN_BLOCK) continue;   // a band change inside (or too early)
                    if (taken++ == blockIdx.x + FR_HELPERS * round) { bsel = b; Gsel = G; wsel = w; isel = i_in; break; }
                }
                if (bsel != ~0ull && __hip_atomic_load(&fdesc[bsel % FR_RING].ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == bsel + 1)
                    bsel = ~0ull;   // prepared already (an earlier post of the same permutation)
            }
            shB = bsel; shG = Gsel; shW = wsel; shIin = isel;
            atomicAdd(&ctl->diag[0], 1ull);                                  // diagnostics: rounds, blocks attempted,
            if (bsel != ~0ull) atomicAdd(&ctl->diag[1], 1ull);               // ... ticks from the post to the start of the work
            atomicAdd(&ctl->diag[2], (unsigned long long)wall_clock64() - ctl->post_t[shSeq & 1u]);
        }
        __syncthreads();
        if (round == 0) last = shSeq;
        const unsigned long long b = shB, G = shG;
        const uint32_t w = shW, i_in = shIin;
        __syncthreads();     // (everybody has read the cells thread 0 rewrites next time)
        if (b == ~0ull) continue;
        // ---- the base trajectory from the fresh guess, as k_phi_events does it ----
        uint32_t u[SCAN_D];
        scan_load(raw, b * SCAN_BLOCK, tau, u);
        ScanRes r;
        uint32_t excl = 0, total_cnt = 0, parity = 0;
        bool easy = block_fixed_point(u, G, i_in, M, top_mask, total_steps, sh, parity, r, excl, total_cnt) == 0;
        const uint32_t mask = mask_of(i_in);
        const uint32_t cap = mask < M ? mask : M, low = (mask >> 1) + 1;
        uint32_t w_pos = 0, w_neg = 0;
        if (easy && i_in >= total_cnt + low) {
            const uint32_t i_out = i_in - total_cnt;
            w_pos = w < i_out - low ? w : i_out - low;
            w_neg = w < cap - i_in ? w : cap - i_in;
        } else easy = false;
        uint32_t totP = 0, totN = 0, offP = 0, offN = 0;
        {
            uint32_t thr = i_in - excl, np = 0, nn = 0;
            if (easy) {
#pragma unroll
                for (int s = 0; s < SCAN_D; ++s) {
                    const int32_t d = (int32_t)(thr - (u[s] & mask));
                    if (d >= 0) { np += ((uint32_t)d < w_pos) ? 1u : 0u; --thr; }
                    else nn += ((uint32_t)(-d - 1) < w_neg) ? 1u : 0u;
                }
            }
            const uint32_t pk = np | (nn << 16);
            const uint32_t incl = wave_inclusive_scan(pk);
            if (lane == 63) wpk[wave] = incl;
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int k = 0; k < SCAN_THREADS / 64; ++k) {
                const uint32_t t = wpk[k];
                before += (k < (int)wave) ? t : 0u;
                all += t;
            }
            const uint32_t ex = before + incl - pk;
            offP = ex & 0xffffu; offN = ex >> 16;
            totP = all & 0xffffu; totN = all >> 16;
        }
        if (totP > FR_MAX_EV || totN > FR_MAX_EV) easy = false;   // (uniform: totals are the same in every thread)
        if (easy) {
            uint32_t thr = i_in - excl;
#pragma unroll
            for (int s = 0; s < SCAN_D; ++s) {
                const int32_t dd = (int32_t)(thr - (u[s] & mask));
                if (dd >= 0) { if ((uint32_t)dd < w_pos) evs[0][offP++] = (uint16_t)dd; --thr; }
                else if ((uint32_t)(-dd - 1) < w_neg) evs[1][offN++] = (uint16_t)(-dd - 1);
            }
        }
        __syncthreads();
        if (tau == 0) atomicAdd(&ctl->diag[easy ? 3 : 4], 1ull);   // diagnostics: base trajectories clean / not clean
        if (!easy) continue;   // (uniform)
        const uint64_t slot = b % FR_RING;
        if (tau == 0) {   // the slot's old table is void from here on
            __hip_atomic_store(&fdesc[slot].ready, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
        }
        // ---- the table: every entry gap of the window walked through the side's events (draw order) ----
        // thread (side, t) of the first eight wavefronts owns the gaps 8 t .. 8 t + 8; an event of slack s takes one step
        // off every gap above s: all of the thread's gaps (a counter), none, or -- rarely -- some of them.  Events are
        // read eight at a time (a dependent LDS read per event was most of the walk's time)
        if (tau < 512) {
            const uint32_t side = tau >> 8, t = tau & 255u;
            const uint32_t nev = side ? totN : totP;
            uint32_t a[9], off = 0;
#pragma unroll
            for (int k = 0; k <= 8; ++k) a[k] = 8u * t + k;
            const uint4 *ev8 = reinterpret_cast<const uint4 *>(evs[side]);
            for (uint32_t e0 = 0; e0 < nev; e0 += 8) {
                const uint4 pk = ev8[e0 >> 3];
                const uint32_t wds[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (e0 + j < nev) {   // (uniform per side: all threads of a wavefront belong to one side)
                        const uint32_t lim = ((wds[j >> 1] >> (16 * (j & 1))) & 0xffffu) + off;   // gap > slack  <=>  a > slack + off
                        if (a[0] > lim) ++off;
                        else if (a[8] > lim) {
#pragma unroll
                            for (int k = 0; k <= 8; ++k) a[k] -= (a[k] > lim) ? 1u : 0u;
                        }
                    }
                }
            }
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) bits |= (a[k + 1] - a[k]) << k;   // surviving increments (gaps move together: 0 or 1)
            tab[side][t] = (uint8_t)bits;
        }
        __syncthreads();
        if (tau < 2 * FR_WORDS) {   // 2 x 32 words of 64 bits
            const uint32_t side = tau / FR_WORDS, wd = tau % FR_WORDS;
            ftbits[(slot * 2 + side) * FR_WORDS + wd] = *reinterpret_cast<const unsigned long long *>(&tab[side][8 * wd]);
        }
        __threadfence();
        __syncthreads();
        if (tau == 0) {
            FreshDesc d;
            d.G = G; d.cnt = total_cnt; d.i_in = i_in; d.w_pos = (uint16_t)w_pos; d.w_neg = (uint16_t)w_neg; d.pad_ = 0; d.ready = 0;
            FreshDesc *dst = fdesc + slot;
            dst->G = d.G; dst->cnt = d.cnt; dst->i_in = d.i_in; dst->w_pos = d.w_pos; dst->w_neg = d.w_neg;
            __hip_atomic_store(&dst->ready, (unsigned long long)b + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&ctl->diag[5], (unsigned long long)wall_clock64() - ctl->post_t[last & 1u]);   // diagnostics: ticks from the post to "ready"
        }
        }   // rounds
    }
}

// The chain's side: resolve block bx, entered with S, from its fresh table if there is one that covers S (every thread of
// the workgroup evaluates it: S stays uniform).  Returns false when there is none.
// ready_known: the caller has seen ready == bx + 1 already (a relaxed load issued long before): only the acquire fence is
// needed.  (A slot is rewritten only for a block FR_RING further on, i.e. after the chain has passed this one: no re-check.)
__device__ __forceinline__ bool fresh_lookup(const FreshDesc *__restrict__ fdesc, const unsigned long long *__restrict__ ftbits,
                                             uint64_t bx, uint64_t &S, uint32_t &rem, bool ready_known)
{
    const uint64_t slot = bx % FR_RING;
    const FreshDesc *fd = fdesc + slot;
    if (ready_known) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (__hip_atomic_load(&fd->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != bx + 1) return false;
    const unsigned long long G = fd->G;
    const uint32_t cnt = fd->cnt, i_in = fd->i_in, w_pos = fd->w_pos, w_neg = fd->w_neg;
    const int64_t g = (int64_t)S - (int64_t)G;
    const bool neg = g < 0;
    const uint64_t idx = (uint64_t)(neg ? -g : g);
    if (idx > (neg ? w_neg : w_pos)) return false;
    uint32_t T = 0;
    if (idx) {
        const uint32_t lane = threadIdx.x & 63;
        uint32_t c = 0;
        if (lane < FR_WORDS && (uint64_t)lane * 64u < idx) {
            const unsigned long long wd = ftbits[(slot * 2 + (neg ? 1 : 0)) * FR_WORDS + lane];
            const uint32_t d = (uint32_t)idx - lane * 64u;
            c = (uint32_t)__popcll(wd & sc_low_mask64(d < 64u ? d : 64u));
        }
        T = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(c), 63);
    }
    S = G + cnt + (unsigned long long)(neg ? -(long long)T : (long long)T);
    rem = i_in - (uint32_t)(S - G);
    return true;
}

// Chain the exact states through blocks [b0, b1) (one workgroup): a SEGMENT of prepared blocks costs one lookup in
// its (composed) table, which every thread evaluates for itself from LDS; the other blocks the full in-block fixed
// point.  While a block is computed, the draws of the next block to compute and the tables of the run of segments
// before it are already on their way (registers, then LDS).  Leaves sblk[b] for every block it computed and for the
// first block of every segment (k_seg_fill adds the blocks inside), hardmask[b], segmode[b] and the accept masks /
// entering counts of the blocks it computed itself.  fault != 0 (testing): corrupt one lookup.
__global__ __launch_bounds__(SCAN_THREADS) void k_chain(const uint32_t *__restrict__ raw, uint64_t n_blocks,
                                                        uint32_t n, uint64_t total_steps, uint64_t B0, uint64_t B1,
                                                        uint64_t S_need, const PhiDesc *__restrict__ desc,
                                                        const unsigned long long *__restrict__ tbits,
                                                        const PhiSeg *__restrict__ seg,
                                                        const unsigned long long *__restrict__ ctbits,
                                                        uint8_t *__restrict__ hardmask, uint8_t *__restrict__ segmode, int fault,
                                                        bits_t *__restrict__ acc_bits, uint32_t *__restrict__ enter,
                                                        unsigned long long *__restrict__ sblk,
                                                        unsigned long long *__restrict__ st, uint32_t *__restrict__ flags,
                                                        uint32_t unit0, FreshCtl *__restrict__ fctl /* + descriptors + tables */,
                                                        uint32_t launch_id)
{
    __shared__ BlockShared sh;
#if PHI_TAIL_LDS
    __shared__ TailShared tsh;
    TailShared *const tshp = &tsh;
#else
    TailShared *const tshp = nullptr;
#endif
    __shared__ uint32_t shReady;
    __shared__ __align__(8) PhiSeg sg[PHI_UNIT];
    __shared__ uint16_t nxt[PHI_UNIT + 2];  // first block >= i (relative to b0) the chain computes itself
    __shared__ ulonglong2 tl[PHI_NS * 2 * PHI_STAGE_PIECES];   // [staged segment][side][64 x 16 B]
    __shared__ uint32_t tpre[PHI_NS * 2 * PHI_STAGE_PIECES];   // set bits in front of each 16-byte piece of its row
    static_assert(PHI_NS * 2 * PHI_STAGE_PIECES <= SCAN_THREADS, "one 16-byte piece per thread");
    __shared__ unsigned long long shS;
    __shared__ uint32_t shRem, shRel;
    const uint32_t tau = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tau >> 6));
    const uint32_t M = n - 1, top_mask = mask_of(M);
    uint64_t S = st[0];
    if (S >= total_steps || st[1] != B0 || B1 > n_blocks) {  // job complete, or an earlier launch gave up (uniform)
        if (tau == 0) {
            __hip_atomic_store(flags, 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // no gate waits for us
            if (fctl) __hip_atomic_store(&fctl->done, launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // nor k_fresh
        }
        return;
    }
    // the exact state S_ at the boundary IN FRONT OF block bx_, the first one seen with at most FR_POST_REM steps of the
    // current permutation left: the reference for the fresh tables of the NEXT permutation's end (k_fresh has a whole
    // permutation of chain time, ~100 us, to build them; the uncertainty grows only with the square root of the distance)
#define PHI_POST(bx_, S_)                                                                                     \
    if (fctl && tau == 0 && !cn[6]) {                                                                         \
        cn[6] = 1;                                                                                            \
        const uint32_t ps_ = ++cn[5];                                                                         \
        fctl->post_b[ps_ & 1u] = (bx_);                                                                       \
        fctl->post_S[ps_ & 1u] = (S_);                                                                        \
        fctl->post_t[ps_ & 1u] = (unsigned long long)wall_clock64();                                          \
        __hip_atomic_store(&fctl->seq, ps_, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);                      \
    }
    uint32_t parity = 0;
    int failed = 0;
    __shared__ unsigned long long shEnd;   // (raw position of the job's last step, seen by at most one thread of one launch)
    if (tau == 0) shEnd = 0;
    // counters of thread 0 live in LDS (the kernel sits at its 128-VGPR cap): [0] blocks by lookup, [1] computed, [2] segment
    // lookups, [3] slow paths, [4] blocks by fresh table, [5] posts to k_fresh
    __shared__ uint32_t cn[7];   // ... [6] the current permutation has been posted
    if (tau < 7) cn[tau] = tau == 5 && fctl ? fctl->seq : 0u;
    __syncthreads();
#ifdef PHI_PROFILE
    unsigned long long pf_easy = 0, pf_hard = 0;
#endif
    uint32_t rem = M - (uint32_t)(S % M);  // steps left in the current permutation, carried along from here
    uint64_t b_next = B0;
    // one launch chains several launch units (each prepared by its own launches; the host waited for all of them)
    uint32_t unit = unit0;
    int gave_up = 0;
    const ulonglong2 *tb2 = reinterpret_cast<const ulonglong2 *>(tbits), *ctb2 = reinterpret_cast<const ulonglong2 *>(ctbits);
    for (uint64_t b0 = B0; b0 < B1 && !failed && S < total_steps; b0 += PHI_UNIT, ++unit) {
    const uint64_t b1 = b0 + PHI_UNIT < B1 ? b0 + PHI_UNIT : B1;
    const uint32_t nb = (uint32_t)(b1 - b0);
    __syncthreads();  // the previous unit's readers of sg / nxt / tl are done
#ifdef PHI_PROFILE
    const long long pf_w0 = clock64();
#endif
    if (tau == 0) shReady = (uint32_t)phi_wait_at_least(flags + 1 + unit % PHI_FLAG_SLOTS, unit + 1, st);
    __syncthreads();
#ifdef PHI_PROFILE
    if (tau == 0) {   // waiting for the unit's preparation: total, units that waited > 20 us, longest wait
        const unsigned long long wt = (unsigned long long)(clock64() - pf_w0);
        atomicAdd(&g_phi_prof[20], wt);
        if (wt > 40000ull) {
            const unsigned long long k = atomicAdd(&g_phi_prof[30], 1ull);
            atomicAdd(&g_phi_prof[19], wt);
            if (k < 128) { g_phi_prof[32 + 2 * k] = (unsigned long long)unit | ((unsigned long long)unit0 << 32); g_phi_prof[33 + 2 * k] = wt; }
        }
        atomicMax(&g_phi_prof[31], wt);
    }
#endif
    if (shReady) { gave_up = (int)shReady; break; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the unit's descriptors and tables, written by other kernels
    if (tau < nb) sg[tau] = seg[(b0 + tau) % PHI_RING];
    __syncthreads();
    if (tau <= nb) {
        uint32_t j = tau;
        while (j < nb && sg[j].kind != 0) ++j;
        nxt[tau] = (uint16_t)j;
    }
    __syncthreads();
    // The tables of up to PHI_NS consecutive segments that start at relative block `first` (a run ends at the next block
    // the chain computes): thread = (staged segment q, side, piece) loads one 16-byte piece into treg.
#define PHI_STAGE_LOAD(first)                                                                              \
    {                                                                                                      \
        treg = make_ulonglong2(0ull, 0ull);                                                                \
        if (tau < PHI_NS * 2 * PHI_STAGE_PIECES) {                                                         \
            const uint32_t q = tau / (2 * PHI_STAGE_PIECES), sd = (tau / PHI_STAGE_PIECES) & 1u;           \
            uint32_t pos = (first);                                                                        \
            for (uint32_t k = 0; k < q && pos < nb && sg[pos].kind == 1; ++k) pos += sg[pos].len;          \
            if (pos < nb && sg[pos].kind == 1) {                                                           \
                const uint64_t slot = (b0 + pos) % PHI_RING;                                               \
                treg = (sg[pos].own ? tb2 : ctb2)[((slot * 2 + sd) * PHI_WORDS) / 2 + (tau % PHI_STAGE_PIECES)]; \
            }                                                                                              \
        }                                                                                                  \
    }
    ulonglong2 treg;
    uint32_t un[SCAN_D];
    uint32_t rel = 0, h = nxt[0];   // rel: next block to resolve; h: the next block the chain computes itself (>= rel)
    // the "ready" word (low half: block + 1) of block h's fresh table, asked for as soon as h is known -- one item (a
    // fixed point, microseconds) before it is looked at: a table that is not there then costs nothing
#if PHI_FRESH_TABLES
#define PHI_FRESH_ASK(hh)                                                                                                    \
    fr_ready = (fctl && (hh) < nb) ? __hip_atomic_load(reinterpret_cast<const uint32_t *>(                                   \
                   &reinterpret_cast<const FreshDesc *>(fctl + 1)[(b0 + (hh)) % FR_RING].ready), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t fr_ready;
#else
#define PHI_FRESH_ASK(hh)
#endif
    PHI_FRESH_ASK(h)
    PHI_STAGE_LOAD(rel)
    if (h < nb) scan_load(raw, (b0 + h) * SCAN_BLOCK, tau, un);
    for (;;) {
        // (only wavefront 0 reads the staged tables, and it is behind the barrier that follows its lookups: no barrier here)
        {   // a wavefront's 64 pieces are one (segment, side) row
            const uint32_t ones = (uint32_t)(__popcll(treg.x) + __popcll(treg.y));
            const uint32_t upto = wave_inclusive_scan(ones);
            if (tau < PHI_NS * 2 * PHI_STAGE_PIECES) { tl[tau] = treg; tpre[tau] = upto - ones; }
        }
        __syncthreads();
#ifdef PHI_PROFILE
        const long long pf_t0 = clock64();
#endif
        // ---- the staged segments, one lookup each, by wavefront 0 (sixteen wavefronts doing the same ~60 dependent
        // instructions take turns on the four SIMDs: four times the clocks of one) ----
        if (wave == 0) {
            bool miss0 = false;
            for (uint32_t q = 0; q < PHI_NS && rel < nb && sg[rel].kind == 1; ++q) {
                const PhiSeg sq = sg[rel];
                const int64_t g = (int64_t)S - (int64_t)sq.G;
                const bool neg = g < 0;
                const uint64_t idx = (uint64_t)(neg ? -g : g);
                if (sq.bad || idx > (neg ? sq.vneg : sq.vpos)) { miss0 = true; break; }   // outside the segment's window
                uint32_t T = 0;
                if (idx) {
                    if (idx <= 128u * PHI_STAGE_PIECES) {
                        const uint32_t piece = ((uint32_t)idx - 1u) >> 7, nbit = (uint32_t)idx - 128u * piece;
                        const uint32_t row = (q * 2 + (neg ? 1u : 0u)) * PHI_STAGE_PIECES + piece;
                        T = tpre[row] + phi_piece_rank(tl[row], nbit);
                    } else {   // beyond the staged bits (|gap| > 8192: ~3 sigma of the widest window)
                        const uint64_t slot = (b0 + rel) % PHI_RING;
                        T = phi_lookup((sq.own ? tbits : ctbits) + (slot * 2 + (neg ? 1 : 0)) * PHI_WORDS, (uint32_t)idx);
                    }
                }
                if (tau == 0) { sblk[b0 + rel] = S; segmode[b0 + rel] = 1; }
                int64_t e = (int64_t)sq.exit0 + (neg ? -(int64_t)T : (int64_t)T);
                if (fault && cn[0] == 0) e += 1;  // testing: the verification must catch this
                S = sq.G + (unsigned long long)e;
                rem = sq.i_in - (uint32_t)e;       // no trajectory of the window leaves G's permutation
                if (tau == 0) { cn[0] += sq.len; ++cn[2]; }
                rel += sq.len;
            }
            // ... and behind them a block the chain would compute itself: at the end of a permutation a FRESH table may
            // cover it (k_fresh; wavefront 0 alone evaluates it: the lookup's registers are not live in the fixed point)
#if PHI_FRESH_TABLES
            while (fctl && !miss0 && rel < nb && sg[rel].kind == 0 && rem <= FR_REM_MAX && (rel != h || fr_ready == (uint32_t)(b0 + rel + 1))) {
                const uint64_t S_in = S;
                if (tau == 0) {   // diagnostics: lookups tried, ticks since the last post, how far its block is from the slot's
                    atomicAdd(&fctl->diag[6], 1ull);
                    atomicAdd(&fctl->diag[7], (unsigned long long)wall_clock64() - fctl->post_t[cn[5] & 1u]);
                    const unsigned long long rd = reinterpret_cast<const FreshDesc *>(fctl + 1)[(b0 + rel) % FR_RING].ready;
                    if (rd == b0 + rel + 1) atomicAdd(&fctl->diag[8], 1ull);   // a table for exactly this block exists
                }
                if (!fresh_lookup(reinterpret_cast<const FreshDesc *>(fctl + 1),
                                  reinterpret_cast<const unsigned long long *>(reinterpret_cast<const FreshDesc *>(fctl + 1) + FR_RING),
                                  b0 + rel, S, rem, rel == h)) break;
                if (tau == 0) { sblk[b0 + rel] = S_in; hardmask[b0 + rel] = 0; ++cn[0]; ++cn[4]; }
                ++rel;
            }
#endif
            if (tau == 0) { shS = S; shRem = rem; shRel = rel | (miss0 ? 0x80000000u : 0u); }
        }
        __syncthreads();
        S = shS;
        rem = shRem;
        rel = shRel & 0x7fffffffu;
        const bool miss = (shRel >> 31) != 0;
#if PHI_FRESH_TABLES
        if (rem <= FR_POST_REM) PHI_POST(b0 + rel, S)
#endif
        if (rel > h) {   // a fresh table resolved the block whose draws were prefetched: the next one to compute, then
            h = nxt[rel];
            if (h < nb) scan_load(raw, (b0 + h) * SCAN_BLOCK, tau, un);
            PHI_FRESH_ASK(h)
        }
#ifdef PHI_PROFILE
        const long long pf_ts = clock64();
#endif
        if (miss) {
            // The entry state lies outside the segment's window (a band edge narrowed it, or the guess was far off):
            // its blocks one by one -- the per-block tables from global memory where they cover the state, the fixed
            // point where they do not.  Rare (about every other permutation at 1M cells), and no slower than r03's path.
            const uint32_t first = rel, last = rel + sg[rel].len;
            for (; rel < last && !failed && S < total_steps; ++rel) {
                const uint64_t bx = b0 + rel, slot = bx % PHI_RING;
                const PhiDesc d = desc[slot];
                const int64_t g = (int64_t)S - (int64_t)d.G;
                const bool neg = g < 0;
                const uint64_t idx = (uint64_t)(neg ? -g : g);
                if (d.prepared && idx <= (neg ? d.w_neg : d.w_pos)) {
                    const uint32_t T = idx ? phi_lookup(tbits + (slot * 2 + (neg ? 1 : 0)) * PHI_WORDS, (uint32_t)idx) : 0u;
                    if (tau == 0) { sblk[bx] = S; hardmask[bx] = 0; }
                    S = d.G + d.cnt + (unsigned long long)(neg ? -(long long)T : (long long)T);
                    rem = d.i_in - (uint32_t)(S - d.G);
                    if (tau == 0) ++cn[0];
                } else {
                    // (a fresh table is not consulted here: a second inlined copy of the lookup made the kernel spill)
                    uint32_t u[SCAN_D];
                    scan_load(raw, bx * SCAN_BLOCK, tau, u);
                    ScanRes r;
                    uint32_t excl, total_cnt;
                    if (block_fixed_point(u, S, rem, M, top_mask, total_steps, sh, parity, r, excl, total_cnt) > 0) { failed = 1; break; }
                    acc_bits[bx * SCAN_THREADS + tau] = r.bits;
                    enter[bx * SCAN_THREADS + tau] = excl;
                    if (tau == 0) { sblk[bx] = S; hardmask[bx] = 1; }
                    if (r.end) shEnd = bx * SCAN_BLOCK + (uint64_t)tau * SCAN_D + r.end;
                    S += total_cnt;
                    const uint32_t rem_new = rem_advance(rem, total_cnt, M);
#if PHI_FRESH_TABLES
                    if (rem_new > rem && tau == 0) cn[6] = 0;   // a new permutation: not posted yet
#endif
                    rem = rem_new;
                    if (tau == 0) ++cn[1];
                }
            }
            if (tau == 0) { segmode[b0 + first] = 2; ++cn[3]; }   // k_seg_fill has nothing to add here
#ifdef PHI_PROFILE
            if (tau == 0) { atomicAdd(&g_phi_prof[21], (unsigned long long)(clock64() - pf_ts)); atomicAdd(&g_phi_prof[22], 1ull); }
#endif
            if (failed || S >= total_steps) break;
        }
#ifdef PHI_PROFILE
        const long long pf_t1 = clock64();
        pf_easy += (unsigned long long)(pf_t1 - pf_t0);
#endif
        if (rel >= nb) { rel = nb; break; }
        if (sg[rel].kind == 1) {   // the run goes on (more segments than staged at once, or behind a slow path)
            PHI_STAGE_LOAD(rel)
#ifdef PHI_PROFILE
            if (tau == 0) atomicAdd(&g_phi_prof[23], 1ull);   // exposed table loads
#endif
            continue;
        }
        // ---- block rel: computed by the chain itself ----
        const uint32_t x = rel;
        const uint32_t hN = nxt[x + 1];
        uint32_t u[SCAN_D];
        if (x == h) {
#pragma unroll
            for (int q = 0; q < SCAN_D; ++q) u[q] = un[q];
        } else {
            scan_load(raw, (b0 + x) * SCAN_BLOCK, tau, u);
        }
        // on their way while block x is computed: the tables of the run behind it and the draws of the block after that
        PHI_STAGE_LOAD(x + 1)
        if (x == h && hN < nb) scan_load(raw, (b0 + hN) * SCAN_BLOCK, tau, un);
        PHI_FRESH_ASK(hN)     // (for the block after this one; this block's word has been used)
        ScanRes r;
        uint32_t excl, total_cnt;
#ifdef PHI_PROFILE
        const uint32_t pf_rem = rem;
        int pf_rounds = 0;
        if (block_fixed_point<PHI_TAIL>(u, S, rem, M, top_mask, total_steps, sh, parity, r, excl, total_cnt, &pf_rounds, tshp) > 0) { failed = 1; break; }
        if (tau == 0) {   // computed blocks by the steps left in their permutation: count, clocks of the fixed point, rounds
            const int cls = pf_rem > 98304u ? 0 : pf_rem > 49152u ? 1 : pf_rem > 24576u ? 2 : pf_rem > 12288u ? 3 : 4;
            atomicAdd(&g_phi_prof[cls * 4], 1ull);
            atomicAdd(&g_phi_prof[cls * 4 + 1], (unsigned long long)(clock64() - pf_t1));
            atomicAdd(&g_phi_prof[cls * 4 + 2], (unsigned long long)pf_rounds);
        }
#else
        if (block_fixed_point<PHI_TAIL>(u, S, rem, M, top_mask, total_steps, sh, parity, r, excl, total_cnt, nullptr, tshp) > 0) { failed = 1; break; }
#endif
        const uint64_t bx = b0 + x;
        acc_bits[bx * SCAN_THREADS + tau] = r.bits;
        enter[bx * SCAN_THREADS + tau] = excl;
        if (tau == 0) { sblk[bx] = S; hardmask[bx] = 1; }
        if (r.end) shEnd = bx * SCAN_BLOCK + (uint64_t)tau * SCAN_D + r.end;
        S += total_cnt;
        {
            const uint32_t rem_new = rem_advance(rem, total_cnt, M);
#if PHI_FRESH_TABLES
            if (rem_new > rem && tau == 0) cn[6] = 0;   // a new permutation: not posted yet
#endif
            rem = rem_new;
        }
        if (tau == 0) ++cn[1];
#ifdef PHI_PROFILE
        pf_hard += (unsigned long long)(clock64() - pf_t1);
#endif
        rel = x + 1;
        h = hN;
        if (S >= total_steps) break;
    }
    b_next = b0 + rel;
    if (tau == 0) {  // entry state of the next unit (the reference of a later unit's guesses), then "unit done"
        sblk[b_next] = S;
        __hip_atomic_store(flags, unit + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    }  // units
#undef PHI_STAGE_LOAD
#undef PHI_FRESH_ASK
#undef PHI_POST
    __syncthreads();
    if (tau == 0) {
        if (shEnd) st[3] = shEnd;
        const uint64_t b = b_next;
        st[0] = S;
        st[1] = b;
#ifdef PHI_PROFILE
        st[6] += (pf_easy >> 6) | ((pf_hard >> 6) << 32);   // clocks / 64 of thread 0: lookup phases, computed blocks
#else
        st[6] += cn[2];   // segment lookups
#endif
        st[7] += (unsigned long long)cn[3] | ((unsigned long long)cn[4] << 32);  // segments whose window missed the entry state | blocks resolved by a fresh table
        st[4] += cn[0];
        st[5] += cn[1];
        sblk[b] = S;  // entry state of the next block (sblk holds n_blocks + 1 entries)
        unsigned long long f = 0;  // (k_block_exact of the previous chunk may be raising its own flag right now)
        if (failed) f |= 1ull;
        if (gave_up == 1) f |= 8ull;  // a unit's preparation did not arrive in time
        if (S < S_need && S < total_steps) f |= 2ull;  // the blocks granted to this chunk did not complete it
        if (f) atomicOr(st + 2, f);
        if (f || gave_up || S >= total_steps)  // nothing more will come from the chain: release every gate
            __hip_atomic_store(flags, 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (fctl) __hip_atomic_store(&fctl->done, launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // k_fresh of this launch ends
    }
}

// Recompute every prepared block of range [range[0], range[1]) from its exact entry state (whole chip) and
// verify the chain: S_b + count_b must be the entry state of block b + 1.
__global__ __launch_bounds__(SCAN_THREADS) void k_block_exact(const uint32_t *__restrict__ raw, uint32_t n,
                                                              uint64_t total_steps,
                                                              const unsigned long long *__restrict__ range,
                                                              const uint8_t *__restrict__ hardmask,
                                                              bits_t *__restrict__ acc_bits,
                                                              uint32_t *__restrict__ enter,
                                                              const unsigned long long *__restrict__ sblk,
                                                              unsigned long long *__restrict__ st)
{
    __shared__ BlockShared sh;
    __shared__ uint32_t shrem;
    const uint64_t b = range[0] + blockIdx.x;
    if (b >= range[1] || hardmask[b]) return;
    const uint32_t tau = threadIdx.x;
    const uint32_t M = n - 1, top_mask = mask_of(M);
    const uint64_t S = sblk[b];
    if (tau == 0) shrem = M - (uint32_t)(S % M);
    uint32_t u[SCAN_D];
    scan_load(raw, b * SCAN_BLOCK, tau, u);
    __syncthreads();
    ScanRes r;
    uint32_t excl, total_cnt, parity = 0;
    const int failed = block_fixed_point(u, S, shrem, M, top_mask, total_steps, sh, parity, r, excl, total_cnt) > 0;
    acc_bits[b * SCAN_THREADS + tau] = r.bits;
    enter[b * SCAN_THREADS + tau] = excl;
    if (tau == 0 && (failed || r.end || S + total_cnt != sblk[b + 1])) atomicOr(st + 2, 4ull);
}

// ------------------------------------------------------------------------------------------------
// B: apply the swaps, one wavefront per permutation
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_apply_swaps(const int32_t *__restrict__ J, int32_t *__restrict__ perm,
                                                    int64_t pstride, uint32_t n, int64_t p0, int64_t n_perm)
{
    const int64_t p = p0 + blockIdx.x;
    if (p >= n_perm) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t M = n - 1;
    int32_t *A = perm + p * pstride;
    const int32_t *Jp = J + p * (int64_t)M;
    for (uint32_t x = lane; x < n; x += 64) A[x] = (int32_t)x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int64_t i_top = (int64_t)n - 1;
    while (i_top >= 1) {
        const int64_t i = i_top - lane;
        const bool valid = i >= 1;
        // step index inside the permutation: s = n-1-i (lanes read consecutive entries)
        int32_t j = valid ? Jp[(int64_t)M - i] : -1;
        if (valid && (uint32_t)j > (uint32_t)i) j = (int32_t)i;  // never index outside [0, i], whatever J holds
        const int32_t ii = valid ? (int32_t)i : -2;
        // loads first (latency overlaps the conflict search); L1 is bypassed so that the values the
        // previous round stored (write-through to L2, completed by the vmcnt wait) are seen
        int32_t a_i = 0, a_j = 0;
        if (valid) {
            a_i = __hip_atomic_load(&A[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a_j = __hip_atomic_load(&A[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lane l conflicts if an EARLIER step m < l targets l's own slot (j_m == i_l) or the same slot
        // (j_m == j_l); (j_l == i_m cannot happen: j_l <= i_l < i_m).  Self swaps j == i are harmless.
        bool flag = false;
        for (int m = 0; m < 63; ++m) {
            const int32_t jm = __builtin_amdgcn_readlane(j, m);
            flag |= ((int)lane > m) && (jm == ii || jm == j);
        }
        const unsigned long long conf = __ballot(flag && valid);
        const unsigned long long vmask = __ballot(valid);
        int count = conf ? (int)__builtin_ctzll(conf) : 64;
        const int nvalid = (int)__builtin_popcountll(vmask);
        if (count > nvalid) count = nvalid;
        if ((int)lane < count) {
            A[i] = a_j;
            if (j != (int32_t)i) A[j] = a_i;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        i_top -= count;
    }
}

// The same rule with a whole workgroup per permutation: SW_T consecutive steps per round.  Two steps of a round
// touch a common slot only if a later step's own slot i_t is an earlier step's target (j_m == i_t, found by index
// arithmetic since the i are consecutive) or two steps share a target (j_m == j_l, found with an LDS hash table
// keyed by the target: CAS insert with linear probing, minimum step index per key).  The longest prefix without
// such a pair is applied in parallel; the round trip to L2 that bounds a round is paid once per ~SW_T steps.
#define SW_T 512
#ifndef SWAPS_WG_MIN_N
#define SWAPS_WG_MIN_N 65536  // shorter permutations: conflicts are frequent, one wavefront per permutation is enough
#endif
#define SW_HASH 2048

// ASC = false: the shuffle itself (steps i = n-1 .. 1), the table numpy returns.
// ASC = true:  the same transpositions applied in the opposite order (i = 1 .. n-1) to the identity.  With position
//              swaps s_1 .. s_m applied in order the array is s_1 o s_2 o .. o s_m (position -> value), so the opposite
//              order yields its inverse: the INVERSE permutation table comes out of the same kernel, no scatter pass.
//              Two steps of a round then collide when a later step's target is an earlier step's own slot
//              (j_l == i_m, again index arithmetic) or two steps share a target.
// r03: (i) the swap partners j of the coming rounds are PREFETCHED into an LDS ring (they do not depend on anything the
// rounds do; only WHICH steps a round holds does, by up to SW_T), so a round's memory latency is one dependent access
// (the values at the partners' slots) instead of two; (ii) in the ascending mode a step's own slot has never been touched
// when its turn comes (every earlier step i' < i writes slots <= i'), so its value is i itself: no load, and no identity
// fill of the row beyond slot 0.  -DSW_NO_PREFETCH: the r02 form (A/B builds).
#define SW_RING 2048   // partners of steps [done, done + <= 1536) live here


// r04: PW permutations per workgroup (PW x SW_T threads, each SW_T-thread half runs its own permutation with its own LDS
// structures, the barriers are shared: a round is latency-bound, two of them in lockstep cost what one costs).  Why: a
// swap workgroup lives ~10 ms, and 128 of them with 8 wavefronts each, spread over the CUs the scoring kernel leaves,
// fragment the wavefront slots that the generator's 1024-thread preparation workgroups need sixteen of on one CU (4.3 of
// the chain's 7-9 ms waits).  With PW = 2 a chunk is 64 workgroups of the preparation kernels' own size.
template <bool ASC, int PW>
__global__ __launch_bounds__(SW_T * PW) void k_apply_swaps_wg(const int32_t *__restrict__ J, int32_t *__restrict__ perm,
                                                              int64_t pstride, uint32_t n, int64_t p0, int64_t n_perm)
{
    __shared__ uint32_t hkey_[PW][SW_HASH], hmin_[PW][SW_HASH];
    __shared__ uint32_t first_conf_[PW][2];
    __shared__ uint32_t act[2];
#ifndef SW_NO_PREFETCH
    __shared__ int32_t jring_[PW][SW_RING];
#endif
    const uint32_t half = PW > 1 ? threadIdx.x / SW_T : 0u;
    const uint32_t l = PW > 1 ? threadIdx.x % SW_T : threadIdx.x;
    uint32_t *hkey = hkey_[half], *hmin = hmin_[half], *first_conf = first_conf_[half];
#ifndef SW_NO_PREFETCH
    int32_t *jring = jring_[half];
#endif
    const int64_t p = p0 + (int64_t)blockIdx.x * PW + half;
    const bool exists = p < n_perm;          // (an odd chunk: the last workgroup's second half has nothing to do but meet the barriers)
    const uint32_t M = n - 1;
    int32_t *A = perm + (exists ? p : p0) * pstride;
    const int32_t *Jp = J + (exists ? p : p0) * (int64_t)M;
#ifndef SW_NO_PREFETCH
    // step k = 0 .. M - 1 of the processing order: i = 1 + k (ascending) or n - 1 - k; its partner is Jp[M - i]
    auto step_i = [&](int64_t k) -> int64_t { return ASC ? 1 + k : (int64_t)n - 1 - k; };
    if (exists) {
        if (!ASC) { for (uint32_t x = l; x < n; x += SW_T) A[x] = (int32_t)x; }
        else if (l == 0) A[0] = 0;
        for (int r = 0; r < 2; ++r) {   // partners of the first 2 SW_T steps
            const int64_t k = (int64_t)r * SW_T + l;
            jring[k & (SW_RING - 1)] = k < (int64_t)M ? Jp[(int64_t)M - step_i(k)] : -1;
        }
    }
    int64_t filled = 2 * SW_T;      // partners of steps [done, filled) are in the ring
#else
    if (exists) for (uint32_t x = l; x < n; x += SW_T) A[x] = (int32_t)x;
#endif
    int64_t i_cur = ASC ? 1 : (int64_t)n - 1;  // first step of the round
    if (!exists) i_cur = ASC ? (int64_t)n : 0; // (done)
    if (l < 2) first_conf[l] = SW_T;
    if (l == 0) act[half] = (ASC ? i_cur <= (int64_t)n - 1 : i_cur >= 1) ? 1u : 0u;
    if (PW == 1 && l == 0) act[1] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    uint32_t round = 0;
    while (act[0] | act[1]) {       // (uniform: the words are rewritten in front of the round's last barrier)
        const int64_t i = ASC ? i_cur + l : i_cur - l;
        const bool valid = exists && (ASC ? (i_cur <= (int64_t)n - 1 && i <= (int64_t)n - 1) : (i_cur >= 1 && i >= 1));
#ifndef SW_NO_PREFETCH
        const int64_t done = ASC ? i_cur - 1 : (int64_t)n - 1 - i_cur;   // steps applied so far
        int32_t j = valid ? jring[(done + l) & (SW_RING - 1)] : -1;
        // the ring's next SW_T partners are on their way while this round works (stored at its end)
        const bool top_up = exists && filled - done <= 2 * SW_T;
        const int64_t kf = filled + l;
        int32_t j_next = -1;
        if (top_up && kf < (int64_t)M) j_next = Jp[(int64_t)M - step_i(kf)];
#else
        int32_t j = valid ? Jp[(int64_t)M - i] : -1;
#endif
        if (valid && (uint32_t)j > (uint32_t)i) j = (int32_t)i;  // never index outside [0, i], whatever J holds
        int32_t a_i = 0, a_j = 0;
        if (valid) {  // L1 is bypassed: the values the previous round stored are in L2 (vmcnt wait + barrier)
#ifndef SW_NO_PREFETCH
            a_i = ASC ? (int32_t)i : __hip_atomic_load(&A[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            a_i = __hip_atomic_load(&A[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            a_j = (ASC && j == (int32_t)i) ? a_i : __hip_atomic_load(&A[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < SW_HASH / SW_T; ++k) { hkey[l + SW_T * k] = 0xffffffffu; hmin[l + SW_T * k] = 0xffffffffu; }
        __syncthreads();
        uint32_t *fc = &first_conf[round & 1];
        uint32_t h = 0;
        if (valid) {
            if (ASC) {
                const int64_t t = j - i_cur;  // the step whose own slot is my target (t <= l; t == l is a self swap)
                if (t >= 0 && t < (int64_t)l) atomicMin(fc, l);
            } else {
                const int64_t t = i_cur - j;  // the step whose own slot is j (t >= l; t == l is a self swap)
                if (t < SW_T && t != (int64_t)l) atomicMin(fc, (uint32_t)t);
            }
            h = ((uint32_t)j * 2654435761u) >> 21;
            for (;;) {
                const uint32_t old = atomicCAS(&hkey[h], 0xffffffffu, (uint32_t)j);
                if (old == 0xffffffffu || old == (uint32_t)j) break;
                h = (h + 1) & (SW_HASH - 1);
            }
            atomicMin(&hmin[h], l);
        }
        __syncthreads();
        if (valid && hmin[h] < l) atomicMin(fc, l);
        if (l == 0) first_conf[(round + 1) & 1] = SW_T;  // next round's cell (nobody touches it this round)
        __syncthreads();
        uint32_t count = *fc;
        const int64_t left = ASC ? (int64_t)n - i_cur : i_cur;  // steps not yet applied (<= 0: this half is done)
        const int64_t nvalid = left < 0 ? 0 : (left < SW_T ? left : SW_T);
        if ((int64_t)count > nvalid) count = (uint32_t)nvalid;
        if (valid && l < count) {
            A[i] = a_j;
            if (j != (int32_t)i) A[j] = a_i;
        }
#ifndef SW_NO_PREFETCH
        if (top_up) { jring[kf & (SW_RING - 1)] = j_next; filled += SW_T; }   // (uniform per half; slots of steps already applied)
#endif
        i_cur += ASC ? (int64_t)count : -(int64_t)count;
        ++round;
        if (l == 0) act[half] = (exists && (ASC ? i_cur <= (int64_t)n - 1 : i_cur >= 1)) ? 1u : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// r04: FULL rounds.  The kernel above ends a round at the first step that shares a slot with an earlier step of the round
// (the birthday bound: ~0.9 sqrt(i) steps, 470 of 512 at i = 10^6, 2670 rounds per 10^6-step permutation, each a round trip
// to L2 and four barriers: 10-12 ms per chunk, and the scoring of a chunk waits for exactly that).  But a round's hazards
// all run through the PARTNER slots of earlier steps, and the hash table that finds them can also resolve them:
//   descending (the shuffle): step k reads its own slot i_k and its partner slot j_k.  An earlier step a of the round can
//     have touched either one only as ITS partner (j_a == i_k or j_a == j_k: own slots of earlier steps lie above i_k), and
//     what it left there is the value v_a its own slot held.  So v_k = v_a of the latest such a for i_k (else memory), the
//     value that ends up in slot i_k is v_a of the latest such a for j_k (else memory), and slot j_k ends up with v of the
//     LAST step of the round that has it as partner -- unless it is a processed step's own slot (written by that step).
//   ascending (the inverse table): a step's own slot is untouched (v_k = i_k); its partner slot may have been touched by
//     an earlier step as partner (leaving that step's i_a) or as own slot (leaving w_a, what that step took from ITS
//     partner slot); slot i_k ends up with i_b of the last LATER step that has it as partner, else with w_k.
// Per key (slot) the table keeps the smallest and the largest step index: enough while no key has three steps below the
// round's end, so a round ends at the first MIDDLE step of a key (~i^(2/3) steps: every round of 1024 is whole down to
// i ~ 30 000).  Chains (v_k = v_a = v_a' ...) are rare and resolved by pointer jumping in LDS.  1054 rounds per 10^6-step
// permutation instead of 2670 (simulation and rule: scripts/swap_rounds_sim.py); sixteen wavefronts per workgroup, the
// size of the generator's preparation workgroups.
#define SF_T 1024
#define SF_HASH 8192      // eight slots per step: a CAS insert seldom probes twice (at two slots per step the slowest wavefront
                          // of sixteen probed ~10 times, 6000 clocks per round); 96 KB of the CU's 160 KB LDS, cleared entry by entry
#define SF_HASH_SHIFT 19
#define SF_RING 4096
#define SF_NONE 0xffffffffu
// A barrier that orders LDS only: __syncthreads() carries a global-memory fence, i.e. a wait for every load in flight,
// and the point of a round is that the hash work runs UNDER the latency of the round's loads (measured, clocks per round
// with __syncthreads(): loads + first barrier 3800, insert 2600, detect + look-up 6000, values 2000, stores 1800).
__device__ __forceinline__ void sf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool ASC>
__global__ __launch_bounds__(SF_T) void k_apply_swaps_full(const int32_t *__restrict__ J, int32_t *__restrict__ perm,
                                                           int64_t pstride, uint32_t n, int64_t p0, int64_t n_perm)
{
    __shared__ uint32_t hkey[SF_HASH], hmin[SF_HASH], hmax[SF_HASH];   // key (slot) | smallest step | 1 + largest step
    __shared__ uint32_t omin[SF_T], omax[SF_T];   // the same two for the round's OWN slots as somebody's partner, by step (no probing)
    __shared__ int32_t jring[SF_RING];
    __shared__ int32_t val[SF_T];       // v_k (descending) / w_k (ascending) once ptr[k] == SF_NONE
    __shared__ uint32_t ptr[SF_T];      // the step whose value step k takes
    __shared__ uint32_t first_conf[2], chains[2];
    const uint32_t l = threadIdx.x;
    const int64_t p = p0 + blockIdx.x;
    if (p >= n_perm) return;
    const uint32_t M = n - 1;           // steps; step s = 0 .. M - 1 handles i = 1 + s (ascending) or n - 1 - s
    int32_t *A = perm + p * pstride;
    const int32_t *Jp = J + p * (int64_t)M;
    // partner of step s: Jp[M - i]
    auto partner_at = [&](uint32_t s) -> int32_t { return Jp[ASC ? M - 1u - s : s]; };
    if (!ASC) { for (uint32_t x = l; x < n; x += SF_T) A[x] = (int32_t)x; }
    else if (l == 0) A[0] = 0;
    for (uint32_t r = 0; r < 2; ++r) {
        const uint32_t s = r * SF_T + l;
        jring[s & (SF_RING - 1)] = s < M ? partner_at(s) : -1;
    }
#pragma unroll
    for (int k = 0; k < SF_HASH / SF_T; ++k) { hkey[l + SF_T * k] = SF_NONE; hmin[l + SF_T * k] = SF_NONE; hmax[l + SF_T * k] = 0u; }
    omin[l] = SF_NONE; omax[l] = 0u;
    uint32_t filled = 2 * SF_T;          // partners of steps [done, filled) are in the ring
    uint32_t done = 0;                  // steps applied so far
    if (l < 2) { first_conf[l] = SF_T; chains[l] = 0u; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // the processed steps of a key below k / below the round's end (see above: at most hmin and hmax)
    auto last_lt = [](uint32_t mn, uint32_t mx1, uint32_t k) -> uint32_t {
        return (mx1 != 0u && mx1 - 1u < k) ? mx1 - 1u : (mn < k ? mn : SF_NONE);
    };
#ifdef SF_PROFILE
    long long pf[6] = {0, 0, 0, 0, 0, 0};
#define SF_MARK(x) { const long long t_ = clock64(); pf[x] += t_ - pf_t; pf_t = t_; }
#else
#define SF_MARK(x)
#endif
    uint32_t round = 0;
    while (done < M) {
#ifdef SF_PROFILE
        long long pf_t = clock64();
#endif
        const uint32_t i_cur = ASC ? 1u + done : n - 1u - done;   // the round's first step
        const uint32_t left = M - done;
        const uint32_t nvalid = left < SF_T ? left : SF_T;
        const bool valid = l < nvalid;
        const uint32_t i = ASC ? i_cur + l : i_cur - l;           // (meaningful if valid)
        int32_t j = valid ? jring[(done + l) & (SF_RING - 1)] : -1;
        const bool top_up = filled - done <= 2 * SF_T;
        const uint32_t sf = filled + l;
        int32_t j_next = -1;
        if (top_up && sf < M) j_next = partner_at(sf);
        if (valid && (uint32_t)j > i) j = (int32_t)i;  // never index outside [0, i], whatever J holds
        int32_t a_i = (int32_t)i, a_j = 0;
        if (valid) {  // L1 is bypassed: the values the previous round stored are in L2 (vmcnt wait + barrier)
            if (!ASC) a_i = __hip_atomic_load(&A[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a_j = __hip_atomic_load(&A[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ascending: may be a slot nobody has written yet; not used then)
        }
        // ---- insert: the tables were cleared behind the previous round's last reads ----
        uint32_t *fc = &first_conf[round & 1];
        uint32_t h = 0;
        if (valid) {
            h = ((uint32_t)j * 2654435761u) >> SF_HASH_SHIFT;
            for (;;) {
                const uint32_t old = atomicCAS(&hkey[h], SF_NONE, (uint32_t)j);
                if (old == SF_NONE || old == (uint32_t)j) break;
                h = (h + 1) & (SF_HASH - 1);
            }
            atomicMin(&hmin[h], l);
            atomicMax(&hmax[h], l + 1u);
            const uint32_t t = ASC ? (uint32_t)j - i_cur : i_cur - (uint32_t)j;   // the step whose own slot is j (if < SF_T)
            if (t < SF_T) { atomicMin(&omin[t], l); atomicMax(&omax[t], l + 1u); }
        }
        sf_lds_barrier();
        SF_MARK(0)
        // ---- hazards: middle steps end the round; every step finds where its two values come from ----
        uint32_t mn = SF_NONE, mx1 = 0u, imn = SF_NONE, imx1 = 0u;   // of the key j / of the key i (my own slot as somebody's partner)
        uint32_t myptr = SF_NONE, p2 = SF_NONE;
        if (valid) {
            mn = hmin[h]; mx1 = hmax[h];
            imn = omin[l]; imx1 = omax[l];
            if (mn < l && l + 1u < mx1) atomicMin(fc, l);   // a middle step of its key
            int32_t v = a_i;
            if (!ASC) {
                const uint32_t p1 = last_lt(imn, imx1, l);
                p2 = j == (int32_t)i ? p1 : last_lt(mn, mx1, l);
                myptr = p1;
            } else if (j != (int32_t)i) {
                const uint32_t a_p = last_lt(mn, mx1, l);
                const uint32_t t = (uint32_t)j - i_cur;       // the step whose own slot is j (wraps to a large number below i_cur)
                const bool own = t < l;
                if (a_p == SF_NONE && !own) v = a_j;
                else if (a_p != SF_NONE && (!own || a_p >= t)) v = (int32_t)(i_cur + a_p);
                else myptr = t;
            }
            val[l] = v;
            ptr[l] = myptr;
            if (myptr != SF_NONE) chains[round & 1] = 1u;
        }
        if (l == 0) { first_conf[(round + 1) & 1] = SF_T; chains[(round + 1) & 1] = 0u; }
        sf_lds_barrier();
        SF_MARK(1)
        uint32_t count = *fc;
        if (count > nvalid) count = nvalid;
        // chains: a step takes the value of an earlier one, which may itself be waiting (rare; usually no pointer at all;
        // a pointer of a step beyond the round's end is resolved too, harmlessly)
        if (chains[round & 1]) {
            for (;;) {
                int32_t got = 0;
                bool ok = false;
                if (myptr != SF_NONE && ptr[myptr] == SF_NONE) { got = val[myptr]; ok = true; }
                const int pending = __syncthreads_or(myptr != SF_NONE && !ok);   // (all reads of the iteration are done)
                if (ok) { val[l] = got; ptr[l] = SF_NONE; myptr = SF_NONE; }
                __syncthreads();
                if (!pending) break;
            }
        }
        SF_MARK(2)
        if (valid && l < count) {
            const uint32_t my_last = (mx1 != 0u && mx1 - 1u < count) ? mx1 - 1u : (mn < count ? mn : SF_NONE);   // last processed step of key j
            if (!ASC) {
                A[i] = p2 == SF_NONE ? a_j : val[p2];
                if (j != (int32_t)i && (uint32_t)j + count <= i_cur && my_last == l) A[j] = val[l];
            } else {
                const uint32_t b = (imx1 != 0u && imx1 - 1u < count) ? imx1 - 1u : (imn < count ? imn : SF_NONE);   // last processed step with partner i
                if (!(b != SF_NONE && b > l)) A[i] = val[l];
                if (j != (int32_t)i && my_last == l) A[j] = (int32_t)i;
            }
        }
        sf_lds_barrier();   // every read of the tables and of val is done: clear what this round wrote, under the stores
        if (valid) {
            hkey[h] = SF_NONE; hmin[h] = SF_NONE; hmax[h] = 0u;
            const uint32_t t = ASC ? (uint32_t)j - i_cur : i_cur - (uint32_t)j;
            if (t < SF_T) { omin[t] = SF_NONE; omax[t] = 0u; }
        }
        if (top_up) { jring[sf & (SF_RING - 1)] = j_next; filled += SF_T; }   // (slots of steps already applied)
        done += count;
        ++round;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SF_MARK(3)
    }
#ifdef SF_PROFILE
    if (blockIdx.x == 0 && (l == 0 || l == 1000)) printf("swaps_full thread %u: %u rounds; clocks per round: loads issued + insert + barrier %lld, hazards + values + barrier %lld, chains %lld, stores + clear + wait + barrier %lld\n",
                                  l, round, pf[0] / round, pf[1] / round, pf[2] / round, pf[3] / round);
#endif
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------

static double expected_draws_per_perm(int64_t n)
{
    double e = 0.0;
    for (int64_t i = 1; i < n; ++i) {
        uint32_t m = (uint32_t)i;
        m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
        e += ((double)m + 1.0) / ((double)i + 1.0);
    }
    return e;
}

bool permgen_is_block_parallel(const sc_ctx *c, int64_t n) { return c->pg_mode != 1 && !c->pg_streams_serial && n >= PHI_MIN_N; }

int permgen_begin(sc_ctx *c, const uint64_t *state6, int64_t n, int64_t n_perm, PermJob *job, hipStream_t s)
{
    job->n = n;
    job->n_perm = n_perm;
    job->h = state6[4] ? 1 : 0;
    job->st_hi = state6[0]; job->st_lo = state6[1]; job->inc_hi = state6[2]; job->inc_lo = state6[3];
    job->buffered = (uint32_t)state6[5];
    job->trivial = (n == 1);
    if (job->trivial) {  // nothing is drawn, every permutation is [0]
        SC_HIP(hipMemsetAsync(c->perm.p, 0, sizeof(int32_t) * (size_t)(c->p_stride * n_perm), s));
        return SC_OK;
    }
    const int64_t M = n - 1;
    const u128 inc = ((u128)job->inc_hi << 64) | job->inc_lo;
    job->total_steps = (uint64_t)n_perm * (uint64_t)M;
    // raw draws: expectation + 0.3 % + slack (the spread of the total is ~sqrt(total), far below that)
    job->draws_per_perm = expected_draws_per_perm(n);
    const double want = (double)n_perm * job->draws_per_perm * 1.003 + 262144.0;
    const uint64_t n_blocks = ((uint64_t)want + SCAN_BLOCK - 1) / SCAN_BLOCK;
    job->hi = n_blocks * SCAN_BLOCK;  // raw draw r = half (r & 1) of 64-bit output r / 2
    SC_TRY(c->pg_raw.ensure(sizeof(uint32_t) * (size_t)job->hi, &c->mem));
    SC_TRY(c->pg_J.ensure(sizeof(int32_t) * (size_t)job->total_steps, &c->mem));
    SC_TRY(c->pg_bits.ensure(sizeof(bits_t) * (size_t)(n_blocks * SCAN_THREADS), &c->mem));
    SC_TRY(c->pg_enter.ensure(sizeof(uint32_t) * (size_t)(n_blocks * SCAN_THREADS), &c->mem));
    SC_TRY(c->pg_sblk.ensure(sizeof(unsigned long long) * (size_t)(n_blocks + 1), &c->mem));
    job->phi = permgen_is_block_parallel(c, n);
    job->B_done = 0; job->unit_no = 0;
    for (int64_t &g : job->gate_seen) g = 0;
    job->ahead = c->pg_ahead >= 1 && c->pg_ahead <= PHI_AHEAD_MAX ? c->pg_ahead : 1;
    if (job->phi && !c->pg_probed) {   // first block-parallel job of this context: can its streams overlap at all?
        for (hipStream_t &sp : c->stream_pg)
            if (!sp) SC_HIP(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
        SC_TRY(permgen_probe_streams(c, s));
        job->phi = permgen_is_block_parallel(c, n);
    }
    if (job->phi) {
        SC_TRY(c->pg_desc.ensure(sizeof(PhiDesc) * (size_t)PHI_RING, &c->mem));
        SC_TRY(c->pg_tbits.ensure(sizeof(unsigned long long) * (size_t)PHI_RING * 2 * PHI_WORDS, &c->mem));
        SC_TRY(c->pg_events.ensure(sizeof(uint16_t) * (size_t)PHI_RING * 2 * PHI_MAX_EV, &c->mem));
        SC_TRY(c->pg_hard.ensure((size_t)n_blocks + 1, &c->mem));
        SC_TRY(c->pg_seg.ensure(sizeof(PhiSeg) * (size_t)PHI_RING, &c->mem));
        SC_TRY(c->pg_ctbits.ensure(sizeof(unsigned long long) * (size_t)PHI_RING * 2 * PHI_WORDS, &c->mem));
        SC_TRY(c->pg_segmode.ensure((size_t)n_blocks + 1, &c->mem));
        SC_TRY(c->pg_seglist.ensure(sizeof(uint32_t) * (size_t)PHI_FLAG_SLOTS * (1 + PHI_UNIT), &c->mem));
        // fresh tables: [control | descriptors | tables]
        SC_TRY(c->pg_fresh.ensure(sizeof(FreshCtl) + sizeof(FreshDesc) * FR_RING + sizeof(unsigned long long) * FR_RING * 2 * FR_WORDS, &c->mem));
        SC_HIP(hipMemsetAsync(c->pg_fresh.p, 0, sizeof(FreshCtl) + sizeof(FreshDesc) * FR_RING, s));
        if (!c->stream_fr) SC_HIP(hipStreamCreateWithFlags(&c->stream_fr, hipStreamNonBlocking));
        SC_HIP(hipMemsetAsync(c->pg_segmode.p, 0, (size_t)n_blocks + 1, s));
        int prio_lo = 0, prio_hi = 0;  // the generator is the critical path of its callers (plain streams if refused)
        const bool prio = getenv("SC_STREAM_PRIORITY") && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess;
        for (hipStream_t &sp : c->stream_pg) {
            if (sp) continue;
            if (!prio || hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, prio_hi) != hipSuccess) {
                (void)hipGetLastError();
                sp = nullptr;
                SC_HIP(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
            }
        }
    }
    for (hipEvent_t &e : c->pg_ev)
        if (!e) SC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // pg_out: [0..3] scan state, [4] prepared blocks used, [5] blocks computed by the chain, then one
    // {first block, end block} pair per chunk for k_expand
    const int64_t chunks = ceil_div64(n_perm, PERM_CHUNK) + 2;  // the fused pipeline splits its first chunk
    SC_TRY(c->pg_out.ensure(sizeof(unsigned long long) * (size_t)(8 + 2 * (chunks + 1)), &c->mem));
    job->chunk_no = 0;
    // A generator that starts with a buffered 32-bit half: that half is the first draw of the
    // stream.  It is consumed here, so that raw draw 0 is always the low half of output 0.
    unsigned long long st0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (job->h) {
        uint32_t mask = (uint32_t)M;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint32_t v = job->buffered & mask;
        if (v <= (uint32_t)M) {  // first Fisher-Yates step (i = n-1) accepts it
            const int32_t j0 = (int32_t)v;
            SC_HIP(hipMemcpyAsync(c->pg_J.p, &j0, sizeof(int32_t), hipMemcpyHostToDevice, s));
            st0[0] = 1;
        }
    }
    // [0] units the chain has completed, [1 .. 16] "unit prepared" words
    SC_TRY(c->pg_flags.ensure(sizeof(uint32_t) * (1 + 2 * PHI_FLAG_SLOTS), &c->mem));
    SC_HIP(hipMemsetAsync(c->pg_flags.p, 0, sizeof(uint32_t) * (1 + 2 * PHI_FLAG_SLOTS), s));
    SC_HIP(hipMemcpyAsync(c->pg_out.p, st0, sizeof(st0), hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(c->pg_sblk.p, st0, sizeof(unsigned long long), hipMemcpyHostToDevice, s));  // state at block 0
    SC_HIP(hipStreamSynchronize(s));  // st0 / j0 are stack variables
    const Affine jb = lcg_pow(inc, SCAN_BLOCK / 2);
    const uint64_t threads = ((n_blocks + RAW_BLOCKS - 1) / RAW_BLOCKS) * (uint64_t)(SCAN_THREADS * SCAN_GROUPS);
    hipLaunchKernelGGL(k_raw_stream, dim3((unsigned)(threads / 256)), dim3(256), 0, s, job->st_hi, job->st_lo,
                       job->inc_hi, job->inc_lo, n_blocks, (uint64_t)(jb.mult >> 64), (uint64_t)jb.mult,
                       (uint64_t)(jb.plus >> 64), (uint64_t)jb.plus, c->pg_raw.as<uint32_t>());
    SC_HIP(hipGetLastError());
    if (job->phi) SC_HIP(hipEventRecord(c->pg_ev[32], s));  // the preparation streams start after the raw stream
    return SC_OK;
}

// Advance the rejection scan until permutations [0, p1) are complete, then expand the accept masks
// of the blocks it processed into J (both on stream s; the expansion uses the whole chip).
int permgen_scan_chunk(sc_ctx *c, PermJob *job, int64_t p1, hipStream_t s, hipStream_t post, hipEvent_t done)
{
    if (job->trivial) return SC_OK;
    const uint64_t n_blocks = job->hi / SCAN_BLOCK;
    unsigned long long *st = c->pg_out.as<unsigned long long>();
    unsigned long long *range = st + 8 + 2 * job->chunk_no;
    // range[0] = first block of this launch (= st[1] now), range[1] = st[1] afterwards
    SC_HIP(hipMemcpyAsync(range, st + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    const uint64_t target = (uint64_t)p1 * (uint64_t)(job->n - 1);
    uint64_t phi_first = 0, phi_end = 0;
    unsigned phi_fill_streams = 0;
    if (job->phi) {
        // Blocks granted to this chunk: the expected draws of permutations [0, p1) + ~10 sigma + one block
        // (k_chain raises a flag if they do not complete the chunk); the last chunk takes all blocks.
        const double need = (double)p1 * job->draws_per_perm + 9000.0 * sqrt((double)p1) + (double)SCAN_BLOCK;
        uint64_t B_end = (uint64_t)(need / SCAN_BLOCK) + 1;
        // ... rounded UP to whole launch units (r04).  A chunk that ends inside a unit leaves a SHORT last unit, which the
        // chain finishes in a fraction of a unit's time -- and the first unit of the next chunk, prepared `ahead` units ahead
        // in chain time, is then not ready: the clock profile of the chain inside the Moran pipeline showed ~13 such waits
        // per 1000 x 1M job, 1-2 ms each (20 of the chain's 130 ms), and nothing in between.  The extra blocks (< 6
        // permutations' worth) are simply scanned one chunk earlier.
        B_end = (B_end + PHI_UNIT - 1) / PHI_UNIT * PHI_UNIT;
        if (B_end > n_blocks || p1 >= job->n_perm) B_end = n_blocks;
        phi_first = job->B_done;
        phi_end = B_end;
        KernelTimerScope ts(c, SC_K_PERM_SCAN, s);
        // the chunk's launch units: each prepared by its own launches (4 rotating streams), all chained by ONE launch
        const uint64_t g0 = job->B_done;
        const int64_t u_first = job->unit_no;
        uint32_t *flags = c->pg_flags.as<uint32_t>();
        unsigned fill_streams = 0;
        while (job->B_done < B_end) {
            const uint64_t b0 = job->B_done;
            const uint64_t b1 = b0 + PHI_UNIT < B_end ? b0 + PHI_UNIT : B_end;
            const int64_t u = job->unit_no;
            hipStream_t sp = c->stream_pg[(size_t)(u % PHI_STREAMS)];
            // The guess of unit u uses the exact state at the start of unit u - ahead, which the chain leaves when it
            // completes unit u - ahead - 1; that unit also is the last reader of the ring slots unit u overwrites.
            const int64_t dep = u - job->ahead - 1;
            uint32_t *seglist = c->pg_seglist.as<uint32_t>() + (size_t)(u % PHI_FLAG_SLOTS) * (1 + PHI_UNIT);
            const uint64_t ref = u >= job->ahead ? job->unit_start[(size_t)((u - job->ahead) % 8)] : 0;
            if (u < PHI_STREAMS) SC_HIP(hipStreamWaitEvent(sp, c->pg_ev[32], 0));  // the raw stream (recorded by permgen_begin)
            // (a gate in front of this stream's last k_seg_fill has waited for the same or a later "unit done" already)
            if (dep >= 0 && dep + 1 > job->gate_seen[(size_t)(u % PHI_STREAMS)])
                hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sp, flags, (uint32_t)(dep + 1), st);
            hipLaunchKernelGGL(k_phi_events, dim3((unsigned)(b1 - b0)), dim3(SCAN_THREADS), 0, sp,
                               c->pg_raw.as<uint32_t>(), (uint32_t)job->n, job->total_steps, job->draws_per_perm, b0,
                               b1, ref, c->pg_sblk.as<unsigned long long>(), c->pg_desc.as<PhiDesc>(),
                               c->pg_events.as<uint16_t>(), seglist);
            hipLaunchKernelGGL(k_phi_tbuild, dim3((unsigned)(b1 - b0)), dim3(128), 0, sp, b0, b1,
                               c->pg_desc.as<PhiDesc>(), c->pg_events.as<uint16_t>(),
                               c->pg_tbits.as<unsigned long long>(), c->pg_seg.as<PhiSeg>(), seglist);
            hipLaunchKernelGGL(k_phi_compose, dim3(PHI_COMPOSE_WGS), dim3(SCAN_THREADS), 0, sp, b0,
                               c->pg_desc.as<PhiDesc>(), c->pg_tbits.as<unsigned long long>(), c->pg_seg.as<PhiSeg>(),
                               c->pg_ctbits.as<unsigned long long>(), seglist);
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, sp, flags, (uint32_t)(1 + u % PHI_FLAG_SLOTS), (uint32_t)(u + 1));
            // behind the chain's "unit u done": the entry states of the blocks inside the unit's segments (this stream's
            // next unit, u + PHI_STREAMS, overwrites the ring slots they are read from and is enqueued behind this)
            hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sp, flags, (uint32_t)(u + 1), st);
            job->gate_seen[(size_t)(u % PHI_STREAMS)] = u + 1;
            hipLaunchKernelGGL(k_seg_fill, dim3((unsigned)(b1 - b0)), dim3(64), 0, sp, b0, b1, c->pg_desc.as<PhiDesc>(),
                               c->pg_seg.as<PhiSeg>(), c->pg_tbits.as<unsigned long long>(), c->pg_segmode.as<uint8_t>(),
                               c->pg_sblk.as<unsigned long long>(), c->pg_hard.as<uint8_t>(), st);
            fill_streams |= 1u << (unsigned)(u % PHI_STREAMS);
            job->unit_start[(size_t)(u % 8)] = b0;
            job->B_done = b1;
            job->unit_no = u + 1;
        }
        // the fresh-table helpers of this chain launch (their own stream; they end with the chain launch)
        static const bool no_fresh = !PHI_FRESH_TABLES || getenv("SC_NO_FRESH") != nullptr;   // development build only; A/B switch
        FreshCtl *fctl = no_fresh ? nullptr : reinterpret_cast<FreshCtl *>(c->pg_fresh.p);
        FreshDesc *fdesc = no_fresh ? nullptr : reinterpret_cast<FreshDesc *>(reinterpret_cast<char *>(c->pg_fresh.p) + sizeof(FreshCtl));
        unsigned long long *ftbits = no_fresh ? nullptr : reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(fdesc) + sizeof(FreshDesc) * FR_RING);
        const uint32_t launch_id = (uint32_t)job->chunk_no + 1;
        if (fctl) {
            if (job->chunk_no == 0) SC_HIP(hipStreamWaitEvent(c->stream_fr, c->pg_ev[32], 0));   // (the raw stream and the zeroed control block)
            hipLaunchKernelGGL(k_fresh, dim3(FR_HELPERS), dim3(SCAN_THREADS), 0, c->stream_fr, c->pg_raw.as<uint32_t>(), n_blocks,
                               (uint32_t)job->n, job->total_steps, job->draws_per_perm, fctl, fdesc, ftbits, st, launch_id);
        }
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(SCAN_THREADS), 0, s, c->pg_raw.as<uint32_t>(), n_blocks,
                           (uint32_t)job->n, job->total_steps, g0, B_end, target,
                           c->pg_desc.as<PhiDesc>(), c->pg_tbits.as<unsigned long long>(), c->pg_seg.as<PhiSeg>(),
                           c->pg_ctbits.as<unsigned long long>(), c->pg_hard.as<uint8_t>(), c->pg_segmode.as<uint8_t>(),
                           (c->pg_mode == 2 && u_first == 0) ? 1 : 0, c->pg_bits.as<bits_t>(),
                           c->pg_enter.as<uint32_t>(), c->pg_sblk.as<unsigned long long>(), st, flags, (uint32_t)u_first,
                           fctl, launch_id);
        SC_HIP(hipGetLastError());
        // the verification / expansion of this chunk reads the entry states k_seg_fill leaves on the preparation streams
        for (unsigned q = 0; q < PHI_STREAMS; ++q)
            if (fill_streams & (1u << q)) SC_HIP(hipEventRecord(c->pg_ev[q], c->stream_pg[q]));
        phi_fill_streams = fill_streams;
    } else {
        KernelTimerScope ts(c, SC_K_PERM_SCAN, s);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_THREADS), 0, s, c->pg_raw.as<uint32_t>(), n_blocks,
                           (uint32_t)job->n, target, job->total_steps, c->pg_bits.as<bits_t>(),
                           c->pg_enter.as<uint32_t>(), c->pg_sblk.as<unsigned long long>(), st);
    }
    SC_HIP(hipMemcpyAsync(range + 1, st + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    // Verification and expansion of the chunk use the whole chip for ~2.5 ms: on `post` (if given) they do not hold
    // up the chain of the next chunk on s.  They only read what this chunk's chain left and write J.
    hipStream_t sp = post ? post : s;
    if (sp != s) {
        SC_HIP(hipEventRecord(c->pg_ev[33], s));
        SC_HIP(hipStreamWaitEvent(sp, c->pg_ev[33], 0));
    }
    for (unsigned q = 0; q < PHI_STREAMS; ++q)
        if (phi_fill_streams & (1u << q)) SC_HIP(hipStreamWaitEvent(sp, c->pg_ev[q], 0));
    // blocks this launch can have covered: the chunk's expected draws + 1 % + 2 blocks
    const double chunk_perms = (double)(p1 - job->p_done);
    uint64_t max_blocks = (uint64_t)(chunk_perms * job->draws_per_perm * 1.01 / SCAN_BLOCK) + 3;
    if (job->phi) {
        max_blocks = phi_end - phi_first;
        if (max_blocks == 0) max_blocks = 1;
        // the prepared blocks again, from their exact entry states, on the whole chip + verification of the chain
        KernelTimerScope ts(c, SC_K_PERM_SCAN, sp);
        hipLaunchKernelGGL(k_block_exact, dim3((unsigned)max_blocks), dim3(SCAN_THREADS), 0, sp,
                           c->pg_raw.as<uint32_t>(), (uint32_t)job->n, job->total_steps, range, c->pg_hard.as<uint8_t>(),
                           c->pg_bits.as<bits_t>(), c->pg_enter.as<uint32_t>(), c->pg_sblk.as<unsigned long long>(), st);
    }
    hipLaunchKernelGGL(k_expand, dim3((unsigned)(max_blocks * SCAN_THREADS / 256)), dim3(256), 0, sp,
                       c->pg_raw.as<uint32_t>(), c->pg_bits.as<bits_t>(), c->pg_enter.as<uint32_t>(),
                       c->pg_sblk.as<unsigned long long>(), range, (uint32_t)job->n, job->total_steps,
                       c->pg_J.as<int32_t>());
    if (done) SC_HIP(hipEventRecord(done, sp));
    SC_HIP(hipGetLastError());
    job->p_done = p1;
    job->chunk_no += 1;
    return SC_OK;
}

bool permgen_can_swap_inverse(int64_t n) { return n >= SWAPS_WG_MIN_N; }

// inverse = false: rows [p0, p1) of the permutation table (c->perm); true: of its inverse (c->inv), by the same
// transpositions in ascending order (workgroup kernel only: see permgen_can_swap_inverse)
int permgen_swap_chunk(sc_ctx *c, PermJob *job, int64_t p0, int64_t p1, hipStream_t s, bool inverse, int pw_req)
{
    if (job->trivial || p1 <= p0) {
        if (job->trivial && inverse && p1 > p0)
            SC_HIP(hipMemsetAsync(c->inv.as<int32_t>() + p0 * c->p_stride, 0, sizeof(int32_t) * (size_t)(c->p_stride * (p1 - p0)), s));
        return SC_OK;
    }
    SC_REQUIRE(!inverse || permgen_can_swap_inverse(job->n), SC_ERR_STATE, "permgen_swap_chunk: inverse tables need n >= %d",
               SWAPS_WG_MIN_N);
    KernelTimerScope ts(c, SC_K_PERM_SWAP, s);
    const char *pw_e = getenv("SC_SWAP_PW");   // (development and tests: A/B; read per call)
    const int pw_env = pw_e ? atoi(pw_e) : 0;
    const int pw = pw_env ? pw_env : pw_req;   // permutations per workgroup
    const unsigned wgs = (unsigned)(pw == 2 ? (p1 - p0 + 1) / 2 : p1 - p0);
    // r04 NEGATIVE RESULT, opt-in (SC_SWAP_FULL_ROUNDS=1): whole rounds of 1024 steps (k_apply_swaps_full).  1058 instead of
    // 2880 rounds per 10^6-step permutation, but a round of sixteen wavefronts on one CU is bound by instruction issue, not by
    // its trip to L2 (13 k clocks against 6 k): 6.2 instead of 7.2 ms per 128-permutation chunk alone; inside the Moran
    // pipeline the swaps take 77 instead of 115 ms per step and the generator's chain, which now finds 2048 instead of 1024
    // long-lived wavefronts and 128 KB of LDS per workgroup in its way, 148 instead of 131 ms: the step 161 against 157.5 ms.
    const bool full_rounds = getenv("SC_SWAP_FULL_ROUNDS") != nullptr;
    if (full_rounds && job->n >= SWAPS_WG_MIN_N) {
        if (inverse) hipLaunchKernelGGL(k_apply_swaps_full<true>, dim3((unsigned)(p1 - p0)), dim3(SF_T), 0, s, c->pg_J.as<int32_t>(),
                                        c->inv.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
        else hipLaunchKernelGGL(k_apply_swaps_full<false>, dim3((unsigned)(p1 - p0)), dim3(SF_T), 0, s, c->pg_J.as<int32_t>(),
                                c->perm.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
    } else if (inverse) {
        if (pw == 2) hipLaunchKernelGGL((k_apply_swaps_wg<true, 2>), dim3(wgs), dim3(2 * SW_T), 0, s, c->pg_J.as<int32_t>(),
                                        c->inv.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
        else hipLaunchKernelGGL((k_apply_swaps_wg<true, 1>), dim3(wgs), dim3(SW_T), 0, s, c->pg_J.as<int32_t>(),
                                c->inv.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
    } else if (job->n >= SWAPS_WG_MIN_N) {
        if (pw == 2) hipLaunchKernelGGL((k_apply_swaps_wg<false, 2>), dim3(wgs), dim3(2 * SW_T), 0, s, c->pg_J.as<int32_t>(),
                                        c->perm.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
        else hipLaunchKernelGGL((k_apply_swaps_wg<false, 1>), dim3(wgs), dim3(SW_T), 0, s, c->pg_J.as<int32_t>(),
                                c->perm.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
    }
    else
        hipLaunchKernelGGL(k_apply_swaps, dim3((unsigned)(p1 - p0)), dim3(64), 0, s, c->pg_J.as<int32_t>(),
                           c->perm.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// after every stream that ran chunks has been synchronised: verify and write the final state
int permgen_finish(sc_ctx *c, PermJob *job, uint64_t *state6)
{
    if (job->trivial) return SC_OK;
    unsigned long long st[8];
    SC_HIP(hipMemcpy(st, c->pg_out.p, sizeof(st), hipMemcpyDeviceToHost));
    if (job->phi) { c->pg_blocks_prepared += (int64_t)st[4]; c->pg_blocks_chain += (int64_t)st[5]; }
    if (job->phi && st[2] != 0) {  // verification of the block-parallel scan failed: the caller reruns sequentially
        c->pg_fallbacks += 1;
        if (st[2] & 24ull) {   // a hand-over wait gave up (flags 8 / 16): later jobs take the sequential scan at once
            c->pg_streams_serial = true;
            c->pg_note = "a hand-over wait of the block-parallel permutation generator gave up after 1 s (kernels of its streams did "
                         "not overlap: GPU shared with another process, a profiler, too few hardware queues); this context now uses "
                         "the sequential scan (same results); sc_ctx_set_permgen_mode(ctx, 0) re-arms the block-parallel form";
        }
        sc_set_error("sc_perm_generate: block-parallel scan failed its verification (flags %llu)", st[2]);
        if (!(st[2] & 24ull)) {   // (a one-off: the context stays on the block-parallel form, but the event is on record)
            char buf[200];
            snprintf(buf, sizeof(buf), "a block-parallel permutation job failed its verification (flags %llu) and was rerun with the "
                     "sequential scan (same results)", st[2]);
            c->pg_note = buf;
        }
        return SC_PERMGEN_RETRY;
    }
    SC_REQUIRE(st[2] == 0, SC_ERR_STATE, "sc_perm_generate: rejection scan did not converge");
    (job->phi ? c->pg_jobs_parallel : c->pg_jobs_sequential) += 1;
    SC_REQUIRE(st[0] == job->total_steps, SC_ERR_STATE,
               "sc_perm_generate: raw stream exhausted after %llu of %llu steps", st[0],
               (unsigned long long)job->total_steps);
    const uint64_t h = job->h;
    const uint64_t pos = st[3] + h;  // draws consumed: raw draws + the buffered half taken on the host
    const u128 state0 = ((u128)job->st_hi << 64) | job->st_lo, inc = ((u128)job->inc_hi << 64) | job->inc_lo;
    if (pos > h) {
        const uint64_t tl = pos - 1;           // last consumed position (>= h)
        const uint64_t m_last = (tl - h) / 2;  // its 64-bit output
        const Affine a = lcg_pow(inc, m_last + 1);
        const u128 sN = a.mult * state0 + a.plus;
        state6[0] = (uint64_t)(sN >> 64);
        state6[1] = (uint64_t)sN;
        state6[4] = ((tl - h) & 1) == 0 ? 1 : 0;  // low half consumed -> high half buffered
        state6[5] = (uint32_t)(xsl_rr(sN) >> 32);
    } else if (pos == 1 && h == 1) {
        state6[4] = 0;  // only the buffered half was consumed; uinteger keeps its value
    }
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// Counter-based permutations (r03; SURVEY 8(e) "alternative", H2) for the paths that have NO reference seed semantics
// (label-permutation enrichment, shared-permutation Lee grids): permutation p is a pure function of (seed, p), so ranks
// and batches can take disjoint ranges of p and merge integer counts.  Definition (documented, reproducible, the same
// on any number of GPUs): Fisher-Yates as numpy runs it -- for i = n-1 .. 1: j uniform on [0, i]; swap a[i], a[j] -- with
//   j = bounded(Philox4x32-10(key = (seed low word, seed high word), counter = (i, r, p low, p high)), i + 1)
// where the first two output words form a 64-bit u and bounded is Lemire's multiply-shift with its exact rejection
// (u * (i + 1) >> 64, rejected -- retry with r + 1 -- when the low half falls below 2^64 mod (i + 1): probability < 2^-43
// per draw, so r is 0 in practice and the draw stays a pure function of its counter).  No sequential stage at all: J is
// filled by the whole chip, the swaps are the generator's stage B.
// ------------------------------------------------------------------------------------------------
__host__ __device__ static inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__host__ __device__ static inline uint32_t counter_bounded(uint32_t k0, uint32_t k1, uint64_t p, uint32_t i)
{
    const uint64_t range = (uint64_t)i + 1;
    for (uint32_t r = 0;; ++r) {
        uint32_t c[4] = {i, r, (uint32_t)p, (uint32_t)(p >> 32)};
        philox4x32_10(c, k0, k1);
        const uint64_t u = ((uint64_t)c[1] << 32) | c[0];
        const u128 m = (u128)u * range;
        const uint64_t low = (uint64_t)m;
        if (low >= range || low >= (0 - range) % range) return (uint32_t)(m >> 64);
    }
}

// J[(p - p_first) * M + (M - i)] = the swap partner of step i of permutation p (the layout stage B reads)
__global__ __launch_bounds__(256) void k_counter_J(uint32_t k0, uint32_t k1, uint32_t n, uint64_t p_first, int64_t n_perm,
                                                   int32_t *__restrict__ J)
{
    const uint32_t M = n - 1;
    const int64_t total = n_perm * (int64_t)M;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / M;
        const uint32_t s = (uint32_t)(t - p * M);
        J[t] = (int32_t)counter_bounded(k0, k1, p_first + (uint64_t)p, M - s);
    }
}

extern "C" int sc_perm_counter_host(uint64_t seed, int64_t n, int64_t p_first, int64_t n_perm, int32_t *perm_out)
{
    SC_REQUIRE(perm_out || n_perm == 0 || n == 0, SC_ERR_INVALID, "sc_perm_counter_host: null pointer");
    SC_REQUIRE(n >= 0 && n <= 0x7fffffffLL && n_perm >= 0 && p_first >= 0, SC_ERR_INVALID, "sc_perm_counter_host: bad sizes");
    for (int64_t p = 0; p < n_perm; ++p) {
        int32_t *a = perm_out + p * n;
        for (int64_t i = 0; i < n; ++i) a[i] = (int32_t)i;
        for (int64_t i = n - 1; i >= 1; --i) {
            const uint32_t j = counter_bounded((uint32_t)seed, (uint32_t)(seed >> 32), (uint64_t)(p_first + p), (uint32_t)i);
            const int32_t t = a[j]; a[j] = a[i]; a[i] = t;
        }
    }
    return SC_OK;
}

// rows [0, n_perm) of the (already allocated, >= n_perm rows) table <- counter-based permutations p_first .., on stream s
int sc_perm_counter_rows(sc_ctx *c, uint64_t seed, int64_t n, int64_t p_first, int64_t n_perm, hipStream_t s)
{
    if (n_perm <= 0) return SC_OK;
    if (n == 1) {
        SC_HIP(hipMemsetAsync(c->perm.p, 0, sizeof(int32_t) * (size_t)(c->p_stride * n_perm), s));
        return SC_OK;
    }
    const int64_t M = n - 1;
    SC_TRY(c->pg_J.ensure(sizeof(int32_t) * (size_t)(M * n_perm + 64), &c->mem));
    const int64_t total = M * n_perm;
    const unsigned grid = (unsigned)(ceil_div64(total, 256) < 65536 ? ceil_div64(total, 256) : 65536);
    hipLaunchKernelGGL(k_counter_J, dim3(grid), dim3(256), 0, s, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)n,
                       (uint64_t)p_first, n_perm, c->pg_J.as<int32_t>());
    if (n >= SWAPS_WG_MIN_N)
        hipLaunchKernelGGL((k_apply_swaps_wg<false, 1>), dim3((unsigned)n_perm), dim3(SW_T), 0, s, c->pg_J.as<int32_t>(),
                           c->perm.as<int32_t>(), c->p_stride, (uint32_t)n, (int64_t)0, n_perm);
    else
        hipLaunchKernelGGL(k_apply_swaps, dim3((unsigned)n_perm), dim3(64), 0, s, c->pg_J.as<int32_t>(),
                           c->perm.as<int32_t>(), c->p_stride, (uint32_t)n, (int64_t)0, n_perm);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_perm_generate_counter(sc_ctx *c, uint64_t seed, int64_t n, int64_t p_first, int64_t n_perm, int32_t *perm_out)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "sc_perm_generate_counter: null context");
    SC_REQUIRE(p_first >= 0, SC_ERR_INVALID, "sc_perm_generate_counter: negative first permutation");
    SC_HIP(hipSetDevice(c->device));
    SC_TRY(sc_perm_alloc(c, n, n_perm));
    {
        KernelTimerScope ts(c, SC_K_PERMGEN);
        SC_TRY(sc_perm_counter_rows(c, seed, n, p_first, n_perm, c->stream));
    }
    c->p_count = n_perm;
    c->perm_bijective = true;
    c->perm_forward_valid = true;
    if (perm_out)
        SC_HIP(hipMemcpy2DAsync(perm_out, sizeof(int32_t) * (size_t)n, c->perm.p, sizeof(int32_t) * (size_t)c->p_stride,
                                sizeof(int32_t) * (size_t)n, (size_t)n_perm, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

static int perm_generate_once(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm)
{
    if (permgen_is_block_parallel(c, n) && n_perm > 3 * PERM_CHUNK) {
        // a long job: the chunked pipeline of the seeded statistics with nothing to consume -- the Fisher-Yates swaps of
        // chunk k run (on their own streams) beside the rejection scan of chunk k + 1 instead of all behind the scan
        // (r03, 999 permutations of 1M cells: 55 ms of swaps out of the call's critical path)
        const int ahead = c->pg_ahead;
        c->pg_ahead = 2;
        const int rc = sc_perm_pipeline(c, state6, n, n_perm, 0, nullptr, [](int64_t, int64_t) -> int { return SC_OK; });
        c->pg_ahead = ahead;
        if (rc == SC_OK) c->perm_forward_valid = true;
        return rc;
    }
    PermJob job;
    SC_TRY(permgen_begin(c, state6, n, n_perm, &job, c->stream));
    for (int64_t p0 = 0; p0 < n_perm; p0 += PERM_CHUNK) {
        const int64_t p1 = p0 + PERM_CHUNK < n_perm ? p0 + PERM_CHUNK : n_perm;
        SC_TRY(permgen_scan_chunk(c, &job, p1, c->stream, nullptr, nullptr));
    }
    SC_TRY(permgen_swap_chunk(c, &job, 0, n_perm, c->stream, false, 1));
    c->perm_forward_valid = true;
    SC_HIP(hipStreamSynchronize(c->stream));
    for (hipStream_t sp : c->stream_pg)
        if (sp) SC_HIP(hipStreamSynchronize(sp));
    if (c->stream_fr) SC_HIP(hipStreamSynchronize(c->stream_fr));
    return permgen_finish(c, &job, state6);
}

int sc_perm_generate_device(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm)
{
    int rc = perm_generate_once(c, state6, n, n_perm);
    if (rc == SC_PERMGEN_RETRY) {  // state6 is only written on success: rerun with the sequential scan
        const int mode = c->pg_mode;
        c->pg_mode = 1;
        rc = perm_generate_once(c, state6, n, n_perm);
        c->pg_mode = mode;
    }
    return rc;
}
