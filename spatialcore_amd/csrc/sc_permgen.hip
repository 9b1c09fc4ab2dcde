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
//  A1  rejection    ONE 1024-thread workgroup walks the raw stream in blocks of 32768 draws.  Each
//                   thread simulates its 32 consecutive draws sequentially (exact semantics) from a
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
#define SCAN_D 32
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
    // first draw of the group: r = b*32768 + 32*tau + 4*g  ->  64-bit output m = r / 2
    const uint64_t m = b * (SCAN_BLOCK / 2) + (uint64_t)(SCAN_D / 2) * tau + 2ull * g;
    const Affine j = lcg_pow(inc, m + 1);  // output m is made from the state after m + 1 steps
    u128 s = j.mult * state0 + j.plus;
    for (int k = 0; k < RAW_BLOCKS && b < n_blocks; ++k, ++b) {
        const uint64_t o0 = xsl_rr(s);
        const uint64_t o1 = xsl_rr(s * mult + inc);
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
    if (i0 > (mask >> 1) + SCAN_D && c_in + SCAN_D < limit) {
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
        return;
    }
    uint32_t i = i0, off = c_in, cnt = 0;
    bits_t bits = 0;
#pragma unroll
    for (int s = 0; s < SCAN_D; ++s) {
        const uint32_t v = u[s] & mask;
        const bool acc = off < limit && v <= i;
        if (acc) {
            bits |= (bits_t)1 << s;
            ++off; ++cnt; --i;
            if (off == limit) r.end = (uint32_t)s + 1;
            if (i == 0) { i = M; mask = top_mask; }
            else if (i <= (mask >> 1)) mask >>= 1;
        }
    }
    r.cnt = cnt; r.bits = bits; r.gap = 0; r.fast = 0;
}

// Is the cached result still the exact result for entering count c_new?  On the fast path every
// threshold moves by -(c_new - c_used); no decision flips while the move stays inside the gaps.
__device__ __forceinline__ bool scan_still_valid(const ScanRes &r, uint32_t c_new, uint32_t M, uint32_t limit)
{
    if (c_new == r.c_used) return true;
    if (!r.fast) return false;
    const int64_t delta = (int64_t)c_new - (int64_t)r.c_used;
    const int64_t i0n = (int64_t)r.i0 - delta;  // new first threshold (same permutation, same band required)
    if (i0n > (int64_t)M || i0n > (int64_t)r.mask || i0n <= (int64_t)(r.mask >> 1) + SCAN_D) return false;
    if ((uint64_t)c_new + SCAN_D >= limit) return false;
    return (uint64_t)(delta > 0 ? delta : -delta) <= r.gap;
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

// Per processed block the scan leaves: sblk[b] = steps completed before the block, and per thread
// acc_bits[b*SCAN_THREADS + tau], enter[b*SCAN_THREADS + tau] (accepted steps of the block in front of the thread).
// k_expand turns these into J with the whole chip; one CU cannot store 4 bytes per step fast enough.
//
// st[0] = steps completed so far, st[1] = next block to process, st[2] = sticky failure flag,
// st[3] = number of raw draws consumed when the job's last step completed.
// A launch processes WHOLE blocks while fewer than S_target steps are complete (the last block may
// run past the target; only the end of the job, total_steps, stops mid-block).
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(const uint32_t *__restrict__ raw, uint64_t n_blocks,
                                                       uint32_t n, uint64_t S_target, uint64_t total_steps,
                                                       bits_t *__restrict__ acc_bits,
                                                       uint32_t *__restrict__ enter,
                                                       unsigned long long *__restrict__ sblk,
                                                       unsigned long long *__restrict__ st)
{
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    __shared__ uint32_t wchg[2][SCAN_THREADS / 64];
    static_assert(SCAN_THREADS % 64 == 0 && SCAN_D % 4 == 0 && SCAN_D <= 64, "scan geometry");
    const uint32_t tau = threadIdx.x, lane = tau & 63, wave = tau >> 6;
    const uint32_t M = n - 1;
    const uint32_t top_mask = mask_of(M);
    uint64_t S_block = st[0];
    uint64_t b = st[1];
    uint32_t rem_block = M - (uint32_t)(S_block % M);  // steps left in the current permutation
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
        const uint64_t left = total_steps - S_block;
        const uint32_t limit = left > 0xffffffffULL ? 0xffffffffu : (uint32_t)left;
        const float p_acc = (float)(rem_block + 1.0) / (float)((double)mask_of(rem_block) + 1.0);
        ScanRes r;
        scan_thread(u, (uint32_t)((float)(tau * SCAN_D) * p_acc), rem_block, M, top_mask, limit, r);
        uint32_t excl = 0, total_cnt = 0;
        // fixed point on the entering counts: a thread recomputes only when its cached result is
        // not provably the result for its new entering count
        for (int iter = 0;; ++iter) {
            const uint32_t incl = wave_inclusive_scan(r.cnt);
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < SCAN_THREADS / 64; ++w) {
                const uint32_t t = wsum[w];
                before += (w < (int)wave) ? t : 0u;
                all += t;
            }
            excl = before + incl - r.cnt;
            total_cnt = all;
            const bool stale = !scan_still_valid(r, excl, M, limit);
            const bool wave_stale = __any(stale);
            if (lane == 0) wchg[parity][wave] = wave_stale ? 1u : 0u;
            __syncthreads();
            uint32_t changed = 0;
#pragma unroll
            for (int w = 0; w < SCAN_THREADS / 64; ++w) changed |= wchg[parity][w];
            parity ^= 1u;
            if (!changed) break;
            if (wave_stale) {
                if (stale) scan_thread(u, excl, rem_block, M, top_mask, limit, r);
            }
            if (iter > SCAN_THREADS + 8) { failed = 1; break; }
        }
        acc_bits[b * SCAN_THREADS + tau] = r.bits;
        enter[b * SCAN_THREADS + tau] = excl;
        if (tau == 0) sblk[b] = S_block;
        if (r.end) endpos = b * SCAN_BLOCK + (uint64_t)tau * SCAN_D + r.end;
        S_block += total_cnt;
        {   // steps left in the current permutation after total_cnt more steps
            uint32_t t = total_cnt;
            if (t >= rem_block) { t = (t - rem_block) % M; rem_block = M; }
            rem_block -= t;
        }
    }
    if (endpos) st[3] = endpos;  // exactly one thread of one launch sees the job's last step
    if (tau == 0) {
        st[0] = S_block;
        st[1] = b;
        if (failed) st[2] = 1;
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
    SC_TRY(c->pg_sblk.ensure(sizeof(unsigned long long) * (size_t)n_blocks, &c->mem));
    // pg_out: [0..3] scan state, then one {first block, end block} pair per chunk for k_expand
    const int64_t chunks = ceil_div64(n_perm, PERM_CHUNK) + 2;  // the fused pipeline splits its first chunk
    SC_TRY(c->pg_out.ensure(sizeof(unsigned long long) * (size_t)(4 + 2 * (chunks + 1)), &c->mem));
    job->chunk_no = 0;
    // A generator that starts with a buffered 32-bit half: that half is the first draw of the
    // stream.  It is consumed here, so that raw draw 0 is always the low half of output 0.
    unsigned long long st0[4] = {0, 0, 0, 0};
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
    SC_HIP(hipMemcpyAsync(c->pg_out.p, st0, sizeof(st0), hipMemcpyHostToDevice, s));
    SC_HIP(hipStreamSynchronize(s));  // st0 / j0 are stack variables
    const Affine jb = lcg_pow(inc, SCAN_BLOCK / 2);
    const uint64_t threads = ((n_blocks + RAW_BLOCKS - 1) / RAW_BLOCKS) * (uint64_t)(SCAN_THREADS * SCAN_GROUPS);
    hipLaunchKernelGGL(k_raw_stream, dim3((unsigned)(threads / 256)), dim3(256), 0, s, job->st_hi, job->st_lo,
                       job->inc_hi, job->inc_lo, n_blocks, (uint64_t)(jb.mult >> 64), (uint64_t)jb.mult,
                       (uint64_t)(jb.plus >> 64), (uint64_t)jb.plus, c->pg_raw.as<uint32_t>());
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// Advance the rejection scan until permutations [0, p1) are complete, then expand the accept masks
// of the blocks it processed into J (both on stream s; the expansion uses the whole chip).
int permgen_scan_chunk(sc_ctx *c, PermJob *job, int64_t p1, hipStream_t s)
{
    if (job->trivial) return SC_OK;
    const uint64_t n_blocks = job->hi / SCAN_BLOCK;
    unsigned long long *st = c->pg_out.as<unsigned long long>();
    unsigned long long *range = st + 4 + 2 * job->chunk_no;
    // range[0] = first block of this launch (= st[1] now), range[1] = st[1] afterwards
    SC_HIP(hipMemcpyAsync(range, st + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    const uint64_t target = (uint64_t)p1 * (uint64_t)(job->n - 1);
    {
        KernelTimerScope ts(c, SC_K_PERM_SCAN, s);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_THREADS), 0, s, c->pg_raw.as<uint32_t>(), n_blocks,
                           (uint32_t)job->n, target, job->total_steps, c->pg_bits.as<bits_t>(),
                           c->pg_enter.as<uint32_t>(), c->pg_sblk.as<unsigned long long>(), st);
    }
    SC_HIP(hipMemcpyAsync(range + 1, st + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    // blocks this launch can have covered: the chunk's expected draws + 1 % + 2 blocks
    const double chunk_perms = (double)(p1 - job->p_done);
    const uint64_t max_blocks = (uint64_t)(chunk_perms * job->draws_per_perm * 1.01 / SCAN_BLOCK) + 3;
    hipLaunchKernelGGL(k_expand, dim3((unsigned)(max_blocks * SCAN_THREADS / 256)), dim3(256), 0, s,
                       c->pg_raw.as<uint32_t>(), c->pg_bits.as<bits_t>(), c->pg_enter.as<uint32_t>(),
                       c->pg_sblk.as<unsigned long long>(), range, (uint32_t)job->n, job->total_steps,
                       c->pg_J.as<int32_t>());
    SC_HIP(hipGetLastError());
    job->p_done = p1;
    job->chunk_no += 1;
    return SC_OK;
}

int permgen_swap_chunk(sc_ctx *c, PermJob *job, int64_t p0, int64_t p1, hipStream_t s)
{
    if (job->trivial || p1 <= p0) return SC_OK;
    KernelTimerScope ts(c, SC_K_PERM_SWAP, s);
    hipLaunchKernelGGL(k_apply_swaps, dim3((unsigned)(p1 - p0)), dim3(64), 0, s, c->pg_J.as<int32_t>(),
                       c->perm.as<int32_t>(), c->p_stride, (uint32_t)job->n, p0, p1);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// after every stream that ran chunks has been synchronised: verify and write the final state
int permgen_finish(sc_ctx *c, PermJob *job, uint64_t *state6)
{
    if (job->trivial) return SC_OK;
    unsigned long long st[4];
    SC_HIP(hipMemcpy(st, c->pg_out.p, sizeof(st), hipMemcpyDeviceToHost));
    SC_REQUIRE(st[2] == 0, SC_ERR_STATE, "sc_perm_generate: rejection scan did not converge");
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

int sc_perm_generate_device(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm)
{
    PermJob job;
    SC_TRY(permgen_begin(c, state6, n, n_perm, &job, c->stream));
    for (int64_t p0 = 0; p0 < n_perm; p0 += PERM_CHUNK) {
        const int64_t p1 = p0 + PERM_CHUNK < n_perm ? p0 + PERM_CHUNK : n_perm;
        SC_TRY(permgen_scan_chunk(c, &job, p1, c->stream));
    }
    SC_TRY(permgen_swap_chunk(c, &job, 0, n_perm, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return permgen_finish(c, &job, state6);
}
