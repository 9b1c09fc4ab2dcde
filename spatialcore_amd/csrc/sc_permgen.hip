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
// A0: raw[e0 + 2*r], raw[e0 + 2*r + 1] = low, high half of 64-bit output (m_start + r), r < count
// ------------------------------------------------------------------------------------------------

#define RAW_ROUNDS 16

__global__ __launch_bounds__(256) void k_raw_stream(uint64_t st_hi, uint64_t st_lo, uint64_t inc_hi,
                                                    uint64_t inc_lo, uint64_t m_start, uint64_t count,
                                                    uint32_t e0, uint64_t a64m_hi, uint64_t a64m_lo,
                                                    uint64_t a64p_hi, uint64_t a64p_lo,
                                                    uint32_t *__restrict__ raw)
{
    const u128 state0 = ((u128)st_hi << 64) | st_lo, inc = ((u128)inc_hi << 64) | inc_lo;
    const u128 a64m = ((u128)a64m_hi << 64) | a64m_lo, a64p = ((u128)a64p_hi << 64) | a64p_lo;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    uint64_t r = wave * (64 * RAW_ROUNDS) + lane;
    if (r >= count) return;
    // output index m is produced from the state after m + 1 steps
    Affine j = lcg_pow(inc, m_start + r + 1);
    u128 s = j.mult * state0 + j.plus;
    for (int k = 0; k < RAW_ROUNDS && r < count; ++k, r += 64) {
        uint64_t o = xsl_rr(s);
        raw[e0 + 2 * r] = (uint32_t)o;
        raw[e0 + 2 * r + 1] = (uint32_t)(o >> 32);
        s = a64m * s + a64p;
    }
}

// ------------------------------------------------------------------------------------------------
// A1: rejection scan by one workgroup
// ------------------------------------------------------------------------------------------------

#define SCAN_THREADS 1024
#define SCAN_D 32
#define SCAN_BLOCK (SCAN_THREADS * SCAN_D)

__device__ __forceinline__ uint32_t mask_of(uint32_t i) { return 0xffffffffu >> __clz((int)i); }  // i >= 1

// One thread's sequential pass over its SCAN_D draws, entering with `c_guess` accepted steps in
// front of it inside the block.  EMIT: accepted values go to the LDS staging tile at their step
// offset.  Returns the number of accepts; *my_end = 1 + local index of the draw that completed the
// job's final step (0 if none).
template <bool EMIT>
__device__ __forceinline__ uint32_t scan_thread(const uint32_t (&u)[SCAN_D], uint32_t valid, uint32_t c_guess,
                                                uint32_t rem_block, uint32_t M, uint32_t top_mask,
                                                uint32_t limit, int32_t *stage, uint32_t *my_end)
{
    uint32_t c = c_guess, rem = rem_block;
    if (c >= rem) { c = (c - rem) % M; rem = M; }
    uint32_t i = rem - c, mask = mask_of(i);
    uint32_t off = c_guess, cnt = 0;
    *my_end = 0;
    const bool simple = (i > (mask >> 1) + SCAN_D) && (off + SCAN_D < limit);
    if (simple) {
        // no mask change, no permutation end, no end of job within this thread's draws
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) {
            uint32_t v = u[s] & mask;
            bool acc = ((valid >> s) & 1u) && v <= i;
            if (EMIT && acc) stage[off] = (int32_t)v;
            off += acc; i -= acc; cnt += acc;
        }
    } else {
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) {
            uint32_t v = u[s] & mask;
            bool acc = ((valid >> s) & 1u) && off < limit && v <= i;
            if (acc) {
                if (EMIT) stage[off] = (int32_t)v;
                ++off; ++cnt; --i;
                if (off == limit) *my_end = (uint32_t)s + 1;
                if (i == 0) { i = M; mask = top_mask; }
                else if (i <= (mask >> 1)) mask >>= 1;
            }
        }
    }
    return cnt;
}

// out[0] = steps completed (absolute), out[1] = raw index one past the last consumed draw,
// out[2] = 1 if a block failed to converge (cannot happen; checked by the host)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(const uint32_t *__restrict__ raw, uint64_t lo,
                                                       uint64_t hi, uint32_t n, uint64_t total_steps,
                                                       uint64_t S0, int32_t *__restrict__ J,
                                                       unsigned long long *__restrict__ out)
{
    __shared__ int32_t stage[SCAN_BLOCK];  // 128 KiB: accepted values of one block, by step offset
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    __shared__ unsigned long long s_endpos;
    const uint32_t tau = threadIdx.x, lane = tau & 63, wave = tau >> 6;
    const uint32_t M = n - 1;
    const uint32_t top_mask = mask_of(M);
    uint64_t S_block = S0;
    uint64_t endpos = lo;
    int failed = 0;
    if (tau == 0) s_endpos = lo;
    __syncthreads();

    for (uint64_t base = 0; base < hi && S_block < total_steps; base += SCAN_BLOCK) {
        const uint64_t first = base + (uint64_t)tau * SCAN_D;
        uint32_t u[SCAN_D];
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(raw + first);
#pragma unroll
            for (int q = 0; q < SCAN_D / 4; ++q) {
                uint4 v = src[q];  // the raw buffer is padded to a whole block
                u[4 * q] = v.x; u[4 * q + 1] = v.y; u[4 * q + 2] = v.z; u[4 * q + 3] = v.w;
            }
        }
        // draws of this thread that belong to the stream: raw index in [lo, hi)
        uint32_t valid = 0;
#pragma unroll
        for (int s = 0; s < SCAN_D; ++s) valid |= ((first + s >= lo) && (first + s < hi)) ? (1u << s) : 0u;

        const uint32_t rem_block = M - (uint32_t)(S_block % M);  // steps left in the current permutation
        const uint64_t left = total_steps - S_block;
        const uint32_t limit = left > 0xffffffffULL ? 0xffffffffu : (uint32_t)left;
        const float p_acc = (float)(rem_block + 1.0) / (float)((double)mask_of(rem_block) + 1.0);
        uint32_t c_guess = (uint32_t)((float)(tau * SCAN_D) * p_acc);
        uint32_t total_cnt = 0;
        uint32_t my_end = 0;

        // ---- count-only rounds until the entering counts are a fixed point ----
        for (int iter = 0;; ++iter) {
            const uint32_t cnt = scan_thread<false>(u, valid, c_guess, rem_block, M, top_mask, limit, nullptr, &my_end);
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(incl, d);
                if ((int)lane >= d) incl += t;
            }
            // (the previous round ended with a barrier after its wsum reads)
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < SCAN_THREADS / 64; ++w) {
                uint32_t t = wsum[w];
                before += (w < (int)wave) ? t : 0u;
                all += t;
            }
            const uint32_t excl = before + incl - cnt;
            const int changed = __syncthreads_or(excl != c_guess);
            total_cnt = all;
            if (!changed) break;
            c_guess = excl;
            if (iter > SCAN_THREADS + 8) { failed = 1; break; }
        }
        // ---- emit round: accepted values -> LDS by step offset -> coalesced copy to J ----
        (void)scan_thread<true>(u, valid, c_guess, rem_block, M, top_mask, limit, stage, &my_end);
        if (my_end) s_endpos = first + my_end;  // only the thread that completed the last step
        __syncthreads();
        for (uint32_t k = tau; k < total_cnt; k += SCAN_THREADS) J[S_block + k] = stage[k];
        S_block += total_cnt;
        endpos = (base + SCAN_BLOCK < hi) ? base + SCAN_BLOCK : hi;
        __syncthreads();
    }
    __syncthreads();
    if (tau == 0) {
        out[0] = S_block;
        out[1] = (S_block >= total_steps && total_steps > S0) ? s_endpos : endpos;
        out[2] = (unsigned long long)failed;
    }
}

// ------------------------------------------------------------------------------------------------
// B: apply the swaps, one wavefront per permutation
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_apply_swaps(const int32_t *__restrict__ J, int32_t *__restrict__ perm,
                                                    int64_t pstride, uint32_t n, int64_t n_perm)
{
    const int64_t p = blockIdx.x;
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
        const int32_t j = valid ? Jp[(int64_t)M - i] : -1;
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

int sc_perm_generate_device(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm)
{
    const int64_t M = n - 1;
    const u128 state0 = ((u128)state6[0] << 64) | state6[1];
    const u128 inc = ((u128)state6[2] << 64) | state6[3];
    const uint64_t h = state6[4] ? 1 : 0;
    if (M == 0) {  // n == 1: nothing is drawn, every permutation is [0]
        SC_HIP(hipMemsetAsync(c->perm.p, 0, sizeof(int32_t) * (size_t)(c->p_stride * n_perm), c->stream));
        SC_HIP(hipStreamSynchronize(c->stream));
        return SC_OK;
    }
    const uint64_t total_steps = (uint64_t)n_perm * (uint64_t)M;
    const double e_d = expected_draws_per_perm(n);
    SC_TRY(c->pg_J.ensure(sizeof(int32_t) * (size_t)total_steps, &c->mem));
    SC_TRY(c->pg_out.ensure(sizeof(unsigned long long) * 4, &c->mem));
    const Affine a64 = lcg_pow(inc, 64);

    uint64_t pos = 0;  // stream position (32-bit draws consumed so far, the buffered half included)
    uint64_t S = 0;    // Fisher-Yates steps completed
    int guard = 0;
    while (S < total_steps) {
        SC_REQUIRE(++guard < 1000, SC_ERR_STATE, "sc_perm_generate: no progress");
        // draws for the remaining steps: expectation + 0.3% + slack, capped at 2^31 per segment
        double want = (double)(total_steps - S) / (double)M * e_d * 1.003 + 262144.0;
        uint64_t seg = want > 2147483648.0 ? 2147483648ULL : (uint64_t)want;
        // raw buffer: index `lo` is the first stream draw of this segment
        uint64_t m_start, lo;
        uint32_t e0;
        if (pos < h) {  // the stream starts with the buffered half word
            m_start = 0; e0 = 1; lo = 0;
        } else {
            m_start = (pos - h) / 2; e0 = 0; lo = (pos - h) & 1;
        }
        const uint64_t hi = lo + seg;
        const uint64_t n_out = (hi - e0 + 1) / 2 + 1;
        const uint64_t raw_len = align_up64((int64_t)(e0 + 2 * n_out), SCAN_BLOCK) + SCAN_BLOCK;
        SC_TRY(c->pg_raw.ensure(sizeof(uint32_t) * (size_t)raw_len, &c->mem));
        if (pos < h) {
            uint32_t bufv = (uint32_t)state6[5];
            SC_HIP(hipMemcpyAsync(c->pg_raw.p, &bufv, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        }
        const uint64_t waves = (n_out + 64 * RAW_ROUNDS - 1) / (64 * RAW_ROUNDS);
        hipLaunchKernelGGL(k_raw_stream, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c->stream,
                           (uint64_t)(state0 >> 64), (uint64_t)state0, (uint64_t)(inc >> 64), (uint64_t)inc, m_start,
                           n_out, e0, (uint64_t)(a64.mult >> 64), (uint64_t)a64.mult, (uint64_t)(a64.plus >> 64),
                           (uint64_t)a64.plus, c->pg_raw.as<uint32_t>());
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_THREADS), 0, c->stream, c->pg_raw.as<uint32_t>(), lo, hi,
                           (uint32_t)n, total_steps, S, c->pg_J.as<int32_t>(), c->pg_out.as<unsigned long long>());
        SC_HIP(hipGetLastError());
        unsigned long long out[3];
        SC_HIP(hipMemcpyAsync(out, c->pg_out.p, sizeof(out), hipMemcpyDeviceToHost, c->stream));
        SC_HIP(hipStreamSynchronize(c->stream));
        SC_REQUIRE(out[2] == 0, SC_ERR_STATE, "sc_perm_generate: rejection scan did not converge");
        SC_REQUIRE(out[0] >= S && out[1] >= lo && out[1] <= hi, SC_ERR_STATE, "sc_perm_generate: bad scan result");
        pos += out[1] - lo;
        S = out[0];
    }
    hipLaunchKernelGGL(k_apply_swaps, dim3((unsigned)n_perm), dim3(64), 0, c->stream, c->pg_J.as<int32_t>(),
                       c->perm.as<int32_t>(), c->p_stride, (uint32_t)n, n_perm);
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(c->stream));

    // final generator state: `pos` draws were consumed
    if (pos > h) {
        const uint64_t tl = pos - 1;           // last consumed position (>= h)
        const uint64_t m_last = (tl - h) / 2;  // its 64-bit output
        const Affine a = lcg_pow(inc, m_last + 1);
        const u128 s = a.mult * state0 + a.plus;
        state6[0] = (uint64_t)(s >> 64);
        state6[1] = (uint64_t)s;
        state6[4] = ((tl - h) & 1) == 0 ? 1 : 0;  // low half consumed -> high half buffered
        state6[5] = (uint32_t)(xsl_rr(s) >> 32);
    } else if (pos == 1 && h == 1) {
        state6[4] = 0;  // only the buffered half was consumed; uinteger keeps its value
    }
    return SC_OK;
}
