// A4: numpy-exact permutation source.  gfx950 only.
//
// The reference draws its permutations from numpy: `rng = np.random.default_rng(seed)` followed by
// `rng.permutation(n)` / `rng.permutation(values)` per permutation (autocorrelation.py:839,879,
// 1109,324, 1367,1404; squidpy's _score_helper likewise).  numpy's algorithm (Generator.shuffle ->
// random_interval on PCG64): PCG64 = 128-bit LCG (mult 0x2360ed051fc65da44385df649fccf645) with the
// XSL-RR 64-bit output; 32-bit draws take the low half first and buffer the high half; a bounded
// draw on [0, i] masks with the next power of two minus one and rejects values > i; the shuffle is
// `for i = n-1 .. 1: j = interval(i); swap(a[i], a[j])`.
//
// The stream is sequential (rejections make the draws per permutation data dependent).  The host
// generator below is the simple exact form (sc_perm_numpy_host); the device table is produced by
// the parallel exact generator of sc_permgen.hip.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sc_ctx.h"

typedef unsigned __int128 u128;

namespace {

struct Pcg64 {
    u128 state, inc;
    int has32;
    uint32_t buf;
};

const u128 kMult = ((u128)0x2360ed051fc65da4ULL << 64) | 0x4385df649fccf645ULL;

inline uint64_t next64(Pcg64 &g)
{
    g.state = g.state * kMult + g.inc;
    uint64_t hi = (uint64_t)(g.state >> 64), lo = (uint64_t)g.state;
    uint64_t x = hi ^ lo;
    unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64 - r) & 63));
}

inline uint32_t next32(Pcg64 &g)
{
    if (g.has32) {
        g.has32 = 0;
        return g.buf;
    }
    uint64_t v = next64(g);
    g.has32 = 1;
    g.buf = (uint32_t)(v >> 32);
    return (uint32_t)v;
}

inline uint32_t bounded32(Pcg64 &g, uint32_t mx, uint32_t mask)
{
    uint32_t v;
    while ((v = next32(g) & mask) > mx) {}
    return v;
}

void load(Pcg64 &g, const uint64_t *s)
{
    g.state = ((u128)s[0] << 64) | s[1];
    g.inc = ((u128)s[2] << 64) | s[3];
    g.has32 = s[4] != 0;
    g.buf = (uint32_t)s[5];
}

void store(const Pcg64 &g, uint64_t *s)
{
    s[0] = (uint64_t)(g.state >> 64);
    s[1] = (uint64_t)g.state;
    s[2] = (uint64_t)(g.inc >> 64);
    s[3] = (uint64_t)g.inc;
    s[4] = (uint64_t)g.has32;
    s[5] = g.buf;
}

// one permutation of length n into a[0..n)
void shuffle_one(Pcg64 &g, int32_t *a, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) a[i] = (int32_t)i;
    if (n < 2) return;
    uint32_t mask = (uint32_t)(n - 1);
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t i = n - 1; i >= 1; --i) {
        // the mask only shrinks when i drops below a power of two
        while ((mask >> 1) >= (uint32_t)i) mask >>= 1;
        uint32_t j = bounded32(g, (uint32_t)i, mask);
        int32_t t = a[j];
        a[j] = a[i];
        a[i] = t;
    }
}

}  // namespace

extern "C" int sc_perm_numpy_host(uint64_t *state6, int64_t n, int64_t n_perm, int32_t *perm_out)
{
    SC_REQUIRE(state6 && (perm_out || n_perm == 0 || n == 0), SC_ERR_INVALID, "sc_perm_numpy_host: null pointer");
    SC_REQUIRE(n >= 0 && n <= 0x7fffffffLL && n_perm >= 0, SC_ERR_INVALID, "sc_perm_numpy_host: bad sizes");
    Pcg64 g;
    load(g, state6);
    for (int64_t p = 0; p < n_perm; ++p) shuffle_one(g, perm_out + p * n, n);
    store(g, state6);
    return SC_OK;
}

__global__ __launch_bounds__(256) void k_check_perm(const int32_t *__restrict__ perm, int64_t n, int64_t stride,
                                                    int64_t rows, int *__restrict__ flag)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = rows * n;
    int bad = 0;
    for (; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t / n, i = t - r * n;
        int32_t v = perm[r * stride + i];
        if (v < 0 || v >= n) bad = 1;
    }
    if (bad) atomicOr(flag, 1);
}

int sc_perm_alloc(sc_ctx *c, int64_t n, int64_t n_perm)
{
    SC_REQUIRE(n >= 1 && n <= 0x7fffffffLL, SC_ERR_INVALID, "permutation length %lld out of range", (long long)n);
    SC_REQUIRE(n_perm >= 1, SC_ERR_INVALID, "n_perm must be >= 1");
    sc_perm_pipe_abort(c);   // a generator job begun with sc_moran_seeded_begin and never finished owns the table
    int64_t stride = align_up64(n, 32);
    // +32 elements of slack so that the 8-wide tail reads of the last row stay inside the buffer
    SC_TRY(c->perm.ensure(sizeof(int32_t) * (size_t)(stride * n_perm + 32), &c->mem));
    c->p_n = n;
    c->p_count = 0;
    c->p_stride = stride;
    c->perm_bijective = false;
    c->perm_checked = false;
    c->perm_forward_valid = true;
    c->inv_rows_valid = 0;
    return SC_OK;
}

extern "C" int sc_perm_set(sc_ctx *c, const int32_t *perm, int64_t n, int64_t n_perm)
{
    SC_REQUIRE(c && perm, SC_ERR_INVALID, "sc_perm_set: null pointer");
    SC_HIP(hipSetDevice(c->device));
    SC_TRY(sc_perm_alloc(c, n, n_perm));
    SC_TRY(c->perm_flag.ensure(sizeof(unsigned long long), &c->mem));
    SC_HIP(hipMemsetAsync(c->perm.p, 0, sizeof(int32_t) * (size_t)(c->p_stride * n_perm + 32), c->stream));
    SC_HIP(hipMemcpy2DAsync(c->perm.p, sizeof(int32_t) * (size_t)c->p_stride, perm, sizeof(int32_t) * (size_t)n,
                            sizeof(int32_t) * (size_t)n, (size_t)n_perm, hipMemcpyHostToDevice, c->stream));
    // every index must be a valid cell: the gather kernels trust the table
    SC_HIP(hipMemsetAsync(c->perm_flag.p, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_check_perm, dim3(2048), dim3(256), 0, c->stream, c->perm.as<int32_t>(), n, c->p_stride,
                       n_perm, c->perm_flag.as<int>());
    int flag = 0;
    SC_HIP(hipMemcpyAsync(&flag, c->perm_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    SC_REQUIRE(!flag, SC_ERR_INVALID, "sc_perm_set: table contains an index outside [0, %lld)", (long long)n);
    c->p_count = n_perm;
    return SC_OK;
}

extern "C" int sc_perm_generate(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm, int32_t *perm_out)
{
    SC_REQUIRE(c && state6, SC_ERR_INVALID, "sc_perm_generate: null pointer");
    SC_HIP(hipSetDevice(c->device));
    SC_TRY(sc_perm_alloc(c, n, n_perm));
    {
        KernelTimerScope ts(c, SC_K_PERMGEN);
        SC_TRY(sc_perm_generate_device(c, state6, n, n_perm));
    }
    c->p_count = n_perm;
    c->perm_bijective = true;  // generated rows are permutations by construction
    if (perm_out) {
        SC_HIP(hipMemcpy2DAsync(perm_out, sizeof(int32_t) * (size_t)n, c->perm.p, sizeof(int32_t) * (size_t)c->p_stride,
                                sizeof(int32_t) * (size_t)n, (size_t)n_perm, hipMemcpyDeviceToHost, c->stream));
        SC_HIP(hipStreamSynchronize(c->stream));
    }
    return SC_OK;
}
