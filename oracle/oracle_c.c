/*
 * oracle_c.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, plain C, scalar, single thread).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (spatialcore_amd/) never does.
 *
 * It restates, for the Moran / Lee permutation hot path:
 *
 *  - the permutation source.  The reference draws `np.random.default_rng(seed)` and then
 *    `rng.permutation(n_cells)` / `rng.permutation(z_y)` once per permutation
 *    (reference src/spatialcore/spatial/autocorrelation.py:839,879 ; :1109,324 ; :1367,1404;
 *    squidpy's `_score_helper` does the same with `default_rng(seed + ix)`).  The algorithm lives
 *    in numpy (installed here: 2.2.6), not in /root/reference: PCG64 (128-bit LCG, XSL-RR 64-bit
 *    output), 32-bit draws taken low half first with the high half buffered, masked rejection
 *    sampling `random_interval`, and a reverse Fisher-Yates shuffle `for i = n-1 .. 1`.
 *    Pinned by the known-answer vectors in tests/golden/rng_kat.npz, which were produced by numpy
 *    itself (oracle/make_golden.py).
 *
 *  - the global Moran's I kernel that squidpy -> scanpy.metrics.morans_i runs for the reference's
 *    `morans_i` (autocorrelation.py:576-583): I = N/W * sum_i z_i * (sum_j w_ij z_j) / sum_i z_i^2
 *    with sequential fp64 accumulation, in the literal "permute the graph rows" form and in the
 *    algebraically identical gather form sum_i z_i * lag[perm[i]].  squidpy/scanpy are NOT
 *    importable here (un-vendored dependency, pyproject.toml:39 `squidpy>=1.3.0`): this part is
 *    "parity unpinned" except through the cross-checks listed in DESIGN.md.
 *
 *  - the Lee's L permutation loop of `_compute_lees_l_core` (autocorrelation.py:322-332).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

typedef struct {
    u128 state;
    u128 inc;
    int has_uint32;
    uint32_t uinteger;
} pcg64_t;

static const uint64_t PCG_MULT_HI = 0x2360ed051fc65da4ULL;
static const uint64_t PCG_MULT_LO = 0x4385df649fccf645ULL;

static inline uint64_t rotr64(uint64_t v, unsigned r) { return (v >> r) | (v << ((-r) & 63)); }

static inline uint64_t pcg64_next64(pcg64_t *g)
{
    const u128 mult = ((u128)PCG_MULT_HI << 64) | PCG_MULT_LO;
    g->state = g->state * mult + g->inc;
    uint64_t hi = (uint64_t)(g->state >> 64), lo = (uint64_t)g->state;
    return rotr64(hi ^ lo, (unsigned)(hi >> 58));
}

static inline uint32_t pcg64_next32(pcg64_t *g)
{
    if (g->has_uint32) {
        g->has_uint32 = 0;
        return g->uinteger;
    }
    uint64_t v = pcg64_next64(g);
    g->has_uint32 = 1;
    g->uinteger = (uint32_t)(v >> 32);
    return (uint32_t)v;
}

/* numpy random_interval(): uniform integer in [0, max] by masked rejection. */
static inline uint64_t random_interval(pcg64_t *g, uint64_t max)
{
    if (max == 0) return 0;
    uint64_t mask = max, value;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
    mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    if (max <= 0xffffffffULL) {
        while ((value = (pcg64_next32(g) & mask)) > max) {}
    } else {
        while ((value = (pcg64_next64(g) & mask)) > max) {}
    }
    return value;
}

static void load_state(pcg64_t *g, const uint64_t *st)
{
    g->state = ((u128)st[0] << 64) | st[1];
    g->inc = ((u128)st[2] << 64) | st[3];
    g->has_uint32 = (int)st[4];
    g->uinteger = (uint32_t)st[5];
}

static void store_state(const pcg64_t *g, uint64_t *st)
{
    st[0] = (uint64_t)(g->state >> 64);
    st[1] = (uint64_t)g->state;
    st[2] = (uint64_t)(g->inc >> 64);
    st[3] = (uint64_t)g->inc;
    st[4] = (uint64_t)g->has_uint32;
    st[5] = g->uinteger;
}

/*
 * st[6] = {state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger}, updated in place.
 * Writes n_perm consecutive `rng.permutation(n)` results into out[p*n + i] (int32).
 */
int orc_perm_numpy(uint64_t *st, int64_t n, int64_t n_perm, int32_t *out)
{
    if (n < 0 || n > 0x7fffffffLL || n_perm < 0) return 1;
    pcg64_t g;
    load_state(&g, st);
    for (int64_t p = 0; p < n_perm; ++p) {
        int32_t *a = out + p * n;
        for (int64_t i = 0; i < n; ++i) a[i] = (int32_t)i;
        for (int64_t i = n - 1; i >= 1; --i) {
            int64_t j = (int64_t)random_interval(&g, (uint64_t)i);
            int32_t t = a[j]; a[j] = a[i]; a[i] = t;
        }
    }
    store_state(&g, st);
    return 0;
}

/* n32 raw 32-bit draws (for the stream known-answer test). */
int orc_raw_uint32(uint64_t *st, int64_t n32, uint32_t *out)
{
    pcg64_t g;
    load_state(&g, st);
    for (int64_t i = 0; i < n32; ++i) out[i] = pcg64_next32(&g);
    store_state(&g, st);
    return 0;
}

/* threads used by the two Moran kernels below (results do not depend on it); returns the count in effect */
int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* lag[i] = sum_j data[j]*z[indices[j]] over row i, sequential fp64 (scanpy _morans_i_vec_W inner loop). */
void orc_csr_lag(const int64_t *indptr, const int32_t *indices, const double *data,
                 int64_t n, const double *z, double *lag)
{
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) s += data[e] * z[indices[e]];
        lag[i] = s;
    }
}

/*
 * Literal squidpy form for one permutation and G genes: the graph's rows are re-ordered
 * (g[idx, :]) and the CSR sweep is redone:  inum = sum_i z[i] * (sum_e data_{idx[i]}[e] * z[col]).
 * vals is gene-major [G][n] (raw expression, fp64); out[g] = n/W * inum / sum z^2.
 */
void orc_moran_rowperm(const int64_t *indptr, const int32_t *indices, const double *data,
                       int64_t n, const double *vals, int64_t n_genes, const int32_t *idx,
                       double *out)
{
    double W = 0.0;
    for (int64_t e = 0; e < indptr[n]; ++e) W += data[e];
    /* genes in parallel, one gene's arithmetic strictly sequential -- the shape of scanpy's
     * @njit(parallel=True) kernel (prange over genes); orc_set_threads(1) gives the n_jobs=1 figure */
#pragma omp parallel
    {
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
#pragma omp for schedule(dynamic, 1)
    for (int64_t g = 0; g < n_genes; ++g) {
        const double *x = vals + g * n;
        double m = 0.0;
        for (int64_t i = 0; i < n; ++i) m += x[i];
        m /= (double)n;
        double z2 = 0.0;
        for (int64_t i = 0; i < n; ++i) { z[i] = x[i] - m; z2 += z[i] * z[i]; }
        double inum = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            int64_t r = idx ? idx[i] : i;
            double s = 0.0;
            for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) s += data[e] * z[indices[e]];
            inum += s * z[i];
        }
        out[g] = (double)n / W * inum / z2;
    }
    free(z);
    }
}

/* Gather form: sims[p*G+g] = scale[g] * sum_i z_g[i] * lag_g[perm_p[i]]; z, lag gene-major [G][n]. */
void orc_gather_dot(const double *z, const double *lag, int64_t n, int64_t n_genes,
                    const int32_t *perm, int64_t n_perm, const double *scale, double *sims)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t p = 0; p < n_perm; ++p) {
        for (int64_t g = 0; g < n_genes; ++g) {
            const int32_t *pi = perm + p * n;
            const double *zg = z + g * n, *lg = lag + g * n;
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += zg[i] * lg[pi[i]];
            sims[p * n_genes + g] = scale[g] * s;
        }
    }
}

/*
 * Integer-lattice form of the same sum for integer counts on a graph whose weights are all equal: with S = A x (A = the
 * 0/1 adjacency) the permutation-dependent part of sum_i z[i] lag[perm[i]] is T = sum_i x[i] * S[perm[i]], an exact
 * integer.  T[p*G+g]; perm == NULL: identity (the observed statistic).  x, S gene-major [G][n] int64.
 */
void orc_lattice_T(const int64_t *x, const int64_t *S, int64_t n, int64_t n_genes, const int32_t *perm,
                   int64_t n_perm, int64_t *T)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t p = 0; p < n_perm; ++p) {
        for (int64_t g = 0; g < n_genes; ++g) {
            const int32_t *pi = perm ? perm + p * n : NULL;
            const int64_t *xg = x + g * n, *sg = S + g * n;
            int64_t t = 0;
            if (pi)
                for (int64_t i = 0; i < n; ++i) t += xg[i] * sg[pi[i]];
            else
                for (int64_t i = 0; i < n; ++i) t += xg[i] * sg[i];
            T[p * n_genes + g] = t;
        }
    }
}

/*
 * Lee's L permutation loop, reference autocorrelation.py:322-328, literal form:
 * z_y is shuffled by the SAME swaps rng.permutation(z_y) applies, then lag = W @ z_y_perm,
 * L_perm = sum_i z_x[i]*lag[i].  Stream state st[] carries over between calls (one rng for all
 * pairs, autocorrelation.py:1109).
 */
int orc_lee_perm_literal(uint64_t *st, const int64_t *indptr, const int32_t *indices,
                         const double *data, int64_t n, const double *zx, const double *zy,
                         int64_t n_perm, double *L_perm)
{
    pcg64_t g;
    load_state(&g, st);
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    if (!a) return 2;
    for (int64_t p = 0; p < n_perm; ++p) {
        memcpy(a, zy, sizeof(double) * (size_t)n);
        for (int64_t i = n - 1; i >= 1; --i) {
            int64_t j = (int64_t)random_interval(&g, (uint64_t)i);
            double t = a[j]; a[j] = a[i]; a[i] = t;
        }
        double L = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double s = 0.0;
            for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) s += data[e] * a[indices[e]];
            L += zx[i] * s;
        }
        L_perm[p] = L;
    }
    free(a);
    store_state(&g, st);
    return 0;
}

/* Exhaustive kNN in 2-D: rdist = fl(fl(dx*dx) + fl(dy*dy)); order by (rdist, index); self excluded
 * by index unless include_self.  O(n^2); small n only.  Build with -ffp-contract=off. */
int orc_knn_bruteforce(const double *xy, int64_t n, int k, int include_self, int32_t *idx_out,
                       double *rdist_out)
{
    if (k < 1 || (int64_t)k > n - (include_self ? 0 : 1)) return 1;
    double *bd = (double *)malloc(sizeof(double) * (size_t)k);
    int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    for (int64_t q = 0; q < n; ++q) {
        int cnt = 0;
        double qx = xy[2 * q], qy = xy[2 * q + 1];
        for (int64_t c = 0; c < n; ++c) {
            if (!include_self && c == q) continue;
            double dx = qx - xy[2 * c], dy = qy - xy[2 * c + 1];
            double dx2 = dx * dx, dy2 = dy * dy;
            double d = dx2 + dy2;
            if (cnt == k && !(d < bd[k - 1])) continue; /* ties keep the lower index (already in) */
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && bd[pos - 1] > d) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
            bd[pos] = d; bi[pos] = (int32_t)c;
            if (cnt < k) ++cnt;
        }
        for (int j = 0; j < k; ++j) {
            idx_out[q * k + j] = bi[j];
            if (rdist_out) rdist_out[q * k + j] = bd[j];
        }
    }
    free(bd); free(bi);
    return 0;
}
