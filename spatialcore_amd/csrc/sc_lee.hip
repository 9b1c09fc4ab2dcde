// A8 at scale: Lee's L for many gene pairs in ONE call (BASELINE configs[2]: 100 x 100 pairs, 1M cells).  gfx950 only.
//
// Reference: lees_l (AC:1113-1155) loops over the pairs in Python; per pair it standardises the two columns,
// takes L = sum_i z_x[i] (W z_y)[i] (AC:307-315) and then P times shuffles z_y with ONE generator shared by all pairs
// and redoes the sparse mat-vec (AC:322-328); p = (#{|L_perm| >= |L|} + 1) / (P + 1) (AC:331-332).
// Here, for all pairs at once:
//   * every distinct gene is standardised once (fp64 z-scores, population sd), lag = W Z and U = W^T Z are one
//     tile-SpMM each;
//   * the observed statistics are a dense contraction over the cells, L[x][y] = sum_i Z[i][x] Lag[i][y]: the one true
//     GEMM on the path, done 16 x 16 genes at a time with v_mfma_f64_16x16x4_f64 straight from the 128-byte tile rows
//     (lane l loads element [cell l >> 4][gene l & 15] of both operands: a fully coalesced 512-byte load per MFMA);
//   * the permutation statistic of pair (x, y) is the gather-dot sum_j U[j][x] z_y[perm[j]]; the numpy-exact
//     rows -- a fresh block of P per live pair, in pair order, zero-variance pairs draw nothing (AC:1129-1140) -- come
//     out of the same generator pipeline as sc_moran_seeded and are scored chunk by chunk while the generator runs; the
//     counts are reduced on the device (no host round trip per pair).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "sc_ctx.h"

// out[k][cell] = T[tile(genes[k])][cell][slot(genes[k])]: gene-major contiguous copies of the genes a kernel gathers from
__global__ __launch_bounds__(256) void k_gene_major(const double *__restrict__ T, int64_t n, const int32_t *__restrict__ genes,
                                                    double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t g = genes[blockIdx.y];
    out[(int64_t)blockIdx.y * n + i] = T[(int64_t)(g >> 4) * n * SC_TILE + i * SC_TILE + (g & 15)];
}

// ---- observed statistics: C[tile pair][16 x][16 y] = sum over cells of Z_xtile[cell][x] * Lag_ytile[cell][y] ----
#define LEE_OBS_CELLS 16384   // cells per workgroup (4 wavefronts x 4096)

typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_lee_observed_mfma(const double *__restrict__ Z, const double *__restrict__ Lag,
                                                           int64_t n, const int2 *__restrict__ tile_pairs,
                                                           double *__restrict__ partial)
{
    __shared__ double red[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int2 tp = tile_pairs[blockIdx.y];
    const double *A = Z + (int64_t)tp.x * n * SC_TILE, *B = Lag + (int64_t)tp.y * n * SC_TILE;
    const int64_t c0 = (int64_t)blockIdx.x * LEE_OBS_CELLS + (int64_t)wave * (LEE_OBS_CELLS / 4);
    int64_t c1 = c0 + LEE_OBS_CELLS / 4;
    if (c1 > n) c1 = n;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    // A[i = gene x = lane & 15][k = cell lane >> 4], B[k = cell lane >> 4][j = gene y = lane & 15]: both are element
    // [cell][gene] of a tile row, i.e. word `lane` of the 4-row block
    for (int64_t c = c0; c < c1; c += 4) {
        const int64_t cell = c + (lane >> 4);
        const double a = cell < c1 ? A[cell * SC_TILE + (lane & 15)] : 0.0;
        const double b = cell < c1 ? B[cell * SC_TILE + (lane & 15)] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    // D[row = (lane >> 4) + 4 reg][col = lane & 15]  (row = gene x, col = gene y)
#pragma unroll
    for (int v = 0; v < 4; ++v) red[wave][((lane >> 4) + 4 * v) * 16 + (lane & 15)] = acc[v];
    __syncthreads();
    const int t = threadIdx.x;
    partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// obs[pair] = sum over cell blocks (ascending) of the pair's element of its tile pair
__global__ __launch_bounds__(256) void k_lee_observed_pick(const double *__restrict__ partial, int blocks,
                                                           const int32_t *__restrict__ pair_tp,
                                                           const int32_t *__restrict__ pair_x,
                                                           const int32_t *__restrict__ pair_y, int64_t n_pairs,
                                                           double *__restrict__ obs)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pairs) return;
    const int tp = pair_tp[q];
    if (tp < 0) { obs[q] = 0.0; return; }
    const int e = (pair_x[q] & 15) * 16 + (pair_y[q] & 15);
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += partial[((int64_t)tp * blocks + b) * 256 + e];
    obs[q] = s;
}

// ---- permutation statistic of a chunk of rows: partial[row][block] = sum_{j in block} U[x(row)][j] * Zy[y(row)][perm_row[j]] ----
#define LEE_PERM_CELLS 8192

__global__ __launch_bounds__(256) void k_lee_rows(const double *__restrict__ Uc, const double *__restrict__ Zc, int64_t n,
                                                  const int32_t *__restrict__ perm, int64_t pstride,
                                                  const int2 *__restrict__ row_slots, double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int row = blockIdx.y;
    const int2 sl = row_slots[row];          // (slot of u_x in Uc, slot of z_y in Zc)
    const double *u = Uc + (int64_t)sl.x * n, *z = Zc + (int64_t)sl.y * n;
    const int32_t *prow = perm + (int64_t)row * pstride;
    const int64_t j0 = (int64_t)blockIdx.x * LEE_PERM_CELLS;
    const int64_t j1 = j0 + LEE_PERM_CELLS < n ? j0 + LEE_PERM_CELLS : n;
    double acc = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) acc = fma(u[j], z[prow[j]], acc);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)row * gridDim.x + blockIdx.x] = sh[0];
}

// L_perm[row] = sum_b partial[row][b] (ascending); count[pair(row)] += |L_perm| >= |obs[pair]| (integer: exact, order-free)
__global__ __launch_bounds__(256) void k_lee_rows_count(const double *__restrict__ partial, int blocks, int rows,
                                                        const int32_t *__restrict__ row_pair,
                                                        const double *__restrict__ obs,
                                                        unsigned long long *__restrict__ count,
                                                        double *__restrict__ lperm_out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += partial[(int64_t)r * blocks + b];
    const int q = row_pair[r];
    if (fabs(s) >= fabs(obs[q])) atomicAdd(&count[q], 1ull);
    if (lperm_out) lperm_out[r] = s;
}

extern "C" int sc_lee_seeded(sc_ctx *c, uint64_t *state6, const int32_t *pair_x, const int32_t *pair_y, int64_t n_pairs,
                             int64_t n_perm, double *L_out, int64_t *count_abs_ge_out, double *L_perm_out)
{
    SC_REQUIRE(c && pair_x && pair_y && L_out && count_abs_ge_out, SC_ERR_INVALID, "sc_lee_seeded: null pointer");
    SC_REQUIRE(n_pairs >= 0 && n_perm >= 0, SC_ERR_INVALID, "sc_lee_seeded: negative size");
    SC_REQUIRE(n_perm == 0 || state6, SC_ERR_INVALID, "sc_lee_seeded: generator state required when n_perm > 0");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "sc_lee_seeded: no expression loaded");
    SC_REQUIRE(c->g_n == c->e_n, SC_ERR_STATE, "sc_lee_seeded: graph missing or size mismatch");
    const int64_t n = c->e_n, T = c->e_tiles, G = c->e_genes;
    for (int64_t q = 0; q < n_pairs; ++q)
        SC_REQUIRE(pair_x[q] >= 0 && pair_x[q] < G && pair_y[q] >= 0 && pair_y[q] < G, SC_ERR_INVALID,
                   "sc_lee_seeded: pair %lld references a gene outside the loaded set", (long long)q);
    if (n_pairs == 0) return SC_OK;
    const size_t tile_bytes = (size_t)n * SC_TILE * sizeof(double);

    // z-scores, lag = W Z, U = W^T Z for every loaded gene
    SC_TRY(sc_expr_zscores(c));
    SC_TRY(c->Lag.ensure((size_t)T * tile_bytes, &c->mem));
    SC_TRY(sc_lag_tiles(c, c->g_indptr, c->g_indices, c->g_data, c->Z.as<double>(), c->Lag.as<double>()));
    std::vector<double> var((size_t)G);
    SC_HIP(hipMemcpyAsync(var.data(), c->g_var.p, sizeof(double) * var.size(), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));

    // ---- observed L of every pair: MFMA contraction per distinct (x tile, y tile) ----
    std::vector<int32_t> pair_tp((size_t)n_pairs, -1), live;
    std::vector<int2> tps;
    {
        std::vector<int64_t> keys;
        for (int64_t q = 0; q < n_pairs; ++q)
            if (var[(size_t)pair_x[q]] > 0.0 && var[(size_t)pair_y[q]] > 0.0) {
                live.push_back((int32_t)q);
                keys.push_back((int64_t)(pair_x[q] >> 4) * T + (pair_y[q] >> 4));
            }
        std::vector<int64_t> uniq(keys);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        for (int64_t k : uniq) tps.push_back(make_int2((int)(k / T), (int)(k % T)));
        for (size_t i = 0; i < live.size(); ++i)
            pair_tp[(size_t)live[i]] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), keys[i]) - uniq.begin());
    }
    const int64_t n_live = (int64_t)live.size();
    SC_TRY(c->lee_obs.ensure(sizeof(double) * (size_t)n_pairs, &c->mem));
    SC_TRY(c->lee_cnt.ensure(sizeof(unsigned long long) * (size_t)n_pairs * 2, &c->mem));   // counts + roll-back copy
    unsigned long long *d_cnt_backup = c->lee_cnt.as<unsigned long long>() + n_pairs;
    SC_HIP(hipMemsetAsync(c->lee_cnt.p, 0, sizeof(unsigned long long) * (size_t)n_pairs, c->stream));
    // small index arrays: [pair_tp | pair_x | pair_y] then the tile pairs
    SC_TRY(c->lee_pairs.ensure(sizeof(int32_t) * (size_t)(3 * n_pairs) + sizeof(int2) * (tps.size() + 1), &c->mem));
    int32_t *d_tp = c->lee_pairs.as<int32_t>(), *d_px = d_tp + n_pairs, *d_py = d_px + n_pairs;
    int2 *d_tps = reinterpret_cast<int2 *>(d_py + n_pairs + (n_pairs & 1));
    SC_HIP(hipMemcpyAsync(d_tp, pair_tp.data(), sizeof(int32_t) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(d_px, pair_x, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(d_py, pair_y, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    const int oblocks = (int)ceil_div64(n, LEE_OBS_CELLS);
    if (!tps.empty()) {
        SC_HIP(hipMemcpyAsync(d_tps, tps.data(), sizeof(int2) * tps.size(), hipMemcpyHostToDevice, c->stream));
        SC_TRY(c->lee_part.ensure(sizeof(double) * tps.size() * (size_t)oblocks * 256, &c->mem));
        hipLaunchKernelGGL(k_lee_observed_mfma, dim3((unsigned)oblocks, (unsigned)tps.size()), dim3(256), 0, c->stream,
                           c->Z.as<double>(), c->Lag.as<double>(), n, d_tps, c->lee_part.as<double>());
    }
    hipLaunchKernelGGL(k_lee_observed_pick, dim3((unsigned)ceil_div64(n_pairs, 256)), dim3(256), 0, c->stream,
                       c->lee_part.as<double>(), oblocks, d_tp, d_px, d_py, n_pairs, c->lee_obs.as<double>());
    SC_HIP(hipGetLastError());

    if (n_perm > 0 && n_live > 0) {
        // ---- U = W^T Z (tiles), gene-major copies of the u_x / z_y vectors the permutation kernel reads ----
        SC_TRY(sc_graph_ensure_transpose(c));
        SC_TRY(c->lee_U.ensure((size_t)T * tile_bytes, &c->mem));
        SC_TRY(sc_lag_tiles(c, c->gt_indptr, c->gt_indices, c->gt_data, c->Z.as<double>(), c->lee_U.as<double>()));
        std::vector<int32_t> xs, ys, slot_x((size_t)G, -1), slot_y((size_t)G, -1);
        for (int32_t q : live) {
            if (slot_x[(size_t)pair_x[q]] < 0) { slot_x[(size_t)pair_x[q]] = (int32_t)xs.size(); xs.push_back(pair_x[q]); }
            if (slot_y[(size_t)pair_y[q]] < 0) { slot_y[(size_t)pair_y[q]] = (int32_t)ys.size(); ys.push_back(pair_y[q]); }
        }
        SC_TRY(c->lee_Uc.ensure(sizeof(double) * xs.size() * (size_t)n, &c->mem));
        SC_TRY(c->lee_Zc.ensure(sizeof(double) * ys.size() * (size_t)n, &c->mem));
        SC_TRY(c->lee_a.ensure(sizeof(int32_t) * (xs.size() + ys.size()), &c->mem));
        int32_t *d_xs = c->lee_a.as<int32_t>(), *d_ys = d_xs + xs.size();
        SC_HIP(hipMemcpyAsync(d_xs, xs.data(), sizeof(int32_t) * xs.size(), hipMemcpyHostToDevice, c->stream));
        SC_HIP(hipMemcpyAsync(d_ys, ys.data(), sizeof(int32_t) * ys.size(), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gene_major, dim3((unsigned)ceil_div64(n, 256), (unsigned)xs.size()), dim3(256), 0, c->stream,
                           c->lee_U.as<double>(), n, d_xs, c->lee_Uc.as<double>());
        hipLaunchKernelGGL(k_gene_major, dim3((unsigned)ceil_div64(n, 256), (unsigned)ys.size()), dim3(256), 0, c->stream,
                           c->Z.as<double>(), n, d_ys, c->lee_Zc.as<double>());
        SC_HIP(hipGetLastError());
        SC_HIP(hipStreamSynchronize(c->stream));  // xs / ys / tps are host vectors

        // ---- permutations: sub-jobs of whole pairs, each a generator / scoring pipeline that continues the stream ----
        const int64_t max_rows = std::max<int64_t>(n_perm, (int64_t)(2.2e9 / (double)n));   // ~40 GB of generator scratch
        const int64_t pairs_per_job = std::max<int64_t>(1, max_rows / n_perm);
        const int pblocks = (int)ceil_div64(n, LEE_PERM_CELLS);
        const int64_t job_rows_max = std::min(n_live, pairs_per_job) * n_perm;
        SC_TRY(c->lee_rowmap.ensure((sizeof(int2) + sizeof(int32_t)) * (size_t)job_rows_max, &c->mem));
        SC_TRY(c->lee_b.ensure(sizeof(double) * (size_t)PERM_CHUNK * (size_t)pblocks, &c->mem));
        if (L_perm_out) SC_TRY(c->lee_lperm.ensure(sizeof(double) * (size_t)job_rows_max, &c->mem));
        int2 *d_slots = c->lee_rowmap.as<int2>();
        int32_t *d_rowpair = reinterpret_cast<int32_t *>(d_slots + job_rows_max);
        for (int64_t l0 = 0; l0 < n_live; l0 += pairs_per_job) {
            const int64_t l1 = std::min(n_live, l0 + pairs_per_job), rows = (l1 - l0) * n_perm;
            std::vector<int2> slots((size_t)rows);
            std::vector<int32_t> rowpair((size_t)rows);
            for (int64_t l = l0; l < l1; ++l)
                for (int64_t p = 0; p < n_perm; ++p) {
                    const int32_t q = live[(size_t)l];
                    slots[(size_t)((l - l0) * n_perm + p)] = make_int2(slot_x[(size_t)pair_x[q]], slot_y[(size_t)pair_y[q]]);
                    rowpair[(size_t)((l - l0) * n_perm + p)] = q;
                }
            SC_HIP(hipMemcpyAsync(d_slots, slots.data(), sizeof(int2) * (size_t)rows, hipMemcpyHostToDevice, c->stream));
            SC_HIP(hipMemcpyAsync(d_rowpair, rowpair.data(), sizeof(int32_t) * (size_t)rows, hipMemcpyHostToDevice, c->stream));
            SC_HIP(hipStreamSynchronize(c->stream));
            auto score = [&](int64_t p0, int64_t p1) -> int {
                const int cnt = (int)(p1 - p0);
                {
                    KernelTimerScope ts(c, SC_K_LEE_PERM);
                    hipLaunchKernelGGL(k_lee_rows, dim3((unsigned)pblocks, (unsigned)cnt), dim3(256), 0, c->stream,
                                       c->lee_Uc.as<double>(), c->lee_Zc.as<double>(), n,
                                       c->perm.as<int32_t>() + p0 * c->p_stride, c->p_stride, d_slots + p0,
                                       c->lee_b.as<double>());
                }
                hipLaunchKernelGGL(k_lee_rows_count, dim3((unsigned)ceil_div64(cnt, 256)), dim3(256), 0, c->stream,
                                   c->lee_b.as<double>(), pblocks, cnt, d_rowpair + p0, c->lee_obs.as<double>(),
                                   c->lee_cnt.as<unsigned long long>(),
                                   L_perm_out ? c->lee_lperm.as<double>() + p0 : (double *)nullptr);
                SC_HIP(hipGetLastError());
                return SC_OK;
            };
            // the block-parallel scan verifies itself at the end of a job; if that fails (never seen without the fault
            // injection mode) the sub-job's counts are rolled back and it is rerun with the sequential scan
            SC_HIP(hipMemcpyAsync(d_cnt_backup, c->lee_cnt.p, sizeof(unsigned long long) * (size_t)n_pairs,
                                  hipMemcpyDeviceToDevice, c->stream));
            const int ahead = c->pg_ahead;
            c->pg_ahead = 2;
            int rc = sc_perm_pipeline(c, state6, n, rows, 0, nullptr, score);
            if (rc == SC_PERMGEN_RETRY) {
                SC_HIP(hipMemcpyAsync(c->lee_cnt.p, d_cnt_backup, sizeof(unsigned long long) * (size_t)n_pairs,
                                      hipMemcpyDeviceToDevice, c->stream));
                const int mode = c->pg_mode;
                c->pg_mode = 1;
                rc = sc_perm_pipeline(c, state6, n, rows, 0, nullptr, score);
                c->pg_mode = mode;
            }
            c->pg_ahead = ahead;
            SC_TRY(rc);
            if (L_perm_out) {
                std::vector<double> lp((size_t)rows);
                SC_HIP(hipMemcpy(lp.data(), c->lee_lperm.p, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost));
                for (int64_t l = l0; l < l1; ++l)
                    memcpy(L_perm_out + (int64_t)live[(size_t)l] * n_perm, lp.data() + (l - l0) * n_perm,
                           sizeof(double) * (size_t)n_perm);
            }
        }
    }
    // ---- results ----
    std::vector<unsigned long long> cnt((size_t)n_pairs);
    SC_HIP(hipMemcpyAsync(L_out, c->lee_obs.p, sizeof(double) * (size_t)n_pairs, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(cnt.data(), c->lee_cnt.p, sizeof(unsigned long long) * (size_t)n_pairs, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    for (int64_t q = 0; q < n_pairs; ++q) {
        const bool dead = pair_tp[(size_t)q] < 0;
        if (dead) L_out[q] = 0.0;
        count_abs_ge_out[q] = dead ? n_perm : (int64_t)cnt[(size_t)q];
        if (dead && L_perm_out)
            for (int64_t p = 0; p < n_perm; ++p) L_perm_out[q * n_perm + p] = 0.0;
    }
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// Float32-faithful observed L.  For a float32 matrix the reference computes everything in float32 with numpy's
// summation order (AC:1118-1146, 307-315): mean = S(x) / n, std = sqrt(S(d * d) / n) with d = x - mean, z = d / std,
// lag = W32 @ z_y (scipy csr_matvec: row-sequential, multiply and add rounded separately), L = float(S(z_x * lag)),
// where S is numpy's sum of a contiguous float32 vector: the vector is cut into chunks of 8192 elements (the ufunc
// buffer size), every chunk is summed PAIRWISE (blocks of <= 128 with 8 strided accumulators, halving above that with
// the split rounded down to a multiple of 8) and the chunk sums are accumulated in order (numpy 2.2, verified against
// numpy itself up to 10^6 elements).  A sum of 10^6 signed float32 terms carries ~1e-5 relative rounding noise, so an
// fp64 L differs from the reference's by that much; to hand back the reference's OWN number the same tree is
// evaluated here with the same float roundings -- in parallel: the tree's shape depends on n alone, so one thread sums
// one leaf and one thread per vector replays the recursion over the leaf sums (sc_pairwise.h).  The permutation
// statistics stay fp64 (the p-values of the reference's goldens are reproduced exactly that way).
// ------------------------------------------------------------------------------------------------

#include "sc_pairwise.h"

#define NP_SUM_CHUNK 8192u   // numpy's ufunc buffer size in elements

// leaves[i] = (start, len) of the i-th leaf of numpy's sum over m elements (chunk after chunk); *nleaves
__global__ void k32_leaves(uint32_t m, uint2 *__restrict__ leaves, uint32_t max_leaves, uint32_t *__restrict__ nleaves)
{
    uint32_t k = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += NP_SUM_CHUNK) {
        const uint32_t len_c = m - c0 < NP_SUM_CHUNK ? m - c0 : NP_SUM_CHUNK;
        (void)pw_walk<float>(len_c, [&](uint32_t start, uint32_t len) {
            if (k < max_leaves) leaves[k] = make_uint2(c0 + start, len);
            ++k;
            return 0.f;
        });
    }
    *nleaves = k;
}

// numpy's sum of m float32 terms from the leaf sums `ls` (in leaf order): chunk sums accumulated in order
__device__ __forceinline__ float np_sum_from_leaves(uint32_t m, const float *__restrict__ ls)
{
    uint32_t i = 0;
    float acc = 0.f;
    for (uint32_t c0 = 0; c0 < m; c0 += NP_SUM_CHUNK) {
        const uint32_t len_c = m - c0 < NP_SUM_CHUNK ? m - c0 : NP_SUM_CHUNK;
        const float part = pw_walk<float>(len_c, [&](uint32_t, uint32_t) { return ls[i++]; });
        acc = c0 == 0 ? part : __fadd_rn(acc, part);
    }
    return acc;
}

// STAT 0: leaf sums of x over cells start ..; STAT 1: of fl(d * d), d = fl(x - mean).  thread = (leaf, gene)
template <int STAT>
__global__ __launch_bounds__(256) void k32_gene_leafsum(const double *__restrict__ X, int64_t n,
                                                        const int32_t *__restrict__ genes,
                                                        const float *__restrict__ mean32, const uint2 *__restrict__ leaves,
                                                        uint32_t nleaves, float *__restrict__ leafsum)
{
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= nleaves) return;
    const int32_t g = genes[blockIdx.y];
    const double *col = X + (int64_t)(g >> 4) * n * SC_TILE + (g & 15);
    const uint2 lf = leaves[leaf];
    const float mu = STAT ? mean32[blockIdx.y] : 0.f;
    leafsum[(int64_t)blockIdx.y * nleaves + leaf] = pw_block<float>(lf.y, [&](uint32_t k) {
        const float x = (float)col[(int64_t)(lf.x + k) * SC_TILE];
        if (!STAT) return x;
        const float d = __fsub_rn(x, mu);
        return __fmul_rn(d, d);
    });
}

// STAT 0: mean32[k] = S(x) / n;  STAT 1: sd32[k] = sqrt(S(d * d) / n)   (IEEE float division / sqrt)
template <int STAT>
__global__ void k32_gene_combine(int64_t n, int n_genes, const float *__restrict__ leafsum, uint32_t nleaves,
                                 float *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_genes) return;
    float res = np_sum_from_leaves((uint32_t)n, leafsum + (int64_t)k * nleaves);
    res = (float)__ddiv_rn((double)res, (double)(float)n);   // correctly rounded float division (53 >= 2 * 24 + 2)
    out[k] = STAT ? (float)__dsqrt_rn((double)res) : res;
}

// z32[k][cell] = fl(fl(x - mean) / sd)
__global__ __launch_bounds__(256) void k32_zscore(const double *__restrict__ X, int64_t n, const int32_t *__restrict__ genes,
                                                  const float *__restrict__ mean32, const float *__restrict__ sd32,
                                                  float *__restrict__ z32)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t g = genes[blockIdx.y];
    const float x = (float)X[(int64_t)(g >> 4) * n * SC_TILE + i * SC_TILE + (g & 15)];
    z32[(int64_t)blockIdx.y * n + i] = (float)__ddiv_rn((double)__fsub_rn(x, mean32[blockIdx.y]), (double)sd32[blockIdx.y]);
}

// lag32[k][i] = scipy's float32 csr_matvec row i of W32 @ z32[k]
__global__ __launch_bounds__(256) void k32_lag(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                               const double *__restrict__ w, const float *__restrict__ z32, int64_t n,
                                               float *__restrict__ lag32)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *z = z32 + (int64_t)blockIdx.y * n;
    float s = 0.f;
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) s = __fadd_rn(s, __fmul_rn((float)w[e], z[indices[e]]));
    lag32[(int64_t)blockIdx.y * n + i] = s;
}

// leaf sums of p = fl(zx * lag_y) over cells start ..   thread = (leaf, pair)
__global__ __launch_bounds__(256) void k32_pair_leafsum(const float *__restrict__ z32, const float *__restrict__ lag32,
                                                        int64_t n, const int2 *__restrict__ pair_slots,
                                                        const uint2 *__restrict__ leaves, uint32_t nleaves,
                                                        float *__restrict__ leafsum)
{
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= nleaves) return;
    const int2 sl = pair_slots[blockIdx.y];
    if (sl.x < 0) return;
    const float *zx = z32 + (int64_t)sl.x * n, *ly = lag32 + (int64_t)sl.y * n;
    const uint2 lf = leaves[leaf];
    leafsum[(int64_t)blockIdx.y * nleaves + leaf] =
        pw_block<float>(lf.y, [&](uint32_t k) { return __fmul_rn(zx[lf.x + k], ly[lf.x + k]); });
}

__global__ void k32_pair_combine(int64_t n, const int2 *__restrict__ pair_slots, int64_t n_pairs,
                                 const float *__restrict__ leafsum, uint32_t nleaves, float *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pairs) return;
    out[q] = pair_slots[q].x < 0 ? 0.f : np_sum_from_leaves((uint32_t)n, leafsum + q * nleaves);
}

extern "C" int sc_lee_observed_f32(sc_ctx *c, const int32_t *pair_x, const int32_t *pair_y, int64_t n_pairs,
                                   float *L32_out, float *mean32_out, float *sd32_out)
{
    SC_REQUIRE(c && pair_x && pair_y && L32_out, SC_ERR_INVALID, "sc_lee_observed_f32: null pointer");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->e_n > 0 && c->g_n == c->e_n, SC_ERR_STATE, "sc_lee_observed_f32: expression / graph missing");
    SC_REQUIRE(c->e_dtype == SC_F32, SC_ERR_STATE, "sc_lee_observed_f32: the loaded matrix is not float32");
    const int64_t n = c->e_n, G = c->e_genes;
    SC_REQUIRE(n < ((int64_t)1 << 31), SC_ERR_INVALID, "sc_lee_observed_f32: too many cells");
    if (n_pairs == 0) return SC_OK;
    // distinct genes of the pair list (first-seen order)
    std::vector<int32_t> genes, slot((size_t)G, -1);
    for (int64_t q = 0; q < n_pairs; ++q) {
        SC_REQUIRE(pair_x[q] >= 0 && pair_x[q] < G && pair_y[q] >= 0 && pair_y[q] < G, SC_ERR_INVALID,
                   "sc_lee_observed_f32: pair %lld references a gene outside the loaded set", (long long)q);
        for (int32_t g : {pair_x[q], pair_y[q]})
            if (slot[(size_t)g] < 0) { slot[(size_t)g] = (int32_t)genes.size(); genes.push_back(g); }
    }
    const int K = (int)genes.size();
    const uint32_t m = (uint32_t)n;
    const uint32_t max_leaves = m / 64 + 66;  // leaves hold >= 64 elements each, except in a ragged last chunk
    // layout of one scratch buffer: [genes K i32][mean K f32][sd K f32][nleaves u32 + pad][leaves][z32 K n][lag32 K n]
    SC_TRY(c->lee_a.ensure(sizeof(int32_t) * (size_t)K * 3 + 16 + sizeof(uint2) * (size_t)max_leaves +
                           sizeof(float) * 2 * (size_t)K * (size_t)n, &c->mem));
    int32_t *d_genes = c->lee_a.as<int32_t>();
    float *d_mean = reinterpret_cast<float *>(d_genes + K), *d_sd = d_mean + K;
    uint32_t *d_nl = reinterpret_cast<uint32_t *>(d_sd + K + (K & 1));
    uint2 *d_leaves = reinterpret_cast<uint2 *>(d_nl + 4);
    float *d_z = reinterpret_cast<float *>(d_leaves + max_leaves), *d_lag = d_z + (int64_t)K * n;
    SC_HIP(hipMemcpyAsync(d_genes, genes.data(), sizeof(int32_t) * (size_t)K, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k32_leaves, dim3(1), dim3(1), 0, c->stream, m, d_leaves, max_leaves, d_nl);
    uint32_t nleaves = 0;
    SC_HIP(hipMemcpyAsync(&nleaves, d_nl, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    SC_REQUIRE(nleaves <= max_leaves, SC_ERR_STATE, "sc_lee_observed_f32: leaf table overflow");
    const size_t ls_elems = (size_t)std::max<int64_t>(K, n_pairs) * (size_t)(nleaves ? nleaves : 1);
    SC_TRY(c->lee_b.ensure(sizeof(float) * ls_elems, &c->mem));
    float *d_ls = c->lee_b.as<float>();
    const dim3 lgrid((unsigned)ceil_div64(nleaves ? nleaves : 1, 256), (unsigned)K), cgrid((unsigned)ceil_div64(n, 256), (unsigned)K);
    const double *X = c->X.as<double>();
    if (nleaves) hipLaunchKernelGGL(k32_gene_leafsum<0>, lgrid, dim3(256), 0, c->stream, X, n, d_genes, d_mean, d_leaves, nleaves, d_ls);
    hipLaunchKernelGGL(k32_gene_combine<0>, dim3((unsigned)ceil_div64(K, 64)), dim3(64), 0, c->stream, n, K, d_ls, nleaves, d_mean);
    if (nleaves) hipLaunchKernelGGL(k32_gene_leafsum<1>, lgrid, dim3(256), 0, c->stream, X, n, d_genes, d_mean, d_leaves, nleaves, d_ls);
    hipLaunchKernelGGL(k32_gene_combine<1>, dim3((unsigned)ceil_div64(K, 64)), dim3(64), 0, c->stream, n, K, d_ls, nleaves, d_sd);
    hipLaunchKernelGGL(k32_zscore, cgrid, dim3(256), 0, c->stream, X, n, d_genes, d_mean, d_sd, d_z);
    hipLaunchKernelGGL(k32_lag, cgrid, dim3(256), 0, c->stream, c->g_indptr.as<int64_t>(), c->g_indices.as<int32_t>(),
                       c->g_data.as<double>(), d_z, n, d_lag);
    SC_HIP(hipGetLastError());
    // pairs: (slot of x, slot of y), or (-1, -1) when a gene has zero float32 variance (AC:1129: x_std == 0)
    std::vector<float> sd((size_t)K), mean((size_t)K);
    SC_HIP(hipMemcpyAsync(sd.data(), d_sd, sizeof(float) * (size_t)K, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(mean.data(), d_mean, sizeof(float) * (size_t)K, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    std::vector<int2> ps((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; ++q) {
        const int sx = slot[(size_t)pair_x[q]], sy = slot[(size_t)pair_y[q]];
        ps[(size_t)q] = (sd[(size_t)sx] == 0.f || sd[(size_t)sy] == 0.f) ? make_int2(-1, -1) : make_int2(sx, sy);
        if (mean32_out) { mean32_out[2 * q] = mean[(size_t)sx]; mean32_out[2 * q + 1] = mean[(size_t)sy]; }
        if (sd32_out) { sd32_out[2 * q] = sd[(size_t)sx]; sd32_out[2 * q + 1] = sd[(size_t)sy]; }
    }
    SC_TRY(c->lee_rowmap.ensure(sizeof(int2) * (size_t)n_pairs + sizeof(float) * (size_t)n_pairs, &c->mem));
    int2 *d_ps = c->lee_rowmap.as<int2>();
    float *d_out = reinterpret_cast<float *>(d_ps + n_pairs);
    SC_HIP(hipMemcpyAsync(d_ps, ps.data(), sizeof(int2) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    for (int64_t q0 = 0; nleaves && q0 < n_pairs; q0 += 32768) {   // gridDim.y limit
        const int64_t qn = std::min<int64_t>(32768, n_pairs - q0);
        hipLaunchKernelGGL(k32_pair_leafsum, dim3((unsigned)ceil_div64(nleaves, 256), (unsigned)qn), dim3(256), 0, c->stream,
                           d_z, d_lag, n, d_ps + q0, d_leaves, nleaves, d_ls + q0 * nleaves);
    }
    hipLaunchKernelGGL(k32_pair_combine, dim3((unsigned)ceil_div64(n_pairs, 64)), dim3(64), 0, c->stream, n, d_ps, n_pairs,
                       d_ls, nleaves, d_out);
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(L32_out, d_out, sizeof(float) * (size_t)n_pairs, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// EXTENSION: all pairs of an x-gene list and a y-gene list under SHARED permutations.
//
// The reference draws a fresh block of permutations for every pair, which makes a 100 x 100 screen cost 2 x 10^6
// permutations of the cells (sc_lee_seeded: 36 ms per pair, generator-bound).  When ONE block of P permutations is
// shared by all pairs (each pair's null is still "y shuffled against x"; the nulls of different pairs are correlated),
// the permutation statistics of the whole grid are P dense contractions over the cells,
//     L_p[x][y] = sum_j U[j][x] * Zy[perm_p[j]][y],
// i.e. the permutation x gene batch becomes a true GEMM with a row-gathered B operand: fp64 matrix cores
// (v_mfma_f64_16x16x4_f64), A = 16 x-genes of 4 cells (coalesced 512 bytes), B = 16 y-genes of the 4 permuted cells
// (four gathered 128-byte rows).  A wavefront keeps the accumulators of up to 8 x-tiles, so every gathered row is
// used for 128 x-genes.
// ------------------------------------------------------------------------------------------------

// out tiles [t][cell][16] = column genes[16 t + s] of the source tiles (0 beyond n_genes)
__global__ __launch_bounds__(256) void k_repack_tiles(const double *__restrict__ T, int64_t n, const int32_t *__restrict__ genes,
                                                      int n_genes, double *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cell = t >> 4;
    const int s = (int)(t & 15);
    if (cell >= n) return;
    const int k = blockIdx.y * 16 + s;
    double v = 0.0;
    if (k < n_genes) { const int32_t g = genes[k]; v = T[(int64_t)(g >> 4) * n * SC_TILE + cell * SC_TILE + (g & 15)]; }
    out[(int64_t)blockIdx.y * n * SC_TILE + cell * SC_TILE + s] = v;
}

#define LEE_SH_CELLS 16384   // cells per workgroup (4 wavefronts x 4096)
#define LEE_SH_XT 8          // x tiles per wavefront pass

// partial[p][yt][xt][block][256]: sums over the block's cells of U_xt[cell][x] * Zy_yt[perm_p[cell]][y]
__global__ __launch_bounds__(256) void k_lee_shared_mfma(const double *__restrict__ Ux, const double *__restrict__ Zy,
                                                         int64_t n, const int32_t *__restrict__ perm, int64_t pstride,
                                                         int xt0, int xt_n, int x_tiles, int y_tiles,
                                                         double *__restrict__ partial)
{
    __shared__ double red[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int yt = blockIdx.y, p = blockIdx.z;
    const double *B = Zy + (int64_t)yt * n * SC_TILE;
    const int32_t *prow = perm + (int64_t)p * pstride;
    const int64_t c0 = (int64_t)blockIdx.x * LEE_SH_CELLS + (int64_t)wave * (LEE_SH_CELLS / 4);
    int64_t c1 = c0 + LEE_SH_CELLS / 4;
    if (c1 > n) c1 = n;
    v4f64 acc[LEE_SH_XT];
#pragma unroll
    for (int k = 0; k < LEE_SH_XT; ++k) acc[k] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int64_t c = c0; c < c1; c += 4) {
        const int64_t cell = c + (lane >> 4);
        const bool live = cell < c1;
        const double b = live ? B[(int64_t)prow[cell] * SC_TILE + (lane & 15)] : 0.0;
#pragma unroll
        for (int k = 0; k < LEE_SH_XT; ++k) {
            if (k < xt_n) {
                const double a = live ? Ux[(int64_t)(xt0 + k) * n * SC_TILE + cell * SC_TILE + (lane & 15)] : 0.0;
                acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LEE_SH_XT; ++k) {   // (unrolled: a runtime index into acc[] would put it in scratch)
        if (k < xt_n) {                       // uniform for the whole workgroup
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 4; ++v) red[wave][((lane >> 4) + 4 * v) * 16 + (lane & 15)] = acc[k][v];
            __syncthreads();
            const int t = threadIdx.x;
            partial[((((int64_t)p * y_tiles + yt) * x_tiles + xt0 + k) * gridDim.x + blockIdx.x) * 256 + t] =
                (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        }
    }
}

// L_p[x][y] = sum over blocks (ascending); count[x][y] += |L_p| >= |obs[x][y]|; optional copy of L_p
__global__ __launch_bounds__(256) void k_lee_shared_count(const double *__restrict__ partial, int blocks, int x_tiles, int y_tiles,
                                                          int n_x, int n_y, int n_perm_chunk, const double *__restrict__ obs,
                                                          unsigned long long *__restrict__ count, double *__restrict__ lperm_out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)n_x * n_y;
    if (t >= per * n_perm_chunk) return;
    const int p = (int)(t / per);
    const int x = (int)((t % per) / n_y), y = (int)(t % n_y);
    const double *src = partial + ((((int64_t)p * y_tiles + (y >> 4)) * x_tiles + (x >> 4)) * blocks) * 256 + (x & 15) * 16 + (y & 15);
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += src[(int64_t)b * 256];
    if (fabs(s) >= fabs(obs[(int64_t)x * n_y + y])) atomicAdd(&count[(int64_t)x * n_y + y], 1ull);
    if (lperm_out) lperm_out[t] = s;
}

extern "C" int sc_lee_shared(sc_ctx *c, uint64_t *state6, const int32_t *genes_x, int32_t n_x, const int32_t *genes_y,
                             int32_t n_y, int64_t n_perm, double *L_out, int64_t *count_abs_ge_out, double *L_perm_out)
{
    SC_REQUIRE(c && genes_x && genes_y && L_out && count_abs_ge_out, SC_ERR_INVALID, "sc_lee_shared: null pointer");
    SC_REQUIRE(n_x >= 1 && n_y >= 1 && n_perm >= 0, SC_ERR_INVALID, "sc_lee_shared: bad sizes");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->e_n > 0 && c->g_n == c->e_n, SC_ERR_STATE, "sc_lee_shared: expression / graph missing");
    // state6 == NULL: the shared block is rows [0, n_perm) of the RESIDENT table (e.g. sc_perm_generate_counter's)
    SC_REQUIRE(n_perm == 0 || state6 || (c->p_count >= n_perm && c->p_n == c->e_n), SC_ERR_STATE,
               "sc_lee_shared: no generator state and no resident table of %lld rows", (long long)n_perm);
    const int64_t n = c->e_n, T = c->e_tiles, G = c->e_genes;
    for (int k = 0; k < n_x; ++k) SC_REQUIRE(genes_x[k] >= 0 && genes_x[k] < G, SC_ERR_INVALID, "sc_lee_shared: x gene out of range");
    for (int k = 0; k < n_y; ++k) SC_REQUIRE(genes_y[k] >= 0 && genes_y[k] < G, SC_ERR_INVALID, "sc_lee_shared: y gene out of range");
    const size_t tile_bytes = (size_t)n * SC_TILE * sizeof(double);
    const int XT = (n_x + 15) / 16, YT = (n_y + 15) / 16;
    const int64_t per = (int64_t)n_x * n_y;
    // z-scores (zero-variance genes -> 0: their L and every L_perm are 0, count = n_perm, p = 1), lag, U
    SC_TRY(sc_expr_zscores(c));
    SC_TRY(c->Lag.ensure((size_t)T * tile_bytes, &c->mem));
    SC_TRY(sc_lag_tiles(c, c->g_indptr, c->g_indices, c->g_data, c->Z.as<double>(), c->Lag.as<double>()));
    SC_TRY(sc_graph_ensure_transpose(c));
    SC_TRY(c->lee_U.ensure((size_t)T * tile_bytes, &c->mem));
    SC_TRY(sc_lag_tiles(c, c->gt_indptr, c->gt_indices, c->gt_data, c->Z.as<double>(), c->lee_U.as<double>()));
    // compact tile sets: Zx (observed), Ux (permutations) over the x genes; LagY (observed), Zy (permutations) over the y genes
    SC_TRY(c->lee_Uc.ensure((size_t)(2 * XT) * tile_bytes, &c->mem));
    SC_TRY(c->lee_Zc.ensure((size_t)(2 * YT) * tile_bytes, &c->mem));
    SC_TRY(c->lee_a.ensure(sizeof(int32_t) * (size_t)(n_x + n_y), &c->mem));
    int32_t *d_gx = c->lee_a.as<int32_t>(), *d_gy = d_gx + n_x;
    SC_HIP(hipMemcpyAsync(d_gx, genes_x, sizeof(int32_t) * (size_t)n_x, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(d_gy, genes_y, sizeof(int32_t) * (size_t)n_y, hipMemcpyHostToDevice, c->stream));
    double *Zx = c->lee_Uc.as<double>(), *Ux = Zx + (size_t)XT * n * SC_TILE;
    double *LagY = c->lee_Zc.as<double>(), *Zy = LagY + (size_t)YT * n * SC_TILE;
    const unsigned gcell = (unsigned)ceil_div64(n * 16, 256);
    hipLaunchKernelGGL(k_repack_tiles, dim3(gcell, (unsigned)XT), dim3(256), 0, c->stream, c->Z.as<double>(), n, d_gx, (int)n_x, Zx);
    hipLaunchKernelGGL(k_repack_tiles, dim3(gcell, (unsigned)XT), dim3(256), 0, c->stream, c->lee_U.as<double>(), n, d_gx, (int)n_x, Ux);
    hipLaunchKernelGGL(k_repack_tiles, dim3(gcell, (unsigned)YT), dim3(256), 0, c->stream, c->Lag.as<double>(), n, d_gy, (int)n_y, LagY);
    hipLaunchKernelGGL(k_repack_tiles, dim3(gcell, (unsigned)YT), dim3(256), 0, c->stream, c->Z.as<double>(), n, d_gy, (int)n_y, Zy);
    // observed grid: the identity "permutation" through the same contraction kernel shape (k_lee_observed_mfma)
    std::vector<int2> tps;
    for (int a = 0; a < XT; ++a) for (int b = 0; b < YT; ++b) tps.push_back(make_int2(a, b));
    const int oblocks = (int)ceil_div64(n, LEE_OBS_CELLS);
    SC_TRY(c->lee_pairs.ensure(sizeof(int2) * tps.size(), &c->mem));
    SC_HIP(hipMemcpyAsync(c->lee_pairs.p, tps.data(), sizeof(int2) * tps.size(), hipMemcpyHostToDevice, c->stream));
    const int sblocks = (int)ceil_div64(n, LEE_SH_CELLS);
    const int64_t chunk_max = n_perm < PERM_CHUNK ? (n_perm > 0 ? n_perm : 1) : PERM_CHUNK;
    const size_t part_obs = tps.size() * (size_t)oblocks * 256, part_perm = (size_t)chunk_max * YT * XT * sblocks * 256;
    SC_TRY(c->lee_part.ensure(sizeof(double) * std::max(part_obs, part_perm), &c->mem));
    SC_TRY(c->lee_obs.ensure(sizeof(double) * (size_t)per, &c->mem));
    SC_TRY(c->lee_cnt.ensure(sizeof(unsigned long long) * (size_t)per * 2, &c->mem));
    unsigned long long *d_cnt_backup = c->lee_cnt.as<unsigned long long>() + per;
    SC_HIP(hipMemsetAsync(c->lee_cnt.p, 0, sizeof(unsigned long long) * (size_t)per, c->stream));
    hipLaunchKernelGGL(k_lee_observed_mfma, dim3((unsigned)oblocks, (unsigned)tps.size()), dim3(256), 0, c->stream, Zx, LagY, n,
                       c->lee_pairs.as<int2>(), c->lee_part.as<double>());
    // pick: obs[x][y] from tile pair (x >> 4) * YT + (y >> 4): reuse k_lee_shared_count's addressing with one "permutation"
    // whose partial layout is [yt][xt] -- the observed kernel wrote [tile pair = xt * YT + yt]; a tiny dedicated pick instead:
    {
        std::vector<int32_t> tp((size_t)per), px((size_t)per), py((size_t)per);
        for (int x = 0; x < n_x; ++x)
            for (int y = 0; y < n_y; ++y) {
                tp[(size_t)x * n_y + y] = (x >> 4) * YT + (y >> 4);
                px[(size_t)x * n_y + y] = x;
                py[(size_t)x * n_y + y] = y;
            }
        SC_TRY(c->lee_rowmap.ensure(sizeof(int32_t) * 3 * (size_t)per, &c->mem));
        int32_t *d_tp = c->lee_rowmap.as<int32_t>();
        SC_HIP(hipMemcpyAsync(d_tp, tp.data(), sizeof(int32_t) * (size_t)per, hipMemcpyHostToDevice, c->stream));
        SC_HIP(hipMemcpyAsync(d_tp + per, px.data(), sizeof(int32_t) * (size_t)per, hipMemcpyHostToDevice, c->stream));
        SC_HIP(hipMemcpyAsync(d_tp + 2 * per, py.data(), sizeof(int32_t) * (size_t)per, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_lee_observed_pick, dim3((unsigned)ceil_div64(per, 256)), dim3(256), 0, c->stream,
                           c->lee_part.as<double>(), oblocks, d_tp, d_tp + per, d_tp + 2 * per, per, c->lee_obs.as<double>());
        SC_HIP(hipGetLastError());
        SC_HIP(hipStreamSynchronize(c->stream));   // host vectors
    }
    if (n_perm > 0) {
        if (L_perm_out) SC_TRY(c->lee_lperm.ensure(sizeof(double) * (size_t)per * (size_t)n_perm, &c->mem));
        auto score = [&](int64_t p0, int64_t p1) -> int {
            const int cnt = (int)(p1 - p0);
            for (int xt0 = 0; xt0 < XT; xt0 += LEE_SH_XT) {
                const int xt_n = XT - xt0 < LEE_SH_XT ? XT - xt0 : LEE_SH_XT;
                KernelTimerScope ts(c, SC_K_LEE_PERM);
                hipLaunchKernelGGL(k_lee_shared_mfma, dim3((unsigned)sblocks, (unsigned)YT, (unsigned)cnt), dim3(256), 0, c->stream,
                                   Ux, Zy, n, c->perm.as<int32_t>() + p0 * c->p_stride, c->p_stride, xt0, xt_n, XT, YT,
                                   c->lee_part.as<double>());
            }
            hipLaunchKernelGGL(k_lee_shared_count, dim3((unsigned)ceil_div64(per * cnt, 256)), dim3(256), 0, c->stream,
                               c->lee_part.as<double>(), sblocks, XT, YT, (int)n_x, (int)n_y, cnt, c->lee_obs.as<double>(),
                               c->lee_cnt.as<unsigned long long>(),
                               L_perm_out ? c->lee_lperm.as<double>() + p0 * per : (double *)nullptr);
            SC_HIP(hipGetLastError());
            return SC_OK;
        };
        if (!state6) {   // the resident table, chunk by chunk
            SC_TRY(sc_perm_forward_ensure(c));
            for (int64_t p0 = 0; p0 < n_perm; p0 += PERM_CHUNK) SC_TRY(score(p0, p0 + PERM_CHUNK < n_perm ? p0 + PERM_CHUNK : n_perm));
        }
        SC_HIP(hipMemcpyAsync(d_cnt_backup, c->lee_cnt.p, sizeof(unsigned long long) * (size_t)per, hipMemcpyDeviceToDevice, c->stream));
        const int ahead = c->pg_ahead;
        c->pg_ahead = 2;
        int rc = state6 ? sc_perm_pipeline(c, state6, n, n_perm, 0, nullptr, score) : SC_OK;
        if (rc == SC_PERMGEN_RETRY) {
            SC_HIP(hipMemcpyAsync(c->lee_cnt.p, d_cnt_backup, sizeof(unsigned long long) * (size_t)per, hipMemcpyDeviceToDevice, c->stream));
            const int mode = c->pg_mode;
            c->pg_mode = 1;
            rc = sc_perm_pipeline(c, state6, n, n_perm, 0, nullptr, score);
            c->pg_mode = mode;
        }
        c->pg_ahead = ahead;
        SC_TRY(rc);
        if (L_perm_out)
            SC_HIP(hipMemcpyAsync(L_perm_out, c->lee_lperm.p, sizeof(double) * (size_t)per * (size_t)n_perm, hipMemcpyDeviceToHost, c->stream));
    }
    std::vector<unsigned long long> cnt((size_t)per);
    SC_HIP(hipMemcpyAsync(L_out, c->lee_obs.p, sizeof(double) * (size_t)per, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(cnt.data(), c->lee_cnt.p, sizeof(unsigned long long) * (size_t)per, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    for (int64_t q = 0; q < per; ++q) count_abs_ge_out[q] = (int64_t)cnt[(size_t)q];
    return SC_OK;
}
