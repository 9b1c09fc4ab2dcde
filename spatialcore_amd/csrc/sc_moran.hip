// Expression tiles, spatial lag, global Moran's I with permutations, Lee's L.  gfx950 only.
//
// Device layout (DESIGN.md "Data layout"): genes are grouped in tiles of SC_TILE = 16; a tile is
// [cell][16] fp64, i.e. one 128-byte row per cell.  A permutation step `lag[perm[i]]` then gathers
// one full cache line that serves 16 genes at once, and the contiguous operand z[i] is a coalesced
// 128-byte row.  The permutation table is [perm][cell] int32 with a row stride that is a multiple
// of 32 elements so that rows can be read as int4.
#include <math.h>
#include <stdlib.h>

#include <functional>
#include <thread>
#include <vector>

#include "sc_ctx.h"
#include "sc_pairwise.h"

// ------------------------------------------------------------------------------------------------
// expression upload
// ------------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void k_scatter_csr(const int64_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const T *__restrict__ data,
                                                      const int32_t *__restrict__ colmap,
                                                      double *__restrict__ X, int64_t rows,
                                                      int64_t n, int64_t n_vars, int64_t row0)
{
    // one wavefront per matrix row; lanes stride over the row's stored entries.
    // X points at the chunk's first row inside tile 0; n is the full cell count (tile stride).
    int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    int64_t e0 = indptr[row] - row0, e1 = indptr[row + 1] - row0;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
        int32_t c = indices[e];
        if ((uint32_t)c >= (uint64_t)n_vars) continue;
        int32_t slot = colmap[c];
        if (slot >= 0)
            X[(int64_t)(slot >> 4) * n * SC_TILE + row * SC_TILE + (slot & 15)] = (double)data[e];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather_dense(const T *__restrict__ data, int64_t ld,
                                                       const int32_t *__restrict__ gene_cols,
                                                       int64_t n_genes, double *__restrict__ X,
                                                       int64_t n, int64_t row_lo, int64_t rows)
{
    // thread = (row, slot) of one tile (blockIdx.y); padded slots are written as 0
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t r = t >> 4;
    int s = (int)(t & 15);
    if (r >= rows) return;
    int64_t g = (int64_t)blockIdx.y * SC_TILE + s;
    double v = 0.0;
    if (g < n_genes) v = (double)data[r * ld + gene_cols[g]];
    X[(int64_t)blockIdx.y * n * SC_TILE + (row_lo + r) * SC_TILE + s] = v;
}

static int expr_alloc(sc_ctx *c, int64_t n, int64_t n_genes)
{
    SC_REQUIRE(n >= 1 && n <= 0x7fffffffLL, SC_ERR_INVALID, "n_cells=%lld out of range", (long long)n);
    SC_REQUIRE(n_genes >= 1 && n_genes <= (1 << 24), SC_ERR_INVALID, "n_genes=%lld out of range",
               (long long)n_genes);
    int64_t tiles = ceil_div64(n_genes, SC_TILE);
    size_t bytes = (size_t)tiles * n * SC_TILE * sizeof(double);
    SC_TRY(c->X.ensure(bytes, &c->mem));
    size_t gb = (size_t)align_up64(tiles, 8) * SC_TILE * sizeof(double);  // the narrow-source kernels read whole groups
    SC_TRY(c->g_mean.ensure(gb, &c->mem));
    SC_TRY(c->g_var.ensure(gb, &c->mem));
    SC_TRY(c->g_z2.ensure(gb, &c->mem));
    SC_TRY(c->g_scale.ensure(gb, &c->mem));
    SC_TRY(c->g_Inum.ensure(gb, &c->mem));
    SC_TRY(c->g_xsum.ensure(gb, &c->mem));
    SC_TRY(c->g_meanc.ensure(gb, &c->mem));
    SC_TRY(c->g_lat.ensure(gb, &c->mem));
    c->e_n = n;
    c->e_genes = n_genes;
    c->e_tiles = tiles;
    c->narrow_bits = 64;
    c->lat_any = false;
    c->lm_valid = false;
    c->prep_early = false;
    return SC_OK;
}


extern "C" int sc_expr_set_csr(sc_ctx *c, const int64_t *indptr, const int32_t *indices,
                               const void *data, int dtype, int64_t n, int64_t n_vars,
                               const int32_t *gene_cols, int64_t n_genes)
{
    SC_REQUIRE(c && indptr && gene_cols, SC_ERR_INVALID, "sc_expr_set_csr: null pointer");
    SC_REQUIRE(dtype == SC_F32 || dtype == SC_F64, SC_ERR_INVALID, "sc_expr_set_csr: bad dtype %d", dtype);
    SC_REQUIRE(n_vars >= 1 && n_vars <= 0x7fffffffLL, SC_ERR_INVALID, "n_vars out of range");
    SC_HIP(hipSetDevice(c->device));
    c->e_n = 0;
    SC_TRY(expr_alloc(c, n, n_genes));
    c->e_dtype = dtype;
    SC_REQUIRE(indptr[0] == 0, SC_ERR_INVALID, "sc_expr_set_csr: indptr[0] must be 0");
    for (int64_t i = 0; i < n; ++i)
        SC_REQUIRE(indptr[i + 1] >= indptr[i], SC_ERR_INVALID, "sc_expr_set_csr: indptr not monotone at row %lld",
                   (long long)i);
    int64_t nnz = indptr[n];
    SC_REQUIRE(nnz == 0 || (indices && data), SC_ERR_INVALID, "sc_expr_set_csr: null indices/data");
    std::vector<int32_t> colmap((size_t)n_vars, -1);
    for (int64_t g = 0; g < n_genes; ++g) {
        SC_REQUIRE(gene_cols[g] >= 0 && gene_cols[g] < n_vars, SC_ERR_INVALID, "gene column %d out of range",
                   gene_cols[g]);
        SC_REQUIRE(colmap[gene_cols[g]] < 0, SC_ERR_INVALID, "gene column %d listed twice", gene_cols[g]);
        colmap[gene_cols[g]] = (int32_t)g;
    }
    SC_TRY(c->e_colmap.ensure(sizeof(int32_t) * (size_t)n_vars, &c->mem));
    SC_HIP(hipMemcpyAsync(c->e_colmap.p, colmap.data(), sizeof(int32_t) * (size_t)n_vars,
                          hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemsetAsync(c->X.p, 0, (size_t)c->e_tiles * n * SC_TILE * sizeof(double), c->stream));
    SC_TRY(c->e_tmp_indptr.ensure(sizeof(int64_t) * (size_t)(n + 1), &c->mem));
    SC_HIP(hipMemcpyAsync(c->e_tmp_indptr.p, indptr, sizeof(int64_t) * (size_t)(n + 1),
                          hipMemcpyHostToDevice, c->stream));
    // stream the stored entries through the device in row chunks of <= 256 Mi entries
    const int64_t max_chunk = (int64_t)1 << 28;
    size_t esz = dtype == SC_F32 ? 4 : 8;
    int64_t r0 = 0;
    while (r0 < n) {
        int64_t r1 = r0 + 1;
        while (r1 < n && indptr[r1 + 1] - indptr[r0] <= max_chunk) ++r1;
        int64_t e0 = indptr[r0], cnt = indptr[r1] - e0;
        if (cnt > 0) {
            SC_TRY(c->e_tmp_indices.ensure(sizeof(int32_t) * (size_t)cnt, &c->mem));
            SC_TRY(c->e_tmp_data.ensure(esz * (size_t)cnt, &c->mem));
            SC_HIP(hipMemcpyAsync(c->e_tmp_indices.p, indices + e0, sizeof(int32_t) * (size_t)cnt,
                                  hipMemcpyHostToDevice, c->stream));
            SC_HIP(hipMemcpyAsync(c->e_tmp_data.p, (const char *)data + esz * (size_t)e0, esz * (size_t)cnt,
                                  hipMemcpyHostToDevice, c->stream));
            int64_t rows = r1 - r0;
            unsigned grid = (unsigned)ceil_div64(rows * 64, 256);
            const int64_t *ip = c->e_tmp_indptr.as<int64_t>() + r0;
            double *Xr = c->X.as<double>() + r0 * SC_TILE;
            // Xr is offset by r0 rows inside every tile: tile stride stays n*16
            if (dtype == SC_F32)
                hipLaunchKernelGGL(k_scatter_csr<float>, dim3(grid), dim3(256), 0, c->stream, ip,
                                   c->e_tmp_indices.as<int32_t>(), c->e_tmp_data.as<float>(),
                                   c->e_colmap.as<int32_t>(), Xr, rows, n, n_vars, e0);
            else
                hipLaunchKernelGGL(k_scatter_csr<double>, dim3(grid), dim3(256), 0, c->stream, ip,
                                   c->e_tmp_indices.as<int32_t>(), c->e_tmp_data.as<double>(),
                                   c->e_colmap.as<int32_t>(), Xr, rows, n, n_vars, e0);
            SC_HIP(hipGetLastError());
            // the staging buffers are reused by the next chunk
            SC_HIP(hipStreamSynchronize(c->stream));
        }
        r0 = r1;
    }
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

extern "C" int sc_expr_set_dense(sc_ctx *c, const void *data, int dtype, int64_t n, int64_t n_vars,
                                 const int32_t *gene_cols, int64_t n_genes)
{
    SC_REQUIRE(c && data && gene_cols, SC_ERR_INVALID, "sc_expr_set_dense: null pointer");
    SC_REQUIRE(dtype == SC_F32 || dtype == SC_F64, SC_ERR_INVALID, "sc_expr_set_dense: bad dtype %d", dtype);
    SC_REQUIRE(n_vars >= 1, SC_ERR_INVALID, "n_vars out of range");
    SC_HIP(hipSetDevice(c->device));
    c->e_n = 0;
    SC_TRY(expr_alloc(c, n, n_genes));
    c->e_dtype = dtype;
    for (int64_t g = 0; g < n_genes; ++g)
        SC_REQUIRE(gene_cols[g] >= 0 && gene_cols[g] < n_vars, SC_ERR_INVALID, "gene column %d out of range",
                   gene_cols[g]);
    SC_TRY(c->e_colmap.ensure(sizeof(int32_t) * (size_t)n_genes, &c->mem));
    SC_HIP(hipMemcpyAsync(c->e_colmap.p, gene_cols, sizeof(int32_t) * (size_t)n_genes, hipMemcpyHostToDevice,
                          c->stream));
    size_t esz = dtype == SC_F32 ? 4 : 8;
    // row chunks of <= 1 GiB of source data
    int64_t rows_per = ((int64_t)1 << 30) / (int64_t)(esz * (size_t)n_vars);
    if (rows_per < 1) rows_per = 1;
    for (int64_t r0 = 0; r0 < n; r0 += rows_per) {
        int64_t rows = (n - r0 < rows_per) ? n - r0 : rows_per;
        size_t bytes = esz * (size_t)rows * (size_t)n_vars;
        SC_TRY(c->e_tmp_data.ensure(bytes, &c->mem));
        SC_HIP(hipMemcpyAsync(c->e_tmp_data.p, (const char *)data + esz * (size_t)r0 * (size_t)n_vars, bytes,
                              hipMemcpyHostToDevice, c->stream));
        dim3 grid((unsigned)ceil_div64(rows * SC_TILE, 256), (unsigned)c->e_tiles);
        if (dtype == SC_F32)
            hipLaunchKernelGGL(k_gather_dense<float>, grid, dim3(256), 0, c->stream, c->e_tmp_data.as<float>(),
                               n_vars, c->e_colmap.as<int32_t>(), n_genes, c->X.as<double>(), n, r0, rows);
        else
            hipLaunchKernelGGL(k_gather_dense<double>, grid, dim3(256), 0, c->stream,
                               c->e_tmp_data.as<double>(), n_vars, c->e_colmap.as<int32_t>(), n_genes,
                               c->X.as<double>(), n, r0, rows);
        SC_HIP(hipGetLastError());
        SC_HIP(hipStreamSynchronize(c->stream));
    }
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// per-gene column reductions over tiles (deterministic two-stage tree)
// ------------------------------------------------------------------------------------------------

enum { OP_ID = 0, OP_SQ = 1, OP_MUL = 2, OP_NZ = 3, OP_SQC = 4 };

#define RED_ROWS_PER_BLOCK 4096

// partial[tile][chunk][16] = sum over the chunk's rows of op(A[row][slot], B[row][slot])
// (OP_SQC: (A[row][slot] - B[tile * 16 + slot])^2, B = the per-gene means: the squares of Z = X - mean without storing Z)
template <int OP>
__global__ __launch_bounds__(256) void k_colsum_partial(const double *__restrict__ A,
                                                        const double *__restrict__ B,
                                                        double *__restrict__ partial, int64_t n)
{
    __shared__ double sh[256];
    const int64_t tile = blockIdx.y;
    const int slot = threadIdx.x & 15, rg = threadIdx.x >> 4;  // 16 row groups
    const double *a = A + tile * n * SC_TILE;
    const double *b = (OP == OP_MUL) ? B + tile * n * SC_TILE : nullptr;
    const double centre = (OP == OP_SQC) ? B[tile * SC_TILE + slot] : 0.0;
    int64_t r0 = (int64_t)blockIdx.x * RED_ROWS_PER_BLOCK;
    int64_t r1 = r0 + RED_ROWS_PER_BLOCK < n ? r0 + RED_ROWS_PER_BLOCK : n;
    double acc = 0.0;
    for (int64_t r = r0 + rg; r < r1; r += 16) {
        double v = a[r * SC_TILE + slot];
        if (OP == OP_SQC) v = v - centre;
        if (OP == OP_SQ || OP == OP_SQC) v = v * v;
        if (OP == OP_NZ) v = (v != 0.0) ? 1.0 : 0.0;
        if (OP == OP_MUL) v = v * b[r * SC_TILE + slot];
        acc += v;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 16; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 16) partial[(tile * gridDim.x + blockIdx.x) * SC_TILE + threadIdx.x] = sh[threadIdx.x];
}

// out[tile*16+slot] = (sum over chunks, ascending) / div; out_raw (optional) gets the sum itself.
// A true division, as numpy's mean takes it: sum * (1/n) turns a constant column c into c(1 +- eps) for ~15 % of
// the cell counts n, and a zero-variance gene would then look alive.
__global__ void k_colsum_final(const double *__restrict__ partial, double *__restrict__ out,
                               double *__restrict__ out_raw, int chunks, double div)
{
    int tile = blockIdx.x, slot = threadIdx.x;
    double s = 0.0;
    for (int ch = 0; ch < chunks; ++ch) s += partial[((int64_t)tile * chunks + ch) * SC_TILE + slot];
    if (out_raw) out_raw[tile * SC_TILE + slot] = s;
    out[tile * SC_TILE + slot] = s / div;
}

template <int OP>
static int colsum(sc_ctx *c, const double *A, const double *B, double *out, double div, double *out_raw = nullptr)
{
    int64_t n = c->e_n;
    int chunks = (int)ceil_div64(n, RED_ROWS_PER_BLOCK);
    SC_TRY(c->red_tmp.ensure(sizeof(double) * (size_t)c->e_tiles * chunks * SC_TILE, &c->mem));
    hipLaunchKernelGGL(k_colsum_partial<OP>, dim3(chunks, (unsigned)c->e_tiles), dim3(256), 0, c->stream, A, B,
                       c->red_tmp.as<double>(), n);
    hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)c->e_tiles), dim3(SC_TILE), 0, c->stream,
                       c->red_tmp.as<double>(), out, out_raw, chunks, div);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// Z = X - mean   (mode 0, scanpy's z)      |  Z = Z / sd  (mode 1, in place; Lee's z-score AC:1142)
__global__ __launch_bounds__(256) void k_center(const double *__restrict__ X, const double *__restrict__ mean,
                                                double *__restrict__ Z, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * SC_TILE) return;
    int64_t tile = blockIdx.y;
    int slot = (int)(t & 15);
    Z[tile * n * SC_TILE + t] = X[tile * n * SC_TILE + t] - mean[tile * SC_TILE + slot];
}

__global__ __launch_bounds__(256) void k_div_sd(double *__restrict__ Z, const double *__restrict__ var,
                                                int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * SC_TILE) return;
    int64_t tile = blockIdx.y;
    int slot = (int)(t & 15);
    double v = var[tile * SC_TILE + slot];
    double sd = sqrt(v);
    // zero-variance genes are standardised to 0 (AC:1357-1359)
    Z[tile * n * SC_TILE + t] = (v > 0.0) ? Z[tile * n * SC_TILE + t] / sd : 0.0;
}

// mean (+ raw column sums), z2 = sum (X - mean)^2, var = z2 / n
static int expr_moments(sc_ctx *c)
{
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "no expression loaded (call sc_expr_set_* first)");
    const int64_t n = c->e_n;
    SC_TRY(colsum<OP_ID>(c, c->X.as<double>(), nullptr, c->g_mean.as<double>(), (double)n, c->g_xsum.as<double>()));
    SC_TRY(colsum<OP_SQC>(c, c->X.as<double>(), c->g_mean.as<double>(), c->g_var.as<double>(), (double)n, c->g_z2.as<double>()));
    return SC_OK;
}

// Z = X - centre (per gene)
static int expr_write_z(sc_ctx *c, const double *centre)
{
    c->lm_valid = false;  // Z is about to be rewritten
    const int64_t n = c->e_n;
    SC_TRY(c->Z.ensure((size_t)c->e_tiles * n * SC_TILE * sizeof(double), &c->mem));
    dim3 grid((unsigned)ceil_div64(n * SC_TILE, 256), (unsigned)c->e_tiles);
    hipLaunchKernelGGL(k_center, grid, dim3(256), 0, c->stream, c->X.as<double>(), centre, c->Z.as<double>(), n);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// mean, Z = X - mean, z2 = sum Z^2, var = z2 / n
static int expr_center(sc_ctx *c)
{
    SC_TRY(expr_moments(c));
    return expr_write_z(c, c->g_mean.as<double>());
}

extern "C" int sc_expr_stats(sc_ctx *c, double *mean_out, double *var_out)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_HIP(hipSetDevice(c->device));
    SC_TRY(expr_center(c));
    if (mean_out)
        SC_HIP(hipMemcpyAsync(mean_out, c->g_mean.p, sizeof(double) * (size_t)c->e_genes, hipMemcpyDeviceToHost,
                              c->stream));
    if (var_out)
        SC_HIP(hipMemcpyAsync(var_out, c->g_var.p, sizeof(double) * (size_t)c->e_genes, hipMemcpyDeviceToHost,
                              c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// A3: spatial lag  Lag[i][g] = sum_e w[e] * Z[col[e]][g]   (row-sequential, mul and add rounded
// separately, like scanpy's `(i_data * z[i_indices]).sum()` and scipy's csr_matvec)
// ------------------------------------------------------------------------------------------------

// unit[gene] != 0 (optional): the gene's rows are summed with weight 1 instead of w[e] -- the unweighted neighbour sums S
// of an integer-lattice gene (Z holds its raw counts then), exact integers in fp64.
//
// Processing order (r03): thread groups walk the cells in the graph's spatially sorted order (`order`: the bin-sorted
// order of the points the graph was built from; identity for a graph of unknown geometry) and each XCD -- blockIdx.x % 8
// under round-robin placement, speed only -- takes one contiguous eighth of that order, so the neighbour rows a
// workgroup gathers were fetched by its neighbours a moment ago and sit in THAT XCD's L2.  In input order (r02) every
// neighbour row came from the Infinity Cache or HBM again: 12.5 ms and 8-16 x the compulsory fetch traffic per launch at
// bench size.  The sums are per row, in edge order: the results do not depend on the processing order.
__global__ __launch_bounds__(256) void k_lag(const int64_t *__restrict__ indptr,
                                             const int32_t *__restrict__ indices,
                                             const double *__restrict__ w, const double *__restrict__ Z,
                                             double *__restrict__ Lag, int64_t n, const double *__restrict__ unit,
                                             const int32_t *__restrict__ order)
{
    // 8 threads per cell, each owning 2 of the tile's 16 genes (one 16-byte slice of the row)
    const int64_t per_xcd = (int64_t)(gridDim.x >> 3);                 // gridDim.x is a multiple of 8
    const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t t = blk * blockDim.x + threadIdx.x;
    const int64_t pos = t >> 3;
    int q = (int)(t & 7);
    if (pos >= n) return;
    const int64_t i = order ? order[pos] : pos;
    const double2 *Zt = reinterpret_cast<const double2 *>(Z + (int64_t)blockIdx.y * n * SC_TILE);
    double2 *Lt = reinterpret_cast<double2 *>(Lag + (int64_t)blockIdx.y * n * SC_TILE);
    const bool ux = unit && unit[(int64_t)blockIdx.y * SC_TILE + 2 * q] != 0.0;
    const bool uy = unit && unit[(int64_t)blockIdx.y * SC_TILE + 2 * q + 1] != 0.0;
    int64_t e0 = indptr[i], e1 = indptr[i + 1];
    double sx = 0.0, sy = 0.0;
    for (int64_t e = e0; e < e1; ++e) {
        int32_t j = indices[e];
        double ww = w[e];
        double2 z = Zt[(int64_t)j * 8 + q];
        sx = __dadd_rn(sx, __dmul_rn(ux ? 1.0 : ww, z.x));
        sy = __dadd_rn(sy, __dmul_rn(uy ? 1.0 : ww, z.y));
    }
    Lt[i * 8 + q] = make_double2(sx, sy);
}

// The same lag from the float32 narrow rows (r04): a float32-source batch has its raw values as 32 genes per 128-byte row
// (k_pack_narrow<32>: the lane's 16 bytes = genes {16 t + 2 q, + 1} of the group's two tiles), so a neighbour costs one
// 16-byte piece per lane for FOUR genes instead of one per tile for two -- half the gathered bytes (what bounds k_lag is
// the rows through the CUs' vector memory path, section 4.2 of DESIGN.md).  z = (double)x - centre is rebuilt in
// registers, the very value the Z tile holds; products and sums rounded separately, edges in ascending order: Lag is
// k_lag's bit for bit.
__global__ __launch_bounds__(256) void k_lag_f32rows(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                     const double *__restrict__ w, const uint4 *__restrict__ narrow,
                                                     const double *__restrict__ centre, double *__restrict__ Lag, int64_t n,
                                                     int tiles16, const double *__restrict__ unit,
                                                     const int32_t *__restrict__ order)
{
    const int64_t per_xcd = (int64_t)(gridDim.x >> 3);                 // gridDim.x is a multiple of 8
    const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t t = blk * blockDim.x + threadIdx.x;
    const int64_t pos = t >> 3;
    const int q = (int)(t & 7);
    if (pos >= n) return;
    const int64_t i = order ? order[pos] : pos;
    const int grp = blockIdx.y;                                        // 32 genes = tiles 2 grp, 2 grp + 1
    const int tiles_left = tiles16 - 2 * grp;
    const uint4 *Xg = narrow + (int64_t)grp * n * 8;
    double cen[4];
    bool un[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t g = (int64_t)(2 * grp + ((k >> 1) < tiles_left ? (k >> 1) : 0)) * SC_TILE + 2 * q + (k & 1);
        cen[k] = centre[g];
        un[k] = unit && unit[g] != 0.0;
    }
    const int64_t e0 = indptr[i], e1 = indptr[i + 1];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t e = e0; e < e1; ++e) {
        const uint4 v = Xg[(int64_t)indices[e] * 8 + q];
        const double ww = w[e];
        const double z[4] = {(double)__uint_as_float(v.x) - cen[0], (double)__uint_as_float(v.y) - cen[1],
                             (double)__uint_as_float(v.z) - cen[2], (double)__uint_as_float(v.w) - cen[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = __dadd_rn(s[k], __dmul_rn(un[k] ? 1.0 : ww, z[k]));
    }
    reinterpret_cast<double2 *>(Lag + (int64_t)(2 * grp) * n * SC_TILE)[i * 8 + q] = make_double2(s[0], s[1]);
    if (tiles_left > 1) reinterpret_cast<double2 *>(Lag + (int64_t)(2 * grp + 1) * n * SC_TILE)[i * 8 + q] = make_double2(s[2], s[3]);
}

static int launch_lag(sc_ctx *c, const DBuf &indptr, const DBuf &indices, const DBuf &data, const double *Z,
                      double *out, const double *unit = nullptr)
{
    int64_t n = c->e_n;
    const int32_t *order = (c->g_order_captured && c->g_n == n && c->g_order.p) ? c->g_order.as<int32_t>() : nullptr;
    KernelTimerScope ts(c, SC_K_LAG);
    hipLaunchKernelGGL(k_lag, dim3((unsigned)align_up64(ceil_div64(n * 8, 256), 8), (unsigned)c->e_tiles), dim3(256), 0,
                       c->stream, indptr.as<int64_t>(), indices.as<int32_t>(), data.as<double>(), Z, out, n, unit, order);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// helpers for the other translation units (sc_lee.hip): population-sd z-scores of the loaded genes in c->Z
// (zero-variance genes -> 0, AC:1357-1359; c->g_var holds the variances), and the lag kernel on any CSR
int sc_expr_zscores(sc_ctx *c)
{
    SC_TRY(expr_center(c));
    hipLaunchKernelGGL(k_div_sd, dim3((unsigned)ceil_div64(c->e_n * SC_TILE, 256), (unsigned)c->e_tiles), dim3(256), 0,
                       c->stream, c->Z.as<double>(), c->g_var.as<double>(), c->e_n);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

int sc_lag_tiles(sc_ctx *c, const DBuf &indptr, const DBuf &indices, const DBuf &data, const double *Z, double *out)
{
    return launch_lag(c, indptr, indices, data, Z, out);
}

// ------------------------------------------------------------------------------------------------
// A5: the permutation kernel (the metric's dominant kernel)
//
//   partial[s][p][g] = sum_{i in split s} Z[i][g] * Lag[perm_p[i]][g]        (one 16-gene tile)
//
// Workgroup = 256 threads = 4 wavefronts; wavefront w of block (s, pt) owns permutations
// pt*32 + w*8 + (lane >> 3) and the gene pair (lane & 7): every lane keeps its two fp64
// accumulators in registers across the whole cell range, so there is no cross-lane reduction at
// all.  Per cell a wavefront issues ONE 16-byte-per-lane gather that pulls 8 full 128-byte Lag
// rows (8 permutations x 16 genes) and one broadcast read of the 128-byte Z row.
// ------------------------------------------------------------------------------------------------

#define MP_PERMS_PER_BLOCK 32

__global__ __launch_bounds__(256) void k_moran_perm(const double *__restrict__ Zt,
                                                    const double *__restrict__ Lt,
                                                    const int32_t *__restrict__ perm,
                                                    double *__restrict__ partial, int64_t n,
                                                    int64_t pstride, int n_perm, int64_t cells_per_split)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane >> 3, q = lane & 7;
    const int pbase = blockIdx.y * MP_PERMS_PER_BLOCK + wave * 8;
    if (pbase >= n_perm) return;  // whole wavefront idle (no barriers in this kernel)
    const int p = pbase + r;
    const int pc = p < n_perm ? p : n_perm - 1;
    const int64_t c0 = (int64_t)blockIdx.x * cells_per_split;
    int64_t c1 = c0 + cells_per_split;
    if (c1 > n) c1 = n;
    const int32_t *prow = perm + (int64_t)pc * pstride;
    const double2 *Z2 = reinterpret_cast<const double2 *>(Zt) + q;
    const double2 *L2 = reinterpret_cast<const double2 *>(Lt) + q;

    double a0x = 0.0, a0y = 0.0, a1x = 0.0, a1y = 0.0;
    int64_t i = c0;  // c0 is a multiple of 8 (cells_per_split is)
    for (; i + 8 <= c1; i += 8) {
        const int4 ia = *reinterpret_cast<const int4 *>(prow + i);
        const int4 ib = *reinterpret_cast<const int4 *>(prow + i + 4);
        const double2 l0 = L2[(int64_t)ia.x * 8];
        const double2 l1 = L2[(int64_t)ia.y * 8];
        const double2 l2 = L2[(int64_t)ia.z * 8];
        const double2 l3 = L2[(int64_t)ia.w * 8];
        const double2 l4 = L2[(int64_t)ib.x * 8];
        const double2 l5 = L2[(int64_t)ib.y * 8];
        const double2 l6 = L2[(int64_t)ib.z * 8];
        const double2 l7 = L2[(int64_t)ib.w * 8];
        const double2 z0 = Z2[(i + 0) * 8];
        const double2 z1 = Z2[(i + 1) * 8];
        const double2 z2 = Z2[(i + 2) * 8];
        const double2 z3 = Z2[(i + 3) * 8];
        const double2 z4 = Z2[(i + 4) * 8];
        const double2 z5 = Z2[(i + 5) * 8];
        const double2 z6 = Z2[(i + 6) * 8];
        const double2 z7 = Z2[(i + 7) * 8];
        a0x = fma(z0.x, l0.x, a0x); a0y = fma(z0.y, l0.y, a0y);
        a1x = fma(z1.x, l1.x, a1x); a1y = fma(z1.y, l1.y, a1y);
        a0x = fma(z2.x, l2.x, a0x); a0y = fma(z2.y, l2.y, a0y);
        a1x = fma(z3.x, l3.x, a1x); a1y = fma(z3.y, l3.y, a1y);
        a0x = fma(z4.x, l4.x, a0x); a0y = fma(z4.y, l4.y, a0y);
        a1x = fma(z5.x, l5.x, a1x); a1y = fma(z5.y, l5.y, a1y);
        a0x = fma(z6.x, l6.x, a0x); a0y = fma(z6.y, l6.y, a0y);
        a1x = fma(z7.x, l7.x, a1x); a1y = fma(z7.y, l7.y, a1y);
    }
    for (; i < c1; ++i) {
        const double2 l = L2[(int64_t)prow[i] * 8];
        const double2 z = Z2[i * 8];
        a0x = fma(z.x, l.x, a0x);
        a0y = fma(z.y, l.y, a0y);
    }
    if (p < n_perm) {
        double2 *out = reinterpret_cast<double2 *>(partial) +
                       ((int64_t)blockIdx.x * n_perm + p) * 8 + q;
        *out = make_double2(a0x + a1x, a0y + a1y);
    }
}

// sims[p0 + p][g0 + slot] = seff * (sum_s partial[s][p][slot] - corr)   (ascending s); raw = the sum itself
__global__ __launch_bounds__(256) void k_moran_finalize(const double *__restrict__ partial,
                                                        const double *__restrict__ seff, const double *__restrict__ corr,
                                                        double *__restrict__ sims, double *__restrict__ raw, int n_perm, int splits,
                                                        int64_t n_genes, int64_t g0, int64_t p0)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int p = t >> 4, slot = t & 15;
    if (p >= n_perm || g0 + slot >= n_genes) return;
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += partial[((int64_t)k * n_perm + p) * SC_TILE + slot];
    raw[(p0 + p) * n_genes + g0 + slot] = s;
    sims[(p0 + p) * n_genes + g0 + slot] = seff[g0 + slot] * (s - corr[g0 + slot]);
}

// Per gene: how a sum over the cells becomes the statistic, and what the permutation count compares.
//   ordinary gene: sum = sum_j lag_j z_j;  seff = n / s0 / z2, corr = 0, I = seff * sum, count: sims >= I
//   lattice gene (integer counts, every graph weight = w): sum = T = sum_j S_j x_j, an exact integer (S = unweighted
//     neighbour sums); sum_j lag_j z_j = w (T - mean * sum_j S_j) in exact arithmetic, so seff = n / s0 / z2 * w,
//     corr = mean * sum_j S_j, I = seff * (T_obs - corr), and the count compares the integers: T_p >= T_obs
__global__ void k_moran_scale(const double *__restrict__ z2, const double *__restrict__ inum, const double *__restrict__ slag,
                              const double *__restrict__ mean, const double *__restrict__ lat, double *__restrict__ seff,
                              double *__restrict__ corr, double *__restrict__ thr, double *__restrict__ I, double n_over_s0,
                              double w, int64_t total)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const bool l = lat[g] != 0.0;
    const double sc = l ? (n_over_s0 / z2[g]) * w : n_over_s0 / z2[g];
    const double co = l ? mean[g] * slag[g] : 0.0;
    const double v = sc * (inum[g] - co);
    seff[g] = sc;
    corr[g] = co;
    I[g] = v;
    thr[g] = l ? inum[g] : v;
}

// per gene: count(sims >= I) -- for lattice genes count(T_p >= T_obs), on the exact integer sums --, sum sims,
// sum sims^2 over permutations (block per gene, fixed tree).  A zero-variance gene (I = NaN) counts nothing.
__global__ __launch_bounds__(256) void k_moran_count(const double *__restrict__ sims, const double *__restrict__ raw,
                                                     const double *__restrict__ thr, const double *__restrict__ lat,
                                                     const double *__restrict__ z2, int n_perm,
                                                     int64_t n_genes, long long *__restrict__ count,
                                                     double *__restrict__ ssum, double *__restrict__ ssq)
{
    __shared__ double sh_a[256], sh_b[256];
    __shared__ int sh_c[256];
    int64_t g = blockIdx.x;
    const double t = thr[g];
    const bool l = lat[g] != 0.0, alive = z2[g] > 0.0;
    double a = 0.0, b = 0.0;
    int cnt = 0;
    for (int p = threadIdx.x; p < n_perm; p += 256) {
        double v = sims[(int64_t)p * n_genes + g];
        const double cv = l ? raw[(int64_t)p * n_genes + g] : v;
        cnt += (alive && cv >= t) ? 1 : 0;
        a += v;
        b += v * v;
    }
    sh_a[threadIdx.x] = a;
    sh_b[threadIdx.x] = b;
    sh_c[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh_a[threadIdx.x] += sh_a[threadIdx.x + s];
            sh_b[threadIdx.x] += sh_b[threadIdx.x + s];
            sh_c[threadIdx.x] += sh_c[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        count[g] = sh_c[0];
        ssum[g] = sh_a[0];
        ssq[g] = sh_b[0];
    }
}

// ------------------------------------------------------------------------------------------------
// A5: the permutation statistic, summed over the TARGET cell,
//
//   sims[p][g] = scale_g * sum_j lag_g[j] * z_g[inv_p[j]],   inv_p = perm_p^-1,   z = (double)x - centre_g
//
// The fp64 lag rows are the streamed, coalesced operand; the INVERSE permutation supplies the gather index; the
// gathered operand is a 128-byte row of the raw values in the narrowest type that holds EVERY loaded gene exactly:
//   * 128 uint8 genes  (BITS = 8)   every value an integer count in [0, 255] and every gene a lattice gene (below),
//   * 64 uint16 genes  (BITS = 16)  every value an integer count in [0, 65535],
//   * 32 float32 genes (BITS = 32)  every value a float32 (AnnData's usual dtype),
//   * 16 fp64 genes    (BITS = 64)  anything else: the rows of the Z tiles themselves.
// z is rebuilt in registers: (double)x - centre is the very subtraction k_center performs for the Z tiles, and every
// width adds the same products in the same order (cells ascending inside a split, splits ascending in the
// finalisation) -- A GENE'S STATISTICS DO NOT DEPEND ON THE WIDTH, i.e. not on the genes it is loaded with.
//
// Lattice genes (integer counts on a graph whose weights all equal w, e.g. kNN: w = 1 / k): lag_j = w S_j - mean with
// the integer neighbour sum S_j, so the part of the statistic that depends on the permutation is T_p = sum_j S_j x[inv_p[j]],
// an integer, and a permutation can TIE the observed value exactly (41 of 250 Poisson genes of the bench matrix have
// such a permutation among 1000).  Rounded lag / z operands decide those ties by summation-order noise -- the
// reference's numba loop as much as any kernel here.  They are therefore scored on the lattice itself: centre = 0,
// streamed operand = S (k_lag with unit weights).  Every product and partial sum is an integer below 2^53 (checked on
// the host: max degree * largest count * sum of counts), so the fp64 arithmetic is EXACT in any order, and the
// permutation count is #{T_p >= T_obs} on integers (k_moran_scale / k_moran_count turn T into I and sims).
// The uint8 kernel has no registers for 16 centres next to its 16 accumulators; it is used when every loaded gene is a
// lattice gene (centre 0); uint8-sized counts on a graph with unequal weights take the uint16 kernel.
//
// Row layout (all widths): the 8 lanes q that share a row own genes {16 t + 2 q, 16 t + 2 q + 1 : t < TG} of the
// group's TG 16-gene tiles (TG = 8 / 4 / 2 / 1), stored as the lane's 16 bytes [t][e]: a lane's lag operands are then TG
// 16-byte LDS reads that are contiguous across q (no bank conflicts), one per lag tile.
//
// What bounds the uint8 form (r02, 1M cells, 128 genes x 128 permutations, 3.56 ms on 248 CUs; diagnostic builds
// -DSC_DIAG=n give wrong sums on purpose): half the LDS reads of lag 3.40 ms, no int -> fp64 conversions 3.50, neither
// byte extraction nor conversion 3.44 -- neither LDS nor VALU issue; the gathered rows alone are 4.6 TB/s, with the
// lag rows and indices ~5.7 TB/s memory-side: the random 128-byte gather at what the fabric sustains.
// ------------------------------------------------------------------------------------------------

// Value class of every gene in one pass over the tiles: flags[g] bit 0 = some value is not an integer in [0, 255],
// bit 1 = ... not an integer in [0, 65535], bit 2 = ... not a float32 (NaN included); xmax[g] = largest integer count
// (0xffffffff as soon as a value is no integer in [0, 2^32))
__global__ __launch_bounds__(256) void k_gene_stats(const double *__restrict__ X, int64_t n, uint32_t *__restrict__ flags,
                                                    uint32_t *__restrict__ xmax)
{
    __shared__ uint32_t sh_f[256], sh_m[256];
    const int64_t tile = blockIdx.y;
    const int slot = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const double *a = X + tile * n * SC_TILE;
    const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS_PER_BLOCK;
    const int64_t r1 = r0 + RED_ROWS_PER_BLOCK < n ? r0 + RED_ROWS_PER_BLOCK : n;
    uint32_t f = 0u, m = 0u;
    for (int64_t r = r0 + rg; r < r1; r += 16) {
        const double v = a[r * SC_TILE + slot];
        const bool isint = v >= 0.0 && v <= 4294967295.0 && (double)(uint32_t)v == v;
        const uint32_t u = isint ? (uint32_t)v : 0xffffffffu;
        f |= (u <= 255u ? 0u : 1u) | (u <= 65535u ? 0u : 2u) | ((double)(float)v == v ? 0u : 4u);
        m = u > m ? u : m;
    }
    sh_f[threadIdx.x] = f;
    sh_m[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s >= 16; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh_f[threadIdx.x] |= sh_f[threadIdx.x + s];
            sh_m[threadIdx.x] = sh_m[threadIdx.x] > sh_m[threadIdx.x + s] ? sh_m[threadIdx.x] : sh_m[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < 16) {
        if (sh_f[threadIdx.x]) atomicOr(&flags[tile * SC_TILE + threadIdx.x], sh_f[threadIdx.x]);
        atomicMax(&xmax[tile * SC_TILE + threadIdx.x], sh_m[threadIdx.x]);
    }
}

// centre[g] = lat[g] ? 0 : mean[g]
__global__ void k_moran_centres(const double *__restrict__ mean, const double *__restrict__ lat, double *__restrict__ centre,
                                int64_t total)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < total) centre[g] = lat[g] != 0.0 ? 0.0 : mean[g];
}

// Narrow[group][cell][q][t][e] = X[TG * group + t][cell][2 q + e] as uint8 / uint16 / float (every value fits: k_gene_stats)
template <int BITS>
__global__ __launch_bounds__(256) void k_pack_narrow(const double *__restrict__ X, uint4 *__restrict__ out, int64_t n,
                                                     int64_t tiles16)
{
    constexpr int TG = BITS == 8 ? 8 : BITS == 16 ? 4 : 2;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (cell, q)
    if (t >= n * 8) return;
    const int64_t cell = t >> 3;
    const int q = (int)(t & 7);
    uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
        const int64_t t16 = TG * (int64_t)blockIdx.y + tt;
        if (t16 >= tiles16) continue;
        const double2 v = reinterpret_cast<const double2 *>(X + t16 * n * SC_TILE + cell * SC_TILE)[q];
        if (BITS == 8) {
            const uint32_t a = (uint32_t)(v.x >= 0.0 && v.x <= 255.0 ? v.x : 0.0), b = (uint32_t)(v.y >= 0.0 && v.y <= 255.0 ? v.y : 0.0);
            o[tt >> 1] |= (a | (b << 8)) << (16 * (tt & 1));
        } else if (BITS == 16) {
            const uint32_t a = (uint32_t)(v.x >= 0.0 && v.x <= 65535.0 ? v.x : 0.0), b = (uint32_t)(v.y >= 0.0 && v.y <= 65535.0 ? v.y : 0.0);
            o[tt] = a | (b << 16);
        } else {
            const float a = (float)v.x, b = (float)v.y;
            o[2 * tt] = __float_as_uint(a);
            o[2 * tt + 1] = __float_as_uint(b);
        }
    }
    out[((int64_t)blockIdx.y * n + cell) * 8 + q] = make_uint4(o[0], o[1], o[2], o[3]);
}

#define INV_BLOCKS_PER_ROW 64

// inv[row][perm[row][i]] = i.  Blocks of one row share blockIdx % 8 (one XCD under round-robin placement,
// speed only) so that the 4n-byte inverse row is assembled in one L2.
__global__ __launch_bounds__(256) void k_invert_perm(const int32_t *__restrict__ perm, int32_t *__restrict__ inv,
                                                     int64_t n, int64_t stride, int rows)
{
    const int id = blockIdx.x;
    const int rest = id >> 3;
    const int row = (rest / INV_BLOCKS_PER_ROW) * 8 + (id & 7);
    const int part = rest % INV_BLOCKS_PER_ROW;
    if (row >= rows) return;
    const int64_t per = (n + INV_BLOCKS_PER_ROW - 1) / INV_BLOCKS_PER_ROW;
    const int64_t i0 = (int64_t)part * per, i1 = i0 + per < n ? i0 + per : n;
    const int32_t *src = perm + (int64_t)row * stride;
    int32_t *dst = inv + (int64_t)row * stride;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) dst[src[i]] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_check_inverse(const int32_t *__restrict__ perm,
                                                       const int32_t *__restrict__ inv, int64_t n, int64_t stride,
                                                       int64_t rows, int *__restrict__ flag)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = rows * n;
    int bad = 0;
    for (; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n, i = t - r * n;
        if (inv[r * stride + perm[r * stride + i]] != (int32_t)i) bad = 1;
    }
    if (bad) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------
// The scoring kernel: PERSISTENT, one 16-wavefront workgroup per compute unit, wavefronts as independent workers.
//
// Work = tasks (gene group, cell split, group of 8 permutations); wavefront w of the grid takes tasks w, w + W,
// w + 2 W, ... (groups slowest: the chip works on one 128-MB narrow table at a time, which stays in the
// Infinity Cache).  Lane = (permutation r of the 8, q): it keeps 2 TG fp64 accumulators over the task's cell range --
// no cross-lane reduction, no workgroup barrier.  A task is software-pipelined in blocks of CB cells:
//   (1) the lag rows of block b + 1 are loaded cooperatively (one coalesced 16-byte load per lane and piece: every
//       lag row is fetched ONCE per wavefront, not once per permutation),
//   (2) all CB random row gathers of block b + 1 are issued (indices were loaded two blocks ahead),
//   (3) the indices of block b + 3 are loaded,
//   (4) [wait: only the loads of (2), (3) stay outstanding -- vmcnt retires in order]
//   (5) the lag rows of block b + 1 are parked in the wavefront's private LDS slice,
//   (6) block b is multiplied: gathered rows from registers, lag rows from LDS (a broadcast to the 8 permutations).
// CB kilobytes of random rows are in flight per wavefront while it computes.
//
// Why persistent and 1024 threads: with > 64 VGPRs per lane a second 16-wavefront workgroup cannot fit on a compute
// unit, so a grid of W workgroups occupies exactly W of the 256 compute units and leaves the others EMPTY for the
// permutation generator that runs beside it (sc_moran_seeded) -- without CU-masked or prioritised streams, which were
// found to corrupt concurrently running kernels on this platform (see moran_seeded_streams).
// Sums: per gene, cells ascending inside a split, splits ascending in k_moran_finalize_groups -- the same order for
// both source widths, so their results are bit-identical (and independent of the grid size).
// ------------------------------------------------------------------------------------------------

#define SCORE_WAVES 16
#define SCORE_PRIVATE_WAVES 8   // k_moran_score (r04): 512-thread workgroups -- under the 128-VGPR cap of a 1024-thread workgroup
                                // its private lag staging spilled 6-10 registers (28-44 B of scratch per lane); its wavefronts
                                // are independent workers, so two workgroups of 8 per compute unit are the same 16 workers

// L16 (BITS == 8 only, r04): the lag operand arrives as the 16-bit neighbour sums k_lag_u8 leaves -- Lag16[group][cell][t][q]
// = S of genes (16 t + 2 q, + 1) as two uint16 in one word, 256 bytes per cell and 128-gene group instead of 1 KB of fp64
// -- and becomes the same fp64 values when it is parked in LDS (integers: exact): a quarter of the streamed bytes.
__device__ __forceinline__ double2 lag16_to_double2(uint32_t v) { return make_double2((double)(v & 0xffffu), (double)(v >> 16)); }

template <int BITS, int CB, bool BIG, bool L16 = false>
__global__ __launch_bounds__(SCORE_PRIVATE_WAVES * 64) void k_moran_score(
    const uint4 *__restrict__ narrow, const double *__restrict__ Lag, int64_t tile_elems, int tiles16,
    const double *__restrict__ mean, const int32_t *__restrict__ inv, double *__restrict__ partial, int64_t n,
    int64_t pstride, int n_perm, int64_t cells_per_split, int n_splits, int n_groups)
{
    static_assert((BITS == 8 || BITS == 16 || BITS == 32 || BITS == 64) && (CB == 4 || CB == 8), "source width / block size");
    static_assert(!L16 || BITS == 8, "16-bit lag rows belong to the uint8 source");
    constexpr int TG = BITS == 8 ? 8 : BITS == 16 ? 4 : BITS == 32 ? 2 : 1;   // 16-gene lag tiles per gene group
    // BITS == 8 (128 genes per row, 16 accumulators per lane): no room for 16 centres in registers -- only launched when
    // every gene of the batch is a lattice gene (centre 0, integer operands: exact).  BITS == 64 gathers rows of the Z
    // tiles, which are centred already.
    constexpr bool CENTER = BITS == 16 || BITS == 32;
    constexpr int ROW = TG * 8;                  // 16-byte pieces of a group's lag row (one cell)
    static_assert((CB * ROW) % 64 == 0, "a block's lag rows are loaded by whole wavefront instructions");
    constexpr int NI = CB / 4;                   // index vectors per block
    constexpr int NL = CB * ROW / 64;            // lag pieces per lane and block
    constexpr int CSTEP = 64 / ROW;              // cells covered by one cooperative lag load
    __shared__ double2 lds_lag[SCORE_PRIVATE_WAVES][2][CB * ROW];   // [wavefront][buffer][cell][ROW]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane >> 3, q = lane & 7;
    const int lcell0 = lane / ROW, ltile = (lane % ROW) >> 3;
    double2 *lw = &lds_lag[wave][0][0];
    const int pgroups = (n_perm + 7) >> 3;
    // (Measured and dropped in r02: giving each XCD -- blockIdx % 8 -- its own subset of the cell splits, so that a
    //  split's lag rows are fetched by one L2 only: same launch time, same PMC traffic.)
    const int64_t n_tasks = (int64_t)n_groups * n_splits * pgroups;
    const int64_t worker = (int64_t)blockIdx.x * SCORE_PRIVATE_WAVES + wave, workers = (int64_t)gridDim.x * SCORE_PRIVATE_WAVES;

    for (int64_t task = worker; task < n_tasks; task += workers) {
        const int pg = (int)(task % pgroups);
        const int64_t rest = task / pgroups;
        const int split = (int)(rest % n_splits), grp = (int)(rest / n_splits);
        const int p = pg * 8 + r;
        const int pc = p < n_perm ? p : n_perm - 1;
        const int64_t c0 = (int64_t)split * cells_per_split;
        int64_t c1 = c0 + cells_per_split;
        if (c1 > n) c1 = n;
        const int32_t *irow = inv + (int64_t)pc * pstride;
        // row i of the group's table at byte offset 128 i (+ 16 q for this lane): a 32-bit offset from a
        // wavefront-uniform base (n < 2^25 cells, else the host picks the BIG form), i.e. no 64-bit address arithmetic per gather
        const char *Xg = reinterpret_cast<const char *>(narrow + (int64_t)grp * n * 8);
        const uint32_t qoff = (uint32_t)q * 16u;
        auto row_of = [&](int32_t i) {
            if constexpr (BIG) return *reinterpret_cast<const uint4 *>(Xg + ((uint64_t)(uint32_t)i * 128u + qoff));
            else return *reinterpret_cast<const uint4 *>(Xg + ((uint32_t)i * 128u + qoff));
        };
        const int tiles_left = tiles16 - TG * grp;                                 // lag tiles this group really has
        const double *lag_g = Lag + (int64_t)TG * grp * tile_elems;
        // (a padded last group re-reads its first tile for the missing ones; those sums are never used)
        const double2 *lsrc = reinterpret_cast<const double2 *>(lag_g + (int64_t)(ltile < tiles_left ? ltile : 0) * tile_elems) + (lane & 7);
        const uint32_t *lag16_g = reinterpret_cast<const uint32_t *>(Lag) + (int64_t)grp * n * ROW;   // L16: [cell][ROW] words
        const uint32_t *lsrc16 = lag16_g + (lane % ROW);
        double m[CENTER ? TG : 1][2], acc[TG][2];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            if constexpr (CENTER) {
                const double *mt = mean + (int64_t)(TG * grp + (t < tiles_left ? t : 0)) * SC_TILE + 2 * q;
                m[t][0] = mt[0]; m[t][1] = mt[1];
            }
            acc[t][0] = acc[t][1] = 0.0;
        }
        const int64_t nblk = (c1 - c0) / CB;

        int4 ida[NI], idb[NI];
        double2 lg0, lg1, lg2, lg3;   // lag pieces on their way to LDS (as many as NL)
        uint32_t lh0 = 0, lh1 = 0, lh2 = 0, lh3 = 0;   // ... as 16-bit pairs (L16)
        uint4 xa[CB], xb[CB];

        auto load_idx = [&](int4 (&id)[NI], int64_t b) {
            const int64_t bb = b < nblk ? b : nblk - 1;   // past the end: a harmless reload of the last block
#pragma unroll
            for (int k = 0; k < NI; ++k) id[k] = *reinterpret_cast<const int4 *>(irow + c0 + bb * CB + 4 * k);
        };
        auto load_lag = [&](int64_t b) {
            const int64_t bb = b < nblk ? b : nblk - 1;
            const int64_t row0 = c0 + bb * CB + lcell0;
            if constexpr (L16) {
                lh0 = lsrc16[row0 * ROW];
                if constexpr (NL > 1) lh1 = lsrc16[(row0 + CSTEP) * ROW];
                if constexpr (NL > 2) { lh2 = lsrc16[(row0 + 2 * CSTEP) * ROW]; lh3 = lsrc16[(row0 + 3 * CSTEP) * ROW]; }
            } else {
                lg0 = lsrc[row0 * 8];
                if constexpr (NL > 1) lg1 = lsrc[(row0 + CSTEP) * 8];
                if constexpr (NL > 2) { lg2 = lsrc[(row0 + 2 * CSTEP) * 8]; lg3 = lsrc[(row0 + 3 * CSTEP) * 8]; }
            }
        };
        auto gather = [&](uint4 (&x)[CB], const int4 (&id)[NI]) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                x[4 * k + 0] = row_of(id[k].x);
                x[4 * k + 1] = row_of(id[k].y);
                x[4 * k + 2] = row_of(id[k].z);
                x[4 * k + 3] = row_of(id[k].w);
            }
        };
        auto park_lag = [&](int buf) {
            double2 *dst = lw + buf * (CB * ROW) + lane;
            if constexpr (L16) {
                dst[0] = lag16_to_double2(lh0);
                if constexpr (NL > 1) dst[64] = lag16_to_double2(lh1);
                if constexpr (NL > 2) { dst[128] = lag16_to_double2(lh2); dst[192] = lag16_to_double2(lh3); }
            } else {
                dst[0] = lg0;
                if constexpr (NL > 1) dst[64] = lg1;
                if constexpr (NL > 2) { dst[128] = lg2; dst[192] = lg3; }
            }
        };
        auto mul_cell = [&](const uint4 &x, const double2 *lr) {
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int t = 0; t < TG; ++t) {
#if defined(SC_DIAG) && SC_DIAG == 1   /* diagnostic builds (wrong results): which resource bounds the kernel */
                const double2 l = lr[(t & ~1) * 8];
#else
                const double2 l = lr[t * 8];
#endif
                double v0, v1;
                if (BITS == 8) {
                    const uint32_t h = w[t >> 1] >> (16 * (t & 1));
#if defined(SC_DIAG) && SC_DIAG == 2
                    v0 = __hiloint2double(0x3ff00000, (int)(h & 0xffu)); v1 = __hiloint2double(0x3ff00000, (int)((h >> 8) & 0xffu));
#elif defined(SC_DIAG) && SC_DIAG == 3
                    v0 = __hiloint2double(0x3ff00000, (int)h); v1 = __hiloint2double(0x3ff00001, (int)h);
#else
                    v0 = (double)(h & 0xffu); v1 = (double)((h >> 8) & 0xffu);
#endif
                } else if (BITS == 16) { v0 = (double)(w[t] & 0xffffu); v1 = (double)(w[t] >> 16); }
                else if (BITS == 32) { v0 = (double)__uint_as_float(w[2 * t]); v1 = (double)__uint_as_float(w[2 * t + 1]); }
                else { v0 = __hiloint2double((int)w[1], (int)w[0]); v1 = __hiloint2double((int)w[3], (int)w[2]); }
                if constexpr (CENTER) { v0 -= m[t][0]; v1 -= m[t][1]; }
                acc[t][0] = fma(l.x, v0, acc[t][0]);
                acc[t][1] = fma(l.y, v1, acc[t][1]);
            }
        };
        auto multiply = [&](const uint4 (&x)[CB], int buf) {
            const double2 *lr = lw + buf * (CB * ROW) + q;
#pragma unroll
            for (int c = 0; c < CB; ++c) mul_cell(x[c], lr + c * ROW);
        };
        // one stage for block b: `cur` holds its gathered rows; `nxt` receives block b + 1; `id_next` holds the indices
        // of block b + 1 and is refilled with those of block b + 3 (its partner holds b + 2)
        auto stage = [&](const uint4 (&cur)[CB], uint4 (&nxt)[CB], int4 (&id_next)[NI], int64_t b) {
            load_lag(b + 1);
            __builtin_amdgcn_sched_barrier(0);   // issue order matters: vmcnt retires in order
            gather(nxt, id_next);
            __builtin_amdgcn_sched_barrier(0);
            load_idx(id_next, b + 3);
            __builtin_amdgcn_sched_barrier(0);
            park_lag((int)((b + 1) & 1));        // (the compiler's s_waitcnt here leaves the loads of (2), (3) outstanding)
            multiply(cur, (int)(b & 1));
            __builtin_amdgcn_sched_barrier(0);
        };

        if (nblk > 0) {
            load_idx(ida, 0);
            load_lag(0);
            gather(xa, ida);
            load_idx(idb, 1);
            load_idx(ida, 2);
            park_lag(0);
            int64_t b = 0;
            for (; b + 2 <= nblk; b += 2) {
                stage(xa, xb, idb, b);        // idb: block b + 1 -> refilled with b + 3
                stage(xb, xa, ida, b + 1);    // ida: block b + 2 -> refilled with b + 4
            }
            if (b < nblk) stage(xa, xb, idb, b);
        }
        for (int64_t j = c0 + nblk * CB; j < c1; ++j) {   // ragged tail of the split: straight from global memory
            const uint4 x = row_of(irow[j]);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                double2 l;
                if constexpr (L16) l = lag16_to_double2(lag16_g[j * ROW + t * 8 + q]);
                else l = reinterpret_cast<const double2 *>(lag_g + (int64_t)(t < tiles_left ? t : 0) * tile_elems)[j * 8 + q];
                double v0, v1;
                if (BITS == 8) {
                    const uint32_t h = w[t >> 1] >> (16 * (t & 1));
                    v0 = (double)(h & 0xffu); v1 = (double)((h >> 8) & 0xffu);
                } else if (BITS == 16) { v0 = (double)(w[t] & 0xffffu); v1 = (double)(w[t] >> 16); }
                else if (BITS == 32) { v0 = (double)__uint_as_float(w[2 * t]); v1 = (double)__uint_as_float(w[2 * t + 1]); }
                else { v0 = __hiloint2double((int)w[1], (int)w[0]); v1 = __hiloint2double((int)w[3], (int)w[2]); }
                if constexpr (CENTER) { v0 -= m[t][0]; v1 -= m[t][1]; }
                acc[t][0] = fma(l.x, v0, acc[t][0]);
                acc[t][1] = fma(l.y, v1, acc[t][1]);
            }
        }
        if (p < n_perm) {
            // partial[group][split][perm][16 t + 2 q + e]
            double2 *out = reinterpret_cast<double2 *>(partial) + (((int64_t)grp * n_splits + split) * n_perm + p) * ROW + q;
#pragma unroll
            for (int t = 0; t < TG; ++t) out[t * 8] = make_double2(acc[t][0], acc[t][1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same scoring with the lag rows SHARED BY THE WORKGROUP (r03).
//
// What bounds k_moran_score (r03 measurement, bench size, all three narrow widths, alone and inside the pipeline): the
// bytes a compute unit pulls through its vector memory path, ~35 GB/s per CU -- gathered rows AND lag rows alike.  There
// every wavefront fetches the lag rows of its cells for itself: 1 KB of lag per 1 KB of gathered rows for the uint8
// source (8 lag tiles per 128-gene row), i.e. HALF of a compute unit's traffic is the same lag rows arriving 16 times.
// Here a task is (gene group, cell split, 128 permutations): the workgroup's 16 wavefronts score 8 permutations each
// over the SAME cells, and the lag rows of a super-block of SB = 16 cells are loaded ONCE per workgroup -- every thread
// one 16-byte piece, a single load instruction per wavefront and super-block -- into a double-buffered LDS tile
// (2 x 16 cells x ROW pieces: 32 KB for uint8), handed over by ONE workgroup barrier per 16 cells.  Bytes through the
// compute unit per gathered row: 128 + 4 + 128 TG / 16 instead of 128 + 4 + 128 TG.  Same products, same order of
// summation per (gene, permutation): bit-identical to k_moran_score.
// A super-block = 16 / CB stages of the software pipeline of k_moran_score (gather rows(b + 1), indices(b + 3), multiply
// block b); its first stage also issues the lag load of the NEXT super-block, its last stage parks it in the other LDS
// buffer and ends with the barrier.  The pipelined region has NO branch: hipcc's wait-count pass answers a divergent
// path with s_waitcnt vmcnt(0), which would drain the gathers in flight (measured in the ISA of two earlier forms: a
// loader-wavefront `if`, and a skip for wavefronts beyond the chunk's permutations INSIDE the loop).  Hence: narrower
// sources, whose 16 cells have fewer than 1024 pieces, load some pieces twice (same bytes to the same LDS address); and a
// wavefront beyond the chunk's permutations (a chunk shorter than 128) takes a loop of its own, chosen by a
// wavefront-uniform branch OUTSIDE the pipelined loop: it only carries its pieces of the lag rows into LDS and meets the
// same barriers.  What bounds the kernel is the bytes a compute unit pulls in, so a short chunk costs about its share of a
// full one while enough wavefronts are left to keep the memory path busy; the host sends chunks with fewer than
// SCORE_WG_MIN_PERMS permutations in their last task to k_moran_score instead.
// ------------------------------------------------------------------------------------------------
#define SCORE_SB 16   // cells per lag super-block
#ifndef SCORE_WG_MIN_PERMS
#define SCORE_WG_MIN_PERMS 24   // permutations in a chunk's last (partial) task from which the workgroup form is used
#endif

template <int BITS, int CB, bool BIG, bool L16 = false>
__global__ __launch_bounds__(SCORE_WAVES * 64) void k_moran_score_wg(
    const uint4 *__restrict__ narrow, const double *__restrict__ Lag, int64_t tile_elems, int tiles16,
    const double *__restrict__ mean, const int32_t *__restrict__ inv, double *__restrict__ partial, int64_t n,
    int64_t pstride, int n_perm, int64_t cells_per_split, int n_splits, int n_groups)
{
    static_assert((BITS == 4 || BITS == 8 || BITS == 16 || BITS == 32 || BITS == 64) && (CB == 4 || CB == 8), "source width / block size");
    // BITS == 4 (r04): 256 nibble slots per row; TG = the lane's four 16-byte pieces of uint16 operands per cell (8 slots each)
    constexpr bool NIB = BITS == 4;
    constexpr int TG = BITS == 8 ? 8 : BITS == 16 ? 4 : BITS == 32 ? 2 : NIB ? 4 : 1;
    constexpr bool CENTER = BITS == 16 || BITS == 32;
    constexpr int ROW = TG * 8;
    constexpr int NI = CB / 4;
    constexpr int SPS = SCORE_SB / CB;            // pipeline stages per super-block
    constexpr int PIECES = SCORE_SB * ROW;        // 16-byte pieces of one super-block's lag rows (<= 1024)
    static_assert(PIECES <= SCORE_WAVES * 64 && (SCORE_WAVES * 64) % PIECES == 0, "every thread loads one piece");
    __shared__ double2 lds_lag[2][PIECES];        // [buffer][cell][ROW]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane >> 3, q = lane & 7;
    const int pid = (int)threadIdx.x % PIECES;    // this thread's piece of every super-block
    const int lcell = pid / ROW, ltile = (pid % ROW) >> 3;
    const int pchunks = (n_perm + 8 * SCORE_WAVES - 1) / (8 * SCORE_WAVES);
    const int64_t n_tasks = (int64_t)n_groups * n_splits * pchunks;

    for (int64_t task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int pch = (int)(task % pchunks);
        const int64_t rest = task / pchunks;
        const int split = (int)(rest % n_splits), grp = (int)(rest / n_splits);
        const int pbase = (pch * SCORE_WAVES + (BITS != 64 ? __builtin_amdgcn_readfirstlane(wave) : wave)) * 8;
        const bool live = pbase < n_perm;          // wavefront-uniform (and known to the compiler as such)
        const int p = pbase + r;
        const int pc = p < n_perm ? p : n_perm - 1;
        const int64_t c0 = (int64_t)split * cells_per_split;
        int64_t c1 = c0 + cells_per_split;
        if (c1 > n) c1 = n;
        const int32_t *irow = inv + (int64_t)pc * pstride;
        const char *Xg = reinterpret_cast<const char *>(narrow + (int64_t)grp * n * 8);
        const uint32_t qoff = (uint32_t)q * 16u;
        auto row_of = [&](int32_t i) {
            if constexpr (BIG) return *reinterpret_cast<const uint4 *>(Xg + ((uint64_t)(uint32_t)i * 128u + qoff));
            else return *reinterpret_cast<const uint4 *>(Xg + ((uint32_t)i * 128u + qoff));
        };
        const int tiles_left = tiles16 - TG * grp;
        const double *lag_g = Lag + (int64_t)TG * grp * tile_elems;
        // (a padded last group re-reads its first tile for the missing ones; those sums are never used)
        // (NIB: the operand rows are [group][cell][ROW pieces] -- 512 bytes of uint16 per cell --, moved as 16-byte pieces like lag tiles)
        const double2 *lsrc = NIB ? reinterpret_cast<const double2 *>(Lag) + (int64_t)grp * n * ROW + (pid % ROW)
                                  : reinterpret_cast<const double2 *>(lag_g + (int64_t)(ltile < tiles_left ? ltile : 0) * tile_elems) + (pid & 7);
        constexpr int LSTRIDE = NIB ? ROW : 8;      // pieces from one cell's row to the next
        const uint32_t *lag16_g = reinterpret_cast<const uint32_t *>(Lag) + (int64_t)grp * n * ROW;   // L16: [cell][ROW] words
        const uint32_t *lsrc16 = lag16_g + (pid % ROW);
        double m[CENTER ? TG : 1][2], acc[TG][2];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            if constexpr (CENTER) {
                const double *mt = mean + (int64_t)(TG * grp + (t < tiles_left ? t : 0)) * SC_TILE + 2 * q;
                m[t][0] = mt[0]; m[t][1] = mt[1];
            }
            acc[t][0] = acc[t][1] = 0.0;
        }
        uint32_t ai[NIB ? 32 : 1];   // NIB: integer sums of the lane's 32 slots
#pragma unroll
        for (int k2 = 0; k2 < (NIB ? 32 : 1); ++k2) ai[k2] = 0u;
        const int64_t nsb = (c1 - c0) / SCORE_SB;   // whole super-blocks of the split (the rest: tail loop below)
        const int64_t nblk = nsb * SPS;

        double2 lg;
        uint32_t lh = 0;   // L16: the thread's piece as a pair of 16-bit sums
        uint4 xa[CB], xb[CB];
        auto park = [&](int buf) {
            if constexpr (L16) lds_lag[buf][pid] = lag16_to_double2(lh);
            else lds_lag[buf][pid] = lg;
        };

        auto load_idx = [&](int4 (&id)[NI], int64_t b) {
            const int64_t bb = b < nblk ? b : nblk - 1;   // past the end: a harmless reload of the last block
#pragma unroll
            for (int k = 0; k < NI; ++k) id[k] = *reinterpret_cast<const int4 *>(irow + c0 + bb * CB + 4 * k);
        };
        auto load_lag = [&](int64_t sb) {
            const int64_t ss = sb < nsb ? sb : nsb - 1;
            if constexpr (L16) lh = lsrc16[(c0 + ss * SCORE_SB + lcell) * ROW];
            else lg = lsrc[(c0 + ss * SCORE_SB + lcell) * LSTRIDE];
        };
        auto gather = [&](uint4 (&x)[CB], const int4 (&id)[NI]) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                x[4 * k + 0] = row_of(id[k].x);
                x[4 * k + 1] = row_of(id[k].y);
                x[4 * k + 2] = row_of(id[k].z);
                x[4 * k + 3] = row_of(id[k].w);
            }
        };
        auto mul_cell = [&](const uint4 &x, const double2 *lr) {
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
            if constexpr (NIB) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint4 sp = *reinterpret_cast<const uint4 *>(&lr[t * 8]);   // the 8 uint16 operands of slots 64 t + 8 q + e
                    const uint32_t sw[4] = {sp.x, sp.y, sp.z, sp.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        ai[t * 8 + e] += __umul24((w[t] >> (4 * e)) & 15u, (sw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                }
                return;
            }
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const double2 l = lr[t * 8];
                double v0, v1;
                if (BITS == 8) {
                    const uint32_t h = w[t >> 1] >> (16 * (t & 1));
                    v0 = (double)(h & 0xffu); v1 = (double)((h >> 8) & 0xffu);
                } else if (BITS == 16) { v0 = (double)(w[t] & 0xffffu); v1 = (double)(w[t] >> 16); }
                else if (BITS == 32) { v0 = (double)__uint_as_float(w[2 * t]); v1 = (double)__uint_as_float(w[2 * t + 1]); }
                else { v0 = __hiloint2double((int)w[1], (int)w[0]); v1 = __hiloint2double((int)w[3], (int)w[2]); }
                if constexpr (CENTER) { v0 -= m[t][0]; v1 -= m[t][1]; }
                acc[t][0] = fma(l.x, v0, acc[t][0]);
                acc[t][1] = fma(l.y, v1, acc[t][1]);
            }
        };
        // one stage for block b = sb * SPS + k: `cur` holds its gathered rows; `nxt` receives block b + 1; `id_next` holds
        // the indices of block b + 1 and is refilled with those of block b + 1 + SPS, i.e. every index vector is consumed in
        // the NEXT trip of the super-block loop (consumed in the same trip, hipcc sinks the load down to its use -- seen in
        // the IR -- and the gather then waits vmcnt(0) for it)
        auto stage = [&](const uint4 (&cur)[CB], uint4 (&nxt)[CB], int4 (&id_next)[NI], int64_t sb, int k) {
            if (k == 0) load_lag(sb + 1);            // (k is a literal at every call site)
            __builtin_amdgcn_sched_barrier(0);       // issue order matters: vmcnt retires in order
            gather(nxt, id_next);
            __builtin_amdgcn_sched_barrier(0);
            load_idx(id_next, sb * SPS + k + 1 + SPS);
            __builtin_amdgcn_sched_barrier(0);
            const double2 *lr = &lds_lag[sb & 1][k * CB * ROW + q];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                mul_cell(cur[c], lr + c * ROW);
                if constexpr (NIB) __builtin_amdgcn_sched_barrier(0);   // (one cell's operand pieces at a time: hoisted together they spill)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (k == SPS - 1) {
                park((int)((sb + 1) & 1));           // (that buffer was last read in super-block sb - 1, before its barrier)
                __syncthreads();
            }
        };

        if (BITS != 64 && nsb > 0 && !live) {
            // (the fp64-row form has no registers to spare for a second loop: its idle wavefronts score their clamped
            // permutation again, and the host keeps short chunks away from it)
            // a wavefront beyond the chunk's permutations only carries its pieces of the lag rows into LDS: the same
            // barriers as the scoring loop below, no gathers (the branch is wavefront-uniform and OUTSIDE that loop)
            load_lag(0);
            park(0);
            __syncthreads();
            for (int64_t sb = 0; sb < nsb; ++sb) {
                load_lag(sb + 1);
                park((int)((sb + 1) & 1));
                __syncthreads();
            }
        } else if (nsb > 0) {
            int4 id0[NI], id1[NI], id2[NI], id3[NI];   // indices of blocks b + 1 .. b + SPS at the top of a trip (id0 .. id1 when SPS == 2)
            load_lag(0);
            load_idx(id3, 0);
            gather(xa, id3);
            load_idx(id0, 1);
            load_idx(id1, 2);
            if constexpr (SPS == 4) { load_idx(id2, 3); load_idx(id3, 4); }
            park(0);
            __syncthreads();
            for (int64_t sb = 0; sb < nsb; ++sb) {
                stage(xa, xb, id0, sb, 0);
                stage(xb, xa, id1, sb, 1);
                if constexpr (SPS == 4) {
                    stage(xa, xb, id2, sb, 2);
                    stage(xb, xa, id3, sb, 3);
                }
            }
        }
        if (live) {
            for (int64_t j = c0 + nsb * SCORE_SB; j < c1; ++j) {   // ragged tail of the split (< 16 cells): straight from global memory
                const uint4 x = row_of(irow[j]);
                const uint32_t w[4] = {x.x, x.y, x.z, x.w};
                if constexpr (NIB) {
                    const uint4 *srow = reinterpret_cast<const uint4 *>(Lag) + ((int64_t)grp * n + j) * ROW + q;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint4 sp = srow[t * 8];
                        const uint32_t sw[4] = {sp.x, sp.y, sp.z, sp.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            ai[t * 8 + e] += __umul24((w[t] >> (4 * e)) & 15u, (sw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                    }
                    continue;
                }
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    double2 l;
                    if constexpr (L16) l = lag16_to_double2(lag16_g[j * ROW + t * 8 + q]);
                    else l = reinterpret_cast<const double2 *>(lag_g + (int64_t)(t < tiles_left ? t : 0) * tile_elems)[j * 8 + q];
                    double v0, v1;
                    if (BITS == 8) {
                        const uint32_t h = w[t >> 1] >> (16 * (t & 1));
                        v0 = (double)(h & 0xffu); v1 = (double)((h >> 8) & 0xffu);
                    } else if (BITS == 16) { v0 = (double)(w[t] & 0xffffu); v1 = (double)(w[t] >> 16); }
                    else if (BITS == 32) { v0 = (double)__uint_as_float(w[2 * t]); v1 = (double)__uint_as_float(w[2 * t + 1]); }
                    else { v0 = __hiloint2double((int)w[1], (int)w[0]); v1 = __hiloint2double((int)w[3], (int)w[2]); }
                    if constexpr (CENTER) { v0 -= m[t][0]; v1 -= m[t][1]; }
                    acc[t][0] = fma(l.x, v0, acc[t][0]);
                    acc[t][1] = fma(l.y, v1, acc[t][1]);
                }
            }
            if (p < n_perm) {
                if constexpr (NIB) {   // partial[group][split][perm][slot 64 t + 8 q + e]
                    double2 *out = reinterpret_cast<double2 *>(partial + ((((int64_t)grp * n_splits + split) * n_perm + p) * 256 + q * 8));
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2)
                            out[t * 32 + e2] = make_double2((double)ai[t * 8 + 2 * e2], (double)ai[t * 8 + 2 * e2 + 1]);
                } else {
                // partial[group][split][perm][16 t + 2 q + e]
                double2 *out = reinterpret_cast<double2 *>(partial) + (((int64_t)grp * n_splits + split) * n_perm + p) * ROW + q;
#pragma unroll
                for (int t = 0; t < TG; ++t) out[t * 8] = make_double2(acc[t][0], acc[t][1]);
                }
            }
        }
    }
}

// sims[p0 + p][GP grp + slot] = seff * (sum_s partial[grp][s][p][slot] - corr) (ascending s), raw = the sum itself, for
// every gene group of a chunk in one launch (GP = genes per group: 128 / 64 / 32 / 16 by source width); seff, corr:
// k_moran_scale
template <int GP>
__global__ __launch_bounds__(256) void k_moran_finalize_groups(const double *__restrict__ partial,
                                                               const double *__restrict__ seff,
                                                               const double *__restrict__ corr,
                                                               double *__restrict__ sims, double *__restrict__ raw, int n_perm,
                                                               int splits, int64_t n_genes, int64_t p0)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = t / GP, slot = t % GP;
    const int64_t g = GP * (int64_t)blockIdx.y + slot;
    if (p >= n_perm || g >= n_genes) return;
    const double *pt = partial + (int64_t)blockIdx.y * splits * n_perm * GP;
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += pt[((int64_t)k * n_perm + p) * GP + slot];
    raw[(p0 + p) * n_genes + g] = s;
    sims[(p0 + p) * n_genes + g] = seff[g] * (s - corr[g]);
}

// cell range of one scoring task: a function of n ALONE (results must not depend on the chunking of the
// permutations or on the source width): >= 2048 cells, a multiple of 8, at most 512 splits
static int64_t score_cells_per_split(int64_t n)
{
    int64_t cps = align_up64(ceil_div64(n, 512), 8);
    return cps < 2048 ? 2048 : cps;
}

static int pick_splits(int64_t n, int n_perm_tiles, int64_t *cells_per_split)
{
    // aim for >= 2048 workgroups per launch (256 CUs x 8 resident), splits <= 256,
    // and a cell range that is a multiple of 8 and not shorter than 2048 cells
    int64_t want = ceil_div64(2048, n_perm_tiles > 0 ? n_perm_tiles : 1);
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    int64_t cps = align_up64(ceil_div64(n, want), 8);
    if (cps < 2048) cps = 2048;
    *cells_per_split = cps;
    return (int)ceil_div64(n, cps);
}

static int moran_check(sc_ctx *c, int64_t n_perm, const double *I_out)
{
    SC_REQUIRE(c && I_out, SC_ERR_INVALID, "sc_moran: null pointer");
    SC_REQUIRE(n_perm >= 0 && n_perm <= (1 << 24), SC_ERR_INVALID, "sc_moran: n_perm=%lld out of range",
               (long long)n_perm);
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "sc_moran: no expression loaded");
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "sc_moran: no graph set");
    SC_REQUIRE(c->g_n == c->e_n, SC_ERR_INVALID, "sc_moran: graph has %lld rows but expression has %lld cells",
               (long long)c->g_n, (long long)c->e_n);
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// The prelude of an all-lattice uint8 batch in one pass (r03).  For count data on a kNN graph the streamed operand of
// the scoring kernel is S = A x, the unweighted neighbour sums of the raw counts.  k_lag makes them from fp64 tiles: 15
// neighbours x 8 lanes x 16 B per cell and 16-gene tile = 61 GB through the CUs' texture path at bench size, 11.8 ms in
// the scoring's serial prelude.  The uint8 rows the scoring kernel gathers hold 128 genes per 128 bytes: summing THOSE is
// an eighth of the gathered bytes, the sums fit 16 bits (degree <= 257), and the two column sums the observed statistic
// needs (T_obs = sum_i x_i S_i, sum_i S_i: integers, exact in fp64 in any order) fall out of the same pass.  Cells are
// walked in the graph's processing order, one XCD per contiguous eighth.  Lag gets the same fp64 values k_lag writes
// for lattice genes (integers), so everything downstream is bit-identical.
// ------------------------------------------------------------------------------------------------
#define LAG8_CELLS_PER_BLOCK 4096

__global__ __launch_bounds__(256) void k_lag_u8(const long long *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                const uint4 *__restrict__ narrow, const int32_t *__restrict__ order, int64_t n,
                                                int tiles16, int64_t tile_elems, double *__restrict__ Lag,
                                                uint32_t *__restrict__ Lag16 /* non-null: the sums as 16-bit pairs instead */,
                                                double *__restrict__ partial /* [2][tile][chunk][16] */, int chunks)
{
    __shared__ double shT[128], shS[128];      // [tile of the group][slot]
    const int grp = blockIdx.y;
    const int64_t per_xcd = (int64_t)(gridDim.x >> 3);
    const int64_t chunk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // gridDim.x is a multiple of 8
    const int q = threadIdx.x & 7, row = threadIdx.x >> 3;
    if (threadIdx.x < 128) { shT[threadIdx.x] = 0.0; shS[threadIdx.x] = 0.0; }
    __syncthreads();
    const uint4 *Xg = narrow + (int64_t)grp * n * 8;
    const int tiles_left = tiles16 - 8 * grp;
    double accT[16], accS[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) accT[b] = accS[b] = 0.0;
    const int64_t p0 = chunk * LAG8_CELLS_PER_BLOCK;
    if (chunk < chunks) {
        for (int it = 0; it < LAG8_CELLS_PER_BLOCK / 32; ++it) {
            const int64_t pos = p0 + it * 32 + row;
            if (pos >= n) break;
            const int64_t cell = order ? order[pos] : pos;
            const uint4 own = Xg[cell * 8 + q];
            uint32_t lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};   // packed 16-bit sums of the even / odd bytes
            for (long long e = indptr[cell]; e < indptr[cell + 1]; ++e) {
                const uint4 v = Xg[(int64_t)indices[e] * 8 + q];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { lo[k] += w[k] & 0x00ff00ffu; hi[k] += (w[k] >> 8) & 0x00ff00ffu; }
            }
            const uint32_t xo[4] = {own.x, own.y, own.z, own.w};
            // byte b = 2 t + e of the lane's 16 bytes is gene 16 t + 2 q + e of the group: word t >> 1, byte 2 (t & 1) + e
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = t >> 1, sh = 16 * (t & 1);
                const double s0 = (double)((lo[k] >> sh) & 0xffffu), s1 = (double)((hi[k] >> sh) & 0xffffu);
                const double x0 = (double)((xo[k] >> sh) & 0xffu), x1 = (double)((xo[k] >> (sh + 8)) & 0xffu);
                if (Lag16) Lag16[((int64_t)grp * n + cell) * 64 + t * 8 + q] = ((lo[k] >> sh) & 0xffffu) | (((hi[k] >> sh) & 0xffffu) << 16);
                else if (t < tiles_left)
                    reinterpret_cast<double2 *>(Lag + (int64_t)(8 * grp + t) * tile_elems)[cell * 8 + q] = make_double2(s0, s1);
                accS[2 * t] += s0; accS[2 * t + 1] += s1;
                accT[2 * t] = fma(x0, s0, accT[2 * t]); accT[2 * t + 1] = fma(x1, s1, accT[2 * t + 1]);
            }
        }
    }
    // integers below 2^53: the order of these additions does not matter
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            atomicAdd(&shT[t * 16 + 2 * q + e], accT[2 * t + e]);
            atomicAdd(&shS[t * 16 + 2 * q + e], accS[2 * t + e]);
        }
    __syncthreads();
    if (threadIdx.x < 128 && chunk < chunks) {
        const int t = threadIdx.x >> 4, slot = threadIdx.x & 15;
        if (t < tiles_left) {
            const int64_t tile = 8 * grp + t;
            partial[((int64_t)tile * chunks + chunk) * SC_TILE + slot] = shT[threadIdx.x];
            partial[((int64_t)(tiles16 + tile) * chunks + chunk) * SC_TILE + slot] = shS[threadIdx.x];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// r04: the 4-BIT source.  Count data is mostly small counts: a gene whose largest count is below 16 needs a nibble per
// cell, and a gene with counts up to 255 is the sum of two such pseudo-genes, x = lo + 16 hi, whose statistics add exactly
// (integer lattice: T = sum_j S_j x[inv(j)] = T_lo + 16 T_hi).  256 nibble SLOTS fit the 128-byte row the scoring kernel
// gathers per (permutation, cell): slot s < G is the low nibble of gene s, slots G .. G + nw - 1 the high nibbles of the nw
// genes that have one.  Used when that takes FEWER rows than 128 genes per uint8 row (the bench's 500 genes, 75 of them
// with a count >= 16: 575 slots = 3 rows instead of 4 -- a quarter of the gathered bytes, which is what bounds the kernel).
// Row layout: lane q of the 8 that share a row holds, in word t (0 .. 3) nibble e (0 .. 7), slot 64 t + 8 q + e; the
// streamed operand is S (neighbour sums of the FULL gene, for both of its slots) as uint16 in slot order (512 bytes per
// cell and group), so the lane's operands are again 16-byte LDS pieces (t, q) that are contiguous across q; products
// (< 2^20) are added in int32 per task (15 x largest S x cells of a split < 2^31: checked on the host), then as doubles.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_nib(const double *__restrict__ X, uint4 *__restrict__ out, int64_t n, int64_t G,
                                                  int64_t NS, const int32_t *__restrict__ wide)
{
    const int64_t tt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (cell, q)
    if (tt >= n * 8) return;
    const int64_t cell = tt >> 3;
    const int q = (int)(tt & 7), grp = blockIdx.y;
    uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t s0 = (int64_t)grp * 256 + t * 64 + q * 8;
        if (s0 >= NS) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t s = s0 + e;
            if (s >= NS) break;
            const int64_t g = s < G ? s : (int64_t)wide[s - G];
            const double x = X[(g >> 4) * n * SC_TILE + cell * SC_TILE + (g & 15)];
            const uint32_t v = (uint32_t)(x >= 0.0 && x <= 255.0 ? x : 0.0);
            o[t] |= (s < G ? (v & 15u) : (v >> 4)) << (4 * e);
        }
    }
    out[((int64_t)grp * n + cell) * 8 + q] = make_uint4(o[0], o[1], o[2], o[3]);
}

// neighbour sums of every slot (uint16, slot order), cells in the graph's processing order like k_lag_u8
__global__ __launch_bounds__(256) void k_lag_nib(const long long *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                 const uint4 *__restrict__ nib, const int32_t *__restrict__ order, int64_t n,
                                                 uint4 *__restrict__ S16, int chunks)
{
    const int grp = blockIdx.y;
    const int64_t per_xcd = (int64_t)(gridDim.x >> 3);
    const int64_t chunk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // gridDim.x is a multiple of 8
    if (chunk >= chunks) return;
    const int q = threadIdx.x & 7, row = threadIdx.x >> 3;
    const uint4 *Xg = nib + (int64_t)grp * n * 8;
    const int64_t p0 = chunk * LAG8_CELLS_PER_BLOCK;
    for (int it = 0; it < LAG8_CELLS_PER_BLOCK / 32; ++it) {
        const int64_t pos = p0 + it * 32 + row;
        if (pos >= n) break;
        const int64_t cell = order ? order[pos] : pos;
        uint32_t acc[4][4];   // [word t][pair k]: 16-bit sums of nibbles e = 2 k (low half) and 2 k + 1 (high half)
        uint32_t ev[4] = {0u, 0u, 0u, 0u}, od[4] = {0u, 0u, 0u, 0u};   // byte sums of the even / odd nibbles (<= 16 neighbours x 15)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[t][k] = 0u;
        int pending = 0;
        auto flush = [&]() {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[t][k] += ((ev[t] >> (8 * k)) & 0xffu) | (((od[t] >> (8 * k)) & 0xffu) << 16);
                ev[t] = od[t] = 0u;
            }
            pending = 0;
        };
        for (long long e = indptr[cell]; e < indptr[cell + 1]; ++e) {
            const uint4 v = Xg[(int64_t)indices[e] * 8 + q];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) { ev[t] += w[t] & 0x0f0f0f0fu; od[t] += (w[t] >> 4) & 0x0f0f0f0fu; }
            if (++pending == 16) flush();
        }
        flush();
        uint4 *dst = S16 + ((int64_t)grp * n + cell) * 32 + q;
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t * 8] = make_uint4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
}

// S of a gene with a high nibble = S_lo + 16 S_hi, for BOTH of its slots
__global__ __launch_bounds__(256) void k_nib_fixup(uint16_t *__restrict__ S16, int64_t n, int64_t G, int nw,
                                                   const int32_t *__restrict__ wide)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nw) return;
    const int64_t cell = t / nw;
    const int j = (int)(t - cell * nw);
    const int64_t sl = wide[j], sh = G + j;
    uint16_t *a = S16 + ((sl >> 8) * n + cell) * 256 + (sl & 255), *b = S16 + ((sh >> 8) * n + cell) * 256 + (sh & 255);
    const uint32_t s = (uint32_t)*a + 16u * (uint32_t)*b;
    *a = (uint16_t)s;
    *b = (uint16_t)s;
}

// per slot and chunk of cells: sum nibble x S and sum S (integers, exact in fp64)
__global__ __launch_bounds__(256) void k_nib_colsum(const uint8_t *__restrict__ nib, const uint16_t *__restrict__ S16, int64_t n,
                                                    double *__restrict__ partT, double *__restrict__ partS, int chunks)
{
    const int grp = blockIdx.y, chunk = blockIdx.x, s = threadIdx.x;
    const int q = (s & 63) >> 3, t = s >> 6, e = s & 7;
    const int64_t c0 = (int64_t)chunk * LAG8_CELLS_PER_BLOCK, c1 = c0 + LAG8_CELLS_PER_BLOCK < n ? c0 + LAG8_CELLS_PER_BLOCK : n;
    unsigned long long T = 0, SS = 0;
    for (int64_t cell = c0; cell < c1; ++cell) {
        const uint32_t S = S16[((int64_t)grp * n + cell) * 256 + s];
        const uint32_t by = nib[((int64_t)grp * n + cell) * 128 + q * 16 + t * 4 + (e >> 1)];
        const uint32_t v = (e & 1) ? (by >> 4) : (by & 15u);
        T += (unsigned long long)v * S;
        SS += S;
    }
    partT[((int64_t)grp * chunks + chunk) * 256 + s] = (double)T;
    partS[((int64_t)grp * chunks + chunk) * 256 + s] = (double)SS;
}

// per gene: T_obs = sum x S = T[lo] + 16 T[hi], sum S (chunks in order)
__global__ void k_nib_colsum_final(const double *__restrict__ partT, const double *__restrict__ partS, int chunks, int64_t G,
                                   const int32_t *__restrict__ hi_slot, double *__restrict__ inum, double *__restrict__ slag)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int64_t sh = hi_slot[g];
    double T = 0.0, Th = 0.0, S = 0.0;
    for (int c = 0; c < chunks; ++c) {
        T += partT[((g >> 8) * chunks + c) * 256 + (g & 255)];
        S += partS[((g >> 8) * chunks + c) * 256 + (g & 255)];
        if (sh >= 0) Th += partT[((sh >> 8) * chunks + c) * 256 + (sh & 255)];
    }
    inum[g] = T + 16.0 * Th;
    slag[g] = S;
}

// sims / raw of the nibble form: the sums of a gene's slot(s) over the splits (ascending), T = T[lo] + 16 T[hi]
__global__ __launch_bounds__(256) void k_moran_finalize_nib(const double *__restrict__ partial, const double *__restrict__ seff,
                                                            const double *__restrict__ corr, const int32_t *__restrict__ hi_slot,
                                                            double *__restrict__ sims, double *__restrict__ raw, int n_perm,
                                                            int splits, int64_t n_genes, int64_t p0)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t / n_genes, g = t - p * n_genes;
    if (p >= n_perm) return;
    const int64_t sh = hi_slot[g];
    double lo = 0.0, hi = 0.0;
    for (int k = 0; k < splits; ++k) {
        lo += partial[((((g >> 8) * splits + k) * n_perm) + p) * 256 + (g & 255)];
        if (sh >= 0) hi += partial[((((sh >> 8) * splits + k) * n_perm) + p) * 256 + (sh & 255)];
    }
    const double s = lo + 16.0 * hi;
    raw[(p0 + p) * n_genes + g] = s;
    sims[(p0 + p) * n_genes + g] = seff[g] * (s - corr[g]);
}

// Everything the permutation kernels need, from the loaded tiles and the active graph:
//   value class + lattice decision per gene (one pass + one host sync), Z = X - centre, Lag = W Z (lattice genes: the
//   unweighted neighbour sums of the raw counts), I, the per-gene finalisation constants, the narrow copy of the batch.
// allow_lattice = false: ordinary arithmetic for every gene (tables that are not permutations: the identity
// sum_j x[idx[j]] = sum_j x[j] behind the lattice form does not hold for them).
// First half of moran_prepare: everything that needs neither a decision of the host nor the permutation count -- the
// weight sum, the graph moments (side stream), the gene moments and value classes -- is enqueued, with its small results
// on their way into pinned host memory, and nothing is waited for.  (r03 measured this half ENQUEUED AHEAD of the
// generator's launches for a resident expression -- the generator takes the host ~10 ms to enqueue -- : scoring started
// at 19 ms instead of 31, but the generator's own first launches queued behind these full-chip kernels, its chain
// started 4.5 ms later, and the step, which ends with the generator, was 1-2 ms LONGER.  Not kept; the split stays.)
static int moran_prepare_early(sc_ctx *c)
{
    const int64_t n = c->e_n, T = c->e_tiles;
    const int64_t Gpad = align_up64(T, 8) * SC_TILE;
    SC_REQUIRE(n > 0 && c->g_n == n, SC_ERR_STATE, "internal: early preparation without expression and graph");
    const bool need_s0 = !(c->s0_valid || c->s0_only_valid);
    const int blocks = need_s0 ? sc_graph_weight_sum_blocks(c) : 0;
    const size_t bytes = sizeof(double) * ((size_t)blocks + (size_t)(T * SC_TILE)) + sizeof(uint32_t) * 2 * (size_t)Gpad;
    if (bytes > c->prep_host_cap) {
        SC_HIP(hipStreamSynchronize(c->stream));   // (copies of an earlier, abandoned first half may still be writing the old buffer)
        if (c->prep_host) (void)hipHostFree(c->prep_host);
        c->prep_host = nullptr; c->prep_host_cap = 0;
        SC_HIP(hipHostMalloc(&c->prep_host, bytes, hipHostMallocDefault));
        c->prep_host_cap = bytes;
    }
    double *h_s0 = reinterpret_cast<double *>(c->prep_host), *h_xsum = h_s0 + blocks;
    uint32_t *h_flags = reinterpret_cast<uint32_t *>(h_xsum + T * SC_TILE), *h_xmax = h_flags + Gpad;
    c->prep_s0_blocks = blocks;
    if (need_s0) SC_TRY(sc_graph_weight_sum_launch(c, h_s0));
    SC_TRY(sc_graph_moments_begin(c));   // s1, s2 (p_norm, z-scores): on the side stream, out of this serial prelude
    SC_TRY(expr_moments(c));
    // ---- value classes ----
    SC_TRY(c->g_flags.ensure(sizeof(uint32_t) * (size_t)Gpad, &c->mem));
    SC_TRY(c->g_xmax.ensure(sizeof(uint32_t) * (size_t)Gpad, &c->mem));
    SC_HIP(hipMemsetAsync(c->g_flags.p, 0, sizeof(uint32_t) * (size_t)Gpad, c->stream));
    SC_HIP(hipMemsetAsync(c->g_xmax.p, 0, sizeof(uint32_t) * (size_t)Gpad, c->stream));
    hipLaunchKernelGGL(k_gene_stats, dim3((unsigned)ceil_div64(n, RED_ROWS_PER_BLOCK), (unsigned)T), dim3(256), 0, c->stream,
                       c->X.as<double>(), n, c->g_flags.as<uint32_t>(), c->g_xmax.as<uint32_t>());
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(h_flags, c->g_flags.p, sizeof(uint32_t) * (size_t)Gpad, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(h_xmax, c->g_xmax.p, sizeof(uint32_t) * (size_t)Gpad, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(h_xsum, c->g_xsum.p, sizeof(double) * (size_t)(T * SC_TILE), hipMemcpyDeviceToHost, c->stream));
    c->prep_early = true;
    return SC_OK;
}

static int moran_prepare(sc_ctx *c, int64_t n_perm, bool allow_lattice)
{
    const int64_t n = c->e_n, T = c->e_tiles, G = c->e_genes;
    const int64_t Tpad = align_up64(T, 8), Gpad = Tpad * SC_TILE;
    if (!c->prep_early) SC_TRY(moran_prepare_early(c));
    c->prep_early = false;
    SC_HIP(hipStreamSynchronize(c->stream));   // the ONE synchronisation of the preparation
    const double *h_s0 = reinterpret_cast<const double *>(c->prep_host), *xsum = h_s0 + c->prep_s0_blocks;
    const uint32_t *flags = reinterpret_cast<const uint32_t *>(xsum + T * SC_TILE), *xmax = flags + Gpad;
    if (c->prep_s0_blocks > 0) sc_graph_weight_sum_collect(c, h_s0, c->prep_s0_blocks);
    std::vector<double> lat((size_t)Gpad, 0.0);
    int bits = c->source_bits_min;
    bool lat_any = false, lat_all = true;
    // equal weights AND equal degrees: sum_i z_i lag[pi(i)] = w (T - mean sum S) drops the term -w mean sum_i z_i deg[pi(i)],
    // which vanishes only when deg is constant (r03 advisor finding: a binary adjacency with unequal rows took this path)
    const bool lattice_graph = allow_lattice && c->g_uniform_w > 0.0 && c->g_regular;
    for (int64_t g = 0; g < G; ++g) {
        const int cls = !(flags[(size_t)g] & 1u) ? 8 : !(flags[(size_t)g] & 2u) ? 16 : !(flags[(size_t)g] & 4u) ? 32 : 64;
        if (cls > bits) bits = cls;
        // every partial sum of T_p = sum_j S_j x[inv_p(j)] stays an integer below 2^53: T <= max_j S_j * sum x
        const bool l = lattice_graph && cls <= 16 &&
                       (double)c->g_deg_max * (double)xmax[(size_t)g] * xsum[(size_t)g] < 4.0e15;
        lat[(size_t)g] = l ? 1.0 : 0.0;
        lat_any |= l;
        lat_all &= l;
    }
    if (bits == 8 && !lat_all) bits = 16;      // the uint8 kernel does not centre
    if (n >= ((int64_t)1 << 25)) bits = 64;    // (rows beyond a 32-bit byte offset: the fp64-row kernel with 64-bit addresses)
    c->narrow_bits = bits;
    c->lat_any = lat_any;
    SC_HIP(hipMemcpyAsync(c->g_lat.p, lat.data(), sizeof(double) * (size_t)Gpad, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_moran_centres, dim3((unsigned)ceil_div64(T * SC_TILE, 256)), dim3(256), 0, c->stream,
                       c->g_mean.as<double>(), c->g_lat.as<double>(), c->g_meanc.as<double>(), T * SC_TILE);
    // ---- operands ----
    const size_t gb = (size_t)Gpad * sizeof(double);
    SC_TRY(c->g_slag.ensure(gb, &c->mem));
    SC_TRY(c->g_seff.ensure(gb, &c->mem));
    SC_TRY(c->g_corr.ensure(gb, &c->mem));
    SC_TRY(c->g_thr.ensure(gb, &c->mem));
    SC_TRY(c->g_I.ensure(gb, &c->mem));
    {   // fp64 tiles, or (uint8 batches) 256 bytes of 16-bit sums per cell and 128-gene group -- more than ONE fp64 tile
        const size_t tiles_b = (size_t)T * n * SC_TILE * sizeof(double), sums16_b = (size_t)ceil_div64(T, 8) * n * 256;
        SC_TRY(c->Lag.ensure(tiles_b > sums16_b ? tiles_b : sums16_b, &c->mem));
    }
    // an all-lattice uint8 batch (count data on a kNN graph): narrow copy first, then neighbour sums + both column sums
    // from the uint8 rows in one pass (k_lag_u8); no Z tiles at all (the lattice operand IS the raw value)
    const bool u8_prelude = n_perm > 0 && bits == 8 && lat_all && c->g_deg_max <= 257 && !getenv("SC_NO_U8_PRELUDE");
    // ... or as NIBBLE slots (r04, "the 4-bit source" above) when that takes fewer gathered rows per (permutation, cell)
    int64_t nw = 0;
    uint32_t xmax_all = 0;
    for (int64_t g = 0; g < G; ++g) {
        if (xmax[(size_t)g] >= 16u) ++nw;
        if (xmax[(size_t)g] > xmax_all) xmax_all = xmax[(size_t)g];
    }
    const int64_t nib_slots = G + nw, nib_groups = ceil_div64(nib_slots, 256);
    const bool nib = u8_prelude && c->source_bits_min <= 4 && nib_groups < ceil_div64(G, 128) &&
                     15.0 * (double)xmax_all * (double)c->g_deg_max * (double)score_cells_per_split(n) < 2.0e9;
    c->nib_groups = nib ? (int)nib_groups : 0;
    // ... with the neighbour sums kept as the 16-bit integers they are (r04): a quarter of the bytes k_lag_u8 writes in the
    // serial prelude and of the lag bytes every scoring launch streams (SC_LAG_FP64: the r03 form, for A/B runs)
    const bool lag16 = u8_prelude && !nib && !getenv("SC_LAG_FP64");
    bool narrow_packed = false;   // the narrow copy of the batch exists already (built in front of the lag that reads it)
    c->lag_u16 = lag16;
    if (nib) {
        c->lm_valid = false;   // (Lag is about to be rewritten)
        bits = 4;
        c->narrow_bits = 4;
        c->lag_u16 = true;
        // slot map: [Gpad] slot of the gene's high nibble (-1: none) | [nw] the genes that have one, ascending
        std::vector<int32_t> map((size_t)Gpad + (size_t)(nw > 0 ? nw : 1), -1);
        for (int64_t g = 0, j = 0; g < G; ++g)
            if (xmax[(size_t)g] >= 16u) { map[(size_t)g] = (int32_t)(G + j); map[(size_t)Gpad + (size_t)j] = (int32_t)g; ++j; }
        SC_TRY(c->nib_map.ensure(sizeof(int32_t) * map.size(), &c->mem));
        SC_HIP(hipMemcpyAsync(c->nib_map.p, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice, c->stream));
        SC_HIP(hipStreamSynchronize(c->stream));   // (`map` is a local)
        const int32_t *hi_slot = c->nib_map.as<int32_t>(), *wide = hi_slot + Gpad;
        SC_TRY(c->X32.ensure(sizeof(uint4) * (size_t)nib_groups * n * 8, &c->mem));
        SC_TRY(c->Lag.ensure(sizeof(uint4) * (size_t)nib_groups * n * 32, &c->mem));
        hipLaunchKernelGGL(k_pack_nib, dim3((unsigned)ceil_div64(n * 8, 256), (unsigned)nib_groups), dim3(256), 0, c->stream,
                           c->X.as<double>(), c->X32.as<uint4>(), n, G, nib_slots, wide);
        const int chunks = (int)ceil_div64(n, LAG8_CELLS_PER_BLOCK);
        const int32_t *order = (c->g_order_captured && c->g_n == n && c->g_order.p) ? c->g_order.as<int32_t>() : nullptr;
        {
            KernelTimerScope ts(c, SC_K_LAG);
            hipLaunchKernelGGL(k_lag_nib, dim3((unsigned)align_up64(chunks, 8), (unsigned)nib_groups), dim3(256), 0, c->stream,
                               c->g_indptr.as<long long>(), c->g_indices.as<int32_t>(), c->X32.as<uint4>(), order, n,
                               c->Lag.as<uint4>(), chunks);
            if (nw > 0)
                hipLaunchKernelGGL(k_nib_fixup, dim3((unsigned)ceil_div64(n * nw, 256)), dim3(256), 0, c->stream,
                                   c->Lag.as<uint16_t>(), n, G, (int)nw, wide);
        }
        SC_TRY(c->red_tmp.ensure(sizeof(double) * 2 * (size_t)nib_groups * chunks * 256, &c->mem));
        double *partT = c->red_tmp.as<double>(), *partS = partT + (size_t)nib_groups * chunks * 256;
        hipLaunchKernelGGL(k_nib_colsum, dim3((unsigned)chunks, (unsigned)nib_groups), dim3(256), 0, c->stream,
                           c->X32.as<uint8_t>(), c->Lag.as<uint16_t>(), n, partT, partS, chunks);
        hipLaunchKernelGGL(k_nib_colsum_final, dim3((unsigned)ceil_div64(G, 256)), dim3(256), 0, c->stream, partT, partS, chunks, G,
                           hi_slot, c->g_Inum.as<double>(), c->g_slag.as<double>());
        SC_HIP(hipGetLastError());
    } else if (u8_prelude) {
        c->lm_valid = false;   // (Lag is about to be rewritten)
        SC_TRY(c->X32.ensure(sizeof(float) * (size_t)((T + 1) / 2) * n * 32, &c->mem));
        hipLaunchKernelGGL(k_pack_narrow<8>, dim3((unsigned)ceil_div64(n * 8, 256), (unsigned)ceil_div64(T, 8)), dim3(256), 0,
                           c->stream, c->X.as<double>(), c->X32.as<uint4>(), n, T);
        const int chunks = (int)ceil_div64(n, LAG8_CELLS_PER_BLOCK);
        SC_TRY(c->red_tmp.ensure(sizeof(double) * 2 * (size_t)T * chunks * SC_TILE, &c->mem));
        const int32_t *order = (c->g_order_captured && c->g_n == n && c->g_order.p) ? c->g_order.as<int32_t>() : nullptr;
        {
            KernelTimerScope ts(c, SC_K_LAG);
            hipLaunchKernelGGL(k_lag_u8, dim3((unsigned)align_up64(chunks, 8), (unsigned)ceil_div64(T, 8)), dim3(256), 0, c->stream,
                               c->g_indptr.as<long long>(), c->g_indices.as<int32_t>(), c->X32.as<uint4>(), order, n, (int)T,
                               (int64_t)n * SC_TILE, c->Lag.as<double>(), lag16 ? c->Lag.as<uint32_t>() : nullptr,
                               c->red_tmp.as<double>(), chunks);
        }
        hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)T), dim3(SC_TILE), 0, c->stream, c->red_tmp.as<double>(),
                           c->g_Inum.as<double>(), (double *)nullptr, chunks, 1.0);
        hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)T), dim3(SC_TILE), 0, c->stream,
                           c->red_tmp.as<double>() + (size_t)T * chunks * SC_TILE, c->g_slag.as<double>(), (double *)nullptr, chunks, 1.0);
        SC_HIP(hipGetLastError());
    } else {
        SC_TRY(expr_write_z(c, c->g_meanc.as<double>()));
        const bool lag_from_rows = n_perm > 0 && bits == 32 && !getenv("SC_LAG_FP64_ROWS");   // (SC_LAG_FP64_ROWS: the r03 form, A/B)
        if (lag_from_rows) {   // the float32 narrow copy first, then the lag from ITS rows (half the gathered bytes of the fp64 tiles)
            SC_TRY(c->X32.ensure(sizeof(float) * (size_t)((T + 1) / 2) * n * 32, &c->mem));
            hipLaunchKernelGGL(k_pack_narrow<32>, dim3((unsigned)ceil_div64(n * 8, 256), (unsigned)ceil_div64(T, 2)), dim3(256), 0,
                               c->stream, c->X.as<double>(), c->X32.as<uint4>(), n, T);
            const int32_t *order = (c->g_order_captured && c->g_n == n && c->g_order.p) ? c->g_order.as<int32_t>() : nullptr;
            KernelTimerScope ts(c, SC_K_LAG);
            hipLaunchKernelGGL(k_lag_f32rows, dim3((unsigned)align_up64(ceil_div64(n * 8, 256), 8), (unsigned)ceil_div64(T, 2)),
                               dim3(256), 0, c->stream, c->g_indptr.as<int64_t>(), c->g_indices.as<int32_t>(),
                               c->g_data.as<double>(), c->X32.as<uint4>(), c->g_meanc.as<double>(), c->Lag.as<double>(), n, (int)T,
                               lat_any ? c->g_lat.as<double>() : nullptr, order);
            SC_HIP(hipGetLastError());
            narrow_packed = true;
        } else
        SC_TRY(launch_lag(c, c->g_indptr, c->g_indices, c->g_data, c->Z.as<double>(), c->Lag.as<double>(),
                          lat_any ? c->g_lat.as<double>() : nullptr));
        SC_TRY(colsum<OP_MUL>(c, c->Z.as<double>(), c->Lag.as<double>(), c->g_Inum.as<double>(), 1.0));
        if (lat_any) SC_TRY(colsum<OP_ID>(c, c->Lag.as<double>(), nullptr, c->g_slag.as<double>(), 1.0));
    }
    SC_TRY(c->sims.ensure(sizeof(double) * (size_t)(T * SC_TILE) * (size_t)(n_perm > 0 ? n_perm : 1), &c->mem));
    hipLaunchKernelGGL(k_moran_scale, dim3((unsigned)ceil_div64(T * SC_TILE, 256)), dim3(256), 0, c->stream,
                       c->g_z2.as<double>(), c->g_Inum.as<double>(), c->g_slag.as<double>(), c->g_mean.as<double>(),
                       c->g_lat.as<double>(), c->g_seff.as<double>(), c->g_corr.as<double>(), c->g_thr.as<double>(),
                       c->g_I.as<double>(), (double)n / c->s0, c->g_uniform_w, T * SC_TILE);
    SC_HIP(hipGetLastError());
    if (n_perm > 0) {
        // partial sums for one chunk of permutations (<= PERM_CHUNK): the persistent kernel keeps one row per
        // (128-gene-padded gene, split, permutation); the index-row kernel one per (16 genes, split, permutation)
        int64_t cps = 0;
        const int splits64 = pick_splits(n, 1, &cps);  // upper bound on the index-row kernel's split count
        const int64_t score_splits = ceil_div64(n, score_cells_per_split(n));
        size_t narrow_rows = (size_t)score_splits * (size_t)align_up64(T * SC_TILE, 128);
        if (nib && (size_t)score_splits * (size_t)nib_groups * 256 > narrow_rows) narrow_rows = (size_t)score_splits * (size_t)nib_groups * 256;
        const size_t wide_rows = (size_t)splits64 * SC_TILE;
        SC_TRY(c->partial.ensure(sizeof(double) * (size_t)n_perm * (narrow_rows > wide_rows ? narrow_rows : wide_rows), &c->mem));
        if (bits < 64 && bits > 4 && !u8_prelude && !narrow_packed) {
            // the gathered operand: the raw values in the narrowest type that holds every gene of the batch exactly
            const int64_t T32 = (T + 1) / 2;
            SC_TRY(c->X32.ensure(sizeof(float) * (size_t)T32 * n * 32, &c->mem));   // >= the uint16 / uint8 copies
            const int tg = bits == 8 ? 8 : bits == 16 ? 4 : 2;
            const dim3 grid((unsigned)ceil_div64(n * 8, 256), (unsigned)ceil_div64(T, tg));
            auto pack = bits == 8 ? k_pack_narrow<8> : bits == 16 ? k_pack_narrow<16> : k_pack_narrow<32>;
            hipLaunchKernelGGL(pack, grid, dim3(256), 0, c->stream, c->X.as<double>(), c->X32.as<uint4>(), n, T);
            SC_HIP(hipGetLastError());
        }
    }
    return SC_OK;
}

// inverse rows [p0, p1) of the active table on stream s
static int invert_rows(sc_ctx *c, int64_t p0, int64_t p1, hipStream_t s)
{
    const int rows = (int)(p1 - p0);
    if (rows <= 0) return SC_OK;
    const int groups = (rows + 7) / 8;
    // (the table's own length, not the expression's: a generator job may run before any expression is loaded)
    hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)(groups * INV_BLOCKS_PER_ROW * 8)), dim3(256), 0, s,
                       c->perm.as<int32_t>() + p0 * c->p_stride, c->inv.as<int32_t>() + p0 * c->p_stride, c->p_n,
                       c->p_stride, rows);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// After a seeded pipeline that only generated the inverse table: materialise the permutation table itself (the
// inverse of the inverse) for callers that use the resident table afterwards.
int sc_perm_forward_ensure(sc_ctx *c)
{
    if (c->perm_forward_valid || c->p_count <= 0) return SC_OK;
    const int rows = (int)c->p_count;
    const int groups = (rows + 7) / 8;
    hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)(groups * INV_BLOCKS_PER_ROW * 8)), dim3(256), 0, c->stream,
                       c->inv.as<int32_t>(), c->perm.as<int32_t>(), c->p_n, c->p_stride, rows);
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(c->stream));
    c->perm_forward_valid = true;
    return SC_OK;
}

// The scoring kernels gather through the INVERSE rows, which exist only for true permutations: a table uploaded by
// the caller is checked once (inverse of the inverse).  *bijective = false: its rows are arbitrary index maps and
// take the index-row kernel (k_moran_perm).
static int moran_table_is_bijective(sc_ctx *c, int64_t n_perm, bool *bijective)
{
    *bijective = false;
    if (n_perm <= 0) return SC_OK;
    const int64_t rows = c->p_count > n_perm ? c->p_count : n_perm;  // the WHOLE table is checked once
    SC_TRY(c->inv.ensure(sizeof(int32_t) * (size_t)(c->p_stride * rows + 32), &c->mem));
    if (!c->perm_bijective && !c->perm_checked) {
        SC_TRY(invert_rows(c, 0, rows, c->stream));
        SC_TRY(c->perm_flag.ensure(sizeof(unsigned long long), &c->mem));
        SC_HIP(hipMemsetAsync(c->perm_flag.p, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_check_inverse, dim3(2048), dim3(256), 0, c->stream, c->perm.as<int32_t>(),
                           c->inv.as<int32_t>(), c->p_n, c->p_stride, rows, c->perm_flag.as<int>());
        int bad = 0;
        SC_HIP(hipMemcpyAsync(&bad, c->perm_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        SC_HIP(hipStreamSynchronize(c->stream));
        c->perm_checked = true;
        c->perm_bijective = (bad == 0);
        if (c->perm_bijective) c->inv_rows_valid = rows;
    }
    *bijective = c->perm_bijective;
    return SC_OK;
}

template <int BITS, int CB, bool BIG, bool L16 = false>
static void launch_score(sc_ctx *c, int wgs, const uint4 *rows, int64_t p0, int cnt, int64_t cps, int splits, int groups)
{
    static const bool private_lag = getenv("SC_SCORE_PRIVATE_LAG") != nullptr;   // development: the r02 form, for A/B runs
    // the workgroup form scores 128 permutations per task; wavefronts beyond a short chunk's permutations only help with
    // the lag rows.  Below SCORE_WG_MIN_PERMS live permutations a compute unit has too few gathers in flight: such chunks
    // take the per-wavefront form -- same results (SC_SCORE_WG_MIN: development, to sweep the threshold)
    static const int wg_min = getenv("SC_SCORE_WG_MIN") ? atoi(getenv("SC_SCORE_WG_MIN")) : SCORE_WG_MIN_PERMS;
    const int last_task = cnt % (8 * SCORE_WAVES);
    if (BITS == 4 || (!private_lag && (last_task == 0 || last_task >= (BITS == 64 ? 6 * SCORE_WAVES : wg_min)))) {
        const int64_t tasks = (int64_t)groups * splits * ((cnt + 8 * SCORE_WAVES - 1) / (8 * SCORE_WAVES));
        if (wgs > tasks) wgs = (int)tasks;
        // (r03 measured the grid rounded to whole rounds of tasks -- 1956 tasks are 13 rounds on 160 workgroups, 12 on 163,
        // and 151 suffice for 13: the launches were 3 % shorter with 163, the generator 4 % slower with 3 compute units
        // fewer, the step the same within its noise either way, and with 151 as well.  Not kept.)
        hipLaunchKernelGGL((k_moran_score_wg<BITS, CB, BIG, L16>), dim3((unsigned)wgs), dim3(SCORE_WAVES * 64), 0, c->stream, rows,
                           c->Lag.as<double>(), (int64_t)c->e_n * SC_TILE, (int)c->e_tiles, c->g_meanc.as<double>(),
                           c->inv.as<int32_t>() + p0 * c->p_stride, c->partial.as<double>(), c->e_n, c->p_stride, cnt, cps,
                           splits, groups);
        return;
    }
    if constexpr (BITS != 4)
    hipLaunchKernelGGL((k_moran_score<BITS, CB, BIG, L16>), dim3((unsigned)(wgs * (SCORE_WAVES / SCORE_PRIVATE_WAVES))),
                       dim3(SCORE_PRIVATE_WAVES * 64), 0, c->stream, rows,
                       c->Lag.as<double>(), (int64_t)c->e_n * SC_TILE, (int)c->e_tiles, c->g_meanc.as<double>(),
                       c->inv.as<int32_t>() + p0 * c->p_stride, c->partial.as<double>(), c->e_n, c->p_stride, cnt, cps,
                       splits, groups);
}

// score permutations [p0, p1) of the active table for every gene (on the context stream).
// bits: 8 / 16 / 32 / 64 = the persistent kernel gathers rows of that element width through the INVERSE permutation
// (needs inverse rows [p0, p1); invert_here launches that inversion first); 0 = the table's rows are arbitrary index
// maps: the index-row kernel over the fp64 tiles.
static int moran_perm_range(sc_ctx *c, int64_t p0, int64_t p1, int bits, bool invert_here)
{
    const int64_t n = c->e_n, G = c->e_genes, T = c->e_tiles;
    const size_t tile_elems = (size_t)n * SC_TILE;
    const int cnt = (int)(p1 - p0);
    if (cnt <= 0) return SC_OK;
    c->last_source_bits = bits ? bits : 64;
    if (bits) {
        const bool big = n >= ((int64_t)1 << 25);
        SC_REQUIRE(!big || bits == 64, SC_ERR_STATE, "internal: %lld cells need the 64-bit-address scoring kernel", (long long)n);
        if (invert_here) SC_TRY(invert_rows(c, p0, p1, c->stream));
        const int GP = bits == 4 ? 256 : bits == 8 ? 128 : bits == 16 ? 64 : bits == 32 ? 32 : 16;
        const int groups = bits == 4 ? c->nib_groups : (int)ceil_div64(T * SC_TILE, GP);
        const int64_t cps = score_cells_per_split(n);
        const int splits = (int)ceil_div64(n, cps);
        // one workgroup per compute unit: all of them, or all but those left to a generator that runs beside us
        if (c->n_cus <= 0) {
            hipDeviceProp_t prop;
            SC_HIP(hipGetDeviceProperties(&prop, c->device));
            c->n_cus = prop.multiProcessorCount;
        }
        int wgs = c->n_cus - (c->score_leave_cus > 0 && c->score_leave_cus < c->n_cus ? c->score_leave_cus : 0);
        const int64_t tasks = (int64_t)groups * splits * ((cnt + 7) / 8);
        if ((int64_t)wgs * SCORE_WAVES > tasks) wgs = (int)ceil_div64(tasks, SCORE_WAVES);
        {
            KernelTimerScope ts(c, SC_K_MORAN_PERM);
            // (8 cells per stage were measured for the narrow sources too: under the 128-VGPR cap of the 1024-thread form they spill)
            if (bits == 4) launch_score<4, 4, false>(c, wgs, c->X32.as<uint4>(), p0, cnt, cps, splits, groups);
            else if (bits == 8 && c->lag_u16) launch_score<8, 4, false, true>(c, wgs, c->X32.as<uint4>(), p0, cnt, cps, splits, groups);
            else if (bits == 8) launch_score<8, 4, false>(c, wgs, c->X32.as<uint4>(), p0, cnt, cps, splits, groups);
            else if (bits == 16) launch_score<16, 4, false>(c, wgs, c->X32.as<uint4>(), p0, cnt, cps, splits, groups);
            else if (bits == 32) launch_score<32, 4, false>(c, wgs, c->X32.as<uint4>(), p0, cnt, cps, splits, groups);
            else if (!big) launch_score<64, 8, false>(c, wgs, c->Z.as<uint4>(), p0, cnt, cps, splits, groups);
            else launch_score<64, 8, true>(c, wgs, c->Z.as<uint4>(), p0, cnt, cps, splits, groups);
        }
        if (bits == 4) {
            hipLaunchKernelGGL(k_moran_finalize_nib, dim3((unsigned)ceil_div64((int64_t)cnt * G, 256)), dim3(256), 0, c->stream,
                               c->partial.as<double>(), c->g_seff.as<double>(), c->g_corr.as<double>(), c->nib_map.as<int32_t>(),
                               c->sims.as<double>(), c->sims_raw.as<double>(), cnt, splits, G, p0);
            SC_HIP(hipGetLastError());
            return SC_OK;
        }
        const dim3 fgrid((unsigned)ceil_div64((int64_t)cnt * GP, 256), (unsigned)groups);
        auto fin = bits == 8 ? k_moran_finalize_groups<128> : bits == 16 ? k_moran_finalize_groups<64>
                 : bits == 32 ? k_moran_finalize_groups<32> : k_moran_finalize_groups<16>;
        hipLaunchKernelGGL(fin, fgrid, dim3(256), 0, c->stream, c->partial.as<double>(), c->g_seff.as<double>(),
                           c->g_corr.as<double>(), c->sims.as<double>(), c->sims_raw.as<double>(), cnt, splits, G, p0);
        SC_HIP(hipGetLastError());
        return SC_OK;
    }
    const int ptiles = (int)ceil_div64(cnt, MP_PERMS_PER_BLOCK);
    int64_t cps = 0;
    const int splits = pick_splits(n, ptiles, &cps);
    for (int64_t t = 0; t < T; ++t) {
        {
            KernelTimerScope ts(c, SC_K_MORAN_PERM);
            hipLaunchKernelGGL(k_moran_perm, dim3((unsigned)splits, (unsigned)ptiles), dim3(256), 0, c->stream,
                               c->Z.as<double>() + t * tile_elems, c->Lag.as<double>() + t * tile_elems,
                               c->perm.as<int32_t>() + p0 * c->p_stride, c->partial.as<double>(), n, c->p_stride,
                               cnt, cps);
        }
        hipLaunchKernelGGL(k_moran_finalize, dim3((unsigned)ceil_div64((int64_t)cnt * SC_TILE, 256)), dim3(256), 0,
                           c->stream, c->partial.as<double>(), c->g_seff.as<double>(), c->g_corr.as<double>(),
                           c->sims.as<double>(), c->sims_raw.as<double>(), cnt, splits, G, t * SC_TILE, p0);
    }
    SC_HIP(hipGetLastError());
    return SC_OK;
}

static int moran_alloc_sims(sc_ctx *c, int64_t n_perm)
{
    const size_t bytes = sizeof(double) * (size_t)(c->e_tiles * SC_TILE) * (size_t)(n_perm > 0 ? n_perm : 1);
    SC_TRY(c->sims.ensure(bytes, &c->mem));
    SC_TRY(c->sims_raw.ensure(bytes, &c->mem));
    return SC_OK;
}

static int moran_finish(sc_ctx *c, int64_t n_perm, double *I_out, double *sims_out, int64_t *count_ge_out,
                        double *sim_sum_out, double *sim_sumsq_out)
{
    const int64_t G = c->e_genes;
    if (n_perm > 0) {
        SC_TRY(c->counts.ensure(sizeof(long long) * (size_t)G, &c->mem));
        SC_TRY(c->sim_sum.ensure(sizeof(double) * (size_t)G, &c->mem));
        SC_TRY(c->sim_sumsq.ensure(sizeof(double) * (size_t)G, &c->mem));
        hipLaunchKernelGGL(k_moran_count, dim3((unsigned)G), dim3(256), 0, c->stream, c->sims.as<double>(),
                           c->sims_raw.as<double>(), c->g_thr.as<double>(), c->g_lat.as<double>(), c->g_z2.as<double>(),
                           (int)n_perm, G, c->counts.as<long long>(), c->sim_sum.as<double>(),
                           c->sim_sumsq.as<double>());
        SC_HIP(hipGetLastError());
        if (sims_out)
            SC_HIP(hipMemcpyAsync(sims_out, c->sims.p, sizeof(double) * (size_t)n_perm * (size_t)G,
                                  hipMemcpyDeviceToHost, c->stream));
        if (count_ge_out)
            SC_HIP(hipMemcpyAsync(count_ge_out, c->counts.p, sizeof(int64_t) * (size_t)G, hipMemcpyDeviceToHost,
                                  c->stream));
        if (sim_sum_out)
            SC_HIP(hipMemcpyAsync(sim_sum_out, c->sim_sum.p, sizeof(double) * (size_t)G, hipMemcpyDeviceToHost,
                                  c->stream));
        if (sim_sumsq_out)
            SC_HIP(hipMemcpyAsync(sim_sumsq_out, c->sim_sumsq.p, sizeof(double) * (size_t)G,
                                  hipMemcpyDeviceToHost, c->stream));
    }
    SC_HIP(hipMemcpyAsync(I_out, c->g_I.p, sizeof(double) * (size_t)G, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

extern "C" int sc_moran(sc_ctx *c, int64_t n_perm, double *I_out, double *sims_out, int64_t *count_ge_out,
                        double *sim_sum_out, double *sim_sumsq_out)
{
    SC_TRY(moran_check(c, n_perm, I_out));
    if (n_perm > 0) {
        SC_REQUIRE(c->p_count >= n_perm, SC_ERR_STATE, "sc_moran: permutation table holds %lld rows, need %lld",
                   (long long)c->p_count, (long long)n_perm);
        SC_REQUIRE(c->p_n == c->e_n, SC_ERR_INVALID, "sc_moran: permutation length %lld != n_cells %lld",
                   (long long)c->p_n, (long long)c->e_n);
    }
    // A table left by sc_moran_seeded may exist only as inverse rows, which is all the scoring kernel reads: the
    // permutation rows themselves are materialised only when rows are needed whose inverse is not there yet.
    const bool inverse_suffices = n_perm > 0 && c->perm_bijective && c->inv_rows_valid >= n_perm;
    if (n_perm > 0 && !inverse_suffices) SC_TRY(sc_perm_forward_ensure(c));
    bool bijective = true;
    SC_TRY(moran_table_is_bijective(c, n_perm, &bijective));
    SC_TRY(moran_prepare(c, n_perm < PERM_CHUNK ? n_perm : PERM_CHUNK, n_perm <= 0 || bijective));
    SC_TRY(moran_alloc_sims(c, n_perm));
    const int bits = bijective ? c->narrow_bits : 0;
    for (int64_t p0 = 0; p0 < n_perm; p0 += PERM_CHUNK) {
        const int64_t p1 = p0 + PERM_CHUNK < n_perm ? p0 + PERM_CHUNK : n_perm;
        SC_TRY(moran_perm_range(c, p0, p1, bits, bits != 0 && p1 > c->inv_rows_valid));
        if (bits != 0 && p1 > c->inv_rows_valid) c->inv_rows_valid = p1;
    }
    return moran_finish(c, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out);
}

// (what the number means and how it was chosen: the comment in front of moran_seeded_streams)
#ifndef SCORE_RESERVED_CUS
#define SCORE_RESERVED_CUS 112   // r04 (two runs of 20 steps each, same box, ms per step): 96 -> 163.1 / 162.2, 112 -> 157.7 / 158.5,
                                 // 128 -> 162.2 / 162.6, 144 -> 171.0 / 169.5; the scoring launches take the same 117 ms on 144 CUs
                                 // as on 160 (13 rounds of tasks either way), the generator's preparation gets its CUs sooner
#endif
#ifndef PIPE_AHEAD
#define PIPE_AHEAD 3         // launch units the generator's preparation runs ahead of its chain inside the pipeline
#endif
#ifndef PIPE_FIRST
#define PIPE_FIRST 32        // permutations of the first pipeline chunk
#endif
#ifndef PIPE_TAIL
#define PIPE_TAIL "96,48,24" // the last chunks, tapering: see pipe_tail_perms
#endif
#ifndef PIPE_SWAP_STREAMS
#define PIPE_SWAP_STREAMS 2  // swap chunks in flight (they are latency-bound: two overlap almost for free)
#endif

// permutations of the pipeline's first chunk (SC_PIPE_FIRST: development, to sweep the schedule)
static int64_t pipe_first_perms()
{
    int64_t v = PIPE_FIRST;
    if (const char *e = getenv("SC_PIPE_FIRST")) v = atoi(e);
    return v < 8 || v > PERM_CHUNK ? PIPE_FIRST : v;
}
// The job ends with what is left once the generator's chain has finished: the swaps of its last chunk (~10 ms whatever
// its size: one workgroup per permutation, latency-bound) and the consumption of every chunk not consumed yet.  Behind
// a 128-permutation chunk that is its swaps AND its 10-ms consumption; tapering chunks leave a few milliseconds (bench
// step, same box, ms: one 32-permutation last chunk 173.5 / 174.6; 64,32: 172.4 / 178.8; 64,32,16: 172.8 / 173.1; 96,48,24: 169.4 / 169.4).
// SC_PIPE_TAIL="a,b,...": development, to sweep the schedule ("0": no short chunks at the end).
static std::vector<int64_t> pipe_tail_perms()
{
    const char *e = getenv("SC_PIPE_TAIL");
    std::vector<int64_t> t;
    for (const char *p = e ? e : PIPE_TAIL; *p;) {
        char *end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p) break;
        if (v >= 8 && v <= PERM_CHUNK) t.push_back(v);
        p = *end == ',' ? end + 1 : end;
    }
    return t;
}
static int64_t pipe_tail_total()
{
    int64_t s = 0;
    for (int64_t v : pipe_tail_perms()) s += v;
    return s;
}

// The generator / consumer pipeline shared by sc_moran_seeded and sc_lee_seeded: numpy-exact permutation rows
// [0, n_perm) of length n are produced chunk by chunk on the generator's streams (stream2: rejection scan chain,
// stream_pg: its preparation, stream_px: verification + expansion, stream3/4: Fisher-Yates swaps) while
// `score(p0, p1)` consumes finished chunks on the context stream.  table: 0 = permutation rows (c->perm),
// 1 = inverse rows only (c->inv; the same transpositions in ascending order), 2 = both (rows + k_invert_perm).
// `after_first` runs on the host right after the first chunk of the generator has been enqueued (the generator is
// the longest chain and depends on nothing else; everything host-blocking of the consumer's set-up goes here).
static int pipe_generate(sc_ctx *c, PermPipe &pp, int64_t k)
{
    hipEvent_t &scanned = pp.ev[(size_t)(2 * k)], &swapped = pp.ev[(size_t)(2 * k + 1)];
    // Two swap kernels in flight only for the job's LAST chunks (r04).  A swap workgroup is 8 wavefronts that live ~10 ms;
    // two chunks' worth of them (256) spread over the ~96 CUs the scoring kernel leaves, next to the table builders' two-
    // wavefront workgroups, left no CU with the 16 free wavefront slots a 1024-thread preparation workgroup needs: the
    // chain's clock profile showed its units arriving 1-8 ms late behind every chunk boundary (55 k clocks of waiting per
    // permutation; 37 k with one swap kernel at a time).  The tapering last chunks arrive 2-6 ms apart after the chain
    // is all but done, and keep overlapping.
    const int64_t chunks = (int64_t)pp.bounds.size() - 1;
    const bool overlap = PIPE_SWAP_STREAMS > 1 && (k & 1) && (k >= chunks - 3 || getenv("SC_SWAP_OVERLAP_ALL") != nullptr);
    hipStream_t sw = overlap ? c->stream4 : c->stream3;
    SC_HIP(hipEventCreateWithFlags(&scanned, hipEventDisableTiming));
    SC_HIP(hipEventCreateWithFlags(&swapped, hipEventDisableTiming));
    SC_TRY(permgen_scan_chunk(c, &pp.job, pp.bounds[(size_t)k + 1], c->stream2, c->stream_px, scanned));
    SC_HIP(hipStreamWaitEvent(sw, scanned, 0));
    // Two permutations per swap workgroup while the chain still runs -- workgroups of the preparation kernels' own size --
    // but only beside the Moran scoring kernel (the one consumer that fills its CUs with wavefronts that live for
    // milliseconds): there the step gains 5 ms (151 against 156).  Two permutations in lockstep take 13-15 ms per chunk
    // instead of 10-12, and a light consumer (Lee's row sums, the local counts) leaves the chain at 12.8 ms per chunk:
    // with the pairs the swap stream became the bottleneck (Lee 10 x 10 pairs: 27.8 against 24.9 ms per pair).
    const int pw = (k < chunks - 3 && c->score_leave_cus > 8) ? 2 : 1;
    SC_TRY(permgen_swap_chunk(c, &pp.job, pp.bounds[(size_t)k], pp.bounds[(size_t)k + 1], sw, pp.table == 1, pw));
    if (pp.table == 2) SC_TRY(invert_rows(c, pp.bounds[(size_t)k], pp.bounds[(size_t)k + 1], sw));
    SC_HIP(hipEventRecord(swapped, sw));
    pp.enqueued = k + 1;
    return SC_OK;
}

static void pipe_drain(sc_ctx *c, PermPipe &pp)
{
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->stream4) (void)hipStreamSynchronize(c->stream4);
    if (c->stream_px) (void)hipStreamSynchronize(c->stream_px);
    if (c->stream_fr) (void)hipStreamSynchronize(c->stream_fr);
    for (hipStream_t sp : c->stream_pg)
        if (sp) (void)hipStreamSynchronize(sp);
    (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : pp.ev)
        if (e) (void)hipEventDestroy(e);
    pp.ev.clear();
}

// Begin: allocations, chunk schedule, the generator's set-up and its first `ahead` chunks (all of them when ahead
// >= the number of chunks).  Needs nothing but n and the generator state -- no graph, no expression.
static int pipe_begin(sc_ctx *c, const uint64_t *state6, int64_t n, int64_t n_perm, int table, PermPipe &pp, int64_t ahead,
                      const std::function<int()> &after_first_chunk = nullptr)
{
    SC_REQUIRE(state6, SC_ERR_INVALID, "permutation pipeline: null generator state");
    SC_REQUIRE(n_perm >= 1, SC_ERR_INVALID, "permutation pipeline: n_perm must be >= 1");
    SC_REQUIRE(table == 0 || permgen_can_swap_inverse(n) || table == 2, SC_ERR_STATE, "inverse-only tables need a longer permutation");
    SC_TRY(sc_perm_alloc(c, n, n_perm));
    // the resident table is being overwritten from here on: nothing may take it for valid until the job has been consumed
    // (sc_moran / sc_local_moran / sc_lee_shared with a resident table then fail with "holds 0 rows" instead of reading
    // rows the generator's streams are still writing)
    c->p_count = 0;
    c->inv_rows_valid = 0;
    c->perm_forward_valid = false;
    if (!c->stream2) {  // (SC_STREAM_PRIORITY=1: the r01 prioritised chain stream, for experiments; no gain measured in r02)
        int prio_lo = 0, prio_hi = 0;
        if (!getenv("SC_STREAM_PRIORITY") || hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess ||
            hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi) != hipSuccess) {
            (void)hipGetLastError();
            c->stream2 = nullptr;
            SC_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        }
    }
    if (!c->stream_px) SC_HIP(hipStreamCreateWithFlags(&c->stream_px, hipStreamNonBlocking));
    if (!c->stream3) SC_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    if (!c->stream4) SC_HIP(hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking));
    // allocations first (hipMalloc synchronises the device), then the streams run freely
    if (table >= 1) SC_TRY(c->inv.ensure(sizeof(int32_t) * (size_t)(c->p_stride * n_perm + 32), &c->mem));
    // chunk schedule: a short first chunk so that the consumer starts early, PERM_CHUNK each in the middle, a short
    // last chunk (the job ends with the swaps and the consumption of the last chunk after the scan is done)
    pp.bounds.clear();
    pp.bounds.push_back(0);
    if (n_perm > 3 * PERM_CHUNK) {
        const int64_t first = pipe_first_perms(), last = pipe_tail_total();
        const std::vector<int64_t> tail = pipe_tail_perms();
        const int64_t rest = (n_perm - last - first) % PERM_CHUNK;
        static const bool join = getenv("SC_PIPE_JOIN") != nullptr;   // development: a small remainder joins the first chunk (r02 / early r03)
        int64_t p = first + (join && rest < PERM_CHUNK / 2 ? rest : 0);
        pp.bounds.push_back(p);
        if (p == first && rest > 0) { p += rest; pp.bounds.push_back(p); }   // the remainder: a chunk of its own, second
        for (; p < n_perm - last; ) { p += PERM_CHUNK; pp.bounds.push_back(p); }
        for (int64_t v : tail) { p += v; pp.bounds.push_back(p); }
    } else {
        for (int64_t p = PERM_CHUNK; p < n_perm; p += PERM_CHUNK) pp.bounds.push_back(p);
        pp.bounds.push_back(n_perm);
    }
    const int64_t chunks = (int64_t)pp.bounds.size() - 1;
    // stream2: scan(0) scan(1) ...   stream3/4: swaps(k) (+ inverse(k)) after scan(k)   stream: score(k) after swaps(k)
    pp.ev.assign((size_t)chunks * 2, nullptr);
    pp.table = table; pp.n = n; pp.n_perm = n_perm; pp.enqueued = 0;
    for (int k = 0; k < 6; ++k) pp.state0[k] = state6[k];
    pp.job = PermJob();
    int rc = permgen_begin(c, state6, n, n_perm, &pp.job, c->stream2);
    for (int64_t k = 0; k < chunks && k < ahead && rc == SC_OK; ++k) {
        rc = pipe_generate(c, pp, k);
        if (k == 0 && rc == SC_OK && after_first_chunk) rc = after_first_chunk();
    }
    if (rc != SC_OK) pipe_drain(c, pp);
    return rc;
}

// Consume: `after_first` (the consumer's host-blocking set-up) runs once, then `score(p0, p1)` behind every chunk's
// swaps on the context stream, with the generator kept TWO chunks ahead in the host's enqueue order (r03 timeline: with
// one chunk ahead the chain sat idle for 5 ms behind the consumer's set-up; a chunk is some 250 API calls).
static int pipe_consume(sc_ctx *c, PermPipe &pp, uint64_t *state6, const std::function<int()> &after_first,
                        const std::function<int(int64_t, int64_t)> &score)
{
    const int64_t chunks = (int64_t)pp.bounds.size() - 1;
    int rc = SC_OK;
    if (after_first) rc = after_first();
    c->perm_bijective = true;  // device-generated rows are permutations by construction
    c->perm_forward_valid = pp.table != 1;
    for (int64_t k = 0; k < chunks && rc == SC_OK; ++k) {
        while (rc == SC_OK && pp.enqueued < chunks && pp.enqueued < k + 3) rc = pipe_generate(c, pp, pp.enqueued);
        if (rc == SC_OK && hipStreamWaitEvent(c->stream, pp.ev[(size_t)(2 * k + 1)], 0) != hipSuccess) {
            sc_set_error("permutation pipeline: event plumbing failed");
            rc = SC_ERR_HIP;
        }
        if (rc == SC_OK) rc = score(pp.bounds[(size_t)k], pp.bounds[(size_t)k + 1]);
    }
    pipe_drain(c, pp);
    if (rc != SC_OK) return rc;
    SC_TRY(permgen_finish(c, &pp.job, state6));
    c->p_count = pp.n_perm;
    c->inv_rows_valid = pp.table >= 1 ? pp.n_perm : 0;
    return SC_OK;
}

void sc_perm_pipe_abort(sc_ctx *c)
{
    if (!c->pipe) return;
    pipe_drain(c, *c->pipe);
    delete c->pipe;
    c->pipe = nullptr;
    c->p_count = 0;
}

int sc_perm_pipeline(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm, int table,
                     const std::function<int()> &after_first, const std::function<int(int64_t, int64_t)> &score)
{
    sc_perm_pipe_abort(c);   // (a job begun with sc_moran_seeded_begin and never finished)
    PermPipe pp;
    SC_TRY(pipe_begin(c, state6, n, n_perm, table, pp, 2));
    return pipe_consume(c, pp, state6, after_first, score);
}

static int moran_seeded_once(sc_ctx *c, uint64_t *state6, int64_t n_perm, double *I_out, double *sims_out,
                             int64_t *count_ge_out, double *sim_sum_out, double *sim_sumsq_out, PermPipe *begun)
{
    SC_REQUIRE(state6, SC_ERR_INVALID, "sc_moran_seeded: null state");
    SC_TRY(moran_check(c, n_perm, I_out));
    SC_REQUIRE(n_perm >= 1, SC_ERR_INVALID, "sc_moran_seeded: n_perm must be >= 1 (use sc_moran for n_perm = 0)");
    const int64_t n = c->e_n;
    SC_TRY(moran_alloc_sims(c, n_perm));
    // every scoring kernel gathers through the inverse table only (the inverse of a Fisher-Yates result is the same
    // transpositions in ascending order: no permutation rows, no scatter pass)
    const bool inverse_only = permgen_can_swap_inverse(n);
    int bits = 64;
    auto prepare = [&]() -> int {
        SC_TRY(moran_prepare(c, n_perm < PERM_CHUNK ? n_perm : PERM_CHUNK, true));
        bits = c->narrow_bits;
        // CUs left to the generator, by source width (r04, bench size, ms per step at 64 / 80 / 96 / 112 CUs left): the uint8
        // step is balanced between generator and scoring (SCORE_RESERVED_CUS); uint16 243.0 / 246.8 / 238.5 / 239.2; float32,
        // whose step is scoring-bound, 389.3 / 390.4 / 398.1 / 408.6
        if (c->score_leave_cus > 8 && !getenv("SC_SCORE_LEAVE_CUS")) c->score_leave_cus = bits == 8 ? SCORE_RESERVED_CUS : bits == 16 ? 96 : 64;
        return SC_OK;
    };
    auto score = [&](int64_t p0, int64_t p1) -> int {
        // The last chunk is scored after the generator has finished (its own swaps are the generator's last launches):
        // it takes the CUs the earlier launches left to the generator; the one before it runs beside the generator's
        // short last chunk only.  SC_SCORE_LEAVE_TAIL="a,b" (development): CUs left by the second-to-last / last
        // chunk's launch; measured at bench size (ms per step): 64,64 -> 211.5, 64,8 -> 209.9, 32,8 -> 207.1, 8,8 -> 207.9.
        const int keep = c->score_leave_cus;
        if (keep > 8) {
            int tail_prev = keep < 32 ? keep : 32, tail_last = 8;
            if (const char *v = getenv("SC_SCORE_LEAVE_TAIL")) sscanf(v, "%d,%d", &tail_prev, &tail_last);
            int64_t tail_from = pipe_tail_total();   // launches with fewer permutations than this still to come use tail_prev
            if (const char *v = getenv("SC_SCORE_TAIL_PERMS")) tail_from = atoi(v);   // development: sweep
            if (p1 == n_perm) c->score_leave_cus = tail_last;
            else if (n_perm > 3 * PERM_CHUNK && n_perm - p1 < tail_from) c->score_leave_cus = tail_prev;
        }
        const int rc = moran_perm_range(c, p0, p1, bits, false);
        c->score_leave_cus = keep;
        return rc;
    };
    if (begun) SC_TRY(pipe_consume(c, *begun, state6, prepare, score));   // the generator has been running since _begin
    else SC_TRY(sc_perm_pipeline(c, state6, n, n_perm, inverse_only ? 1 : 2, prepare, score));
    return moran_finish(c, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out);
}

// SCORE_RESERVED_CUS: compute units the persistent scoring kernel leaves EMPTY for the generator that runs beside it.
//
// The scoring kernel would fill every compute unit with wavefronts that live for milliseconds, and the generator is
// a chain of sub-millisecond launches (most of them 1024-thread workgroups that need a nearly empty compute unit)
// that must not queue behind them.  r01 kept them apart with CU-masked and prioritised streams; r02 does it by the
// scoring kernel's own shape -- a grid of (CUs - reserved) one-per-CU workgroups (see k_moran_score) on plain streams,
// each stream on its own hardware queue (GPU_MAX_HW_QUEUES, see sc_api.hip).  Measured on one box, 1M cells x 500
// genes x 1000 permutations, uint16 source, event-ordered generator: reserved 16 -> 1406 genes/s, 32 -> 1716, 48 -> no
// better; with the runtime's default of 4 shared hardware queues 1174 (the chain's launches queue behind 25-ms scoring
// launches).  With the uint8 source and the flag-ordered generator (one chain launch per chunk; its gate / publish
// launches need free wavefront slots at once): 32 -> 2050, 40 / 48 -> 2177, 64 -> 2327, 72 -> 2332, 80 -> 2302,
// 96 -> 2347, 128 -> 2173 (scoring 125 ms on 224 CUs, 157 on 192, 190 on 128: the step is balanced around 64-96).
// r03, workgroup-shared lag rows (scoring a third faster, the step generator-bound): the chain itself slows down when
// the scoring kernel has more of the chip -- chain busy 170 / 144 / 136 / 134 / 132 ms per step with 48 / 64 / 96 / 128 /
// 160 CUs reserved (memory-system interference: the chain is one workgroup of dependent loads) -- and the step is
// 198 / 179 / 175 / 180 / 196 ms: 96 (with a 48-permutation last chunk).
static int moran_seeded_streams(sc_ctx *c, uint64_t *state6, int64_t n_perm, double *I_out, double *sims_out,
                                int64_t *count_ge_out, double *sim_sum_out, double *sim_sumsq_out, PermPipe *begun = nullptr)
{
    const int ahead = c->pg_ahead;
    int leave = c && c->e_n > 0 && permgen_is_block_parallel(c, c->e_n) ? SCORE_RESERVED_CUS : 8;
    if (const char *v = getenv("SC_SCORE_LEAVE_CUS")) leave = atoi(v);  // development: sweep the reservation
    c->pg_ahead = PIPE_AHEAD;
    if (const char *v = getenv("SC_PIPE_AHEAD")) c->pg_ahead = atoi(v);  // development: sweep the lookahead
    c->score_leave_cus = leave;
    const int rc = moran_seeded_once(c, state6, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out, begun);
    c->score_leave_cus = 0;
    c->pg_ahead = ahead;
    return rc;
}

extern "C" int sc_moran_seeded(sc_ctx *c, uint64_t *state6, int64_t n_perm, double *I_out, double *sims_out,
                               int64_t *count_ge_out, double *sim_sum_out, double *sim_sumsq_out)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "sc_moran_seeded: null context");
    int rc = moran_seeded_streams(c, state6, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out);
    if (rc == SC_PERMGEN_RETRY) {  // the block-parallel scan failed its verification: nothing was returned yet
        const int mode = c->pg_mode;
        c->pg_mode = 1;
        rc = moran_seeded_streams(c, state6, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out);
        c->pg_mode = mode;
    }
    return rc;
}

// The same call in two halves, so that the generator -- the longest chain of the job, which needs nothing but n_cells and
// the seed's state -- runs while the caller is still building the graph and uploading the expression matrix
// (morans_i: kNN + the D2H of the neighbour lists for obsp + 40 ms of PCIe upload used to sit in front of it):
//   sc_moran_seeded_begin(ctx, state6, n_cells, n_perm)   enqueues the whole generator job and returns at once;
//   ... sc_knn_2d / sc_graph_* / sc_expr_set_* on the same context ...
//   sc_moran_seeded_finish(ctx, state6, outputs)          prepares the operands and scores chunk after chunk.
// Results and the final generator state are those of sc_moran_seeded.  A job that is begun and not finished is
// dropped by sc_moran_seeded_abort, by any call that replaces the permutation table, and with the context.
extern "C" int sc_moran_seeded_begin(sc_ctx *c, const uint64_t *state6, int64_t n_cells, int64_t n_perm, int64_t ahead_chunks)
{
    SC_REQUIRE(c && state6, SC_ERR_INVALID, "sc_moran_seeded_begin: null pointer");
    SC_REQUIRE(n_perm >= 1 && n_perm <= (1 << 24), SC_ERR_INVALID, "sc_moran_seeded_begin: n_perm=%lld out of range", (long long)n_perm);
    SC_HIP(hipSetDevice(c->device));
    sc_perm_pipe_abort(c);
    const int ahead = c->pg_ahead;
    c->pg_ahead = PIPE_AHEAD;
    if (const char *v = getenv("SC_PIPE_AHEAD")) c->pg_ahead = atoi(v);
    PermPipe *pp = new PermPipe;
    // ahead_chunks: chunks of the generator enqueued before returning (a chunk is ~250 launches, ~3.5 ms of host time);
    // 0 = all of them (a caller with tens of milliseconds of host-blocking work in front of _finish: an upload), else at
    // least 2 (_finish enqueues the rest, two ahead of the scoring)
    const int64_t ahead_n = ahead_chunks <= 0 ? (int64_t)1 << 40 : (ahead_chunks < 2 ? 2 : ahead_chunks);
    // (r04 measured again, and dropped again: the first half of the scoring's preparation enqueued right behind the
    // generator's FIRST chunk for callers whose operands are resident -- scoring starts ~9 ms earlier, the step is 3.5 ms
    // LONGER (165.4 vs 161.9 ms, same box): the full-chip column-sum kernels delay the generator's first units, and the
    // scoring then only waits longer for its first chunks.)
    const int rc = pipe_begin(c, state6, n_cells, n_perm, permgen_can_swap_inverse(n_cells) ? 1 : 2, *pp, ahead_n);
    c->pg_ahead = ahead;
    if (rc != SC_OK) { delete pp; return rc; }
    c->pipe = pp;
    return SC_OK;
}

extern "C" int sc_moran_seeded_abort(sc_ctx *c)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_HIP(hipSetDevice(c->device));
    sc_perm_pipe_abort(c);
    return SC_OK;
}

extern "C" int sc_moran_seeded_finish(sc_ctx *c, uint64_t *state6, double *I_out, double *sims_out, int64_t *count_ge_out,
                                      double *sim_sum_out, double *sim_sumsq_out)
{
    SC_REQUIRE(c && state6, SC_ERR_INVALID, "sc_moran_seeded_finish: null pointer");
    SC_REQUIRE(c->pipe, SC_ERR_STATE, "sc_moran_seeded_finish: no job begun (sc_moran_seeded_begin)");
    SC_HIP(hipSetDevice(c->device));
    PermPipe *pp = c->pipe;
    const int64_t n_perm = pp->n_perm;
    int rc = SC_OK;
    if (pp->n != c->e_n) {
        sc_set_error("sc_moran_seeded_finish: the job was begun for %lld cells, the expression has %lld", (long long)pp->n, (long long)c->e_n);
        rc = SC_ERR_INVALID;
    }
    if (rc == SC_OK) rc = moran_seeded_streams(c, state6, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out, pp);
    uint64_t state0[6];
    for (int k = 0; k < 6; ++k) state0[k] = pp->state0[k];
    c->pipe = nullptr;
    pipe_drain(c, *pp);       // (no-op after a completed consume; an error path may have left launches in flight)
    delete pp;
    if (rc == SC_PERMGEN_RETRY) {  // the block-parallel scan failed its verification: nothing was returned, rerun in one piece
        for (int k = 0; k < 6; ++k) state6[k] = state0[k];
        const int mode = c->pg_mode;
        c->pg_mode = 1;
        rc = moran_seeded_streams(c, state6, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out);
        c->pg_mode = mode;
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// A8: Lee's L
// ------------------------------------------------------------------------------------------------

// out[i] = T[tile(g)][i][slot(g)]  -- pull one gene out of the tiles into a contiguous vector
__global__ __launch_bounds__(256) void k_extract_col(const double *__restrict__ T, int64_t n, int64_t g,
                                                     double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = T[(g >> 4) * n * SC_TILE + i * SC_TILE + (g & 15)];
}

#define LEE_CELLS_PER_BLOCK 8192

// partial[p][blk] = sum_{j in block range} a[j] * b[perm_p[j]]   (p == n_perm: identity perm with a2)
__global__ __launch_bounds__(256) void k_vec_gather_dot(const double *__restrict__ a,
                                                        const double *__restrict__ b,
                                                        const int32_t *__restrict__ perm, int64_t pstride,
                                                        int64_t n, double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int32_t *prow = perm + (int64_t)blockIdx.y * pstride;
    int64_t j0 = (int64_t)blockIdx.x * LEE_CELLS_PER_BLOCK;
    int64_t j1 = j0 + LEE_CELLS_PER_BLOCK < n ? j0 + LEE_CELLS_PER_BLOCK : n;
    double acc = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) acc = fma(a[j], b[prow[j]], acc);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void k_vec_dot(const double *__restrict__ a, const double *__restrict__ b,
                                                 int64_t n, double *__restrict__ partial)
{
    __shared__ double sh[256];
    int64_t j0 = (int64_t)blockIdx.x * LEE_CELLS_PER_BLOCK;
    int64_t j1 = j0 + LEE_CELLS_PER_BLOCK < n ? j0 + LEE_CELLS_PER_BLOCK : n;
    double acc = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) acc = fma(a[j], b[j], acc);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// out[r] = sum_b partial[r][b]; one thread per row, ascending b
__global__ void k_row_sum(const double *__restrict__ partial, int rows, int blocks, double *__restrict__ out)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += partial[(int64_t)r * blocks + b];
    out[r] = s;
}

extern "C" int sc_lee(sc_ctx *c, const int32_t *pair_x, const int32_t *pair_y, const int64_t *perm_offset,
                      int64_t n_pairs, int64_t n_perm, double *L_out, int64_t *count_abs_ge_out,
                      double *L_perm_out)
{
    SC_REQUIRE(c && pair_x && pair_y && L_out, SC_ERR_INVALID, "sc_lee: null pointer");
    SC_REQUIRE(n_pairs >= 0 && n_perm >= 0, SC_ERR_INVALID, "sc_lee: negative size");
    SC_HIP(hipSetDevice(c->device));
    if (n_perm > 0) SC_TRY(sc_perm_forward_ensure(c));
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "sc_lee: no expression loaded");
    SC_REQUIRE(c->g_n == c->e_n, SC_ERR_STATE, "sc_lee: graph missing or size mismatch");
    SC_REQUIRE(n_perm == 0 || perm_offset, SC_ERR_INVALID, "sc_lee: perm_offset required when n_perm > 0");
    const int64_t n = c->e_n, T = c->e_tiles;
    const size_t tile_bytes = (size_t)n * SC_TILE * sizeof(double);
    for (int64_t q = 0; q < n_pairs; ++q) {
        SC_REQUIRE(pair_x[q] >= 0 && pair_x[q] < c->e_genes && pair_y[q] >= 0 && pair_y[q] < c->e_genes,
                   SC_ERR_INVALID, "sc_lee: pair %lld references a gene outside the loaded set", (long long)q);
        if (n_perm > 0 && perm_offset[q] >= 0)
            SC_REQUIRE(c->p_n == n && perm_offset[q] + n_perm <= c->p_count, SC_ERR_STATE,
                       "sc_lee: pair %lld needs permutation rows [%lld, %lld) but the table has %lld",
                       (long long)q, (long long)perm_offset[q], (long long)(perm_offset[q] + n_perm),
                       (long long)c->p_count);
    }
    // z-scores (population sd), lag = W z, u = W^T z
    SC_TRY(expr_center(c));
    hipLaunchKernelGGL(k_div_sd, dim3((unsigned)ceil_div64(n * SC_TILE, 256), (unsigned)T), dim3(256), 0, c->stream,
                       c->Z.as<double>(), c->g_var.as<double>(), n);
    SC_TRY(c->Lag.ensure((size_t)T * tile_bytes, &c->mem));
    SC_TRY(launch_lag(c, c->g_indptr, c->g_indices, c->g_data, c->Z.as<double>(), c->Lag.as<double>()));
    std::vector<double> var((size_t)c->e_genes);
    SC_HIP(hipMemcpyAsync(var.data(), c->g_var.p, sizeof(double) * var.size(), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));

    const int blocks = (int)ceil_div64(n, LEE_CELLS_PER_BLOCK);
    // vectors: a = z_x, la = (W z_y), u = W^T z_x (via transposed graph on the extracted column), b = z_y
    SC_TRY(c->lee_a.ensure(sizeof(double) * (size_t)n * 4, &c->mem));
    double *va = c->lee_a.as<double>(), *vlag = va + n, *vu = va + 2 * n, *vb = va + 3 * n;
    SC_TRY(c->lee_b.ensure(sizeof(double) * (size_t)blocks * (size_t)(n_perm + 1), &c->mem));
    SC_TRY(c->lee_out.ensure(sizeof(double) * (size_t)(n_perm + 1 > T * SC_TILE ? n_perm + 1 : T * SC_TILE),
                             &c->mem));
    if (n_perm > 0) SC_TRY(sc_graph_ensure_transpose(c));
    std::vector<double> host((size_t)n_perm + 1);
    for (int64_t q = 0; q < n_pairs; ++q) {
        bool degenerate = !(var[pair_x[q]] > 0.0) || !(var[pair_y[q]] > 0.0);
        if (degenerate) {
            L_out[q] = 0.0;
            if (count_abs_ge_out) count_abs_ge_out[q] = n_perm;
            if (L_perm_out)
                for (int64_t p = 0; p < n_perm; ++p) L_perm_out[q * n_perm + p] = 0.0;
            continue;
        }
        unsigned gcol = (unsigned)ceil_div64(n, 256);
        hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Z.as<double>(), n,
                           (int64_t)pair_x[q], va);
        hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Lag.as<double>(), n,
                           (int64_t)pair_y[q], vlag);
        hipLaunchKernelGGL(k_vec_dot, dim3(blocks), dim3(256), 0, c->stream, va, vlag, n,
                           c->lee_b.as<double>() + (size_t)n_perm * blocks);
        bool do_perm = n_perm > 0 && perm_offset[q] >= 0;
        if (do_perm) {
            hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Z.as<double>(), n,
                               (int64_t)pair_y[q], vb);
            // u = W^T z_x : SpMV with the transposed graph on a single contiguous vector
            sc_launch_spmv_vec(c, c->gt_indptr.as<int64_t>(), c->gt_indices.as<int32_t>(), c->gt_data.as<double>(),
                               va, vu, n);
            KernelTimerScope ts(c, SC_K_LEE_PERM);
            hipLaunchKernelGGL(k_vec_gather_dot, dim3(blocks, (unsigned)n_perm), dim3(256), 0, c->stream, vu, vb,
                               c->perm.as<int32_t>() + perm_offset[q] * c->p_stride, c->p_stride, n,
                               c->lee_b.as<double>());
        }
        int rows = do_perm ? (int)n_perm + 1 : 1;
        const double *src = c->lee_b.as<double>() + (do_perm ? 0 : (size_t)n_perm * blocks);
        double *dst = c->lee_out.as<double>() + (do_perm ? 0 : n_perm);
        hipLaunchKernelGGL(k_row_sum, dim3((unsigned)ceil_div64(rows, 256)), dim3(256), 0, c->stream, src, rows,
                           blocks, dst);
        SC_HIP(hipGetLastError());
        SC_HIP(hipMemcpyAsync(host.data() + (do_perm ? 0 : n_perm), dst, sizeof(double) * (size_t)rows,
                              hipMemcpyDeviceToHost, c->stream));
        SC_HIP(hipStreamSynchronize(c->stream));
        double L = host[(size_t)n_perm];
        L_out[q] = L;
        int64_t cnt = 0;
        if (do_perm)
            for (int64_t p = 0; p < n_perm; ++p) cnt += fabs(host[(size_t)p]) >= fabs(L) ? 1 : 0;
        if (count_abs_ge_out) count_abs_ge_out[q] = do_perm ? cnt : 0;
        if (L_perm_out)
            for (int64_t p = 0; p < n_perm; ++p) L_perm_out[q * n_perm + p] = do_perm ? host[(size_t)p] : 0.0;
    }
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// N1: Local Moran's I (AC:804-934) with the reference's float32 arithmetic
//
// The reference standardises in float32, takes lag = W32 @ Z32 with scipy's row-sequential float32
// accumulation, and for every permutation recomputes Zs = Z[perm], lag_s = W @ Zs, I_perm = Zs * lag_s
// into a (P, N, B) tensor before counting |I_perm| >= |I| per cell in a Python loop.  Here the count
// is accumulated on the fly: thread = (cell, 4 genes of a 16-gene float tile), loop over permutations.
// ------------------------------------------------------------------------------------------------

// Z32[tile][cell][16] = (float(x) - mean32) / sd32  (two float32 roundings, AC:858); padded genes -> 0
__global__ __launch_bounds__(256) void k_lm_standardize(const double *__restrict__ X, const float *__restrict__ mean32,
                                                        const float *__restrict__ sd32, float *__restrict__ Z32,
                                                        int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * SC_TILE) return;
    int64_t tile = blockIdx.y;
    int slot = (int)(t & 15);
    float x = (float)X[tile * n * SC_TILE + t];
    float sd = sd32[tile * SC_TILE + slot];
    // IEEE float division via double (innocuous double rounding for 24-bit operands)
    float z = (float)__ddiv_rn((double)__fsub_rn(x, mean32[tile * SC_TILE + slot]), (double)sd);
    Z32[tile * n * SC_TILE + t] = z;
}

// ---- numpy's float summation, reproduced ---------------------------------------------------------
// The reference takes the per-gene mean and E[x^2] with scipy's sparse `.mean(axis=0)` (AC:79-80,
// 102-107): (data * T(1/n)) summed per CSC column by np.add.reduceat, i.e. first stored entry +
// numpy's PAIRWISE sum of the rest (blocks of <= 128 with 8 strided accumulators, halving above
// that with the split rounded down to a multiple of 8), in the matrix dtype T.  On count data the
// per-cell |I_perm| >= |I| test is full of exact ties that are decided by the last bit of z, so the
// float32 mean and sd must be THE SAME floats; a more accurate sum is not good enough.
// The summation tree is fixed by the element count alone, so it is evaluated in parallel with the same rounding:
// (1) the stored (non-zero) values of every gene are compacted in cell order (k_npc_count / k_npc_offsets /
// k_npc_scatter: wavefront ballots over 512-cell blocks), (2) one thread per gene lists the leaves of numpy's
// recursion over elements 1.. (k_npc_leaves), (3) one thread per (gene, statistic, leaf) sums its <= 128 elements
// with the 8 strided accumulators (k_npc_leafsum), (4) one thread per (gene, statistic) replays the recursion over
// the leaf sums (k_npc_combine).  A sequential walk per gene took 1.15 s at 1M cells; this takes milliseconds.

#define NPC_CELLS 512  // cells per wavefront block of the compaction

// cnt[(tile * nblk + w) * 16 + g] = stored entries of gene slot g among the cells of block w
__global__ __launch_bounds__(256) void k_npc_count(const double *__restrict__ X, int64_t n, int64_t nblk,
                                                   uint32_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), tile = blockIdx.y;
    if (w >= nblk) return;
    const double *Xt = X + tile * n * SC_TILE;
    uint32_t mine = 0;
    for (int s = 0; s < NPC_CELLS / 64; ++s) {
        const int64_t cell = w * NPC_CELLS + 64 * s + lane;
        double v[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double2 t = cell < n ? reinterpret_cast<const double2 *>(Xt + cell * SC_TILE)[k] : make_double2(0.0, 0.0);
            v[2 * k] = t.x; v[2 * k + 1] = t.y;
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const uint32_t c = (uint32_t)__popcll(__ballot(v[g] != 0.0));
            mine += (lane == g) ? c : 0u;
        }
    }
    if (lane < 16) cnt[(tile * nblk + w) * 16 + lane] = mine;
}

// exclusive prefix over the blocks of one gene, in place (one thread per padded gene)
__global__ void k_npc_offsets(uint32_t *__restrict__ cnt, int64_t nblk, int64_t genes_padded)
{
    const int64_t gp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gp >= genes_padded) return;
    uint32_t *c = cnt + (gp >> 4) * nblk * 16 + (gp & 15);
    uint32_t run = 0;
    for (int64_t w = 0; w < nblk; ++w) {
        const uint32_t t = c[w * 16];
        c[w * 16] = run;
        run += t;
    }
}

// comp[gene * n + k] = k-th stored value of the gene, in cell order, as T
template <typename T>
__global__ __launch_bounds__(256) void k_npc_scatter(const double *__restrict__ X, int64_t n, int64_t nblk,
                                                     const uint32_t *__restrict__ off, T *__restrict__ comp)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), tile = blockIdx.y;
    if (w >= nblk) return;
    const double *Xt = X + tile * n * SC_TILE;
    uint32_t base[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) base[g] = off[(tile * nblk + w) * 16 + g];
    for (int s = 0; s < NPC_CELLS / 64; ++s) {
        const int64_t cell = w * NPC_CELLS + 64 * s + lane;
        double v[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double2 t = cell < n ? reinterpret_cast<const double2 *>(Xt + cell * SC_TILE)[k] : make_double2(0.0, 0.0);
            v[2 * k] = t.x; v[2 * k + 1] = t.y;
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const bool nz = v[g] != 0.0;
            const unsigned long long bal = __ballot(nz);
            // set bits of the ballot below this lane: the hardware's own mask-below-lane count (no per-lane 64-bit shift)
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (nz) comp[(tile * 16 + g) * n + base[g] + below] = (T)v[g];
            base[g] += (uint32_t)__popcll(bal);
        }
    }
}

// leaves[gene][i] = (start, len) of the i-th leaf of the recursion over elements 1 .. nnz-1; nleaves[gene]
__global__ void k_npc_leaves(const double *__restrict__ nnz, int64_t n_genes, int64_t max_leaves,
                             uint2 *__restrict__ leaves, uint32_t *__restrict__ nleaves)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_genes) return;
    const uint32_t cnt = (uint32_t)nnz[g];
    uint32_t k = 0;
    if (cnt >= 2) {
        uint2 *out = leaves + g * max_leaves;
        (void)pw_walk<float>(cnt - 1, [&](uint32_t start, uint32_t len) {
            if ((int64_t)k < max_leaves) out[k] = make_uint2(start, len);
            ++k;
            return 0.f;
        });
    }
    nleaves[g] = k;
}

// one leaf: numpy's unrolled block sum (8 strided accumulators, pairwise combine, then the tail) of
// val(i) = x_i * inv_n (statistic 0) or (x_i * x_i) * inv_n (statistic 1) over compacted elements 1 + start ..
template <typename T>
__global__ __launch_bounds__(256) void k_npc_leafsum(const T *__restrict__ comp, int64_t n,
                                                     const uint2 *__restrict__ leaves,
                                                     const uint32_t *__restrict__ nleaves, int64_t max_leaves,
                                                     T *__restrict__ leafsum)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = blockIdx.y;
    const int square = blockIdx.z;
    if (i >= (int64_t)nleaves[g] || i >= max_leaves) return;
    const uint2 lf = leaves[g * max_leaves + i];
    const T *a = comp + g * n + 1 + lf.x;
    const T inv_n = (T)(1.0 / (double)n);
    const uint32_t len = lf.y;
    auto val = [&](uint32_t k) { T x = a[k]; if (square) x = x * x; return x * inv_n; };
    T res;
    if (len < 8) {
        res = (T)(-0.0);
        for (uint32_t k = 0; k < len; ++k) res += val(k);
    } else {
        T r0 = val(0), r1 = val(1), r2 = val(2), r3 = val(3), r4 = val(4), r5 = val(5), r6 = val(6), r7 = val(7);
        uint32_t k = 8;
        for (; k < len - (len % 8); k += 8) {
            r0 += val(k); r1 += val(k + 1); r2 += val(k + 2); r3 += val(k + 3);
            r4 += val(k + 4); r5 += val(k + 5); r6 += val(k + 6); r7 += val(k + 7);
        }
        res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; k < len; ++k) res += val(k);
    }
    leafsum[(g * 2 + square) * max_leaves + i] = res;
}

// out[2*g] = numpy mean, out[2*g+1] = numpy mean of squares, as T: first stored entry + pairwise sum of the rest
template <typename T>
__global__ void k_npc_combine(const T *__restrict__ comp, int64_t n, const double *__restrict__ nnz,
                              const T *__restrict__ leafsum, int64_t max_leaves, int64_t n_genes,
                              T *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = t >> 1;
    if (g >= n_genes) return;
    const int square = (int)(t & 1);
    const uint32_t cnt = (uint32_t)nnz[g];
    const T inv_n = (T)(1.0 / (double)n);
    T res = (T)0;
    if (cnt >= 1) {
        T x = comp[g * n];
        if (square) x = x * x;
        res = x * inv_n;
        if (cnt >= 2) {
            const T *ls = leafsum + (g * 2 + square) * max_leaves;
            uint32_t k = 0;
            res = res + pw_walk<T>(cnt - 1, [&](uint32_t, uint32_t) { return ls[k++]; });
        }
    }
    out[t] = res;
}

// mean32 / sd32 exactly as AC:821-830: var = sqmean - mean^2 and sqrt in the matrix dtype T, then float32
template <typename T>
__global__ void k_lm_stats(const T *__restrict__ stats, float *__restrict__ mean32, float *__restrict__ sd32,
                           unsigned char *__restrict__ zero, int64_t n_genes, int64_t total)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    if (g >= n_genes) { mean32[g] = 0.f; sd32[g] = 1.f; zero[g] = 1; return; }
    const T m = stats[2 * g], q = stats[2 * g + 1];
    const T var = q - m * m;
    // sqrt in double, rounded once: correctly rounded for a float operand (53 >= 2*24 + 2 bits); the
    // hardware v_sqrt_f32 alone is a 1-ulp approximation
    const float sd = (float)__dsqrt_rn((double)var);
    const bool z = (sd == 0.0f);
    mean32[g] = (float)m;
    sd32[g] = z ? 1.0f : sd;
    zero[g] = z ? 1 : 0;
}

// observed: lag = W32 @ Z32 (row-sequential float32, mul and add rounded separately), I = Z * lag
__global__ __launch_bounds__(256) void k_lm_observed(const long long *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices,
                                                     const double *__restrict__ w, const float *__restrict__ Z32,
                                                     float *__restrict__ Lag32, float *__restrict__ I32, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t >> 2;
    int q = (int)(t & 3);
    if (i >= n) return;
    const float4 *Zt = reinterpret_cast<const float4 *>(Z32 + (int64_t)blockIdx.y * n * SC_TILE) + q;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long e = indptr[i]; e < indptr[i + 1]; ++e) {
        const float ww = (float)w[e];
        const float4 z = Zt[(int64_t)indices[e] * 4];
        s.x = __fadd_rn(s.x, __fmul_rn(ww, z.x)); s.y = __fadd_rn(s.y, __fmul_rn(ww, z.y));
        s.z = __fadd_rn(s.z, __fmul_rn(ww, z.z)); s.w = __fadd_rn(s.w, __fmul_rn(ww, z.w));
    }
    const float4 zi = Zt[i * 4];
    const int64_t o = (int64_t)blockIdx.y * n * 4 + i * 4 + q;
    reinterpret_cast<float4 *>(Lag32)[o] = s;
    reinterpret_cast<float4 *>(I32)[o] =
        make_float4(__fmul_rn(zi.x, s.x), __fmul_rn(zi.y, s.y), __fmul_rn(zi.z, s.z), __fmul_rn(zi.w, s.w));
}

// count[i][g] += #{p : |Z[perm_p[i]] * sum_e w_e Z[perm_p[col_e]]| >= |I[i]|}
__global__ __launch_bounds__(256) void k_lm_perm_count(const long long *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices,
                                                       const double *__restrict__ w, const float *__restrict__ Z32,
                                                       const float *__restrict__ I32,
                                                       const int32_t *__restrict__ perm, int64_t pstride,
                                                       int n_perm, int32_t *__restrict__ count, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t >> 2;
    int q = (int)(t & 3);
    if (i >= n) return;
    const float4 *Zt = reinterpret_cast<const float4 *>(Z32 + (int64_t)blockIdx.y * n * SC_TILE) + q;
    const int64_t o = (int64_t)blockIdx.y * n * 4 + i * 4 + q;
    const float4 obs = reinterpret_cast<const float4 *>(I32)[o];
    const float ax = fabsf(obs.x), ay = fabsf(obs.y), az = fabsf(obs.z), aw = fabsf(obs.w);
    const long long e0 = indptr[i], e1 = indptr[i + 1];
    int cx = 0, cy = 0, cz = 0, cw = 0;
    for (int p = 0; p < n_perm; ++p) {
        const int32_t *prow = perm + (int64_t)p * pstride;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long e = e0; e < e1; ++e) {
            const float ww = (float)w[e];
            const float4 z = Zt[(int64_t)prow[indices[e]] * 4];
            s.x = __fadd_rn(s.x, __fmul_rn(ww, z.x)); s.y = __fadd_rn(s.y, __fmul_rn(ww, z.y));
            s.z = __fadd_rn(s.z, __fmul_rn(ww, z.z)); s.w = __fadd_rn(s.w, __fmul_rn(ww, z.w));
        }
        const float4 zi = Zt[(int64_t)prow[i] * 4];
        cx += fabsf(__fmul_rn(zi.x, s.x)) >= ax; cy += fabsf(__fmul_rn(zi.y, s.y)) >= ay;
        cz += fabsf(__fmul_rn(zi.z, s.z)) >= az; cw += fabsf(__fmul_rn(zi.w, s.w)) >= aw;
    }
    reinterpret_cast<int4 *>(count)[o] = make_int4(cx, cy, cz, cw);
}

// ---- the same count in two phases per batch of permutations, in the graph's processing order (r02) ----
// k_lm_perm_count above reads, per permutation and cell, k + 1 permutation indices and k + 1 random 64-byte z rows
// per gene tile.  Per cell i the permuted vector y = z[perm] is all that matters: I_perm[i] = y[i] * sum_e w_e y[col_e].
// Phase A materialises y once per (permutation, tile) -- ONE random row per cell -- at the cell's position r in a
// spatially sorted order (Ys[r] = Z[perm[order[r]]]); phase B is then a LOCAL sparse product: the neighbours of a cell
// sit at nearby positions, their rows are served by L1 / L2.  The edges of a row keep their ascending-column order,
// so every sum is the reference's row-sequential float32 sum, bit for bit.
#define LM_PERM_BATCH 8

// Ys[p][tile][r][16] = Z32[tile][perm_p[order[r]]][16]      thread = (r, q), grid.y = tile, grid.z = permutation of the batch
__global__ __launch_bounds__(256) void k_lm_gather_sorted(const float *__restrict__ Z32, const int32_t *__restrict__ order,
                                                          const int32_t *__restrict__ perm, int64_t pstride, int64_t n,
                                                          int64_t tiles, float *__restrict__ Ys)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> 2;
    const int q = (int)(t & 3);
    if (r >= n) return;
    const int32_t src = perm[(int64_t)blockIdx.z * pstride + order[r]];
    const float4 v = reinterpret_cast<const float4 *>(Z32 + (int64_t)blockIdx.y * n * SC_TILE)[(int64_t)src * 4 + q];
    reinterpret_cast<float4 *>(Ys + ((int64_t)blockIdx.z * tiles + blockIdx.y) * n * SC_TILE)[r * 4 + q] = v;
}

// count[tile][cell][16] += #{p in batch : |y[r] * sum_e w_e y[rank(col_e)]| >= |I[cell]|},  cell = order[r]
// The edge loop is the OUTER loop and the batch's permutations the (unrolled) inner one: the LM_PERM_BATCH row loads of
// an edge are independent and in flight together (with the permutations outside, every row load waited for the
// previous one: 6.4 ms per launch at 2.9 TB/s of fabric traffic, latency-bound), and an edge's index and weight are
// read once per batch.  Per permutation the terms are still added in the row's edge order: the reference's sum.
// (An XCD-contiguous block order was measured too: 7.6 ms instead of 6.4 with the old loop order; not kept.)
__global__ __launch_bounds__(256) void k_lm_count_sorted(const long long *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices_r,
                                                         const float *__restrict__ w32, const int32_t *__restrict__ order,
                                                         const float *__restrict__ Ys, const float *__restrict__ I32,
                                                         int n_batch, int64_t tiles, int32_t *__restrict__ count, int64_t n,
                                                         int first)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> 2;
    const int q = (int)(t & 3);
    if (r >= n) return;
    const int64_t i = order[r];
    const int64_t o = (int64_t)blockIdx.y * n * 4 + i * 4 + q;
    const float4 obs = reinterpret_cast<const float4 *>(I32)[o];
    const float ax = fabsf(obs.x), ay = fabsf(obs.y), az = fabsf(obs.z), aw = fabsf(obs.w);
    const long long e0 = indptr[i], e1 = indptr[i + 1];
    const int64_t pstep = tiles * n * 4;   // float4 stride between the permutations of the batch
    const float4 *Y0 = reinterpret_cast<const float4 *>(Ys + (int64_t)blockIdx.y * n * SC_TILE) + q;
    float4 s[LM_PERM_BATCH];
#pragma unroll
    for (int p = 0; p < LM_PERM_BATCH; ++p) s[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long e = e0; e < e1; ++e) {
        const float ww = w32[e];
        const float4 *Ye = Y0 + (int64_t)indices_r[e] * 4;
#pragma unroll
        for (int p = 0; p < LM_PERM_BATCH; ++p) {
            if (p < n_batch) {
                const float4 z = Ye[p * pstep];
                s[p].x = __fadd_rn(s[p].x, __fmul_rn(ww, z.x)); s[p].y = __fadd_rn(s[p].y, __fmul_rn(ww, z.y));
                s[p].z = __fadd_rn(s[p].z, __fmul_rn(ww, z.z)); s[p].w = __fadd_rn(s[p].w, __fmul_rn(ww, z.w));
            }
        }
    }
    int cx = 0, cy = 0, cz = 0, cw = 0;
#pragma unroll
    for (int p = 0; p < LM_PERM_BATCH; ++p) {
        if (p < n_batch) {
            const float4 zi = Y0[r * 4 + p * pstep];
            cx += fabsf(__fmul_rn(zi.x, s[p].x)) >= ax; cy += fabsf(__fmul_rn(zi.y, s[p].y)) >= ay;
            cz += fabsf(__fmul_rn(zi.z, s[p].z)) >= az; cw += fabsf(__fmul_rn(zi.w, s[p].w)) >= aw;
        }
    }
    int4 *dst = reinterpret_cast<int4 *>(count) + o;
    if (first) *dst = make_int4(cx, cy, cz, cw);
    else { const int4 c0 = *dst; *dst = make_int4(c0.x + cx, c0.y + cy, c0.z + cz, c0.w + cw); }
}

// ---- the two phases over CODE rows (r03): count data, every value an integer in [0, LM_CODES) ----
// A gene with few distinct values has few distinct z: z = table[gene][value].  The permuted matrix of a batch is then
// moved around as the uint8 rows of the scoring kernel's narrow copy (128 genes per 128-byte row instead of 16 per
// 64-byte float tile row: an eighth of the gathered, written and re-read bytes), and the float32 z of a neighbour is
// looked up in LDS when it is used.  table[gene][v] is k_lm_standardize's own expression at x = v, so every product
// and every sum is the float path's, bit for bit.  With all weights equal (a row-normalised kNN graph) a second table
// holds w * z, the product the float path rounds before it adds.
#define LM_CODES 32          // values 0 .. 31
#define LM_TAB_STRIDE 36     // floats per table row: (q, value) pairs of one load land in different LDS banks for small values
#define LM_U8_QUAD 4         // permutations in flight per thread
#define LM_U8_BATCH_MAX 32   // permutations per launch (the counts are read and written once per launch)

// table rows in the order the kernel's threads use them: row = 8 b + q holds the gene of byte b of lane q's 16 bytes
// of a narrow row (k_pack_narrow<8>: tile 8 grp + b / 2, slot 2 q + b % 2);  tab[0] = z, tab[1] = w z
__global__ __launch_bounds__(256) void k_lm_ztab(const float *__restrict__ mean32, const float *__restrict__ sd32,
                                                 int64_t tiles16, float w, float *__restrict__ tab, int groups)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * 128 * LM_TAB_STRIDE) return;
    const int v = t % LM_TAB_STRIDE, row = (t / LM_TAB_STRIDE) % 128, grp = t / (LM_TAB_STRIDE * 128);
    const int b = row >> 3, q = row & 7;
    const int64_t tile = 8 * (int64_t)grp + (b >> 1);
    float z = 0.f;
    if (tile < tiles16 && v < LM_CODES) {
        const int64_t g = tile * SC_TILE + 2 * q + (b & 1);
        z = (float)__ddiv_rn((double)__fsub_rn((float)v, mean32[g]), (double)sd32[g]);
    }
    tab[t] = z;
    tab[(size_t)groups * 128 * LM_TAB_STRIDE + t] = __fmul_rn(w, z);
}

// Ys8[p][grp][r] = X8[grp][perm_p[order[r]]]   thread = (r, q), grid.y = group, grid.z = permutation of the batch
__global__ __launch_bounds__(256) void k_lm_gather_u8(const uint4 *__restrict__ X8, const int32_t *__restrict__ order,
                                                      const int32_t *__restrict__ perm, int64_t pstride, int64_t n,
                                                      int groups, uint4 *__restrict__ Ys8)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> 3;
    const int q = (int)(t & 7);
    if (r >= n) return;
    const int32_t src = perm[(int64_t)blockIdx.z * pstride + order[r]];
    Ys8[(((int64_t)blockIdx.z * groups + blockIdx.y) * n + r) * 8 + q] = X8[((int64_t)blockIdx.y * n + src) * 8 + q];
}

// count[tile][cell][16] += #{p in batch : |y[r] * sum_e w_e y[rank(col_e)]| >= |I[cell]|}, y = table[code], cell = order[r]
// thread = (r, q): the 16 genes of lane q's 16 bytes, LM_U8_QUAD permutations at a time; edges in the row's order.
template <bool UNI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(UNI ? 4 : 3, 4))) void k_lm_count_u8(const long long *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices_r, const float *__restrict__ w32,
                                                     const int32_t *__restrict__ order, const uint4 *__restrict__ Ys8,
                                                     const float *__restrict__ I32, const float *__restrict__ tab,
                                                     int n_batch, int64_t tiles, int groups, int32_t *__restrict__ count,
                                                     int64_t n, int first)
{
    __shared__ float tz[128 * LM_TAB_STRIDE];
    __shared__ float tw[UNI ? 128 * LM_TAB_STRIDE : 1];
    const int grp = blockIdx.y;
    for (int k = threadIdx.x; k < 128 * LM_TAB_STRIDE; k += 256) {
        tz[k] = tab[(size_t)grp * 128 * LM_TAB_STRIDE + k];
        if (UNI) tw[k] = tab[((size_t)groups + grp) * 128 * LM_TAB_STRIDE + k];
    }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> 3;
    const int q = (int)(t & 7);
    if (r >= n) return;
    const int64_t i = order[r];
    float a[16];
    uint32_t cnt[4] = {0u, 0u, 0u, 0u};   // 16 counts of <= LM_U8_BATCH_MAX, 8 bits each
    static_assert(LM_U8_BATCH_MAX < 256, "packed per-launch counts");
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int64_t tile = 8 * (int64_t)grp + (b >> 1);
        a[b] = tile < tiles ? fabsf(I32[tile * n * SC_TILE + i * SC_TILE + 2 * q + (b & 1)]) : 0.f;
    }
    const long long e0 = indptr[i], e1 = indptr[i + 1];
    const int64_t pstep = (int64_t)groups * n * 8;   // uint4 stride between the permutations of the batch
    const uint4 *Y0 = Ys8 + (int64_t)grp * n * 8 + q;
    const float *zq = tz + q * LM_TAB_STRIDE;         // + b * 8 * LM_TAB_STRIDE + value
    const float *wq = (UNI ? tw : tz) + q * LM_TAB_STRIDE;
    typedef float v2f __attribute__((ext_vector_type(2)));   // two genes per v_pk_add_f32 / v_pk_mul_f32: IEEE per component
    for (int p0 = 0; p0 < n_batch; p0 += LM_U8_QUAD) {
        v2f s[LM_U8_QUAD][8];
#pragma unroll
        for (int p = 0; p < LM_U8_QUAD; ++p)
#pragma unroll
            for (int b = 0; b < 8; ++b) s[p][b] = (v2f){0.f, 0.f};
        for (long long e = e0; e < e1; ++e) {
            const float ww = w32[e];
            const v2f ww2 = {ww, ww};
            const uint4 *Ye = Y0 + (int64_t)indices_r[e] * 8 + (int64_t)p0 * pstep;
            uint4 row[LM_U8_QUAD];
#pragma unroll
            for (int p = 0; p < LM_U8_QUAD; ++p) row[p] = p0 + p < n_batch ? Ye[p * pstep] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int p = 0; p < LM_U8_QUAD; ++p) {
                const uint32_t wd[4] = {row[p].x, row[p].y, row[p].z, row[p].w};
#pragma unroll
                for (int b = 0; b < 16; b += 2) {
                    const uint32_t v0 = (wd[b >> 2] >> (8 * (b & 3))) & 0xffu, v1 = (wd[b >> 2] >> (8 * (b & 3) + 8)) & 0xffu;
                    v2f term = {wq[b * 8 * LM_TAB_STRIDE + v0], wq[(b + 1) * 8 * LM_TAB_STRIDE + v1]};
                    if (!UNI) term = ww2 * term;          // (-ffp-contract=off: product and sum are rounded separately)
                    s[p][b >> 1] = s[p][b >> 1] + term;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < LM_U8_QUAD; ++p) {
            if (p0 + p < n_batch) {
                const uint4 own = Y0[r * 8 + (int64_t)(p0 + p) * pstep];
                const uint32_t wd[4] = {own.x, own.y, own.z, own.w};
#pragma unroll
                for (int b = 0; b < 16; b += 2) {
                    const uint32_t v0 = (wd[b >> 2] >> (8 * (b & 3))) & 0xffu, v1 = (wd[b >> 2] >> (8 * (b & 3) + 8)) & 0xffu;
                    const v2f zi = {zq[b * 8 * LM_TAB_STRIDE + v0], zq[(b + 1) * 8 * LM_TAB_STRIDE + v1]};
                    const v2f ip = zi * s[p][b >> 1];
                    cnt[b >> 2] += (fabsf(ip.x) >= a[b] ? 1u : 0u) << (8 * (b & 3));
                    cnt[b >> 2] += (fabsf(ip.y) >= a[b + 1] ? 1u : 0u) << (8 * (b & 3) + 8);
                }
            }
        }
    }
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
        const int64_t tile = 8 * (int64_t)grp + tt;
        if (tile >= tiles) continue;
        int2 *dst = reinterpret_cast<int2 *>(count + tile * n * SC_TILE + i * SC_TILE + 2 * q);
        const int ca = (int)((cnt[tt >> 1] >> (16 * (tt & 1))) & 0xffu), cb = (int)((cnt[tt >> 1] >> (16 * (tt & 1) + 8)) & 0xffu);
        if (first) *dst = make_int2(ca, cb);
        else { const int2 c0 = *dst; *dst = make_int2(c0.x + ca, c0.y + cb); }
    }
}

// tile layout [tile][cell][16] -> row-major [cell][n_genes]
template <typename T>
__global__ __launch_bounds__(256) void k_untile(const T *__restrict__ tiles, T *__restrict__ out, int64_t n,
                                                int64_t n_genes)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_genes) return;
    int64_t i = t / n_genes, g = t - i * n_genes;
    out[t] = tiles[(g >> 4) * n * SC_TILE + i * SC_TILE + (g & 15)];
}

// Is every loaded value an integer in [0, LM_CODES)?  (one pass over the tiles + one synchronisation; SC_LM_FLOAT_ROWS
// set: development switch, the float-row form for A/B runs and tests)
static int lm_codes_ok(sc_ctx *c, bool *ok)
{
    *ok = false;
    if (getenv("SC_LM_FLOAT_ROWS") || c->e_n >= ((int64_t)1 << 24)) return SC_OK;   // (16.7M cells: 2 GB of code rows per permutation and group)
    const int64_t n = c->e_n, T = c->e_tiles, G = c->e_genes;
    const int64_t Gpad = align_up64(T, 8) * SC_TILE;
    SC_TRY(c->g_flags.ensure(sizeof(uint32_t) * (size_t)Gpad, &c->mem));
    SC_TRY(c->g_xmax.ensure(sizeof(uint32_t) * (size_t)Gpad, &c->mem));
    SC_HIP(hipMemsetAsync(c->g_flags.p, 0, sizeof(uint32_t) * (size_t)Gpad, c->stream));
    SC_HIP(hipMemsetAsync(c->g_xmax.p, 0, sizeof(uint32_t) * (size_t)Gpad, c->stream));
    hipLaunchKernelGGL(k_gene_stats, dim3((unsigned)ceil_div64(n, RED_ROWS_PER_BLOCK), (unsigned)T), dim3(256), 0, c->stream,
                       c->X.as<double>(), n, c->g_flags.as<uint32_t>(), c->g_xmax.as<uint32_t>());
    SC_HIP(hipGetLastError());
    std::vector<uint32_t> flags((size_t)Gpad), xmax((size_t)Gpad);
    SC_HIP(hipMemcpyAsync(flags.data(), c->g_flags.p, sizeof(uint32_t) * (size_t)Gpad, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(xmax.data(), c->g_xmax.p, sizeof(uint32_t) * (size_t)Gpad, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    for (int64_t g = 0; g < G; ++g)
        if ((flags[(size_t)g] & 1u) || xmax[(size_t)g] >= LM_CODES) return SC_OK;
    *ok = true;
    return SC_OK;
}

// One local Moran job: the operands of the per-cell permutation counts (sc_local_moran, sc_local_moran_seeded)
struct LmJob {
    int64_t n = 0, G = 0, T = 0;
    size_t tile_f = 0;
    float *mean32 = nullptr, *sd32 = nullptr, *Z32 = nullptr, *I32 = nullptr, *Lag32 = nullptr;
    int32_t *cnt = nullptr;
    unsigned char *zero = nullptr;
    dim3 gc;
    int mode = 2;        // 0: one-kernel r01 form, 1: uint8 code rows, 2: float rows
    int groups = 0;      // code rows: 128-gene groups
    int64_t batch = 0;   // permutations per launch
    bool uni = false;
};

// statistics in numpy's order, z, observed lag and I; then the form of the permutation counts and its buffers
static int lm_prepare(sc_ctx *c, int64_t n_perm, LmJob &j)
{
    const int64_t n = c->e_n, G = c->e_genes, T = c->e_tiles;
    c->lm_valid = false;
    const size_t tile_f = (size_t)T * n * SC_TILE;
    // per-gene mean and E[x^2] with numpy's own summation order, in the matrix dtype (see k_npc_*)
    SC_TRY(colsum<OP_NZ>(c, c->X.as<double>(), nullptr, c->g_Inum.as<double>(), 1.0));
    SC_TRY(c->lee_out.ensure(sizeof(double) * 2 * (size_t)(T * SC_TILE), &c->mem));
    {
        const int64_t nblk = ceil_div64(n, NPC_CELLS), max_leaves = n / 32 + 64;
        const size_t tsz = c->e_dtype == SC_F32 ? sizeof(float) : sizeof(double);
        SC_TRY(c->np_cnt.ensure(sizeof(uint32_t) * (size_t)(T * nblk * 16), &c->mem));
        SC_TRY(c->np_comp.ensure(tsz * (size_t)(T * SC_TILE) * (size_t)n, &c->mem));
        SC_TRY(c->np_leaves.ensure(sizeof(uint2) * (size_t)G * (size_t)max_leaves + sizeof(uint32_t) * (size_t)G, &c->mem));
        SC_TRY(c->np_leafsum.ensure(tsz * 2 * (size_t)G * (size_t)max_leaves, &c->mem));
        uint2 *leaves = c->np_leaves.as<uint2>();
        uint32_t *nleaves = reinterpret_cast<uint32_t *>(leaves + (size_t)G * (size_t)max_leaves);
        const dim3 gw((unsigned)ceil_div64(nblk, 4), (unsigned)T);
        hipLaunchKernelGGL(k_npc_count, gw, dim3(256), 0, c->stream, c->X.as<double>(), n, nblk, c->np_cnt.as<uint32_t>());
        hipLaunchKernelGGL(k_npc_offsets, dim3((unsigned)ceil_div64(T * SC_TILE, 64)), dim3(64), 0, c->stream,
                           c->np_cnt.as<uint32_t>(), nblk, T * SC_TILE);
        hipLaunchKernelGGL(k_npc_leaves, dim3((unsigned)ceil_div64(G, 64)), dim3(64), 0, c->stream,
                           c->g_Inum.as<double>(), G, max_leaves, leaves, nleaves);
        const dim3 gl((unsigned)ceil_div64(max_leaves, 256), (unsigned)G, 2);
        if (c->e_dtype == SC_F32) {
            hipLaunchKernelGGL(k_npc_scatter<float>, gw, dim3(256), 0, c->stream, c->X.as<double>(), n, nblk,
                               c->np_cnt.as<uint32_t>(), c->np_comp.as<float>());
            hipLaunchKernelGGL(k_npc_leafsum<float>, gl, dim3(256), 0, c->stream, c->np_comp.as<float>(), n, leaves,
                               nleaves, max_leaves, c->np_leafsum.as<float>());
            hipLaunchKernelGGL(k_npc_combine<float>, dim3((unsigned)ceil_div64(2 * G, 64)), dim3(64), 0, c->stream,
                               c->np_comp.as<float>(), n, c->g_Inum.as<double>(), c->np_leafsum.as<float>(), max_leaves, G,
                               c->lee_out.as<float>());
        } else {
            hipLaunchKernelGGL(k_npc_scatter<double>, gw, dim3(256), 0, c->stream, c->X.as<double>(), n, nblk,
                               c->np_cnt.as<uint32_t>(), c->np_comp.as<double>());
            hipLaunchKernelGGL(k_npc_leafsum<double>, gl, dim3(256), 0, c->stream, c->np_comp.as<double>(), n, leaves,
                               nleaves, max_leaves, c->np_leafsum.as<double>());
            hipLaunchKernelGGL(k_npc_combine<double>, dim3((unsigned)ceil_div64(2 * G, 64)), dim3(64), 0, c->stream,
                               c->np_comp.as<double>(), n, c->g_Inum.as<double>(), c->np_leafsum.as<double>(), max_leaves,
                               G, c->lee_out.as<double>());
        }
        SC_HIP(hipGetLastError());
    }
    // float work buffers: [mean32 | sd32] in g_scale (as float), zero flags in counts, Z32/Lag32/I32 in Z/Lag
    SC_TRY(c->Z.ensure(tile_f * sizeof(double), &c->mem));    // Z32 (first half) + I32 (second half)
    SC_TRY(c->Lag.ensure(tile_f * sizeof(double), &c->mem));  // Lag32 (first half) + counts (second half)
    SC_TRY(c->counts.ensure((size_t)T * SC_TILE + 16, &c->mem));
    float *mean32 = c->g_scale.as<float>(), *sd32 = mean32 + T * SC_TILE;
    float *Z32 = c->Z.as<float>(), *I32 = Z32 + tile_f;
    float *Lag32 = c->Lag.as<float>();
    int32_t *cnt = reinterpret_cast<int32_t *>(Lag32 + tile_f);
    unsigned char *zero = c->counts.as<unsigned char>();
    if (c->e_dtype == SC_F32)
        hipLaunchKernelGGL(k_lm_stats<float>, dim3((unsigned)ceil_div64(T * SC_TILE, 256)), dim3(256), 0, c->stream,
                           c->lee_out.as<float>(), mean32, sd32, zero, G, T * SC_TILE);
    else
        hipLaunchKernelGGL(k_lm_stats<double>, dim3((unsigned)ceil_div64(T * SC_TILE, 256)), dim3(256), 0, c->stream,
                           c->lee_out.as<double>(), mean32, sd32, zero, G, T * SC_TILE);
    dim3 ge((unsigned)ceil_div64(n * SC_TILE, 256), (unsigned)T);
    hipLaunchKernelGGL(k_lm_standardize, ge, dim3(256), 0, c->stream, c->X.as<double>(), mean32, sd32, Z32, n);
    dim3 gc((unsigned)ceil_div64(n * 4, 256), (unsigned)T);
    hipLaunchKernelGGL(k_lm_observed, gc, dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                       c->g_indices.as<int32_t>(), c->g_data.as<double>(), Z32, Lag32, I32, n);
    SC_HIP(hipGetLastError());
    j.n = n; j.G = G; j.T = T; j.tile_f = tile_f;
    j.mean32 = mean32; j.sd32 = sd32; j.Z32 = Z32; j.I32 = I32; j.Lag32 = Lag32; j.cnt = cnt; j.zero = zero; j.gc = gc;
    if (n_perm <= 0) return SC_OK;
    if (c->lm_direct) { j.mode = 0; return SC_OK; }   // r01 form (development A/B, sc_ctx_set_local_moran_direct)
    SC_TRY(sc_graph_ensure_order(c));
    bool codes = false;
    SC_TRY(lm_codes_ok(c, &codes));
    if (codes) {
        // count data: the permuted matrix travels as uint8 code rows, z is looked up where it is used (k_lm_count_u8)
        j.mode = 1;
        j.groups = (int)ceil_div64(T, 8);
        const size_t row_bytes = (size_t)j.groups * (size_t)n * 128;
        SC_TRY(c->X32.ensure(sizeof(float) * (size_t)((T + 1) / 2) * n * 32, &c->mem));
        SC_TRY(c->lm_tab.ensure(sizeof(float) * 2 * (size_t)j.groups * 128 * LM_TAB_STRIDE, &c->mem));
        int64_t batch = (int64_t)(((size_t)4 << 30) / row_bytes) / LM_U8_QUAD * LM_U8_QUAD;
        batch = batch < LM_U8_QUAD ? LM_U8_QUAD : batch > LM_U8_BATCH_MAX ? LM_U8_BATCH_MAX : batch;
        if (batch > n_perm) batch = align_up64(n_perm, LM_U8_QUAD);
        j.batch = batch;
        SC_TRY(c->lm_ys.ensure(row_bytes * (size_t)batch, &c->mem));
        hipLaunchKernelGGL(k_pack_narrow<8>, dim3((unsigned)ceil_div64(n * 8, 256), (unsigned)j.groups), dim3(256), 0, c->stream,
                           c->X.as<double>(), c->X32.as<uint4>(), n, T);
        j.uni = c->g_uniform_w > 0.0;
        hipLaunchKernelGGL(k_lm_ztab, dim3((unsigned)ceil_div64((int64_t)j.groups * 128 * LM_TAB_STRIDE, 256)), dim3(256), 0, c->stream,
                           mean32, sd32, T, j.uni ? (float)c->g_uniform_w : 0.f, c->lm_tab.as<float>(), j.groups);
        SC_HIP(hipGetLastError());
    } else {
        j.mode = 2;
        j.batch = LM_PERM_BATCH;
        SC_TRY(c->lm_ys.ensure(sizeof(float) * (size_t)LM_PERM_BATCH * tile_f, &c->mem));
    }
    return SC_OK;
}

// counts of permutations [p0, p1) of the job (rows row0 + p of the forward table); p0 == 0 starts the counts
static int lm_count(sc_ctx *c, const LmJob &j, int64_t row0, int64_t p0, int64_t p1)
{
    const int64_t n = j.n, T = j.T;
    if (p1 <= p0) return SC_OK;
    KernelTimerScope ts(c, SC_K_LEE_PERM);
    if (j.mode == 0) {
        SC_REQUIRE(p0 == 0, SC_ERR_STATE, "internal: the one-kernel local Moran form counts all permutations at once");
        hipLaunchKernelGGL(k_lm_perm_count, j.gc, dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                           c->g_indices.as<int32_t>(), c->g_data.as<double>(), j.Z32, j.I32,
                           c->perm.as<int32_t>() + row0 * c->p_stride, c->p_stride, (int)(p1 - p0), j.cnt, n);
    } else if (j.mode == 1) {
        const dim3 g8((unsigned)ceil_div64(n * 8, 256), (unsigned)j.groups);
        auto count_u8 = j.uni ? k_lm_count_u8<true> : k_lm_count_u8<false>;
        for (int64_t p = p0; p < p1; p += j.batch) {
            const int nb = (int)(p1 - p < j.batch ? p1 - p : j.batch);
            hipLaunchKernelGGL(k_lm_gather_u8, dim3(g8.x, g8.y, (unsigned)nb), dim3(256), 0, c->stream, c->X32.as<uint4>(),
                               c->g_order.as<int32_t>(), c->perm.as<int32_t>() + (row0 + p) * c->p_stride, c->p_stride,
                               n, j.groups, c->lm_ys.as<uint4>());
            hipLaunchKernelGGL(count_u8, g8, dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                               c->g_indices_r.as<int32_t>(), c->g_w32.as<float>(), c->g_order.as<int32_t>(),
                               c->lm_ys.as<uint4>(), j.I32, c->lm_tab.as<float>(), nb, T, j.groups, j.cnt, n, p == 0 ? 1 : 0);
        }
    } else {
        for (int64_t p = p0; p < p1; p += j.batch) {
            const int nb = (int)(p1 - p < j.batch ? p1 - p : j.batch);
            hipLaunchKernelGGL(k_lm_gather_sorted, dim3(j.gc.x, (unsigned)T, (unsigned)nb), dim3(256), 0, c->stream, j.Z32,
                               c->g_order.as<int32_t>(), c->perm.as<int32_t>() + (row0 + p) * c->p_stride, c->p_stride,
                               n, T, c->lm_ys.as<float>());
            hipLaunchKernelGGL(k_lm_count_sorted, j.gc, dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                               c->g_indices_r.as<int32_t>(), c->g_w32.as<float>(), c->g_order.as<int32_t>(),
                               c->lm_ys.as<float>(), j.I32, nb, T, j.cnt, n, p == 0 ? 1 : 0);
        }
    }
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// un-tile into row-major (cells x genes) staging and copy back
static int lm_finish(sc_ctx *c, const LmJob &j, int64_t n_perm, float *z_out, float *lag_out, float *I_out,
                     int32_t *count_out, uint8_t *zero_var_out, bool arrays_done = false)
{
    const int64_t n = j.n, G = j.G;
    SC_TRY(c->lee_a.ensure(sizeof(float) * (size_t)n * (size_t)G, &c->mem));
    unsigned gu = (unsigned)ceil_div64(n * G, 256);
    struct { const float *src; float *dst; } outs[3] = {{j.Z32, z_out}, {j.Lag32, lag_out}, {j.I32, I_out}};
    for (auto &o : outs) {
        if (arrays_done) break;   // (a helper thread has copied them out beside the pipeline)
        hipLaunchKernelGGL(k_untile<float>, dim3(gu), dim3(256), 0, c->stream, o.src, c->lee_a.as<float>(), n, G);
        SC_HIP(hipMemcpyAsync(o.dst, c->lee_a.p, sizeof(float) * (size_t)n * (size_t)G, hipMemcpyDeviceToHost,
                              c->stream));
    }
    if (n_perm > 0 && count_out) {
        hipLaunchKernelGGL(k_untile<int32_t>, dim3(gu), dim3(256), 0, c->stream, j.cnt, c->lee_a.as<int32_t>(), n, G);
        SC_HIP(hipMemcpyAsync(count_out, c->lee_a.p, sizeof(int32_t) * (size_t)n * (size_t)G, hipMemcpyDeviceToHost,
                              c->stream));
    }
    if (zero_var_out)
        SC_HIP(hipMemcpyAsync(zero_var_out, j.zero, (size_t)G, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(c->stream));
    c->lm_valid = true;  // z / lag / counts stay resident for sc_local_moran_hist / sc_local_moran_classify
    c->lm_perms = n_perm;
    return SC_OK;
}

extern "C" int sc_local_moran(sc_ctx *c, int64_t n_perm, int64_t perm_row0, float *z_out, float *lag_out,
                              float *I_out, int32_t *count_out, uint8_t *zero_var_out)
{
    SC_REQUIRE(c && z_out && lag_out && I_out, SC_ERR_INVALID, "sc_local_moran: null pointer");
    SC_REQUIRE(n_perm >= 0 && perm_row0 >= 0, SC_ERR_INVALID, "sc_local_moran: negative size");
    SC_HIP(hipSetDevice(c->device));
    if (n_perm > 0) SC_TRY(sc_perm_forward_ensure(c));
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "sc_local_moran: no expression loaded");
    SC_REQUIRE(c->g_n == c->e_n, SC_ERR_STATE, "sc_local_moran: graph missing or size mismatch");
    c->lm_valid = false;
    if (n_perm > 0) {
        SC_REQUIRE(c->p_n == c->e_n && perm_row0 + n_perm <= c->p_count, SC_ERR_STATE,
                   "sc_local_moran: needs permutation rows [%lld, %lld) of length %lld", (long long)perm_row0,
                   (long long)(perm_row0 + n_perm), (long long)c->e_n);
    }
    LmJob j;
    SC_TRY(lm_prepare(c, n_perm, j));
    SC_TRY(lm_count(c, j, perm_row0, 0, n_perm));
    return lm_finish(c, j, n_perm, z_out, lag_out, I_out, count_out, zero_var_out);
}

// The same with the permutations drawn here: n_perm numpy-exact permutations of the cells from state6 (as
// sc_perm_generate would draw them; state6 is advanced the same way, the table stays resident), generated chunk
// by chunk while the per-cell counts of the finished chunks are taken -- the generator's chain is the longest part of
// a local Moran call, and the counts hide behind it.  Same outputs as sc_perm_generate + sc_local_moran.
extern "C" int sc_local_moran_seeded(sc_ctx *c, uint64_t *state6, int64_t n_perm, float *z_out, float *lag_out,
                                     float *I_out, int32_t *count_out, uint8_t *zero_var_out)
{
    SC_REQUIRE(c && state6 && z_out && lag_out && I_out, SC_ERR_INVALID, "sc_local_moran_seeded: null pointer");
    SC_REQUIRE(n_perm >= 1 && n_perm <= (1 << 24), SC_ERR_INVALID, "sc_local_moran_seeded: n_perm=%lld out of range", (long long)n_perm);
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "sc_local_moran_seeded: no expression loaded");
    SC_REQUIRE(c->g_n == c->e_n, SC_ERR_STATE, "sc_local_moran_seeded: graph missing or size mismatch");
    c->lm_valid = false;
    LmJob j;
    // r04: z, lag and I are final once the preparation has run -- three (cells x genes) float arrays, 1.2 GB at 10^6 cells x
    // 100 genes, that r03 copied to the caller's (pageable) arrays AFTER the last count, 0.1 s of a 0.5-s call.  A helper
    // thread un-tiles and copies them out on a stream of its own while the generator and the counts run (neither uses
    // the PCIe link); this thread keeps enqueuing the pipeline.
    std::thread copier;
    int copier_rc = SC_OK;
    bool copier_started = false;
    static const bool copy_beside = !getenv("SC_LM_COPY_LATE");   // (A/B)
    auto prepare = [&]() -> int {
        SC_TRY(lm_prepare(c, n_perm, j));
        if (!copy_beside || copier_started) return SC_OK;
        if (!c->stream_out) SC_HIP(hipStreamCreateWithFlags(&c->stream_out, hipStreamNonBlocking));
        SC_TRY(c->lm_out.ensure(sizeof(float) * (size_t)j.n * (size_t)j.G, &c->mem));
        hipEvent_t ready;
        SC_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        SC_HIP(hipEventRecord(ready, c->stream));
        SC_HIP(hipStreamWaitEvent(c->stream_out, ready, 0));
        SC_HIP(hipEventDestroy(ready));
        const LmJob jj = j;
        try {   // (no thread to be had: the arrays are copied at the end, as in r03)
            copier = std::thread([c, jj, z_out, lag_out, I_out, &copier_rc]() {
            if (hipSetDevice(c->device) != hipSuccess) { copier_rc = SC_ERR_HIP; return; }
            const unsigned gu = (unsigned)ceil_div64(jj.n * jj.G, 256);
            const struct { const float *src; float *dst; } outs[3] = {{jj.Z32, z_out}, {jj.Lag32, lag_out}, {jj.I32, I_out}};
            for (const auto &o : outs) {
                hipLaunchKernelGGL(k_untile<float>, dim3(gu), dim3(256), 0, c->stream_out, o.src, c->lm_out.as<float>(), jj.n, jj.G);
                if (hipMemcpyAsync(o.dst, c->lm_out.p, sizeof(float) * (size_t)jj.n * (size_t)jj.G, hipMemcpyDeviceToHost, c->stream_out) != hipSuccess ||
                    hipStreamSynchronize(c->stream_out) != hipSuccess) {   // (the staging buffer is reused by the next array)
                    copier_rc = SC_ERR_HIP;
                    return;
                }
            }
            });
            copier_started = true;
        } catch (...) {
            copier_started = false;
        }
        return SC_OK;
    };
    auto count = [&](int64_t p0, int64_t p1) -> int {
        if (j.mode == 0) return p1 == n_perm ? lm_count(c, j, 0, 0, n_perm) : SC_OK;   // (the one-kernel form: all rows at the end)
        return lm_count(c, j, 0, p0, p1);
    };
    const int ahead = c->pg_ahead;
    c->pg_ahead = 2;
    int rc = sc_perm_pipeline(c, state6, c->e_n, n_perm, 0, prepare, count);
    if (rc == SC_PERMGEN_RETRY) {   // the block-parallel scan failed its verification: the counts restart at permutation 0
        if (copier.joinable()) copier.join();   // (the second preparation rewrites what it reads -- with the same values)
        const int mode = c->pg_mode;
        c->pg_mode = 1;
        rc = sc_perm_pipeline(c, state6, c->e_n, n_perm, 0, prepare, count);
        c->pg_mode = mode;
    }
    c->pg_ahead = ahead;
    if (copier.joinable()) copier.join();
    SC_TRY(rc);
    if (copier_started && copier_rc != SC_OK) {
        sc_set_error("sc_local_moran_seeded: the copy of z / lag / I to the host failed");
        return copier_rc;
    }
    return lm_finish(c, j, n_perm, z_out, lag_out, I_out, count_out, zero_var_out, copier_started);
}

// hist[gene][c] = cells of the gene with permutation count c (LDS-private per workgroup while 16 genes' worth fits)
#define LMH_LDS 12288
__global__ __launch_bounds__(256) void k_lm_hist(const int32_t *__restrict__ cnt, int64_t n, int P1,
                                                 unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t h[LMH_LDS];
    const int64_t tile = blockIdx.y;
    const int32_t *ct = cnt + tile * n * SC_TILE;
    unsigned long long *ht = hist + tile * SC_TILE * P1;
    const bool priv = SC_TILE * P1 <= LMH_LDS;
    if (priv) {
        for (int k = threadIdx.x; k < SC_TILE * P1; k += 256) h[k] = 0;
        __syncthreads();
    }
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n * SC_TILE; t += (int64_t)gridDim.x * 256) {
        int c = ct[t];
        c = c < 0 ? 0 : (c >= P1 ? P1 - 1 : c);
        const int slot = (int)(t & 15);
        if (priv) atomicAdd(&h[slot * P1 + c], 1u);
        else atomicAdd(&ht[slot * P1 + c], 1ull);
    }
    if (priv) {
        __syncthreads();
        for (int k = threadIdx.x; k < SC_TILE * P1; k += 256)
            if (h[k]) atomicAdd(&ht[k], (unsigned long long)h[k]);
    }
}

extern "C" int sc_local_moran_hist(sc_ctx *c, int64_t *hist_out)
{
    SC_REQUIRE(c && hist_out, SC_ERR_INVALID, "sc_local_moran_hist: null pointer");
    SC_REQUIRE(c->lm_valid && c->lm_perms > 0, SC_ERR_STATE, "sc_local_moran_hist: no sc_local_moran result with permutations");
    SC_HIP(hipSetDevice(c->device));
    const int64_t n = c->e_n, G = c->e_genes, T = c->e_tiles;
    const int P1 = (int)c->lm_perms + 1;
    const size_t tile_f = (size_t)T * n * SC_TILE;
    const int32_t *cnt = reinterpret_cast<const int32_t *>(c->Lag.as<float>() + tile_f);
    SC_TRY(c->lee_b.ensure(sizeof(unsigned long long) * (size_t)(T * SC_TILE) * (size_t)P1, &c->mem));
    SC_HIP(hipMemsetAsync(c->lee_b.p, 0, sizeof(unsigned long long) * (size_t)(T * SC_TILE) * (size_t)P1, c->stream));
    hipLaunchKernelGGL(k_lm_hist, dim3(256, (unsigned)T), dim3(256), 0, c->stream, cnt, n, P1,
                       c->lee_b.as<unsigned long long>());
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(hist_out, c->lee_b.p, sizeof(int64_t) * (size_t)G * (size_t)P1, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// p = p_tab[g][count], p_adj = padj_tab[g][count], LISA quadrant (AC:219-265): 1 HH, 2 LL, 3 HL, 4 LH from the signs of
// z and lag, 0 where p_adj >= alpha or the gene is flagged; row-major outputs
__global__ __launch_bounds__(256) void k_lm_classify(const float *__restrict__ Z32, const float *__restrict__ Lag32,
                                                     const int32_t *__restrict__ cnt, int64_t n, int64_t G, int P1,
                                                     const float *__restrict__ p_tab, const float *__restrict__ padj_tab,
                                                     const unsigned char *__restrict__ force_ns, float alpha,
                                                     float *__restrict__ p_out, float *__restrict__ padj_out,
                                                     signed char *__restrict__ q_out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * G) return;
    const int64_t i = t / G, g = t - i * G;
    const int64_t src = (g >> 4) * n * SC_TILE + i * SC_TILE + (g & 15);
    const float z = Z32[src], lag = Lag32[src];
    signed char q = 0;
    if (z > 0.f && lag > 0.f) q = 1;
    if (z < 0.f && lag < 0.f) q = 2;
    if (z > 0.f && lag < 0.f) q = 3;
    if (z < 0.f && lag > 0.f) q = 4;
    if (P1 > 0) {
        int c = cnt[src];
        c = c < 0 ? 0 : (c >= P1 ? P1 - 1 : c);
        const float pa = padj_tab[g * P1 + c];
        p_out[t] = p_tab[g * P1 + c];
        padj_out[t] = pa;
        if (pa >= alpha) q = 0;
    }
    if (force_ns[g]) q = 0;
    q_out[t] = q;
}

extern "C" int sc_local_moran_classify(sc_ctx *c, const float *p_tab, const float *padj_tab, const uint8_t *force_ns,
                                       float alpha, float *p_out, float *padj_out, int8_t *quadrant_out)
{
    SC_REQUIRE(c && force_ns && quadrant_out, SC_ERR_INVALID, "sc_local_moran_classify: null pointer");
    SC_REQUIRE(c->lm_valid, SC_ERR_STATE, "sc_local_moran_classify: no sc_local_moran result");
    SC_HIP(hipSetDevice(c->device));
    const int64_t n = c->e_n, G = c->e_genes, T = c->e_tiles;
    const int P1 = c->lm_perms > 0 ? (int)c->lm_perms + 1 : 0;
    if (P1 > 0) SC_REQUIRE(p_tab && padj_tab && p_out && padj_out, SC_ERR_INVALID, "sc_local_moran_classify: tables and outputs required with permutations");
    const size_t tile_f = (size_t)T * n * SC_TILE, cells = (size_t)n * (size_t)G;
    const float *Z32 = c->Z.as<float>(), *Lag32 = c->Lag.as<float>();
    const int32_t *cnt = reinterpret_cast<const int32_t *>(Lag32 + tile_f);
    // device staging: [p | p_adj | quadrant] row-major, tables, flags
    SC_TRY(c->lee_a.ensure(sizeof(float) * 2 * cells + cells + 64, &c->mem));
    SC_TRY(c->lee_b.ensure(sizeof(float) * 2 * (size_t)G * (size_t)(P1 > 0 ? P1 : 1) + (size_t)G + 64, &c->mem));
    float *d_p = c->lee_a.as<float>(), *d_pa = d_p + cells;
    signed char *d_q = reinterpret_cast<signed char *>(d_pa + cells);
    float *d_pt = c->lee_b.as<float>(), *d_at = d_pt + (size_t)G * (size_t)(P1 > 0 ? P1 : 1);
    unsigned char *d_f = reinterpret_cast<unsigned char *>(d_at + (size_t)G * (size_t)(P1 > 0 ? P1 : 1));
    if (P1 > 0) {
        SC_HIP(hipMemcpyAsync(d_pt, p_tab, sizeof(float) * (size_t)G * P1, hipMemcpyHostToDevice, c->stream));
        SC_HIP(hipMemcpyAsync(d_at, padj_tab, sizeof(float) * (size_t)G * P1, hipMemcpyHostToDevice, c->stream));
    }
    SC_HIP(hipMemcpyAsync(d_f, force_ns, (size_t)G, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_lm_classify, dim3((unsigned)ceil_div64(n * G, 256)), dim3(256), 0, c->stream, Z32, Lag32, cnt, n,
                       G, P1, d_pt, d_at, d_f, alpha, d_p, d_pa, d_q);
    SC_HIP(hipGetLastError());
    if (P1 > 0) {
        SC_HIP(hipMemcpyAsync(p_out, d_p, sizeof(float) * cells, hipMemcpyDeviceToHost, c->stream));
        SC_HIP(hipMemcpyAsync(padj_out, d_pa, sizeof(float) * cells, hipMemcpyDeviceToHost, c->stream));
    }
    SC_HIP(hipMemcpyAsync(quadrant_out, d_q, cells, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// N2: Local Lee's L for one pair (AC:1394-1413): z-scores, lag = W z_y, L_local = z_x * lag, and the
// optional per-cell permutation count  #{p : |float32(z_x[i] * (W z_y[perm_p])[i])| >= |L_local[i]|}
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_lee_local_count(const long long *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices,
                                                         const double *__restrict__ w, const double *__restrict__ zx,
                                                         const double *__restrict__ zy,
                                                         const double *__restrict__ Llocal,
                                                         const int32_t *__restrict__ perm, int64_t pstride,
                                                         int n_perm, int32_t *__restrict__ count, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long e0 = indptr[i], e1 = indptr[i + 1];
    const double x = zx[i], obs = fabs(Llocal[i]);
    int cnt = 0;
    for (int p = 0; p < n_perm; ++p) {
        const int32_t *prow = perm + (int64_t)p * pstride;
        double s = 0.0;
        for (long long e = e0; e < e1; ++e) s = __dadd_rn(s, __dmul_rn(w[e], zy[prow[indices[e]]]));
        // the reference stores the permuted values in a float32 array before comparing (AC:1402,1408)
        const double lp = (double)(float)__dmul_rn(x, s);
        cnt += fabs(lp) >= obs;
    }
    count[i] = cnt;
}

__global__ __launch_bounds__(256) void k_vec_mul(const double *__restrict__ a, const double *__restrict__ b,
                                                 double *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __dmul_rn(a[i], b[i]);
}

// The same count in two phases per batch of permutations, in the graph's processing order (see k_lm_gather_sorted):
// ys[p][r] = z_y[perm_p[order[r]]] once per permutation, then a LOCAL sparse product.  (k_lee_local_count above
// fetched 900 GB for 999 permutations of 1M cells: 7 random 8-byte reads per cell and permutation, 128 bytes each.)
#define LL_PERM_BATCH 16

__global__ __launch_bounds__(256) void k_lee_local_gather(const double *__restrict__ zy, const int32_t *__restrict__ order,
                                                          const int32_t *__restrict__ perm, int64_t pstride, int64_t n,
                                                          double *__restrict__ ys)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    ys[(int64_t)blockIdx.y * n + r] = zy[perm[(int64_t)blockIdx.y * pstride + order[r]]];
}

__global__ __launch_bounds__(256) void k_lee_local_count_sorted(const long long *__restrict__ indptr,
                                                                const int32_t *__restrict__ indices_r,
                                                                const double *__restrict__ w, const int32_t *__restrict__ order,
                                                                const double *__restrict__ zx,
                                                                const double *__restrict__ ys,
                                                                const double *__restrict__ Llocal, int n_batch,
                                                                int32_t *__restrict__ count, int64_t n, int first)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t i = order[r];
    const long long e0 = indptr[i], e1 = indptr[i + 1];
    const double x = zx[i], obs = fabs(Llocal[i]);
    double s[LL_PERM_BATCH];
#pragma unroll
    for (int p = 0; p < LL_PERM_BATCH; ++p) s[p] = 0.0;
    for (long long e = e0; e < e1; ++e) {      // edge loop outside, permutations unrolled inside: independent loads in flight
        const double ww = w[e];
        const double *ye = ys + indices_r[e];
#pragma unroll
        for (int p = 0; p < LL_PERM_BATCH; ++p)
            if (p < n_batch) s[p] = __dadd_rn(s[p], __dmul_rn(ww, ye[(int64_t)p * n]));
    }
    int cnt = 0;
#pragma unroll
    for (int p = 0; p < LL_PERM_BATCH; ++p)
        if (p < n_batch) {
            // the reference stores the permuted values in a float32 array before comparing (AC:1402,1408)
            const double lp = (double)(float)__dmul_rn(x, s[p]);
            cnt += fabs(lp) >= obs;
        }
    count[i] = first ? cnt : count[i] + cnt;
}

extern "C" int sc_lee_local(sc_ctx *c, int32_t gene_x, int32_t gene_y, int64_t n_perm, int64_t perm_row0,
                            double *zx_out, double *lag_out, double *L_local_out, int32_t *count_out)
{
    SC_REQUIRE(c && zx_out && lag_out && L_local_out, SC_ERR_INVALID, "sc_lee_local: null pointer");
    SC_HIP(hipSetDevice(c->device));
    if (n_perm > 0) SC_TRY(sc_perm_forward_ensure(c));
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "sc_lee_local: no expression loaded");
    SC_REQUIRE(c->g_n == c->e_n, SC_ERR_STATE, "sc_lee_local: graph missing or size mismatch");
    SC_REQUIRE(gene_x >= 0 && gene_x < c->e_genes && gene_y >= 0 && gene_y < c->e_genes, SC_ERR_INVALID,
               "sc_lee_local: gene index outside the loaded set");
    if (n_perm > 0) {
        SC_REQUIRE(count_out, SC_ERR_INVALID, "sc_lee_local: count_out required when n_perm > 0");
        SC_REQUIRE(c->p_n == c->e_n && perm_row0 >= 0 && perm_row0 + n_perm <= c->p_count, SC_ERR_STATE,
                   "sc_lee_local: needs permutation rows [%lld, %lld)", (long long)perm_row0,
                   (long long)(perm_row0 + n_perm));
    }
    const int64_t n = c->e_n, T = c->e_tiles;
    SC_TRY(expr_center(c));
    hipLaunchKernelGGL(k_div_sd, dim3((unsigned)ceil_div64(n * SC_TILE, 256), (unsigned)T), dim3(256), 0, c->stream,
                       c->Z.as<double>(), c->g_var.as<double>(), n);
    SC_TRY(c->lee_a.ensure(sizeof(double) * (size_t)n * 5, &c->mem));
    double *vx = c->lee_a.as<double>(), *vy = vx + n, *vlag = vx + 2 * n, *vL = vx + 3 * n;
    int32_t *vcnt = reinterpret_cast<int32_t *>(vx + 4 * n);
    unsigned gcol = (unsigned)ceil_div64(n, 256);
    hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Z.as<double>(), n, (int64_t)gene_x, vx);
    hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Z.as<double>(), n, (int64_t)gene_y, vy);
    sc_launch_spmv_vec(c, c->g_indptr.as<int64_t>(), c->g_indices.as<int32_t>(), c->g_data.as<double>(), vy, vlag, n);
    hipLaunchKernelGGL(k_vec_mul, dim3(gcol), dim3(256), 0, c->stream, vx, vlag, vL, n);
    if (n_perm > 0 && c->lm_direct) {   // r01 form (development A/B)
        KernelTimerScope ts(c, SC_K_LEE_PERM);
        hipLaunchKernelGGL(k_lee_local_count, dim3(gcol), dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                           c->g_indices.as<int32_t>(), c->g_data.as<double>(), vx, vy, vL,
                           c->perm.as<int32_t>() + perm_row0 * c->p_stride, c->p_stride, (int)n_perm, vcnt, n);
    } else if (n_perm > 0) {
        SC_TRY(sc_graph_ensure_order(c));
        SC_TRY(c->lm_ys.ensure(sizeof(double) * (size_t)LL_PERM_BATCH * (size_t)n, &c->mem));
        KernelTimerScope ts(c, SC_K_LEE_PERM);
        for (int64_t p0 = 0; p0 < n_perm; p0 += LL_PERM_BATCH) {
            const int nb = (int)(n_perm - p0 < LL_PERM_BATCH ? n_perm - p0 : LL_PERM_BATCH);
            hipLaunchKernelGGL(k_lee_local_gather, dim3(gcol, (unsigned)nb), dim3(256), 0, c->stream, vy,
                               c->g_order.as<int32_t>(), c->perm.as<int32_t>() + (perm_row0 + p0) * c->p_stride, c->p_stride,
                               n, c->lm_ys.as<double>());
            hipLaunchKernelGGL(k_lee_local_count_sorted, dim3(gcol), dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                               c->g_indices_r.as<int32_t>(), c->g_data.as<double>(), c->g_order.as<int32_t>(), vx,
                               c->lm_ys.as<double>(), vL, nb, vcnt, n, p0 == 0 ? 1 : 0);
        }
    }
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(zx_out, vx, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(lag_out, vlag, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(L_local_out, vL, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (n_perm > 0)
        SC_HIP(hipMemcpyAsync(count_out, vcnt, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// r04: the pair body of lees_l_local as ONE pipeline (r03's verdict: a generator call and two device calls per pair, each
// waiting for the one before).  Equal to
//     sc_perm_generate(state6, n, n_perm_global + n_perm_local);  sc_lee(x, y, offset 0, n_perm_global);
//     sc_lee_local(x, y, n_perm_local, perm_row0 = n_perm_global)
// -- the same kernels on the same rows, results bit for bit, the generator state advanced by the same draws -- with the
// permuted sums of the global statistic and the per-cell counts taken chunk by chunk behind the generator (which is 85 % of
// the three calls' time at 10^6 cells), like sc_local_moran_seeded.
extern "C" int sc_lee_local_seeded(sc_ctx *c, uint64_t *state6, int32_t gene_x, int32_t gene_y, int64_t n_perm_global,
                                   int64_t n_perm_local, double *L_out, int64_t *count_abs_ge_out, double *zx_out,
                                   double *lag_out, double *L_local_out, int32_t *count_out)
{
    SC_REQUIRE(c && state6 && L_out && zx_out && lag_out && L_local_out, SC_ERR_INVALID, "sc_lee_local_seeded: null pointer");
    SC_REQUIRE(n_perm_global >= 0 && n_perm_local >= 0 && n_perm_global + n_perm_local >= 1 &&
               n_perm_global + n_perm_local <= (1 << 24), SC_ERR_INVALID, "sc_lee_local_seeded: permutation counts out of range");
    SC_REQUIRE(n_perm_local == 0 || count_out, SC_ERR_INVALID, "sc_lee_local_seeded: count_out required when n_perm_local > 0");
    SC_REQUIRE(n_perm_global == 0 || count_abs_ge_out, SC_ERR_INVALID, "sc_lee_local_seeded: count_abs_ge_out required when n_perm_global > 0");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "sc_lee_local_seeded: no expression loaded");
    SC_REQUIRE(c->g_n == c->e_n, SC_ERR_STATE, "sc_lee_local_seeded: graph missing or size mismatch");
    SC_REQUIRE(gene_x >= 0 && gene_x < c->e_genes && gene_y >= 0 && gene_y < c->e_genes, SC_ERR_INVALID,
               "sc_lee_local_seeded: gene index outside the loaded set");
    const int64_t n = c->e_n, T = c->e_tiles, Pg = n_perm_global, Pl = n_perm_local;
    const int blocks = (int)ceil_div64(n, LEE_CELLS_PER_BLOCK);
    const unsigned gcol = (unsigned)ceil_div64(n, 256);
    double *va = nullptr, *vlag_g = nullptr, *vu = nullptr, *vb = nullptr, *vlag = nullptr, *vL = nullptr;
    int32_t *vcnt = nullptr;
    auto prepare = [&]() -> int {
        // ---- sc_lee's operands: z-scores (population sd), Lag = W Z, u = W^T z_x ----
        SC_TRY(expr_center(c));
        hipLaunchKernelGGL(k_div_sd, dim3((unsigned)ceil_div64(n * SC_TILE, 256), (unsigned)T), dim3(256), 0, c->stream,
                           c->Z.as<double>(), c->g_var.as<double>(), n);
        SC_TRY(c->Lag.ensure((size_t)T * (size_t)n * SC_TILE * sizeof(double), &c->mem));
        SC_TRY(launch_lag(c, c->g_indptr, c->g_indices, c->g_data, c->Z.as<double>(), c->Lag.as<double>()));
        double var[2];
        SC_HIP(hipMemcpyAsync(&var[0], c->g_var.as<double>() + gene_x, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        SC_HIP(hipMemcpyAsync(&var[1], c->g_var.as<double>() + gene_y, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        SC_HIP(hipStreamSynchronize(c->stream));
        SC_REQUIRE(var[0] > 0.0 && var[1] > 0.0, SC_ERR_INVALID, "sc_lee_local_seeded: a gene of the pair has zero variance");
        SC_TRY(c->lee_a.ensure(sizeof(double) * (size_t)n * 8, &c->mem));
        va = c->lee_a.as<double>(); vlag_g = va + n; vu = va + 2 * n; vb = va + 3 * n;   // va = z_x, vb = z_y: sc_lee_local's vx, vy too
        vlag = va + 4 * n; vL = va + 5 * n;
        vcnt = reinterpret_cast<int32_t *>(va + 6 * n);
        SC_TRY(c->lee_b.ensure(sizeof(double) * (size_t)blocks * (size_t)(Pg + 1), &c->mem));
        SC_TRY(c->lee_out.ensure(sizeof(double) * (size_t)(Pg + 1 > T * SC_TILE ? Pg + 1 : T * SC_TILE), &c->mem));
        if (Pg > 0) SC_TRY(sc_graph_ensure_transpose(c));
        hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Z.as<double>(), n, (int64_t)gene_x, va);
        hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Lag.as<double>(), n, (int64_t)gene_y, vlag_g);
        hipLaunchKernelGGL(k_vec_dot, dim3(blocks), dim3(256), 0, c->stream, va, vlag_g, n, c->lee_b.as<double>() + (size_t)Pg * blocks);
        hipLaunchKernelGGL(k_extract_col, dim3(gcol), dim3(256), 0, c->stream, c->Z.as<double>(), n, (int64_t)gene_y, vb);
        if (Pg > 0)
            sc_launch_spmv_vec(c, c->gt_indptr.as<int64_t>(), c->gt_indices.as<int32_t>(), c->gt_data.as<double>(), va, vu, n);
        // ---- sc_lee_local's: lag = W z_y on the vector, L_local = z_x * lag ----
        sc_launch_spmv_vec(c, c->g_indptr.as<int64_t>(), c->g_indices.as<int32_t>(), c->g_data.as<double>(), vb, vlag, n);
        hipLaunchKernelGGL(k_vec_mul, dim3(gcol), dim3(256), 0, c->stream, va, vlag, vL, n);
        if (Pl > 0 && !c->lm_direct) {
            SC_TRY(sc_graph_ensure_order(c));
            SC_TRY(c->lm_ys.ensure(sizeof(double) * (size_t)LL_PERM_BATCH * (size_t)n, &c->mem));
        }
        SC_HIP(hipGetLastError());
        return SC_OK;
    };
    auto score = [&](int64_t p0, int64_t p1) -> int {
        KernelTimerScope ts(c, SC_K_LEE_PERM);
        const int64_t a1 = p1 < Pg ? p1 : Pg;
        if (p0 < a1)   // rows of the global statistic
            hipLaunchKernelGGL(k_vec_gather_dot, dim3(blocks, (unsigned)(a1 - p0)), dim3(256), 0, c->stream, vu, vb,
                               c->perm.as<int32_t>() + p0 * c->p_stride, c->p_stride, n, c->lee_b.as<double>() + (size_t)p0 * blocks);
        const int64_t b0 = p0 > Pg ? p0 : Pg;
        if (b0 < p1 && c->lm_direct) {   // r01 form (development A/B): all rows at the end
            if (p1 == Pg + Pl)
                hipLaunchKernelGGL(k_lee_local_count, dim3(gcol), dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                                   c->g_indices.as<int32_t>(), c->g_data.as<double>(), va, vb, vL,
                                   c->perm.as<int32_t>() + Pg * c->p_stride, c->p_stride, (int)Pl, vcnt, n);
        } else {
            for (int64_t q0 = b0; q0 < p1; q0 += LL_PERM_BATCH) {   // rows of the per-cell counts
                const int nb = (int)(p1 - q0 < LL_PERM_BATCH ? p1 - q0 : LL_PERM_BATCH);
                hipLaunchKernelGGL(k_lee_local_gather, dim3(gcol, (unsigned)nb), dim3(256), 0, c->stream, vb,
                                   c->g_order.as<int32_t>(), c->perm.as<int32_t>() + q0 * c->p_stride, c->p_stride, n, c->lm_ys.as<double>());
                hipLaunchKernelGGL(k_lee_local_count_sorted, dim3(gcol), dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                                   c->g_indices_r.as<int32_t>(), c->g_data.as<double>(), c->g_order.as<int32_t>(), va,
                                   c->lm_ys.as<double>(), vL, nb, vcnt, n, q0 == Pg ? 1 : 0);
            }
        }
        SC_HIP(hipGetLastError());
        return SC_OK;
    };
    const int ahead = c->pg_ahead;
    c->pg_ahead = 2;
    int rc = sc_perm_pipeline(c, state6, n, Pg + Pl, 0, prepare, score);
    if (rc == SC_PERMGEN_RETRY) {   // the block-parallel scan failed its verification: everything restarts at permutation 0
        const int mode = c->pg_mode;
        c->pg_mode = 1;
        rc = sc_perm_pipeline(c, state6, n, Pg + Pl, 0, prepare, score);
        c->pg_mode = mode;
    }
    c->pg_ahead = ahead;
    SC_TRY(rc);
    std::vector<double> host((size_t)Pg + 1);
    hipLaunchKernelGGL(k_row_sum, dim3((unsigned)ceil_div64(Pg + 1, 256)), dim3(256), 0, c->stream, c->lee_b.as<double>(),
                       (int)(Pg + 1), blocks, c->lee_out.as<double>());
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(host.data(), c->lee_out.p, sizeof(double) * (size_t)(Pg + 1), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(zx_out, va, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(lag_out, vlag, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(L_local_out, vL, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (Pl > 0) SC_HIP(hipMemcpyAsync(count_out, vcnt, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    const double L = host[(size_t)Pg];
    *L_out = L;
    if (count_abs_ge_out) {
        int64_t cnt = 0;
        for (int64_t p = 0; p < Pg; ++p) cnt += fabs(host[(size_t)p]) >= fabs(L) ? 1 : 0;
        *count_abs_ge_out = cnt;
    }
    return SC_OK;
}
