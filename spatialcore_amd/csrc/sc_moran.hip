// Expression tiles, spatial lag, global Moran's I with permutations, Lee's L.  gfx950 only.
//
// Device layout (DESIGN.md "Data layout"): genes are grouped in tiles of SC_TILE = 16; a tile is
// [cell][16] fp64, i.e. one 128-byte row per cell.  A permutation step `lag[perm[i]]` then gathers
// one full cache line that serves 16 genes at once, and the contiguous operand z[i] is a coalesced
// 128-byte row.  The permutation table is [perm][cell] int32 with a row stride that is a multiple
// of 32 elements so that rows can be read as int4.
#include <math.h>

#include <vector>

#include "sc_ctx.h"

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
    size_t gb = (size_t)tiles * SC_TILE * sizeof(double);
    SC_TRY(c->g_mean.ensure(gb, &c->mem));
    SC_TRY(c->g_var.ensure(gb, &c->mem));
    SC_TRY(c->g_z2.ensure(gb, &c->mem));
    SC_TRY(c->g_scale.ensure(gb, &c->mem));
    SC_TRY(c->g_Inum.ensure(gb, &c->mem));
    c->e_n = n;
    c->e_genes = n_genes;
    c->e_tiles = tiles;
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

enum { OP_ID = 0, OP_SQ = 1, OP_MUL = 2 };

#define RED_ROWS_PER_BLOCK 4096

// partial[tile][chunk][16] = sum over the chunk's rows of op(A[row][slot], B[row][slot])
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
    int64_t r0 = (int64_t)blockIdx.x * RED_ROWS_PER_BLOCK;
    int64_t r1 = r0 + RED_ROWS_PER_BLOCK < n ? r0 + RED_ROWS_PER_BLOCK : n;
    double acc = 0.0;
    for (int64_t r = r0 + rg; r < r1; r += 16) {
        double v = a[r * SC_TILE + slot];
        if (OP == OP_SQ) v = v * v;
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

// out[tile*16+slot] = (sum over chunks, ascending) * mul
__global__ void k_colsum_final(const double *__restrict__ partial, double *__restrict__ out, int chunks,
                               double mul)
{
    int tile = blockIdx.x, slot = threadIdx.x;
    double s = 0.0;
    for (int ch = 0; ch < chunks; ++ch) s += partial[((int64_t)tile * chunks + ch) * SC_TILE + slot];
    out[tile * SC_TILE + slot] = s * mul;
}

template <int OP>
static int colsum(sc_ctx *c, const double *A, const double *B, double *out, double mul)
{
    int64_t n = c->e_n;
    int chunks = (int)ceil_div64(n, RED_ROWS_PER_BLOCK);
    SC_TRY(c->red_tmp.ensure(sizeof(double) * (size_t)c->e_tiles * chunks * SC_TILE, &c->mem));
    hipLaunchKernelGGL(k_colsum_partial<OP>, dim3(chunks, (unsigned)c->e_tiles), dim3(256), 0, c->stream, A, B,
                       c->red_tmp.as<double>(), n);
    hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)c->e_tiles), dim3(SC_TILE), 0, c->stream,
                       c->red_tmp.as<double>(), out, chunks, mul);
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

// mean, Z = X - mean, z2 = sum Z^2, var = z2 / n
static int expr_center(sc_ctx *c)
{
    SC_REQUIRE(c->e_n > 0, SC_ERR_STATE, "no expression loaded (call sc_expr_set_* first)");
    int64_t n = c->e_n;
    SC_TRY(c->Z.ensure((size_t)c->e_tiles * n * SC_TILE * sizeof(double), &c->mem));
    SC_TRY(colsum<OP_ID>(c, c->X.as<double>(), nullptr, c->g_mean.as<double>(), 1.0 / (double)n));
    dim3 grid((unsigned)ceil_div64(n * SC_TILE, 256), (unsigned)c->e_tiles);
    hipLaunchKernelGGL(k_center, grid, dim3(256), 0, c->stream, c->X.as<double>(), c->g_mean.as<double>(),
                       c->Z.as<double>(), n);
    SC_TRY(colsum<OP_SQ>(c, c->Z.as<double>(), nullptr, c->g_z2.as<double>(), 1.0));
    SC_TRY(colsum<OP_SQ>(c, c->Z.as<double>(), nullptr, c->g_var.as<double>(), 1.0 / (double)n));
    SC_HIP(hipGetLastError());
    return SC_OK;
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

__global__ __launch_bounds__(256) void k_lag(const int64_t *__restrict__ indptr,
                                             const int32_t *__restrict__ indices,
                                             const double *__restrict__ w, const double *__restrict__ Z,
                                             double *__restrict__ Lag, int64_t n)
{
    // 8 threads per cell, each owning 2 of the tile's 16 genes (one 16-byte slice of the row)
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t >> 3;
    int q = (int)(t & 7);
    if (i >= n) return;
    const double2 *Zt = reinterpret_cast<const double2 *>(Z + (int64_t)blockIdx.y * n * SC_TILE);
    double2 *Lt = reinterpret_cast<double2 *>(Lag + (int64_t)blockIdx.y * n * SC_TILE);
    int64_t e0 = indptr[i], e1 = indptr[i + 1];
    double sx = 0.0, sy = 0.0;
    for (int64_t e = e0; e < e1; ++e) {
        int32_t j = indices[e];
        double ww = w[e];
        double2 z = Zt[(int64_t)j * 8 + q];
        sx = __dadd_rn(sx, __dmul_rn(ww, z.x));
        sy = __dadd_rn(sy, __dmul_rn(ww, z.y));
    }
    Lt[i * 8 + q] = make_double2(sx, sy);
}

static int launch_lag(sc_ctx *c, const DBuf &indptr, const DBuf &indices, const DBuf &data, const double *Z,
                      double *out)
{
    int64_t n = c->e_n;
    KernelTimerScope ts(c, SC_K_LAG);
    hipLaunchKernelGGL(k_lag, dim3((unsigned)ceil_div64(n * 8, 256), (unsigned)c->e_tiles), dim3(256), 0,
                       c->stream, indptr.as<int64_t>(), indices.as<int32_t>(), data.as<double>(), Z, out, n);
    SC_HIP(hipGetLastError());
    return SC_OK;
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

// sims[p0 + p][g0 + slot] = scale[slot] * sum_s partial[s][p][slot]   (ascending s)
__global__ __launch_bounds__(256) void k_moran_finalize(const double *__restrict__ partial,
                                                        const double *__restrict__ scale,
                                                        double *__restrict__ sims, int n_perm, int splits,
                                                        int64_t n_genes, int64_t g0, int64_t p0)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int p = t >> 4, slot = t & 15;
    if (p >= n_perm || g0 + slot >= n_genes) return;
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += partial[((int64_t)k * n_perm + p) * SC_TILE + slot];
    sims[(p0 + p) * n_genes + g0 + slot] = scale[slot] * s;
}

// per gene: I = scale * Inum; scale = n / s0 / z2
__global__ void k_moran_scale(const double *__restrict__ z2, const double *__restrict__ inum,
                              double *__restrict__ scale, double *__restrict__ I, double n_over_s0, int64_t total)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    double sc = n_over_s0 / z2[g];
    scale[g] = sc;
    I[g] = sc * inum[g];
}

// per gene: count(sims >= I), sum sims, sum sims^2 over permutations (block per gene, fixed tree)
__global__ __launch_bounds__(256) void k_moran_count(const double *__restrict__ sims,
                                                     const double *__restrict__ I, int n_perm,
                                                     int64_t n_genes, long long *__restrict__ count,
                                                     double *__restrict__ ssum, double *__restrict__ ssq)
{
    __shared__ double sh_a[256], sh_b[256];
    __shared__ int sh_c[256];
    int64_t g = blockIdx.x;
    double obs = I[g];
    double a = 0.0, b = 0.0;
    int cnt = 0;
    for (int p = threadIdx.x; p < n_perm; p += 256) {
        double v = sims[(int64_t)p * n_genes + g];
        cnt += (v >= obs) ? 1 : 0;
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

// z = x - mean, lag = W z, I = n/s0 * sum z*lag / sum z^2 (device), sims buffer sized for n_perm
static int moran_prepare(sc_ctx *c, int64_t n_perm)
{
    const int64_t n = c->e_n, T = c->e_tiles;
    SC_TRY(sc_graph_ensure_s0(c));
    SC_TRY(expr_center(c));
    SC_TRY(c->Lag.ensure((size_t)T * n * SC_TILE * sizeof(double), &c->mem));
    SC_TRY(launch_lag(c, c->g_indptr, c->g_indices, c->g_data, c->Z.as<double>(), c->Lag.as<double>()));
    SC_TRY(colsum<OP_MUL>(c, c->Z.as<double>(), c->Lag.as<double>(), c->g_Inum.as<double>(), 1.0));
    SC_TRY(c->sims.ensure(sizeof(double) * (size_t)(T * SC_TILE) * (size_t)(n_perm > 0 ? n_perm : 1), &c->mem));
    SC_TRY(c->g_I.ensure(sizeof(double) * (size_t)T * SC_TILE, &c->mem));
    hipLaunchKernelGGL(k_moran_scale, dim3((unsigned)ceil_div64(T * SC_TILE, 256)), dim3(256), 0, c->stream,
                       c->g_z2.as<double>(), c->g_Inum.as<double>(), c->g_scale.as<double>(), c->g_I.as<double>(),
                       (double)n / c->s0, T * SC_TILE);
    SC_HIP(hipGetLastError());
    if (n_perm > 0) {
        // partial sums for one chunk of permutations (<= PERM_CHUNK, or all of them in the unfused call)
        int64_t cps = 0;
        int splits = pick_splits(n, 1, &cps);  // upper bound on the split count
        SC_TRY(c->partial.ensure(sizeof(double) * (size_t)splits * (size_t)n_perm * SC_TILE, &c->mem));
    }
    return SC_OK;
}

// score permutations [p0, p1) of the active table for every gene tile (on the context stream)
static int moran_perm_range(sc_ctx *c, int64_t p0, int64_t p1)
{
    const int64_t n = c->e_n, G = c->e_genes, T = c->e_tiles;
    const size_t tile_elems = (size_t)n * SC_TILE;
    const int cnt = (int)(p1 - p0);
    if (cnt <= 0) return SC_OK;
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
                           c->stream, c->partial.as<double>(), c->g_scale.as<double>() + t * SC_TILE,
                           c->sims.as<double>(), cnt, splits, G, t * SC_TILE, p0);
    }
    SC_HIP(hipGetLastError());
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
                           c->g_I.as<double>(), (int)n_perm, G, c->counts.as<long long>(), c->sim_sum.as<double>(),
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
    SC_TRY(moran_prepare(c, n_perm < PERM_CHUNK ? n_perm : PERM_CHUNK));
    SC_TRY(c->sims.ensure(sizeof(double) * (size_t)(c->e_tiles * SC_TILE) * (size_t)(n_perm > 0 ? n_perm : 1),
                          &c->mem));
    for (int64_t p0 = 0; p0 < n_perm; p0 += PERM_CHUNK)
        SC_TRY(moran_perm_range(c, p0, p0 + PERM_CHUNK < n_perm ? p0 + PERM_CHUNK : n_perm));
    return moran_finish(c, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out);
}

extern "C" int sc_moran_seeded(sc_ctx *c, uint64_t *state6, int64_t n_perm, double *I_out, double *sims_out,
                               int64_t *count_ge_out, double *sim_sum_out, double *sim_sumsq_out)
{
    SC_REQUIRE(state6, SC_ERR_INVALID, "sc_moran_seeded: null state");
    SC_TRY(moran_check(c, n_perm, I_out));
    SC_REQUIRE(n_perm >= 1, SC_ERR_INVALID, "sc_moran_seeded: n_perm must be >= 1 (use sc_moran for n_perm = 0)");
    const int64_t n = c->e_n;
    SC_TRY(sc_perm_alloc(c, n, n_perm));
    if (!c->stream2) SC_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    if (!c->stream3) SC_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    // allocations first (hipMalloc synchronises the device), then the two streams run freely
    SC_TRY(c->sims.ensure(sizeof(double) * (size_t)(c->e_tiles * SC_TILE) * (size_t)n_perm, &c->mem));
    PermJob job;
    SC_TRY(permgen_begin(c, state6, n, n_perm, &job, c->stream2));
    SC_TRY(moran_prepare(c, n_perm < PERM_CHUNK ? n_perm : PERM_CHUNK));
    const int64_t chunks = ceil_div64(n_perm, PERM_CHUNK);
    // stream2: scan(0) scan(1) ...      stream3: swaps(k) after scan(k)      stream: score(k) after swaps(k)
    std::vector<hipEvent_t> ev((size_t)chunks * 2, nullptr);
    int rc = SC_OK;
    for (int64_t k = 0; k < chunks && rc == SC_OK; ++k) {
        const int64_t p0 = k * PERM_CHUNK, p1 = p0 + PERM_CHUNK < n_perm ? p0 + PERM_CHUNK : n_perm;
        hipEvent_t &scanned = ev[(size_t)(2 * k)], &swapped = ev[(size_t)(2 * k + 1)];
        rc = permgen_scan_chunk(c, &job, p1, c->stream2);
        if (rc == SC_OK && (hipEventCreateWithFlags(&scanned, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&swapped, hipEventDisableTiming) != hipSuccess ||
                            hipEventRecord(scanned, c->stream2) != hipSuccess ||
                            hipStreamWaitEvent(c->stream3, scanned, 0) != hipSuccess)) {
            sc_set_error("sc_moran_seeded: event plumbing failed");
            rc = SC_ERR_HIP;
        }
        if (rc == SC_OK) rc = permgen_swap_chunk(c, &job, p0, p1, c->stream3);
        if (rc == SC_OK && (hipEventRecord(swapped, c->stream3) != hipSuccess ||
                            hipStreamWaitEvent(c->stream, swapped, 0) != hipSuccess)) {
            sc_set_error("sc_moran_seeded: event plumbing failed");
            rc = SC_ERR_HIP;
        }
        if (rc == SC_OK) rc = moran_perm_range(c, p0, p1);
    }
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamSynchronize(c->stream3);
    (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    if (rc != SC_OK) return rc;
    SC_TRY(permgen_finish(c, &job, state6));
    c->p_count = n_perm;
    return moran_finish(c, n_perm, I_out, sims_out, count_ge_out, sim_sum_out, sim_sumsq_out);
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
