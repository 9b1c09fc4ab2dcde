// Neighbour graphs over 2-D coordinates (exact kNN / radius on a uniform bin grid), CSR graph
// management, squidpy's graph moments, transpose, neighbourhood composition.  gfx950 only.
//
// kNN design: points are counting-sorted into square bins (SoA x[], y[], id[] in bin order, so a
// row of bins is one contiguous, coalesced range).  One thread owns one query and walks square
// rings of bins around it, keeping its k best (distance, index) pairs in registers (fully unrolled
// insertion network, no scratch memory).  It stops as soon as the k-th best distance is provably
// smaller than the distance to anything outside the visited window, so the result is the exact
// kNN set, ordered by (squared distance, index).  Consecutive threads are consecutive points of the
// same bin, so a wavefront reads the same candidate ranges (L1 broadcast).
#include <float.h>
#include <math.h>

#include <hipcub/hipcub.hpp>
#include <vector>

#include "sc_ctx.h"

// ------------------------------------------------------------------------------------------------
// binning
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_split_xy(const double *__restrict__ xy, double *__restrict__ x,
                                                  double *__restrict__ y, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double2 v = reinterpret_cast<const double2 *>(xy)[i];
    x[i] = v.x;
    y[i] = v.y;
}

__device__ __forceinline__ int bin_coord(double v, double v0, double inv_h, int nb)
{
    int b = (int)floor((v - v0) * inv_h);
    return b < 0 ? 0 : (b >= nb ? nb - 1 : b);
}

__global__ __launch_bounds__(256) void k_bin_keys(const double *__restrict__ x, const double *__restrict__ y,
                                                  int64_t n, double x0, double y0, double inv_h, int nbx, int nby,
                                                  uint32_t *__restrict__ keys, int32_t *__restrict__ ids)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bx = bin_coord(x[i], x0, inv_h, nbx), by = bin_coord(y[i], y0, inv_h, nby);
    keys[i] = (uint32_t)by * (uint32_t)nbx + (uint32_t)bx;
    ids[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_gather_sorted(const double *__restrict__ x, const double *__restrict__ y,
                                                       const int32_t *__restrict__ sid, int64_t n,
                                                       double *__restrict__ sx, double *__restrict__ sy)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t j = sid[i];
    sx[i] = x[j];
    sy[i] = y[j];
}

// bin_start[b] = first sorted position whose key >= b  (keys sorted ascending)
__global__ __launch_bounds__(256) void k_bin_start(const uint32_t *__restrict__ keys, int64_t n, int64_t nbins,
                                                   int32_t *__restrict__ bin_start)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t cur = i < n ? keys[i] : (uint32_t)nbins;
    uint32_t prev = i > 0 ? keys[i - 1] + 1 : 0;  // first bin not yet closed
    for (uint32_t b = prev; b <= cur && b <= (uint32_t)nbins; ++b) bin_start[b] = (int32_t)i;
}

static int build_bins(sc_ctx *c, const double *xy, int64_t n, double target_per_bin, double min_h)
{
    SC_REQUIRE(n >= 1 && n <= 0x7fffffffLL, SC_ERR_INVALID, "n=%lld out of range", (long long)n);
    double xmin = DBL_MAX, xmax = -DBL_MAX, ymin = DBL_MAX, ymax = -DBL_MAX;
    for (int64_t i = 0; i < n; ++i) {
        double x = xy[2 * i], y = xy[2 * i + 1];
        SC_REQUIRE(isfinite(x) && isfinite(y), SC_ERR_INVALID, "coordinate %lld is not finite", (long long)i);
        xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax;
        ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
    }
    double w = xmax - xmin, hgt = ymax - ymin;
    double area = (w > 0 ? w : 1.0) * (hgt > 0 ? hgt : 1.0);
    double h = sqrt(area * target_per_bin / (double)n);
    if (h < min_h) h = min_h;
    double ext = w > hgt ? w : hgt;
    if (!(h > 0)) h = 1.0;
    // cap the grid at 4096 x 4096 bins
    if (ext / h > 4096.0) h = ext / 4096.0;
    int nbx = (int)floor(w / h) + 1, nby = (int)floor(hgt / h) + 1;
    c->nbx = nbx; c->nby = nby; c->gx0 = xmin; c->gy0 = ymin; c->gh = h;
    int64_t nbins = (int64_t)nbx * nby;

    SC_TRY(c->e_tmp_data.ensure(sizeof(double) * 2 * (size_t)n, &c->mem));  // staging for AoS upload
    SC_TRY(c->px.ensure(sizeof(double) * (size_t)n, &c->mem));
    SC_TRY(c->py.ensure(sizeof(double) * (size_t)n, &c->mem));
    SC_TRY(c->sx.ensure(sizeof(double) * (size_t)n, &c->mem));
    SC_TRY(c->sy.ensure(sizeof(double) * (size_t)n, &c->mem));
    SC_TRY(c->sid.ensure(sizeof(int32_t) * (size_t)n, &c->mem));
    SC_TRY(c->sid2.ensure(sizeof(int32_t) * (size_t)n, &c->mem));
    SC_TRY(c->bin_keys.ensure(sizeof(uint32_t) * (size_t)n, &c->mem));
    SC_TRY(c->bin_keys2.ensure(sizeof(uint32_t) * (size_t)n, &c->mem));
    SC_TRY(c->bin_start.ensure(sizeof(int32_t) * (size_t)(nbins + 1), &c->mem));
    SC_HIP(hipMemcpyAsync(c->e_tmp_data.p, xy, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    unsigned grid = (unsigned)ceil_div64(n, 256);
    hipLaunchKernelGGL(k_split_xy, dim3(grid), dim3(256), 0, c->stream, c->e_tmp_data.as<double>(),
                       c->px.as<double>(), c->py.as<double>(), n);
    hipLaunchKernelGGL(k_bin_keys, dim3(grid), dim3(256), 0, c->stream, c->px.as<double>(), c->py.as<double>(), n,
                       xmin, ymin, 1.0 / h, nbx, nby, c->bin_keys.as<uint32_t>(), c->sid2.as<int32_t>());
    int bits = 1;
    while (((int64_t)1 << bits) < nbins) ++bits;
    size_t tmp_bytes = 0;
    SC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, c->bin_keys.as<uint32_t>(),
                                              c->bin_keys2.as<uint32_t>(), c->sid2.as<int32_t>(),
                                              c->sid.as<int32_t>(), (int)n, 0, bits, c->stream));
    SC_TRY(c->cub_tmp.ensure(tmp_bytes, &c->mem));
    SC_HIP(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tmp_bytes, c->bin_keys.as<uint32_t>(),
                                              c->bin_keys2.as<uint32_t>(), c->sid2.as<int32_t>(),
                                              c->sid.as<int32_t>(), (int)n, 0, bits, c->stream));
    hipLaunchKernelGGL(k_gather_sorted, dim3(grid), dim3(256), 0, c->stream, c->px.as<double>(), c->py.as<double>(),
                       c->sid.as<int32_t>(), n, c->sx.as<double>(), c->sy.as<double>());
    hipLaunchKernelGGL(k_bin_start, dim3((unsigned)ceil_div64(n + 1, 256)), dim3(256), 0, c->stream,
                       c->bin_keys2.as<uint32_t>(), n, nbins, c->bin_start.as<int32_t>());
    SC_HIP(hipGetLastError());
    c->pts_n = n;
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// A1: exact kNN
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool cand_better(double d, int id, double ed, int eid)
{
    return d < ed || (d == ed && id < eid);
}

template <int K>
struct TopK {
    double d[K];
    int id[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = DBL_MAX; id[j] = 0x7fffffff; }
    }
    // sorted insert of a candidate known to beat the last entry
    __device__ __forceinline__ void insert(double cd, int cid)
    {
        bool b_hi = cand_better(cd, cid, d[K - 1], id[K - 1]);  // vs entry j
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            bool b_lo = cand_better(cd, cid, d[j - 1], id[j - 1]);  // vs entry j-1
            double nd = b_lo ? d[j - 1] : (b_hi ? cd : d[j]);
            int ni = b_lo ? id[j - 1] : (b_hi ? cid : id[j]);
            d[j] = nd;
            id[j] = ni;
            b_hi = b_lo;
        }
        if (b_hi) { d[0] = cd; id[0] = cid; }
    }
    __device__ __forceinline__ double kth(int k) const
    {
        double v = d[K - 1];
#pragma unroll
        for (int j = 0; j < K; ++j) v = (j == k - 1) ? d[j] : v;
        return v;
    }
};

template <int K>
__global__ __launch_bounds__(256) void k_knn(const double *__restrict__ sx, const double *__restrict__ sy,
                                             const int32_t *__restrict__ sid,
                                             const int32_t *__restrict__ bin_start, int64_t n, int k,
                                             int include_self, double x0, double y0, double h, int nbx, int nby,
                                             int32_t *__restrict__ idx_out, double *__restrict__ rd_out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double qx = sx[t], qy = sy[t];
    const int qid = sid[t];
    const double inv_h = 1.0 / h;
    const int bx = bin_coord(qx, x0, inv_h, nbx), by = bin_coord(qy, y0, inv_h, nby);
    TopK<K> best;
    best.init();
    double kth = DBL_MAX;
    const int rmax = (nbx > nby ? nbx : nby);
    const double slack = 1e-9 * h;
    for (int r = 0; r <= rmax; ++r) {
        const int ylo = by - r, yhi = by + r, xlo = bx - r, xhi = bx + r;
        const int cxlo = xlo < 0 ? 0 : xlo, cxhi = xhi >= nbx ? nbx - 1 : xhi;
        for (int yy = (ylo < 0 ? 0 : ylo); yy <= (yhi >= nby ? nby - 1 : yhi); ++yy) {
            const bool full = (yy == ylo) || (yy == yhi);
            // full row of the ring: bins [cxlo, cxhi]; interior rows: only the two end bins
            for (int seg = 0; seg < (full ? 1 : 2); ++seg) {
                int b0, b1;
                if (full) { b0 = cxlo; b1 = cxhi; }
                else if (seg == 0) { if (xlo < 0) continue; b0 = b1 = xlo; }
                else { if (xhi >= nbx || r == 0) continue; b0 = b1 = xhi; }
                const int s0 = bin_start[yy * nbx + b0], s1 = bin_start[yy * nbx + b1 + 1];
                for (int s = s0; s < s1; ++s) {
                    const int cid = sid[s];
                    const double dx = qx - sx[s], dy = qy - sy[s];
                    const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                    if (cid == qid && !include_self) continue;
                    if (cand_better(d, cid, best.d[K - 1], best.id[K - 1])) {
                        best.insert(d, cid);
                        kth = best.kth(k);
                    }
                }
            }
        }
        // distance from the query to the nearest unvisited region
        const bool l_out = xlo <= 0, r_out = xhi >= nbx - 1, b_out = ylo <= 0, t_out = yhi >= nby - 1;
        if (l_out && r_out && b_out && t_out) break;  // everything visited
        double m = DBL_MAX;
        if (!l_out) m = fmin(m, qx - (x0 + (double)xlo * h));
        if (!r_out) m = fmin(m, (x0 + (double)(xhi + 1) * h) - qx);
        if (!b_out) m = fmin(m, qy - (y0 + (double)ylo * h));
        if (!t_out) m = fmin(m, (y0 + (double)(yhi + 1) * h) - qy);
        m -= slack;
        if (m > 0.0 && kth < m * m) break;
    }
    // scatter to the original query order
    int32_t *o = idx_out + (int64_t)qid * k;
    double *od = rd_out ? rd_out + (int64_t)qid * k : nullptr;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k) {
            o[j] = best.id[j];
            if (od) od[j] = best.d[j];
        }
}

// k > 32 (any k < n): the k best candidates of a query live in a binary MAX-heap in global memory (slot j of query t
// at [j * n + t]: the root, every thread's hottest slot, is a coalesced row), keyed by (squared distance, index); a
// candidate that beats the root replaces it and sifts down.  The ring walk, the tie rule and the stopping rule are
// those of k_knn; at the end the heap is sorted in place (heap sort) and scattered to the query's row.  The register
// form needs 3 K registers per lane: K = 64 spilled 130 of them (r02) -- 33 <= k <= 64 come here too.
__device__ __forceinline__ bool cand_worse(double d, int id, double ed, int eid) { return cand_better(ed, eid, d, id); }

__device__ __forceinline__ void heap_sift_down(double *__restrict__ hd, int32_t *__restrict__ hi, int64_t n, int64_t t,
                                               int size, double cd, int cid)
{
    // place (cd, cid) into the heap of `size` slots starting at the root, whose old content is dropped
    int pos = 0;
    for (;;) {
        const int l = 2 * pos + 1, r = l + 1;
        if (l >= size) break;
        double wd = hd[(int64_t)l * n + t];
        int wi = hi[(int64_t)l * n + t], w = l;
        if (r < size) {
            const double rd = hd[(int64_t)r * n + t];
            const int ri = hi[(int64_t)r * n + t];
            if (cand_worse(rd, ri, wd, wi)) { wd = rd; wi = ri; w = r; }
        }
        if (!cand_worse(wd, wi, cd, cid)) break;     // the candidate is at least as bad as both children: it stays here
        hd[(int64_t)pos * n + t] = wd;
        hi[(int64_t)pos * n + t] = wi;
        pos = w;
    }
    hd[(int64_t)pos * n + t] = cd;
    hi[(int64_t)pos * n + t] = cid;
}

__global__ __launch_bounds__(256) void k_knn_heap(const double *__restrict__ sx, const double *__restrict__ sy,
                                                  const int32_t *__restrict__ sid,
                                                  const int32_t *__restrict__ bin_start, int64_t n, int k,
                                                  int include_self, double x0, double y0, double h, int nbx, int nby,
                                                  double *__restrict__ hd, int32_t *__restrict__ hi,
                                                  int32_t *__restrict__ idx_out, double *__restrict__ rd_out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double qx = sx[t], qy = sy[t];
    const int qid = sid[t];
    const double inv_h = 1.0 / h;
    const int bx = bin_coord(qx, x0, inv_h, nbx), by = bin_coord(qy, y0, inv_h, nby);
    for (int j = 0; j < k; ++j) { hd[(int64_t)j * n + t] = DBL_MAX; hi[(int64_t)j * n + t] = 0x7fffffff; }   // a valid heap
    double root_d = DBL_MAX;
    int root_i = 0x7fffffff;
    const int rmax = (nbx > nby ? nbx : nby);
    const double slack = 1e-9 * h;
    for (int r = 0; r <= rmax; ++r) {
        const int ylo = by - r, yhi = by + r, xlo = bx - r, xhi = bx + r;
        const int cxlo = xlo < 0 ? 0 : xlo, cxhi = xhi >= nbx ? nbx - 1 : xhi;
        for (int yy = (ylo < 0 ? 0 : ylo); yy <= (yhi >= nby ? nby - 1 : yhi); ++yy) {
            const bool full = (yy == ylo) || (yy == yhi);
            for (int seg = 0; seg < (full ? 1 : 2); ++seg) {
                int b0, b1;
                if (full) { b0 = cxlo; b1 = cxhi; }
                else if (seg == 0) { if (xlo < 0) continue; b0 = b1 = xlo; }
                else { if (xhi >= nbx || r == 0) continue; b0 = b1 = xhi; }
                const int s0 = bin_start[yy * nbx + b0], s1 = bin_start[yy * nbx + b1 + 1];
                for (int s = s0; s < s1; ++s) {
                    const int cid = sid[s];
                    const double dx = qx - sx[s], dy = qy - sy[s];
                    const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                    if (cid == qid && !include_self) continue;
                    if (cand_better(d, cid, root_d, root_i)) {
                        heap_sift_down(hd, hi, n, t, k, d, cid);
                        root_d = hd[t];
                        root_i = hi[t];
                    }
                }
            }
        }
        const bool l_out = xlo <= 0, r_out = xhi >= nbx - 1, b_out = ylo <= 0, t_out = yhi >= nby - 1;
        if (l_out && r_out && b_out && t_out) break;  // everything visited
        double m = DBL_MAX;
        if (!l_out) m = fmin(m, qx - (x0 + (double)xlo * h));
        if (!r_out) m = fmin(m, (x0 + (double)(xhi + 1) * h) - qx);
        if (!b_out) m = fmin(m, qy - (y0 + (double)ylo * h));
        if (!t_out) m = fmin(m, (y0 + (double)(yhi + 1) * h) - qy);
        m -= slack;
        if (m > 0.0 && root_d < m * m) break;        // the root is the k-th best so far
    }
    // heap sort: the worst of the remaining heap goes to its end; slots end up ascending by (distance, index)
    for (int m = k - 1; m >= 1; --m) {
        const double ld = hd[(int64_t)m * n + t];
        const int li = hi[(int64_t)m * n + t];
        hd[(int64_t)m * n + t] = hd[t];
        hi[(int64_t)m * n + t] = hi[t];
        heap_sift_down(hd, hi, n, t, m, ld, li);
    }
    int32_t *o = idx_out + (int64_t)qid * k;
    double *od = rd_out ? rd_out + (int64_t)qid * k : nullptr;
    for (int j = 0; j < k; ++j) {
        o[j] = hi[(int64_t)j * n + t];
        if (od) od[j] = hd[(int64_t)j * n + t];
    }
}

template <int K>
static void launch_knn(sc_ctx *c, int64_t n, int k, int include_self)
{
    hipLaunchKernelGGL(k_knn<K>, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream, c->sx.as<double>(),
                       c->sy.as<double>(), c->sid.as<int32_t>(), c->bin_start.as<int32_t>(), n, k, include_self,
                       c->gx0, c->gy0, c->gh, c->nbx, c->nby, c->knn_idx.as<int32_t>(), c->knn_rd.as<double>());
}

extern "C" int sc_knn_2d(sc_ctx *c, const double *xy, int64_t n, int k, int include_self, int32_t *idx_out,
                         double *rdist_out)
{
    SC_REQUIRE(c && xy, SC_ERR_INVALID, "sc_knn_2d: null pointer");
    SC_REQUIRE(k >= 1 && k <= (1 << 16), SC_ERR_INVALID, "sc_knn_2d: k=%d unsupported (1..65536)", k);
    SC_REQUIRE(n >= 1, SC_ERR_INVALID, "sc_knn_2d: n must be >= 1");
    SC_REQUIRE((int64_t)k <= n - (include_self ? 0 : 1), SC_ERR_INVALID,
               "sc_knn_2d: k=%d needs more than the %lld available points", k, (long long)n);
    SC_HIP(hipSetDevice(c->device));
    c->knn_n = 0;
    SC_TRY(build_bins(c, xy, n, 0.5 * (k + 1) > 4.0 ? 0.5 * (k + 1) : 4.0, 0.0));
    SC_TRY(c->knn_idx.ensure(sizeof(int32_t) * (size_t)n * k, &c->mem));
    SC_TRY(c->knn_rd.ensure(sizeof(double) * (size_t)n * k, &c->mem));
    {
        KernelTimerScope ts(c, SC_K_KNN);
        if (k <= 8) launch_knn<8>(c, n, k, include_self);
        else if (k <= 16) launch_knn<16>(c, n, k, include_self);
        else if (k <= 32) launch_knn<32>(c, n, k, include_self);
        else {
            SC_TRY(c->knn_hd.ensure(sizeof(double) * (size_t)n * k, &c->mem));
            SC_TRY(c->knn_hi.ensure(sizeof(int32_t) * (size_t)n * k, &c->mem));
            hipLaunchKernelGGL(k_knn_heap, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream, c->sx.as<double>(),
                               c->sy.as<double>(), c->sid.as<int32_t>(), c->bin_start.as<int32_t>(), n, k, include_self,
                               c->gx0, c->gy0, c->gh, c->nbx, c->nby, c->knn_hd.as<double>(), c->knn_hi.as<int32_t>(),
                               c->knn_idx.as<int32_t>(), c->knn_rd.as<double>());
        }
    }
    SC_HIP(hipGetLastError());
    if (idx_out)
        SC_HIP(hipMemcpyAsync(idx_out, c->knn_idx.p, sizeof(int32_t) * (size_t)n * k, hipMemcpyDeviceToHost,
                              c->stream));
    if (rdist_out)
        SC_HIP(hipMemcpyAsync(rdist_out, c->knn_rd.p, sizeof(double) * (size_t)n * k, hipMemcpyDeviceToHost,
                              c->stream));
    // nothing to hand back: the result stays on the device and every consumer is ordered behind it on the context's
    // stream, so the host need not wait (the caller's coordinate array has been staged by the pageable-memory copy)
    if (idx_out || rdist_out) SC_HIP(hipStreamSynchronize(c->stream));
    else {   // for a later sc_knn_fetch on the copy stream
        if (!c->knn_done) SC_HIP(hipEventCreateWithFlags(&c->knn_done, hipEventDisableTiming));
        SC_HIP(hipEventRecord(c->knn_done, c->stream));
    }
    c->knn_n = n;
    c->knn_k = k;
    return SC_OK;
}

// The result of the last sc_knn_2d that was called without output arrays, copied out on the context's side stream (ordered
// behind the search by an event): the copy neither waits for what the context's stream has been given since, nor holds
// it up -- a caller's thread can fetch the neighbour lists while another uploads the expression (PCIe is full duplex).
extern "C" int sc_knn_fetch(sc_ctx *c, int32_t *idx_out, double *rdist_out)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "sc_knn_fetch: null context");
    SC_REQUIRE(c->knn_n > 0 && c->knn_done, SC_ERR_STATE, "sc_knn_fetch: no resident kNN result (sc_knn_2d without output arrays first)");
    SC_HIP(hipSetDevice(c->device));
    // the side stream of the graph moments serves as the copy stream (a stream more per context would be a hardware queue
    // more: a process whose streams outnumber GPU_MAX_HW_QUEUES has them share queues, which the generator must avoid)
    sc_graph_moments_drain(c);
    if (!c->stream_m) SC_HIP(hipStreamCreateWithFlags(&c->stream_m, hipStreamNonBlocking));
    const size_t nk = (size_t)c->knn_n * (size_t)c->knn_k;
    SC_HIP(hipStreamWaitEvent(c->stream_m, c->knn_done, 0));
    if (idx_out) SC_HIP(hipMemcpyAsync(idx_out, c->knn_idx.p, sizeof(int32_t) * nk, hipMemcpyDeviceToHost, c->stream_m));
    if (rdist_out) SC_HIP(hipMemcpyAsync(rdist_out, c->knn_rd.p, sizeof(double) * nk, hipMemcpyDeviceToHost, c->stream_m));
    SC_HIP(hipStreamSynchronize(c->stream_m));
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// A2: radius graph (closed ball, self removed), two passes
// ------------------------------------------------------------------------------------------------

template <bool FILL>
__global__ __launch_bounds__(256) void k_radius(const double *__restrict__ sx, const double *__restrict__ sy,
                                                const int32_t *__restrict__ sid,
                                                const int32_t *__restrict__ bin_start, int64_t n, double r2,
                                                int rings, double x0, double y0, double h, int nbx, int nby,
                                                long long *__restrict__ counts,
                                                const long long *__restrict__ indptr,
                                                int32_t *__restrict__ indices)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double qx = sx[t], qy = sy[t];
    const int qid = sid[t];
    const double inv_h = 1.0 / h;
    const int bx = bin_coord(qx, x0, inv_h, nbx), by = bin_coord(qy, y0, inv_h, nby);
    const int ylo = by - rings < 0 ? 0 : by - rings, yhi = by + rings >= nby ? nby - 1 : by + rings;
    const int xlo = bx - rings < 0 ? 0 : bx - rings, xhi = bx + rings >= nbx ? nbx - 1 : bx + rings;
    long long cnt = 0;
    int32_t *row = FILL ? indices + indptr[qid] : nullptr;
    // short rows are kept ascending by insertion; long ones (large radii) are appended and heap-sorted at the end:
    // insertion in global memory is O(degree^2) writes
    const bool by_insertion = FILL ? (indptr[qid + 1] - indptr[qid] <= 32) : true;
    for (int yy = ylo; yy <= yhi; ++yy) {
        const int s0 = bin_start[yy * nbx + xlo], s1 = bin_start[yy * nbx + xhi + 1];
        for (int s = s0; s < s1; ++s) {
            const int cid = sid[s];
            const double dx = qx - sx[s], dy = qy - sy[s];
            const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
            if (cid != qid && d <= r2) {
                if (FILL) {
                    long long j = cnt;
                    if (by_insertion)
                        while (j > 0 && row[j - 1] > cid) { row[j] = row[j - 1]; --j; }
                    row[j] = cid;
                }
                ++cnt;
            }
        }
    }
    if (!FILL) counts[qid] = cnt;
    if (FILL && !by_insertion) {   // in-place heapsort, ascending
        auto sift = [&](long long root, long long end) {
            for (;;) {
                long long child = 2 * root + 1;
                if (child >= end) break;
                if (child + 1 < end && row[child] < row[child + 1]) ++child;
                if (row[root] >= row[child]) break;
                const int32_t tmp = row[root]; row[root] = row[child]; row[child] = tmp;
                root = child;
            }
        };
        for (long long k = cnt / 2 - 1; k >= 0; --k) sift(k, cnt);
        for (long long end = cnt - 1; end > 0; --end) {
            const int32_t tmp = row[0]; row[0] = row[end]; row[end] = tmp;
            sift(0, end);
        }
    }
}

extern "C" int sc_radius_count_2d(sc_ctx *c, const double *xy, int64_t n, double radius, int64_t *indptr_out)
{
    SC_REQUIRE(c && xy && indptr_out, SC_ERR_INVALID, "sc_radius_count_2d: null pointer");
    SC_REQUIRE(radius > 0 && isfinite(radius), SC_ERR_INVALID, "radius must be > 0, got %g", radius);
    SC_HIP(hipSetDevice(c->device));
    c->radius = -1.0;
    sc_graph_moments_drain(c);   // (a moments job on the side stream uses gt_cursor and reads the graph's arrays)
    // bins no smaller than the radius: a 3x3 window always covers the closed ball
    SC_TRY(build_bins(c, xy, n, 4.0, radius));
    int rings = (int)ceil(radius / c->gh * (1.0 + 1e-9));
    if (rings < 1) rings = 1;
    SC_TRY(c->rad_indptr.ensure(sizeof(long long) * (size_t)(n + 1), &c->mem));
    SC_TRY(c->gt_cursor.ensure(sizeof(long long) * (size_t)(n + 1), &c->mem));
    long long *counts = c->gt_cursor.as<long long>();
    SC_HIP(hipMemsetAsync(counts, 0, sizeof(long long) * (size_t)(n + 1), c->stream));
    hipLaunchKernelGGL(k_radius<false>, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream,
                       c->sx.as<double>(), c->sy.as<double>(), c->sid.as<int32_t>(), c->bin_start.as<int32_t>(), n,
                       radius * radius, rings, c->gx0, c->gy0, c->gh, c->nbx, c->nby, counts,
                       (const long long *)nullptr, (int32_t *)nullptr);
    SC_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    SC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, c->rad_indptr.as<long long>(), (int)(n + 1),
                                            c->stream));
    SC_TRY(c->cub_tmp.ensure(tmp_bytes, &c->mem));
    SC_HIP(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tmp_bytes, counts, c->rad_indptr.as<long long>(),
                                            (int)(n + 1), c->stream));
    SC_HIP(hipMemcpyAsync(indptr_out, c->rad_indptr.p, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToHost,
                          c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    c->radius = radius;
    return SC_OK;
}

extern "C" int sc_radius_fill_2d(sc_ctx *c, int64_t nnz, int32_t *indices_out)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->radius > 0 && c->pts_n > 0, SC_ERR_STATE, "sc_radius_fill_2d: call sc_radius_count_2d first");
    int64_t n = c->pts_n;
    long long total = 0;
    SC_HIP(hipMemcpyAsync(&total, c->rad_indptr.as<long long>() + n, sizeof(long long), hipMemcpyDeviceToHost,
                          c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    SC_REQUIRE(total == nnz, SC_ERR_INVALID, "sc_radius_fill_2d: nnz=%lld but the count pass found %lld",
               (long long)nnz, total);
    SC_REQUIRE(nnz == 0 || indices_out, SC_ERR_INVALID, "sc_radius_fill_2d: null output");
    sc_graph_moments_drain(c);   // (... and this pass overwrites, maybe reallocates, g_indices)
    SC_TRY(c->g_indices.ensure(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    int rings = (int)ceil(c->radius / c->gh * (1.0 + 1e-9));
    if (rings < 1) rings = 1;
    hipLaunchKernelGGL(k_radius<true>, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream,
                       c->sx.as<double>(), c->sy.as<double>(), c->sid.as<int32_t>(), c->bin_start.as<int32_t>(), n,
                       c->radius * c->radius, rings, c->gx0, c->gy0, c->gh, c->nbx, c->nby, (long long *)nullptr,
                       c->rad_indptr.as<long long>(), c->g_indices.as<int32_t>());
    SC_HIP(hipGetLastError());
    if (nnz > 0)
        SC_HIP(hipMemcpyAsync(indices_out, c->g_indices.p, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost,
                              c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    // the fill pass used g_indices as scratch: the active graph (if any) is gone
    c->g_n = 0;
    c->g_nnz = 0;
    c->gt_valid = false;
    c->s0_valid = false;
    c->s0_only_valid = false;
    c->prep_early = false;
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// A3: CSR graph
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_check_csr(const long long *__restrict__ indptr,
                                                   const int32_t *__restrict__ indices, int64_t n,
                                                   int *__restrict__ flag)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = 0;
    for (long long e = indptr[i]; e < indptr[i + 1]; ++e) {
        int32_t j = indices[e];
        if (j < 0 || j >= n) bad |= 1;
        if (e > indptr[i] && indices[e - 1] >= j) bad |= 2;
    }
    if (bad) atomicOr(flag, bad);
}

// ---- processing order with spatial locality (see sc_ctx.h) ----
__global__ __launch_bounds__(256) void k_iota32(int32_t *__restrict__ a, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_rank_of(const int32_t *__restrict__ order, int64_t n, int32_t *__restrict__ rank)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) rank[order[r]] = (int32_t)r;
}

__global__ __launch_bounds__(256) void k_relabel_edges(const int32_t *__restrict__ indices, const double *__restrict__ w,
                                                       const int32_t *__restrict__ rank, int64_t nnz,
                                                       int32_t *__restrict__ indices_r, float *__restrict__ w32)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    indices_r[e] = rank[indices[e]];
    w32[e] = (float)w[e];
}

__global__ __launch_bounds__(256) void k_edge_rows(const long long *__restrict__ indptr, const int32_t *__restrict__ rank,
                                                   int64_t n, int32_t *__restrict__ erow_r)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = rank[i];
    for (long long e = indptr[i]; e < indptr[i + 1]; ++e) erow_r[e] = r;
}

// Morton (Z-curve) key of a point: 16 bits per axis over the bin grid's extent, x in the even bits
__device__ __forceinline__ uint32_t spread16(uint32_t v)
{
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__global__ __launch_bounds__(256) void k_morton_keys(const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                                                     double x0, double y0, double inv_ext, uint32_t *__restrict__ keys,
                                                     int32_t *__restrict__ ids)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double fx = (x[i] - x0) * inv_ext * 65536.0, fy = (y[i] - y0) * inv_ext * 65536.0;
    fx = fx < 0.0 ? 0.0 : (fx > 65535.0 ? 65535.0 : fx);
    fy = fy < 0.0 ? 0.0 : (fy > 65535.0 ? 65535.0 : fy);
    keys[i] = spread16((uint32_t)fx) | (spread16((uint32_t)fy) << 1);
    ids[i] = (int32_t)i;
}

// The graph setters call this: if the points of the last neighbour search are the graph's cells (same count), a
// spatially compact processing order is kept with the graph (the bin-sorted order of the neighbour search, or a Morton
// curve, see below), otherwise the identity.  A wrong guess costs speed, never correctness: it is only the order in which
// kernels that gather neighbours' rows (k_lag, k_lag_u8, local Moran, enrichment) walk the cells.
int sc_graph_capture_order(sc_ctx *c, int64_t n)
{
    c->g_order_ready = false;
    SC_TRY(c->g_order.ensure(sizeof(int32_t) * (size_t)n, &c->mem));
    // Default: the row-major bin order of the neighbour search (it is there already).  SC_GRAPH_ORDER=morton: a Morton (Z)
    // curve over the bin grid's extent instead (one more radix sort, 1.5 ms at 1M cells) -- measured in r03 on k_lag, the
    // local Moran count kernels and the enrichment: it cuts their L2 misses (k_lag: a third of the fetched bytes) but not
    // their time (all of them are bound by the rows through the CUs' texture path, not by where they come from).
    static const char *order_env = getenv("SC_GRAPH_ORDER");
    const bool morton = order_env && order_env[0] == 'm';
    if (!morton && c->pts_n == n && c->sid.p) {
        SC_HIP(hipMemcpyAsync(c->g_order.p, c->sid.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    } else if (morton && c->pts_n == n && c->px.p && c->py.p && c->bin_keys.p && c->bin_keys2.p && c->sid2.p) {
        const double ext = (double)(c->nbx > c->nby ? c->nbx : c->nby) * c->gh;
        hipLaunchKernelGGL(k_morton_keys, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream, c->px.as<double>(),
                           c->py.as<double>(), n, c->gx0, c->gy0, ext > 0.0 ? 1.0 / ext : 0.0, c->bin_keys.as<uint32_t>(),
                           c->sid2.as<int32_t>());
        size_t tmp_bytes = 0;
        SC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, c->bin_keys.as<uint32_t>(), c->bin_keys2.as<uint32_t>(),
                                                  c->sid2.as<int32_t>(), c->g_order.as<int32_t>(), (int)n, 0, 32, c->stream));
        SC_TRY(c->cub_tmp.ensure(tmp_bytes, &c->mem));
        SC_HIP(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tmp_bytes, c->bin_keys.as<uint32_t>(), c->bin_keys2.as<uint32_t>(),
                                                  c->sid2.as<int32_t>(), c->g_order.as<int32_t>(), (int)n, 0, 32, c->stream));
    } else
        hipLaunchKernelGGL(k_iota32, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream, c->g_order.as<int32_t>(), n);
    SC_HIP(hipGetLastError());
    c->g_order_captured = true;
    return SC_OK;
}

int sc_graph_ensure_order(sc_ctx *c)
{
    if (c->g_order_ready) return SC_OK;
    const int64_t n = c->g_n, nnz = c->g_nnz;
    SC_REQUIRE(n > 0 && c->g_order_captured, SC_ERR_STATE, "no graph set");
    SC_TRY(c->g_rank.ensure(sizeof(int32_t) * (size_t)n, &c->mem));
    SC_TRY(c->g_indices_r.ensure(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    SC_TRY(c->g_w32.ensure(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    SC_TRY(c->g_erow_r.ensure(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    hipLaunchKernelGGL(k_rank_of, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream, c->g_order.as<int32_t>(), n,
                       c->g_rank.as<int32_t>());
    if (nnz > 0)
        hipLaunchKernelGGL(k_relabel_edges, dim3((unsigned)ceil_div64(nnz, 256)), dim3(256), 0, c->stream,
                           c->g_indices.as<int32_t>(), c->g_data.as<double>(), c->g_rank.as<int32_t>(), nnz,
                           c->g_indices_r.as<int32_t>(), c->g_w32.as<float>());
    hipLaunchKernelGGL(k_edge_rows, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                       c->g_rank.as<int32_t>(), n, c->g_erow_r.as<int32_t>());
    SC_HIP(hipGetLastError());
    c->g_order_ready = true;
    return SC_OK;
}

extern "C" int sc_graph_set_csr(sc_ctx *c, const int64_t *indptr, const int32_t *indices, const double *data,
                                int64_t n, int64_t nnz)
{
    SC_REQUIRE(c && indptr, SC_ERR_INVALID, "sc_graph_set_csr: null pointer");
    SC_REQUIRE(n >= 1 && n <= 0x7fffffffLL && nnz >= 0, SC_ERR_INVALID, "sc_graph_set_csr: bad shape");
    SC_REQUIRE(nnz == 0 || (indices && data), SC_ERR_INVALID, "sc_graph_set_csr: null indices/data");
    SC_REQUIRE(indptr[0] == 0 && indptr[n] == nnz, SC_ERR_INVALID, "sc_graph_set_csr: indptr does not span nnz");
    int64_t deg_max = 0, deg_min = INT64_MAX;
    for (int64_t i = 0; i < n; ++i) {
        SC_REQUIRE(indptr[i + 1] >= indptr[i], SC_ERR_INVALID, "sc_graph_set_csr: indptr not monotone");
        const int64_t deg = indptr[i + 1] - indptr[i];
        if (deg > deg_max) deg_max = deg;
        if (deg < deg_min) deg_min = deg;
    }
    // all weights equal (a kNN graph after row normalisation: 1 / k)?  Integer-count genes then sit on an integer
    // lattice and their permutation counts are decided exactly (sc_moran.hip, "lattice genes")
    double uniform_w = nnz > 0 ? data[0] : 0.0;
    for (int64_t e = 1; e < nnz && uniform_w != 0.0; ++e)
        if (data[e] != uniform_w) uniform_w = 0.0;
    if (!(uniform_w > 0.0) || uniform_w > 1e300) uniform_w = 0.0;
    SC_HIP(hipSetDevice(c->device));
    sc_graph_moments_drain(c);   // (a moments computation on the side stream still reads the old arrays)
    c->g_n = 0;
    c->gt_valid = false;
    c->s0_valid = false;
    c->s0_only_valid = false;
    c->prep_early = false;
    SC_TRY(c->g_indptr.ensure(sizeof(int64_t) * (size_t)(n + 1), &c->mem));
    SC_TRY(c->g_indices.ensure(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    SC_TRY(c->g_data.ensure(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    SC_TRY(c->perm_flag.ensure(sizeof(int), &c->mem));
    SC_HIP(hipMemcpyAsync(c->g_indptr.p, indptr, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) {
        SC_HIP(hipMemcpyAsync(c->g_indices.p, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice,
                              c->stream));
        SC_HIP(hipMemcpyAsync(c->g_data.p, data, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, c->stream));
    }
    SC_HIP(hipMemsetAsync(c->perm_flag.p, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_check_csr, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream,
                       c->g_indptr.as<long long>(), c->g_indices.as<int32_t>(), n, c->perm_flag.as<int>());
    int flag = 0;
    SC_HIP(hipMemcpyAsync(&flag, c->perm_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    SC_REQUIRE(!(flag & 1), SC_ERR_INVALID, "sc_graph_set_csr: column index out of range");
    SC_REQUIRE(!(flag & 2), SC_ERR_INVALID,
               "sc_graph_set_csr: rows must have strictly ascending column indices (sort_indices/sum_duplicates)");
    SC_TRY(sc_graph_capture_order(c, n));
    c->g_n = n;
    c->g_nnz = nnz;
    c->g_uniform_w = uniform_w;
    c->g_deg_max = deg_max;
    c->g_regular = deg_min == deg_max;   // (the lattice form of the Moran statistic needs equal weights AND equal degrees)
    return SC_OK;
}

// rows of the kNN result, re-ordered ascending by column index (as scipy's COO->CSR gives, AC:404)
__global__ __launch_bounds__(256) void k_knn_to_csr(const int32_t *__restrict__ knn, int64_t n, int k, double w,
                                                    long long *__restrict__ indptr, int32_t *__restrict__ indices,
                                                    double *__restrict__ data)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    indptr[i] = i * k;
    if (i == n) return;
    int32_t *row = indices + i * k;
    for (int j = 0; j < k; ++j) {
        int32_t v = knn[i * k + j];
        int m = j;
        while (m > 0 && row[m - 1] > v) { row[m] = row[m - 1]; --m; }
        row[m] = v;
        data[i * k + j] = w;
    }
}

extern "C" int sc_graph_from_knn(sc_ctx *c, double weight)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->knn_n > 0, SC_ERR_STATE, "sc_graph_from_knn: no kNN result (call sc_knn_2d first)");
    int64_t n = c->knn_n, nnz = n * c->knn_k;
    sc_graph_moments_drain(c);   // (a moments computation on the side stream still reads the old arrays)
    c->g_n = 0;
    c->gt_valid = false;
    c->s0_valid = false;
    c->s0_only_valid = false;
    c->prep_early = false;
    SC_TRY(c->g_indptr.ensure(sizeof(int64_t) * (size_t)(n + 1), &c->mem));
    SC_TRY(c->g_indices.ensure(sizeof(int32_t) * (size_t)nnz, &c->mem));
    SC_TRY(c->g_data.ensure(sizeof(double) * (size_t)nnz, &c->mem));
    hipLaunchKernelGGL(k_knn_to_csr, dim3((unsigned)ceil_div64(n + 1, 256)), dim3(256), 0, c->stream,
                       c->knn_idx.as<int32_t>(), n, c->knn_k, weight, c->g_indptr.as<long long>(),
                       c->g_indices.as<int32_t>(), c->g_data.as<double>());
    SC_HIP(hipGetLastError());
    SC_TRY(sc_graph_capture_order(c, n));
    c->g_n = n;     // (no host wait: consumers are ordered behind these launches on the context's stream)
    c->g_nnz = nnz;
    c->g_uniform_w = (weight > 0.0 && weight < 1e300) ? weight : 0.0;
    c->g_deg_max = c->knn_k;
    c->g_regular = true;
    return SC_OK;
}

extern "C" int sc_graph_shape(sc_ctx *c, int64_t *n, int64_t *nnz)
{
    SC_REQUIRE(c && n && nnz, SC_ERR_INVALID, "null pointer");
    *n = c->g_n;
    *nnz = c->g_nnz;
    return SC_OK;
}

extern "C" int sc_graph_get(sc_ctx *c, int64_t *indptr_out, int32_t *indices_out, double *data_out)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "sc_graph_get: no graph set");
    if (indptr_out)
        SC_HIP(hipMemcpyAsync(indptr_out, c->g_indptr.p, sizeof(int64_t) * (size_t)(c->g_n + 1), hipMemcpyDeviceToHost,
                              c->stream));
    if (indices_out && c->g_nnz)
        SC_HIP(hipMemcpyAsync(indices_out, c->g_indices.p, sizeof(int32_t) * (size_t)c->g_nnz, hipMemcpyDeviceToHost,
                              c->stream));
    if (data_out && c->g_nnz)
        SC_HIP(hipMemcpyAsync(data_out, c->g_data.p, sizeof(double) * (size_t)c->g_nnz, hipMemcpyDeviceToHost,
                              c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// ---- transpose (deterministic: counting sort by column, rows then sorted by source row) ----------

__global__ __launch_bounds__(256) void k_col_count(const int32_t *__restrict__ indices, int64_t nnz,
                                                   unsigned long long *__restrict__ cnt)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) atomicAdd(&cnt[indices[e]], 1ull);
}

// one thread per source row i (ascending e): slot claimed with an atomic cursor; rows of the
// transpose are sorted afterwards so that the final layout does not depend on the claim order
__global__ __launch_bounds__(256) void k_transpose_fill(const long long *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const double *__restrict__ data, int64_t n,
                                                        unsigned long long *__restrict__ cursor,
                                                        int32_t *__restrict__ t_indices, double *__restrict__ t_data)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (long long e = indptr[i]; e < indptr[i + 1]; ++e) {
        unsigned long long pos = atomicAdd(&cursor[indices[e]], 1ull);
        t_indices[pos] = (int32_t)i;
        t_data[pos] = data[e];
    }
}

__global__ __launch_bounds__(256) void k_sort_rows(const long long *__restrict__ indptr, int32_t *__restrict__ indices,
                                                   double *__restrict__ data, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long e0 = indptr[i], e1 = indptr[i + 1];
    for (long long a = e0 + 1; a < e1; ++a) {
        int32_t v = indices[a];
        double w = data[a];
        long long b = a;
        while (b > e0 && indices[b - 1] > v) { indices[b] = indices[b - 1]; data[b] = data[b - 1]; --b; }
        indices[b] = v;
        data[b] = w;
    }
}

static int graph_build_transpose(sc_ctx *c, bool wait);

int sc_graph_ensure_transpose(sc_ctx *c)
{
    if (c->gt_valid) {
        // built on the side stream by a moments computation that may still be running: order this stream behind it
        if (c->mom_pending && c->mom_done) SC_HIP(hipStreamWaitEvent(c->stream, c->mom_done, 0));
        return SC_OK;
    }
    return graph_build_transpose(c, true);
}

// the transposed graph, built on c->stream (which the caller may have pointed at the side stream)
static int graph_build_transpose(sc_ctx *c, bool wait)
{
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "no graph set");
    int64_t n = c->g_n, nnz = c->g_nnz;
    SC_TRY(c->gt_indptr.ensure(sizeof(long long) * (size_t)(n + 1), &c->mem));
    SC_TRY(c->gt_cursor.ensure(sizeof(long long) * (size_t)(n + 1), &c->mem));
    SC_TRY(c->gt_indices.ensure(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    SC_TRY(c->gt_data.ensure(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1), &c->mem));
    unsigned long long *cur = c->gt_cursor.as<unsigned long long>();
    SC_HIP(hipMemsetAsync(cur, 0, sizeof(long long) * (size_t)(n + 1), c->stream));
    if (nnz > 0)
        hipLaunchKernelGGL(k_col_count, dim3((unsigned)ceil_div64(nnz, 256)), dim3(256), 0, c->stream,
                           c->g_indices.as<int32_t>(), nnz, cur);
    size_t tmp_bytes = 0;
    SC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (long long *)cur, c->gt_indptr.as<long long>(),
                                            (int)(n + 1), c->stream));
    SC_TRY(c->gt_tmp.ensure(tmp_bytes, &c->mem));
    SC_HIP(hipcub::DeviceScan::ExclusiveSum(c->gt_tmp.p, tmp_bytes, (long long *)cur, c->gt_indptr.as<long long>(),
                                            (int)(n + 1), c->stream));
    SC_HIP(hipMemcpyAsync(cur, c->gt_indptr.p, sizeof(long long) * (size_t)(n + 1), hipMemcpyDeviceToDevice,
                          c->stream));
    hipLaunchKernelGGL(k_transpose_fill, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream,
                       c->g_indptr.as<long long>(), c->g_indices.as<int32_t>(), c->g_data.as<double>(), n, cur,
                       c->gt_indices.as<int32_t>(), c->gt_data.as<double>());
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream,
                       c->gt_indptr.as<long long>(), c->gt_indices.as<int32_t>(), c->gt_data.as<double>(), n);
    SC_HIP(hipGetLastError());
    if (wait) SC_HIP(hipStreamSynchronize(c->stream));
    c->gt_valid = true;
    return SC_OK;
}

// ---- moments ---------------------------------------------------------------------------------------

#define MOM_ROWS_PER_BLOCK 1024

// per block partials of: s0 (sum w), t2 (sum over W-edges of the (w_ij + w_ji)^2 contributions), s2
__global__ __launch_bounds__(256) void k_moments(const long long *__restrict__ indptr,
                                                 const int32_t *__restrict__ indices,
                                                 const double *__restrict__ data,
                                                 const long long *__restrict__ t_indptr,
                                                 const double *__restrict__ t_data, int64_t n,
                                                 double *__restrict__ partial)
{
    __shared__ double sh[3][256];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int64_t r0 = (int64_t)blockIdx.x * MOM_ROWS_PER_BLOCK;
    int64_t r1 = r0 + MOM_ROWS_PER_BLOCK < n ? r0 + MOM_ROWS_PER_BLOCK : n;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
        double rs = 0.0, cs = 0.0;
        for (long long e = indptr[i]; e < indptr[i + 1]; ++e) {
            int32_t j = indices[e];
            double w = data[e];
            rs += w;
            // look for the reverse edge (j -> i) in row j (ascending columns)
            long long lo = indptr[j], hi = indptr[j + 1];
            double wr = 0.0;
            bool found = false;
            while (lo < hi) {
                long long mid = (lo + hi) >> 1;
                int32_t v = indices[mid];
                if (v < i) lo = mid + 1;
                else if (v > i) hi = mid;
                else { wr = data[mid]; found = true; break; }
            }
            // (i,j) and (j,i) both stored: this edge contributes (w_ij + w_ji)^2 once (the mirror edge
            // contributes the same again when its own row is visited); one-directional: both (i,j)
            // and (j,i) of W + W^T equal w_ij
            a1 += found ? (w + wr) * (w + wr) : 2.0 * w * w;
        }
        for (long long e = t_indptr[i]; e < t_indptr[i + 1]; ++e) cs += t_data[e];
        a0 += rs;
        a2 += (rs + cs) * (rs + cs);
    }
    sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1; sh[2][threadIdx.x] = a2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int m = 0; m < 3; ++m) sh[m][threadIdx.x] += sh[m][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[(int64_t)blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

// Start the full moments (s0, s1, s2) of the active graph on the SIDE stream, behind everything enqueued on the context
// stream so far, and return without waiting: transpose, reverse-edge search and per-block partials into pinned host
// memory.  The scoring's set-up calls this so that the 6 ms (1M cells x 15) leave its serial prelude; sc_graph_moments
// collects.  Nothing else may replace the graph's arrays before sc_graph_moments_drain.
int sc_graph_moments_begin(sc_ctx *c)
{
    if (c->s0_valid || c->mom_pending) return SC_OK;
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "no graph set");
    const int64_t n = c->g_n;
    const int blocks = (int)ceil_div64(n, MOM_ROWS_PER_BLOCK);
    if (!c->stream_m) SC_HIP(hipStreamCreateWithFlags(&c->stream_m, hipStreamNonBlocking));
    if (!c->mom_ready) SC_HIP(hipEventCreateWithFlags(&c->mom_ready, hipEventDisableTiming));
    if (!c->mom_done) SC_HIP(hipEventCreateWithFlags(&c->mom_done, hipEventDisableTiming));
    if (blocks > c->mom_blocks) {
        if (c->mom_host) (void)hipHostFree(c->mom_host);
        c->mom_host = nullptr;
        SC_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->mom_host), sizeof(double) * 3 * (size_t)blocks, hipHostMallocDefault));
        c->mom_blocks = blocks;
    }
    // every allocation the side stream's work needs happens here, on the host, before anything is enqueued
    SC_TRY(c->gt_indptr.ensure(sizeof(long long) * (size_t)(n + 1), &c->mem));
    SC_TRY(c->mom_dev.ensure(sizeof(double) * 3 * (size_t)blocks, &c->mem));
    SC_HIP(hipEventRecord(c->mom_ready, c->stream));
    SC_HIP(hipStreamWaitEvent(c->stream_m, c->mom_ready, 0));
    hipStream_t main_stream = c->stream;
    c->stream = c->stream_m;
    int rc = c->gt_valid ? SC_OK : graph_build_transpose(c, false);
    if (rc == SC_OK) {
        hipLaunchKernelGGL(k_moments, dim3(blocks), dim3(256), 0, c->stream, c->g_indptr.as<long long>(),
                           c->g_indices.as<int32_t>(), c->g_data.as<double>(), c->gt_indptr.as<long long>(),
                           c->gt_data.as<double>(), n, c->mom_dev.as<double>());
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(c->mom_host, c->mom_dev.p, sizeof(double) * 3 * (size_t)blocks, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipEventRecord(c->mom_done, c->stream) != hipSuccess) {
            sc_set_error("graph moments: launch on the side stream failed");
            rc = SC_ERR_HIP;
        }
    }
    c->stream = main_stream;
    if (rc != SC_OK) { (void)hipStreamSynchronize(c->stream_m); return rc; }
    c->mom_pending = true;
    return SC_OK;
}

void sc_graph_moments_drain(sc_ctx *c)
{
    if (!c->mom_pending) return;
    (void)hipEventSynchronize(c->mom_done);
    c->mom_pending = false;
}

static int graph_moments(sc_ctx *c, double *s0, double *s1, double *s2)
{
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "no graph set");
    if (c->s0_valid) {  // a function of the graph alone: computed once per graph (the Moran path asks twice)
        *s0 = c->s0; *s1 = c->s1; *s2 = c->s2;
        return SC_OK;
    }
    SC_TRY(sc_graph_moments_begin(c));
    SC_HIP(hipEventSynchronize(c->mom_done));
    c->mom_pending = false;
    const int blocks = (int)ceil_div64(c->g_n, MOM_ROWS_PER_BLOCK);
    const double *h = c->mom_host;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int b = 0; b < blocks; ++b) { a0 += h[3 * b]; a1 += h[3 * b + 1]; a2 += h[3 * b + 2]; }
    *s0 = a0; *s1 = a1 / 2.0; *s2 = a2;
    c->s0 = a0; c->s1 = a1 / 2.0; c->s2 = a2;
    c->s0_valid = true;
    return SC_OK;
}

extern "C" int sc_graph_moments(sc_ctx *c, double *s0, double *s1, double *s2)
{
    SC_REQUIRE(c && s0 && s1 && s2, SC_ERR_INVALID, "null pointer");
    SC_HIP(hipSetDevice(c->device));
    return graph_moments(c, s0, s1, s2);
}

// s0 alone -- all the scoring needs -- without the transpose and the reverse-edge search of the full moments (4.7 ms of the
// step's serial prelude at bench size): the same per-row sums and the same reduction tree as k_moments' first partial,
// so the value is bit-identical to graph_moments' s0.
__global__ __launch_bounds__(256) void k_weight_sum(const long long *__restrict__ indptr, const double *__restrict__ data,
                                                    int64_t n, double *__restrict__ partial)
{
    __shared__ double sh[256];
    double a0 = 0.0;
    const int64_t r0 = (int64_t)blockIdx.x * MOM_ROWS_PER_BLOCK;
    const int64_t r1 = r0 + MOM_ROWS_PER_BLOCK < n ? r0 + MOM_ROWS_PER_BLOCK : n;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
        double rs = 0.0;
        for (long long e = indptr[i]; e < indptr[i + 1]; ++e) rs += data[e];
        a0 += rs;
    }
    sh[threadIdx.x] = a0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

int sc_graph_ensure_s0(sc_ctx *c)
{
    if (c->s0_valid || c->s0_only_valid) return SC_OK;
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "no graph set");
    const int64_t n = c->g_n;
    const int blocks = (int)ceil_div64(n, MOM_ROWS_PER_BLOCK);
    SC_TRY(c->red_tmp.ensure(sizeof(double) * 3 * (size_t)blocks, &c->mem));
    hipLaunchKernelGGL(k_weight_sum, dim3(blocks), dim3(256), 0, c->stream, c->g_indptr.as<long long>(), c->g_data.as<double>(), n,
                       c->red_tmp.as<double>());
    SC_HIP(hipGetLastError());
    std::vector<double> h((size_t)blocks);
    SC_HIP(hipMemcpyAsync(h.data(), c->red_tmp.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    double a0 = 0;
    for (int b = 0; b < blocks; ++b) a0 += h[(size_t)b];
    c->s0 = a0;
    c->s0_only_valid = true;
    return SC_OK;
}

// The same sum in two halves for a caller that must not wait (moran_prepare_early): launch + copy of the per-block
// sums into the caller's pinned host array, and the host-side addition once the caller has synchronised.
int sc_graph_weight_sum_blocks(const sc_ctx *c) { return (int)ceil_div64(c->g_n, MOM_ROWS_PER_BLOCK); }

int sc_graph_weight_sum_launch(sc_ctx *c, double *pinned_out)
{
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "no graph set");
    const int64_t n = c->g_n;
    const int blocks = sc_graph_weight_sum_blocks(c);
    SC_TRY(c->s0_tmp.ensure(sizeof(double) * (size_t)blocks, &c->mem));
    hipLaunchKernelGGL(k_weight_sum, dim3(blocks), dim3(256), 0, c->stream, c->g_indptr.as<long long>(), c->g_data.as<double>(), n,
                       c->s0_tmp.as<double>());
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(pinned_out, c->s0_tmp.p, sizeof(double) * (size_t)blocks, hipMemcpyDeviceToHost, c->stream));
    return SC_OK;
}

void sc_graph_weight_sum_collect(sc_ctx *c, const double *partial, int blocks)
{
    double a0 = 0;
    for (int b = 0; b < blocks; ++b) a0 += partial[b];   // (the order of sc_graph_ensure_s0)
    c->s0 = a0;
    c->s0_only_valid = true;
}

// y[i] = sum_e w[e] * x[col[e]]  for one contiguous vector
__global__ __launch_bounds__(256) void k_spmv_vec(const long long *__restrict__ indptr,
                                                  const int32_t *__restrict__ indices, const double *__restrict__ w,
                                                  const double *__restrict__ x, double *__restrict__ y, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (long long e = indptr[i]; e < indptr[i + 1]; ++e) s = __dadd_rn(s, __dmul_rn(w[e], x[indices[e]]));
    y[i] = s;
}

void sc_launch_spmv_vec(sc_ctx *c, const int64_t *indptr, const int32_t *indices, const double *w, const double *x,
                        double *y, int64_t n)
{
    hipLaunchKernelGGL(k_spmv_vec, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream,
                       (const long long *)indptr, indices, w, x, y, n);
}

// ------------------------------------------------------------------------------------------------
// A9: neighbourhood composition on the active graph's pattern
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_profile(const long long *__restrict__ indptr,
                                                 const int32_t *__restrict__ indices,
                                                 const int32_t *__restrict__ labels, int64_t n, int n_types,
                                                 float *__restrict__ counts, unsigned long long *__restrict__ n_empty)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float *row = counts + i * n_types;
    long long e0 = indptr[i], e1 = indptr[i + 1];
    for (long long e = e0; e < e1; ++e) row[labels[indices[e]]] += 1.0f;
    if (e1 == e0) atomicAdd(n_empty, 1ull);
}

extern "C" int sc_profile_counts(sc_ctx *c, const int32_t *labels, int64_t n, int32_t n_types, float *counts_out,
                                 int64_t *n_empty_out)
{
    SC_REQUIRE(c && labels && counts_out, SC_ERR_INVALID, "sc_profile_counts: null pointer");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->g_n > 0, SC_ERR_STATE, "sc_profile_counts: no graph set");
    SC_REQUIRE(n == c->g_n, SC_ERR_INVALID, "sc_profile_counts: %lld labels for a graph of %lld cells", (long long)n,
               (long long)c->g_n);
    SC_REQUIRE(n_types >= 1 && n_types <= 65536, SC_ERR_INVALID, "n_types out of range");
    for (int64_t i = 0; i < n; ++i)
        SC_REQUIRE(labels[i] >= 0 && labels[i] < n_types, SC_ERR_INVALID, "label %d of cell %lld out of range",
                   labels[i], (long long)i);
    size_t cbytes = sizeof(float) * (size_t)n * (size_t)n_types;
    SC_TRY(c->lee_a.ensure(cbytes, &c->mem));
    SC_TRY(c->lee_pairs.ensure(sizeof(int32_t) * (size_t)n + 16, &c->mem));
    SC_TRY(c->perm_flag.ensure(sizeof(unsigned long long), &c->mem));
    SC_HIP(hipMemcpyAsync(c->lee_pairs.p, labels, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemsetAsync(c->lee_a.p, 0, cbytes, c->stream));
    SC_HIP(hipMemsetAsync(c->perm_flag.p, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_profile, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, c->stream,
                       c->g_indptr.as<long long>(), c->g_indices.as<int32_t>(), c->lee_pairs.as<int32_t>(), n,
                       (int)n_types, c->lee_a.as<float>(), c->perm_flag.as<unsigned long long>());
    SC_HIP(hipGetLastError());
    unsigned long long empty = 0;
    SC_HIP(hipMemcpyAsync(counts_out, c->lee_a.p, cbytes, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(&empty, c->perm_flag.p, sizeof(empty), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    if (n_empty_out) *n_empty_out = (int64_t)empty;
    if (empty) {
        sc_set_error("%llu cells have empty neighborhood profiles", empty);
        return SC_ERR_EMPTY;
    }
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// N3: domain distances (reference src/spatialcore/spatial/distance.py)
// ------------------------------------------------------------------------------------------------

// nearest target of every query: ring walk over the TARGET bin grid (queries may lie outside it).
// Ties go to the lowest target index.  dist = sqrt(fl(fl(dx*dx)+fl(dy*dy))), as cKDTree.query / cdist.
// EXCL: a target whose group code equals the query's exclusion code is skipped (idx -1 / +inf when nothing is left).
template <bool EXCL>
__global__ __launch_bounds__(256) void k_nearest(const double *__restrict__ sx, const double *__restrict__ sy,
                                                 const int32_t *__restrict__ sid,
                                                 const int32_t *__restrict__ bin_start,
                                                 const double *__restrict__ qxy, int64_t n_q, double x0, double y0,
                                                 double h, int nbx, int nby, int32_t *__restrict__ idx_out,
                                                 double *__restrict__ dist_out,
                                                 const int32_t *__restrict__ tgt_code,
                                                 const int32_t *__restrict__ q_excl)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_q) return;
    const double qx = qxy[2 * t], qy = qxy[2 * t + 1];
    const int32_t excl = EXCL ? q_excl[t] : -1;
    const double inv_h = 1.0 / h;
    const int bx = bin_coord(qx, x0, inv_h, nbx), by = bin_coord(qy, y0, inv_h, nby);
    double best = DBL_MAX;
    int best_id = 0x7fffffff;
    const int rmax = (nbx > nby ? nbx : nby);
    const double slack = 1e-9 * h;
    for (int r = 0; r <= rmax; ++r) {
        const int ylo = by - r, yhi = by + r, xlo = bx - r, xhi = bx + r;
        const int cxlo = xlo < 0 ? 0 : xlo, cxhi = xhi >= nbx ? nbx - 1 : xhi;
        for (int yy = (ylo < 0 ? 0 : ylo); yy <= (yhi >= nby ? nby - 1 : yhi); ++yy) {
            const bool full = (yy == ylo) || (yy == yhi);
            for (int seg = 0; seg < (full ? 1 : 2); ++seg) {
                int b0, b1;
                if (full) { b0 = cxlo; b1 = cxhi; }
                else if (seg == 0) { if (xlo < 0) continue; b0 = b1 = xlo; }
                else { if (xhi >= nbx || r == 0) continue; b0 = b1 = xhi; }
                const int s0 = bin_start[yy * nbx + b0], s1 = bin_start[yy * nbx + b1 + 1];
                for (int s = s0; s < s1; ++s) {
                    const double dx = qx - sx[s], dy = qy - sy[s];
                    const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                    const int cid = sid[s];
                    if (EXCL && tgt_code[cid] == excl) continue;
                    if (cand_better(d, cid, best, best_id)) { best = d; best_id = cid; }
                }
            }
        }
        const bool l_out = xlo <= 0, r_out = xhi >= nbx - 1, b_out = ylo <= 0, t_out = yhi >= nby - 1;
        if (l_out && r_out && b_out && t_out) break;
        double m = DBL_MAX;
        if (!l_out) m = fmin(m, qx - (x0 + (double)xlo * h));
        if (!r_out) m = fmin(m, (x0 + (double)(xhi + 1) * h) - qx);
        if (!b_out) m = fmin(m, qy - (y0 + (double)ylo * h));
        if (!t_out) m = fmin(m, (y0 + (double)(yhi + 1) * h) - qy);
        m -= slack;
        if (m > 0.0 && best < m * m) break;
    }
    const bool none = best_id == 0x7fffffff;
    idx_out[t] = none ? -1 : best_id;
    dist_out[t] = none ? HUGE_VAL : __dsqrt_rn(best);
}

static int nearest_impl(sc_ctx *c, const char *who, const double *xy_targets, const int32_t *tgt_code,
                        int64_t n_targets, const double *xy_queries, const int32_t *q_excl, int64_t n_queries,
                        int32_t *idx_out, double *dist_out)
{
    SC_REQUIRE(c && xy_targets && xy_queries && idx_out && dist_out, SC_ERR_INVALID, "%s: null pointer", who);
    SC_REQUIRE(n_targets >= 1 && n_queries >= 1 && n_queries <= 0x7fffffffLL, SC_ERR_INVALID,
               "%s: need at least one target and one query", who);
    SC_HIP(hipSetDevice(c->device));
    c->knn_n = 0;
    c->radius = -1.0;
    SC_TRY(build_bins(c, xy_targets, n_targets, 4.0, 0.0));
    for (int64_t i = 0; i < n_queries; ++i)
        SC_REQUIRE(isfinite(xy_queries[2 * i]) && isfinite(xy_queries[2 * i + 1]), SC_ERR_INVALID,
                   "query coordinate %lld is not finite", (long long)i);
    SC_TRY(c->e_tmp_data.ensure(sizeof(double) * 2 * (size_t)n_queries, &c->mem));
    SC_TRY(c->knn_idx.ensure(sizeof(int32_t) * (size_t)n_queries, &c->mem));
    SC_TRY(c->knn_rd.ensure(sizeof(double) * (size_t)n_queries, &c->mem));
    SC_HIP(hipMemcpyAsync(c->e_tmp_data.p, xy_queries, sizeof(double) * 2 * (size_t)n_queries, hipMemcpyHostToDevice,
                          c->stream));
    const bool excl = tgt_code && q_excl;
    if (excl) {
        SC_TRY(c->e_tmp_indices.ensure(sizeof(int32_t) * (size_t)(n_targets + n_queries), &c->mem));
        SC_HIP(hipMemcpyAsync(c->e_tmp_indices.p, tgt_code, sizeof(int32_t) * (size_t)n_targets, hipMemcpyHostToDevice,
                              c->stream));
        SC_HIP(hipMemcpyAsync(c->e_tmp_indices.as<int32_t>() + n_targets, q_excl, sizeof(int32_t) * (size_t)n_queries,
                              hipMemcpyHostToDevice, c->stream));
    }
    {
        KernelTimerScope ts(c, SC_K_KNN);
        const dim3 grid((unsigned)ceil_div64(n_queries, 256));
        if (excl)
            hipLaunchKernelGGL(k_nearest<true>, grid, dim3(256), 0, c->stream, c->sx.as<double>(), c->sy.as<double>(),
                               c->sid.as<int32_t>(), c->bin_start.as<int32_t>(), c->e_tmp_data.as<double>(), n_queries,
                               c->gx0, c->gy0, c->gh, c->nbx, c->nby, c->knn_idx.as<int32_t>(), c->knn_rd.as<double>(),
                               c->e_tmp_indices.as<int32_t>(), c->e_tmp_indices.as<int32_t>() + n_targets);
        else
            hipLaunchKernelGGL(k_nearest<false>, grid, dim3(256), 0, c->stream, c->sx.as<double>(), c->sy.as<double>(),
                               c->sid.as<int32_t>(), c->bin_start.as<int32_t>(), c->e_tmp_data.as<double>(), n_queries,
                               c->gx0, c->gy0, c->gh, c->nbx, c->nby, c->knn_idx.as<int32_t>(), c->knn_rd.as<double>(),
                               (const int32_t *)nullptr, (const int32_t *)nullptr);
    }
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(idx_out, c->knn_idx.p, sizeof(int32_t) * (size_t)n_queries, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipMemcpyAsync(dist_out, c->knn_rd.p, sizeof(double) * (size_t)n_queries, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

extern "C" int sc_nearest_2d(sc_ctx *c, const double *xy_targets, int64_t n_targets, const double *xy_queries,
                             int64_t n_queries, int32_t *idx_out, double *dist_out)
{
    return nearest_impl(c, "sc_nearest_2d", xy_targets, nullptr, n_targets, xy_queries, nullptr, n_queries, idx_out,
                        dist_out);
}

extern "C" int sc_nearest_excluding_2d(sc_ctx *c, const double *xy_targets, const int32_t *target_code,
                                       int64_t n_targets, const double *xy_queries, const int32_t *query_excluded_code,
                                       int64_t n_queries, int32_t *idx_out, double *dist_out)
{
    SC_REQUIRE(target_code && query_excluded_code, SC_ERR_INVALID, "sc_nearest_excluding_2d: null code array");
    return nearest_impl(c, "sc_nearest_excluding_2d", xy_targets, target_code, n_targets, xy_queries,
                        query_excluded_code, n_queries, idx_out, dist_out);
}

// brute-force pairwise euclidean distances between two point sets, LDS-tiled: block = 256 points of A
// (one per thread, in registers) x the whole of B streamed through LDS in tiles of 1024 points.
// partial[block] = {sum of distances, min distance} for the block's A points.
#define PW_BTILE 1024

__global__ __launch_bounds__(256) void k_pairwise(const double *__restrict__ a, int64_t n_a,
                                                  const double *__restrict__ b, int64_t n_b,
                                                  double *__restrict__ partial)
{
    __shared__ double2 tile[PW_BTILE];
    __shared__ double red_s[256], red_m[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_a;
    const double ax = live ? a[2 * i] : 0.0, ay = live ? a[2 * i + 1] : 0.0;
    double sum = 0.0, mn = DBL_MAX;
    for (int64_t j0 = 0; j0 < n_b; j0 += PW_BTILE) {
        const int cnt = (int)(n_b - j0 < PW_BTILE ? n_b - j0 : PW_BTILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) tile[k] = reinterpret_cast<const double2 *>(b)[j0 + k];
        __syncthreads();
        if (live) {
            for (int k = 0; k < cnt; ++k) {
                const double dx = ax - tile[k].x, dy = ay - tile[k].y;
                const double d = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
                sum += d;
                mn = d < mn ? d : mn;
            }
        }
    }
    red_s[threadIdx.x] = sum;
    red_m[threadIdx.x] = mn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red_s[threadIdx.x] += red_s[threadIdx.x + s];
            red_m[threadIdx.x] = red_m[threadIdx.x + s] < red_m[threadIdx.x] ? red_m[threadIdx.x + s] : red_m[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = red_s[0];
        partial[2 * blockIdx.x + 1] = red_m[0];
    }
}

extern "C" int sc_pairwise_2d(sc_ctx *c, const double *xy_a, int64_t n_a, const double *xy_b, int64_t n_b,
                              double *mean_out, double *min_out)
{
    SC_REQUIRE(c && xy_a && xy_b, SC_ERR_INVALID, "sc_pairwise_2d: null pointer");
    SC_REQUIRE(n_a >= 1 && n_b >= 1, SC_ERR_INVALID, "sc_pairwise_2d: empty point set");
    SC_HIP(hipSetDevice(c->device));
    const int blocks = (int)ceil_div64(n_a, 256);
    SC_TRY(c->e_tmp_data.ensure(sizeof(double) * 2 * (size_t)(n_a + n_b), &c->mem));
    SC_TRY(c->red_tmp.ensure(sizeof(double) * 2 * (size_t)blocks, &c->mem));
    double *da = c->e_tmp_data.as<double>(), *db = da + 2 * n_a;
    SC_HIP(hipMemcpyAsync(da, xy_a, sizeof(double) * 2 * (size_t)n_a, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(db, xy_b, sizeof(double) * 2 * (size_t)n_b, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pairwise, dim3(blocks), dim3(256), 0, c->stream, da, n_a, db, n_b, c->red_tmp.as<double>());
    SC_HIP(hipGetLastError());
    std::vector<double> h((size_t)blocks * 2);
    SC_HIP(hipMemcpyAsync(h.data(), c->red_tmp.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    double s = 0.0, m = DBL_MAX;
    for (int k = 0; k < blocks; ++k) { s += h[2 * k]; m = h[2 * k + 1] < m ? h[2 * k + 1] : m; }
    if (mean_out) *mean_out = s / ((double)n_a * (double)n_b);
    if (min_out) *min_out = m;
    return SC_OK;
}

// Every (source group, target group) block of the all-pairs distance matrix in one launch: the A points are sorted by
// group and cut into chunks of <= 256 points that never straddle a group; workgroup (chunk, t) streams target group
// t through LDS and leaves {sum, min} of its block.  Replaces the reference's per-pair cdist(src, tgt).mean()/.min()
// loops (distance.py:266-270, 340-350, 387-398).
__global__ __launch_bounds__(256) void k_pair_table(const double *__restrict__ a, const int64_t *__restrict__ chunk_a0,
                                                    const int32_t *__restrict__ chunk_cnt,
                                                    const double *__restrict__ b, const int64_t *__restrict__ b_off,
                                                    int n_groups_b, double *__restrict__ partial)
{
    __shared__ double2 tile[PW_BTILE];
    __shared__ double red_s[256], red_m[256];
    const int64_t a0 = chunk_a0[blockIdx.x];
    const bool live = (int)threadIdx.x < chunk_cnt[blockIdx.x];
    const double ax = live ? a[2 * (a0 + threadIdx.x)] : 0.0, ay = live ? a[2 * (a0 + threadIdx.x) + 1] : 0.0;
    const int64_t b0 = b_off[blockIdx.y], b1 = b_off[blockIdx.y + 1];
    double sum = 0.0, mn = DBL_MAX;
    for (int64_t j0 = b0; j0 < b1; j0 += PW_BTILE) {
        const int cnt = (int)(b1 - j0 < PW_BTILE ? b1 - j0 : PW_BTILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) tile[k] = reinterpret_cast<const double2 *>(b)[j0 + k];
        __syncthreads();
        if (live) {
            for (int k = 0; k < cnt; ++k) {
                const double dx = ax - tile[k].x, dy = ay - tile[k].y;
                const double d = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
                sum += d;
                mn = d < mn ? d : mn;
            }
        }
    }
    red_s[threadIdx.x] = sum;
    red_m[threadIdx.x] = mn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red_s[threadIdx.x] += red_s[threadIdx.x + s];
            red_m[threadIdx.x] = red_m[threadIdx.x + s] < red_m[threadIdx.x] ? red_m[threadIdx.x + s] : red_m[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double *o = partial + 2 * ((int64_t)blockIdx.x * n_groups_b + blockIdx.y);
        o[0] = red_s[0];
        o[1] = red_m[0];
    }
}

extern "C" int sc_pair_table_2d(sc_ctx *c, const double *xy_a, const int64_t *a_off, int32_t n_groups_a,
                                const double *xy_b, const int64_t *b_off, int32_t n_groups_b, double *sum_out,
                                double *min_out)
{
    SC_REQUIRE(c && xy_a && xy_b && a_off && b_off && sum_out && min_out, SC_ERR_INVALID, "sc_pair_table_2d: null pointer");
    SC_REQUIRE(n_groups_a >= 1 && n_groups_b >= 1 && n_groups_b <= 65535, SC_ERR_INVALID,
               "sc_pair_table_2d: group counts out of range (%d, %d)", n_groups_a, n_groups_b);
    SC_REQUIRE(a_off[0] == 0 && b_off[0] == 0, SC_ERR_INVALID, "sc_pair_table_2d: offsets must start at 0");
    for (int g = 0; g < n_groups_a; ++g)
        SC_REQUIRE(a_off[g + 1] >= a_off[g], SC_ERR_INVALID, "sc_pair_table_2d: source offsets not monotone");
    for (int g = 0; g < n_groups_b; ++g)
        SC_REQUIRE(b_off[g + 1] >= b_off[g], SC_ERR_INVALID, "sc_pair_table_2d: target offsets not monotone");
    const int64_t n_a = a_off[n_groups_a], n_b = b_off[n_groups_b];
    SC_REQUIRE(n_a >= 1 && n_b >= 1, SC_ERR_INVALID, "sc_pair_table_2d: empty point set");
    SC_HIP(hipSetDevice(c->device));
    std::vector<int64_t> ch0;
    std::vector<int32_t> chn, chg;
    for (int g = 0; g < n_groups_a; ++g)
        for (int64_t p = a_off[g]; p < a_off[g + 1]; p += 256) {
            ch0.push_back(p);
            chn.push_back((int32_t)(a_off[g + 1] - p < 256 ? a_off[g + 1] - p : 256));
            chg.push_back(g);
        }
    const size_t chunks = ch0.size();
    SC_REQUIRE(chunks <= 0x7fffffffULL, SC_ERR_INVALID, "sc_pair_table_2d: too many points");
    SC_TRY(c->e_tmp_data.ensure(sizeof(double) * 2 * (size_t)(n_a + n_b), &c->mem));
    SC_TRY(c->e_tmp_indptr.ensure(sizeof(int64_t) * (chunks + (size_t)n_groups_b + 1), &c->mem));
    SC_TRY(c->e_tmp_indices.ensure(sizeof(int32_t) * chunks, &c->mem));
    SC_TRY(c->red_tmp.ensure(sizeof(double) * 2 * chunks * (size_t)n_groups_b, &c->mem));
    double *da = c->e_tmp_data.as<double>(), *db = da + 2 * n_a;
    int64_t *d_ch0 = c->e_tmp_indptr.as<int64_t>(), *d_boff = d_ch0 + chunks;
    SC_HIP(hipMemcpyAsync(da, xy_a, sizeof(double) * 2 * (size_t)n_a, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(db, xy_b, sizeof(double) * 2 * (size_t)n_b, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(d_ch0, ch0.data(), sizeof(int64_t) * chunks, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(d_boff, b_off, sizeof(int64_t) * (size_t)(n_groups_b + 1), hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemcpyAsync(c->e_tmp_indices.p, chn.data(), sizeof(int32_t) * chunks, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pair_table, dim3((unsigned)chunks, (unsigned)n_groups_b), dim3(256), 0, c->stream, da, d_ch0,
                       c->e_tmp_indices.as<int32_t>(), db, d_boff, (int)n_groups_b, c->red_tmp.as<double>());
    SC_HIP(hipGetLastError());
    std::vector<double> h(2 * chunks * (size_t)n_groups_b);
    SC_HIP(hipMemcpyAsync(h.data(), c->red_tmp.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    // chunks of a group are reduced in ascending order: run-to-run reproducible
    for (int64_t k = 0; k < (int64_t)n_groups_a * n_groups_b; ++k) { sum_out[k] = 0.0; min_out[k] = HUGE_VAL; }
    for (size_t ch = 0; ch < chunks; ++ch)
        for (int t = 0; t < n_groups_b; ++t) {
            if (b_off[t + 1] == b_off[t]) continue;
            const size_t o = (size_t)chg[ch] * n_groups_b + t;
            sum_out[o] += h[2 * (ch * n_groups_b + t)];
            const double m = h[2 * (ch * n_groups_b + t) + 1];
            if (m < min_out[o]) min_out[o] = m;
        }
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// N4 (extension, no reference counterpart): cell-type pair counts over the graph's edges under label
// permutations.  counts[p][a][b] = #{edges i -> j : lab[perm_p[i]] == a and lab[perm_p[j]] == b}
// (p == n_perm: identity, i.e. the observed counts).  One workgroup = one permutation x one cell
// range; the T x T histogram lives in LDS (integer atomics: deterministic), then is added to global.
// ------------------------------------------------------------------------------------------------

#define ENR_EDGES_PER_BLOCK 65536

// labp[p][r] = lab[perm_p[order[r]]] (p == n_perm: the identity, i.e. the observed labels): the permuted label of
// the cell at position r of the graph's spatially sorted processing order.  The edge kernel then needs ONE byte per
// edge end from an n-byte array (instead of a 4-byte index gather followed by a byte gather), and the two ends of an
// edge -- spatial neighbours -- sit at nearby positions: the row's k + 1 bytes come from one or two cache lines.
__global__ __launch_bounds__(256) void k_enrich_relabel(const unsigned char *__restrict__ lab,
                                                        const int32_t *__restrict__ order,
                                                        const int32_t *__restrict__ perm, int64_t pstride, int n_perm,
                                                        int64_t n, int64_t lstride, unsigned char *__restrict__ labp)
{
    const int p = blockIdx.y;
    const int32_t *prow = p < n_perm ? perm + (int64_t)p * pstride : nullptr;
    const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (r0 >= n) return;
    unsigned char *dst = labp + (int64_t)p * lstride;
    uint32_t v = 0;
    for (int k = 0; k < 4 && r0 + k < n; ++k) {
        const int32_t cell = order[r0 + k];
        v |= (uint32_t)lab[prow ? prow[cell] : cell] << (8 * k);
    }
    if (r0 + 4 <= n) *reinterpret_cast<uint32_t *>(dst + r0) = v;
    else for (int k = 0; r0 + k < n; ++k) dst[r0 + k] = (unsigned char)(v >> (8 * k));
}

// counts[p][a][b] += #{edges of the block : label(row) = a, label(column) = b}.  One thread per EDGE (coalesced reads of
// the two relabelled end positions; consecutive workgroups are the permutations of ONE edge block, which L2 serves),
// `copies` private T x T histograms per workgroup (lane l adds into copy l % copies, copy stride odd: with ~20 skewed
// cell types most of a wavefront's 64 LDS atomics would otherwise hit a handful of addresses and banks and serialise).
__global__ __launch_bounds__(256) void k_enrich(const int32_t *__restrict__ erow_r, const int32_t *__restrict__ ecol_r,
                                                int64_t nnz, const unsigned char *__restrict__ labp, int64_t lstride,
                                                int n_types, int copies, int cstride, unsigned long long *__restrict__ counts)
{
    extern __shared__ unsigned int hist[];
    const int p = blockIdx.x;
    const int tt = n_types * n_types;
    for (int k = threadIdx.x; k < cstride * copies; k += 256) hist[k] = 0;
    __syncthreads();
    const unsigned char *lp = labp + (int64_t)p * lstride;
    unsigned int *mine = hist + (threadIdx.x & (copies - 1)) * cstride;
    const int64_t e0 = (int64_t)blockIdx.y * ENR_EDGES_PER_BLOCK;
    const int64_t e1 = e0 + ENR_EDGES_PER_BLOCK < nnz ? e0 + ENR_EDGES_PER_BLOCK : nnz;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256)
        atomicAdd(&mine[(int)lp[erow_r[e]] * n_types + lp[ecol_r[e]]], 1u);
    __syncthreads();
    unsigned long long *out = counts + (int64_t)p * tt;
    for (int k = threadIdx.x; k < tt; k += 256) {
        unsigned int v = 0;
        for (int c = 0; c < copies; ++c) v += hist[c * cstride + k];
        if (v) atomicAdd(&out[k], (unsigned long long)v);
    }
}

extern "C" int sc_enrichment_counts(sc_ctx *c, const int32_t *labels, int64_t n, int32_t n_types, int64_t n_perm,
                                    int64_t perm_row0, int64_t *counts_out)
{
    SC_REQUIRE(c && labels && counts_out, SC_ERR_INVALID, "sc_enrichment_counts: null pointer");
    SC_HIP(hipSetDevice(c->device));
    if (n_perm > 0) SC_TRY(sc_perm_forward_ensure(c));
    SC_REQUIRE(c->g_n > 0 && n == c->g_n, SC_ERR_STATE, "sc_enrichment_counts: graph missing or size mismatch");
    SC_REQUIRE(n_types >= 1 && n_types <= 96, SC_ERR_INVALID, "sc_enrichment_counts: n_types must be 1..96");
    SC_REQUIRE(n_perm >= 0 && perm_row0 >= 0, SC_ERR_INVALID, "sc_enrichment_counts: negative size");
    SC_REQUIRE(n_perm + 1 <= 65535 && ceil_div64(c->g_nnz, ENR_EDGES_PER_BLOCK) <= 65535, SC_ERR_INVALID,
               "sc_enrichment_counts: at most 65534 permutations per call (got %lld) and 4.2e9 edges; call it per batch of the table",
               (long long)n_perm);
    if (n_perm > 0)
        SC_REQUIRE(c->p_n == n && perm_row0 + n_perm <= c->p_count, SC_ERR_STATE,
                   "sc_enrichment_counts: needs permutation rows [%lld, %lld)", (long long)perm_row0,
                   (long long)(perm_row0 + n_perm));
    std::vector<unsigned char> lab8((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        SC_REQUIRE(labels[i] >= 0 && labels[i] < n_types, SC_ERR_INVALID, "label %d of cell %lld out of range",
                   labels[i], (long long)i);
        lab8[(size_t)i] = (unsigned char)labels[i];
    }
    const size_t tt = (size_t)n_types * n_types;
    const size_t out_bytes = sizeof(unsigned long long) * tt * (size_t)(n_perm + 1);
    const int64_t lstride = align_up64(n, 16);
    SC_TRY(c->lee_pairs.ensure((size_t)n + 16, &c->mem));
    SC_TRY(c->lee_b.ensure(out_bytes, &c->mem));
    SC_TRY(c->lee_a.ensure((size_t)lstride * (size_t)(n_perm + 1), &c->mem));   // permuted label vectors
    SC_HIP(hipMemcpyAsync(c->lee_pairs.p, lab8.data(), (size_t)n, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemsetAsync(c->lee_b.p, 0, out_bytes, c->stream));
    SC_TRY(sc_graph_ensure_order(c));
    hipLaunchKernelGGL(k_enrich_relabel, dim3((unsigned)ceil_div64(n, 1024), (unsigned)(n_perm + 1)), dim3(256), 0, c->stream,
                       c->lee_pairs.as<unsigned char>(), c->g_order.as<int32_t>(),
                       c->perm.as<int32_t>() + perm_row0 * c->p_stride, c->p_stride, (int)n_perm, n, lstride,
                       c->lee_a.as<unsigned char>());
    if (c->g_nnz > 0) {
        int copies = 16;
        const int cstride = (int)tt | 1;   // odd: copy c starts at a different LDS bank
        while (copies > 1 && (size_t)copies * cstride > 12288) copies >>= 1;   // <= 48 KB of LDS per workgroup
        dim3 grid((unsigned)(n_perm + 1), (unsigned)ceil_div64(c->g_nnz, ENR_EDGES_PER_BLOCK));
        hipLaunchKernelGGL(k_enrich, grid, dim3(256), sizeof(unsigned int) * cstride * copies, c->stream,
                           c->g_erow_r.as<int32_t>(), c->g_indices_r.as<int32_t>(), c->g_nnz, c->lee_a.as<unsigned char>(),
                           lstride, (int)n_types, copies, cstride, c->lee_b.as<unsigned long long>());
    }
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(counts_out, c->lee_b.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// The whole label-permutation test of one rank's range of counter-based permutations in ONE call (r03): the sums the
// p-values and z-scores need are accumulated on the device, and the generation of batch b + 1 (stage B's swaps are
// latency-bound: 21 ms per 512 permutations of 1M cells) runs on a second stream beside the edge counting of batch b
// (37 ms per 512): the relabelled label vectors are the only thing the counting reads, so the table is free again as
// soon as k_enrich_relabel is through.
// ------------------------------------------------------------------------------------------------

// ---- sixteen permutations per edge (r03) ------------------------------------------------------------------------------
// k_enrich re-reads the 8 bytes of every edge once per permutation (123 GB of L2 traffic per 512 permutations of a 30M-edge
// graph: what bounds it, 37 ms).  Here the permuted labels of SIXTEEN permutations of a cell are one 16-byte word (at the
// cell's position in the graph's processing order), so an edge's indices are read once per 16 permutations and its two
// label words bring 16 label pairs.  Each of the 16 permutations has its own T x T histogram in LDS; lane l handles them
// in the rotated order (s + l) % 16, so that a wavefront's 64 atomics of one step spread over 16 histograms (what
// k_enrich's 16 private copies did).
#define ENR16_MAX_TT 768   // 16 histograms of <= 768 bins: 48 KB of LDS (T <= 27)

// lab16[g][rank[cell]] = the labels of `cell` under permutations 16 g .. 16 g + 15 (rows clamped to rows - 1)
__global__ __launch_bounds__(256) void k_enrich_relabel16(const unsigned char *__restrict__ lab, const int32_t *__restrict__ rank,
                                                          const int32_t *__restrict__ perm, int64_t pstride, int rows, int64_t n,
                                                          uint4 *__restrict__ lab16)
{
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= n) return;
    const int g = blockIdx.y;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int row = 16 * g + p < rows ? 16 * g + p : rows - 1;
        w[p >> 2] |= (uint32_t)lab[perm[(int64_t)row * pstride + cell]] << (8 * (p & 3));
    }
    lab16[(int64_t)g * n + rank[cell]] = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void k_enrich16(const int32_t *__restrict__ erow_r, const int32_t *__restrict__ ecol_r,
                                                  int64_t nnz, const uint4 *__restrict__ lab16, int64_t n, int n_types,
                                                  int hstride, int rows, unsigned long long *__restrict__ counts)
{
    extern __shared__ unsigned int hist[];   // [16][hstride]
    const int g = blockIdx.x;
    const int tt = n_types * n_types;
    for (int k = threadIdx.x; k < 16 * hstride; k += 256) hist[k] = 0;
    __syncthreads();
    const uint4 *lp = lab16 + (int64_t)g * n;
    const int rot = threadIdx.x & 15;
    const int64_t e0 = (int64_t)blockIdx.y * ENR_EDGES_PER_BLOCK;
    const int64_t e1 = e0 + ENR_EDGES_PER_BLOCK < nnz ? e0 + ENR_EDGES_PER_BLOCK : nnz;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const uint4 a = lp[erow_r[e]], b = lp[ecol_r[e]];
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int p = (s + rot) & 15;
            const uint32_t la = (aw[p >> 2] >> (8 * (p & 3))) & 0xffu, lb = (bw[p >> 2] >> (8 * (p & 3))) & 0xffu;
            atomicAdd(&hist[p * hstride + (int)la * n_types + (int)lb], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 16 * tt; k += 256) {
        const int p = k / tt, bin = k - p * tt;
        const unsigned int v = hist[p * hstride + bin];
        if (v && 16 * g + p < rows) atomicAdd(&counts[(int64_t)(16 * g + p) * tt + bin], (unsigned long long)v);
    }
}

// sums[0][k] += sum_p (cnt_p[k] - obs[k]), sums[1][k] += sum_p (cnt_p[k] - obs[k])^2, sums[2][k] += #{p : cnt_p[k] >= obs[k]}
__global__ __launch_bounds__(256) void k_enrich_sums(const unsigned long long *__restrict__ counts,
                                                     const unsigned long long *__restrict__ obs, int n_perm, int tt,
                                                     long long *__restrict__ sums)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= tt) return;
    const long long o = (long long)obs[k];
    long long s1 = 0, s2 = 0, ge = 0;
    for (int p = 0; p < n_perm; ++p) {
        const long long d = (long long)counts[(int64_t)p * tt + k] - o;
        s1 += d;
        s2 += d * d;
        ge += d >= 0 ? 1 : 0;
    }
    sums[k] += s1;
    sums[tt + k] += s2;
    sums[2 * tt + k] += ge;
}

int sc_perm_counter_rows(sc_ctx *c, uint64_t seed, int64_t n, int64_t p_first, int64_t n_perm, hipStream_t s);   // sc_permgen.hip

extern "C" int sc_enrichment_counter(sc_ctx *c, const int32_t *labels, int64_t n, int32_t n_types, uint64_t seed,
                                     int64_t p_first, int64_t n_perm, int64_t batch, int64_t *observed_out, int64_t *sums_out)
{
    SC_REQUIRE(c && labels && observed_out && sums_out, SC_ERR_INVALID, "sc_enrichment_counter: null pointer");
    SC_HIP(hipSetDevice(c->device));
    SC_REQUIRE(c->g_n > 0 && n == c->g_n, SC_ERR_STATE, "sc_enrichment_counter: graph missing or size mismatch");
    SC_REQUIRE(n_types >= 1 && n_types <= 96, SC_ERR_INVALID, "sc_enrichment_counter: n_types must be 1..96");
    SC_REQUIRE(n_perm >= 0 && p_first >= 0 && batch >= 1 && batch <= 65534, SC_ERR_INVALID, "sc_enrichment_counter: bad sizes");
    SC_REQUIRE(ceil_div64(c->g_nnz, ENR_EDGES_PER_BLOCK) <= 65535, SC_ERR_INVALID, "sc_enrichment_counter: more than 4.2e9 edges");
    std::vector<unsigned char> lab8((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        SC_REQUIRE(labels[i] >= 0 && labels[i] < n_types, SC_ERR_INVALID, "label %d of cell %lld out of range",
                   labels[i], (long long)i);
        lab8[(size_t)i] = (unsigned char)labels[i];
    }
    if (batch > n_perm) batch = n_perm > 0 ? n_perm : 1;
    const int tt = n_types * n_types;
    const int64_t lstride = align_up64(n, 16);
    const size_t cnt_bytes = sizeof(unsigned long long) * (size_t)tt * (size_t)batch;
    SC_TRY(c->lee_pairs.ensure((size_t)n + 16, &c->mem));
    SC_TRY(c->lee_b.ensure(cnt_bytes + sizeof(unsigned long long) * (size_t)tt * 4, &c->mem));   // counts | observed | 3 sums
    SC_TRY(c->lee_a.ensure((size_t)lstride * (size_t)align_up64(batch, 16), &c->mem));   // (also the 16-wide form: n x 16 B per 16 rows)
    unsigned long long *d_cnt = c->lee_b.as<unsigned long long>(), *d_obs = d_cnt + (size_t)tt * batch;
    long long *d_sums = reinterpret_cast<long long *>(d_obs + tt);
    SC_TRY(sc_graph_ensure_order(c));
    if (n_perm > 0) SC_TRY(sc_perm_alloc(c, n, batch));
    if (!c->stream3) SC_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    SC_HIP(hipMemcpyAsync(c->lee_pairs.p, lab8.data(), (size_t)n, hipMemcpyHostToDevice, c->stream));
    SC_HIP(hipMemsetAsync(d_obs, 0, sizeof(unsigned long long) * (size_t)tt * 4, c->stream));
    int copies = 16;
    const int cstride = tt | 1;   // odd: copy c starts at a different LDS bank
    while (copies > 1 && (size_t)copies * cstride > 12288) copies >>= 1;   // <= 48 KB of LDS per workgroup
    const unsigned eblocks = (unsigned)ceil_div64(c->g_nnz, ENR_EDGES_PER_BLOCK);
    const bool wide = tt <= ENR16_MAX_TT && !getenv("SC_ENRICH_NARROW");   // sixteen permutations per edge (k_enrich16)
    const int hstride = tt | 1;
    auto count = [&](int rows, const int32_t *table, unsigned long long *out) {
        if (wide && table)
            hipLaunchKernelGGL(k_enrich_relabel16, dim3((unsigned)ceil_div64(n, 256), (unsigned)((rows + 15) / 16)), dim3(256), 0,
                               c->stream, c->lee_pairs.as<unsigned char>(), c->g_rank.as<int32_t>(), table, c->p_stride, rows, n,
                               c->lee_a.as<uint4>());
        else
            hipLaunchKernelGGL(k_enrich_relabel, dim3((unsigned)ceil_div64(n, 1024), (unsigned)rows), dim3(256), 0, c->stream,
                               c->lee_pairs.as<unsigned char>(), c->g_order.as<int32_t>(), table, c->p_stride, table ? rows : 0, n,
                               lstride, c->lee_a.as<unsigned char>());
    };
    auto edges = [&](int rows, unsigned long long *out, bool from_table) {
        if (c->g_nnz <= 0) return;
        if (wide && from_table)
            hipLaunchKernelGGL(k_enrich16, dim3((unsigned)((rows + 15) / 16), eblocks), dim3(256), sizeof(unsigned int) * 16 * hstride,
                               c->stream, c->g_erow_r.as<int32_t>(), c->g_indices_r.as<int32_t>(), c->g_nnz, c->lee_a.as<uint4>(), n,
                               (int)n_types, hstride, rows, out);
        else
            hipLaunchKernelGGL(k_enrich, dim3((unsigned)rows, eblocks), dim3(256), sizeof(unsigned int) * cstride * copies, c->stream,
                               c->g_erow_r.as<int32_t>(), c->g_indices_r.as<int32_t>(), c->g_nnz, c->lee_a.as<unsigned char>(),
                               lstride, (int)n_types, copies, cstride, out);
    };
    // observed labels: one "permutation" without a table
    count(1, nullptr, d_obs);
    edges(1, d_obs, false);
    SC_HIP(hipGetLastError());
    const int64_t batches = n_perm > 0 ? ceil_div64(n_perm, batch) : 0;
    std::vector<hipEvent_t> ev((size_t)batches * 2, nullptr);
    int rc = SC_OK;
    auto generate = [&](int64_t b) -> int {   // batch b's rows into the table, on the generator's stream
        const int64_t p0 = b * batch, cnt = p0 + batch < n_perm ? batch : n_perm - p0;
        SC_TRY(sc_perm_counter_rows(c, seed, n, p_first + p0, cnt, c->stream3));
        SC_HIP(hipEventCreateWithFlags(&ev[(size_t)(2 * b)], hipEventDisableTiming));
        SC_HIP(hipEventRecord(ev[(size_t)(2 * b)], c->stream3));
        return SC_OK;
    };
    if (batches > 0) {
        SC_HIP(hipStreamSynchronize(c->stream));   // (the table may still be read by an earlier call's kernels)
        rc = generate(0);
    }
    for (int64_t b = 0; b < batches && rc == SC_OK; ++b) {
        const int64_t p0 = b * batch;
        const int cnt = (int)(p0 + batch < n_perm ? batch : n_perm - p0);
        if (hipStreamWaitEvent(c->stream, ev[(size_t)(2 * b)], 0) != hipSuccess) { rc = SC_ERR_HIP; break; }
        count(cnt, c->perm.as<int32_t>(), d_cnt);
        if (hipEventCreateWithFlags(&ev[(size_t)(2 * b + 1)], hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ev[(size_t)(2 * b + 1)], c->stream) != hipSuccess ||
            hipStreamWaitEvent(c->stream3, ev[(size_t)(2 * b + 1)], 0) != hipSuccess) { rc = SC_ERR_HIP; break; }
        if (b + 1 < batches) rc = generate(b + 1);   // beside the edge counting of batch b
        if (rc != SC_OK) break;
        if (hipMemsetAsync(d_cnt, 0, cnt_bytes, c->stream) != hipSuccess) { rc = SC_ERR_HIP; break; }
        edges(cnt, d_cnt, true);
        hipLaunchKernelGGL(k_enrich_sums, dim3((unsigned)ceil_div64(tt, 256)), dim3(256), 0, c->stream, d_cnt, d_obs, cnt, tt, d_sums);
    }
    if (rc == SC_ERR_HIP) sc_set_error("sc_enrichment_counter: event plumbing failed");
    (void)hipStreamSynchronize(c->stream3);
    (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    if (rc != SC_OK) return rc;
    SC_HIP(hipGetLastError());
    c->p_count = 0;   // the table holds the last batch only: not a table later calls may rely on
    std::vector<unsigned long long> host((size_t)tt * 4);
    SC_HIP(hipMemcpy(host.data(), d_obs, sizeof(unsigned long long) * (size_t)tt * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < tt; ++k) observed_out[k] = (int64_t)host[(size_t)k];
    for (int k = 0; k < 3 * tt; ++k) sums_out[k] = (int64_t)host[(size_t)tt + k];
    return SC_OK;
}
