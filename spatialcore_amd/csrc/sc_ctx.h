// Internal definitions shared by the translation units of libspatialcore_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/spatialcore_hip.h"

#define SC_TILE 16  // genes per tile: one 128-byte fp64 row per cell and tile
#ifndef PERM_CHUNK
#define PERM_CHUNK 128  // permutations per pipeline stage (generator scan -> swaps -> scoring)
#endif

void sc_set_error(const char *fmt, ...);

#define SC_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) {                                                             \
            sc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                         __LINE__);                                                          \
            return e__ == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_HIP;                   \
        }                                                                                    \
    } while (0)

#define SC_TRY(call)              \
    do {                          \
        int rc__ = (call);        \
        if (rc__ != SC_OK) return rc__; \
    } while (0)

#define SC_REQUIRE(cond, code, ...)    \
    do {                               \
        if (!(cond)) {                 \
            sc_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

// Device buffer that only ever grows; freed with the context.
struct DBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes, int64_t *acct);
    void release(int64_t *acct);
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct KTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double ms = 0.0;
    int64_t launches = 0;
};

struct PermPipe;   // a generator job in flight (sc_moran.hip: sc_moran_seeded_begin .. _finish)

struct sc_ctx {
    int device = 0;
    PermPipe *pipe = nullptr;        // the generator / consumer pipeline begun by sc_moran_seeded_begin, until _finish
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // fused permutation/Moran pipeline: rejection scan runs ahead here
    hipStream_t stream3 = nullptr;  // ... and the Fisher-Yates swaps of the scanned chunk here
    hipStream_t stream4 = nullptr;  // ... alternating with this one
    hipStream_t stream_pg[4] = {};    // block-parallel scan: the chip prepares blocks here ahead of the chain
    hipStream_t stream_px = nullptr;     // ... and verifies + expands a finished chunk here, beside the next chunk's chain
    hipStream_t stream_fr = nullptr;     // ... and the fresh-table helpers of a chain launch live here (k_fresh)
    hipStream_t stream_out = nullptr;    // r04: result copies that run beside a pipeline, issued by a helper thread (sc_local_moran_seeded)
    hipEvent_t pg_ev[34] = {};        // rings of events between the preparation and the chain launches + start marker
    int pg_mode = 0;                  // 0 auto, 1 sequential scan only, 2 fault injection (tests)
    bool pg_streams_serial = false;   // a wait on a hand-over word gave up once: the streams of this process do not run
                                      // concurrently (profiler that serialises kernels, shared hardware queues) --
                                      // later jobs take the sequential scan at once instead of waiting 10 s again
    bool pg_probed = false;           // the stream-concurrency probe ran (once per context, before the first block-parallel job)
    std::string pg_note;              // why the generator left the block-parallel form, if it did (sc_ctx_permgen_note)
    std::string pg_form;              // scratch of sc_ctx_permgen_form
    int pg_ahead = 1;                 // launch units the preparation runs ahead of the chain (callers that share the chip raise it)
    int64_t pg_jobs_parallel = 0, pg_jobs_sequential = 0, pg_fallbacks = 0;  // generator jobs by scan form
    int64_t pg_blocks_prepared = 0, pg_blocks_chain = 0;  // block-parallel jobs: blocks resolved by table lookup / by the chain workgroup
    int64_t mem = 0;  // bytes allocated through DBuf
    bool timing = true;
    KTimer timers[SC_K_COUNT_];

    // ---- points (kNN / radius) ----
    int64_t pts_n = 0;
    DBuf px, py;        // SoA coordinates in original order
    DBuf sx, sy, sid;   // coordinates / original ids sorted by bin
    DBuf bin_start;     // [nbins + 1]
    DBuf bin_keys, bin_keys2, sid2, cub_tmp;
    int nbx = 0, nby = 0;
    double gx0 = 0, gy0 = 0, gh = 0;
    int64_t knn_n = 0;
    int knn_k = 0;
    DBuf knn_idx, knn_rd;  // [n][k] device result of the last sc_knn_2d
    hipEvent_t knn_done = nullptr;      // ... recorded behind a search whose result was not fetched (sc_knn_fetch)
    DBuf knn_hd, knn_hi;   // k > 32: the per-query candidate heaps, [slot][query]
    double radius = -1.0;
    DBuf rad_indptr;  // [n+1] int64 of the last radius count

    // ---- graph (CSR, rows sorted by column) + transpose ----
    int64_t g_n = 0, g_nnz = 0;
    DBuf g_indptr, g_indices, g_data;
    double g_uniform_w = 0.0;   // > 0: every stored weight equals this value (kNN graphs: 1 / k); 0: weights differ
    int64_t g_deg_max = 0;      // longest row
    bool g_regular = false;     // every row has g_deg_max entries
    bool gt_valid = false;
    DBuf gt_indptr, gt_indices, gt_data, gt_cursor;
    // the full moments (transpose + reverse-edge search: 6 ms at 1M x 15) may be in flight on a side stream, begun by the
    // scoring's set-up so that they leave its serial prelude (sc_graph.hip: graph_moments_begin / graph_moments)
    hipStream_t stream_m = nullptr;
    hipEvent_t mom_ready = nullptr, mom_done = nullptr;
    bool mom_pending = false;
    double *mom_host = nullptr;      // pinned: per-block partials of k_moments
    int mom_blocks = 0;
    DBuf gt_tmp, mom_dev;            // the transpose's own scan scratch (cub_tmp belongs to the neighbour search); k_moments' partials
    bool s0_valid = false, s0_only_valid = false;   // all three moments / s0 alone (k_weight_sum) are those of the active graph
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;  // graph moments (valid with s0_valid)
    // a processing order with spatial locality for kernels that read neighbours' rows (local Moran): the bin-sorted
    // order of the points the graph was built from (identity for a graph of unknown geometry).  Results never depend on it.
    DBuf g_order, g_rank, g_indices_r, g_w32, g_erow_r;   // [n] sorted position -> cell, [n] cell -> position, [nnz] rank of the column, float weights, [nnz] rank of the row
    bool g_order_captured = false, g_order_ready = false;

    // ---- expression tiles ----
    int64_t e_n = 0, e_genes = 0, e_tiles = 0;
    int e_dtype = SC_F64;  // dtype of the matrix the tiles were loaded from (the reference's float32 paths depend on it)
    DBuf X, Z, Lag;      // [tile][cell][16] fp64: raw, centred/standardised, lagged
    DBuf X32;            // the raw values again in the narrowest exact type, one 128-byte row per cell and gene group:
                         // 32 float (float32-exact values), 64 uint16 (counts < 65536) or 128 uint8 (counts < 256) per row
    int nib_groups = 0;         // > 0: the last moran_prepare built the 4-bit source (256 nibble slots per row) with this many row groups
    DBuf nib_map;               // ... its slot map: [padded genes] slot of the gene's high nibble (-1: none) | the genes that have one
    bool lag_u16 = false;       // the last moran_prepare left Lag as 16-bit neighbour sums ([group][cell][64 words]; uint8 source only)
    int narrow_bits = 64;       // element width of the narrow copy the last moran_prepare built: 8, 16, 32 (64: none, the fp64 Z tiles are gathered)
    int n_cus = 0;              // compute units of the device (filled on first use)
    int score_leave_cus = 0;    // compute units the persistent scoring kernel leaves empty (> 0 only while a generator runs beside it)
    int source_bits_min = 8;    // (4: the nibble source, opt-in -- measured slower at the step level) narrowest source the scoring kernels may gather (sc_ctx_set_moran_source_bits)
    int last_source_bits = 0;   // ... and what the last scoring launch gathered (64 = fp64 kernel)
    DBuf e_tmp_indptr, e_tmp_indices, e_tmp_data, e_colmap;
    DBuf g_mean, g_var, g_z2, g_scale, g_Inum, g_I, red_tmp;  // per padded gene
    DBuf g_slag;         // sum_j lag_g[j] per padded gene (lattice genes: the exact integer sum of the neighbour sums)
    // Integer-lattice genes (DESIGN.md "Ties"): integer counts on a graph whose weights are all equal.  Their statistic
    // is scored as the exact integer T_p = sum_j S_j x[inv_p(j)] (S = unweighted neighbour sums; every product and partial
    // sum is an integer < 2^53, so the fp64 arithmetic is exact in any order) and #{T_p >= T_obs} is decided on integers.
    DBuf g_xsum, g_flags, g_xmax;      // per padded gene: raw column sum, value-class bits (k_gene_stats), largest count
    DBuf g_lat, g_meanc, g_seff, g_corr, g_thr;  // lattice flag (0 / 1), mean used for centring (0 for lattice genes),
                                                 // sims = seff * (sum - corr), threshold the count compares against
    DBuf sims_raw;                     // the sums themselves (lattice genes: T_p), same layout as sims
    bool lat_any = false;

    // ---- permutation table ----
    int64_t p_n = 0, p_count = 0, p_stride = 0;  // row stride in elements (multiple of 32)
    DBuf perm;
    DBuf perm_flag;
    DBuf inv;                      // inverse permutations, same layout as perm (rows valid on demand)
    int64_t inv_rows_valid = 0;    // leading rows of c->inv known to be the inverses of the active table's rows
    bool perm_forward_valid = true; // c->perm holds the active table (false: only its inverse, c->inv, was generated)
    bool perm_bijective = false;   // the active table is known to hold true permutations
    bool perm_checked = false;     // ... or was checked and is not
    DBuf pg_J, pg_raw, pg_out, pg_bits, pg_enter, pg_sblk;  // device generator scratch: accepted j per step, raw 32-bit stream
    DBuf pg_flags;       // hand-over words between the chain workgroup and the preparation launches (sc_permgen.hip)
    DBuf pg_desc, pg_tbits, pg_events, pg_hard;  // block-parallel scan: per-block descriptors + gap-transfer tables (ring), hard flags
    DBuf pg_fresh;                               // ... fresh tables for the ends of the permutations: control block, descriptors, tables
    DBuf pg_seglist;                             // ... per unit in flight: [count | first blocks of the segments k_phi_compose builds]
    DBuf pg_seg, pg_ctbits, pg_segmode;          // ... segments of prepared blocks: descriptors + composed tables (ring), per-block mode

    // ---- Moran / Lee work buffers ----
    DBuf partial, sims, counts, sim_sum, sim_sumsq;
    DBuf lee_a, lee_b, lee_out, lee_pairs;
    DBuf lee_U, lee_Zc, lee_Uc, lee_part, lee_obs, lee_cnt, lee_rowmap, lee_lperm;  // batched Lee (sc_lee.hip)
    bool lm_direct = false;  // local Moran per-cell counts: the r01 one-kernel form instead of the two-phase sorted form (A/B)
    bool lm_valid = false;   // z / lag / counts of the last sc_local_moran are still resident
    int64_t lm_perms = 0;
    DBuf lm_out;             // local Moran: row-major staging of one output array for the helper thread's device-to-host copies
    DBuf lm_ys;              // local Moran: the permuted z rows (or uint8 code rows) of a batch of permutations, in the graph's processing order
    DBuf lm_tab;             // local Moran, code rows: z and w z per (gene, value)
    // first half of the Moran preparation, enqueued ahead of the generator by sc_moran_seeded_begin (sc_moran.hip)
    bool prep_early = false;         // ... is in flight / done for the resident expression and graph
    void *prep_host = nullptr;       // pinned: [weight-sum partials | xsum | flags | xmax]
    size_t prep_host_cap = 0;
    int prep_s0_blocks = 0;
    DBuf s0_tmp;
    DBuf np_cnt, np_comp, np_leaves, np_leafsum;  // numpy-order column sums: block counts, compacted values, leaf table, leaf sums
};

struct KernelTimerScope {
    sc_ctx *c;
    int id;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    KernelTimerScope(sc_ctx *ctx, int kid, hipStream_t on = nullptr);
    ~KernelTimerScope();
};

// One numpy-exact permutation job on the device (sc_permgen.hip): begin -> {scan, swap} per chunk
// of permutations -> finish.  Scan state lives on the device so chunks chain without host syncs.
struct PermJob {
    int64_t n = 0, n_perm = 0;
    uint64_t h = 0;            // 1 if the generator starts with a buffered 32-bit half
    uint64_t st_hi = 0, st_lo = 0, inc_hi = 0, inc_lo = 0;
    uint32_t buffered = 0;
    uint64_t total_steps = 0;
    uint64_t hi = 0;           // raw indices [0, hi) hold stream draws
    bool trivial = false;      // n == 1
    double draws_per_perm = 0; // expectation
    int64_t p_done = 0;        // permutations covered by the scan launches so far
    int64_t chunk_no = 0;
    bool phi = false;          // block-parallel scan in use
    uint64_t B_done = 0;       // blocks covered by the chain launches so far
    uint64_t unit_start[8] = {};  // first block of the last launch units (ring)
    int64_t unit_no = 0;
    int64_t gate_seen[4] = {};   // per preparation stream: the "units completed by the chain" count its last gate waited for
    int ahead = 1;             // units prepared ahead of the chain
};
// A generator job whose chunks are (being) enqueued on the generator's streams while the consumer catches up.
struct PermPipe {
    PermJob job;
    std::vector<int64_t> bounds;       // chunk k = permutations [bounds[k], bounds[k + 1])
    std::vector<hipEvent_t> ev;        // per chunk: scanned, swapped
    int table = 0;                     // 0 rows, 1 inverse rows only, 2 both
    int64_t n = 0, n_perm = 0, enqueued = 0;   // generator chunks enqueued so far
    uint64_t state0[6] = {};           // the generator state the job started from (a sequential rerun starts there again)
};
void sc_perm_pipe_abort(sc_ctx *c);    // drain and drop c->pipe (no results)
#define SC_PERMGEN_RETRY 1000  // internal: the block-parallel scan failed its verification, rerun sequentially
bool permgen_is_block_parallel(const sc_ctx *c, int64_t n);  // which scan form a job of length n takes
int permgen_begin(sc_ctx *c, const uint64_t *state6, int64_t n, int64_t n_perm, PermJob *job, hipStream_t s);
int permgen_scan_chunk(sc_ctx *c, PermJob *job, int64_t p1, hipStream_t s, hipStream_t post, hipEvent_t done);
int permgen_swap_chunk(sc_ctx *c, PermJob *job, int64_t p0, int64_t p1, hipStream_t s, bool inverse, int pw_req);
bool permgen_can_swap_inverse(int64_t n);
int sc_perm_forward_ensure(sc_ctx *c);  // materialise c->perm from c->inv after a pipeline that only made the inverse
int permgen_finish(sc_ctx *c, PermJob *job, uint64_t *state6);
int sc_perm_alloc(sc_ctx *c, int64_t n, int64_t n_perm);
// generator / consumer pipeline (sc_moran.hip): table 0 = permutation rows, 1 = inverse rows only, 2 = both
int sc_permgen_profile(unsigned long long *out32, int reset);   // development builds (-DPHI_PROFILE): computed blocks by class
int sc_perm_pipeline(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm, int table,
                     const std::function<int()> &after_first, const std::function<int(int64_t, int64_t)> &score);

int sc_timer_collect(sc_ctx *c);
int sc_expr_zscores(sc_ctx *c);  // Z = (X - mean) / population sd per gene (0 for zero variance), variances in g_var
int sc_lag_tiles(sc_ctx *c, const DBuf &indptr, const DBuf &indices, const DBuf &data, const double *Z, double *out);

// 64-bit masks and per-lane variable shifts built from 32-bit instructions whose shift amounts are IN RANGE by
// construction.  r02 finding (ISA diff in r03): the one instruction that produced wrong values
// next to kernels of other hardware queues was a v_lshlrev_b64 whose per-lane amount register held 0 - r (the compiler's
// form of (64 - r) & 63, legal only through the instruction's implicit 6-bit masking of the amount); the right shift by
// r itself, in the same kernel, never failed.  Device code therefore keeps per-lane 64-bit shifts out of the ISA: these
// helpers compile to v_lshlrev_b32 / v_lshrrev_b32 / v_alignbit_b32 with amounts masked to [0, 31] in the source.
#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t sc_low_mask64(uint32_t d)   // the d low bits set, d in [0, 64]
{
    const uint32_t part = (1u << (d & 31u)) - 1u;
    const uint32_t lo = d >= 32u ? 0xffffffffu : part;
    const uint32_t hi = d >= 64u ? 0xffffffffu : (d > 32u ? part : 0u);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t sc_bit64(uint32_t pos)      // 1 << pos, pos in [0, 63]
{
    const uint32_t b = 1u << (pos & 31u);
    const uint32_t lo = (pos & 32u) ? 0u : b, hi = (pos & 32u) ? b : 0u;
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t sc_shr64(uint64_t x, uint32_t pos)   // x >> pos, pos in [0, 63]
{
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    if (pos & 32u) { xl = xh; xh = 0u; }
    const uint32_t k = pos & 31u;
    return ((uint64_t)(xh >> k) << 32) | __builtin_amdgcn_alignbit(xh, xl, k);
}
#endif

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up64(int64_t a, int64_t b) { return ceil_div64(a, b) * b; }

// ---- implemented across translation units ----
int sc_graph_ensure_transpose(sc_ctx *c);
int sc_graph_ensure_s0(sc_ctx *c);
int sc_graph_weight_sum_blocks(const sc_ctx *c);
int sc_graph_weight_sum_launch(sc_ctx *c, double *pinned_out);                 // s0 without a host wait: launch + copy ...
void sc_graph_weight_sum_collect(sc_ctx *c, const double *partial, int blocks);  // ... and the addition after the caller's synchronisation
int sc_graph_moments_begin(sc_ctx *c);   // start the full moments on the side stream (no host wait); collected by sc_graph_moments
void sc_graph_moments_drain(sc_ctx *c);  // wait for a begun computation (before the graph's arrays are replaced)
int sc_graph_capture_order(sc_ctx *c, int64_t n);  // called by the graph setters
int sc_graph_ensure_order(sc_ctx *c);           // rank / relabelled columns / float weights, built on first use
int sc_perm_generate_device(sc_ctx *c, uint64_t *state6, int64_t n, int64_t n_perm);
void sc_launch_spmv_vec(sc_ctx *c, const int64_t *indptr, const int32_t *indices, const double *w,
                        const double *x, double *y, int64_t n);
