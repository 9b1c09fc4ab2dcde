/*
 * spatialcore_hip.h -- C ABI of libspatialcore_hip.so (MI355X / gfx950).
 *
 * The reference (mcap91/SpatialCore, /root/reference) is pure Python: it has no FFI of its own.
 * The drop-in boundary is the Python surface of `spatialcore.spatial`
 * (reference src/spatialcore/spatial/__init__.py:11-52); spatialcore_amd/spatial/ mirrors those
 * functions and reaches the GPU only through the entry points declared here (ctypes).
 * Each entry point names the reference call site(s) whose arithmetic it replaces;
 * AC = src/spatialcore/spatial/autocorrelation.py, NB = src/spatialcore/spatial/neighborhoods.py.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns an int status (SC_OK = 0) and never
 *    throws; sc_last_error() returns the text of the last failure on the calling thread.
 *  - the caller owns all host buffers; the library owns device memory behind the opaque handle.
 *  - one handle = one GPU + one HIP stream; a handle is not thread-safe.
 *  - host arrays are C-contiguous.  "gene-major" means [gene][cell].
 */
#ifndef SPATIALCORE_HIP_H
#define SPATIALCORE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_OK 0
#define SC_ERR_INVALID 1  /* bad argument            -> ValueError in the Python shim */
#define SC_ERR_STATE 2    /* call order / missing setup -> RuntimeError                */
#define SC_ERR_HIP 3      /* HIP runtime failure     -> RuntimeError                   */
#define SC_ERR_NOMEM 4    /* device/host allocation  -> MemoryError                    */
#define SC_ERR_EMPTY 5    /* empty neighbourhoods (NB:253-260) -> ValueError           */

/* expression value types accepted by sc_expr_* */
#define SC_F32 0
#define SC_F64 1

/* kernel ids for sc_ctx_kernel_time */
#define SC_K_MORAN_PERM 0 /* gather-dot permutation kernel (the metric's dominant kernel) */
#define SC_K_LAG 1
#define SC_K_KNN 2
#define SC_K_PERMGEN 3
#define SC_K_LEE_PERM 4
#define SC_K_PERM_SCAN 5 /* rejection scan of the permutation generator (chain of exact block states + verification) */
#define SC_K_PERM_SWAP 6 /* Fisher-Yates application, one workgroup (or wavefront) per permutation */
#define SC_K_COUNT_ 8

typedef struct sc_ctx sc_ctx;

int sc_version(void);
const char *sc_last_error(void);
int sc_device_count(int *count);

/* ---- context ------------------------------------------------------------------------------ */
int sc_ctx_create(int device, sc_ctx **out);
int sc_ctx_destroy(sc_ctx *ctx);
int sc_ctx_sync(sc_ctx *ctx);
/* Accumulated HIP-event time (ms) and launch count of one kernel family since the last reset. */
int sc_ctx_kernel_time(sc_ctx *ctx, int kernel_id, double *ms, int64_t *launches);
int sc_ctx_reset_timers(sc_ctx *ctx);
/* Enable/disable per-launch HIP-event timing (default on; events are recorded on the ctx stream). */
int sc_ctx_set_timing(sc_ctx *ctx, int enabled);
int sc_ctx_device_mem(sc_ctx *ctx, int64_t *bytes_in_use);
/* Development aid (scripts/concurrency_probe4.py, scripts/pipeline_soak.py): raw bytes of one of the generator's device buffers
 * (0 J, 1 raw stream, 2 accept masks, 3 entering counts, 4 block states, 5 scan state, 6 table, 7 inverse table). */
int sc_debug_copy(sc_ctx *ctx, int which, int64_t offset_bytes, void *out, int64_t bytes);
/* The permutation kernels of sc_moran / sc_moran_seeded gather the narrowest EXACT copy of the raw expression values,
 * one 128-byte row per cell and gene group: uint8 when every value is an integer count in [0, 255] (128 genes per
 * row; only when every loaded gene is a lattice gene, below), uint16 for counts up to 65535 (64 genes), else float32
 * when every value is a float32 (32 genes), else the fp64 rows of the centred tiles (16 genes).  Every width rebuilds
 * the same operands and adds the same products in the same order: a gene's statistics do not depend on the width, i.e.
 * not on the genes it is loaded with.  Lattice genes (integer counts on a graph whose weights are all equal, e.g. kNN):
 * scored as the exact integer sum_j S_j x[inv_p(j)] (S = unweighted neighbour sums), #{sims >= I} decided on integers.
 * r04, opt-in (min_bits = 4): 4-bit slots, 256 per row, when that takes fewer rows than uint8 (a count below 16 is one
 * nibble, a count up to 255 the exact sum of two nibble pseudo-genes, x = lo + 16 hi; integer sums) -- same results bit for
 * bit; measured: the kernel alone 15 % faster, the pipelined step slower (heavier set-up), hence not the default.
 * min_bits (4, 8, 16, 32 or 64; default 8) forbids the narrower sources; sc_ctx_moran_source_bits reports what the last
 * scoring call gathered. */
int sc_ctx_set_moran_source_bits(sc_ctx *ctx, int min_bits);
int sc_ctx_moran_source_bits(sc_ctx *ctx, int *bits);
/* Element width of the STREAMED operand (the lag rows, `W @ z` of AC:307,864) of the last scoring call: 16 when an
 * all-count uint8 batch on an equal-weight graph kept its neighbour sums as the 16-bit integers they are (converted to
 * the same fp64 values inside the kernel), else 64.  Diagnostic: bench.py prices the kernel's compulsory bytes with it. */
int sc_ctx_moran_lag_bits(sc_ctx *ctx, int *bits);
/* ... and the number of 128-byte rows it gathers per (permutation, cell): the gene groups of the source width in use. */
int sc_ctx_moran_row_groups(sc_ctx *ctx, int *groups);
/* How the device generator of sc_perm_generate / sc_moran_seeded resolves numpy's rejection stream
 * (results are identical in every mode): 0 = automatic (block-parallel scan for n >= 131072, verified on the
 * device, sequential scan otherwise or when the verification fails), 1 = sequential scan only,
 * 2 = inject a fault into the block-parallel scan (exercises the verification + fallback; tests only). */
int sc_ctx_set_permgen_mode(sc_ctx *ctx, int mode);
/* Why the generator is not using its block-parallel form, "" when it is.  The form orders its kernels through words in
 * device memory and needs its streams on different hardware queues (the library asks for GPU_MAX_HW_QUEUES=24 when it is
 * loaded before the HIP runtime initialises -- a host application that initialised HIP first keeps its own setting);
 * the context probes that once (5 rounds of 5-ms waits at worst), falls back to the sequential scan with identical
 * results, and leaves the reason here.  sc_ctx_set_permgen_mode re-arms the probe. */
int sc_ctx_permgen_note(sc_ctx *ctx, const char **message);
/* The same question asked up front instead of discovered inside the first job (spatialcore_amd.init()): probe the
 * context's generator streams now; *concurrent = 1 when they overlap (the block-parallel form is available),
 * *hw_queues_requested = the GPU_MAX_HW_QUEUES value in this process's environment (0: unset).  The reference has no
 * counterpart (single process, no device: AC:580 n_jobs=1); SURVEY section 5 "failure detection": a degradation must
 * surface like an error does. */
int sc_ctx_probe_streams(sc_ctx *ctx, int *concurrent, int *hw_queues_requested);
/* The scan form a permutation job of length n takes on this context right now, in words -- "block-parallel",
 * "sequential (...)" or "sequential: <reason>" -- for the provenance entry the drop-in functions append
 * (src/spatialcore/core/metadata.py:49-77).  Valid until the next call of this function on the context. */
int sc_ctx_permgen_form(sc_ctx *ctx, int64_t n, const char **form);
/* Completed generator jobs by scan form, how often the block-parallel form failed its verification and the
 * job was rerun sequentially (0 unless mode 2 injected a fault), and for the block-parallel jobs (failed ones
 * included) the 16384-draw blocks resolved by a prepared table lookup / computed by the chain workgroup itself. */
int sc_ctx_permgen_stats(sc_ctx *ctx, int64_t *jobs_parallel, int64_t *jobs_sequential, int64_t *fallbacks,
                         int64_t *blocks_prepared, int64_t *blocks_chain);

/* ---- A1: kNN graph ------------------------------------------------------------------------
 * Replaces sklearn NearestNeighbors(k+1, "ball_tree").kneighbors + "drop column 0" (AC:393-401),
 * squidpy's NearestNeighbors(k).kneighbors() behind sq.gr.spatial_neighbors (AC:565-570) and
 * scipy cKDTree.query(k+1) + "drop == i" (NB:213-228).
 * xy: [n][2] float64.  idx_out: [n][k] int32, neighbours ordered by (squared distance, index);
 * squared distance = fl(fl(dx*dx) + fl(dy*dy)) in fp64 (no FMA contraction), as the tree codes
 * compute it.  Self is excluded BY INDEX unless include_self (then k counts self, AC:398).
 * rdist_out (nullable): [n][k] squared distances.  The result also stays on the device and can be
 * turned into the active graph with sc_graph_from_knn.
 * LIFETIME: with both output pointers null the call returns WITHOUT waiting for the device -- the upload of `xy` is
 * then merely enqueued, so `xy` must stay valid and unmodified until the next call on this context that waits
 * (sc_knn_fetch, sc_ctx_sync, any call that returns results); kernel errors of the search surface there as well. */
int sc_knn_2d(sc_ctx *ctx, const double *xy, int64_t n, int k, int include_self,
              int32_t *idx_out, double *rdist_out);
/* The neighbour lists of the last sc_knn_2d that was called WITHOUT output arrays (it then returns without waiting),
 * copied out on a stream of their own: neither behind the work the context has been given since, nor in its way.  For a
 * caller that fetches the lists (squidpy's obsp side effects, AC:565-570) from one thread while another uploads the
 * expression.  Either pointer may be null. */
int sc_knn_fetch(sc_ctx *ctx, int32_t *idx_out, double *rdist_out);

/* ---- A2: radius graph ---------------------------------------------------------------------
 * Replaces cKDTree.query_ball_point(coords, r) with self removed (NB:241-244): closed ball
 * fl(dx*dx+dy*dy) <= fl(r*r).  Two-pass: count fills indptr_out[n+1]; fill writes nnz indices,
 * ascending within each row.  The coordinates of the count call stay resident for the fill call. */
int sc_radius_count_2d(sc_ctx *ctx, const double *xy, int64_t n, double radius, int64_t *indptr_out);
int sc_radius_fill_2d(sc_ctx *ctx, int64_t nnz, int32_t *indices_out);

/* ---- A3: graph / weights ------------------------------------------------------------------
 * The active graph is a general CSR with fp64 weights (user graphs from use_existing_graph,
 * AC:558-561, are general).  sc_graph_from_knn builds it on the device from the last sc_knn_2d
 * result with w = weight for every edge and rows sorted by column index, i.e. the matrix that
 * AC:402-413 (weight = (double)(float)(1/k)) or squidpy + l1 row normalisation (weight = 1.0/k)
 * produce.  sc_graph_get copies the device CSR back (indices ascending per row). */
int sc_graph_set_csr(sc_ctx *ctx, const int64_t *indptr, const int32_t *indices, const double *data,
                     int64_t n, int64_t nnz);
int sc_graph_from_knn(sc_ctx *ctx, double weight);
int sc_graph_get(sc_ctx *ctx, int64_t *indptr_out, int32_t *indices_out, double *data_out);
int sc_graph_shape(sc_ctx *ctx, int64_t *n, int64_t *nnz);
/* A6 [upstream squidpy _g_moments]: s0 = sum w, s1 = 1/2 sum (w_ij + w_ji)^2, s2 = sum_i (row_i + col_i)^2 */
int sc_graph_moments(sc_ctx *ctx, double *s0, double *s1, double *s2);

/* ---- expression operands ------------------------------------------------------------------
 * Select n_genes columns of a cells x n_vars matrix and lay them out on the device as 16-gene
 * tiles [tile][cell][16] fp64 (one 128-byte row per cell and tile).  Replaces
 * `adata[:, genes]` + densify + cast (AC:573 then scanpy's float64 cast; AC:1118-1123).
 * CSR: indptr int64[n+1], indices int32[nnz], data f32/f64.  Dense: row-major, ld = n_vars. */
int sc_expr_set_csr(sc_ctx *ctx, const int64_t *indptr, const int32_t *indices, const void *data,
                    int dtype, int64_t n, int64_t n_vars, const int32_t *gene_cols, int64_t n_genes);
int sc_expr_set_dense(sc_ctx *ctx, const void *data, int dtype, int64_t n, int64_t n_vars,
                      const int32_t *gene_cols, int64_t n_genes);
/* per-gene mean and population variance of the loaded columns (fp64) */
int sc_expr_stats(sc_ctx *ctx, double *mean_out, double *var_out);

/* ---- A4: permutation source ---------------------------------------------------------------
 * numpy-exact `rng.permutation(n)` stream (AC:839,879; AC:1109,324; AC:1367,1404; squidpy
 * _score_helper): state6 = {state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger} of a PCG64
 * Generator, updated in place.  sc_perm_numpy_host fills a host table [n_perm][n] int32 (no GPU);
 * sc_perm_generate makes the same table resident on the device as the active permutation table
 * (perm_out nullable: copy back).  sc_perm_set uploads a caller-made table. */
int sc_perm_numpy_host(uint64_t *state6, int64_t n, int64_t n_perm, int32_t *perm_out);
int sc_perm_generate(sc_ctx *ctx, uint64_t *state6, int64_t n, int64_t n_perm, int32_t *perm_out);
int sc_perm_set(sc_ctx *ctx, const int32_t *perm, int64_t n, int64_t n_perm);
/* EXTENSION (SURVEY 8(e) "alternative", H2) for the paths WITHOUT reference seed semantics (label-permutation
 * enrichment, shared-permutation Lee grids; the reference has neither: NB:48-296, AC:1109-1148): counter-based
 * permutations.  Permutation p is a pure function of (seed, p) -- Fisher-Yates with j = Lemire-bounded(Philox4x32-10(key =
 * seed words, counter = (i, retry, p))) -- so ranks / batches take disjoint ranges [p_first, p_first + n_perm) and merge
 * integer counts (sc_allreduce_sum_i64).  sc_perm_generate_counter makes them the resident table (rows 0 .. n_perm-1);
 * sc_perm_counter_host is the same definition on the host (no GPU). */
int sc_perm_generate_counter(sc_ctx *ctx, uint64_t seed, int64_t n, int64_t p_first, int64_t n_perm, int32_t *perm_out);
int sc_perm_counter_host(uint64_t seed, int64_t n, int64_t p_first, int64_t n_perm, int32_t *perm_out);

/* ---- A3 + A5 + A6 + A7: global Moran's I --------------------------------------------------
 * Replaces sq.gr.spatial_autocorr(mode="moran", n_perms=P, seed=seed) (AC:576-583):
 * z = x - mean; lag = W z (row-sequential fp64); I = n/s0 * sum z*lag / sum z^2;
 * sims[p][g] = n/s0 * sum_i z_g[i] * lag_g[perm_p[i]] / sum z_g^2  (== scoring g[perm_p, :]).
 * Needs: active graph, expression, and (if n_perm > 0) an active permutation table.
 * Outputs (host, all nullable except I_out): I_out[G]; sims_out[n_perm][G];
 * count_ge_out[G] = #{p : sims[p][g] >= I[g]}; sim_sum_out[G], sim_sumsq_out[G] = sum and sum of
 * squares of sims over p (for squidpy's pval_z_sim / var_sim). */
int sc_moran(sc_ctx *ctx, int64_t n_perm, double *I_out, double *sims_out, int64_t *count_ge_out,
             double *sim_sum_out, double *sim_sumsq_out);
/* Same result as sc_perm_generate(state6, n_cells, n_perm) followed by sc_moran(n_perm), but the
 * two are pipelined: the generator's rejection scan runs ahead on its own streams (a chain of exact block states
 * on a few CUs the scoring stream leaves free) while the rest of the chip applies the swaps and scores the
 * previous chunk of permutations.
 * state6 is advanced exactly as n_perm calls of rng.permutation(n_cells) would; the table stays
 * resident as the active permutation table (for a float32 matrix the pipeline only builds its inverse, the same
 * Fisher-Yates transpositions in ascending order; the rows themselves are materialised when a later call needs them). */
int sc_moran_seeded(sc_ctx *ctx, uint64_t *state6, int64_t n_perm, double *I_out, double *sims_out,
                    int64_t *count_ge_out, double *sim_sum_out, double *sim_sumsq_out);
/* sc_moran_seeded in two halves: _begin needs nothing but n_cells and the generator state and returns at once with the
 * whole generator job enqueued, so the longest chain of the call runs while the caller builds the graph and uploads the
 * expression (the reference's call order -- sq.gr.spatial_neighbors AC:565-570, then the matrix AC:573, then
 * spatial_autocorr AC:576-583 -- put 60 ms of graph + PCIe work in front of it); _finish prepares the operands and
 * scores chunk after chunk.  Same results and final state6 as sc_moran_seeded; _abort drops a begun job.
 * ahead_chunks: generator chunks (of <= 128 permutations) enqueued before _begin returns: 0 = all (callers with an upload
 * in front of _finish), n >= 2 = that many (each costs ~3.5 ms of host time; _finish enqueues the rest as it scores). */
int sc_moran_seeded_begin(sc_ctx *ctx, const uint64_t *state6, int64_t n_cells, int64_t n_perm, int64_t ahead_chunks);
int sc_moran_seeded_finish(sc_ctx *ctx, uint64_t *state6, double *I_out, double *sims_out, int64_t *count_ge_out,
                           double *sim_sum_out, double *sim_sumsq_out);
int sc_moran_seeded_abort(sc_ctx *ctx);

/* ---- A8: Lee's L ---------------------------------------------------------------------------
 * Replaces _compute_lees_l_core (AC:307-332) for a list of (x, y) pairs over the loaded genes.
 * Standardisation is the population-std z-score of AC:1126-1143.  L = sum_i zx_i * (W zy)_i.
 * L_perm[p] = sum_j (W^T zx)_j * zy[perm[j]] (== shuffling zy and redoing W @ zy).  Pair q uses
 * rows [perm_offset[q], perm_offset[q]+n_perm) of the active permutation table (the reference
 * draws a fresh block of P permutations per non-degenerate pair from ONE stream, AC:1109-1148);
 * pairs with perm_offset[q] < 0 (zero variance, AC:1129-1140) get L = 0, count = n_perm.
 * Outputs: L_out[n_pairs], count_abs_ge_out[n_pairs] = #{p : |L_perm| >= |L|}, L_perm_out
 * (nullable) [n_pairs][n_perm]. */
int sc_lee(sc_ctx *ctx, const int32_t *pair_x, const int32_t *pair_y, const int64_t *perm_offset,
           int64_t n_pairs, int64_t n_perm, double *L_out, int64_t *count_abs_ge_out,
           double *L_perm_out);
/* sc_lee_seeded: the whole pair loop of lees_l (AC:1113-1155) in one call.  Every distinct gene is standardised once;
 * the observed L[x][y] = sum_i z_x[i] (W z_y)[i] of all pairs is a dense contraction over the cells on the fp64 matrix
 * cores (v_mfma_f64_16x16x4_f64 per 16 x 16 genes); each pair with two live genes then draws its OWN block of n_perm
 * numpy-exact permutations, in pair order, from the one generator `state6` (pairs with a zero-variance gene draw
 * nothing: L = 0, count = n_perm, AC:1129-1140), scored as sum_j (W^T z_x)[j] z_y[perm[j]] while the generator runs.
 * count_abs_ge_out[q] = #{p : |L_perm| >= |L|}; L_perm_out (optional) is [n_pairs][n_perm].  state6 is advanced. */
int sc_lee_seeded(sc_ctx *ctx, uint64_t *state6, const int32_t *pair_x, const int32_t *pair_y, int64_t n_pairs,
                  int64_t n_perm, double *L_out, int64_t *count_abs_ge_out, double *L_perm_out);
/* sc_lee_observed_f32: the reference's OWN observed L for a float32 matrix (AC:1118-1146, 307-315 computed in
 * float32): numpy's pairwise float32 sums for mean / std / L, float32 standardisation, scipy's float32 csr_matvec for
 * the lag -- the same roundings in the same order, evaluated in parallel (the summation tree depends on n alone).
 * L32_out[q] is 0 for a pair with a zero-std gene; mean32_out / sd32_out (optional) are [n_pairs][2] (x, y). */
int sc_lee_observed_f32(sc_ctx *ctx, const int32_t *pair_x, const int32_t *pair_y, int64_t n_pairs, float *L32_out,
                        float *mean32_out, float *sd32_out);
/* sc_lee_shared (EXTENSION, no reference site: the reference draws fresh permutations per pair, AC:1109-1148): the
 * full grid genes_x x genes_y under ONE shared block of n_perm numpy-exact permutations.  The permutation statistics
 * of the grid are then n_perm dense contractions over the cells with a row-gathered operand -- fp64 matrix cores.
 * L_out / count_abs_ge_out are [n_x][n_y]; L_perm_out (optional) is [n_perm][n_x][n_y].  state6 is advanced. */
int sc_lee_shared(sc_ctx *ctx, uint64_t *state6, const int32_t *genes_x, int32_t n_x, const int32_t *genes_y,
                  int32_t n_y, int64_t n_perm, double *L_out, int64_t *count_abs_ge_out, double *L_perm_out);

/* ---- N1: Local Moran's I ---------------------------------------------------------------------
 * Replaces the batch body of local_morans_i (AC:845-896) for the loaded genes (= one batch), with the
 * reference's float32 arithmetic: z = (float(x) - mean32) / sd32, lag = W32 @ z (row-sequential
 * float32), I = z * lag, and per cell #{p : |Zs * (W @ Zs)| >= |I|} with Zs = z[perm_p], accumulated
 * on the fly instead of the reference's (P, N, B) tensor + Python loops.  Permutations are rows
 * [perm_row0, perm_row0 + n_perm) of the active table (the reference continues ONE stream across
 * batches, AC:839,879).  Outputs are row-major [n_cells][n_genes]; zero_var_out[g] = 1 where the
 * float32 sd is 0 (AC:825-830; the caller blanks those columns, AC:902-906). */
int sc_local_moran(sc_ctx *ctx, int64_t n_perm, int64_t perm_row0, float *z_out, float *lag_out,
                   float *I_out, int32_t *count_out, uint8_t *zero_var_out);
/* The same batch with its permutations drawn here: n_perm numpy-exact permutations from state6 (exactly the rows
 * sc_perm_generate would leave; state6 advanced the same way: AC:839,879, one stream across the batches), generated
 * chunk by chunk while the per-cell counts of the finished chunks are taken.  Count data (every value an integer below
 * 32) travels through the counts as uint8 code rows with z looked up per (gene, value); outputs identical either way. */
int sc_local_moran_seeded(sc_ctx *ctx, uint64_t *state6, int64_t n_perm, float *z_out, float *lag_out,
                          float *I_out, int32_t *count_out, uint8_t *zero_var_out);
/* Per-cell finalisation of the last sc_local_moran on the device (its z / lag / counts stay resident; count_out
 * above may then be null), replacing the p-value, FDR and quadrant passes of AC:888-934 over (n_cells x n_genes):
 * sc_local_moran_hist returns hist[g][c] = cells of gene g with permutation count c, c = 0..n_perm, from which the
 * caller builds per-gene lookup tables with the reference's own expressions (p = float32((c+1)/(P+1)), BH / Bonferroni
 * adjusted values per level); sc_local_moran_classify applies them: p = p_tab[g][count], p_adj = padj_tab[g][count],
 * quadrant int8 (AC:219-265: 1 HH, 2 LL, 3 HL, 4 LH by the signs of z and lag; 0 where p_adj >= alpha or
 * force_ns[g]).  Without permutations the tables / p outputs are null and quadrants come from the signs alone. */
int sc_local_moran_hist(sc_ctx *ctx, int64_t *hist_out);
int sc_local_moran_classify(sc_ctx *ctx, const float *p_tab, const float *padj_tab, const uint8_t *force_ns,
                            float alpha, float *p_out, float *padj_out, int8_t *quadrant_out);

/* ---- N2: Local Lee's L -----------------------------------------------------------------------
 * Replaces the per-pair body of lees_l_local (AC:1373-1413): population-std z-scores of the two
 * loaded genes, lag = W z_y, L_local = z_x * lag, and (n_perm > 0) the per-cell count
 * #{p : |float32(z_x[i] * (W z_y[perm_p])[i])| >= |L_local[i]|} over rows [perm_row0, +n_perm). */
int sc_lee_local(sc_ctx *ctx, int32_t gene_x, int32_t gene_y, int64_t n_perm, int64_t perm_row0,
                 double *zx_out, double *lag_out, double *L_local_out, int32_t *count_out);
/* The whole pair body of lees_l_local (AC:1373-1413) as one pipeline behind the numpy-exact generator: equal, bit for bit
 * and in the generator state it leaves, to sc_perm_generate(state6, n_cells, n_perm_global + n_perm_local), sc_lee for the
 * pair on rows [0, n_perm_global) (AC:1394-1400: global L and the count of |L_perm| >= |L|) and sc_lee_local on rows
 * [n_perm_global, +n_perm_local) (AC:1402-1413) -- with the sums and counts taken chunk by chunk while the generator runs.
 * Both genes need a positive variance (the reference skips such pairs before it gets here, AC:1380-1392). */
int sc_lee_local_seeded(sc_ctx *ctx, uint64_t *state6, int32_t gene_x, int32_t gene_y, int64_t n_perm_global,
                        int64_t n_perm_local, double *L_out, int64_t *count_abs_ge_out, double *zx_out, double *lag_out,
                        double *L_local_out, int32_t *count_out);

/* ---- N3: domain distances (reference src/spatialcore/spatial/distance.py) -----------------------
 * sc_nearest_2d replaces cKDTree(target_coords).query(source_coords, k=1) (distance.py:222-232,
 * 359-367): index (into the target array, lowest index on ties) and euclidean distance
 * sqrt(fl(fl(dx*dx)+fl(dy*dy))) of the nearest target for every query point.
 * sc_pairwise_2d replaces cdist(a, b).mean() / .min() (distance.py:269, 349, 397): LDS-tiled
 * brute force over all |a| x |b| pairs. */
int sc_nearest_2d(sc_ctx *ctx, const double *xy_targets, int64_t n_targets, const double *xy_queries,
                  int64_t n_queries, int32_t *idx_out, double *dist_out);
int sc_pairwise_2d(sc_ctx *ctx, const double *xy_a, int64_t n_a, const double *xy_b, int64_t n_b,
                   double *mean_out, double *min_out);
/* sc_nearest_excluding_2d: the same search, but a target whose group code equals the query's excluded code is
 * skipped (idx -1 / +inf when no target is left): the per-cell "nearest other centroid" loop of
 * distance.py:305-326 (own domain skipped when source and target column coincide) as ONE launch.
 * sc_pair_table_2d: all (source group, target group) blocks of the pairwise distance matrix in one launch --
 * points arrive sorted by group with offsets a_off[n_groups_a + 1], b_off[n_groups_b + 1]; sum_out / min_out are
 * row-major [n_groups_a][n_groups_b] (sum of distances, minimum distance; +inf / 0 for an empty block).
 * Replaces the nested `for src ... for tgt ... cdist(...).mean() / .min()` loops of distance.py:329-350, 376-398. */
int sc_nearest_excluding_2d(sc_ctx *ctx, const double *xy_targets, const int32_t *target_code, int64_t n_targets,
                            const double *xy_queries, const int32_t *query_excluded_code, int64_t n_queries,
                            int32_t *idx_out, double *dist_out);
int sc_pair_table_2d(sc_ctx *ctx, const double *xy_a, const int64_t *a_off, int32_t n_groups_a, const double *xy_b,
                     const int64_t *b_off, int32_t n_groups_b, double *sum_out, double *min_out);

/* ---- A9: neighbourhood composition --------------------------------------------------------
 * Replaces the per-cell Python counting loops of NB:226-251 on the active graph's pattern:
 * counts_out[n][n_types] float32 = number of neighbours of each label.  Rows with no neighbour
 * give SC_ERR_EMPTY (NB:253-260) and n_empty_out is set. */
int sc_profile_counts(sc_ctx *ctx, const int32_t *labels, int64_t n, int32_t n_types,
                      float *counts_out, int64_t *n_empty_out);

/* ---- N4 (extension; nothing in the reference computes this) -----------------------------------
 * Cell-type pair counts over the active graph's edges for the observed labels and under label
 * permutations labels[perm_p] (rows [perm_row0, perm_row0 + n_perm) of the active table):
 * counts_out[(p * T + a) * T + b] = #{edges i -> j : type(i) = a, type(j) = b}; p = n_perm holds the
 * observed counts.  Integer arithmetic, exact. */
int sc_enrichment_counts(sc_ctx *ctx, const int32_t *labels, int64_t n, int32_t n_types, int64_t n_perm,
                         int64_t perm_row0, int64_t *counts_out);
/* The same test for ONE RANK'S RANGE [p_first, p_first + n_perm) of counter-based permutations (sc_perm_generate_counter's
 * definition) in one call: batches of `batch` permutations are generated on a second stream beside the edge counting of
 * the batch before, and only the integer sums come back -- observed_out[T*T]; sums_out[3][T*T] = sum_p (count_p - observed),
 * sum_p (count_p - observed)^2, #{p : count_p >= observed}: exact and order-free, ranks add theirs (sc_allreduce_sum_i64). */
int sc_enrichment_counter(sc_ctx *ctx, const int32_t *labels, int64_t n, int32_t n_types, uint64_t seed, int64_t p_first,
                          int64_t n_perm, int64_t batch, int64_t *observed_out, int64_t *sums_out);

/* ---- multi-GPU: the path's one collective -------------------------------
 * The reference is single-process (n_jobs=1 hard-coded at AC:580; no collective anywhere).  Here genes shard across
 * one process per GPU with no data-path communication; at the end ONE ncclAllGather over RCCL (xGMI inside a node)
 * hands every rank the per-gene rows (I, expected_I, z_score, p_value) of all shards.  RCCL is dlopen'ed on first
 * use.  Rendezvous: one rank calls sc_comm_unique_id and passes the 128 bytes to the others by any side channel
 * (spatialcore_amd/parallel.py: a file keyed by the launcher's environment); all ranks call sc_comm_create.
 * sc_allgather: host arrays; every rank contributes `count` doubles, out holds world * count, in rank order.
 * sc_allreduce_max: in-place element-wise maximum over ranks (bench: slowest rank's clock; doubles as a barrier).
 * sc_allreduce_sum_i64: in-place element-wise integer sum over ranks (permutation sharding, SURVEY 8(e) "alternative":
 *   ranks score disjoint permutation ranges and add their exceedance counts -- integers, hence exact and order-free).
 * sc_comm_info: what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice). */
typedef struct sc_comm sc_comm;
int sc_comm_unique_id(uint8_t *id_out_128);
int sc_comm_create(sc_ctx *ctx, const uint8_t *id_128, int world, int rank, sc_comm **out);
int sc_comm_destroy(sc_comm *comm);
int sc_allgather(sc_comm *comm, const double *local, int64_t count, double *out);
int sc_allreduce_max(sc_comm *comm, double *values, int64_t count);
int sc_allreduce_sum_i64(sc_comm *comm, int64_t *values, int64_t count);
int sc_comm_info(sc_comm *comm, int *world_out, int *rank_out, int *device_out);

#ifdef __cplusplus
}
#endif
#endif /* SPATIALCORE_HIP_H */
