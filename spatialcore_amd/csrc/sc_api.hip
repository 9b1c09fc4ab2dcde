// Context, error and timer plumbing of libspatialcore_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "sc_ctx.h"

static thread_local char g_err[1024] = "";

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of
// streams that share a hardware queue run strictly one after the other.  The generator / scoring pipeline of
// sc_moran_seeded uses nine streams whose kernels should overlap (a 25-ms scoring launch in front of the generator's
// sub-millisecond chain launches doubles the step), so the library asks for one hardware queue per stream -- before
// the runtime initialises, i.e. when the library is loaded, and only if the user has not set the variable.  It is a
// REQUEST, not a requirement: a host application that initialised HIP earlier (or set a smaller value) keeps its
// setting; the generator probes its streams before its first block-parallel job (permgen_probe_streams), uses the
// sequential scan when they cannot overlap, and says so through sc_ctx_permgen_note (logged once by the Python layer).
// 24 = two contexts' worth: a context creates up to 11 streams, and once a process's streams outnumber the setting the
// runtime lets them SHARE queues (r03: with 16, a second context in one process found its generator streams on shared
// queues and fell back; 24 and 32 were measured: same bench step as 16, every test with two live contexts green).
__attribute__((constructor)) static void sc_request_hw_queues(void) { setenv("GPU_MAX_HW_QUEUES", "24", 0); }

void sc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int DBuf::ensure(size_t bytes, int64_t *acct)
{
    if (bytes <= cap) return SC_OK;
    if (p) {
        (void)hipFree(p);
        *acct -= (int64_t)cap;
        p = nullptr;
        cap = 0;
    }
    // round up so that small growth does not re-allocate
    size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        sc_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_HIP;
    }
    cap = want;
    *acct += (int64_t)cap;
    return SC_OK;
}

void DBuf::release(int64_t *acct)
{
    if (p) {
        (void)hipFree(p);
        *acct -= (int64_t)cap;
    }
    p = nullptr;
    cap = 0;
}

KernelTimerScope::KernelTimerScope(sc_ctx *ctx, int kid, hipStream_t on) : c(ctx), id(kid), s(on ? on : ctx->stream)
{
    if (!c->timing) return;
    KTimer &t = c->timers[id];
    if (!t.pool.empty()) {
        a = t.pool.back().first;
        b = t.pool.back().second;
        t.pool.pop_back();
    } else {
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            a = b = nullptr;
            return;
        }
    }
    (void)hipEventRecord(a, s);
}

KernelTimerScope::~KernelTimerScope()
{
    if (!a) return;
    (void)hipEventRecord(b, s);
    KTimer &t = c->timers[id];
    t.pending.emplace_back(a, b);
    t.launches += 1;
}

int sc_timer_collect(sc_ctx *c)
{
    SC_HIP(hipStreamSynchronize(c->stream));
    if (c->stream2) SC_HIP(hipStreamSynchronize(c->stream2));
    if (c->stream3) SC_HIP(hipStreamSynchronize(c->stream3));
    if (c->stream4) SC_HIP(hipStreamSynchronize(c->stream4));
    for (hipStream_t sp : c->stream_pg)
        if (sp) SC_HIP(hipStreamSynchronize(sp));
    if (c->stream_px) SC_HIP(hipStreamSynchronize(c->stream_px));
    if (c->stream_fr) SC_HIP(hipStreamSynchronize(c->stream_fr));
    if (c->stream_out) SC_HIP(hipStreamSynchronize(c->stream_out));
    for (int k = 0; k < SC_K_COUNT_; ++k) {
        KTimer &t = c->timers[k];
        for (auto &ev : t.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) t.ms += ms;
            t.pool.push_back(ev);
        }
        t.pending.clear();
    }
    return SC_OK;
}

extern "C" {

int sc_version(void) { return 100; }

const char *sc_last_error(void) { return g_err; }

int sc_device_count(int *count)
{
    SC_REQUIRE(count, SC_ERR_INVALID, "sc_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        sc_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return SC_ERR_HIP;
    }
    *count = n;
    return SC_OK;
}

int sc_ctx_create(int device, sc_ctx **out)
{
    SC_REQUIRE(out, SC_ERR_INVALID, "sc_ctx_create: null out pointer");
    *out = nullptr;
    int n = 0;
    SC_HIP(hipGetDeviceCount(&n));
    SC_REQUIRE(device >= 0 && device < n, SC_ERR_INVALID,
               "sc_ctx_create: device %d out of range (%d visible)", device, n);
    SC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SC_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        sc_set_error("sc_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                     prop.gcnArchName);
        return SC_ERR_STATE;
    }
    sc_ctx *c = new sc_ctx();
    c->device = device;
    if (getenv("SC_LOCAL_MORAN_DIRECT")) c->lm_direct = true;  // development: A/B of the per-cell count kernels
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        sc_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return SC_ERR_HIP;
    }
    *out = c;
    return SC_OK;
}

int sc_ctx_destroy(sc_ctx *c)
{
    if (!c) return SC_OK;
    (void)hipSetDevice(c->device);
    sc_perm_pipe_abort(c);
    sc_graph_moments_drain(c);
    if (c->mom_host) (void)hipHostFree(c->mom_host);
    if (c->prep_host) (void)hipHostFree(c->prep_host);
    if (c->mom_ready) (void)hipEventDestroy(c->mom_ready);
    if (c->mom_done) (void)hipEventDestroy(c->mom_done);
    if (c->stream_m) (void)hipStreamDestroy(c->stream_m);
    if (c->knn_done) (void)hipEventDestroy(c->knn_done);
    (void)hipStreamSynchronize(c->stream);
    DBuf *bufs[] = {&c->px, &c->py, &c->sx, &c->sy, &c->sid, &c->bin_start, &c->bin_keys,
                    &c->bin_keys2, &c->sid2, &c->cub_tmp, &c->knn_idx, &c->knn_rd, &c->knn_hd, &c->knn_hi, &c->rad_indptr,
                    &c->g_indptr, &c->g_indices, &c->g_data, &c->gt_indptr, &c->gt_indices,
                    &c->gt_data, &c->gt_cursor, &c->gt_tmp, &c->mom_dev, &c->X, &c->Z, &c->Lag, &c->X32, &c->inv, &c->e_tmp_indptr,
                    &c->e_tmp_indices, &c->e_tmp_data, &c->e_colmap, &c->g_mean, &c->g_var,
                    &c->g_z2, &c->g_scale, &c->g_Inum, &c->g_I, &c->red_tmp, &c->perm, &c->perm_flag,
                    &c->partial, &c->sims, &c->counts, &c->sim_sum, &c->sim_sumsq, &c->lee_a,
                    &c->lee_b, &c->lee_out, &c->lee_pairs, &c->lee_U, &c->lee_Zc, &c->lee_Uc, &c->lee_part, &c->lee_obs, &c->lee_cnt,
                    &c->lee_rowmap, &c->lee_lperm, &c->g_slag, &c->g_xsum, &c->g_flags, &c->g_xmax, &c->g_lat, &c->g_meanc, &c->g_seff, &c->g_corr, &c->g_thr, &c->sims_raw, &c->g_order, &c->g_rank, &c->g_indices_r, &c->g_w32, &c->g_erow_r, &c->lm_ys, &c->lm_out, &c->lm_tab, &c->s0_tmp, &c->pg_J, &c->pg_raw, &c->pg_out, &c->pg_flags, &c->pg_bits, &c->pg_enter, &c->pg_sblk,
                    &c->pg_desc, &c->pg_tbits, &c->pg_events, &c->pg_hard, &c->pg_seg, &c->pg_ctbits, &c->pg_segmode, &c->pg_seglist, &c->pg_fresh, &c->nib_map,
                    &c->np_cnt, &c->np_comp, &c->np_leaves, &c->np_leafsum};
    for (DBuf *b : bufs) b->release(&c->mem);
    for (int k = 0; k < SC_K_COUNT_; ++k) {
        for (auto &ev : c->timers[k].pending) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        for (auto &ev : c->timers[k].pool) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    }
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    for (hipStream_t sp : c->stream_pg)
        if (sp) (void)hipStreamDestroy(sp);
    if (c->stream_px) (void)hipStreamDestroy(c->stream_px);
    if (c->stream_fr) (void)hipStreamDestroy(c->stream_fr);
    if (c->stream_out) (void)hipStreamDestroy(c->stream_out);
    for (hipEvent_t e : c->pg_ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return SC_OK;
}

int sc_ctx_sync(sc_ctx *c)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_HIP(hipSetDevice(c->device));
    SC_HIP(hipDeviceSynchronize());  // every stream of the device (what torch.cuda.synchronize() would do)
    return SC_OK;
}

// Development aid: copy `bytes` bytes at `offset` of one of the generator's device buffers to the host.
int sc_debug_copy(sc_ctx *c, int which, int64_t offset, void *out, int64_t bytes)
{
    SC_REQUIRE(c && out && offset >= 0 && bytes >= 0, SC_ERR_INVALID, "sc_debug_copy: bad argument");
    SC_HIP(hipSetDevice(c->device));
    if (which == 100) {   // development builds of the generator (-DPHI_PROFILE): 32 words, reset when offset != 0
        SC_REQUIRE(bytes == 32 * (int64_t)sizeof(unsigned long long), SC_ERR_INVALID, "sc_debug_copy: the generator profile is 32 words");
        return sc_permgen_profile(reinterpret_cast<unsigned long long *>(out), offset != 0);
    }
    const DBuf *bufs[] = {&c->pg_J, &c->pg_raw, &c->pg_bits, &c->pg_enter, &c->pg_sblk, &c->pg_out, &c->perm, &c->inv, &c->pg_fresh};
    SC_REQUIRE(which >= 0 && which < (int)(sizeof(bufs) / sizeof(bufs[0])), SC_ERR_INVALID, "sc_debug_copy: unknown buffer %d", which);
    const DBuf *b = bufs[which];
    SC_REQUIRE((size_t)(offset + bytes) <= b->cap, SC_ERR_INVALID, "sc_debug_copy: range exceeds the buffer (%zu bytes)", b->cap);
    SC_HIP(hipDeviceSynchronize());
    SC_HIP(hipMemcpy(out, (const char *)b->p + offset, (size_t)bytes, hipMemcpyDeviceToHost));
    return SC_OK;
}

int sc_ctx_kernel_time(sc_ctx *c, int kernel_id, double *ms, int64_t *launches)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_REQUIRE(kernel_id >= 0 && kernel_id < SC_K_COUNT_, SC_ERR_INVALID, "bad kernel id %d", kernel_id);
    SC_TRY(sc_timer_collect(c));
    if (ms) *ms = c->timers[kernel_id].ms;
    if (launches) *launches = c->timers[kernel_id].launches;
    return SC_OK;
}

int sc_ctx_reset_timers(sc_ctx *c)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_TRY(sc_timer_collect(c));
    for (int k = 0; k < SC_K_COUNT_; ++k) {
        c->timers[k].ms = 0.0;
        c->timers[k].launches = 0;
    }
    return SC_OK;
}

int sc_ctx_set_timing(sc_ctx *c, int enabled)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    c->timing = enabled != 0;
    return SC_OK;
}

int sc_ctx_set_permgen_mode(sc_ctx *c, int mode)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_REQUIRE(mode >= 0 && mode <= 2, SC_ERR_INVALID, "sc_ctx_set_permgen_mode: mode %d not in {0, 1, 2}", mode);
    c->pg_mode = mode;
    // re-arm: a context that fell back to the sequential scan after one stalled hand-over (a transient: GPU shared with
    // another process, a profiler pass) probes its streams again and may return to the block-parallel form
    c->pg_streams_serial = false;
    c->pg_probed = false;
    c->pg_note.clear();
    return SC_OK;
}

int sc_ctx_permgen_note(sc_ctx *c, const char **message)
{
    SC_REQUIRE(c && message, SC_ERR_INVALID, "null pointer");
    *message = c->pg_note.c_str();   // "" while the block-parallel generator is in use; valid until the next generator call
    return SC_OK;
}

int sc_ctx_set_moran_source_bits(sc_ctx *c, int min_bits)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    SC_REQUIRE(min_bits == 4 || min_bits == 8 || min_bits == 16 || min_bits == 32 || min_bits == 64, SC_ERR_INVALID,
               "sc_ctx_set_moran_source_bits: %d not in {4, 8, 16, 32, 64}", min_bits);
    c->source_bits_min = min_bits;
    return SC_OK;
}

int sc_ctx_moran_source_bits(sc_ctx *c, int *bits)
{
    SC_REQUIRE(c && bits, SC_ERR_INVALID, "null pointer");
    *bits = c->last_source_bits;
    return SC_OK;
}

int sc_ctx_moran_lag_bits(sc_ctx *c, int *bits)
{
    SC_REQUIRE(c && bits, SC_ERR_INVALID, "null pointer");
    *bits = c->lag_u16 ? 16 : 64;
    return SC_OK;
}

int sc_ctx_moran_row_groups(sc_ctx *c, int *groups)
{
    SC_REQUIRE(c && groups, SC_ERR_INVALID, "null pointer");
    const int64_t gp = c->narrow_bits == 8 ? 128 : c->narrow_bits == 16 ? 64 : c->narrow_bits == 32 ? 32 : 16;
    *groups = c->narrow_bits == 4 ? c->nib_groups : (int)ceil_div64(c->e_tiles * SC_TILE, gp);
    return SC_OK;
}

int sc_ctx_permgen_stats(sc_ctx *c, int64_t *jobs_parallel, int64_t *jobs_sequential, int64_t *fallbacks,
                         int64_t *blocks_prepared, int64_t *blocks_chain)
{
    SC_REQUIRE(c, SC_ERR_INVALID, "null context");
    if (jobs_parallel) *jobs_parallel = c->pg_jobs_parallel;
    if (jobs_sequential) *jobs_sequential = c->pg_jobs_sequential;
    if (fallbacks) *fallbacks = c->pg_fallbacks;
    if (blocks_prepared) *blocks_prepared = c->pg_blocks_prepared;
    if (blocks_chain) *blocks_chain = c->pg_blocks_chain;
    return SC_OK;
}

int sc_ctx_device_mem(sc_ctx *c, int64_t *bytes)
{
    SC_REQUIRE(c && bytes, SC_ERR_INVALID, "null pointer");
    *bytes = c->mem;
    return SC_OK;
}

}  // extern "C"
