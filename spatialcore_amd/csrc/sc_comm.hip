// The path's one collective: an all-gather of the small per-gene result blocks over RCCL (xGMI inside a node).
//
// RCCL is loaded lazily with dlopen: a single-GPU user never maps librccl.so (hundreds of MB), and the library still
// loads on a box that has no RCCL at all.  The rendezvous is the plain NCCL one: one rank calls
// sc_comm_unique_id and hands the 128 bytes to the others by any side channel (spatialcore_amd/parallel.py uses a
// file keyed by the launcher's environment); every rank then calls sc_comm_create with the same id.
//
// Messages are tiny ((genes / ranks) x 4 doubles, <= 64 KB for 2000 genes): the collective is latency-bound, so
// there is nothing to tune for the point-to-point xGMI topology -- one ncclAllGather, no bucketing.
#include <dlfcn.h>
#include <string.h>

#include "sc_ctx.h"

namespace {

// the few RCCL entry points used, with the prototypes of /opt/rocm/include/rccl/rccl.h (ROCm 7.2)
struct RcclId { char internal[128]; };
typedef void *RcclComm;
typedef int (*fn_get_unique_id)(RcclId *);
typedef int (*fn_comm_init_rank)(RcclComm *, int, RcclId, int);
typedef int (*fn_comm_destroy)(RcclComm);
typedef const char *(*fn_error_string)(int);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, RcclComm, hipStream_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, RcclComm, hipStream_t);
typedef int (*fn_comm_int)(const RcclComm, int *);
const int kRcclFloat64 = 8;  // ncclFloat64
const int kRcclInt64 = 4;    // ncclInt64
const int kRcclSum = 0;      // ncclSum
const int kRcclMax = 2;      // ncclMax

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_int comm_count = nullptr, comm_user_rank = nullptr, comm_device = nullptr;
} g_rccl;

int rccl_load()
{
    if (g_rccl.handle) return SC_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    SC_REQUIRE(h, SC_ERR_STATE, "RCCL is not available: dlopen(librccl.so.1) failed: %s", dlerror());
    Rccl r;
    r.handle = h;
    r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    r.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    r.comm_count = (fn_comm_int)dlsym(h, "ncclCommCount");
    r.comm_user_rank = (fn_comm_int)dlsym(h, "ncclCommUserRank");
    r.comm_device = (fn_comm_int)dlsym(h, "ncclCommCuDevice");
    if (!r.comm_count || !r.comm_user_rank || !r.comm_device || !r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.error_string || !r.all_gather || !r.all_reduce) {
        dlclose(h);
        sc_set_error("librccl.so lacks an expected nccl* entry point");
        return SC_ERR_STATE;
    }
    g_rccl = r;
    return SC_OK;
}

#define SC_RCCL(call)                                                                                  \
    do {                                                                                               \
        int r__ = (call);                                                                              \
        if (r__ != 0) {                                                                                \
            sc_set_error("%s failed: %s (%s:%d)", #call, g_rccl.error_string(r__), __FILE__, __LINE__); \
            return SC_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

}  // namespace

struct sc_comm {
    sc_ctx *ctx = nullptr;
    RcclComm comm = nullptr;
    int world = 1, rank = 0;
    DBuf send, recv;
    int64_t mem = 0;
};

extern "C" int sc_comm_unique_id(uint8_t *id_out)
{
    SC_REQUIRE(id_out, SC_ERR_INVALID, "sc_comm_unique_id: null pointer");
    SC_TRY(rccl_load());
    RcclId id;
    SC_RCCL(g_rccl.get_unique_id(&id));
    memcpy(id_out, id.internal, sizeof(id.internal));
    return SC_OK;
}

extern "C" int sc_comm_create(sc_ctx *c, const uint8_t *id, int world, int rank, sc_comm **out)
{
    SC_REQUIRE(c && id && out, SC_ERR_INVALID, "sc_comm_create: null pointer");
    SC_REQUIRE(world >= 1 && rank >= 0 && rank < world, SC_ERR_INVALID, "sc_comm_create: bad rank/world %d/%d", rank, world);
    *out = nullptr;
    SC_TRY(rccl_load());
    SC_HIP(hipSetDevice(c->device));
    sc_comm *m = new sc_comm;
    m->ctx = c;
    m->world = world;
    m->rank = rank;
    RcclId rid;
    memcpy(rid.internal, id, sizeof(rid.internal));
    int r = g_rccl.comm_init_rank(&m->comm, world, rid, rank);
    if (r != 0) {
        sc_set_error("ncclCommInitRank(world=%d, rank=%d, device=%d) failed: %s", world, rank, c->device,
                     g_rccl.error_string(r));
        delete m;
        return SC_ERR_HIP;
    }
    *out = m;
    return SC_OK;
}

extern "C" int sc_comm_destroy(sc_comm *m)
{
    if (!m) return SC_OK;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(m->comm);
    m->send.release(&m->mem);
    m->recv.release(&m->mem);
    delete m;
    return SC_OK;
}

// out[r * count .. (r + 1) * count) = rank r's `local` (host arrays in and out; fp64).  Every rank passes the same
// count.  Upload, ONE ncclAllGather on the context stream, download.
extern "C" int sc_allgather(sc_comm *m, const double *local, int64_t count, double *out)
{
    SC_REQUIRE(m && local && out, SC_ERR_INVALID, "sc_allgather: null pointer");
    SC_REQUIRE(count >= 1 && count <= ((int64_t)1 << 31), SC_ERR_INVALID, "sc_allgather: count=%lld out of range",
               (long long)count);
    sc_ctx *c = m->ctx;
    SC_HIP(hipSetDevice(c->device));
    const size_t bytes = sizeof(double) * (size_t)count;
    SC_TRY(m->send.ensure(bytes, &m->mem));
    SC_TRY(m->recv.ensure(bytes * (size_t)m->world, &m->mem));
    SC_HIP(hipMemcpyAsync(m->send.p, local, bytes, hipMemcpyHostToDevice, c->stream));
    SC_RCCL(g_rccl.all_gather(m->send.p, m->recv.p, (size_t)count, kRcclFloat64, m->comm, c->stream));
    SC_HIP(hipMemcpyAsync(out, m->recv.p, bytes * (size_t)m->world, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// values[k] <- max over ranks (fp64, in place, host array): the bench's "slowest rank" clock and its barrier.
extern "C" int sc_allreduce_max(sc_comm *m, double *values, int64_t count)
{
    SC_REQUIRE(m && values, SC_ERR_INVALID, "sc_allreduce_max: null pointer");
    SC_REQUIRE(count >= 1 && count <= (1 << 20), SC_ERR_INVALID, "sc_allreduce_max: count out of range");
    sc_ctx *c = m->ctx;
    SC_HIP(hipSetDevice(c->device));
    const size_t bytes = sizeof(double) * (size_t)count;
    SC_TRY(m->send.ensure(bytes, &m->mem));
    SC_HIP(hipMemcpyAsync(m->send.p, values, bytes, hipMemcpyHostToDevice, c->stream));
    SC_RCCL(g_rccl.all_reduce(m->send.p, m->send.p, (size_t)count, kRcclFloat64, kRcclMax, m->comm, c->stream));
    SC_HIP(hipMemcpyAsync(values, m->send.p, bytes, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// values[k] <- sum over ranks (int64, in place, host array): exceedance counts of permutation-sharded runs.
extern "C" int sc_allreduce_sum_i64(sc_comm *m, int64_t *values, int64_t count)
{
    SC_REQUIRE(m && values, SC_ERR_INVALID, "sc_allreduce_sum_i64: null pointer");
    SC_REQUIRE(count >= 1 && count <= ((int64_t)1 << 28), SC_ERR_INVALID, "sc_allreduce_sum_i64: count out of range");
    sc_ctx *c = m->ctx;
    SC_HIP(hipSetDevice(c->device));
    const size_t bytes = sizeof(int64_t) * (size_t)count;
    SC_TRY(m->send.ensure(bytes, &m->mem));
    SC_HIP(hipMemcpyAsync(m->send.p, values, bytes, hipMemcpyHostToDevice, c->stream));
    SC_RCCL(g_rccl.all_reduce(m->send.p, m->send.p, (size_t)count, kRcclInt64, kRcclSum, m->comm, c->stream));
    SC_HIP(hipMemcpyAsync(values, m->send.p, bytes, hipMemcpyDeviceToHost, c->stream));
    SC_HIP(hipStreamSynchronize(c->stream));
    return SC_OK;
}

// what RCCL reports for the communicator: number of ranks, this rank, the device it is bound to
extern "C" int sc_comm_info(sc_comm *m, int *world_out, int *rank_out, int *device_out)
{
    SC_REQUIRE(m && m->comm, SC_ERR_INVALID, "sc_comm_info: null communicator");
    int v = 0;
    if (world_out) { SC_RCCL(g_rccl.comm_count(m->comm, &v)); *world_out = v; }
    if (rank_out) { SC_RCCL(g_rccl.comm_user_rank(m->comm, &v)); *rank_out = v; }
    if (device_out) { SC_RCCL(g_rccl.comm_device(m->comm, &v)); *device_out = v; }
    return SC_OK;
}
