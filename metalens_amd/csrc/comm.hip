// Multi-GPU exchange: one process per GPU, RCCL over xGMI.
//
// The near field is independent per aperture sample and the far-field sum is linear in the
// aperture (which is why the reference's strip chunking, nearfield.py:482-516, is exact), so
// the aperture rows are sharded over the ranks with no data-path communication until the
// partial radiation vectors Nx,Ny,Lx,Ly [mx][my] are summed with ONE all-reduce
// (4*mx*my*2 doubles; RCCL has no complex type, so it is reduced as ncclDouble).
//
// librccl is loaded with dlopen the first time a communicator is needed, so single-GPU use
// never touches it.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdlib>
#include <rccl/rccl.h>

#include "common.h"

namespace ml {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                              ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int load_rccl() {
    if (g_rccl.handle) return ML_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        set_error("cannot load librccl: %s", dlerror());
        return ML_ERCCL;
    }
#define ML_SYM(field, name)                                               \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
    if (!g_rccl.field) {                                                  \
        set_error("librccl lacks %s", name);                              \
        dlclose(h);                                                       \
        return ML_ERCCL;                                                  \
    }
    ML_SYM(GetUniqueId, "ncclGetUniqueId")
    ML_SYM(CommInitRank, "ncclCommInitRank")
    ML_SYM(AllReduce, "ncclAllReduce")
    ML_SYM(CommDestroy, "ncclCommDestroy")
    ML_SYM(GetErrorString, "ncclGetErrorString")
#undef ML_SYM
    g_rccl.handle = h;
    return ML_OK;
}

#define ML_NCCL(call)                                                              \
    do {                                                                           \
        ncclResult_t r_ = (call);                                                  \
        if (r_ != ncclSuccess) {                                                   \
            set_error("%s failed: %s", #call, g_rccl.GetErrorString(r_));          \
            return ML_ERCCL;                                                       \
        }                                                                          \
    } while (0)

void comm_release(ml_ctx *ctx) {
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
}

static int allreduce_dev(ml_ctx *ctx, double *buf, size_t count, int op) {
    if (ctx->n_ranks <= 1 && !ctx->comm) return ML_OK;
    if (!ctx->comm) {
        set_error("ml_comm_init has not been called");
        return ML_ESTATE;
    }
    ML_NCCL(g_rccl.AllReduce(buf, buf, count, ncclDouble, op == 1 ? ncclMax : ncclSum,
                             (ncclComm_t)ctx->comm, ctx->stream));
    return ML_OK;
}

int comm_allreduce_sum(ml_ctx *ctx, double *buf, size_t count) {
    return allreduce_dev(ctx, buf, count, 0);
}

}  // namespace ml

using namespace ml;

extern "C" {

int ml_comm_unique_id(uint8_t id[128]) {
    ML_REQUIRE(id, "id is NULL");
    ML_TRY(load_rccl());
    ncclUniqueId u;
    ML_NCCL(g_rccl.GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId size");
    memcpy(id, &u, 128);
    return ML_OK;
}

int ml_comm_init(ml_ctx *ctx, const uint8_t id[128], int n_ranks, int rank) {
    ML_REQUIRE(ctx && id, "NULL argument");
    ML_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank %d of %d", rank, n_ranks);
    ML_HIP(hipSetDevice(ctx->device));
    comm_release(ctx);
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    // ML_FORCE_RCCL=1 builds a real one-rank communicator (exercises dlopen, the unique-id
    // ABI and the all-reduce on a single-GPU box)
    const char *force = getenv("ML_FORCE_RCCL");
    if (n_ranks == 1 && !(force && atoi(force))) return ML_OK;
    ML_TRY(load_rccl());
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t comm = nullptr;
    // RCCL prints a version banner ("RCCL version : ...") to STDOUT from rank 0 when the first
    // communicator is created.  Programs built on this library report results on stdout
    // (bench.py: one JSON line), so send whatever the library prints during initialisation to
    // stderr instead: point fd 1 at fd 2 for the duration of the call and flush C stdio on both
    // sides of the switch.
    fflush(stdout);
    const int saved_stdout = dup(1);
    if (saved_stdout >= 0) dup2(2, 1);
    const ncclResult_t rc = g_rccl.CommInitRank(&comm, n_ranks, u, rank);
    fflush(stdout);
    if (saved_stdout >= 0) {
        dup2(saved_stdout, 1);
        close(saved_stdout);
    }
    ML_NCCL(rc);
    ctx->comm = comm;
    return ML_OK;
}

int ml_farfield_allreduce(ml_ctx *ctx) {
    ML_REQUIRE(ctx, "ctx is NULL");
    FarfieldPlan &pl = ctx->plan;
    if (!pl.ready || !pl.have_vectors) {
        set_error("no radiation vectors to reduce");
        return ML_ESTATE;
    }
    ML_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)pl.mx * (pl.pair_list ? 1 : pl.my);
    pl.amplitudes_reduced = false;
    return allreduce_dev(ctx, pl.vectors.as<double>(), 4 * n * 2, 0);
}

int ml_comm_allreduce_host(ml_ctx *ctx, double *values, int count, int op) {
    ML_REQUIRE(ctx && values && count >= 1, "bad argument");
    if (ctx->n_ranks <= 1 && !ctx->comm) return ML_OK;
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(ctx->comm_scratch.reserve(count * sizeof(double)));
    ML_HIP(hipMemcpyAsync(ctx->comm_scratch.p, values, count * sizeof(double),
                          hipMemcpyHostToDevice, ctx->stream));
    ML_TRY(allreduce_dev(ctx, ctx->comm_scratch.as<double>(), count, op));
    ML_HIP(hipMemcpyAsync(values, ctx->comm_scratch.p, count * sizeof(double),
                          hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return ML_OK;
}

int ml_comm_barrier(ml_ctx *ctx) {
    double one = 1.0;
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_HIP(hipSetDevice(ctx->device));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return ml_comm_allreduce_host(ctx, &one, 1, 0);
}

}  // extern "C"
