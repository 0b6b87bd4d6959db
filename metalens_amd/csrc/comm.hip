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
    ncclResult_t (*CommInitRankConfig)(ncclComm_t *, int, ncclUniqueId, int, ncclConfig_t *) = nullptr;   // (optional)
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                              ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                                  ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
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
    ML_SYM(ReduceScatter, "ncclReduceScatter")
    ML_SYM(AllGather, "ncclAllGather")
    ML_SYM(CommCount, "ncclCommCount")
    ML_SYM(CommDestroy, "ncclCommDestroy")
    ML_SYM(GetErrorString, "ncclGetErrorString")
#undef ML_SYM
    g_rccl.CommInitRankConfig = reinterpret_cast<decltype(g_rccl.CommInitRankConfig)>(dlsym(h, "ncclCommInitRankConfig"));
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
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    ctx->reduce_in_flight = false;
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
}

// ---- test backend: all-reduce through files -------------------------------------------------
// Every rank writes its operand to <key>.<seq>.<rank>, waits for the other ranks' files of the
// same sequence number, and combines them in rank order (deterministic).  A rank removes its own
// file of sequence n-2 when it starts n: by then every rank has finished reading n-2 (it had
// to, to write n-1, which this rank has read).
static std::string comm_file_name(const ml_ctx *ctx, long seq, int rank) {
    char b[512];
    const char *tmp = getenv("TMPDIR");
    snprintf(b, sizeof b, "%s/mlcomm_%s.%ld.%d", tmp ? tmp : "/tmp", ctx->comm_file_key.c_str(), seq,
             rank);
    return b;
}

// op: 0 sum, 1 max (all-reduce of the whole buffer); 2 reduce-scatter (sum, only this rank's chunk of
// count / n_ranks doubles is summed and written back); 3 all-gather (chunk r from rank r)
static int allreduce_file(ml_ctx *ctx, double *buf, size_t count, int op, hipStream_t stream) {
    ML_HIP(hipStreamSynchronize(stream));
    std::vector<double> mine(count), other(count), acc(count);
    ML_HIP(hipMemcpy(mine.data(), buf, count * sizeof(double), hipMemcpyDeviceToHost));
    const long seq = ctx->comm_file_seq++;
    if (seq >= 2) remove(comm_file_name(ctx, seq - 2, ctx->rank).c_str());
    const std::string final_name = comm_file_name(ctx, seq, ctx->rank), tmp_name = final_name + ".tmp";
    FILE *f = fopen(tmp_name.c_str(), "wb");
    ML_REQUIRE(f, "cannot write %s", tmp_name.c_str());
    const bool wrote = fwrite(mine.data(), sizeof(double), count, f) == count;
    fclose(f);
    ML_REQUIRE(wrote && rename(tmp_name.c_str(), final_name.c_str()) == 0, "cannot publish %s",
               final_name.c_str());
    for (int r = 0; r < ctx->n_ranks; ++r) {
        const double *src = mine.data();
        if (r != ctx->rank) {
            const std::string name = comm_file_name(ctx, seq, r);
            FILE *g = nullptr;
            for (int tries = 0; tries < 120000 && !g; ++tries) {   // up to ~2 minutes
                g = fopen(name.c_str(), "rb");
                if (!g) usleep(1000);
            }
            ML_REQUIRE(g, "timed out waiting for %s", name.c_str());
            const bool ok = fread(other.data(), sizeof(double), count, g) == count;
            fclose(g);
            ML_REQUIRE(ok, "short read from %s", name.c_str());
            src = other.data();
        }
        if (op == 3) {
            const size_t chunk = count / ctx->n_ranks;
            for (size_t k = r * chunk; k < (r + 1) * chunk; ++k) acc[k] = src[k];
            continue;
        }
        for (size_t k = 0; k < count; ++k)
            acc[k] = r == 0 ? src[k] : (op == 1 ? (src[k] > acc[k] ? src[k] : acc[k]) : acc[k] + src[k]);
    }
    if (op == 2) {
        const size_t chunk = count / ctx->n_ranks, at = ctx->rank * chunk;
        ML_HIP(hipMemcpy(buf + at, acc.data() + at, chunk * sizeof(double), hipMemcpyHostToDevice));
        return ML_OK;
    }
    ML_HIP(hipMemcpy(buf, acc.data(), count * sizeof(double), hipMemcpyHostToDevice));
    return ML_OK;
}

static int allreduce_dev(ml_ctx *ctx, double *buf, size_t count, int op, hipStream_t stream) {
    if (ctx->comm_file) return allreduce_file(ctx, buf, count, op, stream);
    if (ctx->n_ranks <= 1 && !ctx->comm) return ML_OK;
    if (!ctx->comm) {
        set_error("ml_comm_init has not been called");
        return ML_ESTATE;
    }
    ML_NCCL(g_rccl.AllReduce(buf, buf, count, ncclDouble, op == 1 ? ncclMax : ncclSum,
                             (ncclComm_t)ctx->comm, stream));
    return ML_OK;
}

int comm_allreduce_sum(ml_ctx *ctx, double *buf, size_t count, hipStream_t stream) {
    return allreduce_dev(ctx, buf, count, 0, stream);
}

int comm_reduce_scatter_sum(ml_ctx *ctx, double *buf, size_t chunk, hipStream_t stream) {
    if (ctx->comm_file) return allreduce_file(ctx, buf, chunk * ctx->n_ranks, 2, stream);
    if (ctx->n_ranks <= 1 && !ctx->comm) return ML_OK;
    if (!ctx->comm) {
        set_error("ml_comm_init has not been called");
        return ML_ESTATE;
    }
    // in place: the received chunk lands where this rank's own chunk of the send buffer is
    ML_NCCL(g_rccl.ReduceScatter(buf, buf + (size_t)ctx->rank * chunk, chunk, ncclDouble, ncclSum,
                                 (ncclComm_t)ctx->comm, stream));
    return ML_OK;
}

int comm_allgather(ml_ctx *ctx, double *buf, size_t chunk, hipStream_t stream) {
    if (ctx->comm_file) return allreduce_file(ctx, buf, chunk * ctx->n_ranks, 3, stream);
    if (ctx->n_ranks <= 1 && !ctx->comm) return ML_OK;
    if (!ctx->comm) {
        set_error("ml_comm_init has not been called");
        return ML_ESTATE;
    }
    ML_NCCL(g_rccl.AllGather(buf + (size_t)ctx->rank * chunk, buf, chunk, ncclDouble, (ncclComm_t)ctx->comm, stream));
    return ML_OK;
}

int comm_join(ml_ctx *ctx, bool host) {
    if (!ctx->reduce_in_flight) return ML_OK;
    // every reduction records reduce_done[slot] behind its power kernel; the latest one is
    // plan.amp_slot's, and the comm stream runs them in order
    hipEvent_t e = ctx->reduce_done[ctx->plan.amp_slot];
    if (host) {
        ML_HIP(hipEventSynchronize(e));
        ctx->reduce_in_flight = false;
    } else {
        ML_HIP(hipStreamWaitEvent(ctx->stream, e, 0));
    }
    return ML_OK;
}

}  // namespace ml

using namespace ml;

extern "C" {

static bool file_backend() {
    const char *e = getenv("ML_COMM_BACKEND");
    return e && strcmp(e, "file") == 0;
}

int ml_comm_unique_id(uint8_t id[128]) {
    ML_REQUIRE(id, "id is NULL");
    if (file_backend()) {   // any bytes that differ from run to run will do
        FILE *f = fopen("/dev/urandom", "rb");
        const bool ok = f && fread(id, 1, 128, f) == 128;
        if (f) fclose(f);
        ML_REQUIRE(ok, "cannot read /dev/urandom");
        return ML_OK;
    }
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
    ML_TRY(comm_join(ctx, true));
    comm_release(ctx);
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    // ML_FORCE_RCCL=1 builds a real one-rank communicator (exercises dlopen, the unique-id
    // ABI and the all-reduce on a single-GPU box)
    ctx->comm_file = false;
    if (file_backend()) {
        char key[40];
        for (int k = 0; k < 16; ++k) snprintf(key + 2 * k, 3, "%02x", id[k]);
        ctx->comm_file = true;
        ctx->comm_file_key = key;
        ctx->comm_file_seq = 0;
        return ML_OK;
    }
    const char *force = getenv("ML_FORCE_RCCL");
    if (n_ranks == 1 && !(force && atoi(force))) return ML_OK;
    // The step's collective (a few MB per rank) runs on its own stream BESIDE the next step's synthesis,
    // whose waves fill every SIMD's register file: each CU a collective kernel occupies is a CU the
    // synthesis loses.  Four channels carry the payload in well under a step; RCCL's default
    // takes several times as many workgroups.  The cap belongs to THIS communicator (ncclConfig_t::maxCTAs,
    // ml_comm_set_max_channels): the process environment - and with it every other RCCL user in the
    // process - is left alone.
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
    ncclResult_t rc;
    if (ctx->comm_max_channels > 0 && g_rccl.CommInitRankConfig) {
        ncclConfig_t config = NCCL_CONFIG_INITIALIZER;
        config.maxCTAs = ctx->comm_max_channels;
        rc = g_rccl.CommInitRankConfig(&comm, n_ranks, u, rank, &config);
        // (a library that does not take this header's config struct refuses it before any rank talks to
        // another - identically on every rank - so the plain call with the same id is still possible)
        if (rc == ncclInvalidArgument) {
            comm = nullptr;
            rc = g_rccl.CommInitRank(&comm, n_ranks, u, rank);
        }
    } else {
        rc = g_rccl.CommInitRank(&comm, n_ranks, u, rank);
    }
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
    ML_TRY(flush_unfold(ctx));
    // an amplitude reduction of an earlier step may still run on comm_stream: two collectives of
    // one communicator must not be in flight on two streams (the ranks could enqueue them in
    // different orders)
    ML_TRY(comm_join(ctx, false));
    const size_t n = (size_t)pl.mx * (pl.pair_list ? 1 : pl.my);
    pl.amplitudes_reduced = false;
    return allreduce_dev(ctx, pl.vectors.as<double>(), 4 * n * 2, 0, ctx->stream);
}

int ml_comm_allreduce_host(ml_ctx *ctx, double *values, int count, int op) {
    ML_REQUIRE(ctx && values && count >= 1, "bad argument");
    if (ctx->n_ranks <= 1 && !ctx->comm && !ctx->comm_file) return ML_OK;
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(comm_join(ctx, true));   // one collective of this communicator at a time
    ML_TRY(ctx->comm_scratch.reserve(count * sizeof(double)));
    ML_HIP(hipMemcpyAsync(ctx->comm_scratch.p, values, count * sizeof(double),
                          hipMemcpyHostToDevice, ctx->stream));
    ML_TRY(allreduce_dev(ctx, ctx->comm_scratch.as<double>(), count, op, ctx->stream));
    ML_HIP(hipMemcpyAsync(values, ctx->comm_scratch.p, count * sizeof(double),
                          hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return ML_OK;
}

int ml_comm_set_reduce(ml_ctx *ctx, int mode) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(mode == ML_REDUCE_SCATTER || mode == ML_REDUCE_ALL, "unknown reduction mode %d", mode);
    ctx->reduce_by_allreduce = mode == ML_REDUCE_ALL;
    return ML_OK;
}

int ml_comm_set_max_channels(ml_ctx *ctx, int channels) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(channels >= 0 && channels <= 256, "channel cap %d out of range [0, 256]", channels);
    ML_REQUIRE(!ctx->comm, "the cap belongs to the communicator: set it before ml_comm_init");
    ctx->comm_max_channels = channels;
    return ML_OK;
}

int ml_comm_info(ml_ctx *ctx, int *n_ranks, int *rank, int *backend) {
    ML_REQUIRE(ctx && n_ranks && rank && backend, "NULL argument");
    *n_ranks = ctx->n_ranks;
    *rank = ctx->rank;
    *backend = ctx->comm_file ? 2 : ctx->comm ? 1 : 0;
    if (ctx->comm) ML_NCCL(g_rccl.CommCount((ncclComm_t)ctx->comm, n_ranks));   // what RCCL itself says
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
