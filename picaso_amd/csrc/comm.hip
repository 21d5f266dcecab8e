// Multi-GPU layer: wavelength-sharded spectra gathered with RCCL over xGMI, inside the library.
//
// The reference fans independent spectra out to joblib worker processes (justdoit.py:4774); here the
// wavelength grid of ONE spectrum is cut into contiguous blocks, one per GPU, the solve needs no
// exchange (every function on the path is pointwise in wavelength), and the only
// collective is the all-gather of the final spectrum shards.  One process per GPU
// (picaso_comm_init_rank; the 128-byte id travels through whatever side channel the host has) or one
// process driving several GPUs (picaso_comm_init_all).  The collectives run on the context's own
// stream, behind the kernels that produced their input: no host synchronisation.
#include <rccl/rccl.h>

#include "common.hpp"

struct picaso_comm {
    ncclComm_t comm = nullptr;
    picaso_ctx *ctx = nullptr;
    int nranks = 1, rank = 0;
    double *scratch = nullptr;       // device scratch of the small host-value collectives
    // overlapped gathers (picaso_all_gather_async_dev): the collective runs on the communicator's own
    // stream behind an event of the compute stream, one completion event per result slot
    static constexpr int NSLOT = 4;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;
    hipEvent_t done[NSLOT] = {};
    bool pending[NSLOT] = {};
};

using namespace pz;

#define PZ_NCCL(ctx, expr)                                                                        \
    do {                                                                                          \
        ncclResult_t r__ = (expr);                                                                \
        if (r__ != ncclSuccess)                                                                   \
            return pz::fail(ctx, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, \
                            __LINE__);                                                            \
    } while (0)

extern "C" {

int picaso_comm_unique_id(void *id128)
{
    if (!id128) return fail(nullptr, "picaso_comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == PICASO_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    PZ_NCCL(nullptr, ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

// releases whatever of a communicator exists (also the clean-up of a failed set-up); no synchronisation
static void comm_release(picaso_comm *c)
{
    if (!c) return;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    for (int i = 0; i < picaso_comm::NSLOT; ++i)
        if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// takes ownership of `c` (destroyed with everything else when the set-up fails)
static int comm_finish(picaso_ctx *ctx, ncclComm_t c, int nranks, int rank, picaso_comm **out)
{
    picaso_comm *pc = new picaso_comm();
    pc->comm = c;
    pc->ctx = ctx;
    pc->nranks = nranks;
    pc->rank = rank;
    hipError_t e = hipMalloc((void **)&pc->scratch, sizeof(double) * (size_t)(nranks + 1));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&pc->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pc->ready, hipEventDisableTiming);
    for (int i = 0; i < picaso_comm::NSLOT && e == hipSuccess; ++i)
        e = hipEventCreateWithFlags(&pc->done[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        comm_release(pc);
        return fail(ctx, "picaso_comm: set-up failed: %s", hipGetErrorString(e));
    }
    *out = pc;
    return 0;
}

// an error inside an open ncclGroupStart / End must close the group before returning: every later collective of
// this thread would otherwise join a group that never ends
#define PZ_NCCL_IN_GROUP(ctx, expr)                                                               \
    do {                                                                                          \
        ncclResult_t r__ = (expr);                                                                \
        if (r__ != ncclSuccess) {                                                                 \
            (void)ncclGroupEnd();                                                                 \
            return pz::fail(ctx, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, \
                            __LINE__);                                                            \
        }                                                                                         \
    } while (0)

int picaso_comm_init_rank(picaso_ctx *ctx, int nranks, int rank, const void *id128, picaso_comm **out)
{
    if (!ctx || !id128 || !out) return fail(ctx, "picaso_comm_init_rank: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks)
        return fail(ctx, "picaso_comm_init_rank: rank %d of %d", rank, nranks);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c;
    PZ_NCCL(ctx, ncclCommInitRank(&c, nranks, id, rank));
    return comm_finish(ctx, c, nranks, rank, out);
}

int picaso_comm_init_all(int ndev, picaso_ctx *const *ctxs, picaso_comm **out)
{
    if (ndev < 1 || !ctxs || !out) return fail(nullptr, "picaso_comm_init_all: bad arguments");
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) {
        if (!ctxs[i]) return fail(nullptr, "picaso_comm_init_all: null context %d", i);
        devs[i] = ctxs[i]->device;
    }
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (devs[i] == devs[j])
                return fail(ctxs[0], "picaso_comm_init_all: device %d listed twice (RCCL takes one rank per device)", devs[i]);
    std::vector<ncclComm_t> cs(ndev);
    PZ_NCCL(ctxs[0], ncclCommInitAll(cs.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; ++i) out[i] = nullptr;
    for (int i = 0; i < ndev; ++i) {
        int rc = (hipSetDevice(devs[i]) == hipSuccess) ? comm_finish(ctxs[i], cs[i], ndev, i, &out[i])
                                                       : fail(ctxs[i], "picaso_comm_init_all: hipSetDevice(%d) failed", devs[i]);
        if (rc != 0) {                       // nothing half-built is left behind
            for (int j = 0; j < ndev; ++j) {
                if (out[j]) comm_release(out[j]);
                else if (j > i) (void)ncclCommDestroy(cs[j]);
                out[j] = nullptr;
            }
            return rc;
        }
    }
    return 0;
}

void picaso_comm_destroy(picaso_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_release(c);
}

int picaso_comm_rank(const picaso_comm *c, int *rank, int *nranks)
{
    if (!c) return fail(nullptr, "picaso_comm_rank: null communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return 0;
}

// recv[r*count + i] = rank r's send[i]; device pointers, enqueued on the context's stream.
int picaso_all_gather_dev(picaso_comm *c, const double *send, double *recv, size_t count)
{
    if (!c) return fail(nullptr, "picaso_all_gather_dev: null communicator");
    picaso_ctx *ctx = c->ctx;
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PZ_TRY(picaso_comm_wait_slot(c, -1));
    PZ_NCCL(ctx, ncclAllGather(send, recv, count, ncclDouble, c->comm, ctx->stream));
    return 0;
}

// The same gather off the compute stream: it starts when everything enqueued on the context's stream so
// far has finished (device-side event) and runs on the communicator's own stream, so the context's
// stream can go on with the next spectrum while the shards of this one travel.  `slot` (0..3) names
// the result buffer: picaso_comm_wait_slot makes the context's stream wait for the last gather of that
// slot before the buffer is written or read again.  counts == NULL: equal blocks of `count`.
int picaso_all_gather_async_dev(picaso_comm *c, const double *send, double *recv, size_t count,
                                const size_t *counts, const size_t *displs, int slot)
{
    if (!c) return fail(nullptr, "picaso_all_gather_async_dev: null communicator");
    picaso_ctx *ctx = c->ctx;
    if (slot < 0 || slot >= picaso_comm::NSLOT) return fail(ctx, "picaso_all_gather_async_dev: slot must be 0..3");
    if (!send || !recv) return fail(ctx, "picaso_all_gather_async_dev: null buffer");
    if (counts && !displs) return fail(ctx, "picaso_all_gather_async_dev: counts without displs");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PZ_HIP(ctx, hipEventRecord(c->ready, ctx->stream));
    PZ_HIP(ctx, hipStreamWaitEvent(c->stream, c->ready, 0));
    if (!counts) {
        PZ_NCCL(ctx, ncclAllGather(send, recv, count, ncclDouble, c->comm, c->stream));
    } else {
        PZ_NCCL(ctx, ncclGroupStart());
        for (int r = 0; r < c->nranks; ++r) {
            const double *src = (r == c->rank) ? send : recv + displs[r];
            PZ_NCCL_IN_GROUP(ctx, ncclBroadcast(src, recv + displs[r], counts[r], ncclDouble, r, c->comm, c->stream));
        }
        PZ_NCCL(ctx, ncclGroupEnd());
    }
    PZ_HIP(ctx, hipEventRecord(c->done[slot], c->stream));
    c->pending[slot] = true;
    return 0;
}

// n spectra in ONE collective launch (ncclGroupStart / End around n all-gathers): each spectrum still lands
// contiguous in its own receive buffer, and the per-collective cost (two stream events, one RCCL launch) is paid
// once per n spectra.  Same ordering rules and slots as picaso_all_gather_async_dev.
int picaso_all_gather_multi_async_dev(picaso_comm *c, int n, const double *const *send, double *const *recv,
                                      size_t count, const size_t *counts, const size_t *displs, int slot)
{
    if (!c) return fail(nullptr, "picaso_all_gather_multi_async_dev: null communicator");
    picaso_ctx *ctx = c->ctx;
    if (slot < 0 || slot >= picaso_comm::NSLOT) return fail(ctx, "picaso_all_gather_multi_async_dev: slot must be 0..3");
    if (n < 1 || !send || !recv) return fail(ctx, "picaso_all_gather_multi_async_dev: bad arguments");
    if (counts && !displs) return fail(ctx, "picaso_all_gather_multi_async_dev: counts without displs");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PZ_HIP(ctx, hipEventRecord(c->ready, ctx->stream));
    PZ_HIP(ctx, hipStreamWaitEvent(c->stream, c->ready, 0));
    PZ_NCCL(ctx, ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        if (!counts) {
            PZ_NCCL_IN_GROUP(ctx, ncclAllGather(send[i], recv[i], count, ncclDouble, c->comm, c->stream));
        } else {
            for (int r = 0; r < c->nranks; ++r) {
                const double *src = (r == c->rank) ? send[i] : recv[i] + displs[r];
                PZ_NCCL_IN_GROUP(ctx, ncclBroadcast(src, recv[i] + displs[r], counts[r], ncclDouble, r, c->comm, c->stream));
            }
        }
    }
    PZ_NCCL(ctx, ncclGroupEnd());
    PZ_HIP(ctx, hipEventRecord(c->done[slot], c->stream));
    c->pending[slot] = true;
    return 0;
}

// work enqueued on the context's stream from now on starts after the last asynchronous gather of `slot`
// (slot < 0: of every slot) has finished; device-side, the host does not block
int picaso_comm_wait_slot(picaso_comm *c, int slot)
{
    if (!c) return fail(nullptr, "picaso_comm_wait_slot: null communicator");
    picaso_ctx *ctx = c->ctx;
    if (slot >= picaso_comm::NSLOT) return fail(ctx, "picaso_comm_wait_slot: slot must be < 4");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < picaso_comm::NSLOT; ++i) {
        if ((slot >= 0 && i != slot) || !c->pending[i]) continue;
        PZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->done[i], 0));
        c->pending[i] = false;
    }
    return 0;
}

// Ragged shards: rank r contributes counts[r] elements that land at displs[r] of every rank's recv
// (one broadcast per rank inside a group: RCCL fuses them into one launch).
int picaso_all_gatherv_dev(picaso_comm *c, const double *send, double *recv, const size_t *counts,
                           const size_t *displs)
{
    if (!c || !counts || !displs) return fail(nullptr, "picaso_all_gatherv_dev: null argument");
    picaso_ctx *ctx = c->ctx;
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PZ_TRY(picaso_comm_wait_slot(c, -1));
    PZ_NCCL(ctx, ncclGroupStart());
    for (int r = 0; r < c->nranks; ++r) {
        const double *src = (r == c->rank) ? send : recv + displs[r];
        PZ_NCCL_IN_GROUP(ctx, ncclBroadcast(src, recv + displs[r], counts[r], ncclDouble, r, c->comm, ctx->stream));
    }
    PZ_NCCL(ctx, ncclGroupEnd());
    return 0;
}

/* --------------------------------------------------------------------------------------------------------
 * One thread driving several communicators (picaso_comm_init_all): RCCL requires the per-device calls of ONE
 * collective to sit inside one ncclGroupStart / End -- issued one communicator after the other, the first call
 * waits for peers that the same thread has not posted yet.  These entry points take all communicators of the
 * process at once; nothing is synchronised before every rank has been posted.
 * -------------------------------------------------------------------------------------------------------- */
static int group_check(int n, picaso_comm *const *cs, const char *what)
{
    if (n < 1 || !cs) return fail(nullptr, "%s: bad arguments", what);
    for (int i = 0; i < n; ++i) {
        if (!cs[i]) return fail(nullptr, "%s: null communicator %d", what, i);
        if (cs[i]->nranks != n || cs[i]->rank != i)
            return fail(cs[i]->ctx, "%s: communicator %d is rank %d of %d; pass all %d communicators of "
                        "picaso_comm_init_all in rank order", what, i, cs[i]->rank, cs[i]->nranks, n);
    }
    return 0;
}

// recv[i] (on device i) <- the blocks send[0..n-1] of all devices; counts == NULL: equal blocks of `count`.
// Enqueued on every context's own stream behind the kernels that produced its block.
int picaso_all_gather_group_dev(int n, picaso_comm *const *cs, const double *const *send, double *const *recv,
                                size_t count, const size_t *counts, const size_t *displs)
{
    PZ_TRY(group_check(n, cs, "picaso_all_gather_group_dev"));
    if (!send || !recv) return fail(cs[0]->ctx, "picaso_all_gather_group_dev: null buffer list");
    if (counts && !displs) return fail(cs[0]->ctx, "picaso_all_gather_group_dev: counts without displs");
    for (int i = 0; i < n; ++i) {
        if (!send[i] || !recv[i]) return fail(cs[i]->ctx, "picaso_all_gather_group_dev: null buffer of rank %d", i);
        PZ_TRY(picaso_comm_wait_slot(cs[i], -1));
    }
    picaso_ctx *ctx0 = cs[0]->ctx;
    PZ_NCCL(ctx0, ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        picaso_ctx *ctx = cs[i]->ctx;
        if (hipSetDevice(ctx->device) != hipSuccess) {
            (void)ncclGroupEnd();
            return fail(ctx, "picaso_all_gather_group_dev: hipSetDevice(%d) failed", ctx->device);
        }
        if (!counts) {
            PZ_NCCL_IN_GROUP(ctx, ncclAllGather(send[i], recv[i], count, ncclDouble, cs[i]->comm, ctx->stream));
        } else {
            for (int r = 0; r < n; ++r) {
                const double *src = (r == i) ? send[i] : recv[i] + displs[r];
                PZ_NCCL_IN_GROUP(ctx, ncclBroadcast(src, recv[i] + displs[r], counts[r], ncclDouble, r, cs[i]->comm,
                                                    ctx->stream));
            }
        }
    }
    PZ_NCCL(ctx0, ncclGroupEnd());
    return 0;
}

// every device's stream has drained and every rank's reduction has arrived; values[i] <- reduction over ranks
static int group_allreduce_host(int n, picaso_comm *const *cs, double *values, ncclRedOp_t op, const char *what)
{
    PZ_TRY(group_check(n, cs, what));
    if (!values) return fail(cs[0]->ctx, "%s: null values", what);
    for (int i = 0; i < n; ++i) {
        picaso_ctx *ctx = cs[i]->ctx;
        PZ_HIP(ctx, hipSetDevice(ctx->device));
        PZ_TRY(picaso_comm_wait_slot(cs[i], -1));
        PZ_HIP(ctx, hipMemcpyAsync(cs[i]->scratch, values + i, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    PZ_NCCL(cs[0]->ctx, ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(cs[i]->ctx->device) != hipSuccess) {
            (void)ncclGroupEnd();
            return fail(cs[i]->ctx, "%s: hipSetDevice(%d) failed", what, cs[i]->ctx->device);
        }
        PZ_NCCL_IN_GROUP(cs[i]->ctx, ncclAllReduce(cs[i]->scratch, cs[i]->scratch, 1, ncclDouble, op, cs[i]->comm,
                                                   cs[i]->ctx->stream));
    }
    PZ_NCCL(cs[0]->ctx, ncclGroupEnd());
    for (int i = 0; i < n; ++i) {            // only now, with every rank posted, is anything waited for
        picaso_ctx *ctx = cs[i]->ctx;
        PZ_HIP(ctx, hipSetDevice(ctx->device));
        PZ_HIP(ctx, hipMemcpyAsync(values + i, cs[i]->scratch, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    for (int i = 0; i < n; ++i) {
        PZ_HIP(cs[i]->ctx, hipSetDevice(cs[i]->ctx->device));
        PZ_HIP(cs[i]->ctx, hipStreamSynchronize(cs[i]->ctx->stream));
    }
    return 0;
}
int picaso_comm_group_max(int n, picaso_comm *const *cs, double *values)
{
    return group_allreduce_host(n, cs, values, ncclMax, "picaso_comm_group_max");
}
int picaso_comm_group_barrier(int n, picaso_comm *const *cs)
{
    std::vector<double> ones((size_t)(n > 0 ? n : 1), 1.0);
    return group_allreduce_host(n, cs, ones.data(), ncclSum, "picaso_comm_group_barrier");
}

// max / sum of one host double over the ranks (timing reductions of the launcher); synchronises.
// ONE THREAD PER COMMUNICATOR: with several communicators in one thread (picaso_comm_init_all) use
// picaso_comm_group_max / picaso_comm_group_barrier, which post every rank before anything is waited for.
static int allreduce_host(picaso_comm *c, double *value, ncclRedOp_t op)
{
    if (!c || !value) return fail(nullptr, "picaso_comm reduce: null argument");
    picaso_ctx *ctx = c->ctx;
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PZ_TRY(picaso_comm_wait_slot(c, -1));       // collectives of one communicator never overlap each other
    PZ_HIP(ctx, hipMemcpyAsync(c->scratch, value, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PZ_NCCL(ctx, ncclAllReduce(c->scratch, c->scratch, 1, ncclDouble, op, c->comm, ctx->stream));
    PZ_HIP(ctx, hipMemcpyAsync(value, c->scratch, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int picaso_comm_max(picaso_comm *c, double *value) { return allreduce_host(c, value, ncclMax); }
int picaso_comm_sum(picaso_comm *c, double *value) { return allreduce_host(c, value, ncclSum); }

// every rank's stream has drained and every rank has arrived
int picaso_comm_barrier(picaso_comm *c)
{
    double one = 1.0;
    return allreduce_host(c, &one, ncclSum);
}

}  // extern "C"
