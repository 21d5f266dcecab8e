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
        delete pc;
        return fail(ctx, "picaso_comm: set-up failed: %s", hipGetErrorString(e));
    }
    *out = pc;
    return 0;
}

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
    std::vector<ncclComm_t> cs(ndev);
    PZ_NCCL(ctxs[0], ncclCommInitAll(cs.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; ++i) {
        PZ_HIP(ctxs[i], hipSetDevice(devs[i]));
        PZ_TRY(comm_finish(ctxs[i], cs[i], ndev, i, &out[i]));
    }
    return 0;
}

void picaso_comm_destroy(picaso_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    for (int i = 0; i < picaso_comm::NSLOT; ++i)
        if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
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
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PZ_HIP(ctx, hipEventRecord(c->ready, ctx->stream));
    PZ_HIP(ctx, hipStreamWaitEvent(c->stream, c->ready, 0));
    if (!counts) {
        PZ_NCCL(ctx, ncclAllGather(send, recv, count, ncclDouble, c->comm, c->stream));
    } else {
        if (!displs) return fail(ctx, "picaso_all_gather_async_dev: counts without displs");
        PZ_NCCL(ctx, ncclGroupStart());
        for (int r = 0; r < c->nranks; ++r) {
            const double *src = (r == c->rank) ? send : recv + displs[r];
            PZ_NCCL(ctx, ncclBroadcast(src, recv + displs[r], counts[r], ncclDouble, r, c->comm, c->stream));
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
            PZ_NCCL(ctx, ncclAllGather(send[i], recv[i], count, ncclDouble, c->comm, c->stream));
        } else {
            for (int r = 0; r < c->nranks; ++r) {
                const double *src = (r == c->rank) ? send[i] : recv[i] + displs[r];
                PZ_NCCL(ctx, ncclBroadcast(src, recv[i] + displs[r], counts[r], ncclDouble, r, c->comm, c->stream));
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
        PZ_NCCL(ctx, ncclBroadcast(src, recv + displs[r], counts[r], ncclDouble, r, c->comm, ctx->stream));
    }
    PZ_NCCL(ctx, ncclGroupEnd());
    return 0;
}

// max / sum of one host double over the ranks (timing reductions of the launcher); synchronises.
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
