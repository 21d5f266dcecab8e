// Host side of the C ABI (include/picaso_hip.h): context, memory plumbing, argument marshalling.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace pz {

thread_local char g_err[512] = {0};

int fail(picaso_ctx *ctx, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", buf);
    snprintf(g_err, sizeof(g_err), "%s", buf);
    return 1;
}

int arena_reset(picaso_ctx *ctx, size_t need)
{
    // host-pointer calls are synchronous: nothing from a previous call may still use the arena
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (need > ctx->arena_bytes) {
        if (ctx->arena) PZ_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
        const size_t want = align_up(need + (need >> 3), 1u << 20);
        PZ_HIP(ctx, hipMalloc((void **)&ctx->arena, want));
        ctx->arena_bytes = want;
    }
    ctx->arena_used = 0;
    return 0;
}

void *arena_take(picaso_ctx *ctx, size_t bytes)
{
    const size_t b = align_up(bytes);
    if (ctx->arena_used + b > ctx->arena_bytes) return nullptr;
    void *p = ctx->arena + ctx->arena_used;
    ctx->arena_used += b;
    return p;
}

static int small_ring_reserve(picaso_ctx *ctx)
{
    if (ctx->small_h) return 0;
    // all or nothing: small_h doubles as the "reserved" flag, so it is published only when the device ring and every
    // event exist; a failure half way frees what it got and the next call tries again from the start
    const size_t ring = picaso_ctx::SMALL_BYTES * picaso_ctx::NSMALL;
    char *h = nullptr, *d = nullptr;
    hipEvent_t ev[picaso_ctx::NSMALL];
    int nev = 0;
    hipError_t e = hipHostMalloc((void **)&h, ring, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&d, ring);
    for (; e == hipSuccess && nev < picaso_ctx::NSMALL; ++nev) e = hipEventCreateWithFlags(&ev[nev], hipEventDisableTiming);
    if (e != hipSuccess) {
        for (int i = 0; i < nev - 1; ++i) (void)hipEventDestroy(ev[i]);      // ev[nev - 1] is the one that failed
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        return fail(ctx, "small table ring: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < picaso_ctx::NSMALL; ++i) ctx->small_ev[i] = ev[i];
    ctx->small_d = (decltype(ctx->small_d))d;
    ctx->small_h = (decltype(ctx->small_h))h;
    return 0;
}

// device address the NEXT table_upload of `bytes` will return (tables that point into themselves are filled first)
int table_next_dev(picaso_ctx *ctx, size_t bytes, const char **dev)
{
    if (bytes <= picaso_ctx::SMALL_BYTES) {
        PZ_TRY(small_ring_reserve(ctx));
        *dev = ctx->small_d + (size_t)ctx->small_next * picaso_ctx::SMALL_BYTES;
    } else {
        *dev = ctx->ring_d + (size_t)ctx->ring_next * picaso_ctx::SLOT_BYTES;
    }
    return 0;
}

int table_upload(picaso_ctx *ctx, const void *host, size_t bytes, const void **dev)
{
    if (bytes > picaso_ctx::SLOT_BYTES)
        return fail(ctx, "geometry/profile table of %zu bytes exceeds the %zu-byte slot", bytes,
                    picaso_ctx::SLOT_BYTES);
    if (bytes <= picaso_ctx::SMALL_BYTES) {
        PZ_TRY(small_ring_reserve(ctx));
        const int s = ctx->small_next;
        ctx->small_next = (s + 1) % picaso_ctx::NSMALL;
        if (ctx->small_pending[s]) {
            PZ_HIP(ctx, hipEventSynchronize(ctx->small_ev[s]));
            ctx->small_pending[s] = false;
        }
        char *h = ctx->small_h + (size_t)s * picaso_ctx::SMALL_BYTES, *d = ctx->small_d + (size_t)s * picaso_ctx::SMALL_BYTES;
        memcpy(h, host, bytes);
        PZ_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        PZ_HIP(ctx, hipEventRecord(ctx->small_ev[s], ctx->stream));
        ctx->small_pending[s] = true;
        *dev = d;
        return 0;
    }
    const int s = ctx->ring_next;
    ctx->ring_next = (s + 1) % picaso_ctx::NSLOT;
    if (ctx->ring_pending[s]) {
        PZ_HIP(ctx, hipEventSynchronize(ctx->ring_ev[s]));
        ctx->ring_pending[s] = false;
    }
    char *h = ctx->ring_h + (size_t)s * picaso_ctx::SLOT_BYTES;
    char *d = ctx->ring_d + (size_t)s * picaso_ctx::SLOT_BYTES;
    memcpy(h, host, bytes);
    PZ_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    PZ_HIP(ctx, hipEventRecord(ctx->ring_ev[s], ctx->stream));
    ctx->ring_pending[s] = true;
    *dev = d;
    return 0;
}

// split `n` angles into ceil(n/MAX_ANGLES) nearly equal chunks
static std::vector<int> angle_chunks(int n)
{
    // 5 angles per launch keeps the LDS-resident sweep state at 70 KB per 256-thread block, i.e. two
    // blocks (two waves per SIMD) per CU; 6..8 angles are split 3+3 / 4+3 / 4+4
    int maxa = 5;
    if (const char *e = getenv("PICASO_AMD_MAX_ANGLES")) {   // tuning knob: angles fused per launch
        const int v = atoi(e);
        if (v >= 1 && v <= MAX_ANGLES) maxa = v;
    }
    const int k = (n + maxa - 1) / maxa;
    std::vector<int> out;
    int left = n;
    for (int i = 0; i < k; ++i) {
        const int c = (left + (k - i) - 1) / (k - i);
        out.push_back(c);
        left -= c;
    }
    return out;
}

// Run each angle as its own wave (grid.y) instead of carrying them in one lane?  Worth it while all
// angle-waves together still fit about one wave per SIMD (1024 SIMDs).  The thermal kernels and reflected
// launches of more than MAX_ANGLES angles decide with this product rule; the reflected kernel with few
// angles chooses among group sizes (reflected_angle_group below).
static bool spread_angles(long ncol, int nang, long limit = 1280L * 64)
{
    if (const char *e = getenv("PICASO_AMD_SPREAD_COLS")) limit = atol(e);
    return nang > 1 && ncol * nang <= limit;
}

// Reflected kernel: how many angles a wave carries (0 = all of them fused in one lane, the launch of the
// large grids).  A sweep is one long dependent chain, so below ~1 wave per SIMD the time of a launch is the time of
// ONE wave, which grows with the angles it carries.  Mid-size grids therefore run groups of g angles as separate
// workgroups (XCD-aware order, see k_reflected_toa) and the disk sum as a separate pass.  (Launches of up to one
// 64-column block per CU with the reference's default options take k_reflected_coop instead, reflected_1d_core.)
//
// The choice is made from a two-constant model of a wave's time per layer, fitted to steady-state measurements on
// the MI355X at 90 layers and 5 angles (PICASO_AMD_ANGLE_GROUP sweeps of round 2):
//     a wave that carries g angles and has its SIMD to itself:   t1(g) = T_SHARED + g * T_ANGLE        per layer
//     ... and shares the SIMD with a second wave of the launch:  t2(g) = PAIRED * t1(g)
// (g = 1, 2, 3 alone: 0.050 / 0.063 / 0.084 ms at 90 layers -> 0.37 + 0.19 g us per layer; paired 0.085 / 0.110 /
// 0.140 -> x 1.7; five angles fused with the state in registers 0.105-0.130.)  Every shape scales with the layer
// count alike, so only the RATIOS matter for the choice; what depends on the device is how many 256-thread
// workgroups a shape may have before it doubles up: one per CU (ctx->ncu) for t1, two per CU for t2, and the
// fused launch keeps one wave per SIMD up to 4 * ncu column-waves.  The result does not depend on the shape
// (explicit-fma arithmetic, disk sum in the reference's order).
static int reflected_angle_group(picaso_ctx *ctx, long ncol, int nang, int nspec = 1)
{
    if (const char *e = getenv("PICASO_AMD_ANGLE_GROUP")) return atoi(e);
    if (nang <= 1) return 0;
    // a batched launch: every spectrum keeps whole workgroups, so the counts are per spectrum x nspec
    const long colwaves = ((ncol + 255) / 256) * 4 * nspec, ncg = ((ncol + 255) / 256) * nspec, ncu = ctx->ncu;
    if (nang > MAX_ANGLES) return spread_angles(ncol * nspec, nang, 2560L * 64) ? 1 : 0;
    if (nspec == 1 ? (ncol + 63) / 64 > 4L * ncu : colwaves > 4L * ncu) return 0;   // more than one fused wave per SIMD: throughput regime
    constexpr double T_SHARED = 0.37, T_ANGLE = 0.19, PAIRED = 1.7;       // us per layer, see above
    auto t1 = [&](int g) { return T_SHARED + g * T_ANGLE; };
    // all angles in one lane, one wave per SIMD: the state is in registers up to three angles and for five (the BIG
    // instantiation) and runs at the model's rate; four and six to eight keep part of it in LDS (~25 % slower)
    double best = t1(nang) * ((nang <= 3 || nang == 5) ? 1.0 : 1.25);
    int group = 0;
    for (int g = 1; g <= 3 && g < nang; ++g) {
        const int ngroups = (nang + g - 1) / g;
        if (ngroups * g > MAX_ANGLES) continue;
        const long blocks = ncg * ngroups;
        const double t = blocks <= ncu ? t1(g) : blocks <= 2 * ncu ? PAIRED * t1(g) : 1e9;
        if (t < best) { best = t; group = g; }
    }
    return group;
}

static int check_phase_options(picaso_ctx *ctx, int single_phase, int multi_phase, int toon)
{
    // the reference raises UnboundLocalError for these; report cleanly instead
    if (single_phase < 0 || single_phase > 3)
        return fail(ctx, "single_phase must be 0..3 (cahoy, OTHG, TTHG, TTHG_ray), got %d", single_phase);
    if (multi_phase < 0 || multi_phase > 1)
        return fail(ctx, "multi_phase must be 0 (N=2) or 1 (N=1), got %d", multi_phase);
    if (toon < 0 || toon > 1)
        return fail(ctx, "toon_coefficients must be 0 (quadrature) or 1 (eddington), got %d", toon);
    return 0;
}

}  // namespace pz

using namespace pz;

extern "C" {

const char *picaso_version(void) { return "picaso_amd 0.1.0 (gfx950)"; }

const char *picaso_last_error(const picaso_ctx *ctx) { return ctx ? ctx->err : g_err; }

int picaso_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return 0;
}

int picaso_ctx_create(int device, picaso_ctx **out)
{
    *out = nullptr;
    int n = 0;
    if (picaso_device_count(&n) != 0 || n == 0) return fail(nullptr, "no HIP device visible");
    if (device < 0 || device >= n) return fail(nullptr, "device %d out of range (%d visible)", device, n);
    picaso_ctx *ctx = new picaso_ctx();
    ctx->device = device;
    PZ_HIP(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    PZ_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    ctx->ncu = prop.multiProcessorCount;
    PZ_HIP(nullptr, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    PZ_HIP(nullptr, hipEventCreate(&ctx->ev0));
    PZ_HIP(nullptr, hipEventCreate(&ctx->ev1));
    const size_t ring = picaso_ctx::SLOT_BYTES * picaso_ctx::NSLOT;
    PZ_HIP(nullptr, hipHostMalloc((void **)&ctx->ring_h, ring, hipHostMallocDefault));
    PZ_HIP(nullptr, hipMalloc((void **)&ctx->ring_d, ring));
    for (int i = 0; i < picaso_ctx::NSLOT; ++i)
        PZ_HIP(nullptr, hipEventCreateWithFlags(&ctx->ring_ev[i], hipEventDisableTiming));
    *out = ctx;
    return 0;
}

void picaso_ctx_destroy(picaso_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool) (void)hipFree(kv.second);
    for (auto &kv : ctx->live) (void)hipFree(kv.first);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->lvl_scratch) (void)hipFree(ctx->lvl_scratch);
    if (ctx->ck_scratch) (void)hipFree(ctx->ck_scratch);
    if (ctx->ring_d) (void)hipFree(ctx->ring_d);
    if (ctx->ring_h) (void)hipHostFree(ctx->ring_h);
    if (ctx->small_d) (void)hipFree(ctx->small_d);
    if (ctx->small_h) (void)hipHostFree(ctx->small_h);
    for (int i = 0; i < picaso_ctx::NSMALL; ++i)
        if (ctx->small_ev[i]) (void)hipEventDestroy(ctx->small_ev[i]);
    for (int i = 0; i < picaso_ctx::NSLOT; ++i)
        if (ctx->ring_ev[i]) (void)hipEventDestroy(ctx->ring_ev[i]);
    if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    free_pairwise_plans(ctx);
    for (auto &kv : ctx->host_pool) (void)hipHostFree(kv.second);
    for (auto &kv : ctx->host_live) (void)hipHostFree(kv.first);
    for (hipEvent_t e : ctx->marks_free) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i)
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int pool_release_all(picaso_ctx *ctx)
{
    if (ctx->pool.empty()) return 0;
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->pool) (void)hipFree(kv.second);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    return 0;
}

int picaso_dev_malloc(picaso_ctx *ctx, size_t bytes, void **dptr)
{
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t size = align_up(bytes ? bytes : 8, 256);
    auto it = ctx->pool.find(size);
    if (it != ctx->pool.end()) {
        *dptr = it->second;
        ctx->pool.erase(it);
        ctx->pool_bytes -= size;
    } else {
        hipError_t e = hipMalloc(dptr, size);
        if (e != hipSuccess) {                       // out of memory: give the cached blocks back, retry
            (void)hipGetLastError();
            PZ_TRY(pool_release_all(ctx));
            PZ_HIP(ctx, hipMalloc(dptr, size));
        }
    }
    ctx->live[*dptr] = size;
    return 0;
}
int picaso_dev_free(picaso_ctx *ctx, void *dptr)
{
    if (!dptr) return 0;
    auto it = ctx->live.find(dptr);
    if (it == ctx->live.end()) return fail(ctx, "picaso_dev_free: %p was not allocated by picaso_dev_malloc", dptr);
    const size_t size = it->second;
    ctx->live.erase(it);
    if (ctx->pool_bytes + size > picaso_ctx::POOL_CAP) {
        PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PZ_HIP(ctx, hipFree(dptr));
        return 0;
    }
    ctx->pool.emplace(size, dptr);
    ctx->pool_bytes += size;
    return 0;
}
int picaso_ctx_mem_stats(picaso_ctx *ctx, size_t *out6)
{
    if (!ctx || !out6) return fail(ctx, "picaso_ctx_mem_stats: null argument");
    size_t live = 0, hbytes = 0;
    for (const auto &kv : ctx->live) live += kv.second;
    for (const auto &kv : ctx->host_live) hbytes += kv.second;
    for (const auto &kv : ctx->host_pool) hbytes += kv.first;
    out6[0] = live;
    out6[1] = ctx->live.size();
    out6[2] = ctx->pool_bytes;
    out6[3] = ctx->pool.size();
    out6[4] = hbytes;
    out6[5] = ctx->host_live.size() + ctx->host_pool.size();
    return 0;
}
int picaso_pool_trim(picaso_ctx *ctx)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    return pool_release_all(ctx);
}
// Host <-> device copies of up to STAGE_MAX_COPY bytes go through the context's pinned bounce buffer
// in two alternating halves (the DMA of one half overlaps the host memcpy of the other).  Handing
// pageable memory straight to hipMemcpyAsync makes the runtime pin the caller's pages for transfers
// above ~1 MB; when the caller then frees that memory (numpy returns large arrays to the OS) the
// next HIP call pays for the teardown -- measured 26-29 ms per climate.get_fluxes call on the
// MI355X box against 0.3 ms for the staged copy of the same 1.9 MB.
static int stage_reserve(picaso_ctx *ctx)
{
    if (ctx->stage) return 0;
    PZ_HIP(ctx, hipHostMalloc((void **)&ctx->stage, picaso_ctx::STAGE_BYTES, hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) PZ_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
    return 0;
}

int picaso_memcpy_h2d(picaso_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return fail(nullptr, "null context");
    if (bytes == 0) return 0;
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes > picaso_ctx::STAGE_MAX_COPY) {
        PZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    PZ_TRY(stage_reserve(ctx));
    const size_t half = picaso_ctx::STAGE_BYTES / 2;
    size_t c = 0;
    for (size_t off = 0; off < bytes; off += half, ++c) {
        const size_t len = bytes - off < half ? bytes - off : half;
        char *h = ctx->stage + (c & 1) * half;
        if (c >= 2) PZ_HIP(ctx, hipEventSynchronize(ctx->stage_ev[c & 1]));   // this half's previous DMA is done
        memcpy(h, (const char *)src + off, len);
        PZ_HIP(ctx, hipMemcpyAsync((char *)dst + off, h, len, hipMemcpyHostToDevice, ctx->stream));
        PZ_HIP(ctx, hipEventRecord(ctx->stage_ev[c & 1], ctx->stream));
    }
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int picaso_memcpy_d2h(picaso_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return fail(nullptr, "null context");
    if (bytes == 0) return 0;
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes > picaso_ctx::STAGE_MAX_COPY) {
        PZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    PZ_TRY(stage_reserve(ctx));
    const size_t half = picaso_ctx::STAGE_BYTES / 2;
    const size_t n = (bytes + half - 1) / half;
    auto issue = [&](size_t c) -> int {
        const size_t off = c * half, len = bytes - off < half ? bytes - off : half;
        PZ_HIP(ctx, hipMemcpyAsync(ctx->stage + (c & 1) * half, (const char *)src + off, len, hipMemcpyDeviceToHost,
                                   ctx->stream));
        PZ_HIP(ctx, hipEventRecord(ctx->stage_ev[c & 1], ctx->stream));
        return 0;
    };
    PZ_TRY(issue(0));
    for (size_t c = 0; c < n; ++c) {
        if (c + 1 < n) PZ_TRY(issue(c + 1));               // the other half, already drained
        PZ_HIP(ctx, hipEventSynchronize(ctx->stage_ev[c & 1]));
        const size_t off = c * half, len = bytes - off < half ? bytes - off : half;
        memcpy((char *)dst + off, ctx->stage + (c & 1) * half, len);
    }
    return 0;
}
// Result copies that do not wait.  A retrieval enqueues the next spectra while the GPU is still solving the last ones:
// the copy of a result is put on the stream behind its kernel, lands in a pinned block, and the host waits for the
// mark of THAT copy only -- not for the stream, which by then holds the next spectra's launches.
int picaso_host_alloc(picaso_ctx *ctx, size_t bytes, void **hptr)
{
    if (!ctx || !hptr) return fail(ctx, "picaso_host_alloc: null argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t size = align_up(bytes ? bytes : 8, 4096);
    auto it = ctx->host_pool.find(size);
    if (it != ctx->host_pool.end()) {
        *hptr = it->second;
        ctx->host_pool.erase(it);
    } else {
        PZ_HIP(ctx, hipHostMalloc(hptr, size, hipHostMallocDefault));
    }
    ctx->host_live[*hptr] = size;
    return 0;
}
int picaso_host_free(picaso_ctx *ctx, void *hptr)
{
    if (!hptr) return 0;
    if (!ctx) return fail(nullptr, "null context");
    auto it = ctx->host_live.find(hptr);
    if (it == ctx->host_live.end()) return fail(ctx, "picaso_host_free: %p was not allocated by picaso_host_alloc", hptr);
    ctx->host_pool.emplace(it->second, hptr);          // reuse is ordered by the caller: it frees after picaso_mark_wait
    ctx->host_live.erase(it);
    return 0;
}
int picaso_memcpy_d2h_async(picaso_ctx *ctx, void *pinned_dst, const void *src, size_t bytes, void **mark)
{
    if (!ctx || !pinned_dst || !src || !mark) return fail(ctx, "picaso_memcpy_d2h_async: null argument");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev;
    if (!ctx->marks_free.empty()) {
        ev = ctx->marks_free.back();
        ctx->marks_free.pop_back();
    } else {
        PZ_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    if (bytes) PZ_HIP(ctx, hipMemcpyAsync(pinned_dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipEventRecord(ev, ctx->stream));
    *mark = (void *)ev;
    return 0;
}
int picaso_mark_wait(picaso_ctx *ctx, void *mark)
{
    if (!ctx || !mark) return fail(ctx, "picaso_mark_wait: null argument");
    hipEvent_t ev = (hipEvent_t)mark;
    hipError_t e = hipEventSynchronize(ev);
    ctx->marks_free.push_back(ev);                     // a mark is waited for once
    if (e != hipSuccess) return fail(ctx, "hipEventSynchronize failed: %s", hipGetErrorString(e));
    return 0;
}
int picaso_memcpy_d2d(picaso_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    PZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
int picaso_memcpy_h2d_2d(picaso_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch,
                         size_t width, size_t height)
{
    PZ_HIP(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice,
                                 ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int picaso_memcpy_d2h_2d(picaso_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch,
                         size_t width, size_t height)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PZ_HIP(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost,
                                 ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int picaso_memset(picaso_ctx *ctx, void *dst, int value, size_t bytes)
{
    PZ_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return 0;
}
int picaso_sync(picaso_ctx *ctx)
{
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int picaso_ctx_device(picaso_ctx *ctx, int *device)
{
    if (!ctx || !device) return fail(ctx, "picaso_ctx_device: null argument");
    *device = ctx->device;
    return 0;
}
int picaso_ctx_wait(picaso_ctx *waiter, picaso_ctx *signaller)
{
    if (!waiter || !signaller) return fail(nullptr, "null context");
    if (waiter == signaller) return 0;
    if (waiter->device != signaller->device) return fail(waiter, "picaso_ctx_wait: contexts on different devices");
    PZ_HIP(waiter, hipSetDevice(waiter->device));
    // ev1 of the signaller is only used between timer_start / timer_stop pairs on the host side;
    // a dedicated event keeps the two uses apart
    if (!signaller->wait_ev) PZ_HIP(waiter, hipEventCreateWithFlags(&signaller->wait_ev, hipEventDisableTiming));
    PZ_HIP(waiter, hipEventRecord(signaller->wait_ev, signaller->stream));
    PZ_HIP(waiter, hipStreamWaitEvent(waiter->stream, signaller->wait_ev, 0));
    return 0;
}
int picaso_timer_start(picaso_ctx *ctx)
{
    PZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}
int picaso_timer_stop(picaso_ctx *ctx, float *ms)
{
    PZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    PZ_HIP(ctx, hipEventSynchronize(ctx->ev1));
    PZ_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return 0;
}
void *picaso_stream(picaso_ctx *ctx) { return (void *)ctx->stream; }

/* ============================================================================================
 * reflected light
 * ============================================================================================ */
static ReflectedArgs::Angle make_refl_angle(double v0, double v1, double gw, double tw = 0.0)
{
    const double NL2E = -1.4426950408889634074;
    ReflectedArgs::Angle g;
    g.u0 = v0; g.u1 = v1;
    g.iu0 = 1.0 / v0;
    g.iu0sq = 1.0 / (v0 * v0);                  // as the reference forms it (fluxes.py:1155)
    g.nl0 = NL2E / v0; g.nl1 = NL2E / v1;
    g.nlm = NL2E * (1.0 / v0 + 1.0 / v1);
    g.wq2 = 2.0 * (v0 / (v0 + v1));
    g.q2 = (3.0 * 0.767 * 0.767 * v1 * v1 - 1.0) / 2.0;   // ubar2 = 0.767 (fluxes.py:1280)
    g.wgt = gw;
    g.wgt2 = tw;
    return g;
}

// A batched call (picaso_get_reflected_1d_batch_dev): host arrays of nspec device pointers and the geometries
struct ReflBatchHost {
    int nspec;
    const double *const *plane[11];     // dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray, dtau_og, tau_og, w0_og, cosb_og
    const double *const *surf_reflect, *const *F0PI;
    double *const *xint, *const *albedo;
    int ngeom;                          // 1: ubar0/ubar1 (numg,numt) and cos_theta[0] for every spectrum; nspec: one each
    const double *cos_theta;
};

static int reflected_1d_core(picaso_ctx *ctx, int nlevel, int nwno, int ncolper, long plane_pitch, int numg,
                                int numt, const double *dtau, const double *tau, const double *w0,
                                const double *cosb, const double *gcos2, const double *ftau_cld,
                                const double *ftau_ray, const double *dtau_og,
                                const double *tau_og, const double *w0_og, const double *cosb_og,
                                const double *surf_reflect, const double *ubar0,
                                const double *ubar1, double cos_theta, const double *F0PI,
                                int single_phase, int multi_phase, double frac_a, double frac_b,
                                double frac_c, double constant_back, double constant_forward,
                                int get_toa_intensity, int get_lvl_flux, int toon_coefficients,
                                double b_top, double *xint_at_top, double *flux_minus_all,
                                double *flux_plus_all, double *flux_minus_midpt_all,
                                double *flux_plus_midpt_all, const double *gweight,
                                const double *tweight, double *albedo, const ReflBatchHost *bt = nullptr)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1)
        return fail(ctx, "get_reflected_1d: bad sizes nlevel=%d nwno=%d numg=%d numt=%d", nlevel, nwno, numg, numt);
    // Planes the kernels can re-derive exactly may be NULL (picaso_reflected_1d_can_derive says when): tau / tau_og
    // (running sums of dtau / dtau_og), gcos2 (0.5 ftau_ray), and -- a column without cloud -- cosb, cosb_og, ftau_cld,
    // ftau_ray (0, 0, 0, 1) with dtau_og, w0_og (no delta-scaling: dtau, w0).  The cloud set goes together.
    if (!dtau || !w0) return fail(ctx, "get_reflected_1d: dtau and w0 are required");
    if (!bt) PZ_NEED(ctx, "get_reflected_1d", surf_reflect, ubar0, ubar1, F0PI, xint_at_top);   // a NULL device pointer would be a GPU fault
    const bool clear_set = !ftau_cld;
    if (clear_set ? (cosb || cosb_og || ftau_ray || gcos2) : (!cosb || !cosb_og || !ftau_ray))
        return fail(ctx, "get_reflected_1d: cosb, cosb_og, ftau_cld, ftau_ray are given together or all left out (gcos2 "
                         "too in the second case)");
    if ((dtau_og == nullptr) != (w0_og == nullptr))
        return fail(ctx, "get_reflected_1d: dtau_og and w0_og are given together or both left out");
    const bool derived = !tau || !tau_og || !gcos2 || !ftau_cld || !dtau_og;
    if (derived && (get_lvl_flux || !get_toa_intensity))
        return fail(ctx, "get_reflected_1d: level fluxes need all eleven planes");
    const int nspec = bt ? bt->nspec : 1;
    const long ncol = (long)nwno * ncolper;
    if (plane_pitch < ncol) return fail(ctx, "get_reflected_1d: plane_pitch %ld < %ld columns", plane_pitch, ncol);
    if (ncolper > 1 && albedo)
        return fail(ctx, "get_reflected_1d: the fused disk sum is a per-wavelength output (ngauss = 1)");
    PZ_TRY(check_phase_options(ctx, single_phase, multi_phase, toon_coefficients));
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nang = numg * numt;
    if (get_lvl_flux) {
        if (!flux_minus_all || !flux_plus_all || !flux_minus_midpt_all || !flux_plus_midpt_all)
            return fail(ctx, "get_reflected_1d: get_lvl_flux=1 needs the four level-flux outputs");
    }
    if (!get_toa_intensity) {   // reference returns zeros (fluxes.py:1113, 1262)
        PZ_HIP(ctx, hipMemsetAsync(xint_at_top, 0, sizeof(double) * (size_t)nang * ncol, ctx->stream));
        if (albedo) PZ_HIP(ctx, hipMemsetAsync(albedo, 0, sizeof(double) * nwno, ctx->stream));
        if (!get_lvl_flux) return 0;
    }
    ReflectedArgs a{};
    a.nlayer = nlevel - 1;
    a.ncol = ncol;
    a.ncolper = ncolper;
    a.pitch = plane_pitch;
    a.nfac = 1;
    a.nwno = nwno;
    a.dtau = dtau; a.tau = tau; a.w0 = w0; a.cosb = cosb; a.gcos2 = gcos2; a.ftau_cld = ftau_cld;
    a.ftau_ray = ftau_ray; a.dtau_og = dtau_og; a.tau_og = tau_og; a.w0_og = w0_og; a.cosb_og = cosb_og;
    a.surf_reflect = surf_reflect; a.F0PI = F0PI; a.cos_theta = cos_theta;
    a.single_phase = single_phase; a.multi_phase = multi_phase; a.toon_coefficients = toon_coefficients;
    a.frac_a = frac_a; a.frac_b = frac_b; a.frac_c = frac_c; a.constant_back = constant_back;
    a.constant_forward = constant_forward; a.b_top = b_top;
    const bool fuse = albedo && gweight && tweight;
    a.albedo = fuse ? albedo : nullptr;
    a.albedo_scale = ((numt == 1) ? 2.0 * 3.14159265358979323846 : 1.0) * 0.5;   // disco.py:140-141
    // Batched launch: one table entry per spectrum with the angles [first, first + count) of this launch chunk
    // (slot k of the chunk = reference angle first + k, or a copy of the last angle when the chunk is padded);
    // xint rows are offset like the single launch's.  The table goes through the ring of pinned slots.
    std::vector<ReflBatchItem> items;
    auto upload_batch = [&](int first, int count, bool weights) -> int {
        items.resize((size_t)nspec);
        bool zp = true, ct1 = true;
        for (int s = 0; s < nspec; ++s) {
            ReflBatchItem &it = items[(size_t)s];
            const double *const *P[11];
            for (int j = 0; j < 11; ++j) P[j] = bt->plane[j];
            it.dtau = P[0][s]; it.tau = P[1][s]; it.w0 = P[2][s]; it.cosb = P[3][s]; it.gcos2 = P[4][s];
            it.ftau_cld = P[5][s]; it.ftau_ray = P[6][s]; it.dtau_og = P[7][s]; it.tau_og = P[8][s];
            it.w0_og = P[9][s]; it.cosb_og = P[10][s];
            it.surf_reflect = bt->surf_reflect[s]; it.F0PI = bt->F0PI[s];
            it.xint = bt->xint[s] + (size_t)first * ncol;
            it.albedo = (fuse && weights) ? bt->albedo[s] : nullptr;
            it.cos_theta = bt->cos_theta[bt->ngeom > 1 ? s : 0];
            it.u0_tab = it.u1_tab = nullptr;
            ct1 = ct1 && it.cos_theta == 1.0;
            const double *u0 = ubar0 + (bt->ngeom > 1 ? (size_t)s * nang : 0), *u1 = ubar1 + (bt->ngeom > 1 ? (size_t)s * nang : 0);
            for (int k = 0; k < count; ++k) {
                const int idx = first + k < nang ? first + k : nang - 1;
                it.ang[k] = make_refl_angle(u0[idx], u1[idx], weights ? gweight[idx / numt] : 0.0,
                                            weights ? tweight[idx % numt] : 0.0);
                zp = zp && u0[idx] == u1[idx];
            }
        }
        const void *d = nullptr;
        PZ_TRY(table_upload(ctx, items.data(), sizeof(ReflBatchItem) * items.size(), &d));
        a.batch = (const ReflBatchItem *)d;
        a.nspec = nspec;
        a.batch_zp = zp ? 1 : 0;
        // the compile-time default-options kernel of the symmetric geometry fixes cos_theta = 1: only when every
        // spectrum's is (fast_options looks at a.cos_theta)
        a.cos_theta = ct1 ? 1.0 : bt->cos_theta[0] == 1.0 ? 2.0 : bt->cos_theta[0];
        // every spectrum reads the same planes (one atmosphere under several geometries): XCD-sharing order
        bool shared = nspec > 1;
        for (int s = 1; s < nspec && shared; ++s)
            for (int j = 0; j < 11; ++j) shared = shared && bt->plane[j][s] == bt->plane[j][0];
        a.batch_interleave = shared ? 1 : 0;
        return 0;
    };
    if (bt) {
        if (get_lvl_flux || !get_toa_intensity)
            return fail(ctx, "get_reflected_1d_batch: level fluxes are a per-spectrum call (get_toa_intensity=1, get_lvl_flux=0)");
        if (sizeof(ReflBatchItem) * (size_t)nspec > picaso_ctx::SLOT_BYTES)
            return fail(ctx, "get_reflected_1d_batch: at most %zu spectra per call", picaso_ctx::SLOT_BYTES / sizeof(ReflBatchItem));
        if (nang > MAX_ANGLES) return fail(ctx, "get_reflected_1d_batch: at most %d disk angles", MAX_ANGLES);
    }
    if (get_lvl_flux) {   // two-sweep kernel, one angle per launch (fluxes.py:1219-1257)
        const size_t plane = (size_t)(nlevel - 1) * ncol;
        PZ_TRY(lvl_scratch_reserve(ctx, sizeof(double) * 4 * plane));
        ReflectedLvlArgs la{};
        la.base = a;
        la.base.na = 1;
        la.base.albedo = nullptr;
        la.scratch = ctx->lvl_scratch;
        const size_t lv = (size_t)nlevel * ncol;
        for (int idx = 0; idx < nang; ++idx) {
            const double v0 = ubar0[idx], v1 = ubar1[idx];
            la.base.ang[0] = make_refl_angle(v0, v1, 0.0);
            la.fm = flux_minus_all + idx * lv; la.fp = flux_plus_all + idx * lv;
            la.fmm = flux_minus_midpt_all + idx * lv; la.fpm = flux_plus_midpt_all + idx * lv;
            PZ_TRY(launch_reflected_lvl(ctx, la));
        }
        if (!get_toa_intensity) return 0;
    }
    // Small launches with the reference's default options: the cooperative kernel (one workgroup per 64 columns:
    // a wave for the angle-independent layer quantities, one wave per disk angle, fused disk sum), bit-identical
    // to the fused launch.  Where it stops paying: DESIGN.md section 7.
    if (nang <= MAX_ANGLES && !bt) {
        long coop_cols = 64L * ctx->ncu;                   // one workgroup (64 columns) per CU
        if (const char *e = getenv("PICASO_AMD_REFL_COOP_COLS")) coop_cols = atol(e);
        a.na = nang;
        a.ny = 1;
        if (ncol <= coop_cols && reflected_coop_ok(a)) {       // (also with the product's two derived plane sets)
            for (int k = 0; k < nang; ++k)
                a.ang[k] = make_refl_angle(ubar0[k], ubar1[k], fuse ? gweight[k / numt] : 0.0,
                                           fuse ? tweight[k % numt] : 0.0);
            a.xint = xint_at_top;
            a.albedo_first = a.albedo_last = 1;
            return launch_reflected_coop(ctx, a);
        }
    }
    int done = 0;
    const int group = reflected_angle_group(ctx, ncol, nang, nspec);
    auto disk_pass = [&]() -> int {                        // separate disk sum of the angle-group shapes
        if (!fuse) return 0;
        if (!bt) return picaso_compress_disco_dev(ctx, nwno, cos_theta, xint_at_top, gweight, numg, tweight, numt, F0PI, albedo);
        for (int s = 0; s < nspec; ++s)
            PZ_TRY(picaso_compress_disco_dev(ctx, nwno, bt->cos_theta[bt->ngeom > 1 ? s : 0], bt->xint[s], gweight, numg,
                                             tweight, numt, bt->F0PI[s], bt->albedo[s]));
        return 0;
    };
    if (group > 1 && group < nang && ((nang + group - 1) / group) * group <= MAX_ANGLES) {
        // Mid-size grids: groups of `group` angles per wave (grid.y), the last group padded with a copy of
        // the last angle (its extra results are not stored); disk sum as a separate pass, like below.
        a.na = group;
        a.ny = (nang + group - 1) / group;
        a.nvalid = nang;
        a.albedo = nullptr;
        for (int k = 0; k < a.ny * group; ++k) {
            const int idx = k < nang ? k : nang - 1;
            a.ang[k] = make_refl_angle(ubar0[idx], ubar1[idx], 0.0);
        }
        a.xint = xint_at_top;
        if (bt) PZ_TRY(upload_batch(0, a.ny * group, false));
        PZ_TRY(launch_reflected_toa(ctx, a, false));
        return disk_pass();
    }
    if (group == 1) {
        // Few columns: the chip is far from full and a lane's serial instruction stream sets the
        // latency, so every angle runs as its own wave (grid.y) and the disk sum is a separate pass.
        a.na = 1;
        a.albedo = nullptr;
        while (done < nang) {
            const int m = (nang - done < MAX_ANGLES) ? nang - done : MAX_ANGLES;
            for (int k = 0; k < m; ++k) a.ang[k] = make_refl_angle(ubar0[done + k], ubar1[done + k], 0.0);
            a.ny = m;
            a.nvalid = m;
            a.xint = xint_at_top + (size_t)done * ncol;
            if (bt) PZ_TRY(upload_batch(done, m, false));
            PZ_TRY(launch_reflected_toa(ctx, a, false));
            done += m;
        }
        return disk_pass();
    }
    a.ny = 1;
    const auto chunks = angle_chunks(nang);
    for (size_t c = 0; c < chunks.size(); ++c) {
        a.na = chunks[c];
        for (int k = 0; k < a.na; ++k) {
            const int idx = done + k;
            const double v0 = ubar0[idx], v1 = ubar1[idx];
            a.ang[k] = make_refl_angle(v0, v1, fuse ? gweight[idx / numt] : 0.0, fuse ? tweight[idx % numt] : 0.0);
        }
        a.xint = xint_at_top + (size_t)done * ncol;
        a.albedo_first = (c == 0);
        a.albedo_last = (c + 1 == chunks.size());
        if (bt) PZ_TRY(upload_batch(done, a.na, true));
        PZ_TRY(launch_reflected_toa(ctx, a, false));
        done += a.na;
    }
    return 0;
}

int picaso_reflected_1d_can_derive(int nlevel, long plane_pitch, int numg, int numt, const double *ubar0,
                                   const double *ubar1, double cos_theta, int single_phase, int multi_phase, double frac_c,
                                   int toon_coefficients, int get_lvl_flux)
{
    if (!ubar0 || !ubar1 || nlevel < 2 || numg < 1 || numt < 1 || get_lvl_flux) return 0;
    if (getenv("PICASO_AMD_REFL_GENERIC") || getenv("PICASO_AMD_REFL_ALL_PLANES")) return 0;
    if ((double)plane_pitch * nlevel * 8.0 >= 4294967296.0) return 0;
    bool zp = true;
    for (int k = 0; k < numg * numt; ++k) zp = zp && (ubar0[k] == ubar1[k]);
    return (toon_coefficients == 0 && single_phase == 3 && multi_phase == 0 && frac_c == 2.0 && (!zp || cos_theta == 1.0))
               ? 1 : 0;
}

int picaso_get_reflected_1d_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, int nwno, long plane_pitch, int numg,
                                      int numt, const double *const *dtau, const double *const *tau,
                                      const double *const *w0, const double *const *cosb,
                                      const double *const *gcos2, const double *const *ftau_cld,
                                      const double *const *ftau_ray, const double *const *dtau_og,
                                      const double *const *tau_og, const double *const *w0_og,
                                      const double *const *cosb_og, const double *const *surf_reflect, int ngeom,
                                      const double *ubar0, const double *ubar1, const double *cos_theta,
                                      const double *const *F0PI, int single_phase, int multi_phase, double frac_a,
                                      double frac_b, double frac_c, double constant_back, double constant_forward,
                                      int toon_coefficients, double b_top, double *const *xint_at_top,
                                      const double *gweight, const double *tweight, double *const *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nspec < 1) return fail(ctx, "get_reflected_1d_batch: nspec must be >= 1, got %d", nspec);
    if (ngeom != 1 && ngeom != nspec)
        return fail(ctx, "get_reflected_1d_batch: ngeom must be 1 (one geometry for all) or nspec, got %d", ngeom);
    const double *const *pl[11] = {dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray, dtau_og, tau_og, w0_og, cosb_og};
    for (int j = 0; j < 11; ++j) {
        if (!pl[j]) return fail(ctx, "get_reflected_1d_batch: all eleven plane pointer arrays are required");
        for (int s = 0; s < nspec; ++s)                  // planes left out (see reflected_1d_core): by every spectrum or by none
            if ((pl[j][s] == nullptr) != (pl[j][0] == nullptr))
                return fail(ctx, "get_reflected_1d_batch: plane %d is given for some spectra and left out for others", j);
    }
    if (!surf_reflect || !F0PI || !xint_at_top || !ubar0 || !ubar1 || !cos_theta)
        return fail(ctx, "get_reflected_1d_batch: null argument");
    const bool fuse = albedo && gweight && tweight;
    for (int s = 0; s < nspec; ++s)
        if (!surf_reflect[s] || !F0PI[s] || !xint_at_top[s] || (fuse && !albedo[s]))
            return fail(ctx, "get_reflected_1d_batch: null per-spectrum pointer (spectrum %d)", s);
    ReflBatchHost bt{};
    bt.nspec = nspec;
    for (int j = 0; j < 11; ++j) bt.plane[j] = pl[j];
    bt.surf_reflect = surf_reflect; bt.F0PI = F0PI; bt.xint = xint_at_top; bt.albedo = albedo;
    bt.ngeom = ngeom; bt.cos_theta = cos_theta;
    return reflected_1d_core(ctx, nlevel, nwno, 1, plane_pitch, numg, numt, dtau[0], tau[0], w0[0], cosb[0], gcos2[0],
                             ftau_cld[0], ftau_ray[0], dtau_og[0], tau_og[0], w0_og[0], cosb_og[0], surf_reflect[0],
                             ubar0, ubar1, cos_theta[0], F0PI[0], single_phase, multi_phase, frac_a, frac_b, frac_c,
                             constant_back, constant_forward, 1, 0, toon_coefficients, b_top, xint_at_top[0], nullptr,
                             nullptr, nullptr, nullptr, gweight, tweight, fuse ? albedo[0] : nullptr, &bt);
}

int picaso_get_reflected_1d_dev(picaso_ctx *ctx, int nlevel, int nwno, long plane_pitch, int numg,
                                int numt, const double *dtau, const double *tau, const double *w0,
                                const double *cosb, const double *gcos2, const double *ftau_cld,
                                const double *ftau_ray, const double *dtau_og,
                                const double *tau_og, const double *w0_og, const double *cosb_og,
                                const double *surf_reflect, const double *ubar0,
                                const double *ubar1, double cos_theta, const double *F0PI,
                                int single_phase, int multi_phase, double frac_a, double frac_b,
                                double frac_c, double constant_back, double constant_forward,
                                int get_toa_intensity, int get_lvl_flux, int toon_coefficients,
                                double b_top, double *xint_at_top, double *flux_minus_all,
                                double *flux_plus_all, double *flux_minus_midpt_all,
                                double *flux_plus_midpt_all, const double *gweight,
                                const double *tweight, double *albedo)
{
    return reflected_1d_core(ctx, nlevel, nwno, 1, plane_pitch, numg, numt, dtau, tau, w0, cosb, gcos2,
                             ftau_cld, ftau_ray, dtau_og, tau_og, w0_og, cosb_og, surf_reflect, ubar0,
                             ubar1, cos_theta, F0PI, single_phase, multi_phase, frac_a, frac_b, frac_c,
                             constant_back, constant_forward, get_toa_intensity, get_lvl_flux,
                             toon_coefficients, b_top, xint_at_top, flux_minus_all, flux_plus_all,
                             flux_minus_midpt_all, flux_plus_midpt_all, gweight, tweight, albedo);
}

int picaso_get_reflected_1d_ck_dev(picaso_ctx *ctx, int nlevel, int nwno, int ngauss, int numg, int numt,
                                   const double *dtau, const double *tau, const double *w0,
                                   const double *cosb, const double *gcos2, const double *ftau_cld,
                                   const double *ftau_ray, const double *dtau_og, const double *tau_og,
                                   const double *w0_og, const double *cosb_og,
                                   const double *surf_reflect, const double *ubar0,
                                   const double *ubar1, double cos_theta, const double *F0PI,
                                   int single_phase, int multi_phase, double frac_a, double frac_b,
                                   double frac_c, double constant_back, double constant_forward,
                                   int get_toa_intensity, int get_lvl_flux, int toon_coefficients,
                                   double b_top, const double *gauss_wts, double *xint_at_top,
                                   double *flux_minus_all, double *flux_plus_all,
                                   double *flux_minus_midpt_all, double *flux_plus_midpt_all,
                                   const double *gweight, const double *tweight, double *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "get_reflected_1d_ck: ngauss must be 1..%d", MAX_CK_GAUSS);
    if (!gauss_wts) return fail(ctx, "get_reflected_1d_ck: gauss_wts is null");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_reflected_1d_ck: bad sizes");
    if (get_lvl_flux && !(flux_minus_all && flux_plus_all && flux_minus_midpt_all && flux_plus_midpt_all))
        return fail(ctx, "get_reflected_1d_ck: get_lvl_flux=1 needs the four level-flux outputs");
    const int nang = numg * numt;
    const long ncol = (long)nwno * ngauss;
    const size_t nx = (size_t)nang * ncol, nl = get_lvl_flux ? (size_t)nang * nlevel * ncol : 0;
    PZ_TRY(ck_scratch_reserve(ctx, sizeof(double) * (nx + 4 * nl)));
    double *xcol = ctx->ck_scratch, *l0 = xcol + nx;
    PZ_TRY(reflected_1d_core(ctx, nlevel, nwno, ngauss, ncol, numg, numt, dtau, tau, w0, cosb, gcos2, ftau_cld,
                             ftau_ray, dtau_og, tau_og, w0_og, cosb_og, surf_reflect, ubar0, ubar1,
                             cos_theta, F0PI, single_phase, multi_phase, frac_a, frac_b, frac_c,
                             constant_back, constant_forward, get_toa_intensity, get_lvl_flux,
                             toon_coefficients, b_top, xcol, nl ? l0 : nullptr, nl ? l0 + nl : nullptr,
                             nl ? l0 + 2 * nl : nullptr, nl ? l0 + 3 * nl : nullptr, nullptr, nullptr,
                             nullptr));
    // Gauss-point sums in ig order (justdoit.py:307-313)
    PZ_TRY(launch_weighted_colsum(ctx, nang, nwno, ngauss, gauss_wts, xcol, xint_at_top));
    if (nl) {
        double *outs[4] = {flux_minus_all, flux_plus_all, flux_minus_midpt_all, flux_plus_midpt_all};
        for (int j = 0; j < 4; ++j)
            PZ_TRY(launch_weighted_colsum(ctx, nang * nlevel, nwno, ngauss, gauss_wts, l0 + j * nl, outs[j]));
    }
    if (albedo && gweight && tweight)
        PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta, xint_at_top, gweight, numg, tweight, numt, F0PI, albedo));
    return 0;
}

int picaso_get_reflected_1d(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                            int numt, const double *dtau, const double *tau, const double *w0,
                            const double *cosb, const double *gcos2, const double *ftau_cld,
                            const double *ftau_ray, const double *dtau_og, const double *tau_og,
                            const double *w0_og, const double *cosb_og, const double *surf_reflect,
                            const double *ubar0, const double *ubar1, double cos_theta,
                            const double *F0PI, int single_phase, int multi_phase, double frac_a,
                            double frac_b, double frac_c, double constant_back,
                            double constant_forward, int get_toa_intensity, int get_lvl_flux,
                            int toon_coefficients, double b_top, double *xint_at_top,
                            double *flux_minus_all, double *flux_plus_all,
                            double *flux_minus_midpt_all, double *flux_plus_midpt_all)
{
    (void)wno;   // accepted but never read, exactly like the reference
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1)
        return fail(ctx, "get_reflected_1d: bad sizes nlevel=%d nwno=%d numg=%d numt=%d", nlevel, nwno, numg, numt);
    PZ_NEED(ctx, "get_reflected_1d", dtau, tau, w0, cosb, gcos2, ftau_cld, ftau_ray, dtau_og, tau_og, w0_og, cosb_og, surf_reflect,
            ubar0, ubar1, F0PI, xint_at_top);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nl = (size_t)(nlevel - 1) * nwno, nv = (size_t)nlevel * nwno;
    const size_t nang = (size_t)numg * numt;
    const size_t lvl_elems = get_lvl_flux ? 4 * nang * nv : 0;
    const size_t need = sizeof(double) * (9 * nl + 2 * nv + 2 * (size_t)nwno + nang * nwno + lvl_elems) + 64 * 256;
    PZ_TRY(arena_reset(ctx, need));
    const double *d_dtau, *d_tau, *d_w0, *d_cosb, *d_gcos2, *d_fc, *d_fr, *d_dto, *d_tauo, *d_w0o, *d_cbo, *d_rs, *d_f0;
    PZ_TRY(arena_upload(ctx, dtau, nl, &d_dtau));
    PZ_TRY(arena_upload(ctx, tau, nv, &d_tau));
    PZ_TRY(arena_upload(ctx, w0, nl, &d_w0));
    PZ_TRY(arena_upload(ctx, cosb, nl, &d_cosb));
    PZ_TRY(arena_upload(ctx, gcos2, nl, &d_gcos2));
    PZ_TRY(arena_upload(ctx, ftau_cld, nl, &d_fc));
    PZ_TRY(arena_upload(ctx, ftau_ray, nl, &d_fr));
    PZ_TRY(arena_upload(ctx, dtau_og, nl, &d_dto));
    PZ_TRY(arena_upload(ctx, tau_og, nv, &d_tauo));
    PZ_TRY(arena_upload(ctx, w0_og, nl, &d_w0o));
    PZ_TRY(arena_upload(ctx, cosb_og, nl, &d_cbo));
    PZ_TRY(arena_upload(ctx, surf_reflect, (size_t)nwno, &d_rs));
    PZ_TRY(arena_upload(ctx, F0PI, (size_t)nwno, &d_f0));
    double *d_x = (double *)arena_take(ctx, sizeof(double) * nang * nwno);
    double *d_lvl[4] = {nullptr, nullptr, nullptr, nullptr};
    if (get_lvl_flux)
        for (int j = 0; j < 4; ++j) d_lvl[j] = (double *)arena_take(ctx, sizeof(double) * nang * nv);
    if (!d_x || (get_lvl_flux && !d_lvl[3])) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_get_reflected_1d_dev(ctx, nlevel, nwno, nwno, numg, numt, d_dtau, d_tau, d_w0, d_cosb,
                                       d_gcos2, d_fc, d_fr, d_dto, d_tauo, d_w0o, d_cbo, d_rs, ubar0, ubar1,
                                       cos_theta, d_f0, single_phase, multi_phase, frac_a, frac_b, frac_c,
                                       constant_back, constant_forward, get_toa_intensity, get_lvl_flux,
                                       toon_coefficients, b_top, d_x, d_lvl[0], d_lvl[1], d_lvl[2], d_lvl[3],
                                       nullptr, nullptr, nullptr));
    PZ_HIP(ctx, hipMemcpyAsync(xint_at_top, d_x, sizeof(double) * nang * nwno, hipMemcpyDeviceToHost, ctx->stream));
    if (get_lvl_flux) {
        double *h[4] = {flux_minus_all, flux_plus_all, flux_minus_midpt_all, flux_plus_midpt_all};
        for (int j = 0; j < 4; ++j)
            PZ_HIP(ctx, hipMemcpyAsync(h[j], d_lvl[j], sizeof(double) * nang * nv, hipMemcpyDeviceToHost, ctx->stream));
    }
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int picaso_get_reflected_3d_dev(picaso_ctx *ctx, int nlevel, int nwno, int numg, int numt,
                                const double *dtau_3d, const double *tau_3d, const double *w0_3d,
                                const double *cosb_3d, const double *gcos2_3d,
                                const double *ftau_cld_3d, const double *ftau_ray_3d,
                                const double *dtau_og_3d, const double *tau_og_3d,
                                const double *w0_og_3d, const double *cosb_og_3d,
                                const double *surf_reflect, const double *ubar0,
                                const double *ubar1, double cos_theta, const double *F0PI,
                                int single_phase, int multi_phase, double frac_a, double frac_b,
                                double frac_c, double constant_back, double constant_forward,
                                double *xint_at_top, const double *gweight, const double *tweight,
                                double *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1)
        return fail(ctx, "get_reflected_3d: bad sizes nlevel=%d nwno=%d numg=%d numt=%d", nlevel, nwno, numg, numt);
    PZ_TRY(check_phase_options(ctx, single_phase, multi_phase, 0));
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    // planes compute_opacity derives exactly from others may be NULL (re-derived in the kernel, see picaso_hip.h)
    if (!dtau_3d || !w0_3d) return fail(ctx, "get_reflected_3d: dtau and w0 are required");
    const int ncld = (cosb_3d != nullptr) + (ftau_cld_3d != nullptr) + (ftau_ray_3d != nullptr) + (cosb_og_3d != nullptr);
    if (ncld != 0 && ncld != 4)
        return fail(ctx, "get_reflected_3d: cosb, ftau_cld, ftau_ray and cosb_og are given together or all NULL (no cloud)");
    if (ncld == 0 && gcos2_3d) return fail(ctx, "get_reflected_3d: gcos2 without ftau_ray");
    if ((dtau_og_3d != nullptr) != (w0_og_3d != nullptr))
        return fail(ctx, "get_reflected_3d: dtau_og and w0_og are given together or both NULL (no delta-scaling)");
    if (!dtau_og_3d && tau_og_3d) return fail(ctx, "get_reflected_3d: tau_og without dtau_og");
    const int nfac = numg * numt;
    std::vector<double> tab(2 * (size_t)nfac);
    for (int i = 0; i < nfac; ++i) { tab[i] = ubar0[i]; tab[nfac + i] = ubar1[i]; }
    const void *d_tab = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), sizeof(double) * tab.size(), &d_tab));
    ReflectedArgs a{};
    a.nlayer = nlevel - 1;
    a.ncol = (long)nwno * nfac;
    a.pitch = a.ncol;
    a.nfac = nfac;
    a.nwno = nwno;
    a.dtau = dtau_3d; a.tau = tau_3d; a.w0 = w0_3d; a.cosb = cosb_3d; a.gcos2 = gcos2_3d;
    a.ftau_cld = ftau_cld_3d; a.ftau_ray = ftau_ray_3d; a.dtau_og = dtau_og_3d; a.tau_og = tau_og_3d;
    a.w0_og = w0_og_3d; a.cosb_og = cosb_og_3d;
    a.surf_reflect = surf_reflect; a.F0PI = F0PI; a.cos_theta = cos_theta;
    a.single_phase = single_phase; a.multi_phase = multi_phase; a.toon_coefficients = 0;
    a.frac_a = frac_a; a.frac_b = frac_b; a.frac_c = frac_c; a.constant_back = constant_back;
    a.constant_forward = constant_forward; a.b_top = 0.0;
    a.na = 1;
    a.u0_tab = (const double *)d_tab;
    a.u1_tab = (const double *)d_tab + nfac;
    a.xint = xint_at_top;
    a.albedo = nullptr;
    PZ_TRY(launch_reflected_toa(ctx, a, true));
    if (albedo && gweight && tweight)
        PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta, xint_at_top, gweight, numg, tweight, numt, F0PI, albedo));
    return 0;
}

int picaso_get_reflected_3d_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, int nwno, int numg, int numt,
                                      const double *const *dtau_3d, const double *const *tau_3d,
                                      const double *const *w0_3d, const double *const *cosb_3d,
                                      const double *const *gcos2_3d, const double *const *ftau_cld_3d,
                                      const double *const *ftau_ray_3d, const double *const *dtau_og_3d,
                                      const double *const *tau_og_3d, const double *const *w0_og_3d,
                                      const double *const *cosb_og_3d, const double *const *surf_reflect,
                                      const double *ubar0, const double *ubar1, const double *cos_theta,
                                      const double *const *F0PI, int single_phase, int multi_phase, double frac_a,
                                      double frac_b, double frac_c, double constant_back, double constant_forward,
                                      double *const *xint_at_top, const double *gweight, const double *tweight,
                                      double *const *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nspec < 1) return fail(ctx, "get_reflected_3d_batch: nspec must be >= 1, got %d", nspec);
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_reflected_3d_batch: bad sizes");
    PZ_TRY(check_phase_options(ctx, single_phase, multi_phase, 0));
    if (!dtau_3d || !w0_3d || !surf_reflect || !F0PI || !xint_at_top || !ubar0 || !ubar1 || !cos_theta)
        return fail(ctx, "get_reflected_3d_batch: null argument");
    // a plane family is given for every spectrum or (NULL array) for none: the launch has ONE presence pattern
    const double *const *pl[11] = {dtau_3d, tau_3d, w0_3d, cosb_3d, gcos2_3d, ftau_cld_3d, ftau_ray_3d, dtau_og_3d,
                                   tau_og_3d, w0_og_3d, cosb_og_3d};
    for (int j = 0; j < 11; ++j)
        for (int s = 0; pl[j] && s < nspec; ++s)
            if (!pl[j][s]) return fail(ctx, "get_reflected_3d_batch: plane %d of spectrum %d is NULL (a plane family is "
                                           "given for every spectrum or left out as a whole)", j, s);
    const int ncld = (cosb_3d != nullptr) + (ftau_cld_3d != nullptr) + (ftau_ray_3d != nullptr) + (cosb_og_3d != nullptr);
    if (ncld != 0 && ncld != 4)
        return fail(ctx, "get_reflected_3d_batch: cosb, ftau_cld, ftau_ray and cosb_og are given together or all NULL");
    if (ncld == 0 && gcos2_3d) return fail(ctx, "get_reflected_3d_batch: gcos2 without ftau_ray");
    if ((dtau_og_3d != nullptr) != (w0_og_3d != nullptr))
        return fail(ctx, "get_reflected_3d_batch: dtau_og and w0_og are given together or both NULL");
    if (!dtau_og_3d && tau_og_3d) return fail(ctx, "get_reflected_3d_batch: tau_og without dtau_og");
    const bool fuse = albedo && gweight && tweight;
    for (int s = 0; s < nspec; ++s)
        if (!surf_reflect[s] || !F0PI[s] || !xint_at_top[s] || (fuse && !albedo[s]))
            return fail(ctx, "get_reflected_3d_batch: null per-spectrum pointer (spectrum %d)", s);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nfac = numg * numt;
    // table: nspec entries, then every spectrum's |ubar0|, |ubar1| facet tables
    const size_t head = align_up(sizeof(ReflBatchItem) * (size_t)nspec, 256);
    std::vector<char> tab(head + sizeof(double) * 2 * (size_t)nfac * nspec);
    if (tab.size() > picaso_ctx::SLOT_BYTES) return fail(ctx, "get_reflected_3d_batch: too many spectra for one call");
    ReflBatchItem *items = (ReflBatchItem *)tab.data();
    double *geo = (double *)(tab.data() + head);
    const char *dbase = nullptr;
    PZ_TRY(table_next_dev(ctx, tab.size(), &dbase));
    for (int s = 0; s < nspec; ++s) {
        ReflBatchItem &it = items[s];
        memset(&it, 0, sizeof(it));
        auto at = [&](int j) { return pl[j] ? pl[j][s] : nullptr; };
        it.dtau = at(0); it.tau = at(1); it.w0 = at(2); it.cosb = at(3); it.gcos2 = at(4); it.ftau_cld = at(5);
        it.ftau_ray = at(6); it.dtau_og = at(7); it.tau_og = at(8); it.w0_og = at(9); it.cosb_og = at(10);
        it.surf_reflect = surf_reflect[s]; it.F0PI = F0PI[s]; it.xint = xint_at_top[s]; it.albedo = nullptr;
        it.cos_theta = cos_theta[s];
        for (int i = 0; i < nfac; ++i) {
            geo[(size_t)2 * s * nfac + i] = ubar0[(size_t)s * nfac + i];
            geo[(size_t)(2 * s + 1) * nfac + i] = ubar1[(size_t)s * nfac + i];
        }
        it.u0_tab = (const double *)(dbase + head) + (size_t)2 * s * nfac;
        it.u1_tab = it.u0_tab + nfac;
    }
    const void *d = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), tab.size(), &d));
    if ((const char *)d != dbase) return fail(ctx, "get_reflected_3d_batch: table slot moved");
    ReflectedArgs a{};
    a.nlayer = nlevel - 1;
    a.ncol = (long)nwno * nfac;
    a.pitch = a.ncol;
    a.nfac = nfac;
    a.nwno = nwno;
    a.dtau = dtau_3d[0]; a.tau = tau_3d ? tau_3d[0] : nullptr; a.w0 = w0_3d[0]; a.cosb = cosb_3d ? cosb_3d[0] : nullptr;
    a.gcos2 = gcos2_3d ? gcos2_3d[0] : nullptr; a.ftau_cld = ftau_cld_3d ? ftau_cld_3d[0] : nullptr;
    a.ftau_ray = ftau_ray_3d ? ftau_ray_3d[0] : nullptr; a.dtau_og = dtau_og_3d ? dtau_og_3d[0] : nullptr;
    a.tau_og = tau_og_3d ? tau_og_3d[0] : nullptr; a.w0_og = w0_og_3d ? w0_og_3d[0] : nullptr;
    a.cosb_og = cosb_og_3d ? cosb_og_3d[0] : nullptr;
    a.single_phase = single_phase; a.multi_phase = multi_phase; a.toon_coefficients = 0;
    a.frac_a = frac_a; a.frac_b = frac_b; a.frac_c = frac_c; a.constant_back = constant_back;
    a.constant_forward = constant_forward; a.b_top = 0.0;
    a.na = 1;
    a.batch = (const ReflBatchItem *)d;
    a.nspec = nspec;
    PZ_TRY(launch_reflected_toa(ctx, a, true));
    if (fuse)
        for (int s = 0; s < nspec; ++s)
            PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta[s], xint_at_top[s], gweight, numg, tweight, numt,
                                             F0PI[s], albedo[s]));
    return 0;
}

int picaso_get_reflected_3d(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                            int numt, const double *dtau_3d, const double *tau_3d,
                            const double *w0_3d, const double *cosb_3d, const double *gcos2_3d,
                            const double *ftau_cld_3d, const double *ftau_ray_3d,
                            const double *dtau_og_3d, const double *tau_og_3d,
                            const double *w0_og_3d, const double *cosb_og_3d,
                            const double *surf_reflect, const double *ubar0, const double *ubar1,
                            double cos_theta, const double *F0PI, int single_phase, int multi_phase,
                            double frac_a, double frac_b, double frac_c, double constant_back,
                            double constant_forward, double *xint_at_top)
{
    (void)wno;
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1)
        return fail(ctx, "get_reflected_3d: bad sizes");
    PZ_NEED(ctx, "get_reflected_3d", dtau_3d, tau_3d, w0_3d, cosb_3d, gcos2_3d, ftau_cld_3d, ftau_ray_3d, dtau_og_3d, tau_og_3d,
            w0_og_3d, cosb_og_3d, surf_reflect, ubar0, ubar1, F0PI, xint_at_top);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nfac = (size_t)numg * numt;
    const size_t nl = (size_t)(nlevel - 1) * nwno * nfac, nv = (size_t)nlevel * nwno * nfac;
    const size_t need = sizeof(double) * (9 * nl + 2 * nv + 2 * (size_t)nwno + nfac * nwno) + 64 * 256;
    PZ_TRY(arena_reset(ctx, need));
    const double *d[11], *d_rs, *d_f0;
    const double *h[11] = {dtau_3d, tau_3d, w0_3d, cosb_3d, gcos2_3d, ftau_cld_3d, ftau_ray_3d,
                           dtau_og_3d, tau_og_3d, w0_og_3d, cosb_og_3d};
    for (int j = 0; j < 11; ++j) PZ_TRY(arena_upload(ctx, h[j], (j == 1 || j == 8) ? nv : nl, &d[j]));
    PZ_TRY(arena_upload(ctx, surf_reflect, (size_t)nwno, &d_rs));
    PZ_TRY(arena_upload(ctx, F0PI, (size_t)nwno, &d_f0));
    double *d_x = (double *)arena_take(ctx, sizeof(double) * nfac * nwno);
    if (!d_x) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_get_reflected_3d_dev(ctx, nlevel, nwno, numg, numt, d[0], d[1], d[2], d[3], d[4], d[5], d[6],
                                       d[7], d[8], d[9], d[10], d_rs, ubar0, ubar1, cos_theta, d_f0,
                                       single_phase, multi_phase, frac_a, frac_b, frac_c, constant_back,
                                       constant_forward, d_x, nullptr, nullptr, nullptr));
    PZ_HIP(ctx, hipMemcpyAsync(xint_at_top, d_x, sizeof(double) * nfac * nwno, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

/* ============================================================================================
 * thermal emission
 * ============================================================================================ */
// A batched call (picaso_get_thermal_1d_batch_dev): host arrays of nspec device pointers; tlevel / plevel of the
// core are then (nspec, nlevel) host tables and ubar1 (ngeom, numg, numt)
struct ThermalBatchHost {
    int nspec;
    const double *const *dtau, *const *w0, *const *cosb, *const *surf_reflect;
    double *const *flux, *const *disk;
    int ngeom;
};

static int thermal_1d_core(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int ncolper,
                              long plane_pitch, int numg, int numt, const double *tlevel,
                              const double *dtau, const double *w0, const double *cosb,
                              const double *plevel, const double *ubar1,
                              const double *surf_reflect, int hard_surface, const double *dwno,
                              int calc_type, double *flux_at_top, double *flux_minus,
                              double *flux_plus, double *flux_minus_mdpt, double *flux_plus_mdpt,
                              const double *gweight, const double *tweight, double *flux_disk,
                              const ThermalBatchHost *bt = nullptr)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1)
        return fail(ctx, "get_thermal_1d: bad sizes nlevel=%d nwno=%d numg=%d numt=%d", nlevel, nwno, numg, numt);
    if (!dtau || !w0 || !cosb) return fail(ctx, "get_thermal_1d: dtau, w0 and cosb are required");
    if (!bt) PZ_NEED(ctx, "get_thermal_1d", wno, tlevel, plevel, ubar1, surf_reflect, flux_at_top);
    const int nspec = bt ? bt->nspec : 1;
    const long ncol = (long)nwno * ncolper;
    if (plane_pitch < ncol) return fail(ctx, "get_thermal_1d: plane_pitch %ld < %ld columns", plane_pitch, ncol);
    if (calc_type != 0 && calc_type != 1) return fail(ctx, "get_thermal_1d: calc_type must be 0 or 1");
    if (calc_type == 1 && !dwno) return fail(ctx, "get_thermal_1d: calc_type=1 needs dwno");
    const bool want_lvl = flux_minus || flux_plus || flux_minus_mdpt || flux_plus_mdpt;
    if (want_lvl && !(flux_minus && flux_plus && flux_minus_mdpt && flux_plus_mdpt))
        return fail(ctx, "get_thermal_1d: pass all four level-flux outputs or none");
    if (ncolper > 1 && flux_disk)
        return fail(ctx, "get_thermal_1d: the fused disk sum is a per-wavelength output (ngauss = 1)");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nang = numg * numt;
    ThermalArgs a{};
    a.nlayer = nlevel - 1;
    a.ncol = ncol;
    a.ncolper = ncolper;
    a.pitch = plane_pitch;
    a.nfac = 1;
    a.nwno = nwno;
    a.wno = wno; a.dwno = dwno;
    a.dtau = dtau; a.w0 = w0; a.cosb = cosb; a.surf_reflect = surf_reflect;
    a.hard_surface = hard_surface; a.calc_type = calc_type;
    const bool fuse = flux_disk && gweight && tweight;
    a.disk = fuse ? flux_disk : nullptr;
    a.disk_scale = (numt == 1) ? 1.0 : 1.0 / (2.0 * 3.14159265358979323846);   // disco.py:174-175
    // Small launches: the cooperative kernel (helper waves compute the layer quantities into LDS, one
    // sweeper wave per 64 columns runs the recurrence for all angles; the level temperatures travel as
    // kernel arguments).  Step time of get_thermal_1d + compress_thermal at 1e4 x 90 x 5 (BASELINE
    // configs[1]) and where it stops paying: DESIGN.md section 7.
    // Batched launch: the table = nspec entries followed by every spectrum's level temperatures and pressures;
    // entry s carries the angles [first, first + count) of this launch chunk
    std::vector<char> btab;
    auto upload_batch = [&](int first, int count, bool weights) -> int {
        const size_t head = align_up(sizeof(ThermalBatchItem) * (size_t)nspec, 256);
        btab.resize(head + sizeof(double) * 2 * (size_t)nlevel * nspec);
        if (btab.size() > picaso_ctx::SLOT_BYTES) return fail(ctx, "get_thermal_1d_batch: too many spectra for one call");
        ThermalBatchItem *items = (ThermalBatchItem *)btab.data();
        double *lv = (double *)(btab.data() + head);
        for (int s = 0; s < nspec; ++s) {
            memcpy(lv + (size_t)2 * s * nlevel, tlevel + (size_t)s * nlevel, sizeof(double) * nlevel);
            memcpy(lv + (size_t)(2 * s + 1) * nlevel, plevel + (size_t)s * nlevel, sizeof(double) * nlevel);
        }
        const void *d = nullptr;
        // device addresses of the level tables are known only after the slot is chosen: fill, then upload once
        const char *dbase = nullptr;
        PZ_TRY(table_next_dev(ctx, btab.size(), &dbase));
        for (int s = 0; s < nspec; ++s) {
            ThermalBatchItem &it = items[s];
            it.dtau = bt->dtau[s]; it.w0 = bt->w0[s]; it.cosb = bt->cosb[s]; it.surf_reflect = bt->surf_reflect[s];
            it.tlevel = (const double *)(dbase + head) + (size_t)2 * s * nlevel;
            it.plevel = it.tlevel + nlevel;
            it.flux = bt->flux[s] + (size_t)first * ncol;
            it.disk = (weights && bt->disk) ? bt->disk[s] : nullptr;
            it.u1_tab = nullptr;
            const double *u1 = ubar1 + (bt->ngeom > 1 ? (size_t)s * nang : 0);
            for (int k = 0; k < count; ++k) it.u1[k] = u1[first + k];
        }
        PZ_TRY(table_upload(ctx, btab.data(), btab.size(), &d));
        if ((const char *)d != dbase) return fail(ctx, "get_thermal_1d_batch: table slot moved");
        a.batch = (const ThermalBatchItem *)d;
        a.nspec = nspec;
        return 0;
    };
    if (bt) {
        if (want_lvl) return fail(ctx, "get_thermal_1d_batch: level fluxes are a per-spectrum call");
        if (nang > MAX_ANGLES) return fail(ctx, "get_thermal_1d_batch: at most %d disk angles", MAX_ANGLES);
    }
    if (!want_lvl && !bt) {
        long coop_cols = 32768;
        if (const char *e = getenv("PICASO_AMD_THERMAL_COOP_COLS")) coop_cols = atol(e);
        a.na = nang;
        a.ny = 1;
        if (nang <= 5 && ncol <= coop_cols && thermal_coop_ok(a)) {
            for (int k = 0; k < nang; ++k) {
                a.u1[k] = ubar1[k];
                a.wgt[k] = fuse ? gweight[k / numt] : 0.0;
                a.wgt2[k] = fuse ? tweight[k % numt] : 0.0;
            }
            a.flux = flux_at_top;
            a.disk_first = a.disk_last = 1;
            return launch_thermal_coop(ctx, a, tlevel, plevel);
        }
    }
    std::vector<double> tab(2 * (size_t)nlevel + (size_t)nang);
    for (int i = 0; i < nlevel; ++i) { tab[i] = tlevel[i]; tab[nlevel + i] = plevel[i]; }
    for (int i = 0; i < nang; ++i) tab[2 * (size_t)nlevel + i] = ubar1[i];
    const void *d_tab = nullptr;
    if (!bt) PZ_TRY(table_upload(ctx, tab.data(), sizeof(double) * tab.size(), &d_tab));
    a.tlevel = (const double *)d_tab; a.plevel = (const double *)d_tab + nlevel;
    if (want_lvl) {   // the reference always fills these (fluxes.py:1851-1907): two-sweep kernel
        const size_t plane = (size_t)(nlevel - 1) * ncol;
        PZ_TRY(lvl_scratch_reserve(ctx, sizeof(double) * (4 * plane + (size_t)nlevel * ncol)));
        ThermalLvlArgs la{};
        la.base = a;
        la.nang = nang;
        la.u1_dev = (const double *)d_tab + 2 * (size_t)nlevel;
        la.flux = flux_at_top;
        la.fm = flux_minus; la.fp = flux_plus; la.fmm = flux_minus_mdpt; la.fpm = flux_plus_mdpt;
        la.scratch = ctx->lvl_scratch;
        PZ_TRY(launch_thermal_lvl(ctx, la));
        if (fuse)
            PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, flux_at_top, gweight, numg, tweight, numt, flux_disk));
        return 0;
    }
    int done = 0;
    if (spread_angles(ncol * nspec, nang)) {     // see reflected_1d_core
        a.na = 1;
        a.disk = nullptr;
        while (done < nang) {
            const int m = (nang - done < MAX_ANGLES) ? nang - done : MAX_ANGLES;
            for (int k = 0; k < m; ++k) { a.u1[k] = ubar1[done + k]; a.wgt[k] = a.wgt2[k] = 0.0; }
            a.ny = m;
            a.flux = flux_at_top + (size_t)done * ncol;
            if (bt) PZ_TRY(upload_batch(done, m, false));
            PZ_TRY(launch_thermal_toa(ctx, a, false));
            done += m;
        }
        if (fuse && !bt)
            PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, flux_at_top, gweight, numg, tweight, numt, flux_disk));
        if (fuse && bt)
            for (int s = 0; s < nspec; ++s)
                PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, bt->flux[s], gweight, numg, tweight, numt, bt->disk[s]));
        return 0;
    }
    a.ny = 1;
    const auto chunks = angle_chunks(nang);
    for (size_t c = 0; c < chunks.size(); ++c) {
        a.na = chunks[c];
        for (int k = 0; k < a.na; ++k) {
            const int idx = done + k;
            a.u1[k] = ubar1[idx];
            a.wgt[k] = fuse ? gweight[idx / numt] : 0.0;
            a.wgt2[k] = fuse ? tweight[idx % numt] : 0.0;
        }
        a.flux = flux_at_top + (size_t)done * ncol;
        a.disk_first = (c == 0);
        a.disk_last = (c + 1 == chunks.size());
        if (bt) PZ_TRY(upload_batch(done, a.na, true));
        PZ_TRY(launch_thermal_toa(ctx, a, false));
        done += a.na;
    }
    return 0;
}

int picaso_get_thermal_1d_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, const double *wno, int nwno,
                                    long plane_pitch, int numg, int numt, const double *tlevel,
                                    const double *const *dtau, const double *const *w0, const double *const *cosb,
                                    const double *plevel, int ngeom, const double *ubar1,
                                    const double *const *surf_reflect, int hard_surface, const double *dwno,
                                    int calc_type, double *const *flux_at_top, const double *gweight,
                                    const double *tweight, double *const *flux_disk)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nspec < 1) return fail(ctx, "get_thermal_1d_batch: nspec must be >= 1, got %d", nspec);
    if (ngeom != 1 && ngeom != nspec)
        return fail(ctx, "get_thermal_1d_batch: ngeom must be 1 (one geometry for all) or nspec, got %d", ngeom);
    if (!dtau || !w0 || !cosb || !surf_reflect || !flux_at_top || !tlevel || !plevel || !ubar1)
        return fail(ctx, "get_thermal_1d_batch: null argument");
    const bool fuse = flux_disk && gweight && tweight;
    for (int s = 0; s < nspec; ++s)
        if (!dtau[s] || !w0[s] || !cosb[s] || !surf_reflect[s] || !flux_at_top[s] || (fuse && !flux_disk[s]))
            return fail(ctx, "get_thermal_1d_batch: null per-spectrum pointer (spectrum %d)", s);
    ThermalBatchHost bt{};
    bt.nspec = nspec;
    bt.dtau = dtau; bt.w0 = w0; bt.cosb = cosb; bt.surf_reflect = surf_reflect;
    bt.flux = flux_at_top; bt.disk = fuse ? flux_disk : nullptr; bt.ngeom = ngeom;
    return thermal_1d_core(ctx, nlevel, wno, nwno, 1, plane_pitch, numg, numt, tlevel, dtau[0], w0[0], cosb[0], plevel,
                           ubar1, surf_reflect[0], hard_surface, dwno, calc_type, flux_at_top[0], nullptr, nullptr,
                           nullptr, nullptr, gweight, tweight, fuse ? flux_disk[0] : nullptr, &bt);
}

int picaso_get_thermal_1d_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno,
                              long plane_pitch, int numg, int numt, const double *tlevel,
                              const double *dtau, const double *w0, const double *cosb,
                              const double *plevel, const double *ubar1,
                              const double *surf_reflect, int hard_surface, const double *dwno,
                              int calc_type, double *flux_at_top, double *flux_minus,
                              double *flux_plus, double *flux_minus_mdpt, double *flux_plus_mdpt,
                              const double *gweight, const double *tweight, double *flux_disk)
{
    return thermal_1d_core(ctx, nlevel, wno, nwno, 1, plane_pitch, numg, numt, tlevel, dtau, w0, cosb, plevel,
                           ubar1, surf_reflect, hard_surface, dwno, calc_type, flux_at_top, flux_minus,
                           flux_plus, flux_minus_mdpt, flux_plus_mdpt, gweight, tweight, flux_disk);
}

int picaso_get_thermal_1d_ck_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int ngauss,
                                 int numg, int numt, const double *tlevel, const double *dtau,
                                 const double *w0, const double *cosb, const double *plevel,
                                 const double *ubar1, const double *surf_reflect, int hard_surface,
                                 const double *dwno, int calc_type, const double *gauss_wts,
                                 double *flux_at_top, double *flux_minus, double *flux_plus,
                                 double *flux_minus_mdpt, double *flux_plus_mdpt, const double *gweight,
                                 const double *tweight, double *flux_disk)
{
    if (!ctx) return fail(nullptr, "null context");
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "get_thermal_1d_ck: ngauss must be 1..%d", MAX_CK_GAUSS);
    if (!gauss_wts) return fail(ctx, "get_thermal_1d_ck: gauss_wts is null");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_1d_ck: bad sizes");
    const bool want_lvl = flux_minus || flux_plus || flux_minus_mdpt || flux_plus_mdpt;
    if (want_lvl && !(flux_minus && flux_plus && flux_minus_mdpt && flux_plus_mdpt))
        return fail(ctx, "get_thermal_1d_ck: pass all four level-flux outputs or none");
    const int nang = numg * numt;
    const long ncol = (long)nwno * ngauss;
    const size_t nx = (size_t)nang * ncol, nl = want_lvl ? (size_t)nang * nlevel * ncol : 0;
    PZ_TRY(ck_scratch_reserve(ctx, sizeof(double) * (nx + 4 * nl)));
    double *xcol = ctx->ck_scratch, *l0 = xcol + nx;
    PZ_TRY(thermal_1d_core(ctx, nlevel, wno, nwno, ngauss, ncol, numg, numt, tlevel, dtau, w0, cosb, plevel, ubar1,
                           surf_reflect, hard_surface, dwno, calc_type, xcol, nl ? l0 : nullptr,
                           nl ? l0 + nl : nullptr, nl ? l0 + 2 * nl : nullptr, nl ? l0 + 3 * nl : nullptr,
                           nullptr, nullptr, nullptr));
    PZ_TRY(launch_weighted_colsum(ctx, nang, nwno, ngauss, gauss_wts, xcol, flux_at_top));   // justdoit.py:380
    if (nl) {
        double *outs[4] = {flux_minus, flux_plus, flux_minus_mdpt, flux_plus_mdpt};
        for (int j = 0; j < 4; ++j)
            PZ_TRY(launch_weighted_colsum(ctx, nang * nlevel, nwno, ngauss, gauss_wts, l0 + j * nl, outs[j]));
    }
    if (flux_disk && gweight && tweight)
        PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, flux_at_top, gweight, numg, tweight, numt, flux_disk));
    return 0;
}

int picaso_get_thermal_1d_ck_tbatch_dev(picaso_ctx *ctx, int nitem, int nlevel, const double *wno, int nwno, int ngauss,
                                        int numg, int numt, const double *tlevel, const double *dtau, const double *w0,
                                        const double *cosb, const double *plevel, const double *ubar1,
                                        const double *surf_reflect, int hard_surface, const double *dwno, int calc_type,
                                        const double *gauss_wts, const double *gweight, const double *tweight,
                                        double *disk4)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nitem < 1 || nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_1d_ck_tbatch: bad sizes");
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "get_thermal_1d_ck_tbatch: ngauss must be 1..%d", MAX_CK_GAUSS);
    if (!wno || !tlevel || !dtau || !w0 || !cosb || !plevel || !ubar1 || !surf_reflect || !gauss_wts || !gweight ||
        !tweight || !disk4)
        return fail(ctx, "get_thermal_1d_ck_tbatch: null argument");
    if (calc_type != 0 && calc_type != 1) return fail(ctx, "get_thermal_1d_ck_tbatch: calc_type must be 0 or 1");
    if (calc_type == 1 && !dwno) return fail(ctx, "get_thermal_1d_ck_tbatch: calc_type=1 needs dwno");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nang = numg * numt, nlayer = nlevel - 1;
    const long per = (long)nwno * ngauss, ncol = per * nitem, nout = (long)nwno * nitem;
    std::vector<double> tab((size_t)nitem * nlevel + (size_t)nlevel + (size_t)nang);
    memcpy(tab.data(), tlevel, sizeof(double) * (size_t)nitem * nlevel);
    memcpy(tab.data() + (size_t)nitem * nlevel, plevel, sizeof(double) * nlevel);
    memcpy(tab.data() + (size_t)nitem * nlevel + nlevel, ubar1, sizeof(double) * nang);
    const void *d_tab = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), sizeof(double) * tab.size(), &d_tab));
    // per-column results before the Gauss sum: flux (nang, ncol) + four (nang, nlevel, ncol); then the Gauss-summed
    // four (nang, nlevel, nitem*nwno)
    const size_t nx = (size_t)nang * ncol, nl = (size_t)nang * nlevel * ncol, ng4 = (size_t)nang * nlevel * nout;
    PZ_TRY(lvl_scratch_reserve(ctx, sizeof(double) * ((size_t)4 * nlayer + nlevel) * ncol));
    PZ_TRY(ck_scratch_reserve(ctx, sizeof(double) * (nx + 4 * nl + 4 * ng4)));
    double *xcol = ctx->ck_scratch, *l0 = xcol + nx, *g0 = l0 + 4 * nl;
    ThermalLvlArgs la{};
    ThermalArgs &a = la.base;
    a.nlayer = nlayer;
    a.ncol = ncol;
    a.ncolper = ngauss;
    a.pitch = per;                                   // the planes hold ONE profile's columns
    a.per_item = per;
    a.nfac = 1;
    a.nwno = nwno;
    a.wno = wno; a.dwno = dwno;
    a.tlevel = (const double *)d_tab;
    a.plevel = a.tlevel + (size_t)nitem * nlevel;
    a.dtau = dtau; a.w0 = w0; a.cosb = cosb; a.surf_reflect = surf_reflect;
    a.hard_surface = hard_surface; a.calc_type = calc_type;
    la.nang = nang;
    la.u1_dev = a.plevel + nlevel;
    la.flux = xcol;
    la.fm = l0; la.fp = l0 + nl; la.fmm = l0 + 2 * nl; la.fpm = l0 + 3 * nl;
    la.scratch = ctx->lvl_scratch;
    PZ_TRY(launch_thermal_lvl(ctx, la));
    for (int j = 0; j < 4; ++j) {
        // Gauss-point sums (justdoit.py:380; the columns of a row are (profile, wavelength, Gauss point)), then the
        // disk sum over the angles (climate.py:1925-1928)
        PZ_TRY(launch_weighted_colsum(ctx, nang * nlevel, nout, ngauss, gauss_wts, l0 + j * nl, g0 + j * ng4));
        PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nlevel * nout, g0 + j * ng4, gweight, numg, tweight, numt,
                                           disk4 + (size_t)j * nlevel * nout));
    }
    return 0;
}

int picaso_axpby_dev(picaso_ctx *ctx, size_t n, double a, const double *x, double b, const double *y,
                     double *out)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    return launch_axpby(ctx, n, a, x, b, y, out);
}

int picaso_get_thermal_1d(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                          int numt, const double *tlevel, const double *dtau, const double *w0,
                          const double *cosb, const double *plevel, const double *ubar1,
                          const double *surf_reflect, int hard_surface, const double *dwno,
                          int calc_type, double *flux_at_top, double *flux_minus,
                          double *flux_plus, double *flux_minus_mdpt, double *flux_plus_mdpt)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_1d: bad sizes");
    PZ_NEED(ctx, "get_thermal_1d", wno, tlevel, dtau, w0, cosb, plevel, ubar1, surf_reflect, flux_at_top);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nl = (size_t)(nlevel - 1) * nwno, nv = (size_t)nlevel * nwno;
    const size_t nang = (size_t)numg * numt;
    const bool lvl = flux_minus || flux_plus || flux_minus_mdpt || flux_plus_mdpt;
    const size_t need = sizeof(double) * (3 * nl + 3 * (size_t)nwno + nang * nwno + (lvl ? 4 * nang * nv : 0)) + 64 * 256;
    PZ_TRY(arena_reset(ctx, need));
    const double *d_dtau, *d_w0, *d_cosb, *d_rs, *d_wno, *d_dwno = nullptr;
    PZ_TRY(arena_upload(ctx, dtau, nl, &d_dtau));
    PZ_TRY(arena_upload(ctx, w0, nl, &d_w0));
    PZ_TRY(arena_upload(ctx, cosb, nl, &d_cosb));
    PZ_TRY(arena_upload(ctx, surf_reflect, (size_t)nwno, &d_rs));
    PZ_TRY(arena_upload(ctx, wno, (size_t)nwno, &d_wno));
    if (dwno) PZ_TRY(arena_upload(ctx, dwno, (size_t)nwno, &d_dwno));
    double *d_f = (double *)arena_take(ctx, sizeof(double) * nang * nwno);
    double *d_lvl[4] = {nullptr, nullptr, nullptr, nullptr};
    if (lvl)
        for (int j = 0; j < 4; ++j) d_lvl[j] = (double *)arena_take(ctx, sizeof(double) * nang * nv);
    if (!d_f || (lvl && !d_lvl[3])) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_get_thermal_1d_dev(ctx, nlevel, d_wno, nwno, nwno, numg, numt, tlevel, d_dtau, d_w0, d_cosb,
                                     plevel, ubar1, d_rs, hard_surface, d_dwno, calc_type, d_f, d_lvl[0],
                                     d_lvl[1], d_lvl[2], d_lvl[3], nullptr, nullptr, nullptr));
    PZ_HIP(ctx, hipMemcpyAsync(flux_at_top, d_f, sizeof(double) * nang * nwno, hipMemcpyDeviceToHost, ctx->stream));
    if (lvl) {
        double *h[4] = {flux_minus, flux_plus, flux_minus_mdpt, flux_plus_mdpt};
        for (int j = 0; j < 4; ++j)
            if (h[j]) PZ_HIP(ctx, hipMemcpyAsync(h[j], d_lvl[j], sizeof(double) * nang * nv, hipMemcpyDeviceToHost, ctx->stream));
    }
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int picaso_get_thermal_3d_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                              int numt, const double *tlevel_3d, const double *dtau_3d,
                              const double *w0_3d, const double *cosb_3d, const double *plevel_3d,
                              const double *ubar1, const double *surf_reflect, int hard_surface,
                              double *int_at_top, const double *gweight, const double *tweight,
                              double *flux_disk)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_3d: bad sizes");
    PZ_NEED(ctx, "get_thermal_3d", wno, tlevel_3d, plevel_3d, ubar1, surf_reflect, int_at_top);
    if (!dtau_3d || !w0_3d) return fail(ctx, "get_thermal_3d: dtau and w0 are required (cosb NULL = no cloud)");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nfac = numg * numt;
    std::vector<double> tab((size_t)nfac * (2 * (size_t)nlevel + 1));
    double *t_u1 = tab.data(), *t_T = t_u1 + nfac, *t_P = t_T + (size_t)nlevel * nfac;
    for (int i = 0; i < nfac; ++i) t_u1[i] = ubar1[i];
    for (size_t i = 0; i < (size_t)nlevel * nfac; ++i) { t_T[i] = tlevel_3d[i]; t_P[i] = plevel_3d[i]; }
    const void *d_tab = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), sizeof(double) * tab.size(), &d_tab));
    ThermalArgs a{};
    a.nlayer = nlevel - 1;
    a.ncol = (long)nwno * nfac;
    a.pitch = a.ncol;
    a.nfac = nfac;
    a.nwno = nwno;
    a.wno = wno; a.dwno = nullptr;
    a.u1_tab = (const double *)d_tab;
    a.tlevel = a.u1_tab + nfac;
    a.plevel = a.tlevel + (size_t)nlevel * nfac;
    a.dtau = dtau_3d; a.w0 = w0_3d; a.cosb = cosb_3d; a.surf_reflect = surf_reflect;
    a.hard_surface = hard_surface; a.calc_type = 0;
    a.na = 1;
    a.flux = int_at_top;
    a.disk = nullptr;
    PZ_TRY(launch_thermal_toa(ctx, a, true));
    if (flux_disk && gweight && tweight)
        PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, int_at_top, gweight, numg, tweight, numt, flux_disk));
    return 0;
}

int picaso_get_thermal_3d_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, const double *wno, int nwno, int numg,
                                    int numt, const double *tlevel_3d, const double *const *dtau_3d,
                                    const double *const *w0_3d, const double *const *cosb_3d,
                                    const double *plevel_3d, const double *ubar1,
                                    const double *const *surf_reflect, int hard_surface, double *const *int_at_top,
                                    const double *gweight, const double *tweight, double *const *flux_disk)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nspec < 1) return fail(ctx, "get_thermal_3d_batch: nspec must be >= 1, got %d", nspec);
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_3d_batch: bad sizes");
    if (!dtau_3d || !w0_3d || !surf_reflect || !int_at_top || !tlevel_3d || !plevel_3d || !ubar1)
        return fail(ctx, "get_thermal_3d_batch: null argument");
    const bool fuse = flux_disk && gweight && tweight;
    for (int s = 0; s < nspec; ++s)
        if (!dtau_3d[s] || !w0_3d[s] || (cosb_3d && !cosb_3d[s]) || !surf_reflect[s] || !int_at_top[s] ||
            (fuse && !flux_disk[s]))
            return fail(ctx, "get_thermal_3d_batch: null per-spectrum pointer (spectrum %d)", s);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const int nfac = numg * numt;
    const size_t per = (size_t)nfac * (2 * (size_t)nlevel + 1);           // u1, T, P of one spectrum
    const size_t head = align_up(sizeof(ThermalBatchItem) * (size_t)nspec, 256);
    std::vector<char> tab(head + sizeof(double) * per * nspec);
    if (tab.size() > picaso_ctx::SLOT_BYTES)
        return fail(ctx, "get_thermal_3d_batch: at most %zu spectra of this size per call",
                    (picaso_ctx::SLOT_BYTES - 4096) / (sizeof(double) * per + sizeof(ThermalBatchItem)));
    ThermalBatchItem *items = (ThermalBatchItem *)tab.data();
    double *lv = (double *)(tab.data() + head);
    const char *dbase = nullptr;
    PZ_TRY(table_next_dev(ctx, tab.size(), &dbase));
    for (int s = 0; s < nspec; ++s) {
        double *t_u1 = lv + per * s, *t_T = t_u1 + nfac, *t_P = t_T + (size_t)nlevel * nfac;
        for (int i = 0; i < nfac; ++i) t_u1[i] = ubar1[(size_t)s * nfac + i];
        memcpy(t_T, tlevel_3d + (size_t)s * nlevel * nfac, sizeof(double) * (size_t)nlevel * nfac);
        memcpy(t_P, plevel_3d + (size_t)s * nlevel * nfac, sizeof(double) * (size_t)nlevel * nfac);
        ThermalBatchItem &it = items[s];
        memset(&it, 0, sizeof(it));
        it.dtau = dtau_3d[s]; it.w0 = w0_3d[s]; it.cosb = cosb_3d ? cosb_3d[s] : nullptr;
        it.surf_reflect = surf_reflect[s];
        it.u1_tab = (const double *)(dbase + head) + per * s;
        it.tlevel = it.u1_tab + nfac;
        it.plevel = it.tlevel + (size_t)nlevel * nfac;
        it.flux = int_at_top[s];
        it.disk = nullptr;
    }
    const void *d = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), tab.size(), &d));
    if ((const char *)d != dbase) return fail(ctx, "get_thermal_3d_batch: table slot moved");
    ThermalArgs a{};
    a.nlayer = nlevel - 1;
    a.ncol = (long)nwno * nfac;
    a.pitch = a.ncol;
    a.nfac = nfac;
    a.nwno = nwno;
    a.wno = wno; a.dwno = nullptr;
    a.dtau = dtau_3d[0]; a.w0 = w0_3d[0]; a.cosb = cosb_3d ? cosb_3d[0] : nullptr;    // cosb: the presence pattern
    a.surf_reflect = surf_reflect[0];
    a.hard_surface = hard_surface; a.calc_type = 0;
    a.na = 1;
    a.disk = nullptr;
    a.batch = (const ThermalBatchItem *)d;
    a.nspec = nspec;
    PZ_TRY(launch_thermal_toa(ctx, a, true));
    if (fuse)
        for (int s = 0; s < nspec; ++s)
            PZ_TRY(picaso_compress_thermal_dev(ctx, (size_t)nwno, int_at_top[s], gweight, numg, tweight, numt, flux_disk[s]));
    return 0;
}

int picaso_get_thermal_3d(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                          int numt, const double *tlevel_3d, const double *dtau_3d,
                          const double *w0_3d, const double *cosb_3d, const double *plevel_3d,
                          const double *ubar1, const double *surf_reflect, int hard_surface,
                          double *int_at_top)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlevel < 2 || nwno < 1 || numg < 1 || numt < 1) return fail(ctx, "get_thermal_3d: bad sizes");
    PZ_NEED(ctx, "get_thermal_3d", wno, tlevel_3d, dtau_3d, w0_3d, cosb_3d, plevel_3d, ubar1, surf_reflect, int_at_top);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nfac = (size_t)numg * numt;
    const size_t nl = (size_t)(nlevel - 1) * nwno * nfac;
    const size_t need = sizeof(double) * (3 * nl + 2 * (size_t)nwno + nfac * nwno) + 64 * 256;
    PZ_TRY(arena_reset(ctx, need));
    const double *d_dtau, *d_w0, *d_cosb, *d_rs, *d_wno;
    PZ_TRY(arena_upload(ctx, dtau_3d, nl, &d_dtau));
    PZ_TRY(arena_upload(ctx, w0_3d, nl, &d_w0));
    PZ_TRY(arena_upload(ctx, cosb_3d, nl, &d_cosb));
    PZ_TRY(arena_upload(ctx, surf_reflect, (size_t)nwno, &d_rs));
    PZ_TRY(arena_upload(ctx, wno, (size_t)nwno, &d_wno));
    double *d_f = (double *)arena_take(ctx, sizeof(double) * nfac * nwno);
    if (!d_f) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_get_thermal_3d_dev(ctx, nlevel, d_wno, nwno, numg, numt, tlevel_3d, d_dtau, d_w0, d_cosb,
                                     plevel_3d, ubar1, d_rs, hard_surface, d_f, nullptr, nullptr, nullptr));
    PZ_HIP(ctx, hipMemcpyAsync(int_at_top, d_f, sizeof(double) * nfac * nwno, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

/* ============================================================================================
 * disk quadrature
 * ============================================================================================ */
// (gweight[g], tweight[t]) pairs in (g,t) loop order; small tables go with the kernel arguments
static int compress_with_weights(picaso_ctx *ctx, size_t ninner, const double *x, const double *gweight, int ng,
                                 const double *tweight, int nt, const double *F0PI, int mode, double c1,
                                 double c2, double *out)
{
    if (ng < 1 || nt < 1) return fail(ctx, "compress: empty weight table");
    std::vector<double> wts(2 * (size_t)ng * nt);
    for (int g = 0; g < ng; ++g)
        for (int t = 0; t < nt; ++t) {
            wts[2 * ((size_t)g * nt + t)] = gweight[g];
            wts[2 * ((size_t)g * nt + t) + 1] = tweight[t];
        }
    if (ng * nt <= 128) return launch_compress_hostw(ctx, ninner, x, wts.data(), ng * nt, F0PI, mode, c1, c2, out);
    const void *d = nullptr;
    PZ_TRY(table_upload(ctx, wts.data(), sizeof(double) * wts.size(), &d));
    return launch_compress_dev(ctx, ninner, x, (const double *)d, ng * nt, F0PI, mode, c1, c2, out);
}

int picaso_compress_disco_dev(picaso_ctx *ctx, int nwno, double cos_theta,
                              const double *xint_at_top, const double *gweight, int ng,
                              const double *tweight, int nt, const double *F0PI, double *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const double sym = (nt == 1) ? 2.0 * 3.14159265358979323846 : 1.0;      // disco.py:140-141
    return compress_with_weights(ctx, (size_t)nwno, xint_at_top, gweight, ng, tweight, nt, F0PI, COMPRESS_DISCO,
                                 sym * 0.5, cos_theta + 1.0, albedo);
}

int picaso_compress_disco(picaso_ctx *ctx, int nwno, double cos_theta, const double *xint_at_top,
                          const double *gweight, int ng, const double *tweight, int nt,
                          const double *F0PI, double *albedo)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nwno < 1 || ng < 1 || nt < 1) return fail(ctx, "compress_disco: bad sizes nwno=%d ng=%d nt=%d", nwno, ng, nt);
    PZ_NEED(ctx, "compress_disco", xint_at_top, gweight, tweight, F0PI, albedo);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nx = (size_t)ng * nt * nwno;
    PZ_TRY(arena_reset(ctx, sizeof(double) * (nx + 2 * (size_t)nwno) + 16 * 256));
    const double *d_x, *d_f0;
    PZ_TRY(arena_upload(ctx, xint_at_top, nx, &d_x));
    PZ_TRY(arena_upload(ctx, F0PI, (size_t)nwno, &d_f0));
    double *d_o = (double *)arena_take(ctx, sizeof(double) * nwno);
    if (!d_o) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_compress_disco_dev(ctx, nwno, cos_theta, d_x, gweight, ng, tweight, nt, d_f0, d_o));
    PZ_HIP(ctx, hipMemcpyAsync(albedo, d_o, sizeof(double) * nwno, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int picaso_compress_thermal_dev(picaso_ctx *ctx, size_t ninner, const double *flux_at_top,
                                const double *gweight, int ng, const double *tweight, int nt,
                                double *flux)
{
    if (!ctx) return fail(nullptr, "null context");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const double sym = (nt == 1) ? 1.0 : 1.0 / (2.0 * 3.14159265358979323846);   // disco.py:174-175
    return compress_with_weights(ctx, ninner, flux_at_top, gweight, ng, tweight, nt, nullptr, COMPRESS_THERMAL, sym,
                                 0.0, flux);
}

int picaso_compress_thermal(picaso_ctx *ctx, size_t ninner, const double *flux_at_top,
                            const double *gweight, int ng, const double *tweight, int nt,
                            double *flux)
{
    if (!ctx) return fail(nullptr, "null context");
    if (ninner < 1 || ng < 1 || nt < 1) return fail(ctx, "compress_thermal: bad sizes");
    PZ_NEED(ctx, "compress_thermal", flux_at_top, gweight, tweight, flux);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nx = (size_t)ng * nt * ninner;
    PZ_TRY(arena_reset(ctx, sizeof(double) * (nx + ninner) + 16 * 256));
    const double *d_x;
    PZ_TRY(arena_upload(ctx, flux_at_top, nx, &d_x));
    double *d_o = (double *)arena_take(ctx, sizeof(double) * ninner);
    if (!d_o) return fail(ctx, "arena exhausted");
    PZ_TRY(picaso_compress_thermal_dev(ctx, ninner, d_x, gweight, ng, tweight, nt, d_o));
    PZ_HIP(ctx, hipMemcpyAsync(flux, d_o, sizeof(double) * ninner, hipMemcpyDeviceToHost, ctx->stream));
    PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
