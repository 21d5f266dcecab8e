// Spectrum-wide integrals of the output dictionary on the device: the Bond-albedo numerator
// np.trapz(x=1/wno, y=albedo*stellar) and the effective-temperature integral np.trapz(x=1/wno[::-1], y=thermal[::-1])
// (reference justdoit.py:552-599).  At 1e5 wavelengths the three numpy passes and the sum of one such integral cost
// the host 0.07 ms -- a fifth of what a retrieval's host thread spends per spectrum -- for 0.8 MB that is already
// resident.  The result must not depend on where it was computed (a spectrum in wavelength blocks on several GPUs
// integrates the gathered arrays with numpy), so the sum is taken in numpy's own order:
//
//   numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum:
//     n < 8            sequential, starting from -0.0
//     n <= 128         r[j] = a[j] (j < 8); r[j] += a[i + j] for i = 8, 16, ... < n - n % 8;
//                      ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the n % 8 last elements one by one
//     n > 128          n2 = n / 2 - (n / 2) % 8;  sum(a, n2) + sum(a + n2, n - n2)
//   and the reduction hands that function 8 192 elements at a time (the ufunc buffer size), adding each chunk's sum
//   to the running result in order -- tools/numpy_sum_order.py shows both against np.sum.
//
// and every element as numpy forms it: (d[j] * (y[j + 1] + y[j])) / 2.0 with y = albedo * stellar rounded first.
// tests/test_integrals_gpu.py holds this against np.trapezoid for lengths 2 ... 1e5 + 7 (np.array_equal).
#include "common.hpp"

#include <algorithm>

namespace pz {

struct PairwisePlan {
    long m = 0;                 // number of summands
    int nleaf = 0, nlevel = 0, root = 0, nvals = 0;
    int *d_leaf = nullptr;      // nleaf + 1 offsets
    int *d_ops = nullptr;       // [level offsets (nlevel + 1)] then (dst, a, b) triples, level by level from the leaves up
    double *d_vals = nullptr;   // nleaf leaf sums, then one value per inner node
};

namespace {

struct Node { int id, level; };
constexpr long REDUCE_CHUNK = 8192;        // numpy's default ufunc buffer size, in elements

// the recursion of pairwise_sum over [start, start + n): leaves left to right, inner nodes after their children
Node build(long start, long n, std::vector<int> &leaf, std::vector<std::vector<int>> &levels, int &ninner)
{
    if (n <= 128) {
        leaf.push_back((int)start);
        return {(int)leaf.size() - 1, 0};
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    const Node l = build(start, n2, leaf, levels, ninner);
    const Node r = build(start + n2, n - n2, leaf, levels, ninner);
    const int level = std::max(l.level, r.level) + 1;
    if ((int)levels.size() < level) levels.resize(level);
    const int id = -(++ninner);                    // numbered after the leaves are known
    levels[level - 1].insert(levels[level - 1].end(), {id, l.id, r.id});
    return {id, level};
}

}  // namespace

struct TrapzArgs {
    long n;                      // points; n - 1 summands
    const double *d, *y, *mult;
    int reverse;
    const int *leaf;
    int nleaf;
    double *vals;
};

__device__ __forceinline__ double trapz_elem(const TrapzArgs &a, long j)
{
#pragma clang fp contract(off)
    const long i0 = a.reverse ? a.n - 1 - j : j, i1 = a.reverse ? a.n - 2 - j : j + 1;
    double y0 = a.y[i0], y1 = a.y[i1];
    if (a.mult) {
        y0 = y0 * a.mult[i0];
        y1 = y1 * a.mult[i1];
    }
    return (a.d[j] * (y1 + y0)) / 2.0;
}

// eight lanes per leaf: lane j carries numpy's partial sum r[j]
__global__ __launch_bounds__(256) void k_trapz_leaves(const TrapzArgs a)
{
#pragma clang fp contract(off)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = t >> 3, j = t & 7;
    const bool live = l < a.nleaf;
    const long start = live ? a.leaf[l] : 0, len = live ? a.leaf[l + 1] - start : 0;
    double res;
    if (len < 8) {                                   // only when the whole sum has fewer than 8 terms
        res = -0.0;
        if (j == 0)
            for (long i = 0; i < len; ++i) res = res + trapz_elem(a, start + i);
    } else {
        const long body = len - len % 8;
        // a leaf has at most 128 terms = 16 per lane: all loads first (they do not depend on each other; the sum does)
        double e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = trapz_elem(a, start + ((i * 8 < body) ? i * 8 : 0) + j);   // no branch around a load
        double r = e[0];
#pragma unroll
        for (int i = 1; i < 16; ++i)
            if (i * 8 < body) r = r + e[i];
        r = r + __shfl_xor(r, 1);                   // r0 + r1 | r2 + r3 | r4 + r5 | r6 + r7
        r = r + __shfl_xor(r, 2);                   // (r0 + r1) + (r2 + r3) | (r4 + r5) + (r6 + r7)
        r = r + __shfl_xor(r, 4);
        res = r;
        if (j == 0)
            for (long i = body; i < len; ++i) res = res + trapz_elem(a, start + i);
    }
    if (live && j == 0) a.vals[l] = res;
}

// the inner nodes, one level per barrier; a single workgroup (a 1e5-term sum: 13 chunks, 1 023 nodes, 19 levels)
constexpr int COMBINE_LDS = 4096;          // values (leaves + inner nodes) combined in LDS: sums of up to ~2e5 terms
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void k_pairwise_combine(const int *__restrict__ ops, int nlevel, int root, int nvals,
                                                           int nleaf, double *gvals, double *__restrict__ out)
{
    __shared__ double lvals[IN_LDS ? COMBINE_LDS : 1];
    double *vals = IN_LDS ? lvals : gvals;
    if (IN_LDS) {
        for (int i = threadIdx.x; i < nleaf; i += blockDim.x) lvals[i] = gvals[i];
        __syncthreads();
    }
    const int *trip = ops + nlevel + 1;
    for (int lev = 0; lev < nlevel; ++lev) {
        for (int o = ops[lev] + (int)threadIdx.x; o < ops[lev + 1]; o += (int)blockDim.x)
            vals[trip[3 * o]] = vals[trip[3 * o + 1]] + vals[trip[3 * o + 2]];
        if (!IN_LDS) __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = vals[root];
}

void free_pairwise_plans(picaso_ctx *ctx)
{
    for (auto &kv : ctx->pairwise_plans) {
        PairwisePlan *p = kv.second;
        if (p->d_leaf) (void)hipFree(p->d_leaf);
        if (p->d_ops) (void)hipFree(p->d_ops);
        if (p->d_vals) (void)hipFree(p->d_vals);
        delete p;
    }
    ctx->pairwise_plans.clear();
}

static int get_plan(picaso_ctx *ctx, long m, PairwisePlan **out)
{
    auto it = ctx->pairwise_plans.find(m);
    if (it != ctx->pairwise_plans.end()) {
        *out = it->second;
        return 0;
    }
    if (ctx->pairwise_plans.size() >= 16) {          // a caller cycling through many grid sizes
        PZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        free_pairwise_plans(ctx);
    }
    std::vector<int> leaf;
    std::vector<std::vector<int>> levels;
    int ninner = 0;
    Node top = build(0, std::min<long>(m, REDUCE_CHUNK), leaf, levels, ninner);
    for (long start = REDUCE_CHUNK; start < m; start += REDUCE_CHUNK) {          // acc += pairwise_sum(next chunk)
        const Node c = build(start, std::min<long>(REDUCE_CHUNK, m - start), leaf, levels, ninner);
        const int level = std::max(top.level, c.level) + 1;
        if ((int)levels.size() < level) levels.resize(level);
        const int id = -(++ninner);
        levels[level - 1].insert(levels[level - 1].end(), {id, top.id, c.id});
        top = {id, level};
    }
    const int nleaf = (int)leaf.size();
    leaf.push_back((int)m);
    auto slot = [&](int id) { return id >= 0 ? id : nleaf + (-id - 1); };
    std::vector<int> ops(levels.size() + 1, 0);
    for (size_t lev = 0; lev < levels.size(); ++lev) ops[lev + 1] = ops[lev] + (int)levels[lev].size() / 3;
    for (auto &lv : levels)
        for (int id : lv) ops.push_back(slot(id));
    PairwisePlan *p = new PairwisePlan;
    p->m = m; p->nleaf = nleaf; p->nlevel = (int)levels.size(); p->root = slot(top.id);
    p->nvals = nleaf + ninner;
    auto bail = [&](hipError_t e, const char *what) {
        if (p->d_leaf) (void)hipFree(p->d_leaf);
        if (p->d_ops) (void)hipFree(p->d_ops);
        if (p->d_vals) (void)hipFree(p->d_vals);
        delete p;
        return fail(ctx, "picaso_trapz_dev: %s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&p->d_leaf, leaf.size() * sizeof(int))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void **)&p->d_ops, ops.size() * sizeof(int))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void **)&p->d_vals, (size_t)(nleaf + ninner) * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc");
    // once per grid size: a blocking copy keeps the host vectors' lifetime out of the picture
    if ((e = hipMemcpy(p->d_leaf, leaf.data(), leaf.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    if ((e = hipMemcpy(p->d_ops, ops.data(), ops.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    ctx->pairwise_plans[m] = p;
    *out = p;
    return 0;
}

}  // namespace pz

using namespace pz;

extern "C" int picaso_trapz_dev(picaso_ctx *ctx, long n, const double *d, const double *y, const double *mult, int reverse,
                                double *out)
{
    if (!ctx || !d || !y || !out) return fail(ctx, "picaso_trapz_dev: null argument");
    if (n < 2) return fail(ctx, "picaso_trapz_dev: needs at least two points, got %ld", n);
    if (n - 1 > 0x7fffffffL) return fail(ctx, "picaso_trapz_dev: %ld points are beyond the 32-bit leaf offsets", n);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    PairwisePlan *p = nullptr;
    PZ_TRY(get_plan(ctx, n - 1, &p));
    TrapzArgs a{n, d, y, mult, reverse, p->d_leaf, p->nleaf, p->d_vals};
    const long threads = (long)p->nleaf * 8;
    hipLaunchKernelGGL(k_trapz_leaves, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, a);
    if (p->nvals <= COMBINE_LDS)
        hipLaunchKernelGGL(k_pairwise_combine<true>, dim3(1), dim3(1024), 0, ctx->stream, p->d_ops, p->nlevel, p->root,
                           p->nvals, p->nleaf, p->d_vals, out);
    else
        hipLaunchKernelGGL(k_pairwise_combine<false>, dim3(1), dim3(1024), 0, ctx->stream, p->d_ops, p->nlevel, p->root,
                           p->nvals, p->nleaf, p->d_vals, out);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}
