// On-the-fly correlated-k gas mixing ("resort-rebin", Amundsen et al. 2017) -- gfx950.
// Replaces deq_chem.mix_all_gases_gasesfly / do_mixing_mono_gasesfly / mix_2_gases
// (reference picaso/deq_chem.py:333-384, :387-477, :537-597), the inner loops of
// RetrieveCKs.mix_my_opacities_gasesfly (picaso/optics.py:1164-1198).
//
// One wavefront per (layer, P-T neighbour, wavenumber bin): with Nk <= 8 Gauss points the Nk^2 <= 64
// random-overlap products of two gases are exactly one lane each (lane = i*Nk + j).  Per gas added:
//   key   = (mix1*k1[i] + mix2*k2[j]) / (mix1+mix2)                      (:571, unfused)
//   stable rank sort across the wave (LDS broadcast of the keys, ties by lane = np.argsort 'mergesort'),
//   keys and weights moved to their sorted lanes with ds_permute,
//   prefix sum of the sorted weights (wave scan), x = cum / cum[last],
//   np.interp(gauss_pts, x, log k) -> k (base 2 here, base 10 in the reference: the base cancels in a
//   linear interpolation) -> the Nk coefficients of the mixture (:586-590),
// everything in registers apart from the 512-byte key row each wave broadcasts from LDS.  HBM traffic
// is the table rows read (ngas * Nk doubles per bin) and Nk doubles written: the kernel is
// VALU/cross-lane bound.
#include "common.hpp"
#include "device_math.hpp"

namespace pz {

constexpr int MIX_MAX_GAUSS = 8;

struct CKMixArgs {
    int ngas, nk, nwno, nlayer, ntemp;
    const double *const *tabs;      // device: ngas pointers to ln(kappa) (npres, ntemp, nwno, nk)
    const double *mixes;            // device (ngas, nlayer)
    const int *indices;             // device (4, nlayer): p_low, p_hi, t_low, t_hi
    double gpts[MIX_MAX_GAUSS], gwts[MIX_MAX_GAUSS];
    double *out;                    // (nlayer, 4, nwno, nk) ln of the mixed coefficients
};

__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// value of `v` moves from this lane to lane `dst`
__device__ __forceinline__ double permute_to_d(double v, int dst)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_permute(dst << 2, (int)b);
    const int hi = __builtin_amdgcn_ds_permute(dst << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// log2(x) for x > 0, ~1 ulp of the result, ~37 instructions (ocml's log10 takes 118): x = m 2^e with
// m in [sqrt(1/2), sqrt(2)), log2 m = (2/ln 2) atanh(s), s = (m - 1)/(m + 1), |s| <= 0.172, odd series
// to s^21 (next term < 3e-18 relative).  The rebinning interpolates linearly in log k, so the base of
// the logarithm cancels: interpolating log2 k and returning 2^r is the reference's
// 10**np.interp(., ., log10 k) up to the rounding of the logarithms themselves.
__device__ __forceinline__ double flog2(double x)
{
    int e;
    double m = frexp(x, &e);                                   // m in [1/2, 1)
    const bool low = m < 0x1.6a09e667f3bcdp-1;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = (m - 1.0) * frcp(m + 1.0);
    const double t = s * s;
    double q = fma(t, 1.0 / 21.0, 1.0 / 19.0);
    q = fma(q, t, 1.0 / 17.0);
    q = fma(q, t, 1.0 / 15.0);
    q = fma(q, t, 1.0 / 13.0);
    q = fma(q, t, 1.0 / 11.0);
    q = fma(q, t, 1.0 / 9.0);
    q = fma(q, t, 1.0 / 7.0);
    q = fma(q, t, 1.0 / 5.0);
    q = fma(q, t, 1.0 / 3.0);
    constexpr double C_HI = 0x1.71547652b82fep+1, C_LO = 0x1.777d0ffda0d24p-55;   // 2/ln 2
    const double cs = C_HI * s;
    const double lo = fma(C_HI, s, -cs) + C_LO * s;
    const double l = cs + fma(cs * t, q, lo);
    const double r = (double)e + l;
    return x == 0.0 ? -__builtin_inf() : (x == __builtin_inf() ? x : r);
}

__device__ __forceinline__ double shfl_d(double v, int src_lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)b);
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__global__ __launch_bounds__(256) void k_ckmix(const CKMixArgs a)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) double s_key[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long iw = blockIdx.x * 4L + wv;
    if (iw >= a.nwno) return;                          // whole wave; no block-level barrier below
    const int il = blockIdx.y >> 2, ct = blockIdx.y & 3;
    const int nk = a.nk, n2 = nk * nk;
    const bool valid = lane < n2;
    const int i = valid ? lane / nk : 0, j = valid ? lane - i * nk : 0;
    // neighbour ct = 2*ip + it of the reference's nested loops (deq_chem.py:371-372)
    const int p_ind = a.indices[(ct >> 1) * a.nlayer + il];
    const int t_ind = a.indices[(2 + (ct & 1)) * a.nlayer + il];
    const size_t off = (((size_t)p_ind * a.ntemp + t_ind) * a.nwno + iw) * nk;
    double wi = 0.0, wj = 0.0, gp_own = 0.0;
    for (int n = 0; n < nk; ++n) {                     // by-value tables: uniform index only
        wi = (i == n) ? a.gwts[n] : wi;
        wj = (j == n) ? a.gwts[n] : wj;
        gp_own = (i == n) ? a.gpts[n] : gp_own;
    }
    const double w_own = valid ? wi * wj : 0.0;        // eq. 10 Amundsen 2017 (:572)
    double *row = s_key[wv];
    const int n2e = (n2 + 1) & ~1;

    double k1 = fexp(a.tabs[0][off + i]);              // (:428) coefficient i of the running mixture
    double mix_t = a.mixes[il];
    for (int g = 1; g < a.ngas; ++g) {
        const double k2 = fexp(a.tabs[g][off + j]);
        const double mix2 = a.mixes[(size_t)g * a.nlayer + il];
        const double mt = mix_t + mix2;
        const double key = valid ? (mix_t * k1 + mix2 * k2) / mt : __builtin_inf();
        // stable rank = number of elements sorting before this one (ties by original index, as
        // np.argsort(kind='mergesort')); the keys are broadcast from LDS two per ds_read_b128
        row[lane] = key;
        __builtin_amdgcn_wave_barrier();
        int rank = 0;
        for (int t = 0; t < n2e; t += 2) {
            const double2 kk = *reinterpret_cast<const double2 *>(row + t);
            rank += ((kk.x < key) || (kk.x == key && t < lane)) ? 1 : 0;
            rank += ((kk.y < key) || (kk.y == key && t + 1 < lane)) ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
        const bool bad = __any(valid && (key != key));  // NaN in -> NaN out
        const int dst = valid ? rank : lane;
        const double ks = permute_to_d(key, dst);
        const double ws = permute_to_d(w_own, dst);
        // cumulative weight (np.cumsum; a log-step wave scan: sums of <= 64 positive weights, the
        // summation order differs from numpy's in the last bit only)
        double cum = ws;
        for (int o = 1; o < 64; o <<= 1) {
            const double up = shfl_d(cum, lane - o);
            cum = (lane >= o) ? cum + up : cum;
        }
        const double x = cum / readlane_d(cum, n2 - 1);       // np.max of an increasing sum (:582)
        const double f = flog2(ks);
        // np.interp(gauss_pts, x, f) (:586): slope of the segment [lane, lane+1] once per lane
        const int nb = lane + 1 < n2 ? lane + 1 : lane;
        const double x1 = shfl_d(x, nb), f1 = shfl_d(f, nb);
        const double slope = (f1 - f) / (x1 - x);
        // every lane looks up the segment of its own Gauss point gauss_pts[i] (the lanes of one i
        // agree): binary search for the last x <= gp over the sorted lanes, 6 ds_bpermute rounds
        const double x_first = readlane_d(x, 0), x_last = readlane_d(x, n2 - 1);
        const double f_first = readlane_d(f, 0), f_last = readlane_d(f, n2 - 1);
        int pos = 0;
#pragma unroll
        for (int st = 32; st >= 1; st >>= 1) {
            const int cand = pos + st;
            const double xc = shfl_d(x, cand & 63);
            pos = (cand < n2 && xc <= gp_own) ? cand : pos;
        }
        const int pos1 = pos + 1 < n2 ? pos + 1 : pos;
        const double xj = shfl_d(x, pos), fj = shfl_d(f, pos), sj = shfl_d(slope, pos);
        const double xj1 = shfl_d(x, pos1), fj1 = shfl_d(f, pos1);
        double rsel = sj * (gp_own - xj) + fj;
        if (rsel != rsel) {                            // numpy's fallbacks for a non-finite slope
            rsel = sj * (gp_own - xj1) + fj1;
            if (rsel != rsel && fj == fj1) rsel = fj;
        }
        rsel = (xj == gp_own) ? fj : rsel;
        rsel = (pos == n2 - 1) ? f_last : rsel;
        rsel = (gp_own < x_first) ? f_first : rsel;
        rsel = (gp_own > x_last) ? f_last : rsel;
        k1 = bad ? __builtin_nan("") : fexp2(rsel);
        mix_t = mt;
    }
    if (valid && j == 0)
        a.out[(((size_t)il * 4 + ct) * a.nwno + iw) * nk + i] = log(k1);
}

}  // namespace pz

using namespace pz;

extern "C" {

int picaso_mix_all_gases_gasesfly_dev(picaso_ctx *ctx, int ngas, const double *const *kappas, int npres,
                                      int ntemp, int nwno, int ngauss, const double *mixes,
                                      const double *gauss_pts, const double *gauss_wts, const int *indices,
                                      int nlayer, double *kappa_mixed)
{
    if (!ctx) return fail(nullptr, "null context");
    if (ngas < 1 || ngas > 256) return fail(ctx, "mix_all_gases_gasesfly: ngas must be 1..256, got %d", ngas);
    if (ngauss < 1 || ngauss > MIX_MAX_GAUSS)
        return fail(ctx, "mix_all_gases_gasesfly: ngauss must be 1..%d (one wavefront holds the ngauss^2 "
                         "overlap products), got %d", MIX_MAX_GAUSS, ngauss);
    if (npres < 1 || ntemp < 1 || nwno < 1 || nlayer < 1)
        return fail(ctx, "mix_all_gases_gasesfly: bad sizes npres=%d ntemp=%d nwno=%d nlayer=%d", npres, ntemp,
                    nwno, nlayer);
    if (!kappas || !mixes || !gauss_pts || !gauss_wts || !indices || !kappa_mixed)
        return fail(ctx, "mix_all_gases_gasesfly: null argument");
    for (int l = 0; l < nlayer; ++l)
        for (int q = 0; q < 4; ++q) {
            const int v = indices[q * nlayer + l], lim = q < 2 ? npres : ntemp;
            if (v < 0 || v >= lim)
                return fail(ctx, "mix_all_gases_gasesfly: index %d of layer %d outside the %s grid (%d)", v, l,
                            q < 2 ? "pressure" : "temperature", lim);
        }
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    // one ring slot: [gas table pointers | mixes | indices]
    const size_t b_ptr = sizeof(double *) * (size_t)ngas, b_mix = sizeof(double) * (size_t)ngas * nlayer,
                 b_idx = sizeof(int) * 4 * (size_t)nlayer;
    std::vector<char> host(b_ptr + b_mix + b_idx);
    memcpy(host.data(), kappas, b_ptr);
    memcpy(host.data() + b_ptr, mixes, b_mix);
    memcpy(host.data() + b_ptr + b_mix, indices, b_idx);
    const void *d = nullptr;
    PZ_TRY(table_upload(ctx, host.data(), host.size(), &d));
    CKMixArgs a{};
    a.ngas = ngas; a.nk = ngauss; a.nwno = nwno; a.nlayer = nlayer; a.ntemp = ntemp;
    a.tabs = (const double *const *)d;
    a.mixes = (const double *)((const char *)d + b_ptr);
    a.indices = (const int *)((const char *)d + b_ptr + b_mix);
    for (int n = 0; n < ngauss; ++n) { a.gpts[n] = gauss_pts[n]; a.gwts[n] = gauss_wts[n]; }
    a.out = kappa_mixed;
    const dim3 grid((unsigned)((nwno + 3) / 4), (unsigned)nlayer * 4u);
    if ((long)nlayer * 4 > 65535) return fail(ctx, "mix_all_gases_gasesfly: nlayer %d exceeds the launch grid", nlayer);
    hipLaunchKernelGGL(k_ckmix, grid, dim3(256), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
