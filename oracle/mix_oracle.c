/* CPU oracle -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C restatement of the reference's on-the-fly correlated-k gas mixing ("resort-rebin",
 * Amundsen et al. 2017 / Molliere et al. 2015 B.2.1):
 *   mix_2_gases              picaso/deq_chem.py:537-597
 *   do_mixing_mono_gasesfly  picaso/deq_chem.py:387-477
 *   mix_all_gases_gasesfly   picaso/deq_chem.py:333-384
 * Operation order follows the reference: unfused (mix1*k1[i] + mix2*k2[j]) / mix_t, a STABLE sort of
 * the Nk^2 mixed coefficients (np.argsort(kind='mergesort')), sequential cumulative sum of the sorted
 * weights, numpy's np.interp arithmetic (slope*(x - xp[j]) + fp[j], exact-hit and edge rules of
 * numpy/core/src/multiarray/compiled_base.c), log10 / pow(10,.) / exp / log from libm.
 * Pinned against tests/golden/mixing.npz (outputs of the reference's own functions).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* stable insertion sort of n (key, idx) pairs by key: same permutation as a stable merge sort */
static void stable_sort(int n, double *key, int *idx)
{
    for (int a = 1; a < n; ++a) {
        const double k = key[a];
        const int id = idx[a];
        int b = a - 1;
        while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; idx[b + 1] = idx[b]; --b; }
        key[b + 1] = k;
        idx[b + 1] = id;
    }
}

/* np.interp(x, xp, fp) for one x; xp increasing, n >= 1 */
static double np_interp1(double x, int n, const double *xp, const double *fp)
{
    if (isnan(x)) return x;
    if (x > xp[n - 1]) return fp[n - 1];
    if (x < xp[0]) return fp[0];
    int j = 0;                                   /* xp[j] <= x < xp[j+1] */
    { int lo = 0, hi = n; while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (x >= xp[mid]) lo = mid; else hi = mid; } j = lo; }
    if (j == n - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double r = slope * (x - xp[j]) + fp[j];
    if (isnan(r)) {
        r = slope * (x - xp[j + 1]) + fp[j + 1];
        if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}

/* deq_chem.py:537-597; k1, k2, out have nk entries (out may alias k1) */
static double mix_2_gases(int nk, const double *k1, const double *k2, double mix1, double mix2,
                          const double *gauss_pts, const double *gauss_wts, double *out, double *work, int *iwork)
{
    const int n2 = nk * nk;
    double *kmix = work, *wts = work + n2, *x = work + 2 * n2, *lk = work + 3 * n2;
    const double mix_t = mix1 + mix2;
    for (int i = 0; i < nk; ++i)
        for (int j = 0; j < nk; ++j) {
            kmix[i * nk + j] = (mix1 * k1[i] + mix2 * k2[j]) / mix_t;
            iwork[i * nk + j] = i * nk + j;
        }
    stable_sort(n2, kmix, iwork);
    double c = 0.0;
    for (int a = 0; a < n2; ++a) {
        const int id = iwork[a];
        wts[a] = gauss_wts[id / nk] * gauss_wts[id % nk];
        c = (a == 0) ? wts[a] : c + wts[a];
        x[a] = c;
    }
    double cmax = x[0];
    for (int a = 1; a < n2; ++a) cmax = x[a] > cmax ? x[a] : cmax;
    for (int a = 0; a < n2; ++a) { x[a] = x[a] / cmax; lk[a] = log10(kmix[a]); }
    for (int i = 0; i < nk; ++i) out[i] = pow(10.0, np_interp1(gauss_pts[i], n2, x, lk));
    return mix_t;
}

/* deq_chem.py:333-384.  kappas[g]: (npres, ntemp, nwno, nk) ln(kappa) of gas g; mixes (ngas, nlayer);
 * indices (4, nlayer) = p_low, p_hi, t_low, t_hi; out (nlayer, nwno, nk, 4) = ln of the mixed k. */
int orc_mix_all_gases_gasesfly(int ngas, const double *const *kappas, int npres, int ntemp, int nwno, int nk,
                               const double *mixes, const double *gauss_pts, const double *gauss_wts,
                               const int *indices, int nlayer, double *out)
{
    (void)npres;
    if (ngas < 1 || nk < 1) return 1;
    const int n2 = nk * nk;
    double *work = (double *)malloc(sizeof(double) * (4 * (size_t)n2 + 2 * (size_t)nk));
    int *iwork = (int *)malloc(sizeof(int) * (size_t)n2);
    if (!work || !iwork) { free(work); free(iwork); return 2; }
    double *kbin = work + 4 * n2, *knext = kbin + nk;
    for (int il = 0; il < nlayer; ++il) {
        int ct = 0;
        for (int ip = 0; ip < 2; ++ip)
            for (int it = 0; it < 2; ++it, ++ct) {
                const int p_ind = indices[ip * nlayer + il], t_ind = indices[(2 + it) * nlayer + il];
                for (int iw = 0; iw < nwno; ++iw) {
                    const size_t off = (((size_t)p_ind * ntemp + t_ind) * nwno + iw) * nk;
                    for (int i = 0; i < nk; ++i) kbin[i] = exp(kappas[0][off + i]);
                    double mix_t = mixes[il];
                    for (int g = 1; g < ngas; ++g) {
                        for (int i = 0; i < nk; ++i) knext[i] = exp(kappas[g][off + i]);
                        mix_t = mix_2_gases(nk, kbin, knext, mix_t, mixes[(size_t)g * nlayer + il], gauss_pts,
                                            gauss_wts, kbin, work, iwork);
                    }
                    for (int i = 0; i < nk; ++i)
                        out[(((size_t)il * nwno + iw) * nk + i) * 4 + ct] = log(kbin[i]);
                }
            }
    }
    free(work);
    free(iwork);
    return 0;
}
