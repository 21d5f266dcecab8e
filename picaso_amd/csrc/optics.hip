// Opacity pre-stage -- gfx950.  HBM-bound plane algebra: one lane per wavelength, every access a
// coalesced row segment of the reference's (nlayer, nwno) layout.
//
//  k_opacity_gas     : RetrieveOpacities.get_opacities[_nearest] interpolation arithmetic
//                      (reference picaso/optics.py:2277-2294, :2350-2351, :2304-2306) fused with
//                      the TAUGAS / TAURAY sums of compute_opacity (optics.py:144-277)
//  k_compute_opacity : mixing + delta-Eddington half of compute_opacity (optics.py:327-431)
#include "common.hpp"
#include "device_math.hpp"

namespace pz {

struct MixArgs {
    int nlayer, nwno, test_mode, delta_eddington, stream;
    int ncolper;      // columns per wavelength of taugas and of every output (tauray, cloud, raman have none)
    int nfac;         // > 1: 3-D facet mode.  Columns are (wavelength, facet) with the facet fastest -- the
                      // layout of the (rows, nwno, ng, nt) planes get_reflected_3d takes and of the cloud
                      // inputs; taugas / tauray / raman are facet-major (nfac, nlayer, nwno), as the per-facet
                      // gas launches write them (ncolper must be 1)
    const double *taugas, *tauray, *taucld, *w0c, *g0c, *raman;
    // fused launch only: cloud tables on their own grid, interpolated where they are used (regrid_value: numpy.interp's
    // bits) instead of three regridded planes -- cld_fp = (3 nlayer, cld_nin): the opd rows, the w0 rows, the g0 rows
    int cld_nin;
    const double *cld_xp, *cld_fp, *cld_x;
    double raman_const;
    int raman_row;    // raman is one row (nwno) for every layer and facet (the Pollack table) instead of a plane
    double *dtau, *tau, *w0, *cosb, *ftau_cld, *ftau_ray, *gcos2, *dtau_og, *tau_og, *w0_og,
        *cosb_og, *w0_no_raman, *f_deltaM;
    // k_compute_opacity only: grid.y facets, FACET-MAJOR in and out (picaso_compute_opacity_facet_major_ck_dev).  Facet
    // f = blockIdx.y reads and writes f * fm_lay elements further on in every (nlayer, ncol) plane, f * fm_lev in the two
    // level planes, f * fm_ray in tauray (and in a Raman plane), f * fm_cld in the cloud planes; fm_lay = 0: one facet
    long fm_lay, fm_lev, fm_ray, fm_cld;
};

__device__ __forceinline__ double ipow(double x, int n)
{
    double r = 1.0;
    for (int i = 0; i < n; ++i) r *= x;      // COSB**stream, stream = 2 or 4 (optics.py:412)
    return r;
}

// One layer of one column: mixing + delta-Eddington (optics.py:327-431), running level sums in tau_run /
// taud_run.  Shared by the plane kernel and the facet kernel; operations as written (no contraction), so
// both give the same bits -- and planes that coincide analytically (w0 and w0_no_raman for a constant Raman
// factor of 0.99999, the delta-scaled and the unscaled set for cosb = 0) coincide bit for bit, which the
// 3-D path relies on when it leaves the duplicates out.
__device__ __forceinline__ void mix_layer(const MixArgs &a, long q, long qn, double tg, double tr, double tc,
                                          double wc, double gc, double rf, double &tau_run, double &taud_run)
{
#pragma clang fp contract(off)
    // every output is optional (kernel arguments: the tests are scalar); a quotient nobody stores is not formed --
    // a cloud-free spectrum asks for dtau, w0 (and tau): two of the six divisions
    const bool want_fc = a.ftau_cld != nullptr, want_fr = a.ftau_ray != nullptr || a.gcos2 != nullptr;
    const bool want_nr = a.w0_no_raman != nullptr;
    double dtau = tg + tr + tc;                                     // optics.py:329
    double fcld = want_fc ? (wc * tc) / (wc * tc + tr) : 0.0;       // :335
    double cosb = gc;                                               // :338
    double fray = want_fr ? tr / (tr + wc * tc) : 0.0;              // :341
    double gcos2 = 0.5 * fray;                                      // :342
    double w0 = (tr * rf + tc * wc) / (tg + tr + tc);               // :346
    double w0nr = want_nr ? (tr * 0.99999 + tc * wc) / (tg + tr + tc) : 0.0;   // :350
    if (a.test_mode) {                                              // :372-399
        if (a.test_mode == 1) {          // 'rayleigh'
            dtau = tr; gcos2 = 0.5; fray = 1.0; fcld = 0.0;
        } else {                         // constant tau from the cloud opd
            dtau = tc; gcos2 = 0.0; fray = 0.0; fcld = 1.0;
        }
        if (dtau <= 0) dtau = 1e-10;
        cosb = gc;
        w0 = (wc <= 0) ? 1e-10 : wc;
        w0nr = w0;
    }
    tau_run += dtau;                                                // numba_cumsum (:353-354)
    if (a.dtau_og) a.dtau_og[q] = dtau;
    if (a.tau_og) a.tau_og[qn] = tau_run;
    if (a.w0_og) a.w0_og[q] = w0;
    if (a.cosb_og) a.cosb_og[q] = cosb;
    if (a.ftau_cld) a.ftau_cld[q] = fcld;
    if (a.ftau_ray) a.ftau_ray[q] = fray;
    if (a.gcos2) a.gcos2[q] = gcos2;
    if (a.w0_no_raman) a.w0_no_raman[q] = w0nr;
    if (a.delta_eddington) {                                        // :401-420
        const double f = ipow(cosb, a.stream);
        const double dtd = dtau * (1. - w0 * f);
        taud_run += dtd;
        if (a.f_deltaM) a.f_deltaM[q] = f;
        if (a.w0) a.w0[q] = w0 * (1. - f) / (1.0 - w0 * f);
        if (a.cosb) a.cosb[q] = (cosb - f) / (1. - f);
        if (a.dtau) a.dtau[q] = dtd;
        if (a.tau) a.tau[qn] = taud_run;
    } else {                                                        // :428-431
        if (a.f_deltaM) a.f_deltaM[q] = 0 * cosb;
        if (a.w0) a.w0[q] = w0;
        if (a.cosb) a.cosb[q] = cosb;
        if (a.dtau) a.dtau[q] = dtau;
        if (a.tau) a.tau[qn] = tau_run;
    }
}

// numpy.interp(x, xp, row) for one x against many rows (reference wavelength.regrid, wavelength.py:46-70): the bracket of x
// in xp once, then per row the slope form numpy evaluates -- shared by k_regrid_rows and the fused gas + mixing launch, so
// a cloud table interpolated where it is used carries the bits of the regridded plane.
struct RegridBracket {
    int j, j0, j1;
    bool knot;
    double xv, x0, x1;
};
__device__ __forceinline__ RegridBracket regrid_bracket(const double *xp, int nin, double xv)
{
    RegridBracket b;
    const int last = nin - 1;
    int j;                       // -1: left of the grid, nin: right of it, else xp[j] <= x (< xp[j+1])
    if (xv != xv) j = -2;
    else if (xv > xp[last]) j = nin;
    else if (xv < xp[0]) j = -1;
    else {
        int lo = 0, hi = last;   // xp[lo] <= x <= xp[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (xv >= xp[mid]) lo = mid;
            else hi = mid;
        }
        j = (xv >= xp[hi]) ? hi : lo;
    }
    const bool edge = (j < 0) || (j >= last);
    b.j = j;
    b.j0 = j < 0 ? 0 : (j >= last ? last : j);
    b.j1 = edge ? b.j0 : b.j0 + 1;
    b.xv = xv;
    b.x0 = xp[b.j0];
    b.x1 = xp[b.j1];
    b.knot = edge || (b.x0 == xv);
    return b;
}
__device__ __forceinline__ double regrid_value(const double *row, const RegridBracket &b)
{
#pragma clang fp contract(off)
    const double y0 = row[b.j0], y1 = row[b.j1];
    if (b.j == -2) return b.xv;
    if (b.knot) return y0;
    const double slope = (y1 - y0) / (b.x1 - b.x0);
    double v = slope * (b.xv - b.x0) + y0;
    if (v != v) {
        v = slope * (b.xv - b.x1) + y1;
        if (v != v && y0 == y1) v = y0;
    }
    return v;
}

struct GasArgs {
    int nlayer, nwno, nmol, ncont, nray;
    int ncolper;      // columns per wavelength of the molecular tables / taugas (correlated-k Gauss points)
    int mol_mode;     // 0 nearest row; 1 10**(sum_4 w log10 kappa) (optics.py:2290-2293);
                      // 2 exp(sum_4 w ln kappa), premixed correlated-k (optics.py:1152-1157)
    int cont_mode;    // 0 nearest-temperature row (optics.py:2298-2306);
                      // 1 exp((1-t) ln k_lo + t ln k_hi), correlated-k continuum (optics.py:1486-1489)
    const double *const *mol_tables, *const *cont_tables, *const *ray_tables;   // device arrays of device ptrs
    const int *mol_rows, *cont_rows;          // device
    const double *mol_wts, *mol_fac, *cont_wts, *cont_fac, *ray_fac;
    unsigned ncg, ntile;      // column groups and layer tiles of the 1-D grid
    double *taugas, *tauray;  // not written with fuse
    int fuse;                 // 1: mix_layer on every element straight from the sums (picaso_gas_compute_opacity_dev):
    MixArgs mix;              //    TAUGAS / TAURAY never travel through HBM; mix.tau / mix.tau_og must be NULL (k_level_sums)
};

// One lane per column, LT consecutive layers per thread: neighbouring layers mostly bracket the
// same (P,T) table rows, so a row value is fetched once per tile instead of once per layer (the
// tables are re-read nlayer/npt times otherwise: 1.4 GB of L2/MALL traffic for 5 molecules at
// 1e5 x 90, against 144 MB written).  Per element the sums run in the reference's order
// (continuum pairs, then molecules; optics.py:172-255), so the tiling does not change a bit.
// (round 1, TAUGAS / TAURAY out: 10 layers per tile.  The fused forms carry the mixing per element and ran fastest with 5 or
// 6: 113 us against 125 at 10, 129 at 3, 145 at 18 for 1e5 x 90 -- tools/gas_time.sh.)
#ifndef PZ_GAS_LT
#define PZ_GAS_LT 10
#endif
#ifndef PZ_GAS_LT_FUSED
#define PZ_GAS_LT_FUSED 6
#endif
#ifndef PZ_GAS_XCD
#define PZ_GAS_XCD 1
#endif
template <int FUSE> struct GasTile { static constexpr int LT = FUSE ? PZ_GAS_LT_FUSED : PZ_GAS_LT; };

template <int FUSE>      // 0: TAUGAS / TAURAY out; 1: mixing fused in; 2: the cloud-free form of 1
__global__ __launch_bounds__(256) void k_opacity_gas(const GasArgs a)
{
    constexpr int GAS_LT = GasTile<FUSE>::LT;
    // 1-D grid in XCD-aware order (consecutive workgroups go to consecutive XCDs, each with its own L2): the layer tiles
    // of one column group are dispatched 8 apart, all to XCD (column group % 8), close in time -- neighbouring tiles
    // bracket mostly the same (P,T) table rows, which the later ones then find in that L2 (HBM fetch of the 1e5 x 90
    // launch 340 -> 82 MB by PMC; 113 -> 110 us: the kernel is latency-, not bandwidth-bound, the traffic matters to what
    // runs next to it); the last ncg % 8 column groups go out in plain (tile, column group) order
    unsigned cg, tile;
    {
        const unsigned b = blockIdx.x, ncg = a.ncg, nrep = a.ntile, nfull = ncg & ~7u;
        if (!PZ_GAS_XCD) { cg = b % ncg; tile = b / ncg; }
        else if (b < nfull * nrep) {
            const unsigned per = 8u * nrep, chunk = b / per, rem = b - chunk * per;
            tile = rem >> 3;
            cg = chunk * 8u + (rem & 7u);
        } else {
            const unsigned idx = b - nfull * nrep, left = ncg - nfull;
            tile = idx / left;
            cg = nfull + (idx - tile * left);
        }
    }
    const long col = cg * (long)blockDim.x + threadIdx.x;
    const int l0 = tile * GAS_LT;
    const int nl = a.nlayer - l0 < GAS_LT ? a.nlayer - l0 : GAS_LT;
    const long nw = a.nwno, ncol = nw * a.ncolper;
    if (col >= ncol) return;
    const long w = (a.ncolper > 1) ? col / a.ncolper : col;
    double tg[GAS_LT];
#pragma unroll
    for (int l = 0; l < GAS_LT; ++l) tg[l] = 0.0;
    for (int c = 0; c < a.ncont; ++c) {
        const double *tab = a.cont_tables[c];
        int r0 = -1, r1 = -1;
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int l = 0; l < GAS_LT; ++l) {
            if (l < nl) {
                const int lay = l0 + l;
                double k;
                if (a.cont_mode == 1) {
                    const int b = (c * a.nlayer + lay) * 2;
                    const int q0 = a.cont_rows[b], q1 = a.cont_rows[b + 1];
                    if (q0 != r0) { r0 = q0; v0 = tab[(long)q0 * nw + w]; }
                    if (q1 != r1) { r1 = q1; v1 = tab[(long)q1 * nw + w]; }
                    k = fexp(a.cont_wts[b] * v0 + a.cont_wts[b + 1] * v1);
                } else {
                    const int q0 = a.cont_rows[c * a.nlayer + lay];
                    if (q0 != r0) { r0 = q0; v0 = tab[(long)q0 * nw + w]; }
                    k = v0;
                }
                tg[l] += k * a.cont_fac[c * a.nlayer + lay];
            }
        }
    }
    for (int m = 0; m < a.nmol; ++m) {
        const double *tab = a.mol_tables[m];
        int r[4] = {-1, -1, -1, -1};
        double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 0; l < GAS_LT; ++l) {
            if (l < nl) {
                const int lay = l0 + l;
                const int base = (m * a.nlayer + lay) * 4;
                double cx;
                if (a.mol_mode) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rq = a.mol_rows[base + q];          // wave-uniform
                        if (rq != r[q]) { r[q] = rq; v[q] = tab[(long)rq * ncol + col]; }
                    }
                    double lg = a.mol_wts[base] * v[0];
                    lg = lg + a.mol_wts[base + 1] * v[1];
                    lg = lg + a.mol_wts[base + 2] * v[2];
                    lg = lg + a.mol_wts[base + 3] * v[3];
                    cx = fexp(a.mol_mode == 1 ? lg * 2.302585092994046 : lg);
                } else {                 // nearest (p,T) row (optics.py:2351)
                    const int rq = a.mol_rows[base];
                    if (rq != r[0]) { r[0] = rq; v[0] = tab[(long)rq * ncol + col]; }
                    cx = v[0];
                }
                tg[l] += (cx * 6.02214086e+23) * a.mol_fac[m * a.nlayer + lay];    // optics.py:2294, :246-250, :1159
            }
        }
    }
    if constexpr (FUSE == 0) {
#pragma unroll
        for (int l = 0; l < GAS_LT; ++l)
            if (l < nl) a.taugas[(long)(l0 + l) * ncol + col] = tg[l];
    }
    if (col == w * a.ncolper) {      // Rayleigh has no Gauss-point axis (optics.py:265-277)
        double tr[GAS_LT];
#pragma unroll
        for (int l = 0; l < GAS_LT; ++l) tr[l] = 0.0;
        for (int q = 0; q < a.nray; ++q) {
            const double rv = a.ray_tables[q][w];
#pragma unroll
            for (int l = 0; l < GAS_LT; ++l)
                if (l < nl) tr[l] += rv * a.ray_fac[q * a.nlayer + l0 + l];   // :265-271
        }
        if constexpr (FUSE == 0) {
#pragma unroll
            for (int l = 0; l < GAS_LT; ++l)
                if (l < nl) a.tauray[(long)(l0 + l) * nw + w] = tr[l];
        }
        if constexpr (FUSE != 0) {
            // (ncolper = 1: col == w.)  The mixing of compute_opacity on the values just formed -- the same function
            // on the same operands as k_compute_opacity reads back from HBM, so the same bits; the level sums, the one
            // thing that runs down a column, are k_level_sums' (tau / tau_og are NULL here).
            const bool rf_plane = a.mix.raman && !a.mix.raman_row;
            const double rf_row = (a.mix.raman && a.mix.raman_row) ? a.mix.raman[w] : a.mix.raman_const;
            double run0 = 0.0, run1 = 0.0;
            if constexpr (FUSE == 2) {
                // The cloud-free spectrum (the host checked: no cloud planes, no test mode, only dtau / w0 /
                // w0_no_raman wanted): the same mix_layer on a copy of the arguments whose other pointers are literal
                // NULLs, so the compiler drops the eleven stores it cannot reach and the quotients only they need --
                // a specialisation by constant propagation, not a second formula.
                MixArgs m{};
                m.nlayer = a.mix.nlayer; m.nwno = a.mix.nwno; m.ncolper = 1;
                m.delta_eddington = a.mix.delta_eddington; m.stream = a.mix.stream;
                m.dtau = a.mix.dtau; m.w0 = a.mix.w0; m.w0_no_raman = a.mix.w0_no_raman;
#pragma unroll
                for (int l = 0; l < GAS_LT; ++l) {
                    if (l < nl) {
                        const long o = (long)(l0 + l) * nw + w;
                        mix_layer(m, o, o, tg[l], tr[l], 0.0, 0.0, 0.0, rf_plane ? a.mix.raman[o] : rf_row, run0, run1);
                    }
                }
            } else {
                const MixArgs &m = a.mix;
                RegridBracket cb{};
                if (m.cld_nin) cb = regrid_bracket(m.cld_xp, m.cld_nin, m.cld_x[w]);
#pragma unroll
                for (int l = 0; l < GAS_LT; ++l) {
                    if (l < nl) {
                        const long o = (long)(l0 + l) * nw + w;
                        double tc, wc, gc;
                        if (m.cld_nin) {
                            const double *row = m.cld_fp + (long)(l0 + l) * m.cld_nin;
                            tc = regrid_value(row, cb);
                            wc = regrid_value(row + (long)m.nlayer * m.cld_nin, cb);
                            gc = regrid_value(row + 2L * m.nlayer * m.cld_nin, cb);
                        } else {
                            tc = m.taucld ? m.taucld[o] : 0.0;
                            wc = m.w0c ? m.w0c[o] : 0.0;
                            gc = m.g0c ? m.g0c[o] : 0.0;
                        }
                        const double rf = rf_plane ? m.raman[o] : rf_row;
                        mix_layer(m, o, o, tg[l], tr[l], tc, wc, gc, rf, run0, run1);
                    }
                }
            }
        }
    }
}

// tau[0] = 0, tau[i + 1] = tau[i] + dtau[i] down every column (numba_cumsum, optics.py:353-354, 418-420): the level
// planes of the fused gas + mixing launch, whose elements are formed layer tile by layer tile.  Two planes per launch
// (the delta-scaled and the unscaled set); either pair may be NULL.
__global__ __launch_bounds__(256) void k_level_sums(int nlayer, long ncol, const double *__restrict__ d0,
                                                    double *__restrict__ t0, const double *__restrict__ d1,
                                                    double *__restrict__ t1)
{
#pragma clang fp contract(off)
    const long col = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const double *d = blockIdx.y == 0 ? d0 : d1;
    double *t = blockIdx.y == 0 ? t0 : t1;
    if (!t) return;
    double run = 0.0;
    t[col] = 0.0;
    // eight loads in flight per lane: a column is a chain of dependent adds, the loads are not
    int i = 0;
    for (; i + 8 <= nlayer; i += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = d[(long)(i + k) * ncol + col];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            run += v[k];
            t[(long)(i + k + 1) * ncol + col] = run;
        }
    }
    for (; i < nlayer; ++i) {
        run += d[(long)i * ncol + col];
        t[(long)(i + 1) * ncol + col] = run;
    }
}

__global__ __launch_bounds__(1024) void k_compute_opacity(const MixArgs a_in)
{
    MixArgs a = a_in;
    if (a.fm_lay) {                  // one facet of a facet-major stack per grid.y (wave-uniform pointer arithmetic)
        const long f = blockIdx.y;
        auto mv = [](auto *&p, long off) { if (p) p += off; };
        mv(a.taugas, f * a.fm_lay); mv(a.tauray, f * a.fm_ray);
        mv(a.taucld, f * a.fm_cld); mv(a.w0c, f * a.fm_cld); mv(a.g0c, f * a.fm_cld);
        if (!a.raman_row) mv(a.raman, f * a.fm_ray);
        mv(a.dtau, f * a.fm_lay); mv(a.w0, f * a.fm_lay); mv(a.cosb, f * a.fm_lay); mv(a.ftau_cld, f * a.fm_lay);
        mv(a.ftau_ray, f * a.fm_lay); mv(a.gcos2, f * a.fm_lay); mv(a.dtau_og, f * a.fm_lay); mv(a.w0_og, f * a.fm_lay);
        mv(a.cosb_og, f * a.fm_lay); mv(a.w0_no_raman, f * a.fm_lay); mv(a.f_deltaM, f * a.fm_lay);
        mv(a.tau, f * a.fm_lev); mv(a.tau_og, f * a.fm_lev);
    }
    const long col = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const bool facets = a.nfac > 1;
    const long nw = a.nwno, ncol = nw * (facets ? a.nfac : a.ncolper);
    if (col >= ncol) return;
    const long w = facets ? col / a.nfac : ((a.ncolper > 1) ? col / a.ncolper : col);
    const long fbase = facets ? (col - w * a.nfac) * a.nlayer : 0;     // facet-major row block of this column
    double tau_run = 0.0, taud_run = 0.0;
    const bool rf_plane = a.raman && !a.raman_row;
    const double rf_row = (a.raman && a.raman_row) ? a.raman[w] : a.raman_const;
    // every output plane is optional: a caller that runs only the thermal (or only the transmission)
    // leg asks for 3 (1) of the 13 planes and the kernel does not write the rest
    if (a.tau_og) a.tau_og[col] = 0.0;
    if (a.tau) a.tau[col] = 0.0;
    for (int i = 0; i < a.nlayer; ++i) {
        const long o = (long)i * ncol + col;
        // gas / Rayleigh / Raman: per wavelength (monochromatic), per column (correlated-k) or per
        // (facet, layer) row; cloud: per wavelength, or per column in facet mode (NULL = no cloud)
        const long og = facets ? (fbase + i) * nw + w : o;
        const long ow = facets ? og : (long)i * nw + w;
        const long oc = facets ? o : (long)i * nw + w;
        const double tg = a.taugas[og], tr = a.tauray[ow];
        const double tc = a.taucld ? a.taucld[oc] : 0.0, wc = a.w0c ? a.w0c[oc] : 0.0, gc = a.g0c ? a.g0c[oc] : 0.0;
        const double rf = rf_plane ? a.raman[ow] : rf_row;
        mix_layer(a, o, o + ncol, tg, tr, tc, wc, gc, rf, tau_run, taud_run);
    }
}

// Facet mode with the transposition staged through LDS.  The gas / Rayleigh / Raman rows arrive facet-major
// (nfac, nlayer, nwno) -- what the batched gas launch writes, one coalesced row per (facet, layer) -- and the
// planes leave with the facet index fastest, (rows, nwno, nfac), what get_reflected_3d reads.  A block owns
// 1024 consecutive output columns = all facets of 1024/nfac (+ partial) wavelengths; per layer its threads
// first copy the [facet][wavelength] tile of each input row set into LDS along the rows (16 consecutive
// wavelengths per 128-byte line at 64 facets), then every thread picks its own (wavelength, facet) element.
// Reading the rows directly, 8 bytes per lane from 64 different lines, relied on the 16 waves of a block
// finding each other's lines in L1 and ran at 1.4 TB/s; the loads of layer i+1 are in flight while layer i
// is mixed.
constexpr int MIXF_BLOCK = 1024, MIXF_TILE = 2048;
__global__ __launch_bounds__(MIXF_BLOCK) void k_compute_opacity_facets(const MixArgs a)
{
    __shared__ double tile[3][MIXF_TILE];
    const long nw = a.nwno;
    const int nfac = a.nfac, n = a.nlayer;
    const long ncol = nw * nfac;
    const long c0 = blockIdx.x * (long)MIXF_BLOCK;
    const long col = c0 + threadIdx.x;
    const bool active = col < ncol;
    const long w_first = c0 / nfac;
    const long c_last = (c0 + MIXF_BLOCK - 1 < ncol) ? c0 + MIXF_BLOCK - 1 : ncol - 1;
    const int nwl = (int)(c_last / nfac - w_first) + 1;          // wavelengths this block touches
    const int ldp = nwl | 1;                                       // odd row pitch of the LDS tile
    const int nel = nfac * nwl;                                    // <= MIXF_TILE (launcher)
    // staging assignment: element e = (facet e / nwl, wavelength e % nwl), at most two per thread
    long src[2];
    int dst[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = threadIdx.x + r * MIXF_BLOCK;
        const int fe = e / nwl, j = e - fe * nwl;
        src[r] = (e < nel) ? ((long)fe * n) * nw + w_first + j : -1;
        dst[r] = fe * ldp + j;
    }
    const long w = active ? col / nfac : 0;
    const int my = active ? (int)(col - w * nfac) * ldp + (int)(w - w_first) : 0;
    const bool has_rf = a.raman != nullptr && !a.raman_row;
    const double rf_row = (active && a.raman && a.raman_row) ? a.raman[w] : a.raman_const;
    double pg[2], pr[2], pf[2];
    auto fetch = [&](int i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (src[r] >= 0) {
                const long o = src[r] + (long)i * nw;
                pg[r] = a.taugas[o];
                pr[r] = a.tauray[o];
                if (has_rf) pf[r] = a.raman[o];
            }
    };
    double tau_run = 0.0, taud_run = 0.0;
    if (active) {
        if (a.tau_og) a.tau_og[col] = 0.0;
        if (a.tau) a.tau[col] = 0.0;
    }
    fetch(0);
    for (int i = 0; i < n; ++i) {
        __syncthreads();                       // the tile of the layer above has been consumed
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (src[r] >= 0) {
                tile[0][dst[r]] = pg[r];
                tile[1][dst[r]] = pr[r];
                if (has_rf) tile[2][dst[r]] = pf[r];
            }
        __syncthreads();
        if (i + 1 < n) fetch(i + 1);
        if (active) {
            const long o = (long)i * ncol + col;
            const double tg = tile[0][my], tr = tile[1][my];
            const double rf = has_rf ? tile[2][my] : rf_row;
            const double tc = a.taucld ? a.taucld[o] : 0.0, wc = a.w0c ? a.w0c[o] : 0.0, gc = a.g0c ? a.g0c[o] : 0.0;
            mix_layer(a, o, o + ncol, tg, tr, tc, wc, gc, rf, tau_run, taud_run);
        }
    }
}

// dst[(r, w, f)] = src[(r, w)] * scale[f]: a (rows, nwno) plane shared by all facets laid out with the facet
// index fastest (what the 3-D solvers and the facet mixing kernel read), optionally scaled per facet.  64-bit
// element indices: a 64-facet plane at 1e5 wavelengths x 91 levels is 4.7 GB.
constexpr int BCAST_MAX_FACETS = 256;
struct BcastArgs {
    long nin;                       // rows * nwno
    int nfac;
    const double *src;
    double *dst;
    int has_scale;
    double scale[BCAST_MAX_FACETS];
};
__global__ __launch_bounds__(256) void k_broadcast_facets(const BcastArgs a)
{
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= a.nin * a.nfac) return;
    const long q = e / a.nfac;
    const int f = (int)(e - q * a.nfac);
    const double v = a.src[q];
    a.dst[e] = a.has_scale ? v * a.scale[f] : v;
}

// Rows of a (nrows, nin) table on the increasing grid xp -> (nrows, nwno) on x, piecewise linear with the end
// values held outside the grid: numpy.interp per row, which is what the reference's wavelength.regrid does for the
// cloud tables (wavelength.py:46-70 from atmsetup.py:609-622) -- on the host, 0.5 s per table at 1e5 wavelengths.
// Bit for bit numpy's arithmetic (numpy/_core/src/multiarray/compiled_base.c:arr_interp): j with
// xp[j] <= x < xp[j+1]; a knot hit returns fp[j]; otherwise slope = (fp[j+1]-fp[j])/(xp[j+1]-xp[j]) (a correctly
// rounded division), slope*(x-xp[j]) + fp[j] as a separate multiply and add, and the NaN retry from the other side.
// One thread per output column: the bracket search is done once for all rows; writes are coalesced.
constexpr int REGRID_LDS = 4096;
struct RegridArgs {
    int nrows, nin;
    long nwno;
    const double *xp, *fp, *x;
    double scale;
    int has_scale;
    double *out;
};
__global__ __launch_bounds__(256) void k_regrid_rows(const RegridArgs a)
{
#pragma clang fp contract(off)
    __shared__ double sxp[REGRID_LDS];
    const bool in_lds = a.nin <= REGRID_LDS;
    if (in_lds) {
        for (int i = threadIdx.x; i < a.nin; i += blockDim.x) sxp[i] = a.xp[i];
        __syncthreads();
    }
    const double *xp = in_lds ? sxp : a.xp;
    const long w = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (w >= a.nwno) return;
    const RegridBracket b = regrid_bracket(xp, a.nin, a.x[w]);
    for (int r = 0; r < a.nrows; ++r) {
        const double v = regrid_value(a.fp + (long)r * a.nin, b);
        a.out[(long)r * a.nwno + w] = a.has_scale ? a.scale * v : v;
    }
}

// The same for cloud tables that differ from facet to facet (the 3-D path: clouds_3d on the 196-point grid of virga,
// justdoit.py:4515-4620 -- the reference regrids facet by facet on the host, atmsetup.py:609-622): fp is
// (nlayer, nfacets, nin), out the (nlayer, nwno, nfacets) plane the facet mixing kernel and the 3-D solvers read, facet
// index fastest.  One thread per (wavelength, facet) element, so that the writes are coalesced; the bracket search is
// repeated per facet (a few LDS reads), the table rows come from L2.  Same arithmetic as k_regrid_rows.
struct RegridFacetArgs {
    int nlayer, nfac, nin;
    long nwno;
    const double *xp, *fp, *x;
    double *out;
};
__global__ __launch_bounds__(256) void k_regrid_facets(const RegridFacetArgs a)
{
#pragma clang fp contract(off)
    __shared__ double sxp[REGRID_LDS];
    const bool in_lds = a.nin <= REGRID_LDS;
    if (in_lds) {
        for (int i = threadIdx.x; i < a.nin; i += blockDim.x) sxp[i] = a.xp[i];
        __syncthreads();
    }
    const double *xp = in_lds ? sxp : a.xp;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long ncol = a.nwno * a.nfac;
    if (e >= ncol) return;
    const long w = e / a.nfac;
    const int f = (int)(e - w * a.nfac);
    const RegridBracket b = regrid_bracket(xp, a.nin, a.x[w]);
    for (int i = 0; i < a.nlayer; ++i)
        a.out[(long)i * ncol + e] = regrid_value(a.fp + ((long)i * a.nfac + f) * a.nin, b);
}

// Oklopcic (2016) Raman factor plane, reference optics.compute_raman (optics.py:434-494): for every transition i of the
// H2 table, Q_i(w) = c_i / w^3 / (w + dnu_i) and (dnu_i != 0) Q_i(w) * shift_i(w) depend on the wavelength only and
// arrive as resident (ntrans, nwno) tables formed once with numpy's own pow and divisions; the layer enters through the
// 10 rotational populations J(j, layer).  Per (layer, wavelength):
//     ray = sum_{dnu_i = 0} J Q_i,  w_shift = sum J (Q_i shift_i),  wo_shift = sum J Q_i,
//     out = min((ray + w_shift) / (ray + wo_shift), cap)
// with the reference's order of accumulation (i ascending, each term one multiply then one add: `acc += np.outer(J, Q)`),
// so the plane is bit for bit the host's -- which takes 0.97 s per call at 1e5 wavelengths x 90 layers.
// One thread per wavelength and RAMAN_LCH layers: 3 x RAMAN_LCH accumulators in registers, the tables read once per pass.
constexpr int RAMAN_LCH = 30;
constexpr int RAMAN_MAX_TRANS = 1024;
struct RamanArgs {
    int nlayer, ntrans;
    long nwno;
    const double *Q, *QS;
    const double *tab;      // device: J (10, nlayer), then ntrans x (j_initial, is_rayleigh) as doubles
    double cap;
    double *out;
};
__global__ __launch_bounds__(256) void k_raman_oklopcic(const RamanArgs a)
{
#pragma clang fp contract(off)
    // the populations of this pass's layers and the transition table in LDS: every lane reads the same element
    // (a broadcast ds_read) instead of a dependent global load per (transition, layer)
    __shared__ double sJ[10][RAMAN_LCH];
    __shared__ int sji[RAMAN_MAX_TRANS], sray[RAMAN_MAX_TRANS];
    const int l0 = blockIdx.y * RAMAN_LCH, n = a.nlayer;
    const double *meta = a.tab + 10 * (long)n;
    for (int e = threadIdx.x; e < 10 * RAMAN_LCH; e += blockDim.x) {
        const int j = e / RAMAN_LCH, l = e - j * RAMAN_LCH;
        sJ[j][l] = a.tab[(long)j * n + min(l0 + l, n - 1)];
    }
    for (int i = threadIdx.x; i < a.ntrans; i += blockDim.x) {
        sji[i] = (int)meta[2 * i];
        sray[i] = meta[2 * i + 1] != 0.0;
    }
    __syncthreads();
    const long w = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (w >= a.nwno) return;
    double ray[RAMAN_LCH], ws[RAMAN_LCH], wo[RAMAN_LCH];
#pragma unroll
    for (int l = 0; l < RAMAN_LCH; ++l) ray[l] = ws[l] = wo[l] = 0.0;
    for (int i = 0; i < a.ntrans; ++i) {
        const double *J = sJ[sji[i]];
        const double q = a.Q[(long)i * a.nwno + w];
        if (sray[i]) {
#pragma unroll
            for (int l = 0; l < RAMAN_LCH; ++l) ray[l] = ray[l] + J[l] * q;
        } else {
            const double qs = a.QS[(long)i * a.nwno + w];
#pragma unroll
            for (int l = 0; l < RAMAN_LCH; ++l) {
                const double j = J[l];
                ws[l] = ws[l] + j * qs;
                wo[l] = wo[l] + j * q;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < RAMAN_LCH; ++l)
        if (l0 + l < n) {
            const double v = (ray[l] + ws[l]) / (ray[l] + wo[l]);
            a.out[(long)(l0 + l) * a.nwno + w] = (v > a.cap) ? a.cap : v;      // np.minimum: NaN stays
        }
}

}  // namespace pz

using namespace pz;

extern "C" {

int picaso_raman_oklopcic_dev(picaso_ctx *ctx, int nlayer, long nwno, int ntrans, const double *Q, const double *QS,
                              const int *j_initial, const int *is_rayleigh, const double *j_at_temp, double cap,
                              double *out)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlayer < 1 || nwno < 1 || ntrans < 1 || !Q || !QS || !j_initial || !is_rayleigh || !j_at_temp || !out)
        return fail(ctx, "raman_oklopcic: bad arguments");
    if (ntrans > RAMAN_MAX_TRANS) return fail(ctx, "raman_oklopcic: at most %d transitions", RAMAN_MAX_TRANS);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> tab((size_t)10 * nlayer + 2 * (size_t)ntrans);
    memcpy(tab.data(), j_at_temp, sizeof(double) * 10 * nlayer);
    for (int i = 0; i < ntrans; ++i) {
        if (j_initial[i] < 0 || j_initial[i] > 9) return fail(ctx, "raman_oklopcic: j_initial[%d] = %d not in 0..9", i, j_initial[i]);
        tab[(size_t)10 * nlayer + 2 * i] = j_initial[i];
        tab[(size_t)10 * nlayer + 2 * i + 1] = is_rayleigh[i] ? 1.0 : 0.0;
    }
    const void *d_tab = nullptr;
    PZ_TRY(table_upload(ctx, tab.data(), sizeof(double) * tab.size(), &d_tab));
    RamanArgs a{};
    a.nlayer = nlayer;
    a.ntrans = ntrans;
    a.nwno = nwno;
    a.Q = Q;
    a.QS = QS;
    a.tab = (const double *)d_tab;
    a.cap = cap;
    a.out = out;
    const dim3 grid((unsigned)((nwno + 255) / 256), (unsigned)((nlayer + RAMAN_LCH - 1) / RAMAN_LCH));
    hipLaunchKernelGGL(k_raman_oklopcic, grid, dim3(256), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int picaso_regrid_rows_dev(picaso_ctx *ctx, int nrows, int nin, long nwno, const double *xp, const double *fp,
                           const double *x, const double *scale, double *out)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nrows < 1 || nin < 2 || nwno < 1 || !xp || !fp || !x || !out) return fail(ctx, "regrid_rows: bad arguments");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    RegridArgs a{};
    a.nrows = nrows;
    a.nin = nin;
    a.nwno = nwno;
    a.xp = xp;
    a.fp = fp;
    a.x = x;
    a.has_scale = scale != nullptr;
    a.scale = scale ? *scale : 1.0;
    a.out = out;
    hipLaunchKernelGGL(k_regrid_rows, dim3((unsigned)((nwno + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int picaso_regrid_facets_dev(picaso_ctx *ctx, int nlayer, int nfacets, int nin, long nwno, const double *xp,
                             const double *fp, const double *x, double *out)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlayer < 1 || nfacets < 1 || nin < 2 || nwno < 1 || !xp || !fp || !x || !out)
        return fail(ctx, "regrid_facets: bad arguments");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    RegridFacetArgs a{};
    a.nlayer = nlayer; a.nfac = nfacets; a.nin = nin; a.nwno = nwno;
    a.xp = xp; a.fp = fp; a.x = x; a.out = out;
    const long total = nwno * nfacets;
    hipLaunchKernelGGL(k_regrid_facets, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int picaso_broadcast_facets_dev(picaso_ctx *ctx, size_t nrows, int nwno, int nfacets, const double *src,
                                const double *facet_scale, double *dst)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nrows < 1 || nwno < 1 || nfacets < 1 || !src || !dst) return fail(ctx, "broadcast_facets: bad arguments");
    if (nfacets > BCAST_MAX_FACETS) return fail(ctx, "broadcast_facets: at most %d facets", BCAST_MAX_FACETS);
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    BcastArgs a{};
    a.nin = (long)nrows * nwno;
    a.nfac = nfacets;
    a.src = src;
    a.dst = dst;
    a.has_scale = facet_scale != nullptr;
    for (int f = 0; f < nfacets; ++f) a.scale[f] = facet_scale ? facet_scale[f] : 1.0;
    const long total = a.nin * nfacets;
    hipLaunchKernelGGL(k_broadcast_facets, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// the gas stage; `mix` != NULL: with the mixing fused in (taugas / tauray may then be NULL)
static int gas_launch(picaso_ctx *ctx, int nlayer, int nwno, int ngauss, int mol_mode, int nmol,
                      const double *const *mol_tables, const int *mol_rows,
                      const double *mol_wts, const double *mol_fac, int cont_mode, int ncont,
                      const double *const *cont_tables, const int *cont_rows,
                      const double *cont_wts, const double *cont_fac, int nray,
                      const double *const *ray_tables, const double *ray_fac, double *taugas,
                      double *tauray, const MixArgs *mix)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlayer < 1 || nwno < 1 || nmol < 0 || ncont < 0 || nray < 0) return fail(ctx, "opacity_gas: bad sizes");
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "opacity_gas: ngauss must be 1..%d", MAX_CK_GAUSS);
    if (mol_mode < 0 || mol_mode > 2 || cont_mode < 0 || cont_mode > 1) return fail(ctx, "opacity_gas: bad mode");
    if (cont_mode == 1 && ncont > 0 && !cont_wts) return fail(ctx, "opacity_gas: cont_mode=1 needs cont_wts");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    // pack the small per-layer host tables into one ring slot
    const int crow = (cont_mode == 1) ? 2 : 1;
    const size_t n_ptr = (size_t)nmol + ncont + nray;
    const size_t b_ptr = align_up(n_ptr * sizeof(double *), 8);
    const size_t b_mrow = align_up((size_t)nmol * nlayer * 4 * sizeof(int), 8);
    const size_t b_crow = align_up((size_t)ncont * nlayer * crow * sizeof(int), 8);
    const size_t b_mw = (size_t)nmol * nlayer * 4 * sizeof(double);
    const size_t b_mf = (size_t)nmol * nlayer * sizeof(double);
    const size_t b_cw = (cont_mode == 1) ? (size_t)ncont * nlayer * 2 * sizeof(double) : 0;
    const size_t b_cf = (size_t)ncont * nlayer * sizeof(double);
    const size_t b_rf = (size_t)nray * nlayer * sizeof(double);
    std::vector<char> buf(b_ptr + b_mrow + b_crow + b_mw + b_mf + b_cw + b_cf + b_rf + 64);
    char *p = buf.data();
    const double **h_ptr = (const double **)p;
    for (int m = 0; m < nmol; ++m) h_ptr[m] = mol_tables[m];
    for (int c = 0; c < ncont; ++c) h_ptr[nmol + c] = cont_tables[c];
    for (int r = 0; r < nray; ++r) h_ptr[nmol + ncont + r] = ray_tables[r];
    size_t off = b_ptr;
    const size_t o_mrow = off; if (nmol) memcpy(p + off, mol_rows, (size_t)nmol * nlayer * 4 * sizeof(int)); off += b_mrow;
    const size_t o_crow = off; if (ncont) memcpy(p + off, cont_rows, (size_t)ncont * nlayer * crow * sizeof(int)); off += b_crow;
    const size_t o_mw = off; if (nmol) memcpy(p + off, mol_wts, b_mw); off += b_mw;
    const size_t o_mf = off; if (nmol) memcpy(p + off, mol_fac, b_mf); off += b_mf;
    const size_t o_cw = off; if (ncont && b_cw) memcpy(p + off, cont_wts, b_cw); off += b_cw;
    const size_t o_cf = off; if (ncont) memcpy(p + off, cont_fac, b_cf); off += b_cf;
    const size_t o_rf = off; if (nray) memcpy(p + off, ray_fac, b_rf); off += b_rf;
    const void *d = nullptr;
    PZ_TRY(table_upload(ctx, p, off, &d));
    const char *dc = (const char *)d;
    GasArgs a{};
    a.nlayer = nlayer; a.nwno = nwno; a.nmol = nmol; a.ncont = ncont; a.nray = nray;
    a.ncolper = ngauss; a.mol_mode = mol_mode; a.cont_mode = cont_mode;
    a.mol_tables = (const double *const *)dc;
    a.cont_tables = a.mol_tables + nmol;
    a.ray_tables = a.cont_tables + ncont;
    a.mol_rows = (const int *)(dc + o_mrow);
    a.cont_rows = (const int *)(dc + o_crow);
    a.mol_wts = (const double *)(dc + o_mw);
    a.mol_fac = (const double *)(dc + o_mf);
    a.cont_wts = (const double *)(dc + o_cw);
    a.cont_fac = (const double *)(dc + o_cf);
    a.ray_fac = (const double *)(dc + o_rf);
    a.taugas = taugas; a.tauray = tauray;
    if (mix) {
        const MixArgs &m = *mix;
        const bool lean = !m.taucld && !m.w0c && !m.g0c && !m.cld_nin && !m.test_mode && !m.cosb && !m.ftau_cld && !m.ftau_ray &&
                          !m.gcos2 && !m.dtau_og && !m.w0_og && !m.cosb_og && !m.f_deltaM;
        a.fuse = lean ? 2 : 1;
        a.mix = m;
    } else if (!taugas || !tauray) {
        return fail(ctx, "opacity_gas: null output");
    }
    const int block = 256;
    const long ncol = (long)nwno * ngauss;
    const int lt = a.fuse ? GasTile<1>::LT : GasTile<0>::LT;
    a.ncg = (unsigned)((ncol + block - 1) / block);
    a.ntile = (unsigned)((nlayer + lt - 1) / lt);
    if ((double)a.ncg * a.ntile > 2147483647.0) return fail(ctx, "opacity_gas: grid too large");
    dim3 grid(a.ncg * a.ntile);
    if (a.fuse == 2) hipLaunchKernelGGL(k_opacity_gas<2>, grid, dim3(block), 0, ctx->stream, a);
    else if (a.fuse == 1) hipLaunchKernelGGL(k_opacity_gas<1>, grid, dim3(block), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_opacity_gas<0>, grid, dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int picaso_opacity_gas_ck_dev(picaso_ctx *ctx, int nlayer, int nwno, int ngauss, int mol_mode, int nmol,
                              const double *const *mol_tables, const int *mol_rows,
                              const double *mol_wts, const double *mol_fac, int cont_mode, int ncont,
                              const double *const *cont_tables, const int *cont_rows,
                              const double *cont_wts, const double *cont_fac, int nray,
                              const double *const *ray_tables, const double *ray_fac, double *taugas,
                              double *tauray)
{
    return gas_launch(ctx, nlayer, nwno, ngauss, mol_mode, nmol, mol_tables, mol_rows, mol_wts, mol_fac, cont_mode, ncont,
                      cont_tables, cont_rows, cont_wts, cont_fac, nray, ray_tables, ray_fac, taugas, tauray, nullptr);
}

int picaso_level_sums_dev(picaso_ctx *ctx, int nlayer, long ncol, const double *dtau, double *tau, const double *dtau_og,
                          double *tau_og)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlayer < 1 || ncol < 1) return fail(ctx, "level_sums: bad sizes");
    if ((tau && !dtau) || (tau_og && !dtau_og)) return fail(ctx, "level_sums: a level plane without its layer plane");
    if (!tau && !tau_og) return 0;
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid((unsigned)((ncol + 255) / 256), (tau && tau_og) ? 2u : 1u);
    if (tau)
        hipLaunchKernelGGL(k_level_sums, grid, dim3(256), 0, ctx->stream, nlayer, ncol, dtau, tau, dtau_og, tau_og);
    else
        hipLaunchKernelGGL(k_level_sums, grid, dim3(256), 0, ctx->stream, nlayer, ncol, dtau_og, tau_og, dtau_og, tau_og);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int picaso_gas_compute_opacity_dev(picaso_ctx *ctx, int nlayer, int nwno, int mol_mode, int nmol,
                                   const double *const *mol_tables, const int *mol_rows, const double *mol_wts,
                                   const double *mol_fac, int cont_mode, int ncont, const double *const *cont_tables,
                                   const int *cont_rows, const double *cont_wts, const double *cont_fac, int nray,
                                   const double *const *ray_tables, const double *ray_fac, const double *taucld,
                                   const double *w0_cld, const double *g0_cld, const double *raman_factor,
                                   int raman_rows, double raman_const, int test_mode, int delta_eddington, int stream,
                                   double *dtau, double *tau, double *w0, double *cosb, double *ftau_cld,
                                   double *ftau_ray, double *gcos2, double *dtau_og, double *tau_og, double *w0_og,
                                   double *cosb_og, double *w0_no_raman, double *f_deltaM, int level_sums, int cld_nin,
                                   const double *cld_xp, const double *cld_fp, const double *cld_x)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlayer < 1 || nwno < 1) return fail(ctx, "gas_compute_opacity: bad sizes");
    if (cld_nin < 0 || (cld_nin > 0 && (cld_nin < 2 || !cld_xp || !cld_fp || !cld_x || taucld || w0_cld || g0_cld)))
        return fail(ctx, "gas_compute_opacity: cloud tables need cld_nin >= 2, their grid, their rows and the wavenumbers, "
                         "and exclude cloud planes");
    if (test_mode < 0 || test_mode > 2) return fail(ctx, "gas_compute_opacity: test_mode must be 0, 1 or 2");
    if (stream != 2 && stream != 4) return fail(ctx, "gas_compute_opacity: stream must be 2 or 4");
    if (raman_factor && raman_rows != 0 && raman_rows != nlayer)
        return fail(ctx, "gas_compute_opacity: raman_rows must be nlayer (planes) or 0 (one row for all layers)");
    if ((tau && !dtau) || (tau_og && !dtau_og))
        return fail(ctx, "gas_compute_opacity: tau / tau_og are the running sums of the dtau / dtau_og planes: ask for those too");
    MixArgs m{};
    m.nlayer = nlayer; m.nwno = nwno; m.ncolper = 1; m.nfac = 0; m.test_mode = test_mode;
    m.delta_eddington = delta_eddington; m.stream = stream;
    m.taucld = taucld; m.w0c = w0_cld; m.g0c = g0_cld; m.raman = raman_factor; m.raman_const = raman_const;
    m.raman_row = raman_factor && raman_rows == 0;
    m.cld_nin = cld_nin; m.cld_xp = cld_xp; m.cld_fp = cld_fp; m.cld_x = cld_x;
    m.dtau = dtau; m.tau = nullptr; m.w0 = w0; m.cosb = cosb; m.ftau_cld = ftau_cld; m.ftau_ray = ftau_ray;
    m.gcos2 = gcos2; m.dtau_og = dtau_og; m.tau_og = nullptr; m.w0_og = w0_og; m.cosb_og = cosb_og;
    m.w0_no_raman = w0_no_raman; m.f_deltaM = f_deltaM;
    PZ_TRY(gas_launch(ctx, nlayer, nwno, 1, mol_mode, nmol, mol_tables, mol_rows, mol_wts, mol_fac, cont_mode, ncont,
                      cont_tables, cont_rows, cont_wts, cont_fac, nray, ray_tables, ray_fac, nullptr, nullptr, &m));
    if (level_sums) PZ_TRY(picaso_level_sums_dev(ctx, nlayer, nwno, dtau, tau, dtau_og, tau_og));
    return 0;
}

int picaso_opacity_gas_dev(picaso_ctx *ctx, int nlayer, int nwno, int linear, int nmol,
                           const double *const *mol_tables, const int *mol_rows,
                           const double *mol_wts, const double *mol_fac, int ncont,
                           const double *const *cont_tables, const int *cont_rows,
                           const double *cont_fac, int nray, const double *const *ray_tables,
                           const double *ray_fac, double *taugas, double *tauray)
{
    return picaso_opacity_gas_ck_dev(ctx, nlayer, nwno, 1, linear ? 1 : 0, nmol, mol_tables, mol_rows, mol_wts,
                                     mol_fac, 0, ncont, cont_tables, cont_rows, nullptr, cont_fac, nray,
                                     ray_tables, ray_fac, taugas, tauray);
}

int picaso_compute_opacity_ck_dev(picaso_ctx *ctx, int nlayer, int nwno, int ngauss, const double *taugas,
                                  const double *tauray, const double *taucld, const double *w0_cld,
                                  const double *g0_cld, const double *raman_factor, int raman_rows,
                                  double raman_const, int test_mode, int delta_eddington, int stream,
                                  double *dtau, double *tau, double *w0, double *cosb,
                                  double *ftau_cld, double *ftau_ray, double *gcos2, double *dtau_og,
                                  double *tau_og, double *w0_og, double *cosb_og, double *w0_no_raman,
                                  double *f_deltaM)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlayer < 1 || nwno < 1) return fail(ctx, "compute_opacity: bad sizes");
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "compute_opacity: ngauss must be 1..%d", MAX_CK_GAUSS);
    if (test_mode < 0 || test_mode > 2) return fail(ctx, "compute_opacity: test_mode must be 0, 1 or 2");
    if (stream != 2 && stream != 4) return fail(ctx, "compute_opacity: stream must be 2 or 4");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    MixArgs a{};
    a.nlayer = nlayer; a.nwno = nwno; a.ncolper = ngauss; a.nfac = 0; a.test_mode = test_mode;
    a.delta_eddington = delta_eddington;
    a.stream = stream; a.taugas = taugas; a.tauray = tauray; a.taucld = taucld; a.w0c = w0_cld;
    a.g0c = g0_cld; a.raman = raman_factor; a.raman_const = raman_const;
    if (raman_factor && raman_rows != 0 && raman_rows != nlayer)
        return fail(ctx, "compute_opacity: raman_rows must be nlayer (planes) or 0 (one row for all layers)");
    a.raman_row = raman_factor && raman_rows == 0;
    a.dtau = dtau; a.tau = tau; a.w0 = w0; a.cosb = cosb; a.ftau_cld = ftau_cld; a.ftau_ray = ftau_ray;
    a.gcos2 = gcos2; a.dtau_og = dtau_og; a.tau_og = tau_og; a.w0_og = w0_og; a.cosb_og = cosb_og;
    a.w0_no_raman = w0_no_raman; a.f_deltaM = f_deltaM;
    const int block = 256;
    const long ncol = (long)nwno * ngauss;
    hipLaunchKernelGGL(k_compute_opacity, dim3((unsigned)((ncol + block - 1) / block)), dim3(block), 0,
                       ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

// The facet loop of the 3-D branch for correlated-k tables (reference justdoit.py:437-471), facet-major: ONE launch, grid.y =
// facet (round 5: it was one k_compute_opacity launch per facet -- 64 x 85 us of a 7.6 ms spectrum at 661 bins x 8 Gauss
// points, each a 21-workgroup launch that lasts as long as one column's 90-layer loop).  Per element the arithmetic of
// picaso_compute_opacity_ck_dev on that facet: same bits.
int picaso_compute_opacity_facet_major_ck_dev(picaso_ctx *ctx, int nfacets, int nlayer, int nwno, int ngauss,
                                              const double *taugas, const double *tauray, const double *taucld,
                                              const double *w0_cld, const double *g0_cld, long cloud_stride,
                                              const double *raman_factor, int raman_rows, double raman_const,
                                              int test_mode, int delta_eddington, int stream, double *dtau, double *tau,
                                              double *w0, double *cosb, double *ftau_cld, double *ftau_ray, double *gcos2,
                                              double *dtau_og, double *tau_og, double *w0_og, double *cosb_og,
                                              double *w0_no_raman, double *f_deltaM)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nfacets < 1 || nfacets > 65535 || nlayer < 1 || nwno < 1)
        return fail(ctx, "compute_opacity_facet_major_ck: bad sizes");
    if (!taugas || !tauray) return fail(ctx, "compute_opacity_facet_major_ck: taugas and tauray are required");
    if (cloud_stride < 0) return fail(ctx, "compute_opacity_facet_major_ck: cloud_stride must be >= 0");
    if (ngauss < 1 || ngauss > MAX_CK_GAUSS) return fail(ctx, "compute_opacity: ngauss must be 1..%d", MAX_CK_GAUSS);
    if (test_mode < 0 || test_mode > 2) return fail(ctx, "compute_opacity: test_mode must be 0, 1 or 2");
    if (stream != 2 && stream != 4) return fail(ctx, "compute_opacity: stream must be 2 or 4");
    if (raman_factor && raman_rows != 0 && raman_rows != nlayer)
        return fail(ctx, "compute_opacity: raman_rows must be nlayer (planes) or 0 (one row for all layers)");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    MixArgs a{};
    a.nlayer = nlayer; a.nwno = nwno; a.ncolper = ngauss; a.nfac = 0; a.test_mode = test_mode;
    a.delta_eddington = delta_eddington;
    a.stream = stream; a.taugas = taugas; a.tauray = tauray; a.taucld = taucld; a.w0c = w0_cld;
    a.g0c = g0_cld; a.raman = raman_factor; a.raman_const = raman_const;
    a.raman_row = raman_factor && raman_rows == 0;
    a.dtau = dtau; a.tau = tau; a.w0 = w0; a.cosb = cosb; a.ftau_cld = ftau_cld; a.ftau_ray = ftau_ray;
    a.gcos2 = gcos2; a.dtau_og = dtau_og; a.tau_og = tau_og; a.w0_og = w0_og; a.cosb_og = cosb_og;
    a.w0_no_raman = w0_no_raman; a.f_deltaM = f_deltaM;
    const long ncol = (long)nwno * ngauss;
    a.fm_lay = (long)nlayer * ncol;
    a.fm_lev = (long)(nlayer + 1) * ncol;
    a.fm_ray = (long)nlayer * nwno;
    a.fm_cld = cloud_stride;
    const int block = 256;
    hipLaunchKernelGGL(k_compute_opacity, dim3((unsigned)((ncol + block - 1) / block), (unsigned)nfacets), dim3(block), 0,
                       ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int picaso_compute_opacity_facets_dev(picaso_ctx *ctx, int nlayer, int nwno, int nfacets,
                                      const double *taugas, const double *tauray, const double *taucld,
                                      const double *w0_cld, const double *g0_cld, const double *raman_factor, int raman_rows,
                                      double raman_const, int test_mode, int delta_eddington, int stream,
                                      double *dtau, double *tau, double *w0, double *cosb,
                                      double *ftau_cld, double *ftau_ray, double *gcos2, double *dtau_og,
                                      double *tau_og, double *w0_og, double *cosb_og, double *w0_no_raman,
                                      double *f_deltaM)
{
    if (!ctx) return fail(nullptr, "null context");
    if (nlayer < 1 || nwno < 1 || nfacets < 1) return fail(ctx, "compute_opacity: bad sizes");
    if (test_mode < 0 || test_mode > 2) return fail(ctx, "compute_opacity: test_mode must be 0, 1 or 2");
    if (stream != 2 && stream != 4) return fail(ctx, "compute_opacity: stream must be 2 or 4");
    PZ_HIP(ctx, hipSetDevice(ctx->device));
    MixArgs a{};
    a.nlayer = nlayer; a.nwno = nwno; a.ncolper = 1; a.nfac = nfacets > 1 ? nfacets : 0;
    a.test_mode = test_mode; a.delta_eddington = delta_eddington;
    a.stream = stream; a.taugas = taugas; a.tauray = tauray; a.taucld = taucld; a.w0c = w0_cld;
    a.g0c = g0_cld; a.raman = raman_factor; a.raman_const = raman_const;
    if (raman_factor && raman_rows != 0 && raman_rows != nlayer)
        return fail(ctx, "compute_opacity: raman_rows must be nlayer (planes) or 0 (one row for all layers)");
    a.raman_row = raman_factor && raman_rows == 0;
    a.dtau = dtau; a.tau = tau; a.w0 = w0; a.cosb = cosb; a.ftau_cld = ftau_cld; a.ftau_ray = ftau_ray;
    a.gcos2 = gcos2; a.dtau_og = dtau_og; a.tau_og = tau_og; a.w0_og = w0_og; a.cosb_og = cosb_og;
    a.w0_no_raman = w0_no_raman; a.f_deltaM = f_deltaM;
    const int block = MIXF_BLOCK;
    const long ncol = (long)nwno * nfacets;
    const dim3 grid((unsigned)((ncol + block - 1) / block));
    // LDS-staged transposition while a block's [facet][wavelength] tile fits (<= 341 facets); else (and for a
    // single facet, where there is nothing to transpose) the direct kernel
    const bool staged = nfacets > 1 && (long)nfacets * ((block / nfacets + 2) | 1) <= MIXF_TILE &&
                        !getenv("PICASO_AMD_MIX_DIRECT");
    if (staged)
        hipLaunchKernelGGL(k_compute_opacity_facets, grid, dim3(block), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_compute_opacity, grid, dim3(block), 0, ctx->stream, a);
    PZ_HIP(ctx, hipGetLastError());
    return 0;
}

int picaso_compute_opacity_dev(picaso_ctx *ctx, int nlayer, int nwno, const double *taugas,
                               const double *tauray, const double *taucld, const double *w0_cld,
                               const double *g0_cld, const double *raman_factor,
                               double raman_const, int test_mode, int delta_eddington, int stream,
                               double *dtau, double *tau, double *w0, double *cosb,
                               double *ftau_cld, double *ftau_ray, double *gcos2, double *dtau_og,
                               double *tau_og, double *w0_og, double *cosb_og, double *w0_no_raman,
                               double *f_deltaM)
{
    return picaso_compute_opacity_ck_dev(ctx, nlayer, nwno, 1, taugas, tauray, taucld, w0_cld, g0_cld,
                                         raman_factor, nlayer, raman_const, test_mode, delta_eddington, stream, dtau,
                                         tau, w0, cosb, ftau_cld, ftau_ray, gcos2, dtau_og, tau_og, w0_og,
                                         cosb_og, w0_no_raman, f_deltaM);
}

}  // extern "C"
