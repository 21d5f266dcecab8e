"""CPU oracle of the climate solver's radiative-transfer call -- TEST INFRASTRUCTURE.

Restates the reference's ``climate.get_fluxes`` (picaso/climate.py:1687-1952) on top of the C
oracle's ``get_reflected_1d`` / ``get_thermal_1d`` (``oracle/oracle.py``): the per-Gauss-point loop,
the patchy-cloud blend, the Gauss-weight / wavenumber sums in the reference's order.  Pinned against
``tests/golden/climate_fluxes.npz`` (outputs of the reference's own function).  Only ``tests/`` may
import this module.
"""
import numpy as np

from . import oracle as orc


def get_fluxes(Atmosphere, OpacityWEd, OpacityNoEd, ScatteringPhase, Disco, Opagrid, F0PI, reflected, thermal,
               do_holes=False, fhole=0.0, hole_OpacityWEd=None, hole_OpacityNoEd=None):
    pressure, temperature, nlevel = Atmosphere.p_level, Atmosphere.t_level, Atmosphere.nlevel
    W, N, sp = OpacityWEd, OpacityNoEd, ScatteringPhase
    ng, nt = Disco.ng, Disco.nt
    nwno, dwni, wno, ngauss, gauss_wts = (Opagrid.nwno, Opagrid.delta_wno, Opagrid.wno, Opagrid.ngauss,
                                          Opagrid.gauss_wts)
    flux_net_v = np.zeros((ng, nt, nlevel))
    flux_net_v_layer = np.zeros((ng, nt, nlevel))
    flux_plus_v = np.zeros((ng, nt, nlevel, nwno))
    flux_minus_v = np.zeros((ng, nt, nlevel, nwno))
    flux_plus_midpt = np.zeros((ng, nt, nlevel, nwno))
    flux_minus_midpt = np.zeros((ng, nt, nlevel, nwno))
    flux_plus = np.zeros((ng, nt, nlevel, nwno))
    flux_minus = np.zeros((ng, nt, nlevel, nwno))
    flux_net_ir = np.zeros(nlevel)
    flux_net_ir_layer = np.zeros(nlevel)
    flux_plus_ir = np.zeros((nlevel, nwno))
    flux_minus_ir = np.zeros((nlevel, nwno))

    def sl(a, ig):
        return np.ascontiguousarray(a[:, :, ig])

    def refl(W_, N_, ig):                                 # climate.py:1806-1814
        half = np.full((1, 1), 0.5)
        _, out = orc.get_reflected_1d(nlevel, wno, nwno, 1, 1, sl(W_.DTAU, ig), sl(W_.TAU, ig), sl(W_.W0, ig),
                                      sl(W_.COSB, ig), sl(W_.GCOS2, ig), sl(W_.ftau_cld, ig), sl(W_.ftau_ray, ig),
                                      sl(N_.DTAU, ig), sl(N_.TAU, ig), sl(N_.W0, ig), sl(N_.COSB, ig),
                                      sp.surf_reflect, half, half, Disco.cos_theta, F0PI, sp.single_phase,
                                      sp.multi_phase, sp.frac_a, sp.frac_b, sp.frac_c, sp.constant_back,
                                      sp.constant_forward, get_toa_intensity=0, get_lvl_flux=1)
        return out

    def therm(W_, N_, ig):                                # climate.py:1888-1892
        _, out = orc.get_thermal_1d(nlevel, wno, nwno, ng, nt, temperature, sl(N_.DTAU, ig),
                                    sl(W_.W0_no_raman, ig), sl(N_.COSB, ig), pressure, Disco.ubar1,
                                    sp.surf_reflect, 0, dwni, calc_type=1)
        return out

    def blend(cloudy, clear):
        return [(1.0 - fhole) * a + fhole * b for a, b in zip(cloudy, clear)]

    if reflected:
        for ig in range(ngauss):
            out = refl(W, N, ig)
            if do_holes:
                out = blend(out, refl(hole_OpacityWEd, hole_OpacityNoEd, ig))
            fm, fp, fmm, fpm = out
            flux_net_v_layer += (np.sum(fpm, axis=3) - np.sum(fmm, axis=3)) * gauss_wts[ig]
            flux_net_v += (np.sum(fp, axis=3) - np.sum(fm, axis=3)) * gauss_wts[ig]
            flux_plus_v += fp * gauss_wts[ig]
            flux_minus_v += fm * gauss_wts[ig]
    if thermal:
        for ig in range(ngauss):
            out = therm(W, N, ig)
            if do_holes:
                out = blend(out, therm(hole_OpacityWEd, hole_OpacityNoEd, ig))
            fm, fp, fmm, fpm = out
            flux_plus += fp * gauss_wts[ig]
            flux_minus += fm * gauss_wts[ig]
            flux_plus_midpt += fpm * gauss_wts[ig]
            flux_minus_midpt += fmm * gauss_wts[ig]
        fp2 = orc.compress_thermal(nwno, flux_plus, Disco.gweight, Disco.tweight)
        fm2 = orc.compress_thermal(nwno, flux_minus, Disco.gweight, Disco.tweight)
        fpm2 = orc.compress_thermal(nwno, flux_plus_midpt, Disco.gweight, Disco.tweight)
        fmm2 = orc.compress_thermal(nwno, flux_minus_midpt, Disco.gweight, Disco.tweight)
        for wvi in range(nwno):                           # climate.py:1931-1936
            flux_net_ir_layer += (fpm2[:, wvi] - fmm2[:, wvi]) * dwni[wvi]
            flux_net_ir += (fp2[:, wvi] - fm2[:, wvi]) * dwni[wvi]
            flux_plus_ir[:, wvi] += fp2[:, wvi] * dwni[wvi]
            flux_minus_ir[:, wvi] += fm2[:, wvi] * dwni[wvi]
    return (flux_net_v_layer, flux_net_v, flux_plus_v, flux_minus_v, flux_net_ir_layer, flux_net_ir,
            flux_plus_ir, flux_minus_ir)
