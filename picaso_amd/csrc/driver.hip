// One C call per spectrum: the launches of picaso()'s 1-D Toon path for every wavelength block of the spectrum.
//
// The reference's driver (justdoit.py:236-385) calls get_opacities -> compute_opacity -> get_reflected_1d /
// get_thermal_1d -> compress_disco / compress_thermal once per spectrum; its multi-process fan-out (justdoit.py:4774)
// runs whole spectra side by side.  Here ONE spectrum may be cut into wavelength blocks, one per GPU,
// and the Python mirror paid ~0.2 ms of interpreter time per block for a dozen ctypes calls whose arguments differ
// only in pointers -- more than the 0.1 ms of GPU work of a 12 500-column block.  picaso_toon_spectrum_blocks takes the
// per-block pointers once (a table the caller builds per opacity object and keeps) and the per-layer tables of THIS
// call (table rows and weights, mixing coefficients, level temperatures: shared by all blocks) and enqueues, per block,
//     picaso_opacity_gas_ck_dev -> picaso_compute_opacity_ck_dev -> picaso_get_reflected_1d_dev (+ fused disk sum)
//                                                                 || picaso_get_thermal_1d_dev  (+ fused disk sum)
// on the block's own context(s); picaso_toon_spectrum_collect then copies one leg's results of every block into the
// caller's full-grid host arrays.  Nothing here computes: it is the same entry points in the same order, so the
// results are the Python path's, bit for bit (tests/test_devices_gpu.py, tests/test_driver_gpu.py).
#include "common.hpp"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace pz;

// The opacity stage of one block: cloud planes (device, host columns, or tables interpolated in the launch), gas stage +
// mixing as one launch, the level planes as a second one that the thermal leg does not wait for (it reads layer planes
// only: `thermal_waits_here` orders its stream in between).  PICASO_AMD_UNFUSED_OPACITY=1: the two launches with
// TAUGAS / TAURAY in HBM (A/B).
static int opacity_stage(const picaso_block &k, const picaso_spectrum_job &j, int b, picaso_ctx *thermal_waits_here)
{
    // cloud tables handed over as full-grid host planes: this block's columns, strided copy (the reference slices
    // nothing: it has one grid; justdoit.py:4774 fans out whole spectra)
    const double *cld[3] = {k.cld_opd, k.cld_w0, k.cld_g0};
    if (k.cld_host_opd) {
        const double *src[3] = {k.cld_host_opd, k.cld_host_w0, k.cld_host_g0};
        double *dst[3] = {k.cld_work_opd, k.cld_work_w0, k.cld_work_g0};
        for (int c = 0; c < 3; ++c) {
            if (!src[c] || !dst[c]) return fail(k.ctx, "toon_spectrum_blocks: block %d: incomplete host cloud planes", b);
            PZ_TRY(picaso_memcpy_h2d_2d(k.ctx, dst[c], sizeof(double) * (size_t)k.nwno, src[c] + k.col0,
                                        sizeof(double) * (size_t)k.cld_host_pitch, sizeof(double) * (size_t)k.nwno,
                                        (size_t)j.nlayer));
            cld[c] = dst[c];
        }
    }
    double *const *o = k.planes;
    const int ngs = j.ngauss > 1 ? j.ngauss : 1;          // correlated-k tables: the two launches, Gauss axis in the columns
    const bool fused = ngs == 1 && (!o[1] || o[0]) && (!o[8] || o[7]) && !getenv("PICASO_AMD_UNFUSED_OPACITY");
    if (fused) {
        PZ_TRY(picaso_gas_compute_opacity_dev(k.ctx, j.nlayer, k.nwno, j.mol_mode, j.nmol, k.mol_tabs, j.mol_rows,
                                              j.mol_wts, j.mol_fac, j.cont_interp, j.ncont, k.cont_tabs, j.cont_rows,
                                              j.cont_wts, j.cont_fac, j.nray, k.ray_tabs, j.ray_fac, cld[0], cld[1],
                                              cld[2], k.raman, j.raman_rows, j.raman_const, j.test_mode,
                                              j.delta_eddington, j.stream, o[0], o[1], o[2], o[3], o[4], o[5], o[6],
                                              o[7], o[8], o[9], o[10], o[11], o[12], 0, k.cld_tab_nin, k.cld_tab_xp,
                                              k.cld_tab_fp, k.cld_tab_nin ? k.wno : nullptr));
        if (thermal_waits_here) PZ_TRY(picaso_ctx_wait(thermal_waits_here, k.ctx));
        PZ_TRY(picaso_level_sums_dev(k.ctx, j.nlayer, k.nwno, o[0], o[1], o[7], o[8]));
    } else {
        if (k.cld_tab_nin) return fail(k.ctx, "toon_spectrum_blocks: cloud tables need the fused opacity launch");
        if (!k.taugas || !k.tauray) return fail(k.ctx, "toon_spectrum_blocks: block %d has no TAUGAS / TAURAY workspace", b);
        PZ_TRY(picaso_opacity_gas_ck_dev(k.ctx, j.nlayer, k.nwno, ngs, j.mol_mode, j.nmol, k.mol_tabs, j.mol_rows,
                                         j.mol_wts, j.mol_fac, j.cont_interp, j.ncont, k.cont_tabs, j.cont_rows,
                                         j.cont_wts, j.cont_fac, j.nray, k.ray_tabs, j.ray_fac, k.taugas, k.tauray));
        PZ_TRY(picaso_compute_opacity_ck_dev(k.ctx, j.nlayer, k.nwno, ngs, k.taugas, k.tauray, cld[0], cld[1], cld[2],
                                             k.raman, j.raman_rows, j.raman_const, j.test_mode, j.delta_eddington,
                                             j.stream, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10],
                                             o[11], o[12]));
        if (thermal_waits_here) PZ_TRY(picaso_ctx_wait(thermal_waits_here, k.ctx));
    }
    return 0;
}

// behind a leg's disk sum: the spectrum-wide integral (one block over the grid) and the result copy on the leg's stream
static int reflected_tail(int nblocks, picaso_block &k)
{
    if (k.trapz_d) {
        if (nblocks != 1) return fail(k.ctx, "toon_spectrum_blocks: device integrals need ONE block over the grid");
        PZ_TRY(picaso_trapz_dev(k.ctx, k.nwno, k.trapz_d, k.albedo, k.stellar, 0, k.albedo + k.nwno));
    }
    if (k.albedo_pin && k.albedo_host)
        PZ_TRY(picaso_memcpy_d2h_async(k.ctx, k.albedo_pin, k.albedo,
                                       sizeof(double) * (size_t)(k.nwno + (k.trapz_d ? 1 : 0)), &k.albedo_mark));
    return 0;
}

static int thermal_tail(int nblocks, picaso_block &k, picaso_ctx *tctx)
{
    if (k.trapz_dr) {
        if (nblocks != 1) return fail(k.ctx, "toon_spectrum_blocks: device integrals need ONE block over the grid");
        PZ_TRY(picaso_trapz_dev(tctx, k.nwno, k.trapz_dr, k.disk, nullptr, 1, k.disk + k.nwno));
    }
    if (k.thermal_pin && k.thermal_host)
        PZ_TRY(picaso_memcpy_d2h_async(tctx, k.thermal_pin, k.disk,
                                       sizeof(double) * (size_t)(k.nwno + (k.trapz_dr ? 1 : 0)), &k.thermal_mark));
    return 0;
}

// One block of a 3-D spectrum (picaso_spectrum_job::nfacets; reference justdoit.py:407-516): ONE fused gas + mixing launch
// over the tall atmosphere of all facets writes facet-major planes, every facet goes into the batched 3-D solver launch as
// a spectrum of its own with one facet (a wave holds 64 wavelengths of one facet), then the disk sums -- the sequence of
// spectrum.Spectrum._plan_3d / _reflected_3d_fm / resident.thermal_3d_fm_batch, with the same entry points.
static int enqueue_block_3d(int nblocks, picaso_block *blocks, const picaso_spectrum_job &j, int b)
{
    picaso_block &k = blocks[b];
    const int nfac = j.nfacets, nlevel = j.nlayer + 1;
    if (!k.ctx || k.nwno < 1) return fail(k.ctx, "toon_spectrum_blocks: block %d has no context or no columns", b);
    if (k.albedo_mark || k.thermal_mark)
        return fail(k.ctx, "toon_spectrum_blocks: block %d still has uncollected results", b);
    if (nfac != j.numg * j.numt) return fail(k.ctx, "toon_spectrum_blocks: nfacets must be numg * numt");
    if (j.rt_method != 0) return fail(k.ctx, "toon_spectrum_blocks: 3-D blocks are Toon only (the reference has no 3-D SH)");
    if (j.ngauss > 1) return fail(k.ctx, "toon_spectrum_blocks: 3-D blocks take monochromatic tables");
    if ((long)nfac * j.nlayer > 2147483647L / 4) return fail(k.ctx, "toon_spectrum_blocks: too many facet-layers");
    double *const *o = k.planes;
    if (o[1] || o[8])
        return fail(k.ctx, "toon_spectrum_blocks: 3-D blocks leave tau / tau_og out (running sums down ONE facet's layers)");
    if (k.cld_opd || k.cld_host_opd)
        return fail(k.ctx, "toon_spectrum_blocks: 3-D blocks take cloud tables on their own grid (cld_tab_*) or none");
    if (j.do_reflected && (!k.refl_planes[0] || !k.refl_planes[2] || !k.xint || !k.albedo || !k.surf_reflect || !k.F0PI))
        return fail(k.ctx, "toon_spectrum_blocks: block %d: the reflected leg needs dtau, w0, xint, albedo, surf_reflect, F0PI", b);
    if (j.do_thermal && (!k.th_dtau || !k.th_w0 || !k.flux || !k.disk || !k.wno || !k.surf_reflect || !j.tlevel || !j.plevel))
        return fail(k.ctx, "toon_spectrum_blocks: block %d: the thermal leg needs dtau, w0, flux, disk, wno, surf_reflect and "
                           "the level tables", b);
    picaso_ctx *tctx = k.tctx ? k.tctx : k.ctx;
    PZ_TRY(picaso_gas_compute_opacity_dev(k.ctx, nfac * j.nlayer, k.nwno, j.mol_mode, j.nmol, k.mol_tabs, j.mol_rows,
                                          j.mol_wts, j.mol_fac, j.cont_interp, j.ncont, k.cont_tabs, j.cont_rows,
                                          j.cont_wts, j.cont_fac, j.nray, k.ray_tabs, j.ray_fac, nullptr, nullptr, nullptr,
                                          k.raman, j.raman_rows, j.raman_const, j.test_mode, j.delta_eddington, j.stream,
                                          o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12], 0,
                                          k.cld_tab_nin, k.cld_tab_xp, k.cld_tab_fp, k.cld_tab_nin ? k.wno : nullptr));
    if (tctx != k.ctx && j.do_thermal) PZ_TRY(picaso_ctx_wait(tctx, k.ctx));
    const size_t slab = (size_t)j.nlayer * (size_t)k.nwno;
    auto slabs = [&](const double *base) {
        std::vector<const double *> v((size_t)nfac);
        for (int f = 0; f < nfac; ++f) v[(size_t)f] = base + (size_t)f * slab;
        return v;
    };
    const std::vector<const double *> rs((size_t)nfac, k.surf_reflect);
    if (j.do_reflected) {
        std::vector<const double *> cols[11];
        const double *const *colp[11];
        for (int p = 0; p < 11; ++p) {
            if (k.refl_planes[p] && (p == 1 || p == 8))
                return fail(k.ctx, "toon_spectrum_blocks: 3-D blocks leave tau / tau_og out");
            if (k.refl_planes[p]) cols[p] = slabs(k.refl_planes[p]);
            colp[p] = k.refl_planes[p] ? cols[p].data() : nullptr;
        }
        const std::vector<const double *> f0((size_t)nfac, k.F0PI);
        const std::vector<double> ct((size_t)nfac, j.cos_theta);
        std::vector<double *> xs((size_t)nfac);
        for (int f = 0; f < nfac; ++f) xs[(size_t)f] = k.xint + (size_t)f * (size_t)k.nwno;
        PZ_TRY(picaso_get_reflected_3d_batch_dev(k.ctx, nfac, nlevel, k.nwno, 1, 1, colp[0], colp[1], colp[2], colp[3],
                                                 colp[4], colp[5], colp[6], colp[7], colp[8], colp[9], colp[10], rs.data(),
                                                 j.ubar0, j.ubar1, ct.data(), f0.data(), j.single_phase, j.multi_phase,
                                                 j.frac_a, j.frac_b, j.frac_c, j.constant_back, j.constant_forward,
                                                 xs.data(), nullptr, nullptr, nullptr));
        PZ_TRY(picaso_compress_disco_dev(k.ctx, k.nwno, j.cos_theta, k.xint, j.gweight, j.numg, j.tweight, j.numt, k.F0PI,
                                         k.albedo));
        PZ_TRY(reflected_tail(nblocks, k));
    }
    if (j.do_thermal) {
        const std::vector<const double *> dt = slabs(k.th_dtau), w0 = slabs(k.th_w0);
        std::vector<const double *> cb;
        if (k.th_cosb) cb = slabs(k.th_cosb);
        std::vector<double *> fx((size_t)nfac);
        for (int f = 0; f < nfac; ++f) fx[(size_t)f] = k.flux + (size_t)f * (size_t)k.nwno;
        PZ_TRY(picaso_get_thermal_3d_batch_dev(tctx, nfac, nlevel, k.wno, k.nwno, 1, 1, j.tlevel, dt.data(), w0.data(),
                                               k.th_cosb ? cb.data() : nullptr, j.plevel, j.ubar1, rs.data(),
                                               j.hard_surface, fx.data(), nullptr, nullptr, nullptr));
        PZ_TRY(picaso_compress_thermal_dev(tctx, (size_t)k.nwno, k.flux, j.gweight, j.numg, j.tweight, j.numt, k.disk));
        PZ_TRY(thermal_tail(nblocks, k, tctx));
    }
    return 0;
}

// one block: opacity stage -> reflected || thermal (+ integrals, result copies) on the block's own context(s)
// phase 0: the whole block.  1: the opacity stage alone (picaso_toon_spectrum_phase: the caller has filled the opacity half
// of the job and of the block and fills the rest while the gas kernel runs).  2: everything behind the opacity stage.
static int enqueue_block(int nblocks, picaso_block *blocks, const picaso_spectrum_job &j, int b, int phase = 0)
{
    if (j.nfacets > 0) {
        if (phase) return fail(blocks[b].ctx, "toon_spectrum_phase: 3-D blocks are enqueued in one piece");
        return enqueue_block_3d(nblocks, blocks, j, b);
    }
    const int nlevel = j.nlayer + 1;
    {
        picaso_block &k = blocks[b];
        if (!k.ctx || k.nwno < 1) return fail(k.ctx, "toon_spectrum_blocks: block %d has no context or no columns", b);
        if (k.albedo_mark || k.thermal_mark)
            return fail(k.ctx, "toon_spectrum_blocks: block %d still has uncollected results", b);
        picaso_ctx *tctx = k.tctx ? k.tctx : k.ctx;
        if (phase != 2) PZ_TRY(opacity_stage(k, j, b, (tctx != k.ctx && j.do_thermal) ? tctx : nullptr));
        if (phase == 1) return 0;
        if (j.rt_method == 1 && j.ngauss > 1)
            return fail(k.ctx, "toon_spectrum_blocks: correlated-k blocks are Toon only");
        if (j.do_reflected && j.rt_method == 1) {
            const double *const *r = k.refl_planes;             // SH argument order (picaso_spectrum_job::rt_method)
            PZ_TRY(picaso_get_reflected_SH_top_dev(k.ctx, nlevel, k.nwno, k.nwno, j.numg, j.numt, r[0], r[1], r[2], r[3], r[4],
                                                   r[5], r[6], r[7], r[8], r[9], r[10], k.surf_reflect, j.ubar0, j.ubar1,
                                                   j.cos_theta, k.F0PI, j.sh_w_single_form, j.sh_w_multi_form,
                                                   j.sh_psingle_form, j.sh_w_single_rayleigh, j.sh_w_multi_rayleigh,
                                                   j.sh_psingle_rayleigh, j.frac_a, j.frac_b, j.frac_c, j.constant_back,
                                                   j.constant_forward, j.stream, j.b_top, 0, j.sh_single_form, 1,
                                                   j.sh_cloud_free_above, k.xint, nullptr, j.gweight, j.tweight, k.albedo));
        } else if (j.do_reflected && j.ngauss > 1) {
            const double *const *r = k.refl_planes;             // (rows, nwno, ngauss): the Gauss loop inside the call
            if (!j.gauss_wts) return fail(k.ctx, "toon_spectrum_blocks: ngauss > 1 without gauss_wts");
            PZ_TRY(picaso_get_reflected_1d_ck_dev(k.ctx, nlevel, k.nwno, j.ngauss, j.numg, j.numt, r[0], r[1], r[2], r[3], r[4],
                                                  r[5], r[6], r[7], r[8], r[9], r[10], k.surf_reflect, j.ubar0, j.ubar1,
                                                  j.cos_theta, k.F0PI, j.single_phase, j.multi_phase, j.frac_a, j.frac_b,
                                                  j.frac_c, j.constant_back, j.constant_forward, 1, 0, j.toon_coefficients,
                                                  j.b_top, j.gauss_wts, k.xint, nullptr, nullptr, nullptr, nullptr, j.gweight,
                                                  j.tweight, k.albedo));
        } else if (j.do_reflected) {
            const double *const *r = k.refl_planes;
            PZ_TRY(picaso_get_reflected_1d_dev(k.ctx, nlevel, k.nwno, k.nwno, j.numg, j.numt, r[0], r[1], r[2], r[3], r[4],
                                               r[5], r[6], r[7], r[8], r[9], r[10], k.surf_reflect, j.ubar0, j.ubar1,
                                               j.cos_theta, k.F0PI, j.single_phase, j.multi_phase, j.frac_a, j.frac_b,
                                               j.frac_c, j.constant_back, j.constant_forward, 1, 0, j.toon_coefficients,
                                               j.b_top, k.xint, nullptr, nullptr, nullptr, nullptr, j.gweight, j.tweight,
                                               k.albedo));
        }
        if (j.do_reflected) PZ_TRY(reflected_tail(nblocks, k));
        if (j.do_thermal) {
            if (j.rt_method == 1)
                // ff = 0 if np.array_equal(cosb, cosb_og) else cosb_og**stream (fluxes.py:3072-3075): with delta-Eddington
                // scaling the two planes differ wherever cosb_og**stream does not vanish (spectrum._thermal_sh)
                PZ_TRY(picaso_get_thermal_SH_dev(tctx, nlevel, k.wno, k.nwno, k.nwno, j.numg, j.numt, j.tlevel, k.th_dtau,
                                                 nullptr, k.th_w0, k.th_cosb, j.plevel, j.ubar1, k.surf_reflect, j.stream,
                                                 j.hard_surface, j.delta_eddington, 0, k.flux, j.gweight, j.tweight, k.disk));
            else if (j.ngauss > 1) {
                if (!j.gauss_wts) return fail(k.ctx, "toon_spectrum_blocks: ngauss > 1 without gauss_wts");
                PZ_TRY(picaso_get_thermal_1d_ck_dev(tctx, nlevel, k.wno, k.nwno, j.ngauss, j.numg, j.numt, j.tlevel, k.th_dtau,
                                                    k.th_w0, k.th_cosb, j.plevel, j.ubar1, k.surf_reflect, j.hard_surface,
                                                    nullptr, 0, j.gauss_wts, k.flux, nullptr, nullptr, nullptr, nullptr,
                                                    j.gweight, j.tweight, k.disk));
            } else
            PZ_TRY(picaso_get_thermal_1d_dev(tctx, nlevel, k.wno, k.nwno, k.nwno, j.numg, j.numt, j.tlevel, k.th_dtau,
                                             k.th_w0, k.th_cosb, j.plevel, j.ubar1, k.surf_reflect, j.hard_surface,
                                             nullptr, 0, k.flux, nullptr, nullptr, nullptr, nullptr, j.gweight, j.tweight,
                                             k.disk));
            PZ_TRY(thermal_tail(nblocks, k, tctx));
        }
    }
    return 0;
}

// The blocks of a spectrum are independent (own contexts, streams, table rings; the job is read-only), and a block costs
// ~0.13 ms of HIP API calls to enqueue -- more than its 0.1 ms of GPU work at 12 500 columns.  Blocks on DIFFERENT devices
// are therefore enqueued from a thread each (the reference's fan-out runs whole spectra in separate processes,
// justdoit.py:4774).  Blocks on ONE device keep the serial loop: the runtime serialises the calls of a device anyway
// (measured on this pool's single GPU, eight blocks: 1.34 ms threaded, 1.22 serial) -- which also means the threaded form
// has not met two real GPUs here.  PICASO_AMD_PARALLEL_BLOCKS=1 forces threads (the tests do: same bits), =0 forbids them.
extern "C" int picaso_toon_spectrum_blocks(int nblocks, picaso_block *blocks, const picaso_spectrum_job *job)
{
    if (nblocks < 1 || !blocks || !job) return fail(nullptr, "toon_spectrum_blocks: null argument");
    const picaso_spectrum_job &j = *job;
    if (j.nlayer < 1 || j.numg < 1 || j.numt < 1) return fail(blocks[0].ctx, "toon_spectrum_blocks: bad sizes");
    const char *force = getenv("PICASO_AMD_PARALLEL_BLOCKS");
    bool parallel = nblocks > 1 && !(force && force[0] == '0');
    bool one_device = false;
    for (int a = 0; a < nblocks && parallel; ++a)
        for (int b = a + 1; b < nblocks && parallel; ++b) {
            const picaso_ctx *ca[2] = {blocks[a].ctx, blocks[a].tctx ? blocks[a].tctx : blocks[a].ctx};
            const picaso_ctx *cb[2] = {blocks[b].ctx, blocks[b].tctx ? blocks[b].tctx : blocks[b].ctx};
            if (!ca[0] || !cb[0] || ca[0] == cb[0] || ca[0] == cb[1] || ca[1] == cb[0] || ca[1] == cb[1]) parallel = false;
            else if (ca[0]->device == cb[0]->device) one_device = true;
        }
    if (one_device && !(force && force[0] == '1')) parallel = false;
    // A failing block leaves its message on its own context and in the thread-local buffer of the thread that ran it.
    // The caller asks picaso_last_error(blocks[0].ctx) or picaso_last_error(NULL) from ITS thread: the text is carried
    // over to both.
    if (!parallel) {
        for (int b = 0; b < nblocks; ++b)
            if (int rc = enqueue_block(nblocks, blocks, j, b)) {
                char msg[sizeof(g_err)];
                snprintf(msg, sizeof(msg), "%s", g_err);
                fail(blocks[0].ctx, "%s", msg);
                return rc;
            }
        return 0;
    }
    std::vector<int> rc((size_t)nblocks, 0);
    std::vector<std::string> msgs((size_t)nblocks);
    std::vector<std::thread> workers;
    workers.reserve((size_t)nblocks - 1);
    for (int b = 1; b < nblocks; ++b)
        workers.emplace_back([&, b] {
            rc[(size_t)b] = enqueue_block(nblocks, blocks, j, b);
            if (rc[(size_t)b]) msgs[(size_t)b] = g_err;          // this worker's thread-local text
        });
    rc[0] = enqueue_block(nblocks, blocks, j, 0);
    if (rc[0]) msgs[0] = g_err;
    for (auto &t : workers) t.join();
    for (int b = 0; b < nblocks; ++b)
        if (rc[(size_t)b]) {
            fail(blocks[0].ctx, "%s", msgs[(size_t)b].c_str());
            return rc[(size_t)b];
        }
    return 0;
}

// The same sequence in two calls (1-D blocks): phase 1 enqueues the opacity stage of every block and returns -- it reads
// the opacity half of the job (table rows / weights / coefficients, Raman, delta-Eddington, stream, do_thermal) and of the
// blocks (tables, planes, cloud inputs, raman, ctx / tctx) and nothing else -- phase 2 the legs, integrals and result copies.
// What the caller has to prepare for the legs (geometry, level tables, resident vectors, result buffers) is prepared
// while the gas kernel runs instead of in front of it; same launches in the same order on the same streams, same bits.
extern "C" int picaso_toon_spectrum_phase(int nblocks, picaso_block *blocks, const picaso_spectrum_job *job, int phase)
{
    if (nblocks < 1 || !blocks || !job) return fail(nullptr, "toon_spectrum_phase: null argument");
    if (phase != 1 && phase != 2) return fail(blocks[0].ctx, "toon_spectrum_phase: phase must be 1 or 2, got %d", phase);
    const picaso_spectrum_job &j = *job;
    if (j.nlayer < 1 || (phase == 2 && (j.numg < 1 || j.numt < 1))) return fail(blocks[0].ctx, "toon_spectrum_phase: bad sizes");
    for (int b = 0; b < nblocks; ++b)
        if (int rc = enqueue_block(nblocks, blocks, j, b, phase)) {
            char msg[sizeof(g_err)];
            snprintf(msg, sizeof(msg), "%s", g_err);
            fail(blocks[0].ctx, "%s", msg);
            return rc;
        }
    return 0;
}

// Copy one leg's results (which = 1: albedo, 2: thermal flux) of every block into the caller's full-grid host arrays
// at [col0, col0 + nwno).  Each copy waits for its own stream only; the other blocks and the other leg keep running --
// the caller integrates the albedo while the thermal kernels finish.
extern "C" int picaso_toon_spectrum_collect(int nblocks, picaso_block *blocks, int which)
{
    if (nblocks < 1 || !blocks) return fail(nullptr, "toon_spectrum_collect: null argument");
    for (int b = 0; b < nblocks; ++b) {
        picaso_block &k = blocks[b];
        picaso_ctx *tctx = k.tctx ? k.tctx : k.ctx;
        if (which == 1) {
            if (!k.albedo_host) return fail(k.ctx, "toon_spectrum_collect: block %d has no host albedo array", b);
            const size_t bytes = sizeof(double) * (size_t)(k.nwno + (k.trapz_d ? 1 : 0));
            if (k.albedo_mark) {
                void *mark = k.albedo_mark;
                k.albedo_mark = nullptr;
                PZ_TRY(picaso_mark_wait(k.ctx, mark));
                memcpy(k.albedo_host + k.col0, k.albedo_pin, bytes);
            } else {
                PZ_TRY(picaso_memcpy_d2h(k.ctx, k.albedo_host + k.col0, k.albedo, bytes));
            }
        } else if (which == 2) {
            if (!k.thermal_host) return fail(k.ctx, "toon_spectrum_collect: block %d has no host thermal array", b);
            const size_t bytes = sizeof(double) * (size_t)(k.nwno + (k.trapz_dr ? 1 : 0));
            if (k.thermal_mark) {
                void *mark = k.thermal_mark;
                k.thermal_mark = nullptr;
                PZ_TRY(picaso_mark_wait(tctx, mark));
                memcpy(k.thermal_host + k.col0, k.thermal_pin, bytes);
            } else {
                PZ_TRY(picaso_memcpy_d2h(tctx, k.thermal_host + k.col0, k.disk, bytes));
            }
            // a second stream read the planes of ctx: ctx's next call (which overwrites them) starts behind it
            if (tctx != k.ctx) PZ_TRY(picaso_ctx_wait(k.ctx, tctx));
        } else {
            return fail(k.ctx, "toon_spectrum_collect: which must be 1 (albedo) or 2 (thermal)");
        }
    }
    return 0;
}

// Results nobody will collect (the caller failed between the two calls): wait for the copies in flight and clear their
// marks, so that the blocks can take the next spectrum.
extern "C" int picaso_toon_spectrum_abandon(int nblocks, picaso_block *blocks)
{
    if (nblocks < 1 || !blocks) return fail(nullptr, "toon_spectrum_abandon: null argument");
    int rc = 0;
    for (int b = 0; b < nblocks; ++b) {
        picaso_block &k = blocks[b];
        picaso_ctx *tctx = k.tctx ? k.tctx : k.ctx;
        if (k.albedo_mark) { if (picaso_mark_wait(k.ctx, k.albedo_mark)) rc = 1; k.albedo_mark = nullptr; }
        if (k.thermal_mark) { if (picaso_mark_wait(tctx, k.thermal_mark)) rc = 1; k.thermal_mark = nullptr; }
    }
    return rc;
}

// sizeof / offsets of the two structs above as this library was compiled, so that a binding (picaso_amd/driver.py's
// ctypes.Structure definitions) can check its layout without a GPU
extern "C" int picaso_driver_abi(size_t *block_bytes, size_t *job_bytes, size_t *off_albedo_host, size_t *off_hard_surface)
{
    if (block_bytes) *block_bytes = sizeof(picaso_block);
    if (job_bytes) *job_bytes = sizeof(picaso_spectrum_job);
    if (off_albedo_host) *off_albedo_host = offsetof(picaso_block, albedo_host);
    if (off_hard_surface) *off_hard_surface = offsetof(picaso_spectrum_job, hard_surface);
    return 0;
}
