/*
 * picaso_hip.h -- C ABI of the MI355X (gfx950) implementation of PICASO's per-wavelength
 * radiative-transfer hot path.
 *
 * The reference (natashabatalha/picaso v4.0.1) is pure Python + numba; it has no FFI.  Its
 * "operator interface" for this path is the set of Python function signatures that
 * justdoit.picaso() calls (reference picaso/justdoit.py:243,260,275,337,365,492,510,530,567).
 * Each entry point below replaces one of those functions and keeps its argument order and
 * meaning; the ctypes binding that a maintainer adds on the reference side is shown in
 * INTEGRATION.md and implemented in picaso_amd/_lib.py.
 *
 * Conventions
 *  - All arrays are float64, C-contiguous, in the reference's own layout: planes are
 *    (nlayer|nlevel, nwno) layer-major / wavelength-contiguous for 1-D and
 *    (nlayer|nlevel, nwno, numg, numt) for 3-D (reference fluxes.py:1032-1047, :355-358).
 *  - `surf_reflect` and `F0PI` are always (nwno) arrays (the Python shim broadcasts scalars).
 *  - Functions without a suffix take HOST pointers, run synchronously (H2D, kernel, D2H) and own
 *    no caller memory.  Functions ending in `_dev` take DEVICE pointers for every plane /
 *    per-wavelength vector / output (geometry tables ubar0, ubar1, gweight, tweight, tlevel,
 *    plevel stay host pointers: they are tiny), enqueue on the context's stream and return
 *    without synchronising; `plane_pitch` is the element stride between consecutive layers
 *    (== nwno for a dense plane, larger for a wavelength-shard view of a bigger plane).
 *  - Return value 0 = ok; non-zero = error, message via picaso_last_error().
 *  - NaN/inf propagate as in the reference (no clamping beyond the reference's own clips).
 *  - One context per process and GPU; calls on one context are serialised by the caller.
 */
#ifndef PICASO_HIP_H
#define PICASO_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct picaso_ctx picaso_ctx;

/* ---- context / plumbing ------------------------------------------------------------------
 * No counterpart in the reference (it has no device, no FFI and no explicit memory): contexts,
 * device buffers, copies and stream timing for callers that keep planes resident in HBM. */
int picaso_device_count(int *count);
int picaso_ctx_create(int device, picaso_ctx **out);
void picaso_ctx_destroy(picaso_ctx *ctx);
const char *picaso_last_error(const picaso_ctx *ctx); /* ctx may be NULL: last global error */
const char *picaso_version(void);

int picaso_dev_malloc(picaso_ctx *ctx, size_t bytes, void **dptr);
int picaso_dev_free(picaso_ctx *ctx, void *dptr);
/* picaso_dev_free keeps blocks for reuse by later picaso_dev_malloc calls of the same size (reuse is
 * ordered on the context's stream); this returns all cached blocks to the driver. */
int picaso_pool_trim(picaso_ctx *ctx);
/* what the context holds right now: out[0..5] = bytes and blocks handed out by picaso_dev_malloc (live), bytes and blocks
 * cached for reuse (free, capped at 64 GB; picaso_pool_trim returns them), bytes and blocks of pinned host memory
 * (handed out + cached).  For a long-running caller (a retrieval) that wants to see a leak as a number. */
int picaso_ctx_mem_stats(picaso_ctx *ctx, size_t *out6);
int picaso_memcpy_h2d(picaso_ctx *ctx, void *dst, const void *src, size_t bytes);
int picaso_memcpy_d2h(picaso_ctx *ctx, void *dst, const void *src, size_t bytes);
int picaso_memcpy_d2d(picaso_ctx *ctx, void *dst, const void *src, size_t bytes);
/* Result copies that do not wait (the retrieval loop of driver.py:405-426 pipelined: the host sets up the next
 * spectra while the GPU solves the last ones).  picaso_host_alloc: a pinned host block, kept for reuse after
 * picaso_host_free.  picaso_memcpy_d2h_async: the copy is enqueued behind the kernels already on the context's
 * stream and *mark identifies it; picaso_mark_wait blocks until that copy has landed -- work enqueued on the stream
 * after it does not delay the wait -- and consumes the mark (wait for every mark exactly once). */
int picaso_host_alloc(picaso_ctx *ctx, size_t bytes, void **hptr);
int picaso_host_free(picaso_ctx *ctx, void *hptr);
int picaso_memcpy_d2h_async(picaso_ctx *ctx, void *pinned_dst, const void *src, size_t bytes, void **mark);
int picaso_mark_wait(picaso_ctx *ctx, void *mark);
/* strided row copy: `height` rows of `width_bytes`, used to upload a wavelength shard
 * [w0, w0+n) of an (nlayer, nwno) host plane */
int picaso_memcpy_h2d_2d(picaso_ctx *ctx, void *dst, size_t dpitch_bytes, const void *src,
                         size_t spitch_bytes, size_t width_bytes, size_t height);
/* the reverse: read a wavelength block of a resident (rows, nwno[, nfacets]) plane back */
int picaso_memcpy_d2h_2d(picaso_ctx *ctx, void *dst, size_t dpitch_bytes, const void *src,
                         size_t spitch_bytes, size_t width_bytes, size_t height);
int picaso_memset(picaso_ctx *ctx, void *dst, int value, size_t bytes);
int picaso_sync(picaso_ctx *ctx);
/* work enqueued on `waiter` after this call starts only when everything enqueued on `signaller` so
 * far has finished (device-side; the host does not block).  Two contexts on one device: e.g. the
 * thermal leg of a spectrum on a second stream next to the reflected leg, both reading planes the
 * first stream produced. */
int picaso_ctx_wait(picaso_ctx *waiter, picaso_ctx *signaller);
/* HIP device ordinal the context was created on */
int picaso_ctx_device(picaso_ctx *ctx, int *device);
/* HIP-event timing on the context's own stream (the stream every kernel here is launched on) */
int picaso_timer_start(picaso_ctx *ctx);
int picaso_timer_stop(picaso_ctx *ctx, float *elapsed_ms);
/* raw hipStream_t of the context (for callers that interleave their own work on it) */
void *picaso_stream(picaso_ctx *ctx);

/* ---- multi-GPU: wavelength shards gathered with RCCL over xGMI -------------------------------
 * Replaces the reference's process fan-out (joblib.Parallel over independent spectra, reference
 * picaso/justdoit.py:4774): here ONE spectrum is cut into contiguous wavelength blocks, one per GPU;
 * the solve needs no exchange and the only collective is the all-gather of the result shards.
 *  - one process per GPU: rank 0 calls picaso_comm_unique_id and hands the 128 bytes to the other
 *    ranks through any side channel (picaso_amd/sharding.py: a TCP socket), every rank then calls
 *    picaso_comm_init_rank with its own context;
 *  - one process, several GPUs: picaso_comm_init_all(ndev, ctxs, comms) (ncclCommInitAll).
 * Collectives take DEVICE pointers and are enqueued on the context's stream behind the kernels that
 * produced their input (no host synchronisation); picaso_comm_max / _sum / _barrier move one host
 * double and synchronise (timing reductions of a launcher). */
#define PICASO_COMM_ID_BYTES 128
typedef struct picaso_comm picaso_comm;
int picaso_comm_unique_id(void *id128);
int picaso_comm_init_rank(picaso_ctx *ctx, int nranks, int rank, const void *id128, picaso_comm **out);
int picaso_comm_init_all(int ndev, picaso_ctx *const *ctxs, picaso_comm **out);
void picaso_comm_destroy(picaso_comm *comm);
int picaso_comm_rank(const picaso_comm *comm, int *rank, int *nranks);
/* recv[r*count + i] = rank r's send[i] */
int picaso_all_gather_dev(picaso_comm *comm, const double *send, double *recv, size_t count);
/* ragged shards: rank r's counts[r] elements land at recv + displs[r] on every rank */
int picaso_all_gatherv_dev(picaso_comm *comm, const double *send, double *recv, const size_t *counts,
                           const size_t *displs);
/* overlapped form: the gather starts when everything enqueued on the context's stream so far is done and
 * runs on the communicator's own stream; `slot` (0..3) names the result buffer.  counts == NULL: equal
 * blocks of `count` elements; else the ragged form.  picaso_comm_wait_slot(comm, slot) orders later work
 * of the context's stream behind the last gather of that slot (slot < 0: of all slots); the synchronous
 * collectives and picaso_comm_max / _sum / _barrier wait for all slots first. */
int picaso_all_gather_async_dev(picaso_comm *comm, const double *send, double *recv, size_t count,
                                const size_t *counts, const size_t *displs, int slot);
/* n spectra in one collective launch (a group of n all-gathers): send[i] / recv[i] as above for each */
int picaso_all_gather_multi_async_dev(picaso_comm *comm, int n, const double *const *send, double *const *recv,
                                      size_t count, const size_t *counts, const size_t *displs, int slot);
int picaso_comm_wait_slot(picaso_comm *comm, int slot);
int picaso_comm_max(picaso_comm *comm, double *value);
int picaso_comm_sum(picaso_comm *comm, double *value);
int picaso_comm_barrier(picaso_comm *comm);
/* One thread driving ALL communicators of picaso_comm_init_all (rank order): the per-device calls of one
 * collective sit inside one ncclGroupStart / End and nothing is waited for before every rank is posted.
 * (picaso_all_gather_dev, picaso_comm_max / _sum / _barrier issue one rank's call and are for one thread -- or
 * process -- per communicator.)  recv[i], on device i, receives every device's block; counts == NULL: equal
 * blocks of `count` elements, else counts[r] elements of rank r at displs[r].  Replaces the gather step of the
 * reference's joblib fan-out (justdoit.py:4774) for the single-process form, SURVEY 8(e). */
int picaso_all_gather_group_dev(int n, picaso_comm *const *comms, const double *const *send, double *const *recv,
                                size_t count, const size_t *counts, const size_t *displs);
int picaso_comm_group_max(int n, picaso_comm *const *comms, double *values /* [n], in and out */);
int picaso_comm_group_barrier(int n, picaso_comm *const *comms);

/* ---- Toon89 two-stream reflected light ---------------------------------------------------- */
/* replaces fluxes.get_reflected_1d (reference picaso/fluxes.py:1009-1413).
 * Outputs: xint_at_top (numg,numt,nwno); the four level-flux arrays (numg,numt,nlevel,nwno) are
 * written only when get_lvl_flux != 0 (pass NULL otherwise; the reference returns zeros). */
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
                            double *flux_minus_midpt_all, double *flux_plus_midpt_all);

/* device-resident form of the above.  Optional fused disk integration
 * (disco.compress_disco, reference picaso/disco.py:117-149): if `albedo` is non-NULL,
 * gweight (numg) / tweight (numt) host tables are used to also write albedo (nwno). */
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
                                const double *tweight, double *albedo);

/* Planes of picaso_get_reflected_1d_dev / _batch_dev that compute_opacity derives exactly from others may be NULL when
 * the launch runs the default-options kernels: tau / tau_og (running sums of dtau / dtau_og from 0 at the top,
 * optics.py:353-354, 418-420), gcos2 (0.5 ftau_ray, optics.py:342), and for a column without cloud cosb, cosb_og,
 * ftau_cld, ftau_ray (0, 0, 0, 1) together with dtau_og, w0_og (cosb = 0: no delta-scaling) -- re-derived in the kernel
 * with the same operations, so the same bits, and 3 to 9 of the 11 planes never travel through HBM.  Returns 1 when a
 * call with these arguments may leave them out (the reference's default options: quadrature coefficients, TTHG_ray,
 * N = 2, frac_c = 2; cos_theta = 1 in the symmetric geometry; no level fluxes), 0 when it needs all eleven. */
int picaso_reflected_1d_can_derive(int nlevel, long plane_pitch, int numg, int numt, const double *ubar0,
                                   const double *ubar1, double cos_theta, int single_phase, int multi_phase, double frac_c,
                                   int toon_coefficients, int get_lvl_flux);
/* `nspec` spectra of one shape and one option set in ONE launch (the reference runs the
 * spectra of a retrieval or the phases of a curve as separate processes -- driver.py:405-426,
 * justdoit.py:4741-4777 -- each calling get_reflected_1d, fluxes.py:1009-1413).  Every pointer argument of
 * picaso_get_reflected_1d_dev becomes a HOST array of nspec device pointers (spectrum s: dtau[s], ...,
 * xint_at_top[s] (numg,numt,nwno), albedo[s] (nwno) when the fused disk sum is asked for); entries may
 * repeat (one atmosphere under several geometries: the launch then orders its workgroups so that the
 * spectra share the planes in L2).  Geometry: ngeom = 1 -- ubar0 / ubar1 (numg,numt) and cos_theta[0] for all
 * spectra -- or ngeom = nspec -- ubar0 / ubar1 (nspec,numg,numt), cos_theta[nspec].  get_toa_intensity = 1,
 * get_lvl_flux = 0 (level fluxes stay per-spectrum calls); at most 8 disk angles.  Spectrum s of the result
 * is bit-identical to picaso_get_reflected_1d_dev on its own arguments: a spectrum keeps whole workgroups
 * in the batched grid and no arithmetic depends on the launch shape.  What it buys: a 1e5-column spectrum
 * alone fills 1.5 of the 2 wave slots per SIMD (the launch lasts as long as a doubled SIMD), four of them in
 * one grid run at the kernel's steady rate; small spectra (1e3-1e4 columns) share one launch latency. */
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
                                      const double *gweight, const double *tweight, double *const *albedo);

/* replaces fluxes.get_reflected_3d (reference picaso/fluxes.py:354-660); planes are
 * (nlayer|nlevel, nwno, numg, numt), output (numg,numt,nwno). */
int picaso_get_reflected_3d(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                            int numt, const double *dtau_3d, const double *tau_3d,
                            const double *w0_3d, const double *cosb_3d, const double *gcos2_3d,
                            const double *ftau_cld_3d, const double *ftau_ray_3d,
                            const double *dtau_og_3d, const double *tau_og_3d,
                            const double *w0_og_3d, const double *cosb_og_3d,
                            const double *surf_reflect, const double *ubar0, const double *ubar1,
                            double cos_theta, const double *F0PI, int single_phase, int multi_phase,
                            double frac_a, double frac_b, double frac_c, double constant_back,
                            double constant_forward, double *xint_at_top);

/* device-resident form (+ optional fused compress_disco into `albedo`).  Planes that compute_opacity
 * derives exactly from other planes may be passed as NULL; the kernel then re-derives them with the same
 * operations instead of reading them from HBM (picaso() does this for the whole 3-D path):
 *   tau_3d / tau_og_3d : running sums of dtau_3d / dtau_og_3d from 0 at the top (optics.py:353-354, 418-420)
 *   gcos2_3d           : 0.5 ftau_ray (optics.py:342)
 *   cosb_3d, ftau_cld_3d, ftau_ray_3d, cosb_og_3d (all four, and gcos2_3d): no cloud in any column:
 *                        0, 0, 1, 0 (and 0.5), what optics.py:335-342 gives for TAUCLD = 0 and TAURAY > 0
 *   dtau_og_3d, w0_og_3d (both; then tau_og_3d too): no delta-scaling, equal to dtau_3d / w0_3d -- the case
 *                        of cosb = 0 (optics.py:412-420 with f = 0) and of delta_eddington = False
 * dtau_3d and w0_3d are always required. */
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
                                double *albedo);

/* `nspec` 3-D spectra (the phases of a phase curve, justdoit.py:4741-4777) in one launch: host arrays of nspec device
 * pointers; a plane family that picaso_get_reflected_3d_dev accepts as NULL is left out for ALL spectra by passing
 * a NULL array; ubar0 / ubar1 (nspec, numg, numt) and cos_theta (nspec) on the host.  Bit-identical per spectrum to
 * picaso_get_reflected_3d_dev (fluxes.py:354-660). */
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
                                      double *const *albedo);

/* ---- Toon89 two-stream thermal emission --------------------------------------------------- */
/* replaces fluxes.get_thermal_1d (reference picaso/fluxes.py:1682-1912).
 * Outputs: flux_at_top (numg,numt,nwno); the four (numg,numt,nlevel,nwno) arrays are written when
 * non-NULL (the reference always fills them; pass NULL for a spectrum-only call). */
int picaso_get_thermal_1d(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                          int numt, const double *tlevel, const double *dtau, const double *w0,
                          const double *cosb, const double *plevel, const double *ubar1,
                          const double *surf_reflect, int hard_surface, const double *dwno,
                          int calc_type, double *flux_at_top, double *flux_minus,
                          double *flux_plus, double *flux_minus_mdpt, double *flux_plus_mdpt);

/* device-resident form; optional fused disco.compress_thermal (reference disco.py:151-181) into
 * `flux_disk` (nwno) when non-NULL. */
int picaso_get_thermal_1d_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno,
                              long plane_pitch, int numg, int numt, const double *tlevel,
                              const double *dtau, const double *w0, const double *cosb,
                              const double *plevel, const double *ubar1,
                              const double *surf_reflect, int hard_surface, const double *dwno,
                              int calc_type, double *flux_at_top, double *flux_minus,
                              double *flux_plus, double *flux_minus_mdpt, double *flux_plus_mdpt,
                              const double *gweight, const double *tweight, double *flux_disk);

/* `nspec` thermal spectra in one launch, see picaso_get_reflected_1d_batch_dev: tlevel / plevel are HOST tables
 * (nspec, nlevel), dtau / w0 / cosb / surf_reflect / flux_at_top / flux_disk host arrays of nspec device
 * pointers, wno / dwno shared; ubar1 (numg,numt) for ngeom = 1 or (nspec,numg,numt).  Spectrum-only form
 * (no level fluxes).  Bit-identical per spectrum to picaso_get_thermal_1d_dev (fluxes.py:1682-1912). */
int picaso_get_thermal_1d_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, const double *wno, int nwno,
                                    long plane_pitch, int numg, int numt, const double *tlevel,
                                    const double *const *dtau, const double *const *w0, const double *const *cosb,
                                    const double *plevel, int ngeom, const double *ubar1,
                                    const double *const *surf_reflect, int hard_surface, const double *dwno,
                                    int calc_type, double *const *flux_at_top, const double *gweight,
                                    const double *tweight, double *const *flux_disk);

/* replaces fluxes.get_thermal_3d (reference picaso/fluxes.py:2147-2352); tlevel_3d / plevel_3d are
 * (nlevel,numg,numt), planes (nlayer,nwno,numg,numt). */
int picaso_get_thermal_3d(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                          int numt, const double *tlevel_3d, const double *dtau_3d,
                          const double *w0_3d, const double *cosb_3d, const double *plevel_3d,
                          const double *ubar1, const double *surf_reflect, int hard_surface,
                          double *int_at_top);

/* device-resident form (+ optional fused compress_thermal into `flux_disk`); cosb_3d NULL = no cloud in any
 * column (cosb_og = 0, optics.py:338): the plane is not read */
int picaso_get_thermal_3d_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                              int numt, const double *tlevel_3d, const double *dtau_3d,
                              const double *w0_3d, const double *cosb_3d, const double *plevel_3d,
                              const double *ubar1, const double *surf_reflect, int hard_surface,
                              double *int_at_top, const double *gweight, const double *tweight,
                              double *flux_disk);

/* `nspec` 3-D thermal spectra in one launch: tlevel_3d / plevel_3d (nspec, nlevel, numg, numt) and ubar1
 * (nspec, numg, numt) on the host, the rest host arrays of nspec device pointers (cosb_3d == NULL: no cloud in any
 * spectrum).  Bit-identical per spectrum to picaso_get_thermal_3d_dev (fluxes.py:2147-2352). */
int picaso_get_thermal_3d_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, const double *wno, int nwno, int numg,
                                    int numt, const double *tlevel_3d, const double *const *dtau_3d,
                                    const double *const *w0_3d, const double *const *cosb_3d,
                                    const double *plevel_3d, const double *ubar1,
                                    const double *const *surf_reflect, int hard_surface, double *const *int_at_top,
                                    const double *gweight, const double *tweight, double *const *flux_disk);

/* ---- spherical harmonics (SH2 / SH4) ------------------------------------------------------- */
/* replaces fluxes.get_reflected_SH with setup_2/4_stream_fluxes + solve_4_stream_banded (reference
 * picaso/fluxes.py:2675-2976, :3189-3628).  stream = 2 or 4.  Output xint_at_top (numg,numt,nwno);
 * with flx=1 also the reference's second return value `flux` (numg,numt,stream*nlevel,nwno), the
 * layer moment fluxes F.X + G of calculate_flux (fluxes.py:3631-3635): per level the downward
 * moments then the upward ones, at the top of layer 0 and the bottom of every layer (pass NULL with
 * flx=0).  `compound_f_deltaM` = 1 reproduces the reference's in-place multiplication of f_deltaM
 * once per angle in the TTHG branch (fluxes.py:2823-2824; angle k sees f_deltaM*fac^(k+1));
 * 0 gives the non-compounding variant.  The library never writes to f_deltaM. */
int picaso_get_reflected_SH(picaso_ctx *ctx, int nlevel, int nwno, int numg, int numt,
                            const double *dtau, const double *tau, const double *w0,
                            const double *cosb, const double *ftau_cld, const double *ftau_ray,
                            const double *f_deltaM, const double *dtau_og, const double *tau_og,
                            const double *w0_og, const double *cosb_og, const double *surf_reflect,
                            const double *ubar0, const double *ubar1, double cos_theta,
                            const double *F0PI, int w_single_form, int w_multi_form,
                            int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                            int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                            double constant_back, double constant_forward, int stream, double b_top,
                            int flx, int single_form, int compound_f_deltaM, double *xint_at_top,
                            double *flux);
/* Cloud-free columns: picaso_get_reflected_SH_dev accepts NULL for tau, cosb, ftau_cld, ftau_ray, f_deltaM, dtau_og,
 * tau_og, w0_og and cosb_og TOGETHER (all of them or none; cosb is never read and may be NULL on its own) -- without
 * cloud they are constants (ftau_cld = 0, ftau_ray = 1, cosb = f_deltaM = 0), copies (dtau_og = dtau, w0_og = w0) and
 * running sums of dtau (tau = tau_og), what optics.compute_opacity writes for such an atmosphere (optics.py:303-431).
 * "Cloud-free" means NO cloud profile -- opd = w0 = g0 = 0 in every layer (atmsetup.py get_clouds): COSB is the cloud's
 * g0 itself (optics.py:338), so a layer with g0 != 0 and no optical depth is still delta-scaled and needs all planes.
 * The launch then reads dtau and w0 only and shares the angle-independent half of every layer (stream coefficients,
 * modes, the matrix recursion of the sweep) between the disk angles of a lane (k_sh4_clear, csrc/sh.hip).  Built for
 * stream = 4, the reference's default SH options (config.json) and flx = 0: this function returns 1 when a call with
 * these options may leave the planes out, 0 when it needs all of them.  Results agree with the full-plane launch to
 * <= 1e-9 relative (as both do with the reference), not bit for bit (the full-plane kernel takes angle-dependent and angle-independent reciprocals of a
 * layer from one Newton iteration); a column's bits do not depend on the launch shape. */
int picaso_reflected_SH_can_derive(int stream, int w_single_form, int w_multi_form, int psingle_form,
                                   int w_single_rayleigh, int w_multi_rayleigh, int psingle_rayleigh, double frac_c,
                                   int single_form, int flx);
/* The level planes alone may be left out as well: tau and tau_og TOGETHER NULL, everything else given (a cloudy
 * atmosphere, where the cloud-free form above does not apply).  compute_opacity forms them as running sums of dtau /
 * dtau_og from 0 at the top (optics.py:353-354, 418-420), so the launch carries the beam exponentials exp(-tau/u0) and
 * exp(-tau_og/u0) down the column as running products of the layers' exp(-dtau/u0) -- in the symmetric geometry (ubar0
 * == ubar1) the factor exp(-dtau/ubar1) the layer forms anyway: two of a layer's five exponentials and three of its
 * eleven loads less.  The reference's clipped exp(-clip35(tau/u0)) is max(., e^-35) of the running product.  Returns 1
 * when a call with these arguments may do so (the default phase-function options, flx = 0, planes below 4 GB), 0 when it
 * needs both planes.  Results agree with the plane-reading launch to the rounding of the product (<= nlayer ulp of the
 * exponentials: ~1e-14 relative in the exponentials, <= 1e-11 in the intensities), not bit for bit; they do not depend on launch shape or wavelength block. */
int picaso_reflected_SH_can_derive_levels(int nlevel, long plane_pitch, int stream, int w_single_form, int w_multi_form,
                                          int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                          int psingle_rayleigh, double frac_c, int single_form, int flx);
int picaso_get_reflected_SH_dev(picaso_ctx *ctx, int nlevel, int nwno, long plane_pitch, int numg,
                                int numt, const double *dtau, const double *tau, const double *w0,
                                const double *cosb, const double *ftau_cld, const double *ftau_ray,
                                const double *f_deltaM, const double *dtau_og, const double *tau_og,
                                const double *w0_og, const double *cosb_og,
                                const double *surf_reflect, const double *ubar0,
                                const double *ubar1, double cos_theta, const double *F0PI,
                                int w_single_form, int w_multi_form, int psingle_form,
                                int w_single_rayleigh, int w_multi_rayleigh, int psingle_rayleigh,
                                double frac_a, double frac_b, double frac_c, double constant_back,
                                double constant_forward, int stream, double b_top, int flx,
                                int single_form, int compound_f_deltaM, double *xint_at_top,
                                double *flux, const double *gweight, const double *tweight,
                                double *albedo);

/* picaso_get_reflected_SH_dev with one more statement from the caller: the first `cloud_free_above` layers (0 ..
 * cloud_free_above - 1, counted from the top) carry no cloud in ANY column -- their planes hold what
 * optics.compute_opacity writes for a layer without a cloud profile entry (ftau_cld = cosb_og = f_deltaM = 0, ftau_ray = 1,
 * dtau_og = dtau, w0_og = w0, tau = tau_og = the running sum of dtau).  A cloud deck sits below some pressure; above it
 * the SH blocks are the same for every disk angle (fluxes.py:2823-2824 compounds an f_deltaM of zero), so those layers go
 * through the cloud-free kernel (two angles per lane share the angle-independent half of a layer), which hands its sweep
 * state to the full kernel at the first cloudy layer.  Taken for stream = 4, the default forms and flx = 0
 * (picaso_reflected_SH_can_derive) from 4 layers on; otherwise, and with cloud_free_above = 0, this IS
 * picaso_get_reflected_SH_dev.  Results agree with the unsplit launch to <= 1e-9 relative (not bit for bit); they do not
 * depend on launch shape or wavelength block, but they do depend on cloud_free_above: every block of a sharded spectrum
 * must be given the same value.  The statement is not verified (PICASO_AMD_SH_CHECK_TOP=1 in the environment checks it
 * on the device and fails the call, synchronising -- a debugging aid). */
int picaso_get_reflected_SH_top_dev(picaso_ctx *ctx, int nlevel, int nwno, long plane_pitch, int numg,
                                    int numt, const double *dtau, const double *tau, const double *w0,
                                    const double *cosb, const double *ftau_cld, const double *ftau_ray,
                                    const double *f_deltaM, const double *dtau_og, const double *tau_og,
                                    const double *w0_og, const double *cosb_og,
                                    const double *surf_reflect, const double *ubar0,
                                    const double *ubar1, double cos_theta, const double *F0PI,
                                    int w_single_form, int w_multi_form, int psingle_form,
                                    int w_single_rayleigh, int w_multi_rayleigh, int psingle_rayleigh,
                                    double frac_a, double frac_b, double frac_c, double constant_back,
                                    double constant_forward, int stream, double b_top, int flx,
                                    int single_form, int compound_f_deltaM, int cloud_free_above,
                                    double *xint_at_top, double *flux, const double *gweight,
                                    const double *tweight, double *albedo);

/* `nspec` SH spectra (flx = 0) of one shape and option set in one launch, see picaso_get_reflected_1d_batch_dev:
 * host arrays of nspec device pointers, ngeom = 1 or nspec geometries (at most 16 disk angles), bit-identical per
 * spectrum to picaso_get_reflected_SH_dev (fluxes.py:2675-2976).  Planes are read, never modified (the reference's
 * in-place f_deltaM compounding is reproduced per angle inside the kernel). */
int picaso_get_reflected_SH_batch_dev(picaso_ctx *ctx, int nspec, int nlevel, int nwno, long plane_pitch, int numg,
                                      int numt, const double *const *dtau, const double *const *tau,
                                      const double *const *w0, const double *const *cosb,
                                      const double *const *ftau_cld, const double *const *ftau_ray,
                                      const double *const *f_deltaM, const double *const *dtau_og,
                                      const double *const *tau_og, const double *const *w0_og,
                                      const double *const *cosb_og, const double *const *surf_reflect, int ngeom,
                                      const double *ubar0, const double *ubar1, const double *cos_theta,
                                      const double *const *F0PI, int w_single_form, int w_multi_form,
                                      int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                      int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                                      double constant_back, double constant_forward, int stream, double b_top,
                                      int single_form, int compound_f_deltaM, double *const *xint_at_top,
                                      const double *gweight, const double *tweight, double *const *albedo);

/* replaces fluxes.get_thermal_SH (reference picaso/fluxes.py:2979-3186), flx = 0.  Of the
 * reference's arguments only tlevel, dtau, w0, cosb_og, plevel, ubar1, surf_reflect are read by
 * its arithmetic; `cosb_differs_from_cosb_og` carries the reference's
 * `np.array_equal(cosb, cosb_og)` test (fluxes.py:3072-3075), evaluated by the caller. */
int picaso_get_thermal_SH(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int numg,
                          int numt, const double *tlevel, const double *dtau, const double *tau,
                          const double *w0, const double *cosb_og, const double *plevel,
                          const double *ubar1, const double *surf_reflect, int stream,
                          int hard_surface, int cosb_differs_from_cosb_og, int flx,
                          double *xint_at_top);
int picaso_get_thermal_SH_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno,
                              long plane_pitch, int numg, int numt, const double *tlevel,
                              const double *dtau, const double *tau, const double *w0,
                              const double *cosb_og, const double *plevel, const double *ubar1,
                              const double *surf_reflect, int stream, int hard_surface,
                              int cosb_differs_from_cosb_og, int flx, double *xint_at_top,
                              const double *gweight, const double *tweight, double *flux_disk);

/* ---- Planck-function tables ------------------------------------------------------------------ */
/* replaces fluxes.blackbody(t, w) (reference picaso/fluxes.py:1660-1680): Planck function per unit wavelength, cgs,
 * `t` (ntemp) in K, `w_cm` (nwave) WAVELENGTH in cm; out (ntemp, nwave) C-order.  An exponential that overflows gives 0
 * (numpy's 1/(inf - 1)).  The solvers evaluate the same device function level by level and never read such a table
 * (fluxes.py:1752); the entry point is for callers that use blackbody() on its own (brightness temperatures). */
int picaso_blackbody(picaso_ctx *ctx, int ntemp, const double *t, long nwave, const double *w_cm, double *out);
int picaso_blackbody_dev(picaso_ctx *ctx, int ntemp, const double *t, long nwave, const double *w_cm, double *out);
/* replaces fluxes.blackbody_integrated(T, wave, dwave) (reference picaso/fluxes.py:1609-1658): mean of the wavenumber
 * Planck function at wave - dwave/2, wave, wave + dwave/2 (nbb = 1), what get_thermal_1d(calc_type=1) uses
 * (fluxes.py:1754); `wave`, `dwave` (nwave) in cm^-1; out (ntemp, nwave). */
int picaso_blackbody_integrated(picaso_ctx *ctx, int ntemp, const double *T, long nwave, const double *wave,
                                const double *dwave, double *out);
int picaso_blackbody_integrated_dev(picaso_ctx *ctx, int ntemp, const double *T, long nwave, const double *wave,
                                    const double *dwave, double *out);

/* ---- disk quadrature ----------------------------------------------------------------------- */
/* replaces disco.compress_disco (reference picaso/disco.py:117-149) */
int picaso_compress_disco(picaso_ctx *ctx, int nwno, double cos_theta, const double *xint_at_top,
                          const double *gweight, int ng, const double *tweight, int nt,
                          const double *F0PI, double *albedo);
/* device form; F0PI == NULL means F0PI = 1 (how the reference disk-integrates its level fluxes,
 * justdoit.py:536-548: pass nwno = nlevel*nwno columns) */
int picaso_compress_disco_dev(picaso_ctx *ctx, int nwno, double cos_theta,
                              const double *xint_at_top, const double *gweight, int ng,
                              const double *tweight, int nt, const double *F0PI, double *albedo);
/* replaces disco.compress_thermal (reference picaso/disco.py:151-181); `ninner` = nwno for a
 * (ng,nt,nwno) input or nlevel*nwno for a (ng,nt,nlevel,nwno) input */
int picaso_compress_thermal(picaso_ctx *ctx, size_t ninner, const double *flux_at_top,
                            const double *gweight, int ng, const double *tweight, int nt,
                            double *flux);
int picaso_compress_thermal_dev(picaso_ctx *ctx, size_t ninner, const double *flux_at_top,
                                const double *gweight, int ng, const double *tweight, int nt,
                                double *flux);

/* ---- on-the-fly correlated-k gas mixing (resort-rebin) ---------------------------------------- */
/* replaces deq_chem.mix_all_gases_gasesfly (reference picaso/deq_chem.py:333-384, with
 * do_mixing_mono_gasesfly :387-477 and mix_2_gases :537-597), the loop nest of
 * RetrieveCKs.mix_my_opacities_gasesfly (picaso/optics.py:1164-1198).
 * kappas: host array of `ngas` DEVICE pointers, each ln(kappa) of one gas laid out
 * (npres, ntemp, nwno, ngauss) as the reference's self.kappas[mol]; mixes: host (ngas, nlayer) layer
 * volume mixing ratios; gauss_pts / gauss_wts: host (ngauss), ngauss <= 8; indices: host (4, nlayer)
 * int32 = [p_low, p_hi, t_low, t_hi] of get_mixing_indices (optics.py:1200-1278).
 * Output (device): ln of the mixed coefficients laid out (nlayer, 4, nwno, ngauss), neighbour index
 * ct = 0 (p_low,t_low), 1 (p_low,t_hi), 2 (p_hi,t_low), 3 (p_hi,t_hi) -- the reference's
 * (nlayer, nwno, ngauss, 4) array with the neighbour axis moved forward, so that each neighbour is a
 * (nwno*ngauss) table row for the ln-bilinear interpolation of picaso_opacity_gas_ck_dev. */
int picaso_mix_all_gases_gasesfly_dev(picaso_ctx *ctx, int ngas, const double *const *kappas, int npres,
                                      int ntemp, int nwno, int ngauss, const double *mixes,
                                      const double *gauss_pts, const double *gauss_wts, const int *indices,
                                      int nlayer, double *kappa_mixed);

/* ---- transmission ------------------------------------------------------------------------- */
/* replaces fluxes.get_transit_1d (reference picaso/fluxes.py:2581-2663): (Rp/Rs)^2 per wavelength
 * from the chord-integrated slant optical depth (Brown 2001, eq. 11).  z, dz, player, tlayer are
 * host arrays of length nlevel indexed exactly as the reference indexes them (its caller passes
 * the level pressure / temperature, justdoit.py:390-394); mmw, colden host arrays of length
 * nlevel-1; dtau (nlevel-1, nwno) = DTAU_OG.  Output rprs2 (nwno). */
int picaso_get_transit_1d(picaso_ctx *ctx, const double *z, const double *dz, int nlevel, int nwno,
                          double rstar, const double *mmw, double k_b, double amu, const double *player,
                          const double *tlayer, const double *colden, const double *dtau, double *rprs2);
/* device-resident dtau / rprs2 (row pitch `plane_pitch` elements) */
int picaso_get_transit_1d_dev(picaso_ctx *ctx, const double *z, const double *dz, int nlevel, int nwno,
                              long plane_pitch, double rstar, const double *mmw, double k_b, double amu,
                              const double *player, const double *tlayer, const double *colden,
                              const double *dtau, double *rprs2);

/* The correlated-k loop of the transmission branch (reference picaso/justdoit.py:388-405):
 * dtau is (nlevel-1, nwno*ngauss) with the Gauss index fastest (= DTAU_OG[:, :, ig] for every ig),
 * rprs2 (nwno) = sum_ig get_transit_1d(DTAU_OG[:,:,ig]) * gauss_wts[ig], summed in ig order.
 * gauss_wts is a host array of ngauss (<= 32) weights. */
int picaso_get_transit_1d_ck_dev(picaso_ctx *ctx, const double *z, const double *dz, int nlevel, int nwno,
                                 int ngauss, double rstar, const double *mmw, double k_b, double amu,
                                 const double *player, const double *tlayer, const double *colden,
                                 const double *dtau, const double *gauss_wts, double *rprs2);

/* ---- correlated-k Gauss-point batch and patchy-cloud blend -------------------------------- */
/* The reference loops the solver over the `ngauss` correlated-k points of every wavelength bin and
 * accumulates `xint_at_top += xint * gauss_wts[ig]` (reference picaso/justdoit.py:256-307 reflected,
 * :328-380 thermal), slicing `DTAU[:,:,ig]` out of the (nlayer|nlevel, nwno, ngauss) arrays that
 * compute_opacity returns (optics.py:423-431).  These entry points take those arrays as they are
 * (Gauss index fastest), solve all nwno*ngauss columns in one launch and return the Gauss-weighted
 * (numg,numt,nwno) intensities; `gauss_wts` is a host array of length ngauss (<= 32).  Optional
 * fused disk quadrature as in the _dev forms above.  The four level-flux outputs
 * (numg,numt,nlevel,nwno) are Gauss-weighted the same way (justdoit.py:309-313, 372-376: the climate
 * caller's inputs) and are written when get_lvl_flux = 1 / when non-NULL. */
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
                                   const double *gweight, const double *tweight, double *albedo);
int picaso_get_thermal_1d_ck_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int ngauss,
                                 int numg, int numt, const double *tlevel, const double *dtau,
                                 const double *w0, const double *cosb, const double *plevel,
                                 const double *ubar1, const double *surf_reflect, int hard_surface,
                                 const double *dwno, int calc_type, const double *gauss_wts,
                                 double *flux_at_top, double *flux_minus, double *flux_plus,
                                 double *flux_minus_mdpt, double *flux_plus_mdpt, const double *gweight,
                                 const double *tweight, double *flux_disk);

/* The same Gauss-point loop around the spherical-harmonics solvers (reference picaso/justdoit.py:256-269 + :307 reflected,
 * :364-370 + :380 thermal: get_reflected_SH / get_thermal_SH once per Gauss point on plane[:, :, ig], accumulated with
 * gauss_wts[ig] in ig order).  Planes (nlayer|nlevel, nwno, ngauss) as compute_opacity returns them, Gauss index fastest;
 * surf_reflect / F0PI / wno (nwno); output (numg,numt,nwno) (+ optional fused disk sum).  All nwno*ngauss columns go
 * through ONE launch of the SH kernels, so every column carries the bits of the per-Gauss-point call
 * picaso_get_reflected_SH_top_dev / picaso_get_thermal_SH_dev (arguments as there; flx = 0: the reference discards the
 * layer fluxes of the loop, justdoit.py:259).  The reference's in-place f_deltaM compounding acts on each Gauss slice
 * separately (a view per ig), i.e. per column as in the monochromatic call.  cloud_free_above as in _top_dev (0: none). */
int picaso_get_reflected_SH_ck_dev(picaso_ctx *ctx, int nlevel, int nwno, int ngauss, int numg, int numt,
                                   const double *dtau, const double *tau, const double *w0, const double *cosb,
                                   const double *ftau_cld, const double *ftau_ray, const double *f_deltaM,
                                   const double *dtau_og, const double *tau_og, const double *w0_og,
                                   const double *cosb_og, const double *surf_reflect, const double *ubar0,
                                   const double *ubar1, double cos_theta, const double *F0PI, int w_single_form,
                                   int w_multi_form, int psingle_form, int w_single_rayleigh, int w_multi_rayleigh,
                                   int psingle_rayleigh, double frac_a, double frac_b, double frac_c,
                                   double constant_back, double constant_forward, int stream, double b_top,
                                   int single_form, int compound_f_deltaM, int cloud_free_above, const double *gauss_wts,
                                   double *xint_at_top, const double *gweight, const double *tweight, double *albedo);
int picaso_get_thermal_SH_ck_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int ngauss, int numg,
                                 int numt, const double *tlevel, const double *dtau, const double *tau,
                                 const double *w0, const double *cosb_og, const double *plevel, const double *ubar1,
                                 const double *surf_reflect, int stream, int hard_surface,
                                 int cosb_differs_from_cosb_og, const double *gauss_wts, double *xint_at_top,
                                 const double *gweight, const double *tweight, double *flux_disk);

/* ... and around the 3-D solvers (reference picaso/justdoit.py:488-500 reflected, :502-516 thermal: get_reflected_3d /
 * get_thermal_3d once per Gauss point on DTAU_3d[:, :, :, :, ig]).  Planes are FACET-MAJOR: (numg*numt, nlayer|nlevel,
 * nwno, ngauss), facet f = g*numt + t, Gauss index fastest -- what picaso_compute_opacity_facet_major_ck_dev writes; the
 * reference's (nlayer, nwno, numg, numt, ngauss) arrays are this with the facet axes moved to the front.  Every facet is
 * solved as a spectrum of nwno*ngauss columns by the batched 3-D launches (a wave = 64 columns of one facet, all loads
 * coalesced), bit-identical per (facet, wavelength, Gauss point) to picaso_get_reflected_3d_dev / picaso_get_thermal_3d_dev
 * on the slice.  Plane families that picaso_get_reflected_3d_dev accepts as NULL may be NULL here (tau / tau_og / gcos2
 * re-derived, no cloud, no delta-scaling; cosb NULL in the thermal call = no cloud).  tlevel_3d / plevel_3d HOST
 * (nlevel, numg, numt), ubar0 / ubar1 HOST (numg, numt); output (numg,numt,nwno) (+ optional fused disk sum). */
int picaso_get_reflected_3d_ck_dev(picaso_ctx *ctx, int nlevel, int nwno, int ngauss, int numg, int numt,
                                   const double *dtau, const double *tau, const double *w0, const double *cosb,
                                   const double *gcos2, const double *ftau_cld, const double *ftau_ray,
                                   const double *dtau_og, const double *tau_og, const double *w0_og,
                                   const double *cosb_og, const double *surf_reflect, const double *ubar0,
                                   const double *ubar1, double cos_theta, const double *F0PI, int single_phase,
                                   int multi_phase, double frac_a, double frac_b, double frac_c, double constant_back,
                                   double constant_forward, const double *gauss_wts, double *xint_at_top,
                                   const double *gweight, const double *tweight, double *albedo);
int picaso_get_thermal_3d_ck_dev(picaso_ctx *ctx, int nlevel, const double *wno, int nwno, int ngauss, int numg,
                                 int numt, const double *tlevel_3d, const double *dtau, const double *w0,
                                 const double *cosb, const double *plevel_3d, const double *ubar1,
                                 const double *surf_reflect, int hard_surface, const double *gauss_wts,
                                 double *int_at_top, const double *gweight, const double *tweight, double *flux_disk);

/* The thermal leg of climate.get_fluxes (reference picaso/climate.py:1879-1941) for `nitem` level-temperature profiles
 * over ONE set of opacity planes in one launch sequence: the climate solver's Jacobian perturbs one level temperature at
 * a time and calls get_fluxes with unchanged opacities (climate.py:1105-1180), ~nlevel calls per Newton step.  Planes
 * dtau / w0 / cosb (nlayer, nwno*ngauss) as picaso_get_thermal_1d_ck_dev takes them; tlevel HOST (nitem, nlevel), plevel
 * HOST (nlevel).  Output disk4 (4, nlevel, nitem*nwno): flux_minus, flux_plus, flux_minus_mdpt, flux_plus_mdpt, Gauss-
 * weighted and disk-integrated (compress_thermal), column = profile * nwno + wavelength.  Every profile's numbers are
 * bit-identical to picaso_get_thermal_1d_ck_dev + picaso_compress_thermal_dev on that profile. */
int picaso_get_thermal_1d_ck_tbatch_dev(picaso_ctx *ctx, int nitem, int nlevel, const double *wno, int nwno, int ngauss,
                                        int numg, int numt, const double *tlevel, const double *dtau, const double *w0,
                                        const double *cosb, const double *plevel, const double *ubar1,
                                        const double *surf_reflect, int hard_surface, const double *dwno, int calc_type,
                                        const double *gauss_wts, const double *gweight, const double *tweight,
                                        double *disk4);
/* the wavenumber sums of climate.get_fluxes (climate.py:1931-1936) on picaso_get_thermal_1d_ck_tbatch_dev's output:
 * net_layer[profile][level] = sum_w (plus_mdpt - minus_mdpt) dwno, net[profile][level] = sum_w (plus - minus) dwno
 * (device (nitem, nlevel) each).  Deterministic tree sums, not numpy's order: ~1e-16 of sum|terms| from get_fluxes' own. */
int picaso_flux_net_sums_dev(picaso_ctx *ctx, int nlevel, int nitem, int nwno, const double *disk4,
                             const double *dwno, double *net_layer, double *net);
/* np.trapezoid(y', x) with d = diff(x) resident: sum_j (d[j] * (y'[j + 1] + y'[j])) / 2.0 over the n - 1 intervals, summed
 * in numpy's own pairwise order (blocks of <= 128 terms with eight partial sums), so the same bits as the reference's
 * spectrum-wide integrals (justdoit.py:552-599: Bond albedo, effective temperature).  y'[k] = y[k] * mult[k] (mult may be
 * NULL), with reverse != 0 y'[k] = y[n - 1 - k] (* mult[n - 1 - k]).  out: one double in device memory. */
int picaso_trapz_dev(picaso_ctx *ctx, long n, const double *d, const double *y, const double *mult, int reverse,
                     double *out);
/* out = a*x + b*y on device arrays: the patchy-cloud blend (1-fhole)*cloudy + fhole*clear
 * (reference picaso/justdoit.py:300-305, 356-361) */
int picaso_axpby_dev(picaso_ctx *ctx, size_t n, double a, const double *x, double b, const double *y,
                     double *out);

/* ---- opacity pre-stage --------------------------------------------------------------------- */
/* Gas + Rayleigh optical depth per (layer, wavelength) from HBM-resident opacity tables.
 * Replaces the arithmetic of RetrieveOpacities.get_opacities / get_opacities_nearest (reference
 * picaso/optics.py:2241-2368: per layer, bilinear interpolation of log10(kappa) in (1/T, log10 P)
 * between the four bracketing table rows -- or the nearest row -- times Avogadro's number) and of
 * the TAUGAS / TAURAY sums of compute_opacity (optics.py:144-277).  The table rows themselves
 * (read from the sqlite DB by host Python) live on the device:
 *   mol_tables[m]  : (n_pt_rows, nwno)  kappa (nearest) or log10(kappa, zero -> 1e-50) (linear)
 *   cont_tables[c] : (n_cia_temps, nwno) continuum opacity
 *   ray_tables[r]  : (nwno)             Rayleigh cross section
 * Small per-layer host tables select and weight them:
 *   mol_rows  [nmol][nlayer][4] row indices   mol_wts [nmol][nlayer][4] interpolation weights
 *   mol_fac   [nmol][nlayer]    colden*x/mmw (times exclude_mol factor)
 *   cont_rows [ncont][nlayer]   nearest-temperature row, cont_fac [ncont][nlayer] layer coefficient
 *   ray_fac   [nray][nlayer]    colden*x/mmw
 * Outputs taugas, tauray: device planes (nlayer, nwno).  `linear` = 1 for query_method='linear'. */
int picaso_opacity_gas_dev(picaso_ctx *ctx, int nlayer, int nwno, int linear, int nmol,
                           const double *const *mol_tables, const int *mol_rows,
                           const double *mol_wts, const double *mol_fac, int ncont,
                           const double *const *cont_tables, const int *cont_rows,
                           const double *cont_fac, int nray, const double *const *ray_tables,
                           const double *ray_fac, double *taugas, double *tauray);

/* replaces the mixing half of optics.compute_opacity (reference picaso/optics.py:327-431):
 * DTAU, TAU, W0, COSB, ftau_cld, ftau_ray, GCOS2, W0_no_raman, f_deltaM and the delta-Eddington
 * scaled set from the gas / Rayleigh / cloud optical depths.  All arrays are device planes
 * (nlayer, nwno) except tau, tau_og (nlevel, nwno).  `raman_factor` may be NULL (then
 * `raman_const`, 0.99999 for raman='none', is used); `taucld`, `w0_cld`, `g0_cld` may be NULL
 * (cloud-free atmosphere: read as zero planes).  Any of the 13 output planes may be NULL: it is then
 * not written (a thermal-only caller needs dtau_og, w0_no_raman and cosb_og only).  test_mode: 0 off, 1 'rayleigh',
 * 2 constant-tau (optics.py:372-399).  Output order = the reference's return tuple
 * (optics.py:423-431). */
int picaso_compute_opacity_dev(picaso_ctx *ctx, int nlayer, int nwno, const double *taugas,
                               const double *tauray, const double *taucld, const double *w0_cld,
                               const double *g0_cld, const double *raman_factor,
                               double raman_const, int test_mode, int delta_eddington, int stream,
                               double *dtau, double *tau, double *w0, double *cosb,
                               double *ftau_cld, double *ftau_ray, double *gcos2, double *dtau_og,
                               double *tau_og, double *w0_og, double *cosb_og, double *w0_no_raman,
                               double *f_deltaM);

/* Correlated-k forms of the two calls above (reference RetrieveCKs.get_pre_mix_ck,
 * picaso/optics.py:1081-1161; RetrieveCKs.get_continuum, :1398-1498; compute_opacity with
 * ngauss > 1, :234-262).  Molecular tables have nwno*ngauss columns per row (Gauss index fastest,
 * the reference's kappa[p,t,wno,gauss] layout) and taugas / all 13 outputs are
 * (nlayer|nlevel, nwno, ngauss); Rayleigh, cloud and Raman planes have no Gauss axis.
 *   mol_mode : 0 nearest row, 1 10**(sum_4 w log10 kappa), 2 exp(sum_4 w ln kappa) (premixed CK)
 *   cont_mode: 0 nearest-temperature row (cont_rows [ncont][nlayer], cont_wts NULL),
 *              1 exp(w0 ln k[row0] + w1 ln k[row1]) (cont_rows, cont_wts [ncont][nlayer][2];
 *                tables hold ln kappa)
 *   raman_rows: nlayer -- raman_factor is a (nlayer, nwno) plane (one per facet, facet-major, in the 3-D form), or
 *               0 -- ONE row of nwno values used for every layer and facet: the Pollack table, which the reference
 *               tiles over the layers with np.repeat (optics.py:296-298, 584-652).  picaso_compute_opacity_dev
 *               takes planes. */
int picaso_opacity_gas_ck_dev(picaso_ctx *ctx, int nlayer, int nwno, int ngauss, int mol_mode, int nmol,
                              const double *const *mol_tables, const int *mol_rows,
                              const double *mol_wts, const double *mol_fac, int cont_mode, int ncont,
                              const double *const *cont_tables, const int *cont_rows,
                              const double *cont_wts, const double *cont_fac, int nray,
                              const double *const *ray_tables, const double *ray_fac, double *taugas,
                              double *tauray);
int picaso_compute_opacity_ck_dev(picaso_ctx *ctx, int nlayer, int nwno, int ngauss, const double *taugas,
                                  const double *tauray, const double *taucld, const double *w0_cld,
                                  const double *g0_cld, const double *raman_factor, int raman_rows,
                                  double raman_const, int test_mode, int delta_eddington, int stream,
                                  double *dtau, double *tau, double *w0, double *cosb,
                                  double *ftau_cld, double *ftau_ray, double *gcos2, double *dtau_og,
                                  double *tau_og, double *w0_og, double *cosb_og, double *w0_no_raman,
                                  double *f_deltaM);

/* Gas stage and compute_opacity as ONE launch for monochromatic tables (ngauss = 1): picaso_opacity_gas_ck_dev followed
 * by picaso_compute_opacity_ck_dev, bit for bit, without the TAUGAS / TAURAY planes travelling to HBM and back (2 x 72 MB
 * written and read at 1e5 x 90; the two launches 0.22 ms, this one ~0.12).  The mixing is element-wise except for the
 * level planes tau / tau_og (running sums down a column, optics.py:353-354, 418-420): those are a second, small launch --
 * level_sums = 1: issued here; 0: left to the caller (picaso_level_sums_dev), who may first let another stream start
 * on the layer planes (the thermal leg reads no level plane).  Output planes may be NULL as for compute_opacity; tau
 * needs dtau, tau_og needs dtau_og.  Cloud: three (nlayer, nwno) planes, or -- cld_nin >= 2 -- tables on their own grid
 * cld_xp (cld_nin, increasing), cld_fp (3 nlayer, cld_nin: opd rows, w0 rows, g0 rows), interpolated to the wavenumbers
 * cld_x (nwno) inside the launch with numpy.interp's bits (picaso_regrid_rows_dev's), no regridded planes in HBM. */
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
                                   const double *cld_xp, const double *cld_fp, const double *cld_x);
/* tau[0] = 0, tau[i + 1] = tau[i] + dtau[i] for (nlayer, ncol) layer planes -> (nlayer + 1, ncol) level planes; either
 * pair may be NULL */
int picaso_level_sums_dev(picaso_ctx *ctx, int nlayer, long ncol, const double *dtau, double *tau, const double *dtau_og,
                          double *tau_og);
/* 3-D path (reference picaso/justdoit.py:444-471 fills `DTAU_3d[:,:,g,t,:] = dtau` facet by facet):
 * one launch mixes all facets.  taugas, tauray (and raman_factor when non-NULL) are facet-major
 * (nfacets, nlayer, nwno) -- picaso_opacity_gas_dev is called once per facet on its slice -- the cloud
 * inputs are (nlayer, nwno, nfacets) as the caller holds them (NULL = no cloud), and the 13 outputs are
 * the (nlayer|nlevel, nwno, nfacets) planes picaso_get_reflected_3d / _thermal_3d take. */
int picaso_compute_opacity_facets_dev(picaso_ctx *ctx, int nlayer, int nwno, int nfacets,
                                      const double *taugas, const double *tauray, const double *taucld,
                                      const double *w0_cld, const double *g0_cld, const double *raman_factor, int raman_rows,
                                      double raman_const, int test_mode, int delta_eddington, int stream,
                                      double *dtau, double *tau, double *w0, double *cosb,
                                      double *ftau_cld, double *ftau_ray, double *gcos2, double *dtau_og,
                                      double *tau_og, double *w0_og, double *cosb_og, double *w0_no_raman,
                                      double *f_deltaM);

/* The facet loop of the 3-D branch for correlated-k tables (reference picaso/justdoit.py:437-471: compute_opacity facet
 * by facet, `DTAU_3d[:, :, g, t, :] = dtau`), facet-major: taugas (nfacets, nlayer, nwno, ngauss) and tauray (nfacets,
 * nlayer, nwno) from picaso_opacity_gas_ck_dev on the tall atmosphere of all facets; the cloud planes (nlayer, nwno) one
 * per facet `cloud_stride` elements apart (0: one set for the whole disk; NULL: no cloud); raman_factor with raman_rows =
 * nlayer one (nlayer, nwno) plane per facet, facet-major, with raman_rows = 0 one row for everything.  The 13 outputs are
 * (nfacets, nlayer|nlevel, nwno, ngauss), any of them NULL = not written; each facet's numbers are those of
 * picaso_compute_opacity_ck_dev on that facet. */
int picaso_compute_opacity_facet_major_ck_dev(picaso_ctx *ctx, int nfacets, int nlayer, int nwno, int ngauss,
                                              const double *taugas, const double *tauray, const double *taucld,
                                              const double *w0_cld, const double *g0_cld, long cloud_stride,
                                              const double *raman_factor, int raman_rows, double raman_const,
                                              int test_mode, int delta_eddington, int stream, double *dtau, double *tau,
                                              double *w0, double *cosb, double *ftau_cld, double *ftau_ray, double *gcos2,
                                              double *dtau_og, double *tau_og, double *w0_og, double *cosb_og,
                                              double *w0_no_raman, double *f_deltaM);

/* A (nrows, nwno) plane shared by all facets -> (nrows, nwno, nfacets), facet index fastest, optionally times
 * facet_scale[f] (host array of nfacets, or NULL): cloud tables that do not vary over the disk, and the synthetic
 * facet planes of bench.py --config 4, without building nfacets copies on the host (the reference tiles
 * `DTAU_3d[:,:,g,t] = dtau` facet by facet on the CPU, justdoit.py:444-471). */
int picaso_broadcast_facets_dev(picaso_ctx *ctx, size_t nrows, int nwno, int nfacets, const double *src,
                                const double *facet_scale, double *dst);

/* Rows of a (nrows, nin) device table on the increasing grid xp (device, nin >= 2) -> out (nrows, nwno) on the grid
 * x (device): numpy.interp per row with numpy's own arithmetic, i.e. the reference's wavelength.regrid
 * (wavelength.py:46-70) as atmsetup.get_clouds applies it to the cloud opd / w0 / g0 tables (atmsetup.py:609-622).
 * scale (host pointer or NULL): out = *scale * interp, the thinned-cloud multiply of optics.py:314-315. */
int picaso_regrid_rows_dev(picaso_ctx *ctx, int nrows, int nin, long nwno, const double *xp, const double *fp,
                           const double *x, const double *scale, double *out);

/* The same for tables that differ from facet to facet (3-D path, clouds_3d on a grid of their own: the reference
 * regrids facet by facet on the host, justdoit.py:437-449 -> atmsetup.py:609-622): fp (nlayer, nfacets, nin) -> out
 * (nlayer, nwno, nfacets), facet index fastest -- the layout picaso_compute_opacity_facets_dev reads its cloud inputs in.
 * numpy.interp's bits per (layer, facet) row. */
int picaso_regrid_facets_dev(picaso_ctx *ctx, int nlayer, int nfacets, int nin, long nwno, const double *xp,
                             const double *fp, const double *x, double *out);


/* Oklopcic (2016) Raman factor plane, replaces optics.compute_raman (reference picaso/optics.py:434-494) as
 * compute_opacity calls it (optics.py:285-294): out (nlayer, nwno) = min((ray + w_shift)/(ray + wo_shift), cap).
 * Q, QS: device tables (ntrans, nwno) of c_i / wno**3 / (wno + deltanu_i) and of that times the stellar shift ratio
 * of transition i (formed once per opacity grid / star by the caller; QS rows of Rayleigh transitions are not read);
 * j_initial, is_rayleigh (deltanu_i == 0): host, ntrans ints; j_at_temp: host (10, nlayer) rotational populations
 * j_fraction(j, T_layer) (optics.py:524-545). */
int picaso_raman_oklopcic_dev(picaso_ctx *ctx, int nlayer, long nwno, int ntrans, const double *Q, const double *QS,
                              const int *j_initial, const int *is_rayleigh, const double *j_at_temp, double cap,
                              double *out);

/* ---- one call per spectrum: every launch of picaso()'s 1-D path (Toon or SH), for every wavelength block ---------
 * (reference picaso/justdoit.py:236-385 is the per-spectrum sequence; its fan-out over processes :4774).  The
 * caller keeps one picaso_block per wavelength block (pointers into that block's resident tables and a workspace it
 * allocated once) and fills one picaso_spectrum_job per call with the per-layer tables all blocks share; the
 * function enqueues gas stage -> compute_opacity -> reflected || thermal (+ fused disk sums) on every block's
 * context(s) and returns; picaso_toon_spectrum_collect copies a leg's results into the caller's full-grid host
 * arrays at [col0, col0 + nwno).  They only chain the entry points above: same results as calling them one by one. */
typedef struct picaso_block {
    picaso_ctx *ctx, *tctx;                /* tctx: the thermal leg's context (its own stream), NULL = ctx */
    int nwno;                              /* columns of this block */
    long col0;                             /* first column of the block in the full grid (host arrays below) */
    const double *const *mol_tabs, *const *cont_tabs, *const *ray_tabs;   /* this block's resident tables (device) */
    const double *cld_opd, *cld_w0, *cld_g0;           /* device cloud planes (nlayer, nwno) or NULL: no cloud */
    const double *cld_host_opd, *cld_host_w0, *cld_host_g0;   /* OR full-grid HOST planes (nlayer, cld_host_pitch) ... */
    long cld_host_pitch;
    double *cld_work_opd, *cld_work_w0, *cld_work_g0;  /* ... copied, this block's columns, into these (nlayer, nwno) */
    const double *raman;                   /* device Raman factor: plane, one row (job.raman_rows = 0) or NULL */
    const double *surf_reflect, *F0PI, *wno;           /* device (nwno) */
    double *taugas, *tauray;               /* workspace (nlayer, nwno) */
    double *planes[13];                    /* compute_opacity outputs in the order of picaso_compute_opacity_ck_dev;
                                              NULL = not written */
    const double *refl_planes[11];         /* what get_reflected_1d reads, its argument order (may alias planes[]
                                              or constant planes: a cloud-free atmosphere writes three) */
    const double *th_dtau, *th_w0, *th_cosb;           /* what get_thermal_1d reads */
    double *xint, *albedo, *flux, *disk;   /* device results: (numg,numt,nwno), (nwno), (numg,numt,nwno), (nwno) */
    double *albedo_host, *thermal_host;    /* full-grid host results (or NULL: leave them on the device) */
    /* The spectrum-wide integrals of justdoit.py:552-599 on the device (picaso_trapz_dev), for ONE block that covers the
     * grid: trapz_d = diff(1/wno), trapz_dr = diff(1/wno[::-1]) (nwno - 1 each), stellar (nwno), all device, or NULL.
     * With trapz_d set, albedo holds nwno + 1 doubles and [nwno] = np.trapz(x=1/wno, y=albedo*stellar); with trapz_dr,
     * disk[nwno] = np.trapz(x=1/wno[::-1], y=thermal[::-1]); collect then copies nwno + 1 doubles into the host array. */
    const double *trapz_d, *trapz_dr, *stellar;
    /* Pinned staging blocks of the caller (picaso_host_alloc, nwno + 1 doubles each) or NULL.  With them the copies of
     * a leg's results are put on its stream by picaso_toon_spectrum_blocks itself, right behind its last kernel
     * (picaso_memcpy_d2h_async; the marks below are written by that call), and picaso_toon_spectrum_collect only
     * waits for the mark and moves the block into the host array: a leg that finishes early is on the host early,
     * whatever order the legs are collected in.  Without them collect issues a synchronous copy. */
    double *albedo_pin, *thermal_pin;
    void *albedo_mark, *thermal_mark;
    /* cloud tables on their own grid (device): cld_tab_nin points cld_tab_xp, rows cld_tab_fp (3 nlayer, cld_tab_nin);
     * interpolated to `wno` inside the fused opacity launch (picaso_gas_compute_opacity_dev); 0 = none */
    int cld_tab_nin;
    const double *cld_tab_xp, *cld_tab_fp;
} picaso_block;
typedef struct picaso_spectrum_job {
    int nlayer;
    int mol_mode, nmol, cont_interp, ncont, nray;      /* as picaso_opacity_gas_ck_dev */
    const int *mol_rows;
    const double *mol_wts, *mol_fac;
    const int *cont_rows;
    const double *cont_wts, *cont_fac, *ray_fac;
    int raman_rows;
    double raman_const;
    int test_mode, delta_eddington, stream;            /* as picaso_compute_opacity_ck_dev */
    int do_reflected, do_thermal;
    int numg, numt;
    const double *ubar0, *ubar1;           /* host (numg, numt) */
    double cos_theta;
    const double *gweight, *tweight;       /* host: the fused disk sums */
    int single_phase, multi_phase, toon_coefficients;
    double frac_a, frac_b, frac_c, constant_back, constant_forward, b_top;
    const double *tlevel, *plevel;         /* host (nlevel) */
    int hard_surface;
    /* rt_method = 1: the spherical-harmonics solvers instead of the Toon ones (reference justdoit.py:259-269, 364-370):
     * picaso_get_reflected_SH_top_dev / picaso_get_thermal_SH_dev with `stream` above (2 or 4), flx = 0 and the
     * reference's per-angle f_deltaM compounding.  The block's refl_planes then hold the SH argument order -- dtau, tau,
     * w0, cosb, ftau_cld, ftau_ray, f_deltaM, dtau_og, tau_og, w0_og, cosb_og (NULL where that entry point takes NULL) --
     * and th_dtau / th_w0 / th_cosb what get_thermal_SH reads (dtau, w0, cosb_og).  0: Toon. */
    int rt_method;
    int sh_w_single_form, sh_w_multi_form, sh_psingle_form, sh_w_single_rayleigh, sh_w_multi_rayleigh,
        sh_psingle_rayleigh, sh_single_form;
    int sh_cloud_free_above;               /* as picaso_get_reflected_SH_top_dev (the same value for every block) */
    /* nfacets > 0: a 3-D spectrum (reference justdoit.py:407-516: one atmosphere per (gangle, tangle) facet) on
     * FACET-MAJOR planes.  nfacets = numg * numt; the per-layer tables above hold nfacets * nlayer rows (the tall
     * atmosphere of all facets: facet f's layers at [f nlayer, (f + 1) nlayer)) and ONE picaso_gas_compute_opacity_dev
     * launch over them writes the block's planes as (nfacets, nlayer, nwno); tlevel / plevel are host (nfacets, nlevel);
     * cloud input: none, or tables on their own grid with 3 nfacets nlayer rows (cld_tab_*).  The solvers are
     * picaso_get_reflected_3d_batch_dev / picaso_get_thermal_3d_batch_dev with every facet as a spectrum of one facet
     * (its slab of the planes, its ubar0 / ubar1), then picaso_compress_disco_dev / picaso_compress_thermal_dev.  The level
     * planes (planes[1], planes[8]: tau, tau_og) must be NULL -- the 3-D kernels form them as running sums.  Toon only. */
    int nfacets;
    /* ngauss > 1: premixed correlated-k tables (reference justdoit.py:256-313, 328-380: the Gauss-point loop around the Toon
     * solvers).  The block's molecular table is ONE table of nwno * ngauss columns (mol_mode = 2, nmol = 1), continua are
     * interpolated (cont_interp = 1, cont_rows / cont_wts [ncont][nlayer][2]), planes / taugas are (rows, nwno, ngauss) with
     * the Gauss index fastest, tauray and the cloud planes (rows, nwno); the opacity stage is picaso_opacity_gas_ck_dev +
     * picaso_compute_opacity_ck_dev, the legs picaso_get_reflected_1d_ck_dev / picaso_get_thermal_1d_ck_dev with
     * gauss_wts (host, ngauss).  0 or 1: monochromatic.  Toon, 1-D only. */
    int ngauss;
    const double *gauss_wts;
} picaso_spectrum_job;
int picaso_toon_spectrum_blocks(int nblocks, picaso_block *blocks, const picaso_spectrum_job *job);
/* picaso_toon_spectrum_blocks in two calls (1-D blocks, enqueued one after the other): phase = 1 the opacity stage alone --
 * it reads nlayer, ngauss, the mol_ / cont_ / ray_ fields, raman_rows / raman_const, test_mode, delta_eddington, stream,
 * do_thermal and nfacets (= 0) of the job and ctx / tctx, the tables, planes, cloud inputs and raman of the blocks, nothing
 * else -- phase = 2 the legs, integrals and result copies.  The caller fills the other half of job and blocks in between,
 * while the gas kernel runs.  Same launches, same order, same streams as the one call: same bits. */
int picaso_toon_spectrum_phase(int nblocks, picaso_block *blocks, const picaso_spectrum_job *job, int phase);
/* copy one leg's results (which = 1: albedo, 2: thermal flux) of every block into albedo_host / thermal_host */
int picaso_toon_spectrum_collect(int nblocks, picaso_block *blocks, int which);
/* wait for and drop result copies that will not be collected (the caller failed in between) */
int picaso_toon_spectrum_abandon(int nblocks, picaso_block *blocks);
/* sizeof(picaso_block), sizeof(picaso_spectrum_job) and two member offsets as compiled (layout check of a binding) */
int picaso_driver_abi(size_t *block_bytes, size_t *job_bytes, size_t *off_albedo_host, size_t *off_hard_surface);

/* ---- host-side set-up of one 1-D spectrum in one call (no GPU work) ------------------------------------------------
 * ATMSETUP's level / layer state (reference atmsetup.py:74-461), the table rows and weights of
 * RetrieveOpacities.get_opacities (optics.py:2048-2123, 2241-2306) and the per-layer coefficients of the TAUGAS / TAURAY
 * sums (optics.py:144-277), bit for bit what the Python mirror computes with ~150 numpy calls.  Scope: strictly
 * increasing pressures, 'linear' interpolation, molecule-pair continua.  Returns 0, or > 0 when the
 * profile is outside that scope (the caller then takes the mirror), < 0 never.  Every pointer is host memory. */
typedef struct picaso_setup_args {
    int nlevel, nmol;                      /* nmol: the recognised molecules of the profile, column order */
    const double *pressure_bar, *temperature;          /* (nlevel) */
    const double *const *mix;              /* nmol x (nlevel) level mixing ratios */
    const double *weights;                 /* (nmol) molecular weights */
    double gravity, radius, GM, p_reference_bar;       /* planet.gravity (cgs); planet.radius (NaN: gravity constant with height,
                                                          else G M / z^2 with GM = G * planet.mass); approx p_reference */
    double pconv, k_b, amu;
    double coef1_scale, coef1_den;         /* rgas * 273.15**2 * .5E5 and 1.01325**2 * gravity/100, as Python forms them */
    /* functions of the pressure grid only, from numpy (kept by the caller while the grid is unchanged) */
    const double *log_pratio;              /* (nlevel - 1) np.log(P[k+1] / P[k]), P = pressure_bar * pconv */
    const double *log10_player;            /* (nlayer) np.log10(sqrt(P[k+1] P[k]) / pconv) */
    const double *pbar_cubed_hi, *pbar_cubed_lo;       /* (nlayer) (P / pconv)[1:] ** 3 and (P / pconv)[:-1] ** 3 */
    /* the opacity object's (P, T) grid */
    int nt, npg;
    const double *t_inv_grid, *p_log_grid;
    const long *nc_p, *row_lut;
    int nlut, ncia_t;
    const double *cia_temps;               /* np.unique(cia_temps) */
    int nopa, ncont, nray;
    const int *opa_idx, *cont_a, *cont_b, *ray_idx;    /* indices into the nmol list */
    /* outputs */
    double *level_pressure, *level_mmw, *level_den, *z, *dz, *scale_height;               /* (nlevel) */
    double *layer_temperature, *layer_pressure, *layer_mmw, *layer_gravity, *colden;       /* (nlayer) */
    double *layer_mix;                     /* (nmol, nlayer) */
    int *rows;                             /* (nopa, nlayer, 4) */
    double *wts;                           /* (nopa, nlayer, 4) */
    int *cia_rows;                         /* (max(ncont, 1), nlayer) */
    double *mol_fac, *cont_fac, *ray_fac;  /* (nopa | ncont | nray, nlayer) */
    int *pt_opa_index, *n_pt_opa_index;    /* (4 nlayer) sorted unique ptids used, and how many */
    double *scratch;                       /* (3 nlevel) */
    /* premixed correlated-k tables (reference RetrieveCKs.get_pre_mix_ck / get_continuum, optics.py:1081-1161, 1398-1498):
     * the table search above is theirs as well (get_mixing_indices, :1200-1278), with these differences, switched on here:
     * premixed = 1: nopa = 1, rows = p * nt + t of the four neighbours (row_lut is not read, pt_opa_index not written),
     * mol_fac[0] = colden / mmw (no mixing ratio, optics.py:256-262);
     * cont_interp = 1: besides the nearest row, the BRACKETING continuum temperatures and the 1/T weight of every layer
     * (:1411-1428, 1474-1478): cia_rows2 (nlayer, 2) = lo, lo + 1 and cia_wts2 (nlayer, 2) = 1 - ti, ti with lo the last
     * temperature <= T_layer clipped to [0, ncia_t - 2], ti = (1/T - 1/t_lo) / (1/t_hi - 1/t_lo).  0 / NULL: as before. */
    int premixed, cont_interp;
    int *cia_rows2;
    double *cia_wts2;
} picaso_setup_args;
int picaso_host_setup(const picaso_setup_args *args);
/* the facets of a 3-D spectrum in one call: facet f reads temperature + f t_stride and mix[m] + f mix_stride[m] (0: one
 * column for all facets) and writes behind facet f - 1 in every output array (facet-major) */
int picaso_host_setup_facets(const picaso_setup_args *args, int nfacets, long t_stride, const long *mix_stride);
size_t picaso_host_setup_abi(void);

#ifdef __cplusplus
}
#endif
#endif /* PICASO_HIP_H */
