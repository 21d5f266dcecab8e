// Internal header shared by the gfx950 translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <initializer_list>

#include <unordered_map>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/picaso_hip.h"

// Round 5: the layer body's arithmetic regrouped for fewer instructions (7 per angle-layer and 2 per layer of ~479 per
// five-angle layer).  The VALUES are the same expressions; their rounding is not, so results differ from rounds 1-4 in the
// last bits (observed <= 2e-13; every launch shape, shard and batch member still agrees bit for bit because every kernel
// takes the same regrouped form, toon_reflected_coop.hip included).  0 = the round-4 operation order (A/B).
//   (1) the beam terms at the bottom of a layer from those at its top (c-dn = c-up e0, c+dn = c+up e0) where the beam
//       exponential is carried as a running product anyway;
//   (2) the common factors of the two mode integrals multiplied once: T/(lu^2 - 1) w0/2pi (1 + Gamma) B0;
//   (3) the layer source S0 = t2 (ssa eo + (w0/2pi A0) B0 fx) where the layer is not delta-scaled;
//   (4) the per-angle reciprocal 1/(den (lu - 1)(lu + 1)) with ONE Newton step (2^-46): it is a common FACTOR of every
//       beam term of the layer (and of the mode integrals), so its error scales the layer's particular solution and the
//       homogeneous response to it alike -- the near-singular cancellation at lambda u0 -> 1 is untouched (checked
//       against the x87 oracle on the headline scene's worst columns, tools/headline_error_x87.py);
//   (5) 1/g2 and 1/EP from one reciprocal of their product.
#ifndef PZ_REFL_DIET
#define PZ_REFL_DIET 1
#endif

namespace pz { struct PairwisePlan; }        // integrals.hip: the summation tree of one array length

struct picaso_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t wait_ev = nullptr;          // recorded on this stream for picaso_ctx_wait
    char err[512] = {0};
    int ncu = 256;
    // staging arena for the host-pointer entry points (grown on demand, reused across calls)
    char *arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    // ring of small pinned-host/device slots for geometry / profile tables handed over as host
    // pointers by the *_dev entry points (which return before the stream has consumed them)
    static constexpr int NSLOT = 8;
    static constexpr size_t SLOT_BYTES = 4u << 20;      // a batched 3-D launch carries every spectrum's level tables
    char *ring_h = nullptr, *ring_d = nullptr;
    hipEvent_t ring_ev[NSLOT] = {};
    bool ring_pending[NSLOT] = {};
    int ring_next = 0;
    // ... and a second ring of many small slots for the tables of 1-D spectra (rows / weights / coefficients of one
    // spectrum: ~30 KB; level temperatures; batch items): a chunk of a retrieval uploads a dozen of them back to back, and
    // with eight slots the host waited for the GPU to consume the chunk's first tables before it could enqueue its last
    static constexpr int NSMALL = 64;
    static constexpr size_t SMALL_BYTES = 64u << 10;
    char *small_h = nullptr, *small_d = nullptr;
    hipEvent_t small_ev[NSMALL] = {};
    bool small_pending[NSMALL] = {};
    int small_next = 0;
    // pinned bounce buffer (two halves) for host<->device copies of small and medium arrays, so the
    // caller's pageable memory is never pinned by the runtime (see picaso_memcpy_h2d)
    static constexpr size_t STAGE_BYTES = 8u << 20;
    static constexpr size_t STAGE_MAX_COPY = 64u << 20;
    char *stage = nullptr;
    hipEvent_t stage_ev[2] = {};
    // per-layer sweep state of the level-flux (two-sweep) kernels: 4 planes (nlayer, ncol)
    double *lvl_scratch = nullptr;
    size_t lvl_scratch_bytes = 0;
    double *ck_scratch = nullptr;          // per-column results of a correlated-k batch before the Gauss sum
    size_t ck_scratch_bytes = 0;
    // picaso_dev_malloc / picaso_dev_free keep released blocks for reuse (hipFree synchronises the
    // device and costs ~0.5 ms; a spectrum() call releases a few dozen planes).  Reuse is safe
    // without a synchronisation because every access to library-owned memory is ordered on `stream`.
    std::unordered_multimap<size_t, void *> pool;      // size -> free block
    std::unordered_map<void *, size_t> live;           // block -> size
    size_t pool_bytes = 0;
    static constexpr size_t POOL_CAP = 64ull << 30;    // cached (free) bytes kept at most
    // pinned host blocks for result copies that do not wait (picaso_host_alloc / picaso_memcpy_d2h_async), kept for
    // reuse like the device blocks, and the events that mark those copies (picaso_mark_wait)
    std::unordered_multimap<size_t, void *> host_pool;
    std::unordered_map<void *, size_t> host_live;
    std::vector<hipEvent_t> marks_free;
    std::unordered_map<long, pz::PairwisePlan *> pairwise_plans;      // picaso_trapz_dev, by number of summands
};

namespace pz {

extern thread_local char g_err[512];

int fail(picaso_ctx *ctx, const char *fmt, ...);
void free_pairwise_plans(picaso_ctx *ctx);

#define PZ_HIP(ctx, expr)                                                                    \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            /* the runtime keeps a failed call as its "last error": drop it, or the hipGetLastError() behind the NEXT  \
             * call's kernel launch reports this failure again and one bad call poisons every later one */           \
            (void)hipGetLastError();                                                         \
            return pz::fail(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                            __FILE__, __LINE__);                                             \
        }                                                                                    \
    } while (0)

// required pointer arguments of an entry point: PZ_NEED(ctx, "who", a, b, c) fails cleanly on the first NULL
inline bool any_null(std::initializer_list<const void *> ps)
{
    for (const void *p : ps)
        if (!p) return true;
    return false;
}
#define PZ_NEED(ctx, who, ...)                                                                   \
    do {                                                                                         \
        if (pz::any_null({__VA_ARGS__})) return pz::fail(ctx, "%s: a required array argument is NULL (%s)", who, #__VA_ARGS__); \
    } while (0)

#define PZ_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != 0) return rc__; \
    } while (0)

// Bump allocator over the context arena: reset at the start of every host-pointer call.
int arena_reset(picaso_ctx *ctx, size_t need_bytes);
void *arena_take(picaso_ctx *ctx, size_t bytes);

template <typename T>
inline int arena_upload(picaso_ctx *ctx, const T *host, size_t count, const T **dev)
{
    if (!host && count) return fail(ctx, "a required input array is NULL");
    T *d = static_cast<T *>(arena_take(ctx, count * sizeof(T)));
    if (!d) return fail(ctx, "arena exhausted");
    PZ_HIP(ctx, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *dev = d;
    return 0;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// launchers implemented in the kernel translation units
// ---------------------------------------------------------------------------------------------
constexpr int MAX_ANGLES = 8;   // angles fused per launch (per-lane register state)

constexpr int MAX_CK_GAUSS = 32;   // Gauss points of a correlated-k table (reference tables use 8 or 20)

struct ReflectedArgs {
    int nlayer;
    long ncol;          // columns: nwno*ncolper (1-D) or nwno*nfac (3-D)
    int ncolper;        // 1-D only: columns per wavelength (correlated-k Gauss points, fastest axis)
    long pitch;         // elements between consecutive layers of a plane
    int nfac;           // 1 for 1-D; numg*numt for 3-D (facet index fastest in memory)
    int nwno;
    const double *dtau, *tau, *w0, *cosb, *gcos2, *ftau_cld, *ftau_ray, *dtau_og, *tau_og, *w0_og,
        *cosb_og;
    const double *surf_reflect, *F0PI;      // (nwno)
    double cos_theta;
    int single_phase, multi_phase, toon_coefficients;
    double frac_a, frac_b, frac_c, constant_back, constant_forward, b_top;
    // 1-D: angles of this launch (shared planes).  3-D: device tables (nfac) of |ubar|.
    // Per-angle constants are derived on the host so they arrive as wave-uniform SGPR values (fp64
    // has no scalar ALU: computing 1/u in the kernel parks uniform values in VGPRs), and are kept
    // together per angle so that one s_load_dwordx16 fetches everything an angle block needs.
    int na;            // angles carried per lane (template NA of the launch)
    int ny;            // grid.y: angle groups of `na` running as separate waves (small problems)
    int nvalid;        // ny > 1: angles [0, nvalid) of the ny*na slots are real, the rest pad the last group
    int ncg;           // ny > 1: column groups (workgroups along the wavelength axis); set by the launcher
    struct Angle {
        double u1, iu0, iu0sq, nl1, q2;     // used by the symmetric-geometry (ubar0 == ubar1) kernel
        double u0, nl0, nlm, wq2, wgt, wgt2;
        // iu0sq = 1/(u0 u0) as the reference forms it (fluxes.py:1155); nl0/nl1/nlm = -log2(e)/u0,
        // -log2(e)/u1, -log2(e)(1/u0 + 1/u1): exp(-x/u) = 2^(x nl); wq2 = 2 u0/(u0+u1);
        // q2 = (3 ubar2^2 u1^2 - 1)/2, ubar2 = 0.767 (fluxes.py:1280); wgt, wgt2 = gweight[g], tweight[t]
    } ang[MAX_ANGLES];
    const double *u0_tab, *u1_tab;          // 3-D
    double *xint;                           // 1-D: this launch's first angle row, (na, nwno); 3-D: (nfac, nwno)
    double *albedo;                         // nullable; 1-D fused compress_disco accumulator
    double albedo_scale;                    // sym_fac*0.5*(cos_theta+1)
    int albedo_first, albedo_last;          // first chunk initialises, last chunk finalises (/F0PI*scale)
    // Batched launch (picaso_get_reflected_1d_batch_dev): `nspec` spectra of the same shape and options in ONE
    // grid.  `batch` is a device table with one entry per spectrum (its planes, outputs and geometry); the
    // members above that the entry also holds are ignored.  Every spectrum keeps whole workgroups (bps blocks
    // each), so a column meets the same wave-mates as in a single launch.
    const struct ReflBatchItem *batch;      // nullptr: single spectrum
    int nspec;
    unsigned bps;                           // workgroups per spectrum (set by the launcher)
    int batch_zp;                           // every angle of every spectrum has ubar0 == ubar1 (the table is on the device)
    int batch_interleave;                   // all spectra read the SAME planes (geometries differ): the workgroups
                                            // of one column group go to one XCD, like angle groups (ny)
};
struct ReflBatchItem {
    const double *dtau, *tau, *w0, *cosb, *gcos2, *ftau_cld, *ftau_ray, *dtau_og, *tau_og, *w0_og, *cosb_og;
    const double *surf_reflect, *F0PI;
    double *xint, *albedo;
    double cos_theta;
    const double *u0_tab, *u1_tab;          // 3-D: this spectrum's (nfac) tables of ubar0, ubar1
    ReflectedArgs::Angle ang[MAX_ANGLES];   // 1-D: this launch chunk's angles of this spectrum
};
int launch_reflected_toa(picaso_ctx *ctx, const ReflectedArgs &a, bool is3d);
// cooperative kernel for small 1-D launches (toon_reflected_coop.hip): one workgroup per 64 columns, a wave for the
// angle-independent layer quantities and one wave per disk angle; a.ang[0 .. a.na) = ALL angles, a.ny = 1
bool reflected_coop_ok(const ReflectedArgs &a);
int reflected_coop_pattern(const ReflectedArgs &a);
int launch_reflected_coop(picaso_ctx *ctx, const ReflectedArgs &a);
// out[r][w] = sum_j wts[j] in[r][w*n + j]  (correlated-k Gauss-point sum, justdoit.py:307, 380)
int launch_weighted_colsum(picaso_ctx *ctx, int nrows, long nwno, int n, const double *wts_host,
                           const double *in, double *out);
// out = a x + b y (patchy-cloud blend, justdoit.py:300-305)
int launch_axpby(picaso_ctx *ctx, size_t n, double a, const double *x, double b, const double *y, double *out);


struct ThermalArgs {
    int nlayer;
    long ncol, pitch;
    int nfac, nwno;
    int ncolper;                            // 1-D only: columns per wavelength (correlated-k Gauss points)
    const double *wno, *dwno;               // (nwno)
    const double *tlevel, *plevel;          // device: (nlevel) for 1-D, (nlevel,nfac) for 3-D
    const double *dtau, *w0, *cosb;
    const double *surf_reflect;
    int hard_surface, calc_type;
    int na, ny;                             // angles per lane, angle groups in grid.y (see ReflectedArgs)
    double u1[MAX_ANGLES], wgt[MAX_ANGLES], wgt2[MAX_ANGLES];   // wgt = gweight[g], wgt2 = tweight[t]
    const double *u1_tab;                   // 3-D
    double *flux;                           // (na,nwno) / (nfac,nwno)
    double *disk;                           // nullable fused compress_thermal accumulator
    double disk_scale;
    int disk_first, disk_last;
    // batched launch (picaso_get_thermal_1d_batch_dev): grid.z = spectrum, see ReflectedArgs::batch
    const struct ThermalBatchItem *batch;
    int nspec;
    // level-flux kernels only (picaso_get_thermal_1d_ck_tbatch_dev): ONE set of planes under `ncol / per_item`
    // temperature profiles -- the columns are (profile, wavelength, Gauss point), a column reads the planes at
    // column % per_item and the level temperatures of profile column / per_item (tlevel: (nprofile, nlevel)).  0: off
    long per_item;
};
struct ThermalBatchItem {
    const double *dtau, *w0, *cosb, *surf_reflect, *tlevel, *plevel;
    double *flux, *disk;
    const double *u1_tab;                   // 3-D: (nfac) table of ubar1
    double u1[MAX_ANGLES];                  // 1-D: this launch chunk's angles of this spectrum
};
int launch_thermal_toa(picaso_ctx *ctx, const ThermalArgs &a, bool is3d);
// cooperative kernel for small 1-D launches (toon_thermal.hip): helper waves + a sweeper wave per 64 columns
bool thermal_coop_ok(const ThermalArgs &a);
int launch_thermal_coop(picaso_ctx *ctx, const ThermalArgs &a, const double *tlevel_host, const double *plevel_host);

// level-flux (two-sweep) variants, toon_lvl.hip.  One angle per launch for reflected light
// (its right-hand side is per angle); all angles inside one launch for thermal (one solve).
struct ReflectedLvlArgs {
    ReflectedArgs base;                     // na == 1: u0[0], u1[0]
    int want_toa;                           // also write base.xint (get_toa_intensity)
    double *fm, *fp, *fmm, *fpm;            // this angle's (nlevel, nwno) output planes
    double *scratch;                        // 4 planes (nlayer, nwno): rho, delta, s, t
};
int launch_reflected_lvl(picaso_ctx *ctx, const ReflectedLvlArgs &a);

struct ThermalLvlArgs {
    ThermalArgs base;
    int nang;
    const double *u1_dev;                   // (nang) device table
    double *flux;                           // (nang, nwno)
    double *fm, *fp, *fmm, *fpm;            // (nang, nlevel, nwno)
    double *scratch;                        // 4 planes (nlayer, nwno)
};
int launch_thermal_lvl(picaso_ctx *ctx, const ThermalLvlArgs &a);
int lvl_scratch_reserve(picaso_ctx *ctx, size_t bytes);
int ck_scratch_reserve(picaso_ctx *ctx, size_t bytes);

// disk integration (disco.hip): `mode` says how the (g,t) sum is finished
enum { COMPRESS_DISCO = 0, COMPRESS_THERMAL = 1 };
// the same with up to 128 host weight pairs carried in the kernel arguments (no table upload)
int launch_compress_hostw(picaso_ctx *ctx, size_t ninner, const double *x, const double *wts_host, int nang,
                          const double *F0PI, int mode, double c1, double c2, double *out);
int launch_compress_dev(picaso_ctx *ctx, size_t ninner, const double *x, const double *wts_dev,
                        int nang, const double *F0PI, int mode, double c1, double c2, double *out);

// copy a small host table into the next ring slot; *dev is valid for kernels enqueued afterwards
int table_upload(picaso_ctx *ctx, const void *host, size_t bytes, const void **dev);

}  // namespace pz
