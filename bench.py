#!/usr/bin/env python
"""Headline benchmark: spectra/s of ONE 1e5-wavelength x 90-layer Toon reflected-light spectrum
(BASELINE.json configs[2]; 5 Gauss angles, TTHG_ray + N=2 + delta-Eddington, fused disk
integration) with all input planes resident in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--config 1|2|3|4] [--scaling strong|weak]

One process per GPU.  `python bench.py --gpus N` starts its own N ranks (spawn_ranks below: plain
subprocesses, each with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT and a random job token in its
environment; rank 0's stdout is this process's stdout) and exits non-zero, saying so, when fewer than N GPUs are
visible.  Any other one-process-per-GPU launcher that sets the same variables works as well (the driver's
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
--gpus N ...`: only its environment is used; nothing here imports PyTorch).  One "step" = one full spectrum: every rank solves its contiguous wavelength
block of the SAME 1e5-point grid (strong scaling, BASELINE's metric) and the albedo shards are
all-gathered inside the timed region by RCCL inside libpicaso_hip.so (picaso_all_gather_multi_async_dev: spectra
are handed to RCCL in batches of --gather-every, one collective launch per batch on the communicator's own stream
behind the kernels that produced them, overlapping the solves of the next batch; each spectrum lands contiguous in
its own buffer and every gather has finished when the timed region ends).  `--scaling weak` gives every rank its own 1e5-point block of an N x 1e5 grid
instead.  Prints ONE JSON line (rank 0).

--config selects the other BASELINE workloads (not the headline): 1 thermal emission 1e4 x 90,
3 SH4 reflected 1e5 x 90, 4 3-D 8x8 facets x 90 x nwno (default 12 500 per GPU: 1e5 over 8 GPUs).  With --gpus N they
shard, gather and report exactly like the headline: `per_rank` (each rank's columns, solve ms, gather ms) and `checks`
(gathered == local shard; gathered == the unsharded spectrum solved on rank 0, bit for bit).

Clock ramp: an idle MI355X takes ~100 launches (30 ms) of this kernel to reach its steady clock
state (tools/refl_time.py --ramp: 0.53, 0.35, 0.32, 0.30, 0.29 ... 0.25 ms per launch in groups of
ten from cold) and falls back within milliseconds of idling, so the W warm-up steps are preceded by
--prewarm-ms of the same launches, untimed; the timed region is exactly K steps between two
barrier + synchronise pairs.
"""
import argparse
import hashlib
import json
import os
import secrets
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
    # the image exports NCCL_DEBUG=VERSION: librccl would print a version banner on stdout next to
    # the one JSON line this script owes its caller
    del os.environ["NCCL_DEBUG"]

from picaso_amd import _lib, device, disco, resident, sharding  # noqa: E402
from picaso_amd import synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)   # frac_a, frac_b, frac_c, constant_back, constant_forward


def kernel_source_hash():
    """Hash of the kernel sources the committed PMC numbers (profiles/traffic.json) refer to."""
    h = hashlib.sha1()
    for f in ("toon_reflected.hip", "device_math.hpp", "common.hpp"):
        with open(os.path.join(ROOT, "picaso_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def live_traffic(kernel_tag="k_reflected_toa<5", timeout_s=150.0):
    """HBM bytes per launch of the headline kernel from the PMC counters, measured NOW: two short child runs of this
    script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes: both do not fit one TCC pass;
    --kernel-trace only, no other trace domain), (2 * FETCH_SIZE + WRITE_SIZE) KB per dispatch -- FETCH_SIZE doubled per
    the gfx950 correction of MI355X_MICROARCH.md (HBM / rocprofv3).  Returns (bytes, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_CTOR",
                                     "ROCPROF_OUTPUT_PATH")) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--prewarm-ms", "0",
             "--cpu-sample", "0", "--steady-steps", "0", "--secondary", "0", "--product", "0", "--traffic", "off"]
    env = dict(os.environ, TMPDIR="/tmp")
    means = {}
    t0 = time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="picaso_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--output-format", "csv", "--pmc", ctr, "-d", out, "-o", "pmc",
                                "--"] + child, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=timeout_s)
            n, tot = 0, 0.0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == ctr and kernel_tag in row.get("Kernel_Name", ""):
                        n += 1
                        tot += float(row.get("Counter_Value", 0))
            if not n:
                return None, "no %s rows for %s (rocprofv3 rc %d)" % (ctr, kernel_tag, r.returncode)
            means[ctr] = (tot / n, n)
        except Exception as e:       # a profiler that is missing a counter, hangs or dies must not take the bench line with it
            return None, "%s pass failed: %s" % (ctr, str(e)[:200])
        finally:
            shutil.rmtree(out, ignore_errors=True)
    nbytes = (2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0]) * 1024.0
    return nbytes, ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of a child "
                    "`bench.py --steps 3` (%d and %d dispatches of the kernel, %.0f s), (2 x FETCH_SIZE + WRITE_SIZE) KB per "
                    "dispatch, FETCH_SIZE doubled per the gfx950 correction"
                    % (means["FETCH_SIZE"][1], means["WRITE_SIZE"][1], time.perf_counter() - t0))


# ---------------------------------------------------------------------------------------------
# workloads: each returns a dict with solve(), the local result DeviceArray, algorithmic bytes of
# the local launch, and how to check the result against the CPU oracle
# ---------------------------------------------------------------------------------------------
def workload_reflected(ctx, args, lo, hi, seed, nwno_total, scene=None):
    """configs[2]: get_reflected_1d + compress_disco (9 layer planes + 2 level planes +
    F0PI + surf_reflect read once, xint_at_top and albedo written once)."""
    nlayer, nlevel, ng = args.nlayer, args.nlayer + 1, args.ngauss
    n = hi - lo
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    ubar0, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    if scene is None:
        scene = syn.make_scene(nlayer, nwno_total, seed=seed)
    scene["F0PI"] = np.ones(nwno_total)
    scene["surf_reflect"] = np.zeros(nwno_total)
    keys = resident.REFLECTED_PLANES + ("F0PI", "surf_reflect")
    d = resident.upload_scene(scene, keys, lo, hi, ctx=ctx)
    xint = device.DeviceArray((ng, 1, n), ctx)

    def solve(albedo):
        resident.reflected_1d(ctx, nlevel, n, ng, 1, d, d["surf_reflect"], ubar0, ubar1, 1.0, d["F0PI"], 3, 0,
                              *TTHG, xint, toon_coefficients=0, b_top=0.0, gweight=gw, tweight=tw, albedo=albedo)

    def oracle(sl):
        from oracle import oracle as orc
        ns = sl.stop - sl.start
        planes = [np.ascontiguousarray(scene[k][:, sl]) for k in resident.REFLECTED_PLANES]
        xo, _ = orc.get_reflected_1d(nlevel, scene["wno"][sl], ns, ng, 1, *planes, 0.0, ubar0, ubar1, 1.0,
                                     np.ones(ns), 3, 0, *TTHG)
        return orc.compress_disco(ns, 1.0, xo, gw, tw, np.ones(ns))

    plane_sets = [d]

    def solve_batch(B, on=None):
        """B plane sets in ONE launch (picaso_get_reflected_1d_batch_dev): set 0 is the headline's, sets 1..B-1 are
        copies of it in their own HBM allocations (B x 0.8 GB read per launch; generating B distinct 1e5 x 90
        scenes on the host would take the bench tens of seconds).  `on`: the context (stream) the launch goes to.
        Returns (launch, per-spectrum albedo arrays)."""
        import ctypes as _ct
        while len(plane_sets) < B:
            c = {}
            for k in keys:
                c[k] = device.DeviceArray(d[k].shape, ctx)
                _lib.check(_lib.load().picaso_memcpy_d2d(ctx, _ct.c_void_p(c[k].addr), _ct.c_void_p(d[k].addr),
                                                         _ct.c_size_t(d[k].nbytes)), ctx)
            plane_sets.append(c)
        device.sync(ctx)
        on = ctx if on is None else on
        sets = plane_sets[:B]
        xs = [device.DeviceArray((ng, 1, n), on) for _ in range(B)]
        albs = [device.DeviceArray((n,), on) for _ in range(B)]

        def launch():
            resident.reflected_1d_batch(on, nlevel, n, ng, 1, sets, [x["surf_reflect"] for x in sets], ubar0, ubar1, 1.0,
                                        [x["F0PI"] for x in sets], 3, 0, *TTHG, xs, gweight=gw, tweight=tw, albedo=albs)
        return launch, albs

    def solve_batch_distinct(B):
        """B DIFFERENT atmospheres in one launch: what a retrieval's batch is.  Member 0 is the headline scene; the others
        are drawn afresh (own gas field, own cloud) with the cloud slab moved up or down by 6 layers per member and every
        third member cloud-free, so the per-wave shortcut flags of the layer body (cloud in this layer or not, delta-scaled
        or not) differ between the members of one launch.  Returns (launch, albedo arrays, the same members one by one)."""
        sets = [d]
        for m in range(1, B):
            comps = list(syn.tau_components(nlayer, n, syn.BASE_SEED + seed + 100 + m, cloud=(m % 3 != 2),
                                            gas_scale=1.0 + 0.15 * m))
            shift = 6 * ((m + 1) // 2) * (1 if m % 2 else -1)
            comps[2:] = [np.roll(c, shift, axis=0) for c in comps[2:]]
            sc = syn.mix_planes(*comps)
            sc["F0PI"], sc["surf_reflect"] = scene["F0PI"][lo:hi], scene["surf_reflect"][lo:hi]
            sets.append(resident.upload_scene(sc, keys, ctx=ctx))
        xs = [device.DeviceArray((ng, 1, n), ctx) for _ in range(B)]
        albs = [device.DeviceArray((n,), ctx) for _ in range(B)]

        def launch():
            resident.reflected_1d_batch(ctx, nlevel, n, ng, 1, sets, [x["surf_reflect"] for x in sets], ubar0, ubar1, 1.0,
                                        [x["F0PI"] for x in sets], 3, 0, *TTHG, xs, gweight=gw, tweight=tw, albedo=albs)

        def singles():
            out = []
            x1, a1 = device.DeviceArray((ng, 1, n), ctx), device.DeviceArray((n,), ctx)
            for x in sets:
                resident.reflected_1d(ctx, nlevel, n, ng, 1, x, x["surf_reflect"], ubar0, ubar1, 1.0, x["F0PI"], 3, 0,
                                      *TTHG, x1, toon_coefficients=0, b_top=0.0, gweight=gw, tweight=tw, albedo=a1)
                device.sync(ctx)
                out.append(a1.to_host())
            return out
        return launch, albs, singles

    return dict(solve=solve, oracle=oracle, nloc=n, scene=scene, solve_batch=solve_batch,
                solve_batch_distinct=solve_batch_distinct,
                abytes=8 * n * (9 * nlayer + 2 * nlevel + 2 + ng + 1),
                # launches of at most one 64-column block per CU (256 CUs) run the cooperative kernel (api.hip)
                kernel=("k_reflected_coop<true>" if n <= 16384 and ng <= 5 and not os.environ.get("PICASO_AMD_REFL_NO_COOP")
                        else "k_reflected_toa<%d, false, true, true, false, 0>" % ng),
                workload="BASELINE configs[2]: Toon two-stream reflected light (get_reflected_1d + "
                         "compress_disco), TTHG_ray, N=2, delta-Eddington, Rayleigh + cloud slab",
                metric="spectra/sec (1e5 wave x 90 layer reflected)")


def workload_thermal(ctx, args, lo, hi, seed, nwno_total, scene=None):
    """configs[1]: get_thermal_1d + compress_thermal (3 layer planes + wno + surf_reflect in, flux +
    disk flux out)."""
    nlayer, nlevel, ng = args.nlayer, args.nlayer + 1, args.ngauss
    n = hi - lo
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    _, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    if scene is None:
        scene = syn.make_scene(nlayer, nwno_total, seed=seed)
    scene["surf_reflect"] = np.zeros(nwno_total)
    scene["dwno"] = scene["wno"] * 0
    d = resident.upload_scene(scene, ("dtau_og", "w0_no_raman", "cosb_og", "wno", "dwno", "surf_reflect"), lo, hi,
                              ctx=ctx)
    flux = device.DeviceArray((ng, 1, n), ctx)

    def solve(disk):
        resident.thermal_1d(ctx, nlevel, d["wno"], n, ng, 1, scene["tlevel"], d["dtau_og"], d["w0_no_raman"],
                            d["cosb_og"], scene["plevel"], ubar1, d["surf_reflect"], 0, flux, dwno=d["dwno"],
                            calc_type=0, gweight=gw, tweight=tw, flux_disk=disk)

    def oracle(sl):
        from oracle import oracle as orc
        ns = sl.stop - sl.start
        fo, _ = orc.get_thermal_1d(nlevel, scene["wno"][sl], ns, ng, 1, scene["tlevel"],
                                   np.ascontiguousarray(scene["dtau_og"][:, sl]),
                                   np.ascontiguousarray(scene["w0_no_raman"][:, sl]),
                                   np.ascontiguousarray(scene["cosb_og"][:, sl]), scene["plevel"], ubar1,
                                   np.zeros(ns), 0, scene["wno"][sl] * 0, 0)
        return orc.compress_thermal(ns, fo, gw, tw)

    def solve_batch(B):
        """B thermal spectra in ONE launch (picaso_get_thermal_1d_batch_dev): B copies of the scene's three planes in
        their own HBM allocations (shared planes would be read from HBM once and from L2 fifteen times: not what B
        atmospheres of a retrieval do) under B level-temperature profiles (T scaled by 1 + 0.01 s; member 0 is the
        single launch's), each with its own outputs."""
        import ctypes as _ct
        sets = [{k: d[k] for k in ("dtau_og", "w0_no_raman", "cosb_og")}]
        for _ in range(1, B):
            c = {}
            for k in sets[0]:
                c[k] = device.DeviceArray(d[k].shape, ctx)
                _lib.check(_lib.load().picaso_memcpy_d2d(ctx, _ct.c_void_p(c[k].addr), _ct.c_void_p(d[k].addr),
                                                         _ct.c_size_t(d[k].nbytes)), ctx)
            sets.append(c)
        fl = [device.DeviceArray((ng, 1, n), ctx) for _ in range(B)]
        dk = [device.DeviceArray((n,), ctx) for _ in range(B)]
        tl = np.stack([scene["tlevel"] * (1.0 + 0.01 * s) for s in range(B)])
        pl = np.stack([scene["plevel"]] * B)

        def launch():
            resident.thermal_1d_batch(ctx, nlevel, d["wno"], n, ng, 1, tl, [x["dtau_og"] for x in sets],
                                      [x["w0_no_raman"] for x in sets], [x["cosb_og"] for x in sets], pl, ubar1,
                                      d["surf_reflect"], 0, fl, dwno=d["dwno"], calc_type=0, gweight=gw, tweight=tw,
                                      flux_disk=dk)
        return launch, dk

    return dict(solve=solve, oracle=oracle, nloc=n, abytes=8 * n * (3 * nlayer + 3 + ng + 1), solve_batch=solve_batch,
                kernel=("k_thermal_coop<%d>" if n <= 32768 else "k_thermal_toa<%d, false>") % ng,
                workload="BASELINE configs[1]: thermal emission (get_thermal_1d + compress_thermal), "
                         "Planck per level, 5 Gauss angles",
                metric="spectra/sec (%d wave x %d layer thermal)" % (nwno_total, nlayer))


def workload_sh4(ctx, args, lo, hi, seed, nwno_total, scene=None, clear=False, top=0, levels=True):
    """configs[3]: get_reflected_SH, stream = 4, + compress_disco.  ``clear``: the scene has no cloud and the launch is
    handed dtau and w0 only (picaso_reflected_SH_can_derive; the oracle still gets all eleven planes).  ``top``: the
    caller's statement that the first ``top`` layers carry no cloud (picaso_get_reflected_SH_top_dev; what spectrum()
    reads off the cloud profile).  ``levels=False``: the level planes tau / tau_og are left out
    (picaso_reflected_SH_can_derive_levels: running products of the beam exponentials in the kernel; spectrum()'s plane set)."""
    nlayer, nlevel, ng = args.nlayer, args.nlayer + 1, args.ngauss
    n = hi - lo
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    ubar0, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    if scene is None:
        scene = syn.make_scene(nlayer, nwno_total, seed=seed, stream=4)
    scene["F0PI"] = np.ones(nwno_total)
    scene["surf_reflect"] = np.zeros(nwno_total)
    d = resident.upload_scene(scene, resident.SH_PLANES + ("F0PI", "surf_reflect"), lo, hi, ctx=ctx)
    xint = device.DeviceArray((ng, 1, n), ctx)
    opts = (0, 0, 0, 1, 1, 1)        # w_single_form, w_multi_form, psingle_form, *_rayleigh (config.json defaults)

    planes = {"dtau": d["dtau"], "w0": d["w0"]} if clear else d
    if not levels and not clear:
        planes = {k: v for k, v in d.items() if k not in ("tau", "tau_og")}

    def solve(albedo):
        resident.reflected_SH(ctx, nlevel, n, ng, 1, planes, d["surf_reflect"], ubar0, ubar1, 1.0, d["F0PI"], *opts,
                              *TTHG, 4, xint, gweight=gw, tweight=tw, albedo=albedo, cloud_free_above=top)

    def oracle(sl):
        from oracle import oracle as orc
        ns = sl.stop - sl.start
        # copies: the reference multiplies f_deltaM in place once per angle (and so does its restatement)
        planes = [np.array(scene[k][:, sl], order="C") for k in resident.SH_PLANES]
        xo, _ = orc.get_reflected_SH(nlevel, ns, ng, 1, *planes, np.zeros(ns), ubar0, ubar1, 1.0, np.ones(ns),
                                     *opts, *TTHG, 4)
        return orc.compress_disco(ns, 1.0, xo, gw, tw, np.ones(ns))

    if clear:
        return dict(solve=solve, oracle=oracle, nloc=n, abytes=8 * n * (2 * nlayer + 2 + ng + 1), kernel="k_sh4_clear<2>",
                    workload="configs[3] on a cloud-free atmosphere: SH4 reflected light from dtau and w0 only (the other "
                             "nine planes are constants, copies and running sums), angle-independent half of each layer "
                             "shared between two disk angles of a lane",
                    metric="spectra/sec (%d wave x %d layer SH4 reflected, cloud-free)" % (nwno_total, nlayer))
    return dict(solve=solve, oracle=oracle, nloc=n,
                abytes=8 * n * (9 * nlayer + 2 * nlevel + 2 + ng + 1),
                kernel=("k_sh_refl<2, true%s>" % ("" if levels else ", true")) if not top else
                       "k_sh4_clear<2> (layers 0-%d) + k_sh_refl<2, true%s>" % (top - 1, "" if levels else ", true"),
                workload="BASELINE configs[3]: spherical-harmonics SH4 reflected light (get_reflected_SH + "
                         "compress_disco), TTHG, delta-M, Rayleigh + cloud slab",
                metric="spectra/sec (%d wave x %d layer SH4 reflected)" % (nwno_total, nlayer))


def workload_3d(ctx, args, lo, hi, seed, nwno_total, scene=None, single_phase=0):
    """configs[4]: get_reflected_3d on 8x8 facets + compress_disco; facet index fastest in memory.  The 64 facet
    plane sets are generated ON THE DEVICE: every (nlayer, n) base plane is uploaded once and tiled
    over the facets by picaso_broadcast_facets_dev, the optical-depth planes times a per-facet factor -- at the
    stated size (--nwno 100000 on one GPU) that is 51 GB of planes from 0.8 GB of host arrays."""
    nlayer, nlevel = args.nlayer, args.nlayer + 1
    ng = nt = 8
    n = hi - lo
    phase = np.pi / 3                                            # ubar0 != ubar1, cos_theta = 0.5
    gang, gw, tang, tw = disco.get_angles_3d(ng, nt)
    ubar0, ubar1, ct, _, _ = disco.compute_disco(ng, nt, gang, tang, phase)
    base = syn.make_scene(nlayer, n, seed=seed + 7 * lo)
    rng = np.random.default_rng(seed)
    fac = 1.0 + 0.05 * rng.standard_normal(ng * nt)          # facet-to-facet variation of the optical depths
    scaled = ("dtau", "tau", "dtau_og", "tau_og")
    d = {}
    for k in resident.REFLECTED_PLANES:
        src = device.DeviceArray.from_host(base[k], ctx)
        d[k] = device.broadcast_facets(src, ng * nt, fac if k in scaled else None, ctx)
        src.free()
    f0 = device.DeviceArray.from_host(np.ones(n), ctx)
    rs = device.DeviceArray.from_host(np.zeros(n), ctx)
    xint = device.DeviceArray((ng, nt, n), ctx)

    def solve(albedo):
        # single_phase = 0 ('cahoy'): the form get_reflected_3d has of its own (fluxes.py:604-615); 3 (TTHG_ray): the
        # reference's default option, which the library's compile-time default-options 3-D kernel takes
        resident.reflected_3d(ctx, nlevel, n, ng, nt, d, rs, ubar0, ubar1, float(ct), f0, single_phase, 0, *TTHG, xint,
                              gweight=gw, tweight=tw, albedo=albedo)

    def oracle(sl):
        from oracle import oracle as orc
        ns = sl.stop - sl.start
        planes = []
        for k in resident.REFLECTED_PLANES:
            a = np.ascontiguousarray(base[k][:, sl])
            a3 = a[:, :, None] * fac[None, None, :] if k in scaled else np.repeat(a[:, :, None], ng * nt, axis=2)
            planes.append(np.ascontiguousarray(a3).reshape(a.shape[0], ns, ng, nt))
        xo = orc.get_reflected_3d(nlevel, base["wno"][sl], ns, ng, nt, *planes, np.zeros(ns), ubar0, ubar1,
                                  float(ct), np.ones(ns), single_phase, 0, *TTHG)
        xo = xo[0] if isinstance(xo, tuple) else xo
        return orc.compress_disco(ns, float(ct), xo, gw, tw, np.ones(ns))

    return dict(solve=solve, oracle=oracle, nloc=n, oracle_sample=256,
                abytes=8 * n * (ng * nt * (9 * nlayer + 2 * nlevel + 1) + 2 + 1),
                kernel="k_reflected_toa<1, true, false, false, false, 0>",
                workload="BASELINE configs[4]: 3-D reflected light, 8x8 facets at phase pi/3 (get_reflected_3d + "
                         "compress_disco), facet planes generated on the device",
                metric="spectra/sec (%d wave x %d layer x 64 facet 3-D reflected)" % (nwno_total, nlayer))


WORKLOADS = {1: (workload_thermal, 10000), 2: (workload_reflected, 100000), 3: (workload_sh4, 100000),
             4: (workload_3d, 12500)}


def spawn_ranks(ngpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, with the
    environment a one-process-per-GPU launcher would give them, and wait.  The parent never touches HIP (the GPUs
    belong to the ranks); rank 0 inherits stdout, so its one JSON line is this process's one JSON line.  The first
    rank that fails takes the job down: the others are terminated (by PID: they are our own children) and the exit
    status is non-zero."""
    probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)      # a free loopback port for the rendezvous
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    base = dict(os.environ, WORLD_SIZE=str(ngpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                PICASO_AMD_RDZV_PORT=str(port), PICASO_AMD_JOB_TOKEN=secrets.token_hex(16),
                PICASO_AMD_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(ngpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    kill_at = None
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                kill_at = time.time() + 5.0      # the others usually fail for the same reason and say so themselves
        if kill_at is not None and time.time() > kill_at:
            for q in alive:                      # ... or they would wait at the rendezvous for ever
                q.terminate()
            kill_at = float("inf")
        time.sleep(0.02)
    if rc != 0:
        print("bench.py: a rank of the %d-GPU job failed (exit status %d)" % (ngpus, rc), file=sys.stderr, flush=True)
    return rc if rc >= 0 else 1


def gpu_clocks(local=0):
    """Current sclk / mclk in MHz of the GPU this rank runs on, best effort: the amdgpu sysfs tables (the line marked
    '*' is the level in use), else ``rocm-smi --showclocks --json``; None where neither can be read."""
    import glob
    import re
    out = {}
    cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
    if cards:
        d = cards[min(local, len(cards) - 1)]
        for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
            try:
                cur = [ln for ln in open(os.path.join(d, fn)).read().splitlines() if ln.rstrip().endswith("*")]
                out[key] = int(re.search(r"(\d+)\s*[Mm][Hh]z", cur[0]).group(1)) if cur else None
            except Exception:
                out[key] = None
        out["source"] = "sysfs " + os.path.join(d, "pp_dpm_*")
        if out.get("sclk_mhz") is not None:
            return out
    try:
        import subprocess
        txt = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=15).stdout
        card = sorted(json.loads(txt).items())[local][1]
        for key, pat in (("sclk_mhz", "sclk"), ("mclk_mhz", "mclk")):
            val = [v for k, v in card.items() if pat in k.lower() and "level" in k.lower()]
            m_ = re.search(r"(\d+)\s*[Mm][Hh]z", val[0]) if val else None
            out[key] = int(m_.group(1)) if m_ else None
        out["source"] = "rocm-smi --showclocks --json"
    except Exception as exc:
        out.setdefault("sclk_mhz", None)
        out.setdefault("mclk_mhz", None)
        out["source"] = "unreadable (%s)" % type(exc).__name__
    return out


def orc_flags():
    try:
        from oracle import oracle as orc
        return orc.build_flags()
    except Exception as exc:
        return "unknown (%s)" % exc


def steady_ms(ctx, launch, steps, prewarm_ms=60.0):
    """HIP-event time per launch in the steady clock state: `prewarm_ms` of the same launches first, no idle gap."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < prewarm_ms:
        for _ in range(10):
            launch()
        device.sync(ctx)
    for _ in range(10):
        launch()
    device.timer_start(ctx)
    for _ in range(steps):
        launch()
    return device.timer_stop(ctx) / steps


def companions(ctx, args, wl, res_single, nwno_total):
    """Untimed by the driver, after the timed region of the default line (N = 1, BASELINE configs[2]):

    throughput_batched -- the SAME workload, `--batch` plane sets per launch (picaso_get_reflected_1d_batch_dev),
        every member checked bit for bit against the single-spectrum result of the timed region;
    secondary -- the other BASELINE configurations on this box's clock: configs[1] (thermal 1e4 x 90), its batched
        form, configs[3] (SH4 1e5 x 90), the 12 500-column per-GPU shards of configs[2] and configs[4]; each with its
        kernel, algorithmic bytes, HBM fraction and the error against the CPU oracle on <= 256 columns.
    profiles/ holds the rocprofv3 kernel stats of the same command."""
    extra = {}
    t_all = time.perf_counter()
    B = max(2, args.batch)
    launch, albs = wl["solve_batch"](B)
    ms = steady_ms(ctx, launch, max(10, 200 // B), prewarm_ms=150.0)
    same = all(bool(np.array_equal(a.to_host(), res_single)) for a in albs)
    per = ms / B
    extra["throughput_batched"] = {
        "value": 1e3 / per, "unit": "spectra/s", "batch": B, "ms_per_launch": ms, "ms_per_spectrum": per,
        "roofline_frac": wl["abytes"] / (per * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "bit_identical_to_single_launch": same,
        "what": "%d plane sets of the headline workload (set 0 + %d copies in their own HBM allocations) in one "
                "launch of k_reflected_toa_batch; HIP events, steady clocks, after the timed region" % (B, B - 1)}
    if not same:
        print("bench.py: CHECK FAILED: a member of the batched launch differs from the single launch", file=sys.stderr)
    # Consecutive batches of a retrieval are independent: with the launches alternating between two streams (two
    # library contexts, the planes are read-only) the last, partly filled generation of one launch overlaps the
    # first of the next instead of draining the chip
    ctx2 = _lib.aux_context(0)
    launch2, albs2 = wl["solve_batch"](B, on=ctx2)

    def both():
        launch()
        launch2()
    device.sync(ctx2)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < 100.0:
        for _ in range(5):
            both()
        device.sync(ctx)
        device.sync(ctx2)
    npair = max(5, 100 // B)
    for _ in range(3):
        both()
    device.sync(ctx)
    device.sync(ctx2)
    t0 = time.perf_counter()
    for _ in range(npair):
        both()
    device.sync(ctx)
    device.sync(ctx2)
    wall = (time.perf_counter() - t0) * 1e3 / (2 * npair * B)
    extra["throughput_batched"]["two_streams"] = {
        "value": 1e3 / wall, "ms_per_spectrum": wall,
        "roofline_frac": wl["abytes"] / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "bit_identical_to_single_launch": all(bool(np.array_equal(a.to_host(), res_single)) for a in albs2),
        "what": "the same B = %d launches alternating between two streams (independent batches in flight); host "
                "clock around %d launches, synchronised at both ends" % (B, 2 * npair)}
    del launch, albs, launch2, albs2
    # The launch is 391 workgroups per spectrum on 512 resident slots: B = 4 is 3.05 generations (the last one 28
    # workgroups), B = 8 is 6.1 -- the same kernel at a batch size whose tail weighs half as much
    B2 = 2 * B
    launch, albs = wl["solve_batch"](B2)
    ms = steady_ms(ctx, launch, max(10, 200 // B2), prewarm_ms=100.0)
    extra["throughput_batched"]["batch_%d" % B2] = {
        "value": 1e3 * B2 / ms, "ms_per_spectrum": ms / B2,
        "roofline_frac": wl["abytes"] / (ms / B2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "bit_identical_to_single_launch": all(bool(np.array_equal(a.to_host(), res_single)) for a in albs)}
    del launch, albs
    # the same launch with B DIFFERENT atmospheres (own gas field, cloud slab at another height or none): the members'
    # wave-uniform shortcut flags diverge inside one launch, as in a retrieval's batch
    if "solve_batch_distinct" in wl:
        launch, albs, singles = wl["solve_batch_distinct"](B)
        ms = steady_ms(ctx, launch, max(10, 200 // B), prewarm_ms=100.0)
        got = [a.to_host() for a in albs]
        one = singles()
        extra["throughput_batched"]["distinct_atmospheres"] = {
            "value": 1e3 * B / ms, "batch": B, "ms_per_spectrum": ms / B,
            "roofline_frac": wl["abytes"] / (ms / B * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "each_member_bit_identical_to_its_own_single_launch": all(bool(np.array_equal(g, o)) for g, o in zip(got, one)),
            "finite": all(bool(np.all(np.isfinite(g))) for g in got),
            "what": "B = %d different atmospheres in one launch: member 0 the headline scene, the others their own gas "
                    "fields with the cloud slab 6 layers higher / lower per member, every third member cloud-free" % B}
        if not extra["throughput_batched"]["distinct_atmospheres"]["each_member_bit_identical_to_its_own_single_launch"]:
            print("bench.py: CHECK FAILED: a member of the distinct-atmosphere batch differs from its own launch",
                  file=sys.stderr)
        del launch, albs, singles

    def entry(w, ms_, n_oracle=256, res=None):
        e = {"ms": ms_, "kernel": w["kernel"], "algorithmic_bytes": w["abytes"],
             "frac": w["abytes"] / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, "workload": w["workload"]}
        if w.get("oracle") is not None and res is not None:
            ns = min(n_oracle, w["nloc"], w.get("oracle_sample", n_oracle))
            cpu = w["oracle"](slice(0, ns))
            e["max_rel_err_vs_oracle"] = float(np.max(np.abs(res[:ns] - cpu) / np.abs(cpu)))
            e["oracle_columns"] = ns
        return e

    sec = {}
    scene = wl["scene"]
    # configs[1]: thermal emission, 1e4 x 90 (single launch: the cooperative kernel) and 16 spectra per launch
    a1 = argparse.Namespace(**vars(args))
    w1 = workload_thermal(ctx, a1, 0, 10000, 3, 10000)
    out1 = device.DeviceArray((10000,), ctx)
    ms1 = steady_ms(ctx, lambda: w1["solve"](out1), 300)
    sec["configs[1]"] = entry(w1, ms1, res=out1.to_host())
    launch1, dk = w1["solve_batch"](16)
    msb = steady_ms(ctx, launch1, 60) / 16
    first = dk[0].to_host()
    sec["configs[1] x16 batched"] = {
        "ms": msb, "kernel": "k_thermal_toa_batch<5, false>", "algorithmic_bytes": w1["abytes"],
        "frac": w1["abytes"] / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "bit_identical_to_single_launch": bool(np.array_equal(first, out1.to_host())),
        "workload": "configs[1], 16 atmospheres per launch (picaso_get_thermal_1d_batch_dev): 16 plane sets in their own HBM "
                    "allocations, 16 level-temperature profiles; ms per spectrum"}
    del launch1, dk, w1
    # BASELINE.md section 3 asks for the thermal workload at 1e5 and 1e6 wavelengths as well: the headline scene's own
    # thermal planes, and ten copies of them side by side (a 1e6-column scene would take the host a minute to draw)
    for nth, reps in ((nwno_total, 1), (10 * nwno_total, 10)):
        if nwno_total != 100000:
            break
        sct = {k: (np.tile(scene[k], (1, reps)) if scene[k].ndim == 2 else np.tile(scene[k], reps))
               for k in ("dtau_og", "w0_no_raman", "cosb_og", "wno")}
        sct["tlevel"], sct["plevel"] = scene["tlevel"], scene["plevel"]
        wt = workload_thermal(ctx, a1, 0, nth, 3, nth, scene=sct)
        outt = device.DeviceArray((nth,), ctx)
        mst = steady_ms(ctx, lambda: wt["solve"](outt), 100 if reps == 1 else 20, prewarm_ms=100.0)
        sec["configs[1] at %d wavelengths" % nth] = entry(wt, mst, n_oracle=128, res=outt.to_host())
        sec["configs[1] at %d wavelengths" % nth]["wavelength_layer_updates_per_s"] = nth * args.nlayer / (mst * 1e-3)
        del wt, outt, sct
    # configs[2], the 12 500-column block one of 8 GPUs solves
    w2 = workload_reflected(ctx, args, 0, 12500, 3, nwno_total, scene=scene)
    out2 = device.DeviceArray((12500,), ctx)
    ms2 = steady_ms(ctx, lambda: w2["solve"](out2), 300)
    sec["configs[2] 12500-column shard"] = entry(w2, ms2, res=out2.to_host())
    sec["configs[2] 12500-column shard"]["bit_identical_to_unsharded"] = bool(np.array_equal(out2.to_host(), res_single[:12500]))
    del w2
    # configs[3]: SH4 reflected, 1e5 x 90: the same optical-depth components mixed for stream = 4
    sc4 = syn.mix_planes(scene["taugas"], scene["tauray"], scene["taucld"], scene["w0_cld"], scene["g0_cld"], stream=4)
    sc4["wno"] = scene["wno"]
    w3 = workload_sh4(ctx, args, 0, nwno_total, 3, nwno_total, scene=sc4)
    out3 = device.DeviceArray((nwno_total,), ctx)
    ms3 = steady_ms(ctx, lambda: w3["solve"](out3), 40, prewarm_ms=100.0)
    sec["configs[3]"] = entry(w3, ms3, n_oracle=128, res=out3.to_host())
    w3s = workload_sh4(ctx, args, 0, 12500, 3, nwno_total, scene=sc4)
    out3s = device.DeviceArray((12500,), ctx)
    ms3s = steady_ms(ctx, lambda: w3s["solve"](out3s), 150)
    sec["configs[3] 12500-column shard"] = entry(w3s, ms3s)
    sec["configs[3] 12500-column shard"]["bit_identical_to_unsharded"] = bool(np.array_equal(out3s.to_host(), out3.to_host()[:12500]))
    # the same launch told where the cloud deck begins (what spectrum() reads off the cloud profile): the layers above it
    # go through the cloud-free kernel, which hands its sweep state to k_sh
    busy = scene["taucld"].any(axis=1) | scene["g0_cld"].any(axis=1)
    deck = int(np.argmax(busy)) if busy.any() else args.nlayer
    w3t = workload_sh4(ctx, args, 0, nwno_total, 3, nwno_total, scene=sc4, top=deck)
    ms3t = steady_ms(ctx, lambda: w3t["solve"](out3), 40, prewarm_ms=100.0)
    sec["configs[3] cloud_free_above=%d" % deck] = entry(w3t, ms3t, n_oracle=128, res=out3.to_host())
    # ... and without the level planes tau / tau_og (running sums: the kernel carries the beam exponentials as running
    # products; the plane set spectrum() hands over): whole grid, shard, and with the cloud deck statement -- the product's
    # launch.  Algorithmic bytes stay SURVEY 8(d)'s eleven planes.
    for tag, n3, tp in (("configs[3] level planes left out", nwno_total, 0),
                        ("configs[3] 12500-column shard, level planes left out", 12500, 0),
                        ("configs[3] level planes left out, cloud_free_above=%d" % deck, nwno_total, deck)):
        w3l = workload_sh4(ctx, args, 0, n3, 3, nwno_total, scene=sc4, top=tp, levels=False)
        o3l = device.DeviceArray((n3,), ctx)
        ms3l = steady_ms(ctx, lambda: w3l["solve"](o3l), 40 if n3 > 20000 else 150, prewarm_ms=100.0)
        sec[tag] = entry(w3l, ms3l, n_oracle=128, res=o3l.to_host())
        del w3l, o3l
    del w3, w3s, w3t, sc4
    # the same atmosphere without its cloud: the cloud-free SH4 form against the full-plane kernel on the same planes
    # (no cloud profile: opd = w0 = g0 = 0 in every layer, what ATMSETUP.get_clouds leaves, atmsetup.py:609-640 -- COSB is
    # the cloud's g0 itself, optics.py:338, so a g0 without optical depth would still delta-scale the layer)
    sc4c = syn.mix_planes(scene["taugas"], scene["tauray"], 0.0 * scene["taucld"], 0.0 * scene["w0_cld"],
                          0.0 * scene["g0_cld"], stream=4)
    sc4c["wno"] = scene["wno"]
    w3c = workload_sh4(ctx, args, 0, nwno_total, 3, nwno_total, scene=sc4c, clear=True)
    ms3c = steady_ms(ctx, lambda: w3c["solve"](out3), 40, prewarm_ms=100.0)
    sec["configs[3] cloud-free"] = entry(w3c, ms3c, n_oracle=128, res=out3.to_host())
    w3f = workload_sh4(ctx, args, 0, nwno_total, 3, nwno_total, scene=sc4c)
    sec["configs[3] cloud-free"]["full_plane_kernel_ms"] = steady_ms(ctx, lambda: w3f["solve"](out3), 40, prewarm_ms=100.0)
    sec["configs[3] cloud-free"]["frac_is"] = "of the HBM peak for the 1 480 B per wavelength this launch reads and writes; the kernel is fp64-issue bound"
    del w3c, w3f, sc4c
    # configs[4]: 64 facets x 90 layers, the 12 500-wavelength block one of 8 GPUs holds
    w4 = workload_3d(ctx, args, 0, 12500, 3, 12500)
    out4 = device.DeviceArray((12500,), ctx)
    ms4 = steady_ms(ctx, lambda: w4["solve"](out4), 40, prewarm_ms=100.0)
    sec["configs[4] 12500-column shard"] = entry(w4, ms4, n_oracle=64, res=out4.to_host())
    del w4
    w4d = workload_3d(ctx, args, 0, 12500, 3, 12500, single_phase=3)
    ms4d = steady_ms(ctx, lambda: w4d["solve"](out4), 40, prewarm_ms=100.0)
    sec["configs[4] 12500-column shard, default phase options (TTHG_ray)"] = entry(w4d, ms4d, n_oracle=64, res=out4.to_host())
    del w4d
    extra["secondary"] = sec
    extra["secondary_seconds"] = time.perf_counter() - t_all
    return extra


def product_companion(ctx, nwno=100000, nlevel=91, ncalls=30, nbatch=32):
    """The product call on this box's clock (untimed by the driver, after everything else): `inputs.spectrum(opa,
    "reflected+thermal")` end to end -- ATMSETUP, table-row search, opacity mixing from resident tables (5 molecules,
    2 CIA pairs, 2 Rayleigh species), both Toon solves, disk sums, integrals, results on the host -- and the same as a
    retrieval would run it, `spectrum_batch` over `nbatch` atmospheres (checked key by key against the single calls)."""
    from picaso_amd import justdoit as jdi
    from picaso_amd import optics as px
    t_all = time.perf_counter()
    wno = np.linspace(2000.0, 33333.0, nwno)
    temps, press = [100.0, 300.0, 700.0, 1500.0, 3000.0], [1e-6, 1e-4, 1e-2, 1e-1, 1.0, 10.0, 100.0, 500.0]
    pt = [(i + 1, p, t) for i, (t, p) in enumerate((t, p) for t in temps for p in press)]
    mols = ["H2O", "CH4", "CO", "NH3", "H2"]
    molecular = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(wno / 2500.0 + k) + 0.4 * np.log10(p) + 0.8 * np.log10(t / 300.0))
                     for (i, p, t) in pt} for k, m in enumerate(mols)}
    cia_t = [75.0, 200.0, 500.0, 1000.0, 2000.0, 4000.0]
    continuum = {pr: {t: 10.0 ** (-7.0 + np.cos(wno / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
                 for k, pr in enumerate(("H2H2", "H2He"))}
    ray = {m: 1e-27 * (wno / 1e4) ** 4 for m in ("H2", "He")}
    opa = px.RetrieveOpacities(wno, pt, molecular, continuum, cia_t, rayleigh_opa=ray, query_method="linear", ctx=ctx)
    plev = np.logspace(-6, 2, nlevel)
    prof = {"pressure": plev, "temperature": 150.0 + 1200.0 * ((np.log10(plev) + 6) / 8) ** 2,
            "H2": np.full(nlevel, 0.84), "He": np.full(nlevel, 0.155), "H2O": np.full(nlevel, 1e-3),
            "CH4": np.full(nlevel, 5e-4), "CO": np.full(nlevel, 1e-4), "NH3": np.full(nlevel, 1e-5)}

    def make(k):
        case = jdi.inputs()
        case.phase_angle(0, num_gangle=5)
        case.gravity(gravity=2500.0)
        case.atmosphere(df=dict(prof, temperature=prof["temperature"] * (1.0 + 0.01 * k)))
        case.approx(raman="none")
        return case
    calc = "reflected+thermal"
    case = make(0)
    for _ in range(15):
        r = case.spectrum(opa, calculation=calc)
    ts = []
    for _ in range(ncalls):
        t0 = time.perf_counter()
        r = case.spectrum(opa, calculation=calc)
        ts.append(time.perf_counter() - t0)
    idle = []                          # what one interactive call sees: the GPU has dropped its clocks meanwhile
    for _ in range(5):
        time.sleep(0.3)
        t0 = time.perf_counter()
        r = case.spectrum(opa, calculation=calc)
        idle.append(time.perf_counter() - t0)
    cases = [make(k) for k in range(nbatch)]
    for _ in range(2):
        rb = jdi.spectrum_batch(cases, opa, calculation=calc)
    tb = []
    for _ in range(3):
        t0 = time.perf_counter()
        rb = jdi.spectrum_batch(cases, opa, calculation=calc)
        tb.append(time.perf_counter() - t0)
    rl = [c.spectrum(opa, calculation=calc) for c in cases[:4]]
    same = all(np.array_equal(a[k], b[k]) for a, b in zip(rb[:4], rl) for k in a if isinstance(a[k], np.ndarray))

    def pipelined():                   # a retrieval's loop: ask for sample i + 1, then read sample i (spectrum_async)
        prev, outs_ = None, []
        for c in cases:
            h = c.spectrum_async(opa, calculation=calc)
            if prev is not None:
                outs_.append(prev.result())
            prev = h
        outs_.append(prev.result())
        return outs_
    for _ in range(2):
        ra = pipelined()
    ta = []
    for _ in range(3):
        t0 = time.perf_counter()
        ra = pipelined()
        ta.append(time.perf_counter() - t0)
    same_async = all(np.array_equal(a[k], b[k]) for a, b in zip(ra[:4], rl) for k in a if isinstance(a[k], np.ndarray))

    def timed(c, n=12):
        for _ in range(4):
            c.spectrum(opa, calculation=calc)
        tt = []
        for _ in range(n):
            t0 = time.perf_counter()
            out = c.spectrum(opa, calculation=calc)
            tt.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(tt)), out
    def guarded(fn):                      # a companion that fails reports its error instead of taking the others down
        try:
            return fn()
        except Exception as exc:
            return {"error": "%s: %s" % (type(exc).__name__, exc)}

    nl = nlevel - 1

    def part_sh4():
        # the same call with the spherical-harmonics solver (SH4): without cloud (dtau and w0 only, angle-independent half
        # of a layer shared between disk angles) and with a grey box cloud below layer 50 (the layers above the deck go
        # through the cloud-free kernel: spectrum() states where the deck begins, justdoit._cloud_free_top)
        sh = make(0)
        sh.approx(raman="none", rt_method="SH", stream=4)
        sh_ms, sh_out = timed(sh)
        shc = make(0)
        shc.approx(raman="none", rt_method="SH", stream=4)
        box = np.zeros((nl, 196))
        box[50:60] = 0.3
        shc.clouds(df={"opd": box, "w0": np.where(box > 0, 0.95, 0.0), "g0": np.where(box > 0, 0.6, 0.0)},
                   wavenumber=np.linspace(wno[0], wno[-1], 196))
        shc_ms, shc_out = timed(shc)
        return {"sh4_spectrum_ms": sh_ms, "sh4_box_cloud_below_layer_50_spectrum_ms": shc_ms,
                "finite": bool(all(np.all(np.isfinite(o[k])) for o in (sh_out, shc_out) for k in ("albedo", "thermal")))}

    def part_3d():
        # the 3-D form of the same call (BASELINE configs[4]'s per-GPU shape: 8 x 8 facets x 12 500 wavelengths x 90 layers,
        # per-facet temperatures): without cloud, with a per-facet cloud map on a 196-point wavenumber grid of its own
        # (clouds_3d as virga hands it over), and an 8-phase reflected-light curve of the cloudy map
        n3 = 12500
        w3 = np.linspace(3000.0, 30000.0, n3)
        mol3 = {m: {i: 10.0 ** (-24.0 + 2.0 * np.sin(w3 / 2500.0 + k) + 0.4 * np.log10(p) + 0.8 * np.log10(t / 300.0))
                    for (i, p, t) in pt} for k, m in enumerate(mols)}
        con3 = {pr: {t: 10.0 ** (-7.0 + np.cos(w3 / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
                for k, pr in enumerate(("H2H2", "H2He"))}
        opa3 = px.RetrieveOpacities(w3, pt, mol3, con3, cia_t, rayleigh_opa={m: 1e-27 * (w3 / 1e4) ** 4 for m in ("H2", "He")},
                                    query_method="linear", ctx=ctx)
        pert = 1.0 + 0.1 * np.cos(np.arange(64).reshape(8, 8))
        prof3 = dict(prof, temperature=prof["temperature"][:, None, None] * pert[None])
        box3 = np.zeros((nl, 196, 8, 8))
        box3[50:60] = 0.3 * (1.0 + 0.3 * np.cos(np.arange(64).reshape(1, 1, 8, 8)))
        cmap = {"opd": box3, "w0": np.where(box3 > 0, 0.95, 0.0), "g0": np.where(box3 > 0, 0.6, 0.0),
                "wavenumber": np.linspace(w3[0], w3[-1], 196)}

        def case3(cloudy):
            c = jdi.inputs()
            c.phase_angle(np.pi / 3, num_gangle=8, num_tangle=8)
            c.gravity(gravity=2500.0)
            c.atmosphere_3d(prof3)
            c.approx(raman="none")
            if cloudy:
                c.clouds_3d(df=cmap)
            return c

        def timed3(c, n=8):
            for _ in range(3):
                c.spectrum(opa3, calculation=calc, dimension="3d")
            tt = []
            for _ in range(n):
                t0 = time.perf_counter()
                out = c.spectrum(opa3, calculation=calc, dimension="3d")
                tt.append(time.perf_counter() - t0)
            return 1e3 * float(np.median(tt)), out
        s3_ms, s3_out = timed3(case3(False))
        s3c_ms, s3c_out = timed3(case3(True))
        phases = list(2 * np.pi * (np.arange(8) + 0.5) / 8)          # (not pi itself: the disk geometry divides by 1 + cos)
        pc = jdi.inputs()
        pc.phase_curve_geometry("reflected", phases, num_gangle=8, num_tangle=8)
        pc.gravity(gravity=2500.0)
        pc.atmosphere_4d([prof3 for _ in phases])
        pc.approx(raman="none")
        pc.phase_curve(opa3, clouds_by_phase=[cmap] * 8)
        t0 = time.perf_counter()
        curve = pc.phase_curve(opa3, clouds_by_phase=[cmap] * 8)
        pc_ms = 1e3 * (time.perf_counter() - t0)
        fin3 = all(np.all(np.isfinite(o[k])) for o in (s3_out, s3c_out) for k in ("albedo", "thermal")) and \
            all(np.all(np.isfinite(v["albedo"])) for v in curve.values())
        return {"workload": "spectrum(dimension='3d', 'reflected+thermal'): 8 x 8 facets x %d wavelengths x %d layers, "
                            "per-facet temperatures" % (n3, nl),
                "cloud_free_ms": s3_ms, "per_facet_cloud_map_on_196_point_grid_ms": s3c_ms,
                "phase_curve_reflected_8_phases_cloudy_ms": pc_ms, "finite": bool(fin3)}

    def part_climate():
        # the climate solver's call between Jacobians: get_fluxes (reference climate.py:1687) at the climate tables' shape --
        # 91 levels, 661 bins x 8 Gauss points, one two-stream angle for the visible and 5 disk angles for the infrared,
        # level fluxes of both legs back on the host -- with the thirteen opacity planes resident (what calculate_atm hands over)
        from picaso_amd import climate as pcl
        from picaso_amd.device import DeviceArray
        nlev_c, nw_c, ng_c = 91, 661, 8
        scs = [syn.make_scene(nlev_c - 1, nw_c, seed=70 + ig, gas_scale=10.0 ** (0.5 * ig - 2)) for ig in range(ng_c)]
        stc = {k: np.ascontiguousarray(np.stack([sc_[k] for sc_ in scs], axis=2)) for k in resident.REFLECTED_PLANES + ("w0_no_raman",)}
        xg, wg = np.polynomial.legendre.leggauss(ng_c)
        gang_c, gw_c, tang_c, tw_c = disco.get_angles_1d(5)
        u0_c, u1_c, _, _, _ = disco.compute_disco(5, 1, gang_c, tang_c, 0.0)
        wno_c = scs[0]["wno"]
        atm_t = pcl.Atmosphere_Tuple(None, None, nlev_c, scs[0]["tlevel"], scs[0]["plevel"], None, None, None, None)
        sp_t = pcl.ScatteringPhase_Tuple(np.zeros(nw_c), 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
        dis_t = pcl.Disco_Tuple(5, 1, gw_c, tw_c, u0_c, u1_c, 1.0)
        og_t = pcl.Opagrid_Tuple(nw_c, np.abs(np.gradient(wno_c)), wno_c, ng_c, 0.5 * wg)
        up = lambda a: DeviceArray.from_host(a, ctx)
        wed = pcl.OpacityWEd_Tuple(*[up(stc[k]) for k in ("dtau", "tau", "w0", "cosb", "ftau_cld", "ftau_ray", "gcos2", "w0_no_raman")], None)
        noed = pcl.OpacityNoEd_Tuple(*[up(stc[k]) for k in ("dtau_og", "tau_og", "w0_og", "cosb_og")])
        f0c = np.ones(nw_c)
        for _ in range(10):
            fl = pcl.get_fluxes(atm_t, wed, noed, sp_t, dis_t, og_t, f0c, True, True, ctx=ctx)
        tc_ = []
        for _ in range(30):
            t0 = time.perf_counter()
            fl = pcl.get_fluxes(atm_t, wed, noed, sp_t, dis_t, og_t, f0c, True, True, ctx=ctx)
            tc_.append(time.perf_counter() - t0)
        clim_ms = 1e3 * float(np.median(tc_))
        clim_ok = all(np.all(np.isfinite(a)) for a in fl)
        return {"workload": "climate.get_fluxes(reflected, thermal): 91 levels, 661 bins x 8 Gauss points, level "
                            "fluxes of both legs, resident opacity planes", "ms": clim_ms, "finite": bool(clim_ok)}

    def part_transmission():
        c = make(0)
        c.gravity(radius=7.1e9, mass=1.9e30)
        c.star(relative_flux=1.0 + 0.2 * np.cos(wno / 900.0), radius=6.9e10, semi_major=7.5e12)
        for _ in range(4):
            c.spectrum(opa, calculation="transmission")
        tt = []
        for _ in range(12):
            t0 = time.perf_counter()
            o = c.spectrum(opa, calculation="transmission")
            tt.append(time.perf_counter() - t0)
        return {"ms": 1e3 * float(np.median(tt)), "finite": bool(np.all(np.isfinite(o["transit_depth"])))}

    def part_ck():
        # correlated-k tables at the climate grid's shape (661 bins x 8 Gauss points, premixed): the Gauss loop is the column
        # axis of ONE launch per leg for every solver (csrc/ckloop.hip) -- SH4 in 1-D, and the 3-D branch on 8 x 8 facets
        # with per-facet temperatures (reference justdoit.py:256-307, 488-516), plus an 8-phase thermal phase curve
        nb, nk = 661, 8
        wck = np.linspace(40.0, 28000.0, nb)
        xg, wg = np.polynomial.legendre.leggauss(4)
        gpts = np.concatenate([0.95 * 0.5 * (xg + 1), 0.95 + 0.05 * 0.5 * (xg + 1)])
        gwts = np.concatenate([0.95 * 0.5 * wg, 0.05 * 0.5 * wg])
        tk, pk = np.array(temps), np.array(press)
        lnk = np.log(10.0) * (-26.0 + 2.0 * np.sin(wck / 2500.0)[None, None, :, None] + 0.5 * np.log10(pk)[:, None, None, None]
                              + 0.9 * np.log10(tk / 300.0)[None, :, None, None] + 0.6 * np.arange(nk)[None, None, None, :])
        cont = {pr: {t: 10.0 ** (-7.0 + np.cos(wck / 4000.0 + k) + 0.3 * np.log10(t / 300.0)) for t in cia_t}
                for k, pr in enumerate(("H2H2", "H2He"))}
        opk = px.RetrieveCKs(wck, gwts, np.tile(pk, tk.size), np.repeat(tk, pk.size), np.full(tk.size, pk.size), lnk,
                             continuum=cont, cia_temps=cia_t, rayleigh_opa={m: 1e-27 * (wck / 1e4) ** 4 for m in ("H2", "He")},
                             gauss_pts=gpts, ctx=ctx)

        def timed_k(c, n=10, **kw):
            for _ in range(3):
                c.spectrum(opk, calculation=calc, **kw)
            tt = []
            for _ in range(n):
                t0 = time.perf_counter()
                out = c.spectrum(opk, calculation=calc, **kw)
                tt.append(time.perf_counter() - t0)
            return 1e3 * float(np.median(tt)), out
        toon = make(0)
        toon_ms, toon_out = timed_k(toon)
        sh = make(0)
        sh.approx(raman="none", rt_method="SH", stream=4)
        sh_ms, sh_out = timed_k(sh)
        pert = 1.0 + 0.1 * np.cos(np.arange(64).reshape(8, 8))
        c3 = jdi.inputs()
        c3.phase_angle(np.pi / 3, num_gangle=8, num_tangle=8)
        c3.gravity(gravity=2500.0)
        c3.atmosphere_3d(dict(prof, temperature=prof["temperature"][:, None, None] * pert[None]))
        c3.approx(raman="none")
        d3_ms, d3_out = timed_k(c3, n=6, dimension="3d")
        phases = list(2 * np.pi * (np.arange(8) + 0.5) / 8)
        pc = jdi.inputs()
        pc.phase_curve_geometry("thermal", phases, num_gangle=8, num_tangle=8)
        pc.gravity(gravity=2500.0)
        pc.atmosphere_4d([dict(prof, temperature=prof["temperature"][:, None, None] * (pert[None] + 0.01 * k)) for k in range(8)])
        pc.approx(raman="none")
        pc.phase_curve(opk)
        t0 = time.perf_counter()
        curve = pc.phase_curve(opk)
        pc_ms = 1e3 * (time.perf_counter() - t0)
        fin = all(np.all(np.isfinite(o[k])) for o in (toon_out, sh_out, d3_out) for k in ("albedo", "thermal")) and \
            all(np.all(np.isfinite(v["thermal"])) for v in curve.values())
        return {"workload": "spectrum(opa = RetrieveCKs: %d bins x %d Gauss points, premixed, %d layers, 'reflected+thermal')"
                            % (nb, nk, nl),
                "toon_1d_ms": toon_ms, "sh4_1d_ms": sh_ms, "toon_3d_8x8_facets_ms": d3_ms,
                "thermal_phase_curve_8_phases_8x8_facets_ms": pc_ms, "finite": bool(fin)}

    sh4 = guarded(part_sh4)
    return {"product": {
        "correlated_k": guarded(part_ck),
        "climate_get_fluxes": guarded(part_climate),
        "transmission_spectrum": guarded(part_transmission),
        "spectrum_3d": guarded(part_3d),
        "workload": "inputs.spectrum(opa, 'reflected+thermal'), %d wavelengths x %d layers, 5 Gauss angles, cloud-free, "
                    "resident opacity tables (5 molecules, 2 CIA pairs, 2 Rayleigh species): set-up, opacity mixing, "
                    "both Toon solves, disk sums, integrals, results on the host" % (nwno, nlevel - 1),
        "spectrum_ms": 1e3 * float(np.median(ts)), "spectrum_ms_min": 1e3 * min(ts),
        "spectrum_ms_after_300ms_idle": 1e3 * float(np.median(idle)),
        "spectrum_batch_ms_per_spectrum": 1e3 * min(tb) / nbatch, "batch_of": nbatch,
        "spectrum_batch_equals_single_calls": bool(same),
        "spectrum_async_pipelined_ms_per_spectrum": 1e3 * min(ta) / nbatch,
        "spectrum_async_equals_single_calls": bool(same_async),
        "sh4_spectrum_ms": sh4.get("sh4_spectrum_ms"),
        "sh4_box_cloud_below_layer_50_spectrum_ms": sh4.get("sh4_box_cloud_below_layer_50_spectrum_ms"),
        **({"sh4_error": sh4["error"]} if "error" in sh4 else {}),
        "finite": bool(np.all(np.isfinite(r["albedo"])) and np.all(np.isfinite(r["thermal"])) and sh4.get("finite", False)),
        "seconds": time.perf_counter() - t_all}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"))
    ap.add_argument("--gather-every", type=int, default=4,
                    help="N > 1: spectra per collective launch (a group of that many all-gathers; each spectrum is "
                         "still gathered into its own contiguous buffer inside the timed region)")
    ap.add_argument("--nwno", type=int, default=0, help="wavelengths of the whole spectrum (0 = the config's)")
    ap.add_argument("--nlayer", type=int, default=90)
    ap.add_argument("--ngauss", type=int, default=5, help="disk Gauss angles (5..8)")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed launches of the same step before the W warm-up steps (clock ramp)")
    ap.add_argument("--cpu-sample", type=int, default=100000,
                    help="wavelengths of the same workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--steady-steps", type=int, default=2000,
                    help="N = 1: steps of an extra untimed run after the timed region, reported as steady_state")
    ap.add_argument("--repeats", type=int, default=9,
                    help="further K-step blocks after the timed one (N = 1): their median / min / max go into the line")
    ap.add_argument("--secondary", type=int, default=1,
                    help="N = 1, default workload: after the timed region also time the batched launch of the same "
                         "workload (throughput_batched) and the other BASELINE configurations (secondary); 0 = skip")
    ap.add_argument("--batch", type=int, default=4, help="spectra per launch of the throughput_batched companion")
    ap.add_argument("--product", type=int, default=1,
                    help="with --secondary: also time inputs.spectrum() and spectrum_batch() end to end (product); 0 = skip")
    ap.add_argument("--traffic", default="live", choices=("live", "committed", "off"),
                    help="roofline.traffic: 'live' = two rocprofv3 --pmc passes of a short child run, now (falls back to the "
                         "committed profiles/traffic.json when the profiler is not usable); 'committed' = that file only")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as subprocesses also for --gpus 1 (the N > 1 code path -- rendezvous, RCCL "
                         "communicator, gather in the timed region, checks -- with one rank, on a 1-GPU box)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, let them find each other (HostGroup) and exit: checks the launch path "
                         "of a node without touching a GPU")
    args = ap.parse_args()

    launched = "RANK" in os.environ
    if (args.gpus > 1 or args.spawn) and not launched:
        raise SystemExit(spawn_ranks(args.gpus))
    rank, world, local_rank, addr, port = sharding.launcher_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.rendezvous_only:
        group = sharding.HostGroup(rank, world, addr, port, timeout=60.0)
        seen = [int(b.decode()) for b in group.all_gather_bytes(str(rank).encode())]
        group.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous": seen, "world": world, "port": group.port,
                              "spawned_by_bench": bool(os.environ.get("PICASO_AMD_BENCH_SPAWNED"))}), flush=True)
        group.close()
        return
    ndev = _lib.device_count()
    if ndev < max(world, 1):
        # one GPU per rank, never two ranks on one device and never fewer ranks than asked for
        raise SystemExit("bench.py [rank %d]: --gpus %d needs %d GPUs, %d visible" % (rank, args.gpus, world, ndev))
    if local_rank >= ndev:
        raise SystemExit("bench.py [rank %d]: LOCAL_RANK %d but %d GPUs visible" % (rank, local_rank, ndev))
    ctx = _lib.context(local_rank)
    group = comm = None
    if launched:
        group = sharding.HostGroup(rank, world, addr, port)
        comm = sharding.Comm.from_launcher(ctx, group)     # RCCL inside the library

    build, nwno_cfg = WORKLOADS[args.config]
    # configs[4] is stated as 1e5 wavelengths over 8 GPUs: 12 500 per GPU; with --gpus N and no --nwno the grid is
    # N x 12 500 wavelengths, cut into N blocks like every other configuration
    nwno = args.nwno or (nwno_cfg * world if args.config == 4 else nwno_cfg)
    if args.scaling == "strong":
        nwno_total = nwno
        lo, hi = sharding.shard_of(nwno_total, world, rank)
        seed = 3
    else:
        nwno_total = nwno * world
        lo, hi = rank * nwno, (rank + 1) * nwno
        seed = 3
    wl = build(ctx, args, lo, hi, seed, nwno_total)
    nloc = hi - lo
    # two result buffers: with N > 1 the gather of spectrum i is still in flight on the communicator's
    # stream while spectrum i+1 is solved into the other buffer
    # N > 1: spectra are gathered in batches of G (one collective launch per batch: the two stream events and
    # the RCCL launch of a collective cost ~10 us, a fifth of a 12 500-column solve); two batches of buffers, so
    # the gather of one batch runs on the communicator's stream while the next batch is being solved
    G = max(1, args.gather_every) if comm else 1
    loc = [device.DeviceArray((nloc,), ctx) for _ in range(2 * G)]
    full = [device.DeviceArray((nwno_total,), ctx) for _ in range(2 * G)] if comm else None
    nstep = [0]
    pending = []                 # buffer indices solved but not yet handed to a collective

    def flush():
        if comm and pending:
            slot = (pending[0] // G) & 1
            if len(pending) == 1:
                comm.all_gather_spectrum_async(loc[pending[0]], full[pending[0]], nwno_total, slot)
            else:
                comm.all_gather_spectra_async([loc[j] for j in pending], [full[j] for j in pending], nwno_total, slot)
            del pending[:]

    def step(gather=True):
        # spectrum i is solved into buffer i % 2G; a batch's buffers are reused only after the batch's previous
        # gather (device-side wait at the start of the batch, no host synchronisation)
        j = nstep[0] % (2 * G)
        nstep[0] += 1
        if comm and j % G == 0:
            comm.wait_slot((j // G) & 1)
        wl["solve"](loc[j])
        if comm and gather:
            pending.append(j)
            if j % G == G - 1:
                flush()

    def barrier():
        if comm:
            flush()                  # a partial last batch
            comm.wait_slot(-1)       # every gather has finished before the stream is drained
        device.sync(ctx)
        if comm:
            comm.barrier()

    def timed(nsteps, gather=True):
        device.timer_start(ctx)
        for _ in range(nsteps):
            step(gather)
        return device.timer_stop(ctx) / nsteps

    # ---- the first launches of a cold process (reported, not the metric): what ONE interactive call sees ----
    device.sync(ctx)
    cold_ms = timed(20, gather=False)
    nstep[0] = 0
    # ---- clock ramp (untimed), warm-up, timed region ----
    # The time-based part launches the solve only (no collective: the ranks may run different numbers
    # of iterations); a fixed number of complete steps follows so that the gather is warm as well.
    t0 = time.perf_counter()
    nprewarm = 0
    while (time.perf_counter() - t0) * 1e3 < args.prewarm_ms:
        for _ in range(20):
            step(gather=False)
        nprewarm += 20
        device.sync(ctx)
    nstep[0] = 0                     # the ranks ran different numbers of ramp launches: same batch phase from here on
    if args.prewarm_ms > 0:
        for _ in range(20):
            step()
        nprewarm += 20
    clocks_before = gpu_clocks(local_rank) if not comm else None      # behind the ramp, in front of the W warm-up steps
    for _ in range(args.warmup):
        step()
    barrier()
    device.timer_start(ctx)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if comm:
        flush()
        comm.wait_slot(-1)
    stream_ms = device.timer_stop(ctx) / args.steps       # HIP events on the kernel's stream (solve [+ last gathers])
    barrier()
    elapsed = time.perf_counter() - t0
    # ---- untimed: the spread of the same K-step block on this box (nine more blocks, each bracketed like the timed
    # one) and the clocks the GPU reports while a block is still on the stream -- what makes `value` comparable across
    # boxes (the timed block above is the metric; these only say where it sits) ----
    repeats = None
    if not comm and args.repeats > 0:
        blocks, clk = [], []
        for _ in range(args.repeats):
            device.sync(ctx)
            device.timer_start(ctx)
            for _ in range(args.steps):
                step()
            if len(clk) < 3:
                clk.append(gpu_clocks(local_rank))          # read while the block runs (the launches above are asynchronous)
            blocks.append(device.timer_stop(ctx) / args.steps)
        repeats = {"n": len(blocks), "steps_per_block": args.steps, "median_ms": float(np.median(blocks)),
                   "min_ms": float(min(blocks)), "max_ms": float(max(blocks)), "ms": [round(b, 5) for b in blocks],
                   "timed_block_ms": stream_ms, "clocks_before_timed_block": clocks_before, "clocks_under_load": clk,
                   "clocks_idle_after": None,
                   "what": "HIP-event time per step of further %d-step blocks after the timed one (untimed by the driver)"
                           % args.steps}
        device.sync(ctx)
        repeats["clocks_idle_after"] = gpu_clocks(local_rank)
    if comm:
        elapsed = comm.max(elapsed)
    last = (nstep[0] - 1) % (2 * G)
    res_local = loc[last].to_host()
    res_full = full[last].to_host() if comm else res_local

    # ---- untimed: per-rank split of the step into solve and gather ----
    kernel_ms = stream_ms
    per_rank = None
    if comm:
        for _ in range(200):                # the copies above idled the GPU: back to steady clocks first
            step()
        kernel_ms = timed(args.steps, gather=False)
        device.timer_start(ctx)                               # the gather alone, serialised behind each solve
        for _ in range(args.steps):
            step(gather=True)
            flush()
            comm.wait_slot(-1)
        both_ms = device.timer_stop(ctx) / args.steps
        info = json.dumps({"rank": rank, "nwno": nloc, "kernel_ms": kernel_ms,
                           "gather_ms": max(both_ms - kernel_ms, 0.0)}).encode()
        per_rank = [json.loads(b.decode()) for b in group.all_gather_bytes(info)]

    out = None
    if rank == 0:
        # ---- parity: the gathered spectrum holds this rank's shard bit-exactly, and (strong scaling)
        # equals the spectrum solved unsharded on this GPU bit for bit (SURVEY 7 test (v)) ----
        checks = {}
        if comm:
            # recorded in the JSON line (and on stderr when false) rather than asserted: a failed check must
            # not cost the run its measurement
            checks["gathered_contains_local_shard"] = bool(np.array_equal(res_full[lo:hi], res_local))
            # configs[4]'s unsharded planes are 64 x the 1-D ones (51 GB at 1e5 wavelengths): solved next to this rank's
            # shard only while both fit comfortably in HBM
            if args.scaling == "strong" and (args.config != 4 or nwno_total * 64 * args.nlayer * 8 * 12 < 120e9):
                wl1 = build(ctx, args, 0, nwno_total, seed, nwno_total)
                one = device.DeviceArray((nwno_total,), ctx)
                wl1["solve"](one)
                device.sync(ctx)
                ref_full = one.to_host()
                checks["bit_identical_to_unsharded"] = bool(np.array_equal(ref_full, res_full))
                checks["max_rel_diff_vs_unsharded"] = float(np.max(np.abs(ref_full - res_full) /
                                                                   np.maximum(np.abs(ref_full), 1e-300)))
            for k, v in checks.items():
                if v is False:
                    print("bench.py: CHECK FAILED: %s" % k, file=sys.stderr, flush=True)
        ms_per_step = 1e3 * elapsed / args.steps
        spectra_per_step = 1 if args.scaling == "strong" else world
        value = spectra_per_step * args.steps / elapsed
        abytes = wl["abytes"]
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        traffic = valu = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        src_hash = kernel_source_hash()
        traffic_source = traffic_committed = None
        if args.config == 2 and world == 1 and nwno == 100000 and args.nlayer == 90 and args.ngauss == 5 \
                and os.path.exists(tfile) and args.traffic != "off":
            try:
                prof = json.load(open(tfile))
                if prof.get("kernel_source_hash") == src_hash:     # counters of THESE kernel sources only
                    traffic, valu = prof.get("hbm_bytes_per_launch"), prof.get("valu_wave_insts_per_launch")
                    traffic_committed = traffic
                    traffic_source = ("profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of these "
                                      "kernel sources, committed; not measured in this run)")
            except Exception:
                traffic = valu = None
            if args.traffic == "live":
                device.sync(ctx)
                live, note = live_traffic()
                if live is not None:
                    traffic, traffic_source = live, note
                elif traffic_source:
                    traffic_source += "; live pass not taken: " + note
        out = {
            "metric": wl["metric"] if args.config != 2 else "spectra/sec (1e5 wave x 90 layer reflected)",
            "value": value, "unit": "spectra/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["workload"], "nwno": nwno_total, "nlayer": args.nlayer,
                       "gauss_angles": args.ngauss if args.config != 4 else 64,
                       "sharding": "%d contiguous wavelength block(s) of %s" % (
                           world, "%d..%d" % (nwno_total // world, -(-nwno_total // world))),
                       "collective": "RCCL all-gather of the albedo shards inside libpicaso_hip.so "
                                     "(picaso_all_gather_multi_async_dev: batches of %d spectra per collective launch, overlapping the next solves), "
                                     "in the timed region" % G
                       if comm else "none"},
            "prewarm": {"ms": args.prewarm_ms, "launches": nprewarm, "why": "GPU clock ramp, untimed"},
            "cold_ms_first_20": cold_ms,
            "repeats": repeats,
            "wavelength_layer_updates_per_s": value * nwno_total / spectra_per_step * args.nlayer,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source, "traffic_committed": traffic_committed,
                         "kernel": wl["kernel"], "kernel_ms": kernel_ms, "algorithmic_bytes": abytes,
                         "kernel_source_hash": src_hash},
        }
        if per_rank:
            out["per_rank"] = per_rank
            out["checks"] = checks
        if world == 1 and args.steady_steps > 0:
            # the HBM ceiling this box reaches with a plain copy ("peak = vendor 8 TB/s and a measured
            # hipMemcpyDtoD ceiling on the same box, both stated"): 1 GiB device-to-device, read + write counted
            try:
                import ctypes as _ct
                from picaso_amd._lib import check as _check, load as _load
                nb = 1 << 30
                src_c, dst_c = device.DeviceArray((nb // 8,), ctx), device.DeviceArray((nb // 8,), ctx)
                src_c.zero()

                def _copy():
                    _check(_load().picaso_memcpy_d2d(ctx, _ct.c_void_p(dst_c.addr), _ct.c_void_p(src_c.addr),
                                                     _ct.c_size_t(nb)), ctx)
                for _ in range(3):
                    _copy()
                device.sync(ctx)
                device.timer_start(ctx)
                for _ in range(10):
                    _copy()
                cms = device.timer_stop(ctx) / 10
                out["roofline"]["measured_copy_peak"] = {"GBps_read_plus_write": 2 * nb / (cms * 1e-3) / 1e9,
                                                         "what": "picaso_memcpy_d2d of 1 GiB on this box, after the run"}
                out["roofline"]["frac_of_measured_copy_peak"] = achieved / out["roofline"]["measured_copy_peak"][
                    "GBps_read_plus_write"]
                src_c.free()
                dst_c.free()
            except Exception as exc:      # a reported extra, never worth the measurement above
                out["roofline"]["measured_copy_peak"] = {"error": str(exc)}
            # a long companion run (untimed by the driver): long enough for a utilisation sampler to see the
            # GPU busy, and the check that the K timed steps above sit in the steady state
            nst = max(50, min(args.steady_steps, int(1000.0 / max(ms_per_step, 1e-3))))     # about a second at most
            for _ in range(50):
                step()
            out["steady_state"] = {"steps": nst, "ms_per_step": timed(nst)}
        if world == 1 and args.config == 2 and args.secondary and args.scaling == "strong":
            try:      # reported extras: never worth the measurement above
                out.update(companions(ctx, args, wl, res_local, nwno_total))
            except Exception as exc:
                out["companions_error"] = "%s: %s" % (type(exc).__name__, exc)
                print("bench.py: the companion measurements failed: %s" % exc, file=sys.stderr, flush=True)
            if args.product:
                try:
                    out.update(product_companion(ctx))
                except Exception as exc:
                    out["product_error"] = "%s: %s" % (type(exc).__name__, exc)
                    print("bench.py: the product companion failed: %s" % exc, file=sys.stderr, flush=True)
        if valu:
            # second ceiling: the kernel is FP64-VALU bound.  PMC instruction count of this launch shape
            # (profiles/) over the live kernel time, against the fp64 issue rate measured on an MI355X
            # with tools/ubench/f64_rates.hip (2.47 ns per wave64 instruction per SIMD, 1024 SIMDs)
            rate = valu * 64 / (kernel_ms * 1e-3) / 1e12
            peak = 1024 * 64 / 2.47e-9 / 1e12
            out["fp64_issue"] = {"achieved": rate, "peak_measured": peak, "unit": "T lane-instr/s",
                                 "frac": rate / peak, "valu_wave_insts_per_launch": valu}
        if world == 1 and args.cpu_sample > 0 and wl["oracle"] is not None:
            ns = min(args.cpu_sample, nloc, wl.get("oracle_sample", nloc))
            # ~10 s of CPU work: whole passes over the sample until that much has been timed (at most five)
            t1 = time.perf_counter()
            cpu = wl["oracle"](slice(0, ns))
            cpu_s, passes = time.perf_counter() - t1, 1
            while cpu_s < 10.0 and passes < 5:
                t1 = time.perf_counter()
                wl["oracle"](slice(0, ns))
                cpu_s += time.perf_counter() - t1
                passes += 1
            err = float(np.max(np.abs(res_local[:ns] - cpu) / np.abs(cpu)))
            ncores = len(os.sched_getaffinity(0))
            out["cpu_baseline"] = {
                "value": passes * (ns / nwno_total) / cpu_s, "unit": "spectra/s", "cores": 1, "kind": "port",
                "sample": "%d pass(es) over %d of %d wavelengths of the same scene, oracle/ C restatement of the reference's "
                          "serial numba path on one core, %.1f s" % (passes, ns, nwno_total, cpu_s),
                "host_cores_available": ncores, "flags": orc_flags()}
            out["max_rel_err_vs_oracle"] = err
            # the same C restatement on ALL host cores: wavelength blocks on a thread pool (the ctypes
            # calls release the GIL; the reference itself is serial, so this is an upper bound on what
            # its algorithm could do on this host, not a measurement of the reference)
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, ncores)
            if nthr > 1:
                edges = np.linspace(0, ns, nthr + 1).astype(int)
                t1 = time.perf_counter()
                with ThreadPoolExecutor(nthr) as ex:
                    parts = list(ex.map(lambda j: wl["oracle"](slice(int(edges[j]), int(edges[j + 1]))),
                                        [j for j in range(nthr) if edges[j + 1] > edges[j]]))
                thr_s = time.perf_counter() - t1
                assert np.array_equal(np.concatenate(parts), cpu)
                out["cpu_baseline_threads"] = {
                    "value": (ns / nwno_total) / thr_s, "unit": "spectra/s", "cores": nthr, "kind": "port",
                    "sample": "same sample, %d wavelength blocks on %d threads (all host cores), %.2f s"
                              % (nthr, nthr, thr_s)}
        print(json.dumps(out), flush=True)
    if comm:
        comm.barrier()
        comm.destroy()
        group.close()


if __name__ == "__main__":
    main()
