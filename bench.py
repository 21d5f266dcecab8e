#!/usr/bin/env python
"""Headline benchmark: spectra/s of a 1e5-wavelength x 90-layer Toon reflected-light spectrum
(BASELINE.json configs[2]; 5 Gauss angles, TTHG_ray + N=2 + delta-Eddington, fused disk
integration) with all input planes resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full spectrum per GPU (get_reflected_1d + compress_disco over the rank's
wavelength shard).  N > 1 is weak scaling: the wavelength grid is N x 1e5 points, each rank owns
a contiguous 1e5-point shard, and the albedo shards are collected with one RCCL all-gather
inside the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from picaso_amd import _lib, device, disco, resident  # noqa: E402
from picaso_amd import synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)   # frac_a, frac_b, frac_c, constant_back, constant_forward


def algorithmic_bytes(nwno, nlayer, nang, with_albedo=True):
    """9 layer planes + 2 level planes + F0PI + surf_reflect read once,
    xint_at_top (+ albedo) written once."""
    nlevel = nlayer + 1
    return 8 * nwno * (9 * nlayer + 2 * nlevel + 2 + nang + (1 if with_albedo else 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nwno", type=int, default=100000, help="wavelengths per GPU")
    ap.add_argument("--nlayer", type=int, default=90)
    ap.add_argument("--ngauss", type=int, default=5, help="disk Gauss angles (5..8)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only for smoke "
                         "tests of the sharded path on a box with fewer GPUs than ranks)")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the same K spectra issued round-robin on two streams (extra JSON object; "
                         "off by default so that a rocprofv3 trace of the default run holds only the "
                         "single-stream launches the roofline refers to)")
    ap.add_argument("--cpu-sample", type=int, default=100000,
                    help="wavelengths of the same workload timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    dist = torch = None
    launched = "RANK" in os.environ          # under torch.distributed.run (also with one rank)
    if launched:
        # torch first: PyTorch-ROCm bundles its own HIP runtime and the process must hold exactly one
        # (picaso_amd/_lib.py then binds libpicaso_hip.so to the copy torch has mapped)
        import torch
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    dev = local_rank if args.backend == "nccl" else local_rank % ndev

    ctx = _lib.context(dev)
    nwno, nlayer, nlevel = args.nwno, args.nlayer, args.nlayer + 1
    ng = args.ngauss
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    ubar0, ubar1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    cos_theta = 1.0     # symmetric 1-D geometry (reference justdoit.py:1532)

    # ---- synthetic shard of the (world*nwno)-point grid, built on the host, uploaded once ----
    scene = syn.make_scene(nlayer, nwno, seed=3 + 1000 * rank)
    scene["F0PI"] = np.ones(nwno)
    scene["surf_reflect"] = np.zeros(nwno)
    d = resident.upload_scene(scene, resident.REFLECTED_PLANES + ("F0PI", "surf_reflect"), ctx=ctx)
    xint = device.DeviceArray((ng, 1, nwno), ctx)
    use_nccl = launched and args.backend == "nccl"
    if use_nccl:
        # two result buffers: the gather of one spectrum overlaps the solve of the next
        alb_tt = [torch.empty(nwno, dtype=torch.float64, device="cuda") for _ in range(2)]
        full_tt = [torch.empty(world * nwno, dtype=torch.float64, device="cuda") for _ in range(2)]
        pending = [None, None]
        alb_t, full_t = alb_tt[0], full_tt[0]
        albedo = alb_t.data_ptr()
    else:
        alb_d = device.DeviceArray((nwno,), ctx)
        albedo = alb_d
        if launched:
            full_t = torch.empty(world * nwno, dtype=torch.float64)

    nstep = [0]

    def step():
        if use_nccl:
            # The library's stream is torch's current stream (ExternalStream below).  The gather of
            # spectrum i runs on RCCL's own stream after the kernel that produced it (async_op: the
            # compute stream does not wait for it), so it overlaps the solve of spectrum i+1, which
            # writes the other result buffer; a buffer is reused only after its previous gather has
            # finished -- a device-side wait, no host synchronisation inside the timed loop.
            b = nstep[0] & 1
            nstep[0] += 1
            if pending[b] is not None:
                pending[b].wait()
            resident.reflected_1d(ctx, nlevel, nwno, ng, 1, d, d["surf_reflect"], ubar0, ubar1,
                                  cos_theta, d["F0PI"], 3, 0, *TTHG, xint, toon_coefficients=0,
                                  b_top=0.0, gweight=gw, tweight=tw, albedo=alb_tt[b].data_ptr())
            pending[b] = dist.all_gather_into_tensor(full_tt[b], alb_tt[b], async_op=True)   # RCCL over xGMI
            return
        resident.reflected_1d(ctx, nlevel, nwno, ng, 1, d, d["surf_reflect"], ubar0, ubar1,
                              cos_theta, d["F0PI"], 3, 0, *TTHG, xint, toon_coefficients=0,
                              b_top=0.0, gweight=gw, tweight=tw, albedo=albedo)
        if launched:                                          # gloo smoke path: gather on the host
            dist.all_gather_into_tensor(full_t, torch.from_numpy(alb_d.to_host()))

    def drain():
        if use_nccl:
            for k in range(2):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None

    def barrier():
        device.sync(ctx)
        if launched:
            if use_nccl:
                torch.cuda.synchronize()
            dist.barrier()
            if use_nccl:
                torch.cuda.synchronize()

    import contextlib
    on_lib_stream = contextlib.nullcontext()
    if use_nccl:
        on_lib_stream = torch.cuda.stream(torch.cuda.ExternalStream(_lib.stream_ptr(ctx)))
    with on_lib_stream:
        for _ in range(args.warmup):
            step()
        drain()
        barrier()
        device.timer_start(ctx)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        kernel_ms_total = device.timer_stop(ctx)              # HIP events on the kernel's stream
        drain()                                               # every spectrum gathered ...
        barrier()                                             # ... and every rank done
        elapsed = time.perf_counter() - t0
    if launched:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if use_nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity + CPU baseline on rank 0 (outside the timed region) ----
    out = None
    if rank == 0:
        if use_nccl:
            last = (nstep[0] - 1) & 1
            alb_t, full_t = alb_tt[last], full_tt[last]
        alb_gpu = alb_t.cpu().numpy() if use_nccl else alb_d.to_host()
        if launched:    # the gathered spectrum must contain this rank's shard bit-exactly
            assert np.array_equal(full_t[:nwno].cpu().numpy(), alb_gpu), "all-gather mismatch"
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.steps / elapsed
        nang = ng
        abytes = algorithmic_bytes(nwno, nlayer, nang)
        kernel_ms = kernel_ms_total / args.steps
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        traffic = valu = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                prof = json.load(open(tfile))
                traffic, valu = prof.get("hbm_bytes_per_launch"), prof.get("valu_wave_insts_per_launch")
            except Exception:
                traffic = valu = None
        out = {
            "metric": "spectra/sec (1e5 wave x 90 layer reflected)",
            "value": value, "unit": "spectra/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Toon two-stream reflected light "
                                   "(get_reflected_1d + compress_disco), TTHG_ray, N=2, "
                                   "delta-Eddington, Rayleigh + cloud slab",
                       "nwno_per_gpu": nwno, "nlayer": nlayer, "gauss_angles": ng,
                       "sharding": "wavelength blocks, %d x %d" % (world, nwno),
                       "collective": ("rccl all_gather of albedo shards" if use_nccl else
                                      "gloo all_gather (smoke)") if launched else "none"},
            "wavelength_layer_updates_per_s": value * nwno * nlayer,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_reflected_toa<%d,false>" % nang,
                         "kernel_ms": kernel_ms, "algorithmic_bytes": abytes},
        }
        if valu and nwno == 100000 and nlayer == 90 and ng == 5:
            # second ceiling: the kernel is FP64-VALU bound.  PMC instruction count of this launch shape
            # (profiles/) over the live kernel time, against the fp64 issue rate measured on an MI355X
            # with tools/ubench/f64_rates.hip (2.47 ns per wave64 instruction per SIMD, 1024 SIMDs)
            rate = valu * 64 / (kernel_ms * 1e-3) / 1e12
            peak = 1024 * 64 / 2.47e-9 / 1e12
            out["fp64_issue"] = {"achieved": rate, "peak_measured": peak, "unit": "T lane-instr/s",
                                 "frac": rate / peak, "valu_wave_insts_per_launch": valu}
        if world == 1 and not launched and args.pipelined:
            # Not the headline: the same K spectra issued round-robin on two streams (two library
            # contexts).  A 1e5-column spectrum is 1.5 waves per SIMD, so on one stream half the SIMDs
            # idle through the tail of every launch; with a second spectrum in flight they do not.
            # Per-kernel durations (what rocprofv3 reports) get longer, spectra per second go up.
            ctx2 = _lib.new_context(dev)
            x2, a2 = device.DeviceArray((ng, 1, nwno), ctx2), device.DeviceArray((nwno,), ctx2)
            lanes = [(ctx, xint, alb_d), (ctx2, x2, a2)]

            def step2(j):
                c, xo, ao = lanes[j % 2]
                resident.reflected_1d(c, nlevel, nwno, ng, 1, d, d["surf_reflect"], ubar0, ubar1, cos_theta,
                                      d["F0PI"], 3, 0, *TTHG, xo, toon_coefficients=0, b_top=0.0, gweight=gw,
                                      tweight=tw, albedo=ao)
            for j in range(4):
                step2(j)
            device.sync(ctx); device.sync(ctx2)
            t2 = time.perf_counter()
            for j in range(args.steps):
                step2(j)
            device.sync(ctx); device.sync(ctx2)
            e2 = time.perf_counter() - t2
            assert np.array_equal(a2.to_host(), alb_gpu)
            out["pipelined_2streams"] = {"value": args.steps / e2, "unit": "spectra/s",
                                         "ms_per_step": 1e3 * e2 / args.steps,
                                         "hbm_frac_throughput": abytes * args.steps / e2 / 1e9 / HBM_PEAK_GBS}
        if world == 1 and args.cpu_sample > 0:
            from oracle import oracle as orc
            ns = min(args.cpu_sample, nwno)
            sl = slice(0, ns)
            planes = [np.ascontiguousarray(scene[k][:, sl]) for k in resident.REFLECTED_PLANES]
            t1 = time.perf_counter()
            xo, _ = orc.get_reflected_1d(nlevel, scene["wno"][sl], ns, ng, 1, *planes, 0.0, ubar0,
                                         ubar1, cos_theta, np.ones(ns), 3, 0, *TTHG)
            alb_cpu = orc.compress_disco(ns, cos_theta, xo, gw, tw, np.ones(ns))
            cpu_s = time.perf_counter() - t1
            err = float(np.max(np.abs(alb_gpu[sl] - alb_cpu) / np.abs(alb_cpu)))
            out["cpu_baseline"] = {
                "value": (ns / nwno) / cpu_s, "unit": "spectra/s", "cores": 1, "kind": "port",
                "sample": "%d of %d wavelengths of the same scene, oracle/picaso_oracle.c "
                          "(single-thread C restatement of the reference's serial numba path), "
                          "%.1f s" % (ns, nwno, cpu_s),
                "host_cores_available": len(os.sched_getaffinity(0))}
            out["max_rel_err_vs_oracle"] = err
            # the same C restatement on many host cores: wavelength blocks on a thread pool (the
            # ctypes calls release the GIL; the reference itself is serial, so this is an upper bound on
            # what its algorithm could do on this host, not a measurement of the reference)
            from concurrent.futures import ThreadPoolExecutor
            nthr = min(64, len(os.sched_getaffinity(0)))
            if nthr > 1:
                edges = np.linspace(0, ns, nthr + 1).astype(int)

                def work(j):
                    a_, b_ = edges[j], edges[j + 1]
                    pl = [np.ascontiguousarray(scene[k][:, a_:b_]) for k in resident.REFLECTED_PLANES]
                    x_, _ = orc.get_reflected_1d(nlevel, scene["wno"][a_:b_], b_ - a_, ng, 1, *pl, 0.0, ubar0,
                                                 ubar1, cos_theta, np.ones(b_ - a_), 3, 0, *TTHG)
                    return orc.compress_disco(b_ - a_, cos_theta, x_, gw, tw, np.ones(b_ - a_))
                t1 = time.perf_counter()
                with ThreadPoolExecutor(nthr) as ex:
                    parts = list(ex.map(work, range(nthr)))
                thr_s = time.perf_counter() - t1
                assert np.array_equal(np.concatenate(parts), alb_cpu)
                out["cpu_baseline_threads"] = {"value": (ns / nwno) / thr_s, "unit": "spectra/s", "cores": nthr,
                                               "kind": "port", "sample": "same sample, %d wavelength blocks on "
                                               "%d threads, %.2f s" % (nthr, nthr, thr_s)}
        print(json.dumps(out), flush=True)
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
