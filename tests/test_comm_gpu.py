"""The library's RCCL layer (csrc/comm.hip) on the GPU box: a one-rank communicator built from a
library-generated id runs every collective entry point (librccl.so gets mapped and called), and the
sharded solve -- two contexts (streams) of one GPU each solving its wavelength block of one spectrum
into its slice of a shared result -- is bit-identical to the unsharded launch.  (More than one rank
needs more than one GPU: RCCL refuses two ranks on one device; bench.py --gpus N covers that on the
multi-GPU node.)"""
import ctypes
import os

import numpy as np
import pytest

from picaso_amd import _lib, device, disco, resident, sharding
from picaso_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TTHG = (1.0, -1.0, 2.0, -0.5, 1.0)


def test_one_rank_rccl_collectives():
    ctx = _lib.context(0)
    group = sharding.HostGroup(0, 1)
    comm = sharding.Comm.from_launcher(ctx, group)
    try:
        n = 4099
        host = np.linspace(-1.0, 1.0, n)
        send = device.DeviceArray.from_host(host, ctx)
        recv = device.DeviceArray.zeros((n,), ctx)
        comm.all_gather(send, recv, n)
        device.sync(ctx)
        assert np.array_equal(recv.to_host(), host)
        recv.zero()
        comm.all_gather_spectrum(send, recv, n)
        device.sync(ctx)
        assert np.array_equal(recv.to_host(), host)
        recv.zero()
        comm.all_gatherv(send, recv, [n], [0])
        device.sync(ctx)
        assert np.array_equal(recv.to_host(), host)
        # overlapped form: the gather is ordered behind the producer on the context's stream, later work of
        # that stream behind the gather (wait_slot); the producer of the next result may overlap it
        recv2 = device.DeviceArray.zeros((n,), ctx)
        sends = [device.DeviceArray.from_host(host * (it + 1), ctx) for it in range(6)]
        for it in range(6):
            slot = it & 1
            comm.wait_slot(slot)
            comm.all_gather_spectrum_async(sends[it], recv if slot == 0 else recv2, n, slot)
        comm.wait_slot(-1)
        device.sync(ctx)
        assert np.array_equal(recv.to_host(), host * 5) and np.array_equal(recv2.to_host(), host * 6)
        with pytest.raises(_lib.PicasoHipError):
            comm.all_gather_spectrum_async(send, recv, n, 7)
        # several spectra in one collective launch
        outs = [device.DeviceArray.zeros((n,), ctx) for _ in range(3)]
        comm.all_gather_spectra_async(sends[:3], outs, n, 2)
        comm.wait_slot(2)
        device.sync(ctx)
        for k in range(3):
            assert np.array_equal(outs[k].to_host(), host * (k + 1))
        assert comm.max(3.25) == 3.25
        comm.barrier()
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.check(_lib.load().picaso_comm_rank(comm.handle, ctypes.byref(r), ctypes.byref(w)), ctx)
        assert (r.value, w.value) == (0, 1)
    finally:
        comm.destroy()
    maps = open("/proc/self/maps").read()
    assert "librccl" in maps


def test_unique_ids_differ():
    lib = _lib.load()
    a, b = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
    _lib.check(lib.picaso_comm_unique_id(a), None)
    _lib.check(lib.picaso_comm_unique_id(b), None)
    assert a.raw != b.raw


@pytest.mark.parametrize("world", [2, 3])
def test_product_sharded_over_contexts_bit_identical(world):
    nlayer, nwno, ng = 40, 30011, 5
    nlevel = nlayer + 1
    gang, gw, tang, tw = disco.get_angles_1d(ng)
    u0, u1, _, _, _ = disco.compute_disco(ng, 1, gang, tang, 0.0)
    sc = syn.make_scene(nlayer, nwno, seed=21)
    sc["F0PI"] = np.linspace(0.5, 1.5, nwno)
    sc["surf_reflect"] = np.full(nwno, 0.2)
    keys = resident.REFLECTED_PLANES + ("F0PI", "surf_reflect")
    ctx0 = _lib.context(0)
    d = resident.upload_scene(sc, keys, ctx=ctx0)
    x = device.DeviceArray((ng, 1, nwno), ctx0)
    alb = device.DeviceArray((nwno,), ctx0)
    resident.reflected_1d(ctx0, nlevel, nwno, ng, 1, d, d["surf_reflect"], u0, u1, 1.0, d["F0PI"], 3, 0, *TTHG, x,
                          gweight=gw, tweight=tw, albedo=alb)
    device.sync(ctx0)
    want = alb.to_host()
    full = device.DeviceArray.zeros((nwno,), ctx0)
    ctxs = [ctx0] + [_lib.new_context(0) for _ in range(world - 1)]
    keep = []
    for r, (lo, hi) in enumerate(sharding.shard_bounds(nwno, world)):
        c = ctxs[r]
        dr = resident.upload_scene(sc, keys, lo, hi, ctx=c)
        xr = device.DeviceArray((ng, 1, hi - lo), c)
        resident.reflected_1d(c, nlevel, hi - lo, ng, 1, dr, dr["surf_reflect"], u0, u1, 1.0, dr["F0PI"], 3, 0,
                              *TTHG, xr, gweight=gw, tweight=tw, albedo=full.addr + 8 * lo)
        keep.append((dr, xr))
    for c in ctxs:
        device.sync(c)
    got = full.to_host()
    for dr, xr in keep:                      # arrays belong to their context: release them before it goes
        for v in dr.values():
            v.free()
        xr.free()
    for c in ctxs[1:]:
        _lib.destroy_context(c)
    assert np.array_equal(got, want)


def test_bench_self_spawned_rank_runs_the_sharded_path():
    """`python bench.py --gpus 1 --spawn`: bench.py's own launcher starts the rank as a subprocess with the
    environment it gives every rank of an N-GPU job; the rank rendezvous, builds the RCCL communicator inside the
    library, gathers in the timed region and prints the one JSON line with `per_rank` and the bit-identity checks --
    everything the driver's `bench.py --gpus N` does, with the one rank a 1-GPU box allows."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--spawn", "--steps", "4",
                        "--warmup", "2", "--nwno", "20000", "--prewarm-ms", "20", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 4 and len(out["per_rank"]) == 1
    assert out["checks"]["gathered_contains_local_shard"] is True
    assert out["checks"]["bit_identical_to_unsharded"] is True
    assert out["config"]["collective"].startswith("RCCL all-gather")
