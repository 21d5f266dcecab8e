"""World-size-2 (and 3, ragged) wavelength sharding over gloo on CPU: each rank solves only its
block of the grid and the all-gathered spectrum must be bit-identical to the unsharded result
(sharding changes no arithmetic: every column is independent).  The per-rank solve is the CPU
oracle here -- this test covers the sharding / collective logic that bench.py and multi-GPU
callers use, not the kernels (those are covered by the -m gpu tests)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nwno, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import oracle as orc
    from picaso_amd import disco, sharding
    from picaso_amd import synthetic as syn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlayer = 20
    sc = syn.make_scene(nlayer, nwno, seed=12)
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    lo, hi = sharding.shard_of(nwno, world, rank)
    names = ("dtau", "tau", "w0", "cosb", "gcos2", "ftau_cld", "ftau_ray", "dtau_og", "tau_og",
             "w0_og", "cosb_og")
    planes = [np.ascontiguousarray(sc[k][:, lo:hi]) for k in names]
    f0 = np.linspace(0.8, 1.2, nwno)
    x, _ = orc.get_reflected_1d(nlayer + 1, sc["wno"][lo:hi], hi - lo, 5, 1, *planes, 0.1, u0, u1,
                                1.0, f0[lo:hi], 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
    alb = orc.compress_disco(hi - lo, 1.0, x, gw, tw, f0[lo:hi])
    full_alb = sharding.all_gather_spectrum(alb, nwno, dist)
    full_x = sharding.all_gather_spectrum(x, nwno, dist)
    if rank == 0:
        q.put((full_alb, full_x))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nwno", [(2, 64), (3, 50)])
def test_sharded_equals_unsharded(world, nwno, oracle):
    from picaso_amd import disco
    from picaso_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, nwno, q)) for r in range(world)]
    for p in procs:
        p.start()
    full_alb, full_x = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nlayer = 20
    sc = syn.make_scene(nlayer, nwno, seed=12)
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    names = ("dtau", "tau", "w0", "cosb", "gcos2", "ftau_cld", "ftau_ray", "dtau_og", "tau_og",
             "w0_og", "cosb_og")
    f0 = np.linspace(0.8, 1.2, nwno)
    x, _ = oracle.get_reflected_1d(nlayer + 1, sc["wno"], nwno, 5, 1, *[sc[k] for k in names], 0.1,
                                   u0, u1, 1.0, f0, 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
    alb = oracle.compress_disco(nwno, 1.0, x, gw, tw, f0)
    assert full_alb.shape == (nwno,) and full_x.shape == x.shape
    assert np.array_equal(full_alb, alb)
    assert np.array_equal(full_x, x)


def test_shard_bounds():
    from picaso_amd.sharding import shard_bounds
    for n, w in ((100000, 8), (10, 3), (5, 8), (1, 1)):
        b = shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1
