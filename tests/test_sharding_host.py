"""Wavelength sharding, N > 1, on CPU: world sizes 2 and 3 (ragged blocks) as real processes.

The ranks find each other the way bench.py's do (picaso_amd.sharding.HostGroup: a TCP star on
127.0.0.1, no PyTorch, no MPI), rank 0's 128-byte communicator id reaches every rank intact, every
rank solves only its block of the grid, and the gathered spectrum is bit-identical to the unsharded
one (sharding changes no arithmetic: every column is independent).  The per-rank solve is the CPU
oracle here -- this covers the sharding / rendezvous / gather logic of the multi-GPU path; the
device collectives (RCCL inside libpicaso_hip.so) are exercised by tests/test_comm_gpu.py.
A second test runs the same shards through torch.distributed's gloo backend (test-side adapter only:
the product does not import torch) as an independent check of the block layout.
"""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("dtau", "tau", "w0", "cosb", "gcos2", "ftau_cld", "ftau_ray", "dtau_og", "tau_og", "w0_og", "cosb_og")


def _solve_block(nwno, lo, hi):
    from oracle import oracle as orc
    from picaso_amd import disco
    from picaso_amd import synthetic as syn
    nlayer = 20
    sc = syn.make_scene(nlayer, nwno, seed=12)
    g, gw, t, tw = disco.get_angles_1d(5)
    u0, u1, ct, _, _ = disco.compute_disco(5, 1, g, t, 0.0)
    planes = [np.ascontiguousarray(sc[k][:, lo:hi]) for k in NAMES]
    f0 = np.linspace(0.8, 1.2, nwno)
    x, _ = orc.get_reflected_1d(nlayer + 1, sc["wno"][lo:hi], hi - lo, 5, 1, *planes, 0.1, u0, u1, 1.0, f0[lo:hi],
                                3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
    return x, orc.compress_disco(hi - lo, 1.0, x, gw, tw, f0[lo:hi])


def _worker(rank, world, port, nwno, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.pop("PICASO_AMD_RDZV_PORT", None)
    from picaso_amd import sharding
    r, w, local, addr, rport = sharding.launcher_env()
    assert (r, w, local, addr, rport) == (rank, world, rank, "127.0.0.1", port + 37)
    group = sharding.HostGroup(r, w, addr, rport)
    uid = bytes((7 * i + 3) % 256 for i in range(sharding.COMM_ID_BYTES)) if rank == 0 else None
    uid = group.broadcast(uid)                       # the RCCL id's path (rank 0 -> all)
    lo, hi = sharding.shard_of(nwno, world, rank)
    x, alb = _solve_block(nwno, lo, hi)
    full_alb = group.all_gather_spectrum(alb, nwno)
    full_x = group.all_gather_spectrum(x, nwno)
    slowest = group.max(float(rank))
    group.barrier()
    q.put((rank, uid, full_alb, full_x, slowest))
    group.barrier()
    group.close()


@pytest.mark.parametrize("world,nwno", [(2, 64), (3, 50)])
def test_sharded_equals_unsharded_host_group(world, nwno, oracle):
    from picaso_amd import sharding
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23000 + (os.getpid() % 2000) + 40 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, nwno, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, alb = _solve_block(nwno, 0, nwno)
    want_uid = bytes((7 * i + 3) % 256 for i in range(sharding.COMM_ID_BYTES))
    assert sorted(g[0] for g in got) == list(range(world))
    for rank, uid, full_alb, full_x, slowest in got:
        assert uid == want_uid and len(uid) == 128
        assert full_alb.shape == (nwno,) and full_x.shape == x.shape
        assert np.array_equal(full_alb, alb)
        assert np.array_equal(full_x, x)
        assert slowest == float(world - 1)


def _rdzv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from picaso_amd import sharding
    group = sharding.HostGroup(rank, world, "127.0.0.1", port, timeout=60.0)
    got = group.broadcast(b"id-of-rank-0" if rank == 0 else None)
    vals = group.all_gather_bytes(bytes([rank]))
    q.put((rank, group.port, got, vals))
    group.barrier()
    group.close()


def test_rendezvous_survives_a_taken_port_and_stray_connections():
    """The first rendezvous port is held by a foreign listener (which answers nothing): rank 0 moves to the
    next candidate and the other ranks find it there; a connection that does not speak the handshake is
    dropped without disturbing the group."""
    import socket
    import threading
    import time
    world = 3
    port = 27000 + (os.getpid() % 2000)
    squatter = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    squatter.bind(("127.0.0.1", port))
    squatter.listen(8)
    stop = []

    def stray():                       # pokes the fall-back port with garbage while the group forms
        while not stop:
            try:
                s = socket.create_connection(("127.0.0.1", port + 1000), timeout=0.5)
                s.sendall(b"GET / HTTP/1.0\r\n\r\n")
                s.close()
            except OSError:
                pass
            time.sleep(0.05)
    t = threading.Thread(target=stray, daemon=True)
    t.start()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stop.append(1)
    squatter.close()
    assert sorted(g[0] for g in got) == [0, 1, 2]
    for rank, used, bid, vals in got:
        assert used == port + 1000
        assert bid == b"id-of-rank-0"
        assert vals == [bytes([r]) for r in range(world)]


def _gloo_worker(rank, world, port, nwno, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from picaso_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bounds = sharding.shard_bounds(nwno, world)
    lo, hi = bounds[rank]
    _, alb = _solve_block(nwno, lo, hi)
    nmax = max(b[1] - b[0] for b in bounds)
    pad = torch.zeros(nmax, dtype=torch.float64)
    pad[: hi - lo] = torch.from_numpy(alb)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    full = np.concatenate([out[r][: b[1] - b[0]].numpy() for r, b in enumerate(bounds)])
    if rank == 0:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_unsharded_gloo(oracle):
    world, nwno = 2, 33
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 25500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, nwno, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, alb = _solve_block(nwno, 0, nwno)
    assert np.array_equal(full, alb)


def test_shard_bounds():
    from picaso_amd.sharding import shard_bounds
    for n, w in ((100000, 8), (10, 3), (5, 8), (1, 1)):
        b = shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_product_does_not_import_torch():
    """north_star: host code calls the kernels through ctypes, no PyTorch anywhere in the product."""
    import re
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "picaso_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(import|from)\s+torch\b", text, re.M):
                    offenders.append(f)
    text = open(os.path.join(ROOT, "bench.py")).read()
    if re.search(r"^\s*(import|from)\s+torch\b", text, re.M):
        offenders.append("bench.py")
    assert not offenders, offenders


def _run_bench(*argv, timeout=180):
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PICASO_AMD_RDZV_PORT",
                        "PICASO_AMD_JOB_TOKEN")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks_and_fails_loudly_without_the_gpus():
    """`python bench.py --gpus 2` with no launcher environment starts two ranks itself; on a box with fewer
    than two GPUs every rank says so and the job exits non-zero -- it never runs one rank and calls it two."""
    import ctypes
    from picaso_amd import _lib
    r = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    ndev = _lib.device_count()
    if ndev >= 2:
        pytest.skip("two GPUs visible: the job would run")
    assert r.returncode != 0
    for rank in (0, 1):
        assert "bench.py [rank %d]: --gpus 2 needs 2 GPUs, %d visible" % (rank, ndev) in r.stderr
    assert r.stdout.strip() == ""          # no JSON line of a job that did not run


def test_bench_ranks_find_each_other_through_the_spawned_environment():
    import json
    r = _run_bench("--gpus", "3", "--rendezvous-only")
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["rendezvous"] == [0, 1, 2] and line["world"] == 3 and line["spawned_by_bench"] is True


def test_bench_refuses_a_launcher_world_that_is_not_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 2 but the launcher started WORLD_SIZE=1" in r.stderr


def _token_worker(rank, world, port, token, q):
    sys.path.insert(0, ROOT)
    from picaso_amd import sharding
    try:
        g = sharding.HostGroup(rank, world, "127.0.0.1", port, timeout=6.0, token=token)
        q.put((rank, "joined", g.all_gather_bytes(bytes([rank]))))
        g.close()
    except Exception as e:                       # noqa: BLE001
        q.put((rank, "refused", type(e).__name__))


def test_rendezvous_refuses_a_rank_of_another_job():
    """Two jobs share a node and a port: a rank carrying job B's token cannot complete job A's handshake (it is
    dropped by rank 0 and gives up), and job A still forms with its own ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 21000 + (os.getpid() % 2000)
    a0 = ctx.Process(target=_token_worker, args=(0, 2, port, b"job-A", q))
    intruder = ctx.Process(target=_token_worker, args=(1, 2, port, b"job-B", q))
    a0.start()
    intruder.start()
    first = q.get(timeout=60)
    assert first[0] == 1 and first[1] == "refused"            # the foreign rank never got in
    a1 = ctx.Process(target=_token_worker, args=(1, 2, port, b"job-A", q))
    a1.start()
    got = sorted([q.get(timeout=60), q.get(timeout=60)])
    for p in (a0, intruder, a1):
        p.join(timeout=30)
    assert [g[:2] for g in got] == [(0, "joined"), (1, "joined")]
    assert got[0][2] == [b"\x00", b"\x01"]
