"""Wavelength sharding across the GPUs of one node (one process per GPU, or one process driving
several GPUs).

Every function on the hot path is pointwise in wavelength, so the grid of ONE
spectrum is cut into contiguous blocks, one per rank, with no exchange inside the solve; the only
collective is the all-gather of the final spectrum shards, which runs inside ``libpicaso_hip.so``
(RCCL over xGMI, ``csrc/comm.hip``) on the context's own stream.  Replaces the reference's joblib
fan-out of independent spectra (reference ``justdoit.py:4774``).

No PyTorch, no MPI: the 128-byte RCCL id travels from rank 0 to the other ranks over a plain TCP
socket (``HostGroup``), using the ``RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT``
environment the one-process-per-GPU launcher provides.  ``HostGroup`` also
carries small host-side collectives (bytes broadcast, array all-gather), which is what the CPU tests
of the sharded path use.
"""
import ctypes
import hashlib
import os
import socket
import struct
import time

import numpy as np

from . import _lib

COMM_ID_BYTES = 128


def shard_bounds(nwno, world):
    """Contiguous [lo, hi) wavelength blocks, sizes differing by at most one."""
    base, rem = divmod(int(nwno), int(world))
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_of(nwno, world, rank):
    return shard_bounds(nwno, world)[rank]


def launcher_env():
    """(rank, world, local_rank, addr, port) from the launcher's environment (1 process: 0, 1, 0)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    # the launcher's own store listens on MASTER_PORT: take a fixed offset from it
    port = int(os.environ.get("PICASO_AMD_RDZV_PORT", str(int(os.environ.get("MASTER_PORT", "29500")) + 37)))
    return rank, world, local, addr, port


# ---------------------------------------------------------------------------------------------
# host side channel: a star over TCP with rank 0 in the middle
# ---------------------------------------------------------------------------------------------
def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("picaso_amd.sharding: peer closed the rendezvous socket")
        buf.extend(chunk)
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


_HELLO, _ACK = b"PZRV", b"PZOK"
PORT_STEPS = (0, 1000, 2000)          # fall-back ports (offsets) when the first one is taken by someone else
TOKEN_BYTES = 16
_LOOPBACK = ("127.0.0.1", "localhost", "::1")


def job_token():
    """16 bytes that identify THIS job at the rendezvous: two jobs on one node (or a stale listener of an
    earlier one) must not be able to complete each other's handshake.  ``PICASO_AMD_JOB_TOKEN`` (hex; bench.py's
    own launcher hands every rank a random one) wins; otherwise a hash of what the launcher gives every rank of
    one job and no rank of another: the store address, the world size and the launcher's run id."""
    tok = os.environ.get("PICASO_AMD_JOB_TOKEN")
    if tok:
        try:
            raw = bytes.fromhex(tok)
        except ValueError:
            raw = tok.encode()
        return hashlib.sha256(raw).digest()[:TOKEN_BYTES]
    ident = ":".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "TORCHELASTIC_RUN_ID",
                                                     "SLURM_JOB_ID", "SLURM_STEP_ID"))
    return hashlib.sha256(ident.encode()).digest()[:TOKEN_BYTES]


class HostGroup:
    """Rank 0 listens on ``port`` of ``addr`` (loopback for a single-node job; all interfaces only when
    ``addr`` is not an address of this host); ranks 1..world-1 connect to ``addr`` and identify themselves with
    their rank and the job token (``job_token()``), which rank 0 checks and echoes, so a rank of another job or a
    stale listener is refused on both sides.  Small host-side collectives only: the spectra themselves never
    travel through here on a GPU run.  If the port is in use by something else, rank 0 moves on to
    ``port + 1000`` and ``port + 2000``; the other ranks try the candidates in turn until one answers with the
    rendezvous handshake of THIS job.  Connections that do not start with the handshake are dropped."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29537, timeout=120.0, token=None):
        self.rank, self.world = int(rank), int(world)
        self.peers = {}          # rank 0: rank -> socket
        self.sock = None         # other ranks: socket to rank 0
        self.port = None
        if self.world == 1:
            return
        token = job_token() if token is None else hashlib.sha256(bytes(token)).digest()[:TOKEN_BYTES]
        ports = [port + d for d in PORT_STEPS if port + d < 65536]
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            bind_host = "127.0.0.1" if addr in _LOOPBACK else addr
            t0 = time.time()
            while self.port is None:         # a previous run's listener may still be closing
                for p in ports:
                    try:
                        srv.bind((bind_host, p))
                        self.port = p
                        break
                    except socket.gaierror:
                        bind_host = ""       # MASTER_ADDR does not resolve here: all interfaces (the token guards)
                    except OSError as e:
                        if e.errno == 99:    # EADDRNOTAVAIL: MASTER_ADDR is not an address of this host
                            bind_host = ""
                        continue
                if self.port is None:
                    if time.time() - t0 > 30.0:
                        raise OSError("picaso_amd.sharding: none of the rendezvous ports %s can be bound" % ports)
                    time.sleep(0.2)
            srv.listen(self.world + 8)
            srv.settimeout(timeout)
            while len(self.peers) < self.world - 1:
                conn, _ = srv.accept()
                try:
                    conn.settimeout(5.0)
                    hello = _recv_exact(conn, 8 + TOKEN_BYTES)
                    (r,) = struct.unpack("<i", hello[4:8])
                    if (hello[:4] != _HELLO or hello[8:] != token or r <= 0 or r >= self.world
                            or r in self.peers):
                        raise ConnectionError("not a rank of this job")
                    conn.sendall(_ACK + token)
                except (OSError, ConnectionError, struct.error):
                    conn.close()             # a stray connection: not ours
                    continue
                conn.settimeout(timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.peers[r] = conn
            srv.close()
        else:
            t0 = time.time()
            hosts = ["127.0.0.1"] if addr in _LOOPBACK else [addr]      # no blind loopback try for a remote rank 0
            while self.sock is None:
                for host in hosts:
                    for p in ports:
                        try:
                            s = socket.create_connection((host, p), timeout=2.0)
                            s.sendall(_HELLO + struct.pack("<i", self.rank) + token)
                            s.settimeout(5.0)
                            if _recv_exact(s, 4 + TOKEN_BYTES) != _ACK + token:
                                raise ConnectionError("no handshake of this job")
                        except (OSError, ConnectionError):
                            try:
                                s.close()
                            except Exception:
                                pass
                            continue
                        self.sock, self.port = s, p
                        break
                    if self.sock is not None:
                        break
                if self.sock is None:
                    if time.time() - t0 > timeout:
                        raise ConnectionError("picaso_amd.sharding: rank %d found no rendezvous at %s ports %s"
                                              % (self.rank, hosts, ports))
                    time.sleep(0.05)
            self.sock.settimeout(timeout)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def broadcast(self, payload=None):
        """bytes of rank 0 on every rank"""
        if self.world == 1:
            return payload
        if self.rank == 0:
            for r in sorted(self.peers):
                _send_msg(self.peers[r], payload)
            return payload
        return _recv_msg(self.sock)

    def gather(self, payload):
        """list of every rank's bytes on rank 0 (None elsewhere)"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            out = [payload] + [None] * (self.world - 1)
            for r in sorted(self.peers):
                out[r] = _recv_msg(self.peers[r])
            return out
        _send_msg(self.sock, payload)
        return None

    def all_gather_bytes(self, payload):
        parts = self.gather(payload)
        blob = None
        if self.rank == 0:
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
        blob = self.broadcast(blob)
        out, o = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, o)
            out.append(blob[o + 8:o + 8 + n])
            o += 8 + n
        return out

    def barrier(self):
        self.all_gather_bytes(b"")

    def max(self, value):
        vals = [struct.unpack("<d", b)[0] for b in self.all_gather_bytes(struct.pack("<d", float(value)))]
        return max(vals)

    def all_gather_spectrum(self, local, nwno):
        """Host arrays: per-rank shards (last axis = this rank's wavelength block) -> full spectrum."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        bounds = shard_bounds(nwno, self.world)
        lo, hi = bounds[self.rank]
        if local.shape[-1] != hi - lo:
            raise ValueError("rank %d holds %d wavelengths, its block is [%d, %d)" % (self.rank, local.shape[-1], lo, hi))
        parts = self.all_gather_bytes(local.tobytes())
        lead = local.shape[:-1]
        pieces = [np.frombuffer(p, dtype=np.float64).reshape(lead + (b[1] - b[0],)) for p, b in zip(parts, bounds)]
        return np.concatenate(pieces, axis=-1)

    def close(self):
        for s in self.peers.values():
            s.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = {}, None


# ---------------------------------------------------------------------------------------------
# device collectives: RCCL inside the library
# ---------------------------------------------------------------------------------------------
def _addr(x):
    return ctypes.c_void_p(int(x.addr if hasattr(x, "addr") else x))


class Comm:
    """RCCL communicator of the library bound to one context (one GPU).  ``Comm.from_launcher`` is the
    one-process-per-GPU form; ``Comm.init_all`` the single-process form."""

    def __init__(self, handle, ctx, rank, world, group=None):
        self.handle, self.ctx, self.rank, self.world, self.group = handle, ctx, rank, world, group

    @classmethod
    def from_launcher(cls, ctx, group):
        """Every rank calls this with its context and the HostGroup it shares with the others."""
        lib = _lib.load()
        uid = None
        if group.rank == 0:
            buf = ctypes.create_string_buffer(COMM_ID_BYTES)
            _lib.check(lib.picaso_comm_unique_id(buf), None)
            uid = buf.raw
        uid = group.broadcast(uid)
        if uid is None or len(uid) != COMM_ID_BYTES:
            raise _lib.PicasoHipError("picaso_amd.sharding: bad RCCL id at the rendezvous")
        h = ctypes.c_void_p()
        _lib.check(lib.picaso_comm_init_rank(ctx, ctypes.c_int(group.world), ctypes.c_int(group.rank),
                                             ctypes.c_char_p(uid), ctypes.byref(h)), ctx)
        return cls(h, ctx, group.rank, group.world, group)

    @classmethod
    def init_all(cls, ctxs):
        lib = _lib.load()
        n = len(ctxs)
        arr = (ctypes.c_void_p * n)(*[c.value if hasattr(c, "value") else c for c in ctxs])
        out = (ctypes.c_void_p * n)()
        _lib.check(lib.picaso_comm_init_all(ctypes.c_int(n), arr, out), ctxs[0])
        return [cls(ctypes.c_void_p(out[i]), ctxs[i], i, n) for i in range(n)]

    def all_gather(self, send, recv, count):
        """recv[r*count + i] = rank r's send[i] (device buffers; asynchronous on the context's stream)"""
        _lib.check(_lib.load().picaso_all_gather_dev(self.handle, _addr(send), _addr(recv),
                                                     ctypes.c_size_t(int(count))), self.ctx)

    def all_gatherv(self, send, recv, counts, displs):
        c = (ctypes.c_size_t * self.world)(*[int(x) for x in counts])
        d = (ctypes.c_size_t * self.world)(*[int(x) for x in displs])
        _lib.check(_lib.load().picaso_all_gatherv_dev(self.handle, _addr(send), _addr(recv), c, d), self.ctx)

    def all_gather_spectrum(self, local, full, nwno):
        """Per-rank shard (this rank's block of ``shard_bounds(nwno, world)``, device) -> ``full`` (nwno,
        device) on every rank.  Equal blocks use ncclAllGather, ragged ones the grouped-broadcast form."""
        bounds = shard_bounds(nwno, self.world)
        counts = [hi - lo for lo, hi in bounds]
        if len(set(counts)) == 1:
            self.all_gather(local, full, counts[0])
        else:
            self.all_gatherv(local, full, counts, [lo for lo, _ in bounds])

    def all_gather_spectrum_async(self, local, full, nwno, slot):
        """As ``all_gather_spectrum`` but on the communicator's own stream, behind everything enqueued on
        the context's stream so far: the next solve overlaps the gather.  ``wait_slot(slot)`` before the
        buffers of that slot are written or read again."""
        bounds = shard_bounds(nwno, self.world)
        counts = [hi - lo for lo, hi in bounds]
        if len(set(counts)) == 1:
            c = d = None
        else:
            c = (ctypes.c_size_t * self.world)(*counts)
            d = (ctypes.c_size_t * self.world)(*[lo for lo, _ in bounds])
        _lib.check(_lib.load().picaso_all_gather_async_dev(self.handle, _addr(local), _addr(full),
                                                           ctypes.c_size_t(counts[0]), c, d, ctypes.c_int(slot)),
                   self.ctx)

    def all_gather_spectra_async(self, locals_, fulls, nwno, slot):
        """Several spectra in one collective launch: ``locals_[i]`` (this rank's block) -> ``fulls[i]``."""
        bounds = shard_bounds(nwno, self.world)
        counts = [hi - lo for lo, hi in bounds]
        if len(set(counts)) == 1:
            c = d = None
        else:
            c = (ctypes.c_size_t * self.world)(*counts)
            d = (ctypes.c_size_t * self.world)(*[lo for lo, _ in bounds])
        n = len(locals_)
        snd = (ctypes.c_void_p * n)(*[int(x.addr) for x in locals_])
        rcv = (ctypes.c_void_p * n)(*[int(x.addr) for x in fulls])
        _lib.check(_lib.load().picaso_all_gather_multi_async_dev(self.handle, ctypes.c_int(n), snd, rcv,
                                                                 ctypes.c_size_t(counts[0]), c, d, ctypes.c_int(slot)),
                   self.ctx)

    def wait_slot(self, slot=-1):
        _lib.check(_lib.load().picaso_comm_wait_slot(self.handle, ctypes.c_int(slot)), self.ctx)

    def max(self, value):
        v = ctypes.c_double(float(value))
        _lib.check(_lib.load().picaso_comm_max(self.handle, ctypes.byref(v)), self.ctx)
        return v.value

    def barrier(self):
        _lib.check(_lib.load().picaso_comm_barrier(self.handle), self.ctx)

    def destroy(self):
        if self.handle:
            _lib.load().picaso_comm_destroy(self.handle)
            self.handle = None


class DeviceGroup:
    """One process driving several GPUs: one context per device, the communicators of ``picaso_comm_init_all``,
    and collectives that post every device's call inside one RCCL group (``picaso_all_gather_group_dev``) --
    a thread that issued them one communicator after the other would wait in the first call for peers it has
    not posted yet.  ``devices``: device indices, each at most once (RCCL takes one rank per device)."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        if len(set(self.devices)) != len(self.devices):
            raise _lib.PicasoHipError("DeviceGroup: every device at most once, got %s" % self.devices)
        ndev = _lib.device_count()
        if not self.devices or max(self.devices) >= ndev or min(self.devices) < 0:
            raise _lib.PicasoHipError("DeviceGroup: devices %s but %d GPU(s) visible" % (self.devices, ndev))
        self.ctxs = [_lib.context(d) for d in self.devices]
        lib = _lib.load()
        n = len(self.ctxs)
        arr = (ctypes.c_void_p * n)(*[c.value for c in self.ctxs])
        out = (ctypes.c_void_p * n)()
        _lib.check(lib.picaso_comm_init_all(ctypes.c_int(n), arr, out), self.ctxs[0])
        self._handles = (ctypes.c_void_p * n)(*[out[i] for i in range(n)])
        self.world = n

    def all_gather_spectrum(self, locals_, fulls, nwno):
        """``locals_[i]`` (device i's block of ``shard_bounds(nwno, n)``) -> ``fulls[i]`` (nwno) on every device."""
        n = self.world
        bounds = shard_bounds(nwno, n)
        counts = [hi - lo for lo, hi in bounds]
        if len(set(counts)) == 1:
            c = d = None
        else:
            c = (ctypes.c_size_t * n)(*counts)
            d = (ctypes.c_size_t * n)(*[lo for lo, _ in bounds])
        snd = (ctypes.c_void_p * n)(*[int(x.addr) for x in locals_])
        rcv = (ctypes.c_void_p * n)(*[int(x.addr) for x in fulls])
        _lib.check(_lib.load().picaso_all_gather_group_dev(ctypes.c_int(n), self._handles, snd, rcv,
                                                           ctypes.c_size_t(counts[0]), c, d), self.ctxs[0])

    def max(self, values):
        n = self.world
        v = (ctypes.c_double * n)(*[float(x) for x in values])
        _lib.check(_lib.load().picaso_comm_group_max(ctypes.c_int(n), self._handles, v), self.ctxs[0])
        return [v[i] for i in range(n)]

    def barrier(self):
        _lib.check(_lib.load().picaso_comm_group_barrier(ctypes.c_int(self.world), self._handles), self.ctxs[0])

    def destroy(self):
        if self._handles is not None:
            for h in self._handles:
                if h:
                    _lib.load().picaso_comm_destroy(ctypes.c_void_p(h))
            self._handles = None


_device_groups = {}


def device_group(devices):
    """The process's DeviceGroup for this device list (communicators are created once and kept)."""
    key = (os.getpid(), tuple(int(d) for d in devices))
    if key not in _device_groups:
        _device_groups[key] = DeviceGroup(devices)
    return _device_groups[key]
