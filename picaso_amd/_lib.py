"""ctypes binding of the C ABI declared in ``include/picaso_hip.h``.

The HIP library is the product: if ``libpicaso_hip.so`` is missing, or no MI355X is visible when
a compute entry point is called, this module raises -- there is no CPU fallback.
"""
import ctypes
import functools
import os
import re
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PICASO_AMD_LIB: another build of the same sources (an A/B build kept beside the current one)
LIB_PATH = os.environ.get("PICASO_AMD_LIB") or os.path.join(_HERE, "libpicaso_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "picaso_hip.h")

c_double_p = ctypes.POINTER(ctypes.c_double)
_lib = None
_ctx = {}


class PicasoHipError(Exception):
    pass


def declared_symbols():
    """Every function name include/picaso_hip.h declares."""
    with open(HEADER) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(picaso_[A-Za-z0-9_]+)\s*\(", text)))


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (the SONAME this
    library links against); two copies of the runtime in one process cannot both own the GPUs (the second
    reports "No HIP GPUs are available").  The product never imports PyTorch -- but when the CALLER's process
    already has (``torch`` in ``sys.modules``), its runtime is mapped first so that the dynamic loader resolves
    our dependency to that same copy.  Nothing happens in a process without torch.  (A caller that imports torch
    AFTER this library gets two runtimes: import torch first; INTEGRATION.md.)"""
    import sys
    torch = sys.modules.get("torch")
    if torch is None or getattr(getattr(torch, "version", None), "hip", None) is None:
        return
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PicasoHipError(
            "picaso_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(LIB_PATH)
    lib.picaso_last_error.restype = ctypes.c_char_p
    lib.picaso_last_error.argtypes = [ctypes.c_void_p]
    lib.picaso_version.restype = ctypes.c_char_p
    lib.picaso_stream.restype = ctypes.c_void_p
    lib.picaso_stream.argtypes = [ctypes.c_void_p]
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc != 0:
        msg = load().picaso_last_error(ctx)
        raise PicasoHipError(msg.decode() if msg else "picaso_hip error %d" % rc)


def stream_ptr(ctx):
    """The hipStream_t all of this context's kernels run on, as an integer."""
    return int(load().picaso_stream(ctx))


def device_count():
    n = ctypes.c_int(0)
    rc = load().picaso_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


# The HIP runtime does not survive fork(): a child of a process that has already created a context segfaults in its first
# HIP call (measured: exit status 139 from the first kernel launch of a forked child) -- what multiprocessing's default
# start method on Linux gives a retrieval that builds its opacity object before starting the pool.  Importing this
# package before a fork is fine (nothing touches the GPU until a context exists); using the GPU on both sides is not,
# and is a clean error here instead of a dead worker.
_gpu_pid = None           # the process that created this module's first context


def _check_not_forked():
    if _gpu_pid is not None and os.getpid() != _gpu_pid:
        raise PicasoHipError(
            "picaso_amd: this process (pid %d) was forked from pid %d after that one had started using the GPU; the HIP "
            "runtime does not survive fork().  Start the workers with multiprocessing.get_context('spawn') (or joblib's "
            "default loky backend), or create the opacity object inside the worker." % (os.getpid(), _gpu_pid))


def _mark_gpu_used():
    global _gpu_pid
    _check_not_forked()
    if _gpu_pid is None:
        _gpu_pid = os.getpid()


def context(device=None):
    """Per-process, per-device context (lazy: safe to import before fork)."""
    if device is None:
        device = int(os.environ.get("PICASO_AMD_DEVICE", "0"))
    key = (os.getpid(), device)
    if key not in _ctx:
        _mark_gpu_used()
        lib = load()
        h = ctypes.c_void_p()
        rc = lib.picaso_ctx_create(ctypes.c_int(device), ctypes.byref(h))
        if rc != 0:
            raise PicasoHipError("picaso_amd needs an MI355X (gfx950) GPU: %s"
                                 % lib.picaso_last_error(None).decode())
        _ctx[key] = h
    return _ctx[key]


# One product call at a time per process.  A spectrum is dozens of C calls that share the context's stream, staging
# arena and block pool and the workspaces kept with the opacity object; two Python threads inside spectrum() at once
# interleave them (ctypes releases the GIL during a C call) and read each other's planes -- measured: wrong spectra
# and "block still has uncollected results" (tools/scratch/threads_probe.py).  The reference fans out with PROCESSES
# (joblib, justdoit.py:4774); here threads are safe but serialised -- parallelism is the batch axis, devices=N, or one
# process per GPU.  Re-entrant: the public functions call each other.
CALL_LOCK = threading.RLock()


def serialized(fn):
    """Decorator of the public entry points: the call runs under ``CALL_LOCK``."""
    @functools.wraps(fn)
    def locked(*args, **kwargs):
        _check_not_forked()
        with CALL_LOCK:
            return fn(*args, **kwargs)
    return locked


def new_context(device=None):
    """An additional context (its own HIP stream, arenas and block cache) on the device, e.g. to keep
    two independent spectra in flight.  Device memory may be read from any context's stream."""
    if device is None:
        device = int(os.environ.get("PICASO_AMD_DEVICE", "0"))
    _mark_gpu_used()
    h = ctypes.c_void_p()
    rc = load().picaso_ctx_create(ctypes.c_int(device), ctypes.byref(h))
    if rc != 0:
        raise PicasoHipError(load().picaso_last_error(None).decode())
    return h


_aux = {}


def aux_context(device=None, index=0):
    """The process's second context on the device (its own stream): the thermal leg of a spectrum runs
    there next to the reflected leg.  Created once per (process, device) and kept, like ``context()``.
    ``index``: a further one per wavelength block when one device carries several blocks of a spectrum
    (``devices=[0, 0, ...]``) -- with one stream for all of them their thermal legs would run one after the other."""
    if device is None:
        device = int(os.environ.get("PICASO_AMD_DEVICE", "0"))
    key = (os.getpid(), device, int(index))
    if key not in _aux:
        _aux[key] = new_context(device)
    return _aux[key]


_destroy_hooks = []


def on_context_destroy(hook):
    """Register ``hook(ctx_value)``: called by ``destroy_context`` so that module-level caches of DeviceArrays drop what
    they hold for that context (a new context may be created at the same address later)."""
    _destroy_hooks.append(hook)


def destroy_context(ctx):
    """Free a context made by ``new_context`` (``picaso_ctx_destroy``) after telling the caches that key device
    memory by context.  The per-process contexts of ``context()`` / ``aux_context()`` live as long as the process."""
    value = getattr(ctx, "value", ctx)
    for hook in _destroy_hooks:
        hook(value)
    for table in (_ctx, _aux):
        for k in [k for k, v in table.items() if getattr(v, "value", v) == value]:
            del table[k]
    load().picaso_ctx_destroy(ctx)


def device_of(ctx):
    d = ctypes.c_int(0)
    check(load().picaso_ctx_device(ctx, ctypes.byref(d)), ctx)
    return d.value


def ctx_wait(waiter, signaller):
    """Work enqueued on ``waiter`` from now on starts after everything enqueued on ``signaller`` so far
    (device-side ordering between two contexts' streams; the host does not block)."""
    check(load().picaso_ctx_wait(waiter, signaller), waiter)


def f64(x, shape=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        a = np.ascontiguousarray(np.broadcast_to(a, shape))
    return a


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(c_double_p)
    return ctypes.cast(ctypes.c_void_p(int(a)), c_double_p)   # raw device address


def per_wave(x, nwno):
    """Scalar-or-(nwno) array -> (nwno) float64 array (the reference accepts both)."""
    return f64(np.zeros(nwno) + np.asarray(x, dtype=np.float64))
