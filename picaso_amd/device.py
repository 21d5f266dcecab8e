"""Device-resident buffers for callers that keep planes in HBM across calls (bench, sharded runs)."""
import ctypes

import numpy as np

from . import _lib


class DeviceArray:
    """A float64 array living in HBM, owned by the library's context."""

    def __init__(self, shape, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.context()
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.size = int(np.prod(self.shape))
        self.nbytes = self.size * 8
        p = ctypes.c_void_p()
        _lib.check(_lib.load().picaso_dev_malloc(self.ctx, ctypes.c_size_t(self.nbytes),
                                                 ctypes.byref(p)), self.ctx)
        self.addr = p.value

    @classmethod
    def from_host(cls, arr, ctx=None):
        a = _lib.f64(arr)
        d = cls(a.shape, ctx)
        _lib.check(_lib.load().picaso_memcpy_h2d(d.ctx, ctypes.c_void_p(d.addr), _lib.ptr(a),
                                                 ctypes.c_size_t(d.nbytes)), d.ctx)
        return d

    @classmethod
    def from_host_columns(cls, arr, w_lo, w_hi, ctx=None):
        """Upload the wavelength shard ``arr[..., w_lo:w_hi]`` of a (rows, nwno) plane."""
        a = _lib.f64(arr)
        rows, nwno = int(np.prod(a.shape[:-1])), a.shape[-1]
        n = w_hi - w_lo
        d = cls(a.shape[:-1] + (n,), ctx)
        src = ctypes.c_void_p(a.ctypes.data + 8 * w_lo)
        _lib.check(_lib.load().picaso_memcpy_h2d_2d(
            d.ctx, ctypes.c_void_p(d.addr), ctypes.c_size_t(8 * n), src, ctypes.c_size_t(8 * nwno),
            ctypes.c_size_t(8 * n), ctypes.c_size_t(rows)), d.ctx)
        return d

    def columns_to_host(self, w_lo, w_hi):
        """Host copy of the wavelength block ``self[:, w_lo:w_hi, ...]`` of a resident
        ``(rows, nwno[, inner...])`` plane (strided rows; 64-bit offsets: a 64-facet plane at 1e5 wavelengths is
        4.7 GB)."""
        rows, nwno = self.shape[0], self.shape[1]
        inner = int(np.prod(self.shape[2:])) if len(self.shape) > 2 else 1
        n = int(w_hi) - int(w_lo)
        out = np.empty((rows, n) + tuple(self.shape[2:]), dtype=np.float64)
        src = ctypes.c_void_p(self.addr + 8 * int(w_lo) * inner)
        _lib.check(_lib.load().picaso_memcpy_d2h_2d(
            self.ctx, _lib.ptr(out), ctypes.c_size_t(8 * n * inner), src, ctypes.c_size_t(8 * nwno * inner),
            ctypes.c_size_t(8 * n * inner), ctypes.c_size_t(rows)), self.ctx)
        return out

    def to_host(self):
        out = np.empty(self.shape, dtype=np.float64)
        _lib.check(_lib.load().picaso_memcpy_d2h(self.ctx, _lib.ptr(out), ctypes.c_void_p(self.addr),
                                                 ctypes.c_size_t(self.nbytes)), self.ctx)
        return out

    def to_host_async(self, pinned, ctx=None):
        """Enqueue the copy into ``pinned`` (a ``PinnedArray`` of this size) behind the kernels already on the stream
        (of ``ctx``: another context of the same device, ordered behind this buffer's producer by the caller; default the
        buffer's own) and return at once; ``pinned.wait()`` blocks until the copy -- not the stream -- is done."""
        if pinned.nbytes != self.nbytes:
            raise ValueError("pinned block of %d bytes for a result of %d" % (pinned.nbytes, self.nbytes))
        ctx = ctx if ctx is not None else self.ctx
        mark = ctypes.c_void_p()
        _lib.check(_lib.load().picaso_memcpy_d2h_async(ctx, ctypes.c_void_p(pinned.addr), ctypes.c_void_p(self.addr),
                                                       ctypes.c_size_t(self.nbytes), ctypes.byref(mark)), ctx)
        pinned._mark, pinned._mark_ctx = mark, ctx
        return pinned

    @classmethod
    def zeros(cls, shape, ctx=None):
        d = cls(shape, ctx)
        d.zero()
        return d

    def zero(self):
        _lib.check(_lib.load().picaso_memset(self.ctx, ctypes.c_void_p(self.addr), 0,
                                             ctypes.c_size_t(self.nbytes)), self.ctx)

    def row_block(self, index):
        """Non-owning view of ``self[index]`` (leading axis) -- e.g. one facet of a facet-major stack."""
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape = self.ctx, self.shape[1:]
        v.size = int(np.prod(v.shape))
        v.nbytes = v.size * 8
        v.addr = self.addr + int(index) * v.nbytes
        v._owner = self            # keeps the parent alive; views are never freed
        return v

    def __deepcopy__(self, memo):
        """A DeviceArray inside a container that is deep-copied (``copy.deepcopy(case)``): the copy is a non-owning view
        of the same allocation that keeps the original alive -- two owners of one device pointer would free it twice."""
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape, v.size, v.nbytes, v.addr = self.ctx, self.shape, self.size, self.nbytes, self.addr
        v._owner = self
        return v

    __copy__ = lambda self: self.__deepcopy__(None)

    def __reduce__(self):
        """Device memory does not travel between processes: an opacity object (or anything else holding resident
        tables) handed to a ``multiprocessing`` / joblib worker fails HERE, with the remedy, not in ctypes."""
        raise TypeError("picaso_amd: a DeviceArray (resident tables of an opacity object, device planes) cannot be "
                        "pickled -- device memory belongs to the process that allocated it.  Create the opacity object "
                        "inside the worker process (opannection(...) there), and send inputs / results, which are "
                        "plain numpy.")

    def row_range(self, lo, hi):
        """Non-owning view of ``self[lo:hi]`` (leading axis)."""
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape = self.ctx, (int(hi) - int(lo),) + tuple(self.shape[1:])
        v.size = int(np.prod(v.shape))
        v.nbytes = v.size * 8
        v.addr = self.addr + int(lo) * (self.nbytes // int(self.shape[0]))
        v._owner = self
        return v

    def head(self, n):
        """Non-owning view of the first ``n`` elements of a 1-D buffer (a result vector allocated with room for a
        spectrum-wide integral behind it)."""
        if len(self.shape) != 1 or not 0 < int(n) <= self.size:
            raise ValueError("head(%r) of a buffer of shape %s" % (n, self.shape))
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape, v.size, v.nbytes, v.addr = self.ctx, (int(n),), int(n), int(n) * 8, self.addr
        v._owner = self
        return v

    def reshape(self, shape):
        """Non-owning view of the same buffer under another shape of equal size."""
        shape = tuple(int(x) for x in shape)
        if int(np.prod(shape)) != self.size:
            raise ValueError("cannot reshape %s into %s" % (self.shape, shape))
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape, v.size, v.nbytes, v.addr = self.ctx, shape, self.size, self.nbytes, self.addr
        v._owner = self
        return v

    def free(self):
        if self.addr and not hasattr(self, "_owner"):
            _lib.load().picaso_dev_free(self.ctx, ctypes.c_void_p(self.addr))
        self.addr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A float64 block of pinned host memory owned by the library's context (``picaso_host_alloc``): the target of
    ``DeviceArray.to_host_async``.  ``array`` is a numpy view of it, valid until ``free()`` / collection."""

    def __init__(self, shape, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.context()
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.nbytes = int(np.prod(self.shape)) * 8
        p = ctypes.c_void_p()
        _lib.check(_lib.load().picaso_host_alloc(self.ctx, ctypes.c_size_t(self.nbytes), ctypes.byref(p)), self.ctx)
        self.addr = p.value
        self._mark = None
        buf = (ctypes.c_double * (self.nbytes // 8)).from_address(self.addr)
        self.array = np.frombuffer(buf, dtype=np.float64).reshape(self.shape)

    def wait(self):
        """Block until the copy enqueued by ``to_host_async`` has landed; returns the numpy view."""
        if self._mark is not None:
            mark, self._mark = self._mark, None
            _lib.check(_lib.load().picaso_mark_wait(self._mark_ctx, mark), self._mark_ctx)
        return self.array

    def free(self):
        if self.addr:
            if self._mark is not None:          # never hand a block back while a copy into it may be in flight
                try:
                    self.wait()
                except Exception:
                    pass
            self.array = None
            _lib.load().picaso_host_free(self.ctx, ctypes.c_void_p(self.addr))
        self.addr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def broadcast_facets(src, nfacets, facet_scale=None, ctx=None):
    """``(rows, nwno)`` DeviceArray -> ``(rows, nwno, nfacets)`` with the facet index fastest, times
    ``facet_scale[f]`` when given (``picaso_broadcast_facets_dev``)."""
    ctx = ctx if ctx is not None else src.ctx
    rows, nwno = src.shape
    out = DeviceArray((rows, nwno, int(nfacets)), ctx)
    sc = _lib.f64(facet_scale, (int(nfacets),)) if facet_scale is not None else None
    _lib.check(_lib.load().picaso_broadcast_facets_dev(ctx, ctypes.c_size_t(rows), ctypes.c_int(nwno),
                                                      ctypes.c_int(int(nfacets)), ctypes.c_void_p(src.addr),
                                                      _lib.ptr(sc), ctypes.c_void_p(out.addr)), ctx)
    return out


def regrid_rows(xp, fp, d_x, ctx, scale=None):
    """``numpy.interp(x, xp, row)`` for every row of the host table ``fp`` ``(nrows, nin)``, on the device grid
    ``d_x`` (a DeviceArray of ``nwno`` wavenumbers): a ``(nrows, nwno)`` DeviceArray, bit for bit what the
    reference's ``wavelength.regrid`` returns (``picaso_regrid_rows_dev``).  ``scale``: result times a scalar."""
    if isinstance(fp, DeviceArray):          # tables already resident (kept by the caller between calls)
        d_fp, d_xp = fp, xp
        nrows, nin = fp.shape
    else:
        fp = np.ascontiguousarray(fp, dtype=np.float64)
        nrows, nin = fp.shape
        d_xp = DeviceArray.from_host(np.ascontiguousarray(xp, dtype=np.float64).reshape(nin), ctx)
        d_fp = DeviceArray.from_host(fp, ctx)
    nwno = int(d_x.shape[0])
    out = DeviceArray((nrows, nwno), ctx)
    sc = ctypes.byref(ctypes.c_double(float(scale))) if scale is not None else None
    _lib.check(_lib.load().picaso_regrid_rows_dev(ctx, ctypes.c_int(nrows), ctypes.c_int(nin), ctypes.c_long(nwno),
                                                  ctypes.c_void_p(d_xp.addr), ctypes.c_void_p(d_fp.addr),
                                                  ctypes.c_void_p(d_x.addr), sc, ctypes.c_void_p(out.addr)), ctx)
    out._inputs = (d_xp, d_fp)          # the launch is asynchronous: its inputs live as long as its output
    return out


def regrid_facets(xp, fp, d_x, ctx):
    """Per-facet tables ``fp`` ``(nlayer, nfacets, nin)`` (host) on the grid ``xp`` -> the ``(nlayer, nwno, nfacets)``
    DeviceArray (facet index fastest) of ``numpy.interp`` along the last axis on the device grid ``d_x``
    (``picaso_regrid_facets_dev``)."""
    if isinstance(fp, DeviceArray):
        d_fp, d_xp = fp, xp
        nlayer, nfac, nin = fp.shape
    else:
        fp = np.ascontiguousarray(fp, dtype=np.float64)
        nlayer, nfac, nin = fp.shape
        d_xp = DeviceArray.from_host(np.ascontiguousarray(xp, dtype=np.float64).reshape(nin), ctx)
        d_fp = DeviceArray.from_host(fp, ctx)
    nwno = int(d_x.shape[0])
    out = DeviceArray((nlayer, nwno, nfac), ctx)
    _lib.check(_lib.load().picaso_regrid_facets_dev(ctx, ctypes.c_int(nlayer), ctypes.c_int(nfac), ctypes.c_int(nin),
                                                    ctypes.c_long(nwno), ctypes.c_void_p(d_xp.addr),
                                                    ctypes.c_void_p(d_fp.addr), ctypes.c_void_p(d_x.addr),
                                                    ctypes.c_void_p(out.addr)), ctx)
    out._inputs = (d_xp, d_fp)
    return out


def sync(ctx=None):
    ctx = ctx if ctx is not None else _lib.context()
    _lib.check(_lib.load().picaso_sync(ctx), ctx)


def timer_start(ctx=None):
    ctx = ctx if ctx is not None else _lib.context()
    _lib.check(_lib.load().picaso_timer_start(ctx), ctx)


def timer_stop(ctx=None):
    ctx = ctx if ctx is not None else _lib.context()
    ms = ctypes.c_float(0)
    _lib.check(_lib.load().picaso_timer_stop(ctx, ctypes.byref(ms)), ctx)
    return ms.value
