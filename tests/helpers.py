"""Shared helpers for the parity tests: golden-fixture access and error metrics."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLANES = ("dtau", "tau", "w0", "cosb", "gcos2", "ftau_cld", "ftau_ray", "dtau_og", "tau_og",
          "w0_og", "cosb_og")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def scene_id(path):
    return os.path.basename(path)[:-4]


class Golden:
    def __init__(self, path):
        self.z = np.load(path)
        self.keys = list(self.z.keys())

    def __getitem__(self, k):
        return self.z[k]

    def inp(self, k):
        return self.z["in/" + k]

    def geo(self, k):
        v = self.z["geo/" + k]
        return v if v.ndim else v.item()

    def opt(self, k):
        return float(self.z["opt/" + k])

    def cases(self, family):
        """Unique case keys 'family/<case>' present in the file."""
        out = []
        for k in self.keys:
            parts = k.split("/")
            if parts[0] == family and parts[1] not in out:
                out.append(parts[1])
        return out

    def tthg(self):
        return tuple(self.opt(k) for k in ("frac_a", "frac_b", "frac_c", "constant_back",
                                           "constant_forward"))


def rel_err(got, ref, floor=0.0):
    """max |got-ref| / max(|ref|, floor); floor lets cancellation-dominated entries be judged
    against the scale of the field instead of their own tiny magnitude."""
    got, ref = np.asarray(got), np.asarray(ref)
    den = np.maximum(np.abs(ref), floor)
    den = np.where(den == 0, 1.0, den)
    return float(np.max(np.abs(got - ref) / den))


def scale_err(got, ref):
    """max |got-ref| relative to the per-wavelength scale (max over leading axes)."""
    got, ref = np.asarray(got), np.asarray(ref)
    lead = tuple(range(ref.ndim - 1))
    scale = np.max(np.abs(ref), axis=lead, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float(np.max(np.abs(got - ref) / scale))


def lvl_err(got4, ref4):
    """Level-flux metric: max |got-ref| over the four (numg,numt,nlevel,nwno) arrays, relative to
    the per-wavelength scale of the whole flux field (max over the four arrays, angles, levels).
    The reference's downward-flux expressions cancel catastrophically in optically thin layers
    (sigma1*(1-e) + sigma2*(mu*e + dtau - mu)), so tiny entries carry no relative precision even
    between two runs of the reference with different libm."""
    ref4 = [np.asarray(r) for r in ref4]
    scale = np.max(np.stack([np.max(np.abs(r), axis=(0, 1, 2)) for r in ref4]), axis=0)
    scale = np.where(scale == 0, 1.0, scale)
    return max(float(np.max(np.abs(np.asarray(g) - r) / scale)) for g, r in zip(got4, ref4))


def lvl_excess(got4, ref4, x80_4, tol):
    """Element-wise level-flux check against the reference's fp64 output AND its extended-precision
    evaluation.  With e_ref = |ref - x80| (how far the reference's own fp64 rounding moved that
    element) the kernel may differ from the reference by tol*scale + 2 e_ref and from the
    extended-precision value by tol*scale + e_ref/500 (x87 extended carries 11 more mantissa bits
    than fp64, so x80 itself is only good to ~e_ref/2048).  Returns the worst excess over the
    allowance in units of the per-wavelength field scale (<= 0 means pass)."""
    ref4 = [np.asarray(r) for r in ref4]
    scale = np.max(np.stack([np.max(np.abs(r), axis=(0, 1, 2)) for r in ref4]), axis=0)
    scale = np.where(scale == 0, 1.0, scale)
    worst = -np.inf
    for g, r, x in zip(got4, ref4, x80_4):
        e_ref = np.abs(r - x)
        ex1 = (np.abs(np.asarray(g) - r) - 2.0 * e_ref) / scale - tol
        ex2 = (np.abs(np.asarray(g) - x) - e_ref / 500.0) / scale - tol
        worst = max(worst, float(np.max(ex1)), float(np.max(ex2)))
    return worst


def excess(got, ref, x80, tol, floor):
    """``lvl_excess`` for one array judged relative to ``max(|ref|, floor)``: with e_ref = |ref - x80| (the reference's
    own fp64 rounding of that element, from its extended-precision evaluation) the result may differ from the
    reference by tol*scale + 2 e_ref and from the extended-precision value by tol*scale + e_ref/500.  <= 0 passes."""
    got, ref, x80 = (np.asarray(a, dtype=np.float64) for a in (got, ref, x80))
    scale = np.maximum(np.abs(ref), floor)
    e_ref = np.abs(ref - x80)
    ex1 = (np.abs(got - ref) - 2.0 * e_ref) / scale - tol
    ex2 = (np.abs(got - x80) - e_ref / 500.0) / scale - tol
    return max(float(np.max(ex1)), float(np.max(ex2)))


# ------------------------------------------------------------------------------------------------
# a stand-in for the h5py package (absent from this image and from the GPU box): the product's HDF5 readers
# (picaso_amd/optics.py read_ck_tables; reference optics.py:725-770, opacity_factory.py:2221-2327) use
# ``h5py.File(path, mode)`` as a context manager, ``f[name][:]`` to read a dataset and ``f[name] = array`` to write
# one.  This module offers exactly that on top of an .npz container kept under the .hdf5 name, so that the reader
# branch EXECUTES in the tests instead of being skipped.  With a real h5py installed the tests use it instead.
# ------------------------------------------------------------------------------------------------
class _StubDataset:
    def __init__(self, a):
        self._a = np.asarray(a)
        self.shape, self.dtype, self.attrs = self._a.shape, self._a.dtype, {}

    def __getitem__(self, key):
        return self._a[key].copy() if self._a.ndim else self._a[()]

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __len__(self):
        return len(self._a)


class _StubFile:
    def __init__(self, path, mode="r"):
        self._path, self._mode, self._data, self.attrs = str(path), mode, {}, {}
        if mode == "r":
            with np.load(self._path, allow_pickle=False) as z:
                self._data = {k: z[k] for k in z.files}
        elif mode not in ("w", "a"):
            raise ValueError("h5py stand-in: mode %r" % mode)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._mode in ("w", "a") and self._data is not None:
            with open(self._path, "wb") as fh:
                np.savez(fh, **self._data)
        self._data = None

    def __getitem__(self, name):
        return _StubDataset(self._data[name])

    def __setitem__(self, name, value):
        if self._mode == "r":
            raise OSError("h5py stand-in: file is read-only")
        self._data[name] = np.asarray(value)

    def create_dataset(self, name, data=None, **kw):
        self[name] = data
        return self[name]

    def __contains__(self, name):
        return name in self._data

    def keys(self):
        return self._data.keys()


def h5py_module():
    """The real h5py when it is installed, else a module object with the ``File`` stand-in above."""
    try:
        import h5py
        return h5py, False
    except ImportError:
        import types
        mod = types.ModuleType("h5py")
        mod.File = _StubFile
        mod.__stand_in__ = True
        return mod, True
