"""Import the *reference* PICASO hot-path modules in THIS container only.

Test infrastructure.  The reference (``/root/reference``) is pure Python + numba; numba,
astropy, bokeh, h5py are absent here, so the modules are imported under identity-decorator
shims.  Used exclusively by ``tests/golden/make_golden.py`` to emit
golden input/output vectors and by ad-hoc checks of the ``oracle/`` restatement.  Nothing here
runs on the GPU box (``/root/reference`` does not exist there) and nothing from the reference is
copied into this repository.
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("PICASO_REFERENCE_ROOT", "/root/reference")


class _Dummy(types.ModuleType):
    """Module whose every attribute is a harmless no-op callable / sub-dummy."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _noop

    def __getitem__(self, key):          # numba type specs such as float64[:]
        return self


def _noop(*a, **k):
    return None


def _jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def install_shims():
    if "numba" not in sys.modules or not hasattr(sys.modules["numba"], "__graft_shim__"):
        nb = types.ModuleType("numba")
        nb.jit = nb.njit = nb.vectorize = nb.guvectorize = _jit
        nb.prange = range
        nb.objmode = _noop
        nb.float32 = nb.float64 = nb.int32 = nb.int64 = _Dummy("numba.types")
        exp = types.ModuleType("numba.experimental")
        exp.jitclass = _jit
        nb.experimental = exp
        sys.modules["numba.experimental"] = exp
        nb.__graft_shim__ = True
        sys.modules["numba"] = nb
    for name in ("bokeh", "bokeh.plotting", "bokeh.palettes", "bokeh.io", "bokeh.models",
                 "astropy", "astropy.io", "astropy.io.fits", "astropy.units", "astropy.constants", "h5py",
                 "virga", "virga.justdoit"):
        if name not in sys.modules:
            sys.modules[name] = _Dummy(name)
    os.environ.setdefault("picaso_refdata", os.path.join(REF_ROOT, "reference"))


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "picaso"))


def _load_by_path(modname, filename):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, "picaso", filename))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load(name):
    """name in {'fluxes','disco','rayleigh','deq_chem','optics','atmsetup','climate'} -> reference module object."""
    if name in _cache:
        return _cache[name]
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    install_shims()
    if name in ("fluxes", "disco", "rayleigh", "deq_chem"):
        mod = _load_by_path("_picaso_ref_" + name, name + ".py")
    elif name in ("optics", "atmsetup", "climate"):
        if "picaso" not in sys.modules:
            pkg = types.ModuleType("picaso")
            pkg.__path__ = [os.path.join(REF_ROOT, "picaso")]
            sys.modules["picaso"] = pkg
        mod = importlib.import_module("picaso." + name)
    else:
        raise KeyError(name)
    _cache[name] = mod
    return mod
