"""On-the-fly correlated-k gas mixing with the reference's call signature.

Drop-in for ``picaso.deq_chem.mix_all_gases_gasesfly`` (reference picaso/deq_chem.py:333-384): numpy
arrays in, the ``(nlayer, nwno, ngauss, 4)`` array of ln(mixed k) out.  The per-gas tables are large
(tens of MB each) and do not change between calls of a climate run, so their device copies are kept
and reused while the host arrays are the same objects with the same content fingerprint.
``picaso_amd.optics.RetrieveCKs(kappas=...)`` is the resident form ``picaso()`` uses.
"""
import numpy as np

from . import _lib, resident
from ._lib import f64
from .device import DeviceArray

_tables = {}            # (address, shape) -> (fingerprint, DeviceArray)


def _fingerprint(a):
    flat = a.reshape(-1)
    step = max(1, flat.size // 4096)
    return float(flat[::step].sum()), float(flat[0]), float(flat[-1])


def _resident_table(a, ctx):
    a = f64(a)
    key = (a.ctypes.data, a.shape)
    fp = _fingerprint(a)
    hit = _tables.get(key)
    if hit is not None and hit[0] == fp:
        return hit[1]
    d = DeviceArray.from_host(a, ctx)
    _tables[key] = (fp, d)
    return d


def clear_table_cache():
    """Drop the device copies of the k-tables kept by ``mix_all_gases_gasesfly``."""
    _tables.clear()


def mix_all_gases_gasesfly(kappas, mixes, gauss_pts, gauss_wts, indices, ctx=None):
    """``kappas``: list of ``(npres, ntemp, nwno, ngauss)`` ln(kappa) arrays, ``mixes``: list of per-layer
    mixing ratios, ``indices`` = [p_low, p_hi, t_low, t_hi] per layer (``get_mixing_indices``).
    Returns ``(nlayer, nwno, ngauss, 4)``: ln of the mixed coefficients at the four P-T neighbours."""
    ctx = ctx if ctx is not None else _lib.context()
    dk = [_resident_table(k, ctx) for k in kappas]
    out = resident.mix_all_gases_gasesfly(ctx, dk, mixes, gauss_pts, gauss_wts, indices).to_host()
    return np.ascontiguousarray(np.moveaxis(out, 1, 3))
