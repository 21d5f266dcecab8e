"""On-the-fly correlated-k gas mixing with the reference's call signature.

Drop-in for ``picaso.deq_chem.mix_all_gases_gasesfly`` (reference picaso/deq_chem.py:333-384): numpy
arrays in, the ``(nlayer, nwno, ngauss, 4)`` array of ln(mixed k) out.  The per-gas tables are large
(tens of MB each) and do not change between calls of a climate run, so their device copies are kept
and reused for host arrays with the same content (keyed by a digest of every byte, not by address).
``picaso_amd.optics.RetrieveCKs(kappas=...)`` is the resident form ``picaso()`` uses.
"""
import os

import numpy as np

from . import _lib, resident
from ._lib import f64
from .device import DeviceArray

_tables = {}            # (pid, context, digest of the table) -> DeviceArray: content-addressed, oldest out first
_TABLES_MAX = 64
_lib.on_context_destroy(lambda value: [_tables.pop(k) for k in [k for k in _tables if k[1] == value]])


def _fingerprint(a):
    """Digest of every byte of the table (``optics.content_digest``): an in-place edit of one coefficient is a new
    table, as it is for the reference, which reads its arguments afresh on every call (deq_chem.py:334-384)."""
    from .optics import content_digest
    return content_digest(a)


def _resident_table(a, ctx):
    a = f64(a)
    key = (os.getpid(), getattr(ctx, "value", ctx), _fingerprint(a))
    hit = _tables.get(key)
    if hit is not None:
        return hit
    while len(_tables) >= _TABLES_MAX:
        del _tables[next(iter(_tables))]
    d = DeviceArray.from_host(a, ctx)
    _tables[key] = d
    return d


def clear_table_cache():
    """Drop the device copies of the k-tables kept by ``mix_all_gases_gasesfly``."""
    _tables.clear()


@_lib.serialized
def mix_all_gases_gasesfly(kappas, mixes, gauss_pts, gauss_wts, indices, ctx=None):
    """``kappas``: list of ``(npres, ntemp, nwno, ngauss)`` ln(kappa) arrays, ``mixes``: list of per-layer
    mixing ratios, ``indices`` = [p_low, p_hi, t_low, t_hi] per layer (``get_mixing_indices``).
    Returns ``(nlayer, nwno, ngauss, 4)``: ln of the mixed coefficients at the four P-T neighbours."""
    ctx = ctx if ctx is not None else _lib.context()
    dk = [_resident_table(k, ctx) for k in kappas]
    out = resident.mix_all_gases_gasesfly(ctx, dk, mixes, gauss_pts, gauss_wts, indices).to_host()
    return np.ascontiguousarray(np.moveaxis(out, 1, 3))
