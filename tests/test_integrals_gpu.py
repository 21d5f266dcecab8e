"""picaso_trapz_dev against numpy: the spectrum-wide integrals of the output dictionary (reference justdoit.py:552-599:
``np.trapz(x=1/wno, y=albedo*stellar)``, ``np.trapz(x=1/wno[::-1], y=thermal[::-1])``) computed on the device must carry
numpy's own bits -- a spectrum integrated on the host (wavelength blocks on several GPUs, full_output paths) and one
integrated on the device are the same spectrum."""
import numpy as np
import pytest

from picaso_amd import _lib, device, resident

pytestmark = pytest.mark.gpu

LENGTHS = [2, 3, 8, 9, 10, 16, 17, 63, 128, 129, 130, 136, 137, 257, 258, 1000, 1025, 4096, 9999, 10000, 65537, 99999,
           100000, 100007, 262145]


def _device_trapz(ctx, x, y, mult=None, reverse=False):
    d = np.diff(x[::-1]) if reverse else np.diff(x)
    out = device.DeviceArray((1,), ctx)
    dd, dy = device.DeviceArray.from_host(d, ctx), device.DeviceArray.from_host(y, ctx)
    dm = device.DeviceArray.from_host(mult, ctx) if mult is not None else None
    resident.trapz(ctx, len(y), dd, dy, out, mult=dm, reverse=reverse)
    return out.to_host()[0]


@pytest.mark.parametrize("n", LENGTHS)
def test_trapz_bits_of_numpy(n):
    ctx = _lib.context(0)
    rng = np.random.default_rng(n)
    wno = np.sort(rng.uniform(300.0, 33000.0, n))
    x = 1 / wno
    y = 10.0 ** rng.uniform(-8, 3, n)
    s = 1.0 + 0.3 * np.cos(wno / 700.0)
    assert _device_trapz(ctx, x, y) == np.trapezoid(x=x, y=y)
    assert _device_trapz(ctx, x, y, mult=s) == np.trapezoid(x=x, y=y * s)
    assert _device_trapz(ctx, x, y, reverse=True) == np.trapezoid(x=x[::-1], y=y[::-1])
    assert _device_trapz(ctx, x, y, mult=s, reverse=True) == np.trapezoid(x=x[::-1], y=(y * s)[::-1])


def test_trapz_every_length_to_300():
    """every remainder of the blocks of eight and both sides of the 128-term split"""
    ctx = _lib.context(0)
    rng = np.random.default_rng(7)
    for n in range(2, 301):
        x = np.cumsum(rng.uniform(0.1, 1.0, n))
        y = rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3, n)       # mixed signs: cancellation shows any reordering
        assert _device_trapz(ctx, x, y) == np.trapezoid(x=x, y=y), n


def test_trapz_repeated_and_interleaved_lengths():
    """plans are cached per length and share nothing: alternate two lengths on one stream"""
    ctx = _lib.context(0)
    rng = np.random.default_rng(11)
    for n in (1000, 5000, 1000, 5000, 1000):
        x = np.cumsum(rng.uniform(0.1, 1.0, n))
        y = rng.normal(size=n)
        assert _device_trapz(ctx, x, y) == np.trapezoid(x=x, y=y)


def test_trapz_argument_errors():
    ctx = _lib.context(0)
    one = device.DeviceArray((1,), ctx)
    with pytest.raises(_lib.PicasoHipError):
        resident.trapz(ctx, 1, one, one, one)
    with pytest.raises(_lib.PicasoHipError):
        resident.trapz(ctx, 10, None, one, one)
