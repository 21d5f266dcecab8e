"""The C ABI: libpicaso_hip.so builds for gfx950 without a GPU, loads, and exports every function
include/picaso_hip.h declares; the Python layer refuses to run without the library or without a
GPU (no CPU fallback)."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from picaso_amd import _lib
    from picaso_amd import build as b
    b.build(force=False)              # hipcc cross-compiles; a no-op when the library is current
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from picaso_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    for must in ("picaso_get_reflected_1d", "picaso_get_reflected_3d", "picaso_get_thermal_1d",
                 "picaso_get_thermal_3d", "picaso_get_reflected_SH", "picaso_get_thermal_SH",
                 "picaso_compress_disco", "picaso_compress_thermal", "picaso_compute_opacity_dev",
                 "picaso_opacity_gas_dev", "picaso_get_transit_1d", "picaso_get_reflected_1d_ck_dev"):
        assert must in names


def test_header_is_plain_c():
    """The header compiles as C (extern "C" boundary, plain pointers and sizes only)."""
    src = '#include "picaso_hip.h"\nint main(void) { return picaso_version() == 0; }\n'
    p = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                       input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()


def test_version_and_error_calls_need_no_gpu(lib):
    lib.picaso_version.restype = ctypes.c_char_p
    assert b"picaso_amd" in lib.picaso_version()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is visible: covered by the -m gpu tests")
def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a GPU every compute entry point refuses to run."""
    import numpy as np
    from picaso_amd import _lib, fluxes
    with pytest.raises(_lib.PicasoHipError):
        _lib.context()
    z = np.ones((1, 4))
    with pytest.raises(_lib.PicasoHipError):
        fluxes.get_thermal_1d(2, np.ones(4), 4, 1, 1, np.ones(2), z, z, z, np.ones(2), [[0.5]], 0.0, 0, np.ones(4), 0)


def test_comm_symbols_and_rccl_linked(lib):
    """The multi-GPU layer lives in the library: comm entry points exported, librccl a direct dependency."""
    for n in ("picaso_comm_unique_id", "picaso_comm_init_rank", "picaso_comm_init_all", "picaso_all_gather_dev",
              "picaso_all_gatherv_dev", "picaso_all_gather_async_dev", "picaso_all_gather_multi_async_dev", "picaso_comm_wait_slot", "picaso_comm_max", "picaso_comm_barrier", "picaso_comm_destroy"):
        assert hasattr(lib, n), n
    from picaso_amd import _lib
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True).stdout.decode()
    assert "librccl" in out


def _no_gpu_here():
    from picaso_amd import _lib
    return _lib.device_count() == 0


def test_product_path_fails_loudly_without_a_gpu():
    """No CPU fallback: on a machine without a HIP device every product entry point raises."""
    import numpy as np
    import pytest
    if not _no_gpu_here():
        pytest.skip("a GPU is visible here")
    from picaso_amd import climate, fluxes, synthetic as syn
    from picaso_amd._lib import PicasoHipError
    sc = syn.make_scene(5, 8, seed=1)
    with pytest.raises(PicasoHipError, match="no HIP device|GPU"):
        fluxes.get_thermal_1d(6, sc["wno"], 8, 5, 1, sc["tlevel"], sc["dtau_og"], sc["w0_no_raman"], sc["cosb_og"],
                              sc["plevel"], np.full((5, 1), 0.5), 0.0, 0, sc["wno"] * 0, 0)
    with pytest.raises(PicasoHipError):
        fluxes.get_transit_1d(np.linspace(2, 1, 6), np.ones(6), 6, 8, 1.0, np.ones(5), 1.0, 1.0, np.ones(6), np.ones(6),
                              np.ones(5), sc["dtau_og"])
    atm = climate.Atmosphere_Tuple(None, None, 6, sc["tlevel"], sc["plevel"], None, None, None, None)
    with pytest.raises(PicasoHipError):
        climate.get_fluxes(atm, None, None, climate.ScatteringPhase_Tuple(0.0, 3, 0, 1, -1, 2, -.5, 1),
                           climate.Disco_Tuple(5, 1, np.ones(5), np.ones(1), None, None, 1.0),
                           climate.Opagrid_Tuple(8, np.ones(8), sc["wno"], 1, np.ones(1)), 1.0, True, True)


def test_missing_library_is_a_clear_error(monkeypatch):
    """Without libpicaso_hip.so the loader says how to build it and that nothing else will run."""
    import pytest
    from picaso_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpicaso_hip.so")
    with pytest.raises(_lib.PicasoHipError, match="no CPU fallback"):
        _lib.load()


def test_committed_pmc_numbers_belong_to_the_committed_kernel_sources():
    """profiles/traffic.json (HBM bytes and VALU instructions per launch, quoted by bench.py as
    roofline.traffic / fp64_issue) carries the hash of the kernel sources it was measured on; bench.py
    drops the numbers when the hash differs.  Committed state: they match."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    prof = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert prof["kernel_source_hash"] == bench.kernel_source_hash()
    assert 0.95 < prof["hbm_bytes_per_launch"] / 8.0e8 < 1.2


def test_driver_structs_match_the_library_layout():
    """picaso_amd/driver.py restates picaso_block / picaso_spectrum_job (include/picaso_hip.h) as ctypes Structures:
    same size and member offsets as the compiled library (no GPU needed)."""
    import ctypes
    from picaso_amd import _lib, driver
    lib = _lib.load()
    vals = [ctypes.c_size_t(0) for _ in range(4)]
    assert lib.picaso_driver_abi(*[ctypes.byref(v) for v in vals]) == 0
    assert vals[0].value == ctypes.sizeof(driver.Block)
    assert vals[1].value == ctypes.sizeof(driver.Job)
    assert vals[2].value == driver.Block.albedo_host.offset
    assert vals[3].value == driver.Job.hard_surface.offset
