"""Run options of the accelerated path in ONE object.

Up to round 4 the alternative code paths of ``picaso()`` (full plane sets, the call-by-call sequence instead of the C
driver, facet-fastest 3-D planes, host-side regridding ...) were switched by environment variables read wherever a
branch happened to be.  They are A/B and test switches -- every one of them selects a path with the same results (same
bits unless its line says otherwise) -- and now live here: ``Options.from_env()`` is the only place of the host
package's spectrum path that reads the environment (once per call), ``picaso(..., options=Options(all_planes=True))``
/ ``inputs.spectrum(..., options=...)`` sets them explicitly.  The variables keep working, so shell-level A/B runs
(tools/ab.sh) and the tests that monkeypatch them are unchanged.

(Kernel-level switches that are read inside the C library -- PICASO_AMD_REFL_NO_COOP, PICASO_AMD_SH_NO_TOP,
PICASO_AMD_ANGLE_GROUP ... -- are launch-shape knobs of csrc/, not part of this object.)
"""
import contextlib
import contextvars
import os
from dataclasses import dataclass, fields, replace

_ENV = {
    # field                env variable                      meaning when set
    "all_planes":         "PICASO_AMD_ALL_PLANES",         # write and read all 13 opacity planes (no derived / lean sets)
    "no_driver":          "PICASO_AMD_NO_DRIVER",          # 1-D Toon spectra through the call-by-call path, not csrc/driver.hip
    "facet_fastest":      "PICASO_AMD_FACET_FASTEST",      # 3-D: (rows, nwno, nfacets) planes instead of facet-major ones
    "facet_loop":         "PICASO_AMD_FACET_LOOP",         # 3-D: one ATMSETUP + one gas launch per facet (the reference's loop)
    "host_regrid":        "PICASO_AMD_HOST_REGRID",        # cloud tables regridded with numpy on the host (1e-11, not bit for bit)
    "raman_planes":       "PICASO_AMD_RAMAN_PLANES",       # Raman factor as a host-made plane instead of the device form
    "unfused_opacity":    "PICASO_AMD_UNFUSED_OPACITY",    # gas stage and compute_opacity as two launches
    "regrid_planes":      "PICASO_AMD_REGRID_PLANES",      # cloud tables regridded into planes before the opacity launch
    "host_integrals":     "PICASO_AMD_HOST_INTEGRALS",     # Bond albedo / effective temperature integrals with numpy
    "sync_copies":        "PICASO_AMD_SYNC_COPIES",        # result copies synchronously after the launches
    "no_post_stream":     "PICASO_AMD_NO_POST_STREAM",     # spectrum_batch: integrals on the solver stream
    "py_setup":           "PICASO_AMD_PY_SETUP",           # ATMSETUP through the Python mirror, not picaso_host_setup
    "one_phase":          "PICASO_AMD_ONE_PHASE",          # the C driver in ONE call (opacity stage not enqueued ahead of the legs' set-up)
}


@dataclass(frozen=True)
class Options:
    all_planes: bool = False
    no_driver: bool = False
    facet_fastest: bool = False
    facet_loop: bool = False
    host_regrid: bool = False
    raman_planes: bool = False
    unfused_opacity: bool = False
    regrid_planes: bool = False
    host_integrals: bool = False
    sync_copies: bool = False
    no_post_stream: bool = False
    py_setup: bool = False
    one_phase: bool = False
    phases_in_flight: int = 16           # PICASO_AMD_PHASES_IN_FLIGHT: phase_curve() phases enqueued before the first is read
    phase_chunk: int = 0                 # PICASO_AMD_PHASE_CHUNK: phases per batched launch (0: from the HBM budget)
    overlap_legs: bool = True            # PICASO_AMD_OVERLAP_LEGS=0: thermal leg behind the reflected one, same stream

    @classmethod
    def from_env(cls, environ=None):
        if environ is None:
            # the process environment, read EVERY call (a test or a shell-level A/B may have changed it) but through the
            # mapping's own dictionary: eighteen `os.environ.get` calls encode and decode their way to ~15 us per spectrum,
            # eighteen dictionary look-ups to ~1; the values seen last time give back the object built from them
            raw = getattr(os.environ, "_data", None)
            if isinstance(raw, dict):
                seen = tuple(raw.get(k) for k in _RAW_KEYS)
                hit = _last_env[0]
                if hit is not None and hit[0] == seen:
                    return hit[1]
                opt = cls.from_env(os.environ.copy())
                _last_env[0] = (seen, opt)
                return opt
        env = os.environ if environ is None else environ
        kw = {f: bool(env.get(v)) for f, v in _ENV.items()}
        kw["overlap_legs"] = env.get("PICASO_AMD_OVERLAP_LEGS", "1") != "0"
        kw["phases_in_flight"] = int(env.get("PICASO_AMD_PHASES_IN_FLIGHT", "16"))
        kw["phase_chunk"] = int(env.get("PICASO_AMD_PHASE_CHUNK", "0"))
        return cls(**kw)

    def with_(self, **kw):
        return replace(self, **kw)


_NAMES = tuple(_ENV.values()) + ("PICASO_AMD_OVERLAP_LEGS", "PICASO_AMD_PHASES_IN_FLIGHT", "PICASO_AMD_PHASE_CHUNK")
_RAW_KEYS = tuple(os.fsencode(n) for n in _NAMES) if os.name == "posix" else ()
_last_env = [None]

_active = contextvars.ContextVar("picaso_amd_options", default=None)


def current(options=None):
    """``options`` when given, else the set a caller up the stack put in force (``use``), else the environment's."""
    if options is not None:
        return options
    hit = _active.get()
    return hit if hit is not None else Options.from_env()


@contextlib.contextmanager
def use(options):
    """``with use(opt):`` -- the helpers called inside (optics.compute_opacity_resident ...) see ``opt`` through
    ``current()`` without it being threaded through every signature."""
    token = _active.set(options)
    try:
        yield options
    finally:
        _active.reset(token)


def names():
    return [f.name for f in fields(Options)]
