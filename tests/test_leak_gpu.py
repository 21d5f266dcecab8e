"""A retrieval-shaped loop leaves nothing to Python's cycle collector: every device array of a finished spectrum() /
spectrum_batch() / 3-D call goes when the caller lets go of the result (tools/leak_check.py: fourteen kinds of calls, every
call with new inputs).  Before round 5's fix a finished ``spectrum.Spectrum`` sat in a reference cycle with its own
collectors, and hundreds of dead planes piled up between two collections (GBs at 1e5 wavelengths)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
def test_no_device_arrays_wait_for_the_cycle_collector(capsys):
    import gc
    import leak_check
    was = gc.isenabled()
    gc.collect()
    gc.disable()                 # nothing may be hidden by a collection that happens to run inside the loop
    try:
        rc = leak_check.main(["--calls", "91", "--nwno", "3000", "--cycles-only"])
    finally:
        if was:
            gc.enable()
    import json
    out = capsys.readouterr().out
    rec = json.loads(out.strip().splitlines()[-1])
    # on failure: which shapes waited, and who held them (referrer chains inside the unreachable set) -- not the samples
    assert rc == 0 and rec["finite"], json.dumps({k: rec[k] for k in (
        "device_arrays_that_waited_for_the_cycle_collector", "held_by", "finite", "calls_by_kind")})
    assert rec["device_arrays_that_waited_for_the_cycle_collector"] == {} and rec["held_by"] == {}
