"""cProfile of inputs.spectrum() at 1e5 wavelengths (run on the GPU box)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BENCH_ONLY"] = "e2e"
import tools.bench_extra as be  # noqa: E402

pr = cProfile.Profile()
pr.enable()
be.main()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
