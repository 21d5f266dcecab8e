import os, sys, time, cProfile, pstats
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["x"]
src = open(os.path.join(ROOT, "tools", "e2e_3d_time.py")).read()
src = src[:src.index("out = {}\nfor tag, env in")]
src = src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", "%r" % ROOT)
exec(src)
P = 8
phases = list(np.linspace(0.0, 2 * np.pi * (P - 1) / P, P))
pt = jdi.inputs()
pt.phase_curve_geometry("thermal", phases, num_gangle=ng, num_tangle=nt)
pt.gravity(gravity=2500.0)
pt.atmosphere_4d([prof3 for _ in phases])
pt.approx(raman="none")
pt.phase_curve(opa)
t0 = time.perf_counter(); pt.phase_curve(opa); print("thermal curve ms", 1e3 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); pt.phase_curve(opa); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
