import os, sys, time, cProfile, pstats
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, "tools", "e2e_3d_time.py")).read()
src = src[:src.index("out = {}\nfor tag, env in")]
src = src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", "%r" % ROOT)
sys.argv = ["x"]
exec(src)
calc = "reflected+thermal"
for devs in (None, [0] * 8):
    for _ in range(4):
        r = c3.spectrum(opa, calculation=calc, dimension="3d", devices=devs)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = c3.spectrum(opa, calculation=calc, dimension="3d", devices=devs); ts.append(time.perf_counter() - t0)
    print("devices", devs and len(devs), "ms", round(1e3 * min(ts), 3), float(r["albedo"].sum()))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    c3.spectrum(opa, calculation=calc, dimension="3d", devices=[0] * 8)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
