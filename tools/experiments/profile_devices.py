import os, sys
os.environ["NWNO"]="100000"
sys.argv=["x"]
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src=open(os.path.join(ROOT,"tools/e2e_1d_time.py")).read()
src=src[:src.index("calc = os.environ.get")]
exec(src)
import cProfile, pstats
devs=[0]*8
for _ in range(20): case.spectrum(opa, calculation="reflected+thermal", devices=devs)
pr=cProfile.Profile(); pr.enable()
for _ in range(30): case.spectrum(opa, calculation="reflected+thermal", devices=devs)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
