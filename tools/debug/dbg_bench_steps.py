import sys, os, time, types, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import PatchLinkage
orig = PatchLinkage.count_pairs
times = []
def timed(self, *a, **k):
    t0 = time.perf_counter(); r = orig(self, *a, **k); times.append((time.perf_counter() - t0) * 1e3); return r
PatchLinkage.count_pairs = timed
sys.argv = ["bench.py", "--n-ref", "1e6", "--n-unk", "1e6", "--patches", "16", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0"]
t0 = time.perf_counter()
bench.main()
print("per-call ms:", " ".join("%.2f" % t for t in times))
