import sys, os, time, types, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine, _lib

args = types.SimpleNamespace(n_ref=10e6, n_unk=10e6, patches=64, zbins=30)
config, ref, unk = bench.make_catalogs(args)
l1 = ref.build_trees(config.binning.edges, closed=config.binning.closed)
l2 = unk.build_trees(None)
links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
for _ in range(5):
    links.count_pairs(ref, unk)
jobs = links.get_patch_pairs(ref, unk)
plans, t = links._angular_setup()
slices, factors = links._dense_spec
ctx, d1, d2 = engine._device_pair(l1, l2, t, links.sort_axis)
n = 200
def avg(fn):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print("np.empty dense      %.4f ms" % avg(lambda: np.empty((1, 30, 64, 64))))
keep = []
print("np.empty kept       %.4f ms" % avg(lambda: keep.append(np.empty((1, 30, 64, 64)))))
def fill():
    a = np.empty((1, 30, 64, 64)); a.fill(0.0)
print("np.empty + fill     %.4f ms" % avg(fill))
tot = [0.0]
def fine_call():
    c, s, st = _lib.count_pairs(ctx, d1, d2, jobs, t); tot[0] += st.total_ms
print("count_pairs (fine)  %.4f ms" % avg(fine_call), " library wall %.4f" % (tot[0] / n))
tot[0] = 0.0
def dense_call():
    d, st = _lib.count_pairs_dense(ctx, d1, d2, jobs, t, slices, factors, False); tot[0] += st.total_ms
print("count_pairs_dense   %.4f ms" % avg(dense_call), " library wall %.4f" % (tot[0] / n))
print("links.count_pairs   %.4f ms" % avg(lambda: links.count_pairs(ref, unk)))
