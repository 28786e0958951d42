import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine, _lib

args = types.SimpleNamespace(n_ref=1e6, n_unk=1e6, patches=16, zbins=30)
config, ref, unk = bench.make_catalogs(args)
l1 = ref.build_trees(config.binning.edges, closed=config.binning.closed)
l2 = unk.build_trees(None)
links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
jobs = links.get_patch_pairs(ref, unk)
plans, t = links._angular_setup()
out = []
for i in range(120):
    t0 = time.perf_counter()
    if mode == "dense":
        links.count_pairs(ref, unk); st = links.last_stats
    else:
        fine, st = engine.count_fine(l1, l2, jobs, t)
    dt = (time.perf_counter() - t0) * 1e3
    out.append((dt, st.total_ms, st.kernel_ms))
for i in list(range(0, 12)) + list(range(20, 120, 10)):
    print(mode, i, "call %.3f ms  lib %.3f  dev %.3f" % out[i])
