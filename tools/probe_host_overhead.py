"""Where the host time of one PatchLinkage.count_pairs call goes (10M x 10M headline, one GPU)."""
import sys, time, types
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine, measurements

args = types.SimpleNamespace(n_ref=10e6, n_unk=10e6, patches=64, zbins=30)
config, ref, unk = bench.make_catalogs(args)
ref.build_trees(config.binning.edges, closed=config.binning.closed)
unk.build_trees(None)
links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
for _ in range(3):
    links.count_pairs(ref, unk)
orig = engine.count_dense
acc = {"fine": 0.0, "lib_total": 0.0, "lib_kernel": 0.0}
def timed(*a, **k):
    t0 = time.perf_counter()
    out = orig(*a, **k)
    acc["fine"] += time.perf_counter() - t0
    acc["lib_total"] += out[1].total_ms * 1e-3
    acc["lib_kernel"] += out[1].kernel_ms * 1e-3
    return out
engine.count_dense = timed
measurements.engine.count_dense = timed
n = 50
t0 = time.perf_counter()
for _ in range(n):
    links.count_pairs(ref, unk)
tot = time.perf_counter() - t0
print(f"per call: count_pairs {tot/n*1e3:.3f} ms | engine.count_dense {acc['fine']/n*1e3:.3f} | library wall {acc['lib_total']/n*1e3:.3f} | "
      f"device {acc['lib_kernel']/n*1e3:.3f} | python around the library {(acc['fine']-acc['lib_total'])/n*1e3:.3f} | "
      f"python epilogue {(tot-acc['fine'])/n*1e3:.3f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    links.count_pairs(ref, unk)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
