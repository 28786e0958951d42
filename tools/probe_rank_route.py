"""Host cost of the several-ranks route of PatchLinkage.count_pairs (rows left on the device, torch wrap, copy back, numpy
epilogue) against the one-process route (dense tensor from one C call), both on ONE GPU without a process group: what a
rank adds per call besides the collective itself.   python tools/probe_rank_route.py [calls]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()  # (before the library touches the device, as bench.py does)
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import measurements

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = types.SimpleNamespace(n_ref=1e7, n_unk=1e7, patches=64, zbins=30)
config, ref, unk = bench.make_catalogs(args)
ref.build_trees(config.binning.edges, closed=config.binning.closed)
unk.build_trees(None)
links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
for route in ("dense (one process)", "rows on the device + epilogue (a rank's route)", "dense (one process)"):
    measurements.FORCE_DEVICE_REDUCE = route.startswith("rows")
    for _ in range(20):
        res = links.count_pairs(ref, unk)
    t0 = time.perf_counter()
    for _ in range(n_calls):
        res = links.count_pairs(ref, unk)
    print(f"{route}: {(time.perf_counter() - t0) / n_calls * 1e3:.4f} ms per call, total {res[0].counts.counts.sum():.0f}")
if len(sys.argv) > 2:  # any second argument: profile of the rank's route
    import cProfile, pstats
    measurements.FORCE_DEVICE_REDUCE = True
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n_calls):
        links.count_pairs(ref, unk)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
