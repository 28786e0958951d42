import time, types, cProfile, pstats, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
args = types.SimpleNamespace(n_ref=10e6, n_unk=10e6, patches=64, zbins=30, scales=1, weights=False, seed=1)
inputs = bench.make_inputs(args)
bench.make_catalogs(args, inputs)
for rep in range(2):
    pr = cProfile.Profile(); pr.enable()
    t0=time.perf_counter(); config, ref, unk = bench.make_catalogs(args, inputs); t1=time.perf_counter()
    ref.build_trees(config.binning.edges, closed=config.binning.closed); unk.build_trees(None); t2=time.perf_counter()
    pr.disable()
    print("make_catalogs", round(t1-t0,3), "build_trees", round(t2-t1,3))
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
