"""Python time of one PatchLinkage.count_pairs call around the library (1M x 1M, 16 patches: the library part is short, so
the interpreter's share stands out).   python tools/probe_python_overhead.py [calls]"""
import cProfile, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import yet_another_wizz_amd as yaw

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
args = types.SimpleNamespace(n_ref=1e6, n_unk=1e6, patches=16, zbins=30)
config, ref, unk = bench.make_catalogs(args)
ref.build_trees(config.binning.edges, closed=config.binning.closed)
unk.build_trees(None)
links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
for _ in range(20):
    links.count_pairs(ref, unk)
lib_ms = 0.0
t0 = time.perf_counter()
for _ in range(n_calls):
    links.count_pairs(ref, unk)
    lib_ms += links.last_stats.total_ms
wall = (time.perf_counter() - t0) / n_calls * 1e3
print(f"per call: {wall:.4f} ms, of which enqueue..finish inside the library {lib_ms / n_calls:.4f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(n_calls):
    links.count_pairs(ref, unk)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
