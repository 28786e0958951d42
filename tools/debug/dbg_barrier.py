import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch
torch.cuda.is_available(); torch.cuda.set_device(0)
from yet_another_wizz_amd import PatchLinkage, engine
args = types.SimpleNamespace(n_ref=1e6, n_unk=1e6, patches=16, zbins=30)
inputs = bench.make_inputs(args)
config, ref, unk = bench.make_catalogs(args, inputs)
ref.build_trees(config.binning.edges, closed=config.binning.closed); unk.build_trees(None)
links = PatchLinkage.from_catalogs(config, ref, unk)
for _ in range(5): links.count_pairs(ref, unk)
def tm(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3
print("sync before: %.3f ms" % tm(torch.cuda.synchronize))
t0 = time.perf_counter()
per = []
for _ in range(20):
    per.append(tm(lambda: links.count_pairs(ref, unk)))
    if per[-1] > 5: print("  stalled call: python %.3f ms, library wall %.3f, device %.3f" % (per[-1], links.last_stats.total_ms, links.last_stats.kernel_ms))
loop = (time.perf_counter() - t0) * 1e3
print("loop %.3f ms (sum of calls %.3f, max %.3f)" % (loop, sum(per), max(per)))
print("sync after: %.3f ms" % tm(torch.cuda.synchronize))
print("sync again: %.3f ms" % tm(torch.cuda.synchronize))
for _ in range(3):
    per = [tm(lambda: links.count_pairs(ref, unk)) for _ in range(20)]
    print("loop again sum %.3f; sync after: %.3f ms" % (sum(per), tm(torch.cuda.synchronize)))
