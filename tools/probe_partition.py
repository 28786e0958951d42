"""Balance of the job partition over N ranks, measured on ONE GPU: every rank's share of the headline job list is
run on its own; the step time of an N-GPU run is the slowest share (+ all-reduce)."""
import sys, types
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine, parallel
from yet_another_wizz_amd.measurements import angular_plans, threshold_table, job_costs

args = types.SimpleNamespace(n_ref=10e6, n_unk=10e6, patches=64, zbins=30)
config, ref, unk = bench.make_catalogs(args)
l1 = ref.build_trees(config.binning.edges, closed=config.binning.closed)
l2 = unk.build_trees(None)
links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
jobs = links.get_patch_pairs(ref, unk)
t = threshold_table(angular_plans(config))
costs = job_costs(l1, l2, jobs, t)
for _ in range(2):
    fine, st = engine.count_fine(l1, l2, jobs, t)
print(f"all {len(jobs)} jobs: device {st.kernel_ms:.3f} ms, evaluated {st.evaluated_pairs:.3e}")
# per-job evaluated pairs (what the device really does), to judge the cost model
ev = np.array([engine.count_fine(l1, l2, jobs[j:j + 1], t)[1].evaluated_pairs for j in range(len(jobs))], dtype=np.float64)
diag = jobs[:, 0] == jobs[:, 1]
print(f"evaluated pairs: diagonal jobs {ev[diag].sum():.3e} ({ev[diag].mean():.3e} each), off-diagonal {ev[~diag].sum():.3e} ({ev[~diag].mean():.3e} each)")
print(f"model cost:      diagonal jobs {costs[diag].sum():.3e}, off-diagonal {costs[~diag].sum():.3e}; corr(model, evaluated) = {np.corrcoef(costs, ev)[0,1]:.3f}")
for n in (2, 4, 8):
    work = engine.job_work(l1, l2, jobs, t).astype(np.float64)
    assert np.array_equal(work, ev)
    from yet_another_wizz_amd.measurements import JOB_FIXED_COST
    for name, c in (("model", costs), ("evaluated", ev), ("job_work + fixed", work + JOB_FIXED_COST)):
        parts = parallel.partition_jobs(c, n)
        ms = []
        for part in parts:
            for _ in range(2):
                _, s = engine.count_fine(l1, l2, jobs[part], t)
            ms.append(s.kernel_ms)
        ms = np.array(ms)
        print(f"N={n} partition by {name}: shares {np.round(ms, 3)} max {ms.max():.3f} mean {ms.mean():.3f} imbalance {ms.max()/ms.mean():.2f} sum {ms.sum():.3f}")
