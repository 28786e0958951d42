"""Lean form of tools/probe_auto.py for the profiler: the three counts of BASELINE config #4 (DD, DR, RR of an
autocorrelation with ten times the randoms, weighted) on the AUTO path, REPS launches each, nothing else -- so that the
k-th group of REPS count-kernel dispatches in a rocprofv3 trace is DD, DR, RR in that order
(tools/profile_cmd.sh + tools/summarize_auto_profile.py).   python tools/probe_auto_prof.py [n_data] [n_random] [w]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import engine
from yet_another_wizz_amd.measurements import angular_plans, threshold_table

REPS = 4
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
nr = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 * n
weighted = len(sys.argv) > 3 and sys.argv[3].startswith("w")
for _opt in os.environ.get("YAW_SET", "").split(","):  # YAW_SET=key=value,key=value: any context option
    if "=" in _opt:
        engine.get_context().set_option(_opt.split("=")[0], int(_opt.split("=")[1]))
config, data, rand = bench.make_auto_catalogs(n, nr, weighted=weighted)
ld, lr = data.build_trees(config.binning.edges), rand.build_trees(config.binning.edges)
links = yaw.PatchLinkage.from_catalogs(config, data, rand)
t = threshold_table(angular_plans(config))
for name, l1, l2, auto in (("DD", ld, ld, True), ("DR", ld, lr, False), ("RR", lr, lr, True)):
    jobs = links.get_patch_pairs(data, None if auto else rand)
    for rep in range(REPS):
        fine, st = engine.count_fine(l1, l2, jobs, t)
    print(f"{name}: jobs={len(jobs)} cand={st.candidate_pairs:.4e} eval={st.evaluated_pairs:.4e} items={st.n_workgroups} "
          f"kernel_ms={st.kernel_ms:.3f} count_ms={st.count_ms:.3f} variant={st.band_variant} triples={st.merged_triples} "
          f"found={fine.sum():.8g}", flush=True)
