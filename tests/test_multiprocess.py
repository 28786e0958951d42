"""N>1 path on CPU: two processes in a gloo group shard the patch-pair jobs, each counts its share
(the oracle stands in for the GPU call) and the dense tensor is combined with one all-reduce.
The result must equal the single-process result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import helpers
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine, parallel

    seen = []

    def counting(layout1, layout2, jobs, thresholds, *, kernel=None, sort_axis=2):
        seen.append(len(jobs))
        return helpers.oracle_count_fine(layout1, layout2, jobs, thresholds)

    engine.count_fine = counting
    # the device's cost estimate has a host model as its CPU stand-in
    from yet_another_wizz_amd import measurements
    engine.job_work = lambda l1, l2, jobs, t, **kw: measurements.job_costs(l1, l2, jobs, t)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.world() == (rank, world)
        inp, cats = helpers.full_catalogs("w")
        config = helpers.full_config(inp, "s2", "right")
        cfs = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
        acf = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), jobs_seen=np.array(seen),
                 dd=cfs[0].dd.counts.counts, rr=cfs[1].rr.counts.counts, add=acf[0].dd.counts.counts,
                 w=cfs[0].sample().data)
    finally:
        dist.destroy_process_group()


def test_two_process_sharding_matches_single_process(tmp_path, monkeypatch):
    import torch.multiprocessing as mp

    import helpers
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    monkeypatch.setattr(engine, "count_fine", helpers.oracle_count_fine)
    inp, cats = helpers.full_catalogs("w")
    config = helpers.full_config(inp, "s2", "right")
    cfs = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
    acf = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"])
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    for r in (r0, r1):  # every rank holds the full, identical result
        assert np.array_equal(r["dd"], cfs[0].dd.counts.counts)
        assert np.array_equal(r["rr"], cfs[1].rr.counts.counts)
        assert np.array_equal(r["add"], acf[0].dd.counts.counts)
        assert np.array_equal(r["w"], cfs[0].sample().data)
    # the jobs were really split: each rank counted a strict subset, together all of them
    n_cross = len(helpers.load_golden("full_w_s2_right.npz")["cross.job_pairs"])
    assert r0["jobs_seen"][0] + r1["jobs_seen"][0] == n_cross
    assert 0 < r0["jobs_seen"][0] < n_cross


def test_partition_is_balanced_and_complete():
    from yet_another_wizz_amd.parallel import partition_jobs

    rng = np.random.default_rng(0)
    cost = rng.uniform(1, 100, 500)
    parts = partition_jobs(cost, 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(500))
    loads = np.array([cost[p].sum() for p in parts])
    assert loads.max() / loads.mean() < 1.05
    assert partition_jobs([], 4)[0].size == 0
