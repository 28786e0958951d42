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


def _worker(rank, world, port, out_dir, fail_rank=-1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import helpers
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine, parallel

    seen = []

    def counting(layout1, layout2, jobs, thresholds, *, kernel=None, sort_axis=2, max_workers=None):
        seen.append(len(jobs))
        if rank == fail_rank:
            raise engine._lib.YawhipError("simulated device failure")
        return helpers.oracle_count_fine(layout1, layout2, jobs, thresholds)

    engine.count_fine = counting
    # the device's cost estimate has a host model as its CPU stand-in
    from yet_another_wizz_amd import measurements
    calls = []

    def job_work(l1, l2, jobs, t, **kw):
        calls.append(rank)
        return measurements.job_costs(l1, l2, jobs, t)

    engine.job_work = job_work
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.world() == (rank, world)
        inp, cats = helpers.full_catalogs("w")
        config = helpers.full_config(inp, "s2", "right")
        if fail_rank >= 0:  # a failing rank must make every rank raise, not leave the others in the collective
            try:
                yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
            except RuntimeError as err:
                with open(os.path.join(out_dir, f"rank{rank}.err"), "w") as f:
                    f.write(str(err))
            return
        cfs = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
        acf = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"])
        assert (len(calls) > 0) == (rank == 0)  # the partition is derived on rank 0 only and broadcast
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), jobs_seen=np.array(seen),
                 dd=cfs[0].dd.counts.counts, rr=cfs[1].rr.counts.counts, add=acf[0].dd.counts.counts,
                 w=cfs[0].sample().data)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_process_group_sharding_matches_single_process(tmp_path, monkeypatch, world):
    import torch.multiprocessing as mp

    import helpers
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    helpers.use_oracle_engine(monkeypatch)
    inp, cats = helpers.full_catalogs("w")
    config = helpers.full_config(inp, "s2", "right")
    cfs = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
    acf = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"])
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0 = ranks[0]
    for r in ranks:  # every rank holds the full, identical result
        assert np.array_equal(r["dd"], cfs[0].dd.counts.counts)
        assert np.array_equal(r["rr"], cfs[1].rr.counts.counts)
        assert np.array_equal(r["add"], acf[0].dd.counts.counts)
        assert np.array_equal(r["w"], cfs[0].sample().data)
    # the jobs were really split: each rank counted a strict subset, together all of them
    n_cross = len(helpers.load_golden("full_w_s2_right.npz")["cross.job_pairs"])
    assert sum(int(r["jobs_seen"][0]) for r in ranks) == n_cross
    assert 0 < r0["jobs_seen"][0] < n_cross


def test_failing_rank_raises_everywhere(tmp_path):
    """One rank's device call fails: the flag that travels with the all-reduce makes every rank raise."""
    import torch.multiprocessing as mp

    world = 3
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), 1), nprocs=world, join=True, start_method="spawn")
    msgs = [open(tmp_path / f"rank{r}.err").read() for r in range(world)]
    assert all("failed on 1 of 3 ranks" in m for m in msgs)


def test_partition_is_balanced_and_complete():
    from yet_another_wizz_amd.parallel import partition_jobs

    rng = np.random.default_rng(0)
    cost = rng.uniform(1, 100, 500)
    parts = partition_jobs(cost, 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(500))
    loads = np.array([cost[p].sum() for p in parts])
    assert loads.max() / loads.mean() < 1.05
    assert partition_jobs([], 4)[0].size == 0


@pytest.mark.parametrize("env, expect", [
    (dict(YAW_AMD_DEVICES="3"), 3),                    # what mpirun / srun launchers set per process
    (dict(YAW_AMD_DEVICE="2", LOCAL_RANK="5"), 2),     # the explicit override wins
    (dict(LOCAL_RANK="4"), 4),                         # torch.distributed.run
    (dict(YAW_AMD_DEVICES="5", LOCAL_RANK="1"), 5),
])
def test_counting_context_and_collectives_agree_on_the_device(monkeypatch, env, expect):
    """One source of truth for a rank's GPU: the device the counting context takes (engine.default_devices) is the one the
    collectives stage on (parallel.local_device_index), whichever variable a launcher sets."""
    from yet_another_wizz_amd import engine, parallel

    for key in ("YAW_AMD_DEVICES", "YAW_AMD_DEVICE", "LOCAL_RANK"):
        monkeypatch.delenv(key, raising=False)
    for key, value in env.items():
        monkeypatch.setenv(key, value)
    assert parallel.local_device_index() == expect
    assert engine.default_devices() == (expect,)


def test_several_ids_in_a_group_fall_back_to_the_host_route(monkeypatch):
    """YAW_AMD_DEVICES naming several ids: a context of several devices -- inside a process group such a rank must not take
    the device-resident route (single-device contexts only); ``default_devices`` reports what ``count_pairs`` checks."""
    from yet_another_wizz_amd import engine

    monkeypatch.setenv("YAW_AMD_DEVICES", "0,1")
    assert engine.default_devices() == (0, 1)
    assert engine.default_devices(max_workers=1) == (0,)
