"""GPU: the N>1 path with the real HIP kernels. Three ranks (gloo group, all on GPU 0) shard the jobs
of the golden 8-patch measurement, count their share on the device and all-reduce the tensor; every
rank must hold the reference's result."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, backend="gloo", device_reduce=False, launcher="torchrun"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if launcher == "torchrun":  # LOCAL_RANK per process; YAW_AMD_DEVICE lets the ranks share the one GPU of this box
        os.environ.update(LOCAL_RANK=str(rank), YAW_AMD_DEVICE="0")
    else:  # mpirun / srun style: no LOCAL_RANK, the launcher names ONE device per process in YAW_AMD_DEVICES
        os.environ.pop("LOCAL_RANK", None)
        os.environ.pop("YAW_AMD_DEVICE", None)
        os.environ["YAW_AMD_DEVICES"] = "0"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import helpers
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import measurements

    measurements.FORCE_DEVICE_REDUCE = device_reduce
    if launcher != "torchrun":  # the counting context and the collectives must name the same device (round-3 advice)
        from yet_another_wizz_amd import engine, parallel

        assert engine.default_devices() == (0,) and parallel.local_device_index() == 0
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        inp, cats = helpers.full_catalogs("u")
        config = helpers.full_config(inp, "s2", "right")
        cfs = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
        exp = helpers.load_golden("full_u_s2_right.npz")
        helpers.check_corrfuncs("cross", cfs, exp, exact=lambda kind: True)  # unweighted: bit-identical on every rank
        acf = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"])
        helpers.check_corrfuncs("auto", acf, exp, exact=lambda kind: True)
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), cfs[0].dd.counts.counts)
    finally:
        dist.destroy_process_group()


def test_three_ranks_share_one_gpu(tmp_path):
    import torch.multiprocessing as mp

    world = 3
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    results = [np.load(tmp_path / f"ok{r}.npy") for r in range(world)]
    assert all(np.array_equal(results[0], r) for r in results[1:]) and results[0].sum() > 0


def test_three_ranks_device_resident_rows(tmp_path):
    """The device-resident route of the RCCL path (``yawhip_count_pairs_rows_device``: every rank's rows scattered into
    their place of the full tensor ON the GPU), rehearsed by three gloo ranks on one GPU: the reduce itself then runs on the
    host, everything before it is the production code."""
    import torch.multiprocessing as mp

    world = 3
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), "gloo", True), nprocs=world, join=True,
                       start_method="spawn")
    results = [np.load(tmp_path / f"ok{r}.npy") for r in range(world)]
    assert all(np.array_equal(results[0], r) for r in results[1:]) and results[0].sum() > 0


def test_one_rank_rccl_reduce_on_the_device(tmp_path):
    """backend "nccl" (RCCL) with the one GPU this box has: a group of ONE rank takes the several-ranks route -- partition,
    rows left in HBM, ``torch.as_tensor`` on the library's buffer through ``__cuda_array_interface__``, an RCCL all-reduce
    on the device, one copy back. (More ranks need more GPUs: the driver's 8-GPU run exercises those.)"""
    import torch.multiprocessing as mp

    mp.start_processes(_worker, args=(1, _free_port(), str(tmp_path), "nccl", True), nprocs=1, join=True, start_method="spawn")
    assert np.load(tmp_path / "ok0.npy").sum() > 0


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_ranks_named_their_device_by_yaw_amd_devices(tmp_path, backend, world):
    """Launchers that start one process per GPU without LOCAL_RANK (mpirun, srun) set a one-id YAW_AMD_DEVICES per process:
    the counting context (engine.default_devices) and the collectives (parallel.local_device_index, DeviceRows.device) then use
    that device -- round 3 counted on GPU k and reduced on cuda:0. The device-resident route, two gloo ranks and a one-rank
    RCCL group on the one GPU of this box."""
    import torch.multiprocessing as mp

    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), backend, True, "mpirun"), nprocs=world, join=True,
                       start_method="spawn")
    results = [np.load(tmp_path / f"ok{r}.npy") for r in range(world)]
    assert all(np.array_equal(results[0], r) for r in results[1:]) and results[0].sum() > 0


def test_multi_device_context_one_call():
    """Several GPUs behind ONE context and ONE call (yawhip_ctx_create_multi): the drop-in route -- crosscorrelate(...,
    max_workers=N) in a single process. Rehearsed on the one GPU of this box by listing its id three times (three
    streams, three replicas of every catalogue): the library splits the job list, every device counts its share, the rows
    land in place. Results equal the single-device ones bit for bit (weighted sums included: a job's rows come from one
    device, the arithmetic per job is unchanged)."""
    import helpers
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import _lib, engine

    inp, cats = helpers.full_catalogs("w")
    config = helpers.full_config(inp, "s2", "right")
    single = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
    auto1 = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"])
    os.environ["YAW_AMD_DEVICES"] = "0,0,0"
    try:
        assert engine.default_devices() == (0, 0, 0) and engine.default_devices(2) == (0, 0)
        ctx = engine.get_context()
        assert ctx.devices == (0, 0, 0)
        multi = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
        auto3 = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"])
        two = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"], max_workers=2)
    finally:
        del os.environ["YAW_AMD_DEVICES"]
    for a, b in list(zip(single, multi)) + list(zip(single, two)) + list(zip(auto1, auto3)):
        for kind in ("dd", "dr", "rd", "rr"):
            x, y = getattr(a, kind), getattr(b, kind)
            assert (x is None) == (y is None)
            if x is not None:
                assert np.array_equal(x.counts.counts, y.counts.counts), kind
                assert np.array_equal(x.sum_weights.sum_weights1, y.sum_weights.sum_weights1)
    helpers.check_corrfuncs("cross", multi, helpers.load_golden("full_w_s2_right.npz"), exact=lambda k: False)  # vs the reference
    # the C ABI directly: job split, statistics, a failing call leaves the context usable
    l1, l2 = cats["ref"]._active_layout, cats["unk"]._active_layout
    c3 = _lib.Context([0, 0, 0])
    c1 = _lib.Context(0)
    up = lambda c, l: _lib.DeviceCatalog(c, l.x, l.y, l.z, l.w, l.num_patches, l.num_bins, l.offsets)
    jobs = np.array([(p, q) for p in range(l1.num_patches) for q in range(l1.num_patches)], dtype=np.int32)
    t = np.array([[1e-7, 4e-6, 3e-5]] * l1.num_bins)
    ref_counts, ref_sums, st1 = _lib.count_pairs(c1, up(c1, l1), up(c1, l2), jobs, t, want_counts=True, want_sums=True)
    d1, d2 = up(c3, l1), up(c3, l2)
    for _ in range(2):  # second call: the cached plan
        counts, sums, st3 = _lib.count_pairs(c3, d1, d2, jobs, t, want_counts=True, want_sums=True)
        assert np.array_equal(counts, ref_counts) and np.array_equal(sums, ref_sums)
        assert st3.candidate_pairs == st1.candidate_pairs and st3.evaluated_pairs == st1.evaluated_pairs
    with pytest.raises(_lib.YawhipError, match="ascending"):
        _lib.count_pairs(c3, d1, d2, jobs, t[:, ::-1].copy())
    counts, _, _ = _lib.count_pairs(c3, d1, d2, jobs[:5], t, want_counts=True)
    assert np.array_equal(counts, ref_counts[:5])
    for cat in cats.values():
        cat.drop_layouts()
