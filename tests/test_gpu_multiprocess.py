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


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), YAW_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import helpers
    import yet_another_wizz_amd as yaw

    dist.init_process_group("gloo", rank=rank, world_size=world)
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
