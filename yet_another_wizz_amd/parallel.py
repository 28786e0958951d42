"""Distribution of patch-pair jobs over GPUs: one process per GPU (``torch.distributed``; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

Replaces the reference's task farm (src/yaw/utils/parallel.py:251-346: multiprocessing pool or
mpi4py root/worker send-recv).  Jobs are independent and write disjoint ``[.,.,i,j]`` slots of the
result tensor, so the only exchange step is one sum all-reduce of that tensor at the end; every
slot is non-zero on exactly one rank, which makes the sum exact and order independent.
"""
from __future__ import annotations

import os

import numpy as np

__all__ = ["world", "partition_jobs", "allreduce_sum", "allreduce_device_rows", "device_collectives", "broadcast_object",
           "local_device_index"]


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # torch missing: single process only
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    dist = _dist()
    return (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)


def local_device_index() -> int:
    """GPU of this process: LOCAL_RANK as set by ``python -m torch.distributed.run``; ``YAW_AMD_DEVICE`` overrides it
    (e.g. to let several ranks share one GPU in a test), and so does a ``YAW_AMD_DEVICES`` that names ONE id (what
    mpirun / srun launchers set per process): the counting context (``engine.default_devices``) and the collectives
    then agree on the device whichever of the two variables a launcher uses."""
    one = os.environ.get("YAW_AMD_DEVICE")
    if one is None:
        ids = [v for v in os.environ.get("YAW_AMD_DEVICES", "").split(",") if v.strip() != ""]
        if len(ids) == 1:
            one = ids[0]
    return int(one if one is not None else os.environ.get("LOCAL_RANK", "0"))


def partition_jobs(costs, num_parts: int) -> list:
    """Longest-processing-time-first greedy assignment. Returns ``num_parts`` sorted index arrays.

    The reference also schedules the heaviest (diagonal) jobs first (measurements.py:262-273);
    with a static assignment the same idea becomes LPT over the candidate-pair cost N1*N2."""
    costs = np.asarray(costs, dtype=np.float64)
    loads = np.zeros(num_parts)
    parts = [[] for _ in range(num_parts)]
    for j in np.argsort(-costs, kind="stable"):
        dest = int(np.argmin(loads))
        parts[dest].append(int(j))
        loads[dest] += costs[j]
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def allreduce_sum(array: np.ndarray, device: int | None = None) -> np.ndarray:
    """Sum ``array`` (int64 or float64) over all ranks; identity for a single process. ``device``: the GPU the RCCL
    backend stages the array on (default: this process' own, ``local_device_index``)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return array
    import torch

    tensor = torch.from_numpy(np.ascontiguousarray(array))
    if dist.get_backend() == "nccl":
        tensor = tensor.to(torch.device("cuda", local_device_index() if device is None else int(device)))
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor.cpu().numpy()


def device_collectives() -> bool:
    """True when the process group reduces device memory directly (backend "nccl" = RCCL over xGMI)."""
    dist = _dist()
    return dist is not None and dist.get_backend() == "nccl"


def allreduce_device_rows(rows, n: int, status: float = 0.0, device: int | None = None, timings: dict | None = None) -> np.ndarray:
    """Sum all-reduce of a float64[n] tensor that already lives on this rank's GPU (``_lib.DeviceRows``: the library's
    result buffer, wrapped by torch without a copy) -- or, for a rank that has nothing to contribute (``rows`` None), of
    zeros. ``status`` is added to the last element (the failure flag of ``PatchLinkage.count_pairs``). One device-to-host
    copy brings the reduced tensor back. The collective runs on the GPU that HOLDS the rows (``rows.device``; ``device``
    for a rank without rows): one source of truth, the counting context's device -- not a second reading of the
    environment. ``timings``: receives ``allreduce_ms`` (the collective, waited for) and ``copy_back_ms``."""
    import time

    import torch

    dist = _dist()
    index = rows.device if rows is not None else (local_device_index() if device is None else int(device))
    device = torch.device("cuda", index)
    tensor = torch.zeros(n, dtype=torch.float64, device=device) if rows is None else torch.as_tensor(rows, device=device)
    if status:
        tensor[-1] += status
    t0 = time.perf_counter()
    if dist is not None and dist.get_backend() == "nccl":
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)  # RCCL, on the device
        if timings is not None:
            torch.cuda.synchronize(device)
            timings["allreduce_ms"] = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
        out = tensor.cpu().numpy()
        if timings is not None:
            timings["copy_back_ms"] = (time.perf_counter() - t0) * 1e3
        return out
    host = tensor.cpu()  # rehearsals on another backend (several ranks sharing one GPU under gloo): reduce on the host
    if timings is not None:
        timings["copy_back_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
    if dist is not None and dist.get_world_size() > 1:
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
    if timings is not None:
        timings["allreduce_ms"] = (time.perf_counter() - t0) * 1e3
    return host.numpy()


def broadcast_object(obj, src: int = 0, device: int | None = None):
    """``obj`` of rank ``src`` on every rank (a plan such as the job partition: small, sent once); identity for a
    single process. ``device``: the GPU the RCCL backend sends it through (default ``local_device_index``)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return obj
    box = [obj if dist.get_rank() == src else None]
    if dist.get_backend() == "nccl":  # the object travels through this rank's GPU: name it (torch's current device may be another)
        import torch

        dist.broadcast_object_list(box, src=src, device=torch.device("cuda", local_device_index() if device is None else int(device)))
    else:
        dist.broadcast_object_list(box, src=src)
    return box[0]
