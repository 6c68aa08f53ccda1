"""Multi-GPU plumbing: independent scan pairs shard across ranks; there is no data-path collective.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).  The only
exchange is the final gather of fixed-size result records (north_star: "RCCL over xGMI used only for the
final gather") plus the barrier / max-over-ranks timing reduction of the bench contract.
"""
from __future__ import annotations

import numpy as np

RECORD_DOUBLES = 24  # T[16], cost, valid, n_clique, n_rot_inliers, n_final, n_corr, pair_id, pad


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition [lo, hi) of range(n_items) for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_record(pair_id: int, res: dict) -> np.ndarray:
    r = np.zeros(RECORD_DOUBLES, dtype=np.float64)
    r[:16] = np.asarray(res["T"], dtype=np.float64).reshape(-1)
    r[16] = res["cost"] if np.isfinite(res["cost"]) else -1.0
    r[17] = float(bool(res["valid"]))
    r[18] = len(res["clique"])
    r[19] = res.get("n_rot_inliers", 0) or 0
    r[20] = len(res["final_inliers"])
    r[21] = res.get("L", 0)
    r[22] = pair_id
    return r


def gather_records(local: "np.ndarray", device=None):
    """Gathers [n_local, RECORD_DOUBLES] records from every rank onto rank 0, in rank order (block partitions of
    shard_range differ by at most one record: blocks are padded to the longest and trimmed after the gather).
    Returns the concatenated array on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1, RECORD_DOUBLES)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else "cpu"
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    nmax = max(max(counts), 1)
    pad = np.zeros((nmax, RECORD_DOUBLES), dtype=np.float64)
    pad[: local.shape[0]] = local
    t = torch.from_numpy(pad).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)  # fixed-size collective on both back ends (RCCL has no gather-to-root primitive)
    if rank != 0:
        return None
    return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(out, counts)], axis=0)


def max_over_ranks(seconds: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_over_ranks(value: float, device=None) -> list:
    """Every rank's `value`, in rank order, on every rank (one fixed-size all-gather: the per-rank figures of a sharded
    leg, so that a scaling line can be read rank by rank without a second run)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def parse_cpulist(text: str) -> list:
    """'64-127,192-255' (sysfs local_cpulist) -> sorted list of CPU numbers; '' -> []."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def pin_to_device_node(device_index: int):
    """Restrict this process to the CPU cores of the GPU's NUMA node (one process per GPU: every hand-over of the path — voxel
    counts, matcher counts, the result — crosses PCIe twice, and from the other socket each costs a microsecond or two more:
    profiles/r6_ab.txt section 14).  Returns the sysfs cpulist that was applied, or None when there is nothing to go by (no sysfs
    entry, an affinity mask that already excludes those cores, a platform without sched_setaffinity)."""
    import os

    try:
        import torch

        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as fh:
            text = fh.read().strip()
        want = set(parse_cpulist(text)) & set(os.sched_getaffinity(0))
        if not want:
            return None
        os.sched_setaffinity(0, want)
        return text
    except Exception:  # (no GPU, no such attribute in this torch, no sysfs entry, ...): leave the affinity as it is
        return None
