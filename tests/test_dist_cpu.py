"""The N>1 path on CPU: pair sharding, the final gather of result records and the max-over-ranks timing,
exercised with world_size 2 over gloo (the same code runs over RCCL on the GPUs)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_shard_range_partitions_everything():
    from quatro_amd.dist import shard_range
    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quatro_amd import dist as qd
    lo, hi = qd.shard_range(6, rank, world)
    recs = []
    for pid in range(lo, hi):
        T = np.eye(4)
        T[0, 3] = pid
        recs.append(qd.pack_record(pid, {"T": T, "cost": 0.5 * pid, "valid": True, "clique": np.arange(pid + 2),
                                         "final_inliers": np.arange(pid + 1), "L": 100 + pid, "n_rot_inliers": pid}))
    g = qd.gather_records(np.stack(recs))
    tmax = qd.max_over_ranks(1.0 + rank)
    every = qd.all_over_ranks(10.0 + rank)  # (the per-rank figures of bench.py's sharded leg)
    assert every == [10.0 + r for r in range(world)]
    if rank == 0:
        q.put((g, tmax))
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_max_over_ranks_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert g.shape == (6, 24)
    assert g[:, 22].tolist() == [0, 1, 2, 3, 4, 5]          # pair ids in rank order
    assert g[:, 3].tolist() == [0, 1, 2, 3, 4, 5]           # T[0,3]
    assert g[:, 18].tolist() == [2, 3, 4, 5, 6, 7]          # clique sizes
    assert tmax == 2.0


def _worker_real(rank, world, port, q):
    """Every rank registers ITS block of pair ids (the CPU oracle stands in for the GPU worker: same records) and the
    records meet on rank 0 — the N > 1 path of bench.py / BASELINE configs[3] with real registrations."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as qo
    from quatro_amd import dist as qd
    from quatro_amd import synth
    qo.set_threads(1)
    lo, hi = qd.shard_range(7, rank, world)  # 7 ids over 2 ranks: blocks of 4 and 3
    recs = []
    for pid in range(lo, hi):
        src, tgt, _, _ = synth.correspondences(200 + 10 * pid, 0.3, seed=pid, noise=0.2)
        r = qo.solve(src, tgt)
        r["L"] = src.shape[0]
        recs.append(qd.pack_record(pid, r))
    g = qd.gather_records(np.stack(recs))
    if rank == 0:
        q.put(g)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_registrations_gather_on_rank0_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    from oracle import oracle as qo
    from quatro_amd import synth
    qo.set_threads(1)
    assert g.shape == (7, 24) and g[:, 22].tolist() == list(range(7))
    for pid in range(7):  # the gathered records equal a serial run of the same ids
        src, tgt, Tgt, _ = synth.correspondences(200 + 10 * pid, 0.3, seed=pid, noise=0.2)
        r = qo.solve(src, tgt)
        assert np.array_equal(g[pid, :16].reshape(4, 4), r["T"]) and g[pid, 18] == len(r["clique"])
        assert g[pid, 17] == 1.0 and np.abs(r["T"][:3, 3] - Tgt[:3, 3]).max() < 0.2


def test_cpulist_parsing_and_pinning_without_a_gpu():
    """bench.py pins each rank to its GPU's NUMA-local cores (sysfs local_cpulist); without a GPU nothing is changed."""
    import os

    from quatro_amd import dist as qd

    assert qd.parse_cpulist("64-127,192-255") == list(range(64, 128)) + list(range(192, 256))
    assert qd.parse_cpulist("3") == [3] and qd.parse_cpulist("") == [] and qd.parse_cpulist("0-1, 5\n") == [0, 1, 5]
    before = os.sched_getaffinity(0)
    assert qd.pin_to_device_node(0) is None
    assert os.sched_getaffinity(0) == before
