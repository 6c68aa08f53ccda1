"""One rank of tests/test_gpu_multi.py: `python tests/gpu_multi_worker.py <rank> <world> <device> <dir>`.

Joins the library's communicator (qtr_comm_unique_id through a file written by rank 0), then runs the scenarios and
writes what it saw to <dir>/rank<r>.json.  Scenarios:
  uneven     qtr_gather_results_v with blocks of 3, 2, 0, 1, ... records
  refuse     a rank with too little room: EVERY rank gets QTR_ERR_CAPACITY (nobody is left in the second collective)
  unequal    qtr_gather_results with different block lengths: QTR_ERR_BAD_ARG on every rank
  work       configs[3] in small: a fixed set of composite pair ids (scan pair + 1500 given correspondences)
             block-partitioned over the ranks (quatro_amd.dist.shard_range), each rank's block through
             qtr_submit_batch / qtr_wait, records gathered with qtr_gather_results_v
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime per process: torch's)

from quatro_amd import dist as qdist  # noqa: E402
from quatro_amd import lib as ql  # noqa: E402
from quatro_amd import synth  # noqa: E402

N_IDS = 11


def work_items():
    scans = [synth.kitti64_pair(i) for i in range(2)]
    corr = [synth.correspondences(1500, 0.08, seed=20 + k, noise=0.1) for k in range(3)]
    return [(scans[i % 2][0], scans[i % 2][1], i % 4, corr[i % 3][0], corr[i % 3][1]) for i in range(N_IDS)]


def rec_tuple(r):
    return [int(r.status), int(r.valid), int(r.n_clique), int(r.n_final), int(r.n_corr), int(r.n_src), int(r.n_tgt),
            [float(x) for x in r.T[:]]]


def main():
    rank, world, device, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    idfile = os.path.join(d, "unique_id.bin")
    if rank == 0:
        uid = ql.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise RuntimeError("rank 0 never wrote the unique id")
            time.sleep(0.02)
        uid = open(idfile, "rb").read()
    h = ql.Handle(device, max_points=131072, max_voxels=32768, max_corr=8192, n_slots=4)
    out = {"rank": rank}
    try:
        h.comm_init(uid, rank, world)
        # ---- uneven blocks
        n_local = [3, 2, 0, 1, 4, 0, 2, 1][rank % 8]
        res = (ql.Result * max(n_local, 1))()
        for i in range(n_local):
            res[i].status = 0
            res[i].valid = 1
            res[i].n_clique = 100 * rank + i
            res[i].cost = rank + 0.25 * i
            for k in range(16):
                res[i].T[k] = rank * 1000 + i * 16 + k
        allr, counts, n_all = h.gather_results_v(res if n_local else None, n_local, world, 64)
        out["uneven"] = {"counts": counts, "n_all": n_all,
                         "records": [[int(allr[i].n_clique), float(allr[i].cost), float(allr[i].T[5])] for i in range(n_all)]}
        # ---- a rank with too little room: every rank is told, nobody hangs
        cap = 1 if rank == world - 1 else 64
        try:
            h.gather_results_v(res if n_local else None, n_local, world, cap)
            out["refuse"] = "ok"
        except ql.QuatroHipError as e:
            out["refuse"] = e.code
        # ---- equal-length gather with unequal blocks
        try:
            h.gather_results(res, world) if n_local else h._check(h._lib.qtr_gather_results(h._h, None, 0, None))
            out["unequal"] = "ok"
        except ql.QuatroHipError as e:
            out["unequal"] = e.code
        # ---- and an equal-length one that must work
        one = (ql.Result * 1)()
        one[0].n_clique = 7 + rank
        eq = h.gather_results(one, world)
        out["equal"] = [int(eq[i].n_clique) for i in range(world)]
        # ---- the sharded batch
        items = work_items()
        lo, hi = qdist.shard_range(N_IDS, rank, world)
        mine = items[lo:hi]
        B = len(mine)
        descs = (ql.PairDesc * max(B, 1))()
        results = (ql.Result * max(B, 1))()
        keep = []
        for i, (s, t, seed, cs, ct) in enumerate(mine):
            arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (s, t, cs, ct)]
            keep.append(arrs)
            descs[i] = ql.PairDesc(arrs[0].ctypes.data, arrs[0].shape[0], arrs[1].ctypes.data, arrs[1].shape[0], seed, None,
                                   None, 0, arrs[2].ctypes.data, arrs[3].ctypes.data, arrs[2].shape[0])
        fp, prm = ql.default_frontend_params(), ql.demo_params()
        h._check(h._lib.qtr_submit_batch(h._h, descs, B, C.byref(fp), C.byref(prm), results, ql.MEM_HOST))
        h._check(h._lib.qtr_wait(h._h))
        allr, counts, n_all = h.gather_results_v(results if B else None, B, world, N_IDS)
        out["work"] = {"counts": counts, "n_all": n_all, "records": [rec_tuple(allr[i]) for i in range(n_all)]}
    finally:
        h.close()
    with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
