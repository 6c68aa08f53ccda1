"""The library's OWN multi-process path (qtr_comm_unique_id / qtr_comm_init / qtr_gather_results[_v], include/quatro_hip.h
"Multi-GPU") with more than one rank — SURVEY section 8(e), BASELINE configs[3]:
  * on a box with >= 2 GPUs: one process per GPU over real RCCL;
  * on a 1-GPU box: three processes sharing the GPU over a transport double (tests/mock/mock_rccl.cpp, selected with
    QTR_RCCL_LIB; real RCCL refuses two ranks on one device) — everything of the library's path except RCCL itself:
    the count exchange, blocks of different lengths padded and trimmed, the rank-uniform early-outs, and a sharded batch
    of composite pairs whose gathered records equal a single-process run.
Each rank is tests/gpu_multi_worker.py in its own process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run_ranks(world, devices, tmp_path, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_multi_worker.py"), str(r), str(world),
                               str(devices[r]), str(tmp_path)], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o[-1500:], e[-3000:]))
    assert all(rc == 0 for rc, _, _ in outs), outs
    return [json.load(open(os.path.join(tmp_path, f"rank{r}.json"))) for r in range(world)]


def _check(world, got):
    from quatro_amd import dist as qdist
    from quatro_amd import lib as ql
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_multi_worker as w
    n_local = [[3, 2, 0, 1, 4, 0, 2, 1][r % 8] for r in range(world)]
    want = [[100 * r + i, r + 0.25 * i, float(r * 1000 + i * 16 + 5)] for r in range(world) for i in range(n_local[r])]
    for g in got:   # every rank holds the same, complete result
        assert g["uneven"]["counts"] == n_local and g["uneven"]["n_all"] == sum(n_local)
        assert g["uneven"]["records"] == want
        assert g["refuse"] == ql.QTR_ERR_CAPACITY      # the last rank had room for one record: everybody is told
        assert g["unequal"] == ql.QTR_ERR_BAD_ARG      # blocks of different lengths through the equal-length entry
        assert g["equal"] == [7 + r for r in range(world)]
    # the sharded batch against one process doing all of it
    items = w.work_items()
    h = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=8192, n_slots=4)
    try:
        ref = h.register_batch(items, want_lists=False)
    finally:
        h.close()
    sizes = [qdist.shard_range(w.N_IDS, r, world) for r in range(world)]
    for g in got:
        assert g["work"]["counts"] == [hi - lo for lo, hi in sizes] and g["work"]["n_all"] == w.N_IDS
        for rec, r in zip(g["work"]["records"], ref):
            assert rec[:7] == [r["status"], int(r["valid"]), r["n_clique"], r["n_final"], r["L"], r["n_src"], r["n_tgt"]]
            assert np.array_equal(np.array(rec[7]).reshape(4, 4), r["T"])
            assert r["valid"] and r["L"] == 1500


def test_library_gather_over_real_rccl_one_process_per_gpu(tmp_path):
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (real RCCL refuses two ranks on one device); the 1-GPU form runs below")
    world = min(n, 4)
    _check(world, _run_ranks(world, list(range(world)), tmp_path, {}))


def test_library_gather_across_three_processes_sharing_one_gpu(tmp_path):
    so = os.path.join(str(tmp_path), "libmock_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-std=c++17", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tests", "mock", "mock_rccl.cpp"), "-o", so, "-lrt"])
    world = 3
    _check(world, _run_ranks(world, [0] * world, tmp_path, {"QTR_RCCL_LIB": so}))


def test_two_processes_solving_side_by_side_stay_correct_and_bounded(tmp_path):
    """k_hcore_async's workgroups wait for one another and want a compute unit each: two PROCESSES (two handles that know
    nothing of each other) solving L = 5000 back to back on one GPU can leave both launches partly resident.  The
    residency timeout (2 ms) then hands the pair to the peeling workgroup: every result stays the oracle's, and no solve
    takes anywhere near the seconds the round-3 kernel waited."""
    code = r'''
import sys, time, json, numpy as np
sys.path.insert(0, %r)
import torch
from quatro_amd import lib as ql, synth
h = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=8192)
s, t, _, _ = synth.correspondences(5000, 0.05, seed=int(sys.argv[1]), noise=0.1)
first = h.solve(s, t)
worst, same = 0.0, True
t_end = time.time() + 6.0
n = 0
while time.time() < t_end:
    t0 = time.perf_counter()
    r = h.solve(s, t)
    worst = max(worst, time.perf_counter() - t0)
    same = same and np.array_equal(r["T"], first["T"]) and np.array_equal(r["clique"], first["clique"])
    n += 1
st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
json.dump({"n": n, "worst_ms": 1e3 * worst, "same": bool(same), "clique": int(first["clique"].size),
           "T": first["T"].tolist()}, open(sys.argv[2], "w"))
h.close()
''' % ROOT
    outs = [os.path.join(str(tmp_path), f"p{i}.json") for i in range(2)]
    procs = [subprocess.Popen([sys.executable, "-c", code, "7", outs[i]], cwd=ROOT, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for i in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
    res = [json.load(open(f)) for f in outs]
    from oracle import oracle as qo
    from quatro_amd import synth
    s, t, _, _ = synth.correspondences(5000, 0.05, seed=7, noise=0.1)
    o = qo.solve(s, t)
    for r in res:
        assert r["same"] and r["n"] > 50
        assert r["clique"] == o["clique"].size and np.array_equal(np.array(r["T"]), o["T"])
        assert r["worst_ms"] < 250.0, r   # (round 3: a partly resident pair spun for seconds before it fell back)
