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
