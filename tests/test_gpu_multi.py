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


def _mock_rccl(tmp_path):
    so = os.path.join(str(tmp_path), "libmock_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-std=c++17", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tests", "mock", "mock_rccl.cpp"), "-o", so, "-lrt"])
    return so


def test_library_gather_with_world_size_eight_sharing_one_gpu(tmp_path):
    """The communicator at the size the 8-GPU run uses: eight processes, qtr_comm_init(world = 8), uneven blocks (3, 2, 0, 1,
    4, 0, 2, 1 records), the rank-uniform refusals, and the eleven composite ids split 2/1/1/2/1/1/2/1 over the ranks."""
    _check(8, _run_ranks(8, [0] * 8, tmp_path, {"QTR_RCCL_LIB": _mock_rccl(tmp_path)}))


def test_bench_line_of_a_four_rank_launch_as_the_driver_starts_it(tmp_path):
    """First-contact insurance for the multi-GPU bench: `python -m torch.distributed.run --nproc-per-node 4 bench.py --gpus 4`
    exactly as the driver launches it — on this one-GPU box under QTR_BENCH_ONE_DEVICE=1 (every rank computes on cuda:0,
    torch's collectives over gloo, the library's gather over the transport double) — prints ONE JSON line whose N > 1 keys
    are all there: the weak-scaling headline, configs[3]'s sharded leg closed by the library's gather, the CPU baseline and
    the parity record from rank 0's host."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", QTR_BENCH_ONE_DEVICE="1", QTR_RCCL_LIB=_mock_rccl(tmp_path),
               MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "1",
           "--legs", "batch", "--cpu-seconds", "2", "--sharded-pairs", "96", "--batch-slots", "8"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["steps"] == 4 and out["scaling"] == "weak" and out["value"] > 0
    assert out["metric"].startswith("scan-pair registrations/sec") and out["unit"] == "registrations/s"
    assert out["config"]["records_gathered"] == 4 * len(out["config"]["pool"])
    sh = out["sharded_leg"]
    assert sh["pairs"] == 96 and sh["n_corr"] == 5000 and sh["identical_to_sequential"] and sh["value"] > 0
    assert sh["gather"]["ok"] is True and sh["gather"]["records"] == 96 and sh["gather"]["all_valid"]
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] in ("port", "reference")
    assert out["parity_vs_oracle"]["all_pool_pairs_ok"] is True
    assert "roofline" in out and out["roofline"]["frac"] > 0
