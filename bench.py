#!/usr/bin/env python
"""bench.py — registrations/sec of the whole hot path (voxel-FPFH -> matching -> consistency graph ->
max-clique -> GNC-TLS -> COTE) on synthetic KITTI-64-shaped scan pairs resident in HBM.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run, one rank per GPU.  A step = one registration of one scan pair (BASELINE.json
configs[1]: a single KITTI-64 pair, whole path on the GPU).  Every rank times exactly K steps: the pair ids
[0, N*K) are block-partitioned over the ranks (quatro_amd.dist.shard_range, the partition of BASELINE configs[3]), rank r
registers ids [r*K, (r+1)*K) one at a time — per-GPU work is fixed as N grows ("weak" scaling) and `value` is the
whole job, N*K registrations over the slowest rank's time.  Pair id -> synthetic pair is id % pool.  Pairs are
independent, so there is no data-path collective; the only exchange is the gather of the fixed-size result records
(RCCL) after the timed region plus the barrier / max-over-ranks of the contract.  For N > 1 the line also carries
`sharded_leg`: configs[3] itself — a FIXED set of 4096 pair ids block-partitioned over the ranks and streamed through
the batched entry points (strong scaling).  Prints ONE JSON line on rank 0.

Objects in the line next to the contract's keys:
  roofline      — dominant kernel k_nn_f16: the 33-D distance matrix nb' - 2 a.b evaluated on the f16 matrix pipe with
                  every f32 operand split in two halves (3 x 33 products + 3 norm slots = 102 per matrix entry, see
                  match.hip).  `achieved` = 2 * 102 * n_query * n_base FLOP per launch (K padding to 112 not counted)
                  / mean launch duration (HIP events recorded by the library on the launch stream), `peak` = the dense
                  f16/bf16 MFMA peak.  `f32_equivalent` restates the same launches in SURVEY.md section 8(d)'s unit
                  (66 * n_query * n_base FLOP of an f32 evaluation) against the FP32 matrix peak — the figure earlier
                  rounds reported for the f32 kernel k_nn_mfma (QTR_NN_ENGINE=mfma32 still runs it, and is then the
                  kernel this object describes).
                  `end_to_end`: the registration's algorithmic FLOP and bytes priced at the
                  FP32-matrix / HBM peaks, over the measured time per step.
  cpu_baseline  — the CPU oracle (a port: the reference cannot be built here; brute-force NN instead of FLANN
                  kd-trees) on this box's host cores, swept over OMP thread counts on a bounded sample; `value` is the
                  best setting, `omp4` the reference README's 4-thread setting.  A reported baseline, not the target.
  solver_L5000_leg, batch256_leg, dense_leg, sharded_leg — the other BASELINE configs, never part of `value`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix (v_mfma_f32_32x32x2_f32) = FP32 vector peak
F16_PEAK_TFLOPS = 2500.0  # same guide: BF16/FP16 MFMA, dense (v_mfma_f32_32x32x16_f16: 32 cycles per SIMD)
NN_ENGINE = os.environ.get("QTR_NN_ENGINE", "f16")  # f16 (default): k_nn_f16; mfma32: k_nn_mfma; exact: no MFMA kernel
F16_FLOP_PER_ENTRY, F32_FLOP_PER_ENTRY = 204.0, 66.0


def nn_roofline(entries_per_launch, mean_launch_s):
    """roofline fields of the nearest-neighbour kernel for one launch of `entries_per_launch` distance-matrix entries"""
    f32_rate = F32_FLOP_PER_ENTRY * entries_per_launch / mean_launch_s / 1e12
    f32eq = {"flop_per_launch": F32_FLOP_PER_ENTRY * entries_per_launch, "achieved": f32_rate, "peak": FP32_PEAK_TFLOPS,
             "unit": "TFLOP/s", "ratio": f32_rate / FP32_PEAK_TFLOPS,
             "note": "SURVEY 8(d) row E unit: 33 multiply-adds per matrix entry as an f32 evaluation would need"}
    if NN_ENGINE == "mfma32":
        return {"kernel": "k_nn_mfma (33-D distance matrix on v_mfma_f32_32x32x2_f32 + per-query top-2, one launch per direction)",
                "bound": "mfma", "dtype": "f32", "achieved": f32_rate, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": f32_rate / FP32_PEAK_TFLOPS, "flop_per_launch": F32_FLOP_PER_ENTRY * entries_per_launch,
                "mean_launch_ms": 1e3 * mean_launch_s}
    rate = F16_FLOP_PER_ENTRY * entries_per_launch / mean_launch_s / 1e12
    return {"kernel": "k_nn_f16 (33-D distance matrix on v_mfma_f32_32x32x16_f16, f32 operands split in two f16 halves, "
                      "+ per-query top-2; one launch per direction)",
            "bound": "mfma", "dtype": "f16 (2-way split of f32 operands, f32 accumulate)", "achieved": rate,
            "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rate / F16_PEAK_TFLOPS,
            "flop_per_launch": F16_FLOP_PER_ENTRY * entries_per_launch, "mean_launch_ms": 1e3 * mean_launch_s,
            "f32_equivalent": f32eq}
HBM_PEAK_GBS = 8000.0
METRIC = "scan-pair registrations/sec (KITTI 64-ch, ~5k corr) + rot/trans err vs ref"  # BASELINE.json, verbatim


def algorithmic_work(P_s, P_t, n_s, n_t, L, M):
    """bytes (each logical array once written + once read at a stage boundary) and FLOP (the
    33-D distance matrix, once) of one registration."""
    b = 0.0
    for P, n in ((P_s, n_s), (P_t, n_t)):
        b += 16.0 * P + 16.0 * n        # A voxel grid
        b += 32.0 * n                   # B normals
        b += 164.0 * n                  # C SPFH
        b += 264.0 * n                  # D FPFH
    b += 140.0 * (n_s + n_t)            # E matching (descriptors + NN tables)
    b += 48.0 * L + L * L / 8.0         # G consistency graph (bit matrix)
    b += L * L / 8.0                    # H clique search (one read)
    b += 48.0 * M + 128.0               # I GNC + COTE
    return b, 66.0 * n_s * n_t


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic pairs (pair id -> pair id %% pool)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--legs", default="solver5k,batch,dense,segment,patchwork",
                    help="comma list of the extra legs to run on rank 0 / all ranks (never part of `value`)")
    ap.add_argument("--batch-pairs", type=int, default=256, help="pairs of the batch256 leg (BASELINE configs[2])")
    ap.add_argument("--sharded-pairs", type=int, default=4096, help="pair ids of the N>1 sharded leg (configs[3])")
    ap.add_argument("--batch-slots", type=int, default=32, help="stream slots of the batched legs (two lanes of half as many pairs)")
    args = ap.parse_args()
    legs = set(x for x in args.legs.split(",") if x)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # QTR_BENCH_ONE_DEVICE=1 (test hook for single-GPU boxes): every rank computes on cuda:0 and the collectives go
    # through gloo on host tensors, so that the N > 1 logic (partition, barriers, gather, sharded leg) can be exercised
    one_dev = os.environ.get("QTR_BENCH_ONE_DEVICE") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_dev:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    cdev = None if (world > 1 and one_dev) else dev  # where the collectives' tensors live

    from quatro_amd import dist as qdist
    from quatro_amd import lib as ql
    from quatro_amd import synth

    h = ql.Handle(local_rank, max_points=131072, max_voxels=32768, max_corr=8192)
    prm = ql.demo_params()
    res = ql.Result()

    # ---- synthetic inputs (the same pool on every rank), resident in HBM before the timed region
    pool = []
    for pid in range(args.pool):
        s, t, Tgt = synth.kitti64_pair_16k(pid)
        pool.append({"id": pid, "src_h": s, "tgt_h": t, "Tgt": Tgt, "src": torch.from_numpy(s).to(dev),
                     "tgt": torch.from_numpy(t).to(dev), "fp": ql.default_frontend_params(seed=pid)})
    torch.cuda.synchronize()

    def step(p, handle=h, r=res, slot=0):
        rc = handle.register_pair_dev(p["src"].data_ptr(), p["src"].shape[0], p["tgt"].data_ptr(), p["tgt"].shape[0],
                                      p["fp"], prm, r, slot)
        if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
            raise ql.QuatroHipError(rc, handle.last_error())

    # sizes of every pool pair (one untimed registration each): needed for the per-launch FLOP accounting
    for p in pool:
        step(p)
        p["n_src"], p["n_tgt"], p["L"], p["M"] = res.n_src, res.n_tgt, res.n_corr, res.n_clique
        p["n_hit"] = int(h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)[7])  # rows the second NN direction is asked for
    lo, hi = qdist.shard_range(world * args.steps, rank, world)  # = [rank * K, (rank + 1) * K)
    for w in range(args.warmup):
        step(pool[w % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # the timed region is the registrations and nothing else: the per-stage events are switched off (stage_ms comes from
    # a short untimed pass below); the two nearest-neighbour launches of every step keep their event pairs on the launch
    # stream and the library adds their elapsed times up (read once, after the region)
    h.set_stage_events(False)
    h.nn_totals(reset=True)
    todo = [pool[k % len(pool)] for k in range(lo, hi)]
    t0 = time.perf_counter()
    for p in todo:
        step(p)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = qdist.max_over_ranks(elapsed, cdev)
    my_steps = max(hi - lo, 1)
    nn_ms, nn_launches = h.nn_totals()
    nn_flop, alg_bytes, alg_flop = 0.0, 0.0, 0.0
    for p in todo:
        # launch 1: every row of the smaller cloud against the larger one; launch 2: the hit rows of the larger cloud
        # against the smaller one
        nn_flop += 66.0 * p["n_src"] * p["n_tgt"] + 66.0 * p["n_hit"] * min(p["n_src"], p["n_tgt"])
        b_, f_ = algorithmic_work(p["src"].shape[0], p["tgt"].shape[0], p["n_src"], p["n_tgt"], p["L"], p["M"])
        alg_bytes += b_
        alg_flop += f_
    h.set_stage_events(True)
    stage_acc, stage_n = {}, 0
    for p in pool:  # untimed: where the time goes, stage by stage (events between the stages)
        step(p)
        for key, v in h.stage_times().items():
            stage_acc[key] = stage_acc.get(key, 0.0) + float(v)
        stage_n += 1

    # ---- result records of the pool, gathered on rank 0 (the path's only collective)
    recs = []
    for p in pool:
        r = h.register_pair(p["src_h"], p["tgt_h"], p["fp"], prm)
        p["result"] = r
        recs.append(qdist.pack_record(p["id"], r))
    gathered = qdist.gather_records(np.stack(recs), cdev)

    extra = {}
    # ---- BASELINE configs[2]: a batch of independent pairs streamed through one GPU
    if "batch" in legs and hasattr(h, "register_batch_dev"):
        extra["batch256_leg" if world == 1 else "sharded_leg"] = batch_leg(args, torch, ql, h, pool, prm, dev, world, dist, qdist, cdev)
    # ---- solver alone at the metric's "~5k corr" (the matcher yields fewer on the synthetic scans)
    if "solver5k" in legs and rank == 0:
        extra["solver_L5000_leg"] = solver_leg(args, torch, ql, synth, h, prm, dev, 5000)
    if "dense" in legs and rank == 0 and world == 1:
        extra.update(dense_legs(args, torch, ql, synth, prm, dev, local_rank))
    seg = pwl = None
    if world == 1 and "segment" in legs:
        seg = segment_leg(args, torch, ql, h, pool, dev)
    if world == 1 and "patchwork" in legs:
        pwl = patchwork_leg(args, torch, ql, synth, h, dev)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    p0 = pool[0]
    r0 = p0["result"]
    value = world * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    out = {
        "metric": METRIC,
        "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (voxel grid, FPFH, 33-D matching: f16-split MFMA filter, exact f32 decision) / f64 (consistency graph, GNC-TLS, COTE)", "data": "synthetic",
        "config": {
            "workload": "synthetic KITTI-64-shaped single pair (quatro_amd.synth.kitti64_pair_16k), voxel 0.3 m, whole "
                        "path on GPU, one registration at a time (BASELINE configs[1])",
            "raw_points": [int(p0["src_h"].shape[0]), int(p0["tgt_h"].shape[0])],
            "n_src": int(r0["n_src"]), "n_tgt": int(r0["n_tgt"]), "n_corr": int(r0["L"]),
            "n_clique": int(r0["clique"].size), "n_final_inliers": int(r0["final_inliers"].size),
            "pool": [{"id": p["id"], "n_src": int(p["n_src"]), "n_tgt": int(p["n_tgt"]), "n_corr": int(p["L"]),
                      "n_hit": int(p["n_hit"])}
                     for p in pool],
            "records_gathered": 0 if gathered is None else int(gathered.shape[0]),
            "parallelism": f"pair ids [0,{world * args.steps}) block-partitioned over {world} GPU(s) ({args.steps} per rank), one "
                           "process per GPU, RCCL "
                           "gather of result records",
        },
        "stage_ms": {k: round(v / max(stage_n, 1), 4) for k, v in stage_acc.items() if k not in ("nn_launches",)},
    }
    out.update(extra)
    if seg is not None:
        out["segment_cloud_leg"] = seg
    raw0 = pwl.pop("_raw0") if pwl is not None else None
    if pwl is not None:
        out["patchwork_leg"] = pwl
    ms = h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)
    out["config"]["nn_rows_exact_recheck"] = [int(ms[8]), int(ms[9])]
    out["config"]["n_cross_checked"] = int(ms[3])
    yaw_gt = float(np.arctan2(p0["Tgt"][1, 0], p0["Tgt"][0, 0]))
    yaw = float(np.arctan2(r0["T"][1, 0], r0["T"][0, 0]))
    out["accuracy_vs_ground_truth"] = {
        "rot_err_rad": abs(float(np.arctan2(np.sin(yaw - yaw_gt), np.cos(yaw - yaw_gt)))),
        "trans_err_m": float(np.linalg.norm(r0["T"][:3, 3] - p0["Tgt"][:3, 3])), "valid": bool(r0["valid"])}

    # ---- roofline of the dominant kernel (rank 0's launches), and of the whole registration
    if nn_launches > 0:
        mean_launch_s = 1e-3 * nn_ms / nn_launches
        bound_ms = 1e3 * max(alg_flop / my_steps / (FP32_PEAK_TFLOPS * 1e12), alg_bytes / my_steps / (HBM_PEAK_GBS * 1e9))
        out["roofline"] = nn_roofline(nn_flop / F32_FLOP_PER_ENTRY / nn_launches, mean_launch_s)
        out["roofline"].update({
            "traffic": None, "launches_timed": nn_launches,
            "end_to_end": {
                "algorithmic_gflop_per_registration": alg_flop / my_steps / 1e9,
                "algorithmic_mbytes_per_registration": alg_bytes / my_steps / 1e6,
                "mfma_bound_ms": 1e3 * alg_flop / my_steps / (FP32_PEAK_TFLOPS * 1e12),
                "hbm_bound_ms": 1e3 * alg_bytes / my_steps / (HBM_PEAK_GBS * 1e9),
                "ms_per_step": ms_per_step, "frac": bound_ms / ms_per_step},
        })
        # HBM-side bytes per launch come from the PMC passes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 runs of
        # this same command); counters cannot be read in-process, so the committed summary is quoted
        import glob
        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r2*_pmc_nn.json")))
        if pmc and NN_ENGINE != "mfma32":
            with open(pmc[-1]) as f:
                if "k_nn_f16" not in json.load(f).get("kernel", ""):
                    pmc = []  # the committed counters are of the other kernel
        if pmc:
            with open(pmc[-1]) as f:
                pj = json.load(f)
            out["roofline"]["traffic"] = pj["traffic_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "bytes per launch (FETCH_SIZE + WRITE_SIZE)"
            out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(pmc[-1])

    # ---- CPU baseline: the oracle (port) on this box's host cores, bounded sample; also the parity check
    if world == 1 and args.cpu_seconds > 0:
        out.update(cpu_baseline_leg(args, ql, h, p0, r0, yaw, value, seg, pwl, raw0))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def batch_leg(args, torch, ql, h, pool, prm, dev, world, dist, qdist, cdev):
    """BASELINE configs[2] (and, for N > 1, configs[3]): B pair ids streamed through the batched entry points
    (qtr_submit_batch / qtr_wait), block-partitioned over the ranks."""
    B = args.batch_pairs if world == 1 else args.sharded_pairs
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = qdist.shard_range(B, rank, world)
    hb = ql.Handle(torch.cuda.current_device(), max_points=131072, max_voxels=32768, max_corr=8192, n_slots=args.batch_slots)
    ids = list(range(lo, hi))
    pairs = [pool[i % len(pool)] for i in ids]
    hb.register_batch_dev(pairs[:64], prm)  # warm-up
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    results = hb.register_batch_dev(pairs, prm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = qdist.max_over_ranks(time.perf_counter() - t0, cdev)
    same = all(bool(np.allclose(r["T"], pool[i % len(pool)]["result"]["T"], rtol=0, atol=0)) for i, r in zip(ids, results))
    hb.close()
    return {"what": f"{B} pair ids, block-partitioned over {world} GPU(s), batched launch chains "
                    "(qtr_submit_batch / qtr_wait)", "pairs": B, "value": B / el, "unit": "registrations/s",
            "ms_per_pair": 1e3 * el / B, "identical_to_sequential": same}


def solver_leg(args, torch, ql, synth, h, prm, dev, L):
    res = ql.Result()
    items = []
    for sid in range(4):
        s, t, _, _ = synth.correspondences(L, 0.05, seed=sid, noise=0.1)
        items.append((torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)))
    for s, t in items:
        h.solve_dev(s.data_ptr(), t.data_ptr(), L, prm, res)
    torch.cuda.synchronize()
    n = max(args.steps, 20)
    g_ms = 0.0
    t0 = time.perf_counter()
    for k in range(n):
        s, t = items[k % len(items)]
        rc = h.solve_dev(s.data_ptr(), t.data_ptr(), L, prm, res)
        if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
            raise ql.QuatroHipError(rc, h.last_error())
        g_ms += h.stage_times()["graph"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gk = 1e-3 * g_ms / n
    gb = 48.0 * L + L * L / 8.0
    return {"what": f"Quatro::computeTransformation alone on {L} synthetic correspondences (5 % planted inliers)",
            "value": n / el, "unit": "solves/s", "ms_per_solve": 1e3 * el / n, "n_clique": int(res.n_clique),
            "graph_build": {"bound": "hbm", "algorithmic_bytes": gb, "stage_ms": 1e3 * gk,
                            "achieved": gb / gk / 1e9 if gk > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": (gb / gk / 1e9 / HBM_PEAK_GBS) if gk > 0 else 0.0}}


def dense_legs(args, torch, ql, synth, prm, dev, device_index):
    """BASELINE configs[4]: dense mode — 50 k-point clouds without voxel down-sampling through FPFH + matching (the NN
    contraction at 50 k x 50 k = 1.65e11 FLOP per direction) and the solver at L = 20 000 (50 MB bit matrix)."""
    out = {}
    hd = ql.Handle(device_index, max_points=65536, max_voxels=65536, max_corr=24576)
    res = ql.Result()
    L = 20000
    s, t, _, _ = synth.correspondences(L, 0.02, seed=7, noise=0.1)
    sd, td = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
    hd.solve_dev(sd.data_ptr(), td.data_ptr(), L, prm, res)
    torch.cuda.synchronize()
    n = 5
    g_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        hd.solve_dev(sd.data_ptr(), td.data_ptr(), L, prm, res)
        g_ms += hd.stage_times()["graph"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gk = 1e-3 * g_ms / n
    gb = 48.0 * L + L * L / 8.0
    out["dense_solver_leg"] = {
        "what": f"solver at L = {L} (2 % planted inliers): O(L^2) consistency graph as a {L * L / 8e6:.0f} MB bit matrix",
        "value": n / el, "unit": "solves/s", "ms_per_solve": 1e3 * el / n, "n_clique": int(res.n_clique),
        "roofline": {"kernel": "k_graph_build", "bound": "hbm", "algorithmic_bytes": gb, "stage_ms": 1e3 * gk,
                     "achieved": gb / gk / 1e9 if gk > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (gb / gk / 1e9 / HBM_PEAK_GBS) if gk > 0 else 0.0}}
    # front end at 50 k points per cloud, no voxel grid: qtr_fpfh + qtr_match on device-resident arrays
    n_pts = 50000
    a, b, _ = synth.dense_pair(n_pts)
    cl = [torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)]
    desc = [torch.zeros((n_pts, 33), dtype=torch.float32, device=dev) for _ in range(2)]
    fp = ql.default_frontend_params(seed=1)
    corr = torch.zeros((n_pts, 2), dtype=torch.int32, device=dev)
    Lout = C.c_int()

    def fe_once():
        f_ms = 0.0
        for c, d in zip(cl, desc):
            rc = hd._lib.qtr_fpfh(hd._h, 0, c.data_ptr(), n_pts, fp.normal_radius, fp.fpfh_radius, None, d.data_ptr(),
                                  ql.MEM_DEVICE)
            if rc != ql.QTR_OK:
                raise ql.QuatroHipError(rc, hd.last_error())
            f_ms += hd.stage_times()["fpfh"]
        rc = hd._lib.qtr_match(hd._h, 0, cl[0].data_ptr(), n_pts, desc[0].data_ptr(), cl[1].data_ptr(), n_pts,
                               desc[1].data_ptr(), C.byref(fp), corr.data_ptr(), n_pts, C.byref(Lout), ql.MEM_DEVICE)
        if rc != ql.QTR_OK:
            raise ql.QuatroHipError(rc, hd.last_error())
        return f_ms, hd.stage_times()
    fe_once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    nn_ms, nn_l, f_acc, m_acc = 0.0, 0, 0.0, 0.0
    for _ in range(reps):
        f_ms, st = fe_once()
        nn_ms += st["nn_kernel"]
        nn_l += st["nn_launches"]
        f_acc += f_ms
        m_acc += st["match"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["dense_frontend_leg"] = {
        "what": f"FPFH + reciprocal matching of two {n_pts}-point clouds, no voxel down-sampling",
        "ms_per_pair": 1e3 * el / reps, "fpfh_ms": f_acc / reps, "match_ms": m_acc / reps, "n_corr": int(Lout.value),
        "roofline": nn_roofline(float(n_pts) * n_pts, 1e-3 * nn_ms / max(nn_l, 1)) if nn_ms > 0 else None}
    hd.close()
    return out


def segment_leg(args, torch, ql, h, pool, dev):
    """next-row leg: range-image projection + sub-cluster rejection of the bench scans."""
    ipp = ql.ip_params()
    NP = ipp.n_scan * ipp.horizon_scan
    ov = torch.zeros((NP, 4), dtype=torch.float32, device=dev)
    oo = torch.zeros((NP, 4), dtype=torch.float32, device=dev)
    nv, no, nsg = C.c_int(), C.c_int(), C.c_int()

    def seg_once(t):
        rc = h._lib.qtr_segment_cloud(h._h, 0, t.data_ptr(), t.shape[0], C.byref(ipp), ov.data_ptr(), NP, C.byref(nv),
                                      oo.data_ptr(), NP, C.byref(no), C.byref(nsg), None, ql.MEM_DEVICE)
        if rc != ql.QTR_OK:
            raise ql.QuatroHipError(rc, h.last_error())
    for _ in range(3):
        seg_once(pool[0]["src"])
    torch.cuda.synchronize()
    ts0 = time.perf_counter()
    nscan, gpu_ms = 0, 0.0
    for k in range(10):
        p = pool[k % len(pool)]
        for t in (p["src"], p["tgt"]):
            seg_once(t)
            gpu_ms += h.stage_times()["total"]
            nscan += 1
    torch.cuda.synchronize()
    seg_once(pool[0]["src"])
    return {"what": "ImageProjection::segmentCloud (Velodyne-64-HDE, 4CrossNeighbor) on the bench scans",
            "scans_per_s": nscan / (time.perf_counter() - ts0), "gpu_ms_per_scan": gpu_ms / nscan,
            "points_in": int(pool[0]["src"].shape[0]), "valid_out": int(nv.value), "segments": int(nsg.value)}


def patchwork_leg(args, torch, ql, synth, h, dev):
    """next-row leg: Patchwork ground segmentation of raw 64-beam scans (with ground)."""
    raws = [synth.kitti64_raw_scan(i)[0] for i in range(2)]
    raw_d = [torch.from_numpy(r).to(dev) for r in raws]
    capp = max(r.shape[0] for r in raws)
    og = torch.zeros((capp, 4), dtype=torch.float32, device=dev)
    on = torch.zeros((capp, 4), dtype=torch.float32, device=dev)
    ngr, nng = C.c_int(), C.c_int()
    pwp = ql.pw_params()

    def pw_once(t):
        rc = h._lib.qtr_patchwork(h._h, 0, t.data_ptr(), t.shape[0], C.byref(pwp), og.data_ptr(), capp, C.byref(ngr),
                                  on.data_ptr(), capp, C.byref(nng), ql.MEM_DEVICE)
        if rc != ql.QTR_OK:
            raise ql.QuatroHipError(rc, h.last_error())
    for _ in range(3):
        pw_once(raw_d[0])
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    nscan, gpu_ms = 0, 0.0
    for k in range(20):
        pw_once(raw_d[k % len(raw_d)])
        gpu_ms += h.stage_times()["total"]
        nscan += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - tp0
    pw_once(raw_d[0])
    return {"what": "PatchWork::estimate_ground (config/patchwork_params.yaml) on synthetic raw 64-beam scans",
            "scans_per_s": nscan / el, "gpu_ms_per_scan": gpu_ms / nscan,
            "points_in": int(raws[0].shape[0]), "ground_out": int(ngr.value), "nonground_out": int(nng.value),
            "_raw0": raws[0]}


def cpu_baseline_leg(args, ql, h, p0, r0, yaw, value, seg, pwl, raw0):
    from oracle import oracle as qo  # cpu_baseline leg: the only place bench.py touches oracle/
    out = {}
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    phys = ncpu
    try:  # physical cores: distinct (package, core id) pairs
        ids = set()
        pk = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pk = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip()
                elif not line.strip():
                    if pk is not None and cid is not None:
                        ids.add((pk, cid))
                    pk = cid = None
        if ids:
            phys = min(len(ids), ncpu)
    except OSError:
        pass
    sweep = sorted(set(t for t in (4, 8, 16, 32, 64, phys) if 1 <= t <= ncpu))

    def cpu_once():
        return qo.register_pair(p0["src_h"], p0["tgt_h"], seed=p0["id"])

    qo.set_threads(min(16, ncpu))
    t1 = time.perf_counter()
    o = cpu_once()  # warm-up, also the parity reference
    first = time.perf_counter() - t1
    budget = max(args.cpu_seconds - first, 1.0)
    per_setting = budget / len(sweep)
    table = {}
    for th in sweep:
        qo.set_threads(th)
        ts = []
        t_start = time.perf_counter()
        while len(ts) < 5 and (not ts or time.perf_counter() - t_start + min(ts) < per_setting):
            t1 = time.perf_counter()
            cpu_once()
            ts.append(time.perf_counter() - t1)
        table[th] = float(np.median(ts))
    best_th = min(table, key=table.get)
    out["cpu_baseline"] = {
        "value": 1.0 / table[best_th], "unit": "registrations/s", "cores": best_th, "kind": "port",
        "omp4": (1.0 / table[4]) if 4 in table else None,
        "threads_swept": {str(k): round(1.0 / v, 3) for k, v in table.items()},
        "host_cpus": ncpu, "physical_cores": phys,
        "sample": f"pair {p0['id']} of the same workload (n = {o['n_src']}/{o['n_tgt']}): median of up to 5 runs of the "
                  f"OpenMP CPU oracle per thread count {sweep}; best = {best_th} threads ({table[best_th]:.3f} s). The "
                  "oracle's 33-D NN is brute force (the reference uses FLANN kd-trees, src/teaser_utils/"
                  "feature_matcher.cc:267-299) and its graph is a bit matrix"}
    yaw_o = float(np.arctan2(o["T"][1, 0], o["T"][0, 0]))
    out["parity_vs_oracle"] = {
        "rot_err_rad": abs(float(np.arctan2(np.sin(yaw - yaw_o), np.cos(yaw - yaw_o)))),
        "trans_err_m": float(np.linalg.norm(r0["T"][:3, 3] - o["T"][:3, 3])),
        "clique_bit_exact": bool(np.array_equal(r0["clique"], o["clique"])),
        "final_inliers_bit_exact": bool(np.array_equal(r0["final_inliers"], o["final_inliers"])),
        "counts_equal": bool((r0["n_src"], r0["n_tgt"], r0["L"]) == (o["n_src"], o["n_tgt"], o["L"]))}
    qo.set_threads(1)
    if seg is not None:  # the same scans through the oracle's breadth-first restatement, one thread (it is serial)
        t1 = time.perf_counter()
        so = qo.segment_cloud(p0["src_h"])
        cpu_s = time.perf_counter() - t1
        gs = h.segment_cloud(p0["src_h"])
        seg["cpu_port_scans_per_s"] = 1.0 / cpu_s
        seg["labels_bit_exact"] = bool(np.array_equal(gs["labels"], so["labels"]))
    if pwl is not None:  # the serial CPU restatement on one of the same scans
        t1 = time.perf_counter()
        po = qo.patchwork(raw0)
        cpu_s = time.perf_counter() - t1
        pg = h.patchwork(raw0)
        pwl["cpu_port_scans_per_s"] = 1.0 / cpu_s
        pwl["outputs_bit_exact"] = bool(np.array_equal(pg["ground"], po["ground"]) and
                                        np.array_equal(pg["nonground"], po["nonground"]))
    return out


if __name__ == "__main__":
    main()
