#!/usr/bin/env python
"""bench.py — registrations/sec of the whole hot path (voxel-FPFH -> matching -> consistency graph ->
max-clique -> GNC-TLS -> COTE) on synthetic KITTI-64-shaped scan pairs resident in HBM.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run, one rank per GPU.  A step = one registration of one scan pair (BASELINE.json
configs[1]); pairs are independent, so ranks shard pairs with no data-path collective and the only
exchange is the final gather of fixed-size result records over RCCL ("weak" scaling: per-GPU work
fixed).  Prints ONE JSON line on rank 0.

Extra objects in the line:
  roofline     — dominant kernel (33-D nearest-neighbour contraction): algorithmic FLOP per launch
                 (66 * n_small * n_large) / mean launch duration from HIP events recorded by the library on
                 the launch stream, against the FP32 matrix/vector peak of MI355X.
  cpu_baseline — the CPU oracle (a port; the reference cannot be built here) timed on this box's host cores
                 on a bounded sample (rank 0, N=1 only).  A reported baseline, not the target.
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

FP32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: FP32 vector = FP32 matrix peak
HBM_PEAK_GBS = 8000.0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="kitti64_pair", choices=["kitti64_pair", "solver5k"])
    ap.add_argument("--pairs", type=int, default=4, help="distinct synthetic pairs cycled through per rank")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--stream-slots", type=int, default=4,
                    help="extra (untimed-for-`value`) leg: the same K steps with this many pairs in flight on "
                         "independent stream slots of one GPU (BASELINE configs[2] style); 0 = skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    from quatro_amd import dist as qdist
    from quatro_amd import lib as ql
    from quatro_amd import synth

    h = ql.Handle(local_rank)
    prm = ql.demo_params()
    res = ql.Result()

    # ---- synthetic inputs, resident in HBM before the timed region
    pool = []
    for i in range(args.pairs):
        pid = rank * args.pairs + i
        if args.workload == "kitti64_pair":
            s, t, Tgt = synth.kitti64_pair(pid)
        else:
            s, t, Tgt, _ = synth.correspondences(5000, 0.05, seed=pid, noise=0.1)
        pool.append({"id": pid, "src_h": s, "tgt_h": t, "Tgt": Tgt, "src": torch.from_numpy(s).to(dev),
                     "tgt": torch.from_numpy(t).to(dev), "fp": ql.default_frontend_params(seed=pid)})
    torch.cuda.synchronize()

    def step(p):
        if args.workload == "kitti64_pair":
            rc = h.register_pair_dev(p["src"].data_ptr(), p["src"].shape[0], p["tgt"].data_ptr(), p["tgt"].shape[0],
                                     p["fp"], prm, res)
        else:
            rc = h.solve_dev(p["src"].data_ptr(), p["tgt"].data_ptr(), p["src"].shape[0], prm, res)
        if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
            raise ql.QuatroHipError(rc, h.last_error())

    for w in range(args.warmup):
        step(pool[w % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    nn_ms, nn_launches, stage_acc = 0.0, 0, {}
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(pool[k % len(pool)])
        st = h.stage_times()
        nn_ms += st["nn_kernel"]
        nn_launches += st["nn_launches"]
        for key, v in st.items():
            stage_acc[key] = stage_acc.get(key, 0.0) + float(v)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = qdist.max_over_ranks(elapsed, dev)

    # ---- extra leg: several pairs in flight per GPU (one host thread + one stream slot each).  Reported next to
    # `value`, never as `value`: configs[1] is the single-pair workload.
    multi = None
    if args.stream_slots > 1 and args.workload == "kitti64_pair":
        import threading
        S = args.stream_slots
        hm = ql.Handle(local_rank, n_slots=S)
        results = [ql.Result() for _ in range(S)]

        def worker(slot, count):
            for k in range(count):
                p = pool[(slot + k * S) % len(pool)]
                rc = hm.register_pair_dev(p["src"].data_ptr(), p["src"].shape[0], p["tgt"].data_ptr(), p["tgt"].shape[0],
                                          p["fp"], prm, results[slot], slot)
                if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
                    raise ql.QuatroHipError(rc, hm.last_error())

        def run(total):
            per = [(total + S - 1 - i) // S for i in range(S)]
            th = [threading.Thread(target=worker, args=(i, per[i])) for i in range(S)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()

        run(max(S, args.warmup))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tm0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tm = qdist.max_over_ranks(time.perf_counter() - tm0, dev)
        multi = {"stream_slots": S, "steps": args.steps, "value": world * args.steps / tm, "unit": "registrations/s",
                 "ms_per_step": 1e3 * tm / args.steps}
        hm.close()

    # ---- next-row leg: range-image projection + sub-cluster rejection of the same scans
    # (device-resident input, host copy of the two small outputs included); never part of `value`
    seg = None
    if world == 1 and args.workload == "kitti64_pair":
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
        for k in range(max(args.steps, 10)):
            p = pool[k % len(pool)]
            for t in (p["src"], p["tgt"]):
                seg_once(t)
                gpu_ms += h.stage_times()["total"]
                nscan += 1
        torch.cuda.synchronize()
        seg = {"what": "ImageProjection::segmentCloud (Velodyne-64-HDE, 4CrossNeighbor) on the bench scans",
               "scans_per_s": nscan / (time.perf_counter() - ts0), "gpu_ms_per_scan": gpu_ms / nscan,
               "points_in": int(pool[0]["src"].shape[0]), "valid_out": int(nv.value), "segments": int(nsg.value)}

    # ---- result records of the pool, gathered on rank 0 (the path's only collective)
    recs = []
    for p in pool:
        if args.workload == "kitti64_pair":
            r = h.register_pair(p["src_h"], p["tgt_h"], p["fp"], prm)
        else:
            r = h.solve(p["src_h"], p["tgt_h"], prm)
        p["result"] = r
        recs.append(qdist.pack_record(p["id"], r))
    gathered = qdist.gather_records(np.stack(recs), dev)

    # ---- next-row leg: Patchwork ground segmentation of raw 64-beam scans (with ground),
    # device-resident input and outputs; never part of `value`
    pwl = None
    if world == 1 and args.workload == "kitti64_pair":
        raws = [synth.kitti64_raw_scan(i)[0] for i in range(4)]
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
        for k in range(max(2 * args.steps, 20)):
            pw_once(raw_d[k % len(raw_d)])
            gpu_ms += h.stage_times()["total"]
            nscan += 1
        torch.cuda.synchronize()
        pw_once(raw_d[0])
        pwl = {"what": "PatchWork::estimate_ground (config/patchwork_params.yaml) on synthetic raw 64-beam scans",
               "scans_per_s": nscan / (time.perf_counter() - tp0), "gpu_ms_per_scan": gpu_ms / nscan,
               "points_in": int(raws[0].shape[0]), "ground_out": int(ngr.value), "nonground_out": int(nng.value)}
        pwl["_raw0"] = raws[0]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    p0 = pool[0]
    r0 = p0["result"]
    value = world * args.steps / elapsed
    out = {
        "metric": "scan-pair registrations/sec (KITTI 64-ch) + rot/trans err vs ref",
        "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (front end) / f64 (solver)", "data": "synthetic",
        "config": {
            "workload": ("synthetic KITTI-64-shaped single pair, voxel 0.3 m, whole path on GPU (BASELINE configs[1])"
                         if args.workload == "kitti64_pair" else "solver only, 5000 synthetic correspondences"),
            "raw_points": [int(p0["src_h"].shape[0]), int(p0["tgt_h"].shape[0])],
            "n_src": int(r0.get("n_src", 0)), "n_tgt": int(r0.get("n_tgt", 0)), "n_corr": int(r0["L"]),
            "n_clique": int(r0["clique"].size), "n_final_inliers": int(r0["final_inliers"].size),
            "pairs_per_rank": len(pool), "records_gathered": 0 if gathered is None else int(gathered.shape[0]),
            "parallelism": f"pairs sharded over {world} GPU(s), one process per GPU, RCCL gather of result records",
        },
        "stage_ms": {k: round(v / args.steps, 4) for k, v in stage_acc.items() if k not in ("nn_launches",)},
    }
    if multi is not None:
        out["pairs_in_flight_leg"] = multi
    if seg is not None:
        out["segment_cloud_leg"] = seg
    raw0 = pwl.pop("_raw0") if pwl is not None else None
    if pwl is not None:
        out["patchwork_leg"] = pwl
    if args.workload == "kitti64_pair":
        ms = h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)
        out["config"]["nn_rows_exact_recheck"] = [int(ms[8]), int(ms[9])]
        out["config"]["nn_rows_pair_compare"] = [int(ms[10]), int(ms[11])]
        out["config"]["n_cross_checked"] = int(ms[3])
    yaw_gt = float(np.arctan2(p0["Tgt"][1, 0], p0["Tgt"][0, 0]))
    yaw = float(np.arctan2(r0["T"][1, 0], r0["T"][0, 0]))
    out["accuracy_vs_ground_truth"] = {
        "rot_err_rad": abs(float(np.arctan2(np.sin(yaw - yaw_gt), np.cos(yaw - yaw_gt)))),
        "trans_err_m": float(np.linalg.norm(r0["T"][:3, 3] - p0["Tgt"][:3, 3])), "valid": bool(r0["valid"])}

    # ---- roofline of the dominant kernel
    if args.workload == "kitti64_pair" and nn_launches > 0:
        ns, nt = int(r0["n_src"]), int(r0["n_tgt"])
        flop_per_launch = 66.0 * ns * nt
        mean_launch_s = 1e-3 * nn_ms / nn_launches
        achieved = flop_per_launch / mean_launch_s / 1e12
        out["roofline"] = {"kernel": "k_nn (33-D reciprocal nearest neighbour, one launch per direction)",
                           "bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / FP32_PEAK_TFLOPS, "traffic": None,
                           "flop_per_launch": flop_per_launch, "mean_launch_ms": 1e3 * mean_launch_s}
        # HBM-side bytes per launch come from the PMC passes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 runs of
        # this same command); counters cannot be read in-process, so the committed summary is quoted
        import glob
        pmc = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_nn.json")))
        if pmc:
            with open(pmc[-1]) as f:
                pj = json.load(f)
            out["roofline"]["traffic"] = pj["traffic_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "bytes per launch (FETCH_SIZE + WRITE_SIZE)"
            out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(pmc[-1])
    else:
        L = int(r0["L"])
        gk = stage_acc.get("graph", 0.0) / max(args.steps, 1)
        alg_bytes = 32.0 * L + L * L / 8.0  # 2 x 16 B per correspondence in + bit matrix out
        achieved = alg_bytes / (1e-3 * gk) / 1e9 if gk > 0 else 0.0
        out["roofline"] = {"kernel": "k_graph_build (stage time)", "bound": "hbm", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None}

    # ---- CPU baseline: the oracle (port) on this box's host cores, bounded sample; also the parity check
    if world == 1 and args.cpu_seconds > 0:
        from oracle import oracle as qo  # cpu_baseline leg: the only place bench.py touches oracle/
        cores = os.cpu_count() or 1
        qo.set_threads(cores)

        def cpu_once():
            if args.workload == "kitti64_pair":
                return qo.register_pair(p0["src_h"], p0["tgt_h"], seed=p0["id"])
            return qo.solve(p0["src_h"], p0["tgt_h"])

        t1 = time.perf_counter()
        o = cpu_once()  # warm-up, also the parity reference
        first = time.perf_counter() - t1
        runs = int(max(1, min(5, args.cpu_seconds / max(first, 1e-3) - 1)))
        ts = []
        for _ in range(runs):
            t1 = time.perf_counter()
            cpu_once()
            ts.append(time.perf_counter() - t1)
        med = float(np.median(ts))
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "registrations/s", "cores": cores, "kind": "port",
                               "sample": f"pair {p0['id']} of the same workload: 1 warm-up + {runs} timed runs of the "
                                         f"OpenMP CPU oracle (median {med:.3f} s)"}
        yaw_o = float(np.arctan2(o["T"][1, 0], o["T"][0, 0]))
        out["parity_vs_oracle"] = {
            "rot_err_rad": abs(float(np.arctan2(np.sin(yaw - yaw_o), np.cos(yaw - yaw_o)))),
            "trans_err_m": float(np.linalg.norm(r0["T"][:3, 3] - o["T"][:3, 3])),
            "clique_bit_exact": bool(np.array_equal(r0["clique"], o["clique"])),
            "final_inliers_bit_exact": bool(np.array_equal(r0["final_inliers"], o["final_inliers"])),
            "counts_equal": bool(args.workload != "kitti64_pair" or
                                 (r0["n_src"], r0["n_tgt"], r0["L"]) == (o["n_src"], o["n_tgt"], o["L"]))}
        out["speedup_vs_cpu_baseline"] = value / (1.0 / med)
        if seg is not None:  # the same scans through the oracle's breadth-first restatement, one thread (it is serial)
            t1 = time.perf_counter()
            so = qo.segment_cloud(p0["src_h"])
            cpu_s = time.perf_counter() - t1
            gs = h.segment_cloud(p0["src_h"])
            out["segment_cloud_leg"]["cpu_port_scans_per_s"] = 1.0 / cpu_s
            out["segment_cloud_leg"]["labels_bit_exact"] = bool(np.array_equal(gs["labels"], so["labels"]))
        if pwl is not None:  # the serial CPU restatement on one of the same scans
            t1 = time.perf_counter()
            po = qo.patchwork(raw0)
            cpu_s = time.perf_counter() - t1
            pg = h.patchwork(raw0)
            out["patchwork_leg"]["cpu_port_scans_per_s"] = 1.0 / cpu_s
            out["patchwork_leg"]["outputs_bit_exact"] = bool(np.array_equal(pg["ground"], po["ground"]) and
                                                             np.array_equal(pg["nonground"], po["nonground"]))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
