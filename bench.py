#!/usr/bin/env python
"""bench.py — registrations/sec of the whole hot path (voxel-FPFH -> matching -> consistency graph ->
max-clique -> GNC-TLS -> COTE) on synthetic KITTI-64-shaped scan pairs resident in HBM.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run, one rank per GPU.  A step = one registration at the size BASELINE.json's metric is quoted on
(configs[1]: a single KITTI-64 pair, ~5k correspondences, whole path on the GPU), timed as ONE step and, since round 4,
issued as ONE call — qtr_register_pair_corr:
  1. the front end of a synthetic 64-beam scan pair (n ~ 16-18 k voxels per cloud at 0.3 m): voxel grid x2, FPFH x2,
     reciprocal 33-D matching with cross check and tuple test — the reference's voxelize + FPFHManager::setFeaturePair;
  2. the back end on 5000 synthetic correspondences with 5 % planted inliers — Quatro::computeTransformation at the
     metric's "~5k corr" — enqueued when the matcher's counters arrive and run after the front end.
The step is a COMPOSITE because FPFH matching on synthetic scans does not produce 5000 correspondences for any
physically plausible scene: the mutual-NN + tuple test keeps ~250-650 (DESIGN.md section 5 and
tests/probe/synth_L_probe.py list what was tried: baselines from 0 to 10 m, porous / solid clutter, near facades, range
noise down to 5 mm; only a jittered COPY of the same sweep gets there, and that is not a second scan; WITHOUT the tuple
test a finer voxel leaf does: `connected_l5k` below registers the front end's own L ~ 5 k).  So the back end
of the step is fed from the solver-only generator SURVEY.md section 8(d) defines for this configuration, not from the
matcher's output; `--workload pair` times qtr_register_pair on the scan pair alone (the `whole_pair_leg`), and the
`connected_leg` times registrations whose back end runs on thousands of the matcher's OWN correspondences.
Every rank times exactly K steps: the pair ids [0, N*K) are block-partitioned over the ranks
(quatro_amd.dist.shard_range, the partition of BASELINE configs[3]), rank r registers ids [r*K, (r+1)*K) one at a time —
per-GPU work is fixed as N grows ("weak" scaling) and `value` is the whole job, N*K registrations over the slowest
rank's time.  Pair id -> synthetic pair is id % pool.  Pairs are independent, so there is no data-path collective; the
only exchange is the gather of the fixed-size result records (RCCL) after the timed region plus the barrier /
max-over-ranks of the contract.  For N > 1 the line also carries `sharded_leg`: configs[3] itself — a FIXED set of 4096
pair ids, each the headline's unit of work (scan pair + 5000 given correspondences), block-partitioned over the ranks,
streamed through the batched entry points (strong scaling) and gathered through the LIBRARY's RCCL path
(qtr_comm_init / qtr_gather_results_v).  Prints ONE JSON line on rank 0.
Two things happen before the W warm-up steps and are named in `config`: the rank pins itself to the cores of its GPU's NUMA
node (`host_cpus`; --no-pin) and registers the pool for a quarter of a second (`settle`; --settle-seconds 0: the device's
clocks have to come back from the seconds of host-side set-up).  The timed region is the K steps and nothing else.

Objects in the line next to the contract's keys:
  roofline      — dominant kernel k_nn_f16: the 33-D distance matrix nb' - 2 a.b evaluated on the f16 matrix pipe with
                  every f32 operand split in two halves (3 x 33 products + 3 norm slots = 102 per matrix entry, see
                  match.hip).  `achieved` = 2 * 102 * n_query * n_base FLOP per launch (K padding to 112 not counted)
                  / mean launch duration (HIP events recorded by the library on the launch stream), `peak` = the dense
                  f16/bf16 MFMA peak.  `f32_equivalent` restates the same launches in SURVEY.md section 8(d)'s unit
                  (66 * n_query * n_base FLOP of an f32 evaluation) against the FP32 matrix peak.
                  `end_to_end`: the step's algorithmic FLOP and bytes (with L = 5000) priced at
                  the FP32-matrix / HBM peaks (`frac`) and on the f16 pipe the matcher actually uses
                  (`frac_on_f16_pipe`), over the measured time per step.
  cpu_baseline  — the CPU oracle (a port: the reference cannot be built here; brute-force NN instead of FLANN
                  kd-trees) running the SAME composite step on this box's host cores, swept over OMP thread counts on
                  a bounded sample; `value` is the best setting, `omp4` the reference README's 4-thread setting.  At
                  N > 1 too (rank 0's host, after the ranks have parted).
  cpu_reference_text — the reference's OWN back-end text (oracle/_ref/libref_solver.so) on the step's 5000
                  correspondences, one thread.  Reported baselines, not the target.
  parity_vs_oracle — every pool pair: front end (counts, correspondence list, keypoints), back end (clique, rotation /
                  final inliers, transform), the timed entry point's own record and the scan pair's whole path against
                  the oracle.
  connected_l5k — beside the composite headline: REAL registrations at the metric's L ~ 5 k, ONE qtr_register_pair call each, the
                  front end's OWN 4.2-5.9 k correspondences into the back end (the pool's scan pairs at a 0.07 m leaf, cross
                  check without tuple test; `dense18k`: two 18 000-point samplings of the structured scene, L = 4999), with
                  their own ms_per_registration (and per pair: the cliques are 400 - 2600 members), errors against the ground
                  truth, and — with --cpu-seconds > 0 — `oracle_equal`: pair 0's counts, clique, final inliers and transform
                  against the oracle's stages composed the same way, whose time is `cpu_port_registrations_per_s`.
  repeat_regions — the timed region four more times after the contract's one: how noisy the box is.
  whole_pair_leg, connected_leg, solver_L5000_leg, batch256_leg (composite pairs; `scan_pairs` = the scans alone),
  raw_batch_leg (raw sweeps through Patchwork + range image + the rest, batched), cpp_driver_leg (the step from a
  compiled caller), dense_*_leg, sharded_leg — the other configurations, never part of `value`.
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
L5K_LEAF = 0.07  # voxel leaf at which the pool's scan pairs yield 4.2-5.9 k mutual nearest neighbours (connected_leg.l5k)
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix (v_mfma_f32_32x32x2_f32) = FP32 vector peak
F16_PEAK_TFLOPS = 2500.0  # same guide: BF16/FP16 MFMA, dense (v_mfma_f32_32x32x16_f16: 32 cycles per SIMD)
NN_ENGINE = os.environ.get("QTR_NN_ENGINE", "f16")  # f16 (default): k_nn_f16; mfma32: k_nn_mfma; exact: no MFMA kernel
F16_FLOP_PER_ENTRY, F32_FLOP_PER_ENTRY = 204.0, 66.0


def nn_kernel_sha():
    """sha256[:16] over the nearest-neighbour kernel's text (the hand-scheduled loop, its generator, match.hip): what a
    committed counter profile was collected on (profiles/summarize_pmc.py records it) against what this run executes."""
    import hashlib
    hh = hashlib.sha256()
    for f in ("nn_f16_core.inc", "gen_nn_f16_core.py", "match.hip"):
        with open(os.path.join(ROOT, "quatro_amd", "csrc", f), "rb") as fh:
            hh.update(fh.read())
    return hh.hexdigest()[:16]


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
    ap.add_argument("--workload", default="composite", choices=("composite", "pair"),
                    help="composite (default): front end + matcher of the scan pair, back end on --corr planted "
                         "correspondences (the metric's '~5k corr'); pair: qtr_register_pair on the scan pair alone "
                         "(its matcher yields a few hundred correspondences)")
    ap.add_argument("--corr", type=int, default=5000, help="correspondences of the composite step's back end")
    ap.add_argument("--composite-calls", type=int, default=1, choices=(1, 2),
                    help="1 (default): the composite step is ONE call, qtr_register_pair_corr (front end of the scans, back end "
                         "on the given correspondences); 2: qtr_feature_pair then qtr_solve, as in rounds 2-3")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--nn-event-stride", type=int, default=5,
                    help="every n-th step of the timed region carries the nearest-neighbour launches' event pairs (1 = all)")
    ap.add_argument("--legs", default="pair,solver5k,batch,dense,connected,cpp,rawbatch,segment,patchwork",
                    help="comma list of the extra legs to run on rank 0 / all ranks (never part of `value`); "
                         "`refdense` adds the reference's own back-end text at L = 20000 on the CPU (minutes, > 16 GB)")
    ap.add_argument("--settle-seconds", type=float, default=0.25,
                    help="untimed registrations before the warm-up steps, until the device's clocks have settled (0 = none)")
    ap.add_argument("--no-pin", action="store_true", help="leave the process's CPU affinity alone (default: the GPU's NUMA-local cores)")
    ap.add_argument("--batch-pairs", type=int, default=256, help="pairs of the batch256 leg (BASELINE configs[2])")
    ap.add_argument("--sharded-pairs", type=int, default=4096, help="pair ids of the N>1 sharded leg (configs[3])")
    ap.add_argument("--batch-slots", type=int, default=32, help="stream slots of the batched legs (two lanes of half as many pairs)")
    args = ap.parse_args()
    legs = set(x for x in args.legs.split(",") if x)
    composite = args.workload == "composite"

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

    # one process per GPU, on the cores of the GPU's NUMA node (every hand-over of the path crosses PCIe twice)
    host_cpus = None if args.no_pin else qdist.pin_to_device_node(local_rank)

    LC = int(args.corr)
    h = ql.Handle(local_rank, max_points=131072, max_voxels=32768, max_corr=max(8192, LC + 64))
    prm = ql.demo_params()
    res = ql.Result()

    # ---- synthetic inputs (the same pool on every rank), resident in HBM before the timed region
    pool = []
    for pid in range(args.pool):
        s, t, Tgt = synth.kitti64_pair_16k(pid)
        cs, ct, Tc, inl = synth.correspondences(LC, 0.05, seed=pid, noise=0.1)
        pool.append({"id": pid, "src_h": s, "tgt_h": t, "Tgt": Tgt, "src": torch.from_numpy(s).to(dev),
                     "tgt": torch.from_numpy(t).to(dev), "fp": ql.default_frontend_params(seed=pid),
                     "cs_h": cs, "ct_h": ct, "Tc": Tc, "planted": inl,
                     "cs": torch.from_numpy(cs).to(dev), "ct": torch.from_numpy(ct).to(dev)})
    torch.cuda.synchronize()

    def step_pair(p, handle=h, r=res, slot=0):
        rc = handle.register_pair_dev(p["src"].data_ptr(), p["src"].shape[0], p["tgt"].data_ptr(), p["tgt"].shape[0],
                                      p["fp"], prm, r, slot)
        if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
            raise ql.QuatroHipError(rc, handle.last_error())

    for p in pool:  # (the device addresses once, not four torch calls per step)
        p["ptrs"] = (p["src"].data_ptr(), int(p["src"].shape[0]), p["tgt"].data_ptr(), int(p["tgt"].shape[0]),
                     p["cs"].data_ptr(), p["ct"].data_ptr())

    def step_composite_1(p, handle=h, r=res, slot=0):
        # voxelize x2 + FPFHManager::setFeaturePair on the scans, then Quatro::computeTransformation on the metric's ~5k
        # correspondences — one call through the C ABI, nothing between the two halves but the library's own hand-over
        sp, sn, tp, tn, cs, ct = p["ptrs"]
        rc = handle.register_pair_corr_dev(sp, sn, tp, tn, p["fp"], cs, ct, LC, prm, r, slot)
        if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
            raise ql.QuatroHipError(rc, handle.last_error())

    def step_composite_2(p, handle=h, r=res, slot=0):
        # FPFHManager::setFeaturePair on the scans (voxel grid, FPFH, reciprocal matching, tuple test) ...
        rc, p["_ns"], p["_nt"], p["_Lm"] = handle.feature_pair_dev(p["src"].data_ptr(), p["src"].shape[0],
                                                                   p["tgt"].data_ptr(), p["tgt"].shape[0], p["fp"], slot)
        if rc != ql.QTR_OK:
            raise ql.QuatroHipError(rc, handle.last_error())
        # ... Quatro::computeTransformation on the metric's ~5k correspondences
        rc = handle.solve_dev(p["cs"].data_ptr(), p["ct"].data_ptr(), LC, prm, r, slot)
        if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
            raise ql.QuatroHipError(rc, handle.last_error())

    step_composite = step_composite_1 if args.composite_calls == 1 else step_composite_2
    step = step_composite if composite else step_pair

    # sizes of every pool pair (one untimed registration each): needed for the per-launch FLOP accounting
    for p in pool:
        step_pair(p)
        p["n_src"], p["n_tgt"], p["L"], p["M"] = res.n_src, res.n_tgt, res.n_corr, res.n_clique
        p["n_hit"] = int(h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)[7])  # rows the second NN direction is asked for
        if composite:
            step_composite(p)
            p["Mc"] = res.n_clique
    lo, hi = qdist.shard_range(world * args.steps, rank, world)  # = [rank * K, (rank + 1) * K)
    # The device has idled through the seconds of host-side set-up above (synthetic scans, oracle library) and comes back from
    # its low-power state over tens of milliseconds: with the driver's 5 warm-up steps (2 ms) the contract's region — 20 steps,
    # 9 ms — read 1.5 - 2 % slower than each of its four repeats (0.464 against 0.455 - 0.458 ms on the final collection's boxes).
    # A fixed 0.25 s of untimed registrations lets the clocks settle BEFORE the W warm-up steps; the timed region is untouched.
    t_settle = time.perf_counter() + args.settle_seconds
    k_settle = 0
    while time.perf_counter() < t_settle:
        step(pool[k_settle % len(pool)])
        k_settle += 1
    for w in range(args.warmup):
        step(pool[w % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # the timed region is the registrations and nothing else: the per-stage events are switched off (stage_ms comes from
    # a short untimed pass below).  The roofline's kernel times come from event pairs attached to the two
    # nearest-neighbour launches, on their launch stream, and the library adds their elapsed times up (read once, after
    # the region).  A pair costs ~5 us of queue time on either side of its launch — four such gaps, 23 us, per step —
    # so every NN_STRIDE-th step of the region carries them, not all: the region stays the product path as a caller
    # without instrumentation runs it, and the average launch duration is still measured inside it (NN_STRIDE is odd so
    # that the timed steps rotate through the pool's pairs).
    NN_STRIDE = max(1, args.nn_event_stride)
    h.set_stage_events(False)
    h.set_nn_event_stride(NN_STRIDE)  # (also makes the region's first step a timed one)
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
    # the contract's region is the one above; the same region four more times shows how noisy the box is (extra key)
    repeats = []
    for _ in range(4):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for p in todo:
            step(p)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        repeats.append(qdist.max_over_ranks(time.perf_counter() - t1, cdev))
    h.set_nn_event_stride(1)
    nn_flop, alg_bytes, alg_flop = 0.0, 0.0, 0.0
    for k, p in enumerate(todo):
        # launch 1: every row of the smaller cloud against the larger one; launch 2: the hit rows of the larger cloud
        # against the smaller one  (of the steps whose launches carried events)
        if k % NN_STRIDE == 0:
            nn_flop += 66.0 * p["n_src"] * p["n_tgt"] + 66.0 * p["n_hit"] * min(p["n_src"], p["n_tgt"])
        b_, f_ = algorithmic_work(p["src"].shape[0], p["tgt"].shape[0], p["n_src"], p["n_tgt"],
                                  LC if composite else p["L"], p["Mc"] if composite else p["M"])
        alg_bytes += b_
        alg_flop += f_
    h.set_stage_events(True)
    stage_acc, stage_n = {}, 0
    for p in pool:  # untimed: where the time goes, stage by stage (events between the stages)
        if composite:
            h.feature_pair_dev(p["src"].data_ptr(), p["src"].shape[0], p["tgt"].data_ptr(), p["tgt"].shape[0], p["fp"])
            st = dict(h.stage_times())
            h.solve_dev(p["cs"].data_ptr(), p["ct"].data_ptr(), LC, prm, res)
            st2 = h.stage_times()
            for key in ("graph", "clique", "solve"):
                st[key] = st2[key]
            st["total"] = float(st["total"]) + float(st2["total"])
        else:
            step_pair(p)
            st = h.stage_times()
        for key, v in st.items():
            stage_acc[key] = stage_acc.get(key, 0.0) + float(v)
        stage_n += 1

    # ---- result records of the pool, gathered on rank 0 (the path's only collective)
    recs = []
    for p in pool:
        p["whole"] = h.register_pair(p["src_h"], p["tgt_h"], p["fp"], prm)  # the scan pair through the whole path
        if composite:
            p["front"] = h.feature_pair(p["src_h"], p["tgt_h"], p["fp"])
            p["result"] = h.solve(p["cs_h"], p["ct_h"], prm)
            p["result"]["L"] = LC
            p["one_call"] = h.register_pair_corr(p["src_h"], p["tgt_h"], p["cs_h"], p["ct_h"], p["fp"], prm)
        else:
            p["result"] = p["whole"]
        recs.append(qdist.pack_record(p["id"], p["result"]))
    gathered = qdist.gather_records(np.stack(recs), cdev)

    extra = {}
    # ---- the scan pair alone through qtr_register_pair (the matcher's own few hundred correspondences)
    if composite and "pair" in legs and rank == 0:
        n = max(min(args.steps, 40), 8)
        h.set_stage_events(False)  # (as in the headline's timed region: a production loop records no events)
        h.set_nn_event_stride(0)
        for k in range(4):
            step_pair(pool[k % len(pool)])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(n):
            step_pair(pool[k % len(pool)])
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        h.set_stage_events(True)
        h.set_nn_event_stride(1)
        extra["whole_pair_leg"] = {
            "what": "qtr_register_pair on the same scan pairs, one at a time: the whole path fed by its own matcher "
                    "(which keeps only a few hundred tuple-consistent correspondences on the synthetic scans)",
            "value": n / el, "unit": "registrations/s", "ms_per_step": 1e3 * el / n,
            "n_corr": [int(p["L"]) for p in pool]}
    # ---- BASELINE configs[2]: a batch of independent pairs streamed through one GPU
    if "batch" in legs and hasattr(h, "register_batch_dev"):
        extra["batch256_leg" if world == 1 else "sharded_leg"] = batch_leg(args, torch, ql, h, pool, prm, dev, world, dist,
                                                                         qdist, cdev, composite, LC)
    # ---- solver alone at the metric's "~5k corr"
    if "solver5k" in legs and rank == 0:
        extra["solver_L5000_leg"] = solver_leg(args, torch, ql, synth, h, prm, dev, 5000)
    if "dense" in legs and rank == 0 and world == 1:
        extra.update(dense_legs(args, torch, ql, synth, prm, dev, local_rank))
    if "rawbatch" in legs and rank == 0 and world == 1:
        extra["raw_batch_leg"] = raw_batch_leg(args, torch, ql, synth, prm, dev, local_rank)
    if "cpp" in legs and composite and rank == 0 and world == 1:
        extra["cpp_driver_leg"] = cpp_driver_leg(args, pool, LC)
    if "connected" in legs and rank == 0 and world == 1:
        extra["connected_leg"] = connected_leg(args, torch, ql, synth, pool, prm, dev, local_rank)
    seg = pwl = None
    if world == 1 and "segment" in legs:
        seg = segment_leg(args, torch, ql, h, pool, dev)
    if world == 1 and "patchwork" in legs:
        pwl = patchwork_leg(args, torch, ql, synth, h, dev)

    hung = any(isinstance(v, dict) and v.pop("_hung", False) for v in extra.values())
    # every GPU leg is done: the ranks part here (rank 0 goes on to the CPU baseline / parity legs on the host cores and
    # prints the line; nothing below is a collective)
    if world > 1 and not hung:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        if hung:
            os._exit(0)
        return

    p0 = pool[0]
    r0 = p0["result"]
    value = world * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    if composite:
        workload = (f"composite (BASELINE configs[1]: single KITTI-64 pair, ~5k correspondences): front end + matcher of a "
                    f"synthetic KITTI-64-shaped scan pair (quatro_amd.synth.kitti64_pair_16k, voxel 0.3 m, n ~ 16-18 k "
                    f"per cloud) through qtr_feature_pair [voxelize x2 + FPFHManager::setFeaturePair], then "
                    f"Quatro::computeTransformation [qtr_solve] on {LC} synthetic correspondences with 5 % planted "
                    f"inliers (quatro_amd.synth.correspondences, SURVEY 8(d) config 2) INSTEAD of the matcher's own "
                    f"output: FPFH on synthetic scans keeps only ~250-650 tuple-consistent correspondences for any "
                    f"physically plausible scene (DESIGN.md section 5; tests/probe/synth_L_probe.py), so the metric's "
                    f"'~5k corr' back end is fed from the solver-only generator; one step = " +
                    ("ONE call, qtr_register_pair_corr (the back end is enqueued when the matcher's counters arrive and runs "
                     "after the front end)" if args.composite_calls == 1 else "both calls, qtr_feature_pair then qtr_solve") +
                    ", one pair at a time, inputs resident in HBM")
    else:
        workload = ("synthetic KITTI-64-shaped single pair (quatro_amd.synth.kitti64_pair_16k), voxel 0.3 m, whole "
                    "path on GPU through qtr_register_pair, one registration at a time; NOTE n_corr is the matcher's "
                    "own output (a few hundred), not the metric's ~5k")
    out = {
        "metric": METRIC,
        "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (voxel grid, FPFH, 33-D matching: f16-split MFMA filter, exact f32 decision) / f64 (consistency graph, GNC-TLS, COTE)", "data": "synthetic",
        "config": {
            "workload": workload,
            "raw_points": [int(p0["src_h"].shape[0]), int(p0["tgt_h"].shape[0])],
            "n_src": int(p0["n_src"]), "n_tgt": int(p0["n_tgt"]), "n_corr": int(LC if composite else p0["L"]),
            "n_corr_matcher": int(p0["L"]),
            "n_clique": int(r0["clique"].size), "n_final_inliers": int(r0["final_inliers"].size),
            "pool": [{"id": p["id"], "n_src": int(p["n_src"]), "n_tgt": int(p["n_tgt"]),
                      "n_corr": int(LC if composite else p["L"]), "n_corr_matcher": int(p["L"]), "n_hit": int(p["n_hit"]),
                      "n_clique": int(p["result"]["clique"].size)}
                     for p in pool],
            "host_cpus": host_cpus or "unpinned",
            "settle": {"seconds": args.settle_seconds, "untimed_registrations_before_the_warmup": k_settle},
            "records_gathered": 0 if gathered is None else int(gathered.shape[0]),
            "parallelism": f"pair ids [0,{world * args.steps}) block-partitioned over {world} GPU(s) ({args.steps} per rank), one "
                           "process per GPU, RCCL "
                           "gather of result records",
        },
        "stage_ms": {k: round(v / max(stage_n, 1), 4) for k, v in stage_acc.items() if k not in ("nn_launches",)},
        "repeat_regions": {"what": "the timed region run four more times after the contract's one (same steps, same barriers)",
                           "ms_per_step": [round(1e3 * r / args.steps, 5) for r in repeats],
                           "min": round(1e3 * min(repeats) / args.steps, 5),
                           "median": round(1e3 * float(np.median(repeats)) / args.steps, 5),
                           "max": round(1e3 * max(repeats) / args.steps, 5)},
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

    def gt_err(T, Tgt):
        yaw_gt = float(np.arctan2(Tgt[1, 0], Tgt[0, 0]))
        yaw = float(np.arctan2(T[1, 0], T[0, 0]))
        return {"rot_err_rad": abs(float(np.arctan2(np.sin(yaw - yaw_gt), np.cos(yaw - yaw_gt)))),
                "trans_err_m": float(np.linalg.norm(T[:3, 3] - Tgt[:3, 3]))}
    out["accuracy_vs_ground_truth"] = {
        "solver_input": [dict(gt_err(p["result"]["T"], p["Tc"] if composite else p["Tgt"]), valid=bool(p["result"]["valid"]))
                         for p in pool],
        "scan_pair_whole_path": [dict(gt_err(p["whole"]["T"], p["Tgt"]), valid=bool(p["whole"]["valid"])) for p in pool]}

    # ---- roofline of the dominant kernel (rank 0's launches), and of the whole registration
    if nn_launches > 0:
        mean_launch_s = 1e-3 * nn_ms / nn_launches
        bound_ms = 1e3 * max(alg_flop / my_steps / (FP32_PEAK_TFLOPS * 1e12), alg_bytes / my_steps / (HBM_PEAK_GBS * 1e9))
        out["roofline"] = nn_roofline(nn_flop / F32_FLOP_PER_ENTRY / nn_launches, mean_launch_s)
        out["roofline"].update({
            "traffic": None, "launches_timed": nn_launches, "launches_in_region": 2 * my_steps,
            "timed_every": NN_STRIDE,
            "end_to_end": {
                "algorithmic_gflop_per_registration": alg_flop / my_steps / 1e9,
                "algorithmic_mbytes_per_registration": alg_bytes / my_steps / 1e6,
                "mfma_bound_ms": 1e3 * alg_flop / my_steps / (FP32_PEAK_TFLOPS * 1e12),
                "mfma_f16_bound_ms": 1e3 * (alg_flop / F32_FLOP_PER_ENTRY * F16_FLOP_PER_ENTRY) / my_steps / (F16_PEAK_TFLOPS * 1e12),
                "hbm_bound_ms": 1e3 * alg_bytes / my_steps / (HBM_PEAK_GBS * 1e9),
                "ms_per_step": ms_per_step, "frac": bound_ms / ms_per_step,
                "frac_on_f16_pipe": 1e3 * max((alg_flop / F32_FLOP_PER_ENTRY * F16_FLOP_PER_ENTRY) / my_steps / (F16_PEAK_TFLOPS * 1e12),
                                              alg_bytes / my_steps / (HBM_PEAK_GBS * 1e9)) / ms_per_step},
        })
        # HBM-side bytes per launch come from the PMC passes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 runs of
        # this same command); counters cannot be read in-process, so the committed summary is quoted
        import glob
        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc_nn.json")))
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
            # (a committed profile can go stale behind a kernel change: say which kernel text it was collected on)
            out["roofline"]["traffic_kernel_sha"] = pj.get("kernel_source_sha", "unrecorded")
            out["roofline"]["kernel_sha_now"] = nn_kernel_sha()
        # ... and so do the matrix pipe's own counters (SQ_VALU_MFMA_BUSY_CYCLES against the waves' lifetime)
        mf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc_mfma.json")))
        if mf and NN_ENGINE != "mfma32":
            with open(mf[-1]) as f:
                mj = json.load(f)
            if "k_nn_f16" in mj.get("kernel", ""):
                # NOT measured in this run (counters cannot be read in-process): a figure of the committed profile,
                # named as such, with the kernel text it was collected on beside the one this run executes
                out["roofline"]["mfma_busy_from_profile"] = mj["mfma_busy_of_wave_lifetime"]
                out["roofline"]["mfma_busy_source"] = "profiles/" + os.path.basename(mf[-1])
                out["roofline"]["mfma_busy_kernel_sha"] = mj.get("kernel_source_sha", "unrecorded")
                out["roofline"]["kernel_sha_now"] = nn_kernel_sha()

    # ---- CPU baseline: the oracle (port) on this box's host cores, bounded sample; also the parity check
    # (for N > 1 too: north_star wants the CPU path timed in the same run next to the multi-GPU numbers — rank 0's host)
    if args.cpu_seconds > 0:
        out.update(cpu_baseline_leg(args, ql, h, pool, composite, LC, value, seg, pwl, raw0, legs, extra))
    # ---- beside the composite headline: a REAL registration at the metric's L ~ 5 k — the front end's own 4.2-5.9 k
    # correspondences into the back end, one qtr_register_pair call (connected_leg.l5k), with its own ms_per_registration
    cl = out.get("connected_leg") or {}
    kept = cl.pop("_l5k", None)
    chk = out.pop("_l5k_check", None) or {}
    if "l5k" in cl:
        def brief(name):
            c = cl[name]
            d = {"ms_per_registration": c["ms_per_registration"], "value": c["value"], "unit": "registrations/s",
                 "frontend_params": c["frontend_params"], "n_voxels": [[r["n_src"], r["n_tgt"]] for r in c["pairs"]],
                 "n_corr": c["n_corr"], "n_clique": [r["n_clique"] for r in c["pairs"]],
                 "n_final": [r["n_final"] for r in c["pairs"]], "valid": [r["valid"] for r in c["pairs"]],
                 "rot_err_vs_gt_rad": [r["rot_err_vs_gt_rad"] for r in c["pairs"]],
                 "trans_err_vs_gt_m": [r["trans_err_vs_gt_m"] for r in c["pairs"]],
                 "ms_per_registration_by_pair": [r.get("ms_per_registration") for r in c["pairs"]]}
            if name in chk:
                k = chk[name]
                d["oracle_check_first_pair"] = k
                d["oracle_equal"] = bool(k["counts_equal"] and k["clique_bit_exact"] and k["final_inliers_bit_exact"] and
                                         k["rot_err_rad"] <= 1e-4 and k["trans_err_m"] <= 1e-3)
                d["cpu_port_registrations_per_s"] = 1.0 / k["cpu_port_seconds"]
            return d
        out["connected_l5k"] = dict(
            what="ONE qtr_register_pair call per registration, the matcher's OWN correspondences into the back end at the "
                 "metric's L ~ 5 k: the pool's 64-beam scan pairs at a %.2f m leaf, cross check without tuple test "
                 "(reference feature_matcher.cc:113-181; examples/run_global_registration.cpp:206-246)" % L5K_LEAF,
            **brief("l5k"))
        if "l5k_dense18k" in cl:
            out["connected_l5k"]["dense18k"] = brief("l5k_dense18k")
    print(json.dumps(out), flush=True)
    if hung:
        os._exit(0)


def batch_leg(args, torch, ql, h, pool, prm, dev, world, dist, qdist, cdev, composite, LC):
    """BASELINE configs[2] (and, for N > 1, configs[3]): B pair ids streamed through the batched entry points
    (qtr_submit_batch / qtr_wait), block-partitioned over the ranks.  With the composite workload every pair id is the
    headline's unit of work: the scan pair's front end AND the back end on the pair's --corr given correspondences
    (qtr_pair_desc.src_corr4 / tgt_corr4: the reference's loop hands Quatro whatever matched clouds its caller has,
    examples/run_global_registration.cpp:97-108,243-246).  `scan_pairs` repeats the leg on the scans alone (the
    matcher's own few hundred correspondences — the previous rounds' batch256 number).  For N > 1 the per-rank record
    blocks are gathered through the LIBRARY's RCCL path (qtr_comm_init / qtr_gather_results_v), torch's as fallback."""
    B = args.batch_pairs if world == 1 else args.sharded_pairs
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = qdist.shard_range(B, rank, world)
    hb = ql.Handle(torch.cuda.current_device(), max_points=131072, max_voxels=32768, max_corr=max(8192, LC + 64),
                   n_slots=args.batch_slots)
    ids = list(range(lo, hi))
    pairs = [pool[i % len(pool)] for i in ids]

    def run(corr):
        hb.register_batch_dev(pairs[:64], prm, corr=corr)  # warm-up
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        results = hb.register_batch_dev(pairs, prm, corr=corr)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0  # this rank's own block, before it waits for the others
        if world > 1:
            dist.barrier()
        el = qdist.max_over_ranks(time.perf_counter() - t0, cdev)
        per_rank.append(qdist.all_over_ranks(len(ids) / max(own, 1e-9), cdev))
        return el, results
    per_rank = []  # per run(): every rank's pairs/s on its own block (rank order)
    out = {"what": f"{B} pair ids, block-partitioned over {world} GPU(s), batched launch chains "
                   "(qtr_submit_batch / qtr_wait)", "pairs": B}
    if composite:
        el, results = run(True)
        # identical to the sequential calls of the headline's step: the solver's record on the same correspondences,
        # the front end's voxel counts
        same = all(bool(np.array_equal(r["T"], pool[i % len(pool)]["result"]["T"])) and r["L"] == LC and
                   r["n_clique"] == pool[i % len(pool)]["result"]["clique"].size and
                   r["n_final"] == pool[i % len(pool)]["result"]["final_inliers"].size and
                   (r["n_src"], r["n_tgt"]) == (pool[i % len(pool)]["front"]["n_src"], pool[i % len(pool)]["front"]["n_tgt"])
                   for i, r in zip(ids, results))
        out.update({"unit_of_work": f"front end of the scan pair (n ~ 16-18 k voxels per cloud) + back end on {LC} given "
                                    "correspondences per pair id — the headline step's unit, batched",
                    "n_corr": LC, "value": B / el, "unit": "registrations/s", "ms_per_pair": 1e3 * el / B,
                    "identical_to_sequential": same, "per_rank_pairs_per_s": per_rank[-1],
                    "pairs_per_rank": [qdist.shard_range(B, r_, world)[1] - qdist.shard_range(B, r_, world)[0]
                                       for r_ in range(world)]})
    el2, results2 = run(False)
    same2 = all(bool(np.array_equal(r["T"], pool[i % len(pool)]["whole"]["T"])) for i, r in zip(ids, results2))
    scan = {"what": "the same ids on the scans alone: the matcher's own correspondences feed the back end",
            "n_corr": [int(p["L"]) for p in pool], "value": B / el2, "unit": "registrations/s",
            "ms_per_pair": 1e3 * el2 / B, "identical_to_sequential": same2, "per_rank_pairs_per_s": per_rank[-1]}
    if composite:
        out["scan_pairs"] = scan
    else:
        out.update(scan)
    hung = False
    # the gather that closes configs[3]: through the library's own RCCL path.  (QTR_BENCH_ONE_DEVICE: real RCCL refuses
    # two ranks on one device, so the hook's ranks take this path only when QTR_RCCL_LIB names a transport double —
    # tests/test_gpu_multi.py — and the unique id then travels over gloo on host tensors.)
    if world > 1 and (cdev is not None or os.environ.get("QTR_RCCL_LIB")):
        g = {"path": "qtr_comm_init + qtr_gather_results_v (RCCL via the C ABI, dlopen)"}
        src_results = results if composite else results2
        udev = cdev if cdev is not None else torch.device("cpu")

        def lib_gather():
            try:
                t_init = time.perf_counter()
                uid = torch.zeros(128, dtype=torch.uint8, device=udev)
                if rank == 0:
                    uid = torch.frombuffer(bytearray(ql.comm_unique_id()), dtype=torch.uint8).to(udev)
                dist.broadcast(uid, 0)
                hb.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
                n_loc = len(ids)
                loc = (ql.Result * max(n_loc, 1))()
                for k, r in enumerate(src_results):
                    loc[k].status, loc[k].valid, loc[k].n_clique, loc[k].n_final = r["status"], int(r["valid"]), r["n_clique"], r["n_final"]
                    loc[k].n_corr, loc[k].n_src, loc[k].n_tgt, loc[k].cost = r["L"], r["n_src"], r["n_tgt"], r["cost"]
                    for q in range(16):
                        loc[k].T[q] = float(r["T"].reshape(-1)[q])
                t0 = time.perf_counter()
                allr, counts, n_all = hb.gather_results_v(loc if n_loc else None, n_loc, world, B)
                want = [qdist.shard_range(B, r_, world)[1] - qdist.shard_range(B, r_, world)[0] for r_ in range(world)]
                g.update({"ok": bool(n_all == B and counts == want), "records": int(n_all),
                          "seconds": time.perf_counter() - t0,  # the collective alone (counts + padded records)
                          "comm_init_seconds": t0 - t_init,  # unique id broadcast + ncclCommInitRank + packing the block
                          "bytes_gathered": int(n_all) * C.sizeof(ql.Result),
                          "all_valid": bool(all(allr[i].valid for i in range(n_all)))})
            except Exception as e:  # the bench line must survive a site whose RCCL cannot be opened from the library
                g.update({"ok": False, "error": repr(e)[:300]})
        import threading
        th = threading.Thread(target=lib_gather, daemon=True)
        th.start()
        th.join(timeout=120.0)
        if th.is_alive():  # (a rank that never arrives must not cost the line: report and leave without joining)
            g.update({"ok": False, "error": "timed out after 120 s"})
            hung = True
        out["gather"] = g
    if hung:
        out["_hung"] = True
        return out
    hb.close()
    return out


def connected_leg(args, torch, ql, synth, pool, prm, dev, device_index):
    """Data-connected registrations whose back end runs on THOUSANDS of the matcher's own correspondences — ONE
    qtr_register_pair call each, timed one at a time like the headline:
      mutual_nn   the 16-18 k-voxel pool pairs with use_tuple_test = 0 (reference feature_matcher.cc:187-247 skipped):
                  every mutual nearest-neighbour pair, L ~ 2 k
      no_cross    use_crosscheck = 0 as well (:124-181: corres_ij + corres_ji, de-duplicated): L ~ n_s + n_hit ~ 20 k
      dense       BASELINE configs[4] end to end: two INDEPENDENT 50 000-point samplings of one structured scene
                  (synth.dense_scene_pair), a leaf so small that the voxel grid would overflow int32 (pcl::VoxelGrid then
                  passes the cloud through — "no voxel downsample"), FPFH, matching with cross check and tuple test
                  (L ~ 2.1 k) and the back end: a registration that LANDS (clique ~160, centimetres from the truth)
      dense_mutual  the same with use_tuple_test = 0: every mutual nearest-neighbour pair of the 50 k x 50 k search goes
                  into the back end — L ~ 14 k of the matcher's own correspondences (configs[4]'s "~20 k corr")
      l5k         THE METRIC'S SIZE, DATA-CONNECTED: the pool's 64-beam scan pairs at a 0.07 m leaf (n ~ 35-40 k voxels per
                  cloud) with use_tuple_test = 0 — the front end's OWN output is L = 4.2-5.9 k mutual nearest neighbours
                  (reference feature_matcher.cc:113-181), and the back end registers them: cliques of 400-2600, centimetres
                  from the truth
      l5k_dense18k  two independent 18 000-point samplings of the structured scene, no voxel step: n = 18 000 per cloud
                  (the headline's n) and L = 4999 of the matcher's own
    (parity of each against the oracle: tests/test_gpu_baseline_sizes.py; of l5k pair 0 and l5k_dense18k also in this run:
    `connected_l5k.oracle_equal`)."""
    hc = ql.Handle(device_index, max_points=131072, max_voxels=65536, max_corr=32768)
    res = ql.Result()
    out = {}
    a, b, Td = synth.dense_scene_pair(50000)
    dense = {"src": torch.from_numpy(a).to(dev), "tgt": torch.from_numpy(b).to(dev), "Tgt": Td}
    a18, b18, T18 = synth.dense_scene_pair(18000)
    dense18 = {"src": torch.from_numpy(a18).to(dev), "tgt": torch.from_numpy(b18).to(dev), "Tgt": T18, "src_h": a18, "tgt_h": b18}
    cases = [("mutual_nn", dict(use_tuple_test=0), pool, 12),
             ("no_cross", dict(use_crosscheck=0, use_tuple_test=0), pool, 8),
             ("dense", dict(voxel_size=0.001), [dense], 6),
             ("dense_mutual", dict(voxel_size=0.001, use_tuple_test=0), [dense], 6),
             ("l5k", dict(voxel_size=L5K_LEAF, use_tuple_test=0), pool, 12),
             ("l5k_dense18k", dict(voxel_size=0.001, use_tuple_test=0), [dense18], 8)]
    keep = []  # (name, item, frontend params, the registration's full record) of the L ~ 5 k cases: the CPU leg checks them
    hc.set_stage_events(False)
    hc.set_nn_event_stride(0)
    for name, kw, items, n in cases:
        fps = [ql.default_frontend_params(seed=k, **kw) for k in range(len(items))]
        recs = []
        for k, it in enumerate(items):  # warm-up (also switches the handle to long neighbour lists for the dense clouds)
            rc = hc.register_pair_dev(it["src"].data_ptr(), it["src"].shape[0], it["tgt"].data_ptr(), it["tgt"].shape[0],
                                      fps[k], prm, res)
            if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
                raise ql.QuatroHipError(rc, hc.last_error())
            T = np.array(res.T[:]).reshape(4, 4)
            yaw_gt = float(np.arctan2(it["Tgt"][1, 0], it["Tgt"][0, 0]))
            yaw = float(np.arctan2(T[1, 0], T[0, 0]))
            recs.append({"n_src": int(res.n_src), "n_tgt": int(res.n_tgt), "n_corr": int(res.n_corr),
                         "n_clique": int(res.n_clique), "n_final": int(res.n_final), "valid": bool(res.valid),
                         "rot_err_vs_gt_rad": abs(float(np.arctan2(np.sin(yaw - yaw_gt), np.cos(yaw - yaw_gt)))),
                         "trans_err_vs_gt_m": float(np.linalg.norm(T[:3, 3] - it["Tgt"][:3, 3]))})
        torch.cuda.synchronize()
        per_item = [[] for _ in items]  # (a call returns with its result: the host clock around it is the registration)
        t0 = time.perf_counter()
        for k in range(n):
            it = items[k % len(items)]
            t1 = time.perf_counter()
            hc.register_pair_dev(it["src"].data_ptr(), it["src"].shape[0], it["tgt"].data_ptr(), it["tgt"].shape[0],
                                 fps[k % len(items)], prm, res)
            per_item[k % len(items)].append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        for r, ts in zip(recs, per_item):
            r["ms_per_registration"] = 1e3 * float(np.mean(ts)) if ts else None
        out[name] = {"frontend_params": kw, "value": n / el, "unit": "registrations/s", "ms_per_registration": 1e3 * el / n,
                     "n_corr": [r["n_corr"] for r in recs], "pairs": recs}
        if name.startswith("l5k"):  # the first item's lists (host call: clique, final inliers, transform) for the oracle check
            keep.append((name, items[0], fps[0], hc.register_pair(items[0]["src_h"], items[0]["tgt_h"], fps[0], prm)))
    out["_l5k"] = keep
    hc.close()
    out["what"] = ("qtr_register_pair, one call per registration, the matcher's own correspondences into the back end "
                   "(no generator in between)")
    return out


def raw_batch_leg(args, torch, ql, synth, prm, dev, device_index):
    """The demo's WHOLE sequence per pair (examples/run_global_registration.cpp:136-160,206-246) batched: raw 64-beam sweeps
    with their ground returns -> PatchWork::estimate_ground -> ImageProjection::segmentCloud -> voxel grid -> FPFH ->
    matching -> Quatro, through qtr_set_batch_preprocess + qtr_submit_batch.  The two raw-sweep stages run for all pairs
    of a chunk side by side (one slot each), the rest as launch chains over the group."""
    scans = [torch.from_numpy(synth.kitti64_raw_scan(i)[0]).to(dev) for i in range(4)]
    B = 64
    items = [{"src": scans[i % 4], "tgt": scans[(i + 1) % 4], "fp": ql.default_frontend_params(seed=i % 7)} for i in range(B)]
    hb = ql.Handle(device_index, max_points=131072, max_voxels=32768, max_corr=8192, n_slots=args.batch_slots)
    try:
        hb.set_batch_preprocess()
        hb.register_batch_dev(items[:32], prm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = hb.register_batch_dev(items, prm)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    finally:
        hb.close()
    return {"what": f"{B} raw-sweep pairs ({int(scans[0].shape[0])} points per sweep, ground included) through "
                    "qtr_set_batch_preprocess + qtr_submit_batch", "pairs": B, "value": B / el, "unit": "registrations/s",
            "ms_per_pair": 1e3 * el / B, "valid": int(sum(1 for r in res if r["valid"])),
            "n_src_after_preprocessing_and_voxel_grid": int(res[0]["n_src"])}


def cpp_driver_leg(args, pool, LC):
    """The headline's composite step driven by a compiled caller (tests/bench_cpp/bench_step.cpp, built here with hipcc):
    the same two C-ABI calls per step without the Python harness in between."""
    import shutil
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="qtr_cpp_")
    try:
        for k, p in enumerate(pool):
            for name, arr in (("src", p["src_h"]), ("tgt", p["tgt_h"]), ("cs", p["cs_h"]), ("ct", p["ct_h"])):
                np.ascontiguousarray(arr, dtype=np.float32).tofile(os.path.join(d, f"pair{k}_{name}.bin"))
        exe = os.path.join(d, "bench_step")
        libdir = os.path.join(ROOT, "quatro_amd")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "--offload-arch=gfx950",
                               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "bench_cpp", "bench_step.cpp"),
                               "-o", exe, "-L", libdir, "-lquatro_hip", "-Wl,-rpath," + libdir],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        n = max(args.steps, 40)
        best = None
        for _ in range(3):  # (three processes: the first pays the image's cold start)
            o = subprocess.run([exe, d, str(len(pool)), str(n), str(max(args.warmup, 8))], capture_output=True, text=True,
                               timeout=300)
            if o.returncode != 0:
                return {"error": (o.stderr or o.stdout)[-300:]}
            j = json.loads(o.stdout.strip().splitlines()[-1])
            if best is None or j["ms_per_step"] < best["ms_per_step"]:
                best = j
        best["runs"] = 3
        return best
    except Exception as e:  # a box without hipcc: the leg is informative only
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


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


def nn_roofline_two(n_rows1, n_rows2, n_cols1, n_cols2, ms_dir1, ms_dir2):
    """roofline of the two nearest-neighbour launches of ONE match, each priced with ITS OWN rows: direction 1 evaluates
    n_rows1 x n_cols1 entries (every row of the smaller cloud against the larger one), direction 2 n_rows2 x n_cols2 (only
    the rows of the larger cloud some row of the smaller one chose, feature_matcher.cc:113-122).  The leading fields are
    the two launches together (entries of both / time of both)."""
    e1, e2 = float(n_rows1) * n_cols1, float(n_rows2) * n_cols2
    r = nn_roofline((e1 + e2) / 2.0, 1e-3 * (ms_dir1 + ms_dir2) / 2.0)
    d1, d2 = nn_roofline(e1, 1e-3 * ms_dir1), nn_roofline(e2, 1e-3 * ms_dir2)
    r["accounting"] = "each launch priced with its own rows x columns; leading fields = both launches together"
    r["direction1"] = {"rows": int(n_rows1), "cols": int(n_cols1), "launch_ms": ms_dir1, "achieved": d1["achieved"], "frac": d1["frac"]}
    r["direction2"] = {"rows": int(n_rows2), "cols": int(n_cols2), "launch_ms": ms_dir2, "achieved": d2["achieved"], "frac": d2["frac"]}
    return r


def dense_legs(args, torch, ql, synth, prm, dev, device_index):
    """BASELINE configs[4]: dense mode — 50 k-point clouds, no voxel down-sampling, ~20 k correspondences.
      dense_step_leg      configs[4] as ONE step and ONE call (qtr_register_pair_corr): the front end of two 50 000-point
                          scans of one structured scene (synth.dense_scene_pair; the voxel grid passes them through, as
                          pcl::VoxelGrid does when the leaf would overflow its index) + the back end on 20 000 given
                          correspondences (2 % planted) — the dense analogue of the headline's composite step
      dense_solver_leg    the back end alone at L = 20 000 (50 MB bit matrix)
      dense_frontend_leg  FPFH + matching alone on the same clouds (qtr_fpfh x2 + qtr_match)
    (the data-CONNECTED dense registrations — the matcher's own 2 k / 14 k correspondences into the back end — are
    connected_leg.dense / dense_mutual)."""
    out = {}
    hd = ql.Handle(device_index, max_points=65536, max_voxels=65536, max_corr=24576)
    res = ql.Result()
    L = 20000
    s, t, Tc, planted = synth.correspondences(L, 0.02, seed=7, noise=0.1)
    sd, td = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
    n_pts = 50000
    a, b, Td = synth.dense_scene_pair(n_pts)
    cl = [torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)]
    fpd = ql.default_frontend_params(voxel_size=0.001, seed=1)

    # ---- configs[4] as one step
    def dstep():
        rc = hd.register_pair_corr_dev(cl[0].data_ptr(), n_pts, cl[1].data_ptr(), n_pts, fpd, sd.data_ptr(), td.data_ptr(), L,
                                       prm, res)
        if rc not in (ql.QTR_OK, ql.QTR_ERR_CLIQUE_TOO_SMALL):
            raise ql.QuatroHipError(rc, hd.last_error())
    hd.set_stage_events(True)
    hd.set_nn_event_stride(1)
    dstep()   # (the first call on a handle also sizes its lazily allocated arenas)
    dstep()
    st = dict(hd.stage_times())
    n_hit = int(hd.debug_fetch(ql.DBG_MATCH_STATS, np.int32)[7])
    T = np.array(res.T[:]).reshape(4, 4)
    yaw_gt, yaw = float(np.arctan2(Tc[1, 0], Tc[0, 0])), float(np.arctan2(T[1, 0], T[0, 0]))
    rec = {"n_src": int(res.n_src), "n_tgt": int(res.n_tgt), "n_corr": int(res.n_corr), "n_hit": n_hit,
           "n_clique": int(res.n_clique), "n_final": int(res.n_final), "valid": bool(res.valid),
           "rot_err_vs_gt_rad": abs(float(np.arctan2(np.sin(yaw - yaw_gt), np.cos(yaw - yaw_gt)))),
           "trans_err_vs_gt_m": float(np.linalg.norm(T[:3, 3] - Tc[:3, 3]))}
    hd.set_stage_events(False)
    hd.set_nn_event_stride(0)
    dstep()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        dstep()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms_step = 1e3 * el / reps
    # SURVEY.md section 8(d) counts the 33-D distance matrix ONCE (66 n_s n_t, as the headline's end_to_end does); the second
    # direction's rows (n_hit x n_s: not small change at 50 k) are reported beside it, never inside `frac` (round 5 added
    # them in and read 0.63 where the survey's unit gives 0.42)
    ab, af = algorithmic_work(n_pts, n_pts, rec["n_src"], rec["n_tgt"], L, rec["n_clique"])
    af_both = af + 66.0 * n_hit * min(rec["n_src"], rec["n_tgt"])
    roof = nn_roofline_two(min(rec["n_src"], rec["n_tgt"]), n_hit, max(rec["n_src"], rec["n_tgt"]),
                           min(rec["n_src"], rec["n_tgt"]), st["nn_dir1"], st["nn_dir2"]) if st["nn_dir1"] > 0 else None
    f16_ms = 1e3 * (af / F32_FLOP_PER_ENTRY * F16_FLOP_PER_ENTRY) / (F16_PEAK_TFLOPS * 1e12)
    hbm_ms = 1e3 * ab / (HBM_PEAK_GBS * 1e9)
    if roof is not None:
        roof["end_to_end"] = {"algorithmic_gflop_per_registration": af / 1e9, "algorithmic_mbytes_per_registration": ab / 1e6,
                              "mfma_bound_ms": 1e3 * af / (FP32_PEAK_TFLOPS * 1e12), "mfma_f16_bound_ms": f16_ms,
                              "hbm_bound_ms": hbm_ms, "ms_per_step": ms_step,
                              "frac": max(1e3 * af / (FP32_PEAK_TFLOPS * 1e12), hbm_ms) / ms_step,
                              "frac_on_f16_pipe": max(f16_ms, hbm_ms) / ms_step,
                              "unit_note": "SURVEY 8(d): 66 n_s n_t FLOP, the distance matrix once",
                              "gflop_with_second_direction": af_both / 1e9,
                              "frac_with_second_direction": max(1e3 * af_both / (FP32_PEAK_TFLOPS * 1e12), hbm_ms) / ms_step}
    out["dense_step_leg"] = {
        "what": f"BASELINE configs[4] as ONE call (qtr_register_pair_corr): front end of two {n_pts}-point scans (no voxel "
                f"down-sampling) + back end on {L} given correspondences",
        "value": reps / el, "unit": "registrations/s", "ms_per_step": ms_step, "steps": reps, "record": rec,
        "stage_ms": {k: round(float(v), 4) for k, v in st.items() if k != "nn_launches"}, "roofline": roof}

    # ---- the back end alone
    hd.set_stage_events(True)
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
                     "frac": (gb / gk / 1e9 / HBM_PEAK_GBS) if gk > 0 else 0.0,
                     "note": "the kernel is bound by its vector-instruction count, not by HBM (profiles/r*_graph_sq.txt): "
                             "L^2/2 predicates per launch is the unit that matters"}}
    # ---- the front end alone at 50 k points per cloud, no voxel grid: qtr_fpfh + qtr_match on device-resident arrays
    desc = [torch.zeros((n_pts, 33), dtype=torch.float32, device=dev) for _ in range(2)]
    fp = ql.default_frontend_params(seed=1)
    corr = torch.zeros((n_pts, 2), dtype=torch.int32, device=dev)
    Lout = C.c_int()
    hd.set_nn_event_stride(1)

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
    d1_ms, d2_ms, f_acc, m_acc = 0.0, 0.0, 0.0, 0.0
    for _ in range(reps):
        f_ms, st = fe_once()
        d1_ms += st["nn_dir1"]
        d2_ms += st["nn_dir2"]
        f_acc += f_ms
        m_acc += st["match"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n_hit = int(hd.debug_fetch(ql.DBG_MATCH_STATS, np.int32)[7])
    out["dense_frontend_leg"] = {
        "what": f"FPFH + reciprocal matching of two {n_pts}-point clouds, no voxel down-sampling",
        "ms_per_pair": 1e3 * el / reps, "fpfh_ms": f_acc / reps, "match_ms": m_acc / reps, "n_corr": int(Lout.value),
        "n_hit": n_hit,
        "roofline": nn_roofline_two(n_pts, n_hit, n_pts, n_pts, d1_ms / reps, d2_ms / reps) if d1_ms > 0 else None}
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


def host_cores():
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
    return ncpu, phys


def cpu_baseline_leg(args, ql, h, pool, composite, LC, value, seg, pwl, raw0, legs, extra):
    from oracle import oracle as qo  # cpu_baseline leg: the only place bench.py touches oracle/
    out = {}
    ncpu, phys = host_cores()
    sweep = sorted(set(t for t in (4, 8, 16, 32, 64, phys) if 1 <= t <= ncpu))
    p0 = pool[0]
    fp0 = p0["fp"]

    def cpu_front(p):
        """voxelize x2 + FPFH x2 + matching through the oracle's stage functions: (vs, vt, corr)"""
        f = p["fp"]
        vs, vt = qo.voxelize(p["src_h"], f.voxel_size), qo.voxelize(p["tgt_h"], f.voxel_size)
        ds = qo.fpfh(vs, f.normal_radius, f.fpfh_radius)[2]
        dt = qo.fpfh(vt, f.normal_radius, f.fpfh_radius)[2]
        corr = qo.match(vs, ds, vt, dt, bool(f.use_crosscheck), bool(f.use_tuple_test), f.tuple_scale, int(f.seed))
        return vs, vt, corr

    def cpu_once():
        if composite:
            cpu_front(p0)
            return qo.solve(p0["cs_h"], p0["ct_h"])
        return qo.register_pair(p0["src_h"], p0["tgt_h"], seed=p0["id"])

    qo.set_threads(min(16, ncpu))
    t1 = time.perf_counter()
    cpu_once()  # warm-up
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
    what = (f"front end + matcher of pair {p0['id']} (n = {p0['n_src']}/{p0['n_tgt']}) + back end on the same {LC} planted "
            "correspondences" if composite else f"pair {p0['id']} of the same workload (n = {p0['n_src']}/{p0['n_tgt']})")
    out["cpu_baseline"] = {
        "value": 1.0 / table[best_th], "unit": "registrations/s", "cores": best_th, "kind": "port",
        "omp4": (1.0 / table[4]) if 4 in table else None,
        "threads_swept": {str(k): round(1.0 / v, 3) for k, v in table.items()},
        "host_cpus": ncpu, "physical_cores": phys,
        "sample": f"{what}: median of up to 5 runs of the OpenMP CPU oracle per thread count {sweep}; best = {best_th} "
                  f"threads ({table[best_th]:.3f} s). The oracle's 33-D NN is brute force (the reference uses FLANN "
                  "kd-trees, src/teaser_utils/feature_matcher.cc:267-299) and its graph is a bit matrix; the reference's "
                  "own back-end text is timed separately (cpu_reference_text)"}

    # ---- parity of EVERY pool pair against the oracle (bit-exact integer outputs, transform within 1e-4 rad / 1e-3 m)
    qo.set_threads(best_th)

    def t_err(Ta, Tb):
        ya, yb = float(np.arctan2(Ta[1, 0], Ta[0, 0])), float(np.arctan2(Tb[1, 0], Tb[0, 0]))
        return abs(float(np.arctan2(np.sin(ya - yb), np.cos(ya - yb)))), float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
    par = []
    for p in pool:
        e = {"id": p["id"]}
        o = qo.register_pair(p["src_h"], p["tgt_h"], seed=p["id"])
        w = p["whole"]
        re_, te_ = t_err(w["T"], o["T"])
        e["whole_path"] = {
            "rot_err_rad": re_, "trans_err_m": te_,
            "clique_bit_exact": bool(np.array_equal(w["clique"], o["clique"])),
            "final_inliers_bit_exact": bool(np.array_equal(w["final_inliers"], o["final_inliers"])),
            "counts_equal": bool((w["n_src"], w["n_tgt"], w["L"]) == (o["n_src"], o["n_tgt"], o["L"]))}
        if composite:
            vs, vt, corr = cpu_front(p)
            f = p["front"]
            e["front_end"] = {
                "counts_equal": bool((f["n_src"], f["n_tgt"], f["L"]) == (vs.shape[0], vt.shape[0], corr.shape[0])),
                "correspondences_bit_exact": bool(np.array_equal(f["corr"], corr)),
                "keypoints_bit_exact": bool(corr.shape[0] == f["L"] and
                                            np.array_equal(f["src_kps"][:, :3], vs[corr[:, 0], :3]) and
                                            np.array_equal(f["tgt_kps"][:, :3], vt[corr[:, 1], :3]))}
            so = qo.solve(p["cs_h"], p["ct_h"])
            r = p["result"]
            re_, te_ = t_err(r["T"], so["T"])
            e["back_end"] = {
                "rot_err_rad": re_, "trans_err_m": te_,
                "clique_bit_exact": bool(np.array_equal(r["clique"], so["clique"])),
                "rot_inliers_bit_exact": bool(np.array_equal(r["rot_inliers"], so["rot_inliers"])),
                "final_inliers_bit_exact": bool(np.array_equal(r["final_inliers"], so["final_inliers"])),
                "planted_inliers_recovered": int(np.intersect1d(r["final_inliers"], p["planted"]).size),
                "planted": int(p["planted"].size)}
            oc = p["one_call"]  # the timed step's own entry point: the same record as the two stage calls
            e["one_call"] = {
                "counts_equal": bool((oc["n_src"], oc["n_tgt"], oc["L"], oc["n_matched"]) == (f["n_src"], f["n_tgt"], LC, f["L"])),
                "clique_bit_exact": bool(np.array_equal(oc["clique"], so["clique"])),
                "final_inliers_bit_exact": bool(np.array_equal(oc["final_inliers"], so["final_inliers"])),
                "transform_bit_exact": bool(np.array_equal(oc["T"], r["T"]))}
        par.append(e)

    def all_ok(e):
        ok = True
        for part in ("whole_path", "front_end", "back_end", "one_call"):
            if part in e:
                d = e[part]
                ok = ok and all(v for k, v in d.items() if k.endswith("_exact") or k.endswith("_equal"))
                ok = ok and d.get("rot_err_rad", 0.0) <= 1e-4 and d.get("trans_err_m", 0.0) <= 1e-3
        return bool(ok)
    out["parity_vs_oracle"] = {"all_pool_pairs_ok": all(all_ok(e) for e in par), "pairs": par,
                               "tolerance": "integer outputs bit-exact; 1e-4 rad / 1e-3 m"}

    # ---- the data-connected registrations at the metric's L ~ 5 k (connected_leg.l5k*): the oracle's stages composed the
    # same way on the same scans — every output equal — and timed: the CPU port's rate on a REAL registration of that size
    l5k = (extra.get("connected_leg") or {}).get("_l5k") or []
    chk = {}
    for name, it, f, g in l5k:
        t1 = time.perf_counter()
        vs, vt = qo.voxelize(it["src_h"], f.voxel_size), qo.voxelize(it["tgt_h"], f.voxel_size)
        ds, dt = qo.fpfh(vs, f.normal_radius, f.fpfh_radius)[2], qo.fpfh(vt, f.normal_radius, f.fpfh_radius)[2]
        corr = qo.match(vs, ds, vt, dt, bool(f.use_crosscheck), bool(f.use_tuple_test), f.tuple_scale, int(f.seed))
        o = qo.solve(vs[corr[:, 0]], vt[corr[:, 1]])
        cpu_s = time.perf_counter() - t1
        re_, te_ = t_err(g["T"], o["T"])
        chk[name] = {
            "counts_equal": bool((g["n_src"], g["n_tgt"], g["L"]) == (vs.shape[0], vt.shape[0], corr.shape[0])),
            "clique_bit_exact": bool(np.array_equal(g["clique"], o["clique"])),
            "final_inliers_bit_exact": bool(np.array_equal(g["final_inliers"], o["final_inliers"])),
            "transform_bit_exact": bool(np.array_equal(g["T"], o["T"])), "rot_err_rad": re_, "trans_err_m": te_,
            "n_corr": int(corr.shape[0]), "n_clique": int(o["clique"].size), "n_final": int(o["final_inliers"].size),
            "cpu_port_seconds": cpu_s, "cpu_port_threads": best_th}
    if chk:
        out["_l5k_check"] = chk

    # ---- the reference's OWN back-end text (oracle/_ref/libref_solver.so: computeTIMs with materialised TIMs,
    # solveForScale, teaser::Graph adjacency lists with find-before-insert, solveForRotation2D, COTE — cut out of
    # include/quatro.hpp at build time) timed on this box's host cores; the clique search inside it is the oracle's
    if qo.ref_solver_available():
        def ref_time(cs, ct, L):
            t1 = time.perf_counter()
            rr = qo.ref_compute_transformation(cs, ct)
            return time.perf_counter() - t1, rr
        el, rr = ref_time(p0["cs_h"], p0["ct_h"], LC)
        r = p0["result"] if composite else h.solve(p0["cs_h"], p0["ct_h"], ql.demo_params())
        entry = {
            "what": f"Quatro::computeTransformation of the reference's own text on the same {LC} correspondences "
                    "(/root/reference/include/quatro.hpp:307-386,430-747,769-936 + include/teaser/graph.h compiled "
                    "against an eager Eigen stand-in; findMaxClique answered by the oracle)",
            "seconds": el, "solves_per_s": 1.0 / el, "threads": 1,
            "note": "single thread: the stand-in Eigen evaluates eagerly and the reference's `#pragma omp parallel for` "
                    "in computeTIMs is compiled without -fopenmp; a reported baseline, not the target",
            "same_clique_as_gpu": bool(np.array_equal(np.sort(rr["clique"]), np.sort(r["clique"]))),
            "same_final_inliers_as_gpu": bool(np.array_equal(np.sort(rr["final_inliers"]), np.sort(r["final_inliers"]))),
            "max_abs_dT_vs_gpu": float(np.abs(rr["T"] - r["T"]).max())}
        out["cpu_reference_text"] = {f"L{LC}": entry}
        if "solver_L5000_leg" in extra and LC == 5000:
            extra["solver_L5000_leg"]["cpu_reference_text_solves_per_s"] = 1.0 / el
        if "refdense" in legs:
            from quatro_amd import synth
            s20, t20, _, _ = synth.correspondences(20000, 0.02, seed=7, noise=0.1)
            el20, _ = ref_time(s20, t20, 20000)
            out["cpu_reference_text"]["L20000"] = {"seconds": el20, "solves_per_s": 1.0 / el20, "threads": 1}
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
