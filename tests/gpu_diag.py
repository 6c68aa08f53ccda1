"""GPU-vs-oracle diagnostic sweep (not a pytest module): runs every stage comparison, never stops at the
first mismatch, prints a summary and writes gpurun_out/diag.json.  Used while bringing kernels up."""
import json
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as qo  # noqa: E402
from quatro_amd import lib as ql  # noqa: E402
from quatro_amd import synth  # noqa: E402

OUT = {}


def rec(name, ok, **info):
    OUT[name] = {"ok": bool(ok), **{k: (v if isinstance(v, (int, float, str, bool, list)) else str(v)) for k, v in info.items()}}
    print(("PASS " if ok else "FAIL ") + name + " " + json.dumps(OUT[name])[:400], flush=True)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def main():
    qo.set_threads(min(8, qo.max_threads()))
    h = ql.Handle(0)
    rng = np.random.default_rng(1)
    # ---- math
    try:
        a = np.concatenate([rng.uniform(-1.2, 1.2, 200000), [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan]]).astype(np.float32)
        b = np.concatenate([rng.uniform(-1.2, 1.2, 200000), [0.0, -0.0, -0.0, 0.0, np.inf, 1.0, 1.0]]).astype(np.float32)
        for fn, nm in [(0, "atan2f"), (1, "acosf")]:
            d, o = h.debug_math(fn, a, b), qo.math_fn(fn, a, b)
            bad = int(((bits(d) != bits(o)) & ~(np.isnan(d) & np.isnan(o))).sum())
            rec("math_" + nm, bad == 0, mismatches=bad)
        th = rng.uniform(0, 1.2, 200000).astype(np.float32)
        for fn, nm in [(2, "sinf"), (3, "cosf")]:
            d, o = h.debug_math(fn, th), qo.math_fn(fn, th)
            rec("math_" + nm, int((bits(d) != bits(o)).sum()) == 0, mismatches=int((bits(d) != bits(o)).sum()))
    except Exception as e:
        rec("math", False, err=traceback.format_exc()[-600:])

    # ---- solver on synthetic correspondences
    for (L, frac, seed, noise) in [(2, 1.0, 0, 0.1), (50, 0.3, 1, 0.1), (300, 0.2, 2, 0.3), (1000, 0.1, 3, 0.35),
                                   (5000, 0.05, 4, 0.1), (5000, 0.05, 7, 0.3), (5000, 0.02, 5, 0.4), (3000, 0.0, 6, 0.1),
                                   (8000, 0.1, 8, 0.3)]:
        name = f"solve_L{L}_s{seed}"
        try:
            src, tgt, T, inl = synth.correspondences(L, frac, seed, noise=noise)
            t0 = time.time()
            r = h.solve(src, tgt)
            tg = time.time() - t0
            o = qo.solve(src, tgt)
            bm_g = h.debug_fetch(ql.DBG_GRAPH_BITMAP, np.uint64).reshape(L, -1)
            bm_o = qo.build_graph(src, tgt, 0.3, 1.0)
            bm_bad = int((bm_g != bm_o).sum())
            core_g = h.debug_fetch(ql.DBG_CORE, np.int32)
            core_o, _, mc_o = qo.kcore(bm_o)
            core_bad = int((core_g != core_o).sum())
            ok = (bm_bad == 0 and core_bad == 0 and r["valid"] == o["valid"] and np.array_equal(r["clique"], o["clique"])
                  and np.array_equal(r["rot_inliers"], o["rot_inliers"]) and np.array_equal(r["final_inliers"], o["final_inliers"])
                  and np.array_equal(r["T"], o["T"]) and r["gnc_iters"] == o["gnc_iters"]
                  and (r["cost"] == o["cost"] or (np.isinf(r["cost"]) and np.isinf(o["cost"]))))
            rec(name, ok, bm_bad=bm_bad, core_bad=core_bad, valid=[r["valid"], o["valid"]],
                clique=[int(r["clique"].size), int(o["clique"].size)], clique_eq=bool(np.array_equal(r["clique"], o["clique"])),
                rot=[int(r["rot_inliers"].size), int(o["rot_inliers"].size)],
                final=[int(r["final_inliers"].size), int(o["final_inliers"].size)],
                Tmaxdiff=float(np.abs(r["T"] - o["T"]).max()), iters=[r["gnc_iters"], o["gnc_iters"]],
                cost=[r["cost"], o["cost"]], ncard=[r["n_card"], o["n_card"]], max_core=[r["max_core"], o["max_core"]],
                edges=[r["n_edges"], o["n_edges"]], gpu_s=round(tg, 4), times=h.stage_times())
        except Exception:
            rec(name, False, err=traceback.format_exc()[-800:])

    # ---- front end on a synthetic scan pair
    try:
        src, tgt, Tgt = synth.kitti64_pair(0)
        # voxelize
        vs_o, vt_o = qo.voxelize(src, 0.3), qo.voxelize(tgt, 0.3)
        vs_g = h.voxelize(src, 0.3)
        tv = h.stage_times()
        okv = vs_g.shape == vs_o.shape and np.array_equal(bits(vs_g), bits(vs_o))
        rec("voxelize", okv, n=[int(vs_g.shape[0]), int(vs_o.shape[0])],
            bad=int((bits(vs_g) != bits(vs_o)).sum()) if vs_g.shape == vs_o.shape else -1, times=tv)
        # fpfh
        nrm_o, sp_o, de_o = qo.fpfh(vs_o, 0.5, 0.75)
        nrm_g, de_g = h.fpfh(vs_o, 0.5, 0.75)
        tf = h.stage_times()
        off_o, idx_o, d2_o = qo.radius_neighbors(vs_o, 0.75)
        off_g = h.debug_fetch(ql.DBG_NBR_OFFSETS, np.int32)
        n = vs_o.shape[0]
        cnt_ok = np.array_equal(off_g.astype(np.int64), off_o)
        rec("nbr_counts", cnt_ok, total=[int(off_g[-1]), int(off_o[-1])], kmax=int(np.diff(off_o).max()))
        sp_g = h.debug_fetch(ql.DBG_SPFH, np.float32).reshape(n, 33)
        nb = bits(nrm_g) != bits(nrm_o)
        nanboth = np.isnan(nrm_g) & np.isnan(nrm_o)
        nbad = int((nb & ~nanboth).any(axis=1).sum())
        rec("normals", nbad == 0, bad_points=nbad, nan=[int(np.isnan(nrm_g[:, 0]).sum()), int(np.isnan(nrm_o[:, 0]).sum())],
            maxabs=float(np.nanmax(np.abs(nrm_g - nrm_o))))
        sbad = int((bits(sp_g) != bits(sp_o)).any(axis=1).sum())
        rec("spfh", sbad == 0, bad_points=sbad, maxabs=float(np.abs(sp_g - sp_o).max()))
        fbad = int((bits(de_g) != bits(de_o)).any(axis=1).sum())
        rec("fpfh", fbad == 0, bad_points=fbad, maxabs=float(np.abs(de_g - de_o).max()), times=tf)
        # match on oracle descriptors (isolates the matcher)
        nrm_t, sp_t, de_t = qo.fpfh(vt_o, 0.5, 0.75)
        fp = ql.default_frontend_params(seed=7)
        corr_g = h.match(vs_o, de_o, vt_o, de_t, fp)
        tm = h.stage_times()
        corr_o, nn_ij, nn_ji = qo.match(vs_o, de_o, vt_o, de_t, seed=7, debug=True)
        nn_s = h.debug_fetch(ql.DBG_NN_LARGE_OF_SMALL, np.int32)
        nn_l = h.debug_fetch(ql.DBG_NN_SMALL_OF_LARGE, np.int32)
        hit = nn_ji >= 0
        rec("nn_tables", np.array_equal(nn_s, nn_ij) and np.array_equal(nn_l[hit], nn_ji[hit]),
            bad_small=int((nn_s != nn_ij).sum()), bad_large_hit=int((nn_l[hit] != nn_ji[hit]).sum()))
        rec("match", np.array_equal(corr_g, corr_o), L=[int(corr_g.shape[0]), int(corr_o.shape[0])], times=tm,
            stats=h.debug_fetch(ql.DBG_MATCH_STATS, np.int32).tolist())
        # whole path
        t0 = time.time()
        r = h.register_pair(src, tgt, fp)
        tg = time.time() - t0
        o = qo.register_pair(src, tgt, seed=7)
        ok = (r["valid"] == o["valid"] and r["n_src"] == o["n_src"] and r["n_tgt"] == o["n_tgt"] and r["L"] == o["L"]
              and np.array_equal(r["clique"], o["clique"]) and np.array_equal(r["final_inliers"], o["final_inliers"])
              and np.array_equal(r["T"], o["T"]))
        rec("register_pair", ok, n=[r["n_src"], r["n_tgt"], o["n_src"], o["n_tgt"]], L=[r["L"], o["L"]],
            clique=[int(r["clique"].size), int(o["clique"].size)], final=[int(r["final_inliers"].size), int(o["final_inliers"].size)],
            Tmaxdiff=float(np.abs(r["T"] - o["T"]).max()), gpu_s=round(tg, 4), times=h.stage_times(),
            yaw_err_vs_gt=float(np.arctan2(r["T"][1, 0], r["T"][0, 0]) - np.arctan2(Tgt[1, 0], Tgt[0, 0])),
            t_err_vs_gt=(r["T"][:3, 3] - Tgt[:3, 3]).tolist())
        for rep in range(3):
            t0 = time.time()
            h.register_pair(src, tgt, fp)
            print("register_pair wall %.4f s" % (time.time() - t0), h.stage_times(), flush=True)
    except Exception:
        rec("frontend", False, err=traceback.format_exc()[-1500:])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/diag.json", "w") as f:
        json.dump(OUT, f, indent=1)
    nfail = sum(1 for v in OUT.values() if not v["ok"])
    print(f"SUMMARY: {len(OUT) - nfail} pass / {nfail} fail")


if __name__ == "__main__":
    main()
