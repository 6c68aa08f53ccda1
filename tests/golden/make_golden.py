"""Generates the committed fixtures under tests/golden/: matcher_ref.npz from the REFERENCE's own teaser::Matcher and
solver_ref.npz from the reference's Eigen-only back-end functions (both compiled from /root/reference by
oracle/Makefile, targets `ref` and `ref_solver`), the others from the CPU oracle (the reference has no golden vectors
and its other stages cannot run here — see oracle/quatro_oracle.cpp header).  They pin the oracle against
accidental drift and give the GPU tests inputs/outputs that do not depend on the oracle being rebuilt.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import oracle as qo  # noqa: E402
from quatro_amd import synth  # noqa: E402


def main():
    qo.set_threads(1)
    # --- back end: 300 correspondences, 20 % inliers
    src, tgt, T, inl = synth.correspondences(300, 0.2, seed=42, noise=0.3)
    r = qo.solve(src, tgt)
    bm = qo.build_graph(src, tgt)
    core, _, mc = qo.kcore(bm)
    np.savez_compressed(os.path.join(HERE, "solver_L300.npz"), src=src, tgt=tgt, T_gt=T, planted=inl, T=r["T"],
                        clique=r["clique"], rot_inliers=r["rot_inliers"], final_inliers=r["final_inliers"],
                        cost=r["cost"], gnc_iters=r["gnc_iters"], bitmap=bm, core=core, max_core=mc)
    # --- front end: a 700-point patch of a synthetic scan
    s, t, _ = synth.kitti64_pair(7)
    v = qo.voxelize(s, 0.3)
    sel = v[np.argsort(np.linalg.norm(v[:, :2] - v[100, :2], axis=1))[:700]]
    sel = sel[np.lexsort((sel[:, 0], sel[:, 1], sel[:, 2]))].copy()
    nrm, sp, de = qo.fpfh(sel, 0.5, 0.75)
    raw = s[:6000].copy()
    vox = qo.voxelize(raw, 0.3)
    np.savez_compressed(os.path.join(HERE, "frontend_patch.npz"), cloud=sel, normals=nrm, spfh=sp, fpfh=de, raw=raw,
                        vox=vox)
    # --- matcher: descriptors of two overlapping patches
    v2 = qo.voxelize(t, 0.3)
    a = v[:900].copy()
    b = v2[:800].copy()
    _, _, da = qo.fpfh(a, 0.5, 0.75)
    _, _, db = qo.fpfh(b, 0.5, 0.75)
    corr, nn_ij, nn_ji = qo.match(a, da, b, db, seed=5, debug=True)
    np.savez_compressed(os.path.join(HERE, "matcher_small.npz"), xyz_s=a, desc_s=da, xyz_t=b, desc_t=db, corr=corr,
                        nn_large_of_small=nn_ij, nn_small_of_large=nn_ji, seed=5)
    # --- the REFERENCE's own matcher (oracle/_ref, compiled from /root/reference): outputs the repository's code did not
    # produce.  Cases: source larger / smaller (the swap), cross-check off, tuple test off.
    if qo.build_ref():
        v3 = qo.voxelize(synth.kitti64_pair(3)[0], 0.4)
        v4 = qo.voxelize(synth.kitti64_pair(3)[1], 0.4)
        a2, b2 = v3[:1400].copy(), v4[:1100].copy()
        _, _, da2 = qo.fpfh(a2, 0.6, 0.9)
        _, _, db2 = qo.fpfh(b2, 0.6, 0.9)
        out = {"xyz_a": a2, "desc_a": da2, "xyz_b": b2, "desc_b": db2}
        for name, (x1, d1, x2, d2, cross, tup, seed) in {
                "ab": (a2, da2, b2, db2, True, True, 11), "ba": (b2, db2, a2, da2, True, True, 12),
                "ab_nocross": (a2, da2, b2, db2, False, True, 13), "ba_nocross_notuple": (b2, db2, a2, da2, False, False, 14),
                "ab_notuple": (a2, da2, b2, db2, True, False, 15)}.items():
            out["corr_" + name] = qo.ref_match(x1, d1, x2, d2, crosscheck=cross, tuple_test=tup, seed=seed)
        np.savez_compressed(os.path.join(HERE, "matcher_ref.npz"), **out)
    # --- the REFERENCE's own back-end functions (oracle/_ref/libref_solver.so: computeTIMs, solveForScale,
    # solveForRotation2D, solveForTranslation, estimate of include/quatro.hpp, compiled from the header's text against an
    # Eigen-subset stand-in): inputs and what they returned
    if qo.build_ref_solver():
        g = np.random.default_rng(2024)
        out = {"gnc_noise_bound": qo.REF_GNC_NOISE_BOUND}
        # consistency graph: TIMs of both clouds + scale test
        src, tgt, _, _ = synth.correspondences(150, 0.3, seed=9, noise=0.05)
        ts, mp = qo.ref_compute_tims(src[:, :3].astype(np.float64))
        tt, _ = qo.ref_compute_tims(tgt[:, :3].astype(np.float64))
        out.update(graph_src=src, graph_tgt=tgt, tims_src=ts, tims_tgt=tt, tims_map=mp,
                   scale_mask=qo.ref_scale_mask(ts, tt, 0.3, 1.0))
        # GNC-TLS yaw on TIM-like 2-D pairs, several sizes / outlier rates
        for k, (M, frac) in enumerate(((6, 1.0), (40, 0.8), (150, 0.5), (400, 0.3))):
            ang = 0.25 + 0.07 * k
            Rt = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
            a = g.normal(0, 5, (M, 2))
            b = a @ Rt.T + g.normal(0, 0.05, (M, 2))
            bad = g.random(M) > frac
            b[bad] = g.normal(0, 5, (int(bad.sum()), 2))
            R, cost, inl = qo.ref_gnc_rotation2d(a, b)
            out.update({f"gnc{k}_src": a, f"gnc{k}_dst": b, f"gnc{k}_R": R, f"gnc{k}_cost": cost, f"gnc{k}_inl": inl})
        # COTE: estimate(), uniform and per-element ranges, both selection modes
        for k, N in enumerate((2, 3, 10, 77, 300)):
            X = g.normal(0, 1, N)
            X[:max(2, N // 2)] = 0.5 + g.normal(0, 0.05, max(2, N // 2))
            rg = g.uniform(0.05, 0.6, N)
            out[f"cote{k}_X"], out[f"cote{k}_ranges"] = X, rg
            for tag, ranges in (("u", np.full(N, 0.3)), ("r", rg)):
                for median in (1, 0):
                    e, m = qo.ref_cote_estimate(X, ranges, bool(median))
                    out[f"cote{k}_{tag}{median}_est"], out[f"cote{k}_{tag}{median}_inl"] = e, m
        # translation of a rotated cloud
        a = g.normal(0, 5, (120, 3))
        b = a + np.array([1.0, -2.0, 0.5]) + g.normal(0, 0.05, (120, 3))
        b[::4] += 3.0
        t, m = qo.ref_translation(a, b, 0.3, 1.0, True)
        out.update(trans_src=a, trans_dst=b, trans_t=t, trans_inl=m)
        # the whole back end: Quatro::computeTransformation of the reference (PMC's clique search answered by the
        # oracle's on the graph the reference code built), several parameter sets
        for k, (L, frac, noise, seed, kw) in enumerate((
                (3, 1.0, 0.0, 1, {}), (60, 0.3, 0.05, 2, {}), (200, 0.2, 0.1, 3, {"cote_median": 0}),
                (300, 0.2, 0.3, 42, {"using_rot_inliers_when_estimating_cote": 1}),
                (150, 0.05, 0.05, 5, {"inlier_selection_mode": 2}), (4, 0.0, 0.0, 9, {}))):
            src, tgt, _, _ = synth.correspondences(L, frac, seed=seed, noise=noise)
            r = qo.ref_compute_transformation(src, tgt, cote_median=bool(kw.get("cote_median", 1)),
                                              use_rot_inliers=bool(kw.get("using_rot_inliers_when_estimating_cote", 0)),
                                              inlier_selection_mode=kw.get("inlier_selection_mode", 1))
            out.update({f"ct{k}_src": src, f"ct{k}_tgt": tgt, f"ct{k}_valid": r["valid"], f"ct{k}_T": r["T"],
                        f"ct{k}_clique": r["clique"], f"ct{k}_rot": r["rot_inliers"], f"ct{k}_final": r["final_inliers"],
                        f"ct{k}_cote_median": int(kw.get("cote_median", 1)),
                        f"ct{k}_use_rot": int(kw.get("using_rot_inliers_when_estimating_cote", 0)),
                        f"ct{k}_mode": int(kw.get("inlier_selection_mode", 1))})
        np.savez_compressed(os.path.join(HERE, "solver_ref.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
