"""The oracle must keep reproducing the committed fixtures (tests/golden/, made by make_golden.py)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _b(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_solver_fixture(qo):
    g = np.load(os.path.join(G, "solver_L300.npz"))
    r = qo.solve(g["src"], g["tgt"])
    assert np.array_equal(r["clique"], g["clique"]) and np.array_equal(r["final_inliers"], g["final_inliers"])
    assert np.array_equal(r["rot_inliers"], g["rot_inliers"]) and np.array_equal(r["T"], g["T"])
    assert r["gnc_iters"] == int(g["gnc_iters"]) and r["cost"] == float(g["cost"])
    assert np.array_equal(qo.build_graph(g["src"], g["tgt"]), g["bitmap"])
    assert np.array_equal(qo.kcore(g["bitmap"])[0], g["core"])
    # the planted inliers are (nearly all) in the clique, and the transform is near ground truth
    assert len(set(g["planted"]) - set(r["clique"])) <= 0.25 * len(g["planted"])
    assert np.linalg.norm(r["T"][:3, 3] - g["T_gt"][:3, 3]) < 0.3


def test_frontend_fixture(qo):
    g = np.load(os.path.join(G, "frontend_patch.npz"))
    nrm, sp, de = qo.fpfh(g["cloud"], 0.5, 0.75)
    both_nan = np.isnan(nrm) & np.isnan(g["normals"])
    assert np.all((_b(nrm) == _b(g["normals"])) | both_nan)
    assert np.array_equal(_b(sp), _b(g["spfh"])) and np.array_equal(_b(de), _b(g["fpfh"]))
    assert np.array_equal(_b(qo.voxelize(g["raw"], 0.3)), _b(g["vox"]))


def test_matcher_fixture(qo):
    g = np.load(os.path.join(G, "matcher_small.npz"))
    corr = qo.match(g["xyz_s"], g["desc_s"], g["xyz_t"], g["desc_t"], seed=int(g["seed"]))
    assert np.array_equal(corr, g["corr"])
