"""Anchors of the CPU oracle that the repository's own code did not produce.

* the REFERENCE's teaser::Matcher, compiled from /root/reference (oracle/Makefile target `ref`): live comparison when
  oracle/_ref/libref_matcher.so is present, and the committed outputs it generated (tests/golden/matcher_ref.npz);
* library cross-checks of the third-party semantics the oracle restates: scipy's cKDTree for the radius sets,
  numpy's eigh for pcl::eigen33, and a known answer of the published FPFH definition (Rusu 2009) on a plane.
"""
import os

import numpy as np
import pytest

from quatro_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def qo():
    from oracle import oracle as o
    o.build()
    o.set_threads(min(4, o.max_threads()))
    return o


CASES = {"ab": ("a", "b", True, True, 11), "ba": ("b", "a", True, True, 12), "ab_nocross": ("a", "b", False, True, 13),
         "ba_nocross_notuple": ("b", "a", False, False, 14), "ab_notuple": ("a", "b", True, False, 15)}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matcher_equals_reference_generated_golden(qo, name):
    """tests/golden/matcher_ref.npz holds what the reference's own advancedMatching returned (source larger / smaller,
    cross-check on / off, tuple test on / off): the oracle's restatement has to return the same lists."""
    g = np.load(os.path.join(G, "matcher_ref.npz"))
    s, t, cross, tup, seed = CASES[name]
    got = qo.match(g["xyz_" + s], g["desc_" + s], g["xyz_" + t], g["desc_" + t], crosscheck=cross, tuple_test=tup,
                   seed=seed)
    assert np.array_equal(got, g["corr_" + name])
    assert got.shape[0] > 20


def test_oracle_matcher_equals_compiled_reference_live(qo):
    """The same comparison against the compiled reference itself, on inputs with exact ties (duplicated descriptors),
    clouds of very different size and the degenerate single-point cloud."""
    if not (qo.build_ref() and qo.ref_available()):
        pytest.skip("oracle/_ref/libref_matcher.so is not available (no /root/reference, no prebuilt file)")
    rng = np.random.default_rng(3)
    for trial in range(6):
        na, nb = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        xa = rng.uniform(-20, 20, size=(na, 4)).astype(np.float32)
        xb = rng.uniform(-20, 20, size=(nb, 4)).astype(np.float32)
        da = rng.uniform(0, 100, size=(na, 33)).astype(np.float32)
        db = rng.uniform(0, 100, size=(nb, 33)).astype(np.float32)
        if trial % 2 and na > 10 and nb > 10:  # ties: copies inside a cloud and across the two clouds
            da[na // 2:na // 2 + 5] = da[0]
            db[:7] = da[0]
            db[nb // 2] = db[nb - 1]
        for cross in (True, False):
            for tup in (True, False):
                ref = qo.ref_match(xa, da, xb, db, crosscheck=cross, tuple_test=tup, seed=trial)
                got = qo.match(xa, da, xb, db, crosscheck=cross, tuple_test=tup, seed=trial)
                assert np.array_equal(ref, got), (trial, na, nb, cross, tup)
    s, t, _ = synth.kitti64_pair(5)
    vs, vt = qo.voxelize(s, 0.45), qo.voxelize(t, 0.45)
    _, _, ds = qo.fpfh(vs, 0.7, 1.0)
    _, _, dt = qo.fpfh(vt, 0.7, 1.0)
    for (a, da, b, db, seed) in ((vs, ds, vt, dt, 1), (vt, dt, vs, ds, 2)):
        assert np.array_equal(qo.ref_match(a, da, b, db, seed=seed), qo.match(a, da, b, db, seed=seed))


def test_radius_sets_against_scipy_ckdtree(qo):
    """pcl::search::KdTree::radiusSearch restated (sorted (d^2, index) lists, query included) vs scipy's cKDTree: the same
    sets, up to points within rounding of the radius."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(11)
    pts = np.zeros((3000, 4), dtype=np.float32)
    pts[:, :3] = rng.uniform(-6, 6, size=(3000, 3))
    pts[:50, :3] = pts[50:100, :3]  # duplicates
    r = 0.75
    off, idx, d2 = qo.radius_neighbors(pts, r)
    tree = cKDTree(pts[:, :3].astype(np.float64))
    inner = tree.query_ball_point(pts[:, :3].astype(np.float64), r * (1 - 1e-5))
    outer = tree.query_ball_point(pts[:, :3].astype(np.float64), r * (1 + 1e-5))
    for i in range(pts.shape[0]):
        mine = idx[off[i]:off[i + 1]]
        assert set(inner[i]) <= set(mine.tolist()) <= set(outer[i])
        dd = d2[off[i]:off[i + 1]]
        assert np.all(np.diff(dd) >= 0) and i in mine
        exact = ((pts[mine, :3].astype(np.float64) - pts[i, :3].astype(np.float64)) ** 2).sum(axis=1)
        assert np.allclose(dd, exact, rtol=1e-5, atol=1e-9)


def test_eigen33_against_numpy_eigh_on_random_covariances(qo):
    """pcl::eigen33 (closed-form smallest eigenpair, float) vs numpy.linalg.eigh (float64) on 100 000 random covariance
    matrices: eigenvalue to float accuracy (99.9 % below 2e-5 relative, all below 5e-4), eigenvector to 2e-3 rad whenever the smallest eigenvalue is separated."""
    rng = np.random.default_rng(5)
    n = 100000
    pts = rng.normal(size=(n, 12, 3)) * rng.uniform(0.05, 2.0, size=(n, 1, 3))
    rot = np.linalg.qr(rng.normal(size=(n, 3, 3)))[0]
    pts = pts @ rot
    c = pts - pts.mean(axis=1, keepdims=True)
    cov = (c.transpose(0, 2, 1) @ c / 12.0)
    ev, vec = qo.eigen33(cov.astype(np.float32))
    w, v = np.linalg.eigh(cov.astype(np.float32).astype(np.float64))
    scale = np.abs(cov).max(axis=(1, 2))
    err = np.abs(ev - w[:, 0]) / scale  # the closed form (trigonometric roots in float) is not an iterative solver:
    assert np.quantile(err, 0.999) < 2e-5 and err.max() < 5e-4  # a few ill-conditioned cases lose 3 more digits
    sep = (w[:, 1] - w[:, 0]) / scale
    ok = sep > 0.05
    assert ok.mean() > 0.8
    cosang = np.abs((vec.astype(np.float64) * v[:, :, 0]).sum(axis=1))
    assert np.all(np.arccos(np.clip(cosang[ok], -1, 1)) < 2e-3)
    assert np.allclose(np.linalg.norm(vec, axis=1), 1.0, atol=1e-5)


def test_fpfh_known_answer_on_a_plane(qo):
    """Published definition (Rusu et al. 2009; PCL's 11-bin layout): on a plane every Darboux frame gives
    f1 = atan2(w.n2, n1.n2) = 0, f2 = v.n2 = 0, f3 = n1.d/|d| = 0, i.e. the centre bin (5) of each 11-bin block.  The SPFH
    and the FPFH of every interior point are therefore (0,...,0,100,0,...,0) per block — exactly."""
    xs, ys = np.meshgrid(np.arange(-4, 4.001, 0.25), np.arange(-4, 4.001, 0.25))
    pts = np.zeros((xs.size, 4), dtype=np.float32)
    pts[:, 0] = xs.ravel()
    pts[:, 1] = ys.ravel()
    pts[:, 2] = -1.5  # below the viewpoint: every normal flips to +z
    nrm, sp, de = qo.fpfh(pts, 0.6, 0.9)
    assert np.allclose(np.abs(nrm[:, 2]), 1.0, atol=1e-5) and np.all(nrm[:, 2] > 0)
    interior = (np.abs(pts[:, 0]) < 2.5) & (np.abs(pts[:, 1]) < 2.5)
    want = np.zeros(33, dtype=np.float32)
    want[[5, 16, 27]] = 100.0
    assert np.allclose(sp[interior], want[None, :], atol=2e-3)
    assert np.allclose(de[interior], want[None, :], atol=2e-3)
