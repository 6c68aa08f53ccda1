"""Parity tests proper: the HIP path (called ONLY through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed fixtures, and — at full BASELINE sizes — through size-independent
properties.  Bar: bit-exact for integer/index outputs (inlier sets, cliques, correspondences, neighbour
lists) and, since both sides evaluate identical IEEE operation sequences, bit-exact floats as well; the
stated tolerance of the north star (1e-4 rad / 1e-3 m) is asserted explicitly on the transforms."""
import os
import sys
import time

import numpy as np
import pytest

from quatro_amd import lib as ql
from quatro_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROT_TOL, TRANS_TOL = 1e-4, 1e-3


def _b(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _yaw(T):
    return float(np.arctan2(T[1, 0], T[0, 0]))


def _assert_same_solution(r, o):
    assert r["valid"] == o["valid"]
    assert np.array_equal(r["clique"], o["clique"])
    assert np.array_equal(r["final_inliers"], o["final_inliers"])
    if r["rot_inliers"] is not None and "rot_inliers" in o:
        assert np.array_equal(r["rot_inliers"], o["rot_inliers"])
    if r["valid"]:
        d = _yaw(r["T"]) - _yaw(o["T"])
        assert abs(np.arctan2(np.sin(d), np.cos(d))) <= ROT_TOL
        assert np.abs(r["T"][:3, 3] - o["T"][:3, 3]).max() <= TRANS_TOL
        assert np.array_equal(r["T"], o["T"])  # identical operation sequences -> identical bits


# ---------------------------------------------------------------------------------------------- shared math
def test_device_math_is_bit_identical_to_host(hip, qo):
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(-1.2, 1.2, 300000), rng.uniform(-1e-6, 1e-6, 1000),
                        [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan]]).astype(np.float32)
    b = np.concatenate([rng.uniform(-1.2, 1.2, 300000), rng.uniform(-1e-6, 1e-6, 1000),
                        [0.0, -0.0, -0.0, 0.0, np.inf, 1.0, 1.0]]).astype(np.float32)
    for fn in (0, 1):
        d, o = hip.debug_math(fn, a, b), qo.math_fn(fn, a, b)
        assert np.all((_b(d) == _b(o)) | (np.isnan(d) & np.isnan(o)))
    th = rng.uniform(0, 1.2, 300000).astype(np.float32)
    for fn in (2, 3):
        assert np.array_equal(_b(hip.debug_math(fn, th)), _b(qo.math_fn(fn, th)))


@pytest.mark.gpu
def test_spfh_role_swap_shortcut_equals_the_reference_arithmetic(hip, qo):
    """k2_spfh decides the Darboux-frame role swap, acosf(|a1|) > acosf(|a2|), from the arguments and evaluates the two
    arc cosines only where rounding could matter (frontend.hip: spfh_swap_roles).  Against the oracle's plain arithmetic
    on 600 k pairs, half of them with arguments at most 2e-6 apart, plus equal, out-of-range and non-finite ones."""
    rng = np.random.default_rng(7)
    x = rng.uniform(0, 1.0000005, 600000)
    y = np.where(rng.random(600000) < 0.5, x + rng.uniform(-2e-6, 2e-6, 600000), rng.uniform(0, 1.0000005, 600000))
    sg = rng.choice([-1.0, 1.0], 600000)
    sp_a = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.0, 1e-40, 1.0, 0.5, 1.0000001, 2.0], dtype=np.float32)
    sp_b = np.array([0.0, -0.0, -0.0, 0.0, np.inf, 1.0, 1.0, -1.0, 1e-40, 1.0, 0.5, 1.0, np.nan], dtype=np.float32)
    a, b = np.concatenate([(x * sg).astype(np.float32), sp_a]), np.concatenate([y.astype(np.float32), sp_b])
    d, o = hip.debug_math(5, a, b), qo.math_fn(5, a, b)
    bad = np.nonzero(d != o)[0]
    assert bad.size == 0, (a[bad[:5]], b[bad[:5]], d[bad[:5]], o[bad[:5]])


# ---------------------------------------------------------------------------------------------- back end
def assert_cores(h, core_o):
    """QTR_DBG_CORE against the oracle's core numbers: equal at or above the floor the solve worked with (k_hcore_async
    stops lowering values below it; QTR_DBG_SOLVER_STATE[29], 0 = everything exact), an upper bound below the floor under
    it.  Returns the floor."""
    core_o = np.asarray(core_o)
    core_g = h.debug_fetch(ql.DBG_CORE, np.int32)[:core_o.size]
    floor = int(h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)[29])
    hi = core_o >= floor
    assert np.array_equal(core_g[hi], core_o[hi])
    assert np.all(core_g[~hi] >= core_o[~hi]) and np.all(core_g[~hi] < floor)
    return floor


@pytest.mark.parametrize("L,frac,seed,noise", [
    (2, 1.0, 0, 0.1), (3, 0.0, 1, 0.1), (50, 0.3, 1, 0.1), (64, 0.5, 2, 0.2), (65, 0.5, 3, 0.2), (300, 0.2, 2, 0.3),
    (1000, 0.1, 3, 0.35), (1281, 0.1, 10, 0.2), (2000, 0.05, 11, 0.2), (3000, 0.0, 6, 0.1), (5000, 0.05, 4, 0.1), (5000, 0.05, 7, 0.3), (5000, 0.02, 5, 0.4),
    (8192, 0.1, 8, 0.3), (8193, 0.03, 9, 0.3)])
def test_solver_matches_oracle(hip, qo, L, frac, seed, noise):
    src, tgt, T, inl = synth.correspondences(L, frac, seed, noise=noise)
    r, o = hip.solve(src, tgt), qo.solve(src, tgt)
    bm_g = hip.debug_fetch(ql.DBG_GRAPH_BITMAP, np.uint64).reshape(L, -1)
    bm_o = qo.build_graph(src, tgt, 0.3, 1.0)
    assert np.array_equal(bm_g, bm_o)
    assert_cores(hip, qo.kcore(bm_o)[0])
    assert r["max_core"] == o["max_core"] and r["n_edges"] == o["n_edges"]
    _assert_same_solution(r, o)
    assert r["gnc_iters"] == o["gnc_iters"] and r["n_card"] == o["n_card"]
    assert r["cost"] == o["cost"] or (np.isinf(r["cost"]) and np.isinf(o["cost"]))


def test_solver_edge_cases(hip, qo):
    e = np.zeros((0, 4), dtype=np.float32)
    r = hip.solve(e, e)
    assert not r["valid"] and r["status"] == ql.QTR_ERR_CLIQUE_TOO_SMALL
    one, one_t, _, _ = synth.correspondences(1, 1.0, 0)
    assert not hip.solve(one, one_t)["valid"]
    # identical duplicated correspondences: zero-length TIMs (0/0 in the reference predicate)
    src, tgt, _, _ = synth.correspondences(40, 0.5, 3)
    src[5], tgt[5] = src[4], tgt[4]
    _assert_same_solution(hip.solve(src, tgt), qo.solve(src, tgt))
    # unsupported / invalid modes are reported, not silently changed
    with pytest.raises(ql.QuatroHipError) as ei:
        hip.solve(src, tgt, ql.demo_params(inlier_selection_mode=ql.INLIER_NONE))
    assert ei.value.code == ql.QTR_ERR_UNSUPPORTED
    with pytest.raises(ql.QuatroHipError) as ei:
        hip.solve(src, tgt, ql.demo_params(noise_bound=-1.0))
    assert ei.value.code == ql.QTR_ERR_BAD_ARG


@pytest.mark.parametrize("kw", [dict(inlier_selection_mode=2, kcore_heuristic_threshold=0.05), dict(cote_median=0),
                                dict(using_rot_inliers_when_estimating_cote=1), dict(noise_bound=0.1, cbar2=2.0),
                                dict(rotation_max_iterations=3), dict(cote_noise_bound=0.15),
                                dict(using_pre_estimated_ryrx=1, ryrx=[0.9998, 0, 0.02, 0, 1, 0, -0.02, 0, 0.9998])])
def test_solver_parameter_variants(hip, qo, kw):
    src, tgt, _, _ = synth.correspondences(800, 0.15, seed=21, noise=0.3)
    _assert_same_solution(hip.solve(src, tgt, ql.demo_params(**kw)), qo.solve(src, tgt, qo.default_params(**kw)))


def test_solver_fixture(hip):
    g = np.load(os.path.join(G, "solver_L300.npz"))
    r = hip.solve(g["src"], g["tgt"])
    assert np.array_equal(r["clique"], g["clique"]) and np.array_equal(r["final_inliers"], g["final_inliers"])
    assert np.array_equal(r["T"], g["T"]) and r["gnc_iters"] == int(g["gnc_iters"])
    assert np.array_equal(hip.debug_fetch(ql.DBG_GRAPH_BITMAP, np.uint64).reshape(300, -1), g["bitmap"])
    assert np.array_equal(hip.debug_fetch(ql.DBG_CORE, np.int32), g["core"])


def test_solver_properties_at_full_size(hip):
    """BASELINE sizes (L = 5000 and the dense-mode L = 20000) through properties that need no oracle."""
    for L, frac in ((5000, 0.05), (20000, 0.02)):
        src, tgt, T, inl = synth.correspondences(L, frac, seed=77, noise=0.1)
        r = hip.solve(src, tgt)
        assert r["valid"]
        bm = hip.debug_fetch(ql.DBG_GRAPH_BITMAP, np.uint64).reshape(L, -1)
        C = r["clique"]
        rows = np.unpackbits(bm[C].view(np.uint8), axis=1, bitorder="little")[:, :L][:, C]
        assert rows.sum() == len(C) * (len(C) - 1)           # the clique is a clique of the device bitmap
        assert set(inl).issubset(set(C))                      # planted mutually-consistent inliers recovered
        assert set(r["final_inliers"]).issubset(set(C))
        d = _yaw(r["T"]) - _yaw(T)
        assert abs(np.arctan2(np.sin(d), np.cos(d))) < 5e-3 and np.linalg.norm(r["T"][:3, 3] - T[:3, 3]) < 0.15
        # idempotence: the same call again returns the same answer
        r2 = hip.solve(src, tgt)
        assert np.array_equal(r2["clique"], C) and np.array_equal(r2["T"], r["T"])
        # linearity: a rigid motion of the target composes with the estimate
        R2 = synth.yaw_matrix(0.4)
        tgt2 = tgt.copy()
        tgt2[:, :3] = (tgt[:, :3].astype(np.float64) @ R2.T + np.array([1.0, -2.0, 0.5])).astype(np.float32)
        r3 = hip.solve(src, tgt2)
        d = _yaw(r3["T"]) - (_yaw(r["T"]) + 0.4)
        assert abs(np.arctan2(np.sin(d), np.cos(d))) < 2e-3


# ---------------------------------------------------------------------------------------------- front end
def test_voxelize_matches_oracle(hip, qo, small_pair):
    s, t, _ = small_pair
    for cloud, leaf in ((s, 0.3), (t, 0.3), (s[:777], 0.5), (s[:1], 0.3), (np.repeat(s[:1], 50, axis=0), 0.3)):
        g, o = hip.voxelize(cloud, leaf), qo.voxelize(cloud, leaf)
        assert g.shape == o.shape and np.array_equal(_b(g), _b(o))


def test_voxelize_properties_full_size(hip):
    s, t, _ = synth.kitti64_pair(11)
    v = hip.voxelize(s, 0.3)
    inv = np.float32(1.0) / np.float32(0.3)
    key = np.floor(v[:, :3] * inv).astype(np.int64)
    assert len(np.unique(key, axis=0)) == v.shape[0]       # one centroid per occupied voxel
    mn = np.floor(s[:, :3].min(0) * inv).astype(np.int64)
    dv = np.floor(s[:, :3].max(0) * inv).astype(np.int64) - mn + 1
    lin = (key - mn) @ np.array([1, dv[0], dv[0] * dv[1]])
    assert np.all(np.diff(lin) > 0)                         # ascending linear voxel index (PCL output order)
    assert np.array_equal(_b(hip.voxelize(v, 0.3)), _b(v))  # idempotent on its own output


def test_fpfh_matches_oracle(hip, qo, small_pair):
    s, t, _ = small_pair
    v = qo.voxelize(s, 0.3)
    nrm_o, sp_o, de_o = qo.fpfh(v, 0.5, 0.75)
    nrm_g, de_g = hip.fpfh(v, 0.5, 0.75)
    n = v.shape[0]
    off_o, idx_o, d2_o = qo.radius_neighbors(v, 0.75)
    off_g = hip.debug_fetch(ql.DBG_NBR_OFFSETS, np.int32)
    assert np.array_equal(off_g.astype(np.int64), off_o)
    sp_g = hip.debug_fetch(ql.DBG_SPFH, np.float32).reshape(n, 33)
    assert np.all((_b(nrm_g) == _b(nrm_o)) | (np.isnan(nrm_g) & np.isnan(nrm_o)))
    assert np.isnan(nrm_o[:, 0]).sum() > 0                  # the NaN-normal branch is exercised
    assert np.array_equal(_b(sp_g), _b(sp_o))
    assert np.array_equal(_b(de_g), _b(de_o))


def test_fpfh_normalisation_sums_on_clustered_points(hip, qo):
    """FPFH on a cloud of tight clusters (neighbour distances from 1e-4 m to 0.7 m: weights 1/d^2 spread over 2^25).  The
    kernel's private binary64 sums are only used when every term of a block lies within 17 binary exponents — here they
    do not, and the reference-order fallback has to give the oracle's bits; long lists (> one staged chunk) too."""
    g = np.random.default_rng(11)
    centres = g.uniform(-6, 6, (220, 3))
    pts = []
    for c in centres:
        m = int(g.integers(2, 9))
        scale = 10.0 ** g.uniform(-4, -0.5)
        pts.append(c + g.normal(0, scale, (m, 3)))
    pts.append(g.uniform(-1, 1, (400, 3)) * [1.0, 1.0, 0.05])  # a dense slab: neighbour lists of 60-250 entries
    v = np.zeros((sum(len(x) for x in pts), 4), np.float32)
    v[:, :3] = np.concatenate(pts)
    nrm_o, sp_o, de_o = qo.fpfh(v, 0.5, 0.75)
    nrm_g, de_g = hip.fpfh(v, 0.5, 0.75)
    sp_g = hip.debug_fetch(ql.DBG_SPFH, np.float32).reshape(v.shape[0], 33)
    assert np.all((_b(nrm_g) == _b(nrm_o)) | (np.isnan(nrm_g) & np.isnan(nrm_o)))
    assert np.array_equal(_b(sp_g), _b(sp_o))
    assert np.array_equal(_b(de_g), _b(de_o))
    off_o, idx_o, d2_o = qo.radius_neighbors(v, 0.75)
    w = 1.0 / d2_o[d2_o > 0]
    assert w.max() / w.min() > 2.0 ** 20 and np.diff(off_o).max() > 64


def test_fpfh_fixture_and_argument_check(hip):
    g = np.load(os.path.join(G, "frontend_patch.npz"))
    nrm, de = hip.fpfh(g["cloud"], 0.5, 0.75)
    assert np.all((_b(nrm) == _b(g["normals"])) | (np.isnan(nrm) & np.isnan(g["normals"])))
    assert np.array_equal(_b(de), _b(g["fpfh"]))
    assert np.array_equal(_b(hip.voxelize(g["raw"], 0.3)), _b(g["vox"]))
    with pytest.raises(ql.QuatroHipError) as ei:            # reference include/fpfh_manager.hpp:99-102
        hip.fpfh(g["cloud"], 0.9, 0.5)
    assert ei.value.code == ql.QTR_ERR_BAD_ARG


def test_match_matches_oracle(hip, qo, small_pair):
    s, t, _ = small_pair
    vs, vt = qo.voxelize(s, 0.3), qo.voxelize(t, 0.3)
    _, _, ds = qo.fpfh(vs, 0.5, 0.75)
    _, _, dt = qo.fpfh(vt, 0.5, 0.75)
    for (a, da, b, db, seed) in ((vs, ds, vt, dt, 7), (vt, dt, vs, ds, 8), (vs[:500], ds[:500], vt[:2000], dt[:2000], 9)):
        corr_g = hip.match(a, da, b, db, ql.default_frontend_params(seed=seed))
        corr_o, nn_ij, nn_ji = qo.match(a, da, b, db, seed=seed, debug=True)
        assert np.array_equal(hip.debug_fetch(ql.DBG_NN_LARGE_OF_SMALL, np.int32), nn_ij)
        # the second direction is only asked for the rows the first one points at (reference feature_matcher.cc:113-122):
        # -1 everywhere else, on both sides
        assert np.array_equal(hip.debug_fetch(ql.DBG_NN_SMALL_OF_LARGE, np.int32), nn_ji)
        assert np.array_equal(corr_g, corr_o)
        nt = hip.match(a, da, b, db, ql.default_frontend_params(seed=seed, use_tuple_test=0))
        assert np.array_equal(nt, qo.match(a, da, b, db, tuple_test=False))
        # use_crosscheck = false (reference feature_matcher.cc:146-181): corres_ij + corres_ji, with and without tuple test
        for tup in (1, 0):
            nc = hip.match(a, da, b, db, ql.default_frontend_params(seed=seed, use_crosscheck=0, use_tuple_test=tup))
            assert np.array_equal(nc, qo.match(a, da, b, db, crosscheck=False, tuple_test=bool(tup), seed=seed))


def test_match_without_crosscheck_sorts_hub_lists_like_the_oracle(qo):
    """use_crosscheck = false (reference feature_matcher.cc:146-181: corres_ij + corres_ji, sorted, unique) when ONE descriptor is
    the nearest neighbour of thousands of the other cloud's (flat ground does that to FPFH): the per-source target lists are
    then thousands of entries long (k_nc_unique's workgroup path: a bit set per NC_RANGE = 16384 targets), here with either
    cloud as the source, with targets on both sides of a range boundary, and with every descriptor identical (all of a cloud on one list)."""
    rng = np.random.default_rng(77)
    h = ql.Handle(0, max_points=32768, max_voxels=32768, max_corr=32768)

    def cloud(n):
        c = np.zeros((n, 4), dtype=np.float32)
        c[:, :3] = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
        return c

    def check(a, da, b, db, seed):
        for tup in (0, 1):
            got = h.match(a, da, b, db, ql.default_frontend_params(seed=seed, use_crosscheck=0, use_tuple_test=tup))
            want = qo.match(a, da, b, db, crosscheck=False, tuple_test=bool(tup), seed=seed)
            assert np.array_equal(got, want), (a.shape[0], b.shape[0], tup)
            if tup == 0:
                untested = want
        return untested  # (the list before the tuple test thins it out)

    try:
        # (a list's entries are the SMALLER cloud's rows that are nearest to one row of the larger, source, cloud)
        # 1400 rows of the smaller cloud nearest to row 5000 of the larger one; either cloud as the source
        a, b = cloud(3000), cloud(6000)
        da = rng.uniform(0, 100, (3000, 33)).astype(np.float32)
        db = rng.uniform(0, 100, (6000, 33)).astype(np.float32)
        da[100:1500] = db[5000] + rng.normal(0, 0.01, (1400, 33)).astype(np.float32)
        check(a, da, b, db, 21)
        w = check(b, db, a, da, 22)
        assert np.count_nonzero(w[:, 0] == 5000) >= 1400
        # 18 000 sources, 17 000 targets, 3000 of them (14 000 .. 16 999: across the bit set's range boundary) on one list
        a, b = cloud(18000), cloud(17000)
        da = rng.uniform(0, 100, (18000, 33)).astype(np.float32)
        db = rng.uniform(0, 100, (17000, 33)).astype(np.float32)
        db[14000:] = da[123] + rng.normal(0, 0.01, (3000, 33)).astype(np.float32)
        w = check(a, da, b, db, 23)
        on = w[w[:, 0] == 123, 1]
        assert on.size >= 3000 and on.min() < 16384 < on.max()
        # every descriptor of a cloud identical: all of the smaller cloud on one list
        a, b = cloud(2000), cloud(3000)
        da = np.tile(rng.uniform(0, 100, (1, 33)).astype(np.float32), (2000, 1))
        db = np.tile(rng.uniform(0, 100, (1, 33)).astype(np.float32), (3000, 1))
        check(a, da, b, db, 25)
        w = check(b, db, a, da, 26)
        assert np.bincount(w[:, 0]).max() >= 2000
    finally:
        h.close()


@pytest.mark.parametrize("name,order,cross,tup,seed", [("ab", "ab", 1, 1, 11), ("ba", "ba", 1, 1, 12),
                                                       ("ab_notuple", "ab", 1, 0, 15), ("ab_nocross", "ab", 0, 1, 13),
                                                       ("ba_nocross_notuple", "ba", 0, 0, 14)])
def test_match_equals_reference_generated_golden(hip, name, order, cross, tup, seed):
    """tests/golden/matcher_ref.npz: what the REFERENCE's own teaser::Matcher (compiled from /root/reference, oracle/
    Makefile target `ref`) returned for these inputs — the HIP matcher has to return the same lists."""
    g = np.load(os.path.join(G, "matcher_ref.npz"))
    s_, t_ = order[0], order[1]
    corr = hip.match(g["xyz_" + s_], g["desc_" + s_], g["xyz_" + t_], g["desc_" + t_],
                     ql.default_frontend_params(seed=seed, use_tuple_test=tup, use_crosscheck=cross))
    assert np.array_equal(corr, g["corr_" + name])


def test_match_fixture(hip):
    g = np.load(os.path.join(G, "matcher_small.npz"))
    corr = hip.match(g["xyz_s"], g["desc_s"], g["xyz_t"], g["desc_t"], ql.default_frontend_params(seed=int(g["seed"])))
    assert np.array_equal(corr, g["corr"])


# ---------------------------------------------------------------------------------------------- whole path
@pytest.mark.parametrize("pair_id", [0, 1, 2])
def test_register_pair_matches_oracle(hip, qo, pair_id):
    s, t, Tgt = synth.kitti64_pair(pair_id)
    r = hip.register_pair(s, t, ql.default_frontend_params(seed=pair_id))
    o = qo.register_pair(s, t, seed=pair_id)
    assert (r["n_src"], r["n_tgt"], r["L"]) == (o["n_src"], o["n_tgt"], o["L"])
    _assert_same_solution(r, o)
    d = _yaw(r["T"]) - _yaw(Tgt)
    assert abs(np.arctan2(np.sin(d), np.cos(d))) < 0.02 and np.linalg.norm(r["T"][:3, 3] - Tgt[:3, 3]) < 0.3


def test_back_end_stages_equal_reference_generated_golden(hip):
    """The device's stage entry points against what the reference's OWN functions returned (tests/golden/solver_ref.npz,
    made by the compiled include/quatro.hpp functions: computeTIMs, solveForScale, solveForRotation2D, estimate): TIMs,
    index map, scale mask, COTE estimate and inliers bit for bit; the GNC-TLS yaw with the same inliers and the
    rotation / cost to rounding (fixed 64-lane summation order, closed-form 2 x 2 rotation)."""
    g = np.load(os.path.join(G, "solver_ref.npz"))
    src, tgt = g["graph_src"], g["graph_tgt"]
    ts, mp = hip.compute_tims(np.ascontiguousarray(src[:, :3].astype(np.float64).T))
    tt, _ = hip.compute_tims(np.ascontiguousarray(tgt[:, :3].astype(np.float64).T))
    assert np.array_equal(ts.T, g["tims_src"]) and np.array_equal(tt.T, g["tims_tgt"]) and np.array_equal(mp.T, g["tims_map"])
    assert np.array_equal(hip.scale_mask(ts, tt, 0.3, 1.0), g["scale_mask"])
    nb = float(g["gnc_noise_bound"])
    for k in range(4):
        R, cost, iters, inl = hip.gnc_rotation2d(g[f"gnc{k}_src"], g[f"gnc{k}_dst"], nb)
        assert np.array_equal(inl, g[f"gnc{k}_inl"]), k
        assert np.abs(R - g[f"gnc{k}_R"]).max() < 1e-12, k
        rc = float(g[f"gnc{k}_cost"])
        assert (cost == rc) or abs(cost - rc) <= 1e-9 * abs(rc) + 1e-18, k
    for k in range(5):
        X, rg = g[f"cote{k}_X"], g[f"cote{k}_ranges"]
        for tag, ranges in (("u", np.full(X.shape[0], 0.3)), ("r", rg)):
            for median in (1, 0):
                e, m, _ = hip.cote_estimate_ranges(X, ranges, bool(median))
                assert e == float(g[f"cote{k}_{tag}{median}_est"]), (k, tag, median)
                assert np.array_equal(m, g[f"cote{k}_{tag}{median}_inl"]), (k, tag, median)


def test_solve_equals_reference_compute_transformation_golden(hip):
    """qtr_solve against the reference's OWN Quatro::computeTransformation (tests/golden/solver_ref.npz, ct* entries:
    include/quatro.hpp:769-936 compiled from its text; only PMC's clique search was answered by the oracle's): the same
    clique, rotation inliers, final inliers; the 4 x 4 to rounding; the same invalid case."""
    g = np.load(os.path.join(G, "solver_ref.npz"))
    for k in range(6):
        kw = dict(cote_median=int(g[f"ct{k}_cote_median"]),
                  using_rot_inliers_when_estimating_cote=int(g[f"ct{k}_use_rot"]), inlier_selection_mode=int(g[f"ct{k}_mode"]))
        o = hip.solve(g[f"ct{k}_src"], g[f"ct{k}_tgt"], ql.demo_params(**kw))
        assert o["valid"] == bool(g[f"ct{k}_valid"]), k
        if not o["valid"]:
            assert g[f"ct{k}_clique"].size <= 1  # "Clique size too small. Abort." (include/quatro.hpp:809-813)
            continue
        assert np.array_equal(np.sort(o["clique"]), g[f"ct{k}_clique"]), k
        assert np.array_equal(o["rot_inliers"], g[f"ct{k}_rot"]) and np.array_equal(o["final_inliers"], g[f"ct{k}_final"]), k
        assert np.abs(o["T"] - g[f"ct{k}_T"]).max() < 1e-12, k


def test_host_mirror_reads_like_the_reference_demo(hip, qo, small_pair):
    """examples/run_global_registration.cpp:103-108, 206-221, 243-246 through quatro_amd.api."""
    from quatro_amd import api
    s, t, Tgt = small_pair
    quatro = api.Quatro(handle=hip)
    params = api.Params(rotation_max_iterations=50, rotation_cost_threshold=1.1e-4)
    quatro.reset(params)
    src_feat, tgt_feat = api.voxelize(s, 0.3, handle=hip), api.voxelize(t, 0.3, handle=hip)
    fm = api.FPFHManager(0.5, 0.75, handle=hip, seed=2)
    fm.flushAllFeatures()
    fm.setFeaturePair(src_feat, tgt_feat)
    quatro.setInputSource(fm.getSrcKps())
    quatro.setInputTarget(fm.getTgtKps())
    out = np.eye(4)
    quatro.computeTransformation(out)
    o = qo.register_pair(s, t, seed=2)
    assert quatro.solution_.valid and np.array_equal(out, o["T"])
    assert quatro.getFinalInliersIndices() == o["final_inliers"].tolist()
    assert quatro.getNumMaxCliqueInliers() == o["clique"].size
    assert len(fm.getCorrespondences()) == o["L"]


@pytest.mark.parametrize("pcl_shaped", [False, True])
def test_cpp_dropin_demo_matches_python_path(hip, qo, small_pair, tmp_path, pcl_shaped):
    """The reference demo's call sequence compiled against include/quatro.hpp + include/fpfh_manager.hpp
    (tests/cpp/dropin_demo.cpp) gives the oracle's answer, digit for digit — with the headers' built-in pcl:: / Eigen::
    stand-ins and with the QUATRO_HAVE_PCL branch over tests/cpp/pcl_stub (boost::shared_ptr, aligned-allocator storage,
    column-major Eigen)."""
    import subprocess

    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    s, t, _ = small_pair
    synth.save_kitti_bin(str(tmp_path / "src.bin"), s)
    synth.save_kitti_bin(str(tmp_path / "tgt.bin"), t)
    exe = str(tmp_path / "dropin_demo")
    libdir = os.path.join(root, "quatro_amd")
    stub = ["-I", os.path.join(root, "tests", "cpp", "pcl_stub")] if pcl_shaped else []
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + stub + ["-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "dropin_demo.cpp"), "-o", exe, "-L", libdir,
                           "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = None
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):  # system HIP runtime, else torch's copy
        env = dict(os.environ)
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        p = subprocess.run([exe, str(tmp_path / "src.bin"), str(tmp_path / "tgt.bin"), "2"], env=env,
                           capture_output=True, text=True, timeout=300)
        if p.returncode == 0:
            out = p.stdout
            break
    assert out is not None, p.stderr[-500:]
    o = qo.register_pair(s, t, seed=2)
    lines = out.strip().splitlines()
    head = dict(zip(lines[0].split()[0::2], lines[0].split()[1::2]))
    assert (int(head["n_src"]), int(head["n_tgt"]), int(head["L"])) == (o["n_src"], o["n_tgt"], o["L"])
    assert int(head["clique"]) == o["clique"].size and int(head["valid"]) == 1
    T = np.array([[float(x) for x in ln.split()[1:]] for ln in lines[1:5]])
    assert np.array_equal(T, o["T"])
    final = [int(x) for x in lines[5].split()[1:]]
    assert final == o["final_inliers"].tolist()


def _nn_tables(engine, vs, ds, vt, dt, seed=4):
    """engine "" = the product library (one path: the f16-split filter); "exact" / "mfma32" = the comparison engines, which
    only exist in the -DQTR_TEST_ENGINES build (quatro_amd/build.py) and are selected there with QTR_NN_ENGINE"""
    if engine:
        os.environ["QTR_NN_ENGINE"] = engine
    try:
        h = ql.Handle(0, lib_path=ql.TEST_ENGINES_LIB_PATH if engine else None)
    finally:
        os.environ.pop("QTR_NN_ENGINE", None)
    corr = h.match(vs, ds, vt, dt, ql.default_frontend_params(seed=seed))
    out = (corr, h.debug_fetch(ql.DBG_NN_LARGE_OF_SMALL, np.int32), h.debug_fetch(ql.DBG_NN_SMALL_OF_LARGE, np.int32),
           h.debug_fetch(ql.DBG_MATCH_STATS, np.int32))
    h.close()
    return out


def test_nn_engines_produce_identical_tables(qo, small_pair):
    """The three nearest-neighbour engines — f16-split MFMA filter (default), f32 MFMA filter, all-exact VALU — give the
    same tables (the filters certify most rows and hand the rest to the exact re-check), and the re-check fraction stays
    small."""
    s, t, _ = small_pair
    vs, vt = qo.voxelize(s, 0.3), qo.voxelize(t, 0.3)
    _, _, ds = qo.fpfh(vs, 0.5, 0.75)
    _, _, dt = qo.fpfh(vt, 0.5, 0.75)
    tables = {e: _nn_tables(e, vs, ds, vt, dt) for e in ("exact", "mfma32", "")}
    for e in ("mfma32", ""):
        for a, b in zip(tables["exact"][:3], tables[e][:3]):
            assert np.array_equal(a, b), e
        stats = tables[e][3]
        assert stats[12] == 0  # descriptor values inside the f16 engine's range
        assert stats[8] + stats[9] < 0.2 * (vs.shape[0] + vt.shape[0]), e  # rows that needed the exact re-check


def test_mfma_f16_accumulation_stays_inside_the_budget(tmp_path):
    """The one hardware assumption of the f16-split filter's rounding bound — how far the f32 accumulation of
    v_mfma_f32_32x32x16_f16 (one instruction, and a chain of seven) can be from the exact sum of its exact products — is
    measured on this device (tests/gpu_checks/mfma_f16_accumulation.hip: the kernel's operand shapes and adversarial
    exponent spreads, 2 million sums): it has to stay below half of the 16 u the bound budgets."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = str(tmp_path / "mfma_f16_accumulation")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value",
                           os.path.join(root, "tests", "gpu_checks", "mfma_f16_accumulation.hip"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-800:] + p.stderr[-400:]
    assert "worst" in p.stdout


@pytest.mark.parametrize("case", ["big_values", "big_norm", "tiny_values", "wide_range", "near_ties"])
def test_f16_filter_holds_on_adversarial_descriptors(qo, case):
    """The f16-split filter's rounding bound and range guard on descriptors that are nothing like FPFH histograms: values
    outside the f16 range (every row must go to the exact re-check), norms outside it, values far below the f16 normal
    range, twelve orders of magnitude inside one cloud, and clusters of near-identical rows.  Tables equal the all-exact
    engine's."""
    g = np.random.default_rng({"big_values": 1, "big_norm": 2, "tiny_values": 3, "wide_range": 4, "near_ties": 5}[case])
    ns, nt = 1500, 1700
    vs = np.zeros((ns, 4), np.float32)
    vt = np.zeros((nt, 4), np.float32)
    vs[:, :3] = g.uniform(-20, 20, (ns, 3))
    vt[:, :3] = g.uniform(-20, 20, (nt, 3))
    if case == "big_values":
        ds = g.uniform(0, 900, (ns, 33)).astype(np.float32)
        dt = g.uniform(0, 900, (nt, 33)).astype(np.float32)
    elif case == "big_norm":
        ds = g.uniform(0, 250, (ns, 33)).astype(np.float32)  # values in range, |b|^2 ~ 7e5 is not
        dt = g.uniform(0, 250, (nt, 33)).astype(np.float32)
    elif case == "tiny_values":
        ds = (g.uniform(0, 1, (ns, 33)) * 1e-6).astype(np.float32)
        dt = (g.uniform(0, 1, (nt, 33)) * 1e-6).astype(np.float32)
    elif case == "wide_range":
        ds = (10.0 ** g.uniform(-10, 2, (ns, 33))).astype(np.float32)
        dt = (10.0 ** g.uniform(-10, 2, (nt, 33))).astype(np.float32)
    else:
        centres = g.uniform(0, 40, (40, 33))
        ds = (centres[g.integers(0, 40, ns)] * (1 + 1e-7 * g.integers(-3, 4, (ns, 33)))).astype(np.float32)
        dt = (centres[g.integers(0, 40, nt)] * (1 + 1e-7 * g.integers(-3, 4, (nt, 33)))).astype(np.float32)
    ref = _nn_tables("exact", vs, ds, vt, dt)
    got = _nn_tables("", vs, ds, vt, dt)
    for a, b in zip(ref[:3], got[:3]):
        assert np.array_equal(a, b)
    assert (got[3][12] != 0) == (case in ("big_values", "big_norm"))


def test_instrumentation_entry_points(small_pair):
    """qtr_set_stage_events / qtr_get_nn_totals: with the stage events off a registration gives the same record and the
    nearest-neighbour launches keep being timed; the totals add up over calls and reset on request."""
    s, t, _ = small_pair
    h = ql.Handle(0)
    fp = ql.default_frontend_params(seed=3)
    a = h.register_pair(s, t, fp)
    assert h.stage_times()["match"] > 0
    h.nn_totals(reset=True)
    h.set_stage_events(False)
    for _ in range(3):
        b = h.register_pair(s, t, fp)
    ms, n = h.nn_totals()
    assert n == 6 and 0.0 < ms < 50.0
    assert h.nn_totals(reset=True)[1] == 6 and h.nn_totals()[1] == 0
    h.set_stage_events(True)
    c = h.register_pair(s, t, fp)
    assert h.stage_times()["match"] > 0 and h.nn_totals()[1] == 2
    # qtr_set_nn_event_stride: every second match carries the event pairs (the first one after the call does), none at 0
    h.set_nn_event_stride(2)
    h.nn_totals(reset=True)
    for _ in range(4):
        d = h.register_pair(s, t, fp)
    assert h.nn_totals(reset=True)[1] == 4
    h.set_nn_event_stride(0)
    e = h.register_pair(s, t, fp)
    assert h.nn_totals()[1] == 0 and h.stage_times()["nn_kernel"] == 0
    h.set_nn_event_stride(1)
    for r in (b, c, d, e):
        assert np.array_equal(a["T"], r["T"]) and np.array_equal(a["final_inliers"], r["final_inliers"])
    h.close()


def test_stream_slots_run_concurrently_and_agree(qo):
    """Three pairs in flight on three stream slots (one host thread each) give the answers of sequential runs."""
    import threading
    h = ql.Handle(0, n_slots=3)
    pairs = [synth.kitti64_pair(i) for i in range(3)]
    seq = [h.register_pair(s, t, ql.default_frontend_params(seed=i), slot=0) for i, (s, t, _) in enumerate(pairs)]
    out = [None] * 3

    def work(i):
        s, t, _ = pairs[i]
        for _ in range(3):
            out[i] = h.register_pair(s, t, ql.default_frontend_params(seed=i), slot=i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    for a, b in zip(seq, out):
        assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["final_inliers"], b["final_inliers"])
        assert np.array_equal(a["clique"], b["clique"]) and a["L"] == b["L"]
    h.close()


# ---------------------------------------------------------------------------------------------- edge cases
def test_edge_cases_through_the_abi(hip, qo):
    rng = np.random.default_rng(0)
    # empty and tiny clouds
    e4 = np.zeros((0, 4), dtype=np.float32)
    assert hip.voxelize(e4, 0.3).shape == (0, 4)
    for n in (1, 2, 3, 5):
        c = np.zeros((n, 4), dtype=np.float32)
        c[:, :3] = rng.uniform(-1, 1, (n, 3))
        nrm_g, de_g = hip.fpfh(c, 0.5, 0.75)
        nrm_o, _, de_o = qo.fpfh(c, 0.5, 0.75)
        assert np.all((_b(nrm_g) == _b(nrm_o)) | (np.isnan(nrm_g) & np.isnan(nrm_o)))
        assert np.array_equal(_b(de_g), _b(de_o))
    # isolated points (no neighbour inside either radius): NaN normals, all-zero descriptors
    far = np.zeros((6, 4), dtype=np.float32)
    far[:, 0] = np.arange(6) * 10.0
    nrm, de = hip.fpfh(far, 0.5, 0.75)
    assert np.isnan(nrm[:, :3]).all() and not de.any()
    # exact duplicates inside a cloud (d2 == 0 neighbours are skipped by the FPFH weighting)
    dup = np.zeros((40, 4), dtype=np.float32)
    dup[:, :3] = rng.uniform(-0.6, 0.6, (40, 3))
    dup[7] = dup[3]
    nrm_g, de_g = hip.fpfh(dup, 0.5, 0.75)
    nrm_o, _, de_o = qo.fpfh(dup, 0.5, 0.75)
    assert np.array_equal(_b(de_g), _b(de_o))
    # matcher with very unequal / minimal sizes, and tuple test off
    a = np.zeros((1, 4), dtype=np.float32)
    da = rng.uniform(0, 50, (1, 33)).astype(np.float32)
    b = np.zeros((700, 4), dtype=np.float32)
    b[:, :3] = rng.uniform(-5, 5, (700, 3))
    db = rng.uniform(0, 50, (700, 33)).astype(np.float32)
    for (x, dx, y, dy) in ((a, da, b, db), (b, db, a, da)):
        for tup in (0, 1):
            fp = ql.default_frontend_params(seed=3, use_tuple_test=tup)
            assert np.array_equal(hip.match(x, dx, y, dy, fp), qo.match(x, dx, y, dy, seed=3, tuple_test=bool(tup)))
    assert hip.match(e4, np.zeros((0, 33), np.float32), b, db).shape[0] == 0
    # identical descriptors everywhere: every NN is a tie -> lowest index, through the exact re-check path
    same = np.tile(da, (300, 1))
    pts = np.zeros((300, 4), dtype=np.float32)
    pts[:, :3] = rng.uniform(-5, 5, (300, 3))
    assert np.array_equal(hip.match(pts, same, b[:200], np.tile(da, (200, 1)), ql.default_frontend_params(seed=1)),
                          qo.match(pts, same, b[:200], np.tile(da, (200, 1)), seed=1))
    # whole path on clouds too small / too sparse to register: soft failure, not a crash
    tiny = np.zeros((30, 4), dtype=np.float32)
    tiny[:, :3] = rng.uniform(-20, 20, (30, 3))
    r = hip.register_pair(tiny, tiny[::-1].copy(), ql.default_frontend_params(seed=0))
    o = qo.register_pair(tiny, tiny[::-1].copy(), seed=0)
    assert r["valid"] == o["valid"] and r["L"] == o["L"] and np.array_equal(r["clique"], o["clique"])
    # capacity errors are reported as such
    small = ql.Handle(0, max_points=4096, max_voxels=1024, max_corr=64)
    big = np.zeros((5000, 4), dtype=np.float32)
    with pytest.raises(ql.QuatroHipError) as ei:
        small.voxelize(big, 0.3)
    assert ei.value.code == ql.QTR_ERR_CAPACITY
    s100, t100, _, _ = synth.correspondences(100, 0.5, 1)
    with pytest.raises(ql.QuatroHipError) as ei:
        small.solve(s100, t100)
    assert ei.value.code == ql.QTR_ERR_CAPACITY
    small.close()


def test_full_size_raw_cloud_and_voxel_grid_extremes(qo):
    """Loader-cap sized scan (250 000 points) and degenerate grids."""
    hip = ql.Handle(0, max_points=262144, max_voxels=262144, max_corr=1024)
    rng = np.random.default_rng(5)
    P = 250000
    c = np.zeros((P, 4), dtype=np.float32)
    c[:, :3] = rng.normal(0, 12, (P, 3)).astype(np.float32)
    c[:, 2] *= 0.1
    g, o = hip.voxelize(c, 0.3), qo.voxelize(c, 0.3)
    assert g.shape == o.shape and np.array_equal(_b(g), _b(o))
    # all points in one voxel (a 60 000-point run: exercises the long-run path of the centroid kernel)
    one = np.zeros((60000, 4), dtype=np.float32)
    one[:, :3] = rng.uniform(0.01, 0.29, (60000, 3)).astype(np.float32)
    g, o = hip.voxelize(one, 0.3), qo.voxelize(one, 0.3)
    assert g.shape == (1, 4) and np.array_equal(_b(g), _b(o))
    # leaf too small for the extent: PCL passes the input through
    far = np.array([[0, 0, 0, 0], [1e5, 1e5, 1e5, 0], [5, 5, 5, 0]], dtype=np.float32)
    assert np.array_equal(hip.voxelize(far, 0.001), far) and qo.voxelize(far, 0.001).shape[0] == 3
    hip.close()


# ------------------------------------------------------------------------- teaser::Graph / MaxCliqueSolver
def _random_graph_bitmap(L, p, seed, planted=0):
    rng = np.random.default_rng(seed)
    A = np.triu(rng.random((L, L)) < p, 1)
    if planted:
        mem = rng.choice(L, planted, replace=False)
        A[np.ix_(mem, mem)] = True
    A = np.triu(A, 1)
    A = A | A.T
    W = (L + 63) // 64
    bits = np.zeros((L, W * 64), dtype=np.uint8)
    bits[:, :L] = A
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, W), A


def test_core_numbers_when_only_the_first_rows_have_to_move(hip, qo):
    """k_hcore_async evaluates every row at least once: workgroup 0's first snapshot is the one its floor was computed from,
    and in a graph where every OTHER vertex's degree is its core number nothing ever moves again — vertex 0 (degree 2 or 5
    between leaves) used to keep its degree as its core number, and the heuristic then started from it instead of from the
    reference's highest-ranked vertex (found by tests/gpu_fuzz.py seed 72 on a 64-correspondence pair inside a batch group)."""
    for L, edges in ((2000, [(0, 700), (0, 900), (1500, 1900)]),
                     (3000, [(0, 10), (0, 1000), (0, 2000), (0, 2500), (0, 2999), (5, 6), (2400, 2800)]),
                     (1500, [(1, 40), (1, 50), (1, 60), (2, 70), (2, 80), (1400, 1499)])):
        A = np.zeros((L, L), dtype=bool)
        for a, b in edges:
            A[a, b] = A[b, a] = True
        W = (L + 63) // 64
        bits = np.zeros((L, W * 64), dtype=np.uint8)
        bits[:, :L] = A
        bm = np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, W)
        core, _, mc = qo.kcore(bm)
        for mode, thr in ((1, 0.5), (2, 0.5)):
            got, max_core = hip.max_clique(bm, mode, thr)
            assert max_core == mc == 1
            assert np.array_equal(hip.debug_fetch(ql.DBG_CORE, np.int32)[:L], core), (L, mode)
            assert np.array_equal(got, qo.max_clique(bm, mode, thr)), (L, mode)


def test_core_numbers_and_cliques_of_degenerate_graphs(hip, qo):
    """Graphs the consistency test of real correspondences rarely builds, through qtr_max_clique above the size where
    k_hcore_async takes over (1280): no edge at all, a perfect matching, a path (its h-index iteration is a chain of L / 2
    dependent rounds: the bounded iteration gives up and the peeling workgroup takes over), a cycle, a star, a star whose
    centre is the LAST vertex, two disjoint cliques of equal size — core numbers and the heuristic's clique against the oracle."""
    def graph(L, edges):
        A = np.zeros((L, L), dtype=bool)
        e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        A[e[:, 0], e[:, 1]] = True
        A[e[:, 1], e[:, 0]] = True
        W = (L + 63) // 64
        bits = np.zeros((L, W * 64), dtype=np.uint8)
        bits[:, :L] = A
        return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, W)

    L = 1600
    k1, k2 = list(range(100, 112)), list(range(1500, 1512))
    cases = {
        "empty": [],
        "matching": [(2 * i, 2 * i + 1) for i in range(L // 2)],
        "path": [(i, i + 1) for i in range(L - 1)],
        "cycle": [(i, (i + 1) % L) for i in range(L)],
        "star": [(0, i) for i in range(1, L)],
        "star_last": [(L - 1, i) for i in range(L - 1)],
        "two_cliques": [(a, b) for k in (k1, k2) for a in k for b in k if a < b],
    }
    for name, edges in cases.items():
        bm = graph(L, edges)
        core, _, mc = qo.kcore(bm)
        for mode, thr in ((1, 0.5), (2, 0.5)):
            got, max_core = hip.max_clique(bm, mode, thr)
            assert max_core == mc, (name, mode)
            assert np.array_equal(hip.debug_fetch(ql.DBG_CORE, np.int32)[:L], core), (name, mode)
            assert np.array_equal(got, qo.max_clique(bm, mode, thr)), (name, mode)


def test_batch_group_with_a_tiny_pair_whose_largest_clique_is_an_edge(qo):
    """tests/golden/batch_tie_case*.npz: the two groups of tests/gpu_fuzz.py seed 72 (correspondence-only pair descriptors of
    5000 / 64 / 300 / 300 / 2500 and 0 / 300 / 5000 / 64 / 9000 correspondences, noise bound 0.05) whose 64-correspondence
    member came back with another edge than the oracle's: inside a group every pair runs the large pair's kernels."""
    hb = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=12288, n_slots=8)
    try:
        for f in ("batch_tie_case2176.npz", "batch_tie_case5740.npz"):
            d = np.load(os.path.join(G, f))
            nb = float(d["noise_bound"])
            sets = [(d[f"src{j}"], d[f"tgt{j}"]) for j in range(len(d["sizes"]))]
            got = hb.register_batch([(None, None, 0, a, b) for a, b in sets], params=ql.demo_params(noise_bound=nb))
            for j, (a, b) in enumerate(sets):
                o = qo.solve(a, b, qo.default_params(noise_bound=nb))
                assert np.array_equal(got[j]["clique"], np.sort(o["clique"])), (f, j)
                if o["valid"]:
                    assert np.array_equal(got[j]["T"], o["T"]), (f, j)
    finally:
        hb.close()


@pytest.mark.parametrize("L,p,planted,seed", [(1, 0.0, 0, 0), (2, 1.0, 0, 0), (65, 0.3, 0, 1), (200, 0.05, 12, 2),
                                              (777, 0.02, 25, 3), (1500, 0.3, 40, 4), (3000, 0.01, 30, 5),
                                              (2600, 0.01, 1300, 6), (6000, 0.005, 2500, 7)])
def test_max_clique_entry_matches_oracle(hip, qo, L, p, planted, seed):
    """qtr_max_clique (the teaser::MaxCliqueSolver boundary) on arbitrary graphs, both heuristic modes."""
    bm, A = _random_graph_bitmap(L, p, seed, planted)
    core, _, mc = qo.kcore(bm)
    for mode, thr in ((1, 0.5), (2, 0.5), (2, 0.0), (2, 1.0)):
        ref = qo.max_clique(bm, mode, thr)
        got, max_core = hip.max_clique(bm, mode, thr)
        assert np.array_equal(got, ref), (mode, thr)
        assert max_core == mc
        if got.size:
            sub = A[np.ix_(got, got)]
            assert sub.sum() == got.size * (got.size - 1) or (mode == 2)  # KCORE_HEU returns the top core, not a clique
    assert np.array_equal(hip.debug_fetch(ql.DBG_CORE, np.int32)[:L], core)


def _bitmap_of(A):
    L = A.shape[0]
    A = np.triu(A, 1)
    A = A | A.T
    bits = np.zeros((L, ((L + 63) // 64) * 64), dtype=np.uint8)
    bits[:, :L] = A
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, -1)


def test_core_number_floor_and_second_run(hip, qo):
    """k_hcore_async's floor (values below half the h-index H of the degrees are not lowered any further, where H stands
    clear of the mean degree) and what the clique search makes of it.  (a) a planted clique far above the graph's bulk:
    the floor is in force, the bulk's numbers are upper bounds, the clique and the largest core are the oracle's.  (b) a
    dense block that is NOT a clique (200 vertices, every other pair joined) lifts H to ~100 while the largest clique has
    a dozen members: the search under the injected bound comes back empty and the stage runs a second time with exact
    numbers — the oracle's clique again.  (c) no structure at all: H is the bulk's own, no floor, one run."""
    bm, _ = _random_graph_bitmap(4000, 0.01, 31, planted=300)
    core, _, mc = qo.kcore(bm)
    got, max_core = hip.max_clique(bm, 1)
    st = hip.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    assert st[29] >= 100 and st[22] == 0, st
    assert assert_cores(hip, core) == st[29]
    assert np.array_equal(got, qo.max_clique(bm, 1, 0.5)) and max_core == mc and got.size >= 300
    L = 3000
    rng = np.random.default_rng(5)
    A = np.triu(rng.random((L, L)) < 0.005, 1)
    blk = rng.choice(L, 200, replace=False)
    A[np.ix_(blk, blk)] |= rng.random((200, 200)) < 0.5
    bm = _bitmap_of(A)
    core, _, mc = qo.kcore(bm)
    got, max_core = hip.max_clique(bm, 1)
    st = hip.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    assert st[22] == 1 and st[29] == 0, st  # second run, exact
    assert np.array_equal(got, qo.max_clique(bm, 1, 0.5)) and max_core == mc
    assert_cores(hip, core)
    bm, _ = _random_graph_bitmap(3000, 0.02, 8)
    got, max_core = hip.max_clique(bm, 1)
    st = hip.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    assert st[22] == 0 and st[29] == 0, st
    assert np.array_equal(got, qo.max_clique(bm, 1, 0.5))


def test_clique_search_under_the_floor_sweep(hip, qo):
    """The regime the floor's argument has to carry: a dense block (not a clique) sets the floor F ~ 0.75 x its size / 2,
    and cliques are planted AROUND F — F - 1, F, F + 1, F + 2 (the search under the injected bound accepts sizes above
    F only), far above it, several of similar size, none — in graphs of the size k_hcore_async takes, over a sweep of
    densities.  Every clique must be the oracle's, whichever way the stage went."""
    rng = np.random.default_rng(77)
    ways = {"floor": 0, "second run": 0, "exact": 0}
    for case in range(60):
        L = int(rng.integers(1300, 3400))
        p = float(rng.choice([0.004, 0.01, 0.02, 0.04]))
        m = max(6, int(p * L))  # ~ the bulk's mean degree
        A = np.triu(rng.random((L, L)) < p, 1)
        B = 6 * m  # the block: degrees ~ 3 m + m inside, h-index ~ 3.5 m, floor ~ 1.75 m
        blk = rng.choice(L, min(B, L), replace=False)
        if case % 6 != 5:
            A[np.ix_(blk, blk)] |= rng.random((blk.size, blk.size)) < 0.5
        bm0 = _bitmap_of(A)
        deg = np.unpackbits(bm0.view(np.uint8), axis=1).sum(1)
        H = int(np.sum(np.sort(deg)[::-1] >= np.arange(1, L + 1)))
        F = H // 2
        sizes = [[], [F - 1], [F], [F + 1], [F + 2], [F + 5], [2 * F], [F + 1, F + 1], [F + 2, F, F - 1], [3 * F]][case % 10]
        for sz in sizes:
            mem = rng.choice(L, max(2, min(sz, L)), replace=False)
            A[np.ix_(mem, mem)] = True
        bm = _bitmap_of(A)
        got, max_core = hip.max_clique(bm, 1)
        st = hip.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
        ref = qo.max_clique(bm, 1, 0.5)
        assert np.array_equal(got, ref), (case, L, p, sizes, st[22], st[29])
        core, _, mc = qo.kcore(bm)
        assert max_core == mc, (case, L, p, sizes)
        assert_cores(hip, core)
        ways["second run" if st[22] else "floor" if st[29] else "exact"] += 1
    assert ways["floor"] >= 10 and ways["second run"] >= 5, ways


def test_scout_floor_on_large_graphs(hip, qo):
    """Above 8192 vertices a single pair's k_hcore_async launch carries a scout workgroup (solver.hip: hca_scout) that
    peels the sub-graph of the largest values down to a clique and publishes a floor below its size — no bet on the degree
    sequence.  (a) the generator's correspondences with 2.5 % planted at L = 12 000 (h-index of the degrees 1.6 x their
    mean: the bet stays away): the floor is the scout's, one run, clique / largest core / core numbers at or above the
    floor the oracle's.  (b) nothing planted: no clique of sixteen to see, no floor.  (c) a graph handed in as a bit
    matrix, a clique beside a dense block that is none."""
    src, tgt, _, inl = synth.correspondences(12000, 0.025, 21, noise=0.05)
    r, o = hip.solve(src, tgt), qo.solve(src, tgt)
    st = hip.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    bm_o = qo.build_graph(src, tgt, 0.3, 1.0)
    assert st[22] == 0 and st[29] >= 200, (st[22], st[29])  # (the clique has 300 members: 300 - 1 - 37)
    assert assert_cores(hip, qo.kcore(bm_o)[0]) == st[29]
    assert r["max_core"] == o["max_core"] and r["n_edges"] == o["n_edges"]
    _assert_same_solution(r, o)
    src, tgt, _, _ = synth.correspondences(9000, 0.0, 22, noise=0.05)
    r, o = hip.solve(src, tgt), qo.solve(src, tgt)
    st = hip.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    assert st[29] == 0, st[29]
    assert_cores(hip, qo.kcore(qo.build_graph(src, tgt, 0.3, 1.0))[0])
    _assert_same_solution(r, o)
    L = 8600
    rng = np.random.default_rng(9)
    A = np.triu(rng.random((L, L), dtype=np.float32) < 0.006, 1)
    blk = rng.choice(L, 400, replace=False)
    A[np.ix_(blk, blk)] |= rng.random((400, 400)) < 0.5
    mem = rng.choice(L, 120, replace=False)
    A[np.ix_(mem, mem)] = True
    bm = _bitmap_of(A)
    core, _, mc = qo.kcore(bm)
    got, max_core = hip.max_clique(bm, 1)
    assert np.array_equal(got, qo.max_clique(bm, 1, 0.5)) and max_core == mc
    assert_cores(hip, core)


def test_scout_floor_sweep_on_small_graphs():
    """The scout on the graphs of the floor's own sweep (dense blocks, cliques around the bet's floor, overlapping
    near-cliques, a clique inside a block): tests/gpu_scout_sweep.py under the test-engine build, whose
    QTR_HCORE_SCOUT_MIN_L lets the scout run below 8192 vertices; every case against the oracle."""
    import subprocess, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, QTR_HCORE_SCOUT_MIN_L="1000", GRAFT_REPO_ROOT=root)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scout_sweep.py"), "56"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and "SCOUT_SWEEP_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    ways = eval(p.stdout.strip().splitlines()[-2].split("ways", 1)[1])
    assert ways["scout"] >= 3, ways


def test_max_clique_entry_hygiene_and_errors(hip, qo):
    bm, _ = _random_graph_bitmap(130, 0.2, 9, 10)
    ref, _ = hip.max_clique(bm, 1)
    dirty = bm.copy()
    for i in range(130):  # self loops and garbage past column L must be ignored
        dirty[i, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
        dirty[i, 2] |= np.uint64(0xFFFF) << np.uint64(40)
    got, _ = hip.max_clique(dirty, 1)
    assert np.array_equal(got, ref)
    with pytest.raises(ql.QuatroHipError) as e:
        hip.max_clique(bm, 3)  # NONE is not a clique solver mode
    assert e.value.code == ql.QTR_ERR_UNSUPPORTED
    empty, mcore = hip.max_clique(np.zeros((0, 0), dtype=np.uint64), 1)
    assert empty.size == 0 and mcore == 0
    none, _ = hip.max_clique(np.zeros((10, 1), dtype=np.uint64), 1)
    assert none.size == 0 and qo.max_clique(np.zeros((10, 1), dtype=np.uint64)).size == 0
    # the solver still works on the slot afterwards (shared arenas)
    src, tgt, _, _ = synth.correspondences(300, 0.2, seed=1, noise=0.05)
    _assert_same_solution(hip.solve(src, tgt), qo.solve(src, tgt))


def test_cpp_teaser_graph_dropin(hip, qo, tmp_path):
    """teaser::Graph + teaser::MaxCliqueSolver from include/teaser/graph.h (tests/cpp/graph_demo.cpp)."""
    import subprocess

    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    bm, A = _random_graph_bitmap(400, 0.05, 11, 18)
    ii, jj = np.nonzero(np.triu(A, 1))
    with open(tmp_path / "edges.txt", "w") as f:
        f.write("400\n")
        for a, b in zip(ii, jj):
            f.write(f"{a} {b}\n")
            if (a + b) % 7 == 0:
                f.write(f"{b} {a}\n")  # duplicate: Graph::addEdge must refuse it
    exe = str(tmp_path / "graph_demo")
    libdir = os.path.join(root, "quatro_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "graph_demo.cpp"), "-o", exe, "-L", libdir,
                           "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])

    def run(mode, thr):
        for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
            env = dict(os.environ)
            if extra:
                env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
            p = subprocess.run([exe, str(tmp_path / "edges.txt"), str(mode), str(thr)], env=env, capture_output=True,
                               text=True, timeout=300)
            if p.returncode == 0:
                return p.stdout.strip()
        raise AssertionError(p.stderr[-500:])

    out = run(1, 0.5).split()
    assert out[:2] == ["vertices", "400"] and int(out[3]) == ii.size
    assert [int(x) for x in out[7:]] == qo.max_clique(bm, 1, 0.5).tolist()
    assert int(out[5]) == qo.kcore(bm)[2]
    out = run(2, 0.0).split()
    assert [int(x) for x in out[7:]] == qo.max_clique(bm, 2, 0.0).tolist()
    out = run(0, 0.5).split()  # PMC_EXACT, the class's default mode
    assert [int(x) for x in out[7:]] == np.sort(qo.max_clique(bm, 0)).tolist()


def test_cpp_objects_on_several_threads_use_their_own_slots(hip, tmp_path):
    """Independent drop-in objects driven from four threads at once (tests/cpp/threads_demo.cpp): every wrapper call
    leases a free stream slot of the process-wide handle instead of queueing behind one mutex, and every thread gets the
    answers it gets alone."""
    import subprocess

    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    bm, A = _random_graph_bitmap(500, 0.05, 5, 20)
    ii, jj = np.nonzero(np.triu(A, 1))
    with open(tmp_path / "edges.txt", "w") as f:
        f.write("500\n")
        for a, b in zip(ii, jj):
            f.write(f"{a} {b}\n")
    exe = str(tmp_path / "threads_demo")
    libdir = os.path.join(root, "quatro_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "threads_demo.cpp"), "-o", exe, "-L", libdir,
                           "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    p = None
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
        env = dict(os.environ)
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        p = subprocess.run([exe, str(tmp_path / "edges.txt"), "4", "25"], env=env, capture_output=True, text=True, timeout=300)
        if p.returncode == 0:
            break
    assert p.returncode == 0, (p.stdout[-300:], p.stderr[-500:])
    assert "mismatches 0 slots 4" in p.stdout


def test_dense_mode_front_end_at_50k_points(qo):
    """BASELINE configs[4] (dense mode: 50 000-point clouds, no voxel step) through the front end: FPFH + matching
    at n = 50 k via properties (descriptor blocks sum to 100 or 0, rigid-motion invariance of the matching) and
    a sampled exact check of the nearest-neighbour tables against brute force."""
    n = 50000
    src, tgt, perm = synth.dense_pair(n, seed=7, independent=False)  # moved copy: the invariance properties need it
    rng = np.random.default_rng(70)
    h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=24576)
    try:
        ns, ds = h.fpfh(src, 0.5, 0.75)
        nt, dt = h.fpfh(tgt, 0.5, 0.75)
        for d in (ds, dt):
            blocks = d.reshape(n, 3, 11).sum(axis=2)
            ok = np.isclose(blocks, 100.0, atol=1e-2) | np.isclose(blocks, 0.0, atol=1e-6) | ~np.isfinite(blocks)
            assert ok.mean() > 0.999
        # rigid-motion invariance: the descriptor of a point and of its moved copy agree closely
        good = np.isfinite(ds).all(axis=1) & np.isfinite(dt[np.argsort(perm)]).all(axis=1)
        diff = np.abs(ds[good] - dt[np.argsort(perm)][good]).max(axis=1)
        assert np.median(diff) < 2.0
        corr = h.match(src, ds, tgt, dt, ql.default_frontend_params(seed=3))
        assert corr.shape[0] > 1000
        # most correspondences are the planted ones (src i <-> tgt position of i)
        inv = np.argsort(perm)
        assert (inv[corr[:, 0]] == corr[:, 1]).mean() > 0.5
        # sampled exactness of the NN table (small cloud's NN in the large one): brute force in flann order
        nn = h.debug_fetch(ql.DBG_NN_LARGE_OF_SMALL, np.int32)
        small, large = dt, ds  # equal sizes: the source plays the larger cloud
        qs = rng.choice(n, 64, replace=False)
        ref = qo.nn33(small[qs], large)
        assert np.array_equal(nn[qs], ref)
    finally:
        h.close()


# ------------------------------------------------------------- individually callable stages (reference :307-747)
def test_stage_entry_points_match_oracle_and_numpy(hip, qo):
    """qtr_compute_tims / qtr_scale_mask / qtr_gnc_rotation2d / qtr_cote_estimate — the public stage methods of the
    reference class served by the same device code as the fused path."""
    rng = np.random.default_rng(5)
    # computeTIMs: column order and index map of the reference (segment start i*N - i(i+1)/2)
    N = 57
    v = rng.standard_normal((3, N)) * 10
    tims, mp = hip.compute_tims(v)
    ii, jj = np.triu_indices(N, 1)
    assert np.array_equal(mp[0], ii) and np.array_equal(mp[1], jj)
    assert np.array_equal(tims, v[:, jj] - v[:, ii])
    # solveForScale mask over TIMs == the consistency graph of the same points
    src, tgt, _, _ = synth.correspondences(300, 0.2, seed=3, noise=0.05)
    ts, _ = hip.compute_tims(src[:, :3].T.astype(np.float64))
    tt, m2 = hip.compute_tims(tgt[:, :3].T.astype(np.float64))
    mask = hip.scale_mask(ts, tt, 0.3, 1.0)
    bm = qo.build_graph(src, tgt)
    bits = np.unpackbits(bm.view(np.uint8), axis=1, bitorder="little")[:, :300].astype(bool)
    assert np.array_equal(mask, bits[m2[0], m2[1]])
    # solveForRotation2D
    for seed in range(4):
        g = np.random.default_rng(seed)
        M = [7, 64, 129, 500][seed]
        th = g.uniform(-3, 3)
        Rz = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        a = g.standard_normal((M, 2)) * 8
        b = a @ Rz.T + 0.05 * g.standard_normal((M, 2))
        out = g.random(M) < 0.4
        b[out] = g.standard_normal((int(out.sum()), 2)) * 8
        Ro, co, io, mo = qo.gnc_rotation2d(a, b, 0.6)
        Rg, cg, ig, mg = hip.gnc_rotation2d(a, b, 0.6)
        assert np.array_equal(Rg, Ro) and cg == co and ig == io and np.array_equal(mg, mo)
    # estimate (COTE)
    # (129 .. 256 members: the split ranking + merge of the interval endpoints; the "ties" cases put lower and upper
    # endpoints of different members, and members themselves, on exactly equal keys: X on a 0.5 grid, range 0.25)
    for seed, Nn in enumerate([2, 3, 50, 130, 200, 256, 301, 700, -180, -256]):
        g = np.random.default_rng(100 + seed)
        ties = Nn < 0
        Nn = abs(Nn)
        X = np.concatenate([1.5 + 0.1 * g.standard_normal(Nn - Nn // 3), g.uniform(-20, 20, Nn // 3)])
        if ties:
            X = np.round(X * 2) / 2
            for median in (True, False):
                eo, mo, no = qo.cote_estimate(X, 0.25, median)
                eg, mg, ng = hip.cote_estimate(X, 0.25, median)
                assert eg == eo and ng == no and np.array_equal(mg, mo), (Nn, median, "ties")
        for median in (True, False):
            eo, mo, no = qo.cote_estimate(X, 0.3, median)
            eg, mg, ng = hip.cote_estimate(X, 0.3, median)
            assert eg == eo and ng == no and np.array_equal(mg, mo), (Nn, median)
            R = g.uniform(0.05, 0.6, Nn)  # one range per element (include/quatro.hpp:618-630 takes a vector)
            eo, mo, no = qo.cote_estimate_ranges(X, R, median)
            eg, mg, ng = hip.cote_estimate_ranges(X, R, median)
            assert eg == eo and ng == no and np.array_equal(mg, mo), (Nn, median, "ranges")


@pytest.mark.parametrize("pcl_shaped", [False, True])
def test_cpp_stage_methods_and_front_end_classes(hip, qo, small_pair, tmp_path, pcl_shaped):
    """tests/cpp/stages_demo.cpp: teaser::FPFHEstimation / teaser::Matcher and the public stage methods of class
    Quatro from this repository's headers give the oracle's numbers, digit for digit (also over tests/cpp/pcl_stub,
    i.e. the QUATRO_HAVE_PCL branch with column-major Eigen matrices)."""
    import subprocess

    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    s, t, _ = small_pair
    vs, vt = qo.voxelize(s, 0.3), qo.voxelize(t, 0.3)
    synth.save_kitti_bin(str(tmp_path / "src.bin"), vs)
    synth.save_kitti_bin(str(tmp_path / "tgt.bin"), vt)
    exe = str(tmp_path / "stages_demo")
    libdir = os.path.join(root, "quatro_amd")
    stub = ["-I", os.path.join(root, "tests", "cpp", "pcl_stub")] if pcl_shaped else []
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + stub + ["-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "stages_demo.cpp"), "-o", exe, "-L", libdir,
                           "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = None
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
        env = dict(os.environ)
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        p = subprocess.run([exe, str(tmp_path / "src.bin"), str(tmp_path / "tgt.bin"), "2"], env=env,
                           capture_output=True, text=True, timeout=300)
        if p.returncode == 0:
            out = p.stdout
            break
    assert out is not None, p.stderr[-500:]
    lines = out.strip().splitlines()
    _, _, ds = qo.fpfh(vs, 0.5, 0.75)
    _, _, dt = qo.fpfh(vt, 0.5, 0.75)
    corr = qo.match(vs, ds, vt, dt, seed=2)
    head = dict(zip(lines[0].split()[0::2], lines[0].split()[1::2]))
    assert (int(head["n_src"]), int(head["n_tgt"]), int(head["L"])) == (vs.shape[0], vt.shape[0], corr.shape[0])
    assert lines[1].split()[1:] == [f"{a}:{b}" for a, b in corr[:8]]
    N = min(200, corr.shape[0])
    a = vs[corr[:N, 0]]
    b = vt[corr[:N, 1]]
    bm = qo.build_graph(a, b)
    n_pairs = int(np.unpackbits(bm.view(np.uint8), bitorder="little").sum()) // 2
    l2 = lines[2].split()
    assert int(l2[1]) == N * (N - 1) // 2 and float(l2[3]) == 1.0 and int(l2[5]) == n_pairs
    assert (int(l2[7]), int(l2[8])) == (N - 2, N - 1)
    R, _, _, _ = qo.gnc_rotation2d(a[:, :2].astype(np.float64), b[:, :2].astype(np.float64), 0.3)
    assert [float(x) for x in lines[3].split()[1:]] == [R[0, 0], R[0, 1], R[1, 0], R[1, 1]]
    R3 = np.eye(3)
    R3[:2, :2] = R
    a64 = a[:, :3].astype(np.float64)
    ra = np.stack([(R3[r, 0] * a64[:, 0] + R3[r, 1] * a64[:, 1]) + R3[r, 2] * a64[:, 2] for r in range(3)], axis=1)
    tt = [qo.cote_estimate(b[:, r].astype(np.float64) - ra[:, r], 0.3, True)[0] for r in range(3)]
    assert [float(x) for x in lines[4].split()[1:]] == tt
    # reg_name "TEASER": stage method, then the whole back end
    b64 = b[:, :3].astype(np.float64)
    R3o, _, _, _ = qo.gnc_rotation3d(a64, b64, 0.3)
    assert [float(x) for x in lines[5].split()[1:]] == R3o.ravel().tolist()
    o3 = qo.solve(vs[corr[:, 0]], vt[corr[:, 1]], qo.default_params(reg_mode=1))
    assert [float(x) for x in lines[6].split()[1:]] == o3["T"].ravel().tolist()


# ------------------------------------------------------------- (f)1 range image + sub-cluster rejection
@pytest.mark.parametrize("lidar,mode,minpts", [("Velodyne-64-HDE", "4CrossNeighbor", 30), ("Velodyne-64-HDE", "4Neighbor", 30),
                                               ("Velodyne-64-HDE", "8Neighbor", 10), ("Ouster-OS1-64", "4CrossNeighbor", 30),
                                               ("VLP-16", "4Neighbor", 30), ("HDL-32E", "8Neighbor", 30)])
def test_segment_cloud_matches_oracle(hip, qo, lidar, mode, minpts):
    """qtr_segment_cloud (ImageProjection::segmentCloud, "Patchwork" mode): label image, valid segments and
    rejected sub-clusters bit-identical to the oracle's breadth-first restatement."""
    for pid in (0, 3):
        s, t, _ = synth.kitti64_pair(pid)
        for cloud in (s, t):
            o = qo.segment_cloud(cloud, qo.ip_params(lidar, mode, minpts))
            g = hip.segment_cloud(cloud, ql.ip_params(lidar, mode, minpts))
            assert np.array_equal(g["labels"], o["labels"])
            assert np.array_equal(_b(g["valid"]), _b(o["valid"]))
            assert np.array_equal(_b(g["outliers"]), _b(o["outliers"]))
            assert g["n_segments"] == (o["labels"][(o["labels"] > 0) & (o["labels"] < 999999)].max(initial=0))


def test_segment_cloud_edge_cases_and_pipeline(hip, qo):
    ipp = ql.ip_params()
    r = hip.segment_cloud(np.zeros((0, 4), dtype=np.float32), ipp)
    assert r["valid"].shape[0] == 0 and r["outliers"].shape[0] == 0 and (r["labels"] == -1).all()
    pts = np.array([[10, 0, 0, 0], [0.01, 0.01, 0, 0], [1, 0, 5, 0]], dtype=np.float32)
    r = hip.segment_cloud(pts, ipp)
    assert r["valid"].shape[0] == 0 and r["outliers"].shape[0] == 1 and (r["labels"] == 999999).sum() == 1
    # many returns in one pixel: the last one in scan order owns it
    dup = np.tile(np.array([[20, 1, 0.5, 0]], dtype=np.float32), (500, 1))
    dup[:, 0] += np.linspace(0, 1e-3, 500, dtype=np.float32)
    o = qo.segment_cloud(dup)
    g = hip.segment_cloud(dup)
    assert np.array_equal(g["labels"], o["labels"]) and np.array_equal(_b(g["outliers"]), _b(o["outliers"]))
    with pytest.raises(ValueError):
        ql.ip_params("no-such-lidar")
    # the reference demo's order: segment -> voxelize -> ... -> transform, against the oracle doing the same
    s, t, _ = synth.kitti64_pair(2)
    vs, vt = hip.segment_cloud(s)["valid"], hip.segment_cloud(t)["valid"]
    os_, ot = qo.segment_cloud(s)["valid"], qo.segment_cloud(t)["valid"]
    _assert_same_solution(hip.register_pair(vs, vt, ql.default_frontend_params(seed=2)), qo.register_pair(os_, ot, seed=2))


def test_cpp_dropin_demo_with_image_projection(hip, qo, tmp_path):
    """The demo's STEP 3 + STEP 4 order (ImageProjection::segmentCloud -> voxelize -> FPFHManager -> Quatro) through
    include/imageProjection.hpp, against the oracle running the same order."""
    import subprocess

    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    s, t, _ = synth.kitti64_pair(1)
    synth.save_kitti_bin(str(tmp_path / "src.bin"), s)
    synth.save_kitti_bin(str(tmp_path / "tgt.bin"), t)
    exe = str(tmp_path / "dropin_demo")
    libdir = os.path.join(root, "quatro_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "dropin_demo.cpp"), "-o", exe, "-L", libdir,
                           "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = None
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
        env = dict(os.environ)
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        p = subprocess.run([exe, str(tmp_path / "src.bin"), str(tmp_path / "tgt.bin"), "5", "segment"], env=env,
                           capture_output=True, text=True, timeout=300)
        if p.returncode == 0:
            out = p.stdout
            break
    assert out is not None, p.stderr[-500:]
    so, to = qo.segment_cloud(s), qo.segment_cloud(t)
    lines = out.strip().splitlines()
    w = lines[0].split()
    nseg = lambda r: int(r["labels"][(r["labels"] > 0) & (r["labels"] < 999999)].max(initial=0))
    assert [int(x) for x in (w[1], w[2], w[4], w[5], w[7], w[8])] == [nseg(so), nseg(to), so["valid"].shape[0],
                                                                   to["valid"].shape[0], so["outliers"].shape[0],
                                                                   to["outliers"].shape[0]]
    vs, vt = so["valid"].copy(), to["valid"].copy()
    vs[:, 3] = 0
    vt[:, 3] = 0
    o = qo.register_pair(vs, vt, seed=5)
    head = dict(zip(lines[1].split()[0::2], lines[1].split()[1::2]))
    assert (int(head["n_src"]), int(head["n_tgt"]), int(head["L"])) == (o["n_src"], o["n_tgt"], o["L"])
    T = np.array([[float(x) for x in ln.split()[1:]] for ln in lines[2:6]])
    assert np.array_equal(T, o["T"])


# ------------------------------------------------------------------------------------------------
# "next" row (f)2: Patchwork ground segmentation
def _pw_equal(g, o):
    assert g["ground"].shape == o["ground"].shape and g["nonground"].shape == o["nonground"].shape
    assert np.array_equal(g["ground"].view(np.uint32), o["ground"].view(np.uint32))
    assert np.array_equal(g["nonground"].view(np.uint32), o["nonground"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("scan_id", [0, 1, 2])
def test_patchwork_matches_oracle(hip, qo, scan_id):
    """qtr_patchwork (PatchWork::estimate_ground, include/patchwork.hpp:329-476): ground and non-ground clouds in the
    reference's output order, bit for bit against the CPU restatement, on raw 64-beam scans with ground."""
    xyzi, is_ground = synth.kitti64_raw_scan(scan_id)
    o = qo.patchwork(xyzi)
    g = hip.patchwork(xyzi)
    _pw_equal(g, o)
    assert g["ground"].shape[0] > 0.5 * is_ground.sum()  # a sane split, not an empty one


@pytest.mark.gpu
def test_patchwork_parameter_variants(hip, qo):
    xyzi, _ = synth.kitti64_raw_scan(3)
    for mut in ("global_thr", "no_margin", "one_iter", "small_lpr", "thr8", "two_zones"):
        po, pg = qo.pw_params(), ql.pw_params()
        for p in (po, pg):
            if mut == "global_thr":
                p.using_global_thr, p.global_elevation_thr = 1, -1.4
            elif mut == "no_margin":
                p.sensor_height = 0.0
            elif mut == "one_iter":
                p.num_iter = 1
            elif mut == "small_lpr":
                p.num_lpr, p.num_min_pts = 5, 10
            elif mut == "thr8":
                p.num_thr = 8
                p.elevation_thr[:] = [-1.2, -0.9984, -0.851, -0.605, -0.5, -0.4, -0.3, -0.2]
                p.flatness_thr[:] = [1e-4, 1.25e-4, 1.85e-4, 1.85e-4, 2e-4, 2e-4, 3e-4, 3e-4]
            elif mut == "two_zones":
                p.num_zones = 2
                p.num_sectors_each_zone[:] = [16, 32, 0, 0]
                p.num_rings_each_zone[:] = [2, 4, 0, 0]
                p.min_ranges[:] = [2.7, 12.3625, 0, 0]
                p.max_range = 40.0
        _pw_equal(hip.patchwork(xyzi, pg), qo.patchwork(xyzi, po))


@pytest.mark.gpu
def test_patchwork_edge_cases(hip, qo):
    r = hip.patchwork(np.zeros((0, 4), dtype=np.float32))
    assert r["ground"].shape[0] == 0 and r["nonground"].shape[0] == 0
    rng = np.random.default_rng(5)
    # everything inside min_range / beyond max_range: no patch receives a point, both outputs are empty (the
    # reference drops such points, patchwork.hpp:541-560)
    near = (rng.standard_normal((500, 4)) * 0.5).astype(np.float32)
    far = near.copy()
    far[:, 0] += 500.0
    for pts in (near, far):
        _pw_equal(hip.patchwork(pts), qo.patchwork(pts))
    # a handful of points (every patch below num_min_pts), duplicates, and one dense flat patch
    few = np.zeros((40, 4), dtype=np.float32)
    few[:, 0] = np.linspace(5, 60, 40)
    few[:, 2] = -1.7
    _pw_equal(hip.patchwork(few), qo.patchwork(few))
    dup = np.repeat(few[:4], 100, axis=0)
    _pw_equal(hip.patchwork(dup), qo.patchwork(dup))
    flat = np.zeros((5000, 4), dtype=np.float32)
    flat[:, 0] = rng.uniform(5.0, 8.0, 5000)
    flat[:, 1] = rng.uniform(-0.3, 0.3, 5000)
    flat[:, 2] = -1.723 + rng.normal(0, 0.01, 5000)
    flat[:, 3] = rng.uniform(0, 1, 5000)
    _pw_equal(hip.patchwork(flat), qo.patchwork(flat))
    # inconsistent parameters are refused like check_input_parameters_are_correct (:588-614)
    bad = ql.pw_params()
    bad.min_range = 3.0  # != min_ranges[0]
    with pytest.raises(Exception):
        hip.patchwork(few, bad)
    bad = ql.pw_params()
    bad.num_zones = 5
    with pytest.raises(Exception):
        hip.patchwork(few, bad)


@pytest.mark.gpu
def test_raw_scan_pipeline_matches_oracle(hip, qo):
    """The reference demo's whole order on raw scans: Patchwork -> ImageProjection -> voxelize -> FPFH -> Quatro."""
    a, _ = synth.kitti64_raw_scan(0)
    b, _ = synth.kitti64_raw_scan(1)
    outs = []
    for be in (hip, qo):
        clouds = []
        for raw in (a, b):
            ng = be.patchwork(raw)["nonground"]
            clouds.append(be.segment_cloud(ng)["valid"])
        outs.append(clouds)
    for cg, co in zip(outs[0], outs[1]):
        assert np.array_equal(cg.view(np.uint32), co.view(np.uint32))
    _assert_same_solution(hip.register_pair(outs[0][0], outs[0][1], ql.default_frontend_params(seed=4)),
                          qo.register_pair(outs[1][0], outs[1][1], seed=4))


@pytest.mark.gpu
def test_cpp_dropin_demo_raw_scans(hip, qo, tmp_path):
    """The demo's STEP 2-4 on raw scans through include/patchwork.hpp + imageProjection.hpp + quatro.hpp
    (PatchWork::estimate_ground -> ImageProjection::segmentCloud -> voxelize -> FPFHManager -> Quatro)."""
    import subprocess

    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    a, _ = synth.kitti64_raw_scan(0)
    b, _ = synth.kitti64_raw_scan(1)
    a[:, 3] = 0  # the demo's getCloud keeps x, y, z only
    b[:, 3] = 0
    synth.save_kitti_bin(str(tmp_path / "src.bin"), a)
    synth.save_kitti_bin(str(tmp_path / "tgt.bin"), b)
    exe = str(tmp_path / "dropin_demo")
    libdir = os.path.join(root, "quatro_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "dropin_demo.cpp"), "-o", exe, "-L", libdir,
                           "-lquatro_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = None
    for extra in ("", os.path.join(os.path.dirname(torch.__file__), "lib")):
        env = dict(os.environ)
        if extra:
            env["LD_LIBRARY_PATH"] = extra + ":" + env.get("LD_LIBRARY_PATH", "")
        p = subprocess.run([exe, str(tmp_path / "src.bin"), str(tmp_path / "tgt.bin"), "4", "raw"], env=env,
                           capture_output=True, text=True, timeout=300)
        if p.returncode == 0:
            out = p.stdout
            break
    assert out is not None, p.stderr[-500:]
    pa, pb = qo.patchwork(a), qo.patchwork(b)
    lines = out.strip().splitlines()
    w = lines[0].split()
    assert [int(w[1]), int(w[2]), int(w[4]), int(w[5])] == [pa["ground"].shape[0], pb["ground"].shape[0],
                                                            pa["nonground"].shape[0], pb["nonground"].shape[0]]
    so, to = qo.segment_cloud(pa["nonground"]), qo.segment_cloud(pb["nonground"])
    w = lines[1].split()
    assert [int(w[4]), int(w[5])] == [so["valid"].shape[0], to["valid"].shape[0]]
    vs, vt = so["valid"].copy(), to["valid"].copy()
    vs[:, 3] = 0
    vt[:, 3] = 0
    o = qo.register_pair(vs, vt, seed=4)
    head = dict(zip(lines[2].split()[0::2], lines[2].split()[1::2]))
    assert (int(head["n_src"]), int(head["n_tgt"]), int(head["L"])) == (o["n_src"], o["n_tgt"], o["L"])
    T = np.array([[float(x) for x in ln.split()[1:]] for ln in lines[3:7]])
    assert np.array_equal(T, o["T"])


# ------------------------------------------------------------------------------------------------
# "next" row (f)4: PMC_EXACT
@pytest.mark.gpu
@pytest.mark.parametrize("L,p,planted,seed", [(1, 0.0, 0, 0), (2, 1.0, 0, 0), (60, 0.3, 0, 1), (120, 0.2, 10, 2),
                                              (200, 0.1, 12, 3), (150, 0.5, 0, 4), (300, 0.3, 25, 5), (400, 0.05, 0, 6),
                                              (90, 0.7, 0, 7), (1000, 0.1, 30, 8), (5000, 0.02, 40, 9)])
def test_exact_max_clique_matches_oracle(hip, qo, L, p, planted, seed):
    """qtr_max_clique in PMC_EXACT mode (reference src/graph.cc:106-127): the same clique as the CPU restatement's
    sequential branch and bound (the heuristic's when it is maximum, else the first maximum clique in the canonical
    depth-first order), and a clique of the input graph."""
    bm, A = _random_graph_bitmap(L, p, seed, planted)
    ref = np.sort(qo.max_clique(bm, 0))
    got, _ = hip.max_clique(bm, 0)
    assert np.array_equal(got, ref)
    heu, _ = hip.max_clique(bm, 1)
    assert got.size >= heu.size and got.size >= planted
    if got.size == heu.size:
        assert np.array_equal(got, heu)
    if got.size > 1:
        sub = A[np.ix_(got, got)]
        assert sub.sum() == got.size * (got.size - 1)
    st = hip.exact_stats()
    assert not st["aborted"]


@pytest.mark.gpu
def test_exact_max_clique_is_maximum_vs_networkx(hip):
    nx = pytest.importorskip("networkx")
    for seed, (L, p) in enumerate([(80, 0.4), (150, 0.3), (250, 0.15), (64, 0.8)]):
        bm, A = _random_graph_bitmap(L, p, 100 + seed)
        got, _ = hip.max_clique(bm, 0)
        omega = max(len(c) for c in nx.find_cliques(nx.from_numpy_array(A.astype(int))))
        assert got.size == omega


@pytest.mark.gpu
def test_exact_mode_through_the_solver(hip, qo):
    """Quatro with INLIER_SELECTION_MODE::PMC_EXACT: whole back end against the oracle, on an easy case (heuristic
    already maximum) and on one where the exact search replaces the clique before the estimation."""
    for L, frac, seed in ((300, 0.3, 1), (800, 0.1, 2), (2000, 0.05, 3)):
        src, tgt, _, _ = synth.correspondences(L, frac, seed=seed, noise=0.05)
        _assert_same_solution(hip.solve(src, tgt, ql.demo_params(inlier_selection_mode=ql.INLIER_PMC_EXACT)),
                              qo.solve(src, tgt, qo.default_params(inlier_selection_mode=0)))
    # points scattered inside a box the size of a few noise bounds: a dense, random-looking consistency graph on which
    # the heuristic is beaten, so the estimation runs a second time from the exact clique
    for L, size, seed in ((120, 1.5, 1), (200, 2.0, 3), (250, 2.5, 5), (180, 1.8, 6)):
        rng = np.random.default_rng(seed)
        src = np.zeros((L, 4), dtype=np.float32)
        tgt = np.zeros((L, 4), dtype=np.float32)
        src[:, :3] = rng.uniform(0, size, (L, 3))
        tgt[:, :3] = rng.uniform(0, size, (L, 3))
        g = hip.solve(src, tgt, ql.demo_params(inlier_selection_mode=ql.INLIER_PMC_EXACT))
        o = qo.solve(src, tgt, qo.default_params(inlier_selection_mode=0))
        _assert_same_solution(g, o)
        assert g["clique"].size > hip.solve(src, tgt)["clique"].size


@pytest.mark.gpu
def test_exact_time_limit_returns_heuristic(hip, qo):
    """Params::max_clique_time_limit: a dense random graph with an absurdly small limit comes back with the heuristic
    clique and the abort flag, quickly."""
    bm, _ = _random_graph_bitmap(600, 0.6, 11)
    hip.set_clique_time_limit(1e-4)
    try:
        got, _ = hip.max_clique(bm, 0)
        st = hip.exact_stats()
    finally:
        hip.set_clique_time_limit(3600.0)
    heu, _ = hip.max_clique(bm, 1)
    assert st["aborted"] and np.array_equal(got, heu)


# ------------------------------------------------------------------------------------------------
# "next" row (f)4, second half: 3-DoF rotation (reg_name "TEASER")
def _planted_3dof(L, frac, seed, noise=0.02):
    from scipy.spatial.transform import Rotation as Rt
    rng = np.random.default_rng(seed)
    src = np.zeros((L, 4), dtype=np.float32)
    src[:, :3] = rng.uniform(-20, 20, (L, 3))
    Rm = Rt.from_euler("zyx", rng.uniform(-60, 60, 3), degrees=True).as_matrix()
    tv = rng.uniform(-3, 3, 3)
    tgt = np.zeros((L, 4), dtype=np.float32)
    tgt[:, :3] = src[:, :3] @ Rm.T + tv + rng.normal(0, noise, (L, 3))
    out = rng.random(L) >= frac
    tgt[out, :3] = rng.uniform(-20, 20, (int(out.sum()), 3))
    return src, tgt, Rm, tv


@pytest.mark.gpu
@pytest.mark.parametrize("M,seed", [(1, 0), (2, 1), (63, 2), (64, 3), (200, 4), (1000, 5), (5000, 6)])
def test_gnc_rotation3d_matches_oracle(hip, qo, M, seed):
    """qtr_gnc_rotation3d (teaser::utils::svdRot inside TEASER++'s GNC-TLS loop): R, cost, iteration count and inlier
    mask bit for bit against the CPU restatement; R within 1e-4 rad of the planted rotation when there is one."""
    from scipy.spatial.transform import Rotation as Rt
    rng = np.random.default_rng(seed)
    X = rng.uniform(-10, 10, (M, 3))
    Rm = Rt.from_euler("zyx", rng.uniform(-90, 90, 3), degrees=True).as_matrix()
    Y = X @ Rm.T + rng.normal(0, 0.02, (M, 3))
    bad = rng.random(M) < 0.3
    Y[bad] = rng.uniform(-10, 10, (int(bad.sum()), 3))
    for nb, it in ((0.6, 50), (0.1, 100), (0.6, 1)):
        Ro, co, io, mo = qo.gnc_rotation3d(X, Y, nb, 1.4, it, 1.1e-4)
        Rg, cg, ig, mg = hip.gnc_rotation3d(X, Y, nb, 1.4, it, 1.1e-4)
        assert np.array_equal(Rg, Ro) and ig == io and np.array_equal(mg, mo)
        assert cg == co or (np.isinf(cg) and np.isinf(co))
    if M >= 200:
        ang = np.arccos(np.clip((np.trace(Rg.T @ Rm) - 1) / 2, -1, 1))
        Rg, *_ = hip.gnc_rotation3d(X, Y, 0.6, 1.4, 50, 1.1e-4)
        assert np.arccos(np.clip((np.trace(Rg.T @ Rm) - 1) / 2, -1, 1)) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("L,frac,seed", [(300, 0.4, 1), (1000, 0.2, 2), (3000, 0.1, 3), (12000, 0.3, 4)])
def test_teaser_mode_through_the_solver(hip, qo, L, frac, seed):
    """qtr_solve with reg_mode = QTR_REG_TEASER: clique, rotation inliers, 3-DoF rotation, COTE translation against the
    oracle, and the planted 6-DoF transform recovered (which the yaw-only mode cannot do)."""
    src, tgt, Rm, tv = _planted_3dof(L, frac, seed)
    g = hip.solve(src, tgt, ql.demo_params(reg_mode=ql.REG_TEASER))
    o = qo.solve(src, tgt, qo.default_params(reg_mode=1))
    _assert_same_solution(g, o)
    assert g["valid"]
    ang = np.arccos(np.clip((np.trace(g["T"][:3, :3].T @ Rm) - 1) / 2, -1, 1))
    assert ang < 5e-3 and np.abs(g["T"][:3, 3] - tv).max() < 0.05
    with pytest.raises(ql.QuatroHipError) as e:
        hip.solve(src, tgt, ql.demo_params(reg_mode=ql.REG_TEASER, using_pre_estimated_ryrx=1))
    assert e.value.code == ql.QTR_ERR_BAD_ARG
    with pytest.raises(ql.QuatroHipError):
        hip.solve(src, tgt, ql.demo_params(reg_mode=7))


# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_phase_counters_are_never_stale_across_calls(hip, qo):
    """Alternating registrations of very different sizes on one handle: every call must report ITS OWN voxel counts and
    correspondence count.  The phase-ending kernels hand these counters to the host through a pinned mailbox; before the
    payload was tagged, about one phase in 10^4 let the host run ahead with the previous call's numbers (found by
    tests/gpu_fuzz.py on a 5 % subsample of a scan pair, which is what the small input here is)."""
    rng = np.random.default_rng(2024)
    s, t, _ = synth.kitti64_pair(1)
    s, t = s[rng.random(s.shape[0]) < 0.05], t[rng.random(t.shape[0]) < 0.05]
    fp = ql.default_frontend_params(seed=1)
    o = qo.register_pair(s, t, seed=1)
    want = (o["n_src"], o["n_tgt"], o["L"])
    S, T, _ = synth.kitti64_pair(0)
    big = hip.register_pair(S, T, ql.default_frontend_params(seed=3))
    want_big = (big["n_src"], big["n_tgt"], big["L"])
    t_end = time.time() + 6.0
    n = 0
    while time.time() < t_end:
        r = hip.register_pair(s, t, fp)
        assert (r["n_src"], r["n_tgt"], r["L"]) == want, n
        b = hip.register_pair(S, T, ql.default_frontend_params(seed=3))
        assert (b["n_src"], b["n_tgt"], b["L"]) == want_big, n
        n += 1
    assert n > 500
    _assert_same_solution(hip.register_pair(s, t, fp), o)


@pytest.mark.gpu
def test_voxel_sort_pass_speculation_recovers(hip, qo):
    """The whole-path driver launches as many radix passes for the voxel sort as the previous pair on the slot needed
    (qtr_register_pair, capi.hip) and runs the stage again when a pair needs more.  A scan pair on a grid below 2^24
    cells (3 passes) is registered until the driver has stepped down, then the same pair with two far outliers that
    stretch the grid past 2^24 cells (4 passes), then the first one again: every record equals the oracle's."""
    S, T, _ = synth.kitti64_pair(0)
    rng = np.random.default_rng(5)
    keep = rng.random(S.shape[0]) < 0.25
    S, T = S[keep], T[rng.random(T.shape[0]) < 0.25]
    far = np.array([[900.0, -900.0, 3.0, 0.0], [-900.0, 900.0, -2.0, 0.0]], dtype=np.float32)
    Sw, Tw = np.concatenate([S, far]), np.concatenate([T, far])
    span = Sw[:, :3].max(0) - Sw[:, :3].min(0)
    assert np.prod(np.floor(span / 0.3) + 1) > 2 ** 24 > np.prod(np.floor((S[:, :3].max(0) - S[:, :3].min(0)) / 0.3) + 1)
    fp = ql.default_frontend_params(seed=11)
    o_narrow, o_wide = qo.register_pair(S, T, seed=11), qo.register_pair(Sw, Tw, seed=11)
    for _ in range(7):  # (the driver steps down after four calls that needed fewer passes than it launched)
        _assert_same_solution(hip.register_pair(S, T, fp), o_narrow)
    for _ in range(2):
        _assert_same_solution(hip.register_pair(Sw, Tw, fp), o_wide)
        _assert_same_solution(hip.register_pair(S, T, fp), o_narrow)


@pytest.mark.gpu
def test_randomised_sweep_against_oracle():
    """Ten seconds of tests/gpu_fuzz.py (random solver / clique / pair / raw-scan stage / stand-alone stage cases against
    the oracle) in its own process; any mismatch fails."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_fuzz.py"), "5", "10"], cwd=root, capture_output=True,
                       text=True, timeout=280)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-500:])
    assert "0 mismatches" in p.stdout


# ---------------------------------------------------------------------------------------------- batched path
def _seq_results(h, pairs):
    return [h.register_pair(s, t, ql.default_frontend_params(seed=seed), slot=0) for (s, t, seed) in pairs]


def _assert_same_record(a, b):
    assert a["status"] == b["status"] and a["valid"] == b["valid"]
    assert (a["n_src"], a["n_tgt"], a["L"]) == (b["n_src"], b["n_tgt"], b["L"])
    assert np.array_equal(a["clique"], b["clique"]) and np.array_equal(a["final_inliers"], b["final_inliers"])
    assert np.array_equal(a["T"], b["T"]) and (a["cost"] == b["cost"] or (np.isinf(a["cost"]) and np.isinf(b["cost"])))


def test_batch_is_bit_identical_to_sequential_runs():
    """qtr_submit_batch / qtr_wait (groups of pairs per launch chain, two lanes) against one qtr_register_pair per pair:
    every record, clique and inlier list identical.  11 pairs on 8 slots: full chunks, a ragged last chunk, both lanes."""
    pairs = []
    for i in range(11):
        s, t, _ = synth.kitti64_pair(i % 4) if i % 3 else synth.kitti64_pair(4 + i % 2, n_boxes=60)
        if i == 5:
            s = s[: s.shape[0] // 3]       # a much smaller pair inside a group
        pairs.append((s, t, 100 + i))
    h1 = ql.Handle(0)
    seq = _seq_results(h1, pairs)
    h1.close()
    hb = ql.Handle(0, n_slots=8, max_points=131072, max_voxels=32768, max_corr=8192)
    try:
        got = hb.register_batch(pairs)
        assert len(got) == len(seq)
        for a, b in zip(got, seq):
            _assert_same_record(a, b)
        again = hb.register_batch(pairs[:3])   # the handle is reusable, a batch smaller than one lane
        for a, b in zip(again, seq[:3]):
            _assert_same_record(a, b)
        one = hb.register_pair(*pairs[2][:2], ql.default_frontend_params(seed=pairs[2][2]), slot=3)  # slots still work
        _assert_same_record(one, seq[2])
    finally:
        hb.close()


def test_batch_256_pair_ids_agree_with_sequential_runs():
    """BASELINE configs[2]: 256 pair ids streamed through one GPU (pool of 8 distinct synthetic pairs, distinct tuple-test
    seeds per id): the batched records equal the sequential ones for every id."""
    pool = [synth.kitti64_pair(i) for i in range(8)]
    ids = list(range(256))
    h1 = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=8192)
    # sequential reference: one run per distinct (pool pair, seed) — seeds cycle with period 16
    ref = {}
    for i in ids:
        key = (i % 8, i % 16)
        if key not in ref:
            s, t, _ = pool[key[0]]
            ref[key] = h1.register_pair(s, t, ql.default_frontend_params(seed=key[1]))
    h1.close()
    hb = ql.Handle(0, n_slots=32, max_points=131072, max_voxels=32768, max_corr=8192)
    try:
        got = hb.register_batch([(pool[i % 8][0], pool[i % 8][1], i % 16) for i in ids])
        for i, g in zip(ids, got):
            _assert_same_record(g, ref[(i % 8, i % 16)])
    finally:
        hb.close()


def test_batch_reports_per_pair_failures_and_keeps_going():
    """A pair that exceeds a limit gets its own QTR_ERR_CAPACITY; the rest of the batch is unaffected."""
    good = synth.kitti64_pair(1)
    big = np.zeros((70000, 4), dtype=np.float32)
    big[:, :3] = np.random.default_rng(1).uniform(-40, 40, size=(70000, 3))
    hb = ql.Handle(0, n_slots=4, max_points=65536, max_voxels=32768, max_corr=8192)
    h1 = ql.Handle(0)
    try:
        ref = h1.register_pair(good[0], good[1], ql.default_frontend_params(seed=9))
        got = hb.register_batch([(good[0], good[1], 9), (big, good[1], 1), (good[0], good[1], 9)])
        assert got[1]["status"] == ql.QTR_ERR_CAPACITY
        _assert_same_record(got[0], ref)
        _assert_same_record(got[2], ref)
    finally:
        hb.close()
        h1.close()


def test_rccl_gather_of_result_records_through_the_c_abi():
    """qtr_comm_unique_id / qtr_comm_init / qtr_gather_results on a communicator of one rank (the box has one GPU): the
    RCCL plumbing — run-time loading, communicator, device staging, ncclAllGather — returns the records unchanged."""
    import ctypes as C
    h = ql.Handle(0)
    try:
        h.comm_init(ql.comm_unique_id(), 0, 1)
        res = (ql.Result * 3)()
        for i in range(3):
            res[i].status = i
            res[i].valid = 1
            res[i].cost = 0.25 * i
            res[i].n_clique = 10 + i
            for k in range(16):
                res[i].T[k] = i + 0.5 * k
        out = h.gather_results(res, 1)
        assert bytes(C.string_at(C.addressof(out), C.sizeof(res))) == bytes(C.string_at(C.addressof(res), C.sizeof(res)))
        # the variable-length form (uneven block partitions): counts first, then the records; a rank may hold none
        out2, counts, n_all = h.gather_results_v(res, 2, 1, 8)
        assert counts == [2] and n_all == 2
        assert bytes(C.string_at(C.addressof(out2), 2 * C.sizeof(ql.Result))) == bytes(C.string_at(C.addressof(res), 2 * C.sizeof(ql.Result)))
        _, counts, n_all = h.gather_results_v(None, 0, 1, 8)
        assert counts == [0] and n_all == 0
        with pytest.raises(ql.QuatroHipError) as ei:   # more records than the caller made room for: refused, not overrun
            h.gather_results_v(res, 3, 1, 2)
        assert ei.value.code == ql.QTR_ERR_CAPACITY
        src, tgt, _, _ = synth.correspondences(300, 0.2, seed=1, noise=0.2)   # the handle still registers afterwards
        assert h.solve(src, tgt)["valid"]
    finally:
        h.close()
