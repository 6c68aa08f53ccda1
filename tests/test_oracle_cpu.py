"""The oracle restates un-vendored third-party algorithms (PCL / FLANN / PMC / Eigen) and the reference has
no golden vectors (PARITY UNPINNED).  These tests pin the oracle against INDEPENDENT
implementations and closed-form properties instead (SURVEY.md §8c "what stands in for golden vectors")."""
import networkx as nx
import numpy as np
import pytest

from quatro_amd import synth


def _cloud(n, seed, extent=6.0):
    rng = np.random.default_rng(seed)
    c = np.zeros((n, 4), dtype=np.float32)
    c[:, :3] = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    c[:, 2] *= 0.2
    return c


def _bitmap_to_dense(bm, L):
    bits = np.unpackbits(bm.view(np.uint8), axis=1, bitorder="little")[:, :L]
    return bits.astype(bool)


# ------------------------------------------------------------------------------------------- K1
def test_voxelize_matches_independent_numpy(qo):
    rng = np.random.default_rng(0)
    pts = np.zeros((3000, 4), dtype=np.float32)
    pts[:, :3] = rng.normal(0, 3, (3000, 3)).astype(np.float32)
    leaf = np.float32(0.3)
    out = qo.voxelize(pts, float(leaf))
    inv = np.float32(1.0) / leaf
    mn, mx = pts[:, :3].min(0), pts[:, :3].max(0)
    minb = np.floor(mn * inv).astype(np.int64)
    maxb = np.floor(mx * inv).astype(np.int64)
    div = maxb - minb + 1
    ijk = (np.floor(pts[:, :3] * inv) - minb.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.lexsort((np.arange(len(idx)), idx))
    exp = []
    i = 0
    while i < len(order):
        j = i
        acc = np.zeros(3, dtype=np.float32)
        while j < len(order) and idx[order[j]] == idx[order[i]]:
            acc = acc + pts[order[j], :3]
            j += 1
        exp.append(acc / np.float32(j - i))
        i = j
    exp = np.array(exp, dtype=np.float32)
    assert out.shape[0] == exp.shape[0]
    assert np.array_equal(out[:, :3].view(np.uint32), exp.view(np.uint32))
    # output order = ascending voxel index; every centroid lies in its voxel
    assert np.all(out[:, 3] == 0)


def test_voxelize_edge_cases(qo):
    one = np.array([[1.0, 2.0, 3.0, 9.0]], dtype=np.float32)
    assert np.array_equal(qo.voxelize(one, 0.3)[:, :3], one[:, :3])
    dup = np.repeat(one, 7, axis=0)
    assert qo.voxelize(dup, 0.3).shape[0] == 1
    # leaf far too small for the extent -> PCL passes the input through
    far = np.array([[0, 0, 0, 0], [1e5, 1e5, 1e5, 0]], dtype=np.float32)
    assert qo.voxelize(far, 0.001).shape[0] == 2


# ------------------------------------------------------------------------------------------- radius search
def test_radius_neighbors_vs_bruteforce(qo):
    c = _cloud(1200, 1)
    off, idx, d2 = qo.radius_neighbors(c, 0.75)
    r2 = np.float32(0.75 * 0.75)
    p = c[:, :3]
    for i in range(0, 1200, 37):
        dx = p[i, 0] - p[:, 0]
        dy = p[i, 1] - p[:, 1]
        dz = p[i, 2] - p[:, 2]
        dd = ((dx * dx) + dy * dy) + dz * dz  # float32, L2_Simple order
        sel = np.nonzero(dd < r2)[0]
        order = sel[np.lexsort((sel, dd[sel]))]
        got = idx[off[i]:off[i + 1]]
        assert np.array_equal(got, order)
        assert np.array_equal(d2[off[i]:off[i + 1]].view(np.uint32), dd[order].view(np.uint32))
        assert got[0] == i  # the query itself, distance 0


# ------------------------------------------------------------------------------------------- K2-K4
def test_normals_close_to_eigh_and_oriented(qo):
    rng = np.random.default_rng(2)
    n = 1500
    c = np.zeros((n, 4), dtype=np.float32)
    c[:, 0] = rng.uniform(2, 8, n)
    c[:, 1] = rng.uniform(-3, 3, n)
    c[:, 2] = 0.3 * c[:, 0] + 0.02 * rng.normal(size=n)  # a tilted plane
    nrm, sp, de = qo.fpfh(c, 0.5, 0.75)
    ok = ~np.isnan(nrm[:, 0])
    assert ok.sum() > 0.5 * n
    v = nrm[ok, :3]
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-4)
    true_n = np.array([-0.3, 0, 1.0]) / np.linalg.norm([-0.3, 0, 1.0])
    cosang = np.abs(v @ true_n)
    assert np.median(cosang) > 0.98
    # flipped toward the viewpoint (origin): n . (0 - p) >= 0
    assert np.all(np.einsum("ij,ij->i", v, -c[ok, :3]) >= -1e-5)
    # curvature = lambda0 / trace in [0, 1/3]
    assert np.all((nrm[ok, 3] >= 0) & (nrm[ok, 3] <= 1.0 / 3 + 1e-3))


def _ref_fpfh_float64(xyz, rn, rf):
    """pcl::NormalEstimation + FPFHEstimation restated from SURVEY.md Appendix A.2 in plain float64 numpy (O(n^2))."""
    P = xyz.astype(np.float64)
    n = P.shape[0]
    D2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
    normals = np.full((n, 3), np.nan)
    for i in range(n):
        nb = np.nonzero(D2[i] <= rn * rn)[0]
        if nb.size < 3:
            continue
        Q = P[nb]
        m = Q.mean(0)
        C = (Q[:, :, None] * Q[:, None, :]).mean(0) - np.outer(m, m)
        w, V = np.linalg.eigh(C)
        v = V[:, 0]
        if np.dot(v, -P[i]) < 0:
            v = -v
        normals[i] = v
    spfh = np.zeros((n, 33))
    nbrs = []
    for i in range(n):
        nb = np.nonzero(D2[i] <= rf * rf)[0]
        nbrs.append(nb)
        k = nb.size
        if k < 2:
            continue
        incr = 100.0 / (k - 1)
        for q in nb:
            if q == i:
                continue
            d = P[q] - P[i]
            f4 = np.linalg.norm(d)
            if f4 == 0:
                continue
            n_p, n_q = normals[i], normals[q]
            a1 = np.dot(n_p, d) / f4
            a2 = np.dot(n_q, d) / f4
            if np.arccos(abs(a1)) > np.arccos(abs(a2)):
                n1, n2, dd, f3 = n_q, n_p, -d, -a2
            else:
                n1, n2, dd, f3 = n_p, n_q, d, a1
            v = np.cross(dd, n1)
            nv = np.linalg.norm(v)
            if nv == 0:
                continue
            v = v / nv
            w = np.cross(n1, v)
            f2 = np.dot(v, n2)
            f1 = np.arctan2(np.dot(w, n2), np.dot(n1, n2))
            feats = [(f1 + np.pi) / (2 * np.pi), (f2 + 1) / 2, (f3 + 1) / 2]
            if any(np.isnan(feats)):
                bins = [0, 0, 0]
            else:
                bins = [min(10, max(0, int(np.floor(11 * f)))) for f in feats]
            for b, bb in enumerate(bins):
                spfh[i, 11 * b + bb] += incr
    fpfh = np.zeros((n, 33))
    for i in range(n):
        for q in nbrs[i]:
            if D2[i, q] == 0:
                continue
            fpfh[i] += spfh[q] / D2[i, q]
        for b in range(3):
            s = fpfh[i, 11 * b:11 * b + 11].sum()
            if s != 0:
                fpfh[i, 11 * b:11 * b + 11] *= 100.0 / s
    return normals, spfh, fpfh


def test_normals_spfh_fpfh_against_independent_float64_restatement(qo):
    """The oracle's K2-K4 (float32, PCL's operation order) against an independent float64 numpy restatement of the same
    published algorithm: identical NaN pattern, normals within 1e-3 rad, SPFH rows identical except where a pair feature
    sits within float32 reach of a bin edge or of the p/q role swap (then one or two neighbours' mass moves between
    bins), FPFH within the mass that propagates from those rows."""
    rng = np.random.default_rng(0)
    n = 350
    c = np.zeros((n, 4), dtype=np.float32)
    c[:, 0] = rng.uniform(6, 10, n)
    c[:, 1] = rng.uniform(-2, 2, n)
    c[:, 2] = 0.3 * np.sin(c[:, 0]) + 0.2 * np.cos(1.5 * c[:, 1]) + 0.01 * rng.normal(size=n)
    no, so, fo = qo.fpfh(c, 0.5, 0.75)
    nr, sr, fr = _ref_fpfh_float64(c[:, :3], 0.5, 0.75)
    ok = ~np.isnan(nr[:, 0])
    assert np.array_equal(np.isnan(no[:, 0]), ~ok) and ok.sum() > 0.9 * n
    dots = np.einsum("ij,ij->i", no[ok, :3].astype(np.float64), nr[ok])
    assert np.all(dots > 1 - 1e-6)  # same orientation, < 1.5e-3 rad apart (single-pass float32 covariance)
    D2 = ((c[:, None, :3].astype(np.float64) - c[None, :, :3].astype(np.float64)) ** 2).sum(-1)
    k = (D2 <= 0.75 ** 2).sum(1)
    ds = np.abs(so.astype(np.float64) - sr).sum(1)
    assert (ds < 1e-3).mean() > 0.9
    assert np.all(ds <= 4 * 100.0 / np.maximum(k - 1, 1) + 1e-3)  # at most two neighbours change bins in a row
    assert np.allclose(sr.reshape(n, 3, 11).sum(2)[k > 1], 100.0) and np.allclose(so.reshape(n, 3, 11).sum(2)[k > 1], 100.0, atol=1e-2)
    df = np.abs(fo.astype(np.float64) - fr).sum(1)
    assert np.median(df) < 0.5 and df.max() < 15.0  # out of 300 per row


def test_fpfh_blocks_sum_to_100(qo):
    s, t, _ = synth.kitti64_pair(5)
    v = qo.voxelize(s, 0.3)[:2500]
    nrm, sp, de = qo.fpfh(v, 0.5, 0.75)
    assert np.all(de >= 0) and np.all(np.isfinite(de))
    blocks = de.reshape(-1, 3, 11).sum(axis=2)
    nz = blocks > 0
    assert np.allclose(blocks[nz], 100.0, atol=1e-2)
    # SPFH rows: each block sums to 100 when the point has >= 1 usable neighbour
    sb = sp.reshape(-1, 3, 11).sum(axis=2)
    assert np.all((np.abs(sb - 100.0) < 0.05) | (sb == 0))


def test_fpfh_rigid_motion_invariance(qo):
    """FPFH is built from relative angles: descriptors of a rotated+translated copy stay close (float
    rounding and the viewpoint flip aside)."""
    rng = np.random.default_rng(3)
    n = 800
    c = np.zeros((n, 4), dtype=np.float32)
    c[:, 0] = rng.uniform(10, 14, n)
    c[:, 1] = rng.uniform(-2, 2, n)
    c[:, 2] = 0.2 * np.sin(c[:, 0]) + 0.1 * np.cos(2 * c[:, 1])
    _, _, d0 = qo.fpfh(c, 0.5, 0.75)
    R = synth.yaw_matrix(0.3)
    c2 = c.copy()
    c2[:, :3] = (c[:, :3].astype(np.float64) @ R.T + np.array([1.0, 0.5, 0.0])).astype(np.float32)
    _, _, d1 = qo.fpfh(c2, 0.5, 0.75)
    assert np.median(np.abs(d0 - d1).sum(axis=1)) < 20.0  # out of 300 total mass


# ------------------------------------------------------------------------------------------- K5-K8
def test_nn33_vs_sklearn_and_exact_order(qo):
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.default_rng(4)
    A = rng.uniform(0, 100, (400, 33)).astype(np.float32)
    B = rng.uniform(0, 100, (900, 33)).astype(np.float32)
    got = qo.nn33(A, B)
    ref = NearestNeighbors(n_neighbors=1, algorithm="brute").fit(B.astype(np.float64)).kneighbors(
        A.astype(np.float64), return_distance=False)[:, 0]
    assert (got == ref).mean() > 0.995
    # exact flann::L2 accumulation order, float32
    for q in range(0, 400, 41):
        d = np.zeros(900, dtype=np.float32)
        for g in range(8):
            df = A[q, 4 * g:4 * g + 4][None, :] - B[:, 4 * g:4 * g + 4]
            sq = df * df
            d = d + (((sq[:, 0] + sq[:, 1]) + sq[:, 2]) + sq[:, 3])
        t = A[q, 32] - B[:, 32]
        d = d + t * t
        assert got[q] == int(np.argmin(d))


def test_nn33_tie_goes_to_lowest_index(qo):
    B = np.zeros((5, 33), dtype=np.float32)
    B[1] = 1.0
    B[3] = 1.0  # duplicates of row 1
    A = np.ones((1, 33), dtype=np.float32)
    assert qo.nn33(A, B)[0] == 1


def test_match_properties(qo):
    s, t, _ = synth.kitti64_pair(6)
    vs, vt = qo.voxelize(s, 0.3)[:3000], qo.voxelize(t, 0.3)[:2600]
    _, _, ds = qo.fpfh(vs, 0.5, 0.75)
    _, _, dt = qo.fpfh(vt, 0.5, 0.75)
    cross = qo.match(vs, ds, vt, dt, tuple_test=False)
    full = qo.match(vs, ds, vt, dt, seed=9)
    full2 = qo.match(vs, ds, vt, dt, seed=9)
    assert np.array_equal(full, full2)  # deterministic given the seed
    # sorted lexicographically, unique, each index used at most once
    assert np.all(np.diff(cross[:, 0]) > 0) and len(set(cross[:, 1])) == len(cross)
    # mutual nearest neighbours
    nn_t = qo.nn33(ds, dt)
    nn_s = qo.nn33(dt, ds)
    for a, b in cross[::17]:
        assert nn_t[a] == b and nn_s[b] == a
    # the tuple test only removes pairs
    cs = set(map(tuple, cross))
    assert all(tuple(p) in cs for p in full)
    # swapped roles (larger cloud second) give the transposed answer without the tuple test
    crossT = qo.match(vt, dt, vs, ds, tuple_test=False)
    assert set(map(tuple, crossT[:, ::-1])) == cs


def test_match_empty_inputs(qo):
    e4 = np.zeros((0, 4), dtype=np.float32)
    e33 = np.zeros((0, 33), dtype=np.float32)
    c = _cloud(10, 0)
    d = np.ones((10, 33), dtype=np.float32)
    assert qo.match(e4, e33, c, d).shape[0] == 0


# ------------------------------------------------------------------------------------------- K9-K12
def test_graph_predicate_vs_numpy(qo):
    src, tgt, T, inl = synth.correspondences(400, 0.2, seed=1, noise=0.3)
    bm = qo.build_graph(src, tgt, 0.3, 1.0)
    A = _bitmap_to_dense(bm, 400)
    s = src[:, :3].astype(np.float64)
    t = tgt[:, :3].astype(np.float64)
    ds = s[None, :, :] - s[:, None, :]
    dt = t[None, :, :] - t[:, None, :]
    a = np.sqrt(ds[..., 0] ** 2 + (ds[..., 1] ** 2 + ds[..., 2] ** 2))
    b = np.sqrt(dt[..., 0] ** 2 + (dt[..., 1] ** 2 + dt[..., 2] ** 2))
    beta = 2 * 0.3 * np.sqrt(1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        fwd = np.abs(b / a - 1.0) <= beta * (1.0 / a)
        rev = np.abs(a / b - 1.0) <= beta * (1.0 / b)
    exp = fwd & rev
    np.fill_diagonal(exp, False)
    assert np.array_equal(A, exp)
    assert np.array_equal(A, A.T)


def test_kcore_vs_networkx(qo):
    for seed in range(4):
        src, tgt, _, _ = synth.correspondences(300, 0.15, seed=seed, noise=0.3)
        bm = qo.build_graph(src, tgt)
        A = _bitmap_to_dense(bm, 300)
        core, order, mc = qo.kcore(bm)
        G = nx.from_numpy_array(A.astype(int))
        ref = nx.core_number(G)
        assert all(core[v] == ref[v] for v in range(300))
        assert mc == max(ref.values())
        assert sorted(order.tolist()) == list(range(300))
        # BZ order is a valid degeneracy order: core numbers are non-decreasing along it
        assert np.all(np.diff(core[order]) >= 0)


def test_clique_invariants_and_planted_recovery(qo):
    for seed in range(5):
        src, tgt, _, inl = synth.correspondences(600, 0.1, seed=seed, noise=0.05)
        bm = qo.build_graph(src, tgt)
        A = _bitmap_to_dense(bm, 600)
        core, _, mc = qo.kcore(bm)
        for order_mode in (0, 1):
            C = qo.max_clique(bm, 1, 0.5, order_mode)
            assert len(C) >= 2 and np.all(np.diff(C) > 0)
            sub = A[np.ix_(C, C)]
            assert sub.sum() == len(C) * (len(C) - 1)  # a clique
            assert len(C) <= mc + 1
            assert np.all(core[C] >= len(C) - 1)
            # planted inliers with noise << noise bound are mutually consistent and must be found
            assert len(set(inl) - set(C)) == 0
        assert len(qo.max_clique(bm, 1, 0.5, 0)) == len(qo.max_clique(bm, 1, 0.5, 1))


def test_clique_degenerate_graphs(qo):
    # no edges -> empty clique
    bm = np.zeros((10, 1), dtype=np.uint64)
    assert qo.max_clique(bm).size == 0
    # a single edge
    bm[2, 0] = 1 << 7
    bm[7, 0] = 1 << 2
    assert qo.max_clique(bm).tolist() == [2, 7]


# ------------------------------------------------------------------------------------------- K14-K15
def test_rotation_closed_form_matches_svd(qo):
    rng = np.random.default_rng(5)
    for _ in range(20):
        M = 64
        X = rng.normal(size=(M, 2))
        th = rng.uniform(-np.pi, np.pi)
        Rt = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        Y = X @ Rt.T + 0.01 * rng.normal(size=(M, 2))
        R, cost, iters, inl = qo.gnc_rotation2d(X, Y, 10.0, max_iter=1)  # one iteration = plain weighted fit
        H = X.T @ Y  # X W Y^T with W = I (2 x 2)
        U, S, Vt = np.linalg.svd(H)
        V = Vt.T
        if np.linalg.det(U) * np.linalg.det(V) < 0:
            V[:, 1] *= -1
        Rs = V @ U.T  # svdRot2d, reference include/teaser/utils.h:151-166
        assert np.abs(R - Rs).max() < 1e-12
        assert abs(np.arctan2(R[1, 0], R[0, 0]) - th) < 0.01


def test_gnc_rejects_outliers(qo):
    rng = np.random.default_rng(6)
    M = 200
    X = rng.uniform(-20, 20, (M, 2))
    th = 0.7
    Rt = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    Y = X @ Rt.T + rng.uniform(-0.05, 0.05, (M, 2))
    out = rng.choice(M, 60, replace=False)
    Y[out] = rng.uniform(-20, 20, (60, 2))
    R, cost, iters, inl = qo.gnc_rotation2d(X, Y, 0.6)
    assert abs(np.arctan2(R[1, 0], R[0, 0]) - th) < 2e-3
    assert inl.sum() >= M - 60 - 2 and not inl[out].all()
    assert 1 < iters <= 50 and np.isfinite(cost)


def test_cote_majority_and_median(qo):
    rng = np.random.default_rng(7)
    X = np.concatenate([3.0 + rng.uniform(-0.2, 0.2, 80), rng.uniform(-30, 30, 40)])
    est, inl, ncard = qo.cote_estimate(X, 0.3, True)
    assert abs(est - 3.0) < 0.1 and inl[:80].all() and ncard >= 80
    est_w, inl_w, _ = qo.cote_estimate(X, 0.3, False)
    assert abs(est_w - 3.0) < 0.1
    # two measurements: defined behaviour (reference indexes out of range for n_card == 1)
    est2, inl2, nc2 = qo.cote_estimate(np.array([0.0, 10.0]), 0.3, True)
    assert np.isfinite(est2) and nc2 == 1


# ------------------------------------------------------------------------------------------- whole back end
@pytest.mark.parametrize("L,frac,noise", [(200, 0.3, 0.1), (1000, 0.1, 0.3), (2000, 0.05, 0.2)])
def test_solve_recovers_planted_transform(qo, L, frac, noise):
    src, tgt, T, inl = synth.correspondences(L, frac, seed=L, noise=noise)
    r = qo.solve(src, tgt)
    assert r["valid"] and r["status"] == 0
    yaw = np.arctan2(r["T"][1, 0], r["T"][0, 0])
    yaw_gt = np.arctan2(T[1, 0], T[0, 0])
    assert abs(np.arctan2(np.sin(yaw - yaw_gt), np.cos(yaw - yaw_gt))) < 0.02
    assert np.linalg.norm(r["T"][:3, 3] - T[:3, 3]) < 0.35
    assert set(r["final_inliers"]).issubset(set(r["clique"]))
    assert np.allclose(r["T"][2, :3], [0, 0, 1]) and np.allclose(r["T"][3], [0, 0, 0, 1])
    assert abs(np.linalg.det(r["T"][:3, :3]) - 1) < 1e-12


def test_solve_soft_failure_and_modes(qo):
    src, tgt, _, _ = synth.correspondences(1, 1.0, seed=0)
    r = qo.solve(src, tgt)
    assert not r["valid"] and r["status"] == 1
    src, tgt, _, _ = synth.correspondences(300, 0.2, seed=3)
    assert qo.solve(src, tgt, qo.default_params(inlier_selection_mode=3))["status"] == 2
    rk = qo.solve(src, tgt, qo.default_params(inlier_selection_mode=2, kcore_heuristic_threshold=0.1))
    assert rk["valid"]
    rw = qo.solve(src, tgt, qo.default_params(cote_median=0))
    assert rw["valid"]


def test_register_pair_end_to_end(qo, small_pair):
    s, t, Tgt = small_pair
    r = qo.register_pair(s, t, seed=2)
    assert r["valid"] and r["L"] > 50
    yaw = np.arctan2(r["T"][1, 0], r["T"][0, 0])
    yaw_gt = np.arctan2(Tgt[1, 0], Tgt[0, 0])
    assert abs(np.arctan2(np.sin(yaw - yaw_gt), np.cos(yaw - yaw_gt))) < 0.02
    assert np.linalg.norm(r["T"][:3, 3] - Tgt[:3, 3]) < 0.3


# ------------------------------------------------------------------- (f)1 range image + sub-cluster rejection
def test_segment_cloud_against_independent_numpy_restatement(qo):
    """ImageProjection::segmentCloud restated twice: the oracle's breadth-first labelling vs projection + scipy
    connected components + the validity rule in numpy (float64 trigonometry, so pixels / edges that sit within
    1e-4 of a decision boundary are excluded from the comparison)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from quatro_amd import synth
    s, _, _ = synth.kitti64_pair(1)
    checked = 0
    for lidar, mode in (("Velodyne-64-HDE", "4CrossNeighbor"), ("Velodyne-64-HDE", "4Neighbor"),
                        ("Ouster-OS1-64", "8Neighbor")):
        ipp = qo.ip_params(lidar, mode)
        r = qo.segment_cloud(s, ipp)
        NS, H = ipp.n_scan, ipp.horizon_scan
        x, y, z = (s[:, k].astype(np.float64) for k in range(3))
        va = np.degrees(np.arctan2(z, np.hypot(x, y)))
        rowf = (va + ipp.ang_bottom) / ipp.ang_res_y
        ha = np.degrees(np.arctan2(x, y))
        colr = (ha - 90.0) / ipp.ang_res_x
        col = (-np.round(colr) + H // 2).astype(np.int64)
        col = np.where(col >= H, col - H, col)
        rng_ = np.sqrt(x * x + y * y + z * z)
        row = np.trunc(rowf).astype(np.int64)
        ok = (rowf > -1) & (row < NS) & (col >= 0) & (col < H) & (rng_ >= 0.1)
        sure = (np.abs(rowf - np.round(rowf)) > 1e-3) & (np.abs(colr - np.floor(colr) - 0.5) > 1e-3)
        owner = np.full(NS * H, -1, dtype=np.int64)
        for i in np.nonzero(ok)[0]:  # last writer wins
            owner[row[i] * H + col[i]] = i
        lab = r["labels"].reshape(-1)
        if sure.all():
            assert np.array_equal(owner >= 0, lab != -1)
            assert np.allclose(r["ranges"].reshape(-1)[owner >= 0], rng_[owner[owner >= 0]], rtol=1e-6)
        # edges
        R = np.where(owner >= 0, rng_[np.maximum(owner, 0)], np.inf).reshape(NS, H)
        ax, ay = np.radians(np.float32(ipp.ang_res_x)), np.radians(np.float32(ipp.ang_res_y))
        offs = {0: [(0, 1), (1, 0)], 1: [(0, 1), (1, 0), (1, 1), (1, -1)], 2: [(1, 1), (1, -1)]}[ipp.neighbor_mode]
        ii, jj, uu, vv = [], [], [], []
        for dr, dc in offs:
            A = R[:NS - dr] if dr else R
            B = np.roll(R, -dc, axis=1)[dr:] if dr else np.roll(R, -dc, axis=1)
            d1, d2 = np.maximum(A, B), np.minimum(A, B)
            al = ax if dr == 0 else ay
            with np.errstate(invalid="ignore"):
                ang = np.arctan2(d2 * np.sin(al), d1 - d2 * np.cos(al))
            valid = np.isfinite(A) & np.isfinite(B)
            near = valid & (np.abs(ang - ipp.segment_theta) < 3e-6)  # float32 vs float64: undecided here
            e = valid & (ang > ipp.segment_theta) & ~near
            for mask, (li, lj) in ((e, (ii, jj)), (near, (uu, vv))):
                rr, cc = np.nonzero(mask)
                li.append(rr * H + cc)
                lj.append((rr + dr) * H + (cc + dc) % H)
        ii, jj, uu, vv = (np.concatenate(q) for q in (ii, jj, uu, vv))
        assert sure.all()

        def comps(a, b):
            return connected_components(coo_matrix((np.ones(a.size), (a, b)), shape=(NS * H, NS * H)), directed=False)[1]
        comp = comps(ii, jj)
        comp_hi = comps(np.concatenate([ii, uu]), np.concatenate([jj, vv]))
        occ = owner >= 0
        # components untouched by an undecided edge are the ones both restatements must agree on exactly
        touched = np.zeros(NS * H, dtype=bool)
        touched[np.isin(comp_hi, comp_hi[np.concatenate([uu, vv])])] = True if uu.size else False
        stable = occ & ~touched
        assert stable.sum() > 0.5 * occ.sum(), (stable.sum(), occ.sum(), uu.size)
        first = {}
        for p in np.nonzero(stable)[0]:
            first.setdefault(comp[p], p)
        sizes = np.bincount(comp[occ], minlength=comp.max() + 1)
        prev_label = 0
        for cid, p0 in sorted(first.items(), key=lambda kv: kv[1]):
            members = np.nonzero(occ & (comp == cid))[0]
            assert len(set(lab[members].tolist())) == 1  # one oracle label per component
            rows_pushed = {int(p) // H for p in members if p != p0}
            good = sizes[cid] >= ipp.num_min_pts or (sizes[cid] >= 5 and len(rows_pushed) >= 3)
            if good:
                assert 0 < lab[p0] < 999999 and lab[p0] > prev_label  # numbered in row-major order of first pixels
                assert (lab == lab[p0]).sum() == members.size       # ... and nobody else carries that label
                prev_label = lab[p0]
            else:
                assert lab[p0] == 999999
        valid_mask = occ & (lab != 999999)
        assert r["valid"].shape[0] == int(valid_mask.sum()) and r["outliers"].shape[0] == int((lab == 999999).sum())
        assert np.array_equal(r["valid"][:, 3].astype(np.int64), lab[valid_mask])
        assert np.array_equal(r["valid"][:, :3], s[owner[valid_mask], :3])
        checked += 1
    assert checked == 3


def test_segment_cloud_edge_cases(qo):
    ipp = qo.ip_params()
    r = qo.segment_cloud(np.zeros((0, 4), dtype=np.float32), ipp)
    assert r["valid"].shape[0] == 0 and r["outliers"].shape[0] == 0 and (r["labels"] == -1).all()
    # one isolated return -> a rejected sub-cluster; points closer than 0.1 m or outside the vertical FOV are dropped
    pts = np.array([[10, 0, 0, 0], [0.01, 0.01, 0, 0], [1, 0, 5, 0]], dtype=np.float32)
    r = qo.segment_cloud(pts, ipp)
    assert r["valid"].shape[0] == 0 and r["outliers"].shape[0] == 1 and (r["labels"] == 999999).sum() == 1
    with pytest.raises(KeyError):
        qo.ip_params("no-such-lidar")


def test_patchwork_against_independent_numpy_checks(qo):
    """Patchwork restatement (reference include/patchwork.hpp:329-476): binning against a float64 numpy zone model,
    partition/ordering properties of the two outputs, and ground quality on a synthetic scan with known ground."""
    xyzi, is_ground = synth.kitti64_raw_scan(0)
    pp = qo.pw_params()
    r = qo.patchwork(xyzi, pp)
    P = xyzi.shape[0]
    x, y = xyzi[:, 0].astype(np.float64), xyzi[:, 1].astype(np.float64)
    rad = np.sqrt(x * x + y * y)
    theta = np.arctan2(y, x)
    theta = np.where(theta < 0, theta + 2 * np.pi, theta)
    mins = list(pp.min_ranges[:4]) + [pp.max_range]
    nsec, nring = list(pp.num_sectors_each_zone[:4]), list(pp.num_rings_each_zone[:4])
    want = np.full(P, -1)
    base = 0
    for k in range(4):
        m = (rad >= mins[k]) & (rad < mins[k + 1]) if k else (rad > mins[0]) & (rad < mins[1])
        ring = np.minimum(((rad - mins[k]) / ((mins[k + 1] - mins[k]) / nring[k])).astype(int), nring[k] - 1)
        sec = np.minimum((theta / (2 * np.pi / nsec[k])).astype(int), nsec[k] - 1)
        want[m] = (base + ring * nsec[k] + sec)[m]
        base += nring[k] * nsec[k]
    got = r["patch"]
    # away from ring / sector / range boundaries the float64 model must agree exactly
    ringf = np.concatenate([(rad - mins[k]) / ((mins[k + 1] - mins[k]) / nring[k]) for k in range(4)]).reshape(4, P)
    secf = np.stack([theta / (2 * np.pi / nsec[k]) for k in range(4)])
    safe = np.ones(P, bool)
    for k in range(4):
        safe &= np.abs(ringf[k] - np.round(ringf[k])) > 1e-4
        safe &= np.abs(secf[k] - np.round(secf[k])) > 1e-4
    for b in mins:
        safe &= np.abs(rad - b) > 1e-4
    assert safe.mean() > 0.95
    assert np.array_equal(got[safe], want[safe])
    # every binned point of a patch with more than num_min_pts members comes out exactly once
    sizes = np.bincount(got[got >= 0], minlength=base)
    kept = (got >= 0) & (sizes[np.maximum(got, 0)] > pp.num_min_pts)
    out = np.concatenate([r["ground"], r["nonground"]])
    assert out.shape[0] == kept.sum()
    key = lambda a: np.sort(np.ascontiguousarray(a).view([("", np.uint32)] * 4).ravel())
    assert np.array_equal(key(out), key(xyzi[kept]))
    # quality against the generator's own ground truth
    rec = lambda a: {bytes(v) for v in np.ascontiguousarray(a).view(np.uint8).reshape(-1, 16)}
    gset = rec(r["ground"])
    truth = rec(xyzi[is_ground & kept])
    tp = len(gset & truth)
    assert tp / max(len(gset), 1) > 0.93 and tp / max(len(truth), 1) > 0.90


def test_patchwork_edge_cases_and_determinism(qo):
    xyzi, _ = synth.kitti64_raw_scan(1)
    a, b = qo.patchwork(xyzi), qo.patchwork(xyzi.copy())
    assert np.array_equal(a["ground"], b["ground"]) and np.array_equal(a["nonground"], b["nonground"])
    e = qo.patchwork(np.zeros((0, 4), dtype=np.float32))
    assert e["ground"].shape[0] == 0 and e["nonground"].shape[0] == 0
    # a tilted wall inside one patch is rejected by the uprightness rule: everything non-ground
    rng = np.random.default_rng(2)
    wall = np.zeros((3000, 4), dtype=np.float32)
    wall[:, 0] = 6.0 + rng.normal(0, 0.005, 3000)
    wall[:, 1] = rng.uniform(-0.3, 0.3, 3000)
    wall[:, 2] = rng.uniform(-1.7, 0.5, 3000)
    w = qo.patchwork(wall)
    assert w["nonground"].shape[0] == 3000 and w["ground"].shape[0] == 0
    # a flat floor at sensor height is all ground
    floor = wall.copy()
    floor[:, 0] = rng.uniform(5.0, 8.0, 3000)
    floor[:, 2] = -1.723 + rng.normal(0, 0.01, 3000)
    f = qo.patchwork(floor)
    assert f["ground"].shape[0] == 3000


def test_exact_clique_is_maximum_and_defined(qo):
    """PMC_EXACT restatement (reference src/graph.cc:106-127): the size equals networkx's exact clique number, the
    result is a clique, and it is the heuristic's own clique whenever that is already maximum (D10)."""
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(0)
    beaten = 0
    for L, p, plant in [(60, 0.3, 0), (120, 0.2, 10), (200, 0.1, 12), (150, 0.5, 0), (300, 0.3, 25), (400, 0.05, 0),
                        (90, 0.7, 0), (1, 0.0, 0), (2, 1.0, 0), (5, 0.0, 0)]:
        A = np.triu(rng.random((L, L)) < p, 1)
        A = A | A.T
        if plant:
            idx = rng.choice(L, plant, replace=False)
            A[np.ix_(idx, idx)] = True
            A[idx, idx] = False
        W = (L + 63) // 64
        bits = np.zeros((L, W * 64), dtype=np.uint8)
        bits[:, :L] = A
        bm = np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, W)
        ce, ch = qo.max_clique(bm, 0), qo.max_clique(bm, 1)
        omega = max(len(c) for c in nx.find_cliques(nx.from_numpy_array(A.astype(int))))
        if A.sum() == 0:
            omega = ch.size  # PMC reports no clique on an edgeless graph; exact mode keeps that
        assert ce.size == omega >= ch.size
        assert all(A[a, b] for a in ce for b in ce if a != b)
        if ce.size == ch.size:
            assert np.array_equal(np.sort(ce), np.sort(ch))
        else:
            beaten += 1
        assert np.array_equal(qo.max_clique(bm, 0), ce)  # deterministic
    assert beaten >= 2


def test_rot3_matches_svd_construction(qo):
    """qm_rot3_from_h (Horn quaternion form) against teaser::utils::svdRot's construction with numpy's SVD
    (reference include/teaser/utils.h:123-149: V diag(1, 1, det) U^T): same matrix when H is well conditioned, same
    objective and a proper rotation always."""
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(500):
        n = int(rng.integers(1, 30))
        X, Y = rng.standard_normal((3, n)), rng.standard_normal((3, n))
        if trial % 7 == 0:
            X[2] = 0
            Y[2] = 0
        w = rng.random(n)
        H = (X * w) @ Y.T
        U, S, Vt = np.linalg.svd(H)
        V = Vt.T
        if np.linalg.det(U) * np.linalg.det(V) < 0:
            V[:, 2] *= -1
        Rref = V @ U.T
        R = qo.rot3_from_h(H)
        assert abs(np.linalg.det(R) - 1) < 1e-12 and np.abs(R @ R.T - np.eye(3)).max() < 1e-12
        assert np.trace(R @ H) >= np.trace(Rref @ H) - 1e-9 * max(1.0, abs(np.trace(Rref @ H)))
        if n >= 3 and trial % 7 and S[1] > 1e-6 * S[0]:
            worst = max(worst, np.abs(R - Rref).max())
    assert worst < 1e-10
    assert np.array_equal(qo.rot3_from_h(np.zeros((3, 3))), np.eye(3))


def test_gnc_rotation3d_recovers_rotation_and_flags_outliers(qo):
    from scipy.spatial.transform import Rotation as Rt
    rng = np.random.default_rng(1)
    M = 400
    X = rng.uniform(-10, 10, (M, 3))
    Rm = Rt.from_euler("zyx", [50, -20, 10], degrees=True).as_matrix()
    Y = X @ Rm.T + rng.normal(0, 0.02, (M, 3))
    bad = rng.random(M) < 0.4
    Y[bad] = rng.uniform(-10, 10, (int(bad.sum()), 3))
    R, cost, iters, mask = qo.gnc_rotation3d(X, Y, 0.2, 1.4, 100, 1e-6)
    ang = np.arccos(np.clip((np.trace(R.T @ Rm) - 1) / 2, -1, 1))
    assert ang < 2e-3 and 1 < iters < 100
    assert mask[~bad].mean() > 0.98 and mask[bad].mean() < 0.05
    # no outliers and a generous bound: the degenerate-mu exit after one iteration, all inliers
    R1, _, it1, m1 = qo.gnc_rotation3d(X, X @ Rm.T, 5.0, 1.4, 100, 1e-6)
    assert it1 == 1 and m1.all() and np.abs(R1 - Rm).max() < 1e-9


def _ref_cote_python(X, r, median):
    """Quatro::estimate (reference include/quatro.hpp:618-747) written again, directly from the reference text, as plain
    Python floats (IEEE binary64, the same operation order); r: one range or one per element; std::sort's unspecified
    order among equal keys taken as insertion order (the oracle's definition); ranges.sum() taken sequentially."""
    N = len(X)
    R = [float(r)] * N if np.isscalar(r) else [float(v) for v in r]
    h = []
    for i in range(N):
        h.append((X[i] - R[i], i + 1))
        h.append((X[i] + R[i], -i - 1))
    h.sort(key=lambda e: e[0])  # stable
    ranges_inverse_sum = 0.0
    for i in range(N):
        ranges_inverse_sum += R[i]
    dot_X_weights = dot_weights_consensus = sum_xi = sum_xi_square = 0.0
    card = 0
    x_hat, x_cost, set_card = [], [], []
    for key, tag in h:
        idx = abs(tag) - 1
        eps = 1 if tag > 0 else -1
        w = 1.0 / (R[idx] * R[idx])
        card += eps
        dot_weights_consensus += eps * w
        dot_X_weights += eps * w * X[idx]
        ranges_inverse_sum -= eps * R[idx]
        sum_xi += eps * X[idx]
        sum_xi_square += eps * X[idx] * X[idx]
        set_card.append(card)
        xh = dot_X_weights / dot_weights_consensus if dot_weights_consensus != 0 else float("nan")
        x_hat.append(xh)
        x_cost.append(card * xh * xh + sum_xi_square - 2 * sum_xi * xh + ranges_inverse_sum)
    mi = min(range(2 * N), key=lambda i: (x_cost[i], i))  # Eigen minCoeff: first minimum
    n_card = set_card[mi]
    if median:
        cand = sorted(X[abs(h[mi - j][1]) - 1] for j in range(n_card))
        est = (cand[len(cand) // 2 - 1] + cand[len(cand) // 2]) / 2.0 if n_card >= 2 else (cand[0] if n_card == 1 else x_hat[mi])
    else:
        est = x_hat[mi]
    return est, n_card, [abs(x - est) <= ri for x, ri in zip(X, R)]


def test_cote_against_a_second_restatement_of_the_reference(qo):
    rng = np.random.default_rng(7)
    checked = 0
    for N, spread, r in [(2, 0.1, 0.3), (5, 0.5, 0.3), (64, 0.2, 0.3), (300, 1.0, 0.3), (301, 0.05, 0.15), (1000, 3.0, 0.6)]:
        for trial in range(4):
            X = rng.normal(0.3, spread, N)
            if trial == 1:
                X = np.round(X, 1)  # exact ties among the interval endpoints
            if trial == 2:
                X[: N // 3] += 5.0  # a far cluster of outliers
            for median in (True, False):
                e_ref, card_ref, inl_ref = _ref_cote_python([float(v) for v in X], r, median)
                if any(np.isnan(v) for v in [e_ref]):
                    continue
                e, inl, card = qo.cote_estimate(X, r, median)
                assert card == card_ref
                assert e == e_ref, (N, trial, median, e, e_ref)
                assert inl.tolist() == inl_ref
                checked += 1
    assert checked >= 40
    # one range per element (estimate() takes a vector of ranges; the class only ever passes equal ones)
    for N in (2, 7, 120, 500):
        X = rng.normal(0.3, 0.5, N)
        R = rng.uniform(0.05, 0.6, N)
        for median in (True, False):
            e_ref, card_ref, inl_ref = _ref_cote_python([float(v) for v in X], R, median)
            e, inl, card = qo.cote_estimate_ranges(X, R, median)
            assert card == card_ref and e == e_ref and inl.tolist() == inl_ref


def _ref_gnc_rotation2d_numpy(src, dst, noise_bound, gnc_factor, max_iter, cost_thr):
    """solveForRotation2D (reference include/quatro.hpp:430-572) with teaser::utils::svdRot2d (include/teaser/utils.h:
    151-166) written again from the reference text: numpy SVD, sequential cost sum."""
    X, Y = np.asarray(src, dtype=np.float64).T, np.asarray(dst, dtype=np.float64).T  # 2 x M
    M = X.shape[1]
    w = np.ones(M)
    mu, prev_cost, cost = 1.0, np.inf, np.inf
    nb_sq = noise_bound ** 2
    if nb_sq < 1e-16:
        nb_sq = 1e-2
    R = np.eye(2)
    iters = 0
    for i in range(max_iter):
        iters = i + 1
        H = (X * w) @ Y.T
        U, _, Vt = np.linalg.svd(H)
        V = Vt.T
        if np.linalg.det(U) * np.linalg.det(V) < 0:
            V[:, 1] *= -1
        R = V @ U.T
        res = ((Y - R @ X) ** 2).sum(0)
        if i == 0:
            mu = 1 / (2 * res.max() / nb_sq - 1)
            if mu <= 0:
                break
        th1, th2 = (mu + 1) / mu * nb_sq, mu / (mu + 1) * nb_sq
        cost = 0.0
        for j in range(M):
            cost += w[j] * res[j]
            if res[j] >= th1:
                w[j] = 0
            elif res[j] <= th2:
                w[j] = 1
            else:
                w[j] = np.sqrt(nb_sq * mu * (mu + 1) / res[j]) - mu
        cost_diff = abs(cost - prev_cost)
        mu *= gnc_factor
        prev_cost = cost
        if cost_diff < cost_thr:
            break
    return R, cost, iters, w >= 0.4


def test_gnc_rotation2d_against_a_second_restatement_of_the_reference(qo):
    """Same loop, SVD instead of the closed form and a different summation order: the iteration count and the inlier
    mask agree exactly, the rotation to 1e-9 and the cost to 1e-9 relative, on clean, noisy and outlier-heavy inputs."""
    rng = np.random.default_rng(11)
    for M, noise, frac_out, nb in [(2, 0.0, 0.0, 0.6), (50, 0.02, 0.2, 0.6), (300, 0.05, 0.5, 0.6), (1000, 0.05, 0.7, 0.2),
                                   (200, 0.0, 0.0, 0.6), (400, 0.3, 0.3, 0.1)]:
        X = rng.uniform(-10, 10, (M, 2))
        a = rng.uniform(-np.pi, np.pi)
        Rm = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        Y = X @ Rm.T + rng.normal(0, noise, (M, 2))
        out = rng.random(M) < frac_out
        Y[out] = rng.uniform(-10, 10, (int(out.sum()), 2))
        R, cost, iters, mask = qo.gnc_rotation2d(X, Y, nb, 1.4, 50, 1.1e-4)
        Rr, costr, itr, maskr = _ref_gnc_rotation2d_numpy(X, Y, nb, 1.4, 50, 1.1e-4)
        assert iters == itr and np.array_equal(mask, maskr)
        assert np.abs(R - Rr).max() < 1e-9
        assert (np.isinf(cost) and np.isinf(costr)) or abs(cost - costr) <= 1e-9 * max(1.0, abs(costr))


def _ref_pmc_heu_python(A):
    """pmc_heu::search_bounds with the "kcore" strategy, single thread, restated once more from SURVEY.md Appendix A.4
    with Python sets and networkx core numbers (canonical (core, id) orders, as the oracle defines them)."""
    import networkx as nx
    V = A.shape[0]
    core = nx.core_number(nx.from_numpy_array(A.astype(int)))
    K = [core[v] + 1 for v in range(V)]
    nbrs = [set(np.nonzero(A[v])[0].tolist()) for v in range(V)]
    order = sorted(range(V), key=lambda v: (K[v], v))
    ub = max(K) if V else 0
    mc, C_max = 0, []
    for v in reversed(order):
        if not K[v] > mc:
            continue
        P = sorted((u for u in nbrs[v] if K[u] > mc), key=lambda u: (K[u], u))
        if len(P) <= mc:
            continue
        mc_cur, depth, chosen = mc, 1, []
        while P:
            u = P.pop()
            chosen.append(u)
            P = [w for w in P if w in nbrs[u] and K[w] > mc_cur]
            depth += 1
        if depth > mc_cur:
            mc_cur = depth
        if mc_cur > mc:
            mc = mc_cur
            C_max = chosen + [v]
            if mc >= ub:
                break
    return sorted(C_max)


def test_clique_heuristic_against_a_second_restatement(qo):
    rng = np.random.default_rng(13)
    for L, p, planted in [(30, 0.3, 0), (80, 0.2, 8), (200, 0.05, 12), (150, 0.5, 0), (400, 0.02, 15), (300, 0.3, 25),
                          (64, 0.9, 0), (500, 0.01, 0)]:
        A = np.triu(rng.random((L, L)) < p, 1)
        A = A | A.T
        if planted:
            idx = rng.choice(L, planted, replace=False)
            A[np.ix_(idx, idx)] = True
            A[idx, idx] = False
        W = (L + 63) // 64
        bits = np.zeros((L, W * 64), dtype=np.uint8)
        bits[:, :L] = A
        bm = np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, W)
        got = sorted(qo.max_clique(bm, 1).tolist())
        assert got == _ref_pmc_heu_python(A), (L, p, planted)


def _ref_patchwork_float64(xyz, pp):
    """PatchWork::estimate_ground (reference include/patchwork.hpp:264-318, 329-476, 492-586) written again from the
    reference text in float64 numpy with numpy's SVD of the patch covariance.  Returns a label per input point:
    1 ground, 0 non-ground, -1 not emitted."""
    P = np.asarray(xyz, dtype=np.float64)
    n = P.shape[0]
    label = np.full(n, -1)
    order = np.argsort(P[:, 2], kind="stable")
    order = order[P[order, 2] >= -1.8 * pp.sensor_height]
    r = np.sqrt(P[:, 0] ** 2 + P[:, 1] ** 2)
    th = np.arctan2(P[:, 1], P[:, 0])
    th = np.where(th > 0, th, th + 2 * np.pi)
    mins = [pp.min_ranges[i] for i in range(pp.num_zones)] + [pp.max_range]
    patches = {}
    for i in order:
        if not (r[i] <= pp.max_range and r[i] > pp.min_range):
            continue
        k = pp.num_zones - 1
        for z in range(1, pp.num_zones):
            if r[i] < mins[z]:
                k = z - 1
                break
        nr, ns = pp.num_rings_each_zone[k], pp.num_sectors_each_zone[k]
        ring = min(int((r[i] - mins[k]) / ((mins[k + 1] - mins[k]) / nr)), nr - 1)
        sec = min(int(th[i] / (2 * np.pi / ns)), ns - 1)
        patches.setdefault((k, ring, sec), []).append(i)
    margin = -0.1 if pp.sensor_height == 0.0 else pp.adaptive_seed_selection_margin * pp.sensor_height
    cidx = 0
    for k in range(pp.num_zones):
        for ring in range(pp.num_rings_each_zone[k]):
            for sec in range(pp.num_sectors_each_zone[k]):
                ids = np.array(patches.get((k, ring, sec), []), dtype=int)
                if not ids.size > pp.num_min_pts:
                    continue
                Q = P[ids]
                init = 0
                if k == 0:
                    while init < ids.size and Q[init, 2] < margin:
                        init += 1
                lpr = Q[init:init + pp.num_lpr, 2]
                lpr_h = lpr.mean() if lpr.size else 0.0
                g = Q[:, 2] < lpr_h + pp.th_seeds
                for it in range(pp.num_iter):
                    G = Q[g]
                    mean = G.mean(0)
                    cov = (G - mean).T @ (G - mean) / G.shape[0]
                    U, S, _ = np.linalg.svd(cov)
                    normal = U[:, 2]
                    d = -normal @ mean
                    g = Q @ normal < pp.th_dist - d
                zvec, elev = abs(normal[2]), mean[2]
                surf = S.min() / S.sum()
                reject = False
                if zvec < pp.uprightness_thr:
                    reject = True
                elif cidx < pp.num_thr:
                    ti = ring + 2 * k
                    if elev > pp.elevation_thr[ti] and not (pp.flatness_thr[ti] > surf):
                        reject = True
                elif pp.using_global_thr and elev > pp.global_elevation_thr:
                    reject = True
                label[ids] = 0
                if not reject:
                    label[ids[g]] = 1
            cidx += 1
    return label


def test_patchwork_against_a_second_restatement_of_the_reference(qo):
    """The oracle's float32, fixed-order plane fits against a float64 SVD restatement of the same text: the same points are
    emitted, and all but a sliver of them (points within rounding of th_dist, patches within rounding of a threshold)
    get the same ground / non-ground label."""
    for scan_id, mutate in [(0, None), (2, None), (1, "iter1"), (3, "global")]:
        xyzi, _ = synth.kitti64_raw_scan(scan_id)
        pp = qo.pw_params()
        if mutate == "iter1":
            pp.num_iter, pp.num_lpr = 1, 5
        if mutate == "global":
            pp.using_global_thr, pp.global_elevation_thr = 1, -1.5
        r = qo.patchwork(xyzi, pp)
        key = lambda a: [bytes(v) for v in np.ascontiguousarray(a).view(np.uint8).reshape(-1, 16)]
        lab_o = {}
        for k_ in key(r["ground"]):
            lab_o[k_] = 1
        for k_ in key(r["nonground"]):
            lab_o[k_] = 0
        lab_r = _ref_patchwork_float64(xyzi[:, :3], pp)
        recs = key(xyzi)
        got = np.array([lab_o.get(k_, -1) for k_ in recs])
        assert np.array_equal(got == -1, lab_r == -1)  # the same points are emitted
        emitted = lab_r >= 0
        assert (got[emitted] == lab_r[emitted]).mean() > 0.995, (scan_id, mutate)


def _ref_advanced_matching_python(feat_src, feat_tgt):
    """Matcher::advancedMatching with use_crosscheck = true and the tuple test off (reference
    src/teaser_utils/feature_matcher.cc:77-170, 252-265) written again from the reference text: brute-force exact nearest
    neighbours in float64 (FLANN's single kd-tree with eps = 0 is exact), lazy i_to_j, Mi / Mj cross check, un-swap, sort,
    unique."""
    feats = [np.asarray(feat_src, dtype=np.float64), np.asarray(feat_tgt, dtype=np.float64)]
    fi, fj, swapped = 0, 1, False
    if feats[fj].shape[0] > feats[fi].shape[0]:
        fi, fj, swapped = 1, 0, True

    def nn(tree, q):
        return int(np.argmin(((tree - q) ** 2).sum(1)))
    nPti, nPtj = feats[fi].shape[0], feats[fj].shape[0]
    i_to_j = [-1] * nPti
    corres_ji = []
    for j in range(nPtj):
        i = nn(feats[fi], feats[fj][j])
        if i_to_j[i] == -1:
            i_to_j[i] = nn(feats[fj], feats[fi][i])
        corres_ji.append((i, j))
    corres_ij = [(i, i_to_j[i]) for i in range(nPti) if i_to_j[i] != -1]
    Mi = [[] for _ in range(nPti)]
    Mj = [[] for _ in range(nPtj)]
    for ci, cj in corres_ij:
        Mi[ci].append(cj)
    for ci, cj in corres_ji:
        Mj[cj].append(ci)
    corres = []
    for i in range(nPti):
        for j in Mi[i]:
            for ii in Mj[j]:
                if ii == i:
                    corres.append((i, j))
    if swapped:
        corres = [(b, a) for a, b in corres]
    return sorted(set(corres))


def test_matcher_cross_check_against_a_second_restatement(qo):
    rng = np.random.default_rng(17)
    for ns, nt in [(40, 55), (300, 180), (257, 257), (1, 7), (500, 640)]:
        # clustered descriptors so that mutual nearest neighbours are neither rare nor universal
        centres = rng.uniform(0, 100, (max(ns, nt) // 3 + 1, 33))
        fs = (centres[rng.integers(0, centres.shape[0], ns)] + rng.normal(0, 3.0, (ns, 33))).astype(np.float32)
        ft = (centres[rng.integers(0, centres.shape[0], nt)] + rng.normal(0, 3.0, (nt, 33))).astype(np.float32)
        xs = rng.uniform(-10, 10, (ns, 4)).astype(np.float32)
        xt = rng.uniform(-10, 10, (nt, 4)).astype(np.float32)
        got = qo.match(xs, fs, xt, ft, crosscheck=True, tuple_test=False)
        want = _ref_advanced_matching_python(fs, ft)
        assert [tuple(r) for r in got.tolist()] == want, (ns, nt)
        assert len(want) > 0


def _ref_compute_transformation_python(src4, tgt4, noise_bound=0.3, cbar2=1.0):
    """Quatro::computeTransformation (reference include/quatro.hpp:769-936) composed once more from the second
    restatements above: scale-consistency graph (:355-386), PMC heuristic, chain TIMs (:817-844), GNC-TLS yaw with the
    doubled noise bound (:850-852), the rotation-inlier chain rule (:857-874), COTE per axis on dst - R src (:585-615)."""
    s = np.asarray(src4, dtype=np.float32)[:, :3].astype(np.float64)
    t = np.asarray(tgt4, dtype=np.float32)[:, :3].astype(np.float64)
    L = s.shape[0]
    beta = 2 * noise_bound * np.sqrt(cbar2)
    A = np.zeros((L, L), dtype=bool)
    for i in range(L):
        v1 = np.sqrt(((s[i + 1:] - s[i]) ** 2).sum(1))
        v2 = np.sqrt(((t[i + 1:] - t[i]) ** 2).sum(1))
        with np.errstate(divide="ignore", invalid="ignore"):
            ok = (np.abs(v2 / v1 - 1) <= beta / v1) & (np.abs(v1 / v2 - 1) <= beta / v2)
        A[i, i + 1:] = ok
    A = A | A.T
    clique = _ref_pmc_heu_python(A)
    if len(clique) <= 1:
        return dict(valid=False, clique=clique)
    M = len(clique)
    leaf = clique[1:] + clique[:1]
    ps, pd = s[leaf] - s[clique], t[leaf] - t[clique]
    R2, cost, iters, mask = _ref_gnc_rotation2d_numpy(ps[:, :2], pd[:, :2], 2 * noise_bound, 1.4, 50, 1.1e-4)
    R = np.eye(3)
    R[:2, :2] = R2
    rot = [i for i in range(M) if mask[i - 1 if i else M - 1] and mask[i]]
    raw = t[clique] - s[clique] @ R.T
    r = 0.3 * np.sqrt(cbar2)
    tr, inl = [], np.ones(M, dtype=bool)
    for a in range(3):
        e, _, ia = _ref_cote_python([float(v) for v in raw[:, a]], r, True)
        tr.append(e)
        inl &= np.array(ia)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = tr
    return dict(valid=True, clique=clique, rot_inliers=rot, final_inliers=[clique[i] for i in np.nonzero(inl)[0]], T=T,
                gnc_iters=iters)


@pytest.mark.parametrize("L,frac,noise,seed", [(60, 0.5, 0.02, 1), (200, 0.3, 0.05, 2), (400, 0.1, 0.05, 3), (150, 0.0, 0.0, 4),
                                               (300, 0.9, 0.1, 5)])
def test_back_end_against_the_composed_second_restatement(qo, L, frac, noise, seed):
    src, tgt, _, _ = synth.correspondences(L, frac, seed=seed, noise=noise)
    o = qo.solve(src, tgt)
    r = _ref_compute_transformation_python(src, tgt)
    assert o["valid"] == r["valid"]
    assert sorted(o["clique"].tolist()) == r["clique"]
    if not r["valid"]:
        return
    assert o["rot_inliers"].tolist() == r["rot_inliers"] and o["gnc_iters"] == r["gnc_iters"]
    assert o["final_inliers"].tolist() == r["final_inliers"]
    assert np.abs(o["T"] - r["T"]).max() < 1e-9


def test_gnc_rotation3d_against_an_svd_restatement(qo):
    """The 3-DoF loop with numpy's SVD-based svdRot (reference include/teaser/utils.h:123-149) and a sequential cost sum:
    same iteration count and inlier mask, rotation to 1e-9."""
    from scipy.spatial.transform import Rotation as Rt

    def ref(X, Y, nb, factor, max_iter, thr):
        X, Y = X.T, Y.T
        M = X.shape[1]
        w = np.ones(M)
        mu, prev, cost, iters = 1.0, np.inf, np.inf, 0
        nb_sq = nb * nb if nb * nb >= 1e-16 else 1e-2
        R = np.eye(3)
        for i in range(max_iter):
            iters = i + 1
            U, _, Vt = np.linalg.svd((X * w) @ Y.T)
            V = Vt.T
            if np.linalg.det(U) * np.linalg.det(V) < 0:
                V[:, 2] *= -1
            R = V @ U.T
            res = ((Y - R @ X) ** 2).sum(0)
            if i == 0:
                mu = 1 / (2 * res.max() / nb_sq - 1)
                if mu <= 0:
                    break
            th1, th2 = (mu + 1) / mu * nb_sq, mu / (mu + 1) * nb_sq
            cost = 0.0
            for j in range(M):
                cost += w[j] * res[j]
                w[j] = 0 if res[j] >= th1 else 1 if res[j] <= th2 else np.sqrt(nb_sq * mu * (mu + 1) / res[j]) - mu
            d = abs(cost - prev)
            mu *= factor
            prev = cost
            if d < thr:
                break
        return R, cost, iters, w >= 0.4
    rng = np.random.default_rng(19)
    for M, noise, frac_out, nb in [(3, 0.0, 0.0, 0.6), (80, 0.02, 0.25, 0.6), (400, 0.05, 0.5, 0.3), (900, 0.05, 0.7, 0.2)]:
        X = rng.uniform(-10, 10, (M, 3))
        Rm = Rt.from_euler("zyx", rng.uniform(-80, 80, 3), degrees=True).as_matrix()
        Y = X @ Rm.T + rng.normal(0, noise, (M, 3))
        out = rng.random(M) < frac_out
        Y[out] = rng.uniform(-10, 10, (int(out.sum()), 3))
        R, cost, iters, mask = qo.gnc_rotation3d(X, Y, nb, 1.4, 60, 1.1e-4)
        Rr, costr, itr, maskr = ref(X, Y, nb, 1.4, 60, 1.1e-4)
        assert iters == itr and np.array_equal(mask, maskr)
        assert np.abs(R - Rr).max() < 1e-9


def _splitmix_u32(seed, counter):
    M = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    return mix(mix(seed) ^ ((counter * 0xD1342543DE82EF95) & M)) >> 32


def test_tuple_test_against_a_second_restatement(qo):
    """The tuple constraint (reference src/teaser_utils/feature_matcher.cc:172-247) with the oracle's declared RNG (D1:
    trial t draws qm_rand_u32(seed, 3t + k) mod ncorr) written again in Python with float32 arithmetic: normalizePoints'
    sequential float means, ncorr * 100 trials, the six scale tests, un-swap, sort, unique."""
    f32 = np.float32
    rng = np.random.default_rng(23)
    for ns, nt, seed in [(60, 80, 0), (200, 150, 5), (120, 120, 123456789)]:
        centres = rng.uniform(0, 100, (max(ns, nt) // 2, 33))
        pick_s, pick_t = rng.integers(0, centres.shape[0], ns), rng.integers(0, centres.shape[0], nt)
        fs = (centres[pick_s] + rng.normal(0, 2.0, (ns, 33))).astype(f32)
        ft = (centres[pick_t] + rng.normal(0, 2.0, (nt, 33))).astype(f32)
        # geometry consistent with the descriptor clusters, so that a good share of the triples passes
        anchor = rng.uniform(-20, 20, (centres.shape[0], 3))
        xs = np.zeros((ns, 4), dtype=f32)
        xt = np.zeros((nt, 4), dtype=f32)
        xs[:, :3] = anchor[pick_s] + rng.normal(0, 0.05, (ns, 3))
        xt[:, :3] = anchor[pick_t] + rng.normal(0, 0.05, (nt, 3)) + np.array([3.0, -1.0, 0.5])
        cross = qo.match(xs, fs, xt, ft, crosscheck=True, tuple_test=False)
        got = qo.match(xs, fs, xt, ft, crosscheck=True, tuple_test=True, tuple_scale=0.95, seed=seed)

        def centred(x):  # Matcher::normalizePoints with use_absolute_scale: sequential float mean, subtracted
            m = np.zeros(3, dtype=f32)
            for p in x[:, :3]:
                m = (m + p).astype(f32)
            m = (m / f32(x.shape[0])).astype(f32)
            return (x[:, :3] - m).astype(f32)
        pc = [centred(xs), centred(xt)]
        swapped = nt > ns
        fi, fj = (1, 0) if swapped else (0, 1)
        corres = [(int(b), int(a)) if swapped else (int(a), int(b)) for a, b in cross]  # (i in fi, j in fj), ascending i
        corres.sort()
        ncorr = len(corres)
        assert ncorr > 3

        def norm3(a, b):
            d = (a - b).astype(f32)
            return np.sqrt(f32(d[0] * d[0]) + (f32(d[1] * d[1]) + f32(d[2] * d[2])), dtype=f32)
        scale = f32(0.95)
        kept = set()
        for t in range(ncorr * 100):
            r = [_splitmix_u32(seed, 3 * t + k) % ncorr for k in range(3)]
            (i0, j0), (i1, j1), (i2, j2) = corres[r[0]], corres[r[1]], corres[r[2]]
            li = [norm3(pc[fi][i0], pc[fi][i1]), norm3(pc[fi][i1], pc[fi][i2]), norm3(pc[fi][i2], pc[fi][i0])]
            lj = [norm3(pc[fj][j0], pc[fj][j1]), norm3(pc[fj][j1], pc[fj][j2]), norm3(pc[fj][j2], pc[fj][j0])]
            if all((f32(a * scale) < b) and (b < f32(a / scale)) for a, b in zip(li, lj)):
                kept.update([(i0, j0), (i1, j1), (i2, j2)])
        want = sorted((j, i) if swapped else (i, j) for i, j in kept)
        assert [tuple(v) for v in got.tolist()] == want, (ns, nt, seed)
        assert 0 < len(want) <= ncorr


def test_spfh_role_swap_shortcut_logic_against_the_oracle_arithmetic(qo):
    """numpy mirror of the device's role-swap shortcut (frontend.hip: spfh_swap_roles) against the oracle's plain
    arithmetic (qo_math 5): wherever the shortcut claims to be sure it must give the oracle's answer; the rest falls
    back to the two arc cosines on the device."""
    rng = np.random.default_rng(3)
    x1 = rng.uniform(0, 1.0000005, 300000).astype(np.float32)
    x2 = np.where(rng.random(300000) < 0.5, x1 + rng.uniform(-2e-6, 2e-6, 300000), rng.uniform(0, 1.0000005, 300000)).astype(np.float32)
    x2 = np.abs(x2)
    want = qo.math_fn(5, x1, x2)
    eq = x1 == x2
    fast = (~eq) & (x1 <= 1) & (x2 <= 1) & (np.abs(x1 - x2) > np.float32(5e-7))
    assert fast.mean() > 0.4
    assert not want[eq].any()
    assert np.array_equal((x1 < x2)[fast].astype(np.float32), want[fast])


def test_spfh_angle_bin_shortcut_logic_against_the_oracle_arithmetic(qo):
    """numpy mirror (binary32, no contraction) of the device's shortcut for the first SPFH block's bin (frontend.hip:
    spfh_bin_of_angle: five cross products against the boundary directions instead of atan2f) against the oracle's plain
    arithmetic — qm_atan2f (qo_math 0), then floor(11 (f1 + pi) d_pi) in binary64 as pcl::computePointSPFHSignature does:
    wherever the shortcut claims to be sure it must give that bin.  Inputs: uniform directions, directions within 4e-6 rad of
    every boundary (both sides, also of the cut at +-pi and of 0), y = +-0, tiny and huge magnitudes."""
    f32 = np.float32
    rng = np.random.default_rng(6)
    d_pi = float(f32(1.0) / (f32(2.0) * f32(np.pi)))
    T = np.array([k / (11 * d_pi) - np.pi for k in range(0, 12)])
    n = 400000
    phi = np.concatenate([rng.uniform(-np.pi, np.pi, n),
                          (T[rng.integers(0, 12, n)] + rng.uniform(-4e-6, 4e-6, n)),
                          rng.uniform(-4e-6, 4e-6, n // 4), np.pi - rng.uniform(0, 4e-6, n // 4), -np.pi + rng.uniform(0, 4e-6, n // 4)])
    r = np.exp(rng.uniform(np.log(1e-12), np.log(1e6), phi.size))
    y, x = (r * np.sin(phi)).astype(f32), (r * np.cos(phi)).astype(f32)
    y[:2000:4], y[1:2000:4] = f32(0.0), f32(-0.0)                      # atan2f(-0, x < 0) = -pi rounds below -pi: bin 0
    f1 = qo.math_fn(0, y, x).astype(np.float64)
    g = 11.0 * ((f1 + np.pi) * d_pi)
    want = np.where(g != g, 0, np.clip(np.floor(g), 0, 10)).astype(np.int64)
    ya, s = np.abs(y), np.abs(y) + np.abs(x)
    C = [(0.95949297361449748, 0.2817325568414295), (0.6548607339452851, 0.75574957435425827),
         (0.14231483827328512, 0.98982144188093268), (-0.41541501300188632, 0.90963199535451844),
         (-0.84125353283118109, 0.54064081745559778)]
    c = [f32(ca) * ya - f32(sa) * x for ca, sa in C]                   # (float32 products, float32 difference)
    low = np.minimum.reduce([np.abs(v) for v in c])
    sure = (low > f32(1e-6) * s) & (s > f32(1e-30)) & (s < f32(1e30)) & (ya != 0)
    m = sum((v > 0).astype(np.int64) for v in c)
    got = np.where(y > 0, 5 + m, 5 - m)
    assert sure[2000:n].mean() > 0.9999 and sure.mean() > 0.4         # (uniform directions: practically always sure)
    assert not sure[:2000:4].any() and not sure[1:2000:4].any()
    assert np.array_equal(got[sure], want[sure])
    assert set(np.unique(want[sure])) == set(range(11))


def test_shared_math_header_against_an_independent_libm(qo):
    """include/qtr_math.h (atan2f / acosf / sinf / cosf) is compiled into BOTH the HIP library and the oracle, so their
    agreement is self-agreement; this pins the header itself against numpy's float64 functions rounded once to float32
    — what a correctly rounded libm returns.  2e6 inputs per function, including the quadrant borders and tiny
    arguments; at most 1e-6 of them may differ (0 were seen), and never by more than 1 ulp."""
    rng = np.random.default_rng(20260925)
    n = 2_000_000

    def ulp_diff(got, want):
        g = np.ascontiguousarray(got, dtype=np.float32).view(np.int32).astype(np.int64)
        w = np.ascontiguousarray(want, dtype=np.float32).view(np.int32).astype(np.int64)
        g = np.where(g < 0, -(g & 0x7fffffff), g)
        w = np.where(w < 0, -(w & 0x7fffffff), w)
        return np.abs(g - w)

    y = np.concatenate([rng.uniform(-1.5, 1.5, n - 4000), rng.uniform(-1e-6, 1e-6, 2000), rng.uniform(-1e3, 1e3, 1992),
                        [0.0, -0.0, 1.0, -1.0, 0.0, -0.0, 1e-30, -1e-30]]).astype(np.float32)
    x = np.concatenate([rng.uniform(-1.5, 1.5, n - 4000), rng.uniform(-1e-6, 1e-6, 2000), rng.uniform(-1e3, 1e3, 1992),
                        [1.0, 1.0, 0.0, 0.0, -1.0, -1.0, -1.0, -1.0]]).astype(np.float32)
    cases = [
        ("atan2f", qo.math_fn(0, y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64))),
    ]
    a = np.concatenate([rng.uniform(-1.0, 1.0, n - 6), [1.0, -1.0, 0.0, -0.0, 0.99999994, -0.99999994]]).astype(np.float32)
    cases.append(("acosf", qo.math_fn(1, a), np.arccos(a.astype(np.float64))))
    th = np.concatenate([rng.uniform(-np.pi, np.pi, n - 4), [0.0, -0.0, 1.5707964, 3.1415927]]).astype(np.float32)
    cases.append(("sinf", qo.math_fn(2, th), np.sin(th.astype(np.float64))))
    cases.append(("cosf", qo.math_fn(3, th), np.cos(th.astype(np.float64))))
    for name, got, want64 in cases:
        want = want64.astype(np.float32)
        d = ulp_diff(got, want)
        # +0 / -0 results (atan2 of signed zeros) are compared as values
        d = np.where((got == 0) & (want == 0), 0, d)
        assert d.max() <= 1, (name, int(d.max()))
        assert (d > 0).mean() <= 1e-6, (name, int((d > 0).sum()))


def test_dense_scene_pair_is_deterministic_and_what_its_docstring_says():
    """synth.dense_scene_pair (BASELINE configs[4]'s registering input): same seed, same clouds; the two clouds share no
    sample point (independent samplings of one scene); the surface density puts a few dozen neighbours inside r = 0.75 m
    and none of the neighbour lists overflows QTR_KMAX = 256; tgt ~ T @ src as SURFACES (nearest-surface distance after
    un-doing T stays within the sampling spacing)."""
    import scipy.spatial as sp
    n = 6000
    a, b, T = synth.dense_scene_pair(n, seed=11)
    a2, b2, T2 = synth.dense_scene_pair(n, seed=11)
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and np.array_equal(T, T2)
    assert a.shape == (n, 4) and a.dtype == np.float32 and np.all(a[:, 3] == 0)
    back = (b[:, :3].astype(np.float64) - T[:3, 3]) @ T[:3, :3]          # T^-1 applied to the target
    d, _ = sp.cKDTree(a[:, :3]).query(back)
    assert np.median(d) > 0.02 and np.percentile(d, 99) < 1.0           # different sample points of the same surfaces
    cnt = sp.cKDTree(a[:, :3]).query_ball_point(a[:, :3], 0.75, return_length=True)
    assert 15 < np.median(cnt) < 60 and cnt.max() < 256


def test_scout_model_finds_the_generators_planted_cliques():
    """The CPU model of k_hcore_async's scout workgroup (tests/probe/scout_sim.py; the kernel is solver.hip: hca_scout):
    from the degrees alone, the min-degree peel of the 768 largest values' sub-graph ends on the planted clique — the
    oracle's clique — and its floor s - 1 - s / 8 lies above the bulk's core numbers; nothing planted: nothing found."""
    import importlib.util
    import os
    from oracle import oracle as qo
    from quatro_amd import synth
    spec = importlib.util.spec_from_file_location("scout_sim", os.path.join(os.path.dirname(__file__), "probe", "scout_sim.py"))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    for L, frac in ((6000, 0.03), (6000, 0.0)):
        src, tgt, _, _ = synth.correspondences(L, frac, seed=5, noise=0.05)
        bm = qo.build_graph(src, tgt)
        adj = sim.unpack(np.asarray(bm).reshape(L, -1), L)
        res = sim.scout(adj, adj.sum(axis=1).astype(np.int64))
        if frac:
            clique = np.asarray(qo.max_clique(bm))
            core = np.asarray(qo.kcore(bm)[0])
            assert res[0] == clique.size == int(round(L * frac))
            assert res[0] - 1 - (res[0] >> 3) > np.median(core)
        else:
            assert res[0] == 0

