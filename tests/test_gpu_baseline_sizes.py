"""The BASELINE.json sizes against the ORACLE (not through properties): the 16-18 k-voxel scan pairs bench.py times, the
front-end entry point of the composite step, a batch of 64 ids of that pool, the back end at L = 20000, and the full
n = 50 k nearest-neighbour tables of dense mode.  Everything through the C ABI; integer outputs bit-exact."""
import numpy as np
import pytest

from quatro_amd import lib as ql
from quatro_amd import synth

pytestmark = pytest.mark.gpu
ROT_TOL, TRANS_TOL = 1e-4, 1e-3
LIMITS = dict(max_points=131072, max_voxels=32768, max_corr=8192)


def _yaw(T):
    return float(np.arctan2(T[1, 0], T[0, 0]))


def _same(r, o):
    assert r["valid"] == o["valid"]
    assert np.array_equal(r["clique"], o["clique"])
    assert np.array_equal(r["final_inliers"], o["final_inliers"])
    if r["valid"]:
        d = _yaw(r["T"]) - _yaw(o["T"])
        assert abs(np.arctan2(np.sin(d), np.cos(d))) <= ROT_TOL
        assert np.abs(r["T"][:3, 3] - o["T"][:3, 3]).max() <= TRANS_TOL
        assert np.array_equal(r["T"], o["T"])


@pytest.fixture(scope="module")
def big():
    h = ql.Handle(0, **LIMITS)
    yield h
    h.close()


@pytest.fixture(scope="module")
def qo16(qo):
    qo.set_threads(min(16, qo.max_threads()))
    yield qo
    qo.set_threads(min(8, qo.max_threads()))


@pytest.fixture(scope="module")
def pool16k():
    return [synth.kitti64_pair_16k(i) for i in range(4)]


def _oracle_front(qo, s, t, seed):
    vs, vt = qo.voxelize(s, 0.3), qo.voxelize(t, 0.3)
    ds, dt = qo.fpfh(vs, 0.5, 0.75)[2], qo.fpfh(vt, 0.5, 0.75)[2]
    return vs, vt, ds, dt, qo.match(vs, ds, vt, dt, True, True, 0.95, seed)


@pytest.mark.parametrize("pid", [0, 1, 2, 3])
def test_register_pair_on_the_bench_pool_matches_oracle(big, qo16, pool16k, pid):
    """qtr_register_pair on every pair of the bench pool (n = 15-20 k voxels per cloud) against the oracle's whole path."""
    s, t, Tgt = pool16k[pid]
    r = big.register_pair(s, t, ql.default_frontend_params(seed=pid))
    o = qo16.register_pair(s, t, seed=pid)
    assert r["n_src"] >= 13000 and r["n_tgt"] >= 13000
    assert (r["n_src"], r["n_tgt"], r["L"]) == (o["n_src"], o["n_tgt"], o["L"])
    _same(r, o)
    d = _yaw(r["T"]) - _yaw(Tgt)
    assert abs(np.arctan2(np.sin(d), np.cos(d))) < 0.02 and np.linalg.norm(r["T"][:3, 3] - Tgt[:3, 3]) < 0.5


@pytest.mark.parametrize("pid", [0, 3])
def test_feature_pair_matches_oracle_stages_and_register_pair(big, qo16, pool16k, pid):
    """qtr_feature_pair (voxelize x2 + FPFHManager::setFeaturePair, reference include/fpfh_manager.hpp:98-153): counts,
    correspondence list and matched keypoints equal the oracle's stage functions; a qtr_solve on its keypoints equals
    qtr_register_pair (the composite bench step is these two calls)."""
    s, t, _ = pool16k[pid]
    f = big.feature_pair(s, t, ql.default_frontend_params(seed=pid))
    vs, vt, ds, dt, corr = _oracle_front(qo16, s, t, pid)
    assert (f["n_src"], f["n_tgt"], f["L"]) == (vs.shape[0], vt.shape[0], corr.shape[0])
    assert np.array_equal(f["corr"], corr)
    assert np.array_equal(f["src_kps"][:, :3], vs[corr[:, 0], :3]) and np.array_equal(f["tgt_kps"][:, :3], vt[corr[:, 1], :3])
    whole = big.register_pair(s, t, ql.default_frontend_params(seed=pid))
    again = big.solve(f["src_kps"], f["tgt_kps"])
    assert np.array_equal(again["clique"], whole["clique"]) and np.array_equal(again["T"], whole["T"])
    # device-resident form: the counts come back, the matched clouds stay in the slot, the next call may follow at once
    import torch
    sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    rc, ns, nt, L = big.feature_pair_dev(sd.data_ptr(), s.shape[0], td.data_ptr(), t.shape[0],
                                         ql.default_frontend_params(seed=pid))
    assert (rc, ns, nt, L) == (ql.QTR_OK, f["n_src"], f["n_tgt"], f["L"])
    assert np.array_equal(big.debug_fetch(ql.DBG_CORR, np.int32).reshape(-1, 2)[:L], corr)


def test_feature_pair_rejects_what_the_reference_rejects(big):
    s, t, _ = synth.kitti64_pair(1)
    with pytest.raises(ql.QuatroHipError) as ei:  # fpfh_manager.hpp:101
        big.feature_pair(s, t, ql.default_frontend_params(normal_radius=1.0, fpfh_radius=0.5))
    assert ei.value.code == ql.QTR_ERR_BAD_ARG
    with pytest.raises(ql.QuatroHipError):
        big.feature_pair(np.zeros((0, 4), np.float32), t)


def test_batch_of_64_ids_of_the_bench_pool_against_oracle_records(qo16, pool16k):
    """qtr_submit_batch / qtr_wait on 64 pair ids of the 16 k pool (4 pairs x 4 tuple-test seeds) against the ORACLE's
    records — not only against sequential device runs."""
    ids = list(range(64))
    ref = {}
    for i in ids:
        key = (i % 4, (i // 4) % 4)
        if key not in ref:
            s, t, _ = pool16k[key[0]]
            ref[key] = qo16.register_pair(s, t, seed=key[1])
    hb = ql.Handle(0, n_slots=16, **LIMITS)
    try:
        got = hb.register_batch([(pool16k[i % 4][0], pool16k[i % 4][1], (i // 4) % 4) for i in ids])
        for i, g in zip(ids, got):
            o = ref[(i % 4, (i // 4) % 4)]
            assert (g["n_src"], g["n_tgt"], g["L"]) == (o["n_src"], o["n_tgt"], o["L"]), i
            _same(g, o)
    finally:
        hb.close()


def test_solver_at_L20000_matches_oracle(qo16):
    """BASELINE configs[4]'s back end: 20000 correspondences, 2 % planted inliers — bit matrix (50 MB), core numbers,
    clique, inlier sets and transform against the oracle."""
    L = 20000
    src, tgt, T, inl = synth.correspondences(L, 0.02, seed=7, noise=0.1)
    h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=24576)
    try:
        r = h.solve(src, tgt)
        bm_g = h.debug_fetch(ql.DBG_GRAPH_BITMAP, np.uint64).reshape(L, -1)
        core_g = h.debug_fetch(ql.DBG_CORE, np.int32)
        floor = int(h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)[29])
    finally:
        h.close()
    o = qo16.solve(src, tgt)
    bm_o = qo16.build_graph(src, tgt, 0.3, 1.0)
    assert np.array_equal(bm_g, bm_o)
    core_o = qo16.kcore(bm_o)[0]  # (exact at or above the floor k_hcore_async worked with, upper bounds under it)
    hi = core_o >= floor
    assert np.array_equal(core_g[:L][hi], core_o[hi])
    assert np.all(core_g[:L][~hi] >= core_o[~hi]) and np.all(core_g[:L][~hi] < floor)
    assert r["max_core"] == o["max_core"] and r["n_edges"] == o["n_edges"]
    _same(r, o)
    assert np.array_equal(r["rot_inliers"], o["rot_inliers"]) and r["gnc_iters"] == o["gnc_iters"]
    assert set(inl).issubset(set(r["clique"]))


@pytest.mark.parametrize("L,p,planted", [(3500, 0.2, 40), (4000, 0.45, 0), (8000, 0.4, 0)])
def test_core_numbers_of_dense_graphs_match_oracle(qo16, L, p, planted):
    """k_hcore_async beyond the consistency graphs of the benches (a few hundred neighbours per row): rows of 700
    neighbours (16 list entries per lane), of 1800 (lists longer than the registers hold, re-read per probe) and — at
    L = 8000, 3200 neighbours per row — more neighbours per workgroup than its LDS pool holds, so that the last rows of
    every workgroup fall back to their bit rows.  KCORE_HEU with threshold 0 returns the top core: a function of the core
    numbers alone, and cheap on the oracle's side."""
    from test_gpu_parity import _random_graph_bitmap
    bm, _ = _random_graph_bitmap(L, p, 1000 + L, planted)
    h = ql.Handle(0, max_points=4096, max_voxels=4096, max_corr=8192)
    try:
        got, max_core = h.max_clique(bm, 2, 0.0)
        core_g = h.debug_fetch(ql.DBG_CORE, np.int32)[:L]
    finally:
        h.close()
    core_o, _, mc = qo16.kcore(bm)
    assert max_core == mc
    assert np.array_equal(core_g, core_o)
    assert np.array_equal(got, qo16.max_clique(bm, 2, 0.0))


def test_dense_mode_front_end_at_50k_points_matches_oracle(qo16):
    """BASELINE configs[4]'s front end: two INDEPENDENTLY sampled 50 000-point clouds (no voxel step) through FPFH and
    matching — descriptors, both complete nearest-neighbour tables (50 k x 50 k brute force on the oracle's side) and the
    correspondence list against the oracle."""
    n = 50000
    src, tgt, T = synth.dense_pair(n, seed=7)
    h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=24576)
    try:
        nsrc, ds = h.fpfh(src, 0.5, 0.75)
        ntgt, dt = h.fpfh(tgt, 0.5, 0.75)
        os_, ot_ = qo16.fpfh(src, 0.5, 0.75), qo16.fpfh(tgt, 0.5, 0.75)
        for d, o in ((ds, os_[2]), (dt, ot_[2])):
            assert np.array_equal(np.isnan(d), np.isnan(o))
            assert np.array_equal(np.nan_to_num(d).view(np.uint32), np.nan_to_num(o).view(np.uint32))
        corr = h.match(src, ds, tgt, dt, ql.default_frontend_params(seed=3))
        nn_ij_g = h.debug_fetch(ql.DBG_NN_LARGE_OF_SMALL, np.int32)
        nn_ji_g = h.debug_fetch(ql.DBG_NN_SMALL_OF_LARGE, np.int32)
    finally:
        h.close()
    corr_o, nn_ij, nn_ji = qo16.match(src, os_[2], tgt, ot_[2], seed=3, debug=True)
    assert np.array_equal(nn_ij_g, nn_ij)          # every row of one cloud against the other: 2.5e9 distances
    assert np.array_equal(nn_ji_g, nn_ji)          # the rows the first direction pointed at (-1 elsewhere)
    assert np.array_equal(corr, corr_o)


def test_batch_of_mixed_sizes_in_one_lane_group_equals_sequential_calls():
    """Group launches pick kernel variants (level-parallel / peeling / h-index core numbers, merged first clique round,
    LDS sizing, fused matcher tails) from the LARGEST pair of the group, so a small pair batched with a large one runs
    other code than in qtr_register_pair.  One lane group holding L = 0, L < 1280, 1280 < L < 3000 and L > 3000, clouds
    on both sides of 16384 and 32768 voxels, against sequential calls."""
    rng = np.random.default_rng(5)
    s16, t16, _ = synth.kitti64_pair_16k(0)
    s9, t9, _ = synth.kitti64_pair(1)

    def jitter(a, sig):
        b = a.copy()
        b[:, :3] += rng.normal(0, sig, (a.shape[0], 3)).astype(np.float32)
        return b
    iso = np.zeros((3, 4), dtype=np.float32)
    iso[:, 0] = [0.0, 50.0, 100.0]
    wide = np.concatenate([s16, s16 + np.float32([300, 0, 0, 0]), s9 + np.float32([0, 300, 0, 0])])  # > 32768 voxels
    pairs = [(s9, t9, 1), (s16, t16, 0), (s16, jitter(s16, 0.005), 2), (s16, jitter(s16, 0.02), 3), (iso, iso.copy(), 4),
             (wide, jitter(wide, 0.01), 5), (s9, t9, 6), (iso, t9, 7)]
    lim = dict(max_points=262144, max_voxels=65536, max_corr=24576)
    h1 = ql.Handle(0, **lim)
    try:
        seq = []
        for s, t, seed in pairs:
            try:
                seq.append(h1.register_pair(s, t, ql.default_frontend_params(seed=seed)))
            except ql.QuatroHipError as e:
                seq.append({"status": e.code})
    finally:
        h1.close()
    Ls = [r.get("L", -1) for r in seq]
    assert any(L == 0 for L in Ls) and any(0 < L < 1280 for L in Ls) and any(1280 < L < 3000 for L in Ls) and \
        any(L > 3000 for L in Ls), Ls
    assert any(r.get("n_src", 0) > 32768 for r in seq) and any(16384 < r.get("n_tgt", 0) < 32768 for r in seq)
    hb = ql.Handle(0, n_slots=16, **lim)   # two lanes of 8: all eight pairs share one lane group
    try:
        got = hb.register_batch(pairs)
    finally:
        hb.close()
    for i, (g, r) in enumerate(zip(got, seq)):
        assert g["status"] == r["status"], (i, g["status"], r["status"])
        if "T" not in r:
            continue
        assert (g["n_src"], g["n_tgt"], g["L"]) == (r["n_src"], r["n_tgt"], r["L"]), i
        assert np.array_equal(g["clique"], r["clique"]) and np.array_equal(g["final_inliers"], r["final_inliers"]), i
        assert np.array_equal(g["T"], r["T"]) or not r["valid"], i


def _near_field_patch(n, seed, side=6.0):
    """An un-voxelised near-field patch: n points on a gently curved, slightly thick surface of side x side metres —
    hundreds to thousands of neighbours inside r = 0.75 m."""
    g = np.random.default_rng(seed)
    p = np.zeros((n, 4), dtype=np.float32)
    p[:, 0] = g.uniform(0, side, n)
    p[:, 1] = g.uniform(0, side, n)
    p[:, 2] = 0.15 * np.sin(1.3 * p[:, 0]) * np.cos(0.9 * p[:, 1]) + g.normal(0, 0.01, n)
    return p


@pytest.mark.parametrize("n,side", [(12000, 6.0), (20000, 4.0), (9000, 1.4)])
def test_fpfh_with_neighbour_lists_longer_than_256_matches_oracle(qo16, n, side):
    """pcl's radius search has no cap (reference src/teaser_utils/fpfh.cc:58-72): an un-voxelised near-field patch with
    ~300 - 2000 neighbours per point goes through qtr_fpfh — lists of more than 256 entries live in the long-list arena —
    and normals, SPFH and FPFH equal the oracle's bit for bit."""
    pts = _near_field_patch(n, 5 + n, side)
    off_o, idx_o, d2_o = qo16.radius_neighbors(pts, 0.75)
    cnt = np.diff(off_o)
    assert cnt.max() > 256 and np.median(cnt) > 256, (int(cnt.max()), float(np.median(cnt)))
    h = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=8192, max_long_neighbors=int(off_o[-1] * 1.2) + 65536)
    try:
        nrm_g, de_g = h.fpfh(pts, 0.5, 0.75)
        off_g = h.debug_fetch(ql.DBG_NBR_OFFSETS, np.int32)
        sp_g = h.debug_fetch(ql.DBG_SPFH, np.float32).reshape(n, 33)
    finally:
        h.close()
    nrm_o, sp_o, de_o = qo16.fpfh(pts, 0.5, 0.75)
    assert np.array_equal(off_g.astype(np.int64), off_o)

    def same(a, b):
        a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        return np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))
    assert same(nrm_g, nrm_o) and same(sp_g, sp_o) and same(de_g, de_o)


def test_register_pair_on_dense_patches_switches_to_long_lists(qo16):
    """The whole-path entry point leaves k2_neighbors_big out of its chain until a cloud needs it: a pair of dense
    patches voxelised at a 5 cm leaf (hundreds of centroids inside r = 0.75 m) makes it run the stage again with the
    long lists — same answer as the oracle, no QTR_ERR_CAPACITY — and an arena that is too small is reported as such."""
    s = _near_field_patch(30000, 3, 5.0)
    R = synth.yaw_matrix(0.3)
    t = s.copy()
    t[:, :3] = (s[:, :3].astype(np.float64) @ R.T + np.array([0.4, -0.2, 0.05])).astype(np.float32)
    t[:, :3] += np.random.default_rng(9).normal(0, 0.002, (t.shape[0], 3)).astype(np.float32)
    fp = ql.default_frontend_params(seed=2, voxel_size=0.05)
    h = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=16384, max_long_neighbors=24 << 20)
    try:
        r = h.register_pair(s, t, fp)
        r2 = h.register_pair(s, t, fp)  # now with the launch in the chain from the start
    finally:
        h.close()
    o = qo16.register_pair(s, t, leaf=0.05, seed=2)
    assert (r["n_src"], r["n_tgt"], r["L"]) == (o["n_src"], o["n_tgt"], o["L"])
    _same(r, o)
    assert np.array_equal(r2["T"], r["T"]) and np.array_equal(r2["clique"], r["clique"])
    hs = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=16384, max_long_neighbors=4096)
    try:
        with pytest.raises(ql.QuatroHipError) as ei:
            hs.register_pair(s, t, fp)
        assert ei.value.code == ql.QTR_ERR_CAPACITY and "max_long_neighbors" in str(ei.value)
    finally:
        hs.close()


@pytest.mark.parametrize("tuple_test", [1, 0])
def test_batch_without_cross_check_equals_sequential_calls_and_oracle(qo16, tuple_test):
    """calculateCorrespondences(..., use_crosscheck = false, ...) (reference feature_matcher.cc:146-181) through the batched
    entry points: corres_ij + corres_ji, with and without the tuple test, for a group of pairs — the records equal the
    per-pair entry point's and the oracle's."""
    pairs = [synth.kitti64_pair(i) for i in (1, 2)]
    fp = ql.default_frontend_params(use_crosscheck=0, use_tuple_test=tuple_test)
    lim = dict(max_points=131072, max_voxels=32768, max_corr=24576)
    h1 = ql.Handle(0, **lim)
    try:
        seq = []
        for i, (s, t, _) in enumerate(pairs):
            f = ql.default_frontend_params(use_crosscheck=0, use_tuple_test=tuple_test, seed=10 + i)
            seq.append(h1.register_pair(s, t, f))
    finally:
        h1.close()
    hb = ql.Handle(0, n_slots=4, **lim)
    try:
        got = hb.register_batch([(s, t, 10 + i) for i, (s, t, _) in enumerate(pairs)], fp)
    finally:
        hb.close()
    for i, (g, r) in enumerate(zip(got, seq)):
        assert (g["n_src"], g["n_tgt"], g["L"]) == (r["n_src"], r["n_tgt"], r["L"]), i
        assert r["L"] > (5000 if not tuple_test else 100)
        assert np.array_equal(g["clique"], r["clique"]) and np.array_equal(g["final_inliers"], r["final_inliers"]), i
        assert np.array_equal(g["T"], r["T"]), i
    # the oracle on the first pair: the same list length and solution
    s, t, _ = pairs[0]
    vs, vt = qo16.voxelize(s, 0.3), qo16.voxelize(t, 0.3)
    ds, dt = qo16.fpfh(vs, 0.5, 0.75)[2], qo16.fpfh(vt, 0.5, 0.75)[2]
    corr = qo16.match(vs, ds, vt, dt, False, bool(tuple_test), 0.95, 10)
    assert corr.shape[0] == got[0]["L"]
    o = qo16.solve(vs[corr[:, 0]], vt[corr[:, 1]])
    assert np.array_equal(got[0]["clique"], o["clique"]) and np.array_equal(got[0]["T"], o["T"])


def test_batch_on_raw_sweeps_runs_the_demo_sequence(qo16):
    """qtr_set_batch_preprocess: the batched entry on raw sweeps WITH their ground returns — per scan
    PatchWork::estimate_ground -> ImageProjection::segmentCloud -> valid segments, then the usual chain (reference
    examples/run_global_registration.cpp:136-160, 206-246) — against the same sequence through the oracle."""
    scans = [synth.kitti64_raw_scan(i)[0] for i in range(3)]
    pairs = [(scans[0], scans[1], 4), (scans[1], scans[2], 5), (scans[2], scans[0], 6)]
    # (four slots = two lanes of two: the pre-processing of a chunk's pairs runs side by side, one slot each)
    ref = []
    for a, b, seed in pairs:
        clouds = [qo16.segment_cloud(qo16.patchwork(raw)["nonground"])["valid"] for raw in (a, b)]
        ref.append(qo16.register_pair(clouds[0], clouds[1], seed=seed))
    hb = ql.Handle(0, n_slots=4, max_points=131072, max_voxels=32768, max_corr=8192)
    try:
        hb.set_batch_preprocess()
        got = hb.register_batch(pairs)
        hb.set_batch_preprocess(on=False)
        plain = hb.register_batch([(scans[0], scans[1], 4)])  # the same handle without it: the raw sweep as it is
    finally:
        hb.close()
    for i, (g, o) in enumerate(zip(got, ref)):
        assert (g["n_src"], g["n_tgt"], g["L"]) == (o["n_src"], o["n_tgt"], o["L"]), i
        _same(g, o)
    assert plain[0]["n_src"] != got[0]["n_src"]


def test_batch_on_raw_sweeps_many_pairs_in_flight_and_a_scan_that_is_all_ground(qo16):
    """Eight slots: the raw-sweep stages of four pairs per lane advance side by side (a host-driven step per scan and
    stage).  Every record equals the same pair registered alone through the same entry; a sweep that is nothing but ground
    fails in its own record (QTR_ERR_BAD_ARG: an empty cloud) without disturbing its neighbours; host memory."""
    scans = [synth.kitti64_raw_scan(i)[0] for i in range(3)]
    flat = np.zeros((20000, 4), dtype=np.float32)   # a ground plane at the sensor's height and nothing else
    g = np.random.default_rng(3)
    flat[:, 0] = g.uniform(-30, 30, 20000)
    flat[:, 1] = g.uniform(-30, 30, 20000)
    flat[:, 2] = -1.723 + g.normal(0, 0.01, 20000)
    pairs = [(scans[i % 3], scans[(i + 1) % 3], 10 + i) for i in range(7)]
    pairs.insert(3, (scans[0], flat, 99))
    h1 = ql.Handle(0, n_slots=2, max_points=131072, max_voxels=32768, max_corr=8192)
    hb = ql.Handle(0, n_slots=8, max_points=131072, max_voxels=32768, max_corr=8192)
    try:
        h1.set_batch_preprocess()
        hb.set_batch_preprocess()
        alone = [h1.register_batch([p])[0] for p in pairs]
        got = hb.register_batch(pairs)
    finally:
        h1.close()
        hb.close()
    assert got[3]["status"] == ql.QTR_ERR_BAD_ARG and alone[3]["status"] == ql.QTR_ERR_BAD_ARG
    for i, (a, b) in enumerate(zip(got, alone)):
        if i == 3:
            continue
        assert (a["n_src"], a["n_tgt"], a["L"]) == (b["n_src"], b["n_tgt"], b["L"]), i
        assert np.array_equal(a["clique"], b["clique"]) and np.array_equal(a["T"], b["T"]), i
    o = qo16.register_pair(qo16.segment_cloud(qo16.patchwork(pairs[5][0])["nonground"])["valid"],
                           qo16.segment_cloud(qo16.patchwork(pairs[5][1])["nonground"])["valid"], seed=pairs[5][2])
    _same(got[5], o)


# ------------------------------------------------------------------------------------------------
# batched entry on pre-matched correspondences (qtr_pair_desc.src_corr4 / tgt_corr4 / n_corr): the reference's loop is one
# Quatro object, reset(), setInputSource / setInputTarget / computeTransformation on whatever it is handed
# (examples/run_global_registration.cpp:97-108,243-246; include/quatro.hpp:769)
def _same_back_end(g, r):
    assert g["status"] == r["status"] and g["valid"] == r["valid"]
    assert g["L"] == r["L"]
    assert np.array_equal(g["clique"], r["clique"]) and np.array_equal(g["final_inliers"], r["final_inliers"])
    assert np.array_equal(g["T"], r["T"]) or not r["valid"]


def test_batch_of_composite_pairs_at_the_metrics_size_against_sequential_calls_and_oracle(qo16, pool16k):
    """BASELINE configs[2] in the unit of work the metric is quoted on: every pair id = the front end of a 16-18 k-voxel
    scan pair AND the back end on 5000 given correspondences (5 % planted inliers).  64 ids through qtr_submit_batch /
    qtr_wait against (a) sequential qtr_feature_pair + qtr_solve and (b) the ORACLE's solve on the same correspondences."""
    LC = 5000
    corr = [synth.correspondences(LC, 0.05, seed=k, noise=0.1) for k in range(8)]
    ids = list(range(64))
    h1 = ql.Handle(0, **LIMITS)
    try:
        seq_front = [h1.feature_pair(s, t, ql.default_frontend_params(seed=0)) for (s, t, _) in pool16k]
        seq_back = [h1.solve(c[0], c[1]) for c in corr]
    finally:
        h1.close()
    ora = [qo16.solve(c[0], c[1]) for c in corr]
    hb = ql.Handle(0, n_slots=16, **LIMITS)
    try:
        got = hb.register_batch([(pool16k[i % 4][0], pool16k[i % 4][1], 0, corr[i % 8][0], corr[i % 8][1]) for i in ids])
    finally:
        hb.close()
    for i, g in zip(ids, got):
        f, b, o = seq_front[i % 4], seq_back[i % 8], ora[i % 8]
        assert (g["n_src"], g["n_tgt"], g["L"]) == (f["n_src"], f["n_tgt"], LC), i
        _same_back_end(g, dict(b, L=LC))
        _same(g, o)
        assert g["valid"] and np.intersect1d(g["final_inliers"], corr[i % 8][3]).size >= 0.9 * corr[i % 8][3].size


def test_batch_of_correspondence_only_pairs_of_mixed_sizes_against_sequential_calls_and_oracle(qo16):
    """Pair descriptors without scans: the back end alone, batched.  One lane group mixing L = 0, 3, 100, 1281 (the lower
    end of the h-index kernel), 2000, 5000 and 8000 — kernel variants are picked from the LARGEST pair of a group —
    against sequential qtr_solve calls and the oracle; PMC_HEU and KCORE_HEU."""
    sizes = [5000, 100, 0, 1281, 8000, 3, 2000, 5000, 640, 1279]
    sets = []
    for k, L in enumerate(sizes):
        if L == 0:
            z = np.zeros((0, 4), dtype=np.float32)
            sets.append((z, z.copy()))
        else:
            c = synth.correspondences(max(L, 3), 0.06 if L >= 100 else 1.0, seed=40 + k, noise=0.1)
            sets.append((c[0][:L], c[1][:L]))
    h1 = ql.Handle(0, **LIMITS)
    hb = ql.Handle(0, n_slots=24, **LIMITS)   # two lanes of 12: the ten pairs share one lane group
    try:
        for mode in (ql.INLIER_PMC_HEU, ql.INLIER_KCORE_HEU):
            prm = ql.demo_params(inlier_selection_mode=mode)
            seq = [h1.solve(s, t, prm) for (s, t) in sets]
            got = hb.register_batch([(None, None, 0, s, t) for (s, t) in sets], params=prm)
            for i, (g, r) in enumerate(zip(got, seq)):
                assert (g["n_src"], g["n_tgt"]) == (0, 0)
                _same_back_end(g, dict(r, L=sizes[i]))
            if mode == ql.INLIER_PMC_HEU:
                for i in (0, 3, 4, 6):
                    _same(got[i], qo16.solve(*sets[i]))
    finally:
        h1.close()
        hb.close()


def _noisy_block_correspondences(L, nblk, amp, seed):
    """nblk correspondences that follow one transform up to a displacement of up to `amp` metres (every other pair of them
    consistent at amp = 1.2: a dense block of the consistency graph that is not a clique), the rest random."""
    src, tgt, _, inl = synth.correspondences(L, nblk / L, seed, noise=0.05)
    rng = np.random.default_rng(seed + 100)
    e = rng.normal(size=(len(inl), 3))
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    e *= amp * rng.random((len(inl), 1)) ** (1 / 3)
    tgt = tgt.copy()
    tgt[inl, :3] += e.astype(np.float32)
    return src, tgt


def test_second_run_of_the_clique_stage_single_and_inside_a_batch(qo16):
    """A dense block that is not a clique lifts the h-index of the degrees (floor ~ 94) far above the largest clique (41):
    the clique search under k_hcore_async's floor comes back empty and the stage runs a second time with exact core numbers
    — as a single call (the state says so) and for pairs in the middle of a lane group of a batch, beside pairs that keep
    their floor; every record equals the sequential call's and the oracle's."""
    blk = _noisy_block_correspondences(3000, 300, 1.2, 3)
    blk2 = _noisy_block_correspondences(4000, 360, 1.3, 9)
    good = synth.correspondences(5000, 0.05, seed=4, noise=0.1)
    good2 = synth.correspondences(3000, 0.1, seed=5, noise=0.2)
    sets = [good[:2], blk, good2[:2], blk2, good[:2]]
    h1 = ql.Handle(0, **LIMITS)
    hb = ql.Handle(0, n_slots=12, **LIMITS)   # two lanes of 6: the five pairs share one lane group
    try:
        seq = []
        for k, (s, t) in enumerate(sets):
            seq.append(h1.solve(s, t))
            st = h1.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
            assert (st[22] == 1 and st[29] == 0) if k in (1, 3) else (st[22] == 0 and st[29] > 0), (k, st[22], st[29])
            _same(seq[-1], qo16.solve(s, t))
        got = hb.register_batch([(None, None, 0, s, t) for (s, t) in sets])
        for i, (g, r) in enumerate(zip(got, seq)):
            _same_back_end(g, dict(r, L=len(sets[i][0])))
    finally:
        h1.close()
        hb.close()


def test_batch_mixing_the_three_kinds_of_pairs_and_per_pair_failures(qo16):
    """One batch holding scan-only pairs (the matcher's own correspondences), correspondence-only pairs and pairs with
    both, plus descriptors the entry has to refuse pair by pair: each record equals the matching sequential call, the
    failures carry their own status and the rest of the batch is unaffected.  Host memory (the staging copies)."""
    s9, t9, _ = synth.kitti64_pair(1)
    c1 = synth.correspondences(1500, 0.1, seed=3, noise=0.1)
    c2 = synth.correspondences(400, 0.2, seed=4, noise=0.1)
    too_many = synth.correspondences(9000, 0.05, seed=5, noise=0.1)
    pairs = [(s9, t9, 7), (None, None, 0, c1[0], c1[1]), (s9, t9, 7, c2[0], c2[1]), (None, None, 0, too_many[0], too_many[1]),
             (s9, t9, 8), (s9, t9, 7, c1[0], c1[1]), (None, None, 0, c2[0], c2[1])]
    h1 = ql.Handle(0, **LIMITS)
    hb = ql.Handle(0, n_slots=4, **LIMITS)    # two lanes of 2: several chunks, both lanes
    try:
        whole7 = h1.register_pair(s9, t9, ql.default_frontend_params(seed=7))
        whole8 = h1.register_pair(s9, t9, ql.default_frontend_params(seed=8))
        b1, b2 = h1.solve(c1[0], c1[1]), h1.solve(c2[0], c2[1])
        got = hb.register_batch(pairs)
        assert got[3]["status"] == ql.QTR_ERR_CAPACITY
        for g, r in ((got[0], whole7), (got[4], whole8)):
            assert (g["n_src"], g["n_tgt"], g["L"]) == (r["n_src"], r["n_tgt"], r["L"])
            _same_back_end(g, r)
        for g, r, L in ((got[1], b1, 1500), (got[6], b2, 400)):
            assert (g["n_src"], g["n_tgt"]) == (0, 0)
            _same_back_end(g, dict(r, L=L))
        for g, r, L in ((got[2], b2, 400), (got[5], b1, 1500)):
            assert (g["n_src"], g["n_tgt"]) == (whole7["n_src"], whole7["n_tgt"])
            _same_back_end(g, dict(r, L=L))
        _same(got[1], qo16.solve(c1[0], c1[1]))
        # descriptors with half a correspondence set, or nothing at all: QTR_ERR_BAD_ARG in the pair's own record
        import ctypes as C
        descs = (ql.PairDesc * 3)()
        res = (ql.Result * 3)()
        a, b = np.ascontiguousarray(c2[0]), np.ascontiguousarray(c2[1])
        descs[0] = ql.PairDesc(None, 0, None, 0, 0, None, None, 0, a.ctypes.data, None, 400)
        descs[1] = ql.PairDesc(None, 0, None, 0, 0, None, None, 0, None, None, 0)
        descs[2] = ql.PairDesc(None, 0, None, 0, 0, None, None, 0, a.ctypes.data, b.ctypes.data, 400)
        fp, prm = ql.default_frontend_params(), ql.demo_params()
        assert hb._lib.qtr_submit_batch(hb._h, descs, 3, C.byref(fp), C.byref(prm), res, ql.MEM_HOST) == ql.QTR_OK
        assert hb._lib.qtr_wait(hb._h) == ql.QTR_OK
        assert [res[i].status for i in range(3)] == [ql.QTR_ERR_BAD_ARG, ql.QTR_ERR_BAD_ARG, ql.QTR_OK]
        assert res[2].n_clique == b2["clique"].size and res[2].n_corr == 400
    finally:
        h1.close()
        hb.close()


def test_batch_on_device_resident_scans_and_correspondences(pool16k):
    """The same through QTR_MEM_DEVICE (what bench.py's batch256 leg runs): torch tensors in HBM, nothing staged."""
    import torch
    dev = torch.device("cuda", 0)
    items = []
    for k in range(6):
        s, t, _ = pool16k[k % 4]
        c = synth.correspondences(5000, 0.05, seed=k, noise=0.1)
        items.append({"src": torch.from_numpy(s).to(dev), "tgt": torch.from_numpy(t).to(dev),
                      "fp": ql.default_frontend_params(seed=k), "cs": torch.from_numpy(c[0]).to(dev),
                      "ct": torch.from_numpy(c[1]).to(dev), "host": c})
    torch.cuda.synchronize()
    h1 = ql.Handle(0, **LIMITS)
    hb = ql.Handle(0, n_slots=8, **LIMITS)
    try:
        prm = ql.demo_params()
        seq = [h1.solve(it["host"][0], it["host"][1], prm) for it in items]
        for kw in (dict(scans=True, corr=True), dict(scans=False, corr=True)):
            got = hb.register_batch_dev(items, prm, **kw)
            for g, r in zip(got, seq):
                assert g["status"] == r["status"] and g["L"] == 5000 and g["n_clique"] == r["clique"].size
                assert g["n_final"] == r["final_inliers"].size and np.array_equal(g["T"], r["T"])
                assert (g["n_src"] > 10000) == kw["scans"]
    finally:
        h1.close()
        hb.close()


# ------------------------------------------------------------------------------------------------
# data-connected registrations with L in the thousands: the matcher's OWN output feeds the back end
def _oracle_connected(qo, s, t, leaf, crosscheck, tuple_test, seed):
    vs, vt = qo.voxelize(s, leaf), qo.voxelize(t, leaf)
    ds, dt = qo.fpfh(vs, 0.5, 0.75)[2], qo.fpfh(vt, 0.5, 0.75)[2]
    corr = qo.match(vs, ds, vt, dt, crosscheck, tuple_test, 0.95, seed)
    return vs, vt, corr, qo.solve(vs[corr[:, 0]], vt[corr[:, 1]])


@pytest.mark.parametrize("crosscheck,lo,hi", [(1, 1200, 6000), (0, 12000, 32768)])
def test_connected_registration_without_tuple_test_on_the_bench_pool_matches_oracle(qo16, pool16k, crosscheck, lo, hi):
    """qtr_register_pair on a 16-18 k-voxel pair with use_tuple_test = 0 (mutual nearest neighbours: L ~ 2 k,
    feature_matcher.cc:187-247 skipped) and with use_crosscheck = 0 as well (corres_ij + corres_ji de-duplicated,
    :124-181: L ~ n_s + n_hit ~ 20 k): ONE registration whose back end runs on thousands of the matcher's own
    correspondences, against the oracle's stages composed the same way."""
    s, t, _ = pool16k[0]
    h = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=32768)
    try:
        fp = ql.default_frontend_params(use_crosscheck=crosscheck, use_tuple_test=0, seed=0)
        g = h.register_pair(s, t, fp)
    finally:
        h.close()
    vs, vt, corr, o = _oracle_connected(qo16, s, t, 0.3, bool(crosscheck), False, 0)
    assert lo < corr.shape[0] < hi, corr.shape
    assert (g["n_src"], g["n_tgt"], g["L"]) == (vs.shape[0], vt.shape[0], corr.shape[0])
    _same(g, o)


@pytest.mark.parametrize("pid", [0, 1, 2, 3])
def test_connected_registration_at_the_metrics_5k_correspondences_matches_oracle(qo16, pool16k, pid):
    """The metric's size DATA-CONNECTED (bench.py connected_leg.l5k / `connected_l5k`): qtr_register_pair on a 64-beam scan
    pair at a 0.07 m leaf (n ~ 35-40 k voxels per cloud) with cross check and without the tuple test — the front end's OWN
    output is 4.2-5.9 k mutual nearest neighbours (reference src/teaser_utils/feature_matcher.cc:113-181), and ONE call
    registers them (examples/run_global_registration.cpp:206-246).  Every output equals the oracle's stages composed the same
    way; the registration lands inside the noise bound."""
    s, t, Tgt = pool16k[pid]
    h = ql.Handle(0, max_points=131072, max_voxels=65536, max_corr=8192)
    try:
        g = h.register_pair(s, t, ql.default_frontend_params(voxel_size=0.07, use_tuple_test=0, seed=pid))
    finally:
        h.close()
    vs, vt, corr, o = _oracle_connected(qo16, s, t, 0.07, True, False, pid)
    assert 4000 < corr.shape[0] < 6000, corr.shape
    assert (g["n_src"], g["n_tgt"], g["L"]) == (vs.shape[0], vt.shape[0], corr.shape[0])
    _same(g, o)
    assert g["valid"] and g["clique"].size > 300 and g["final_inliers"].size > 300
    dy = _yaw(g["T"]) - _yaw(Tgt)
    assert abs(np.arctan2(np.sin(dy), np.cos(dy))) < 5e-3 and np.linalg.norm(g["T"][:3, 3] - Tgt[:3, 3]) < 0.3  # noise bound


def test_connected_registration_of_18k_point_clouds_with_5k_correspondences_matches_oracle(qo16):
    """The headline's n AND L from one input: two independent 18 000-point samplings of the structured scene (no voxel
    step: the grid would overflow and passes the cloud through), use_tuple_test = 0 -> 4999 mutual nearest neighbours."""
    a, b, Tgt = synth.dense_scene_pair(18000)
    h = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=8192)
    try:
        g = h.register_pair(a, b, ql.default_frontend_params(voxel_size=0.001, use_tuple_test=0, seed=1))
    finally:
        h.close()
    ds, dt = qo16.fpfh(a, 0.5, 0.75)[2], qo16.fpfh(b, 0.5, 0.75)[2]
    corr = qo16.match(a, ds, b, dt, True, False, 0.95, 1)
    assert 4500 < corr.shape[0] < 5500 and (g["n_src"], g["n_tgt"], g["L"]) == (18000, 18000, corr.shape[0])
    _same(g, qo16.solve(a[corr[:, 0]], b[corr[:, 1]]))
    assert g["valid"] and g["final_inliers"].size > 20
    dy = _yaw(g["T"]) - _yaw(Tgt)
    assert abs(np.arctan2(np.sin(dy), np.cos(dy))) < 5e-3 and np.linalg.norm(g["T"][:3, 3] - Tgt[:3, 3]) < 0.3


def test_neighbour_grid_cell_counters_are_clean_after_a_refused_registration(pool16k):
    """The FPFH chain's dense cell table is zero between uses: k2_cell_count fills it and k2_cell_scan leaves it zero
    (frontend.hip).  Registrations that are REFUSED behind the voxel stage (more voxels than max_voxels), and grids of different
    sizes on the same table, must leave nothing behind: the next registrations on the same handle equal a fresh handle's."""
    s, t, _ = pool16k[2]
    fp = ql.default_frontend_params(seed=2)
    fresh = ql.Handle(0, **LIMITS)
    try:
        want = fresh.register_pair(s, t, fp)
    finally:
        fresh.close()
    h = ql.Handle(0, max_points=131072, max_voxels=8192, max_corr=8192)
    try:
        with pytest.raises(ql.QuatroHipError) as e:
            h.register_pair(s, t, fp)                       # ~16 k voxels per cloud: refused after the voxel stage
        assert e.value.code == ql.QTR_ERR_CAPACITY
        coarse = ql.default_frontend_params(voxel_size=0.6, seed=2)
        a = h.register_pair(s, t, coarse)                   # fits: runs on the table the refused call left behind
        with pytest.raises(ql.QuatroHipError):
            h.register_pair(s, t, fp)
        b = h.register_pair(s, t, coarse)
    finally:
        h.close()
    h2 = ql.Handle(0, **LIMITS)
    try:
        c = h2.register_pair(s, t, coarse)
        d = h2.register_pair(s, t, fp)                      # (and a second grid of another size on the same table)
    finally:
        h2.close()
    for x in (a, b):
        assert (x["n_src"], x["n_tgt"], x["L"]) == (c["n_src"], c["n_tgt"], c["L"])
        assert np.array_equal(x["clique"], c["clique"]) and np.array_equal(x["T"], c["T"])
    assert (d["n_src"], d["n_tgt"], d["L"]) == (want["n_src"], want["n_tgt"], want["L"])
    assert np.array_equal(d["clique"], want["clique"]) and np.array_equal(d["T"], want["T"])


def test_dense_mode_end_to_end_through_the_whole_path_entry_matches_oracle(qo16):
    """BASELINE configs[4] as ONE registration: two independently sampled 50 000-point clouds and a leaf so small that the
    voxel grid would overflow int32 — pcl::VoxelGrid passes the cloud through unchanged, and so does the reference's
    `voxelize` (include/quatro.hpp:49-68) — then FPFH, matching (its own ~1.1 k correspondences) and the back end.
    Per pair and through the batched entry, against the oracle's stages."""
    a, b, T = synth.dense_pair(50000)
    lim = dict(max_points=65536, max_voxels=65536, max_corr=24576)
    fp = ql.default_frontend_params(voxel_size=0.001, seed=1)
    h = ql.Handle(0, **lim)
    try:
        g = h.register_pair(a, b, fp)
    finally:
        h.close()
    assert (g["n_src"], g["n_tgt"]) == (50000, 50000) and g["L"] > 500
    assert np.array_equal(qo16.voxelize(a, 0.001), a)          # the oracle's voxel grid passes through too
    ds, dt = qo16.fpfh(a, 0.5, 0.75)[2], qo16.fpfh(b, 0.5, 0.75)[2]
    corr = qo16.match(a, ds, b, dt, True, True, 0.95, 1)
    assert corr.shape[0] == g["L"]
    o = qo16.solve(a[corr[:, 0]], b[corr[:, 1]])
    _same(g, o)   # (parity only: two independent samplings of smooth surfaces give the matcher little to agree on)
    hb = ql.Handle(0, n_slots=2, **lim)
    try:
        gb = hb.register_batch([(a, b, 1), (a[:30000], b[:30000], 1)], fp)
    finally:
        hb.close()
    assert (gb[0]["n_src"], gb[0]["n_tgt"], gb[0]["L"]) == (50000, 50000, g["L"])
    assert np.array_equal(gb[0]["clique"], g["clique"]) and np.array_equal(gb[0]["T"], g["T"])
    assert (gb[1]["n_src"], gb[1]["n_tgt"]) == (30000, 30000)


def test_single_call_composite_equals_the_two_stage_calls_and_the_batched_entry(qo16, pool16k):
    """qtr_register_pair_corr (front end of the scans + back end on given correspondences, ONE call — the bench's step) gives
    the voxel counts / matcher count of qtr_feature_pair and the record of qtr_solve on the same correspondences, on host
    and device memory, and refuses what the two calls refuse."""
    s, t, _ = pool16k[1]
    c = synth.correspondences(5000, 0.05, seed=3, noise=0.1)
    h = ql.Handle(0, **LIMITS)
    try:
        fp = ql.default_frontend_params(seed=1)
        f = h.feature_pair(s, t, fp)
        b = h.solve(c[0], c[1])
        g = h.register_pair_corr(s, t, c[0], c[1], fp)
        assert (g["n_src"], g["n_tgt"], g["n_matched"], g["L"]) == (f["n_src"], f["n_tgt"], f["L"], 5000)
        _same_back_end(g, dict(b, L=5000))
        _same(g, qo16.solve(c[0], c[1]))
        g0 = h.register_pair_corr(s, t, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), fp)
        assert g0["status"] == ql.QTR_ERR_CLIQUE_TOO_SMALL and not g0["valid"] and g0["n_src"] == f["n_src"]
        big = synth.correspondences(9000, 0.05, seed=5, noise=0.1)
        with pytest.raises(ql.QuatroHipError) as ei:
            h.register_pair_corr(s, t, big[0], big[1], fp)
        assert ei.value.code == ql.QTR_ERR_CAPACITY
        with pytest.raises(ql.QuatroHipError):
            h.register_pair_corr(np.zeros((0, 4), np.float32), t, c[0], c[1], fp)
    finally:
        h.close()


@pytest.mark.parametrize("host_mem", [True, False])
def test_batch_long_list_fallback_with_given_correspondences_and_preprocessed_input(qo16, host_mem):
    """A fresh handle's batch chains leave k2_neighbors_big out until a cloud needs it; the pair that does is registered on
    its own slot — on the CLOUDS THE GROUP'S VOXEL GRID READ and on the caller's correspondences when it brought some (the
    fallback of round 3 went back to the descriptor's raw scans and the matcher's own list).  Dense patches at a 5 cm
    leaf force the fallback; the pair brings 1500 correspondences; host and device memory."""
    s = _near_field_patch(30000, 3, 5.0)
    R = synth.yaw_matrix(0.3)
    t = s.copy()
    t[:, :3] = (s[:, :3].astype(np.float64) @ R.T + np.array([0.4, -0.2, 0.05])).astype(np.float32)
    t[:, :3] += np.random.default_rng(9).normal(0, 0.002, (t.shape[0], 3)).astype(np.float32)
    c = synth.correspondences(1500, 0.1, seed=3, noise=0.1)
    s9, t9, _ = synth.kitti64_pair(1)
    fp = ql.default_frontend_params(seed=2, voxel_size=0.05)
    lim = dict(max_points=65536, max_voxels=32768, max_corr=16384, max_long_neighbors=24 << 20)
    h1 = ql.Handle(0, **lim)
    try:
        whole = h1.register_pair(s, t, fp)      # (switches h1 to long lists on the way)
        back = h1.solve(c[0], c[1])
    finally:
        h1.close()
    hb = ql.Handle(0, n_slots=4, **lim)           # fresh: its first chains run without the long-list launch
    try:
        if host_mem:
            got = hb.register_batch([(s, t, 2, c[0], c[1]), (s, t, 2)], fp)
        else:
            import torch
            dev = torch.device("cuda", 0)
            it = {"src": torch.from_numpy(s).to(dev), "tgt": torch.from_numpy(t).to(dev), "fp": fp,
                  "cs": torch.from_numpy(c[0]).to(dev), "ct": torch.from_numpy(c[1]).to(dev)}
            torch.cuda.synchronize()
            got = hb.register_batch_dev([it], ql.demo_params(), fp, corr=True) + hb.register_batch_dev([it], ql.demo_params(), fp)
    finally:
        hb.close()
    assert (got[0]["n_src"], got[0]["n_tgt"], got[0]["L"]) == (whole["n_src"], whole["n_tgt"], 1500)
    assert got[0]["valid"] == back["valid"] and np.array_equal(got[0]["T"], back["T"])
    assert (got[1]["n_src"], got[1]["n_tgt"], got[1]["L"]) == (whole["n_src"], whole["n_tgt"], whole["L"])
    assert np.array_equal(got[1]["T"], whole["T"])
    if host_mem:
        assert np.array_equal(got[0]["clique"], back["clique"]) and np.array_equal(got[1]["clique"], whole["clique"])


# ---- BASELINE configs[4] as ONE registration (round 5) ----------------------------------------------------------------
DENSE_LIMITS = dict(max_points=65536, max_voxels=65536, max_corr=24576)


@pytest.fixture(scope="module")
def dense_scene(qo16):
    """Two independent 50 000-point samplings of one structured scene + the oracle's descriptors of both (shared by the
    dense tests below: ~2 s of FPFH and one 50 k x 50 k brute-force matcher run per flag combination on the CPU side)."""
    a, b, T = synth.dense_scene_pair(50000)
    assert np.array_equal(qo16.voxelize(a, 0.001), a) and np.array_equal(qo16.voxelize(b, 0.001), b)  # pass-through
    return {"a": a, "b": b, "T": T, "da": qo16.fpfh(a, 0.5, 0.75)[2], "db": qo16.fpfh(b, 0.5, 0.75)[2]}


def test_dense_step_front_end_of_50k_clouds_and_back_end_on_20000_correspondences_in_one_call(qo16, dense_scene):
    """configs[4] as the bench's dense_step_leg runs it: ONE qtr_register_pair_corr call — the front end of two
    50 000-point clouds (no voxel down-sampling) and the back end on 20 000 given correspondences — against the oracle's
    front end (voxel counts, the matcher's own count) and the oracle's solve of the same correspondences."""
    d = dense_scene
    c = synth.correspondences(20000, 0.02, seed=7, noise=0.1)
    fp = ql.default_frontend_params(voxel_size=0.001, seed=1)
    h = ql.Handle(0, **DENSE_LIMITS)
    try:
        g = h.register_pair_corr(d["a"], d["b"], c[0], c[1], fp)
        g2 = h.register_pair_corr(d["a"], d["b"], c[0], c[1], fp)   # (a second call on warm arenas: the same record)
    finally:
        h.close()
    corr = qo16.match(d["a"], d["da"], d["b"], d["db"], True, True, 0.95, 1)
    assert (g["n_src"], g["n_tgt"], g["n_matched"], g["L"]) == (50000, 50000, corr.shape[0], 20000)
    o = qo16.solve(c[0], c[1])
    _same(g, o)
    assert set(c[3]).issubset(set(g["clique"])) and g["valid"]
    assert np.array_equal(g2["clique"], g["clique"]) and np.array_equal(g2["T"], g["T"])


@pytest.mark.parametrize("tuple_test,lo,hi", [(1, 1500, 3000), (0, 11000, 17000)])
def test_dense_scene_registers_through_the_whole_path_entry_and_matches_oracle(qo16, dense_scene, tuple_test, lo, hi):
    """configs[4] DATA-CONNECTED: qtr_register_pair on two independent 50 000-point samplings of one structured scene (no
    point of one cloud is a moved copy of a point of the other), the matcher's own correspondences into the back end —
    L ~ 2.1 k with the tuple test, L ~ 14 k without it (every mutual nearest-neighbour pair of the 50 k x 50 k search).
    The registration LANDS (final inliers, centimetres / 1e-3 rad from the truth) and every output equals the oracle's."""
    d = dense_scene
    fp = ql.default_frontend_params(voxel_size=0.001, use_tuple_test=tuple_test, seed=1)
    h = ql.Handle(0, **DENSE_LIMITS)
    try:
        g = h.register_pair(d["a"], d["b"], fp)
    finally:
        h.close()
    corr = qo16.match(d["a"], d["da"], d["b"], d["db"], True, bool(tuple_test), 0.95, 1)
    assert lo < corr.shape[0] < hi, corr.shape
    assert (g["n_src"], g["n_tgt"], g["L"]) == (50000, 50000, corr.shape[0])
    o = qo16.solve(d["a"][corr[:, 0]], d["b"][corr[:, 1]])
    _same(g, o)
    assert g["valid"] and g["clique"].size > 100 and g["final_inliers"].size > 50
    dy = _yaw(g["T"]) - _yaw(d["T"])
    assert abs(np.arctan2(np.sin(dy), np.cos(dy))) < 2e-3            # the scene's sampling noise, not the tolerance of _same
    assert np.linalg.norm(g["T"][:3, 3] - d["T"][:3, 3]) < 0.1
