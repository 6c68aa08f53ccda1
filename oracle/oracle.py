"""ctypes binding of oracle/libquatro_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinning status (see quatro_oracle.cpp header and DESIGN.md section 4): the reference ships no golden vectors and most of
it cannot be built here (PCL / FLANN / Eigen / PMC absent).  Two parts can, and are compiled from the reference's own
text where it lies (oracle/Makefile; outputs under oracle/_ref/):
  * teaser::Matcher — src/teaser_utils/feature_matcher.cc — libref_matcher.so, checked against match() (ref_match);
  * the back end of class Quatro — computeTIMs, solveForScale, solveForRotation[2D], solveForTranslation, estimate and
    computeTransformation itself, with teaser::Graph and teaser/utils.h's svdRot / svdRot2d — libref_solver.so (Eigen
    replaced by the subset in ref_shim_solver/: the reference's formulae and control flow are pinned, Eigen's summation
    orders and SVD are not; PMC's clique search answered by max_clique() below through a callback), checked against
    build_graph(), gnc_rotation2d(), cote_estimate*() and solve() (ref_* functions below).
tests/test_ref_cpu.py runs both; tests/golden/matcher_ref.npz and solver_ref.npz hold their outputs for the GPU box.
Voxel grid, normals / FPFH (PCL) and the clique search (PMC) remain unpinned: there the oracle defines the semantics.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libquatro_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "quatro_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "qtr_math.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_REF_PATH = os.path.join(_HERE, "_ref", "libref_matcher.so")
REFERENCE_ROOT = os.environ.get("QUATRO_REFERENCE", "/root/reference")
_ref = None


def ref_buildable() -> bool:
    return os.path.exists(os.path.join(REFERENCE_ROOT, "src", "teaser_utils", "feature_matcher.cc"))


def build_ref(force: bool = False):
    """Compiles the reference's teaser::Matcher where it lies (never copied).  Returns the .so path, or None when
    /root/reference is absent (GPU box: the prebuilt file travels with the snapshot)."""
    if ref_buildable():
        subprocess.check_call(["make", "-C", _HERE, "ref", "REF=" + REFERENCE_ROOT] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return _REF_PATH if os.path.exists(_REF_PATH) else None


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


_REF_SOLVER_PATH = os.path.join(_HERE, "_ref", "libref_solver.so")
_refs = None


def build_ref_solver(force: bool = False):
    """Compiles the Eigen-only member functions of the reference's class Quatro (see the module docstring)."""
    if os.path.exists(os.path.join(REFERENCE_ROOT, "include", "quatro.hpp")):
        subprocess.check_call(["make", "-C", _HERE, "ref_solver", "REF=" + REFERENCE_ROOT] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return _REF_SOLVER_PATH if os.path.exists(_REF_SOLVER_PATH) else None


def ref_solver_available() -> bool:
    return os.path.exists(_REF_SOLVER_PATH)


def _rs():
    global _refs
    if _refs is None:
        _refs = C.CDLL(_REF_SOLVER_PATH)
    return _refs


def ref_compute_tims(v3):
    """Quatro::computeTIMs (include/quatro.hpp:307-344): N x 3 points -> (K x 3 TIMs, K x 2 index map)."""
    v = np.ascontiguousarray(np.asarray(v3, dtype=np.float64).T)  # 3 x N row-major
    N = v.shape[1]
    K = N * (N - 1) // 2
    out = np.zeros((3, K))
    mp = np.zeros((2, K), dtype=np.int32)
    _rs().qref_compute_tims(_p(v, C.c_double), N, _p(out, C.c_double), _p(mp, C.c_int))
    return out.T.copy(), mp.T.copy()


def ref_scale_mask(src_tims, dst_tims, noise_bound=0.3, cbar2=1.0):
    """Quatro::solveForScale's inlier mask (:355-386) over K x 3 TIM arrays."""
    a = np.ascontiguousarray(np.asarray(src_tims, dtype=np.float64).T)
    b = np.ascontiguousarray(np.asarray(dst_tims, dtype=np.float64).T)
    K = a.shape[1]
    mask = np.zeros(K, dtype=np.uint8)
    _rs().qref_scale_mask(_p(a, C.c_double), _p(b, C.c_double), C.c_longlong(K), C.c_double(noise_bound), C.c_double(cbar2),
                          _p(mask, C.c_ubyte))
    return mask.astype(bool)


REF_GNC_NOISE_BOUND = 0.6  # solveForRotation2D freezes its bound at the process's first call (a function-local static)


def ref_gnc_rotation2d(src2, dst2, gnc_factor=1.4, max_iter=50, cost_thr=1.1e-4):
    """Quatro::solveForRotation2D (:430-572) on M x 2 TIMs with noise bound REF_GNC_NOISE_BOUND -> (R, cost, inliers)."""
    a = np.ascontiguousarray(np.asarray(src2, dtype=np.float64).T)
    b = np.ascontiguousarray(np.asarray(dst2, dtype=np.float64).T)
    M = a.shape[1]
    R = np.zeros(4)
    inl = np.zeros(M, dtype=np.uint8)
    cost = C.c_double()
    _rs().qref_gnc_rotation2d(_p(a, C.c_double), _p(b, C.c_double), M, C.c_double(REF_GNC_NOISE_BOUND), C.c_double(gnc_factor),
                              int(max_iter), C.c_double(cost_thr), _p(R, C.c_double), _p(inl, C.c_ubyte), C.byref(cost))
    return R.reshape(2, 2), cost.value, inl.astype(bool)


def ref_cote_estimate(X, ranges, median=True):
    """Quatro::estimate (:618-747): (estimate, inlier mask)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    R = np.ascontiguousarray(np.broadcast_to(np.asarray(ranges, dtype=np.float64), X.shape))
    inl = np.zeros(X.shape[0], dtype=np.uint8)
    est = C.c_double()
    _rs().qref_cote_estimate(_p(X, C.c_double), _p(R, C.c_double), X.shape[0], int(median), C.byref(est), _p(inl, C.c_ubyte))
    return est.value, inl.astype(bool)


def ref_translation(src3, dst3, cote_noise_bound=0.3, cbar2=1.0, median=True):
    """Quatro::solveForTranslation (:585-616) on N x 3 (rotated source, target) -> (t, inlier mask)."""
    a = np.ascontiguousarray(np.asarray(src3, dtype=np.float64).T)
    b = np.ascontiguousarray(np.asarray(dst3, dtype=np.float64).T)
    N = a.shape[1]
    t = np.zeros(3)
    inl = np.zeros(N, dtype=np.uint8)
    _rs().qref_translation(_p(a, C.c_double), _p(b, C.c_double), N, C.c_double(cote_noise_bound), C.c_double(cbar2),
                           int(median), _p(t, C.c_double), _p(inl, C.c_ubyte))
    return t, inl.astype(bool)


_clique_cb_keepalive = None


def ref_compute_transformation(src4, tgt4, noise_bound=0.3, cbar2=1.0, gnc_factor=1.4, max_iter=50, cost_thr=1.1e-4,
                               inlier_selection_mode=1, kcore_thr=0.5, cote_noise_bound=0.3, cote_median=True,
                               use_rot_inliers=False, clique_order=0):
    """Quatro::computeTransformation (include/quatro.hpp:769-936) of the REFERENCE on L matched keypoint pairs.  Everything
    is the reference's text except teaser::MaxCliqueSolver::findMaxClique (PMC is absent): that call is answered by THIS
    module's max_clique() on the graph the reference code built.  The noise bound of the rotation stage is the first
    call's in the process (a function-local static of the reference): keep noise_bound = REF_GNC_NOISE_BOUND / 2."""
    global _clique_cb_keepalive
    rs = _rs()
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_ulonglong), C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int))

    def cb(bm, L, W, mode, thr, out):
        bitmap = np.ctypeslib.as_array(bm, shape=(L, max(W, 1))).astype(np.uint64) if L > 0 else np.zeros((0, 1), np.uint64)
        cl = max_clique(bitmap, mode, thr, clique_order)
        for i, v in enumerate(cl):
            out[i] = int(v)
        return len(cl)
    _clique_cb_keepalive = CB(cb)
    rs.qref_set_clique_callback(_clique_cb_keepalive)
    a = np.ascontiguousarray(np.asarray(src4, dtype=np.float32)[:, :3])
    b = np.ascontiguousarray(np.asarray(tgt4, dtype=np.float32)[:, :3])
    L = a.shape[0]
    T = np.zeros(16)
    cl = np.zeros(max(L, 1), dtype=np.int32)
    rot = np.zeros(max(L, 1), dtype=np.int32)
    fin = np.zeros(max(L, 1), dtype=np.int32)
    ncl, nrot, nfin = C.c_int(), C.c_int(), C.c_int()
    valid = rs.qref_compute_transformation(_p(a, C.c_float), _p(b, C.c_float), L, C.c_double(noise_bound), C.c_double(cbar2),
                                           C.c_double(gnc_factor), int(max_iter), C.c_double(cost_thr),
                                           int(inlier_selection_mode), C.c_double(kcore_thr), C.c_double(cote_noise_bound),
                                           int(cote_median), int(use_rot_inliers), _p(T, C.c_double), _p(cl, C.c_int),
                                           C.byref(ncl), _p(rot, C.c_int), C.byref(nrot), _p(fin, C.c_int), C.byref(nfin))
    return {"valid": bool(valid), "T": T.reshape(4, 4), "clique": cl[:ncl.value].copy(), "rot_inliers": rot[:nrot.value].copy(),
            "final_inliers": fin[:nfin.value].copy()}


def ref_svd_rot(X, Y, W):
    """teaser::utils::svdRot2d / svdRot (include/teaser/utils.h:123-166) on N x 2 or N x 3 pairs."""
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64).T)
    Y = np.ascontiguousarray(np.asarray(Y, dtype=np.float64).T)
    W = np.ascontiguousarray(W, dtype=np.float64)
    d, N = X.shape
    R = np.zeros(d * d)
    (_rs().qref_svd_rot2d if d == 2 else _rs().qref_svd_rot3d)(_p(X, C.c_double), _p(Y, C.c_double), _p(W, C.c_double), N,
                                                               _p(R, C.c_double))
    return R.reshape(d, d)


def ref_match(xyz_s, desc_s, xyz_t, desc_t, crosscheck=True, tuple_test=True, tuple_scale=0.95, seed=0):
    """teaser::Matcher::calculateCorrespondences of the REFERENCE (compiled from /root/reference), with
    use_absolute_scale = true as FPFHManager calls it (include/fpfh_manager.hpp:126-127).  The only substitutions are the
    absent libraries (exact brute-force scan behind FLANN's interface) and the tuple test's random draws (counter RNG
    instead of srand(time) / rand()).  Prints the reference's own progress lines on stdout."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF_PATH)
        _ref.ref_calculate_correspondences.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                       C.c_int, C.c_int, C.c_int, C.c_float, C.c_ulonglong, C.c_void_p,
                                                       C.c_int]
    xs = np.ascontiguousarray(np.asarray(xyz_s)[:, :3], dtype=np.float32)
    xt = np.ascontiguousarray(np.asarray(xyz_t)[:, :3], dtype=np.float32)
    ds = np.ascontiguousarray(desc_s, dtype=np.float32)
    dt = np.ascontiguousarray(desc_t, dtype=np.float32)
    cap = 3 * 100 * (len(xs) + len(xt)) + 16
    corr = np.zeros((cap, 2), dtype=np.int32)
    n = _ref.ref_calculate_correspondences(xs.ctypes.data, len(xs), ds.ctypes.data, xt.ctypes.data, len(xt),
                                           dt.ctypes.data, 1, int(crosscheck), int(tuple_test), tuple_scale, int(seed),
                                           corr.ctypes.data, cap)
    assert n <= cap
    return corr[:n].copy()


class Params(C.Structure):
    _fields_ = [
        ("noise_bound", C.c_double), ("cbar2", C.c_double), ("rotation_gnc_factor", C.c_double),
        ("rotation_cost_threshold", C.c_double), ("kcore_heuristic_threshold", C.c_double),
        ("cote_noise_bound", C.c_double), ("ryrx", C.c_double * 9),
        ("rotation_max_iterations", C.c_int), ("inlier_selection_mode", C.c_int), ("cote_median", C.c_int),
        ("using_rot_inliers_when_estimating_cote", C.c_int), ("using_pre_estimated_ryrx", C.c_int),
        ("clique_order", C.c_int), ("reg_mode", C.c_int),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("valid", C.c_int), ("T", C.c_double * 16), ("cost", C.c_double),
        ("gnc_iters", C.c_int), ("n_clique", C.c_int), ("n_rot_inliers", C.c_int), ("n_final", C.c_int),
        ("max_core", C.c_int), ("n_edges", C.c_int), ("n_card", C.c_int * 3),
    ]


def default_params(**kw) -> Params:
    """Demo defaults actually run by the reference (config/params.yaml:22-44)."""
    p = Params()
    p.noise_bound = 0.3
    p.cbar2 = 1.0
    p.rotation_gnc_factor = 1.4
    p.rotation_cost_threshold = 1.1e-4
    p.kcore_heuristic_threshold = 0.5
    p.cote_noise_bound = 0.3
    for i, v in enumerate([1, 0, 0, 0, 1, 0, 0, 0, 1]):
        p.ryrx[i] = float(v)
    p.rotation_max_iterations = 50
    p.inlier_selection_mode = 1
    p.cote_median = 1
    p.using_rot_inliers_when_estimating_cote = 0
    p.using_pre_estimated_ryrx = 0
    p.clique_order = 0
    for k, v in kw.items():
        if k == "ryrx":
            for i, x in enumerate(np.asarray(v, dtype=np.float64).reshape(-1)):
                p.ryrx[i] = float(x)
        else:
            setattr(p, k, v)
    return p


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.qo_radius_neighbors.restype = C.c_longlong
        _lib.qo_cote_estimate.restype = C.c_double
    return _lib


def _f4(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4
    return a


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def set_threads(n: int) -> None:
    lib().qo_set_threads(int(n))


def max_threads() -> int:
    return int(lib().qo_get_max_threads())


def math_fn(fn: int, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else a, dtype=np.float32)
    out = np.zeros_like(a)
    lib().qo_math(int(fn), _p(a, C.c_float), _p(b, C.c_float), _p(out, C.c_float), a.size)
    return out


def voxelize(xyz4, leaf: float):
    xyz4 = _f4(xyz4)
    out = np.zeros_like(xyz4)
    n = lib().qo_voxelize(_p(xyz4, C.c_float), xyz4.shape[0], C.c_float(leaf), _p(out, C.c_float), xyz4.shape[0])
    if n < 0:
        return xyz4.copy()
    return out[:n].copy()


def radius_neighbors(xyz4, radius: float):
    xyz4 = _f4(xyz4)
    n = xyz4.shape[0]
    off = np.zeros(n + 1, dtype=np.int64)
    tot = lib().qo_radius_neighbors(_p(xyz4, C.c_float), n, C.c_double(radius), _p(off, C.c_longlong), None, None,
                                    C.c_longlong(0))
    idx = np.zeros(max(tot, 1), dtype=np.int32)
    d2 = np.zeros(max(tot, 1), dtype=np.float32)
    lib().qo_radius_neighbors(_p(xyz4, C.c_float), n, C.c_double(radius), _p(off, C.c_longlong), _p(idx, C.c_int),
                              _p(d2, C.c_float), C.c_longlong(tot))
    return off, idx[:tot], d2[:tot]


def fpfh(xyz4, r_normal: float, r_fpfh: float):
    """returns (normals4 [n,4] = nx,ny,nz,curvature ; spfh [n,33] ; fpfh [n,33])"""
    xyz4 = _f4(xyz4)
    n = xyz4.shape[0]
    nrm = np.zeros((n, 4), dtype=np.float32)
    sp = np.zeros((n, 33), dtype=np.float32)
    de = np.zeros((n, 33), dtype=np.float32)
    lib().qo_fpfh(_p(xyz4, C.c_float), n, C.c_double(r_normal), C.c_double(r_fpfh), _p(nrm, C.c_float),
                  _p(sp, C.c_float), _p(de, C.c_float))
    return nrm, sp, de


def eigen33(cov):
    """pcl::eigen33 restatement on n symmetric 3x3 float matrices -> (smallest eigenvalues [n], eigenvectors [n,3])."""
    cov = np.ascontiguousarray(cov, dtype=np.float32).reshape(-1, 9)
    n = cov.shape[0]
    ev = np.zeros(n, dtype=np.float32)
    vec = np.zeros((n, 3), dtype=np.float32)
    lib().qo_eigen33(_p(cov, C.c_float), n, _p(ev, C.c_float), _p(vec, C.c_float))
    return ev, vec


def nn33(query, data):
    query = np.ascontiguousarray(query, dtype=np.float32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    out = np.zeros(query.shape[0], dtype=np.int32)
    lib().qo_nn33(_p(query, C.c_float), query.shape[0], _p(data, C.c_float), data.shape[0], _p(out, C.c_int))
    return out


def match(xyz_s, desc_s, xyz_t, desc_t, crosscheck=True, tuple_test=True, tuple_scale=0.95, seed=0, debug=False):
    xyz_s, xyz_t = _f4(xyz_s), _f4(xyz_t)
    desc_s = np.ascontiguousarray(desc_s, dtype=np.float32)
    desc_t = np.ascontiguousarray(desc_t, dtype=np.float32)
    ns, nt = xyz_s.shape[0], xyz_t.shape[0]
    cap = 3 * 100 * (ns + nt) if not crosscheck else ns + nt
    corr = np.zeros((cap, 2), dtype=np.int32)
    nn_ij = np.zeros(max(min(ns, nt), 1), dtype=np.int32)
    nn_ji = np.zeros(max(max(ns, nt), 1), dtype=np.int32)
    L = lib().qo_match(_p(xyz_s, C.c_float), ns, _p(desc_s, C.c_float), _p(xyz_t, C.c_float), nt,
                       _p(desc_t, C.c_float), int(crosscheck), int(tuple_test), C.c_float(tuple_scale),
                       C.c_ulonglong(seed), _p(corr, C.c_int), cap, _p(nn_ij, C.c_int), _p(nn_ji, C.c_int))
    if debug:
        return corr[:L].copy(), nn_ij, nn_ji
    return corr[:L].copy()


def graph_words(L: int) -> int:
    return (L + 63) // 64


def build_graph(src4, tgt4, noise_bound=0.3, cbar2=1.0):
    src4, tgt4 = _f4(src4), _f4(tgt4)
    L = src4.shape[0]
    bm = np.zeros((L, graph_words(L)), dtype=np.uint64)
    lib().qo_build_graph(_p(src4, C.c_float), _p(tgt4, C.c_float), L, C.c_double(noise_bound), C.c_double(cbar2),
                         _p(bm, C.c_ulonglong))
    return bm


def kcore(bitmap):
    bitmap = np.ascontiguousarray(bitmap, dtype=np.uint64)
    L = bitmap.shape[0]
    core = np.zeros(L, dtype=np.int32)
    order = np.zeros(L, dtype=np.int32)
    mc = lib().qo_kcore(_p(bitmap, C.c_ulonglong), L, _p(core, C.c_int), _p(order, C.c_int))
    return core, order, mc


def max_clique(bitmap, mode=1, kcore_thr=0.5, order_mode=0):
    bitmap = np.ascontiguousarray(bitmap, dtype=np.uint64)
    L = bitmap.shape[0]
    cl = np.zeros(max(L, 1), dtype=np.int32)
    m = lib().qo_max_clique(_p(bitmap, C.c_ulonglong), L, mode, C.c_double(kcore_thr), order_mode, _p(cl, C.c_int))
    return cl[:m].copy()


def rot3_from_h(H):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    R = np.zeros(9)
    lib().qo_rot3_from_h(_p(H, C.c_double), _p(R, C.c_double))
    return R.reshape(3, 3)


def gnc_rotation3d(src3, dst3, noise_bound, gnc_factor=1.4, max_iter=50, cost_thr=1.1e-4):
    """M x 3 row-major TIMs -> (R 3x3, cost, iters, inlier mask)."""
    src3 = np.ascontiguousarray(src3, dtype=np.float64)
    dst3 = np.ascontiguousarray(dst3, dtype=np.float64)
    M = src3.shape[0]
    R = np.zeros(9)
    cost = C.c_double()
    iters = C.c_int()
    inl = np.zeros(max(M, 1), dtype=np.uint8)
    lib().qo_gnc_rotation3d(_p(src3, C.c_double), _p(dst3, C.c_double), M, C.c_double(noise_bound), C.c_double(gnc_factor),
                            max_iter, C.c_double(cost_thr), _p(R, C.c_double), C.byref(cost), C.byref(iters),
                            _p(inl, C.c_ubyte))
    return R.reshape(3, 3), cost.value, iters.value, inl[:M].astype(bool)


def gnc_rotation2d(src2, dst2, noise_bound, gnc_factor=1.4, max_iter=50, cost_thr=1.1e-4):
    src2 = np.ascontiguousarray(src2, dtype=np.float64)
    dst2 = np.ascontiguousarray(dst2, dtype=np.float64)
    M = src2.shape[0]
    R = np.zeros(4)
    cost = C.c_double()
    iters = C.c_int()
    inl = np.zeros(M, dtype=np.uint8)
    lib().qo_gnc_rotation2d(_p(src2, C.c_double), _p(dst2, C.c_double), M, C.c_double(noise_bound),
                            C.c_double(gnc_factor), max_iter, C.c_double(cost_thr), _p(R, C.c_double),
                            C.byref(cost), C.byref(iters), _p(inl, C.c_ubyte))
    return R.reshape(2, 2), cost.value, iters.value, inl.astype(bool)


def cote_estimate(X, rng, median=True):
    X = np.ascontiguousarray(X, dtype=np.float64)
    inl = np.zeros(X.shape[0], dtype=np.uint8)
    nc = C.c_int()
    e = lib().qo_cote_estimate(_p(X, C.c_double), X.shape[0], C.c_double(rng), int(median), _p(inl, C.c_ubyte),
                               C.byref(nc))
    return e, inl.astype(bool), nc.value


def cote_estimate_ranges(X, ranges, median=True):
    X = np.ascontiguousarray(X, dtype=np.float64)
    R = np.ascontiguousarray(ranges, dtype=np.float64)
    inl = np.zeros(X.shape[0], dtype=np.uint8)
    nc = C.c_int()
    lib().qo_cote_estimate_ranges.restype = C.c_double
    e = lib().qo_cote_estimate_ranges(_p(X, C.c_double), _p(R, C.c_double), X.shape[0], int(median), _p(inl, C.c_ubyte),
                                      C.byref(nc))
    return e, inl.astype(bool), nc.value


class PwParams(C.Structure):
    _fields_ = [("sensor_height", C.c_double), ("num_iter", C.c_int), ("num_lpr", C.c_int), ("num_min_pts", C.c_int),
                ("th_seeds", C.c_double), ("th_dist", C.c_double), ("max_range", C.c_double), ("min_range", C.c_double),
                ("uprightness_thr", C.c_double), ("adaptive_seed_selection_margin", C.c_double),
                ("using_global_thr", C.c_int), ("global_elevation_thr", C.c_double), ("num_zones", C.c_int),
                ("num_sectors_each_zone", C.c_int * 4), ("num_rings_each_zone", C.c_int * 4),
                ("min_ranges", C.c_double * 4), ("num_thr", C.c_int), ("elevation_thr", C.c_double * 8),
                ("flatness_thr", C.c_double * 8)]


def pw_params():
    """config/patchwork_params.yaml of the reference (the values its demo runs with)."""
    p = PwParams()
    p.sensor_height = 1.723
    p.num_iter, p.num_lpr, p.num_min_pts = 3, 20, 80
    p.th_seeds, p.th_dist, p.max_range, p.min_range = 0.25, 0.125, 80.0, 2.7
    p.uprightness_thr, p.adaptive_seed_selection_margin = 0.707, -1.1
    p.using_global_thr, p.global_elevation_thr = 0, -0.5
    p.num_zones = 4
    p.num_sectors_each_zone[:] = [16, 32, 54, 32]
    p.num_rings_each_zone[:] = [2, 4, 4, 4]
    p.min_ranges[:] = [2.7, 12.3625, 22.025, 41.35]
    p.num_thr = 4
    p.elevation_thr[:4] = [-1.2, -0.9984, -0.851, -0.605]
    p.flatness_thr[:4] = [0.0001, 0.000125, 0.000185, 0.000185]
    return p


def patchwork(xyz4, pp: PwParams | None = None):
    """PatchWork::estimate_ground -> ground, nonground (reference output order), patch id per input point."""
    xyz4 = _f4(xyz4)
    pp = pp or pw_params()
    P = xyz4.shape[0]
    g = np.zeros((max(P, 1), 4), dtype=np.float32)
    n = np.zeros((max(P, 1), 4), dtype=np.float32)
    pid = np.zeros(max(P, 1), dtype=np.int32)
    ng, nn = C.c_int(), C.c_int()
    lib().qo_patchwork(_p(xyz4, C.c_float), P, C.byref(pp), _p(g, C.c_float), C.byref(ng), _p(n, C.c_float),
                       C.byref(nn), _p(pid, C.c_int))
    return dict(ground=g[:ng.value].copy(), nonground=n[:nn.value].copy(), patch=pid[:P].copy())


class IpParams(C.Structure):
    _fields_ = [("n_scan", C.c_int), ("horizon_scan", C.c_int), ("ang_res_x", C.c_float), ("ang_res_y", C.c_float),
                ("ang_bottom", C.c_float), ("neighbor_mode", C.c_int), ("num_min_pts", C.c_int),
                ("segment_theta", C.c_float), ("valid_point_num", C.c_int), ("valid_line_num", C.c_int)]


def ip_params(lidar="Velodyne-64-HDE", neighbor_mode="4CrossNeighbor", num_min_pts=30):
    """The reference's ImageProjection constructor table (include/imageProjection.hpp:85-133)."""
    f32 = np.float32
    table = {  # n_scan, horizon_scan, ang_res_x, ang_res_y, ang_bottom
        "Velodyne-64-HDE": (64, 1800, f32(360.0) / f32(1800), f32(26.9) / f32(63), f32(25.0)),
        "VLP-16": (16, 1800, f32(0.2), f32(2.0), f32(15.0 + 0.1)),
        "HDL-32E": (32, 1800, f32(360.0) / f32(1800), f32(41.33) / f32(31), f32(30.67)),
        "Ouster-OS1-16": (16, 1024, f32(360.0) / f32(1024), f32(33.2) / f32(15), f32(16.6 + 0.1)),
        "Ouster-OS1-64": (64, 1024, f32(360.0) / f32(1024), f32(33.2) / f32(63), f32(16.6 + 0.1)),
    }
    ns, hs, rx, ry, ab = table[lidar]
    mode = {"4Neighbor": 0, "8Neighbor": 1, "4CrossNeighbor": 2}[neighbor_mode]
    return IpParams(ns, hs, float(rx), float(ry), float(ab), mode, num_min_pts,
                    float(f32(60.0 / 180.0 * np.pi)), 5, 3)


def segment_cloud(xyz4, ipp: IpParams | None = None):
    """ImageProjection::segmentCloud ("Patchwork" mode) -> valid segments (x,y,z,label), outliers, label / range images."""
    xyz4 = _f4(xyz4)
    ipp = ipp or ip_params()
    NP = ipp.n_scan * ipp.horizon_scan
    out = np.zeros((NP, 4), dtype=np.float32)
    outl = np.zeros((NP, 4), dtype=np.float32)
    lab = np.zeros(NP, dtype=np.int32)
    rng = np.zeros(NP, dtype=np.float32)
    nv, no = C.c_int(), C.c_int()
    lib().qo_segment_cloud(_p(xyz4, C.c_float), xyz4.shape[0], C.byref(ipp), _p(out, C.c_float), C.byref(nv),
                           _p(outl, C.c_float), C.byref(no), _p(lab, C.c_int), _p(rng, C.c_float))
    return dict(valid=out[:nv.value].copy(), outliers=outl[:no.value].copy(),
                labels=lab.reshape(ipp.n_scan, ipp.horizon_scan), ranges=rng.reshape(ipp.n_scan, ipp.horizon_scan))


def solve(src4, tgt4, params: Params | None = None):
    src4, tgt4 = _f4(src4), _f4(tgt4)
    L = src4.shape[0]
    prm = params or default_params()
    res = Result()
    cl = np.zeros(max(L, 1), dtype=np.int32)
    rot = np.zeros(max(L, 1), dtype=np.int32)
    fin = np.zeros(max(L, 1), dtype=np.int32)
    lib().qo_solve(_p(src4, C.c_float), _p(tgt4, C.c_float), L, C.byref(prm), C.byref(res), _p(cl, C.c_int),
                   _p(rot, C.c_int), _p(fin, C.c_int))
    return {
        "status": res.status, "valid": bool(res.valid), "T": np.array(res.T[:]).reshape(4, 4), "cost": res.cost,
        "gnc_iters": res.gnc_iters, "clique": cl[:res.n_clique].copy(), "rot_inliers": rot[:res.n_rot_inliers].copy(),
        "final_inliers": fin[:res.n_final].copy(), "max_core": res.max_core, "n_edges": res.n_edges,
        "n_card": list(res.n_card),
    }


def register_pair(src_raw4, tgt_raw4, leaf=0.3, r_normal=0.5, r_fpfh=0.75, tuple_scale=0.95, seed=0,
                  params: Params | None = None):
    src_raw4, tgt_raw4 = _f4(src_raw4), _f4(tgt_raw4)
    prm = params or default_params()
    res = Result()
    cap = max(src_raw4.shape[0], tgt_raw4.shape[0])
    counts = np.zeros(3, dtype=np.int32)
    cl = np.zeros(cap, dtype=np.int32)
    fin = np.zeros(cap, dtype=np.int32)
    lib().qo_register_pair(_p(src_raw4, C.c_float), src_raw4.shape[0], _p(tgt_raw4, C.c_float), tgt_raw4.shape[0],
                           C.c_float(leaf), C.c_double(r_normal), C.c_double(r_fpfh), C.c_float(tuple_scale),
                           C.c_ulonglong(seed), C.byref(prm), C.byref(res), _p(counts, C.c_int), _p(cl, C.c_int),
                           _p(fin, C.c_int), cap)
    return {
        "status": res.status, "valid": bool(res.valid), "T": np.array(res.T[:]).reshape(4, 4), "cost": res.cost,
        "gnc_iters": res.gnc_iters, "clique": cl[:res.n_clique].copy(), "final_inliers": fin[:res.n_final].copy(),
        "n_src": int(counts[0]), "n_tgt": int(counts[1]), "L": int(counts[2]), "max_core": res.max_core,
        "n_edges": res.n_edges,
    }
