"""ctypes binding of libquatro_hip.so (the C ABI in include/quatro_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises, loudly.
Build it with ``python -m quatro_amd.build`` (hipcc cross-compiles for gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# QTR_LIB: another build of the same library (the diagnostic / candidate builds of tests/probe); it has to exist like the default
LIB_PATH = os.environ.get("QTR_LIB") or os.path.join(_HERE, "libquatro_hip.so")

QTR_OK, QTR_ERR_BAD_ARG, QTR_ERR_CLIQUE_TOO_SMALL, QTR_ERR_CAPACITY, QTR_ERR_HIP, QTR_ERR_UNSUPPORTED = range(6)
MEM_HOST, MEM_DEVICE = 0, 1
INLIER_PMC_EXACT, INLIER_PMC_HEU, INLIER_KCORE_HEU, INLIER_NONE = range(4)
REG_QUATRO, REG_TEASER = 0, 1

DBG_GRAPH_BITMAP, DBG_CORE, DBG_PERM, DBG_NBR_OFFSETS, DBG_NBR_INDEX, DBG_NBR_DIST2, DBG_SPFH = 1, 2, 3, 4, 5, 6, 7
DBG_NN_LARGE_OF_SMALL, DBG_NN_SMALL_OF_LARGE, DBG_VOX_SRC, DBG_VOX_TGT, DBG_CORR, DBG_MATCH_STATS = 8, 9, 10, 11, 12, 13
DBG_SOLVER_STATE = 14


class Limits(C.Structure):
    _fields_ = [("max_points", C.c_int), ("max_voxels", C.c_int), ("max_corr", C.c_int), ("n_slots", C.c_int),
                ("max_long_neighbors", C.c_int)]


class Params(C.Structure):
    _fields_ = [
        ("noise_bound", C.c_double), ("cbar2", C.c_double), ("rotation_gnc_factor", C.c_double),
        ("rotation_cost_threshold", C.c_double), ("kcore_heuristic_threshold", C.c_double),
        ("cote_noise_bound", C.c_double), ("ryrx", C.c_double * 9),
        ("rotation_max_iterations", C.c_int), ("inlier_selection_mode", C.c_int), ("cote_median", C.c_int),
        ("using_rot_inliers_when_estimating_cote", C.c_int), ("using_pre_estimated_ryrx", C.c_int),
        ("reg_mode", C.c_int), ("max_clique_time_limit", C.c_double),
    ]


class FrontendParams(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("normal_radius", C.c_float), ("fpfh_radius", C.c_float),
                ("tuple_scale", C.c_float), ("use_crosscheck", C.c_int), ("use_tuple_test", C.c_int),
                ("seed", C.c_ulonglong)]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("valid", C.c_int), ("T", C.c_double * 16), ("cost", C.c_double),
        ("gnc_iters", C.c_int), ("n_clique", C.c_int), ("n_rot_inliers", C.c_int), ("n_final", C.c_int),
        ("max_core", C.c_int), ("n_edges", C.c_int), ("n_card", C.c_int * 3),
        ("n_src", C.c_int), ("n_tgt", C.c_int), ("n_corr", C.c_int),
    ]


class PairDesc(C.Structure):
    _fields_ = [("src_raw4", C.c_void_p), ("n_src", C.c_int), ("tgt_raw4", C.c_void_p), ("n_tgt", C.c_int),
                ("seed", C.c_ulonglong), ("clique", C.c_void_p), ("final_inliers", C.c_void_p), ("cap", C.c_int),
                ("src_corr4", C.c_void_p), ("tgt_corr4", C.c_void_p), ("n_corr", C.c_int)]


class StageTimes(C.Structure):
    _fields_ = ([(n, C.c_float) for n in ("voxelize", "fpfh", "match", "graph", "clique", "solve", "total",
                                          "nn_kernel")] + [("nn_launches", C.c_int), ("graph_kernel", C.c_float)])


class PwParams(C.Structure):
    _fields_ = [("sensor_height", C.c_double), ("num_iter", C.c_int), ("num_lpr", C.c_int), ("num_min_pts", C.c_int),
                ("th_seeds", C.c_double), ("th_dist", C.c_double), ("max_range", C.c_double), ("min_range", C.c_double),
                ("uprightness_thr", C.c_double), ("adaptive_seed_selection_margin", C.c_double),
                ("using_global_thr", C.c_int), ("global_elevation_thr", C.c_double), ("num_zones", C.c_int),
                ("num_sectors_each_zone", C.c_int * 4), ("num_rings_each_zone", C.c_int * 4),
                ("min_ranges", C.c_double * 4), ("num_thr", C.c_int), ("elevation_thr", C.c_double * 8),
                ("flatness_thr", C.c_double * 8)]


class IpParams(C.Structure):
    _fields_ = [("n_scan", C.c_int), ("horizon_scan", C.c_int), ("ang_res_x", C.c_float), ("ang_res_y", C.c_float),
                ("ang_bottom", C.c_float), ("neighbor_mode", C.c_int), ("num_min_pts", C.c_int),
                ("segment_theta", C.c_float), ("valid_point_num", C.c_int), ("valid_line_num", C.c_int)]


EXPORTS = [
    "qtr_create", "qtr_destroy", "qtr_last_error", "qtr_default_limits", "qtr_default_params", "qtr_demo_params",
    "qtr_default_frontend_params", "qtr_num_slots", "qtr_slot_stream", "qtr_voxelize", "qtr_fpfh", "qtr_match",
    "qtr_solve", "qtr_max_clique", "qtr_compute_tims", "qtr_scale_mask", "qtr_gnc_rotation2d",
    "qtr_cote_estimate", "qtr_cote_estimate_ranges", "qtr_ip_default_params", "qtr_segment_cloud", "qtr_pw_default_params", "qtr_patchwork", "qtr_gnc_rotation3d", "qtr_exact_stats", "qtr_read_kitti_bin", "qtr_write_pcd_xyz", "qtr_read_pcd_xyz", "qtr_register_pair", "qtr_register_pair_corr", "qtr_feature_pair", "qtr_get_stage_times", "qtr_get_nn_dir_times", "qtr_set_stage_events", "qtr_set_nn_event_stride", "qtr_get_nn_totals", "qtr_debug_fetch", "qtr_debug_math", "qtr_submit_batch", "qtr_wait", "qtr_set_batch_preprocess", "qtr_comm_unique_id", "qtr_comm_init", "qtr_gather_results", "qtr_gather_results_v", "qtr_comm_destroy",
]

_lib = None


QTR_ERR_IO = 6
QTR_ERR_NOT_RUN = 7


def read_kitti_bin(path: str, max_points: int = 250000) -> np.ndarray:
    """getCloud of the demo (examples/run_global_registration.cpp:377-402): (n, 4) float32 x, y, z, intensity."""
    out = np.zeros((max(max_points, 1), 4), dtype=np.float32)
    n = C.c_int()
    rc = load().qtr_read_kitti_bin(os.fsencode(path), out.ctypes.data, max_points, C.byref(n))
    if rc != QTR_OK:
        raise OSError(f"error: failed to load {path}")
    return out[:n.value].copy()


def write_pcd_xyz(path: str, xyz, binary: bool = False) -> None:
    a = np.asarray(xyz, dtype=np.float32).reshape(-1, np.asarray(xyz).shape[-1] if np.asarray(xyz).ndim == 2 else 3)
    if a.shape[1] == 3:
        a = np.concatenate([a, np.zeros((a.shape[0], 1), dtype=np.float32)], axis=1)
    a = _f4(a[:, :4])
    rc = load().qtr_write_pcd_xyz(os.fsencode(path), a.ctypes.data, a.shape[0], 1 if binary else 0)
    if rc != QTR_OK:
        raise OSError(f"failed to write {path}")


def read_pcd_xyz(path: str) -> np.ndarray:
    """(n, 4) float32 x, y, z, 0 from an ascii / binary / binary_compressed PCD holding fields x y z."""
    n = C.c_int()
    rc = load().qtr_read_pcd_xyz(os.fsencode(path), None, 0, C.byref(n))
    if rc not in (QTR_OK, QTR_ERR_CAPACITY):
        raise OSError(f"failed to read {path}")
    out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
    rc = load().qtr_read_pcd_xyz(os.fsencode(path), out.ctypes.data, n.value, C.byref(n))
    if rc != QTR_OK:
        raise OSError(f"failed to read {path}")
    return out[:n.value].copy()


def pw_params() -> PwParams:
    p = PwParams()
    load().qtr_pw_default_params(C.byref(p))
    return p


def ip_params(lidar: str = "Velodyne-64-HDE", neighbor_mode: str = "4CrossNeighbor", num_min_pts: int = 30) -> IpParams:
    p = IpParams()
    rc = load().qtr_ip_default_params(lidar.encode(), neighbor_mode.encode(), C.byref(p))
    if rc != QTR_OK:
        raise ValueError("[ImageProjection]:Check your paramter. Lidar Type / neighbor selection mode is wrong!")
    p.num_min_pts = num_min_pts
    return p


class QuatroHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libquatro_hip status {code}: {msg}")
        self.code = code


_libs = {}
TEST_ENGINES_LIB_PATH = os.path.join(_HERE, "libquatro_hip_testengines.so")  # the -DQTR_TEST_ENGINES build (tests only)


def load(path: str | None = None):
    """Loads libquatro_hip.so (or another build of it: the tests' comparison-engine build, a probe's candidate).  Raises
    if it has not been built — there is no software fallback."""
    global _lib
    LIB_PATH = path or globals()["LIB_PATH"]
    if LIB_PATH in _libs:
        return _libs[LIB_PATH]
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -m quatro_amd.build` "
            "(or __graft_entry__.build()). quatro_amd has no CPU fallback.")
    # torch ships its own libamdhip64.so.7; importing it first makes both share ONE HIP runtime.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.qtr_last_error.restype = C.c_char_p
    lib.qtr_slot_stream.restype = C.c_void_p
    lib.qtr_debug_fetch.restype = C.c_longlong
    lib.qtr_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.qtr_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.qtr_destroy.argtypes = [C.c_void_p]
    lib.qtr_last_error.argtypes = [C.c_void_p]
    lib.qtr_slot_stream.argtypes = [C.c_void_p, C.c_int]
    lib.qtr_num_slots.argtypes = [C.c_void_p]
    lib.qtr_voxelize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                 C.POINTER(C.c_int), C.c_int]
    lib.qtr_fpfh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                             C.c_int]
    lib.qtr_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                              C.POINTER(FrontendParams), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.qtr_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Params),
                              C.POINTER(Result), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.qtr_register_pair.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.POINTER(FrontendParams), C.POINTER(Params), C.POINTER(Result), C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int]
    lib.qtr_register_pair_corr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                           C.POINTER(FrontendParams), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Params),
                                           C.POINTER(Result), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.qtr_feature_pair.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.POINTER(FrontendParams), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.qtr_set_batch_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qtr_get_stage_times.argtypes = [C.c_void_p, C.c_int, C.POINTER(StageTimes)]
    if hasattr(lib, "qtr_get_nn_dir_times"):
        lib.qtr_get_nn_dir_times.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.qtr_set_stage_events.argtypes = [C.c_void_p, C.c_int]
    lib.qtr_set_nn_event_stride.argtypes = [C.c_void_p, C.c_int]
    lib.qtr_get_nn_totals.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]
    lib.qtr_max_clique.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.qtr_compute_tims.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.qtr_scale_mask.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_double, C.c_double,
                                   C.c_void_p]
    lib.qtr_gnc_rotation2d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                       C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                       C.c_void_p]
    lib.qtr_gnc_rotation3d.argtypes = lib.qtr_gnc_rotation2d.argtypes
    lib.qtr_cote_estimate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int,
                                      C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)]
    lib.qtr_cote_estimate_ranges.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)]
    lib.qtr_exact_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
    lib.qtr_read_kitti_bin.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.qtr_write_pcd_xyz.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    lib.qtr_read_pcd_xyz.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.qtr_pw_default_params.argtypes = [C.POINTER(PwParams)]
    lib.qtr_pw_default_params.restype = None
    lib.qtr_patchwork.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(PwParams), C.c_void_p, C.c_int,
                                  C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.qtr_ip_default_params.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(IpParams)]
    lib.qtr_segment_cloud.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(IpParams), C.c_void_p, C.c_int,
                                      C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.c_void_p, C.c_int]
    lib.qtr_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.qtr_submit_batch.argtypes = [C.c_void_p, C.POINTER(PairDesc), C.c_int, C.POINTER(FrontendParams),
                                     C.POINTER(Params), C.POINTER(Result), C.c_int]
    lib.qtr_wait.argtypes = [C.c_void_p]
    lib.qtr_comm_unique_id.argtypes = [C.c_char_p]
    lib.qtr_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    lib.qtr_gather_results.argtypes = [C.c_void_p, C.POINTER(Result), C.c_int, C.POINTER(Result)]
    lib.qtr_gather_results_v.argtypes = [C.c_void_p, C.POINTER(Result), C.c_int, C.POINTER(Result), C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.qtr_comm_destroy.argtypes = [C.c_void_p]
    lib.qtr_comm_destroy.restype = None
    _libs[LIB_PATH] = lib
    if path is None:
        _lib = lib
    return lib


def comm_unique_id() -> bytes:
    """The rendezvous id rank 0 creates for qtr_comm_init (128 bytes)."""
    buf = C.create_string_buffer(128)
    if load().qtr_comm_unique_id(buf) != QTR_OK:
        raise RuntimeError("qtr_comm_unique_id failed (librccl not available?)")
    return buf.raw


def default_params() -> Params:
    p = Params()
    load().qtr_default_params(C.byref(p))
    return p


def demo_params(**kw) -> Params:
    """config/params.yaml values (what the reference demo actually runs)."""
    p = Params()
    load().qtr_demo_params(C.byref(p))
    for k, v in kw.items():
        if k == "ryrx":
            for i, x in enumerate(np.asarray(v, dtype=np.float64).reshape(-1)):
                p.ryrx[i] = float(x)
        else:
            setattr(p, k, v)
    return p


def default_frontend_params(**kw) -> FrontendParams:
    p = FrontendParams()
    load().qtr_default_frontend_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _ptr(a):
    """numpy array -> (host pointer, MEM_HOST); torch CUDA tensor -> (device pointer, MEM_DEVICE)."""
    if a is None:
        return None, None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data, MEM_HOST
    # torch tensor
    assert a.is_contiguous()
    return a.data_ptr(), (MEM_DEVICE if a.is_cuda else MEM_HOST)


def _f4(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4, "points must be [N,4] float32 (x,y,z,pad)"
    return a


class Handle:
    """One device + n_slots stream slots (qtr_handle)."""

    def __init__(self, device: int = 0, max_points: int = 262144, max_voxels: int = 65536, max_corr: int = 24576,
                 n_slots: int = 1, max_long_neighbors: int = 0, lib_path: str | None = None):
        self._lib = load(lib_path)
        lim = Limits(max_points, max_voxels, max_corr, n_slots, max_long_neighbors)
        self._h = C.c_void_p()
        rc = self._lib.qtr_create(device, C.byref(lim), C.byref(self._h))
        if rc != QTR_OK:
            msg = self._lib.qtr_last_error(self._h).decode() if self._h else "qtr_create failed"
            if self._h:
                self._lib.qtr_destroy(self._h)
                self._h = C.c_void_p()
            raise QuatroHipError(rc, msg)
        self.limits = lim
        self._time_limit = 3600.0  # MaxCliqueSolver::Params::time_limit default (reference include/teaser/graph.h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.qtr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        return self._lib.qtr_last_error(self._h).decode()

    def _check(self, rc: int, ok=(QTR_OK,)):
        if rc not in ok:
            raise QuatroHipError(rc, self.last_error())
        return rc

    def stream_ptr(self, slot: int = 0) -> int:
        return int(self._lib.qtr_slot_stream(self._h, slot) or 0)

    # ---- host-array convenience wrappers (numpy in / numpy out) ---------------------------------
    def voxelize(self, xyz4, leaf: float, slot: int = 0):
        xyz4 = _f4(xyz4)
        out = np.zeros_like(xyz4)
        n = C.c_int()
        self._check(self._lib.qtr_voxelize(self._h, slot, xyz4.ctypes.data, xyz4.shape[0], leaf, out.ctypes.data,
                                           out.shape[0], C.byref(n), MEM_HOST))
        return out[: n.value].copy()

    def fpfh(self, xyz4, r_normal: float, r_fpfh: float, slot: int = 0):
        xyz4 = _f4(xyz4)
        n = xyz4.shape[0]
        nrm = np.zeros((n, 4), dtype=np.float32)
        desc = np.zeros((n, 33), dtype=np.float32)
        self._check(self._lib.qtr_fpfh(self._h, slot, xyz4.ctypes.data, n, r_normal, r_fpfh, nrm.ctypes.data,
                                       desc.ctypes.data, MEM_HOST))
        return nrm, desc

    def match(self, xyz_s, desc_s, xyz_t, desc_t, fp: FrontendParams | None = None, slot: int = 0):
        xyz_s, xyz_t = _f4(xyz_s), _f4(xyz_t)
        desc_s = np.ascontiguousarray(desc_s, dtype=np.float32)
        desc_t = np.ascontiguousarray(desc_t, dtype=np.float32)
        fp = fp or default_frontend_params()
        # cross-checked lists hold at most min(n_s, n_t) pairs; without the cross-check up to n_s + n_t
        cap = max(min(xyz_s.shape[0], xyz_t.shape[0]) if fp.use_crosscheck else xyz_s.shape[0] + xyz_t.shape[0], 1)
        corr = np.zeros((cap, 2), dtype=np.int32)
        L = C.c_int()
        self._check(self._lib.qtr_match(self._h, slot, xyz_s.ctypes.data, xyz_s.shape[0], desc_s.ctypes.data,
                                        xyz_t.ctypes.data, xyz_t.shape[0], desc_t.ctypes.data, C.byref(fp),
                                        corr.ctypes.data, cap, C.byref(L), MEM_HOST))
        return corr[: L.value].copy()

    def solve(self, src4, tgt4, params: Params | None = None, slot: int = 0):
        src4, tgt4 = _f4(src4), _f4(tgt4)
        L = src4.shape[0]
        assert tgt4.shape[0] == L
        prm = params or demo_params()
        res = Result()
        cap = max(L, 1)
        cl = np.zeros(cap, dtype=np.int32)
        rot = np.zeros(cap, dtype=np.int32)
        fin = np.zeros(cap, dtype=np.int32)
        rc = self._lib.qtr_solve(self._h, slot, src4.ctypes.data, tgt4.ctypes.data, L, C.byref(prm), C.byref(res),
                                 cl.ctypes.data, rot.ctypes.data, fin.ctypes.data, cap, MEM_HOST)
        self._check(rc, ok=(QTR_OK, QTR_ERR_CLIQUE_TOO_SMALL))
        return _result_dict(res, cl, rot, fin)

    def max_clique(self, bitmap, mode: int = 1, kcore_thr: float = 0.5, slot: int = 0, time_limit=None):
        """teaser::MaxCliqueSolver::findMaxClique on a bit-matrix graph [L][ceil(L/64)] uint64 -> (ids, max_core)."""
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint64)
        L = bitmap.shape[0]
        assert bitmap.ndim == 2 and (L == 0 or bitmap.shape[1] == (L + 63) // 64)
        cl = np.zeros(max(L, 1), dtype=np.int32)
        n, mcore = C.c_int(), C.c_int()
        tl = self._time_limit if time_limit is None else float(time_limit)
        self._check(self._lib.qtr_max_clique(self._h, slot, bitmap.ctypes.data, L, mode, kcore_thr, tl, cl.ctypes.data,
                                             cl.size, C.byref(n), C.byref(mcore), MEM_HOST))
        return cl[: n.value].copy(), mcore.value

    def set_clique_time_limit(self, seconds: float):
        """default MaxCliqueSolver::Params::time_limit of this wrapper's max_clique() calls (the C ABI takes the limit per
        call: qtr_max_clique's time_limit, qtr_params.max_clique_time_limit)"""
        self._time_limit = float(seconds)

    def exact_stats(self, slot: int = 0):
        n, a = C.c_ulonglong(), C.c_int()
        self._check(self._lib.qtr_exact_stats(self._h, slot, C.byref(n), C.byref(a)))
        return dict(nodes=int(n.value), aborted=bool(a.value))

    # ---- Patchwork ground segmentation (PatchWork::estimate_ground)
    def patchwork(self, xyz4, pp: "PwParams | None" = None, slot: int = 0):
        xyz4 = _f4(xyz4)
        pp = pp or pw_params()
        P = xyz4.shape[0]
        g = np.zeros((max(P, 1), 4), dtype=np.float32)
        n = np.zeros((max(P, 1), 4), dtype=np.float32)
        ng, nn = C.c_int(), C.c_int()
        self._check(self._lib.qtr_patchwork(self._h, slot, xyz4.ctypes.data, P, C.byref(pp), g.ctypes.data, max(P, 1),
                                            C.byref(ng), n.ctypes.data, max(P, 1), C.byref(nn), MEM_HOST))
        return dict(ground=g[:ng.value].copy(), nonground=n[:nn.value].copy())

    # ---- range-image projection + sub-cluster rejection (ImageProjection::segmentCloud, "Patchwork" mode)
    def segment_cloud(self, xyz4, ipp: "IpParams | None" = None, slot: int = 0, want_labels: bool = True):
        xyz4 = _f4(xyz4)
        ipp = ipp or ip_params()
        NP = ipp.n_scan * ipp.horizon_scan
        out = np.zeros((NP, 4), dtype=np.float32)
        outl = np.zeros((NP, 4), dtype=np.float32)
        lab = np.zeros(NP, dtype=np.int32) if want_labels else None
        nv, no, nseg = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.qtr_segment_cloud(self._h, slot, xyz4.ctypes.data, xyz4.shape[0], C.byref(ipp),
                                                out.ctypes.data, NP, C.byref(nv), outl.ctypes.data, NP, C.byref(no),
                                                C.byref(nseg), lab.ctypes.data if want_labels else None, MEM_HOST))
        return dict(valid=out[:nv.value].copy(), outliers=outl[:no.value].copy(), n_segments=nseg.value,
                    labels=lab.reshape(ipp.n_scan, ipp.horizon_scan) if want_labels else None)

    # ---- the reference class's individually callable stages (row-major matrices) -------------------
    def compute_tims(self, v3n, slot: int = 0):
        v = np.ascontiguousarray(v3n, dtype=np.float64)
        N = v.shape[1]
        K = N * (N - 1) // 2
        tims = np.zeros((3, K), dtype=np.float64)
        mp = np.zeros((2, K), dtype=np.int32)
        self._check(self._lib.qtr_compute_tims(self._h, slot, v.ctypes.data, N, tims.ctypes.data, mp.ctypes.data))
        return tims, mp

    def scale_mask(self, tims_src, tims_dst, noise_bound: float, cbar2: float = 1.0, slot: int = 0):
        a = np.ascontiguousarray(tims_src, dtype=np.float64)
        b = np.ascontiguousarray(tims_dst, dtype=np.float64)
        K = a.shape[1]
        mask = np.zeros(K, dtype=np.uint8)
        self._check(self._lib.qtr_scale_mask(self._h, slot, a.ctypes.data, b.ctypes.data, K, noise_bound, cbar2,
                                             mask.ctypes.data))
        return mask.astype(bool)

    def gnc_rotation2d(self, src2, dst2, noise_bound: float, gnc_factor: float = 1.4, max_iter: int = 50,
                       cost_thr: float = 1.1e-4, slot: int = 0):
        """src2/dst2: (M, 2) arrays as the oracle's gnc_rotation2d takes them."""
        s2 = np.ascontiguousarray(np.asarray(src2, dtype=np.float64).T)
        d2 = np.ascontiguousarray(np.asarray(dst2, dtype=np.float64).T)
        M = s2.shape[1]
        R = np.zeros(4)
        cost, iters = C.c_double(), C.c_int()
        inl = np.zeros(M, dtype=np.uint8)
        self._check(self._lib.qtr_gnc_rotation2d(self._h, slot, s2.ctypes.data, d2.ctypes.data, M, noise_bound,
                                                 gnc_factor, max_iter, cost_thr, R.ctypes.data, C.byref(cost),
                                                 C.byref(iters), inl.ctypes.data))
        return R.reshape(2, 2), cost.value, iters.value, inl.astype(bool)

    def gnc_rotation3d(self, src3, dst3, noise_bound: float, gnc_factor: float = 1.4, max_iter: int = 50,
                       cost_thr: float = 1.1e-4, slot: int = 0):
        """src3/dst3: (M, 3) TIMs -> (R 3x3, cost, iterations, inlier mask); reg_name "TEASER"."""
        s3 = np.ascontiguousarray(np.asarray(src3, dtype=np.float64).T)
        d3 = np.ascontiguousarray(np.asarray(dst3, dtype=np.float64).T)
        M = s3.shape[1]
        R = np.zeros(9)
        cost, iters = C.c_double(), C.c_int()
        inl = np.zeros(M, dtype=np.uint8)
        self._check(self._lib.qtr_gnc_rotation3d(self._h, slot, s3.ctypes.data, d3.ctypes.data, M, noise_bound,
                                                 gnc_factor, max_iter, cost_thr, R.ctypes.data, C.byref(cost),
                                                 C.byref(iters), inl.ctypes.data))
        return R.reshape(3, 3), cost.value, iters.value, inl.astype(bool)

    def cote_estimate(self, X, rng: float, median: bool = True, slot: int = 0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        inl = np.zeros(X.shape[0], dtype=np.uint8)
        est, nc = C.c_double(), C.c_int()
        self._check(self._lib.qtr_cote_estimate(self._h, slot, X.ctypes.data, X.shape[0], rng, 1 if median else 0,
                                                C.byref(est), inl.ctypes.data, C.byref(nc)))
        return est.value, inl.astype(bool), nc.value

    def cote_estimate_ranges(self, X, ranges, median: bool = True, slot: int = 0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        R = np.ascontiguousarray(ranges, dtype=np.float64)
        inl = np.zeros(X.shape[0], dtype=np.uint8)
        est, nc = C.c_double(), C.c_int()
        self._check(self._lib.qtr_cote_estimate_ranges(self._h, slot, X.ctypes.data, R.ctypes.data, X.shape[0],
                                                       1 if median else 0, C.byref(est), inl.ctypes.data, C.byref(nc)))
        return est.value, inl.astype(bool), nc.value

    def register_pair(self, src_raw4, tgt_raw4, fp: FrontendParams | None = None, params: Params | None = None,
                      slot: int = 0):
        src_raw4, tgt_raw4 = _f4(src_raw4), _f4(tgt_raw4)
        fp = fp or default_frontend_params()
        prm = params or demo_params()
        res = Result()
        cap = int(self.limits.max_corr)
        cl = np.zeros(cap, dtype=np.int32)
        fin = np.zeros(cap, dtype=np.int32)
        rc = self._lib.qtr_register_pair(self._h, slot, src_raw4.ctypes.data, src_raw4.shape[0], tgt_raw4.ctypes.data,
                                         tgt_raw4.shape[0], C.byref(fp), C.byref(prm), C.byref(res), cl.ctypes.data,
                                         fin.ctypes.data, cap, MEM_HOST)
        self._check(rc, ok=(QTR_OK, QTR_ERR_CLIQUE_TOO_SMALL))
        return _result_dict(res, cl, None, fin)

    def feature_pair(self, src_raw4, tgt_raw4, fp: FrontendParams | None = None, slot: int = 0):
        """voxelize x2 + FPFHManager::setFeaturePair (reference include/fpfh_manager.hpp:98-153) in one launch chain:
        raw scans -> {n_src, n_tgt, L, src_kps [L,4], tgt_kps [L,4], corr [L,2]}."""
        src_raw4, tgt_raw4 = _f4(src_raw4), _f4(tgt_raw4)
        fp = fp or default_frontend_params()
        cap = int(self.limits.max_corr)
        sk = np.zeros((cap, 4), dtype=np.float32)
        tk = np.zeros((cap, 4), dtype=np.float32)
        corr = np.zeros((cap, 2), dtype=np.int32)
        ns, nt, L = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.qtr_feature_pair(self._h, slot, src_raw4.ctypes.data, src_raw4.shape[0],
                                               tgt_raw4.ctypes.data, tgt_raw4.shape[0], C.byref(fp), C.byref(ns),
                                               C.byref(nt), C.byref(L), sk.ctypes.data, tk.ctypes.data,
                                               corr.ctypes.data, cap, MEM_HOST))
        return {"n_src": ns.value, "n_tgt": nt.value, "L": L.value, "src_kps": sk[:L.value].copy(),
                "tgt_kps": tk[:L.value].copy(), "corr": corr[:L.value].copy()}

    # ---- device-resident entry points (torch CUDA tensors or raw pointers) -----------------------
    def feature_pair_dev(self, src_ptr: int, Ps: int, tgt_ptr: int, Pt: int, fp: FrontendParams, slot: int = 0):
        """front end on device-resident scans; the matched clouds stay in the slot.  Returns (rc, n_src, n_tgt, L)."""
        ns, nt, L = C.c_int(), C.c_int(), C.c_int()
        rc = self._lib.qtr_feature_pair(self._h, slot, src_ptr, Ps, tgt_ptr, Pt, C.byref(fp), C.byref(ns), C.byref(nt),
                                        C.byref(L), None, None, None, 0, MEM_DEVICE)
        return rc, ns.value, nt.value, L.value

    def register_pair_dev(self, src_ptr: int, Ps: int, tgt_ptr: int, Pt: int, fp: FrontendParams, prm: Params,
                          res: Result, slot: int = 0) -> int:
        return self._lib.qtr_register_pair(self._h, slot, src_ptr, Ps, tgt_ptr, Pt, C.byref(fp), C.byref(prm),
                                           C.byref(res), None, None, 0, MEM_DEVICE)

    def register_pair_corr_dev(self, src_ptr: int, Ps: int, tgt_ptr: int, Pt: int, fp: FrontendParams, cs_ptr: int,
                               ct_ptr: int, n_corr: int, prm: Params, res: Result, slot: int = 0) -> int:
        """qtr_register_pair_corr on device-resident scans and correspondences: front end of the scans, back end on the given
        correspondences, one call"""
        return self._lib.qtr_register_pair_corr(self._h, slot, src_ptr, Ps, tgt_ptr, Pt, C.byref(fp), cs_ptr, ct_ptr, n_corr,
                                                C.byref(prm), C.byref(res), None, None, None, 0, MEM_DEVICE)

    def register_pair_corr(self, src_raw4, tgt_raw4, corr_src4, corr_tgt4, fp: FrontendParams | None = None,
                           params: Params | None = None, slot: int = 0):
        src_raw4, tgt_raw4, corr_src4, corr_tgt4 = _f4(src_raw4), _f4(tgt_raw4), _f4(corr_src4), _f4(corr_tgt4)
        fp = fp or default_frontend_params()
        prm = params or demo_params()
        res = Result()
        L = corr_src4.shape[0]
        cap = max(L, 1)
        cl = np.zeros(cap, dtype=np.int32)
        fin = np.zeros(cap, dtype=np.int32)
        nm = C.c_int()
        rc = self._lib.qtr_register_pair_corr(self._h, slot, src_raw4.ctypes.data, src_raw4.shape[0], tgt_raw4.ctypes.data,
                                              tgt_raw4.shape[0], C.byref(fp), corr_src4.ctypes.data, corr_tgt4.ctypes.data, L,
                                              C.byref(prm), C.byref(res), C.addressof(nm), cl.ctypes.data, fin.ctypes.data, cap,
                                              MEM_HOST)
        self._check(rc, ok=(QTR_OK, QTR_ERR_CLIQUE_TOO_SMALL))
        out = _result_dict(res, cl, None, fin)
        out["n_matched"] = nm.value
        return out

    def solve_dev(self, src_ptr: int, tgt_ptr: int, L: int, prm: Params, res: Result, slot: int = 0) -> int:
        return self._lib.qtr_solve(self._h, slot, src_ptr, tgt_ptr, L, C.byref(prm), C.byref(res), None, None, None,
                                   0, MEM_DEVICE)

    # ---- batched registration (qtr_submit_batch / qtr_wait): every slot of the handle is used
    def register_batch(self, pairs, fp: FrontendParams | None = None, params: Params | None = None, want_lists=True):
        """pairs: sequence of (src [n,4] float32, tgt [m,4] float32, seed) — or (src, tgt, seed, corr_src [L,4], corr_tgt
        [L,4]) for a pair that brings pre-matched correspondences (qtr_pair_desc.src_corr4 / tgt_corr4): src = tgt = None
        runs the back end alone on them, scans AND correspondences run the scans' front end and the back end on the given
        correspondences.  Returns one result dict per pair, in order — the same dicts register_pair returns."""
        fp = fp or default_frontend_params()
        prm = params or demo_params()
        B = len(pairs)
        descs = (PairDesc * max(B, 1))()
        results = (Result * max(B, 1))()
        keep = []
        cap = int(self.limits.max_corr)
        for i, item in enumerate(pairs):
            s_, t_, seed = item[0], item[1], item[2]
            cs_, ct_ = (item[3], item[4]) if len(item) > 3 else (None, None)
            s_, t_ = (None if s_ is None else _f4(s_)), (None if t_ is None else _f4(t_))
            cs_, ct_ = (None if cs_ is None else _f4(cs_)), (None if ct_ is None else _f4(ct_))
            n_c = 0 if cs_ is None else cs_.shape[0]
            if cs_ is not None and n_c == 0:  # (an empty array may have no address: "zero correspondences" needs pointers)
                cs_, ct_ = np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32)
            cl = np.zeros(cap if want_lists else 1, dtype=np.int32)
            fin = np.zeros(cap if want_lists else 1, dtype=np.int32)
            keep.append((s_, t_, cl, fin, cs_, ct_))
            descs[i] = PairDesc(None if s_ is None else s_.ctypes.data, 0 if s_ is None else s_.shape[0],
                                None if t_ is None else t_.ctypes.data, 0 if t_ is None else t_.shape[0], int(seed),
                                cl.ctypes.data if want_lists else None, fin.ctypes.data if want_lists else None, cap,
                                None if cs_ is None else cs_.ctypes.data, None if ct_ is None else ct_.ctypes.data, n_c)
        self._check(self._lib.qtr_submit_batch(self._h, descs, B, C.byref(fp), C.byref(prm), results, MEM_HOST))
        self._check(self._lib.qtr_wait(self._h))
        out = []
        for i in range(B):
            cl, fin = keep[i][2], keep[i][3]
            r = results[i]
            if want_lists and r.status in (QTR_OK, QTR_ERR_CLIQUE_TOO_SMALL):
                out.append(_result_dict(r, cl, None, fin))
            else:
                out.append({"status": r.status, "valid": bool(r.valid), "T": np.array(r.T[:]).reshape(4, 4),
                            "cost": r.cost, "n_src": r.n_src, "n_tgt": r.n_tgt, "L": r.n_corr,
                            "n_clique": r.n_clique, "n_final": r.n_final, "n_rot_inliers": r.n_rot_inliers})
        return out

    def set_batch_preprocess(self, pw: "PwParams | None" = None, ip: "IpParams | None" = None, on: bool = True):
        """Raw sweeps through register_batch: Patchwork ground removal + range-image segmentation in front of the voxel
        grid (qtr_set_batch_preprocess); on=False switches it off again."""
        if not on:
            self._check(self._lib.qtr_set_batch_preprocess(self._h, None, None))
            return
        self._pre = (pw or pw_params(), ip or ip_params())
        self._check(self._lib.qtr_set_batch_preprocess(self._h, C.byref(self._pre[0]), C.byref(self._pre[1])))

    def register_batch_dev(self, items, prm: Params, fp: FrontendParams | None = None, scans: bool = True,
                           corr: bool = False):
        """items: dicts with device tensors "src" / "tgt" and a FrontendParams "fp" (its seed is the pair's seed), and —
        with corr=True — the pre-matched correspondences "cs" / "ct" the back end runs on (scans=False: the back end
        alone).  Inputs stay in HBM; only the result records come back.  Returns a list of dicts (status, valid, T,
        sizes)."""
        fp = fp or default_frontend_params()
        B = len(items)
        descs = (PairDesc * max(B, 1))()
        results = (Result * max(B, 1))()
        for i, it in enumerate(items):
            descs[i] = PairDesc(it["src"].data_ptr() if scans else None, it["src"].shape[0] if scans else 0,
                                it["tgt"].data_ptr() if scans else None, it["tgt"].shape[0] if scans else 0,
                                int(it["fp"].seed), None, None, 0,
                                it["cs"].data_ptr() if corr else None, it["ct"].data_ptr() if corr else None,
                                it["cs"].shape[0] if corr else 0)
        self._check(self._lib.qtr_submit_batch(self._h, descs, B, C.byref(fp), C.byref(prm), results, MEM_DEVICE))
        self._check(self._lib.qtr_wait(self._h))
        return [{"status": r.status, "valid": bool(r.valid), "T": np.array(r.T[:]).reshape(4, 4), "cost": r.cost,
                 "n_src": r.n_src, "n_tgt": r.n_tgt, "L": r.n_corr, "n_clique": r.n_clique, "n_final": r.n_final,
                 "n_rot_inliers": r.n_rot_inliers, "gnc_iters": r.gnc_iters}
                for r in results[:B]]

    # ---- multi-GPU: RCCL all-gather of the result records through the C ABI (one handle = one rank)
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        self._check(self._lib.qtr_comm_init(self._h, unique_id, rank, world))

    def gather_results(self, results, world: int):
        """results: ctypes array (Result * n_local).  Returns a (Result * (world * n_local)) array, rank order."""
        n = len(results)
        out = (Result * max(world * n, 1))()
        self._check(self._lib.qtr_gather_results(self._h, results, n, out))
        return out

    def gather_results_v(self, results, n_local: int, world: int, cap_all: int):
        """Blocks of different lengths (qtr_gather_results_v).  results: ctypes array holding at least n_local records
        (None for n_local = 0).  Returns (array of the gathered records in rank order, per-rank counts)."""
        out = (Result * max(cap_all, 1))()
        counts = (C.c_int * max(world, 1))()
        n_all = C.c_int()
        self._check(self._lib.qtr_gather_results_v(self._h, results, n_local, out, cap_all, counts, C.byref(n_all)))
        return out, [int(c) for c in counts[:world]], n_all.value

    def set_stage_events(self, on: bool) -> None:
        self._lib.qtr_set_stage_events(self._h, 1 if on else 0)

    def set_nn_event_stride(self, every: int) -> None:
        """every n-th match of a slot carries the nearest-neighbour event pairs (1: all, 0: none)"""
        self._check(self._lib.qtr_set_nn_event_stride(self._h, int(every)))

    def nn_totals(self, slot: int = 0, reset: bool = False):
        """(summed milliseconds, launches) of the nearest-neighbour kernel since the last reset"""
        ms, n = C.c_double(0), C.c_longlong(0)
        self._check(self._lib.qtr_get_nn_totals(self._h, slot, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def stage_times(self, slot: int = 0) -> dict:
        t = StageTimes()
        self._lib.qtr_get_stage_times(self._h, slot, C.byref(t))
        out = {n: getattr(t, n) for n, _ in StageTimes._fields_}
        d1, d2 = C.c_float(), C.c_float()
        if hasattr(self._lib, "qtr_get_nn_dir_times"):  # (QTR_LIB may name an older build of the library: A/B runs)
            self._lib.qtr_get_nn_dir_times(self._h, slot, C.byref(d1), C.byref(d2))
        out["nn_dir1"], out["nn_dir2"] = d1.value, d2.value  # the two nearest-neighbour launches behind nn_kernel apart
        return out

    def debug_fetch(self, what: int, dtype, slot: int = 0) -> np.ndarray:
        nbytes = self._lib.qtr_debug_fetch(self._h, slot, what, None, 0)
        if nbytes < 0:
            raise QuatroHipError(-1, "qtr_debug_fetch failed")
        out = np.zeros(max(nbytes // np.dtype(dtype).itemsize, 0), dtype=dtype)
        if nbytes:
            self._lib.qtr_debug_fetch(self._h, slot, what, out.ctypes.data, nbytes)
        return out

    def debug_math(self, fn: int, a, b=None) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b if b is not None else a, dtype=np.float32)
        out = np.zeros_like(a)
        self._check(self._lib.qtr_debug_math(self._h, fn, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
        return out


def _result_dict(res: Result, cl, rot, fin) -> dict:
    return {
        "status": res.status, "valid": bool(res.valid), "T": np.array(res.T[:]).reshape(4, 4), "cost": res.cost,
        "gnc_iters": res.gnc_iters, "clique": cl[: res.n_clique].copy(),
        "rot_inliers": None if rot is None else rot[: res.n_rot_inliers].copy(),
        "final_inliers": fin[: res.n_final].copy(), "max_core": res.max_core, "n_edges": res.n_edges,
        "n_card": list(res.n_card), "n_src": res.n_src, "n_tgt": res.n_tgt, "L": res.n_corr,
        "n_rot_inliers": res.n_rot_inliers,
    }
