"""Host-side mirror of the reference's interface for this path, in Python, above the C ABI.

Same names, argument meaning and error behaviour as the reference so that tests read like the
reference's demo (examples/run_global_registration.cpp:103-108, 206-221, 243-246):

    quatro = Quatro(); quatro.reset(params)
    src_feat, tgt_feat = voxelize(src, 0.3), voxelize(tgt, 0.3)
    fm = FPFHManager(normal_radius, fpfh_radius); fm.flushAllFeatures(); fm.setFeaturePair(src_feat, tgt_feat)
    quatro.setInputSource(fm.getSrcKps()); quatro.setInputTarget(fm.getTgtKps())
    T = quatro.computeTransformation()

Clouds are numpy [N,4] float32 (x,y,z,pad) — the layout of pcl::PointXYZ.  Everything numerical happens
in libquatro_hip.so; this module holds state and argument checks only.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field

import numpy as np

from . import lib as _ql


class INLIER_SELECTION_MODE(enum.IntEnum):  # reference include/quatro.hpp:184-189
    PMC_EXACT = 0
    PMC_HEU = 1
    KCORE_HEU = 2
    NONE = 3


class ROTATION_ESTIMATION_ALGORITHM(enum.IntEnum):  # :172-175
    GNC_TLS = 0
    FGR = 1


class INLIER_GRAPH_FORMULATION(enum.IntEnum):  # :197-200
    CHAIN = 0
    COMPLETE = 1


@dataclass
class Params:
    """Quatro::Params (reference include/quatro.hpp:202-268), same field names and defaults."""
    reg_name: str = "Quatro"
    cote_mode: str = "median"
    using_rot_inliers_when_estimating_cote: bool = False
    noise_bound: float = 0.3
    cbar2: float = 1.0
    estimate_scaling: bool = True            # accepted, ignored (reference :361 forces scale = 1)
    rotation_estimation_algorithm: ROTATION_ESTIMATION_ALGORITHM = ROTATION_ESTIMATION_ALGORITHM.GNC_TLS
    rotation_gnc_factor: float = 1.4
    rotation_max_iterations: int = 100
    rotation_cost_threshold: float = 1e-6
    rotation_tim_graph: INLIER_GRAPH_FORMULATION = INLIER_GRAPH_FORMULATION.CHAIN
    inlier_selection_mode: INLIER_SELECTION_MODE = INLIER_SELECTION_MODE.PMC_HEU
    kcore_heuristic_threshold: float = 0.5
    use_max_clique: bool = True              # deprecated in the reference, unused
    max_clique_exact_solution: bool = True   # deprecated in the reference, unused
    max_clique_time_limit: float = 3600.0


@dataclass
class RegistrationSolution:  # reference include/quatro.hpp:161-168
    valid: bool = True
    scale: float = 1.0
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3))
    rotation: np.ndarray = field(default_factory=lambda: np.eye(3))


_shared_handle = None


def _handle() -> "_ql.Handle":
    global _shared_handle
    if _shared_handle is None:
        _shared_handle = _ql.Handle(0)
    return _shared_handle


def _as_cloud(c) -> np.ndarray:
    a = np.asarray(c, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (3, 4):
        raise ValueError("cloud must be [N,3] or [N,4]")
    if a.shape[1] == 3:
        a = np.concatenate([a, np.zeros((a.shape[0], 1), dtype=np.float32)], axis=1)
    return np.ascontiguousarray(a)


def voxelize(src, voxelSize: float, handle=None) -> np.ndarray:
    """voxelize<T>() (reference include/quatro.hpp:49-68): pcl::VoxelGrid centroid down-sampling."""
    return (handle or _handle()).voxelize(_as_cloud(src), float(voxelSize))


class PatchWork:
    """Reference include/patchwork.hpp:36-233 (ground segmentation on the concentric zone model).  The constructor
    takes the "/patchwork/..." parameters as keyword arguments (names of config/patchwork_params.yaml, e.g.
    sensor_height=1.723, czm={"num_zones": 4, ...}) instead of a ros::NodeHandle; unspecified ones keep the yaml values."""

    _SCALARS = {"sensor_height": "sensor_height", "num_iter": "num_iter", "num_lpr": "num_lpr",
                "num_min_pts": "num_min_pts", "th_seeds": "th_seeds", "th_dist": "th_dist", "max_r": "max_range",
                "min_r": "min_range", "uprightness_thr": "uprightness_thr",
                "adaptive_seed_selection_margin": "adaptive_seed_selection_margin",
                "using_global_elevation": "using_global_thr", "global_elevation_threshold": "global_elevation_thr"}

    def __init__(self, handle=None, czm: dict | None = None, **kw):
        p = _ql.pw_params()
        for k, v in kw.items():
            if k not in self._SCALARS:
                raise TypeError(f"unknown Patchwork parameter {k!r}")
            setattr(p, self._SCALARS[k], type(getattr(p, self._SCALARS[k]))(v))
        if czm:
            nz = int(czm.get("num_zones", p.num_zones))
            for key, field in (("num_sectors_each_zone", "num_sectors_each_zone"),
                               ("num_rings_each_zone", "num_rings_each_zone"),
                               ("min_ranges_each_zone", "min_ranges")):
                if key in czm:
                    if len(czm[key]) != nz:  # patchwork.hpp:598-604
                        raise ValueError("Some parameters are wrong! the size of parameters should be same")
                    arr = getattr(p, field)
                    for i in range(4):
                        arr[i] = czm[key][i] if i < nz else 0
            p.num_zones = nz
            if "elevation_thresholds" in czm or "flatness_thresholds" in czm:
                e, f = czm.get("elevation_thresholds", []), czm.get("flatness_thresholds", [])
                if len(e) != len(f) or len(e) > 8:  # :610
                    raise ValueError("Some parameters are wrong! Check the elevation/flatness_thresholds")
                p.num_thr = len(e)
                for i in range(len(e)):
                    p.elevation_thr[i], p.flatness_thr[i] = e[i], f[i]
        if p.min_range != p.min_ranges[0]:  # :606
            raise ValueError("Setting min. ranges are weired! The first term should be eqaul to min_range_")
        self.params = p
        self._h = handle

    def estimate_ground(self, cloudIn):
        """-> (cloudOut (ground), cloudNonground, time_taken [s]); (n, 4) float32 records in the reference's order."""
        import time
        t0 = time.perf_counter()
        r = (self._h or _handle()).patchwork(_as_cloud(cloudIn), self.params)
        return r["ground"], r["nonground"], time.perf_counter() - t0


class ImageProjection:
    """Reference include/imageProjection.hpp:31-581 (range-image projection + sub-cluster rejection, "Patchwork"
    ground mode): segmentCloud then getValidSegments / getOutliers."""

    def __init__(self, lidarType: str = "Velodyne-64-HDE", neighborSelectionMode: str = "4CrossNeighbor",
                 groundSegmentationMode: str = "Patchwork", numSubclusteringCriteria: int = 30, handle=None):
        try:
            self.params = _ql.ip_params(lidarType, neighborSelectionMode, numSubclusteringCriteria)
        except ValueError:
            try:
                _ql.ip_params(lidarType, "4Neighbor")
            except ValueError:
                raise ValueError("[ImageProjection]:Check your paramter. Lidar Type is wrong!") from None
            raise ValueError("[ImageProjection]:Check your paramter. Neighbor selection mode is wrong!") from None
        if groundSegmentationMode not in ("LeGO-LOAM", "Patchwork"):
            raise ValueError("[ImageProjection]: Check your paramter. Ground Segmentation mode is wrong!")
        if groundSegmentationMode == "LeGO-LOAM":
            raise ValueError("[ImageProjection]: the LeGO-LOAM ground removal is not part of the device path")
        self._h = handle
        self._r = None

    def segmentCloud(self, cloud):
        self._r = (self._h or _handle()).segment_cloud(_as_cloud(cloud), self.params)

    def getValidSegments(self) -> np.ndarray:
        """(n, 4) float32: x, y, z, segment label (pcl::PointXYZI view; drop the last column for PointXYZ)."""
        return self._r["valid"]

    def getOutliers(self) -> np.ndarray:
        return self._r["outliers"]

    def getGround(self) -> np.ndarray:
        return np.zeros((0, 4), dtype=np.float32)

    def getLabelMat(self) -> np.ndarray:
        return self._r["labels"]


class FPFHManager:
    """Reference include/fpfh_manager.hpp:25-238 (front-end orchestrator)."""

    def __init__(self, normal_radius: float = 0.5, fpfh_radius: float = 0.6, interval: int = 1, handle=None,
                 seed: int = 0):
        self.normal_radius_ = float(normal_radius)
        self.fpfh_radius_ = float(fpfh_radius)
        self.interval_ = interval
        self.is_initial_ = True
        self.is_odometry_test_ = False
        self.corr = np.zeros((0, 2), dtype=np.int32)
        self.src_cloud = self.tgt_cloud = None
        self._obj = self._scene = None
        self._src_normals = self._tgt_normals = None
        self.src_matched = self.tgt_matched = None
        self.tgt_normals = None
        self._h = handle
        self.seed = seed  # tuple-test RNG seed (the reference seeds from the clock)

    def flushAllFeatures(self):
        self.is_initial_ = True

    # matched-pair PCD cache (reference :91-96, 179-232): "%06d_to_%06d.pcd", source half then target half
    def setLoadDir(self, loaddir):
        self.loaddir_ = str(loaddir)

    def setSaveDir(self, savedir):
        self.savedir_ = str(savedir)

    def saveFeaturePair(self, src_idx: int, tgt_idx: int, verbose: bool = False):
        if not getattr(self, "savedir_", ""):
            raise ValueError("Save dir. is not set")
        name = "%s/%06d_to_%06d.pcd" % (self.savedir_, src_idx, tgt_idx)
        merge = np.concatenate([self.getSrcKps(), self.getTgtKps()])
        if verbose:
            print(f"[SAVER]: {name}")
        _ql.write_pcd_xyz(name, merge)

    def loadFeaturePair(self, src_idx: int, tgt_idx: int, verbose: bool = False):
        if not getattr(self, "loaddir_", ""):
            raise ValueError("Load dir. is not set")
        name = "%s/%06d_to_%06d.pcd" % (self.loaddir_, src_idx, tgt_idx)
        try:
            merge = _ql.read_pcd_xyz(name)
        except OSError:
            raise ValueError("[FPFHManager]: Load feature set failed.") from None
        half = merge.shape[0] // 2
        self._src_kps, self._tgt_kps = merge[:half].copy(), merge[half:].copy()
        self.src_matched = self._src_kps[:, :3].astype(np.float64).T.copy()
        self.tgt_matched = self._tgt_kps[:, :3].astype(np.float64).T.copy()
        if verbose:
            print(f"[LOADER]: Loaded data from {name}...=>{half} {merge.shape[0] - half}")


    def setParams(self, normal_radius, fpfh_radius, interval):
        self.normal_radius_, self.fpfh_radius_, self.interval_ = float(normal_radius), float(fpfh_radius), interval

    def clearInputs(self):
        self.is_initial_ = True
        self.src_cloud = self.tgt_cloud = self._obj = self._scene = None

    def swapTgt2Src(self):
        self.src_cloud, self._obj, self._src_normals = self.tgt_cloud, self._scene, self._tgt_normals

    def setFeaturePair(self, src, target):
        if self.normal_radius_ > self.fpfh_radius_:  # reference :99-102
            raise ValueError("[FPFHManager]: Normal should be lower than fpfh_radius!!!!")
        h = self._h or _handle()
        if self.is_initial_ and not self.is_odometry_test_:
            self.src_cloud = _as_cloud(src)
            self._src_normals, self._obj = h.fpfh(self.src_cloud, self.normal_radius_, self.fpfh_radius_)
            self.is_initial_ = False
        else:
            self.swapTgt2Src()
        self.tgt_cloud = _as_cloud(target)
        self._tgt_normals, self._scene = h.fpfh(self.tgt_cloud, self.normal_radius_, self.fpfh_radius_)
        fp = _ql.default_frontend_params(normal_radius=self.normal_radius_, fpfh_radius=self.fpfh_radius_,
                                         tuple_scale=0.95, use_crosscheck=1, use_tuple_test=1, seed=self.seed)
        self.corr = h.match(self.src_cloud, self._obj, self.tgt_cloud, self._scene, fp)
        self._src_kps = self._tgt_kps = None
        self.src_matched = self.src_cloud[self.corr[:, 0], :3].astype(np.float64).T.copy()  # 3 x L, as Eigen
        self.tgt_matched = self.tgt_cloud[self.corr[:, 1], :3].astype(np.float64).T.copy()
        self.tgt_normals = self._tgt_normals[self.corr[:, 1], :3].astype(np.float64).T.copy()

    def getSrcMatched(self):
        return self.src_matched

    def getTgtMatched(self):
        return self.tgt_matched

    def getTgtNormals(self):
        return self.tgt_normals

    def getObjDescriptor(self):
        return self._obj

    def getSceneDescriptor(self):
        return self._scene

    def getSrcKps(self) -> np.ndarray:
        if getattr(self, "_src_kps", None) is not None:  # loaded from the pair cache
            return self._src_kps
        return _as_cloud(self.src_cloud[self.corr[:, 0], :3])

    def getTgtKps(self) -> np.ndarray:
        if getattr(self, "_tgt_kps", None) is not None:
            return self._tgt_kps
        return _as_cloud(self.tgt_cloud[self.corr[:, 1], :3])

    def getCorrespondences(self):
        return [(int(a), int(b)) for a, b in self.corr]


class Quatro:
    """Reference include/quatro.hpp:70-1061 — the PCL-Registration-derived back-end surface."""

    def __init__(self, handle=None):
        self.reg_name_ = "Quatro"
        self.noise_bound_ = 0.3                       # public member, used by COTE (reference :115, :601)
        self.cost_ = float("inf")
        self.using_pre_estimated_RyRx_ = False
        self.estimated_RyRx_ = np.eye(3)
        self.solution_ = RegistrationSolution()
        self.params_ = Params()
        self.input_ = None
        self.target_ = None
        self.max_iterations_ = 0
        self._h = handle
        self._clear()

    def _clear(self):
        self.max_clique_ = np.zeros(0, dtype=np.int32)
        self.rotation_inliers_ = np.zeros(0, dtype=np.int32)
        self.final_inliers_ = np.zeros(0, dtype=np.int32)
        self.num_rot_inliers_ = 0
        self.num_maxclique_ = 0

    def getParams(self) -> Params:
        return self.params_

    def setParams(self, params: Params):
        self.params_ = params

    def setPreEstaimatedRyRx(self, estimated_RyRx):  # sic (reference :276-279)
        self.estimated_RyRx_ = np.asarray(estimated_RyRx, dtype=np.float64)[:3, :3].copy()
        self.using_pre_estimated_RyRx_ = True

    def setInputSource(self, cloud):
        self.input_ = _as_cloud(cloud)

    def setInputTarget(self, cloud):
        c = _as_cloud(cloud)
        if c.shape[0] == 0:  # reference :298-302: PCL_ERROR + return
            print("[pcl::Quatro::setInputSource] Invalid or empty point cloud dataset given!")
            return
        self.target_ = c

    def reset(self, params: Params):
        self.reg_name_ = params.reg_name
        self.params_ = params
        self._clear()

    def setMaximumIterations(self, nr_iterations: int):
        self.max_iterations_ = nr_iterations

    def _c_params(self) -> "_ql.Params":
        p = self.params_
        if p.cote_mode not in ("median", "weighted_mean"):
            raise ValueError("[COTE]: Wrong parameter comes!")  # reference :911
        if self.reg_name_ not in ("Quatro", "TEASER"):
            raise ValueError("[solveForRotation] The param is wrong! It should be 'TEASER' or 'Quatro'")  # :410
        if self.reg_name_ == "TEASER" and self.using_pre_estimated_RyRx_:
            raise ValueError("Wrong reg type name is coming!")  # :424-426
        cp = _ql.default_params()
        cp.reg_mode = _ql.REG_TEASER if self.reg_name_ == "TEASER" else _ql.REG_QUATRO
        cp.noise_bound = p.noise_bound
        cp.cbar2 = p.cbar2
        cp.rotation_gnc_factor = p.rotation_gnc_factor
        cp.rotation_cost_threshold = p.rotation_cost_threshold
        cp.kcore_heuristic_threshold = p.kcore_heuristic_threshold
        cp.cote_noise_bound = self.noise_bound_
        for i, v in enumerate(np.asarray(self.estimated_RyRx_, dtype=np.float64).reshape(-1)):
            cp.ryrx[i] = float(v)
        cp.rotation_max_iterations = int(p.rotation_max_iterations)
        cp.inlier_selection_mode = int(p.inlier_selection_mode)
        cp.cote_median = 1 if p.cote_mode == "median" else 0
        cp.using_rot_inliers_when_estimating_cote = int(bool(p.using_rot_inliers_when_estimating_cote))
        cp.using_pre_estimated_ryrx = int(bool(self.using_pre_estimated_RyRx_))
        return cp

    def computeTransformation(self, output=None):
        """computeTransformation(Eigen::Matrix4d& output) (reference :769-936).  Returns the 4x4; when a
        numpy array is passed it is overwritten in place — and left untouched if the clique is too small,
        as in the reference (:809-813)."""
        if self.input_ is None or self.target_ is None:
            raise ValueError("input clouds not set")
        if self.input_.shape[0] != self.target_.shape[0]:
            raise ValueError("source and target keypoint clouds must have equal length")
        h = self._h or _handle()
        cp = self._c_params()
        cp.max_clique_time_limit = float(self.params_.max_clique_time_limit)  # :800 (PMC_EXACT only)
        r = h.solve(self.input_, self.target_, cp)
        # the reference persists params_.noise_bound *= 2/scale for later calls (:850-852)
        self.params_.noise_bound = self.params_.noise_bound * 2.0
        self.max_clique_ = r["clique"]
        self.num_maxclique_ = int(r["clique"].size)
        if not r["valid"]:
            self.solution_.valid = False
            return output
        self.rotation_inliers_ = r["rot_inliers"]
        self.num_rot_inliers_ = int(r["rot_inliers"].size)
        self.final_inliers_ = r["final_inliers"]
        self.cost_ = r["cost"]
        self.solution_ = RegistrationSolution(True, 1.0, r["T"][:3, 3].copy(), r["T"][:3, :3].copy())
        if output is None:
            return r["T"].copy()
        output[...] = r["T"]
        return output

    def getMaxCliques(self):
        return self.input_[self.max_clique_], self.target_[self.max_clique_]

    def getFinalInliers(self):
        return self.input_[self.final_inliers_], self.target_[self.final_inliers_]

    def getFinalInliersIndices(self):
        return [int(i) for i in self.final_inliers_]

    def getNumRotaionInliers(self) -> int:  # sic
        return self.num_rot_inliers_

    def getNumMaxCliqueInliers(self) -> int:
        return self.num_maxclique_
