/* quatro_hip.h — C ABI of libquatro_hip.so, the MI355X (gfx950) back end behind the Quatro API.
 *
 * Plain C types only (no STL / Eigen / PCL / torch), never throws across the boundary: every call
 * returns an int status and the handle keeps a human-readable message (qtr_last_error).
 *
 * Each entry point replaces one call site of the reference (url-kaist/Quatro, cited as file:line):
 *   qtr_voxelize       <- voxelize<T>()                    include/quatro.hpp:49-68 (pcl::VoxelGrid)
 *   qtr_fpfh           <- teaser::FPFHEstimation::computeFPFHFeatures (4-arg)
 *                                                          src/teaser_utils/fpfh.cc:44-75
 *   qtr_match          <- teaser::Matcher::calculateCorrespondences
 *                                                          include/teaser_utils/feature_matcher.h:42-74,
 *                                                          src/teaser_utils/feature_matcher.cc:18-265
 *   qtr_solve          <- Quatro::computeTransformation(Eigen::Matrix4d&)
 *                                                          include/quatro.hpp:769-936
 *                         (computeTIMs :307, solveForScale :355, teaser::Graph + MaxCliqueSolver
 *                          include/teaser/graph.h:29-274 + src/graph.cc:12-104, solveForRotation2D :430,
 *                          solveForTranslation/estimate :585-747)
 *   qtr_max_clique     <- teaser::MaxCliqueSolver::findMaxClique(teaser::Graph)
 *                                                          include/teaser/graph.h:219-274, src/graph.cc:12-104
 *   qtr_compute_tims / qtr_scale_mask / qtr_gnc_rotation2d / qtr_cote_estimate
 *                      <- the public stage methods computeTIMs :307-344, solveForScale :355-386,
 *                         solveForRotation2D :430-572, estimate :618-747 of include/quatro.hpp
 *   qtr_patchwork      <- PatchWork::estimate_ground    include/patchwork.hpp:329-476
 *   qtr_segment_cloud  <- ImageProjection::segmentCloud ("Patchwork" mode) + getValidSegments / getOutliers
 *                                                          include/imageProjection.hpp:244-258,273-581
 *   qtr_submit_batch / qtr_wait <- the demo's loop over scan pairs (one Quatro object, reset() between
 *                         registrations)                   examples/run_global_registration.cpp:97-108
 *   qtr_register_pair  <- the demo's whole path        examples/run_global_registration.cpp:206-246
 *                         (voxelize x2, FPFHManager::setFeaturePair include/fpfh_manager.hpp:98-153,
 *                          setInputSource/setInputTarget/computeTransformation)
 *
 * Point layout everywhere: float32 x,y,z,pad — 16 bytes per point, i.e. pcl::PointXYZ and the KITTI
 * .bin record (x,y,z,intensity; reference examples/run_global_registration.cpp:377-402) can be passed
 * without repacking.  The 4th float is ignored on input and written as 0 on output.
 *
 * Memory: `mem` selects where caller buffers live — QTR_MEM_HOST (pageable/pinned host memory; the
 * library stages through its own device arenas) or QTR_MEM_DEVICE (HBM pointers valid on the handle's
 * device; nothing is staged, results are written to the device buffers and the small qtr_result
 * record to host).  The caller owns all buffers; the library owns only its arenas inside the handle.
 *
 * Threading: one handle = one device + n_slots independent stream slots.  Calls on different slots
 * may be issued from different host threads; calls on the same slot must be externally serialised.
 */
#ifndef QUATRO_HIP_H
#define QUATRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points below are its whole dynamic symbol table. */
#if defined(__GNUC__)
#define QTR_API __attribute__((visibility("default")))
#else
#define QTR_API
#endif

#define QTR_OK 0
#define QTR_ERR_BAD_ARG 1          /* std::invalid_argument in the reference */
#define QTR_ERR_CLIQUE_TOO_SMALL 2 /* reference: solution_.valid = false, output untouched (quatro.hpp:809-813) */
#define QTR_ERR_CAPACITY 3         /* a qtr_limits bound or a caller buffer capacity was exceeded */
#define QTR_ERR_HIP 4              /* HIP runtime error (message in qtr_last_error) */
#define QTR_ERR_UNSUPPORTED 5      /* mode accepted by the reference's API but not built here */
#define QTR_ERR_IO 6               /* file missing / unreadable / malformed (qtr_read_*, qtr_write_*) */
#define QTR_ERR_NOT_RUN 7          /* batched pair whose record was never produced: the job failed before it was started
                                      (qtr_submit_batch fills every record with this until the pair's result is final) */

#define QTR_MEM_HOST 0
#define QTR_MEM_DEVICE 1

/* INLIER_SELECTION_MODE, reference include/quatro.hpp:184-189 */
#define QTR_INLIER_PMC_EXACT 0
#define QTR_INLIER_PMC_HEU 1
#define QTR_INLIER_KCORE_HEU 2
#define QTR_INLIER_NONE 3

typedef struct qtr_handle qtr_handle;

typedef struct qtr_limits {
  int max_points; /* raw points per cloud            (default 262144; reference loader caps at 250000) */
  int max_voxels; /* down-sampled points per cloud   (default 65536; at most 2^20 - 32: qtr_create refuses more) */
  int max_corr;   /* correspondences into the solver (default 24576) */
  int n_slots;    /* independent stream slots        (default 1) */
  int max_long_neighbors; /* per cloud: total entries of the radius-neighbour lists that are LONGER than 256 entries
                             (dense / un-voxelised clouds; a list of up to 256 entries costs nothing here).
                             0 = default = 128 * max_voxels.  pcl's radius search has no cap
                             (reference src/teaser_utils/fpfh.cc:58-72): exceeding this returns QTR_ERR_CAPACITY */
} qtr_limits;

/* The fields of Quatro::Params that the path consumes (reference include/quatro.hpp:202-268), plus the
 * public member noise_bound_ (:269, used by COTE :600-601) and estimated_RyRx_ (:159). */
#define QTR_REG_QUATRO 0
#define QTR_REG_TEASER 1
typedef struct qtr_params {
  double noise_bound;               /* 0.3 */
  double cbar2;                     /* 1.0 */
  double rotation_gnc_factor;       /* 1.4 */
  double rotation_cost_threshold;   /* 1e-6 (Params default); demo yaml 1.1e-4 */
  double kcore_heuristic_threshold; /* 0.5 */
  double cote_noise_bound;          /* Quatro::noise_bound_ = 0.3 */
  double ryrx[9];                   /* row-major estimated_RyRx_, identity */
  int rotation_max_iterations;      /* 100 (Params default); demo yaml 50 */
  int inlier_selection_mode;        /* QTR_INLIER_PMC_HEU */
  int cote_median;                  /* 1 = cote_mode "median", 0 = "weighted_mean" */
  int using_rot_inliers_when_estimating_cote; /* 0 */
  int using_pre_estimated_ryrx;     /* 0 */
  int reg_mode;                     /* QTR_REG_QUATRO (Params::reg_name "Quatro", yaw) or QTR_REG_TEASER (3-DoF, row (f)4) */
  double max_clique_time_limit;     /* 3600 s (include/quatro.hpp:267,800).  PMC_EXACT only; <= 0 = none.  When the limit
                                       is hit the heuristic's clique is returned (the reference returns PMC's best so far)
                                       and qtr_exact_stats reports it.  Per call: two threads with different limits on
                                       different slots do not see each other's */
} qtr_params;

/* Front-end knobs of the demo (reference examples/run_global_registration.cpp:37-55, config/params.yaml:22-25)
 * and of FPFHManager::setFeaturePair's matcher call (include/fpfh_manager.hpp:126-127). */
typedef struct qtr_frontend_params {
  float voxel_size;      /* 0.3 */
  float normal_radius;   /* 0.5 */
  float fpfh_radius;     /* 0.75 */
  float tuple_scale;     /* 0.95 */
  int use_crosscheck;    /* 1 */
  int use_tuple_test;    /* 1 */
  unsigned long long seed; /* tuple-test RNG seed (the reference seeds with time(NULL)) */
} qtr_frontend_params;

typedef struct qtr_result {
  int status;        /* same value the call returned */
  int valid;         /* solution_.valid */
  double T[16];      /* row-major 4x4 [R t; 0 1] */
  double cost;       /* Quatro::cost_ */
  int gnc_iters;
  int n_clique;      /* getNumMaxCliqueInliers() */
  int n_rot_inliers; /* getNumRotaionInliers() */
  int n_final;       /* getFinalInliersIndices().size() */
  int max_core;
  int n_edges;       /* undirected edges of the consistency graph */
  int n_card[3];     /* COTE consensus-set cardinality per axis */
  int n_src, n_tgt;  /* voxelised cloud sizes (qtr_register_pair) */
  int n_corr;        /* correspondences handed to the solver */
} qtr_result;

/* Per-stage GPU time of the last call on a slot, milliseconds (hipEvent based). */
typedef struct qtr_stage_times {
  float voxelize, fpfh, match, graph, clique, solve, total;
  float nn_kernel;  /* sum of the nearest-neighbour kernel launches of the last match (events on the launch stream) */
  int nn_launches;
  float graph_kernel; /* k_graph_build alone */
} qtr_stage_times;

QTR_API int qtr_create(int device, const qtr_limits* limits /* NULL = defaults */, qtr_handle** out);
QTR_API void qtr_destroy(qtr_handle* h);
QTR_API const char* qtr_last_error(const qtr_handle* h);
QTR_API void qtr_default_limits(qtr_limits* l);
QTR_API void qtr_default_params(qtr_params* p);                   /* Quatro::Params defaults */
QTR_API void qtr_demo_params(qtr_params* p);                      /* config/params.yaml values */
QTR_API void qtr_default_frontend_params(qtr_frontend_params* p);
QTR_API int qtr_num_slots(const qtr_handle* h);
QTR_API void* qtr_slot_stream(qtr_handle* h, int slot); /* hipStream_t of a slot */

/* K1.  out_xyz4 capacity `cap` points; *n_out receives the voxel count (output order = ascending
 * linear voxel index, as PCL).  If the grid would overflow int32 PCL passes the input through; so
 * does this call (then *n_out == P). */
QTR_API int qtr_voxelize(qtr_handle* h, int slot, const float* xyz4, int P, float leaf, float* out_xyz4, int cap, int* n_out,
                 int mem);

/* K2-K4.  normals4 (nx,ny,nz,curvature; may be NULL) and desc33 (n x 33 floats) */
QTR_API int qtr_fpfh(qtr_handle* h, int slot, const float* xyz4, int n, float r_normal, float r_fpfh, float* normals4,
             float* desc33, int mem);

/* K5-K8.  corr2 = L x (src index, tgt index), sorted lexicographically, capacity `cap` pairs. */
QTR_API int qtr_match(qtr_handle* h, int slot, const float* xyz4_s, int n_s, const float* desc33_s, const float* xyz4_t,
              int n_t, const float* desc33_t, const qtr_frontend_params* fp, int* corr2, int cap, int* L_out, int mem);

/* K9-K16.  src4/tgt4: the two equal-length matched keypoint clouds (setInputSource / setInputTarget).
 * clique / rot_inliers / final_inliers: optional int buffers of capacity `cap` (counts in *res). */
QTR_API int qtr_solve(qtr_handle* h, int slot, const float* src4, const float* tgt4, int L, const qtr_params* prm,
              qtr_result* res, int* clique, int* rot_inliers, int* final_inliers, int cap, int mem);

/* K10-K12 alone: the clique search on a caller-supplied graph.  adj = symmetric bit matrix, L rows of
 * ceil(L/64) uint64 words (bit j of row i set <=> edge i-j; the diagonal and bits >= L are ignored).
 * mode: QTR_INLIER_PMC_EXACT, QTR_INLIER_PMC_HEU or QTR_INLIER_KCORE_HEU (teaser CLIQUE_SOLVER_MODE 0 / 1 / 2).
 * PMC_EXACT ("next" row (f)4, src/graph.cc:106-127): the heuristic's clique when it is maximum, otherwise the first
 * maximum clique in the canonical depth-first order; at most 32768 vertices.  clique receives
 * the member ids in ascending order (capacity cap); *n_out their count; *max_core_out (may be NULL) the
 * largest core number. */
QTR_API int qtr_max_clique(qtr_handle* h, int slot, const unsigned long long* adj, int L, int mode, double kcore_thr,
                   double time_limit /* PMC_EXACT: MaxCliqueSolver::Params::time_limit, seconds; <= 0 = none */,
                   int* clique, int cap, int* n_out, int* max_core_out, int mem);

/* The reference class keeps its stages individually callable (computeTIMs :307, solveForScale :355,
 * solveForRotation2D :430, estimate :618); these entry points serve them from the same device code the fused
 * path uses.  Host buffers only; matrices are ROW-major (3 x N means three rows of N doubles).
 * Capacity: every call stages through the slot's solver arena — K (TIM columns) and M / N are limited to what
 * max_corr provides (3 K doubles <= 36 * max_corr doubles per operand); QTR_ERR_CAPACITY otherwise. */
QTR_API int qtr_compute_tims(qtr_handle* h, int slot, const double* v3n, int N, double* tims3k /* 3 x N(N-1)/2 */,
                     int* map2k /* 2 x N(N-1)/2: (i, j) of every column */);
QTR_API int qtr_scale_mask(qtr_handle* h, int slot, const double* tims_src3k, const double* tims_dst3k, long long K,
                   double noise_bound, double cbar2, unsigned char* mask /* K */);
QTR_API int qtr_gnc_rotation2d(qtr_handle* h, int slot, const double* src2m, const double* dst2m, int M, double noise_bound,
                       double gnc_factor, int max_iterations, double cost_threshold, double* R4 /* row-major 2x2 */,
                       double* cost, int* iterations, unsigned char* inliers /* M, weight >= 0.4 */);
/* "Next" row (f)4, second half: the 3-DoF rotation of reg_name "TEASER" (solveForRotation throws for it in the
 * reference, include/quatro.hpp:409-411; teaser::utils::svdRot, include/teaser/utils.h:123-149, is what it would
 * call): TEASER++'s GNC-TLS loop over 3-D TIMs.  Also reachable through qtr_solve with reg_mode = QTR_REG_TEASER. */
QTR_API int qtr_gnc_rotation3d(qtr_handle* h, int slot, const double* src3m, const double* dst3m, int M, double noise_bound,
                       double gnc_factor, int max_iterations, double cost_threshold, double* R9 /* row-major 3x3 */,
                       double* cost, int* iterations, unsigned char* inliers /* M, weight >= 0.4 */);
QTR_API int qtr_cote_estimate(qtr_handle* h, int slot, const double* X, int N, double range /* uniform */, int median_selection,
                      double* estimate, unsigned char* inliers /* N */, int* n_card);
/* the same with one range per element (estimate() takes a RowVectorXd of ranges, include/quatro.hpp:618-630) */
QTR_API int qtr_cote_estimate_ranges(qtr_handle* h, int slot, const double* X, const double* ranges, int N, int median_selection,
                             double* estimate, unsigned char* inliers /* N */, int* n_card);

/* PMC_EXACT only: search-tree nodes of the slot's last exact search and whether its time limit
 * (qtr_params.max_clique_time_limit / qtr_max_clique's time_limit) was hit. */
QTR_API int qtr_exact_stats(qtr_handle* h, int slot, unsigned long long* nodes, int* aborted);

/* "Next" row (f)3: on-disk formats either side of the path (host code, no GPU work, no handle).
 *   qtr_read_kitti_bin <- getCloud, examples/run_global_registration.cpp:377-402: float32 x,y,z,intensity records,
 *                         at most max_points of them (the demo reads 1 000 000 floats = 250 000 points).
 *   qtr_write_pcd_xyz / qtr_read_pcd_xyz <- the matched-pair cache of FPFHManager::saveFeaturePair / loadFeaturePair
 *                         (include/fpfh_manager.hpp:179-232): PCD v0.7, fields x y z.  binary = 0 writes what
 *                         pcl::io::savePCDFile writes by default (DATA ascii, 8 significant digits); the reader takes
 *                         ascii, binary and binary_compressed files and picks x, y, z by field name.
 * Points are 16-byte x,y,z,w records like everywhere else in this ABI (w: intensity for .bin, 0 for PCD).
 * qtr_read_pcd_xyz always reports the file's point count; QTR_ERR_CAPACITY when cap is smaller. */
QTR_API int qtr_read_kitti_bin(const char* path, float* xyzi, int max_points, int* n_points);
QTR_API int qtr_write_pcd_xyz(const char* path, const float* xyz4, int n, int binary);
QTR_API int qtr_read_pcd_xyz(const char* path, float* xyz4, int cap, int* n_points);

/* "Next" row (f)2: Patchwork ground segmentation, the first stage of the reference demo on raw scans
 * (PatchWork::estimate_ground, include/patchwork.hpp:329-476; parameters config/patchwork_params.yaml).
 * ground_xyzw / nonground_xyzw: the input records (16 bytes each, 4th float preserved) in the reference's output
 * order (zone, ring, sector; ascending height inside a patch); capacities in points (P always suffices). */
typedef struct qtr_pw_params {
  double sensor_height;                     /* 1.723 */
  int num_iter, num_lpr, num_min_pts;       /* 3, 20, 80 */
  double th_seeds, th_dist, max_range, min_range, uprightness_thr, adaptive_seed_selection_margin;
  int using_global_thr;
  double global_elevation_thr;
  int num_zones;                            /* <= 4 */
  int num_sectors_each_zone[4], num_rings_each_zone[4];
  double min_ranges[4];
  int num_thr;                              /* size of the two threshold vectors (<= 8) */
  double elevation_thr[8], flatness_thr[8];
} qtr_pw_params;
QTR_API void qtr_pw_default_params(qtr_pw_params* p); /* config/patchwork_params.yaml */
QTR_API int qtr_patchwork(qtr_handle* h, int slot, const float* xyz4, int P, const qtr_pw_params* pw, float* ground_xyzw,
                  int cap_ground, int* n_ground, float* nonground_xyzw, int cap_nonground, int* n_nonground, int mem);

/* "Next" row (f)1: range-image projection + sub-cluster rejection, the stage before voxelisation in the reference
 * demo (ImageProjection::segmentCloud in "Patchwork" mode + getValidSegments / getOutliers,
 * include/imageProjection.hpp:244-258,273-294).  Input: the non-ground points of one scan in sensor order.
 * valid_xyzl: x,y,z,label of every pixel of a valid segment, row-major over the range image; outl_xyzi: the
 * rejected sub-clusters (x, y, z, row + col/10000 as in the reference :345).  Capacities in points
 * (n_scan * horizon_scan always suffices).  labelmat (optional, host only): n_scan x horizon_scan int32,
 * -1 no return / 999999 rejected / label >= 1. */
typedef struct qtr_ip_params {
  int n_scan, horizon_scan;            /* 64, 1800 for "Velodyne-64-HDE" */
  float ang_res_x, ang_res_y, ang_bottom;
  int neighbor_mode;                   /* 0 "4Neighbor", 1 "8Neighbor", 2 "4CrossNeighbor" */
  int num_min_pts;                     /* numMinPtsForSubclustering, 30 */
  float segment_theta;                 /* 60 deg in rad */
  int valid_point_num, valid_line_num; /* 5, 3 */
} qtr_ip_params;
/* lidar_type: "Velodyne-64-HDE", "VLP-16", "HDL-32E", "Ouster-OS1-16", "Ouster-OS1-64"; neighbor_mode: as above.
 * Returns QTR_ERR_BAD_ARG for names the reference's constructor rejects (:131, :140). */
QTR_API int qtr_ip_default_params(const char* lidar_type, const char* neighbor_mode, qtr_ip_params* p);
QTR_API int qtr_segment_cloud(qtr_handle* h, int slot, const float* xyz4, int P, const qtr_ip_params* ip, float* valid_xyzl,
                      int cap_valid, int* n_valid, float* outl_xyzi, int cap_outl, int* n_outl, int* n_segments,
                      int* labelmat, int mem);

/* Whole path on one slot: raw scans -> transform. */
QTR_API int qtr_register_pair(qtr_handle* h, int slot, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                      const qtr_frontend_params* fp, const qtr_params* prm, qtr_result* res, int* clique,
                      int* final_inliers, int cap, int mem);

/* The whole path of one pair whose back end runs on correspondences the CALLER brings (a cache of matched keypoints:
 * FPFHManager::loadFeaturePair, include/fpfh_manager.hpp:211-232; another matcher) while the scans still go through the
 * front end — the single-pair form of a qtr_pair_desc with both scans and src_corr4 / tgt_corr4 set (see
 * qtr_submit_batch), and the unit of work BASELINE's metric is quoted on: a KITTI-64 pair's voxel grid + FPFH + matching
 * AND a ~5 k-correspondence computeTransformation, as ONE call (the back end is enqueued as soon as the matcher's counters
 * arrive; it starts after the front end like in qtr_register_pair — nothing overlaps that a registration would
 * serialise).  res->n_src / n_tgt report the voxel counts, res->n_corr = n_corr; n_matched (optional) receives the
 * matcher's own correspondence count.  corr_*4: n_corr 16-byte records each, same `mem` as the scans. */
QTR_API int qtr_register_pair_corr(qtr_handle* h, int slot, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                                   const qtr_frontend_params* fp, const float* corr_src4, const float* corr_tgt4, int n_corr,
                                   const qtr_params* prm, qtr_result* res, int* n_matched, int* clique, int* final_inliers,
                                   int cap, int mem);

/* Front end of one pair on one slot: raw scans -> matched keypoint clouds.  What the reference's caller does between
 * loading two scans and handing the keypoints to Quatro (examples/run_global_registration.cpp:206-221): `voxelize` x2
 * (include/quatro.hpp:49-68), FPFHManager::setFeaturePair (include/fpfh_manager.hpp:98-153: FPFH x2 + reciprocal
 * matching with cross check and tuple test), getSrcKps / getTgtKps / getCorrespondences (:172-177,234).  The same
 * launch chain as qtr_register_pair up to the solver.  n_src / n_tgt (optional) receive the voxelised cloud sizes, *L
 * the number of correspondences; src_kps4 / tgt_kps4 (optional, capacity `cap` 16-byte records) the matched keypoints in
 * correspondence order, corr2 (optional, cap x 2 ints) the (source, target) voxel indices.  mem = QTR_MEM_HOST: outputs
 * complete on return; QTR_MEM_DEVICE: outputs are written on the slot's stream (qtr_slot_stream).  The matched clouds
 * also stay in the slot, where a following qtr_solve on device pointers can be issued without a copy. */
QTR_API int qtr_feature_pair(qtr_handle* h, int slot, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                     const qtr_frontend_params* fp, int* n_src, int* n_tgt, int* L, float* src_kps4, float* tgt_kps4,
                     int* corr2, int cap, int mem);

/* Batched registration (BASELINE configs[2] / [3]; the reference's usage is one Quatro object reused over many pairs,
 * examples/run_global_registration.cpp:97-108).  B independent pairs go through the SAME kernels as qtr_register_pair,
 * a group of pairs per launch (blockIdx.z = pair): the handle's stream slots are split into two lanes of
 * n_slots / 2 pairs each, a lane runs voxelise -> FPFH + matching -> solver as three launch chains with ONE host
 * read-back of the device-side sizes per chain and group (not per pair), and the two lanes alternate so that one's
 * kernels cover the other's read-back.  Results are bit-identical to B sequential qtr_register_pair calls.
 *   qtr_submit_batch  validates, records the job and enqueues the first chains; returns without waiting.
 *   qtr_wait          drives the job to completion (call it from the same thread).  results[i] receives pair i's
 *                     record (its own status: QTR_OK, QTR_ERR_CLIQUE_TOO_SMALL, QTR_ERR_CAPACITY ...); the return
 *                     value is QTR_OK unless the job itself failed (HIP error, bad argument) — then every pair that
 *                     had finished keeps its record, pairs that were in flight read QTR_ERR_HIP, pairs never started
 *                     QTR_ERR_NOT_RUN, and the handle is drained and usable.
 * pairs / results must stay valid until qtr_wait returns; one job at a time per handle; every slot of the handle is
 * used (do not run slot calls concurrently).  mem as elsewhere (raw scans and the optional index lists). */
typedef struct qtr_pair_desc {
  const float* src_raw4; /* raw source scan, 16-byte x,y,z,* records (NULL together with tgt_raw4: no front end, see below) */
  int n_src;
  const float* tgt_raw4;
  int n_tgt;
  unsigned long long seed; /* tuple-test RNG seed of this pair (qtr_frontend_params.seed is ignored) */
  int* clique;             /* optional: getMaxCliques indices, capacity `cap` ints (NULL: not wanted) */
  int* final_inliers;      /* optional: getFinalInliersIndices */
  int cap;
  /* Pre-matched correspondences (optional; all three zero: the matcher's own output feeds the back end).  The
   * reference's loop hands Quatro whatever matched keypoint clouds its caller has — setInputSource / setInputTarget /
   * computeTransformation, examples/run_global_registration.cpp:243-246, include/quatro.hpp:769 — so a pair may bring
   * them along: src_corr4[i] <-> tgt_corr4[i], n_corr 16-byte records each, same `mem` as the scans.
   *   scans NULL, correspondences given : the back end alone (qtr_solve's work, batched); n_src / n_tgt ignored
   *   scans AND correspondences given   : the front end runs on the scans (result.n_src / n_tgt report its voxel
   *                                       counts) and the back end runs on the GIVEN correspondences instead of the
   *                                       matcher's — the unit of work BASELINE's metric is quoted on (a KITTI-64
   *                                       pair's front end + a ~5 k-correspondence back end) for callers whose
   *                                       correspondences come from elsewhere (a cache: FPFHManager::loadFeaturePair,
   *                                       include/fpfh_manager.hpp:211-232)
   * result.n_corr reports the correspondences the back end ran on.  n_corr = 0 with both pointers set is the
   * reference's "clique too small" outcome; n_corr > max_corr is QTR_ERR_CAPACITY in the pair's record. */
  const float* src_corr4;
  const float* tgt_corr4;
  int n_corr;
} qtr_pair_desc;
QTR_API int qtr_submit_batch(qtr_handle* h, const qtr_pair_desc* pairs, int B, const qtr_frontend_params* fp,
                     const qtr_params* prm, qtr_result* results, int mem);
QTR_API int qtr_wait(qtr_handle* h);
/* Raw sweeps through the batched entry: with parameters set here qtr_submit_batch runs the demo's STEP 2 and 3 in front of
 * the voxel grid on every scan it is given (reference examples/run_global_registration.cpp:136-160:
 * PatchWork::estimate_ground -> non-ground points -> ImageProjection::segmentCloud -> getValidSegments), i.e. the pair
 * descriptors then carry raw scans WITH their ground returns and a batch reproduces the demo's whole sequence per pair.
 * pw = ip = NULL switches it off again (the default).  A scan that is all ground gets QTR_ERR_BAD_ARG in its own record. */
QTR_API int qtr_set_batch_preprocess(qtr_handle* h, const qtr_pw_params* pw, const qtr_ip_params* ip);

/* Multi-GPU (BASELINE configs[3]): pairs are independent, so every process / device registers its own block of pair
 * ids and the ONLY exchange is the final gather of the fixed-size result records — RCCL over xGMI (one ncclAllGather of
 * n_local * sizeof(qtr_result) bytes per rank; latency-bound, far from link bandwidth).  One handle = one rank.
 *   qtr_comm_unique_id   rank 0 creates the 128-byte rendezvous id; the host application hands it to the other ranks
 *                        (MPI, a file, torch.distributed.broadcast ...)
 *   qtr_comm_init        joins the communicator on the handle's device (collective: every rank calls it)
 *   qtr_gather_results   all-gather: `all` receives world * n_local records in rank order on EVERY rank; n_local must
 *                        be the same on all ranks (QTR_ERR_BAD_ARG on every rank otherwise — nothing is overrun)
 *   qtr_gather_results_v the same for blocks of DIFFERENT lengths (a block partition of B pairs over `world` ranks
 *                        differs by one record between ranks; BASELINE configs[3]: 4096 pairs over any world size):
 *                        the counts are exchanged first, the blocks padded to the longest for the fixed-size
 *                        collective and trimmed on the way out.  `all` (capacity cap_all records) receives the
 *                        sum(counts) records in rank order, counts[world] (optional) every rank's count, *n_all
 *                        (optional) the total.  Collective: every rank calls it, also with n_local = 0.  When the
 *                        records do not fit SOME rank's cap_all, EVERY rank returns QTR_ERR_CAPACITY (the capacities
 *                        travel with the counts, so no rank is left waiting in the second collective).
 * librccl is opened at run time (dlopen), so single-GPU users do not need it.  QTR_ERR_HIP with the RCCL message in
 * qtr_last_error on failure. */
#define QTR_COMM_ID_BYTES 128
QTR_API int qtr_comm_unique_id(char id[QTR_COMM_ID_BYTES]);
QTR_API int qtr_comm_init(qtr_handle* h, const char id[QTR_COMM_ID_BYTES], int rank, int world);
QTR_API int qtr_gather_results(qtr_handle* h, const qtr_result* local, int n_local, qtr_result* all);
QTR_API int qtr_gather_results_v(qtr_handle* h, const qtr_result* local, int n_local, qtr_result* all, int cap_all, int* counts,
                         int* n_all);
QTR_API void qtr_comm_destroy(qtr_handle* h);

QTR_API int qtr_get_stage_times(qtr_handle* h, int slot, qtr_stage_times* out);
/* Instrumentation (no reference counterpart; the demo times its stages with std::chrono around the calls,
 * examples/run_global_registration.cpp:206-246).  qtr_set_stage_events(0) stops recording events altogether (every
 * stage field of qtr_stage_times and its total then read 0: an event record is a marker the queue retires before the
 * next launch starts, and a call recorded three to six of them); the two nearest-neighbour launches keep their event pairs, whose
 * elapsed times accumulate per slot: qtr_get_nn_totals returns (and optionally resets) the sum and the launch count
 * without a per-call query.  An event pair attached to a launch costs ~5 us of queue time on either side of it (four
 * such gaps per registration): qtr_set_nn_event_stride(h, n) attaches the pairs to every n-th match of a slot only
 * (default 1: every match; 0: never) — the totals then cover the launches that were timed. */
QTR_API int qtr_set_stage_events(qtr_handle* h, int on);
QTR_API int qtr_set_nn_event_stride(qtr_handle* h, int every);
QTR_API int qtr_get_nn_totals(qtr_handle* h, int slot, double* total_ms, long long* launches, int reset);
/* The two launches behind qtr_stage_times.nn_kernel apart, milliseconds (0 when the last match carried no events): every
 * row of the smaller cloud against the larger one, and the rows of the larger cloud that were chosen against the smaller
 * one — the two FLANN searches of teaser::Matcher::advancedMatching (src/teaser_utils/feature_matcher.cc:100-122).  A
 * call of its own rather than two more fields: qtr_stage_times keeps its size for callers built against earlier headers. */
QTR_API int qtr_get_nn_dir_times(qtr_handle* h, int slot, float* dir1_ms, float* dir2_ms);

/* Inspection of intermediates of the LAST call on a slot (tests / parity debugging).  Copies up to
 * `bytes` bytes to host memory `dst`; returns the number of bytes the item holds, or <0 on error. */
#define QTR_DBG_GRAPH_BITMAP 1   /* uint64[L][ceil(L/64)] adjacency, original labels */
#define QTR_DBG_CORE 2           /* int32[L] core numbers: exact at or above QTR_DBG_SOLVER_STATE[29] (the floor of the last
                                    solve, 0 = all exact; always 0 in the k-core heuristic mode), below it an upper bound
                                    that is itself below the floor */
#define QTR_DBG_PERM 3           /* int32[L] vertex id at each rank of the (core,id) order */
#define QTR_DBG_NBR_OFFSETS 4    /* int32[n+1] CSR offsets of the sorted radius-neighbour lists (last qtr_fpfh) */
#define QTR_DBG_NBR_INDEX 5      /* int32[...] neighbour indices */
#define QTR_DBG_NBR_DIST2 6      /* float[...] squared distances */
#define QTR_DBG_SPFH 7           /* float[n][33] */
#define QTR_DBG_NN_LARGE_OF_SMALL 8 /* int32[n_small] */
#define QTR_DBG_NN_SMALL_OF_LARGE 9 /* int32[n_large] (-1 where not queried) */
#define QTR_DBG_VOX_SRC 10       /* float4[n_src] voxelised source of the last qtr_register_pair */
#define QTR_DBG_VOX_TGT 11
#define QTR_DBG_CORR 12          /* int32[L][2] */
#define QTR_DBG_MATCH_STATS 13   /* int32[16]: [0] L, [3] cross-checked pairs, [4] tuple-test survivors, [5] swapped,
                                    [8],[9] rows sent to the exact NN re-check (dir 0/1), [10],[11] rows settled by the
                                    two-candidate exact compare */
#define QTR_DBG_SOLVER_STATE 14  /* int32[32]: mc, best_r, pos, done, t0, ub, batch, max_core, n_edges2, clique rounds,
                                    [10] k-core peeling rounds / iterations, [22] 1: the clique stage ran twice (second
                                    time with exact core numbers), [29] floor of the core numbers (0: all exact) */
QTR_API long long qtr_debug_fetch(qtr_handle* h, int slot, int what, void* dst, size_t bytes);

/* Evaluates the shared deterministic math (include/qtr_math.h) ON THE DEVICE, for the test that pins
 * host/device bit-equality: fn 0 atan2f(a,b), 1 acosf(a), 2 sinf(a) (theta in [0,1.2]), 3 cosf(a). */
QTR_API int qtr_debug_math(qtr_handle* h, int fn, const float* a, const float* b, float* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* QUATRO_HIP_H */
