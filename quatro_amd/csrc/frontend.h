// frontend.h — buffers and launchers of the front-end kernels (frontend.hip, match.hip).
#pragma once
#include "common.h"

#define QTR_KMAX 256        // capacity of one point's radius-neighbour list (entries)
#define RADIX_TILE 1024     // elements per radix-sort workgroup (one wavefront)

// per-cloud device counters (CloudBufs::counts, 16 ints)
enum { CNT_NVOX = 0, CNT_VOX_OVERFLOW = 1, CNT_NBR_TOTAL = 2, CNT_NBR_OVERFLOW = 3, CNT_GRID_OVERFLOW = 4, CNT_KMAX = 5 };
// matcher device counters (FrontBufs::mcounts, 16 ints)
// MC_RECHECKx: rows sent to the exact re-check; MC_RECHECKx + 2: rows settled by the two-candidate exact compare
enum { MC_NCORR = 0, MC_RECHECK0 = 8, MC_RECHECK1 = 9, MC_PAIRCMP0 = 10, MC_PAIRCMP1 = 11, MC_NCROSS = 3, MC_NTUPLE = 4, MC_SWAPPED = 5 };

// Host mailbox (ints): the kernel that finishes a phase stores the few counters the host needs straight into
// pinned host memory, so a phase boundary costs one stream synchronisation and no copy launches.
// MAIL_SEQ_*: written LAST (after a system-scope fence) with the sequence number the host passed to the phase, so
// the host can simply watch that word instead of going through the runtime's stream wait.
enum { MAIL_VOX0 = 0, MAIL_VOX1 = 16, MAIL_MATCH = 32, MAIL_CNT0 = 48, MAIL_CNT1 = 64, MAIL_SEQ_VOX0 = 96, MAIL_SEQ_VOX1 = 97,
       MAIL_SEQ_MATCH = 98, MAIL_SEQ_SOLVE = 99, MAIL_SOLVER = 128, MAIL_INTS = 512 };

struct CloudBufs {
  int* counts = nullptr;       // 16
  u32* mm = nullptr;           // 8: order-preserving encodings of min x,y,z / max x,y,z
  float4* vox = nullptr;       // [max_voxels] down-sampled cloud (or the qtr_fpfh input)
  float4* normals = nullptr;   // [max_voxels] nx,ny,nz,curvature
  float* spfh = nullptr;       // [max_voxels][33]
  float* fpfh = nullptr;       // [max_voxels][33]
  u64* keys_a = nullptr;       // [max_points] sort ping
  u64* keys_b = nullptr;       // [max_points] sort pong
  u32* hist = nullptr;         // radix histograms / block counters
  int* nbr_cnt = nullptr;      // [max_voxels]
  int* nbr_off = nullptr;      // [max_voxels+1] CSR view (exclusive scan of nbr_cnt) for inspection
  int* nbr_idx = nullptr;      // [max_voxels][QTR_KMAX]   (strided) ... compacted copy lives in nbr_idx_c
  float* nbr_d2 = nullptr;     // [max_voxels][QTR_KMAX]
  float4* spts = nullptr;      // [max_voxels] points in cell-sorted order, w = original index
  float4* raw_sorted = nullptr; // [max_points] raw scan gathered into voxel-sorted order
  int* ranges = nullptr;       // [max_voxels][9][2] candidate key ranges
  float* mean = nullptr;       // 4 floats: sequential float mean of the cloud (Matcher::normalizePoints)
  float* baseT = nullptr;      // [34][n_pad] k-major descriptors + |b|^2 row   (MFMA streamed operand)
  float* queryT = nullptr;     // [34][n_pad] -2*descriptor + ones row            (MFMA stationary operand)
  float* norms = nullptr;      // [max_voxels] |desc|^2
  u32* max_norm = nullptr;     // 1: bits of the largest |desc|^2
};

struct FrontBufs {
  int max_points = 0, max_voxels = 0;
  CloudBufs cloud[2];
  u64* best_small = nullptr;   // [max_voxels] packed (dist bits << 32 | index) running minima
  u64* best_large = nullptr;
  int* nn_of_small = nullptr;  // [n_small] index into the larger cloud
  int* nn_of_large = nullptr;  // [n_large] index into the smaller cloud
  int* cross_i = nullptr;      // cross-checked pairs in ascending i (index into larger cloud)
  int* cross_j = nullptr;
  int* flags = nullptr;        // [max_voxels] scratch flags
  int* scan = nullptr;         // [max_voxels+1] scratch scan
  int* passed = nullptr;       // [max_voxels] tuple-test pass flags
  int* tgt_of_src = nullptr;   // [max_voxels]
  int* corr = nullptr;         // [max_voxels][2]
  int* mcounts = nullptr;      // 16
  void* nn_partial = nullptr;  // [max_voxels_pad][32] NnPartial (16 B)
  int* recheck_rows = nullptr; // [max_voxels]
  float* recheck_thr = nullptr; // [max_voxels] per listed row: approximate best + 2 eps (candidates above it cannot win)
  int* mail = nullptr;         // device view of the slot's pinned host mailbox (see MAIL_* above); may be null
  float4* m_src = nullptr;     // where the matcher's last kernel should leave the matched keypoint clouds (or null)
  float4* m_tgt = nullptr;
  int m_cap = 0;               // capacity of m_src / m_tgt in points (the handle's max_corr)
  bool gathered = false;       // set by match_enqueue when it did
  int mail_seq = 0;            // sequence number the next phase-ending kernel publishes (set by the caller)
  int nn_engine = 1;           // 1 = MFMA + exact re-check, 0 = exact VALU only (QTR_NN_ENGINE=exact)
  int nn_target_waves = 0;     // QTR_NN_WAVES: waves per k_nn_mfma launch to aim at; 0 = one workgroup per compute unit
  int n_cu = 256;              // compute units of the device
  int nn_trace = 0;            // QTR_NN_TRACE=1: k_nn_mfma (first direction) leaves clocks per tile / workgroup lives in mcounts[12..15]
  int nn_events = 1;           // 0: do not bracket the nearest-neighbour launches with events
  hipEvent_t ev_nn[4] = {};    // brackets of the two nearest-neighbour launches (created by the handle)
};

size_t frontend_scratch_bytes(int max_points, int max_voxels);
void frontend_carve(FrontBufs& F, void* base, int max_points, int max_voxels);
hipError_t frontend_init_attributes();

hipError_t voxelize_enqueue(FrontBufs& F, int nc, const float4* const* raw, const int* P, float leaf, hipStream_t st);
hipError_t set_count_enqueue(CloudBufs& C, int which, int value, hipStream_t st);
hipError_t fpfh_enqueue(FrontBufs& F, int first, int nc, const int* n, float r_normal, float r_fpfh, hipStream_t st,
                        bool with_mean);
hipError_t mean_enqueue(FrontBufs& F, int first, int nc, const int* n, hipStream_t st);
hipError_t match_enqueue(FrontBufs& F, int ns, int nt, const qtr_frontend_params& fp, hipStream_t st);
hipError_t gather_matched_enqueue(FrontBufs& F, int L, float4* m_src, float4* m_tgt, hipStream_t st);

// shared small kernels (defined in frontend.hip)
hipError_t exclusive_scan_i32(const int* in, int* out, int n, hipStream_t st);  // out has n+1 entries
