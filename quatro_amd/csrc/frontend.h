// frontend.h — buffers and launchers of the front-end kernels (frontend.hip, match.hip).
#pragma once
#include "common.h"

#define QTR_KMAX 256        // entries of a point's radius-neighbour list kept in its own fixed-stride slot; a longer list
                            // lives in the cloud's long-list arena (nbr_big_*), its offset in word 0 of the slot
#define NBIG_LDS_KEYS 8192  // longest list k2_neighbors_big sorts in LDS (longer ones are sorted in the arena itself)
#define RADIX_TILE 1024     // elements per radix-sort workgroup (four wavefronts, 256 keys each)
#define NORM_BINS 192       // bins of width 1 over sqrt(|descriptor|^2) (<= sqrt(3 * 100^2) = 173.3)

// per-cloud device counters (CloudBufs::counts, 16 ints)
// the largest cloud the matcher takes: k_recheck_filter's per-wave lists hold (listed column << 20 | base row) in 32 bits —
// 20 bits of base row (padded to whole 32-row tiles), 12 of column; qtr_create refuses limits above it
#define QTR_NN_MAX_ROWS ((1 << 20) - 32)
enum { CNT_NVOX = 0, CNT_VOX_OVERFLOW = 1, CNT_NBR_TOTAL = 2, CNT_NBR_OVERFLOW = 3, CNT_GRID_OVERFLOW = 4, CNT_KMAX = 5,
       CNT_SORT_BITS = 6 /* significant bits of the voxel sort's keys */,
       CNT_NBR_ARENA = 7 /* entries of the long-list arena handed out */, CNT_NBR_CAPACITY = 8 /* ... it was too small */,
       CNT_VOX_TAILERR = 9 /* sticky: SOME tile of k2_vox_centroids (or of k2_cell_scan) gave up its look-back */,
       CNT_NCELL = 10 /* cells of the neighbour-search grid over this cloud's bounding box (voxel stage, for the cell side
                         the caller named): up to QTR_CELL_CAP the FPFH chain places the points by a dense cell table */ };
// the largest neighbour-search grid served by the dense cell table (k2_cell_count / _scan / _place); above it — 160 x 160 x 30 m
// at 0.75 m cells is 1.8 M — the chain sorts packed cell keys as it did until round 6
#define QTR_CELL_CAP (1 << 21)
// CNT_NBR_OVERFLOW: some point of the cloud has more than QTR_KMAX neighbours (k2_neighbors_big has work to do);
// CNT_KMAX: the longest such list
// matcher device counters (FrontBufs::mcounts, 16 ints)
// MC_RECHECKx: rows sent to the exact re-check; MC_RECHECKx + 2: rows settled by the two-candidate exact compare
// MC_NQ0 / MC_NHIT: query counts of the two nearest-neighbour directions (device-side: the second direction only asks
// for the rows of the larger cloud that some row of the smaller cloud points at); MC_HIDDEN_I / _J: descriptor rows
// hidden from the base tables because a lower row holds the bit-identical descriptor
enum { MC_NCORR = 0, MC_HIDDEN_I = 1, MC_HIDDEN_J = 2, MC_NCROSS = 3, MC_NTUPLE = 4, MC_SWAPPED = 5, MC_NQ0 = 6, MC_NHIT = 7,
       MC_RECHECK0 = 8, MC_RECHECK1 = 9, MC_NSPLIT0 = 10, MC_NSPLIT1 = 11 /* plan words of the two k_nn_f16 launches: slices per query block | base tiles per slice << 8 */,
       MC_UNSAFE = 12,
       MC_TAILERR = 13 /* a multi-workgroup compaction of the tail gave up waiting for a predecessor's count */ };

// Host mailbox (ints): the kernel that finishes a phase stores the few counters the host needs straight into
// pinned host memory, so a phase boundary costs one stream synchronisation and no copy launches.
// MAIL_SEQ_*: written LAST (after a system-scope fence) with the sequence number the host passed to the phase, so
// the host can simply watch that word instead of going through the runtime's stream wait.
enum { MAIL_VOX0 = 0, MAIL_VOX1 = 16, MAIL_MATCH = 32, MAIL_CNT0 = 48, MAIL_CNT1 = 64, MAIL_SEQ_VOX0 = 96, MAIL_SEQ_VOX1 = 97,
       MAIL_SEQ_MATCH = 98, MAIL_SEQ_SOLVE = 99, MAIL_SOLVER = 128, MAIL_INTS = 512 };

struct CloudBufs {
  int* counts = nullptr;       // 16
  u32* mm = nullptr;           // 8: order-preserving encodings of min x,y,z / max x,y,z
  u32* mm_part = nullptr;      // one such record per workgroup of k2_minmax (folded by k2_keys_hist)
  float4* vox = nullptr;       // [max_voxels] down-sampled cloud (or the qtr_fpfh input)
  float4* normals = nullptr;   // [max_voxels] nx,ny,nz,curvature
  float* spfh = nullptr;       // [max_voxels][33]
  float* fpfh = nullptr;       // [max_voxels][33]
  u64* keys_a = nullptr;       // [max_points] sort ping
  u64* keys_b = nullptr;       // [max_points] sort pong
  u32* hist = nullptr;         // radix histograms / block counters
  int* vox_look = nullptr;     // one look-back word per 1024-point tile of the voxel grid's centroid kernel
  int* nbr_cnt = nullptr;      // [max_voxels]
  int* nbr_off = nullptr;      // [max_voxels+1] CSR view (exclusive scan of nbr_cnt) for inspection
  int* nbr_idx = nullptr;      // [max_voxels][QTR_KMAX]   (strided) ... compacted copy lives in nbr_idx_c
  float* nbr_d2 = nullptr;     // [max_voxels][QTR_KMAX]
  int* nbr_big_idx = nullptr;  // [nbr_big_cap] long-list arena (lists of more than QTR_KMAX entries), separately allocated
  float* nbr_big_d2 = nullptr;
  int nbr_big_cap = 0;
  float4* spts = nullptr;      // [max_voxels] points in cell-sorted order, w = original index
  float4* raw_sorted = nullptr; // [max_points] raw scan gathered into voxel-sorted order
  int* ranges = nullptr;       // [max_voxels][9][2] candidate key ranges
  int* cell_cnt = nullptr;     // [QTR_CELL_CAP + 4096] points per cell of the neighbour-search grid: zero between uses (k2_cell_scan leaves it so)
  int* cell_start = nullptr;   // [QTR_CELL_CAP + 4096] first place of every cell in spts (exclusive scan of cell_cnt)
  float* mean = nullptr;       // 4 floats: sequential float mean of the cloud (Matcher::normalizePoints)
  float* baseT = nullptr;      // [34][n_pad] k-major descriptors + |b|^2 row   (MFMA streamed operand)
  float* queryT = nullptr;     // [34][n_pad] -2*descriptor + ones row            (MFMA stationary operand)
  float* norms = nullptr;      // [max_voxels] |desc|^2
  u32* max_norm = nullptr;     // 1: bits of the largest |desc|^2
  float* baseTb = nullptr;     // [34][n_pad] baseT with its columns in norm-bin order (exact re-check)
  int* nb_row = nullptr;       // [max_voxels] row of every column of baseTb (rows sorted by bin of sqrt|desc|^2)
  int* nb_start = nullptr;     // [NORM_BINS + 1] first column of every bin
  uint4* baseH = nullptr;      // [n_pad/32][14][32] x 16 B: the f16-split base operand of k_nn_f16 (see match.hip)
  uint4* queryH = nullptr;     // same layout: the f16-split query operand (all rows of the cloud)
  u64* dd_hash = nullptr;      // [max_voxels] 64-bit hash of the descriptor bits
  u64* dd_table = nullptr;     // [dd_slots] open-addressing table: (hash tag << 32) | lowest row holding that hash
};

// What a front-end kernel needs to know about one cloud; kernels pick theirs with blockIdx.y (see Clouds2).
struct CloudView {
  const float4* raw;   // raw scan (voxelise)
  int P;               // raw points
  int n;               // voxel count (known on the host after the voxelise read-back)
  int* counts;
  u32* mm;
  u32* mm_part;        // k2_minmax's per-workgroup records
  float4* vox;
  float4* normals;
  float* spfh;
  float* fpfh;
  u64* keys_a;         // sort ping / pong; a kernel argument says which one holds the input of the pass
  u64* keys_b;
  u32* hist;
  int* mail;           // host mailbox slot of this cloud's voxelise counters (or null)
  int* mail_seq_slot;  // ... and the word that receives the sequence number after them
  int seq;             // ... and that number
  int* blkcnt;
  int* blkoff;
  int* vox_look;
  int* nbr_cnt;
  int* nbr_off;
  int* nbr_idx;
  float* nbr_d2;
  int* nbr_big_idx;
  float* nbr_big_d2;
  int nbr_big_cap;
  float4* spts;
  int* ranges;
  int* cell_cnt;
  int* cell_start;
  float* mean;
  // the matcher's per-descriptor preparation at the end of k2_fpfh (whole-path chains; null / 0: k_desc_prep does it):
  // |d|^2, the hash and the row's entry in the duplicate table, which k2_normals cleared earlier in the chain
  float* norms;
  u64* dd_hash;
  u64* dd_table;
  int dd_mask;
};
struct Clouds2 {
  CloudView c[2];        // up to two clouds travel in the kernel arguments ...
  const CloudView* ext;  // ... a batch of pairs puts the array in device memory (null otherwise)
};

// One direction of the 33-D nearest-neighbour search of one pair.
struct NnDir {
  const float* baseT;   // [34][nb_pad] k-major base table (row 33: scaled |b|^2; hidden / pad rows 1e30)
  const float* bnorm;   // [nb] |b|^2 as rounded once from binary64
  int nb, nb_pad;
  const float* queryT;  // [34][nq_pad] the table k_nn_mfma reads (direction 1: the compacted hit rows)
  const uint4* baseH;   // f16-split operands of k_nn_f16: [rows/32][14][32] x 16 B (direction 1's queryH: compacted hit rows)
  const uint4* queryH;
  const float* qnorm;   // [.] |a|^2 per column of that table
  const int* qmap;      // column -> row of the query cloud (null: identity)
  int nq_pad;
  const float* A;       // [n][33] descriptors of the query cloud (exact re-check)
  const float* QT;      // its full k-major table (-2a rows), stride qt_pad
  int qt_pad;
  const float* baseTb;  // baseT with its columns in norm-bin order; brow: row of each column; bstart: first column per bin
  const int* brow;
  const int* bstart;
  const int* qorder;    // queries visited in this order by k_nn_finish (norm-bin order of the query cloud), or null
  u64* best;            // per row of the query cloud: packed (distance bits << 32 | index), or the bare index
  int nq_slot, rc_slot; // mcounts indices: number of queries, re-check counter
};
struct NnPartial {
  float b1, b2;
  int i1;
  int pad;
};
// Everything the matcher kernels need for one pair; picked with blockIdx.z (see MatchArgs).
#define TAIL_MAXWG 256  // look-back words per multi-workgroup compaction of the matcher's tail (two of them, in MatchView::scan)
struct MatchView {
  NnDir d[2];                  // 0: rows of the smaller cloud ask the larger one; 1: hit rows of the larger ask the smaller
  const float4 *vox_i, *vox_j; // i = larger cloud (fi), j = smaller (fj), reference feature_matcher.cc:84-92
  const float *mean_i, *mean_j;
  const float *fpfh_i, *fpfh_j;
  float *baseT_i, *queryT_i, *norms_i, *baseT_j, *queryT_j, *norms_j;
  float *baseTb_i, *baseTb_j;
  uint4 *baseH_i, *baseH_j, *queryH_i, *queryH_j, *queryH_c;
  int *nb_row_i, *nb_row_j, *nb_start_i, *nb_start_j;
  u64 *hash_i, *hash_j, *table_i, *table_j;
  int dd_mask;                 // table slots - 1
  int n_large, n_small, pad_large, pad_small, swapped, ns, nt;
  u64 *best_small, *best_large;
  int *nn_of_small, *nn_of_large, *cross_i, *cross_j, *flags, *scan, *passed, *tgt_of_src, *corr, *mcounts;
  NnPartial* partial;
  int* recheck_rows;
  float* recheck_thr;
  int2* recheck_span;          // per listed row: [first, last) column of the base cloud's norm-bin order that can hold its arg-min
  int* recheck_q;              // per listed row: its column of the direction's query table (k_recheck_filter)
  int* hit_rows;               // ascending rows of the larger cloud that direction 0 points at
  float* queryT_c;             // [34][pad_large] their columns of queryT_i
  float* norms_c;
  const float4 *vox_s, *vox_t; // source / target voxel clouds (un-swapped)
  float4 *m_src, *m_tgt;       // matched keypoint clouds (or null)
  int m_cap;
  int* mail;                   // host mailbox of the pair's slot (or null)
  const int *counts0, *counts1;
  int seq;
  int *nc_cnt, *nc_fill, *nc_off, *nc_list;  // use_crosscheck = 0: per-source counts, fill cursors, offsets, targets
  int crosscheck;              // 0: corres_ij + corres_ji go to the tuple test unfiltered (feature_matcher.cc:146-181)
  int tuple;                   // 1: run the tuple test
  float tuple_scale;
  u64 seed;
};
struct MatchArgs {
  MatchView one;
  const MatchView* ext;
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
  int2* recheck_span = nullptr; // [max_voxels]
  int* recheck_q = nullptr;    // [max_voxels]
  int* hit_rows = nullptr;     // [max_voxels]
  float* queryT_c = nullptr;   // [34][max_voxels_pad]
  float* norms_c = nullptr;    // [max_voxels_pad]
  uint4* queryH_c = nullptr;   // [max_voxels_pad/32][14][32] x 16 B: f16-split query operand of the hit rows
  int dd_slots = 0;            // slots of each cloud's dedup table (power of two >= 2 * max_voxels)
  int *nc_cnt = nullptr, *nc_fill = nullptr, *nc_off = nullptr, *nc_list = nullptr;  // cross-check off: per-source target lists
  int* mail = nullptr;         // device view of the slot's pinned host mailbox (see MAIL_* above); may be null
  float4* m_src = nullptr;     // where the matcher's last kernel should leave the matched keypoint clouds (or null)
  float4* m_tgt = nullptr;
  int m_cap = 0;               // capacity of m_src / m_tgt in points (the handle's max_corr)
  int vox_passes = 4;          // whole-path driver: radix passes the voxel sort needed last time (speculation, see capi.hip)
  int vox_fewer = 0;           // ... and for how many calls in a row fewer would have done
  bool gathered = false;       // set by match_enqueue when it did
  int mail_seq = 0;            // sequence number the next phase-ending kernel publishes (set by the caller)
  int nn_engine = 2;           // 2 = f16-split MFMA filter + exact re-check (default), 1 = f32 MFMA + exact re-check
                               // (QTR_NN_ENGINE=mfma32), 0 = exact VALU only (QTR_NN_ENGINE=exact)
  int n_cu = 256;              // compute units of the device
  int nn_events = 1;           // 0: do not bracket the nearest-neighbour launches with events
  hipEvent_t ev_nn[4] = {};    // brackets of the two nearest-neighbour launches (created by the handle)
};

size_t frontend_scratch_bytes(int max_points, int max_voxels);
void frontend_carve(FrontBufs& F, void* base, int max_points, int max_voxels);
hipError_t frontend_init_attributes();

// passes: radix passes of the voxel sort to launch.  The keys' significant bits (CNT_SORT_BITS, in the mailbox with the
// voxel counts) are only known on the device: a caller that launches fewer than 4 must check them afterwards and run the
// stage again with enough passes when ceil(bits / 8) exceeds what it launched (the output is then unsorted garbage inside
// its bounds)
// cell_side: the cell of the FPFH chain that follows (r_fpfh * 1.001), or 0 — the voxel stage then mails the neighbour grid's
// cell count (CNT_NCELL) with its counters
hipError_t voxelize_enqueue(FrontBufs& F, int nc, const float4* const* raw, const int* P, float leaf, hipStream_t st,
                            int passes = 4, float cell_side = 0.f);
hipError_t set_count_enqueue(CloudBufs& C, int which, int value, hipStream_t st);
// origin_known: the clouds are the voxel centroids voxelize_enqueue just produced in the same CloudBufs (its bounding box
// is still there and serves as the neighbour grid's origin)
hipError_t fpfh_enqueue(FrontBufs& F, int first, int nc, const int* n, float r_normal, float r_fpfh, hipStream_t st,
                        bool with_mean, bool origin_known, bool long_lists, bool desc_prep = false,
                        int max_ncell = 0 /* > 0 (and <= QTR_CELL_CAP, origin_known): the dense cell table, see frontend.hip */);
hipError_t mean_enqueue(FrontBufs& F, int first, int nc, const int* n, hipStream_t st, int cap = 0);
// init_done: match_init_enqueue already ran for this pair (same ns, nt, fp) — the whole-path driver issues it beside the
// FPFH chain, which takes one launch off the critical path
// prep_done: the clouds' k2_fpfh did k_desc_prep's work (fpfh_enqueue* with desc_prep) — then the init may not clear the
// duplicate tables (k2_normals did, before they were filled)
hipError_t match_enqueue(FrontBufs& F, int ns, int nt, const qtr_frontend_params& fp, hipStream_t st, bool init_done = false,
                         bool prep_done = false);
hipError_t match_init_enqueue(FrontBufs& F, int ns, int nt, const qtr_frontend_params& fp, hipStream_t st,
                              bool clear_tables, int* zero_words = nullptr, int n_zero = 0);  // zero_words: n_zero ints the launch clears for the caller
hipError_t gather_matched_enqueue(FrontBufs& F, int L, float4* m_src, float4* m_tgt, hipStream_t st);

// The same stages for G pairs at once (qtr_submit_batch): one launch chain, the views of all pairs in device memory
// (pushed through `stage`).  F[g] is pair g's arena; raw / P / n hold two entries per pair (source, target).
hipError_t voxelize_enqueue_group(FrontBufs* const* F, int G, const float4* const* raw, const int* P, float leaf,
                                  ViewStage* stage, hipStream_t st, float cell_side = 0.f);
hipError_t mean_enqueue_group(FrontBufs* const* F, int G, const int* n, ViewStage* stage, hipStream_t st);
hipError_t fpfh_enqueue_group(FrontBufs* const* F, int G, const int* n, float r_normal, float r_fpfh, ViewStage* stage,
                              hipStream_t st, bool long_lists, bool desc_prep = false, int max_ncell = 0);
hipError_t match_enqueue_group(FrontBufs* const* F, int G, const int* n, const qtr_frontend_params* fp,
                               const unsigned long long* seeds, ViewStage* stage, hipStream_t st, bool prep_done = false);

// shared small kernels (defined in frontend.hip)
hipError_t exclusive_scan_i32(const int* in, int* out, int n, hipStream_t st);  // out has n+1 entries
