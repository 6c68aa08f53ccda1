// capi.hip — implementation of the C ABI declared in include/quatro_hip.h.
// Host-side orchestration only: arenas, stream slots, staging copies, kernel sequencing.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "common.h"
#include "frontend.h"
#include "solver.h"

struct RcclIdByValue {  // ncclUniqueId: passed by value to ncclCommInitRank
  char internal[QTR_COMM_ID_BYTES];
};

struct Slot {
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // second cloud's front end runs concurrently
  hipEvent_t ev[8] = {};
  hipEvent_t ev_vox = nullptr;    // voxel centroids complete (stream) -> Matcher means may start (stream2)
  void* solver_arena = nullptr;
  void* front_arena = nullptr;
  SolverBufs sb;
  FrontBufs fb;
  float4* in_src = nullptr;  // staging for host-resident inputs (max_points each)
  float4* in_tgt = nullptr;
  float4* m_src = nullptr;  // matched keypoints (max_corr each)
  float4* m_tgt = nullptr;
  int* pinned_i32 = nullptr;     // >= 64 ints
  qtr_result* pinned_res = nullptr;
  SegBufs seg;                   // range-image segmentation arena (allocated on first use)
  void* seg_arena = nullptr;
  ExactBufs exb;                 // exact clique search arena (allocated / grown on demand)
  void* ex_arena = nullptr;
  size_t ex_bytes = 0;
  PwBufs pwb;                    // ground segmentation arena (allocated on first use)
  void* pw_arena = nullptr;
  int* mail = nullptr;           // pinned host mailbox the phase-ending kernels write into (frontend.h MAIL_*)
  int seq = 0;                   // last sequence number handed to a phase-ending kernel
  unsigned long long exact_nodes = 0;  // search-tree nodes of the last PMC_EXACT run
  int exact_aborted = 0;               // 1: its time limit was hit (heuristic clique returned)
  int times_pending = 0;         // 1: qtr_solve, 2: qtr_register_pair — stage times are read off the events lazily
  int nn_pending = 0;            // the nearest-neighbour events of the last match have not been added to the totals yet
  int nn_timed_last = 0;         // the last match had its event pairs attached (qtr_set_nn_event_stride)
  long long n_matches = 0;       // matches this slot has run
  float nn_dir_ms[2] = {0, 0};   // qtr_get_nn_dir_times: the two launches of the last timed match apart
  double nn_total_ms = 0;        // qtr_get_nn_totals
  long long nn_total_launches = 0;
  qtr_stage_times times = {};
  int last_L = 0;  // correspondences of the last solve
  int last_Wb = 0; // row stride (words) of the bit matrix it left behind (solver.h SolverView::Wb)
  int last_n = 0;  // points of the last qtr_fpfh
  int last_ns = 0, last_nt = 0;
};

// One lane of the batch driver (qtr_submit_batch): a contiguous group of slots stepped through the three launch
// chains in lockstep on the first slot's streams.
struct Lane {
  int first_slot = 0, cap = 0;  // slots [first_slot, first_slot + cap)
  ViewStage stage;              // views of the group (pinned host + device)
  int phase = 0;                // 0 idle, 1 voxelise pending, 2 matching pending, 3 solver pending
  int first_pair = 0, count = 0;  // pairs [first_pair, first_pair + count) of the job are on this lane
  std::vector<int> active;      // indices g (0..count) of the pairs still alive after each chain's checks
  std::vector<int> ns, nt, L;   // per g
  std::vector<int> corr_only;   // g of the pairs that bring their own correspondences and no scans: solver chain only
  std::vector<const float4*> csrc, ctgt;  // per g: the matched clouds the solver reads (slot's m_src / m_tgt or the caller's)
  std::vector<const float4*> raw_s, raw_t;  // per g: the clouds the voxel grid read (device pointers) and their sizes
  std::vector<int> Ps, Pt;
  bool long_lists = false;      // the chunk's FPFH chain included k2_neighbors_big
};
struct BatchJob {
  const qtr_pair_desc* pairs = nullptr;
  int B = 0, next = 0, done = 0;
  qtr_frontend_params fp;
  qtr_params prm;
  qtr_result* results = nullptr;
  int mem = QTR_MEM_HOST;
  bool active = false;
  std::vector<unsigned char> finished;  // per pair: its record is final (result or per-pair failure)
};

struct qtr_handle {
  int device = 0;
  qtr_limits lim;
  std::vector<Slot> slots;
  std::vector<Lane> lanes;
  BatchJob job;
  void* comm = nullptr;        // ncclComm_t of this rank (qtr_comm_init)
  int comm_rank = 0, comm_world = 1;
  void* comm_buf = nullptr;    // device staging of the gather
  size_t comm_bytes = 0;
  std::atomic<int> solves_in_flight{0};  // back-end chains enqueued and not yet read back (slot calls from several threads)
  int spin_wait = 1;  // QTR_HOST_WAIT=block turns the mailbox polling off
  bool pre_on = false;      // qtr_set_batch_preprocess: the batched entry takes RAW sweeps (ground removal + range-image
  qtr_pw_params pre_pw;     // segmentation in front of the voxel grid)
  qtr_ip_params pre_ip;
  std::atomic<bool> long_lists{false};  // some cloud of the whole-path entry points had a point with more than QTR_KMAX
                            // neighbours: from then on their FPFH chains include k2_neighbors_big (see front_device); written
                            // by whichever slot call meets such a cloud first (calls from several threads)
  // LIVE host threads that have run a back-end chain on this handle (InFlight): the block outlives the handle as long as a
  // thread still holds it, so a thread that ends after qtr_destroy can still take itself off the count
  struct Callers {
    std::atomic<int> live{0};
  };
  std::shared_ptr<Callers> callers = std::make_shared<Callers>();
  std::mutex share_mu;                 // guards share_val / share_gen
  int share_val = 1, share_gen = 0;    // the share in force and how often it has changed
  std::atomic<int> gen_inflight[4] = {};  // chains in flight per generation of the share (gen & 3)
  unsigned long long uid = 0;     // process-unique id of this handle (a thread remembers the handle it registered with)
  int stage_events = 1;  // QTR_STAGE_EVENTS=0: only the first/last event of a call are recorded (stage times read 0)
  int nn_event_stride = 1;  // every n-th match of a slot carries the nearest-neighbour event pairs (0: none)
  char err[512];
};

static std::atomic<unsigned long long> g_handle_uid{0};

#define QTR_TRY(expr)                  \
  do {                                 \
    const int rc_ = (expr);            \
    if (rc_ != QTR_OK) return rc_;     \
  } while (0)

// ---- RCCL, opened at run time (the soname torch ships resolves to the copy that is already loaded)
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclIdByValue, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi* rccl_api(char* err, size_t errn) {
  static RcclApi api;
  static int state = 0;  // 0 untried, 1 ok, -1 failed
  static char why[256] = "symbols missing";  // dlerror() is read ONCE per failure (a second call returns NULL)
  if (state == 0) {
    // QTR_RCCL_LIB: a site's own RCCL build, tried first (tests/test_gpu_multi.py points it at a transport double so
    // that several processes sharing one GPU can run the gather end to end)
    const char* names[] = {getenv("QTR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (api.lib) break;
      if (!n || !*n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (!api.lib) {
        const char* e = dlerror();
        if (e) snprintf(why, sizeof(why), "%s", e);
      }
    }
    if (api.lib) {
      snprintf(why, sizeof(why), "symbols missing");
      api.GetUniqueId = (int (*)(void*))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (int (*)(void**, int, RcclIdByValue, int))dlsym(api.lib, "ncclCommInitRank");
      api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
      api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
      api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
      const char* e = dlerror();
      if (e) snprintf(why, sizeof(why), "%s", e);
    }
    state = (api.lib && api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy) ? 1 : -1;
  }
  if (state != 1) {
    if (err) snprintf(err, errn, "librccl could not be opened (%s)", why);
    return nullptr;
  }
  return &api;
}

extern "C" {

int qtr_comm_unique_id(char id[QTR_COMM_ID_BYTES]) {
  RcclApi* a = rccl_api(nullptr, 0);
  if (!a || !id) return QTR_ERR_HIP;
  return a->GetUniqueId(id) == 0 ? QTR_OK : QTR_ERR_HIP;
}

int qtr_comm_init(qtr_handle* h, const char id[QTR_COMM_ID_BYTES], int rank, int world) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return QTR_ERR_BAD_ARG;
  RcclApi* a = rccl_api(h->err, sizeof(h->err));
  if (!a) return QTR_ERR_HIP;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  qtr_comm_destroy(h);
  RcclIdByValue v;
  memcpy(v.internal, id, QTR_COMM_ID_BYTES);
  const int rc = a->CommInitRank(&h->comm, world, v, rank);
  if (rc != 0) {
    snprintf(h->err, sizeof(h->err), "ncclCommInitRank: %s", a->GetErrorString ? a->GetErrorString(rc) : "error");
    h->comm = nullptr;
    return QTR_ERR_HIP;
  }
  h->comm_rank = rank;
  h->comm_world = world;
  return QTR_OK;
}

// one all-gather of `bytes` per rank from d_send into d_recv (rank order), on slot 0's stream
static int comm_allgather(qtr_handle* h, RcclApi* a, const void* d_send, void* d_recv, size_t bytes, hipStream_t st) {
  const int rc = a->AllGather(d_send, d_recv, bytes, /* ncclChar */ 0, h->comm, st);
  if (rc != 0) {
    snprintf(h->err, sizeof(h->err), "ncclAllGather: %s", a->GetErrorString ? a->GetErrorString(rc) : "error");
    return QTR_ERR_HIP;
  }
  return QTR_OK;
}

// require_equal: qtr_gather_results — blocks of different lengths are refused.  Every decision to leave before the second
// collective is taken from the GATHERED line (counts, capacities, flags: identical on every rank), never from a rank's
// own arguments: a rank that bailed out alone would leave the others waiting in ncclAllGather for good.
static int gather_impl(qtr_handle* h, const qtr_result* local, int n_local, qtr_result* all, int cap_all, int* counts,
                       int* n_all, bool require_equal) {
  if (!h || n_local < 0 || (n_local > 0 && !local) || cap_all < 0 || (cap_all > 0 && !all)) return QTR_ERR_BAD_ARG;
  if (!h->comm) {
    snprintf(h->err, sizeof(h->err), "qtr_gather_results: call qtr_comm_init first");
    return QTR_ERR_BAD_ARG;
  }
  RcclApi* a = rccl_api(h->err, sizeof(h->err));
  if (!a) return QTR_ERR_HIP;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  const int world = h->comm_world;
  hipStream_t st = h->slots[0].stream;
  auto reserve = [&](size_t bytes) -> int {
    if (h->comm_bytes >= bytes) return QTR_OK;
    if (h->comm_buf) (void)hipFree(h->comm_buf);
    h->comm_buf = nullptr;
    h->comm_bytes = 0;
    QTR_HIP_TRY(h, hipMalloc(&h->comm_buf, bytes));
    h->comm_bytes = bytes;
    return QTR_OK;
  };
  // 1. every rank's record count, the room it has for the result and whether it insists on equal blocks (a collective:
  //    ranks with nothing to send take part too)
  QTR_TRY(reserve(256 + 256 * (size_t)world));
  std::vector<int> cnt((size_t)world * 64, 0);  // one 256-byte line per rank
  int mine_line[64] = {n_local, cap_all, require_equal ? 1 : 0};
  QTR_HIP_TRY(h, hipMemcpyAsync(h->comm_buf, mine_line, 256, hipMemcpyHostToDevice, st));
  QTR_TRY(comm_allgather(h, a, h->comm_buf, (char*)h->comm_buf + 256, 256, st));
  QTR_HIP_TRY(h, hipMemcpyAsync(cnt.data(), (char*)h->comm_buf + 256, 256 * (size_t)world, hipMemcpyDeviceToHost, st));
  QTR_HIP_TRY(h, hipStreamSynchronize(st));
  int nmax = 0, nmin = 0x7fffffff;
  long long total = 0;
  bool want_equal = false;
  for (int r = 0; r < world; ++r) {
    const int c = cnt[(size_t)r * 64];
    if (counts) counts[r] = c;
    nmax = std::max(nmax, c);
    nmin = std::min(nmin, c);
    total += c;
    want_equal = want_equal || cnt[(size_t)r * 64 + 2] != 0;
  }
  if (n_all) *n_all = (int)total;
  if (want_equal && nmin != nmax) {  // (the same verdict on every rank)
    snprintf(h->err, sizeof(h->err), "qtr_gather_results: ranks hold different record counts (%d .. %d; use qtr_gather_results_v)",
             nmin, nmax);
    return QTR_ERR_BAD_ARG;
  }
  for (int r = 0; r < world; ++r)
    if (total > cnt[(size_t)r * 64 + 1]) {  // some rank has no room: every rank leaves here, together
      snprintf(h->err, sizeof(h->err), "qtr_gather_results: %lld records, rank %d has room for %d", total, r,
               cnt[(size_t)r * 64 + 1]);
      return QTR_ERR_CAPACITY;
    }
  if (nmax == 0) return QTR_OK;
  // 2. the records, every block padded to the longest (fixed-size collective), trimmed on the way out
  const size_t block = (size_t)nmax * sizeof(qtr_result);
  QTR_TRY(reserve(block * (size_t)(world + 1)));
  char* d_send = (char*)h->comm_buf;
  char* d_recv = d_send + block;
  QTR_HIP_TRY(h, hipMemsetAsync(d_send, 0, block, st));
  if (n_local > 0)
    QTR_HIP_TRY(h, hipMemcpyAsync(d_send, local, (size_t)n_local * sizeof(qtr_result), hipMemcpyHostToDevice, st));
  QTR_TRY(comm_allgather(h, a, d_send, d_recv, block, st));
  size_t off = 0;
  for (int r = 0; r < world; ++r) {
    const int c = cnt[(size_t)r * 64];
    if (c > 0)
      QTR_HIP_TRY(h, hipMemcpyAsync(all + off, d_recv + (size_t)r * block, (size_t)c * sizeof(qtr_result),
                                    hipMemcpyDeviceToHost, st));
    off += (size_t)c;
  }
  QTR_HIP_TRY(h, hipStreamSynchronize(st));
  return QTR_OK;
}

int qtr_gather_results_v(qtr_handle* h, const qtr_result* local, int n_local, qtr_result* all, int cap_all, int* counts,
                         int* n_all) {
  return gather_impl(h, local, n_local, all, cap_all, counts, n_all, false);
}

int qtr_gather_results(qtr_handle* h, const qtr_result* local, int n_local, qtr_result* all) {
  if (!h || n_local < 0 || (n_local > 0 && (!local || !all))) return QTR_ERR_BAD_ARG;
  // the caller sized `all` for world * n_local records: blocks of another length are refused (on every rank), not overrun
  const long long room = (long long)n_local * std::max(h->comm_world, 1);
  return gather_impl(h, local, n_local, all, (int)std::min<long long>(room, 0x7fffffff), nullptr, nullptr, true);
}

void qtr_comm_destroy(qtr_handle* h) {
  if (!h) return;
  if (h->comm) {
    RcclApi* a = rccl_api(nullptr, 0);
    if (a) (void)a->CommDestroy(h->comm);
    h->comm = nullptr;
  }
  if (h->comm_buf) {
    (void)hipFree(h->comm_buf);
    h->comm_buf = nullptr;
    h->comm_bytes = 0;
  }
}

int qtr_exact_stats(qtr_handle* h, int slot, unsigned long long* nodes, int* aborted) {
  if (!h || slot < 0 || slot >= (int)h->slots.size()) return QTR_ERR_BAD_ARG;
  if (nodes) *nodes = h->slots[slot].exact_nodes;
  if (aborted) *aborted = h->slots[slot].exact_aborted;
  return QTR_OK;
}

void qtr_default_limits(qtr_limits* l) {
  l->max_points = 262144;
  l->max_voxels = 65536;
  l->max_corr = 24576;
  l->n_slots = 1;
  l->max_long_neighbors = 0;  // = 128 * max_voxels
}

void qtr_default_params(qtr_params* p) {  // Quatro::Params defaults, reference include/quatro.hpp:202-268
  memset(p, 0, sizeof(*p));
  p->noise_bound = 0.3;
  p->cbar2 = 1.0;
  p->rotation_gnc_factor = 1.4;
  p->rotation_cost_threshold = 1e-6;
  p->kcore_heuristic_threshold = 0.5;
  p->cote_noise_bound = 0.3;
  p->ryrx[0] = p->ryrx[4] = p->ryrx[8] = 1.0;
  p->rotation_max_iterations = 100;
  p->inlier_selection_mode = QTR_INLIER_PMC_HEU;
  p->cote_median = 1;
  p->max_clique_time_limit = 3600;
}

void qtr_demo_params(qtr_params* p) {  // reference config/params.yaml:22-44
  qtr_default_params(p);
  p->rotation_cost_threshold = 1.1e-4;
  p->rotation_max_iterations = 50;
}

void qtr_default_frontend_params(qtr_frontend_params* p) {
  p->voxel_size = 0.3f;
  p->normal_radius = 0.5f;
  p->fpfh_radius = 0.75f;
  p->tuple_scale = 0.95f;
  p->use_crosscheck = 1;
  p->use_tuple_test = 1;
  p->seed = 0;
}

const char* qtr_last_error(const qtr_handle* h) { return h ? h->err : "null handle"; }
int qtr_num_slots(const qtr_handle* h) { return h ? (int)h->slots.size() : 0; }
void* qtr_slot_stream(qtr_handle* h, int slot) {
  if (!h || slot < 0 || slot >= (int)h->slots.size()) return nullptr;
  return (void*)h->slots[slot].stream;
}

void qtr_destroy(qtr_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  qtr_comm_destroy(h);
  for (auto& l : h->lanes) {
    if (l.stage.h) (void)hipHostFree(l.stage.h);
    if (l.stage.d) (void)hipFree(l.stage.d);
  }
  for (auto& s : h->slots) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.stream2) (void)hipStreamSynchronize(s.stream2);
    for (auto& e : s.ev)
      if (e) (void)hipEventDestroy(e);
    if (s.ev_vox) (void)hipEventDestroy(s.ev_vox);
    for (auto& e : s.fb.ev_nn)
      if (e) (void)hipEventDestroy(e);
    if (s.solver_arena) (void)hipFree(s.solver_arena);
    if (s.front_arena) (void)hipFree(s.front_arena);
    for (int c = 0; c < 2; ++c) {
      if (s.fb.cloud[c].nbr_big_idx) (void)hipFree(s.fb.cloud[c].nbr_big_idx);
      if (s.fb.cloud[c].nbr_big_d2) (void)hipFree(s.fb.cloud[c].nbr_big_d2);
    }
    if (s.in_src) (void)hipFree(s.in_src);
    if (s.in_tgt) (void)hipFree(s.in_tgt);
    if (s.m_src) (void)hipFree(s.m_src);
    if (s.m_tgt) (void)hipFree(s.m_tgt);
    if (s.pinned_i32) (void)hipHostFree(s.pinned_i32);
    if (s.pinned_res) (void)hipHostFree(s.pinned_res);
    if (s.mail) (void)hipHostFree(s.mail);
    if (s.seg_arena) (void)hipFree(s.seg_arena);
    if (s.pw_arena) (void)hipFree(s.pw_arena);
    if (s.ex_arena) (void)hipFree(s.ex_arena);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.stream2) (void)hipStreamDestroy(s.stream2);
  }
  delete h;
}

static int create_impl(qtr_handle* h) {
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  QTR_HIP_TRY(h, solver_init_attributes());
  QTR_HIP_TRY(h, frontend_init_attributes());
  // A slot's second stream carries what must run BESIDE the first one's chain (the matcher's sequential means: 70 - 200 us of
  // one workgroup per cloud).  HIP multiplexes a process's streams onto a few hardware queues, and two streams that land on
  // the same one do not overlap: a handle created after another one (or after a framework's own streams) read the dense step
  // at 2.11 instead of 1.93 ms for that reason alone — its FPFH stage was chain + means, 0.48 ms, not their maximum, 0.27
  // (profiles/r6_ab.txt section 19).  Streams of different PRIORITY come from different queue pools, so the second stream
  // is created at the highest priority: its few small kernels are also the ones that should never wait behind a chain.
  int prio_least = 0, prio_greatest = 0;
  QTR_HIP_TRY(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  for (auto& s : h->slots) {
    QTR_HIP_TRY(h, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    if (prio_greatest != prio_least) QTR_HIP_TRY(h, hipStreamCreateWithPriority(&s.stream2, hipStreamNonBlocking, prio_greatest));
    else QTR_HIP_TRY(h, hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking));
    for (auto& e : s.ev) QTR_HIP_TRY(h, hipEventCreate(&e));
    QTR_HIP_TRY(h, hipEventCreateWithFlags(&s.ev_vox, hipEventDisableTiming));
    for (auto& e : s.fb.ev_nn) QTR_HIP_TRY(h, hipEventCreate(&e));
    const size_t sbytes = solver_scratch_bytes(h->lim.max_corr) + 65536;
    QTR_HIP_TRY(h, hipMalloc(&s.solver_arena, sbytes));
    solver_carve(s.sb, s.solver_arena, h->lim.max_corr);
    const size_t fbytes = frontend_scratch_bytes(h->lim.max_points, h->lim.max_voxels) + 65536;
    QTR_HIP_TRY(h, hipMalloc(&s.front_arena, fbytes));
    {
      hipEvent_t keep[4] = {s.fb.ev_nn[0], s.fb.ev_nn[1], s.fb.ev_nn[2], s.fb.ev_nn[3]};
      frontend_carve(s.fb, s.front_arena, h->lim.max_points, h->lim.max_voxels);
      for (int q = 0; q < 4; ++q) s.fb.ev_nn[q] = keep[q];
      s.fb.nn_events = h->stage_events;
      // (the neighbour grid's cell counters start at zero and are left at zero by every chain that uses them: k2_cell_scan)
      for (int c = 0; c < 2; ++c) QTR_HIP_TRY(h, hipMemset(s.fb.cloud[c].cell_cnt, 0, (size_t)(QTR_CELL_CAP + 4096) * 4));
    }
    // (the long-list arenas — lists of more than QTR_KMAX neighbours: 8 bytes x max_long_neighbors per cloud, 134 MB per
    // slot at the defaults — are allocated the first time a chain with k2_neighbors_big is enqueued: ensure_long_arenas)
    QTR_HIP_TRY(h, hipMalloc((void**)&s.in_src, (size_t)h->lim.max_points * 16));
    QTR_HIP_TRY(h, hipMalloc((void**)&s.in_tgt, (size_t)h->lim.max_points * 16));
    QTR_HIP_TRY(h, hipMalloc((void**)&s.m_src, (size_t)h->lim.max_corr * 16));
    QTR_HIP_TRY(h, hipMalloc((void**)&s.m_tgt, (size_t)h->lim.max_corr * 16));
    QTR_HIP_TRY(h, hipHostMalloc((void**)&s.pinned_i32, 256 * sizeof(int)));
    QTR_HIP_TRY(h, hipHostMalloc((void**)&s.pinned_res, sizeof(qtr_result)));
    QTR_HIP_TRY(h, hipHostMalloc((void**)&s.mail, MAIL_INTS * sizeof(int), hipHostMallocMapped));
    memset(s.mail, 0, MAIL_INTS * sizeof(int));
    {
      void* dv = nullptr;
      QTR_HIP_TRY(h, hipHostGetDevicePointer(&dv, s.mail, 0));
      s.fb.mail = (int*)dv;
      s.sb.mail = (int*)dv;
      s.fb.m_src = s.m_src;
      s.fb.m_tgt = s.m_tgt;
      s.fb.m_cap = h->lim.max_corr;
    }
  }
  // batch lanes: two groups of slots that alternate (one's kernels cover the other's host read-back)
  const int S = (int)h->slots.size();
  int NL = S >= 2 ? 2 : 1;
  if (const char* e = QTR_ENGINE_ENV("QTR_BATCH_LANES")) {  // experiment knob: more, smaller groups in flight
    const int v = atoi(e);
    if (v >= 1 && v <= 8 && v <= S) NL = v;
  }
  h->lanes.resize((size_t)NL);
  for (int l = 0; l < NL; ++l) {
    Lane& ln = h->lanes[l];
    ln.cap = min(S / NL, 64);  // a launch chain serves at most 64 pairs (match.hip NN_MAXG)
    ln.first_slot = l * (S / NL);
    ln.stage.cap = (size_t)64 * 1024 + (size_t)ln.cap * 8192;
    QTR_HIP_TRY(h, hipHostMalloc((void**)&ln.stage.h, ln.stage.cap));
    QTR_HIP_TRY(h, hipMalloc((void**)&ln.stage.d, ln.stage.cap));
  }
  // experiment knob (test build only): the lanes' launch streams on DISJOINT sets of compute
  // units (hipExtStreamCreateWithCUMask; on this part bit k of the mask is compute unit k / 8 of XCD k % 8).
  //   QTR_LANE_CU_MASK=halves   lane l of NL takes the l-th NL-th of every XCD's units
  //   QTR_LANE_CU_MASK=xcd      lane l takes XCDs [8 l / NL, 8 (l + 1) / NL)
  // measured and NOT adopted: profiles/r6_ab.txt
  if (const char* e = QTR_ENGINE_ENV("QTR_LANE_CU_MASK")) {
    for (int l = 0; l < NL && NL > 1; ++l) {
      unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < 256; ++k) {
        const int xcd = k % 8, cu = k / 8;
        const bool mine = (strcmp(e, "xcd") == 0) ? (xcd * NL / 8 == l) : (cu * NL / 32 == l);
        if (mine) mask[k >> 5] |= 1u << (k & 31);
      }
      Slot& lead = h->slots[h->lanes[l].first_slot];
      (void)hipStreamDestroy(lead.stream);
      (void)hipStreamDestroy(lead.stream2);
      QTR_HIP_TRY(h, hipExtStreamCreateWithCUMask(&lead.stream, 8, mask));
      QTR_HIP_TRY(h, hipExtStreamCreateWithCUMask(&lead.stream2, 8, mask));
    }
  }
  return QTR_OK;
}

int qtr_create(int device, const qtr_limits* limits, qtr_handle** out) {
  if (!out) return QTR_ERR_BAD_ARG;
  *out = nullptr;
  qtr_handle* h = new (std::nothrow) qtr_handle();
  if (!h) return QTR_ERR_CAPACITY;
  h->err[0] = 0;
  h->device = device;
  h->uid = g_handle_uid.fetch_add(1, std::memory_order_relaxed) + 1;
  {
    const char* hw = getenv("QTR_HOST_WAIT");
    h->spin_wait = (hw && strcmp(hw, "block") == 0) ? 0 : 1;
    const char* se = getenv("QTR_STAGE_EVENTS");
    h->stage_events = (se && atoi(se) == 0) ? 0 : 1;
    h->nn_event_stride = h->stage_events ? 1 : 0;
  }
  if (limits)
    h->lim = *limits;
  else
    qtr_default_limits(&h->lim);
  if (h->lim.max_points < 64 || h->lim.max_voxels < 64 || h->lim.max_corr < 64 || h->lim.n_slots < 1 ||
      h->lim.max_corr > 32768 || h->lim.max_voxels > h->lim.max_points || h->lim.max_long_neighbors < 0 ||
      h->lim.max_voxels > QTR_NN_MAX_ROWS) {  // (k_recheck_filter's lists pack a base row into 20 bits: match.hip)
    delete h;
    return QTR_ERR_BAD_ARG;
  }
  if (h->lim.max_long_neighbors == 0)
    h->lim.max_long_neighbors = (int)std::min<long long>(128LL * h->lim.max_voxels, 1LL << 30);
  h->lim.max_long_neighbors = (h->lim.max_long_neighbors + 63) & ~63;
  h->slots.resize((size_t)h->lim.n_slots);
  const int rc = create_impl(h);
  *out = h;  // returned even on failure so that qtr_last_error can be read; caller destroys it
  return rc;
}

// adds the nearest-neighbour kernel times of the last match (events on its launch stream) to the slot's totals; called
// before the events are recorded again and by qtr_get_nn_totals
static void flush_nn_totals(Slot& s) {
  if (!s.nn_pending) return;
  s.nn_pending = 0;
  float a = 0, b = 0;
  if (hipEventElapsedTime(&a, s.fb.ev_nn[0], s.fb.ev_nn[1]) == hipSuccess &&
      hipEventElapsedTime(&b, s.fb.ev_nn[2], s.fb.ev_nn[3]) == hipSuccess) {
    s.nn_total_ms += (double)a + (double)b;
    s.nn_total_launches += 2;
  }
  (void)hipGetLastError();
}

static void fill_nn_times(Slot& s) {
  float a = 0, b = 0;
  s.nn_dir_ms[0] = s.nn_dir_ms[1] = 0.f;
  if (!s.nn_timed_last) return;  // (the fields keep their zeros)
  if (hipEventElapsedTime(&a, s.fb.ev_nn[0], s.fb.ev_nn[1]) == hipSuccess &&
      hipEventElapsedTime(&b, s.fb.ev_nn[2], s.fb.ev_nn[3]) == hipSuccess) {
    s.times.nn_kernel = a + b;
    s.times.nn_launches = 2;
    s.nn_dir_ms[0] = a;
    s.nn_dir_ms[1] = b;
  }
}

// Waits until the phase-ending kernel has published `seq` in mailbox word `idx` (frontend.h MAIL_SEQ_*).  The
// default is to WATCH the pinned word — the store crosses PCIe in ~1-2 us, the runtime's stream wait costs
// 20-30 us per phase boundary (three per registration) — and to fall back to the runtime every 64k polls so
// that a failed launch or a lost device ends the wait.  QTR_HOST_WAIT=block uses hipStreamSynchronize only.
// the payload a sequence word announces is consumed only once its tag matches (common.h, mail_store_line): the
// sequence word can reach host memory before the counters
static bool mail_payload_ok(const Slot& s, int idx, int seq) {
  auto line_ok = [&](int base) {
    const volatile int* l = s.mail + base;
    int x = 0;
    for (int i = 0; i < 15; ++i) x ^= l[i];
    return l[15] == (seq ^ x ^ MAIL_TAG_SALT);
  };
  auto solver_ok = [&]() {
    const volatile int* m = s.mail + MAIL_SOLVER;
    int x = 0, y = 0;
    for (int i = 0; i < (int)(sizeof(qtr_result) / 4); ++i) x ^= m[i];
    for (int i = 0; i < (int)(sizeof(SolverState) / 4); ++i) y ^= m[64 + i];
    return m[63] == (seq ^ x ^ MAIL_TAG_SALT) && m[96] == (seq ^ y ^ MAIL_TAG_SALT);
  };
  switch (idx) {
    case MAIL_SEQ_VOX0: return line_ok(MAIL_VOX0);
    case MAIL_SEQ_VOX1: return line_ok(MAIL_VOX1);
    case MAIL_SEQ_MATCH: return line_ok(MAIL_MATCH) && line_ok(MAIL_CNT0) && line_ok(MAIL_CNT1);
    case MAIL_SEQ_SOLVE: return solver_ok();
    default: return true;
  }
}
// non-blocking: has the phase-ending kernel published `seq` in word `idx`, payload complete?
static bool mail_ready(const Slot& s, int idx, int seq) {
  if (__atomic_load_n((volatile int*)(s.mail + idx), __ATOMIC_ACQUIRE) != seq) return false;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return mail_payload_ok(s, idx, seq);
}

static int wait_mail(qtr_handle* h, Slot& s, int idx, int seq) {
  volatile int* p = s.mail + idx;
  if (!h->spin_wait) {
    QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  } else {
    for (unsigned long spins = 1;; ++spins) {
      if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) break;  // now make sure the payload it announces is complete
      if ((spins & 0xffff) == 0) {
        const hipError_t q = hipStreamQuery(s.stream);
        if (q == hipSuccess) break;  // stream drained: the word must be there now (checked below)
        if (q != hipErrorNotReady) QTR_HIP_TRY(h, q);
        (void)hipGetLastError();  // hipErrorNotReady is not an error of ours
      }
      __builtin_ia32_pause();
    }
  }
  if (__atomic_load_n(p, __ATOMIC_ACQUIRE) != seq) {
    snprintf(h->err, sizeof(h->err), "mailbox word %d holds %d, expected %d (phase kernel did not run)", idx, (int)*p, seq);
    return QTR_ERR_HIP;
  }
  auto payload_ok = [&]() { return mail_payload_ok(s, idx, seq); };
  bool drained = false;
  for (unsigned long spins = 1;; ++spins) {
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (payload_ok()) return QTR_OK;
    if ((spins & 0x3fff) == 0) {
      if (drained) break;  // the stream was idle one round ago and the payload still does not add up
      const hipError_t q = hipStreamSynchronize(s.stream);
      if (q != hipSuccess) QTR_HIP_TRY(h, q);
      drained = true;
    }
    __builtin_ia32_pause();
  }
  snprintf(h->err, sizeof(h->err), "mailbox payload of word %d does not match its tag (sequence %d)", idx, seq);
  return QTR_ERR_HIP;
}

static void compute_times(Slot& s) {
  if (!s.times_pending) return;
  float ms = 0;
  if (s.times_pending == 4) {  // the call ran with the stage events off: nothing was recorded
    s.times = qtr_stage_times{};
    fill_nn_times(s);
    (void)hipGetLastError();
    s.times_pending = 0;
    return;
  }
  if (s.times_pending == 3) {  // qtr_feature_pair: the front end alone
    (void)hipEventSynchronize(s.ev[7]);
    s.times = qtr_stage_times{};
    if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) s.times.voxelize = ms;
    if (hipEventElapsedTime(&ms, s.ev[1], s.ev[6]) == hipSuccess) s.times.fpfh = ms;
    if (hipEventElapsedTime(&ms, s.ev[6], s.ev[7]) == hipSuccess) s.times.match = ms;
    if (hipEventElapsedTime(&ms, s.ev[0], s.ev[7]) == hipSuccess) s.times.total = ms;
    fill_nn_times(s);
    (void)hipGetLastError();
    s.times_pending = 0;
    return;
  }
  (void)hipEventSynchronize(s.ev[4]);
  s.times = qtr_stage_times{};
  if (s.times_pending == 1) {
    if (hipEventElapsedTime(&ms, s.ev[1], s.ev[2]) == hipSuccess) s.times.graph = ms;
  } else {
    if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) s.times.voxelize = ms;
    if (hipEventElapsedTime(&ms, s.ev[1], s.ev[6]) == hipSuccess) s.times.fpfh = ms;
    if (hipEventElapsedTime(&ms, s.ev[6], s.ev[7]) == hipSuccess) s.times.match = ms;
    if (hipEventElapsedTime(&ms, s.ev[7], s.ev[2]) == hipSuccess) s.times.graph = ms;
    fill_nn_times(s);
  }
  if (hipEventElapsedTime(&ms, s.ev[2], s.ev[3]) == hipSuccess) s.times.clique = ms;
  if (hipEventElapsedTime(&ms, s.ev[3], s.ev[4]) == hipSuccess) s.times.solve = ms;
  if (hipEventElapsedTime(&ms, s.ev[0], s.ev[4]) == hipSuccess) s.times.total = ms;
  (void)hipGetLastError();
  s.times_pending = 0;
}

// The long-list arenas of a slot's two clouds, allocated on first need: voxel-grid centroids at the demo's leaf never have
// more than QTR_KMAX neighbours, so most handles never pay for them.
static int ensure_long_arenas(qtr_handle* h, Slot& s) {
  for (int c = 0; c < 2; ++c) {
    CloudBufs& cb = s.fb.cloud[c];
    // (each pointer on its own: a failed second allocation must not make the next call allocate the first one again)
    if (!cb.nbr_big_idx) QTR_HIP_TRY(h, hipMalloc((void**)&cb.nbr_big_idx, (size_t)h->lim.max_long_neighbors * 4));
    if (!cb.nbr_big_d2) QTR_HIP_TRY(h, hipMalloc((void**)&cb.nbr_big_d2, (size_t)h->lim.max_long_neighbors * 4));
    cb.nbr_big_cap = h->lim.max_long_neighbors;
  }
  return QTR_OK;
}

static Slot* get_slot(qtr_handle* h, int slot) {
  if (!h) return nullptr;
  if (slot < 0 || slot >= (int)h->slots.size()) {
    snprintf(h->err, sizeof(h->err), "slot %d out of range", slot);
    return nullptr;
  }
  return &h->slots[slot];
}

static int check_params(qtr_handle* h, const qtr_params* prm) {
  if (!prm) {
    snprintf(h->err, sizeof(h->err), "params is NULL");
    return QTR_ERR_BAD_ARG;
  }
  if (prm->inlier_selection_mode == QTR_INLIER_NONE) {
    snprintf(h->err, sizeof(h->err), "inlier_selection_mode NONE not supported (undefined behaviour in the reference)");
    return QTR_ERR_UNSUPPORTED;
  }
  if (prm->reg_mode != QTR_REG_QUATRO && prm->reg_mode != QTR_REG_TEASER) {
    snprintf(h->err, sizeof(h->err), "[solveForRotation] The param is wrong! It should be 'TEASER' or 'Quatro'");
    return QTR_ERR_BAD_ARG;
  }
  if (prm->reg_mode == QTR_REG_TEASER && prm->using_pre_estimated_ryrx) {  // reference include/quatro.hpp:424-426
    snprintf(h->err, sizeof(h->err), "Wrong reg type name is coming!");
    return QTR_ERR_BAD_ARG;
  }
  if (prm->inlier_selection_mode < 0 || prm->inlier_selection_mode > 3 || !(prm->noise_bound > 0) ||
      !(prm->rotation_gnc_factor > 1) || prm->rotation_max_iterations < 1) {
    snprintf(h->err, sizeof(h->err), "invalid solver parameter");
    return QTR_ERR_BAD_ARG;
  }
  return QTR_OK;
}

// PMC_EXACT after the heuristic has finished (state `hs` = the device SolverState as the host last saw it): proves the
// heuristic clique maximum or replaces it (exact.hip).  *improved tells the caller that st->mc / best_r / picks changed.
// time_limit: Params::max_clique_time_limit of THIS call, seconds (reference include/quatro.hpp:267,800; <= 0: none).
static int exact_phase(qtr_handle* h, Slot& s, int L, const SolverState& hs, bool* improved, double time_limit) {
  *improved = false;
  s.exact_nodes = 0;
  s.exact_aborted = 0;
  if (L <= 0 || hs.mc >= hs.ub || hs.mc < 1) return QTR_OK;  // lb == ub: reference src/graph.cc:100-102
  const int W = (L + 63) / 64;
  if (exact_nw(W) == 0) {
    snprintf(h->err, sizeof(h->err), "PMC_EXACT supports at most 32768 vertices (L=%d)", L);
    return QTR_ERR_CAPACITY;
  }
  const int depth_cap = hs.ub + 1;
  const size_t small_lds = exact_small_lds(L, W, hs.t0, depth_cap);
  int nwaves = small_lds ? 1024 : 2048;
  while (nwaves > 64 && exact_scratch_bytes(W, depth_cap, nwaves) > ((size_t)768 << 20)) nwaves >>= 1;
  if (!small_lds && nwaves > L) nwaves = L < 1 ? 1 : L;
  const size_t need = exact_scratch_bytes(W, depth_cap, nwaves);
  if (need > s.ex_bytes) {
    if (s.ex_arena) (void)hipFree(s.ex_arena);
    s.ex_arena = nullptr;
    s.ex_bytes = 0;
    QTR_HIP_TRY(h, hipMalloc(&s.ex_arena, need));
    s.ex_bytes = need;
  }
  exact_carve(s.exb, s.ex_arena, depth_cap, nwaves);
  const long long ticks = time_limit > 0 ? (long long)(time_limit * 1e8) : 0;  // 100 MHz counter
  hipLaunchKernelGGL(k_exact_init, dim3(1), dim3(1), 0, s.stream, s.sb.Kp, L, s.sb.st, s.exb.ctl);
  exact_launch_search(s.sb, s.exb, L, depth_cap, nwaves, 0, ticks, s.stream, hs.t0, small_lds);
  ExactCtl* hc = (ExactCtl*)(s.pinned_i32 + 192);
  QTR_HIP_TRY(h, hipMemcpyAsync(hc, s.exb.ctl, sizeof(ExactCtl), hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  s.exact_nodes = hc->nodes;
  if (hc->abort) {  // time limit: the heuristic clique stands (the reference returns PMC's best so far)
    s.exact_aborted = 1;
    return QTR_OK;
  }
  if (hc->gbest <= hs.mc) return QTR_OK;
  QTR_HIP_TRY(h, hipMemsetAsync(s.exb.cliq, 0xff, (size_t)nwaves * (depth_cap + 2) * sizeof(int), s.stream));
  hipLaunchKernelGGL(k_exact_phase_b, dim3(1), dim3(1), 0, s.stream, s.sb.Kp, L, s.exb.ctl);
  exact_launch_search(s.sb, s.exb, L, depth_cap, nwaves, 1, ticks, s.stream, hs.t0, small_lds);
  hipLaunchKernelGGL(k_exact_commit, dim3(1), dim3(256), 0, s.stream, s.exb.ctl, s.exb.cliq, nwaves, depth_cap, s.sb.st,
                     s.sb.picks);
  QTR_HIP_TRY(h, hipMemcpyAsync(hc, s.exb.ctl, sizeof(ExactCtl), hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  s.exact_nodes += hc->nodes;
  if (hc->abort || hc->winner < 0) {
    s.exact_aborted = 1;
    return QTR_OK;
  }
  *improved = true;
  return QTR_OK;
}

// the cells of the larger of a pair's two neighbour-search grids (CNT_NCELL words of the voxel stage's mail) when BOTH fit the
// dense cell table of the FPFH chain, else 0 = sort packed cell keys (frontend.hip, d_cell_count)
static int cell_table_cells(int ncell_s, int ncell_t) {
  static const bool off = QTR_ENGINE_ENV("QTR_CELL_TABLE") != nullptr && atoi(QTR_ENGINE_ENV("QTR_CELL_TABLE")) == 0;
  if (off || ncell_s <= 0 || ncell_t <= 0 || ncell_s > QTR_CELL_CAP || ncell_t > QTR_CELL_CAP) return 0;
  return std::max(ncell_s, ncell_t);
}

// k_hcore_async's workgroups of one chain have to be resident together, so chains that may overlap must not ask for more
// than the compute units between them.  Who may overlap is decided by WHO CALLS, not by what happens to be in flight at
// the moment of the call (round 4 looked at the in-flight count: the first of several threads took the whole device and
// the others' launches were only partly resident until a residency timeout sent them to the peeling fallback): every
// host thread registers with the handle on its first back-end call, and from the moment a second thread has registered
// every chain takes 1 / min(LIVE threads, slots) of the units: a thread leaves the count when it ends (a pool of short-lived
// threads that run one at a time keeps the whole device), and whenever the share CHANGES — a third thread arriving, one
// leaving — the chains enqueued under the old share are waited for once (1/2 + 1/2 + 1/3 would oversubscribe as surely as
// 1 + 1/2).  A single-threaded caller never pays anything.  (QTR_DBG_CORE and the floor statistics
// st[22] / st[29] depend on timing either way: the floor is decided by what has been published 200 us into the launch.)
// (the handles a thread has registered with; its destructor — the thread's end — takes the thread off their counts)
struct ThreadCallerList {
  struct Entry {
    unsigned long long uid;
    std::shared_ptr<qtr_handle::Callers> c;
  };
  std::vector<Entry> e;
  ~ThreadCallerList() {
    for (Entry& x : e) x.c->live.fetch_sub(1, std::memory_order_acq_rel);
  }
  void enter(qtr_handle* h) {
    for (const Entry& x : e)
      if (x.uid == h->uid) return;  // (a thread that works with several handles in turn is ONE caller of each, once)
    if (e.size() >= 64) {  // handles come and go: keep the list short — drop the entries of destroyed handles first; if
      size_t victim = 0;   // every one is alive the oldest goes AND leaves its handle's count (it re-registers if it returns)
      for (size_t i = 0; i < e.size(); ++i)
        if (e[i].c.use_count() == 1) {
          victim = i;
          break;
        }
      e[victim].c->live.fetch_sub(1, std::memory_order_acq_rel);
      e.erase(e.begin() + (long)victim);
    }
    h->callers->live.fetch_add(1, std::memory_order_acq_rel);
    e.push_back(Entry{h->uid, h->callers});
  }
};

struct InFlight {
  qtr_handle* h;
  int gen;
  explicit InFlight(qtr_handle* h_) : h(h_) {
    static thread_local ThreadCallerList t_callers;
    t_callers.enter(h);
    const int want = std::max(1, std::min(h->callers->live.load(std::memory_order_acquire), (int)h->slots.size()));
    bool changed = false;
    {
      std::lock_guard<std::mutex> lk(h->share_mu);
      if (want != h->share_val) {  // a thread came or went: chains enqueued under the old share drain before this one starts
        h->share_val = want;
        ++h->share_gen;
        changed = true;
      }
      gen = h->share_gen;
      h->gen_inflight[gen & 3].fetch_add(1, std::memory_order_acq_rel);
    }
    h->solves_in_flight.fetch_add(1, std::memory_order_acq_rel);
    if (changed)  // (bounded: a chain another thread never reads back must not hang this one; three generations back is every
      for (int back = 1; back <= 3; ++back)  // chain that can still be in flight under an older share)
        for (int spins = 0; h->gen_inflight[(gen - back) & 3].load(std::memory_order_acquire) > 0 && spins < 2000000; ++spins)
          std::this_thread::yield();
    solver_set_hca_share(want);
  }
  ~InFlight() {
    h->gen_inflight[gen & 3].fetch_sub(1, std::memory_order_acq_rel);
    h->solves_in_flight.fetch_sub(1, std::memory_order_acq_rel);
  }
};

// Runs the back end on device-resident matched clouds and brings the result record to the host.
static int solve_device(qtr_handle* h, Slot& s, const float4* d_src, const float4* d_tgt, int L, const qtr_params* prm,
                        qtr_result* res, bool reset_done = false) {
  s.last_L = L;
  s.last_Wb = (((L + 63) / 64) + 3) & ~3;
  s.sb.mail_seq = ++s.seq;
  // k_hcore_async's workgroups have to be resident together: a chain that starts while another slot's is in flight (calls
  // from several threads) takes its share of the compute units only (solver.hip, solver_set_hca_share)
  InFlight in_flight(h);
  QTR_HIP_TRY(h, solver_enqueue(s.sb, d_src, d_tgt, L, *prm, s.stream, s.pinned_i32, h->stage_events ? s.ev[2] : nullptr,
                                h->stage_events ? s.ev[3] : nullptr, reset_done));
  if (h->stage_events) QTR_HIP_TRY(h, hipEventRecord(s.ev[4], s.stream));
  QTR_TRY(wait_mail(h, s, MAIL_SEQ_SOLVE, s.seq));  // k_finalize left the record and the state in the mailbox
  if (L > 0 && !((const SolverState*)(s.mail + MAIL_SOLVER + 64))->done) {  // rare: more clique rounds needed
    s.sb.mail_seq = ++s.seq;
    QTR_HIP_TRY(h, solver_continue(s.sb, d_src, d_tgt, L, *prm, s.stream, s.pinned_i32 + 128,
                                   ((const SolverState*)(s.mail + MAIL_SOLVER + 64))->redo_cores));
    if (h->stage_events) QTR_HIP_TRY(h, hipEventRecord(s.ev[4], s.stream));
    QTR_TRY(wait_mail(h, s, MAIL_SEQ_SOLVE, s.seq));
  }
  if (L > 0 && prm->inlier_selection_mode == QTR_INLIER_PMC_EXACT) {
    SolverState hs;
    memcpy(&hs, s.mail + MAIL_SOLVER + 64, sizeof(hs));
    bool improved = false;
    QTR_TRY(exact_phase(h, s, L, hs, &improved, prm->max_clique_time_limit));
    if (improved) {  // estimate again from the larger clique
      s.sb.mail_seq = ++s.seq;
      QTR_HIP_TRY(h, solver_refinalize(s.sb, d_src, d_tgt, L, *prm, s.stream));
      if (h->stage_events) QTR_HIP_TRY(h, hipEventRecord(s.ev[4], s.stream));
      QTR_TRY(wait_mail(h, s, MAIL_SEQ_SOLVE, s.seq));
    }
  }
  memcpy(s.pinned_res, s.mail + MAIL_SOLVER, sizeof(qtr_result));
  const int keep_ns = res->n_src, keep_nt = res->n_tgt, keep_nc = res->n_corr;
  *res = *s.pinned_res;
  res->n_src = keep_ns;
  res->n_tgt = keep_nt;
  res->n_corr = keep_nc;
  return res->status;
}

static int copy_out_lists(qtr_handle* h, Slot& s, const qtr_result* res, int* clique, int* rot_inliers,
                          int* final_inliers, int cap, int mem) {
  const hipMemcpyKind kind = (mem == QTR_MEM_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (clique && res->n_clique > 0) {
    if (res->n_clique > cap) return QTR_ERR_CAPACITY;
    QTR_HIP_TRY(h, hipMemcpyAsync(clique, s.sb.clique, sizeof(int) * (size_t)res->n_clique, kind, s.stream));
  }
  if (rot_inliers && res->n_rot_inliers > 0) {
    if (res->n_rot_inliers > cap) return QTR_ERR_CAPACITY;
    QTR_HIP_TRY(h, hipMemcpyAsync(rot_inliers, s.sb.rot_inl, sizeof(int) * (size_t)res->n_rot_inliers, kind, s.stream));
  }
  if (final_inliers && res->n_final > 0) {
    if (res->n_final > cap) return QTR_ERR_CAPACITY;
    QTR_HIP_TRY(h, hipMemcpyAsync(final_inliers, s.sb.final_inl, sizeof(int) * (size_t)res->n_final, kind, s.stream));
  }
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  return QTR_OK;
}

int qtr_solve(qtr_handle* h, int slot, const float* src4, const float* tgt4, int L, const qtr_params* prm,
              qtr_result* res, int* clique, int* rot_inliers, int* final_inliers, int cap, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !res) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  memset(res, 0, sizeof(*res));
  int rc = check_params(h, prm);
  if (rc != QTR_OK) return res->status = rc;
  if (L < 0 || (L > 0 && (!src4 || !tgt4))) {
    snprintf(h->err, sizeof(h->err), "bad input clouds");
    return res->status = QTR_ERR_BAD_ARG;
  }
  if (L > h->lim.max_corr) {
    snprintf(h->err, sizeof(h->err), "L=%d exceeds max_corr=%d", L, h->lim.max_corr);
    return res->status = QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  const float4 *d_src = (const float4*)src4, *d_tgt = (const float4*)tgt4;
  // (an event record is a marker packet the queue has to retire before the next launch starts: with the stage events
  // off a call records none at all, and every field of qtr_stage_times reads 0)
  const bool ev_on = h->stage_events != 0;
  if (ev_on) QTR_HIP_TRY(h, hipEventRecord(s.ev[0], s.stream));
  if (mem == QTR_MEM_HOST && L > 0) {
    QTR_HIP_TRY(h, hipMemcpyAsync(s.m_src, src4, (size_t)L * 16, hipMemcpyHostToDevice, s.stream));
    QTR_HIP_TRY(h, hipMemcpyAsync(s.m_tgt, tgt4, (size_t)L * 16, hipMemcpyHostToDevice, s.stream));
    d_src = s.m_src;
    d_tgt = s.m_tgt;
  }
  if (ev_on) QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
  res->n_corr = L;
  rc = solve_device(h, s, d_src, d_tgt, L, prm, res);
  if (rc != QTR_OK && rc != QTR_ERR_CLIQUE_TOO_SMALL) return rc;
  s.times_pending = ev_on ? 1 : 4;
  const int rc2 = copy_out_lists(h, s, res, clique, rot_inliers, final_inliers, cap, mem);
  if (rc2 != QTR_OK) return res->status = rc2;
  return rc;
}

int qtr_max_clique(qtr_handle* h, int slot, const unsigned long long* adj, int L, int mode, double kcore_thr,
                   double time_limit, int* clique, int cap, int* n_out, int* max_core_out, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !n_out || L < 0 || (L > 0 && (!adj || !clique))) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  *n_out = 0;
  if (max_core_out) *max_core_out = 0;
  if (mode != QTR_INLIER_PMC_HEU && mode != QTR_INLIER_KCORE_HEU && mode != QTR_INLIER_PMC_EXACT) {
    snprintf(h->err, sizeof(h->err), "clique solver mode %d not supported", mode);
    return QTR_ERR_UNSUPPORTED;
  }
  if (L > h->lim.max_corr) {
    snprintf(h->err, sizeof(h->err), "L=%d exceeds max_corr=%d", L, h->lim.max_corr);
    return QTR_ERR_CAPACITY;
  }
  if (L == 0) return QTR_OK;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  const int W = (L + 63) / 64;
  const u64* d_adj = (const u64*)adj;
  if (mem == QTR_MEM_HOST) {
    QTR_HIP_TRY(h, hipMemcpyAsync(s.sb.bm, adj, (size_t)L * W * 8, hipMemcpyHostToDevice, s.stream));
    d_adj = s.sb.bm;
  }
  s.last_L = L;
  s.last_Wb = (L + 63) / 64;  // the caller's packed layout
  InFlight in_flight(h);
  QTR_HIP_TRY(h, clique_only_enqueue(s.sb, d_adj, L, mode, kcore_thr, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.pinned_i32 + 128, s.sb.st, sizeof(SolverState), hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  if (!((const SolverState*)(s.pinned_i32 + 128))->done) {
    qtr_params dummy;
    qtr_default_params(&dummy);
    QTR_HIP_TRY(h, solver_continue(s.sb, nullptr, nullptr, L, dummy, s.stream, s.pinned_i32 + 128,
                                   ((const SolverState*)(s.pinned_i32 + 128))->redo_cores));
  }
  if (mode == QTR_INLIER_PMC_EXACT) {
    SolverState hs;
    memcpy(&hs, s.pinned_i32 + 128, sizeof(hs));
    bool improved = false;
    QTR_TRY(exact_phase(h, s, L, hs, &improved, time_limit));
  }
  s.sb.mail_seq = ++s.seq;
  QTR_HIP_TRY(h, clique_only_finish(s.sb, L, s.stream));
  QTR_TRY(wait_mail(h, s, MAIL_SEQ_SOLVE, s.seq));
  memcpy(s.pinned_res, s.mail + MAIL_SOLVER, sizeof(qtr_result));
  const int M = s.pinned_res->n_clique;
  if (max_core_out) *max_core_out = s.pinned_res->max_core;
  *n_out = M;
  if (M > cap) return QTR_ERR_CAPACITY;
  if (M > 0) {
    const hipMemcpyKind kind = (mem == QTR_MEM_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    QTR_HIP_TRY(h, hipMemcpyAsync(clique, s.sb.clique, sizeof(int) * (size_t)M, kind, s.stream));
    QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  }
  return QTR_OK;
}

// ------------------------------------------------------------------------------------------------
// individually callable stages (stages.hip).  All staging goes through the slot's solver arena.
static bool stage_fits(qtr_handle* h, Slot& s, size_t doubles, size_t ints) {
  if (doubles <= (size_t)72 * s.sb.Lcap && ints <= (size_t)24 * s.sb.Lcap) return true;
  snprintf(h->err, sizeof(h->err), "stage call needs %zu doubles / %zu ints of scratch; raise max_corr (now %d)",
           doubles, ints, s.sb.Lcap);
  return false;
}

int qtr_compute_tims(qtr_handle* h, int slot, const double* v3n, int N, double* tims3k, int* map2k) {
  Slot* sp = get_slot(h, slot);
  if (!sp || N < 0 || (N > 1 && (!v3n || !tims3k || !map2k))) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  const long long K = (long long)N * (N - 1) / 2;
  if (K <= 0) return QTR_OK;
  if (!stage_fits(h, s, (size_t)3 * N + (size_t)3 * K, (size_t)2 * K)) return QTR_ERR_CAPACITY;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  double* d_v = s.sb.f64;
  double* d_t = d_v + (size_t)3 * N;
  QTR_HIP_TRY(h, hipMemcpyAsync(d_v, v3n, sizeof(double) * 3 * (size_t)N, hipMemcpyHostToDevice, s.stream));
  hipLaunchKernelGGL(k_compute_tims, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s.stream, d_v, N, K, d_t, s.sb.i32);
  QTR_HIP_TRY(h, hipGetLastError());
  QTR_HIP_TRY(h, hipMemcpyAsync(tims3k, d_t, sizeof(double) * 3 * (size_t)K, hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(map2k, s.sb.i32, sizeof(int) * 2 * (size_t)K, hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  return QTR_OK;
}

int qtr_scale_mask(qtr_handle* h, int slot, const double* a, const double* b, long long K, double noise_bound,
                   double cbar2, unsigned char* mask) {
  Slot* sp = get_slot(h, slot);
  if (!sp || K < 0 || (K > 0 && (!a || !b || !mask)) || !(noise_bound > 0)) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  if (K == 0) return QTR_OK;
  if (!stage_fits(h, s, (size_t)6 * K, (size_t)(K + 3) / 4)) return QTR_ERR_CAPACITY;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  double* d_a = s.sb.f64;
  double* d_b = d_a + (size_t)3 * K;
  unsigned char* d_m = (unsigned char*)s.sb.i32;
  QTR_HIP_TRY(h, hipMemcpyAsync(d_a, a, sizeof(double) * 3 * (size_t)K, hipMemcpyHostToDevice, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(d_b, b, sizeof(double) * 3 * (size_t)K, hipMemcpyHostToDevice, s.stream));
  const double beta = 2 * noise_bound * sqrt(cbar2);
  hipLaunchKernelGGL(k_scale_mask, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s.stream, d_a, d_b, K, beta, d_m);
  QTR_HIP_TRY(h, hipGetLastError());
  QTR_HIP_TRY(h, hipMemcpyAsync(mask, d_m, (size_t)K, hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  return QTR_OK;
}

int qtr_gnc_rotation2d(qtr_handle* h, int slot, const double* src2m, const double* dst2m, int M, double noise_bound,
                       double gnc_factor, int max_iterations, double cost_threshold, double* R4, double* cost,
                       int* iterations, unsigned char* inliers) {
  Slot* sp = get_slot(h, slot);
  if (!sp || M < 1 || !src2m || !dst2m || !R4 || !(gnc_factor > 1) || max_iterations < 1) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  if (!stage_fits(h, s, (size_t)5 * M + 8, (size_t)(M + 3) / 4)) return QTR_ERR_CAPACITY;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  double* d_s = s.sb.f64;
  double* d_d = d_s + (size_t)2 * M;
  double* d_w = d_d + (size_t)2 * M;
  double* d_o = d_w + (size_t)M;
  unsigned char* d_i = (unsigned char*)s.sb.i32;
  QTR_HIP_TRY(h, hipMemcpyAsync(d_s, src2m, sizeof(double) * 2 * (size_t)M, hipMemcpyHostToDevice, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(d_d, dst2m, sizeof(double) * 2 * (size_t)M, hipMemcpyHostToDevice, s.stream));
  hipLaunchKernelGGL(k_gnc_only, dim3(1), dim3(64), 0, s.stream, d_s, d_d, M, noise_bound, gnc_factor, max_iterations,
                     cost_threshold, d_w, d_o, d_i);
  QTR_HIP_TRY(h, hipGetLastError());
  double out[6];
  QTR_HIP_TRY(h, hipMemcpyAsync(out, d_o, sizeof(out), hipMemcpyDeviceToHost, s.stream));
  if (inliers) QTR_HIP_TRY(h, hipMemcpyAsync(inliers, d_i, (size_t)M, hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  for (int i = 0; i < 4; ++i) R4[i] = out[i];
  if (cost) *cost = out[4];
  if (iterations) *iterations = (int)out[5];
  return QTR_OK;
}

int qtr_gnc_rotation3d(qtr_handle* h, int slot, const double* src3m, const double* dst3m, int M, double noise_bound,
                       double gnc_factor, int max_iterations, double cost_threshold, double* R9, double* cost,
                       int* iterations, unsigned char* inliers) {
  Slot* sp = get_slot(h, slot);
  if (!sp || M < 1 || !src3m || !dst3m || !R9 || !(gnc_factor > 1) || max_iterations < 1) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  if (!stage_fits(h, s, (size_t)7 * M + 16, (size_t)(M + 3) / 4)) return QTR_ERR_CAPACITY;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  double* d_s = s.sb.f64;
  double* d_d = d_s + (size_t)3 * M;
  double* d_w = d_d + (size_t)3 * M;
  double* d_o = d_w + (size_t)M;
  unsigned char* d_i = (unsigned char*)s.sb.i32;
  QTR_HIP_TRY(h, hipMemcpyAsync(d_s, src3m, sizeof(double) * 3 * (size_t)M, hipMemcpyHostToDevice, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(d_d, dst3m, sizeof(double) * 3 * (size_t)M, hipMemcpyHostToDevice, s.stream));
  hipLaunchKernelGGL(k_gnc3d_only, dim3(1), dim3(64), 0, s.stream, d_s, d_d, M, noise_bound, gnc_factor, max_iterations,
                     cost_threshold, d_w, d_o, d_i);
  QTR_HIP_TRY(h, hipGetLastError());
  double out[11];
  QTR_HIP_TRY(h, hipMemcpyAsync(out, d_o, sizeof(out), hipMemcpyDeviceToHost, s.stream));
  if (inliers) QTR_HIP_TRY(h, hipMemcpyAsync(inliers, d_i, (size_t)M, hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  for (int i = 0; i < 9; ++i) R9[i] = out[i];
  if (cost) *cost = out[9];
  if (iterations) *iterations = (int)out[10];
  return QTR_OK;
}

static int cote_impl(qtr_handle* h, int slot, const double* X, const double* ranges, int N, double range,
                     int median_selection, double* estimate, unsigned char* inliers, int* n_card) {
  Slot* sp = get_slot(h, slot);
  if (!sp || N < 1 || !X) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  if (!stage_fits(h, s, (size_t)16 * N + 8, (size_t)2 * N + (size_t)(N + 3) / 4)) return QTR_ERR_CAPACITY;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  double* d_x = s.sb.f64;
  double* d_r = d_x + (size_t)N;
  double* d_f = d_r + (size_t)N;
  double* d_o = d_f + (size_t)14 * N;
  int* d_si = s.sb.i32;
  unsigned char* d_i = (unsigned char*)(d_si + (size_t)2 * N);
  QTR_HIP_TRY(h, hipMemcpyAsync(d_x, X, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, s.stream));
  if (ranges) QTR_HIP_TRY(h, hipMemcpyAsync(d_r, ranges, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, s.stream));
  hipLaunchKernelGGL(k_cote_only, dim3(1), dim3(256), 0, s.stream, d_x, N, range, ranges ? d_r : (const double*)nullptr,
                     median_selection ? 1 : 0, d_f, d_si, d_o, d_i);
  QTR_HIP_TRY(h, hipGetLastError());
  double out[2];
  QTR_HIP_TRY(h, hipMemcpyAsync(out, d_o, sizeof(out), hipMemcpyDeviceToHost, s.stream));
  if (inliers) QTR_HIP_TRY(h, hipMemcpyAsync(inliers, d_i, (size_t)N, hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  if (estimate) *estimate = out[0];
  if (n_card) *n_card = (int)out[1];
  return QTR_OK;
}

int qtr_cote_estimate(qtr_handle* h, int slot, const double* X, int N, double range, int median_selection,
                      double* estimate, unsigned char* inliers, int* n_card) {
  if (!(range > 0)) return QTR_ERR_BAD_ARG;
  return cote_impl(h, slot, X, nullptr, N, range, median_selection, estimate, inliers, n_card);
}

int qtr_cote_estimate_ranges(qtr_handle* h, int slot, const double* X, const double* ranges, int N, int median_selection,
                             double* estimate, unsigned char* inliers, int* n_card) {
  if (!ranges || N < 1) return QTR_ERR_BAD_ARG;
  for (int i = 0; i < N; ++i)
    if (!(ranges[i] > 0)) return QTR_ERR_BAD_ARG;
  return cote_impl(h, slot, X, ranges, N, ranges[0], median_selection, estimate, inliers, n_card);
}

// ------------------------------------------------------------------------------------------------
// Patchwork ground segmentation (patchwork.hip)
void qtr_pw_default_params(qtr_pw_params* p) {  // reference config/patchwork_params.yaml
  memset(p, 0, sizeof(*p));
  p->sensor_height = 1.723;
  p->num_iter = 3;
  p->num_lpr = 20;
  p->num_min_pts = 80;
  p->th_seeds = 0.25;
  p->th_dist = 0.125;
  p->max_range = 80.0;
  p->min_range = 2.7;
  p->uprightness_thr = 0.707;
  p->adaptive_seed_selection_margin = -1.1;
  p->using_global_thr = 0;
  p->global_elevation_thr = -0.5;
  p->num_zones = 4;
  const int ns[4] = {16, 32, 54, 32}, nr[4] = {2, 4, 4, 4};
  const double mr[4] = {2.7, 12.3625, 22.025, 41.35};
  const double et[4] = {-1.2, -0.9984, -0.851, -0.605}, ft[4] = {0.0001, 0.000125, 0.000185, 0.000185};
  for (int i = 0; i < 4; ++i) {
    p->num_sectors_each_zone[i] = ns[i];
    p->num_rings_each_zone[i] = nr[i];
    p->min_ranges[i] = mr[i];
    p->elevation_thr[i] = et[i];
    p->flatness_thr[i] = ft[i];
  }
  p->num_thr = 4;
}

// Ground segmentation of one scan in two halves, so that a batch can keep several slots' scans in flight: pw_begin
// validates and enqueues (the two output counts follow the kernels into the slot's pinned words, s.ev[1] marks the end),
// pw_end reads them once the stream has got there.
static int pw_begin(qtr_handle* h, Slot& s, const float* xyz4, int P, const qtr_pw_params* pw, int mem) {
  if (!pw || P < 0 || (P > 0 && !xyz4)) return QTR_ERR_BAD_ARG;
  // check_input_parameters_are_correct (:588-614) + what the kernels rely on
  bool ok = pw->num_zones >= 1 && pw->num_zones <= 4 && pw->num_thr >= 0 && pw->num_thr <= 8 && pw->num_iter >= 1 &&
            pw->num_lpr >= 1 && pw->min_range == pw->min_ranges[0] && pw->max_range > pw->min_range;
  int npatch = 0, nrings = 0;
  for (int k = 0; ok && k < pw->num_zones; ++k) {
    ok = pw->num_sectors_each_zone[k] >= 1 && pw->num_rings_each_zone[k] >= 1 &&
         (k == 0 || pw->min_ranges[k] > pw->min_ranges[k - 1]) && pw->min_ranges[k] < pw->max_range;
    npatch += pw->num_sectors_each_zone[k] * pw->num_rings_each_zone[k];
    nrings += pw->num_rings_each_zone[k];
  }
  if (ok) {  // the reference indexes the threshold vectors with ring + 2 * zone (:395)
    for (int k = 0, ci = 0; k < pw->num_zones; ++k)
      for (int r = 0; r < pw->num_rings_each_zone[k]; ++r, ++ci)
        if (ci < pw->num_thr && r + 2 * k >= pw->num_thr) ok = false;
  }
  if (!ok || npatch > 1024) {
    snprintf(h->err, sizeof(h->err), "Some parameters are wrong! the size of parameters should be same");
    return QTR_ERR_BAD_ARG;
  }
  if (P > h->lim.max_points) {
    snprintf(h->err, sizeof(h->err), "P=%d exceeds max_points=%d", P, h->lim.max_points);
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  if (!s.pw_arena) {
    QTR_HIP_TRY(h, hipMalloc(&s.pw_arena, patchwork_scratch_bytes(h->lim.max_points)));
    patchwork_carve(s.pwb, s.pw_arena, h->lim.max_points);
  }
  const float4* d_in = (const float4*)xyz4;
  if (mem == QTR_MEM_HOST && P > 0) {
    QTR_HIP_TRY(h, hipMemcpyAsync(s.in_src, xyz4, (size_t)P * 16, hipMemcpyHostToDevice, s.stream));
    d_in = s.in_src;
  }
  PwDev d;
  memset(&d, 0, sizeof(d));
  d.sensor_height = pw->sensor_height;
  d.num_iter = pw->num_iter;
  d.num_lpr = pw->num_lpr;
  d.num_min_pts = pw->num_min_pts;
  d.th_seeds = pw->th_seeds;
  d.th_dist = pw->th_dist;
  d.max_range = pw->max_range;
  d.min_range = pw->min_range;
  d.uprightness_thr = pw->uprightness_thr;
  d.margin = (pw->sensor_height == 0.0) ? -0.1 : pw->adaptive_seed_selection_margin * pw->sensor_height;  // :294
  d.using_global_thr = pw->using_global_thr;
  d.global_elevation_thr = pw->global_elevation_thr;
  d.num_zones = pw->num_zones;
  d.num_thr = pw->num_thr;
  for (int k = 0, rb = 0; k < pw->num_zones; ++k) {
    d.nsec[k] = pw->num_sectors_each_zone[k];
    d.nring[k] = pw->num_rings_each_zone[k];
    d.min_ranges[k] = pw->min_ranges[k];
    const double hi = (k + 1 < pw->num_zones) ? pw->min_ranges[k + 1] : pw->max_range;
    d.ring_size[k] = (hi - pw->min_ranges[k]) / pw->num_rings_each_zone[k];
    d.sector_size[k] = 2 * M_PI / pw->num_sectors_each_zone[k];
    d.base[k + 1] = d.base[k] + d.nring[k] * d.nsec[k];
    d.ring_base[k] = rb;
    rb += d.nring[k];
  }
  for (int i = 0; i < 8; ++i) {
    d.elevation_thr[i] = pw->elevation_thr[i];
    d.flatness_thr[i] = pw->flatness_thr[i];
  }
  QTR_HIP_TRY(h, hipEventRecord(s.ev[0], s.stream));
  QTR_HIP_TRY(h, patchwork_enqueue(s.fb, s.pwb, d_in, P, d, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.pinned_i32, s.pwb.offs + 2 * 1024, 2 * sizeof(int), hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
  return QTR_OK;
}
static void pw_end(Slot& s, int* n_ground, int* n_nonground) {  // (after s.ev[1] has completed)
  *n_ground = s.pinned_i32[0];
  *n_nonground = s.pinned_i32[1];
}

int qtr_patchwork(qtr_handle* h, int slot, const float* xyz4, int P, const qtr_pw_params* pw, float* ground_xyzw,
                  int cap_ground, int* n_ground, float* nonground_xyzw, int cap_nonground, int* n_nonground, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !pw || !n_ground || !n_nonground || P < 0 || (P > 0 && !xyz4)) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  *n_ground = *n_nonground = 0;
  QTR_TRY(pw_begin(h, s, xyz4, P, pw, mem));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  int ng = 0, nn = 0;
  pw_end(s, &ng, &nn);
  *n_ground = ng;
  *n_nonground = nn;
  if ((ground_xyzw && ng > cap_ground) || (nonground_xyzw && nn > cap_nonground)) {
    snprintf(h->err, sizeof(h->err), "output capacity too small (%d ground, %d non-ground)", ng, nn);
    return QTR_ERR_CAPACITY;
  }
  const hipMemcpyKind kout = mem == QTR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (ground_xyzw && ng > 0) QTR_HIP_TRY(h, hipMemcpyAsync(ground_xyzw, s.pwb.out_g, (size_t)ng * 16, kout, s.stream));
  if (nonground_xyzw && nn > 0) QTR_HIP_TRY(h, hipMemcpyAsync(nonground_xyzw, s.pwb.out_n, (size_t)nn * 16, kout, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  float ms = 0;
  s.times_pending = 0;
  s.times = qtr_stage_times{};
  if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) s.times.total = ms;
  return QTR_OK;
}

// ------------------------------------------------------------------------------------------------
// range-image projection + sub-cluster rejection (segment.hip)
int qtr_ip_default_params(const char* lidar, const char* nbr, qtr_ip_params* p) {
  if (!lidar || !nbr || !p) return QTR_ERR_BAD_ARG;
  struct Row {
    const char* name;
    int ns, hs;
    float rx, ry, ab;
  };
  const Row rows[] = {  // reference include/imageProjection.hpp:85-131
      {"Velodyne-64-HDE", 64, 1800, 360.0f / 1800.0f, 26.9f / 63.0f, 25.0f},
      {"VLP-16", 16, 1800, 0.2f, 2.0f, (float)(15.0 + 0.1)},
      {"HDL-32E", 32, 1800, 360.0f / 1800.0f, 41.33f / 31.0f, 30.67f},
      {"Ouster-OS1-16", 16, 1024, 360.0f / 1024.0f, 33.2f / 15.0f, (float)(16.6 + 0.1)},
      {"Ouster-OS1-64", 64, 1024, 360.0f / 1024.0f, 33.2f / 63.0f, (float)(16.6 + 0.1)},
  };
  const Row* r = nullptr;
  for (const Row& q : rows)
    if (strcmp(q.name, lidar) == 0) r = &q;
  int mode = -1;
  if (strcmp(nbr, "4Neighbor") == 0) mode = 0;
  if (strcmp(nbr, "8Neighbor") == 0) mode = 1;
  if (strcmp(nbr, "4CrossNeighbor") == 0) mode = 2;
  if (!r || mode < 0) return QTR_ERR_BAD_ARG;
  p->n_scan = r->ns;
  p->horizon_scan = r->hs;
  p->ang_res_x = r->rx;
  p->ang_res_y = r->ry;
  p->ang_bottom = r->ab;
  p->neighbor_mode = mode;
  p->num_min_pts = 30;
  p->segment_theta = (float)(60.0 / 180.0 * M_PI);
  p->valid_point_num = 5;
  p->valid_line_num = 3;
  return QTR_OK;
}

// (the same split as pw_begin / pw_end: three counts — valid points, outliers, segments — into the slot's pinned words)
static int seg_begin(qtr_handle* h, Slot& s, const float* xyz4, int P, const qtr_ip_params* ip, int mem) {
  if (!ip || P < 0 || (P > 0 && !xyz4)) return QTR_ERR_BAD_ARG;
  if (ip->n_scan < 1 || ip->n_scan > 64 || ip->horizon_scan < 4 || ip->horizon_scan > 8192 || !(ip->ang_res_x > 0) ||
      !(ip->ang_res_y > 0) || ip->neighbor_mode < 0 || ip->neighbor_mode > 2) {
    snprintf(h->err, sizeof(h->err), "[ImageProjection]:Check your paramter. (n_scan <= 64, horizon_scan <= 8192)");
    return QTR_ERR_BAD_ARG;
  }
  if (P > h->lim.max_points) {
    snprintf(h->err, sizeof(h->err), "P=%d exceeds max_points=%d", P, h->lim.max_points);
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  const int NP = ip->n_scan * ip->horizon_scan;
  if (!s.seg_arena || s.seg.np_cap < NP) {
    if (s.seg_arena) QTR_HIP_TRY(h, hipFree(s.seg_arena));
    s.seg_arena = nullptr;
    QTR_HIP_TRY(h, hipMalloc(&s.seg_arena, segment_scratch_bytes(NP)));
    segment_carve(s.seg, s.seg_arena, NP);
  }
  const float4* d_in = (const float4*)xyz4;
  if (mem == QTR_MEM_HOST && P > 0) {
    QTR_HIP_TRY(h, hipMemcpyAsync(s.in_src, xyz4, (size_t)P * 16, hipMemcpyHostToDevice, s.stream));
    d_in = s.in_src;
  }
  IpDev d;
  d.n_scan = ip->n_scan;
  d.horizon_scan = ip->horizon_scan;
  d.ang_res_x = ip->ang_res_x;
  d.ang_res_y = ip->ang_res_y;
  d.ang_bottom = ip->ang_bottom;
  d.neighbor_mode = ip->neighbor_mode;
  d.num_min_pts = ip->num_min_pts;
  d.segment_theta = ip->segment_theta;
  d.valid_point_num = ip->valid_point_num;
  d.valid_line_num = ip->valid_line_num;
  {  // segmentAlphaX / Y = ang_res / 180 * pi stored in float (:132-133); their sin / cos through qtr_math.h
    const float ax = (float)((double)ip->ang_res_x / 180.0 * M_PI), ay = (float)((double)ip->ang_res_y / 180.0 * M_PI);
    qm_sincosf(ax, &d.sx, &d.cx);
    qm_sincosf(ay, &d.sy, &d.cy);
  }
  QTR_HIP_TRY(h, hipEventRecord(s.ev[0], s.stream));
  int* d_tot = s.seg.blk + 3 * ((NP + 1023) / 1024 + 1);
  QTR_HIP_TRY(h, segment_enqueue(s.seg, d_in, P, d, d_tot, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.pinned_i32, d_tot, 3 * sizeof(int), hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
  return QTR_OK;
}

int qtr_segment_cloud(qtr_handle* h, int slot, const float* xyz4, int P, const qtr_ip_params* ip, float* valid_xyzl,
                      int cap_valid, int* n_valid, float* outl_xyzi, int cap_outl, int* n_outl, int* n_segments,
                      int* labelmat, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !ip || !n_valid || P < 0 || (P > 0 && !xyz4)) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  *n_valid = 0;
  if (n_outl) *n_outl = 0;
  if (n_segments) *n_segments = 0;
  QTR_TRY(seg_begin(h, s, xyz4, P, ip, mem));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  const int NP = ip->n_scan * ip->horizon_scan;
  const int nv = s.pinned_i32[0], no = s.pinned_i32[1], nseg = s.pinned_i32[2];
  *n_valid = nv;
  if (n_outl) *n_outl = no;
  if (n_segments) *n_segments = nseg;
  if ((valid_xyzl && nv > cap_valid) || (outl_xyzi && no > cap_outl)) {
    snprintf(h->err, sizeof(h->err), "output capacity too small (%d valid, %d outliers)", nv, no);
    return QTR_ERR_CAPACITY;
  }
  const hipMemcpyKind kout = mem == QTR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (valid_xyzl && nv > 0) QTR_HIP_TRY(h, hipMemcpyAsync(valid_xyzl, s.seg.out_valid, (size_t)nv * 16, kout, s.stream));
  if (outl_xyzi && no > 0) QTR_HIP_TRY(h, hipMemcpyAsync(outl_xyzi, s.seg.out_outl, (size_t)no * 16, kout, s.stream));
  if (labelmat) QTR_HIP_TRY(h, hipMemcpyAsync(labelmat, s.seg.labelmat, (size_t)NP * 4, hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  float ms = 0;
  s.times_pending = 0;
  s.times = qtr_stage_times{};
  if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) s.times.total = ms;
  return QTR_OK;
}

int qtr_set_stage_events(qtr_handle* h, int on) {
  if (!h) return QTR_ERR_BAD_ARG;
  h->stage_events = on ? 1 : 0;
  return QTR_OK;
}

int qtr_set_nn_event_stride(qtr_handle* h, int every) {
  if (!h || every < 0) return QTR_ERR_BAD_ARG;
  h->nn_event_stride = every;
  for (auto& s : h->slots) s.n_matches = 0;  // the next match of every slot is a timed one
  return QTR_OK;
}

int qtr_get_nn_dir_times(qtr_handle* h, int slot, float* dir1_ms, float* dir2_ms) {
  Slot* sp = get_slot(h, slot);
  if (!sp) return QTR_ERR_BAD_ARG;
  if (sp->times_pending) {  // (the events of the last call are read on demand, like qtr_get_stage_times does)
    qtr_stage_times t;
    (void)qtr_get_stage_times(h, slot, &t);
  }
  if (dir1_ms) *dir1_ms = sp->nn_dir_ms[0];
  if (dir2_ms) *dir2_ms = sp->nn_dir_ms[1];
  return QTR_OK;
}

int qtr_get_nn_totals(qtr_handle* h, int slot, double* total_ms, long long* launches, int reset) {
  Slot* sp = get_slot(h, slot);
  if (!sp) return QTR_ERR_BAD_ARG;
  (void)hipStreamSynchronize(sp->stream);
  flush_nn_totals(*sp);
  if (total_ms) *total_ms = sp->nn_total_ms;
  if (launches) *launches = sp->nn_total_launches;
  if (reset) {
    sp->nn_total_ms = 0;
    sp->nn_total_launches = 0;
  }
  return QTR_OK;
}

int qtr_get_stage_times(qtr_handle* h, int slot, qtr_stage_times* out) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !out) return QTR_ERR_BAD_ARG;
  compute_times(*sp);
  *out = sp->times;
  return QTR_OK;
}

// ------------------------------------------------------------------------------------------------
// front end
int qtr_voxelize(qtr_handle* h, int slot, const float* xyz4, int P, float leaf, float* out_xyz4, int cap, int* n_out,
                 int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !n_out || P < 0 || (P > 0 && (!xyz4 || !out_xyz4)) || !(leaf > 0)) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  *n_out = 0;
  if (P == 0) return QTR_OK;
  if (P > h->lim.max_points) {
    snprintf(h->err, sizeof(h->err), "P=%d exceeds max_points=%d", P, h->lim.max_points);
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  const float4* d_in = (const float4*)xyz4;
  if (mem == QTR_MEM_HOST) {
    QTR_HIP_TRY(h, hipMemcpyAsync(s.in_src, xyz4, (size_t)P * 16, hipMemcpyHostToDevice, s.stream));
    d_in = s.in_src;
  }
  QTR_HIP_TRY(h, hipEventRecord(s.ev[0], s.stream));
  CloudBufs& cb = s.fb.cloud[0];
  {
    const float4* raws[1] = {d_in};
    const int Ps[1] = {P};
    QTR_HIP_TRY(h, voxelize_enqueue(s.fb, 1, raws, Ps, leaf, s.stream));
  }
  QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.pinned_i32, cb.counts, 16 * sizeof(int), hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  int n = s.pinned_i32[CNT_NVOX];
  if (n < 0 || s.pinned_i32[CNT_VOX_TAILERR]) {  // (see front_device; the sticky word: ANY tile gave up, not only the last)
    snprintf(h->err, sizeof(h->err), "voxel grid: look-back timed out");
    return QTR_ERR_HIP;
  }
  const bool passthrough = s.pinned_i32[CNT_VOX_OVERFLOW] != 0;
  const float4* d_out = cb.vox;
  if (passthrough) {  // PCL: "Leaf size is too small" -> output = input
    n = P;
    d_out = d_in;
  }
  if (n > h->lim.max_voxels && !passthrough) {
    snprintf(h->err, sizeof(h->err), "voxel count %d exceeds max_voxels=%d", n, h->lim.max_voxels);
    return QTR_ERR_CAPACITY;
  }
  if (n > cap) {
    snprintf(h->err, sizeof(h->err), "voxel count %d exceeds output capacity %d", n, cap);
    *n_out = n;
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipMemcpyAsync(out_xyz4, d_out, (size_t)n * 16,
                                mem == QTR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  *n_out = n;
  float ms = 0;
  s.times_pending = 0;
  s.times = qtr_stage_times{};
  if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) s.times.voxelize = s.times.total = ms;
  return QTR_OK;
}

int qtr_fpfh(qtr_handle* h, int slot, const float* xyz4, int n, float r_normal, float r_fpfh, float* normals4,
             float* desc33, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || n < 0 || (n > 0 && (!xyz4 || !desc33))) return QTR_ERR_BAD_ARG;
  if (r_normal > r_fpfh || !(r_normal > 0)) {  // reference include/fpfh_manager.hpp:99-102
    snprintf(h->err, sizeof(h->err), "[FPFHManager]: Normal should be lower than fpfh_radius!!!!");
    return QTR_ERR_BAD_ARG;
  }
  Slot& s = *sp;
  s.last_n = n;
  if (n == 0) return QTR_OK;
  if (n > h->lim.max_voxels) {
    snprintf(h->err, sizeof(h->err), "n=%d exceeds max_voxels=%d", n, h->lim.max_voxels);
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  CloudBufs& cb = s.fb.cloud[0];
  const hipMemcpyKind kin = mem == QTR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  const hipMemcpyKind kout = mem == QTR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  QTR_HIP_TRY(h, hipMemcpyAsync(cb.vox, xyz4, (size_t)n * 16, kin, s.stream));
  QTR_HIP_TRY(h, hipMemsetAsync(cb.counts, 0, 16 * sizeof(int), s.stream));
  QTR_HIP_TRY(h, set_count_enqueue(cb, CNT_NVOX, n, s.stream));
  QTR_HIP_TRY(h, hipEventRecord(s.ev[0], s.stream));
  {
    const int ns1[1] = {n};
    // an arbitrary cloud (dense mode: no voxel grid in front): lists of any length
    QTR_TRY(ensure_long_arenas(h, s));
    QTR_HIP_TRY(h, fpfh_enqueue(s.fb, 0, 1, ns1, r_normal, r_fpfh, s.stream, true, false, true));
  }
  QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.pinned_i32, cb.counts, 16 * sizeof(int), hipMemcpyDeviceToHost, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  if (s.pinned_i32[CNT_NBR_CAPACITY]) {
    snprintf(h->err, sizeof(h->err), "radius-neighbour lists longer than %d entries need %d entries of the long-list arena "
             "(longest list %d); qtr_limits.max_long_neighbors is %d", QTR_KMAX, s.pinned_i32[CNT_NBR_ARENA],
             s.pinned_i32[CNT_KMAX], h->lim.max_long_neighbors);
    return QTR_ERR_CAPACITY;
  }
  if (normals4) QTR_HIP_TRY(h, hipMemcpyAsync(normals4, cb.normals, (size_t)n * 16, kout, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(desc33, cb.fpfh, (size_t)n * 33 * 4, kout, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  float ms = 0;
  s.times_pending = 0;
  s.times = qtr_stage_times{};
  if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) s.times.fpfh = s.times.total = ms;
  return QTR_OK;
}

// Matching on device-resident clouds/descriptors held in fb.cloud[0] (source) and fb.cloud[1] (target).
static int match_device(qtr_handle* h, Slot& s, int ns, int nt, const qtr_frontend_params* fp, int* L_out,
                        bool init_done = false, bool prep_done = false, bool tolerate_tail = false) {
  flush_nn_totals(s);
  // (an event pair attached to a launch costs ~5 us of queue time on either side of it: a caller that only wants the
  // average duration of the launches — the bench's roofline — has every n-th match timed, qtr_set_nn_event_stride)
  const int stride = h->nn_event_stride;
  s.fb.nn_events = (stride > 0 && (s.n_matches++ % stride) == 0) ? 1 : 0;
  s.nn_timed_last = (s.fb.nn_events && s.fb.nn_engine != 0) ? 1 : 0;
  s.nn_pending = s.nn_timed_last;
  s.fb.mail_seq = ++s.seq;
  QTR_HIP_TRY(h, match_enqueue(s.fb, ns, nt, *fp, s.stream, init_done, prep_done));
  QTR_TRY(wait_mail(h, s, MAIL_SEQ_MATCH, s.seq));  // k_corr_compact2 left the counters in the mailbox
  *L_out = s.mail[MAIL_MATCH + MC_NCORR];
  if (*L_out < 0) {  // a multi-workgroup compaction of the tail gave up waiting for a predecessor's count (match.hip)
    if (tolerate_tail) return QTR_OK;  // (nobody reads the matcher's list: the caller reports -1 as the matched count)
    *L_out = 0;
    (void)hipStreamSynchronize(s.stream);
    snprintf(h->err, sizeof(h->err), "matcher tail: look-back timed out (no correspondences were written)");
    return QTR_ERR_HIP;
  }
  return QTR_OK;
}

int qtr_match(qtr_handle* h, int slot, const float* xyz4_s, int n_s, const float* desc33_s, const float* xyz4_t,
              int n_t, const float* desc33_t, const qtr_frontend_params* fp, int* corr2, int cap, int* L_out, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !fp || !L_out || n_s < 0 || n_t < 0) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  *L_out = 0;
  s.last_ns = n_s;
  s.last_nt = n_t;
  if (n_s == 0 || n_t == 0) return QTR_OK;
  if (!xyz4_s || !xyz4_t || !desc33_s || !desc33_t || !corr2) return QTR_ERR_BAD_ARG;
  if (n_s > h->lim.max_voxels || n_t > h->lim.max_voxels) {
    snprintf(h->err, sizeof(h->err), "cloud size exceeds max_voxels=%d", h->lim.max_voxels);
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  const hipMemcpyKind kin = mem == QTR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  const hipMemcpyKind kout = mem == QTR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[0].vox, xyz4_s, (size_t)n_s * 16, kin, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[1].vox, xyz4_t, (size_t)n_t * 16, kin, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[0].fpfh, desc33_s, (size_t)n_s * 132, kin, s.stream));
  QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[1].fpfh, desc33_t, (size_t)n_t * 132, kin, s.stream));
  QTR_HIP_TRY(h, hipEventRecord(s.ev[0], s.stream));
  {
    const int n2[2] = {n_s, n_t};  // Matcher::normalizePoints means of the two clouds handed in
    QTR_HIP_TRY(h, mean_enqueue(s.fb, 0, 2, n2, s.stream));
  }
  int L = 0;
  int rc = match_device(h, s, n_s, n_t, fp, &L);
  if (rc != QTR_OK) return rc;
  QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
  *L_out = L;
  if (L > cap) {
    snprintf(h->err, sizeof(h->err), "L=%d exceeds output capacity %d", L, cap);
    return QTR_ERR_CAPACITY;
  }
  if (L > 0) QTR_HIP_TRY(h, hipMemcpyAsync(corr2, s.fb.corr, (size_t)L * 8, kout, s.stream));
  QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  float ms = 0;
  s.times_pending = 0;
  s.times = qtr_stage_times{};
  if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) s.times.match = s.times.total = ms;
  fill_nn_times(s);
  return QTR_OK;
}

// Front end of one pair on one slot: voxel grid x2 -> FPFH x2 -> reciprocal matching -> matched keypoint clouds gathered
// into s.m_src / s.m_tgt (device).  What the reference does in `voxelize` x2 (include/quatro.hpp:49-68) +
// FPFHManager::setFeaturePair (include/fpfh_manager.hpp:98-153).  Counts go to *ns_out / *nt_out / *L_out; with
// `for_solver` the solver's clean slate is enqueued beside the FPFH chain (qtr_register_pair).  An error return leaves
// nothing in flight that still reads the caller's scans.
static int front_device(qtr_handle* h, Slot& s, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                        const qtr_frontend_params* fp, int mem, bool for_solver, int* ns_out, int* nt_out, int* L_out,
                        bool corr_given = false) {
  // corr_given: the back end will run on the CALLER's correspondences — the matcher's list is only counted (*L_out; -1 when
  // its tail gave up), so neither a list longer than max_corr nor a tail failure fails the registration, and the matched
  // clouds are not gathered.
  int rc = QTR_OK;
  if (fp->normal_radius > fp->fpfh_radius) {
    snprintf(h->err, sizeof(h->err), "[FPFHManager]: Normal should be lower than fpfh_radius!!!!");
    return QTR_ERR_BAD_ARG;
  }
  if (Ps <= 0 || Pt <= 0 || !src_raw4 || !tgt_raw4) {
    snprintf(h->err, sizeof(h->err), "Invalid or empty point cloud dataset given!");
    return QTR_ERR_BAD_ARG;
  }
  if (Ps > h->lim.max_points || Pt > h->lim.max_points) {
    snprintf(h->err, sizeof(h->err), "cloud exceeds max_points=%d", h->lim.max_points);
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  const float4 *d_s = (const float4*)src_raw4, *d_t = (const float4*)tgt_raw4;
  if (mem == QTR_MEM_HOST) {
    QTR_HIP_TRY(h, hipMemcpyAsync(s.in_src, src_raw4, (size_t)Ps * 16, hipMemcpyHostToDevice, s.stream));
    QTR_HIP_TRY(h, hipMemcpyAsync(s.in_tgt, tgt_raw4, (size_t)Pt * 16, hipMemcpyHostToDevice, s.stream));
    d_s = s.in_src;
    d_t = s.in_tgt;
  }
  if (h->stage_events) QTR_HIP_TRY(h, hipEventRecord(s.ev[0], s.stream));
  // k2_vox_centroids publishes the voxel counters from its first block while other blocks may still be reading the raw
  // scans: an error return taken right after the mail must not hand the scans back to the caller (who may free them)
  // before the stream has drained
  auto fail_drained = [&](int code) {
    (void)hipStreamSynchronize(s.stream);
    return code;
  };
  // both clouds go through every front-end kernel together (blockIdx.y = cloud)
  {
    const float4* raws[2] = {d_s, d_t};
    const int Ps2[2] = {Ps, Pt};
    // The voxel sort needs ceil(bits / 8) radix passes, bits = significant bits of the grid's cell index — known on the
    // device only.  A launch that returns at once still costs ~5 us on this chain, so the driver launches what the
    // previous pair on this slot needed (3 for a lidar scan at 0.3 m) and checks: if this pair needs more, the stage
    // runs again with enough (first call on a slot: 4, which always suffices).
    for (int attempt = 0;; ++attempt) {
      const int launched = s.fb.vox_passes;
      s.fb.mail_seq = ++s.seq;
      QTR_HIP_TRY(h, voxelize_enqueue(s.fb, 2, raws, Ps2, fp->voxel_size, s.stream, launched, fp->fpfh_radius * 1.001f));
      // block 0 of k2_vox_centroids publishes the counters while other blocks are still writing centroids: anything
      // that reads the centroids from another stream has to wait for the kernel itself
      QTR_HIP_TRY(h, hipEventRecord(s.ev_vox, s.stream));
      // The matcher's sequential means (70 us of one dependent chain per cloud: as long as the whole FPFH chain since round
      // 6's cuts) start right behind the centroids, on the second stream, with the voxel counts read on the device — not
      // after the mail, the host's checks and the FPFH chain's ten launches.
      {
        const int on_device[2] = {-1, -1};
        QTR_HIP_TRY(h, hipStreamWaitEvent(s.stream2, s.ev_vox, 0));
        QTR_HIP_TRY(h, mean_enqueue(s.fb, 0, 2, on_device, s.stream2, h->lim.max_voxels));
      }
      // k2_vox_centroids leaves both clouds' counters in the mailbox
      if ((rc = wait_mail(h, s, MAIL_SEQ_VOX0, s.seq)) != QTR_OK || (rc = wait_mail(h, s, MAIL_SEQ_VOX1, s.seq)) != QTR_OK)
        return fail_drained(rc);
      const int bits = std::max(s.mail[MAIL_VOX0 + CNT_SORT_BITS], s.mail[MAIL_VOX1 + CNT_SORT_BITS]);
      const int needed = std::min(4, std::max(1, (bits + 7) / 8));
      if (needed > launched && attempt == 0) {  // under-launched: the centroids are garbage, run the stage again
        s.fb.vox_passes = 4;
        s.fb.vox_fewer = 0;
        continue;
      }
      if (needed < launched) {  // step down only after a run of calls that agree (alternating scenes would thrash)
        if (++s.fb.vox_fewer >= 4) {
          s.fb.vox_passes = needed;
          s.fb.vox_fewer = 0;
        }
      } else {
        s.fb.vox_fewer = 0;
      }
      break;
    }
  }
  int ns = s.mail[MAIL_VOX0 + CNT_NVOX], nt = s.mail[MAIL_VOX1 + CNT_NVOX];
  bool passed_through = false;
  if (ns < 0 || nt < 0) {  // k2_vox_centroids: a tile never published its count (bounded look-back): nothing usable was written
    snprintf(h->err, sizeof(h->err), "voxel grid: look-back timed out");
    return fail_drained(QTR_ERR_HIP);
  }
  {
    // pcl::VoxelGrid::applyFilter: "Leaf size is too small for the input dataset. Integer indices would overflow" ->
    // output = input.  So does `voxelize` of the reference (include/quatro.hpp:49-68 calls it unconditionally), and the
    // demo goes on with the cloud as it is — BASELINE's dense mode ("no voxel downsample") through the whole-path entry.
    const bool pass[2] = {s.mail[MAIL_VOX0 + CNT_VOX_OVERFLOW] != 0, s.mail[MAIL_VOX1 + CNT_VOX_OVERFLOW] != 0};
    if (pass[0] || pass[1]) {
      passed_through = true;
      if ((pass[0] && Ps > h->lim.max_voxels) || (pass[1] && Pt > h->lim.max_voxels)) {
        snprintf(h->err, sizeof(h->err), "voxel grid would overflow int32 (leaf too small): the cloud passes through as it "
                 "is (pcl::VoxelGrid), and its %d / %d points exceed max_voxels=%d", Ps, Pt, h->lim.max_voxels);
        return fail_drained(QTR_ERR_CAPACITY);
      }
      if (pass[0]) {
        QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[0].vox, d_s, (size_t)Ps * 16, hipMemcpyDeviceToDevice, s.stream));
        ns = Ps;
      }
      if (pass[1]) {
        QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[1].vox, d_t, (size_t)Pt * 16, hipMemcpyDeviceToDevice, s.stream));
        nt = Pt;
      }
      QTR_HIP_TRY(h, hipEventRecord(s.ev_vox, s.stream));  // (the second stream's means wait for the clouds)
    }
  }
  if (ns > h->lim.max_voxels || nt > h->lim.max_voxels) {
    snprintf(h->err, sizeof(h->err), "voxel count (%d,%d) exceeds max_voxels=%d", ns, nt, h->lim.max_voxels);
    return fail_drained(QTR_ERR_CAPACITY);
  }
  *ns_out = ns;
  *nt_out = nt;
  s.last_ns = ns;
  s.last_nt = nt;
  if (h->stage_events) QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
  {
    const int n2[2] = {ns, nt};
    // Beside the FPFH chain, on the second stream: the matcher's sequential means (already running: enqueued behind the
    // voxel stage above; a cloud that passed through is summed again from the copy), and the matcher's and the solver's
    // clean slates (they depend on the voxel counts alone).
    if (passed_through) {
      QTR_HIP_TRY(h, hipStreamWaitEvent(s.stream2, s.ev_vox, 0));
      QTR_HIP_TRY(h, mean_enqueue(s.fb, 0, 2, n2, s.stream2));
    }
    if (h->long_lists) QTR_TRY(ensure_long_arenas(h, s));
    // (k2_fpfh also does the matcher's per-descriptor preparation: norms, hashes, duplicate table — see frontend.hip)
    QTR_HIP_TRY(h, fpfh_enqueue(s.fb, 0, 2, n2, fp->normal_radius, fp->fpfh_radius, s.stream, false, true, h->long_lists, true,
                                cell_table_cells(s.mail[MAIL_VOX0 + CNT_NCELL], s.mail[MAIL_VOX1 + CNT_NCELL])));
    // (the solver's clean slate rides in the matcher's: one launch fewer beside the FPFH chain)
    QTR_HIP_TRY(h, match_init_enqueue(s.fb, ns, nt, *fp, s.stream2, false, for_solver ? (int*)s.sb.st : nullptr,
                                      (int)(sizeof(SolverState) / 4)));
    QTR_HIP_TRY(h, hipEventRecord(s.ev[5], s.stream2));
    QTR_HIP_TRY(h, hipStreamWaitEvent(s.stream, s.ev[5], 0));
  }
  if (h->stage_events) QTR_HIP_TRY(h, hipEventRecord(s.ev[6], s.stream));
  int L = 0;
  rc = match_device(h, s, ns, nt, fp, &L, true, true, corr_given);
  if (rc != QTR_OK) return rc;
  if (s.mail[MAIL_CNT0 + CNT_VOX_TAILERR] || s.mail[MAIL_CNT1 + CNT_VOX_TAILERR]) {
    // a MIDDLE tile of k2_vox_centroids gave up its look-back (its centroids were never written) although the last tile's
    // came out whole and mailed a valid count: the counter lines the matcher's tail mails after that kernel carry the word
    snprintf(h->err, sizeof(h->err), "voxel grid: look-back timed out in a tile (centroids incomplete)");
    return QTR_ERR_HIP;
  }
  if (s.mail[MAIL_CNT0 + CNT_NBR_CAPACITY] || s.mail[MAIL_CNT1 + CNT_NBR_CAPACITY]) {
    snprintf(h->err, sizeof(h->err), "radius-neighbour lists longer than %d entries (longest %d / %d) exceed the long-list "
             "arena: qtr_limits.max_long_neighbors is %d", QTR_KMAX, s.mail[MAIL_CNT0 + CNT_KMAX],
             s.mail[MAIL_CNT1 + CNT_KMAX], h->lim.max_long_neighbors);
    return QTR_ERR_CAPACITY;
  }
  if (!h->long_lists && (s.mail[MAIL_CNT0 + CNT_NBR_OVERFLOW] || s.mail[MAIL_CNT1 + CNT_NBR_OVERFLOW])) {
    // A point with more than QTR_KMAX neighbours, and the chain ran without k2_neighbors_big (voxel-grid centroids at
    // the demo's leaf never have that many, so the launch is left out until a cloud needs it): descriptors and matches
    // of this call are not usable.  From now on the handle's chains include it; this pair goes round again.
    h->long_lists = true;
    return front_device(h, s, src_raw4, Ps, tgt_raw4, Pt, fp, mem, for_solver, ns_out, nt_out, L_out, corr_given);
  }
  *L_out = L;
  if (corr_given) return QTR_OK;
  s.last_L = L;
  if (L > h->lim.max_corr) {
    snprintf(h->err, sizeof(h->err), "L=%d exceeds max_corr=%d", L, h->lim.max_corr);
    return QTR_ERR_CAPACITY;
  }
  QTR_HIP_TRY(h, gather_matched_enqueue(s.fb, L, s.m_src, s.m_tgt, s.stream));
  if (h->stage_events) QTR_HIP_TRY(h, hipEventRecord(s.ev[7], s.stream));
  return QTR_OK;
}

// The whole path on one slot.  mem_in: where the scans live; mem_out: where the index lists go.  corr_src / corr_tgt
// (device pointers, n_corr >= 0): the back end runs on THESE matched clouds instead of the matcher's output (the batched
// entry's "scans + pre-matched correspondences" pairs); n_corr < 0: the matcher's own correspondences.
static int register_pair_impl(qtr_handle* h, Slot& s, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                              const qtr_frontend_params* fp, const qtr_params* prm, qtr_result* res, int* clique,
                              int* final_inliers, int cap, int mem_in, int mem_out, const float4* corr_src,
                              const float4* corr_tgt, int n_corr) {
  int L = 0;
  int rc = front_device(h, s, src_raw4, Ps, tgt_raw4, Pt, fp, mem_in, true, &res->n_src, &res->n_tgt, &L, n_corr >= 0);
  res->n_corr = L;
  if (rc != QTR_OK) return res->status = rc;
  if (n_corr >= 0) {
    res->n_corr = n_corr;
    rc = solve_device(h, s, corr_src, corr_tgt, n_corr, prm, res, true);
  } else {
    rc = solve_device(h, s, s.m_src, s.m_tgt, L, prm, res, true);
  }
  if (rc != QTR_OK && rc != QTR_ERR_CLIQUE_TOO_SMALL) return rc;
  s.times_pending = h->stage_events ? 2 : 4;
  const int rc2 = copy_out_lists(h, s, res, clique, nullptr, final_inliers, cap, mem_out);
  if (rc2 != QTR_OK) return res->status = rc2;
  return rc;
}

int qtr_register_pair(qtr_handle* h, int slot, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                      const qtr_frontend_params* fp, const qtr_params* prm, qtr_result* res, int* clique,
                      int* final_inliers, int cap, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !res || !fp) return QTR_ERR_BAD_ARG;
  memset(res, 0, sizeof(*res));
  const int rc = check_params(h, prm);
  if (rc != QTR_OK) return res->status = rc;
  return register_pair_impl(h, *sp, src_raw4, Ps, tgt_raw4, Pt, fp, prm, res, clique, final_inliers, cap, mem, mem, nullptr,
                            nullptr, -1);
}

int qtr_register_pair_corr(qtr_handle* h, int slot, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                           const qtr_frontend_params* fp, const float* corr_src4, const float* corr_tgt4, int n_corr,
                           const qtr_params* prm, qtr_result* res, int* n_matched, int* clique, int* final_inliers, int cap,
                           int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !res || !fp) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  memset(res, 0, sizeof(*res));
  if (n_matched) *n_matched = 0;
  int rc = check_params(h, prm);
  if (rc != QTR_OK) return res->status = rc;
  if (n_corr < 0 || (n_corr > 0 && (!corr_src4 || !corr_tgt4))) {
    snprintf(h->err, sizeof(h->err), "bad correspondence clouds");
    return res->status = QTR_ERR_BAD_ARG;
  }
  if (n_corr > h->lim.max_corr) {
    snprintf(h->err, sizeof(h->err), "n_corr=%d exceeds max_corr=%d", n_corr, h->lim.max_corr);
    return res->status = QTR_ERR_CAPACITY;
  }
  int Lm = 0;
  rc = front_device(h, s, src_raw4, Ps, tgt_raw4, Pt, fp, mem, true, &res->n_src, &res->n_tgt, &Lm, true);
  if (n_matched) *n_matched = Lm;  // (-1: the matcher's tail gave up — its list is not used here)
  res->n_corr = n_corr;
  if (rc != QTR_OK) return res->status = rc;
  const float4 *cs = (const float4*)corr_src4, *ct = (const float4*)corr_tgt4;
  if (mem == QTR_MEM_HOST && n_corr > 0) {  // (the matcher's own matched clouds in m_src / m_tgt are not needed: behind it)
    QTR_HIP_TRY(h, hipMemcpyAsync(s.m_src, corr_src4, (size_t)n_corr * 16, hipMemcpyHostToDevice, s.stream));
    QTR_HIP_TRY(h, hipMemcpyAsync(s.m_tgt, corr_tgt4, (size_t)n_corr * 16, hipMemcpyHostToDevice, s.stream));
    cs = s.m_src;
    ct = s.m_tgt;
  }
  rc = solve_device(h, s, cs, ct, n_corr, prm, res, true);
  if (rc != QTR_OK && rc != QTR_ERR_CLIQUE_TOO_SMALL) return rc;
  s.times_pending = h->stage_events ? 2 : 4;
  const int rc2 = copy_out_lists(h, s, res, clique, nullptr, final_inliers, cap, mem);
  if (rc2 != QTR_OK) return res->status = rc2;
  return rc;
}

int qtr_feature_pair(qtr_handle* h, int slot, const float* src_raw4, int Ps, const float* tgt_raw4, int Pt,
                     const qtr_frontend_params* fp, int* n_src, int* n_tgt, int* L_out, float* src_kps4, float* tgt_kps4,
                     int* corr2, int cap, int mem) {
  Slot* sp = get_slot(h, slot);
  if (!sp || !fp || !L_out) return QTR_ERR_BAD_ARG;
  Slot& s = *sp;
  int ns = 0, nt = 0, L = 0;
  *L_out = 0;
  const int rc = front_device(h, s, src_raw4, Ps, tgt_raw4, Pt, fp, mem, false, &ns, &nt, &L);
  if (n_src) *n_src = ns;
  if (n_tgt) *n_tgt = nt;
  *L_out = L;
  if (rc != QTR_OK) return rc;
  s.times_pending = h->stage_events ? 3 : 4;
  if ((src_kps4 || tgt_kps4 || corr2) && L > cap) {
    snprintf(h->err, sizeof(h->err), "L=%d exceeds output capacity %d", L, cap);
    return QTR_ERR_CAPACITY;
  }
  const hipMemcpyKind kout = mem == QTR_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (L > 0) {
    if (src_kps4) QTR_HIP_TRY(h, hipMemcpyAsync(src_kps4, s.m_src, (size_t)L * 16, kout, s.stream));
    if (tgt_kps4) QTR_HIP_TRY(h, hipMemcpyAsync(tgt_kps4, s.m_tgt, (size_t)L * 16, kout, s.stream));
    if (corr2) QTR_HIP_TRY(h, hipMemcpyAsync(corr2, s.fb.corr, (size_t)L * 8, kout, s.stream));
  }
  // host outputs are complete on return; device outputs are ordered on the slot's stream (qtr_slot_stream), like the
  // next call on this slot
  if (mem == QTR_MEM_HOST) QTR_HIP_TRY(h, hipStreamSynchronize(s.stream));
  return QTR_OK;
}

// ------------------------------------------------------------------------------------------------
// batched registration: lanes of slots stepped through the three launch chains in lockstep
static void batch_fail_pair(qtr_handle* h, int pair, int status) {
  qtr_result& r = h->job.results[pair];
  r.status = status;
  r.valid = 0;
  h->job.finished[pair] = 1;
  ++h->job.done;
}

// The job failed (HIP error, a lane that never published): drain what is in flight so the handle stays usable, and make
// every record say what happened to its pair — pairs that had been started but did not finish get QTR_ERR_HIP, pairs
// that were never started QTR_ERR_NOT_RUN; finished pairs keep their records.
static void batch_abort(qtr_handle* h) {
  BatchJob& J = h->job;
  for (auto& ln : h->lanes) {
    Slot& lead = h->slots[ln.first_slot];
    (void)hipStreamSynchronize(lead.stream);
    (void)hipStreamSynchronize(lead.stream2);
    ln.phase = 0;
  }
  (void)hipGetLastError();
  for (int i = 0; i < J.B; ++i) {
    if (J.finished[i]) continue;
    memset(&J.results[i], 0, sizeof(qtr_result));
    J.results[i].status = i < J.next ? QTR_ERR_HIP : QTR_ERR_NOT_RUN;
  }
  J.active = false;
}

// What a pair descriptor asks for: scans (front end + back end), pre-matched correspondences (back end only — the
// reference's setInputSource / setInputTarget / computeTransformation on what it is handed,
// examples/run_global_registration.cpp:243-246), or both (front end of the scans, back end on the given correspondences).
static inline bool pair_has_scans(const qtr_pair_desc& pd) { return pd.src_raw4 != nullptr || pd.tgt_raw4 != nullptr; }
static inline bool pair_has_corr(const qtr_pair_desc& pd) {
  return pd.src_corr4 != nullptr || pd.tgt_corr4 != nullptr || pd.n_corr != 0;
}

static int lane_start_chunk(qtr_handle* h, Lane& ln);

// Solver chain of the chunk: the pairs of `from_match` (survivors of the matching chain, ln.L / ln.csrc / ln.ctgt set)
// plus the chunk's correspondence-only pairs.
static int lane_enqueue_solver(qtr_handle* h, Lane& ln, const std::vector<int>& from_match) {
  BatchJob& J = h->job;
  Slot& lead = h->slots[ln.first_slot];
  std::vector<int> all(from_match);
  all.insert(all.end(), ln.corr_only.begin(), ln.corr_only.end());
  ln.corr_only.clear();
  ln.active.swap(all);
  if (ln.active.empty()) return lane_start_chunk(h, ln);
  std::vector<int> Ls;
  std::vector<SolverBufs*> SB;
  std::vector<const float4*> srcs, tgts;
  for (int g : ln.active) {
    Slot& s = h->slots[ln.first_slot + g];
    s.last_L = ln.L[g];
    s.last_Wb = (((ln.L[g] + 63) / 64) + 3) & ~3;
    s.sb.mail_seq = ++s.seq;
    SB.push_back(&s.sb);
    srcs.push_back(ln.csrc[g]);
    tgts.push_back(ln.ctgt[g]);
    Ls.push_back(ln.L[g]);
  }
  solver_set_hca_share((int)h->lanes.size());  // the lanes' solver chains overlap: each keeps to its share of the device
  QTR_HIP_TRY(h, solver_enqueue_group(SB.data(), (int)SB.size(), srcs.data(), tgts.data(), Ls.data(), J.prm, &ln.stage,
                                      lead.stream));
  ln.phase = 3;
  return QTR_OK;
}

static int lane_start_chunk(qtr_handle* h, Lane& ln) {
  BatchJob& J = h->job;
  if (J.next >= J.B) {
    ln.phase = 0;
    ln.count = 0;
    return QTR_OK;
  }
  ln.first_pair = J.next;
  ln.count = min(ln.cap, J.B - J.next);
  J.next += ln.count;
  ln.stage.off = 0;
  ln.active.clear();
  ln.corr_only.clear();
  ln.ns.assign((size_t)ln.count, 0);
  ln.nt.assign((size_t)ln.count, 0);
  ln.L.assign((size_t)ln.count, 0);
  ln.csrc.assign((size_t)ln.count, nullptr);
  ln.ctgt.assign((size_t)ln.count, nullptr);
  ln.raw_s.assign((size_t)ln.count, nullptr);
  ln.raw_t.assign((size_t)ln.count, nullptr);
  ln.Ps.assign((size_t)ln.count, 0);
  ln.Pt.assign((size_t)ln.count, 0);
  Slot& lead = h->slots[ln.first_slot];
  std::vector<FrontBufs*> F;
  std::vector<const float4*> raws;
  std::vector<int> Ps;
  // The demo's STEP 2 and 3 on raw sweeps (reference examples/run_global_registration.cpp:136-160): per scan
  // PatchWork::estimate_ground -> non-ground points -> ImageProjection::segmentCloud -> valid segments.  Both stages hand a
  // count to the next one through the host, so a scan is a chain of four host-visible steps — run for ALL pairs of the
  // chunk side by side, every pair on its own slot's stream and arenas (target first, then source: the two scans of a pair
  // share the slot), the host advancing whichever slot's step has finished.  (Rounds 2-3 ran them pair after pair with a
  // blocking read-back per step: a chunk of sixteen raw pairs waited 64 times.)  The valid segments land in the slot's
  // staging buffers, which are the voxel grid's input; the lane's stream waits for the copies, not the host.
  struct PreScan {
    int stage = 0;  // 0 start, 1 ground segmentation of scan c in flight, 2 range-image stage in flight, 3 done
    int c = 1;      // scan being processed: 1 target, then 0 source
    int status = QTR_OK;
    int n[2] = {0, 0};  // valid points of source / target
  };
  std::vector<PreScan> pre((size_t)ln.count);
  if (h->pre_on) {
    int open = 0;
    for (int g = 0; g < ln.count; ++g) {
      const qtr_pair_desc& pd = J.pairs[ln.first_pair + g];
      const bool usable = pair_has_scans(pd) && pd.src_raw4 && pd.tgt_raw4 && pd.n_src > 0 && pd.n_tgt > 0 &&
                          pd.n_src <= h->lim.max_points && pd.n_tgt <= h->lim.max_points;
      if (!usable) pre[(size_t)g].stage = 3;  // (refused by the checks below, or a pair without scans)
      else ++open;
    }
    unsigned long spins = 0;
    while (open > 0) {
      bool moved = false;
      for (int g = 0; g < ln.count; ++g) {
        PreScan& q = pre[(size_t)g];
        if (q.stage == 3) continue;
        const qtr_pair_desc& pd = J.pairs[ln.first_pair + g];
        Slot& s = h->slots[ln.first_slot + g];
        int rc = QTR_OK;
        if (q.stage == 0) {
          rc = pw_begin(h, s, q.c ? pd.tgt_raw4 : pd.src_raw4, q.c ? pd.n_tgt : pd.n_src, &h->pre_pw, J.mem);
          q.stage = 1;
          moved = true;
        } else {
          const hipError_t e = hipEventQuery(s.ev[1]);
          if (e == hipErrorNotReady) continue;
          if (e != hipSuccess) {
            snprintf(h->err, sizeof(h->err), "batch pre-processing: %s", hipGetErrorString(e));
            return QTR_ERR_HIP;
          }
          moved = true;
          if (q.stage == 1) {  // ground removed: its non-ground points go through the range image
            int ng = 0, nn = 0;
            pw_end(s, &ng, &nn);
            rc = seg_begin(h, s, (const float*)s.pwb.out_n, nn, &h->pre_ip, QTR_MEM_DEVICE);
            q.stage = 2;
          } else {  // valid segments known: into the staging buffer; next scan, or done
            const int nv = s.pinned_i32[0];
            q.n[q.c] = nv;
            if (nv > h->lim.max_points) rc = QTR_ERR_CAPACITY;
            else if (nv > 0)
              // (on the slot's own stream: the next scan's stages, which reuse the buffers this copy reads, queue up behind it)
              QTR_HIP_TRY(h, hipMemcpyAsync(q.c ? s.in_tgt : s.in_src, s.seg.out_valid, (size_t)nv * 16, hipMemcpyDeviceToDevice,
                                            s.stream));
            if (rc == QTR_OK && q.c == 1) {
              q.c = 0;
              q.stage = 0;
            } else {
              if (rc == QTR_OK) {  // the group's chain (lane stream) reads the staging buffers: it waits for this slot
                QTR_HIP_TRY(h, hipEventRecord(s.ev[1], s.stream));
                QTR_HIP_TRY(h, hipStreamWaitEvent(lead.stream, s.ev[1], 0));
              }
              q.stage = 3;
              --open;
            }
          }
        }
        if (rc == QTR_ERR_HIP) return rc;
        if (rc != QTR_OK && q.stage != 3) {  // this pair's own failure (capacity, a scan the stage refuses)
          q.status = rc;
          q.stage = 3;
          --open;
        } else if (rc != QTR_OK) {
          q.status = rc;
        }
      }
      if (!moved) {
        __builtin_ia32_pause();
        if ((++spins & 0xffffff) == 0) {  // nothing moved for a long while: a lost device must not hang the host
          for (int g = 0; g < ln.count; ++g)
            if (pre[(size_t)g].stage != 3) QTR_HIP_TRY(h, hipStreamSynchronize(h->slots[ln.first_slot + g].stream));
        }
      } else {
        spins = 0;
      }
    }
  }
  for (int g = 0; g < ln.count; ++g) {
    const qtr_pair_desc& pd = J.pairs[ln.first_pair + g];
    Slot& s = h->slots[ln.first_slot + g];
    qtr_result& r = J.results[ln.first_pair + g];
    memset(&r, 0, sizeof(r));
    const bool scans = pair_has_scans(pd), corr = pair_has_corr(pd);
    if ((!scans && !corr) || (scans && (pd.n_src <= 0 || pd.n_tgt <= 0 || !pd.src_raw4 || !pd.tgt_raw4)) ||
        (corr && (pd.n_corr < 0 || !pd.src_corr4 || !pd.tgt_corr4))) {
      batch_fail_pair(h, ln.first_pair + g, QTR_ERR_BAD_ARG);
      continue;
    }
    if ((scans && (pd.n_src > h->lim.max_points || pd.n_tgt > h->lim.max_points)) || (corr && pd.n_corr > h->lim.max_corr)) {
      batch_fail_pair(h, ln.first_pair + g, QTR_ERR_CAPACITY);
      continue;
    }
    s.times_pending = 0;
    if (corr) {
      r.n_corr = pd.n_corr;
      ln.L[g] = pd.n_corr;
    }
    if (!scans) {  // correspondences only: nothing to do before the solver chain
      if (J.mem == QTR_MEM_HOST) {
        if (pd.n_corr > 0) {
          QTR_HIP_TRY(h, hipMemcpyAsync(s.m_src, pd.src_corr4, (size_t)pd.n_corr * 16, hipMemcpyHostToDevice, lead.stream));
          QTR_HIP_TRY(h, hipMemcpyAsync(s.m_tgt, pd.tgt_corr4, (size_t)pd.n_corr * 16, hipMemcpyHostToDevice, lead.stream));
        }
        ln.csrc[g] = s.m_src;
        ln.ctgt[g] = s.m_tgt;
      } else {
        ln.csrc[g] = (const float4*)pd.src_corr4;
        ln.ctgt[g] = (const float4*)pd.tgt_corr4;
      }
      ln.corr_only.push_back(g);
      continue;
    }
    const float4 *d_s = (const float4*)pd.src_raw4, *d_t = (const float4*)pd.tgt_raw4;
    int P_s = pd.n_src, P_t = pd.n_tgt;
    if (h->pre_on) {
      // (the sweeps were pre-processed above, all slots of the chunk side by side: the valid segments are in the slot's
      // staging buffers, the lane's stream already waits for the copies)
      const PreScan& q = pre[(size_t)g];
      if (q.status == QTR_OK && (q.n[0] <= 0 || q.n[1] <= 0)) {  // nothing but ground: an empty cloud
        batch_fail_pair(h, ln.first_pair + g, QTR_ERR_BAD_ARG);
        continue;
      }
      if (q.status != QTR_OK) {
        batch_fail_pair(h, ln.first_pair + g, q.status);
        continue;
      }
      P_s = q.n[0];
      P_t = q.n[1];
      d_s = s.in_src;
      d_t = s.in_tgt;
    } else if (J.mem == QTR_MEM_HOST) {
      QTR_HIP_TRY(h, hipMemcpyAsync(s.in_src, pd.src_raw4, (size_t)pd.n_src * 16, hipMemcpyHostToDevice, lead.stream));
      QTR_HIP_TRY(h, hipMemcpyAsync(s.in_tgt, pd.tgt_raw4, (size_t)pd.n_tgt * 16, hipMemcpyHostToDevice, lead.stream));
      d_s = s.in_src;
      d_t = s.in_tgt;
    }
    s.fb.mail_seq = ++s.seq;
    ln.raw_s[g] = d_s;  // (device pointers: what the voxel grid of this pair reads)
    ln.raw_t[g] = d_t;
    ln.Ps[g] = P_s;
    ln.Pt[g] = P_t;
    ln.active.push_back(g);
    F.push_back(&s.fb);
    raws.push_back(d_s);
    raws.push_back(d_t);
    Ps.push_back(P_s);
    Ps.push_back(P_t);
  }
  if (ln.active.empty()) return lane_enqueue_solver(h, ln, {});  // no scans in this chunk (or nothing valid at all)
  QTR_HIP_TRY(h, voxelize_enqueue_group(F.data(), (int)F.size(), raws.data(), Ps.data(), J.fp.voxel_size, &ln.stage,
                                        lead.stream, J.fp.fpfh_radius * 1.001f));
  QTR_HIP_TRY(h, hipEventRecord(lead.ev_vox, lead.stream));
  ln.phase = 1;
  return QTR_OK;
}

// advances the lane by at most one chain; *progress is set when it did
static int lane_poll(qtr_handle* h, Lane& ln, bool* progress) {
  BatchJob& J = h->job;
  if (ln.phase == 0) return QTR_OK;
  Slot& lead = h->slots[ln.first_slot];
  if (ln.phase == 1) {
    for (int g : ln.active) {
      Slot& s = h->slots[ln.first_slot + g];
      if (!mail_ready(s, MAIL_SEQ_VOX0, s.seq) || !mail_ready(s, MAIL_SEQ_VOX1, s.seq)) return QTR_OK;
    }
    *progress = true;
    bool passed_through = false;
    std::vector<int> keep;
    std::vector<FrontBufs*> F;
    std::vector<int> n2;
    std::vector<unsigned long long> seeds;
    for (int g : ln.active) {
      Slot& s = h->slots[ln.first_slot + g];
      qtr_result& r = J.results[ln.first_pair + g];
      int ns = s.mail[MAIL_VOX0 + CNT_NVOX], nt = s.mail[MAIL_VOX1 + CNT_NVOX];
      if (ns < 0 || nt < 0) {  // the centroid kernel's look-back timed out for this pair (see front_device)
        batch_fail_pair(h, ln.first_pair + g, QTR_ERR_HIP);
        continue;
      }
      // (a grid that would overflow int32 passes its cloud through, as pcl::VoxelGrid does: see front_device)
      const bool pass_s = s.mail[MAIL_VOX0 + CNT_VOX_OVERFLOW] != 0, pass_t = s.mail[MAIL_VOX1 + CNT_VOX_OVERFLOW] != 0;
      if (pass_s) ns = ln.Ps[g];
      if (pass_t) nt = ln.Pt[g];
      if (pass_s && ns <= h->lim.max_voxels) {
        QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[0].vox, ln.raw_s[g], (size_t)ns * 16, hipMemcpyDeviceToDevice, lead.stream));
        passed_through = true;
      }
      if (pass_t && nt <= h->lim.max_voxels) {
        QTR_HIP_TRY(h, hipMemcpyAsync(s.fb.cloud[1].vox, ln.raw_t[g], (size_t)nt * 16, hipMemcpyDeviceToDevice, lead.stream));
        passed_through = true;
      }
      r.n_src = ns;
      r.n_tgt = nt;
      if (ns > h->lim.max_voxels || nt > h->lim.max_voxels || ns <= 0 || nt <= 0) {
        batch_fail_pair(h, ln.first_pair + g, QTR_ERR_CAPACITY);
        continue;
      }
      ln.ns[g] = ns;
      ln.nt[g] = nt;
      s.last_ns = ns;
      s.last_nt = nt;
      keep.push_back(g);
      F.push_back(&s.fb);
      n2.push_back(ns);
      n2.push_back(nt);
      seeds.push_back(J.pairs[ln.first_pair + g].seed);
    }
    ln.active.swap(keep);
    if (ln.active.empty()) return lane_enqueue_solver(h, ln, {});
    const int G = (int)F.size();
    if (passed_through) QTR_HIP_TRY(h, hipEventRecord(lead.ev_vox, lead.stream));
    QTR_HIP_TRY(h, hipStreamWaitEvent(lead.stream2, lead.ev_vox, 0));
    QTR_HIP_TRY(h, mean_enqueue_group(F.data(), G, n2.data(), &ln.stage, lead.stream2));  // beside the FPFH chain
    QTR_HIP_TRY(h, hipEventRecord(lead.ev[5], lead.stream2));
    ln.long_lists = h->long_lists;
    if (ln.long_lists)
      for (int g : ln.active) QTR_TRY(ensure_long_arenas(h, h->slots[ln.first_slot + g]));
    int max_ncell = 1;  // the largest neighbour grid of the group (0: some pair's does not fit the dense cell table)
    for (int g : ln.active) {
      const Slot& sg = h->slots[ln.first_slot + g];
      const int nc2 = cell_table_cells(sg.mail[MAIL_VOX0 + CNT_NCELL], sg.mail[MAIL_VOX1 + CNT_NCELL]);
      max_ncell = (max_ncell == 0 || nc2 == 0) ? 0 : std::max(max_ncell, nc2);
    }
    QTR_HIP_TRY(h, fpfh_enqueue_group(F.data(), G, n2.data(), J.fp.normal_radius, J.fp.fpfh_radius, &ln.stage, lead.stream,
                                      ln.long_lists, true, max_ncell));
    QTR_HIP_TRY(h, hipStreamWaitEvent(lead.stream, lead.ev[5], 0));
    for (int g : ln.active) {
      Slot& s = h->slots[ln.first_slot + g];
      s.fb.mail_seq = ++s.seq;
    }
    QTR_HIP_TRY(h, match_enqueue_group(F.data(), G, n2.data(), &J.fp, seeds.data(), &ln.stage, lead.stream, true));
    ln.phase = 2;
    return QTR_OK;
  }
  if (ln.phase == 2) {
    for (int g : ln.active) {
      Slot& s = h->slots[ln.first_slot + g];
      if (!mail_ready(s, MAIL_SEQ_MATCH, s.seq)) return QTR_OK;
    }
    *progress = true;
    std::vector<int> keep;
    bool drained = false;  // lead.stream has been synchronised by a fallback below
    for (int g : ln.active) {
      Slot& s = h->slots[ln.first_slot + g];
      qtr_result& r = J.results[ln.first_pair + g];
      const qtr_pair_desc& pd = J.pairs[ln.first_pair + g];
      const bool given = pair_has_corr(pd);  // the back end runs on the caller's correspondences, not the matcher's
      const int Lm = s.mail[MAIL_MATCH + MC_NCORR];
      if (!given) r.n_corr = Lm;
      if (Lm < 0 && !given) {  // the tail's look-back timed out (match.hip): nothing usable was written for this pair
        r.n_corr = 0;
        batch_fail_pair(h, ln.first_pair + g, QTR_ERR_HIP);
        continue;
      }
      if (s.mail[MAIL_CNT0 + CNT_VOX_TAILERR] || s.mail[MAIL_CNT1 + CNT_VOX_TAILERR]) {  // (see front_device)
        batch_fail_pair(h, ln.first_pair + g, QTR_ERR_HIP);
        continue;
      }
      // (a pair that brought its correspondences does not read the matcher's list: its length is no reason to fail it)
      if (s.mail[MAIL_CNT0 + CNT_NBR_CAPACITY] || s.mail[MAIL_CNT1 + CNT_NBR_CAPACITY] ||
          (!given && Lm > h->lim.max_corr)) {
        batch_fail_pair(h, ln.first_pair + g, QTR_ERR_CAPACITY);
        continue;
      }
      if (!ln.long_lists && (s.mail[MAIL_CNT0 + CNT_NBR_OVERFLOW] || s.mail[MAIL_CNT1 + CNT_NBR_OVERFLOW])) {
        // A point with more than QTR_KMAX neighbours and a chain without k2_neighbors_big (see front_device): this pair
        // goes through the per-pair path on its own slot, and the handle's later chains include the launch.  The
        // group's chain (lane stream) may still be writing this slot's lists — the mail is published before the
        // kernel's last workgroup is done — so the lane stream is drained first; the pair is registered on the clouds
        // the group's voxel grid read (the pre-processed sweeps when qtr_set_batch_preprocess is on, the staged copies
        // of host scans: device pointers either way), and on the caller's correspondences when it brought some.
        h->long_lists = true;
        if (!drained) QTR_HIP_TRY(h, hipStreamSynchronize(lead.stream));
        drained = true;
        qtr_frontend_params f1 = J.fp;
        f1.seed = pd.seed;
        const float4 *cs = nullptr, *ct = nullptr;
        if (given) {
          cs = (const float4*)pd.src_corr4;
          ct = (const float4*)pd.tgt_corr4;
          if (J.mem == QTR_MEM_HOST) {  // (front_device overwrites m_src / m_tgt: the copy goes behind it, see below)
            cs = s.m_src;
            ct = s.m_tgt;
          }
        }
        int rc1;
        if (given && J.mem == QTR_MEM_HOST) {
          // front end first, then the caller's correspondences into the (now free) matched-cloud buffers, then the back end
          int L1 = 0;
          rc1 = front_device(h, s, (const float*)ln.raw_s[g], ln.Ps[g], (const float*)ln.raw_t[g], ln.Pt[g], &f1,
                             QTR_MEM_DEVICE, true, &r.n_src, &r.n_tgt, &L1, true);
          if (rc1 == QTR_OK) {
            if (pd.n_corr > 0) {
              QTR_HIP_TRY(h, hipMemcpyAsync(s.m_src, pd.src_corr4, (size_t)pd.n_corr * 16, hipMemcpyHostToDevice, s.stream));
              QTR_HIP_TRY(h, hipMemcpyAsync(s.m_tgt, pd.tgt_corr4, (size_t)pd.n_corr * 16, hipMemcpyHostToDevice, s.stream));
            }
            r.n_corr = pd.n_corr;
            rc1 = solve_device(h, s, s.m_src, s.m_tgt, pd.n_corr, &J.prm, &r, true);
            if (rc1 == QTR_OK || rc1 == QTR_ERR_CLIQUE_TOO_SMALL) {
              const int rc2 = copy_out_lists(h, s, &r, pd.clique, nullptr, pd.final_inliers, pd.cap, J.mem);
              if (rc2 != QTR_OK) r.status = rc1 = rc2;
            }
          } else {
            r.status = rc1;
          }
        } else {
          rc1 = register_pair_impl(h, s, (const float*)ln.raw_s[g], ln.Ps[g], (const float*)ln.raw_t[g], ln.Pt[g], &f1, &J.prm,
                                   &r, pd.clique, pd.final_inliers, pd.cap, QTR_MEM_DEVICE, J.mem, cs, ct,
                                   given ? pd.n_corr : -1);
        }
        if (rc1 == QTR_ERR_HIP) return rc1;
        J.finished[ln.first_pair + g] = 1;
        ++J.done;
        continue;
      }
      if (!given) QTR_HIP_TRY(h, gather_matched_enqueue(s.fb, Lm, s.m_src, s.m_tgt, lead.stream));  // no-op after the fused tail
      if (given) {
        if (J.mem == QTR_MEM_HOST) {  // behind the matching chain on the lane's stream: the matched clouds are not needed
          if (pd.n_corr > 0) {
            QTR_HIP_TRY(h, hipMemcpyAsync(s.m_src, pd.src_corr4, (size_t)pd.n_corr * 16, hipMemcpyHostToDevice, lead.stream));
            QTR_HIP_TRY(h, hipMemcpyAsync(s.m_tgt, pd.tgt_corr4, (size_t)pd.n_corr * 16, hipMemcpyHostToDevice, lead.stream));
          }
          ln.csrc[g] = s.m_src;
          ln.ctgt[g] = s.m_tgt;
        } else {
          ln.csrc[g] = (const float4*)pd.src_corr4;
          ln.ctgt[g] = (const float4*)pd.tgt_corr4;
        }
        ln.L[g] = pd.n_corr;
      } else {
        ln.csrc[g] = s.m_src;
        ln.ctgt[g] = s.m_tgt;
        ln.L[g] = Lm;
      }
      keep.push_back(g);
    }
    return lane_enqueue_solver(h, ln, keep);
  }
  // phase 3
  for (int g : ln.active) {
    Slot& s = h->slots[ln.first_slot + g];
    if (!mail_ready(s, MAIL_SEQ_SOLVE, s.seq)) return QTR_OK;
  }
  *progress = true;
  bool copies = false;
  for (int g : ln.active) {
    Slot& s = h->slots[ln.first_slot + g];
    const int pair = ln.first_pair + g;
    qtr_result& r = J.results[pair];
    const qtr_pair_desc& pd = J.pairs[pair];
    const int L = ln.L[g];
    const float4 *c_src = ln.csrc[g], *c_tgt = ln.ctgt[g];
    int rc = QTR_OK;
    // Follow-up work of ONE pair of the group runs on the lane's stream (the lane's chain owns the slot's arenas), and
    // everything that waits for it — wait_mail's liveness check, exact_phase's enqueues — must look at THAT stream:
    // the slot's own stream is idle, so a wait that queries it gives up before the kernels have run.
    struct StreamSwap {
      Slot& s;
      hipStream_t keep;
      StreamSwap(Slot& s_, hipStream_t st) : s(s_), keep(s_.stream) { s.stream = st; }
      ~StreamSwap() { s.stream = keep; }
    } on_lane_stream(s, lead.stream);
    if (L > 0 && !((const SolverState*)(s.mail + MAIL_SOLVER + 64))->done) {  // rare: more clique rounds needed
      s.sb.mail_seq = ++s.seq;
      QTR_HIP_TRY(h, solver_continue(s.sb, c_src, c_tgt, L, J.prm, lead.stream, s.pinned_i32 + 128,
                                     ((const SolverState*)(s.mail + MAIL_SOLVER + 64))->redo_cores));
      QTR_TRY(wait_mail(h, s, MAIL_SEQ_SOLVE, s.seq));
    }
    if (L > 0 && J.prm.inlier_selection_mode == QTR_INLIER_PMC_EXACT) {
      SolverState hs;
      memcpy(&hs, s.mail + MAIL_SOLVER + 64, sizeof(hs));
      bool improved = false;
      rc = exact_phase(h, s, L, hs, &improved, J.prm.max_clique_time_limit);
      if (rc != QTR_OK) return rc;
      if (improved) {
        s.sb.mail_seq = ++s.seq;
        QTR_HIP_TRY(h, solver_refinalize(s.sb, c_src, c_tgt, L, J.prm, lead.stream));
        QTR_TRY(wait_mail(h, s, MAIL_SEQ_SOLVE, s.seq));
      }
    }
    const int keep_ns = r.n_src, keep_nt = r.n_tgt, keep_nc = r.n_corr;
    memcpy(&r, s.mail + MAIL_SOLVER, sizeof(qtr_result));
    r.n_src = keep_ns;
    r.n_tgt = keep_nt;
    r.n_corr = keep_nc;
    const hipMemcpyKind kind = (J.mem == QTR_MEM_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (pd.clique && r.n_clique > 0) {
      if (r.n_clique > pd.cap) r.status = QTR_ERR_CAPACITY;
      else {
        QTR_HIP_TRY(h, hipMemcpyAsync(pd.clique, s.sb.clique, sizeof(int) * (size_t)r.n_clique, kind, lead.stream));
        copies = true;
      }
    }
    if (pd.final_inliers && r.n_final > 0) {
      if (r.n_final > pd.cap) r.status = QTR_ERR_CAPACITY;
      else {
        QTR_HIP_TRY(h, hipMemcpyAsync(pd.final_inliers, s.sb.final_inl, sizeof(int) * (size_t)r.n_final, kind, lead.stream));
        copies = true;
      }
    }
    J.finished[pair] = 1;
    ++J.done;
  }
  if (copies) QTR_HIP_TRY(h, hipStreamSynchronize(lead.stream));
  return lane_start_chunk(h, ln);
}

int qtr_set_batch_preprocess(qtr_handle* h, const qtr_pw_params* pw, const qtr_ip_params* ip) {
  if (!h || (pw == nullptr) != (ip == nullptr)) return QTR_ERR_BAD_ARG;
  if (h->job.active) {
    snprintf(h->err, sizeof(h->err), "a batch is in flight on this handle (call qtr_wait first)");
    return QTR_ERR_BAD_ARG;
  }
  h->pre_on = pw != nullptr;
  if (pw) {
    h->pre_pw = *pw;
    h->pre_ip = *ip;
  }
  return QTR_OK;
}

int qtr_submit_batch(qtr_handle* h, const qtr_pair_desc* pairs, int B, const qtr_frontend_params* fp,
                     const qtr_params* prm, qtr_result* results, int mem) {
  if (!h) return QTR_ERR_BAD_ARG;
  if (h->job.active) {
    snprintf(h->err, sizeof(h->err), "a batch is already in flight on this handle (call qtr_wait first)");
    return QTR_ERR_BAD_ARG;
  }
  if (B < 0 || (B > 0 && (!pairs || !results)) || !fp) {
    snprintf(h->err, sizeof(h->err), "bad batch arguments");
    return QTR_ERR_BAD_ARG;
  }
  const int rc = check_params(h, prm);
  if (rc != QTR_OK) return rc;
  if (fp->normal_radius > fp->fpfh_radius) {
    snprintf(h->err, sizeof(h->err), "[FPFHManager]: Normal should be lower than fpfh_radius!!!!");
    return QTR_ERR_BAD_ARG;
  }
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  BatchJob& J = h->job;
  J.pairs = pairs;
  J.B = B;
  J.next = 0;
  J.done = 0;
  J.fp = *fp;
  J.prm = *prm;
  J.results = results;
  J.mem = mem;
  J.active = true;
  J.finished.assign((size_t)B, 0);
  for (int i = 0; i < B; ++i) {  // until a pair's record is final it says so (a caller that looks only at the
    memset(&results[i], 0, sizeof(qtr_result));  // per-pair status can tell "never ran" from "ran, no solution")
    results[i].status = QTR_ERR_NOT_RUN;
  }
  for (auto& ln : h->lanes) {
    const int r = lane_start_chunk(h, ln);
    if (r != QTR_OK) {
      batch_abort(h);  // earlier lanes may already have work in flight
      return r;
    }
  }
  return QTR_OK;
}

int qtr_wait(qtr_handle* h) {
  if (!h) return QTR_ERR_BAD_ARG;
  BatchJob& J = h->job;
  if (!J.active) return QTR_OK;
  (void)hipSetDevice(h->device);
  int rc = QTR_OK;
  unsigned long idle = 0;
  while (true) {
    bool any_active = false, progress = false;
    for (auto& ln : h->lanes) {
      if (ln.phase == 0) continue;
      any_active = true;
      rc = lane_poll(h, ln, &progress);
      if (rc != QTR_OK) break;
    }
    if (rc != QTR_OK || !any_active) break;
    if (progress) {
      idle = 0;
      continue;
    }
    __builtin_ia32_pause();
    if ((++idle & 0xfffff) == 0) {  // nothing moved for a long while: has a launch failed, is the device gone?
      for (auto& ln : h->lanes) {
        if (ln.phase == 0) continue;
        Slot& lead = h->slots[ln.first_slot];
        const hipError_t q = hipStreamQuery(lead.stream);
        if (q == hipSuccess) {  // stream drained, yet a mailbox never arrived
          bool all = true;
          for (int g : ln.active) {
            Slot& s = h->slots[ln.first_slot + g];
            const int idx = ln.phase == 1 ? MAIL_SEQ_VOX1 : ln.phase == 2 ? MAIL_SEQ_MATCH : MAIL_SEQ_SOLVE;
            all = all && mail_ready(s, idx, s.seq);
          }
          if (!all) {
            snprintf(h->err, sizeof(h->err), "batch lane drained its stream without publishing phase %d", ln.phase);
            rc = QTR_ERR_HIP;
          }
        } else if (q != hipErrorNotReady) {
          snprintf(h->err, sizeof(h->err), "batch lane: %s", hipGetErrorString(q));
          rc = QTR_ERR_HIP;
        }
        (void)hipGetLastError();
      }
      if (rc != QTR_OK) break;
    }
  }
  if (rc != QTR_OK) batch_abort(h);  // leave the handle usable, every record says what happened to its pair
  J.active = false;
  return rc;
}

// ------------------------------------------------------------------------------------------------
long long qtr_debug_fetch(qtr_handle* h, int slot, int what, void* dst, size_t bytes) {
  Slot* sp = get_slot(h, slot);
  if (!sp) return -1;
  Slot& s = *sp;
  if (hipSetDevice(h->device) != hipSuccess) return -1;
  const void* src = nullptr;
  size_t have = 0;
  const int L = s.last_L, W = (L + 63) / 64;
  CloudBufs& c0 = s.fb.cloud[0];
  switch (what) {
    case QTR_DBG_GRAPH_BITMAP: src = s.sb.bm; have = (size_t)L * W * 8; break;
    case QTR_DBG_CORE: src = s.sb.core; have = (size_t)L * 4; break;
    case QTR_DBG_PERM: src = s.sb.perm; have = (size_t)L * 4; break;
    case QTR_DBG_NBR_OFFSETS:
      // the CSR view is only built for inspection
      if (exclusive_scan_i32(c0.nbr_cnt, c0.nbr_off, s.last_n, s.stream) != hipSuccess) return -1;
      src = c0.nbr_off;
      have = (size_t)(s.last_n + 1) * 4;
      break;
    case QTR_DBG_NBR_INDEX:
    case QTR_DBG_NBR_DIST2: {
      int tot = 0;
      if (exclusive_scan_i32(c0.nbr_cnt, c0.nbr_off, s.last_n, s.stream) != hipSuccess) return -1;
      if (hipStreamSynchronize(s.stream) != hipSuccess) return -1;
      if (hipMemcpy(&tot, c0.nbr_off + s.last_n, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
      src = (what == QTR_DBG_NBR_INDEX) ? (const void*)c0.nbr_idx : (const void*)c0.nbr_d2;
      have = (size_t)tot * 4;
      break;
    }
    case QTR_DBG_SPFH: src = c0.spfh; have = (size_t)s.last_n * 132; break;
    case QTR_DBG_NN_LARGE_OF_SMALL: src = s.fb.nn_of_small; have = (size_t)(s.last_ns < s.last_nt ? s.last_ns : s.last_nt) * 4; break;
    case QTR_DBG_NN_SMALL_OF_LARGE: src = s.fb.nn_of_large; have = (size_t)(s.last_ns < s.last_nt ? s.last_nt : s.last_ns) * 4; break;
    case QTR_DBG_VOX_SRC: src = s.fb.cloud[0].vox; have = (size_t)s.last_ns * 16; break;
    case QTR_DBG_VOX_TGT: src = s.fb.cloud[1].vox; have = (size_t)s.last_nt * 16; break;
    case QTR_DBG_CORR: src = s.fb.corr; have = (size_t)s.last_L * 8; break;
    case QTR_DBG_MATCH_STATS: src = s.fb.mcounts; have = 16 * 4; break;
    case QTR_DBG_SOLVER_STATE: src = s.sb.st; have = sizeof(SolverState); break;
    default: return -1;
  }
  const size_t n = have < bytes ? have : bytes;
  if (dst && n > 0) {
    if (hipStreamSynchronize(s.stream) != hipSuccess) return -1;
    if (what == QTR_DBG_GRAPH_BITMAP && s.last_Wb > W) {  // rows start on 32-byte boundaries on the device: pack them
      const size_t rows = n / ((size_t)W * 8);
      if (rows > 0 && hipMemcpy2D(dst, (size_t)W * 8, src, (size_t)s.last_Wb * 8, (size_t)W * 8, rows, hipMemcpyDeviceToHost) !=
                          hipSuccess)
        return -1;
    } else if (hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) {
      return -1;
    }
  }
  return (long long)have;
}

__global__ void k_debug_math(int fn, const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r = 0.f, s, c;
  switch (fn) {
    case 0: r = qm_atan2f(a[i], b[i]); break;
    case 1: r = qm_acosf(a[i]); break;
    case 2: qm_sincosf(a[i], &s, &c); r = s; break;
    case 3: qm_sincosf(a[i], &s, &c); r = c; break;
    case 5: r = spfh_swap_roles(a[i], b[i]) ? 1.f : 0.f; break;  // the SPFH kernel's shortcut (tests compare it with the
                                                                 // plain arithmetic)
    default: break;
  }
  out[i] = r;
}

int qtr_debug_math(qtr_handle* h, int fn, const float* a, const float* b, float* out, int n) {
  if (!h || !a || !out || n <= 0) return QTR_ERR_BAD_ARG;
  QTR_HIP_TRY(h, hipSetDevice(h->device));
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  QTR_HIP_TRY(h, hipMalloc((void**)&da, (size_t)n * 4));
  QTR_HIP_TRY(h, hipMalloc((void**)&db, (size_t)n * 4));
  QTR_HIP_TRY(h, hipMalloc((void**)&dout, (size_t)n * 4));
  QTR_HIP_TRY(h, hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
  QTR_HIP_TRY(h, hipMemcpy(db, b ? b : a, (size_t)n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, 0, fn, da, db, dout, n);
  QTR_HIP_TRY(h, hipGetLastError());
  QTR_HIP_TRY(h, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  return QTR_OK;
}

}  // extern "C"
