// solver.h — device state and buffer carving for the back-end kernels (solver.hip).
#pragma once
#include "common.h"
#include "frontend.h"

#define CLIQUE_BATCH 1024

struct SolverState {  // device resident; mirrored to pinned host memory between clique rounds
  int mc;        // best clique size so far (pmc_heu's `mc`)
  int best_r;    // rank of the start vertex that produced it (-1 none, -2 member bitset prebuilt)
  int pos;       // rank of the next start vertex (descending)
  int done;
  int t0;        // lowest rank whose Kp exceeds mc
  int ub;        // max_core + 1  (reference src/graph.cc:84-86)
  int batch;     // starts evaluated by the current k_clique_batch
  int max_core;
  int n_edges2;  // sum of degrees
  int rounds;
  int pad[19];   // [0] k-core peeling rounds, [1..4] k_finalize phase clocks / 16, [6..11] COTE step clocks / 16,
                 // [12] 1: the clique stage ran twice (second time with exact core numbers, see solver_continue)
  int core_floor;  // k_hcore_async stopped lowering values below this (0: every core number is exact); the clique search then
                   // runs with this many members as an injected lower bound, see k_rank_sort
  int tainted;     // ... and a start was turned down by the |P| > mc rule while the bound was the injected one
  int redo_cores;  // the search under the injected bound found nothing it can vouch for: the host runs the stage again, exact
};

struct SolverBufs {
  int Lcap = 0;
  u64* bm = nullptr;    // adjacency, original labels  [L][W]
  u64* adjP = nullptr;  // adjacency, rank labels      [L][W]
  int *deg = nullptr, *core = nullptr, *perm = nullptr, *rankof = nullptr, *Kp = nullptr, *picks = nullptr,
      *gsz = nullptr;
  int *clique = nullptr, *rot_inl = nullptr, *final_inl = nullptr;
  double* f64 = nullptr;
  int* i32 = nullptr;
  u64* member_bits = nullptr;
  int* picks_buf = nullptr;  // [CLIQUE_BATCH][L] greedy picks of every start of the current batch
  SolverState* st = nullptr;
  qtr_result* res = nullptr;
  // [Lcap + 1] doubles: [0] the COTE range the table was made for, [n] = that range added up n times in sequence (the
  // reference's sum of N ranges, include/quatro.hpp:660, for every N: the host lays the table down when the range changes
  // — solver_enqueue* — so that k_finalize need not run the chain of N dependent additions); range_rg: the host's copy of [0]
  double* range_pre = nullptr;
  mutable double range_rg = -1.0;
  int* mail = nullptr;  // device view of the slot's pinned host mailbox (frontend.h MAIL_*), or null
  int mail_seq = 0;     // sequence number the next k_finalize / k_clique_only publishes
};

// What the back-end kernels need to know about one pair; picked with blockIdx.z (one pair travels in the kernel
// arguments, a group of pairs sits in device memory — see ViewExt in common.h).
struct SolverView {
  const float4* src;  // matched keypoint clouds (L each)
  const float4* tgt;
  int L, W;           // correspondences, words per bit-matrix row
  int Wb;             // row stride of bm in words (>= W; a multiple of four for matrices built by k_graph_build)
  u64* bm;
  u64* adjP;
  int *deg, *core, *perm, *rankof, *Kp, *picks, *gsz, *clique, *rot_inl, *final_inl;
  double* f64;
  int* i32;
  u64* member_bits;
  int* picks_buf;
  SolverState* st;
  qtr_result* res;
  int* mail;
  int seq;
  // degrees as k_graph_build leaves them: one byte per (64-column block k, vertex v) = popcount of word k of row v, at
  // degp[k * Lp + v] (every entry has exactly one writer: no atomics); null when V.deg already holds the degrees
  // (qtr_max_clique).  Readers go through solver_degree().
  const unsigned char* degp;
  int Lp;
  const double* range_pre;  // SolverBufs::range_pre
};
struct SolverArgs {
  SolverView one;
  const SolverView* ext;
};

size_t solver_scratch_bytes(int Lcap);
void solver_carve(SolverBufs& B, void* base, int Lcap);
// reset_done: solver_reset_enqueue already cleared the state for this run (on another stream, off the critical path)
hipError_t solver_enqueue(const SolverBufs& B, const float4* src, const float4* tgt, int L, const qtr_params& prm,
                          hipStream_t stream, int* pinned_state, hipEvent_t ev_graph, hipEvent_t ev_clique,
                          bool reset_done = false);
hipError_t solver_reset_enqueue(const SolverBufs& B, hipStream_t stream);
// how many back-end launch chains may run side by side on the device with the calling thread's next enqueue (1: it has
// the device to itself): bounds k_hcore_async's resident workgroups per pair (thread-local, see solver.hip)
void solver_set_hca_share(int share);
hipError_t solver_init_attributes();
// the same for G pairs at once (qtr_submit_batch): B[g] is pair g's arena, views go through `stage`
hipError_t solver_enqueue_group(SolverBufs* const* B, int G, const float4* const* src, const float4* const* tgt,
                                const int* L, const qtr_params& prm, ViewStage* stage, hipStream_t stream);
hipError_t solver_continue(const SolverBufs& B, const float4* src, const float4* tgt, int L, const qtr_params& prm,
                           hipStream_t stream, int* pinned_state, int redo_cores);
hipError_t solver_refinalize(const SolverBufs& B, const float4* src, const float4* tgt, int L, const qtr_params& prm,
                             hipStream_t stream);
// clique search alone (qtr_max_clique): enqueue, [solver_continue(src = nullptr) if !done], finish
hipError_t clique_only_enqueue(const SolverBufs& B, const u64* d_adj, int L, int mode, double kcore_thr,
                               hipStream_t stream);
hipError_t clique_only_finish(const SolverBufs& B, int L, hipStream_t stream);
