// teaser/graph.h — drop-in for url-kaist/Quatro's include/teaser/graph.h: the same two classes,
// teaser::Graph (:29-211) and teaser::MaxCliqueSolver (:219-274), with findMaxClique (src/graph.cc:12-104,
// which calls the PMC library) replaced by one call into libquatro_hip.so (qtr_max_clique: k-core peeling,
// rank relabelling and the greedy clique heuristic as gfx950 kernels on a bit-matrix graph).
//
// Written against the reference's public surface, not its implementation: Graph keeps plain adjacency
// lists like the reference so user code that walks getEdges() keeps working; the bit matrix the device
// wants is assembled once inside findMaxClique.
//
// Differences a caller can observe:
//   * PMC_HEU and KCORE_HEU follow the documented deterministic semantics (divergences D2/D3:
//     sequential PMC heuristic, canonical (core, id) vertex order).  PMC_EXACT returns the heuristic's clique when it
//     is maximum, otherwise the first maximum clique of the canonical depth-first order (D10) — PMC's threads race
//     for the incumbent, so the reference does not define which maximum clique comes back.
//   * the returned ids are in ascending order (PMC returns them in search order; Quatro sorts them itself,
//     include/quatro.hpp:805).
//   * time_limit bounds the exact search only; when it is hit the heuristic's clique is returned.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <vector>

#include "../quatro_hip_cxx.hpp"

namespace teaser {

class Graph {
 public:
  Graph() : num_edges_(0) {}

  // adjacency-list constructor; vertices numbered 0..N-1.  numEdges() then reports the number of listed
  // entries (an undirected edge listed from both ends counts twice), as the reference's :40-50 does.
  explicit Graph(const std::map<int, std::vector<int>>& adj_list) : num_edges_(0) {
    adj_list_.resize(adj_list.size());
    for (const auto& kv : adj_list) {
      if (kv.first < 0 || static_cast<size_t>(kv.first) >= adj_list_.size()) continue;
      adj_list_[static_cast<size_t>(kv.first)] = kv.second;
      num_edges_ += static_cast<int>(kv.second.size());
    }
  }

  void addVertex(const int& id) {
    if (id >= 0 && static_cast<size_t>(id) >= adj_list_.size()) adj_list_.resize(static_cast<size_t>(id) + 1);
  }
  void populateVertices(const int& num_vertices) { adj_list_.resize(static_cast<size_t>(num_vertices)); }

  bool hasVertex(const int& vertex) const { return vertex >= 0 && static_cast<size_t>(vertex) < adj_list_.size(); }
  bool hasEdge(const int& a, const int& b) const {
    if (!hasVertex(a) || !hasVertex(b)) return false;
    const std::vector<int>& e = adj_list_[static_cast<size_t>(a)];
    return std::find(e.begin(), e.end(), b) != e.end();
  }
  void addEdge(const int& a, const int& b) {
    if (!hasVertex(a) || !hasVertex(b) || hasEdge(a, b)) return;
    adj_list_[static_cast<size_t>(a)].push_back(b);
    adj_list_[static_cast<size_t>(b)].push_back(a);
    ++num_edges_;
  }
  void removeEdge(const int& a, const int& b) {
    if (!hasEdge(a, b)) return;
    std::vector<int>& ea = adj_list_[static_cast<size_t>(a)];
    std::vector<int>& eb = adj_list_[static_cast<size_t>(b)];
    ea.erase(std::remove(ea.begin(), ea.end(), b), ea.end());
    eb.erase(std::remove(eb.begin(), eb.end(), a), eb.end());
    --num_edges_;
  }

  int numVertices() const { return static_cast<int>(adj_list_.size()); }
  int numEdges() const { return num_edges_; }
  const std::vector<int>& getEdges(int id) const { return adj_list_[static_cast<size_t>(id)]; }
  std::vector<int> getVertices() const {
    std::vector<int> v(adj_list_.size());
    for (size_t i = 0; i < v.size(); ++i) v[i] = static_cast<int>(i);
    return v;
  }
  void reserve(const int& num_vertices) { adj_list_.reserve(static_cast<size_t>(num_vertices)); }
  void clear() {
    adj_list_.clear();
    num_edges_ = 0;
  }
  void reserveForCompleteGraph(const int& num_vertices) {
    adj_list_.clear();
    adj_list_.resize(static_cast<size_t>(num_vertices));
    for (auto& e : adj_list_) e.reserve(static_cast<size_t>(num_vertices > 0 ? num_vertices - 1 : 0));
  }

  // Row-major symmetric bit matrix, ceil(N/64) words per row: the layout qtr_max_clique takes.
  std::vector<unsigned long long> bitMatrix() const {
    const size_t N = adj_list_.size(), W = (N + 63) / 64;
    std::vector<unsigned long long> bm(N * W, 0ULL);
    for (size_t i = 0; i < N; ++i)
      for (int j : adj_list_[i]) {
        if (j < 0 || static_cast<size_t>(j) >= N || static_cast<size_t>(j) == i) continue;
        bm[i * W + (static_cast<size_t>(j) >> 6)] |= 1ULL << (j & 63);
        bm[static_cast<size_t>(j) * W + (i >> 6)] |= 1ULL << (i & 63);
      }
    return bm;
  }

 private:
  std::vector<std::vector<int>> adj_list_;
  int num_edges_;
};

class MaxCliqueSolver {
 public:
  enum class CLIQUE_SOLVER_MODE { PMC_EXACT = 0, PMC_HEU = 1, KCORE_HEU = 2 };

  struct Params {
    CLIQUE_SOLVER_MODE solver_mode = CLIQUE_SOLVER_MODE::PMC_EXACT;
    bool solve_exactly = true;  // deprecated in the reference: false forces PMC_HEU (src/graph.cc:15-17)
    double kcore_heuristic_threshold = 1;
    double time_limit = 3600;
  };

  MaxCliqueSolver() = default;
  explicit MaxCliqueSolver(Params params) : params_(params) {}

  std::vector<int> findMaxClique(const Graph& graph) {
    const int N = graph.numVertices();
    std::vector<int> clique(static_cast<size_t>(N > 0 ? N : 1));
    if (N == 0) return {};
    if (!params_.solve_exactly) params_.solver_mode = CLIQUE_SOLVER_MODE::PMC_HEU;  // src/graph.cc:15-17
    int mode = QTR_INLIER_PMC_EXACT;
    if (params_.solver_mode == CLIQUE_SOLVER_MODE::PMC_HEU) mode = QTR_INLIER_PMC_HEU;
    if (params_.solver_mode == CLIQUE_SOLVER_MODE::KCORE_HEU) mode = QTR_INLIER_KCORE_HEU;
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    const std::vector<unsigned long long> bm = graph.bitMatrix();
    int n = 0, max_core = 0;
    quatro_hip::check(h, qtr_max_clique(h, slot_lease.slot, bm.data(), N, mode, params_.kcore_heuristic_threshold, params_.time_limit,
                                        clique.data(),
                                        static_cast<int>(clique.size()), &n, &max_core, QTR_MEM_HOST));
    clique.resize(static_cast<size_t>(n));
    max_core_ = max_core;
    return clique;
  }
  int lastMaxCore() const { return max_core_; }

 private:
  Params params_;
  int max_core_ = 0;
};

}  // namespace teaser
