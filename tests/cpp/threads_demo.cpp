// threads_demo.cpp — independent teaser::MaxCliqueSolver objects driven from several threads at once: every wrapper
// call leases its own stream slot of the process-wide handle (include/quatro_hip_cxx.hpp: SlotLease), so the objects
// proceed side by side and each thread gets the answers it gets alone.
// usage: threads_demo edges.txt n_threads repeats      (edges.txt: first line N, then one "a b" per line)
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "teaser/graph.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::FILE* f = std::fopen(argv[1], "r");
  if (!f) return 3;
  int N = 0;
  if (std::fscanf(f, "%d", &N) != 1) return 4;
  std::vector<std::pair<int, int>> edges;
  int a, b;
  while (std::fscanf(f, "%d %d", &a, &b) == 2) edges.emplace_back(a, b);
  std::fclose(f);
  const int T = std::atoi(argv[2]), reps = std::atoi(argv[3]);
  // thread t solves the graph without its last t edges (so the threads' answers differ), in modes 1 / 2 alternating
  auto solve = [&](int t, int mode) {
    teaser::Graph g;
    g.populateVertices(N);
    for (size_t e = 0; e + static_cast<size_t>(t) < edges.size(); ++e) g.addEdge(edges[e].first, edges[e].second);
    teaser::MaxCliqueSolver::Params p;
    p.solver_mode = static_cast<teaser::MaxCliqueSolver::CLIQUE_SOLVER_MODE>(mode);
    p.kcore_heuristic_threshold = 0.0;
    teaser::MaxCliqueSolver solver(p);
    return solver.findMaxClique(g);
  };
  std::vector<std::vector<int>> alone(static_cast<size_t>(T));
  for (int t = 0; t < T; ++t) alone[static_cast<size_t>(t)] = solve(t, 1 + (t & 1));
  std::atomic<int> bad(0);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      for (int r = 0; r < reps; ++r)
        if (solve(t, 1 + (t & 1)) != alone[static_cast<size_t>(t)]) ++bad;
    });
  for (auto& x : th) x.join();
  std::printf("threads %d repeats %d mismatches %d slots %d\n", T, reps, bad.load(), quatro_hip::default_slot_count());
  return bad.load() ? 1 : 0;
}
