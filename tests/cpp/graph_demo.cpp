// graph_demo.cpp — teaser::Graph + teaser::MaxCliqueSolver of THIS repository's include/teaser/graph.h used
// the way the reference's computeTransformation uses them (include/quatro.hpp:786-805): populateVertices,
// addEdge per consistent pair, findMaxClique.
// usage: graph_demo edges.txt mode thr     (edges.txt: first line N, then one "a b" per line)
#include <cstdio>
#include <cstdlib>

#include "teaser/graph.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::FILE* f = std::fopen(argv[1], "r");
  if (!f) return 3;
  int N = 0;
  if (std::fscanf(f, "%d", &N) != 1) return 4;
  teaser::Graph g;
  g.populateVertices(N);
  int a, b;
  while (std::fscanf(f, "%d %d", &a, &b) == 2) g.addEdge(a, b);
  std::fclose(f);
  teaser::MaxCliqueSolver::Params p;
  p.solver_mode = static_cast<teaser::MaxCliqueSolver::CLIQUE_SOLVER_MODE>(std::atoi(argv[2]));
  p.kcore_heuristic_threshold = std::atof(argv[3]);
  teaser::MaxCliqueSolver solver(p);
  try {
    const std::vector<int> c = solver.findMaxClique(g);
    std::printf("vertices %d edges %d max_core %d clique", g.numVertices(), g.numEdges(), solver.lastMaxCore());
    for (int v : c) std::printf(" %d", v);
    std::printf("\n");
  } catch (const std::invalid_argument& e) {
    std::printf("invalid_argument %s\n", e.what());
  }
  return 0;
}
