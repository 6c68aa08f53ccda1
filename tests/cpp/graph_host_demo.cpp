// graph_host_demo.cpp — the host-only part of include/teaser/graph.h (teaser::Graph, reference include/teaser/graph.h:29-211):
// reads "N" then edge operations "+ a b" / "- a b" from stdin, prints numVertices, numEdges, every adjacency list and
// the bit matrix the device entry point takes.  No GPU call is made.
#include <cstdio>

#include "teaser/graph.h"

int main() {
  int n = 0;
  if (std::scanf("%d", &n) != 1) return 2;
  teaser::Graph g;
  g.populateVertices(n);
  char op;
  int a, b;
  while (std::scanf(" %c %d %d", &op, &a, &b) == 3) {
    if (op == '+') g.addEdge(a, b);
    if (op == '-') g.removeEdge(a, b);
  }
  std::printf("%d %d\n", g.numVertices(), g.numEdges());
  for (int v = 0; v < g.numVertices(); ++v) {
    std::printf("%d:", v);
    for (int u : g.getEdges(v)) std::printf(" %d", u);
    std::printf("\n");
  }
  const auto bm = g.bitMatrix();
  for (unsigned long long w : bm) std::printf("%llx\n", w);
  std::printf("has %d %d %d\n", g.hasEdge(0, 1) ? 1 : 0, g.hasEdge(-1, 0) ? 1 : 0, g.hasVertex(n) ? 1 : 0);
  return 0;
}
