"""Timing of the exact clique search on the GPU box: python tests/gpu_exact_perf.py (prints size, nodes, seconds)."""
import sys
import time

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from quatro_amd import lib as ql  # noqa: E402
from test_gpu_parity import _random_graph_bitmap  # noqa: E402

h = ql.Handle(0)
h.set_clique_time_limit(20.0)
for a in [(100, 0.9, 0), (200, 0.7, 0), (300, 0.5, 0), (500, 0.3, 0), (1000, 0.2, 0), (2500, 0.1, 0)]:
    bm, A = _random_graph_bitmap(a[0], a[1], 5, a[2])
    h.max_clique(bm, 1)
    t = time.time()
    got, _ = h.max_clique(bm, 0)
    dt = time.time() - t
    st = h.exact_stats()
    print(a, got.size, st, "%.3fs" % dt, flush=True)
