"""One registration of a rocprofv3 kernel trace (rocpd sqlite), dispatch by dispatch: start relative to the first kernel,
duration, idle gap since everything before it had ended (negative: it overlaps an earlier dispatch on another stream),
stream, kernel.   usage: python profiles/timeline.py gpurun_out/prof_x/x_results.db [which registration, default 30]
The profiler slows the host's launches down, so the gaps at the three host synchronisation points (before the first
k2_keys_hist of the FPFH chain, before k_graph_build, before the next k2_minmax) are longer than in an unprofiled run."""
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
first = [i for i, r in enumerate(rows) if r[0].startswith("void k2_minmax")]
a, b = first[which], first[which + 1]
busy_until, t0 = rows[a - 1][2], rows[a][1]
print(f"# {db}: registration {which} ({b - a} dispatches)")
print(f"{'start_us':>9} {'dur_us':>7} {'gap_us':>7} stream kernel")
for name, start, end, stream in rows[a:b]:
    print(f"{(start - t0) / 1e3:9.1f} {(end - start) / 1e3:7.1f} {(start - busy_until) / 1e3:7.1f} s{stream:<5} {name[:70]}")
    busy_until = max(busy_until, end)
print(f"# next registration starts {(rows[b][1] - t0) / 1e3:.1f} us after this one")
