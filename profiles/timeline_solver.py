"""Like timeline.py for a trace of the back end alone (tests/gpu_solver_prof.py): one solve = from a k_graph_build* dispatch to
the next.   usage: python profiles/timeline_solver.py trace.db [which solve, default 10]"""
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
first = [i for i, r in enumerate(rows) if r[0].startswith("void k_graph_build")]
a, b = first[which], first[which + 1]
busy_until, t0 = rows[a - 1][2], rows[a][1]
print(f"# {db}: solve {which} ({b - a} dispatches)")
print(f"{'start_us':>9} {'dur_us':>7} {'gap_us':>7} stream kernel")
for name, start, end, stream in rows[a:b]:
    print(f"{(start - t0) / 1e3:9.1f} {(end - start) / 1e3:7.1f} {(start - busy_until) / 1e3:7.1f} s{stream:<5} {name[:70]}")
    busy_until = max(busy_until, end)
print(f"# next solve starts {(rows[b][1] - t0) / 1e3:.1f} us after this one")
