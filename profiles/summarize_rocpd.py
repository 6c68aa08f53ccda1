"""Turns a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary committed under profiles/.
usage: python profiles/summarize_rocpd.py gpurun_out/prof_x/x_results.db [steps] > profiles/x_kernel_stats.txt"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
# steps: a number, or "auto" = the registrations the trace holds (every registration opens with one k2_minmax launch)
steps = 0
if len(sys.argv) > 2:
    steps = (c.execute("select count(*) from kernels where name like 'void k2_minmax%'").fetchone()[0]
             if sys.argv[2] == "auto" else int(sys.argv[2]))
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace summary of {db}")
print(f"# total kernel time {tot:.1f} us over {sum(r[1] for r in rows)} dispatches" +
      (f"; {steps} registrations -> {tot / steps:.1f} us of kernel time per registration" if steps else ""))
print(f"{'kernel':<64} {'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}")
for r in rows:
    print(f"{r[0][:64]:<64} {r[1]:>6} {r[2]:>11.1f} {r[3]:>9.2f} {r[4]:>9.2f} {r[5]:>9.2f} {100 * r[2] / tot:>5.1f}%")
