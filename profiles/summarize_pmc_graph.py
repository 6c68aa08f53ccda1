"""k_graph_build's memory-side counters from the --pmc passes of profiles/collect_r3.sh (rocpd sqlite):
FETCH_SIZE / WRITE_SIZE (KiB per launch, raw — the kernel reads 16-byte point records through the scalar-friendly
path and writes 8-byte words; the guide's x2 rule is calibrated for 16 B/lane streaming reads only) and the L2 hit rate.
usage: python profiles/summarize_pmc_graph.py TAG > profiles/TAG_pmc_graph.json"""
import glob
import json
import sqlite3
import sys

tag = sys.argv[1]
PAT = "void k_graph_build%"


def mean(dbdir, counter):
    f = sorted(glob.glob(f"gpurun_out/prof_{tag}_{dbdir}/*.db"))
    if not f:
        return None
    c = sqlite3.connect(f[0])
    r = c.execute("select count(*), avg(value) from counters_collection where counter_name = ? and kernel_name like ?",
                  (counter, PAT)).fetchone()
    return {"launches": r[0], "mean": r[1]}


out = {"kernel": "k_graph_build", "note": "algorithmic bytes: 32 L in + L^2 / 8 bit matrix + L^2 / 64 degree bytes out"}
for L in (5000, 20000):
    f, w = mean(f"gf{L}", "FETCH_SIZE"), mean(f"gw{L}", "WRITE_SIZE")
    h, m = mean(f"gh{L}", "TCC_HIT_sum"), mean(f"gh{L}", "TCC_MISS_sum")
    e = {"algorithmic_bytes": 32.0 * L + L * L / 8.0 + L * L / 64.0}
    if f and f["mean"] is not None:
        e["fetch_kib_per_launch"] = f["mean"]
    if w and w["mean"] is not None:
        e["write_kib_per_launch"] = w["mean"]
    if f and w and f["mean"] is not None and w["mean"] is not None:
        e["traffic_bytes_per_launch"] = (f["mean"] + w["mean"]) * 1024.0
    if h and m and h["mean"] is not None and m["mean"] is not None and (h["mean"] + m["mean"]) > 0:
        e["l2_hit_rate"] = h["mean"] / (h["mean"] + m["mean"])
    out[f"L{L}"] = e
print(json.dumps(out, indent=1))
