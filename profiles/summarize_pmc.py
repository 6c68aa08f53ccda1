"""Per-launch HBM-side traffic of one kernel from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
rocpd sqlite output).  usage: python profiles/summarize_pmc.py fetch.db write.db 'void k_nn_f16' > profiles/<tag>_pmc_nn.json
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes
of wide coalesced streaming reads (16 bytes per lane) and is to be doubled for them; other widths are uncalibrated.
k_nn_f16 reads its operand tables with global_load_dwordx4 (16 B per lane, 512 contiguous bytes per half-wave): doubled.
k_nn_mfma (the f32 engine) issues 4-byte-per-lane loads: reported raw and flagged as such."""
import json
import sqlite3
import sys


def mean(db, counter, pat):
    c = sqlite3.connect(db)
    r = c.execute("select count(*), avg(value), min(value), max(value) from counters_collection "
                  "where counter_name = ? and kernel_name like ?", (counter, pat + "%")).fetchone()
    return dict(launches=r[0], mean_kib=r[1], min_kib=r[2], max_kib=r[3])


fetch_db, write_db, pat = sys.argv[1], sys.argv[2], sys.argv[3]
f, w = mean(fetch_db, "FETCH_SIZE", pat), mean(write_db, "WRITE_SIZE", pat)
wide = "k_nn_f16" in pat
out = dict(kernel=pat, fetch=f, write=w,
           traffic_bytes_per_launch=((2.0 if wide else 1.0) * f["mean_kib"] + w["mean_kib"]) * 1024.0,
           correction=("FETCH_SIZE x 2 (16 B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE raw" if wide else
                       "none applied (dword loads; FETCH_SIZE x2 rule is calibrated for 16 B/lane reads only)"),
           command="rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 8 --warmup 2 "
                   "--legs '' --cpu-seconds 0 (two separate passes; profiles/collect_r3.sh)")
print(json.dumps(out, indent=1))
