"""profiles/r6_pmc_nn.json (HBM-side traffic of k_nn_f16) and profiles/r6_pmc_mfma.json (its matrix-pipe counters) from a collection of
profiles/collect_r6.sh:   python profiles/make_r6_nn.py gpurun_out/TAG
inputs: TAG/pmc_seq.json (profiles/pmc_all.sh: every kernel of the headline loop, separate --pmc passes) and TAG/mfma.txt
(profiles/pmc_mfma.sh: raw means per launch).  bench.py quotes both files in its roofline object, with the kernel-source hash beside."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_counters import source_sha  # noqa: E402

tag_dir = sys.argv[1]
tag = os.path.basename(os.path.normpath(tag_dir))
nn_sha, _ = source_sha()
here = os.path.dirname(os.path.abspath(__file__))

seq = json.load(open(os.path.join(tag_dir, "pmc_seq.json")))
k = next(v for name, v in seq["kernels"].items() if name.startswith("k_nn_f16"))
m = k["per_launch_means"]
fetch_kib, write_kib = m["FETCH_SIZE"], m["WRITE_SIZE"]
nn = {"kernel": "void k_nn_f16", "kernel_source_sha": nn_sha, "launches": k["duration"]["launches"],
      "mean_launch_us": k["duration"]["mean_us"], "fetch": {"mean_kib": fetch_kib}, "write": {"mean_kib": write_kib},
      "traffic_bytes_per_launch": round((2.0 * fetch_kib + write_kib) * 1024.0),
      "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]),
      "correction": "FETCH_SIZE x 2 (16 B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE raw",
      "command": f"bash profiles/pmc_all.sh {tag} seq (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum, "
                 "separate passes of: python bench.py --steps 8 --warmup 2 --legs '' --cpu-seconds 0)"}
json.dump(nn, open(os.path.join(here, "r6_pmc_nn.json"), "w"), indent=1)

raw = open(os.path.join(tag_dir, "mfma.txt")).read()
open(os.path.join(here, "r6_pmc_mfma_raw.txt"), "w").write(raw)
c, launches = {}, 0
for line in raw.splitlines():
    mm = re.match(r"m\d (\w+)\s+launches (\d+)\s+mean\s+([\d.]+)", line)
    if mm:
        c[mm.group(1)] = float(mm.group(3))
        launches = int(mm.group(2))
wc = c["SQ_WAVE_CYCLES"]
launch_cycles = c["SQ_BUSY_CYCLES"] / 32.0
mf = {"kernel": "void k_nn_f16", "kernel_source_sha": nn_sha, "launches": launches, "per_launch_means": c,
      "units": "SQ_VALU_MFMA_BUSY_CYCLES: cycles, 32 per v_mfma_f32_32x32x16_f16, summed over SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / "
               "SQ_ACTIVE_INST_*: quad-cycles; SQ_BUSY_CYCLES: summed over the 32 shader engines",
      "mfma_busy_of_wave_lifetime": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 4),
      "mfma_busy_of_sq_busy_time": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * launch_cycles), 4),
      "derivation": "one wave per SIMD (1024 waves on 1024 SIMDs): busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)",
      "wave_cycle_split": {"issuing": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3), "issue_stall": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                           "parked": round(c["SQ_WAIT_ANY"] / wc, 3)},
      "shader_clock_ghz_from_sq_busy": round(launch_cycles / (nn["mean_launch_us"] * 1e3), 3),
      "note": "the nearest-neighbour loop is round 5's (unchanged this round: the LDS-staged variant measured slower, "
              "profiles/r6_ab.txt section 5)",
      "command": f"bash profiles/pmc_mfma.sh {tag}"}
json.dump(mf, open(os.path.join(here, "r6_pmc_mfma.json"), "w"), indent=1)
print(json.dumps({"nn": {a: nn[a] for a in ("mean_launch_us", "traffic_bytes_per_launch", "l2_hit_rate")},
                  "mfma": {a: mf[a] for a in ("mfma_busy_of_wave_lifetime", "mfma_busy_of_sq_busy_time", "shader_clock_ghz_from_sq_busy")}}))
