"""Per-kernel counter summary of a set of rocprofv3 --pmc passes (rocpd sqlite), one JSON for all kernels of a workload.
usage: python profiles/summarize_counters.py CLEAN_TRACE.db PASS_A.db PASS_B.db ... [--only PREFIX,PREFIX] > profiles/<tag>_pmc_<workload>.json

CLEAN_TRACE.db is a --kernel-trace run WITHOUT counters (durations under --pmc are inflated by the counter reads and
are not used for rates).  Units (MI355X_MICROARCH.md, PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over the 32 shader engines; FETCH_SIZE / WRITE_SIZE are KiB, and
gfx950's FETCH_SIZE reports half of the bytes of 16-byte-per-lane streaming reads (both the raw and the doubled figure are
given: most kernels here mix 4-, 8- and 16-byte loads, so the truth lies between them).

Derived per kernel (means per launch):
  wave_cycle_split   issuing / issue_stall / parked  = ACTIVE_INST_ANY / WAIT_INST_ANY / WAIT_ANY over SQ_WAVE_CYCLES
  valu_share         SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY       (what the issue slots are spent on)
  lds_share          SQ_ACTIVE_INST_LDS / SQ_ACTIVE_INST_ANY, lds_issue_stall = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  simd_valu_busy     4 x SQ_ACTIVE_INST_VALU quad-cycles ... / (1024 SIMDs x launch cycles): the launch's cycles come from
                     SQ_BUSY_CYCLES / 32 (the shader engines' busy time), i.e. the fraction of every SIMD's time a vector
                     instruction occupied it.  1.0 = bound by the COUNT of vector instructions
  occupancy_waves    SQ_WAVE_CYCLES x 4 / (1024 x launch cycles): mean resident waves per SIMD
  gbytes_per_s       (FETCH x {1,2} + WRITE) / clean duration
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("void ", "")
    i = n.find("(")
    return n[:i] if i > 0 else n


def source_sha():
    hh = hashlib.sha256()
    for f in ("nn_f16_core.inc", "gen_nn_f16_core.py", "match.hip"):
        with open(os.path.join(ROOT, "quatro_amd", "csrc", f), "rb") as fh:
            hh.update(fh.read())
    all_ = hashlib.sha256()
    d = os.path.join(ROOT, "quatro_amd", "csrc")
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            all_.update(fh.read())
    return hh.hexdigest()[:16], all_.hexdigest()[:16]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only"):
            only = a.split("=", 1)[1].split(",") if "=" in a else None
    clean, passes = args[0], args[1:]
    dur = {}
    c = sqlite3.connect(clean)
    for name, n, avg, mn, mx in c.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                                          "from kernels group by name"):
        dur[short(name)] = dict(launches=n, mean_us=avg, min_us=mn, max_us=mx)
    cnt = {}
    for p in passes:
        c = sqlite3.connect(p)
        try:
            rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                             "group by kernel_name, counter_name").fetchall()
        except Exception as e:  # a pass that failed leaves no table: say so instead of dying
            print(f"# {p}: {e}", file=sys.stderr)
            continue
        for k, ctr, n, v in rows:
            cnt.setdefault(short(k), {})[ctr] = v
            cnt[short(k)].setdefault("_launches", {})[ctr] = n
    out = {}
    for k in sorted(cnt, key=lambda k: -dur.get(k, {}).get("mean_us", 0) * dur.get(k, {}).get("launches", 0)):
        if only and not any(k.startswith(o) for o in only):
            continue
        v = cnt[k]
        d = dur.get(k)
        rec = dict(duration=d, per_launch_means={a: b for a, b in v.items() if not a.startswith("_")})
        g = v.get
        der = {}
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            der["wave_cycle_split"] = dict(issuing=(g("SQ_ACTIVE_INST_ANY") or 0) / wc, issue_stall=(g("SQ_WAIT_INST_ANY") or 0) / wc,
                                           parked=(g("SQ_WAIT_ANY") or 0) / wc)
            if g("SQ_WAIT_INST_LDS") is not None:
                der["lds_issue_stall"] = g("SQ_WAIT_INST_LDS") / wc
        if g("SQ_ACTIVE_INST_ANY"):
            der["valu_share"] = (g("SQ_ACTIVE_INST_VALU") or 0) / g("SQ_ACTIVE_INST_ANY")
            der["lds_share"] = (g("SQ_ACTIVE_INST_LDS") or 0) / g("SQ_ACTIVE_INST_ANY")
        if g("SQ_BUSY_CYCLES"):
            launch_cycles = g("SQ_BUSY_CYCLES") / 32.0
            der["launch_cycles_from_sq_busy"] = launch_cycles
            if g("SQ_ACTIVE_INST_VALU") is not None:
                der["simd_valu_busy"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * launch_cycles)
            if wc:
                der["occupancy_waves_per_simd"] = 4.0 * wc / (1024.0 * launch_cycles)
            if d:
                der["shader_clock_ghz"] = launch_cycles / (d["mean_us"] * 1e3)
        if g("SQ_INSTS_VALU") is not None and g("SQ_WAVES"):
            der["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
            der["lds_insts_per_wave"] = (g("SQ_INSTS_LDS") or 0) / g("SQ_WAVES")
            der["salu_insts_per_wave"] = (g("SQ_INSTS_SALU") or 0) / g("SQ_WAVES")
            der["vmem_rd_per_wave"] = (g("SQ_INSTS_VMEM_RD") or 0) / g("SQ_WAVES")
            der["vmem_wr_per_wave"] = (g("SQ_INSTS_VMEM_WR") or 0) / g("SQ_WAVES")
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            der["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        if g("FETCH_SIZE") is not None or g("WRITE_SIZE") is not None:
            f, w = (g("FETCH_SIZE") or 0) * 1024.0, (g("WRITE_SIZE") or 0) * 1024.0
            der["fetch_bytes_raw"], der["fetch_bytes_x2"], der["write_bytes"] = f, 2 * f, w
            if d:
                der["gbytes_per_s_raw"] = (f + w) / (d["mean_us"] * 1e-6) / 1e9
                der["gbytes_per_s_fetch_x2"] = (2 * f + w) / (d["mean_us"] * 1e-6) / 1e9
                der["frac_of_8TBs_fetch_x2"] = der["gbytes_per_s_fetch_x2"] / 8000.0
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
            der["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        rec["derived"] = der
        out[k] = rec
    nn_sha, all_sha = source_sha()
    print(json.dumps(dict(kernel_source_sha=nn_sha, csrc_sha=all_sha, clean_trace=os.path.basename(clean),
                          passes=[os.path.basename(p) for p in passes], kernels=out), indent=1))


if __name__ == "__main__":
    main()
