"""profiles/r6_pmc_fpfh.json: the counters of the FPFH chain's kernels (and of k_graph_build / k_recheck_filter, which round 5's
verdict asked for as well) BEFORE and AFTER round 6's instruction cuts, with the bound each kernel is stated to have.
usage: python profiles/make_r6_fpfh.py BEFORE_seq.json AFTER_seq.json BEFORE_dense.json AFTER_dense.json > profiles/r6_pmc_fpfh.json
(the inputs are profiles/pmc_all.sh summaries: profiles/r6_pmc_seq_before.json etc.)"""
import json
import sys

NOTES = {
    "k2_ranges": "latency: nine pairs of dependent binary searches per point over an L2-resident key array (85 % of the wave-cycles "
                 "parked); round 6 advances a pair's two searches together — no change in the trace: the searches were not what it waits for",
    "k2_neighbors": "instruction issue: one wave per point, as many scalar as vector instructions before round 6 (the bitonic network's "
                    "loop control); SIMDs 65 % busy with vector instructions alone.  Round 6: rank sort for lists of up to 64 entries",
    "k2_normals": "latency of ONE wave per SIMD: ~1350 dependent vector instructions of pcl::eigen33 per point behind four round trips; "
                  "half a wave per SIMD (occupancy 0.4), nothing to overlap with",
    "k2_spfh": "vector-instruction count: binary64 software arc tangent + five IEEE divisions per (point, neighbour) pair; SIMDs 56 % busy "
               "on a single pair's launch, 95 % in the batched path.  Round 6: the arc tangent left the common path",
    "k2_fpfh": "vector + scalar issue: ~20 instructions per (point, neighbour, bin) value before round 6, eight of them scalar bookkeeping "
               "of two guards; SIMDs 68 % busy, 1.8 TB/s of gathered SPFH rows.  Round 6: branch-free, ~9 per value",
    "k_graph_build": "vector-instruction count: SIMDs 98 % busy at 17 instructions per 64 predicates (strips, L > 8192); WRITE_SIZE 2.26x "
                     "the matrix (the transposed words are 8-byte stores)",
    "k_graph_build_tiles": "vector-instruction count at 24 per 64 predicates (SIMDs 78 % busy: a single pair's graph is one wave of "
                           "short workgroups); WRITE_SIZE 3.5x the matrix",
    "k_recheck_filter": "before round 6: latency — one dependent chain of seven MFMAs per 32 x 32 tile behind the tile's L2 round trip "
                        "(1455 clocks per tile against 224 of matrix pipe, occupancy 1.3 waves per SIMD).  Round 6: 128 listed rows per "
                        "wave above 2048 listed rows (two chains interleaved, sign-bit masks): 2870 clocks per 4096 entries",
}


def pick(j, names):
    out = {}
    for k, v in j["kernels"].items():
        base = k.split("<")[0]
        if base in names:
            d, du = v["derived"], v["duration"] or {}
            out[base] = {
                "launches": du.get("launches"), "mean_us": round(du.get("mean_us", 0.0), 2),
                "wave_cycle_split": {a: round(b, 3) for a, b in d.get("wave_cycle_split", {}).items()},
                "lds_issue_stall": round(d.get("lds_issue_stall", 0.0), 4),
                "simd_valu_busy": round(d.get("simd_valu_busy", 0.0), 3),
                "occupancy_waves_per_simd": round(d.get("occupancy_waves_per_simd", 0.0), 2),
                "per_wave": {"valu": round(d.get("valu_insts_per_wave", 0.0)), "salu": round(d.get("salu_insts_per_wave", 0.0)),
                             "lds": round(d.get("lds_insts_per_wave", 0.0)), "vmem_rd": round(d.get("vmem_rd_per_wave", 0.0)),
                             "vmem_wr": round(d.get("vmem_wr_per_wave", 0.0))},
                "lds_bank_conflict_share": round(d.get("lds_bank_conflict_share", 0.0), 3),
                "fetch_mb_raw": round(d.get("fetch_bytes_raw", 0.0) / 1e6, 2), "write_mb": round(d.get("write_bytes", 0.0) / 1e6, 2),
                "gbytes_per_s_fetch_x2": round(d.get("gbytes_per_s_fetch_x2", 0.0)), "l2_hit_rate": round(d.get("l2_hit_rate", 0.0), 3)}
    return out


b_seq, a_seq, b_den, a_den = (json.load(open(p)) for p in sys.argv[1:5])
chain = ["k2_ranges", "k2_neighbors", "k2_normals", "k2_spfh", "k2_fpfh", "k_graph_build_tiles", "k_recheck_filter"]
dense = ["k2_neighbors", "k2_spfh", "k2_fpfh", "k_graph_build", "k_recheck_filter"]
out = {
    "what": "SQ / TCC counters of the FPFH chain's kernels, k_graph_build and k_recheck_filter (separate rocprofv3 --pmc passes beside "
            "--kernel-trace only, profiles/pmc_all.sh; durations from a clean trace of the same command), before and after round 6's "
            "cuts, with the bound each is stated to have",
    "units": "wave_cycle_split: shares of SQ_WAVE_CYCLES (issuing / issue_stall / parked); simd_valu_busy: 4 x SQ_ACTIVE_INST_VALU / "
             "(1024 SIMDs x launch cycles from SQ_BUSY_CYCLES / 32); per_wave: instructions per wavefront; FETCH raw (gfx950 reports half "
             "of 16-byte-per-lane streaming reads: gbytes_per_s uses x2 as the upper estimate)",
    "headline_loop": {"before": {"kernel_source": b_seq.get("csrc_sha"), "kernels": pick(b_seq, chain)},
                      "after": {"kernel_source": a_seq.get("csrc_sha"), "kernels": pick(a_seq, chain)}},
    "dense_step": {"before": {"kernel_source": b_den.get("csrc_sha"), "kernels": pick(b_den, dense)},
                   "after": {"kernel_source": a_den.get("csrc_sha"), "kernels": pick(a_den, dense)}},
    "stated_bound": NOTES,
}
print(json.dumps(out, indent=1))
