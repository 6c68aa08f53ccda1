"""Randomised GPU-vs-oracle sweep (run on the GPU box: python tests/gpu_fuzz.py [seed] [budget_s]).  Prints one line per
mismatch and a summary; exits 1 on any mismatch.  Not collected by pytest (no test_ prefix)."""
import faulthandler
import sys
import time

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")

import torch  # noqa: F401,E402
from oracle import oracle as qo  # noqa: E402
from quatro_amd import lib as ql  # noqa: E402
from quatro_amd import synth  # noqa: E402
from test_gpu_parity import _random_graph_bitmap  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
import os  # noqa: E402
DRY = os.environ.get("FUZZ_ORACLE_ONLY") == "1"  # exercise only the CPU half (checks that no case explodes on the host)


class _Dry:
    def __getattr__(self, name):
        def f(*a, **k):
            return None
        return f


h = _Dry() if DRY else ql.Handle(0)
if not DRY:
    h.set_clique_time_limit(5.0)
qo.build()
qo.set_threads(qo.max_threads())
hb = None  # a second handle with eight slots, for the batched cases
# FUZZ_KINDS=batch,solve: only these kinds (a sweep of the batched entry's mixed-size groups alone finds what one case in twelve of
# the full mix takes an hour to reach)
KINDS = ["solve", "solve", "clique", "pair", "match", "match", "patchwork", "segment", "gnc3", "cote", "batch", "scout"]
if os.environ.get("FUZZ_KINDS"):
    KINDS = [k for k in os.environ["FUZZ_KINDS"].split(",") if k in KINDS]
bad, n_cases, t_end = 0, 0, time.time() + budget
faulthandler.dump_traceback_later(int(budget) + 90, exit=True)  # a hung kernel must not eat the GPU budget


def report(kind, desc, what):
    global bad
    bad += 1
    print(f"MISMATCH {kind} {desc}: {what}", flush=True)


def same_solution(g, o):
    if g["status"] != o["status"] and not (g["status"] == ql.QTR_ERR_CLIQUE_TOO_SMALL and o["status"] == 1):
        return f"status {g['status']} vs {o['status']}"
    if g["valid"] != o["valid"]:
        return "valid"
    if not np.array_equal(g["clique"], np.sort(o["clique"])):
        return f"clique {g['clique'].size} vs {o['clique'].size}"
    if not g["valid"]:
        return None
    if g.get("rot_inliers") is not None and "rot_inliers" in o and not np.array_equal(g["rot_inliers"], o["rot_inliers"]):
        return "rot_inliers"
    if not np.array_equal(g["final_inliers"], o["final_inliers"]):
        return "final_inliers"
    if g["gnc_iters"] != o["gnc_iters"]:
        return "gnc_iters"
    if not np.array_equal(g["T"], o["T"]):
        return f"T max diff {np.abs(g['T'] - o['T']).max()}"
    return None


while time.time() < t_end:
    kind = rng.choice(KINDS)
    n_cases += 1
    print(f"case {n_cases} {kind}", file=sys.stderr, flush=True)
    try:
        if kind == "match":
            # descriptors that are NOT FPFH output: the f16-split filter's bound and range guard against the oracle's
            # exact matcher — clusters of near-identical rows, exact duplicates, sparse rows, values near and beyond the
            # f16 range, tiny values
            ns, nt = int(rng.integers(40, 2500)), int(rng.integers(40, 2500))
            ncl = int(rng.choice([3, 20, 200]))
            centres = rng.uniform(0, 1, (ncl, 33)) * (rng.random((ncl, 33)) < rng.choice([0.3, 1.0]))
            rel = float(rng.choice([0.0, 1e-7, 1e-5, 1e-2, 0.5]))
            scale = float(rng.choice([1e-4, 1.0, 100.0, 250.0, 600.0]))

            def draw(n):
                d = centres[rng.integers(0, ncl, n)] * (1.0 + rel * rng.standard_normal((n, 33)))
                return np.abs(d * scale).astype(np.float32)
            ds, dt = draw(ns), draw(nt)
            vs = np.zeros((ns, 4), np.float32)
            vt = np.zeros((nt, 4), np.float32)
            vs[:, :3] = rng.uniform(-30, 30, (ns, 3))
            vt[:, :3] = rng.uniform(-30, 30, (nt, 3))
            sd = int(rng.integers(0, 1000))
            desc = f"ns={ns} nt={nt} clusters={ncl} rel={rel} scale={scale} seed={sd}"
            co, nn_ij, nn_ji = qo.match(vs, ds, vt, dt, seed=sd, debug=True)
            cg = h.match(vs, ds, vt, dt, ql.default_frontend_params(seed=sd))
            if not DRY:
                g_ij = h.debug_fetch(ql.DBG_NN_LARGE_OF_SMALL, np.int32)[:nn_ij.size]
                if not np.array_equal(g_ij, nn_ij):
                    report(kind, desc, f"NN table differs in {int((g_ij != nn_ij).sum())} of {nn_ij.size} rows")
                elif not np.array_equal(cg, co):
                    report(kind, desc, f"correspondences {cg.shape[0]} vs {co.shape[0]}")
        elif kind == "batch":
            # correspondence-only pair descriptors through qtr_submit_batch: groups of mixed sizes (the group picks kernel
            # variants from its largest pair; above 8192 correspondences the strip form of the graph kernel), small noise
            # bounds (the binary32 screen narrows, then switches off), against the oracle pair by pair
            if hb is None and not DRY:
                hb = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=12288, n_slots=8)
            nb_ = float(rng.choice([0.3, 0.3, 0.05, 0.004]))
            sizes = [int(rng.choice([0, 2, 64, 65, 300, 1279, 1281, 2500, 5000, 8200, 9000])) for _ in range(int(rng.integers(2, 7)))]
            sets = []
            for L in sizes:
                if L == 0:
                    sets.append((np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32)))
                else:
                    c = synth.correspondences(max(L, 3), float(rng.choice([0.03, 0.2, 1.0])), seed=int(rng.integers(1 << 30)),
                                              noise=nb_ / 3)
                    sets.append((c[0][:L].copy(), c[1][:L].copy()))
            desc = f"sizes={sizes} noise_bound={nb_}"
            outs = [qo.solve(a_, b_, qo.default_params(noise_bound=nb_)) for a_, b_ in sets]
            if not DRY:
                got = hb.register_batch([(None, None, 0, a_, b_) for a_, b_ in sets], params=ql.demo_params(noise_bound=nb_))
                for i, (g, o) in enumerate(zip(got, outs)):
                    w = same_solution(dict(g, rot_inliers=None), o) if "clique" in g else f"status {g['status']}"
                    if w:
                        report(kind, desc + f" pair {i}", w)
                        dump = os.environ.get("FUZZ_DUMP_DIR")
                        if dump:  # the whole group, for a reproducer off the random stream
                            np.savez(os.path.join(dump, f"fuzz_batch_seed{seed}_case{n_cases}_pair{i}.npz"), noise_bound=nb_, pair=i,
                                     sizes=np.array(sizes), got_clique=g.get("clique", np.zeros(0, np.int32)),
                                     want_clique=np.sort(o["clique"]),
                                     **{f"src{j}": a_ for j, (a_, b_) in enumerate(sets)}, **{f"tgt{j}": b_ for j, (a_, b_) in enumerate(sets)})
        elif kind == "solve":
            L = int(rng.choice([2, 3, 5, 17, 64, 65, 300, 1281, 2000, 4097, 7000]))
            frac = float(rng.choice([0.0, 0.02, 0.1, 0.5, 0.9, 1.0]))
            noise = float(rng.choice([0.0, 0.02, 0.1]))
            kw = {}
            if rng.random() < 0.4:
                kw["inlier_selection_mode"] = int(rng.choice([0, 1, 2]))
            if rng.random() < 0.3:
                kw["reg_mode"] = 1
            if rng.random() < 0.3:
                kw["cote_median"] = 0
            if rng.random() < 0.3:
                kw["using_rot_inliers_when_estimating_cote"] = 1
            if rng.random() < 0.3:
                kw["noise_bound"] = float(rng.choice([0.05, 0.6]))
            if kw.get("inlier_selection_mode") == 0 and (L > 2000 or noise > 0.02 or frac > 0.5):
                kw["inlier_selection_mode"] = 1  # keep the exact search short (near-complete noisy graphs explode)
            src, tgt, _, _ = synth.correspondences(L, frac, seed=int(rng.integers(1 << 30)), noise=noise)
            desc = f"L={L} frac={frac} noise={noise} {kw}"
            g = h.solve(src, tgt, ql.demo_params(**kw))
            o = qo.solve(src, tgt, qo.default_params(**kw))
            w = same_solution(g, o)
            if w:
                report(kind, desc, w)
        elif kind == "clique":
            L = int(rng.choice([1, 2, 63, 64, 65, 129, 500, 1280, 1281, 2500, 5000]))
            p = float(rng.choice([0.0, 0.01, 0.05, 0.3, 0.9, 1.0]))
            if L > 600 and p > 0.3:
                p = 0.05
            planted = int(rng.choice([0, 0, 5, L // 10, L // 2])) if L > 10 else 0
            bm, A = _random_graph_bitmap(L, p, int(rng.integers(1 << 30)), planted)
            modes = [(1, 0.5), (2, 0.5), (2, 0.1)]
            if (L <= 300 and p <= 0.3) or p <= 0.02 or L <= 65:  # beyond that the search is exponential, not a parity case
                modes.append((0, 0.5))
            refs = {m: qo.max_clique(bm, m[0], m[1]) for m in modes}  # (all CPU work first: the dry mode covers it)
            for mode, thr in modes:
                ref = np.sort(refs[(mode, thr)])
                got, _ = h.max_clique(bm, mode, thr)
                if mode == 0 and h.exact_stats()["aborted"]:
                    print(f"note: exact search hit the 5 s limit on L={L} p={p} planted={planted}", flush=True)
                    continue
                if not np.array_equal(got, ref if mode != 2 else refs[(mode, thr)]):
                    report(kind, f"L={L} p={p} planted={planted} mode={mode} thr={thr}", f"{got.size} vs {ref.size}")
        elif kind == "scout":
            # graphs of more than 8192 vertices: a single pair's k_hcore_async launch carries the scout workgroup, whose
            # floor (a clique it found among the largest values) must never show in the result — the generator's
            # correspondences, and bit matrices with cliques / dense blocks / overlapping near-cliques in a sparse bulk
            if rng.random() < 0.5:
                L = int(rng.choice([8193, 9500, 12000]))
                frac = float(rng.choice([0.0, 0.005, 0.01, 0.02, 0.04]))
                noise = float(rng.choice([0.02, 0.1]))
                src, tgt, _, _ = synth.correspondences(L, frac, seed=int(rng.integers(1 << 30)), noise=noise)
                desc = f"L={L} frac={frac} noise={noise}"
                g = h.solve(src, tgt, ql.demo_params())
                o = qo.solve(src, tgt, qo.default_params())
                w = same_solution(g, o)
                if w:
                    report(kind, desc, w)
            else:
                L = int(rng.choice([8200, 9000]))
                p = float(rng.choice([0.003, 0.01, 0.02]))
                A = np.triu(rng.random((L, L), dtype=np.float32) < p, 1)
                desc = f"L={L} p={p}"
                for _ in range(int(rng.integers(1, 4))):
                    k = int(rng.choice([12, 20, 60, 150, 400]))
                    mem = rng.choice(L, k, replace=False)
                    q = float(rng.choice([1.0, 1.0, 0.97, 0.8, 0.5]))
                    A[np.ix_(mem, mem)] |= rng.random((k, k)) < q
                    desc += f" block({k}, {q})"
                A = np.triu(A, 1)
                A = A | A.T
                bits = np.zeros((L, ((L + 63) // 64) * 64), dtype=np.uint8)
                bits[:, :L] = A
                bm = np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, -1)
                ref = np.sort(qo.max_clique(bm, 1, 0.5))
                core, _, mc = qo.kcore(bm)
                got, max_core = h.max_clique(bm, 1, 0.5)
                if not np.array_equal(got, ref) or max_core != mc:
                    report(kind, desc, f"{got.size} vs {ref.size}, largest core {max_core} vs {mc}")
                elif not DRY:
                    core = np.asarray(core)
                    core_g = h.debug_fetch(ql.DBG_CORE, np.int32)[:L]
                    floor = int(h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)[29])
                    hi = core >= floor
                    if not (np.array_equal(core_g[hi], core[hi]) and np.all(core_g[~hi] >= core[~hi]) and np.all(core_g[~hi] < floor)):
                        report(kind, desc, f"core numbers under floor {floor}")
        elif kind == "pair":
            pid = int(rng.integers(0, 50))
            s, t, _ = synth.kitti64_pair(pid)
            keep = float(rng.choice([0.05, 0.3, 1.0]))
            s = s[rng.random(s.shape[0]) < keep]
            t = t[rng.random(t.shape[0]) < keep]
            leaf = float(rng.choice([0.3, 0.5, 1.0]))
            fp = ql.default_frontend_params(seed=pid, voxel_size=leaf, normal_radius=leaf * 5 / 3, fpfh_radius=leaf * 2.5)
            desc = f"pair={pid} keep={keep} leaf={leaf} n=({s.shape[0]},{t.shape[0]})"
            g = h.register_pair(s, t, fp)
            o = qo.register_pair(s, t, leaf=leaf, r_normal=leaf * 5 / 3, r_fpfh=leaf * 2.5, seed=pid)
            if (g["n_src"], g["n_tgt"], g["L"]) != (o["n_src"], o["n_tgt"], o["L"]):
                report(kind, desc, f"counts {(g['n_src'], g['n_tgt'], g['L'])} vs {(o['n_src'], o['n_tgt'], o['L'])}")
                # stage-level diagnosis on the same handle, in the state the failure happened in
                g2 = h.register_pair(s, t, fp)
                print("  again:", (g2["n_src"], g2["n_tgt"], g2["L"]), "stats", h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)[:16], flush=True)
                vs, vt = qo.voxelize(s, leaf), qo.voxelize(t, leaf)
                print("  vox equal", np.array_equal(h.voxelize(s, leaf), vs), np.array_equal(h.voxelize(t, leaf), vt), flush=True)
                _, _, do = qo.fpfh(vs, leaf * 5 / 3, leaf * 2.5)
                _, _, dto = qo.fpfh(vt, leaf * 5 / 3, leaf * 2.5)
                _, dg = h.fpfh(vs, leaf * 5 / 3, leaf * 2.5)
                _, dgt = h.fpfh(vt, leaf * 5 / 3, leaf * 2.5)
                print("  desc equal", np.array_equal(dg, do), np.array_equal(dgt, dto), flush=True)
                cg = h.match(vs, do, vt, dto, fp)
                co, nn_ij, nn_ji = qo.match(vs, do, vt, dto, seed=pid, debug=True)
                g_ij = h.debug_fetch(ql.DBG_NN_LARGE_OF_SMALL, np.int32)[:nn_ij.size]
                print("  match L", cg.shape[0], co.shape[0], "nn ij diff", int((g_ij != nn_ij).sum()), "of", nn_ij.size,
                      "stats", h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)[:16], flush=True)
                g3 = h.register_pair(s, t, fp)
                print("  third:", (g3["n_src"], g3["n_tgt"], g3["L"]), flush=True)
                h2 = ql.Handle(0)
                g4 = h2.register_pair(s, t, fp)
                print("  fresh handle:", (g4["n_src"], g4["n_tgt"], g4["L"]), flush=True)
                h2.close()
            else:
                w = same_solution(g, o)
                if w:
                    report(kind, desc, w)
        elif kind == "patchwork":
            sid = int(rng.integers(0, 50))
            xyzi, _ = synth.kitti64_raw_scan(sid)
            keep = float(rng.choice([0.02, 0.3, 1.0]))
            xyzi = xyzi[rng.random(xyzi.shape[0]) < keep]
            po, pg = qo.pw_params(), ql.pw_params()
            if rng.random() < 0.5:
                ni, nl, nm = int(rng.integers(1, 5)), int(rng.choice([1, 20, 300])), int(rng.choice([0, 10, 80]))
                td = float(rng.choice([0.05, 0.125, 0.3]))
                for p in (po, pg):
                    p.num_iter, p.num_lpr, p.num_min_pts, p.th_dist = ni, nl, nm, td
            a, b = h.patchwork(xyzi, pg), qo.patchwork(xyzi, po)
            if not (np.array_equal(a["ground"].view(np.uint32), b["ground"].view(np.uint32)) and
                    np.array_equal(a["nonground"].view(np.uint32), b["nonground"].view(np.uint32))):
                report(kind, f"scan={sid} keep={keep} it={pg.num_iter} lpr={pg.num_lpr} min={pg.num_min_pts}",
                       f"{a['ground'].shape[0]}/{a['nonground'].shape[0]} vs {b['ground'].shape[0]}/{b['nonground'].shape[0]}")
        elif kind == "segment":
            pid = int(rng.integers(0, 50))
            s, _, _ = synth.kitti64_pair(pid)
            keep = float(rng.choice([0.05, 0.5, 1.0]))
            s = s[rng.random(s.shape[0]) < keep]
            lidar = str(rng.choice(["Velodyne-64-HDE", "VLP-16", "HDL-32E", "Ouster-OS1-64"]))
            mode = str(rng.choice(["4Neighbor", "4CrossNeighbor", "8Neighbor"]))
            mp = int(rng.choice([5, 30, 100]))
            try:
                ip_o, ip_g = qo.ip_params(lidar, mode, mp), ql.ip_params(lidar, mode, mp)
            except Exception:
                continue
            a, b = h.segment_cloud(s, ip_g), qo.segment_cloud(s, ip_o)
            if not (np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["valid"], b["valid"]) and
                    np.array_equal(a["outliers"], b["outliers"])):
                report(kind, f"pair={pid} keep={keep} {lidar} {mode} {mp}", "labels/valid/outliers")
        elif kind == "gnc3":
            M = int(rng.choice([1, 2, 3, 64, 65, 500, 4000]))
            X = rng.uniform(-10, 10, (M, 3))
            Y = X + rng.normal(0, float(rng.choice([0.0, 0.05, 3.0])), (M, 3))
            nb = float(rng.choice([1e-9, 0.1, 0.6]))
            a, b = h.gnc_rotation3d(X, Y, nb, 1.4, 50, 1.1e-4), qo.gnc_rotation3d(X, Y, nb, 1.4, 50, 1.1e-4)
            if not (np.array_equal(a[0], b[0]) and a[2] == b[2] and np.array_equal(a[3], b[3])):
                report(kind, f"M={M} nb={nb}", f"iters {a[2]} vs {b[2]}")
        elif kind == "cote":
            N = int(rng.choice([1, 2, 3, 64, 255, 256, 257, 1000, 5000]))
            X = rng.normal(0, float(rng.choice([0.0, 0.1, 5.0])), N)
            if rng.random() < 0.3:
                X = np.round(X, 1)  # ties
            r = float(rng.choice([0.05, 0.3]))
            med = bool(rng.integers(0, 2))
            a, b = h.cote_estimate(X, r, med), qo.cote_estimate(X, r, med)
            if not (a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2]):
                report(kind, f"N={N} r={r} med={med}", f"{a[0]} vs {b[0]} ncard {a[2]} vs {b[2]}")
    except ql.QuatroHipError as e:
        report(kind, "exception", f"{e}")
    except (TypeError, KeyError, AttributeError):
        if not DRY:
            raise
print(f"fuzz seed={seed}: {n_cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
