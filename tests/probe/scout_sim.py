"""CPU model of k_hcore_async's scout workgroup (solver.hip): the values after a few h-index iterations from the degrees,
the top-M vertices by value, the min-degree peel of their sub-graph down to a clique, and what floor that clique gives —
against the oracle's core numbers and heuristic clique.  usage: python tests/probe/scout_sim.py [L] [fractions...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quatro_amd import synth
from oracle import oracle

M = 768


def unpack(bm, L):
    bits = np.unpackbits(bm.view(np.uint8), axis=1, bitorder="little")[:, :L]
    return bits.astype(bool)


def h_index_rows(adj, vals):
    out = vals.copy()
    for v in range(adj.shape[0]):
        x = np.sort(vals[adj[v]])[::-1]
        k = np.arange(1, x.size + 1)
        ok = x >= k
        out[v] = min(vals[v], int(k[ok].max()) if ok.any() else 0)
    return out


def scout(adj, vals):
    hist = np.bincount(np.minimum(vals, 1023), minlength=1024)
    cnt_ge = hist[::-1].cumsum()[::-1]
    ok = np.nonzero((cnt_ge <= M) & (np.arange(1024) >= 2))[0]
    if ok.size == 0 or ok[0] >= 1023:
        return 0, 0, 0
    theta = int(ok[0])
    cand = np.nonzero(vals >= theta)[0]
    if cand.size < 16:
        return 0, theta, cand.size
    sub = adj[np.ix_(cand, cand)]
    alive = np.ones(cand.size, dtype=bool)
    rounds = 0
    while True:
        rounds += 1
        d = (sub[:, alive].sum(axis=1))
        n = int(alive.sum())
        if n < 8 or rounds > 512:
            return 0, theta, cand.size
        mn, mx = int(d[alive].min()), int(d[alive].max())
        if mn == n - 1:
            return n, theta, cand.size, rounds
        thr = mn + ((mx - mn) >> 3)
        alive &= d > thr


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    fracs = [float(a) for a in sys.argv[2:]] or [0.02, 0.015, 0.01, 0.0]
    for fr in fracs:
        src, tgt, _, _ = synth.correspondences(L, fr, seed=3, noise=0.05)
        t0 = time.time()
        bm = oracle.build_graph(src, tgt)
        core = np.asarray(oracle.kcore(bm)[0])
        clique = oracle.max_clique(bm)
        adj = unpack(np.asarray(bm).reshape(L, -1), L)
        vals = adj.sum(axis=1).astype(np.int64)
        for it in range(4):
            res = scout(adj, vals)
            s = res[0]
            f2 = s - 1 - (s >> 3) if s >= 16 else 0
            print(f"L {L} planted {fr}: after {it} iterations: scout clique {s} (theta {res[1]}, candidates {res[2]}, rounds "
                  f"{res[3] if len(res) > 3 else '-'}) floor {f2} | oracle clique {len(clique)} median core {int(np.median(core))} "
                  f"max core {int(core.max())} vertices with core >= floor {(core >= f2).sum() if f2 else L}", flush=True)
            if it < 3:
                vals = h_index_rows(adj, vals)
        print(f"  ({time.time() - t0:.1f} s)")


if __name__ == "__main__":
    main()
