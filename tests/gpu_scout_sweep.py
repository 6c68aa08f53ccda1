"""k_hcore_async's scout workgroup (solver.hip: hca_scout) on graphs of the floor's regime, small enough for a sweep.
The product runs the scout above 8192 vertices only; this script loads the -DQTR_TEST_ENGINES build, where
QTR_HCORE_SCOUT_MIN_L moves that threshold (tests/test_gpu_parity.py starts it in a process of its own: the library
reads the variable once).  A dense block that is not a clique sets the bet's floor F = H / 2; cliques are planted around
it and above it; every clique, largest core and core number at or above the floor in force must be the oracle's,
whichever floor — the bet's, the scout's — the stage worked with and whether it ran once or twice.
usage: QTR_HCORE_SCOUT_MIN_L=1000 python tests/gpu_scout_sweep.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from quatro_amd import lib as ql
from oracle import oracle as qo


def bitmap_of(A):
    A = np.triu(A, 1)
    A = A | A.T
    L = A.shape[0]
    bits = np.zeros((L, ((L + 63) // 64) * 64), dtype=np.uint8)
    bits[:, :L] = A
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, -1)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    h = ql.Handle(0, lib_path=ql.TEST_ENGINES_LIB_PATH)
    rng = np.random.default_rng(4242)
    ways = {"bet": 0, "scout": 0, "second run": 0, "exact": 0}
    for case in range(cases):
        L = int(rng.integers(1300, 3400))
        p = float(rng.choice([0.004, 0.01, 0.02, 0.04]))
        m = max(6, int(p * L))
        A = np.triu(rng.random((L, L)) < p, 1)
        kind = case % 4  # 0: dense block + cliques around its floor; 1: cliques only (no bet: the scout's ground);
        #                  2: two overlapping near-cliques (the peel has to choose); 3: a clique inside a dense block
        B = 6 * m
        blk = rng.choice(L, min(B, L), replace=False)
        if kind in (0, 3):
            A[np.ix_(blk, blk)] |= rng.random((blk.size, blk.size)) < 0.5
        deg = (A | A.T).sum(1)
        H = int(np.sum(np.sort(deg)[::-1] >= np.arange(1, L + 1)))
        F = H // 2
        if kind == 0:
            sizes = [[F - 1], [F], [F + 1], [F + 2], [2 * F], [F + 1, F + 1], [3 * F]][(case // 4) % 7]
        elif kind == 1:
            sizes = [[m + 5], [2 * m], [3 * m, 3 * m - 1], [20], [16, 17, 18], [H], [H + 9, H // 2]][(case // 4) % 7]
        else:
            sizes = []
        for sz in sizes:
            mem = rng.choice(L, max(2, min(sz, L)), replace=False)
            A[np.ix_(mem, mem)] = True
        if kind == 2:
            base = rng.choice(L, 5 * m, replace=False)
            a, b = base[: 3 * m], base[2 * m:]
            A[np.ix_(a, a)] = True
            A[np.ix_(b, b)] |= rng.random((b.size, b.size)) < 0.97
        if kind == 3:
            mem = rng.choice(blk, max(2, blk.size // 3), replace=False)
            A[np.ix_(mem, mem)] = True
        bm = bitmap_of(A)
        got, max_core = h.max_clique(bm, 1)
        st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
        ref = qo.max_clique(bm, 1, 0.5)
        core, _, mc = qo.kcore(bm)
        core = np.asarray(core)
        assert np.array_equal(got, ref), (case, kind, L, p, sizes, int(st[22]), int(st[29]), got.size, ref.size)
        assert max_core == mc, (case, kind, L, p, sizes)
        core_g = h.debug_fetch(ql.DBG_CORE, np.int32)[:L]
        floor = int(st[29])
        hi = core >= floor
        assert np.array_equal(core_g[hi], core[hi]), (case, kind, floor)
        assert np.all(core_g[~hi] >= core[~hi]) and np.all(core_g[~hi] < floor), (case, kind, floor)
        deg2 = np.unpackbits(bm.view(np.uint8), axis=1).sum(1)
        H2 = int(np.sum(np.sort(deg2)[::-1] >= np.arange(1, L + 1)))
        way = "second run" if st[22] else "exact" if floor == 0 else "bet" if floor == H2 // 2 else "scout"
        ways[way] += 1
        print(f"case {case} kind {kind} L {L} p {p} planted {sizes} clique {got.size} floor {floor} (H/2 = {H2 // 2}) {way}", flush=True)
    print("ways", ways)
    print("SCOUT_SWEEP_OK")


if __name__ == "__main__":
    main()
