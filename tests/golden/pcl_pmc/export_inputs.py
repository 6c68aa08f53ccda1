"""Writes the fixed inputs of make_pcl_golden (README.md in this directory): the two scans of synth.kitti64_pair(2) as
KITTI .bin records and three random graphs (G(600, 0.05) + planted 30-clique, G(1500, 0.02) + planted 60-clique,
G(300, 0.4)) as the CSR arrays teaser::MaxCliqueSolver::findMaxClique builds (int32 n, int32 m, int64 offsets[n + 1],
int32 neighbours[m], ascending per vertex).  Deterministic; run from the repository root."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "..")))
from quatro_amd import synth  # noqa: E402

GRAPHS = [(600, 0.05, 30, 1), (1500, 0.02, 60, 2), (300, 0.4, 0, 3)]


def random_graph(n, p, planted, seed):
    rng = np.random.default_rng(seed)
    a = np.triu(rng.random((n, n)) < p, 1)
    if planted:
        mem = rng.choice(n, planted, replace=False)
        a[np.ix_(mem, mem)] = True
        a = np.triu(a, 1)
    a = a | a.T
    np.fill_diagonal(a, False)
    return a


def bitmap(a):
    n = a.shape[0]
    w = (n + 63) // 64
    pad = np.zeros((n, w * 64), dtype=bool)
    pad[:, :n] = a
    return np.packbits(pad.reshape(n, w, 64)[:, :, ::-1], axis=2).view(">u8").astype(np.uint64).reshape(n, w)


def main():
    out = os.path.join(HERE, "inputs")
    os.makedirs(out, exist_ok=True)
    s, t, _ = synth.kitti64_pair(2)
    synth.save_kitti_bin(os.path.join(out, "src.bin"), s)
    synth.save_kitti_bin(os.path.join(out, "tgt.bin"), t)
    for g, (n, p, planted, seed) in enumerate(GRAPHS):
        a = random_graph(n, p, planted, seed)
        off = np.concatenate([[0], np.cumsum(a.sum(1))]).astype(np.int64)
        adj = np.concatenate([np.nonzero(a[i])[0] for i in range(n)]).astype(np.int32)
        with open(os.path.join(out, f"graph{g}.csr"), "wb") as f:
            f.write(np.array([n, adj.size], dtype=np.int32).tobytes())
            f.write(off.tobytes())
            f.write(adj.tobytes())
    print("wrote", out)


if __name__ == "__main__":
    main()
