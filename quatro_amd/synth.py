"""Synthetic inputs for tests and bench.

The reference's toy pair (materials/000540.bin, 001319.bin) is downloaded at configure time and is
not in the tree (reference CMakeLists.txt:57-58), so every workload here is generated:

* ``kitti64_pair``  — a KITTI-64-shaped scan pair: 64 beams, vertical FOV +2.0 .. -24.9 deg
  (reference include/imageProjection.hpp:85-91), 1800 azimuth steps, range <= 80 m, ground plane at
  z = -1.723 m (reference config/patchwork_params.yaml:1) removed analytically (stand-in for
  Patchwork, which is out of scope), axis-aligned boxes (buildings, vehicles, walls, poles, clutter)
  with a smooth world-anchored surface displacement, Gaussian range noise.
  Output layout is the reference loader's: float32 x,y,z,intensity per point
  (reference examples/run_global_registration.cpp:377-402).
* ``correspondences`` — solver-only input: L matched pairs with a planted inlier fraction.

All generators are pure numpy and seeded: seed = 0x5154524F + pair_id.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x5154524F
SENSOR_HEIGHT = 1.723
MAX_RANGE = 80.0


def yaw_matrix(yaw: float) -> np.ndarray:
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def _scene(rng: np.random.Generator, n_boxes: int, n_poles: int, n_clutter: int, n_far: int = 0,
           far_r0: float = 50.0):
    """Axis-aligned boxes as (lo[3], hi[3]) rows in the world frame: buildings, vehicles, wall segments,
    poles (thin tall boxes) and small clutter boxes (vegetation stand-in)."""
    lo_l, hi_l = [], []

    def add(cx, cy, sx, sy, z0, sz):
        lo_l.append([cx - sx / 2, cy - sy / 2, z0])
        hi_l.append([cx + sx / 2, cy + sy / 2, z0 + sz])

    for _ in range(n_boxes):
        kind = rng.random()
        r = np.sqrt(rng.uniform(10.0 ** 2, 78.0 ** 2))  # area-uniform
        a = rng.uniform(-np.pi, np.pi)
        cx, cy = r * np.cos(a), r * np.sin(a)
        if kind < 0.35:  # building
            sx, sy, sz = rng.uniform(6, 25), rng.uniform(6, 25), rng.uniform(3, 12)
            if abs(cx) - sx / 2 < 16 and abs(cy) - sy / 2 < 16:
                continue  # keep the sensor neighbourhood (both poses) free of buildings
        elif kind < 0.8:  # car / van
            sx, sy, sz = rng.uniform(1.6, 2.2), rng.uniform(3.5, 5.5), rng.uniform(1.4, 2.2)
            if rng.random() < 0.5:
                sx, sy = sy, sx
        else:  # wall / fence segment
            sx, sy, sz = rng.uniform(0.3, 0.6), rng.uniform(5, 25), rng.uniform(1.0, 2.5)
            if rng.random() < 0.5:
                sx, sy = sy, sx
        add(cx, cy, sx, sy, -SENSOR_HEIGHT, sz)
    for _ in range(n_far):  # perimeter of tall facades so the upper beams return at 40-78 m
        r = rng.uniform(far_r0, 78.0)
        a = rng.uniform(-np.pi, np.pi)
        add(r * np.cos(a), r * np.sin(a), rng.uniform(8, 30), rng.uniform(8, 30), -SENSOR_HEIGHT, rng.uniform(8, 25))
    for _ in range(n_poles):
        r = np.sqrt(rng.uniform(6.0 ** 2, 60.0 ** 2))
        a = rng.uniform(-np.pi, np.pi)
        w = rng.uniform(0.15, 0.4)
        add(r * np.cos(a), r * np.sin(a), w, w, -SENSOR_HEIGHT, rng.uniform(3.0, 9.0))
    for _ in range(n_clutter):  # bushes / crowns: small boxes, some floating (tree crowns)
        r = np.sqrt(rng.uniform(7.0 ** 2, 78.0 ** 2))
        a = rng.uniform(-np.pi, np.pi)
        s3 = rng.uniform(0.4, 2.5, size=3)
        z0 = -SENSOR_HEIGHT + (rng.uniform(1.5, 4.0) if rng.random() < 0.3 else 0.0)
        add(r * np.cos(a), r * np.sin(a), s3[0], s3[1], z0, s3[2])
    return np.array(lo_l), np.array(hi_l)


N_BEAMS, N_AZ = 64, 1800
_ELEV = np.deg2rad(np.linspace(2.0, -24.9, N_BEAMS))
_AZ = np.linspace(-np.pi, np.pi, N_AZ, endpoint=False)


def _ray_dirs() -> np.ndarray:
    ce, se = np.cos(_ELEV)[:, None], np.sin(_ELEV)[:, None]
    return np.stack([ce * np.cos(_AZ)[None, :], ce * np.sin(_AZ)[None, :], np.broadcast_to(se, (N_BEAMS, N_AZ))],
                    axis=-1)


def _scan(origin: np.ndarray, yaw: float, lo: np.ndarray, hi: np.ndarray, rng: np.random.Generator, sigma: float,
          bump_k: np.ndarray, bump_ph: np.ndarray, bump_a: float, ground: bool = False):
    """Ray-cast one 64 x 1800 sweep.  Each box is only tested against the (beam, azimuth) window its
    corners subtend.  Surfaces get a smooth world-anchored range displacement (sum of sinusoids) so
    that local geometry is distinctive and repeatable between the two views."""
    d_local = _ray_dirs()                      # [64,1800,3] sensor frame
    R = yaw_matrix(yaw)
    d = d_local @ R.T                          # world frame
    t_hit = np.full((N_BEAMS, N_AZ), np.inf)
    daz = 2 * np.pi / N_AZ
    corners_sel = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)])
    for b in range(lo.shape[0]):
        l, h = lo[b], hi[b]
        if np.all(origin > l - 0.5) and np.all(origin < h + 0.5):
            continue
        cs = np.where(corners_sel == 0, l[None, :], h[None, :]) - origin[None, :]
        az = np.arctan2(cs[:, 1], cs[:, 0]) - yaw
        rel = np.arctan2(np.sin(az - az[0]), np.cos(az - az[0]))  # unwrap around first corner
        a0, a1 = az[0] + rel.min(), az[0] + rel.max()
        if a1 - a0 > np.pi:           # box wraps around the sensor: test all columns
            cols = np.arange(N_AZ)
        else:
            c0 = int(np.floor((a0 + np.pi) / daz)) - 1
            c1 = int(np.ceil((a1 + np.pi) / daz)) + 1
            cols = np.arange(c0, c1 + 1) % N_AZ
        rho = np.maximum(np.hypot(cs[:, 0], cs[:, 1]).min() * 0.7, 0.3)
        el_hi = np.arctan2(cs[:, 2].max(), rho)
        el_lo = np.arctan2(cs[:, 2].min(), rho)
        rows = np.nonzero((_ELEV <= el_hi + 0.02) & (_ELEV >= el_lo - 0.02))[0]
        if rows.size == 0:
            continue
        dd = d[np.ix_(rows, cols)]
        inv = 1.0 / np.where(np.abs(dd) < 1e-12, 1e-12, dd)
        t1 = (l - origin) * inv
        t2 = (h - origin) * inv
        tmin = np.minimum(t1, t2).max(axis=-1)
        tmax = np.maximum(t1, t2).min(axis=-1)
        ok = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.5)
        cur = t_hit[np.ix_(rows, cols)]
        t_hit[np.ix_(rows, cols)] = np.where(ok & (tmin < cur), tmin, cur)
    # ground: rays whose first hit is the ground plane give no return (ground removed analytically)
    dz = d[..., 2]
    tg = np.where(dz < -1e-9, (-SENSOR_HEIGHT - origin[2]) / np.where(dz < -1e-9, dz, -1.0), np.inf)
    keep = np.isfinite(t_hit) & (t_hit < tg) & (t_hit <= MAX_RANGE)
    pw = origin[None, :] + d[keep] * t_hit[keep][:, None]
    bump = bump_a * np.sin(pw @ bump_k.T + bump_ph[None, :]).sum(axis=1)
    t_meas = t_hit[keep] + bump + rng.normal(0.0, sigma, size=int(keep.sum()))
    pts_local = d_local[keep] * t_meas[:, None]
    out = np.zeros((pts_local.shape[0], 4), dtype=np.float32)
    out[:, :3] = pts_local.astype(np.float32)
    out[:, 3] = rng.random(pts_local.shape[0]).astype(np.float32)  # intensity (unused by the path)
    if not ground:
        return out
    # raw-scan mode (input of the ground-segmentation stage): the ground returns are kept, gently undulating,
    # and the points come in sweep order (beam-major) with a flag telling which ones are ground
    gk = np.isfinite(tg) & (tg <= MAX_RANGE) & ~(np.isfinite(t_hit) & (t_hit < tg))
    pg = origin[None, :] + d[gk] * tg[gk][:, None]
    und = 0.03 * np.sin(0.21 * pg[:, 0] + 0.4) * np.cos(0.17 * pg[:, 1] - 0.3)
    tgm = tg[gk] + und / np.maximum(-dz[gk], 0.05) + rng.normal(0.0, sigma, size=int(gk.sum()))
    gl = d_local[gk] * tgm[:, None]
    full = np.zeros((N_BEAMS, N_AZ, 4), dtype=np.float32)
    flag = np.zeros((N_BEAMS, N_AZ), dtype=np.int8)   # 0 none, 1 object, 2 ground
    full[keep, :3] = pts_local.astype(np.float32)
    flag[keep] = 1
    full[gk, :3] = gl.astype(np.float32)
    flag[gk] = 2
    sel = flag.reshape(-1) > 0
    return full.reshape(-1, 4)[sel], flag.reshape(-1)[sel] == 2


def kitti64_pair(pair_id: int = 0, n_boxes: int = 100, n_poles: int = 60, n_clutter: int = 150, n_far: int = 300,
                 sigma: float = 0.02, max_yaw: float = np.pi, max_xy: float = 10.0, bump_a: float = 0.12):
    """Returns (src_xyzi, tgt_xyzi, T_gt) with tgt ~= T_gt @ src (4x4, yaw + translation only)."""
    rng = np.random.default_rng(SEED_BASE + pair_id)
    lo, hi = _scene(rng, n_boxes, n_poles, n_clutter, n_far)
    bump_k = rng.normal(0.0, 2.0, size=(6, 3))
    bump_ph = rng.uniform(0.0, 2 * np.pi, size=6)
    yaw = rng.uniform(-max_yaw, max_yaw)
    t = np.array([rng.uniform(-max_xy, max_xy), rng.uniform(-max_xy, max_xy), rng.uniform(-0.2, 0.2)])
    src = _scan(np.zeros(3), 0.0, lo, hi, rng, sigma, bump_k, bump_ph, bump_a)   # sensor pose A = identity
    tgt = _scan(t, yaw, lo, hi, rng, sigma, bump_k, bump_ph, bump_a)              # sensor pose B = (yaw, t)
    # world point p: src coords = p ; tgt coords = R^T (p - t)  =>  tgt = R^T src - R^T t
    R = yaw_matrix(yaw)
    T = np.eye(4)
    T[:3, :3] = R.T
    T[:3, 3] = -R.T @ t
    return src, tgt, T


def kitti64_raw_scan(scan_id: int = 0, **kw):
    """One raw KITTI-64-shaped sweep WITH its ground returns: (xyzi float32 in sweep order, is_ground bool)."""
    rng = np.random.default_rng(SEED_BASE + 7919 * (scan_id + 1))
    lo, hi = _scene(rng, kw.get("n_boxes", 100), kw.get("n_poles", 60), kw.get("n_clutter", 150), kw.get("n_far", 300))
    bump_k = rng.normal(0.0, 2.0, size=(6, 3))
    bump_ph = rng.uniform(0.0, 2 * np.pi, size=6)
    return _scan(np.zeros(3), 0.0, lo, hi, rng, kw.get("sigma", 0.02), bump_k, bump_ph, kw.get("bump_a", 0.12), ground=True)


def correspondences(L: int = 5000, inlier_frac: float = 0.05, seed: int = 0, noise: float = 0.1,
                    max_yaw: float = np.pi, max_xy: float = 10.0):
    """Solver-only input: L pairs in [-50,50]^2 x [-2,6] m; returns
    (src_xyz4 f32, tgt_xyz4 f32, T_gt, inlier_idx)."""
    rng = np.random.default_rng(SEED_BASE + 100003 * seed + L)
    src = np.zeros((L, 4), dtype=np.float32)
    src[:, 0] = rng.uniform(-50, 50, L)
    src[:, 1] = rng.uniform(-50, 50, L)
    src[:, 2] = rng.uniform(-2, 6, L)
    yaw = rng.uniform(-max_yaw, max_yaw)
    t = np.array([rng.uniform(-max_xy, max_xy), rng.uniform(-max_xy, max_xy), rng.uniform(-0.2, 0.2)])
    R = yaw_matrix(yaw)
    tgt = np.zeros((L, 4), dtype=np.float32)
    n_in = int(round(L * inlier_frac))
    inl = np.sort(rng.choice(L, size=n_in, replace=False))
    mask = np.zeros(L, dtype=bool)
    mask[inl] = True
    clean = src[:, :3].astype(np.float64) @ R.T + t[None, :]
    clean += rng.uniform(-noise, noise, size=(L, 3))
    rnd = np.stack([rng.uniform(-50, 50, L), rng.uniform(-50, 50, L), rng.uniform(-2, 6, L)], axis=1)
    tgt[:, :3] = np.where(mask[:, None], clean, rnd).astype(np.float32)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return src, tgt, T, inl


def save_kitti_bin(path: str, xyzi: np.ndarray) -> None:
    np.asarray(xyzi, dtype=np.float32).reshape(-1, 4).tofile(path)


def load_kitti_bin(path: str, max_points: int = 250000) -> np.ndarray:
    """Reference loader semantics (examples/run_global_registration.cpp:377-402): float32 x,y,z,i,
    at most 250 000 points."""
    a = np.fromfile(path, dtype=np.float32)
    a = a[: (a.size // 4) * 4].reshape(-1, 4)
    return a[:max_points].copy()
