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
           far_r0: float = 50.0, n_trees: int = 0, n_hedges: int = 0, leaf_p: float = 0.5, crown: float = 1.0):
    """Axis-aligned boxes as (lo[3], hi[3]) rows in the world frame: buildings, vehicles, wall segments,
    poles (thin tall boxes) and small clutter boxes (vegetation stand-in).  With n_trees / n_hedges > 0 a third
    array marks POROUS boxes (tree crowns, hedges): a ray entering one returns from a random depth inside it with
    probability < 1 and otherwise passes through — the volumetric returns that make real vegetation occupy many
    voxels per beam (what lifts a KITTI scan to ~16 k voxels at 0.3 m)."""
    lo_l, hi_l, por_l = [], [], []

    def add(cx, cy, sx, sy, z0, sz, porous=0.0):
        lo_l.append([cx - sx / 2, cy - sy / 2, z0])
        hi_l.append([cx + sx / 2, cy + sy / 2, z0 + sz])
        por_l.append(porous)

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
    for _ in range(n_trees):  # trunk (solid) + crown (porous)
        r = np.sqrt(rng.uniform(7.0 ** 2, 76.0 ** 2))
        a = rng.uniform(-np.pi, np.pi)
        cx, cy = r * np.cos(a), r * np.sin(a)
        w = rng.uniform(0.2, 0.5)
        h_trunk = rng.uniform(1.8, 3.5)
        add(cx, cy, w, w, -SENSOR_HEIGHT, h_trunk)
        cw, ch = rng.uniform(3.0, 7.0) * crown, rng.uniform(3.0, 7.0) * crown
        add(cx, cy, cw, cw, -SENSOR_HEIGHT + h_trunk - 0.3, ch, porous=rng.uniform(0.5, 1.0) * leaf_p)
    for _ in range(n_hedges):  # low porous strips
        r = np.sqrt(rng.uniform(8.0 ** 2, 70.0 ** 2))
        a = rng.uniform(-np.pi, np.pi)
        sx, sy = rng.uniform(1.0, 2.0), rng.uniform(4.0, 18.0)
        if rng.random() < 0.5:
            sx, sy = sy, sx
        add(r * np.cos(a), r * np.sin(a), sx, sy, -SENSOR_HEIGHT, rng.uniform(1.0, 2.4),
            porous=min(1.0, rng.uniform(0.8, 1.4) * leaf_p))
    if n_trees or n_hedges:
        return np.array(lo_l), np.array(hi_l), np.array(por_l)
    return np.array(lo_l), np.array(hi_l)


N_BEAMS, N_AZ = 64, 1800
_ELEV = np.deg2rad(np.linspace(2.0, -24.9, N_BEAMS))
_AZ = np.linspace(-np.pi, np.pi, N_AZ, endpoint=False)


def _ray_dirs() -> np.ndarray:
    ce, se = np.cos(_ELEV)[:, None], np.sin(_ELEV)[:, None]
    return np.stack([ce * np.cos(_AZ)[None, :], ce * np.sin(_AZ)[None, :], np.broadcast_to(se, (N_BEAMS, N_AZ))],
                    axis=-1)


def _scan(origin: np.ndarray, yaw: float, lo: np.ndarray, hi: np.ndarray, rng: np.random.Generator, sigma: float,
          bump_k: np.ndarray, bump_ph: np.ndarray, bump_a: float, ground: bool = False, porous: np.ndarray | None = None):
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
        if porous is not None and porous[b] > 0.0:
            # foliage: a ray entering the box returns from a random depth inside it with probability porous[b] and
            # passes through otherwise (leaves are far below the 0.3 m voxel scale, so two scans taken metres apart do
            # not see the same returns — as in real vegetation)
            ok &= rng.random(tmin.shape) < porous[b]
            tmin = tmin + rng.random(tmin.shape) * np.maximum(tmax - tmin, 0.0)
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
                 sigma: float = 0.02, max_yaw: float = np.pi, max_xy: float = 10.0, bump_a: float = 0.12,
                 n_trees: int = 0, n_hedges: int = 0, far_r0: float = 50.0, leaf_p: float = 0.5, crown: float = 1.0,
                 clear_r: float = 0.0):
    """Returns (src_xyzi, tgt_xyzi, T_gt) with tgt ~= T_gt @ src (4x4, yaw + translation only)."""
    rng = np.random.default_rng(SEED_BASE + pair_id)
    sc = _scene(rng, n_boxes, n_poles, n_clutter, n_far, far_r0, n_trees, n_hedges, leaf_p, crown)
    lo, hi = sc[0], sc[1]
    porous = sc[2] if len(sc) > 2 else None
    bump_k = rng.normal(0.0, 2.0, size=(6, 3))
    bump_ph = rng.uniform(0.0, 2 * np.pi, size=6)
    yaw = rng.uniform(-max_yaw, max_yaw)
    t = np.array([rng.uniform(-max_xy, max_xy), rng.uniform(-max_xy, max_xy), rng.uniform(-0.2, 0.2)])
    if clear_r > 0.0:  # keep a disc around BOTH sensor poses free of objects (a vehicle drives on a clear lane)
        keep = np.ones(lo.shape[0], dtype=bool)
        for c in (np.zeros(2), t[:2]):
            dx = np.maximum(np.maximum(lo[:, 0] - c[0], c[0] - hi[:, 0]), 0.0)
            dy = np.maximum(np.maximum(lo[:, 1] - c[1], c[1] - hi[:, 1]), 0.0)
            keep &= np.hypot(dx, dy) > clear_r
        lo, hi = lo[keep], hi[keep]
        porous = porous[keep] if porous is not None else None
    src = _scan(np.zeros(3), 0.0, lo, hi, rng, sigma, bump_k, bump_ph, bump_a, porous=porous)  # pose A = identity
    tgt = _scan(t, yaw, lo, hi, rng, sigma, bump_k, bump_ph, bump_a, porous=porous)             # pose B = (yaw, t)
    # world point p: src coords = p ; tgt coords = R^T (p - t)  =>  tgt = R^T src - R^T t
    R = yaw_matrix(yaw)
    T = np.eye(4)
    T[:3, :3] = R.T
    T[:3, 3] = -R.T @ t
    return src, tgt, T


# The bench workload of BASELINE.json configs[1] / SURVEY.md section 8(d) config 2: the structural scene above plus
# vegetation, which is what lifts a 64-beam scan from ~9 k to ~16 k voxels at 0.3 m (mean over pair ids 0..5:
# n_src 16.0 k, n_tgt 15.5 k; every pair still registers to its ground truth).
KITTI16K = dict(n_trees=1000, n_hedges=300, leaf_p=0.35, crown=1.4, clear_r=5.0)


def kitti64_pair_16k(pair_id: int = 0):
    """kitti64_pair with the KITTI16K profile (n_s ~ n_t ~ 16 k voxels at leaf 0.3 m)."""
    return kitti64_pair(pair_id, **KITTI16K)


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


def _dense_surfaces(n: int, rng: np.random.Generator) -> np.ndarray:
    """n points sampled on six large planes (20 mm thick) and a cylinder — surface-like, ~20-60 neighbours inside
    r = 0.75 m."""
    parts = []
    for k in range(6):
        u = rng.random((n // 8, 2)) * np.array([120.0, 25.0])
        plane = np.zeros((n // 8, 3))
        plane[:, 0] = u[:, 0] - 60 + 3 * k
        plane[:, 1] = (k - 3) * 9.0 + 0.02 * rng.standard_normal(n // 8)
        plane[:, 2] = u[:, 1] - 2
        if k % 2:
            plane = plane[:, [1, 0, 2]]
        parts.append(plane)
    th = rng.random(n - sum(p.shape[0] for p in parts)) * 2 * np.pi
    cyl = np.stack([40 * np.cos(th), 40 * np.sin(th), rng.random(th.size) * 20 - 2], axis=1)
    parts.append(cyl)
    return np.concatenate(parts)


def dense_pair(n: int = 50000, seed: int = 7, yaw: float = 0.7, t=(3.0, -2.0, 0.4), independent: bool = True):
    """BASELINE configs[4] front-end input (dense mode: n-point clouds, no voxel step).  With `independent` (default) the
    two clouds are two INDEPENDENT samplings of the same surfaces — two scans, no point of one is a moved copy of a point
    of the other — and the call returns (src, tgt, T) with tgt ~ T @ src as surfaces.  independent=False is round 2's
    input: tgt[i] is the moved copy of src[perm[i]]; returns (src, tgt, perm)."""
    rng = np.random.default_rng(seed)
    pts = _dense_surfaces(n, rng).astype(np.float32)
    src = np.zeros((n, 4), dtype=np.float32)
    src[:, :3] = pts
    R = yaw_matrix(yaw)[:3, :3]
    tgt = np.zeros((n, 4), dtype=np.float32)
    if not independent:
        perm = rng.permutation(n)
        tgt[:, :3] = (pts[perm].astype(np.float64) @ R.T + np.asarray(t)).astype(np.float32)
        return src, tgt, perm
    other = _dense_surfaces(n, np.random.default_rng(seed + 7919))
    other = other[np.random.default_rng(seed + 1).permutation(n)]
    tgt[:, :3] = (other @ R.T + np.asarray(t)).astype(np.float32)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(t)
    return src, tgt, T


def _dense_scene_boxes(rng: np.random.Generator, area_target: float) -> np.ndarray:
    """Rows (cx, cy, z0, sx, sy, sz, yaw) of upright boxes — buildings, cars, street clutter — until their visible faces
    (four sides + top) add up to `area_target` m^2."""
    rows, area = [], 0.0
    while area < area_target:
        kind = rng.random()
        if kind < 0.15:
            sx, sy, sz = rng.uniform(4, 12), rng.uniform(4, 12), rng.uniform(3, 8)
        elif kind < 0.5:
            sx, sy, sz = rng.uniform(3.5, 5), rng.uniform(1.6, 2.2), rng.uniform(1.3, 2.0)
        else:
            sx, sy, sz = rng.uniform(0.4, 2.5), rng.uniform(0.4, 2.5), rng.uniform(0.4, 3.0)
        rows.append((rng.uniform(-60, 60), rng.uniform(-60, 60), rng.uniform(-1.8, -1.0), sx, sy, sz, rng.uniform(0, np.pi)))
        area += 2 * sx * sz + 2 * sy * sz + sx * sy
    return np.array(rows)


def _sample_box_faces(rng: np.random.Generator, boxes: np.ndarray, n: int) -> np.ndarray:
    """n points drawn uniformly by AREA over the boxes' four side faces and tops."""
    sx, sy, sz = boxes[:, 3], boxes[:, 4], boxes[:, 5]
    areas = np.stack([sx * sz, sx * sz, sy * sz, sy * sz, sx * sy], axis=1).ravel()
    pick = rng.choice(areas.size, size=n, p=areas / areas.sum())
    B, f = boxes[pick // 5], pick % 5
    u, v = rng.random(n) - 0.5, rng.random(n)
    sx, sy, sz = B[:, 3], B[:, 4], B[:, 5]
    x = np.select([f == 2, f == 3], [-0.5 * sx, 0.5 * sx], u * sx)
    y = np.select([f == 0, f == 1, f == 4], [-0.5 * sy, 0.5 * sy, (v - 0.5) * sy], u * sy)
    z = np.where(f == 4, sz, v * sz)
    c, s = np.cos(B[:, 6]), np.sin(B[:, 6])
    return np.stack([B[:, 0] + c * x - s * y, B[:, 1] + s * x + c * y, B[:, 2] + z], axis=1)


def dense_scene_pair(n: int = 50000, seed: int = 7, yaw: float = 0.7, t=(3.0, -2.0, 0.4), sigma: float = 0.01,
                     density: float = 17.0):
    """BASELINE configs[4] as a pair that REGISTERS: two dense scans (n points each, no voxel step) of one structured
    scene — a few hundred upright boxes from building to street-clutter size, ~`density` points per m^2 of surface, so a
    point has ~30 neighbours inside r = 0.75 m (6 - 110) and corners / edges give FPFH something to hold on to.  The two
    clouds are INDEPENDENT samplings of the same surfaces (different sample points, `sigma` m of range noise each): no
    point of one is a moved copy of a point of the other.  The matcher's own correspondences (cross check + tuple test:
    L ~ 2 k of which ~200 lie within 0.6 m of their true place; use_tuple_test = 0: L ~ 14 k) carry a clique of ~170 and
    the registration lands within a few centimetres / 5e-4 rad.  Returns (src_xyz4 f32, tgt_xyz4 f32, T_gt) with
    tgt ~ T_gt @ src as surfaces."""
    boxes = _dense_scene_boxes(np.random.default_rng(seed), n / density)
    a = _sample_box_faces(np.random.default_rng(seed + 1), boxes, n)
    b = _sample_box_faces(np.random.default_rng(seed + 2), boxes, n)
    a += sigma * np.random.default_rng(seed + 3).standard_normal(a.shape)
    b += sigma * np.random.default_rng(seed + 4).standard_normal(b.shape)
    R = yaw_matrix(yaw)[:3, :3]
    src = np.zeros((n, 4), dtype=np.float32)
    tgt = np.zeros((n, 4), dtype=np.float32)
    src[:, :3] = a
    tgt[:, :3] = (b @ R.T + np.asarray(t)).astype(np.float32)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(t)
    return src, tgt, T


def save_kitti_bin(path: str, xyzi: np.ndarray) -> None:
    np.asarray(xyzi, dtype=np.float32).reshape(-1, 4).tofile(path)


def load_kitti_bin(path: str, max_points: int = 250000) -> np.ndarray:
    """Reference loader semantics (examples/run_global_registration.cpp:377-402): float32 x,y,z,i,
    at most 250 000 points."""
    a = np.fromfile(path, dtype=np.float32)
    a = a[: (a.size // 4) * 4].reshape(-1, 4)
    return a[:max_points].copy()
