"""Seeded synthetic frame pairs shaped like the reference's KITTI odometry samples.

The reference reads ``N x 4`` float32 ``.bin`` frames, moves them to the camera frame
(x right, y down, z forward), drops ground (``y > 1.1``) and far points (``|x|,|z| >= 30``)
and samples 8192 points (``slam/dataset/kitti_odometry_dataset.py:149-172,375-397``).
There is no dataset on the GPU box, so the benchmark and the parity tests use the two
generators below.  numpy only; deterministic for a given seed.
"""
import numpy as np

SENSOR_HEIGHT = 1.73  # m above the ground plane (KITTI HDL-64E mount)


def uniform_pair(seed, npoints=8192, batch=1):
    """Fixture generator: x,z ~ U(-30,30), y ~ U(-3,1.1); frame 2 = small rigid motion of an
    independent draw.  Duplicate-free with probability 1, no point with |p|^2 <= 1e-3 in
    practice (asserted).  Returns two float32 arrays (batch, npoints, 4) -- the 4th channel
    mirrors the ``.bin`` layout and is ignored by the model."""
    out = []
    for f in range(2):
        frames = []
        for b in range(batch):
            r = np.random.default_rng([seed, b, f])
            p = np.empty((npoints, 4), dtype=np.float32)
            p[:, 0] = r.uniform(-30, 30, npoints)
            p[:, 1] = r.uniform(-3.0, 1.1, npoints)
            p[:, 2] = r.uniform(-30, 30, npoints)
            p[:, 3] = r.uniform(0, 1, npoints)
            assert (np.sum(p[:, :3].astype(np.float64) ** 2, axis=1) > 1e-2).all()
            frames.append(p)
        out.append(np.stack(frames))
    return out[0], out[1]


def _scene(r):
    """A few finite vertical walls around the sensor, in the frame-1 velodyne frame
    (x forward, y left, z up, origin at the sensor)."""
    walls = []
    for _ in range(int(r.integers(6, 12))):
        phi = r.uniform(0, 2 * np.pi)
        walls.append(dict(n=np.array([np.cos(phi), np.sin(phi), 0.0]),
                          m=np.array([-np.sin(phi), np.cos(phi), 0.0]),
                          r=r.uniform(4.0, 28.0),            # distance of the plane from the origin
                          u0=r.uniform(-15, 5), w=r.uniform(3, 20),  # lateral extent [u0, u0+w]
                          h=r.uniform(1.5, 7.0)))            # height above ground
    return walls


def _cast(r, walls, origin, yaw, n_azimuth=2048, noise=0.02, max_range=80.0):
    """Ray-cast a 64-beam spinning lidar at `origin`/`yaw` (world = frame-1 velodyne frame).
    Returns hit points in the sensor's own velodyne frame, float64 (M,3)."""
    elev = np.deg2rad(np.linspace(2.0, -24.8, 64))
    azim = np.linspace(0, 2 * np.pi, n_azimuth, endpoint=False) + r.uniform(0, 2 * np.pi / n_azimuth)
    e, a = np.meshgrid(elev, azim, indexing="ij")
    d = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], axis=-1).reshape(-1, 3)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    dw = d @ R.T                                     # ray directions in the world frame
    t = np.full(len(d), np.inf)
    # ground plane z = -SENSOR_HEIGHT
    dz = dw[:, 2]
    tg = np.where(dz < -1e-6, (-SENSOR_HEIGHT - origin[2]) / np.where(dz < -1e-6, dz, -1.0), np.inf)
    t = np.minimum(t, np.where(tg > 0, tg, np.inf))
    for wl in walls:
        nd = dw @ wl["n"]
        nd = np.where(np.abs(nd) > 1e-6, nd, 1e-6)  # grazing rays: far hit, rejected by range
        tw = np.clip((wl["r"] - origin @ wl["n"]) / nd, -1e6, 1e6)
        hit = origin[None, :] + tw[:, None] * dw
        u = hit @ wl["m"]
        ok = (nd > 1e-6) & (tw > 0.5) & (u >= wl["u0"]) & (u <= wl["u0"] + wl["w"]) & \
             (hit[:, 2] <= wl["h"] - SENSOR_HEIGHT) & (hit[:, 2] >= -SENSOR_HEIGHT)
        t = np.minimum(t, np.where(ok, tw, np.inf))
    keep = np.isfinite(t) & (t < max_range)
    t = t[keep] + r.normal(0.0, noise, int(keep.sum()))
    return d[keep] * t[:, None]


def _to_camera_and_filter(r, pts_velo, npoints):
    """velodyne (x fwd, y left, z up) -> camera (x right, y down, z fwd), then the reference's
    ``filter_pcd`` (kitti_odometry_dataset.py:149-172): drop ground / far points, sample."""
    cam = np.stack([-pts_velo[:, 1], -pts_velo[:, 2], pts_velo[:, 0]], axis=1)
    near = (cam[:, 1] <= 1.1) & (np.abs(cam[:, 0]) < 30) & (np.abs(cam[:, 2]) < 30)
    idx = np.nonzero(near)[0]
    if len(idx) >= npoints:
        sel = r.choice(idx, npoints, replace=False)
    else:  # the reference pads by re-sampling; keep the fixture duplicate-free instead
        raise RuntimeError(f"synthetic scene produced only {len(idx)} usable points")
    return cam[sel]


def kitti_like_pair(seed, npoints=8192, batch=1):
    """Benchmark generator: 64-beam ray cast of a ground plane + random walls, frame 2 cast
    from a pose moved by yaw ~ U(-2,2) deg and 0.5-1.5 m forward.  Returns
    (pc1, pc2, q_gt, t_gt): float32 (batch, npoints, 4) x2, (batch,4) scalar-first, (batch,3)."""
    pcs1, pcs2, qs, ts = [], [], [], []
    for b in range(batch):
        r = np.random.default_rng([seed, b])
        walls = _scene(r)
        yaw = np.deg2rad(r.uniform(-2.0, 2.0))
        fwd = r.uniform(0.5, 1.5)
        p1 = _to_camera_and_filter(r, _cast(r, walls, np.zeros(3), 0.0), npoints)
        p2 = _to_camera_and_filter(r, _cast(r, walls, np.array([fwd, 0.0, 0.0]), yaw), npoints)
        for dst, p in ((pcs1, p1), (pcs2, p2)):
            arr = np.empty((npoints, 4), dtype=np.float32)
            arr[:, :3] = p
            arr[:, 3] = r.uniform(0, 1, npoints)
            dst.append(arr)
        # camera frame: yaw about velodyne z = rotation about camera -y
        qs.append(np.array([np.cos(yaw / 2), 0.0, -np.sin(yaw / 2), 0.0], dtype=np.float32))
        ts.append(np.array([0.0, 0.0, fwd], dtype=np.float32))
    return np.stack(pcs1), np.stack(pcs2), np.stack(qs), np.stack(ts)
