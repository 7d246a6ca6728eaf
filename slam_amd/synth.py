"""Deterministic synthetic inputs for the ktl_vo feature-tracking path (no dataset is available).

Two generators, both pure functions of their seed:

* ``texture_pair``  - a band-limited value-noise image and the same texture evaluated at a
  sub-pixel offset (analytic shift, no resampling): ground-truth flow for LK unit tests.
* ``CorridorScene`` - a KITTI-shaped stereo corridor (ground plane, two side walls, end wall)
  rendered by per-pixel ray/plane intersection with exact pose and depth; intrinsics default
  to the reference's (src/ktl_vo/src/cameraParameters.h:16-27).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    """splitmix64 finaliser on uint64 arrays."""
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _lattice(ix, iy, salt):
    """value in [-1,1) attached to integer lattice point (ix, iy)."""
    with np.errstate(over="ignore"):
        k = (ix.astype(np.int64).astype(np.uint64) * np.uint64(0x8DA6B343)
             + iy.astype(np.int64).astype(np.uint64) * np.uint64(0xD8163841) + np.uint64(salt))
        h = _mix(k)
    return (h >> np.uint64(40)).astype(np.float64) * (2.0 / (1 << 24)) - 1.0


def value_noise(u, v, salt):
    """C1-smooth value noise with unit lattice spacing."""
    fu, fv = np.floor(u), np.floor(v)
    a, b = u - fu, v - fv
    a = a * a * (3 - 2 * a)
    b = b * b * (3 - 2 * b)
    ix, iy = fu.astype(np.int64), fv.astype(np.int64)
    v00 = _lattice(ix, iy, salt)
    v01 = _lattice(ix + 1, iy, salt)
    v10 = _lattice(ix, iy + 1, salt)
    v11 = _lattice(ix + 1, iy + 1, salt)
    return (v00 * (1 - a) + v01 * a) * (1 - b) + (v10 * (1 - a) + v11 * a) * b


def fractal(u, v, seed, periods, footprint=None):
    """Sum of value-noise octaves with the given lattice periods (same unit as u, v).

    ``footprint`` (same shape as u) is the size of one pixel in texture units; octaves finer
    than two footprints are faded out so distant surfaces do not alias."""
    acc = np.zeros_like(u, dtype=np.float64)
    norm = 0.0
    for o, p in enumerate(periods):
        amp = 1.0
        n = value_noise(u / p, v / p, (seed * 1000003 + o * 7919) & 0xFFFFFFFF)
        if footprint is not None:
            n = n * np.clip(p / (2.0 * footprint) - 0.5, 0.0, 1.0)
        acc += amp * n
        norm += amp * amp
    return acc / np.sqrt(norm)


def _to_u8(tex, seed, frame, noise=True, sigma=40.0):
    img = 128.0 + sigma * 1.9 * tex  # value noise has std ~0.5 per octave sum
    if noise:
        h, w = img.shape
        yy, xx = np.mgrid[0:h, 0:w]
        img = img + _lattice(xx, yy, (seed * 2654435761 + frame * 40503 + 17) & 0xFFFFFFFF)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def texture_image(w, h, seed, dx=0.0, dy=0.0, periods=(48.0, 24.0, 12.0, 6.0), noise=False):
    """Texture sampled at pixel centres shifted by (dx, dy): I(x, y) = T(x - dx, y - dy)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    tex = fractal(xx - dx, yy - dy, seed, periods)
    return _to_u8(tex, seed, 0, noise=noise)


def texture_pair(w, h, seed, dx, dy, **kw):
    """(prev, next): a point at p in prev appears at p + (dx, dy) in next."""
    return texture_image(w, h, seed, 0.0, 0.0, **kw), texture_image(w, h, seed, dx, dy, **kw)


class CorridorScene:
    """Synthetic stereo corridor.  Camera frame: x right, y down, z forward."""

    def __init__(self, width=1241, height=376, seed=0x5EED0001, fx=707.0912, fy=707.0912, cx=None, cy=None,
                 baseline=0.5461, step=0.8, yaw_amp_deg=0.5, yaw_period=50):
        self.w, self.h, self.seed = width, height, seed & 0xFFFFFFFF
        self.fx, self.fy = fx, fy
        # reference principal point (601.8873, 183.1104) is for 1242x375; keep its offset from centre
        self.cx = cx if cx is not None else 601.8873 + (width - 1242) * 0.5
        self.cy = cy if cy is not None else 183.1104 + (height - 375) * 0.5
        self.baseline = baseline
        self.bf = baseline * fx
        self.step, self.yaw_amp, self.yaw_period = step, np.deg2rad(yaw_amp_deg), yaw_period
        self.periods = (0.9, 0.45, 0.225, 0.1125)
        yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
        self._ray = np.stack([(xx - self.cx) / fx, (yy - self.cy) / fy, np.ones_like(xx)], -1)

    def pose(self, frame):
        """camera-to-world (R, C): p_world = R p_cam + C for the LEFT camera."""
        yaw = self.yaw_amp * np.sin(2 * np.pi * frame / self.yaw_period)
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        C = np.array([0.0, 0.0, self.step * frame])
        return R, C

    def relative_pose(self, f0, f1):
        """T such that p_cam(f1) = T p_cam(f0) (4x4), i.e. the reference's mTcw_local."""
        R0, C0 = self.pose(f0)
        R1, C1 = self.pose(f1)
        T = np.eye(4)
        T[:3, :3] = R1.T @ R0
        T[:3, 3] = R1.T @ (C0 - C1)
        return T

    def render(self, frame, right=False, with_depth=False, noise=True):
        R, C = self.pose(frame)
        if right:
            C = C + R @ np.array([self.baseline, 0.0, 0.0])
        d = self._ray @ R.T
        big = 1e30
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(d[..., 1] > 1e-9, (1.65 - C[1]) / d[..., 1], big)
            tl = np.where(d[..., 0] < -1e-9, (-6.0 - C[0]) / d[..., 0], big)
            tr = np.where(d[..., 0] > 1e-9, (6.0 - C[0]) / d[..., 0], big)
            te = np.where(d[..., 2] > 1e-9, (120.0 - C[2]) / d[..., 2], big)
        ts = np.stack([tg, tl, tr, te], -1)
        ts = np.where(ts > 0, ts, big)
        plane = np.argmin(ts, -1)
        t = np.take_along_axis(ts, plane[..., None], -1)[..., 0]
        P = C + d * t[..., None]
        u = np.choose(plane, [P[..., 0], P[..., 2], P[..., 2], P[..., 0]])
        v = np.choose(plane, [P[..., 2], P[..., 1], P[..., 1], P[..., 1]])
        u = u + plane * 1000.0  # decorrelate the four surfaces
        tex = fractal(u, v, self.seed, self.periods, footprint=t / self.fx)
        img = _to_u8(tex, self.seed, frame * 2 + int(right), noise=noise)
        if with_depth:
            return img, t  # t is depth along z because ray z-component is 1 in camera frame
        return img

    def stereo_pair(self, frame, noise=True):
        return self.render(frame, False, noise=noise), self.render(frame, True, noise=noise)


def render_corridor_frames(width, height, seed, n_frames, workers=8):
    """Frames 0 .. n_frames-1 of CorridorScene(width, height, seed) (left camera), rendered by `workers`
    fresh interpreter processes that import nothing but numpy and this file (safe to call from a
    process that holds a GPU context; no fork of it takes place)."""
    import os
    import subprocess
    import sys
    import tempfile
    workers = max(1, min(workers, n_frames))
    if workers == 1:
        sc = CorridorScene(width, height, seed=seed)
        return [sc.render(f) for f in range(n_frames)]
    here = os.path.abspath(__file__)
    with tempfile.TemporaryDirectory(prefix="ktlvo_frames_") as tmp:
        procs = []
        for k in range(workers):
            out = os.path.join(tmp, "part%d.npy" % k)
            procs.append((out, subprocess.Popen([sys.executable, here, str(width), str(height), str(seed), str(k),
                                                 str(workers), str(n_frames), out])))
        parts = []
        for out, pr in procs:
            if pr.wait() != 0:
                raise RuntimeError("frame renderer failed")
            parts.append(np.load(out))
    frames = [None] * n_frames
    for k, part in enumerate(parts):
        for i, f in enumerate(range(k, n_frames, workers)):
            frames[f] = np.ascontiguousarray(part[i])
    return frames


if __name__ == "__main__":
    import sys
    w_, h_, seed_, first_, stride_, n_, out_ = sys.argv[1:8]
    sc_ = CorridorScene(int(w_), int(h_), seed=int(seed_))
    np.save(out_, np.stack([sc_.render(f) for f in range(int(first_), int(n_), int(stride_))]))
