"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/).

The reference holds no fixtures and its arithmetic (OpenCV 3.x) cannot run here, so these
vectors pin the oracle against itself over time and give the GPU tests committed data to
compare with; they do not pin the oracle to OpenCV.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from slam_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # 1. pyramid + Scharr on an odd-sized image
    img = synth.texture_image(197, 123, seed=11, periods=(24.0, 9.0, 4.0))
    p = O.Pyramid(img, max_level=4, win=21)
    d = {"img": img, "max_level": np.int32(p.max_level)}
    for l in range(p.max_level + 1):
        d["level%d" % l] = p.level(l)
        d["scharr%d" % l] = p.deriv(l)
    np.savez_compressed(os.path.join(OUT, "pyramid_197x123.npz"), **d)

    # 2. LK on a 320x240 pair with a known shift, points include out-of-frame and flat cases
    a, b = synth.texture_pair(320, 240, 21, 2.6, -1.3)
    a[150:, 200:] = 90  # flat patch: min-eigenvalue rejections
    b[150:, 200:] = 90
    rng = np.random.default_rng(21)
    pts = np.stack([rng.uniform(-15, 335, 400), rng.uniform(-15, 255, 400)], 1).astype(np.float32)
    nxt, st, err = O.pyrlk(a, b, pts)                        # serial float sums (OpenCV's scalar order)
    xn, xs, xe = O.pyrlk(a, b, pts, sums=O.SUMS_EXACT)       # exact integer totals rounded once
    np.savez_compressed(os.path.join(OUT, "lk_320x240.npz"), prev=a, next=b, pts=pts, out=nxt, status=st, err=err,
                        out_exact=xn, status_exact=xs, err_exact=xe)

    # 3. corners on ROIs of a 260x180 image (Harris k=0 as the reference, and min-eig)
    img = synth.texture_image(260, 180, seed=31, periods=(20.0, 8.0, 3.0))
    rois = np.array([[30, 30, 59, 39], [89, 30, 59, 39], [10, 100, 64, 64], [196, 116, 64, 64]], np.int32)
    d = {"img": img, "rois": rois}
    for i, r in enumerate(rois):
        xy, rs = O.good_features(img, tuple(r), 100, 0.01, 10.0, True, 0.0)
        d["harris_xy%d" % i], d["harris_resp%d" % i] = xy, rs
        xy, rs = O.good_features(img, tuple(r), 50, 0.01, 5.0, False, 0.0)
        d["mineig_xy%d" % i], d["mineig_resp%d" % i] = xy, rs
    d["resp_harris0"] = O.corner_response(img, tuple(rois[0]), True, 0.0)
    d["resp_mineig0"] = O.corner_response(img, tuple(rois[0]), False, 0.0)
    np.savez_compressed(os.path.join(OUT, "corners_260x180.npz"), **d)

    # 4. RANSAC PnP
    rng = np.random.default_rng(41)
    K = np.array([707.0912, 707.0912, 601.8873, 183.1104])
    n = 400
    R = O.rodrigues_v2m(np.array([0.004, -0.012, 0.002]))
    t = np.array([0.02, -0.01, -0.8])
    Xw = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(4, 38, n)], 1).astype(np.float32)
    Xc = Xw.astype(np.float64) @ R.T + t
    uv = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
    uv += rng.normal(size=uv.shape) * 0.04
    uv[:80] += rng.normal(size=(80, 2)) * 6
    uv = uv.astype(np.float32)
    r = O.solve_pnp_ransac(Xw, uv, K)
    np.savez_compressed(os.path.join(OUT, "ransac_400.npz"), X3=Xw, x2=uv, K=K, rvec=r["rvec"], tvec=r["tvec"],
                        mask=r["mask"], n_inliers=np.int32(r["n_inliers"]), iters=np.int32(r["iters"]), log=r["log"])
    # 5. the frame of the randomised sweep (tests/test_gpu_fuzz.py, seed 1111, 240 cases, case 176:
    #    164x140 white noise with saturated patches, shifted by whole pixels) on which LK results are most
    #    sensitive to the order of the float additions
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fuzz as T
    rng = np.random.default_rng(1111)
    for i, (w, h) in enumerate(T._sizes(rng, 240)):
        a, b = T._images(rng, w, h, i % 3)
        pts = np.stack([rng.uniform(-30, w + 30, 600), rng.uniform(-30, h + 30, 600)], 1).astype(np.float32)
        pts[:8] = [[0, 0], [w - 1, h - 1], [w, h], [-21, -21], [w - 0.5, 0.25], [0.25, h - 0.5], [w / 2, h / 2],
                   [10.5, 10.5]]
        if i == 176:
            assert (w, h) == (164, 140)
            a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
            xn, xs, xe = O.pyrlk(a, b, pts, sums=O.SUMS_EXACT)   # order-independent, so these outputs can be pinned
            np.savez_compressed(os.path.join(OUT, "lk_noise_164x140.npz"), prev=a, next=b, pts=pts,
                                out_exact=xn, status_exact=xs, err_exact=xe)
            break
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
