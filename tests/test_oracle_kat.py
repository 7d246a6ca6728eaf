"""Known-answer and property tests that pin the CPU oracle (CPU only).

The reference has no tests and its arithmetic lives in OpenCV 3.x, which is not available here
(parity unpinned); what can be checked without OpenCV is checked: closed-form answers of the
published kernels, geometric ground truth, and the committed golden vectors.
"""
import os

import numpy as np
import pytest

import oracle as O
from slam_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_pyrdown_constant_and_impulse():
    c = np.full((37, 53), 77, np.uint8)
    assert (O.pyr_down(c) == 77).all()
    img = np.zeros((40, 60), np.uint8)
    img[20, 30] = 255                      # even (x, y): centre weight 6*6 = 36 of 256
    d = O.pyr_down(img)
    assert d.shape == (20, 30)
    assert d[10, 15] == (255 * 36 + 128) >> 8 == 36
    assert d[10, 14] == (255 * 6 + 128) >> 8   # horizontal neighbour sees weight 1*6
    assert d[9, 14] == (255 * 1 + 128) >> 8
    img[:] = 0
    img[21, 31] = 255                      # odd (x, y): weight 4*4 shared by four outputs
    d = O.pyr_down(img)
    assert d[10, 15] == d[11, 16] == d[10, 16] == d[11, 15] == (255 * 16 + 128) >> 8


def test_pyrdown_odd_sizes_and_reflect101():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (31, 45), dtype=np.uint8)
    d = O.pyr_down(img)
    assert d.shape == (16, 23)
    k = np.array([1, 4, 6, 4, 1])
    pad = np.pad(img.astype(np.int64), 2, mode="reflect")  # numpy 'reflect' == reflect-101
    ref = np.zeros_like(d, dtype=np.int64)
    for y in range(16):
        for x in range(23):
            ref[y, x] = (k[:, None] * k[None, :] * pad[2 * y:2 * y + 5, 2 * x:2 * x + 5]).sum()
    assert np.array_equal(d, ((ref + 128) >> 8).astype(np.uint8))


def test_scharr_ramp_and_borders():
    ramp = np.tile(np.arange(100, dtype=np.uint8), (50, 1))
    s = O.scharr(ramp)
    assert (s[1:-1, 1:-1, 0] == 32).all() and (s[..., 1] == 0).all()  # gain 2*(3+10+3) per unit slope
    assert (s[:, 0, 0] == 0).all() and (s[:, -1, 0] == 0).all()        # reflect-101: t(-1) == t(1)
    s = O.scharr(np.ascontiguousarray(ramp.T))
    assert (s[1:-1, 1:-1, 1] == 32).all() and (s[..., 0] == 0).all()
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    s = O.scharr(img)
    p = np.pad(img.astype(np.int32), 1, mode="reflect")
    dx = 3 * (p[:-2, 2:] - p[:-2, :-2]) + 10 * (p[1:-1, 2:] - p[1:-1, :-2]) + 3 * (p[2:, 2:] - p[2:, :-2])
    dy = 3 * (p[2:, :-2] - p[:-2, :-2]) + 10 * (p[2:, 1:-1] - p[:-2, 1:-1]) + 3 * (p[2:, 2:] - p[:-2, 2:])
    assert np.array_equal(s[..., 0], dx) and np.array_equal(s[..., 1], dy)


def test_pyramid_level_rule():
    img = np.zeros((375, 1242), np.uint8)
    p = O.Pyramid(img, 4, 21)
    assert p.max_level == 4
    assert [p.level(l).shape for l in range(5)] == [(375, 1242), (188, 621), (94, 311), (47, 156), (24, 78)]
    p = O.Pyramid(np.zeros((80, 100), np.uint8), 4, 21)   # 40x50 ok, 20x25 not larger than the window
    assert p.max_level == 1


@pytest.mark.parametrize("shift", [(0.3, -0.2), (2.7, 1.4), (-5.5, 3.25), (11.3, -7.6)])
def test_lk_recovers_subpixel_shift(shift):
    dx, dy = shift
    a, b = synth.texture_pair(480, 360, 3, dx, dy)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(60, 420, 200), rng.uniform(60, 300, 200)], 1).astype(np.float32)
    out, st, err = O.pyrlk(a, b, pts)
    assert st.all()
    e = np.abs(out - pts - np.array([dx, dy], np.float32))
    assert e.max() < 0.05 and np.median(e) < 0.01
    out2, st2, err2 = O.pyrlk(a, b, pts, nthreads=4)      # threading splits ranges, not arithmetic
    assert np.array_equal(out, out2) and np.array_equal(st, st2) and np.array_equal(err, err2)


def _float_lk(prev, nxt, pts, levels=5, win=21, iters=30, eps=0.01):
    """Plain float64 pyramidal Lucas-Kanade as published (Bouguet): Gaussian 1-4-6-4-1 pyramid, Scharr
    derivatives of the previous image, bilinear sampling, 21x21 window centred on the point, Gauss-Newton
    steps from the coarsest level down.  No fixed point, no OpenCV-specific rounding."""
    from scipy import ndimage as ndi
    k = np.array([1, 4, 6, 4, 1]) / 16.0

    def down(a):
        a = ndi.correlate1d(ndi.correlate1d(a, k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
        return a[::2, ::2]

    P, N = [prev.astype(np.float64)], [nxt.astype(np.float64)]
    for _ in range(levels - 1):
        P.append(down(P[-1])); N.append(down(N[-1]))
    out = []
    h = (win - 1) / 2
    oy, ox = np.mgrid[-h:h + 1, -h:h + 1]
    for (x, y) in pts:
        g = np.zeros(2)
        for l in range(levels - 1, -1, -1):
            I, J = P[l], N[l]
            Ix = ndi.correlate1d(ndi.correlate1d(I, [-1, 0, 1], axis=1, mode="mirror"), [3, 10, 3], axis=0, mode="mirror") / 32
            Iy = ndi.correlate1d(ndi.correlate1d(I, [-1, 0, 1], axis=0, mode="mirror"), [3, 10, 3], axis=1, mode="mirror") / 32
            px, py = x / 2 ** l, y / 2 ** l
            samp = lambda A, cx, cy: ndi.map_coordinates(A, [oy + cy, ox + cx], order=1, mode="mirror")
            i0, ix, iy = samp(I, px, py), samp(Ix, px, py), samp(Iy, px, py)
            G = np.array([[np.sum(ix * ix), np.sum(ix * iy)], [np.sum(ix * iy), np.sum(iy * iy)]])
            for _ in range(iters):
                d = samp(J, px + g[0], py + g[1]) - i0
                step = -np.linalg.solve(G, [np.sum(d * ix), np.sum(d * iy)])
                g += step
                if step @ step <= eps * eps:
                    break
            if l:
                g *= 2
        out.append([x + g[0], y + g[1]])
    return np.array(out)


def test_lk_against_float_restatement():
    """The fixed-point oracle and a float64 textbook LK land on the same positions (to a few
    hundredths of a pixel) on smooth texture with a 3.3 / -1.7 px shift."""
    a, b = synth.texture_pair(320, 240, 21, 3.3, -1.7, noise=False)
    rng = np.random.default_rng(4)
    pts = np.stack([rng.uniform(40, 280, 40), rng.uniform(40, 200, 40)], 1).astype(np.float32)
    oo, st, _ = O.pyrlk(a, b, pts)
    ref = _float_lk(a, b, pts)
    ok = st == 1
    assert ok.mean() > 0.9
    assert np.abs(oo - ref)[ok].max() < 0.03, np.abs(oo - ref)[ok].max()


def test_lk_status_rules():
    a, b = synth.texture_pair(320, 240, 5, 1.0, 0.5)
    a[100:, 150:] = 128
    b[100:, 150:] = 128
    pts = np.array([[-40.0, 50.0],      # window origin left of -win: status 0, err 0
                    [250.0, 180.0],     # flat area: min eigenvalue below 1e-4
                    [80.0, 60.0]], np.float32)
    out, st, err = O.pyrlk(a, b, pts)
    assert list(st) == [0, 0, 1] and err[0] == 0
    out, st, err = O.pyrlk(a, b, np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2)


def test_corner_response_closed_form():
    # vertical step edge: Dy = 0 everywhere, so det(M) = 0 and min-eig = 0
    img = np.zeros((40, 40), np.uint8)
    img[:, 20:] = 200
    assert np.abs(O.corner_response(img, None, True, 0.0)).max() == 0
    assert np.abs(O.corner_response(img, None, False, 0.0)).max() < 1e-9
    # a single bright quadrant: the response peaks near its corner
    img = np.zeros((41, 41), np.uint8)
    img[20:, 20:] = 200
    r = O.corner_response(img, None, True, 0.0)
    y, x = np.unravel_index(np.argmax(r), r.shape)
    assert abs(x - 20) <= 1 and abs(y - 20) <= 1
    xy, resp = O.good_features(img, None, 10, 0.01, 10.0, True, 0.0)
    assert len(xy) >= 1 and abs(xy[0, 0] - 20) <= 1 and abs(xy[0, 1] - 20) <= 1


def test_corner_response_against_scipy_evaluation():
    """cornerHarris / cornerMinEigenVal as published: Sobel 3x3 scaled by 1 / (255 * 4 * blockSize) = 1/3060
    (8-bit input, aperture 3, block 3), unnormalised 3x3 box sums of dx^2, dxdy, dy^2, then
    det - k tr^2 or the smaller eigenvalue.  Evaluated independently with scipy.ndimage in float64
    ('mirror' is reflect-101); the oracle works in float32 with double box sums."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(8)
    img = (ndi.gaussian_filter(rng.uniform(0, 255, (70, 90)), 1.5) * 1.0).clip(0, 255).astype(np.uint8)
    f = img.astype(np.float64)
    sx = ndi.correlate1d(ndi.correlate1d(f, [-1, 0, 1], axis=1, mode="mirror"), [1, 2, 1], axis=0, mode="mirror") / 3060.0
    sy = ndi.correlate1d(ndi.correlate1d(f, [-1, 0, 1], axis=0, mode="mirror"), [1, 2, 1], axis=1, mode="mirror") / 3060.0
    box = lambda a: ndi.correlate(a, np.ones((3, 3)), mode="mirror")
    a, b, c = box(sx * sx), box(sx * sy), box(sy * sy)
    for k in (0.0, 0.04):
        want = a * c - b * b - k * (a + c) ** 2
        got = O.corner_response(img, None, True, k).astype(np.float64)
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max() + 1e-12
    want = (a + c) / 2 - np.sqrt(((a - c) / 2) ** 2 + b * b)
    got = O.corner_response(img, None, False, 0.0).astype(np.float64)
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max() + 1e-12


def test_good_features_min_distance_and_order():
    img = synth.texture_image(200, 150, seed=7, periods=(16.0, 6.0, 3.0))
    xy, resp = O.good_features(img, (20, 20, 120, 100), 100, 0.01, 10.0, True, 0.0)
    assert 5 < len(xy) <= 100
    assert (np.diff(resp) <= 0).all()
    d = np.linalg.norm(xy[:, None] - xy[None], axis=2) + np.eye(len(xy)) * 1e9
    assert d.min() >= 10.0
    assert (xy >= 1).all() and (xy[:, 0] <= 118).all() and (xy[:, 1] <= 98).all()


def test_roi_sobel_reads_parent_box_reflects_roi():
    img = synth.texture_image(120, 90, seed=9, periods=(10.0, 4.0))
    roi = (30, 20, 40, 30)
    r_roi = O.corner_response(img, roi, True, 0.0)
    r_full = O.corner_response(img, None, True, 0.0)
    # interior of the ROI (away from its box-filter border) equals the full-frame response
    assert np.array_equal(r_roi[1:-1, 1:-1], r_full[21:49, 31:69])
    crop = np.ascontiguousarray(img[20:50, 30:70])
    r_crop = O.corner_response(crop, None, True, 0.0)     # isolated crop reflects Sobel too: differs on the rim
    assert np.array_equal(r_roi[2:-2, 2:-2], r_crop[2:-2, 2:-2])
    assert not np.array_equal(r_roi, r_crop)


def test_anms_keep_all_under_opencv3_and_real_mode():
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 500, (1500, 2)).astype(np.float32)
    zero = np.zeros(1500, np.float32)
    oxy, oresp, perm = O.anms(xy, zero, 1000)
    assert len(oxy) == 1500 and sorted(perm) == list(range(1500))   # keep-all, only a permutation remains
    oxy2, _, perm2 = O.anms(xy[:900], zero[:900], 1000)
    assert np.array_equal(oxy2, xy[:900]) and list(perm2) == list(range(900))  # n < numToKeep: untouched
    resp = rng.uniform(0.1, 1, 1500).astype(np.float32)
    oxy3, oresp3, perm3 = O.anms(xy, resp, 1000)
    assert 1000 <= len(oxy3) <= 1010
    assert perm3[0] == int(np.argmax(resp))                           # strongest has infinite radius


def test_gates_are_one_sided():
    f1 = np.array([[100, 100], [100, 100], [100, 100], [100, 100], [100, 100]], np.float32)
    f2 = np.array([[110, 100], [110, 100], [110, 100], [25, 100], [210, 100]], np.float32)
    chk = np.array([[100, 100], [94, 100], [130, 130], [100, 100], [100, 100]], np.float32)
    keep = O.fb_gate(f1, f2, chk, 1242, 375, y_threshold=1)
    # f1-chk = +6 > 5 drops; f1-chk = -30 passes (signed test); x < 30 drops; trackDis 110 > 100 drops
    assert list(keep) == [1, 0, 1, 0, 0]
    a = np.array([[10, 10], [10, 10], [50, 10]], np.float32)
    b = np.array([[9, 9], [100, 100], [47, 10]], np.float32)
    assert list(O.circular_gate(a, b)) == [1, 1, 0]                  # -90 < 2 passes, +3 fails
    keep, depth = O.depth_gate(np.array([[100, 50], [100, 50], [100, 50]], np.float32),
                               np.array([[90, 50], [101, 50], [99.9, 50]], np.float32), 386.1448, 707.0912)
    assert abs(depth[0] - 38.61448) < 1e-4 and keep[0] == 0          # 38.6 m is beyond bf*70/fx = 38.23 m
    assert keep[1] == 0                                              # negative disparity


def test_p3p_recovers_pose():
    """Gao P3P on exact projections of random poses.  With the closed-form roots as they come (OpenCV 3.x) a
    few per cent of the triples are resolved to 1e-6 .. 1e-3 only and now and then one not at all (depths up to
    40 m over a 1 m motion are badly conditioned and the closed forms lose digits: 1 of 296 here comes out 0.5
    off), so the bars on that mode are quantiles; with two Newton steps on the roots every case has a hard bound."""
    K = np.array([707.0912, 707.0912, 601.8873, 183.1104])
    for polish in (0, 2):
        O.p3p_set_polish(polish)
        try:
            rng = np.random.default_rng(1)
            e3, e4 = [], []
            for _ in range(300):
                rv = rng.normal(size=3) * 0.3
                R = O.rodrigues_v2m(rv)
                t = rng.normal(size=3) * np.array([0.5, 0.2, 1.0])
                Xw = np.stack([rng.uniform(-8, 8, 4), rng.uniform(-2, 2, 4), rng.uniform(4, 40, 4)], 1)
                Xc = Xw @ R.T + t
                if (Xc[:, 2] < 1).any():
                    continue
                uv = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
                Rs, ts = O.p3p(Xw[:3], uv[:3], K)
                assert 1 <= len(Rs) <= 4
                e3.append(min(max(np.abs(Ri - R).max(), np.abs(ti - t).max()) for Ri, ti in zip(Rs, ts)))
                for Ri in Rs:
                    assert np.abs(Ri @ Ri.T - np.eye(3)).max() < 1e-9 and np.linalg.det(Ri) > 0
                ok, R4, t4 = O.solve_p3p4(Xw, uv, K)
                assert ok
                e4.append(max(np.abs(R4 - R).max(), np.abs(t4 - t).max()))
                assert np.abs(O.rodrigues_m2v(R) - rv).max() < 1e-9
        finally:
            O.p3p_set_polish(0)
        assert len(e3) > 200
        for e in (np.array(e3), np.array(e4)):
            assert np.median(e) < 1e-9 and np.percentile(e, 85) < 1e-6 and np.percentile(e, 97) < 1e-2
            if polish:
                assert e.max() < 1e-3 and np.percentile(e, 99) < 1e-5      # every case, polished
            else:
                assert e.max() < 1.0 and (e > 1e-2).sum() <= 2             # the closed forms' failures are rare


def test_ransac_pnp_finds_pose_and_is_deterministic():
    g = np.load(os.path.join(GOLD, "ransac_400.npz"))
    r = O.solve_pnp_ransac(g["X3"], g["x2"], g["K"])
    assert r["ok"] and r["n_inliers"] == int(g["n_inliers"]) and r["iters"] == int(g["iters"])
    assert np.array_equal(r["rvec"], g["rvec"]) and np.array_equal(r["tvec"], g["tvec"])
    assert np.array_equal(r["mask"], g["mask"]) and np.array_equal(r["log"], g["log"])
    assert np.abs(r["rvec"] - np.array([0.004, -0.012, 0.002])).max() < 2e-3
    assert np.abs(r["tvec"] - np.array([0.02, -0.01, -0.8])).max() < 2e-2
    # score of the returned model equals the reported inlier count
    Rt = np.concatenate([O.rodrigues_v2m(r["rvec"]).reshape(9), r["tvec"]])
    thr_sq = np.float32(float(np.float32(0.1)) ** 2)
    cnt, mask, _ = O.project_score(g["X3"], g["x2"], Rt, g["K"], thr_sq)
    assert cnt == r["n_inliers"] and np.array_equal(mask, r["mask"])
    r0 = O.solve_pnp_ransac(g["X3"][:3], g["x2"][:3], g["K"])
    assert not r0["ok"]


def test_golden_pyramid_lk_corners():
    g = np.load(os.path.join(GOLD, "pyramid_197x123.npz"))
    p = O.Pyramid(g["img"], 4, 21)
    assert p.max_level == int(g["max_level"]) == 2
    for l in range(p.max_level + 1):
        assert np.array_equal(p.level(l), g["level%d" % l]) and np.array_equal(p.deriv(l), g["scharr%d" % l])
    g = np.load(os.path.join(GOLD, "lk_320x240.npz"))
    out, st, err = O.pyrlk(g["prev"], g["next"], g["pts"])
    assert np.array_equal(out, g["out"]) and np.array_equal(st, g["status"]) and np.array_equal(err, g["err"])
    assert 0.3 < st.mean() < 0.95            # the fixture exercises both outcomes
    g = np.load(os.path.join(GOLD, "corners_260x180.npz"))
    for i, r in enumerate(g["rois"]):
        xy, rs = O.good_features(g["img"], tuple(int(v) for v in r), 100, 0.01, 10.0, True, 0.0)
        assert np.array_equal(xy, g["harris_xy%d" % i]) and np.array_equal(rs, g["harris_resp%d" % i])
        xy, rs = O.good_features(g["img"], tuple(int(v) for v in r), 50, 0.01, 5.0, False, 0.0)
        assert np.array_equal(xy, g["mineig_xy%d" % i]) and np.array_equal(rs, g["mineig_resp%d" % i])


def test_lk_on_noise_is_sensitive_to_summation_order():
    """On a frame of shifted white noise LK does not settle, and the last bits of its float sums decide where
    it ends: the oracle against ITSELF with the 441 terms of each window sum added in another order (four
    accumulators as a SIMD build has them, or pairwise) moves 5 % of the tracked points by more than 0.05 px,
    with every integer term identical.  This is the frame of the randomised sweep on which the GPU kernel,
    whose sums run in a fixed cross-lane tree, shows 0.059 px at the 95 % quantile against the serial oracle:
    a property of the input, not of a kernel.  (Bit-identical in mode 0: the golden LK vectors.)"""
    g = np.load(os.path.join(GOLD, "lk_noise_164x140.npz"))
    ref = O.pyrlk(g["prev"], g["next"], g["pts"], nthreads=8)
    try:
        for mode in (1, 2):
            O.set_lk_sum_mode(mode)
            out, st, _ = O.pyrlk(g["prev"], g["next"], g["pts"], nthreads=8)
            assert (st == ref[1]).mean() >= 0.99
            both = (st == 1) & (ref[1] == 1)
            d = np.abs(out - ref[0]).max(1)[both]
            assert both.sum() > 200
            assert np.median(d) < 1e-3                      # most points are untouched ...
            assert 0.03 < np.quantile(d, 0.95) < 0.2         # ... 5 % move by about the 0.05 px bar
            assert (d > 0.05).sum() >= 8 and d.max() > 0.3   # and a few by far more
    finally:
        O.set_lk_sum_mode(0)
