"""Randomised differential tests of the HIP path against the CPU oracle through the C ABI: image
sizes from tiny to medium (so the level count, the block mapping and every edge case of the padded
planes vary), random motion, random content (including pure noise, which keeps LK iterating to its
limit and moving its window), points inside, on and outside the image."""
import os

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def _sizes(rng, n):
    out = [(22, 22), (23, 45), (64, 48), (43, 301)]  # below / at the pyramid rule's edges, very thin
    while len(out) < n:
        out.append((int(rng.integers(24, 720)), int(rng.integers(24, 520))))
    return out


def _images(rng, w, h, kind):
    if kind == 0:   # smooth texture, sub-pixel to several-pixel shift
        dx, dy = rng.uniform(-6, 6, 2)
        return synth.texture_pair(w, h, int(rng.integers(1, 1 << 30)), float(dx), float(dy), noise=True)
    if kind == 1:   # white noise in both frames: no consistent motion, long iteration chains
        return (rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h, w), dtype=np.uint8))
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)   # integer translation of noise with saturated patches
    a[: h // 3, : w // 2] = 255
    a[h // 2:, w // 2:] = 0
    s = int(rng.integers(-3, 4))
    return a, np.roll(a, (s, -s), (0, 1))


def test_pyramid_scharr_lk_random_sizes(make_ctx, oracle, klib):
    # KTLVO_FUZZ_SEED / KTLVO_FUZZ_CASES widen the sweep for an occasional long run
    rng = np.random.default_rng(int(os.environ.get("KTLVO_FUZZ_SEED", "20240611")))
    checked = 0
    for i, (w, h) in enumerate(_sizes(rng, int(os.environ.get("KTLVO_FUZZ_CASES", "14")))):
        a, b = _images(rng, w, h, i % 3)
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        ctx = make_ctx(width=w, height=h, num_slots=2, max_points=1024)
        ctx.upload_image(0, a)
        ctx.upload_image(1, b)
        ctx.build_pyramids([0, 1])
        po = oracle.Pyramid(a, 4, 21)
        assert ctx.pyramid_levels(0) == po.max_level, (w, h)
        for l in range(po.max_level + 1):
            lv = po.level(l)
            assert np.array_equal(ctx.download_level(0, l), lv), (w, h, l)
            assert np.array_equal(ctx.download_scharr(0, l, lv.shape[1], lv.shape[0]), po.deriv(l)), (w, h, l)
        n = 600
        pts = np.stack([rng.uniform(-30, w + 30, n), rng.uniform(-30, h + 30, n)], 1).astype(np.float32)
        pts[:8] = [[0, 0], [w - 1, h - 1], [w, h], [-21, -21], [w - 0.5, 0.25], [0.25, h - 0.5], [w / 2, h / 2],
                   [10.5, 10.5]]
        go, gs, ge = ctx.pyrlk(0, 1, pts)
        oo, os_, oe = oracle.pyrlk(a, b, pts, nthreads=8)
        agree = gs == os_
        # a status may flip where a float sum sits on a threshold; positions of agreeing points must match
        assert agree.mean() >= 0.99, (w, h, agree.mean())
        both = agree & (gs == 1)
        if both.any():
            d = np.abs(go - oo).max(1)[both]
            de = np.abs(ge - oe)[both]
            if i % 3 == 0:   # real motion: LK converges, everything agrees closely
                # (points whose window hangs over the image edge may run into the iteration limit on the
                # mirrored border; the order of the float sums then shows, as on noise)
                inside = ((pts[:, 0] >= 11) & (pts[:, 0] <= w - 12) & (pts[:, 1] >= 11) & (pts[:, 1] <= h - 12))[both]
                assert d[inside].max(initial=0) <= 0.05, (w, h, d[inside].max(initial=0))
                assert de[inside].max(initial=0) <= 0.05, (w, h, de[inside].max(initial=0))
                assert np.quantile(d, 0.99) <= 0.05, (w, h, np.quantile(d, 0.99))
                assert np.all(de <= 0.05 + 200.0 * d), (w, h, float((de - 200.0 * d).max()))
            else:
                # noise: LK does not converge, so the last-bit differences of the float sums (the kernel
                # adds 63 integer partials in a tree, the oracle 441 floats serially) grow over up to 150
                # steps; the bulk must still agree, and the residual must agree where the position does
                # (on white noise it changes by up to a hundred-odd grey levels per pixel of displacement)
                if np.quantile(d, 0.95) > 0.05:
                    # a frame on which the order of the float additions alone moves that many points that far
                    # (a long sweep finds one in a few hundred): the kernel must then be no further from the
                    # serial oracle than the oracle is from itself with pairwise sums
                    oracle.set_lk_sum_mode(2)
                    try:
                        po_, ps_, _ = oracle.pyrlk(a, b, pts, nthreads=8)
                    finally:
                        oracle.set_lk_sum_mode(0)
                    ok = both & (ps_ == 1)
                    d_gpu, d_ord = np.abs(go - oo).max(1)[ok], np.abs(po_ - oo).max(1)[ok]
                    # as many points leave the serial oracle's tracks as leave them when the oracle itself sums
                    # pairwise, and the kernel's tracks lie with one of the oracle's two (95 % within 0.05 px of the
                    # nearer).  Not the ratio of the two distances' 95th percentiles: where about one point in twenty
                    # diverges that percentile falls on the edge of the diverging set, among values that jump from
                    # 0.003 to 0.1 px from one point to the next (seed 777777, 64 x 48: 11 and 10 of 177 points
                    # diverge, percentiles 0.121 and 0.055 px, while only 2 points are beyond 0.05 px of the
                    # pairwise oracle's)
                    assert (d_gpu > 0.05).sum() <= 1.5 * (d_ord > 0.05).sum() + 2, (w, h, (d_gpu > 0.05).sum(), (d_ord > 0.05).sum())
                    d_near = np.minimum(d_gpu, np.abs(go - po_).max(1)[ok])
                    assert np.quantile(d_near, 0.95) <= 0.05, (w, h, np.quantile(d_near, 0.95))
                assert np.all(de <= 0.05 + 200.0 * d), (w, h, float((de - 200.0 * d).max()))
            checked += int(both.sum())
        lost = agree & (gs == 0)
        assert np.array_equal(ge[lost], oe[lost])
        ctx.close()
    assert checked > 1500


def test_batched_pyramid_builds_random(make_ctx, oracle):
    """Pyramid + Scharr builds of 1 to 14 images at random sizes in one call: every route of the small levels (a launch
    per level, levels 2-4 of a small batch from one launch, levels 3-4 of a large batch from one launch, fewer than five
    levels) against the oracle, every level and derivative plane of every image."""
    rng = np.random.default_rng(int(os.environ.get("KTLVO_FUZZ_SEED", "31337")))
    for i in range(int(os.environ.get("KTLVO_FUZZ_CASES", "6"))):
        w, h = int(rng.integers(24, 900)), int(rng.integers(24, 600))
        n = int(rng.integers(1, 15))
        imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        ctx = make_ctx(width=w, height=h, num_slots=n)
        ctx.upload_images(list(range(n)), imgs)
        ctx.build_pyramids(list(range(n)))
        for s_ in sorted(set([0, n - 1, int(rng.integers(0, n))])):
            po = oracle.Pyramid(imgs[s_], 4, 21)
            assert ctx.pyramid_levels(s_) == po.max_level, (w, h, n)
            for l in range(po.max_level + 1):
                lv = po.level(l)
                assert np.array_equal(ctx.download_level(s_, l), lv), (w, h, n, s_, l)
                assert np.array_equal(ctx.download_scharr(s_, l, lv.shape[1], lv.shape[0]), po.deriv(l)), (w, h, n, s_, l)
        ctx.close()


def test_corners_and_gates_random(make_ctx, oracle, klib):
    rng = np.random.default_rng(int(os.environ.get("KTLVO_FUZZ_SEED", "77")))
    for i in range(int(os.environ.get("KTLVO_FUZZ_CASES", "6"))):
        w, h = int(rng.integers(80, 700)), int(rng.integers(60, 500))
        img = synth.texture_image(w, h, seed=int(rng.integers(1, 1 << 30)), periods=(31.0, 11.0, 5.0), noise=True)
        ctx = make_ctx(width=w, height=h, num_slots=1)
        ctx.upload_image(0, img)
        rois = []
        for _ in range(12):
            rw, rh = int(rng.integers(8, min(64, w))), int(rng.integers(8, min(64, h)))
            rois.append((int(rng.integers(0, w - rw + 1)), int(rng.integers(0, h - rh + 1)), rw, rh))
        harris = i % 2
        gp = klib.gftt_params(int(rng.integers(1, 120)), 0.01, float(rng.choice([1.0, 3.0, 10.0])), 3, harris,
                              0.04 if i % 3 == 0 else 0.0)
        got = ctx.detect_gftt(0, rois, gp)
        for (rx, ry, rw, rh), (xy, resp) in zip(rois, got):
            oxy, oresp = oracle.good_features(img, (rx, ry, rw, rh), gp.max_corners, gp.quality, gp.min_dist,
                                              bool(harris), gp.harris_k)
            oxy, oresp = np.asarray(oxy, np.float32).reshape(-1, 2), np.asarray(oresp, np.float32)
            assert np.array_equal(xy, oxy), (w, h, rx, ry, rw, rh)
            assert np.array_equal(resp, oresp)
        # gates on random tracks
        n = int(rng.integers(1, 900))
        f1 = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
        f2 = f1 + rng.normal(0, 40, (n, 2)).astype(np.float32)
        chk = f1 + rng.normal(0, 1.0, (n, 2)).astype(np.float32)
        yth = int(rng.integers(1, 60))
        keep = ctx.fb_gate(f1, f2, chk, yth)
        okeep = oracle.fb_gate(f1, f2, chk, w, h, 30, 30, 100.0, yth)
        assert np.array_equal(np.asarray(keep), np.flatnonzero(okeep))
        ckeep = ctx.circular_gate(f1, chk)
        assert np.array_equal(np.asarray(ckeep), np.flatnonzero(oracle.circular_gate(f1, chk)))
        ctx.close()


def test_large_rois_random(make_ctx, oracle, klib):
    """ROIs beyond 4096 pixels (the register candidates kernel, the LDS or the global sort, the greedy pass with its grid
    in LDS or in global memory) at random image sizes, ROI placements, corner counts, qualities and minimum distances:
    points, order and responses identical to the oracle's goodFeaturesToTrack."""
    rng = np.random.default_rng(int(os.environ.get("KTLVO_FUZZ_SEED", "4711")))
    for i in range(int(os.environ.get("KTLVO_FUZZ_CASES", "5"))):
        w, h = int(rng.integers(100, 900)), int(rng.integers(80, 600))
        img = synth.texture_image(w, h, seed=int(rng.integers(1, 1 << 30)),
                                  periods=tuple(float(v) for v in rng.choice([5.0, 9.0, 17.0, 31.0, 47.0], 3, replace=False)),
                                  noise=bool(i % 2))
        ctx = make_ctx(width=w, height=h, num_slots=1)
        ctx.upload_image(0, img)
        rois = [(0, 0, w, h)]
        for _ in range(2):
            rw, rh = int(rng.integers(65, w + 1)), int(rng.integers(65, h + 1))
            rois.append((int(rng.integers(0, w - rw + 1)), int(rng.integers(0, h - rh + 1)), rw, rh))
        harris = int(rng.integers(0, 2))
        gp = klib.gftt_params(int(rng.choice([1, 50, 700, 3000, 20000])), float(rng.choice([0.001, 0.01, 0.1])),
                              float(rng.choice([0.0, 1.0, 2.5, 5.0, 10.0, 17.0, 40.0])), 3, harris,
                              0.04 if harris and i % 3 == 0 else 0.0)
        got = ctx.detect_gftt(0, rois, gp)
        for roi, (xy, resp) in zip(rois, got):
            oxy, oresp = oracle.good_features(img, roi, gp.max_corners, gp.quality, gp.min_dist, bool(harris), gp.harris_k)
            oxy, oresp = np.asarray(oxy, np.float32).reshape(-1, 2), np.asarray(oresp, np.float32)
            assert np.array_equal(xy, oxy), (w, h, roi, gp.max_corners, gp.quality, gp.min_dist, harris, len(xy), len(oxy))
            assert np.array_equal(resp, oresp), (w, h, roi)
        (bxy, bresp), = ctx.detect_gftt_frames([0], gp)                      # the batched route on the whole frame
        assert np.array_equal(bxy, got[0][0]) and np.array_equal(bresp, got[0][1]), (w, h)
        ctx.close()


def test_batched_detection_random(make_ctx, oracle, klib):
    """Batches of 2 to 6 frames of one random size (textures, a flat frame, white noise now and then: key counts from
    none to beyond what the per-frame LDS sort holds) through ktlvo_detect_gftt_frames at random regions and parameters:
    frame by frame the oracle's goodFeaturesToTrack."""
    rng = np.random.default_rng(int(os.environ.get("KTLVO_FUZZ_SEED", "8086")))
    for i in range(int(os.environ.get("KTLVO_FUZZ_CASES", "4"))):
        w, h = int(rng.integers(100, 1300)), int(rng.integers(80, 500))
        n = int(rng.integers(2, 7))
        imgs = []
        for f in range(n):
            kind = int(rng.integers(0, 8))
            if kind == 0:
                imgs.append(np.full((h, w), int(rng.integers(0, 256)), np.uint8))
            elif kind == 1:
                imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
            else:
                imgs.append(synth.texture_image(w, h, seed=int(rng.integers(1, 1 << 30)), periods=(31.0, 11.0, 5.0), noise=bool(kind % 2)))
        ctx = make_ctx(width=w, height=h, num_slots=n)
        ctx.upload_images(list(range(n)), np.stack(imgs))
        roi = None
        if i % 3 == 2:
            rw, rh = int(rng.integers(65, w + 1)), int(rng.integers(65, h + 1))
            roi = (int(rng.integers(0, w - rw + 1)), int(rng.integers(0, h - rh + 1)), rw, rh)
        harris = int(rng.integers(0, 2))
        gp = klib.gftt_params(int(rng.choice([50, 700, 3000, 20000])), float(rng.choice([0.001, 0.01, 0.1])),
                              float(rng.choice([0.0, 1.0, 3.0, 7.0, 10.0, 25.0])), 3, harris, 0.0)
        order = [int(v) for v in rng.permutation(n)]
        for s_, (xy, resp) in zip(order, ctx.detect_gftt_frames(order, gp, roi)):
            oxy, oresp = oracle.good_features(imgs[s_], roi, gp.max_corners, gp.quality, gp.min_dist, bool(harris), 0.0)
            oxy, oresp = np.asarray(oxy, np.float32).reshape(-1, 2), np.asarray(oresp, np.float32)
            assert np.array_equal(xy, oxy), (w, h, n, s_, roi, gp.max_corners, gp.quality, gp.min_dist, harris, len(xy), len(oxy))
            assert np.array_equal(resp, oresp), (w, h, n, s_)
        ctx.close()


def test_greedy_relaxation_random(make_ctx, oracle, klib):
    """Frames whose cell grid does not fit LDS (small minimum distances on mid-sized frames, whole and fractional, also
    2.5, whose cell is narrower than the distance as in OpenCV): the greedy pass runs as a relaxation over all candidates
    at once.  One to three frames a call through ktlvo_detect_gftt_frames, the first also through ktlvo_detect_gftt,
    against the oracle's goodFeaturesToTrack."""
    rng = np.random.default_rng(int(os.environ.get("KTLVO_FUZZ_SEED", "8087")))
    for i in range(int(os.environ.get("KTLVO_FUZZ_CASES", "4"))):
        w, h = int(rng.integers(500, 1100)), int(rng.integers(300, 700))
        n = int(rng.integers(1, 4))
        imgs = []
        for f in range(n):
            kind = int(rng.integers(0, 8))
            if kind == 0:
                imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
            else:
                imgs.append(synth.texture_image(w, h, seed=int(rng.integers(1, 1 << 30)), periods=(31.0, 11.0, 5.0), noise=bool(kind % 2)))
        ctx = make_ctx(width=w, height=h, num_slots=n, max_points=50000)
        ctx.upload_images(list(range(n)), np.stack(imgs))
        harris = int(rng.integers(0, 2))
        gp = klib.gftt_params(int(rng.choice([300, 5000, 50000])), float(rng.choice([0.0005, 0.01, 0.05])),
                              float(rng.choice([1.0, 1.5, 2.0, 2.5, 2.9, 3.5, 4.2])), 3, harris, 0.0)
        want = []
        for f in range(n):
            oxy, oresp = oracle.good_features(imgs[f], None, gp.max_corners, gp.quality, gp.min_dist, bool(harris), 0.0)
            want.append((np.asarray(oxy, np.float32).reshape(-1, 2), np.asarray(oresp, np.float32)))
        for f, (xy, resp) in enumerate(ctx.detect_gftt_frames(list(range(n)), gp)):
            assert np.array_equal(xy, want[f][0]), (w, h, n, f, gp.max_corners, gp.quality, gp.min_dist, harris, len(xy), len(want[f][0]))
            assert np.array_equal(resp, want[f][1]), (w, h, n, f)
        xy, resp = ctx.detect_gftt(0, [(0, 0, w, h)], gp)[0]
        assert np.array_equal(xy, want[0][0]) and np.array_equal(resp, want[0][1]), (w, h, gp.max_corners, gp.quality, gp.min_dist, harris)
        ctx.close()


def test_lk_noise_frame_within_order_sensitivity(make_ctx, oracle):
    """The sweep's worst frame (seed 1111, 164x140 shifted noise): the kernel's distance from the serial
    oracle is no larger than the distance of the oracle from itself under a different order of the float
    additions (tests/test_oracle_kat.py::test_lk_on_noise_is_sensitive_to_summation_order)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_noise_164x140.npz"))
    a, b, pts = g["prev"], g["next"], g["pts"]
    h, w = a.shape
    ctx = make_ctx(width=w, height=h, num_slots=2, max_points=1024)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.build_pyramids([0, 1])
    go, gs, _ = ctx.pyrlk(0, 1, pts)
    oo, os_, _ = oracle.pyrlk(a, b, pts, nthreads=8)
    oracle.set_lk_sum_mode(2)
    try:
        po, ps, _ = oracle.pyrlk(a, b, pts, nthreads=8)
    finally:
        oracle.set_lk_sum_mode(0)
    assert (gs == os_).mean() >= 0.99
    both = (gs == 1) & (os_ == 1) & (ps == 1)
    d_gpu = np.abs(go - oo).max(1)[both]
    d_ord = np.abs(po - oo).max(1)[both]
    print("q95: gpu vs oracle %.4f px, oracle pairwise vs oracle serial %.4f px; points beyond 0.05 px: %d vs %d"
          % (np.quantile(d_gpu, 0.95), np.quantile(d_ord, 0.95), (d_gpu > 0.05).sum(), (d_ord > 0.05).sum()))
    assert np.quantile(d_gpu, 0.90) <= 0.05
    assert np.quantile(d_gpu, 0.95) <= 1.5 * np.quantile(d_ord, 0.95)
    assert (d_gpu > 0.05).sum() <= 1.5 * (d_ord > 0.05).sum() + 2
