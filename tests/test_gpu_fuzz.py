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
        # exact window sums (the contexts of these tests ask for them): the same bits as the oracle summing exactly, on every kind of frame -- white
        # noise, where LK runs into its iteration limit and any difference in a sum's last bit grows over up to 150
        # steps, included
        go, gs, ge = ctx.pyrlk(0, 1, pts)
        oo, os_, oe = oracle.pyrlk(a, b, pts, nthreads=8, sums=oracle.SUMS_EXACT)
        assert np.array_equal(gs, os_), (w, h, int((gs != os_).sum()))
        assert np.array_equal(go.view(np.uint32), oo.view(np.uint32)), (w, h, float(np.abs(go - oo).max()))
        assert np.array_equal(ge.view(np.uint32), oe.view(np.uint32)), (w, h)
        checked += int((gs == 1).sum())
        # float-tree sums against the serial float accumulator (OpenCV's scalar order), on frames with real motion,
        # where LK converges: the task's bars (on noise the two orders part by construction, see
        # test_lk_tree_sums_on_noise_within_order_sensitivity)
        if i % 3 == 0:
            to, ts, te = ctx.pyrlk(0, 1, pts, klib.lk_params(sums=klib.LK_SUMS_TREE))
            so, ss, se = oracle.pyrlk(a, b, pts, nthreads=8, sums=oracle.SUMS_SERIAL)
            assert (ts == ss).mean() >= 0.995, (w, h, (ts == ss).mean())
            both = (ts == 1) & (ss == 1)
            if both.any():
                d = np.abs(to - so).max(1)[both]
                de = np.abs(te - se)[both]
                # (points whose window hangs over the image edge may run into the iteration limit on the
                # mirrored border; the order of the float sums then shows, as on noise)
                inside = ((pts[:, 0] >= 11) & (pts[:, 0] <= w - 12) & (pts[:, 1] >= 11) & (pts[:, 1] <= h - 12))[both]
                assert d[inside].max(initial=0) <= 0.05, (w, h, d[inside].max(initial=0))
                assert de[inside].max(initial=0) <= 0.05, (w, h, de[inside].max(initial=0))
                assert np.quantile(d, 0.99) <= 0.05, (w, h, np.quantile(d, 0.99))
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


def test_lk_tree_sums_on_noise_within_order_sensitivity(make_ctx, oracle, klib):
    """Float-tree sums on the worst frame an earlier sweep found (seed 1111, 164x140 shifted noise): the kernel's
    distance from the serial oracle is no larger than the distance of the oracle from itself under a different order
    of the float additions (tests/test_oracle_kat.py::test_lk_on_noise_is_sensitive_to_summation_order).  With exact
    sums the same frame is bit-equal to the oracle (test_gpu_parity.py::test_lk_golden_and_edge_cases)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_noise_164x140.npz"))
    a, b, pts = g["prev"], g["next"], g["pts"]
    h, w = a.shape
    ctx = make_ctx(width=w, height=h, num_slots=2, max_points=1024)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.build_pyramids([0, 1])
    go, gs, _ = ctx.pyrlk(0, 1, pts, klib.lk_params(sums=klib.LK_SUMS_TREE))
    oo, os_, _ = oracle.pyrlk(a, b, pts, nthreads=8, sums=oracle.SUMS_SERIAL)
    po, ps, _ = oracle.pyrlk(a, b, pts, nthreads=8, sums=oracle.SUMS_PAIRWISE)
    assert (gs == os_).mean() >= 0.99
    both = (gs == 1) & (os_ == 1) & (ps == 1)
    d_gpu = np.abs(go - oo).max(1)[both]
    d_ord = np.abs(po - oo).max(1)[both]
    print("q95: gpu vs oracle %.4f px, oracle pairwise vs oracle serial %.4f px; points beyond 0.05 px: %d vs %d"
          % (np.quantile(d_gpu, 0.95), np.quantile(d_ord, 0.95), (d_gpu > 0.05).sum(), (d_ord > 0.05).sum()))
    assert np.quantile(d_gpu, 0.90) <= 0.05
    assert np.quantile(d_gpu, 0.95) <= 1.5 * np.quantile(d_ord, 0.95)
    assert (d_gpu > 0.05).sum() <= 1.5 * (d_ord > 0.05).sum() + 2
