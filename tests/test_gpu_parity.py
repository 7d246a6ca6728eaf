"""GPU parity tests: every call goes through the C ABI of libktlvo_hip.so and is compared with
the CPU oracle on the same seeded inputs and with the committed golden vectors.

Bars: bit-exact for integer/byte/index work (pyramid, Scharr, corner sets, gates, inlier
counts).  LK has two readings of its five window sums per level.  With exact sums (integer totals
rounded to float once, KTLVO_LK_SUMS_EXACT: what make_ctx asks for unless a test says otherwise) positions, status flags and
errors are BIT-EQUAL to the oracle summing the same way, whatever the image.  With the float tree
(KTLVO_LK_SUMS_TREE, the library's default) against the oracle's serial float accumulator (OpenCV's scalar order) the
bars are the task statement's: positions within 0.05 px, >= 99.5 % identical status flags."""
import os

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
LK_TOL_PX = 0.05
LK_STATUS_AGREE = 0.995


def _lk_compare(go, gs, ge, oo, os_, oe):
    assert (gs == os_).mean() >= LK_STATUS_AGREE
    both = (gs == 1) & (os_ == 1)
    if both.any():
        d = np.abs(go - oo).max(1)[both]
        assert d.max() <= LK_TOL_PX, d.max()
        assert np.abs(ge - oe)[both].max() <= 0.05
    return both


def _lk_identical(go, gs, ge, oo, os_, oe):
    """Same bits: positions (lost points included: they keep the last level's estimate), status, error."""
    assert np.array_equal(gs, os_), int((gs != os_).sum())
    bad = (go.view(np.uint32) != oo.view(np.uint32)).any(1)
    assert not bad.any(), (int(bad.sum()), float(np.abs(go - oo).max()))
    assert np.array_equal(ge.view(np.uint32), oe.view(np.uint32))


def _lk_both(ctx, oracle, klib, a, b, pts, slots=(0, 1), nthreads=8, **kw):
    """Both readings of the window sums against the oracle's: exact totals bit for bit, the float tree within the
    bars of the serial order.  Returns the exact-sum results and the mask of points tracked by both sides."""
    go, gs, ge = ctx.pyrlk(slots[0], slots[1], pts, klib.lk_params(sums=klib.LK_SUMS_EXACT, **kw))
    _lk_identical(go, gs, ge, *oracle.pyrlk(a, b, pts, nthreads=nthreads, sums=oracle.SUMS_EXACT, **kw))
    to, ts, te = ctx.pyrlk(slots[0], slots[1], pts, klib.lk_params(sums=klib.LK_SUMS_TREE, **kw))
    _lk_compare(to, ts, te, *oracle.pyrlk(a, b, pts, nthreads=nthreads, sums=oracle.SUMS_SERIAL, **kw))
    _lk_compare(to, ts, te, go, gs, ge)       # and the library's two readings within the same bars of each other
    return go, gs, ge, gs == 1


# ---------------------------------------------------------------- pyramid / Scharr ------------
@pytest.mark.parametrize("size", [(1242, 375), (1241, 376), (197, 123), (640, 481)])
def test_pyramid_and_scharr_bit_exact(make_ctx, oracle, size):
    w, h = size
    img = synth.texture_image(w, h, seed=w + h, periods=(24.0, 9.0, 4.0), noise=True)
    ctx = make_ctx(width=w, height=h)
    ctx.upload_image(0, img)
    ctx.build_pyramid(0)
    p = oracle.Pyramid(img, 4, 21)
    assert ctx.pyramid_levels(0) == p.max_level
    for l in range(p.max_level + 1):
        o = p.level(l)
        assert np.array_equal(ctx.download_level(0, l), o), l
        assert np.array_equal(ctx.download_scharr(0, l, o.shape[1], o.shape[0]), p.deriv(l)), l


def test_pyramid_kat_and_golden(make_ctx):
    img = np.zeros((64, 96), np.uint8)
    img[20, 30] = 255
    ctx = make_ctx(width=96, height=64)
    ctx.upload_image(0, img)
    ctx.build_pyramid(0)
    assert ctx.pyramid_levels(0) == 1
    l1 = ctx.download_level(0, 1)
    assert l1[10, 15] == 36 and l1[10, 14] == 6 and l1[9, 14] == 1
    g = np.load(os.path.join(GOLD, "pyramid_197x123.npz"))
    ctx = make_ctx(width=197, height=123)
    ctx.upload_image(0, g["img"])
    ctx.build_pyramid(0)
    for l in range(int(g["max_level"]) + 1):
        assert np.array_equal(ctx.download_level(0, l), g["level%d" % l])
        lv = g["level%d" % l]
        assert np.array_equal(ctx.download_scharr(0, l, lv.shape[1], lv.shape[0]), g["scharr%d" % l])


def test_upload_from_padded_and_pinned_rows(make_ctx):
    """The three ways a frame can arrive (packed pageable, padded rows, pinned memory read by DMA in
    place, padded or not) give the same planes."""
    import torch
    W, H = 333, 77
    img = synth.texture_image(W, H, seed=9, noise=True)
    ctx = make_ctx(width=W, height=H, num_slots=5)
    padded = np.zeros((H, W + 59), np.uint8)
    padded[:, :W] = img
    pin_packed = torch.from_numpy(img.copy()).pin_memory()
    pin_padded = torch.from_numpy(padded.copy()).pin_memory()
    ctx.upload_image(0, img)
    ctx.upload_image(1, padded[:, :W])
    ctx.upload_image(2, pin_packed.numpy())
    ctx.upload_image(3, pin_padded.numpy()[:, :W])
    ctx.build_pyramids([0, 1, 2, 3])
    for l in range(ctx.pyramid_levels(0) + 1):
        ref = ctx.download_level(0, l)
        for s in (1, 2, 3):
            assert np.array_equal(ctx.download_level(s, l), ref), (s, l)
    assert np.array_equal(ctx.download_level(0, 0), img)


def test_pyramid_batched_build_equals_single(make_ctx):
    ctx = make_ctx(width=320, height=240, num_slots=6)
    imgs = [synth.texture_image(320, 240, seed=s) for s in range(6)]
    for s, im in enumerate(imgs):
        ctx.upload_image(s, im)
    ctx.build_pyramids(list(range(6)))
    ctx2 = make_ctx(width=320, height=240, num_slots=1)
    for s, im in enumerate(imgs):
        ctx2.upload_image(0, im)
        ctx2.build_pyramid(0)
        for l in range(ctx.pyramid_levels(s) + 1):
            assert np.array_equal(ctx.download_level(s, l), ctx2.download_level(0, l))


def test_scharr_large_batch_store_path(make_ctx, oracle):
    """A batch whose Scharr planes exceed 512 MiB is written with non-temporal stores (the benchmark
    step's path); the planes must be the same bits as the small-batch path gives and the oracle derives."""
    W, H, n = 1241, 376, 224                     # 224 x 2.37 MiB of planes = 531 MiB
    ctx = make_ctx(width=W, height=H, num_slots=n)
    imgs = [synth.texture_image(W, H, seed=40 + s, noise=True) for s in range(3)]
    for s in range(n):
        ctx.upload_image(s, imgs[s % 3])
    ctx.build_pyramids(list(range(n)))
    ctx1 = make_ctx(width=W, height=H, num_slots=1)
    for s in (0, 1, 2, n // 2, n - 1):
        ctx1.upload_image(0, imgs[s % 3])
        ctx1.build_pyramid(0)
        po = oracle.Pyramid(imgs[s % 3], 4, 21)
        for l in range(ctx.pyramid_levels(s) + 1):
            lv = po.level(l)
            big = ctx.download_scharr(s, l, lv.shape[1], lv.shape[0])
            assert np.array_equal(big, ctx1.download_scharr(0, l, lv.shape[1], lv.shape[0])), (s, l)
            assert np.array_equal(big, po.deriv(l)), (s, l)


# ---------------------------------------------------------------- LK --------------------------
@pytest.mark.parametrize("shift", [(0.3, -0.2), (3.3, -1.7), (-9.5, 6.25)])
def test_lk_matches_oracle_and_truth(make_ctx, oracle, klib, shift):
    W, H = 1241, 376
    a, b = synth.texture_pair(W, H, 5, shift[0], shift[1], noise=True)
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.build_pyramids([0, 1])
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-5, W + 5, 3000), rng.uniform(-5, H + 5, 3000)], 1).astype(np.float32)
    assert ctx.cfg.lk_sums == klib.LK_SUMS_EXACT and klib.default_config().lk_sums == klib.LK_SUMS_TREE   # the tests' reading; the library's default
    _lk_identical(*ctx.pyrlk(0, 1, pts), *oracle.pyrlk(a, b, pts, nthreads=8, sums=oracle.SUMS_EXACT))
    go, gs, ge, both = _lk_both(ctx, oracle, klib, a, b, pts)
    inner = both & (pts[:, 0] > 40) & (pts[:, 0] < W - 40) & (pts[:, 1] > 40) & (pts[:, 1] < H - 40)
    e = np.abs(go - pts - np.array(shift, np.float32))[inner]
    assert np.median(e) < 0.02 and np.percentile(e, 95) < 0.1


def test_lk_golden_and_edge_cases(make_ctx, klib):
    g = np.load(os.path.join(GOLD, "lk_320x240.npz"))
    ctx = make_ctx(width=320, height=240)
    ctx.upload_image(0, g["prev"])
    ctx.upload_image(1, g["next"])
    ctx.build_pyramids([0, 1])
    _lk_identical(*ctx.pyrlk(0, 1, g["pts"]), g["out_exact"], g["status_exact"], g["err_exact"])
    go, gs, ge = ctx.pyrlk(0, 1, g["pts"], klib.lk_params(sums=klib.LK_SUMS_TREE))
    _lk_compare(go, gs, ge, g["out"], g["status"], g["err"])
    n = np.load(os.path.join(GOLD, "lk_noise_164x140.npz"))    # white noise: LK iterates to its limit everywhere
    nctx = make_ctx(width=164, height=140)
    nctx.upload_image(0, n["prev"])
    nctx.upload_image(1, n["next"])
    nctx.build_pyramids([0, 1])
    _lk_identical(*nctx.pyrlk(0, 1, n["pts"]), n["out_exact"], n["status_exact"], n["err_exact"])
    # empty list is legal
    o, s, e = ctx.pyrlk(0, 1, np.zeros((0, 2), np.float32))
    assert o.shape == (0, 2)
    # far outside the frame: status 0, err 0, position = input
    o, s, e = ctx.pyrlk(0, 1, np.array([[-500.0, -500.0], [5000.0, 100.0]], np.float32))
    assert list(s) == [0, 0] and list(e) == [0, 0]
    # unsupported window is refused, not silently changed
    with pytest.raises(klib.KtlvoError) as ei:
        ctx.pyrlk(0, 1, g["pts"], klib.lk_params(win=15))
    assert ei.value.status == klib.ERR_UNSUPPORTED
    # tracking a slot without an image is a state error
    with pytest.raises(klib.KtlvoError) as ei:
        ctx.pyrlk(0, 3, g["pts"])
    assert ei.value.status == klib.ERR_STATE
    with pytest.raises(klib.KtlvoError) as ei:
        ctx.pyrlk(0, 1, np.zeros((ctx.cfg.max_points + 1, 2), np.float32))
    assert ei.value.status == klib.ERR_CAPACITY


def test_argument_checks_mirror_opencv_assertions(make_ctx, klib):
    ctx = make_ctx(width=160, height=120)
    img = synth.texture_image(160, 120, seed=5, noise=True)
    ctx.upload_image(0, img)
    ctx.upload_image(1, img)
    ctx.build_pyramids([0, 1])
    pts = np.array([[80.0, 60.0]], np.float32)
    for bad in (klib.lk_params(max_level=-1),):
        with pytest.raises(klib.KtlvoError) as ei:
            ctx.pyrlk(0, 1, pts, bad)
        assert ei.value.status == klib.ERR_INVALID
    for bad in (klib.gftt_params(quality=0.0), klib.gftt_params(min_dist=-1.0)):
        with pytest.raises(klib.KtlvoError) as ei:
            ctx.detect_gftt(0, [(0, 0, 64, 64)], bad)
        assert ei.value.status == klib.ERR_INVALID
    with pytest.raises(klib.KtlvoError) as ei:
        ctx.anms(np.zeros((4, 2), np.float32), np.zeros(4, np.float32), -1)
    assert ei.value.status == klib.ERR_INVALID
    xy, resp = ctx.anms(np.zeros((0, 2), np.float32), np.zeros(0, np.float32), 0)   # nothing in, nothing out
    assert len(xy) == 0 and len(resp) == 0
    # the context still works after refused calls
    o, s, e = ctx.pyrlk(0, 1, pts)
    assert s[0] == 1 and np.abs(o - pts).max() < 1e-3


def test_lk_max_level_3_and_iteration_limits(make_ctx, oracle, klib):
    a, b = synth.texture_pair(640, 480, 8, 4.2, 2.1)
    ctx = make_ctx(width=640, height=480)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.build_pyramids([0, 1])
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(0, 640, 1000), rng.uniform(0, 480, 1000)], 1).astype(np.float32)
    for ml, mc, eps in [(3, 30, 0.01), (4, 5, 0.01), (0, 30, 0.03), (2, 1, 0.5)]:
        _lk_both(ctx, oracle, klib, a, b, pts, nthreads=4, max_level=ml, max_count=mc, epsilon=eps)


def test_lk_unusual_thresholds(make_ctx, oracle, klib):
    """Thresholds outside the usual range switch the kernel's quick float tests off (zero or negative
    minEigThreshold, epsilon 0); a threshold inside a flat region's minEig range exercises both the
    quick accept and the quick reject; results must still be the oracle's."""
    a, b = synth.texture_pair(640, 480, 8, 1.3, -0.6, noise=True)
    a = a.copy(); b = b.copy()
    a[:, :200] = (a[:, :200] // 32) * 32 + 7      # weak-texture band: small eigenvalues
    b[:, :200] = (b[:, :200] // 32) * 32 + 7
    ctx = make_ctx(width=640, height=480)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.build_pyramids([0, 1])
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(0, 640, 1500), rng.uniform(0, 480, 1500)], 1).astype(np.float32)
    for thr, eps in [(0.0, 0.01), (-1.0, 0.01), (1e-13, 0.01), (1e-4, 0.0), (5e-3, 0.01), (0.02, 0.03), (0.05, 0.01), (3.0, 0.01)]:
        go, gs, ge, _ = _lk_both(ctx, oracle, klib, a, b, pts, nthreads=4, min_eig_thr=thr, epsilon=eps)
        if 5e-3 <= thr <= 0.05:
            assert 0 < gs.sum() < len(gs)  # the threshold really splits this point set


def test_lk_negative_remainder_weight(make_ctx, oracle, klib):
    """The fourth bilinear weight is 2^14 minus the three rounded ones and can be -1; points whose
    sub-pixel fractions trigger that must still match (signed arithmetic in the samplers)."""
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 0.02, 400000).astype(np.float32)
    b = rng.uniform(0, 0.02, 400000).astype(np.float32)
    one, s = np.float32(1), np.float32(16384)
    w = [np.rint((one - a) * (one - b) * s), np.rint(a * (one - b) * s), np.rint((one - a) * b * s)]
    neg = np.nonzero(s - w[0] - w[1] - w[2] < 0)[0][:500]
    assert len(neg) == 500
    W, H = 640, 480
    img = synth.texture_image(W, H, seed=12)
    ix = rng.integers(40, W - 40, 500).astype(np.float32)
    iy = rng.integers(40, H - 40, 500).astype(np.float32)
    pts = np.stack([ix + a[neg], iy + b[neg]], 1).astype(np.float32)
    fa = (pts[:, 0] - np.float32(10)) - np.floor(pts[:, 0] - np.float32(10))
    assert np.abs(fa - a[neg]).max() < 1e-4                  # the fraction survives the float add
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, img)
    ctx.upload_image(1, img)
    ctx.build_pyramids([0, 1])
    go, gs, ge, _ = _lk_both(ctx, oracle, klib, img, img, pts, nthreads=1, max_level=0)
    assert gs.mean() > 0.9 and (ge[gs == 1] < 0.5).all()     # an unsigned sampler is off by 128 grey levels here


def test_lk_full_size_properties(make_ctx):
    """Size-independent properties at the benchmark's size: zero motion maps every trackable point
    onto itself, and forward-backward tracking returns to the start."""
    W, H = 1241, 376
    scene = synth.CorridorScene(W, H, seed=0x5EED0002)
    a, b = scene.render(0), scene.render(1)
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.upload_image(2, a)
    ctx.build_pyramids([0, 1, 2])
    rng = np.random.default_rng(7)
    pts = np.stack([rng.uniform(40, W - 40, 2000), rng.uniform(40, H - 40, 2000)], 1).astype(np.float32)
    o, s, e = ctx.pyrlk(0, 2, pts)
    assert (np.abs(o - pts)[s == 1] < 1e-3).all() and (e[s == 1] < 1e-2).all()   # (x - 10) + 10 moves x by ulps
    f, sf, _ = ctx.pyrlk(0, 1, pts)
    r, sr, _ = ctx.pyrlk(1, 0, f)
    ok = (sf == 1) & (sr == 1)
    assert ok.mean() > 0.8
    assert np.median(np.abs(r - pts)[ok]) < 0.05


def test_lk_batch_equals_single_calls(make_ctx, klib):
    import torch
    W, H = 640, 360
    ctx = make_ctx(width=W, height=H, num_slots=6)
    for s in range(6):
        ctx.upload_image(s, synth.texture_image(W, H, 30, dx=0.7 * s, dy=-0.4 * s))
    ctx.build_pyramids(list(range(6)))
    rng = np.random.default_rng(3)
    n = 777
    pts = np.stack([rng.uniform(0, W, (5, n)), rng.uniform(0, H, (5, n))], 2).astype(np.float32)
    dp = torch.from_numpy(pts).cuda()
    dn = torch.zeros_like(dp)
    ds = torch.zeros((5, n), dtype=torch.uint8, device="cuda")
    de = torch.zeros((5, n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.pyrlk_batch_dev([0, 1, 2, 3, 4], [1, 2, 3, 4, 5], dp.data_ptr(), n, dn.data_ptr(), ds.data_ptr(),
                        de.data_ptr(), klib.lk_params())
    ctx.sync()
    for i in range(5):
        o, s, e = ctx.pyrlk(i, i + 1, pts[i])
        assert np.array_equal(o, dn[i].cpu().numpy()) and np.array_equal(s, ds[i].cpu().numpy())
        assert np.array_equal(e, de[i].cpu().numpy())


def test_caller_stream_and_device_buffers(make_ctx, klib):
    """ktlvo_set_stream + the *_dev entry points: frames and point lists that already live in HBM, work
    enqueued on the caller's stream without any synchronisation inside the library."""
    import torch
    W, H = 640, 360
    a, b = synth.texture_pair(W, H, 31, 1.7, -2.2, noise=True)
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.build_pyramids([0, 1])
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(20, W - 20, 700), rng.uniform(20, H - 20, 700)], 1).astype(np.float32)
    ref = ctx.pyrlk(0, 1, pts)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        da, db = torch.from_numpy(a).cuda(non_blocking=True), torch.from_numpy(b).cuda(non_blocking=True)
        dp = torch.from_numpy(pts).cuda(non_blocking=True)
        dn = torch.zeros_like(dp)
        ds = torch.zeros(len(pts), dtype=torch.uint8, device="cuda")
        de = torch.zeros(len(pts), dtype=torch.float32, device="cuda")
        ctx.set_image_dev(2, da.data_ptr(), W, H, W)
        ctx.set_image_dev(3, db.data_ptr(), W, H, W)
        ctx.build_pyramids([2, 3])
        ctx.pyrlk_dev(2, 3, dp.data_ptr(), len(pts), dn.data_ptr(), ds.data_ptr(), de.data_ptr(), klib.lk_params())
    stream.synchronize()
    assert np.array_equal(dn.cpu().numpy(), ref[0]) and np.array_equal(ds.cpu().numpy(), ref[1])
    assert np.array_equal(de.cpu().numpy(), ref[2])
    ctx.set_stream(None)     # back to the context's own stream
    again = ctx.pyrlk(2, 3, pts)
    assert np.array_equal(again[0], ref[0])


@pytest.mark.parametrize("n_pairs,n", [(8, 5), (11, 333), (19, 64), (256, 9)])
def test_lk_batch_block_mapping(make_ctx, klib, n_pairs, n):
    """Pair counts around the kernel's grouping of pairs by 8 (full groups, groups plus left-over
    pairs, the launch capacity) and point counts that are not multiples of the 4 features of a
    workgroup: every pair's results equal a single-pair call's."""
    import torch
    W, H = 320, 200
    ctx = make_ctx(width=W, height=H, num_slots=5)
    for s in range(5):
        ctx.upload_image(s, synth.texture_image(W, H, 41, dx=0.9 * s, dy=0.3 * s))
    ctx.build_pyramids(list(range(5)))
    rng = np.random.default_rng(n_pairs)
    prev = [int(x) for x in rng.integers(0, 5, n_pairs)]
    nxt = [int(x) for x in rng.integers(0, 5, n_pairs)]
    pts = np.stack([rng.uniform(0, W, (n_pairs, n)), rng.uniform(0, H, (n_pairs, n))], 2).astype(np.float32)
    dp = torch.from_numpy(pts).cuda()
    dn = torch.zeros_like(dp)
    ds = torch.full((n_pairs, n), 7, dtype=torch.uint8, device="cuda")
    de = torch.zeros((n_pairs, n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.pyrlk_batch_dev(prev, nxt, dp.data_ptr(), n, dn.data_ptr(), ds.data_ptr(), de.data_ptr(), klib.lk_params())
    ctx.sync()
    gn, gs, ge = dn.cpu().numpy(), ds.cpu().numpy(), de.cpu().numpy()
    for i in range(n_pairs):
        o, st, e = ctx.pyrlk(prev[i], nxt[i], pts[i])
        assert np.array_equal(o, gn[i]) and np.array_equal(st, gs[i]) and np.array_equal(e, ge[i]), i
    if n_pairs == 256:
        with pytest.raises(klib.KtlvoError) as ei:
            ctx.pyrlk_batch_dev(prev + [0], nxt + [1], dp.data_ptr(), n, dn.data_ptr(), ds.data_ptr(), de.data_ptr(),
                                klib.lk_params())
        assert ei.value.status == klib.ERR_CAPACITY


# ---------------------------------------------------------------- corners ---------------------
def _same_corner_set(gxy, gresp, oxy, oresp):
    assert gxy.shape == oxy.shape
    assert np.array_equal(gxy, oxy)            # same corners in the same (response, address) order
    assert np.array_equal(gresp, oresp)


def test_corner_response_bit_exact(make_ctx, oracle):
    img = synth.texture_image(260, 180, seed=31, periods=(20.0, 8.0, 3.0))
    ctx = make_ctx(width=260, height=180)
    ctx.upload_image(0, img)
    for roi in [(30, 30, 59, 39), (0, 0, 260, 180), (201, 141, 59, 39), (5, 7, 33, 70)]:
        for harris, k in [(True, 0.0), (True, 0.04), (False, 0.0)]:
            g = ctx.corner_response(0, roi, harris, k)
            o = oracle.corner_response(img, roi, harris, k)
            assert np.array_equal(g, o), (roi, harris, k)


def test_corner_response_large_rois_both_kernels(make_ctx, oracle):
    """ROIs beyond 4096 pixels take the register kernel (a wave per 58-column strip and band of rows): response maps
    bit-exact against the oracle for widths around the strip boundaries (58 k - 1, 58 k, 58 k + 1, 58 k + 2), heights
    that make one, two and several bands, ROIs inside the image (Sobel reads the parent, the box filter reflects at the
    ROI) and the whole image; the LDS-tile kernel (KTLVO_CANDIDATES_TILES=1) must give the same on the same ROIs."""
    W, H = 400, 300
    img = synth.texture_image(W, H, seed=57, periods=(23.0, 9.0, 4.0), noise=True)
    wave = make_ctx(width=W, height=H)
    os.environ["KTLVO_CANDIDATES_TILES"] = "1"
    try:
        tiles = make_ctx(width=W, height=H)
    finally:
        del os.environ["KTLVO_CANDIDATES_TILES"]
    for ctx in (wave, tiles):
        ctx.upload_image(0, img)
    rois = [(0, 0, W, H), (3, 5, 115, 40), (7, 0, 116, 73), (1, 2, 117, 97), (40, 30, 118, 200), (0, 10, 57, 290),
            (283, 1, 117, 299), (11, 13, 59, 71), (100, 100, 174, 48), (5, 250, 390, 50), (0, 0, 8, 513 // 2 + 30)]
    for roi in rois:
        for harris, k in [(False, 0.0), (True, 0.04)]:
            o = oracle.corner_response(img, roi, harris, k)
            assert np.array_equal(wave.corner_response(0, roi, harris, k), o), ("wave", roi, harris)
            assert np.array_equal(tiles.corner_response(0, roi, harris, k), o), ("tiles", roi, harris)


def test_gftt_large_rois_of_different_heights_in_one_call(make_ctx, oracle, klib):
    """Several large ROIs in one call share one band height (the tallest ROI's): a shorter ROI can then end one row into
    a band, and the band before it reaches the ROI's bottom edge, where the box filter reflects.  (Found by the
    random sweep: 548 x 120 with an ROI of 81 rows lost the corners of row 79.)"""
    w, h = 548, 120
    img = synth.texture_image(w, h, seed=1234, periods=(17.0, 9.0, 5.0), noise=True)
    ctx = make_ctx(width=w, height=h)
    ctx.upload_image(0, img)
    rois = [(0, 0, w, h), (17, 30, 529, 81), (11, 21, 508, 93), (3, 0, 500, 41), (0, 79, w, 41), (100, 5, 300, 82)]
    for gp in (klib.gftt_params(700, 0.1, 1.0, 3, 1, 0.0), klib.gftt_params(3000, 0.01, 5.0, 3, 0, 0.0)):
        for roi, (xy, resp) in zip(rois, ctx.detect_gftt(0, rois, gp)):
            oxy, oresp = oracle.good_features(img, roi, gp.max_corners, gp.quality, gp.min_dist, bool(gp.use_harris), gp.harris_k)
            assert np.array_equal(xy, np.asarray(oxy, np.float32).reshape(-1, 2)), roi
            assert np.array_equal(resp, np.asarray(oresp, np.float32)), roi


def test_greedy_pass_in_rounds_equals_sequential_pass(make_ctx, oracle, klib):
    """Frames whose cell grid does not fit LDS take the greedy pass as a relaxation over all candidates at once; a
    context created with KTLVO_GREEDY_SEQUENTIAL=1 walks them in rank order, a workgroup per frame.  Same corners, order
    and responses, per call and for a batch, with the corner limit cutting early, late and not at all, at small and
    fractional minimum distances, and on white noise (long lists, every cell occupied); and against the oracle."""
    W, H = 960, 540
    rng = np.random.default_rng(404)
    imgs = [synth.texture_image(W, H, seed=90 + i, periods=(19.0, 7.0, 4.0), noise=True) for i in range(3)]
    imgs.append(rng.integers(0, 256, (H, W), dtype=np.uint8))
    imgs.append(np.full((H, W), 13, np.uint8))
    n = len(imgs)
    par = make_ctx(width=W, height=H, num_slots=n, max_points=20000)
    os.environ["KTLVO_GREEDY_SEQUENTIAL"] = "1"
    try:
        seq = make_ctx(width=W, height=H, num_slots=n, max_points=20000)
    finally:
        del os.environ["KTLVO_GREEDY_SEQUENTIAL"]
    for ctx in (par, seq):
        ctx.upload_images(list(range(n)), np.stack(imgs))
    for gp in (klib.gftt_params(8000, 0.01, 5.0, 3, 0, 0.0), klib.gftt_params(300, 0.01, 3.0, 3, 0, 0.0),
               klib.gftt_params(20000, 0.001, 2.5, 3, 0, 0.0), klib.gftt_params(20000, 0.05, 1.0, 3, 1, 0.04),
               klib.gftt_params(5000, 0.01, 4.0, 3, 1, 0.0)):
        a = par.detect_gftt_frames(list(range(n)), gp)
        b = seq.detect_gftt_frames(list(range(n)), gp)
        for f, ((xa, ra), (xb, rb)) in enumerate(zip(a, b)):
            assert np.array_equal(xa, xb) and np.array_equal(ra, rb), (f, gp.max_corners, gp.min_dist, len(xa), len(xb))
        for f in (0, 3):
            (xa, ra), = par.detect_gftt(f, [(0, 0, W, H)], gp)
            assert np.array_equal(xa, a[f][0]) and np.array_equal(ra, a[f][1]), (f, gp.min_dist)
        oxy, oresp = oracle.good_features(imgs[1], None, gp.max_corners, gp.quality, gp.min_dist, bool(gp.use_harris), gp.harris_k)
        assert np.array_equal(a[1][0], np.asarray(oxy, np.float32).reshape(-1, 2)) and np.array_equal(a[1][1], np.asarray(oresp, np.float32))
    assert len(a[4][0]) == 0


def test_sort_by_response_ranges_equals_global_sort(make_ctx, oracle, klib):
    """Frames with more keys above the threshold than one workgroup sorts (here 20,000 to 170,000) are sorted by ranges
    of the response, a workgroup per range; a context created with KTLVO_FRAMES_GLOBAL_SORT=1 runs the bitonic network
    in global memory.  Same corners, order and responses, per call and for a batch; a frame of few distinct responses
    (tens of thousands of equal keys in one bin) is turned down by the plan and takes the global sort in both; against
    the oracle for one frame."""
    W, H = 1600, 900
    rng = np.random.default_rng(77)
    imgs = [synth.texture_image(W, H, seed=190 + i, periods=(13.0, 6.0, 4.0), noise=True) for i in range(2)]
    imgs.append(rng.integers(0, 256, (H, W), dtype=np.uint8))
    yy, xx = np.mgrid[0:H, 0:W]
    imgs.append((((xx // 3 + yy // 3) % 2) * 200).astype(np.uint8)) # a checkerboard of 3 x 3 squares: equal responses
    n = len(imgs)
    lds = make_ctx(width=W, height=H, num_slots=n, max_points=60000)
    os.environ["KTLVO_FRAMES_GLOBAL_SORT"] = "1"
    try:
        glob = make_ctx(width=W, height=H, num_slots=n, max_points=60000)
    finally:
        del os.environ["KTLVO_FRAMES_GLOBAL_SORT"]
    for ctx in (lds, glob):
        ctx.upload_images(list(range(n)), np.stack(imgs))
    for gp, frames in ((klib.gftt_params(60000, 0.001, 2.0, 3, 0, 0.0), [0, 1, 2]), (klib.gftt_params(8000, 0.01, 6.0, 3, 0, 0.0), [2, 0]),
                       (klib.gftt_params(60000, 0.0005, 0.0, 3, 1, 0.04), [1, 2]), (klib.gftt_params(30000, 0.01, 3.0, 3, 0, 0.0), [0, 3, 1])):
        a = lds.detect_gftt_frames(frames, gp)
        b = glob.detect_gftt_frames(frames, gp)
        for f, (xa, ra), (xb, rb) in zip(frames, a, b):
            assert np.array_equal(xa, xb) and np.array_equal(ra, rb), (f, gp.max_corners, gp.min_dist, len(xa), len(xb))
        for k in (0, 1): # per call: the same route queued blind, done again step by step for the frame that is turned down
            (xa, ra), = lds.detect_gftt(frames[k], [(0, 0, W, H)], gp)
            assert np.array_equal(xa, a[k][0]) and np.array_equal(ra, a[k][1]), (frames[k], gp.min_dist)
        f = frames[-1]
        oxy, oresp = oracle.good_features(imgs[f], None, gp.max_corners, gp.quality, gp.min_dist, bool(gp.use_harris), gp.harris_k)
        assert np.array_equal(a[-1][0], np.asarray(oxy, np.float32).reshape(-1, 2)) and np.array_equal(a[-1][1], np.asarray(oresp, np.float32))


def test_relaxation_that_does_not_settle_is_done_again(make_ctx, oracle, klib):
    """A context created with KTLVO_RELAX_TRIES=0 never looks at an undecided predecessor twice within a launch, so the
    two launches queued blind leave most candidates undecided: the flag they raise sends the call through again step by
    step, where the relaxation gets launches until it settles (or, past 32, the walk takes over).  Frames whose grid
    fits LDS and frames whose grid does not, per call and three to a call: the corners of an ordinary context and of the
    oracle."""
    rng = np.random.default_rng(606)
    for (W, H), gp in (((640, 360), klib.gftt_params(1500, 0.01, 8.0, 3, 0, 0.0)), ((960, 540), klib.gftt_params(6000, 0.005, 3.0, 3, 0, 0.0))):
        imgs = [synth.texture_image(W, H, seed=700 + i, periods=(17.0, 8.0, 4.0), noise=True) for i in range(2)]
        imgs.append(rng.integers(0, 256, (H, W), dtype=np.uint8))
        plain = make_ctx(width=W, height=H, num_slots=3, max_points=8192)
        os.environ["KTLVO_RELAX_TRIES"] = "0"
        try:
            slow = make_ctx(width=W, height=H, num_slots=3, max_points=8192)
        finally:
            del os.environ["KTLVO_RELAX_TRIES"]
        for ctx in (plain, slow):
            ctx.upload_images([0, 1, 2], np.stack(imgs))
        a = plain.detect_gftt_frames([0, 1, 2], gp)
        b = slow.detect_gftt_frames([0, 1, 2], gp)
        for f in range(3):
            assert np.array_equal(a[f][0], b[f][0]) and np.array_equal(a[f][1], b[f][1]), (W, H, f, len(a[f][0]), len(b[f][0]))
            (xs, rs), = slow.detect_gftt(f, [(0, 0, W, H)], gp)
            assert np.array_equal(xs, a[f][0]) and np.array_equal(rs, a[f][1]), (W, H, f)
        oxy, oresp = oracle.good_features(imgs[0], None, gp.max_corners, gp.quality, gp.min_dist, False, 0.0)
        assert np.array_equal(a[0][0], np.asarray(oxy, np.float32).reshape(-1, 2)) and np.array_equal(a[0][1], np.asarray(oresp, np.float32))


def test_sub_region_relaxation_equals_walk(make_ctx, oracle, klib):
    """Sub-regions with up to 256 keys above the threshold take the greedy pass as a relaxation inside their workgroup;
    with more keys, or in a context created with KTLVO_GREEDY_SEQUENTIAL=1, one wave walks the sorted keys.  A grid of the
    reference's 59 x 39 regions on a textured frame, regions of a noise patch with 100 to 500 keys, thin and one-pixel
    regions; minimum distances 1 to 10 and 2.5 (a cell narrower than the distance), limits that cut early and not at
    all: both contexts and the oracle agree per region."""
    W, H = 640, 360
    rng = np.random.default_rng(5150)
    img = synth.texture_image(W, H, seed=321, periods=(23.0, 9.0, 5.0), noise=True)
    img[200:300, 400:520] = rng.integers(0, 256, (100, 120), dtype=np.uint8)
    rois = [(10 + 62 * j, 8 + 42 * i, 59, 39) for i in range(4) for j in range(6)]
    rois += [(400, 200, 64, 64), (410, 210, 40, 30), (430, 220, 24, 20), (400, 200, 120, 34), (5, 5, 1, 1), (7, 300, 3, 50), (300, 350, 200, 4)]
    relax = make_ctx(width=W, height=H)
    os.environ["KTLVO_GREEDY_SEQUENTIAL"] = "1"
    try:
        walk = make_ctx(width=W, height=H)
    finally:
        del os.environ["KTLVO_GREEDY_SEQUENTIAL"]
    for ctx in (relax, walk):
        ctx.upload_image(0, img)
    for gp in (klib.gftt_params(100, 0.01, 10.0, 3, 1, 0.04), klib.gftt_params(5, 0.01, 4.0, 3, 0, 0.0),
               klib.gftt_params(400, 0.001, 1.0, 3, 0, 0.0), klib.gftt_params(400, 0.001, 2.5, 3, 0, 0.0),
               klib.gftt_params(60, 0.05, 2.0, 3, 1, 0.04)):
        a = relax.detect_gftt(0, rois, gp)
        b = walk.detect_gftt(0, rois, gp)
        for r, ((xa, ra), (xb, rb)) in enumerate(zip(a, b)):
            assert np.array_equal(xa, xb) and np.array_equal(ra, rb), (rois[r], gp.max_corners, gp.min_dist, len(xa), len(xb))
            oxy, oresp = oracle.good_features(img, rois[r], gp.max_corners, gp.quality, gp.min_dist, bool(gp.use_harris), gp.harris_k)
            assert np.array_equal(xa, np.asarray(oxy, np.float32).reshape(-1, 2)) and np.array_equal(ra, np.asarray(oresp, np.float32)), (rois[r], gp.min_dist)


def test_gftt_rois_match_oracle_and_golden(make_ctx, oracle, klib):
    g = np.load(os.path.join(GOLD, "corners_260x180.npz"))
    ctx = make_ctx(width=260, height=180)
    ctx.upload_image(0, g["img"])
    rois = [tuple(int(v) for v in r) for r in g["rois"]]
    res = ctx.detect_gftt(0, rois, klib.gftt_params(100, 0.01, 10.0, 3, 1, 0.0))
    for i, (xy, resp) in enumerate(res):
        _same_corner_set(xy, resp, g["harris_xy%d" % i], g["harris_resp%d" % i])
    res = ctx.detect_gftt(0, rois, klib.gftt_params(50, 0.01, 5.0, 3, 0, 0.0))
    for i, (xy, resp) in enumerate(res):
        _same_corner_set(xy, resp, g["mineig_xy%d" % i], g["mineig_resp%d" % i])
    assert ctx.detect_gftt(0, [], klib.gftt_params()) == []
    with pytest.raises(klib.KtlvoError):
        ctx.detect_gftt(0, [(250, 170, 59, 39)], klib.gftt_params())   # ROI leaves the frame


def test_gftt_all_negative_response(make_ctx, oracle, klib):
    """Harris with k > 0 on a pure ramp: every response is negative (det = 0, trace > 0), so the ROI maximum is
    negative and quality * max lies ABOVE it: goodFeaturesToTrack keeps nothing.  Small ROI (LDS selection) and
    whole frame (global selection), against the oracle; then a ramp with one blob, whose few positive responses
    must sort above the negative ones."""
    W, H = 320, 200
    ramp = np.tile((np.arange(W) * 0.7).astype(np.uint8), (H, 1))
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, ramp)
    gp = klib.gftt_params(50, 0.01, 5.0, 3, 1, 0.04)
    resp = ctx.corner_response(0, (40, 40, 59, 39), True, 0.04)
    assert resp.max() < 0
    assert np.array_equal(resp, oracle.corner_response(ramp, (40, 40, 59, 39), True, 0.04))
    for roi in ((40, 40, 59, 39), (0, 0, W, H)):
        (xy, rs), = ctx.detect_gftt(0, [roi], gp)
        oxy, ors = oracle.good_features(ramp, roi, 50, 0.01, 5.0, True, 0.04)
        assert len(xy) == 0 and len(oxy) == 0
    img = ramp.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    img[(yy - 100) ** 2 + (xx - 150) ** 2 < 36] = 255
    ctx.upload_image(0, img)
    for roi in ((120, 80, 60, 40), (0, 0, W, H)):
        (xy, rs), = ctx.detect_gftt(0, [roi], gp)
        oxy, ors = oracle.good_features(img, roi, 50, 0.01, 5.0, True, 0.04)
        assert len(xy) > 0 and np.array_equal(xy, oxy) and np.array_equal(rs, ors)


def test_gftt_large_rois_variants(make_ctx, oracle, klib):
    """Device-side selection for ROIs beyond the LDS path: no minimum distance (the first N of the sorted list),
    several large ROIs in one call, a fractional minimum distance, Harris and min-eigenvalue, a count no frame
    can reach (the greedy pass runs to the end of the list), a featureless frame."""
    W, H = 700, 420
    img = synth.texture_image(W, H, seed=99, periods=(37.0, 13.0, 5.0), noise=True)
    ctx = make_ctx(width=W, height=H, max_points=30000)
    ctx.upload_image(0, img)
    rois = [(0, 0, W, H), (10, 20, 300, 200), (350, 100, 333, 311)]
    for gp in (klib.gftt_params(3000, 0.01, 0.0, 3, 0, 0.0), klib.gftt_params(1500, 0.02, 3.7, 3, 1, 0.04),
               klib.gftt_params(20000, 0.001, 6.0, 3, 0, 0.0), klib.gftt_params(500, 0.3, 25.0, 3, 1, 0.0)):
        got = ctx.detect_gftt(0, rois, gp)
        for roi, (xy, rs) in zip(rois, got):
            oxy, ors = oracle.good_features(img, roi, gp.max_corners, gp.quality, gp.min_dist, bool(gp.use_harris),
                                            gp.harris_k)
            assert np.array_equal(xy, np.asarray(oxy, np.float32).reshape(-1, 2)), (roi, gp.min_dist)
            assert np.array_equal(rs, np.asarray(ors, np.float32))
        assert len(got[0][0]) > 100
    ctx.upload_image(0, np.full((H, W), 77, np.uint8))
    (xy, rs), = ctx.detect_gftt(0, [(0, 0, W, H)], klib.gftt_params(100, 0.01, 5.0, 3, 0, 0.0))
    assert len(xy) == 0


def test_gftt_reference_grid_and_full_frame(make_ctx, oracle, klib):
    W, H = 1242, 375
    scene = synth.CorridorScene(W, H, seed=3)
    img = scene.render(0)
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, img)
    subW, subH = (W - 60) / 20.0, (H - 60) / 8.0                      # featureTracking.cpp:76-77
    rois = [(30 + int(subW * j), 30 + int(subH * i), int(subW), int(subH)) for i in range(8) for j in range(20)]
    res = ctx.detect_gftt(0, rois, klib.gftt_params())
    total = 0
    for roi, (xy, resp) in zip(rois, res):
        oxy, oresp = oracle.good_features(img, roi, 100, 0.01, 10.0, True, 0.0)
        _same_corner_set(xy, resp, oxy, oresp)
        total += len(xy)
    assert total > 500
    # whole frame, Shi-Tomasi, 2000 corners (the benchmark's detector setting): host-side selection path
    (xy, resp), = ctx.detect_gftt(0, [(0, 0, W, H)], klib.gftt_params(2000, 0.01, 10.0, 3, 0, 0.0))
    oxy, oresp = oracle.good_features(img, None, 2000, 0.01, 10.0, False, 0.0)
    _same_corner_set(xy, resp, oxy, oresp)
    assert len(xy) > 500


def test_anms_matches_oracle(make_ctx, oracle):
    ctx = make_ctx(width=320, height=240)
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 500, (1500, 2)).astype(np.float32)
    for resp in [np.zeros(1500, np.float32), rng.uniform(0.1, 1, 1500).astype(np.float32)]:
        gxy, gresp = ctx.anms(xy, resp, 1000)
        oxy, oresp, _ = oracle.anms(xy, resp, 1000)
        assert np.array_equal(gxy, oxy) and np.array_equal(gresp, oresp)
    gxy, _ = ctx.anms(xy[:900], np.zeros(900, np.float32), 1000)
    assert np.array_equal(gxy, xy[:900])
    gxy, _ = ctx.anms(xy[:1000], np.zeros(1000, np.float32), 1000)     # n == numToKeep: keep all
    assert len(gxy) == 1000


# ---------------------------------------------------------------- gates -----------------------
def test_gates_match_oracle(make_ctx, oracle):
    ctx = make_ctx()
    rng = np.random.default_rng(5)
    n = 5000
    f1 = np.stack([rng.uniform(0, 1242, n), rng.uniform(0, 375, n)], 1).astype(np.float32)
    f2 = (f1 + rng.normal(size=(n, 2)) * 40).astype(np.float32)
    chk = (f1 + rng.normal(size=(n, 2)) * 4).astype(np.float32)
    for yt in (1, 5):
        keep = oracle.fb_gate(f1, f2, chk, 1242, 375, 30, 30, 100.0, yt)
        idx = ctx.fb_gate(f1, f2, chk, yt)
        assert np.array_equal(idx, np.nonzero(keep)[0])
    pcr = (f1 + rng.normal(size=(n, 2)) * 2).astype(np.float32)
    assert np.array_equal(ctx.circular_gate(pcr, f1), np.nonzero(oracle.circular_gate(pcr, f1))[0])
    assert len(ctx.fb_gate(f1[:0], f2[:0], chk[:0], 1)) == 0
    # none kept / all kept
    assert len(ctx.circular_gate(f1 + 10, f1)) == 0
    assert np.array_equal(ctx.circular_gate(f1, f1), np.arange(n))


def test_get_optical_flow_matches_composition(make_ctx, oracle):
    W, H = 1242, 375
    a, b = synth.texture_pair(W, H, 9, 1.4, 0.3, noise=True)
    ctx = make_ctx()
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.build_pyramids([0, 1])
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(0, W, 1500), rng.uniform(0, H, 1500)], 1).astype(np.float32)
    s1, s2, idx = ctx.get_optical_flow(0, 1, pts, 1)
    f2, _, _ = ctx.pyrlk(0, 1, pts)
    chk, _, _ = ctx.pyrlk(1, 0, f2)
    keep = np.nonzero(oracle.fb_gate(pts, f2, chk, W, H, 30, 30, 100.0, 1))[0]
    assert np.array_equal(idx, keep) and np.array_equal(s1, pts[keep]) and np.array_equal(s2, f2[keep])
    assert 100 < len(idx) < 1500


@pytest.mark.parametrize("size", [(1241, 376), (1242, 375), (640, 360), (353, 370), (368, 353), (1920, 1080)])
def test_fused_small_levels_equal_per_level_launches(make_ctx, oracle, size):
    """Levels 2-4 from ONE launch (pyr_down_tail3) against a launch per level (KTLVO_PYR_PER_LEVEL=1) and against
    the oracle: identical levels, and identical LK results for points hugging and leaving every edge, which read
    the levels' reflect-101 frames at all five levels."""
    W, H = size
    rng = np.random.default_rng(W * 7 + H)
    a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    b = np.roll(a, (1, -2), (0, 1))
    os.environ["KTLVO_PYR_PER_LEVEL"] = "1"
    try:
        ref = make_ctx(width=W, height=H, num_slots=2, max_points=4096)
    finally:
        del os.environ["KTLVO_PYR_PER_LEVEL"]
    new = make_ctx(width=W, height=H, num_slots=2, max_points=4096)
    po = oracle.Pyramid(a, 4, 21)
    for ctx in (ref, new):
        ctx.upload_image(0, a)
        ctx.upload_image(1, b)
        ctx.build_pyramids([0, 1])
        assert ctx.pyramid_levels(0) == po.max_level == 4
    for l in range(5):
        lv = po.level(l)
        assert np.array_equal(new.download_level(0, l), lv), l
        assert np.array_equal(ref.download_level(0, l), lv), l
        assert np.array_equal(new.download_scharr(0, l, lv.shape[1], lv.shape[0]), po.deriv(l)), l
    n = 3000
    edge = rng.integers(0, 4, n)
    t_ = rng.uniform(0, 1, n)
    off = rng.uniform(-24, 24, n)
    x = np.where(edge == 0, off, np.where(edge == 1, W - 1 + off, t_ * W))
    y = np.where(edge == 2, off, np.where(edge == 3, H - 1 + off, t_ * H))
    pts = np.stack([x, y], 1).astype(np.float32)
    ra, rb = ref.pyrlk(0, 1, pts), new.pyrlk(0, 1, pts)
    assert all(np.array_equal(p, q) for p, q in zip(ra, rb))
    assert 0.05 < ra[1].mean() < 1.0


@pytest.mark.parametrize("size", [(1241, 376), (705, 401), (640, 360)])
def test_large_batch_small_levels_equal_per_level_launches(make_ctx, oracle, size):
    """A batch of more than eight images forms levels 3 and 4 from one launch, a workgroup per image
    (pyr_down_tail2), the second level from what the workgroup has just stored.  Against a launch per level
    (KTLVO_PYR_PER_LEVEL=1) and the oracle: identical levels and derivative planes for every image, and identical
    LK results for points hugging and leaving every edge, which read the levels' reflect-101 frames."""
    W, H = size
    n = 12
    rng = np.random.default_rng(W * 11 + H)
    imgs = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    os.environ["KTLVO_PYR_PER_LEVEL"] = "1"
    try:
        ref = make_ctx(width=W, height=H, num_slots=n, max_points=4096)
    finally:
        del os.environ["KTLVO_PYR_PER_LEVEL"]
    new = make_ctx(width=W, height=H, num_slots=n, max_points=4096)
    for ctx in (ref, new):
        ctx.upload_images(list(range(n)), imgs)
        ctx.build_pyramids(list(range(n)))
    for s_ in (0, 5, 11):
        po = oracle.Pyramid(imgs[s_], 4, 21)
        assert po.max_level == 4 == new.pyramid_levels(s_)
        for l in range(5):
            lv = po.level(l)
            assert np.array_equal(new.download_level(s_, l), lv), (s_, l)
            assert np.array_equal(new.download_scharr(s_, l, lv.shape[1], lv.shape[0]), po.deriv(l)), (s_, l)
    m = 2000
    edge = rng.integers(0, 4, m)
    t_ = rng.uniform(0, 1, m)
    off = rng.uniform(-24, 24, m)
    x = np.where(edge == 0, off, np.where(edge == 1, W - 1 + off, t_ * W))
    y = np.where(edge == 2, off, np.where(edge == 3, H - 1 + off, t_ * H))
    pts = np.stack([x, y], 1).astype(np.float32)
    for a_, b_ in ((0, 1), (10, 11), (7, 3)):
        ra, rb = ref.pyrlk(a_, b_, pts), new.pyrlk(a_, b_, pts)
        assert all(np.array_equal(p, q) for p, q in zip(ra, rb)), (a_, b_)


def test_detect_and_build_side_by_side_equals_calls_in_turn(make_ctx, klib):
    """ktlvo_detect_build_frames_dev (pyramid build on a side stream beside the detection) against
    ktlvo_detect_gftt_frames_dev followed by ktlvo_build_pyramids: same corners, same planes at every level, and the
    tracks a following ktlvo_pyrlk_batch_dev makes of them are the same bits; twice in a row with new frames in the
    same slots (the second round's ingest, build and detection must wait for the first round's tracks), with pinned
    frames uploaded in bulk; plus the empty halves and the refusals, which queue nothing."""
    import torch
    W, H, n = 640, 360, 6
    a, b = make_ctx(width=W, height=H, num_slots=n), make_ctx(width=W, height=H, num_slots=n)
    params = klib.gftt_params(600, 0.01, 7.0, 3, 0, 0.0)
    lk = klib.lk_params()
    prev, nxt = [0, 1, 2, 3, 4], [1, 2, 3, 4, 5]
    out = {}
    for tag, ctx in (("turn", a), ("side", b)):
        d_xy = torch.zeros((5, 600, 2), device="cuda")
        d_r = torch.zeros((5, 600), device="cuda")
        d_c = torch.zeros(5, dtype=torch.int32, device="cuda")
        d_n = torch.zeros((5, 600, 2), device="cuda")
        d_s = torch.zeros((5, 600), dtype=torch.uint8, device="cuda")
        d_e = torch.zeros((5, 600), device="cuda")
        torch.cuda.synchronize()
        res = []
        for rnd in range(2):
            imgs = np.stack([synth.texture_image(W, H, 50 + 10 * rnd, dx=0.8 * s, dy=-0.3 * s, noise=True) for s in range(n)])
            pinned = torch.from_numpy(imgs).pin_memory()
            ctx.upload_images(list(range(n)), pinned.numpy())
            if tag == "turn":
                ctx.detect_gftt_frames_dev(prev, params, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), 600)
                ctx.build_pyramids(list(range(n)))
            else:
                ctx.detect_build_frames_dev(prev, params, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), 600, list(range(n)))
            ctx.pyrlk_batch_dev(prev, nxt, d_xy.data_ptr(), 600, d_n.data_ptr(), d_s.data_ptr(), d_e.data_ptr(), lk)
            ctx.sync()
            planes = [(ctx.download_level(s_, l), ctx.download_scharr(s_, l, *ctx.download_level(s_, l).shape[::-1]))
                      for s_ in (0, n - 1) for l in range(ctx.pyramid_levels(s_) + 1)]
            res.append((d_xy.cpu().numpy().copy(), d_r.cpu().numpy().copy(), d_c.cpu().numpy().copy(), d_n.cpu().numpy().copy(),
                        d_s.cpu().numpy().copy(), d_e.cpu().numpy().copy(), planes))
        out[tag] = res
    for rnd in range(2):
        t, s_ = out["turn"][rnd], out["side"][rnd]
        cnt = t[2]
        assert np.array_equal(cnt, s_[2]) and cnt.min() > 100
        for f in range(5):
            k = int(cnt[f])
            assert np.array_equal(t[0][f, :k], s_[0][f, :k]) and np.array_equal(t[1][f, :k], s_[1][f, :k])
            assert np.array_equal(t[3][f, :k], s_[3][f, :k]) and np.array_equal(t[4][f, :k], s_[4][f, :k])
            assert np.array_equal(t[5][f, :k], s_[5][f, :k])
        for (la, da), (lb, db) in zip(t[6], s_[6]):
            assert np.array_equal(la, lb) and np.array_equal(da, db)
    assert not np.array_equal(out["turn"][0][3], out["turn"][1][3])          # the rounds really differ
    # empty halves
    d_xy = torch.zeros((2, 600, 2), device="cuda"); d_r = torch.zeros((2, 600), device="cuda")
    d_c = torch.zeros(2, dtype=torch.int32, device="cuda")
    b.detect_build_frames_dev([], params, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), 600, [0, 1])   # build only
    b.detect_build_frames_dev([0, 1], params, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), 600, [])   # detect only
    b.sync()
    assert np.array_equal(d_c.cpu().numpy(), out["side"][1][2][:2])
    # refusals: a slot without an image in either list, a window the context was not sized for
    c = make_ctx(width=W, height=H, num_slots=3)
    c.upload_image(0, synth.texture_image(W, H, 3))
    for det, bld in (([0], [0, 1]), ([1], [0])):
        with pytest.raises(klib.KtlvoError) as ei:
            c.detect_build_frames_dev(det, params, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), 600, bld)
        assert ei.value.status == klib.ERR_STATE
    with pytest.raises(klib.KtlvoError) as ei:
        c.detect_build_frames_dev([0], params, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), 600, [0], win=15)
    assert ei.value.status == klib.ERR_UNSUPPORTED
    assert c.pyramid_levels(0) == 0                                           # nothing was built by the refused calls


def test_stream_step_in_one_call_equals_the_two_calls(make_ctx, klib):
    """ktlvo_detect_build_track_dev (detection, pyramid build and the LK launch queued in one go, the host waiting for the
    detection's flags only) against ktlvo_detect_build_frames_dev followed by ktlvo_pyrlk_batch_dev: same corners, counts,
    tracks, status and error bits.  Two rounds on textured frames (the route queued blind stands), then a round with a
    noise frame whose corners exceed what the sort in LDS holds (the detection is done again by the other route, and so
    is the LK launch), then the empty forms and a refusal."""
    import torch
    W, H, n, cap = 640, 360, 11, 600   # more than 8 frames: a workgroup per frame sorts in LDS (16384 keys at most)
    a, b = make_ctx(width=W, height=H, num_slots=n), make_ctx(width=W, height=H, num_slots=n)
    params = klib.gftt_params(cap, 0.01, 7.0, 3, 0, 0.0)
    lk = klib.lk_params()
    prev, nxt, slots = list(range(n - 1)), list(range(1, n)), list(range(n))
    P = n - 1
    rng = np.random.default_rng(5)
    rounds = []
    for rnd in range(3):
        imgs = [synth.texture_image(W, H, 70 + 10 * rnd, dx=0.8 * s, dy=-0.3 * s, noise=True) for s in range(n)]
        if rnd == 2:
            imgs[2] = rng.integers(0, 256, (H, W), dtype=np.uint8)
        rounds.append(np.stack(imgs))
    out = {}
    for tag, ctx in (("two", a), ("one", b)):
        bufs = [torch.zeros((P, cap, 2), device="cuda"), torch.zeros((P, cap), device="cuda"),
                torch.zeros(P, dtype=torch.int32, device="cuda"), torch.zeros((P, cap, 2), device="cuda"),
                torch.zeros((P, cap), dtype=torch.uint8, device="cuda"), torch.zeros((P, cap), device="cuda")]
        d_xy, d_r, d_c, d_n, d_s, d_e = [t.data_ptr() for t in bufs]
        torch.cuda.synchronize()
        res = []
        for imgs in rounds:
            ctx.upload_images(slots, imgs)
            if tag == "two":
                ctx.detect_build_frames_dev(prev, params, d_xy, d_r, d_c, cap, slots)
                ctx.pyrlk_batch_dev(prev, nxt, d_xy, cap, d_n, d_s, d_e, lk)
            else:
                ctx.detect_build_track_dev(prev, nxt, params, d_xy, d_r, d_c, cap, slots, d_n, d_s, d_e, lk)
            ctx.sync()
            res.append([t.cpu().numpy().copy() for t in bufs])
        out[tag] = res
    for rnd in range(3):
        t, o = out["two"][rnd], out["one"][rnd]
        cnt = t[2]
        assert np.array_equal(cnt, o[2]) and cnt.min() > 100
        for f in range(P):
            k = int(cnt[f])
            for q in (0, 1, 3, 4, 5):
                assert np.array_equal(t[q][f, :k], o[q][f, :k]), (rnd, f, q)
    assert out["two"][2][2][2] == cap                                        # the noise frame fills its list
    assert not np.array_equal(out["two"][0][3], out["two"][1][3])
    # no pairs: the build alone; no pairs and nothing to build: nothing
    bufs = [torch.zeros((2, cap, 2), device="cuda"), torch.zeros((2, cap), device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda"),
            torch.zeros((2, cap, 2), device="cuda"), torch.zeros((2, cap), dtype=torch.uint8, device="cuda"), torch.zeros((2, cap), device="cuda")]
    d_xy, d_r, d_c, d_n, d_s, d_e = [t.data_ptr() for t in bufs]
    b.detect_build_track_dev([], [], params, d_xy, d_r, d_c, cap, [0, 1], d_n, d_s, d_e, lk)
    b.detect_build_track_dev([], [], params, d_xy, d_r, d_c, cap, [], d_n, d_s, d_e, lk)
    # pairs without a build list: the pyramids of the last round are tracked on
    b.detect_build_track_dev(prev[:2], nxt[:2], params, d_xy, d_r, d_c, cap, [], d_n, d_s, d_e, lk)
    b.sync()
    o = out["one"][2]
    assert np.array_equal(bufs[2].cpu().numpy(), o[2][:2])
    k = int(o[2][0])
    assert np.array_equal(bufs[3].cpu().numpy()[0, :k], o[3][0, :k]) and np.array_equal(bufs[4].cpu().numpy()[0, :k], o[4][0, :k])
    # a second frame without an image: refused before anything is queued
    c = make_ctx(width=W, height=H, num_slots=3)
    c.upload_image(0, synth.texture_image(W, H, 3))
    with pytest.raises(klib.KtlvoError) as ei:
        c.detect_build_track_dev([0], [1], params, d_xy, d_r, d_c, cap, [0], d_n, d_s, d_e, lk)
    assert ei.value.status == klib.ERR_STATE
    assert c.pyramid_levels(0) == 0


def test_batched_detection_equals_per_frame_detection(make_ctx, klib):
    """ktlvo_detect_gftt_frames: the whole-frame detector over a batch of frames with the launches of one
    detection.  Every frame's corners, order and responses equal ktlvo_detect_gftt's for that frame: frames of
    very different candidate counts in one batch (texture, a nearly flat frame, a flat one), with and without a
    minimum distance, a region instead of the whole frame, a count no frame can reach, and the device-output form."""
    import torch
    W, H, n = 640, 360, 7
    rng = np.random.default_rng(33)
    from slam_amd.synth import texture_image
    imgs = [texture_image(W, H, seed=40 + i) for i in range(4)]
    imgs.append(np.full((H, W), 90, np.uint8))                                  # flat: no corners
    low = np.full((H, W), 120, np.uint8)
    low[100:140, 200:260] = texture_image(60, 40, seed=5)                        # nearly flat: a handful
    imgs.append(low)
    imgs.append(rng.integers(0, 256, (H, W), dtype=np.uint8))                    # noise: candidates everywhere
    ctx = make_ctx(width=W, height=H, num_slots=n)
    ctx.upload_images(list(range(n)), np.stack(imgs))
    order = [6, 0, 4, 2, 5, 1, 3, 0]                                             # any order, a slot twice
    for params, roi in ((klib.gftt_params(800, 0.01, 7.0, 3, 0, 0.0), None),
                        (klib.gftt_params(3000, 0.02, 0.0, 3, 1, 0.04), None),
                        (klib.gftt_params(20000, 0.001, 3.0, 3, 0, 0.0), None),
                        (klib.gftt_params(500, 0.01, 10.0, 3, 0, 0.0), (37, 21, 411, 300))):
        got = ctx.detect_gftt_frames(order, params, roi)
        r = roi if roi is not None else (0, 0, W, H)
        for s, (xy, resp) in zip(order, got):
            (rxy, rresp), = ctx.detect_gftt(s, [r], params)
            assert np.array_equal(xy, rxy) and np.array_equal(resp, rresp), (s, len(xy), len(rxy))
        assert len(got[2][0]) == 0 and 0 < len(got[4][0]) < 200 and len(got[0][0]) > 300
    params = klib.gftt_params(800, 0.01, 7.0, 3, 0, 0.0)
    d_xy = torch.zeros((n, 800, 2), device="cuda")
    d_resp = torch.zeros((n, 800), device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.detect_gftt_frames_dev(list(range(n)), params, d_xy.data_ptr(), d_resp.data_ptr(), d_cnt.data_ptr(), 800)
    ctx.sync()
    for s in range(n):
        (rxy, rresp), = ctx.detect_gftt(s, [(0, 0, W, H)], params)
        k = int(d_cnt[s])
        assert k == len(rxy) and np.array_equal(d_xy[s, :k].cpu().numpy(), rxy) and np.array_equal(d_resp[s, :k].cpu().numpy(), rresp)
    with pytest.raises(klib.KtlvoError) as ei:
        ctx.detect_gftt_frames([0, n], params)
    assert ei.value.status == klib.ERR_INVALID
    assert ctx.detect_gftt_frames([], params) == []                      # nothing to do is not an error
    with pytest.raises(klib.KtlvoError) as ei:                            # a region that leaves the frame
        ctx.detect_gftt_frames([0], params, (600, 0, 100, 100))
    assert ei.value.status == klib.ERR_INVALID
    with pytest.raises(klib.KtlvoError) as ei:                            # block sizes other than the reference's
        ctx.detect_gftt_frames([0], klib.gftt_params(100, 0.01, 5.0, 5, 0, 0.0))
    assert ei.value.status == klib.ERR_UNSUPPORTED
    with pytest.raises(klib.KtlvoError) as ei:                            # more frames than one launch takes
        ctx.detect_gftt_frames([0] * 257, params)
    assert ei.value.status == klib.ERR_CAPACITY
    empty = make_ctx(width=W, height=H, num_slots=2)
    with pytest.raises(klib.KtlvoError) as ei:                            # a slot that holds no image yet
        empty.detect_gftt_frames([0, 1], params)
    assert ei.value.status == klib.ERR_STATE


def test_batched_detection_lds_sort_equals_global_sort(make_ctx, klib):
    """The batched detector filters and sorts each frame in LDS by one workgroup and keeps the greedy pass's cell grid
    in LDS when it fits; a frame with more keys above its threshold than the LDS sort holds (16384) sends the whole
    batch through the global sort instead.  Both routes must give what a context created with
    KTLVO_FRAMES_GLOBAL_SORT=1 gives, and what per-frame detection gives: KITTI-sized frames (12,000 keys each; the
    LDS route), the same batch with a white-noise frame in it (50,000 keys; the fallback), a minimum distance whose
    grid does not fit LDS, and no minimum distance at all."""
    from slam_amd.synth import texture_image
    W, H = 1241, 376
    rng = np.random.default_rng(91)
    imgs = [texture_image(W, H, seed=70 + i, noise=True) for i in range(3)]
    imgs.append(np.full((H, W), 77, np.uint8))
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    fast = make_ctx(width=W, height=H, num_slots=5)
    os.environ["KTLVO_FRAMES_GLOBAL_SORT"] = "1"
    try:
        slow = make_ctx(width=W, height=H, num_slots=5)
    finally:
        del os.environ["KTLVO_FRAMES_GLOBAL_SORT"]
    for ctx in (fast, slow):
        ctx.upload_images([0, 1, 2, 3, 4], np.stack(imgs + [noise]))
    for params in (klib.gftt_params(2000, 0.01, 10.0, 3, 0, 0.0), klib.gftt_params(2000, 0.01, 9.0, 3, 0, 0.0),
                   klib.gftt_params(4000, 0.003, 3.0, 3, 0, 0.0), klib.gftt_params(5000, 0.01, 0.0, 3, 0, 0.0),
                   klib.gftt_params(300, 0.05, 40.0, 3, 1, 0.04)):
        for order in ([0, 1, 2, 3, 1], [2, 4, 0, 3]):                           # without / with the noise frame
            a = fast.detect_gftt_frames(order, params)
            b = slow.detect_gftt_frames(order, params)
            for s_, (xa, ra), (xb, rb) in zip(order, a, b):
                assert np.array_equal(xa, xb) and np.array_equal(ra, rb), (s_, len(xa), len(xb))
        for s_, (xa, ra) in zip([0, 1, 2, 3, 1], fast.detect_gftt_frames([0, 1, 2, 3, 1], params)):
            (rxy, rresp), = fast.detect_gftt(s_, [(0, 0, W, H)], params)
            assert np.array_equal(xa, rxy) and np.array_equal(ra, rresp), (s_, len(xa), len(rxy))
        # per-call detection takes the same two routes: several large ROIs in one call, the noise frame (fallback) too
        rois = [(0, 0, W, H), (100, 20, 700, 300), (0, 0, 333, H)]
        for s_ in (0, 4):
            for (xa, ra), (xb, rb) in zip(fast.detect_gftt(s_, rois, params), slow.detect_gftt(s_, rois, params)):
                assert np.array_equal(xa, xb) and np.array_equal(ra, rb), (s_, len(xa), len(xb))
    got = fast.detect_gftt_frames([0, 3], klib.gftt_params(2000, 0.01, 10.0, 3, 0, 0.0))
    assert len(got[0][0]) == 2000 and len(got[1][0]) == 0


def test_batched_upload_equals_per_frame_upload(make_ctx):
    """ktlvo_upload_images (one DMA, one ingest launch) fills the slots exactly as n ktlvo_upload_image calls do,
    from pageable and from pinned memory, into slots given in any order."""
    import torch
    W, H, n = 322, 203, 5
    rng = np.random.default_rng(8)
    imgs = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    a, b, c = (make_ctx(width=W, height=H, num_slots=n) for _ in range(3))
    order = [3, 0, 4, 1, 2]
    for i, s in enumerate(order):
        a.upload_image(s, imgs[i])
    b.upload_images(order, imgs)
    pinned = torch.from_numpy(imgs).pin_memory()
    c.upload_images(order, pinned.numpy())
    c.sync()
    pinned.fill_(7)                      # after the sync the pinned source is free again
    for ctx in (a, b, c):
        ctx.build_pyramids(list(range(n)))
    for s in range(n):
        for l in range(a.pyramid_levels(s) + 1):
            ref = a.download_level(s, l)
            assert np.array_equal(b.download_level(s, l), ref) and np.array_equal(c.download_level(s, l), ref)
    pts = np.stack([rng.uniform(0, W, 200), rng.uniform(0, H, 200)], 1).astype(np.float32)
    ra, rb = a.pyrlk(0, 1, pts), b.pyrlk(0, 1, pts)        # the frames (reflect-101 borders) are the same too
    assert all(np.array_equal(x, y) for x, y in zip(ra, rb))


def test_batch_upload_runs_ahead_of_queued_work(make_ctx, klib):
    """ktlvo_upload_images puts its DMA on the context's copy stream and unpacks on the context's stream.  A caller
    that never waits (upload, pyramids, track, again with the next batch into the SAME slots, ...) gets, for every
    batch, the result of a context that is waited for after every call; pageable and pinned sources, batches of
    changing size (the staging buffers grow), and a per-frame upload in between."""
    import torch
    W, H, n = 640, 360, 6
    rng = np.random.default_rng(21)
    from slam_amd.synth import texture_image
    base = texture_image(W + 64, H + 64, seed=21)
    a, b = (make_ctx(width=W, height=H, num_slots=n, max_points=1500) for _ in range(2))
    NF = 1500
    pts = np.stack([rng.uniform(0, W, NF), rng.uniform(0, H, NF)], 1).astype(np.float32)
    d_prev = torch.from_numpy(np.tile(pts, (n - 1, 1))).cuda()
    lk = klib.lk_params()
    batches, keep = [], []
    for rep in range(7):
        m = (3, 6, 4, 6, 2, 6, 5)[rep]
        imgs = np.stack([np.ascontiguousarray(base[4 * rep + 3 * i:4 * rep + 3 * i + H, rep + 5 * i:rep + 5 * i + W])
                         for i in range(m)])
        batches.append(imgs)
    outs = [tuple((torch.zeros((n - 1) * NF, 2, device="cuda"), torch.zeros((n - 1) * NF, dtype=torch.uint8, device="cuda"),
                   torch.zeros((n - 1) * NF, device="cuda")) for _ in batches) for _ in range(2)]
    for which, ctx in enumerate((a, b)):
        for rep, imgs in enumerate(batches):
            m = len(imgs)
            src = imgs
            if rep % 2:
                keep.append(torch.from_numpy(imgs).pin_memory())   # must outlive the DMA
                src = keep[-1].numpy()
            ctx.upload_images(list(range(m)), src)
            if rep == 3:
                ctx.upload_image(1, imgs[0])
            ctx.build_pyramids(list(range(m)))
            o = outs[which][rep]
            ctx.pyrlk_batch_dev(list(range(m - 1)), list(range(1, m)), d_prev.data_ptr(), NF, o[0].data_ptr(),
                                o[1].data_ptr(), o[2].data_ptr(), lk)
            if which == 0:
                ctx.sync()
        ctx.sync()
    for rep, imgs in enumerate(batches):
        k = (len(imgs) - 1) * NF
        for x, y in zip(outs[0][rep], outs[1][rep]):
            assert torch.equal(x[:k], y[:k]), rep
        assert int(outs[0][rep][1][:k].sum()) > k // 2


# ---------------------------------------------------------------- RANSAC ----------------------
def test_ransac_score_counts_exact(make_ctx, oracle):
    g = np.load(os.path.join(GOLD, "ransac_400.npz"))
    ctx = make_ctx()
    rng = np.random.default_rng(9)
    H = 64
    Rt = np.zeros((H, 12))
    for h in range(H):
        Rt[h, :9] = oracle.rodrigues_v2m(g["rvec"] + rng.normal(size=3) * 1e-3).reshape(9)
        Rt[h, 9:] = g["tvec"] + rng.normal(size=3) * 5e-3
    for thr_sq in (np.float32(float(np.float32(0.1)) ** 2), np.float32(4.0)):
        counts, mask = ctx.ransac_score(g["X3"], g["x2"], Rt, g["K"], thr_sq, mask_hyp=3)
        for h in range(H):
            c, m, _ = oracle.project_score(g["X3"], g["x2"], Rt[h], g["K"], thr_sq)
            assert counts[h] == c, h
            if h == 3:
                assert np.array_equal(mask, m)
    assert counts.max() > 50
    # the device-pointer form of the same call: arrays resident in HBM, nothing synchronised inside
    import torch
    dX, dx = torch.from_numpy(g["X3"].astype(np.float32)).cuda(), torch.from_numpy(g["x2"].astype(np.float32)).cuda()
    dRt = torch.from_numpy(Rt).cuda()
    dcnt = torch.full((H,), -1, dtype=torch.int32, device="cuda")
    dmask = torch.zeros(len(g["X3"]), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.ransac_score_dev(dX.data_ptr(), dx.data_ptr(), len(g["X3"]), dRt.data_ptr(), H, g["K"], thr_sq, dcnt.data_ptr(),
                         3, dmask.data_ptr())
    ctx.sync()
    assert np.array_equal(dcnt.cpu().numpy(), counts) and np.array_equal(dmask.cpu().numpy(), mask)
    ctx.ransac_score_dev(dX.data_ptr(), dx.data_ptr(), 0, dRt.data_ptr(), H, g["K"], thr_sq, dcnt.data_ptr())
    ctx.sync()
    assert not dcnt.cpu().numpy().any()      # no points: every count is zero


def test_ransac_score_large_batch(make_ctx, oracle):
    """BASELINE config 3 shape: 1024 hypotheses x 8000 points."""
    rng = np.random.default_rng(11)
    K = np.array([1093.5, 1093.5, 960.0, 540.0])
    n, H = 8000, 1024
    R = oracle.rodrigues_v2m(np.array([0.003, -0.01, 0.001]))
    t = np.array([0.02, -0.01, -0.8])
    X = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(4, 60, n)], 1).astype(np.float32)
    Xc = X.astype(np.float64) @ R.T + t
    uv = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
    uv = (uv + rng.normal(size=uv.shape) * 0.5).astype(np.float32)
    Rt = np.zeros((H, 12))
    for h in range(H):
        Rt[h, :9] = oracle.rodrigues_v2m(np.array([0.003, -0.01, 0.001]) + rng.normal(size=3) * 2e-4).reshape(9)
        Rt[h, 9:] = t + rng.normal(size=3) * 2e-3
    ctx = make_ctx(width=1920, height=1080, fx=K[0], fy=K[1], cx=K[2], cy=K[3])
    counts = ctx.ransac_score(X, uv, Rt, K, np.float32(4.0))
    for h in range(0, H, 37):
        c, _, _ = oracle.project_score(X, uv, Rt[h], K, np.float32(4.0))
        assert counts[h] == c
    assert counts.max() > 4000


def test_solve_pnp_ransac_matches_oracle(make_ctx, oracle):
    """Both hypothesis paths of the library (P3P on the host, P3P in a kernel) against the golden result of the
    oracle's RANSAC.  Library and oracle run the reference's minimal solver (Gao et al., closed-form roots,
    no polishing): same selected hypothesis, same inlier mask, pose to 1e-12 on the host path (same operations,
    same libm) and 1e-9 on the device path (device libm)."""
    g = np.load(os.path.join(GOLD, "ransac_400.npz"))
    live = oracle.solve_pnp_ransac(g["X3"], g["x2"], g["K"])
    assert live["n_inliers"] == int(g["n_inliers"]) and np.array_equal(live["mask"], g["mask"])
    for dev in (0, 1):
        ctx = make_ctx(device_p3p=dev)
        r = ctx.solve_pnp_ransac(g["X3"], g["x2"])
        assert r["ok"] and r["n_inliers"] == int(g["n_inliers"]), dev
        assert np.array_equal(r["mask"], g["mask"]), dev
        tol = 1e-9 if dev else 1e-12
        assert np.abs(r["rvec"] - g["rvec"]).max() < tol and np.abs(r["tvec"] - g["tvec"]).max() < tol, dev
        r = ctx.solve_pnp_ransac(g["X3"][:3], g["x2"][:3])
        assert not r["ok"]
        r = ctx.solve_pnp_ransac(g["X3"][:0], g["x2"][:0])
        assert not r["ok"]


# ---------------------------------------------------------------- full benchmark sizes -----------
@pytest.mark.parametrize("size,ncorners,min_dist", [((1920, 1080), 8000, 5.0), ((3840, 2160), 16384, 8.0)])
def test_full_size_frames(make_ctx, oracle, klib, size, ncorners, min_dist):
    """BASELINE configs 3 and 5 shapes: whole-frame Shi-Tomasi detection and LK at 1080p / 4K.
    Direct parity with the oracle plus size-independent properties (zero motion is the identity,
    a known global shift is recovered, forward-backward returns to the start)."""
    W, H = size
    shift = (2.5, -1.25)
    a, b = synth.texture_pair(W, H, 5, shift[0], shift[1], periods=(32.0, 8.0))
    ctx = make_ctx(width=W, height=H, max_points=ncorners, num_slots=3)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    ctx.upload_image(2, a)
    ctx.build_pyramids([0, 1, 2])
    pa, pb = oracle.Pyramid(a), oracle.Pyramid(b, with_deriv=False)
    assert ctx.pyramid_levels(0) == pa.max_level == 4
    for l in (1, 4):
        assert np.array_equal(ctx.download_level(0, l), pa.level(l))
    (xy, resp), = ctx.detect_gftt(0, [(0, 0, W, H)], klib.gftt_params(ncorners, 0.01, min_dist, 3, 0, 0.0))
    oxy, oresp = oracle.good_features(a, None, ncorners, 0.01, min_dist, False, 0.0)
    assert len(xy) == ncorners and np.array_equal(xy, oxy) and np.array_equal(resp, oresp)
    go, gs, ge, both = _lk_both(ctx, oracle, klib, pa, pb, xy)
    assert both.mean() > 0.95
    e = np.abs(go - xy - np.array(shift, np.float32))[both]
    assert np.median(e) < 0.02
    z, zs, _ = ctx.pyrlk(0, 2, xy)                                  # zero motion
    assert (np.abs(z - xy)[zs == 1] < 1e-3).all()
    r, rs, _ = ctx.pyrlk(1, 0, go)                                   # forward-backward
    ok = (gs == 1) & (rs == 1)
    assert np.median(np.abs(r - xy)[ok]) < 0.02


def test_pnp_ransac_1024_hypotheses_against_oracle(make_ctx, oracle):
    """Maximum batch (1024 hypotheses, 3000 points): host-P3P and device-P3P paths against the oracle's
    sequential RANSAC, closed-form roots on both sides and, as a second case, two Newton steps on both sides
    (ktlvo_config.p3p_polish): identical inlier counts and masks, poses to 1e-12 (host) / 1e-9 (device), and
    within 2e-3 of the planted rotation."""
    rng = np.random.default_rng(5)
    X = np.stack([rng.uniform(-8, 8, 3000), rng.uniform(-2, 2, 3000), rng.uniform(4, 38, 3000)], 1).astype(np.float32)
    R = oracle.rodrigues_v2m(np.array([0.01, -0.02, 0.005]))
    Xc = X.astype(np.float64) @ R.T + np.array([0.05, 0.0, -0.7])
    K = np.array([707.0912, 707.0912, 601.8873, 183.1104])
    uv = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
    uv = (uv + rng.normal(size=uv.shape) * 0.03).astype(np.float32)
    for polish in (0, 2):
        oracle.p3p_set_polish(polish)
        try:
            o = oracle.solve_pnp_ransac(X, uv, K, max_iters=1024)
        finally:
            oracle.p3p_set_polish(0)
        assert o["ok"]
        for dev in (0, 1):
            r = make_ctx(device_p3p=dev, p3p_polish=polish).solve_pnp_ransac(X, uv, max_iters=1024)
            assert r["ok"] and r["n_inliers"] == o["n_inliers"], (polish, dev, r["n_inliers"], o["n_inliers"])
            assert np.array_equal(r["mask"], o["mask"]), (polish, dev)
            tol = 1e-9 if dev else 1e-12
            assert np.abs(r["rvec"] - o["rvec"]).max() < tol and np.abs(r["tvec"] - o["tvec"]).max() < tol, (polish, dev)
            assert np.abs(r["rvec"] - np.array([0.01, -0.02, 0.005])).max() < 2e-3


def test_profiling_mask_selects_kernels(make_ctx, klib):
    W, H = 320, 240
    a, b = synth.texture_pair(W, H, 3, 1.0, 0.5)
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, a)
    ctx.upload_image(1, b)
    pts = np.array([[100.0, 100.0], [200.0, 120.0]], np.float32)
    for mode, expect_pyr in ((True, True), (2 << klib.K_PYRLK, False)):
        ctx.profile_enable(mode)
        ctx.profile_reset()
        ctx.build_pyramids([0, 1])
        ctx.pyrlk(0, 1, pts)
        ms_lk, n_lk = ctx.profile_get(klib.K_PYRLK)
        ms_pd, n_pd = ctx.profile_get(klib.K_PYRDOWN)
        assert n_lk == 1 and ms_lk > 0
        assert (n_pd == 1) == expect_pyr, (mode, n_pd)
    ctx.profile_enable(False)
    ctx.profile_reset()
    ctx.pyrlk(0, 1, pts)
    assert ctx.profile_get(klib.K_PYRLK)[1] == 0
