"""ktlvo_detect_gftt_noalloc (the entry point that allocates nothing) against ktlvo_detect_gftt, through the
C ABI: identical corners, responses and counts on two calls in a row, the second with a smaller input, so
that whatever the first call left in the library's fixed scratch array must not leak into the second."""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def _run_noalloc(ctx, klib, torch, rois, params, cap):
    n = len(rois)
    d_xy = torch.full((n, cap, 2), -7.0, dtype=torch.float32, device="cuda")
    d_resp = torch.full((n, cap), -7.0, dtype=torch.float32, device="cuda")
    d_cnt = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = ctx.detect_gftt_noalloc(0, rois, params, d_xy.data_ptr(), d_resp.data_ptr(), d_cnt.data_ptr(), cap)
    assert rc == 0, klib.strerror(rc)
    ctx.sync()
    cnt = d_cnt.cpu().numpy()
    xy, resp = d_xy.cpu().numpy(), d_resp.cpu().numpy()
    return [(xy[i, :cnt[i]], resp[i, :cnt[i]]) for i in range(n)]


def _same(a, b):
    assert len(a) == len(b)
    for (xa, ra), (xb, rb) in zip(a, b):
        assert np.array_equal(xa, xb) and np.array_equal(ra, rb)


def test_noalloc_detect_matches_detect(make_ctx, klib):
    import torch
    W, H = 1242, 375
    img = synth.CorridorScene(W, H, seed=3).render(0)
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, img)
    ctx.build_pyramids([0])
    subW, subH = (W - 60) / 20.0, (H - 60) / 8.0                      # featureTracking.cpp:76-77
    grid = [(30 + int(subW * j), 30 + int(subH * i), int(subW), int(subH)) for i in range(8) for j in range(20)]
    # call 1: the reference's key-frame detection, 160 sub-regions, Harris k = 0, 100 corners each
    p1 = klib.gftt_params()
    want1 = ctx.detect_gftt(0, grid, p1)
    got1 = _run_noalloc(ctx, klib, torch, grid, p1, 100)
    _same(got1, want1)
    assert sum(len(x) for x, _ in got1) > 500
    # call 2, right after: a smaller input (7 smaller regions, other detector settings)
    small = [(100 + 70 * i, 60 + 20 * (i % 3), 48 - 2 * i, 33 + i) for i in range(7)]
    p2 = klib.gftt_params(40, 0.02, 6.0, 3, 0, 0.0)
    want2 = ctx.detect_gftt(0, small, p2)
    got2 = _run_noalloc(ctx, klib, torch, small, p2, 40)
    _same(got2, want2)
    assert sum(len(x) for x, _ in got2) > 20
    # and the first input once more, now after the smaller one
    _same(_run_noalloc(ctx, klib, torch, grid, p1, 100), want1)
    # limits are refused up front, with a clear code, and nothing is enqueued
    dummy = torch.zeros(8, dtype=torch.float32, device="cuda")
    cap_err = klib.lib.ktlvo_strerror(ctx.detect_gftt_noalloc(0, [(30, 30, 10, 10)] * 193, p1, dummy.data_ptr(),
                                                            dummy.data_ptr(), dummy.data_ptr(), 100))
    assert b"capacity" in cap_err
    rc = ctx.detect_gftt_noalloc(0, [(30, 30, 100, 100)], p1, dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), 100)
    assert b"capacity" in klib.lib.ktlvo_strerror(rc)
    assert ctx.detect_gftt_noalloc(0, [], p1, None, None, None, 100) == 0


def test_noalloc_on_a_fresh_context_allocates_nothing_and_captures_into_a_graph(make_ctx, klib):
    """The entry point's contract, on a context that has made no other detection call (so nothing it needs was
    allocated or warmed by one): free device memory is the same before and after the call, and the call can be
    recorded into a HIP graph -- stream capture refuses allocation, synchronisation and copies from pageable memory --
    whose replays give ktlvo_detect_gftt's corners."""
    import torch
    W, H = 640, 360
    img = synth.texture_image(W, H, seed=21, noise=True)
    img2 = synth.texture_image(W, H, seed=22, noise=True)
    ctx = make_ctx(width=W, height=H)
    ctx.upload_image(0, img)
    ctx.sync()
    rois = [(30 + 58 * j, 30 + 40 * i, 58, 40) for i in range(7) for j in range(10)]
    p = klib.gftt_params(60, 0.01, 8.0, 3, 1, 0.0)
    n, cap = len(rois), 60
    d_xy = torch.zeros((n, cap, 2), dtype=torch.float32, device="cuda")
    d_resp = torch.zeros((n, cap), dtype=torch.float32, device="cuda")
    d_cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    rc = ctx.detect_gftt_noalloc(0, rois, p, d_xy.data_ptr(), d_resp.data_ptr(), d_cnt.data_ptr(), cap)   # the context's first detection
    assert rc == 0, klib.strerror(rc)
    ctx.sync()
    assert torch.cuda.mem_get_info()[0] == free_before

    def result():
        cnt, xy, resp = d_cnt.cpu().numpy(), d_xy.cpu().numpy(), d_resp.cpu().numpy()
        return [(xy[i, :cnt[i]].copy(), resp[i, :cnt[i]].copy()) for i in range(n)]
    first = result()
    # under stream capture
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        rc = ctx.detect_gftt_noalloc(0, rois, p, d_xy.data_ptr(), d_resp.data_ptr(), d_cnt.data_ptr(), cap)
    assert rc == 0, klib.strerror(rc)
    d_cnt.fill_(-3)
    graph.replay()
    torch.cuda.synchronize()
    _same(result(), first)
    ctx.upload_image(0, img2)          # another frame in the slot: the replayed kernels read it
    ctx.sync()
    graph.replay()
    torch.cuda.synchronize()
    second = result()
    ctx.set_stream(None)
    _same(first, make_other(make_ctx, W, H, img).detect_gftt(0, rois, p))
    _same(second, ctx.detect_gftt(0, rois, p))
    assert sum(len(x) for x, _ in first) > 300 and not all(np.array_equal(a[0], b[0]) for a, b in zip(first, second))


def make_other(make_ctx, W, H, img):
    c = make_ctx(width=W, height=H)
    c.upload_image(0, img)
    return c
