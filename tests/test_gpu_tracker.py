"""Frame-level parity on the GPU: ktlvo_track_stereo_frame / ktlvo_estimate_pose against the
oracle's restatement of the reference's two callbacks on a synthetic stereo corridor."""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def _ate(traj, gt):
    return float(np.sqrt(np.mean(np.sum((traj - gt) ** 2, axis=1))))


def test_tracker_matches_oracle_on_corridor(make_ctx, oracle):
    W, H, NF = 1242, 375, 16   # long enough for a second key frame (tracked count falls below 300)
    scene = synth.CorridorScene(W, H, seed=0x5EED0001)
    ctx = make_ctx()
    ocfg = oracle.default_config(nthreads=8)
    otr = oracle.Tracker(ocfg)
    g_traj, o_traj, gt = [], [], []
    Tgt = np.eye(4)
    kf_seen = 0
    for f in range(NF):
        L, R = scene.stereo_pair(f)
        gl, gc, gkf = ctx.track_stereo_frame(L, R)
        ol, oc, okf = otr.track(L, R)
        assert gkf == okf
        kf_seen += int(gkf)
        if f == 0:
            assert len(gl) == 0 and len(ol) == 0
            continue
        # same features survive the gates, in the same order, with the same bookkeeping
        assert len(gl) == len(ol) and len(gl) > 100, (f, len(gl), len(ol))
        assert np.array_equal(gl["ind"], ol["ind"]) and np.array_equal(gc["ind"], oc["ind"])
        assert np.array_equal(gl["newKeyFrame"], ol["newKeyFrame"])
        # Positions are carried from frame to frame, so the per-call LK tolerance (0.05 px) applies
        # to almost all points while a few chained ones may drift a little further.
        d = np.concatenate([np.abs(gl["u"] - ol["u"]), np.abs(gl["v"] - ol["v"]), np.abs(gc["u"] - oc["u"]),
                            np.abs(gc["v"] - oc["v"])])
        worst, frac_ok = float(d.max()), float((d <= 0.05).mean())
        assert frac_ok >= 0.995 and worst <= 0.25, (f, worst, frac_ok)
        depth_dev = float(np.abs(gl["depth"] / ol["depth"] - 1).max())
        assert depth_dev < 0.02, (f, depth_dev)
        gn, gTl, gTg = ctx.estimate_pose(gl, gc)
        on, oTl, oTg = otr.estimate_pose(ol, oc)
        g_traj.append(gTg[:3, 3].copy())
        o_traj.append(oTg[:3, 3].copy())
        Tgt = Tgt @ scene.relative_pose(f - 1, f)
        gt.append(Tgt[:3, 3].copy())
    assert kf_seen >= 2
    g_traj, o_traj, gt = np.array(g_traj), np.array(o_traj), np.array(gt)
    ate_g, ate_o = _ate(g_traj, gt), _ate(o_traj, gt)
    path = np.abs(gt[-1]).max()
    assert path > 5.0
    # both follow the ground truth, and they agree with each other within 1 % of the path length
    assert ate_g < 0.1 * path and ate_o < 0.1 * path, (ate_g, ate_o)
    assert _ate(g_traj, o_traj) <= 0.01 * path, _ate(g_traj, o_traj)


def test_tracker_edge_cases(make_ctx, klib):
    ctx = make_ctx()
    flat = np.full((375, 1242), 128, np.uint8)
    last, cur, kf = ctx.track_stereo_frame(flat, flat)
    assert len(last) == 0 and not kf
    last, cur, kf = ctx.track_stereo_frame(flat, flat)      # key frame on a featureless image: no corners, no crash
    assert len(last) == 0 and kf
    n, Tl, Tg = ctx.estimate_pose(last, cur)                # empty input: pose stays identity
    assert n == -1 and np.array_equal(Tl, np.eye(4, dtype=np.float32)) and np.array_equal(Tg, Tl)


def test_tracker_output_capacity(make_ctx, klib):
    """Arrays too small for a frame's points: the call says so, fills what fits, and the tracker's own
    state stays whole (the next frame goes on as if nothing had happened)."""
    W, H = 1242, 375
    scene = synth.CorridorScene(W, H, seed=0x5EED0001)
    a, b = make_ctx(), make_ctx()
    for f in range(4):
        L, R = scene.stereo_pair(f)
        ref = a.track_stereo_frame(L, R)
        if f == 2:
            with pytest.raises(klib.KtlvoError) as ei:
                b.track_stereo_frame(L, R, cap=10)
            assert ei.value.status == klib.ERR_CAPACITY
        else:
            got = b.track_stereo_frame(L, R)
            assert len(got[0]) == len(ref[0]) and got[0].tobytes() == ref[0].tobytes()
            assert got[1].tobytes() == ref[1].tobytes() and got[2] == ref[2]


def test_contexts_on_concurrent_host_threads(make_ctx):
    """Different contexts are independent: three host threads, each driving its own context over the
    same frames at the same time, get the very same clouds and poses."""
    import threading
    W, H, NF = 1242, 375, 6
    scene = synth.CorridorScene(W, H, seed=0x5EED0001)
    frames = [scene.stereo_pair(f) for f in range(NF)]
    ctxs = [make_ctx() for _ in range(3)]
    out = [None] * len(ctxs)

    def drive(i):
        res = []
        for L, R in frames:
            last, cur, kf = ctxs[i].track_stereo_frame(L, R)
            _, Tl, Tg = ctxs[i].estimate_pose(last, cur)
            res.append((last.tobytes(), cur.tobytes(), kf, Tl.tobytes(), Tg.tobytes()))
        out[i] = res

    threads = [threading.Thread(target=drive, args=(i,)) for i in range(len(ctxs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(o is not None for o in out)
    assert out[1] == out[0] and out[2] == out[0]
    assert len(out[0][-1][0]) > 100 * 24


def test_hundred_frame_sequence_trajectory(make_ctx, oracle):
    """BASELINE configs[0] shape: 100 stereo frames through tracker + pose on the GPU and through the
    oracle's restatement of the two callbacks.  Last-bit differences of the LK float sums may flip a
    gate for a single feature over such a run, so the bar is on the trajectories: each within 2 % of
    the path length of the ground truth, and within 1 % of each other."""
    W, H, NF = 1242, 375, 100
    scene = synth.CorridorScene(W, H, seed=0x5EED0001)
    ctx = make_ctx()
    otr = oracle.Tracker(oracle.default_config(nthreads=16))
    g_traj, o_traj, gt = [], [], []
    Tgt = np.eye(4)
    same_sets = 0
    for f in range(NF):
        L, R = scene.stereo_pair(f)
        gl, gc, gkf = ctx.track_stereo_frame(L, R)
        ol, oc, okf = otr.track(L, R)
        if f == 0:
            continue
        same_sets += int(len(gl) == len(ol) and np.array_equal(gl["ind"], ol["ind"]))
        _, _, gTg = ctx.estimate_pose(gl, gc)
        _, _, oTg = otr.estimate_pose(ol, oc)
        g_traj.append(gTg[:3, 3].copy())
        o_traj.append(oTg[:3, 3].copy())
        Tgt = Tgt @ scene.relative_pose(f - 1, f)
        gt.append(Tgt[:3, 3].copy())
    g_traj, o_traj, gt = np.array(g_traj), np.array(o_traj), np.array(gt)
    path = float(np.abs(gt[-1]).max())
    assert path > 70.0
    assert _ate(g_traj, gt) < 0.02 * path and _ate(o_traj, gt) < 0.02 * path, (_ate(g_traj, gt), _ate(o_traj, gt))
    assert _ate(g_traj, o_traj) <= 0.01 * path, _ate(g_traj, o_traj)
    assert same_sets >= 0.9 * (NF - 1), same_sets     # in practice every frame keeps the same features
