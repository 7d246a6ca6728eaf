#!/usr/bin/env python3
"""Benchmark of the ktl_vo feature-tracking hot path on MI355X.

Metric (BASELINE.json): tracked features/sec @1241x376, 21x21 pyramidal LK, maxLevel 4.
A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM: P consecutive frame pairs of each of Q synthetic KITTI-shaped sequences (BASELINE
configs[1]: 2000 Shi-Tomasi corners per frame, frame-pair stream), i.e. the Gaussian pyramids of
the Q(P+1) frames (pyr_down_u8, scharr_planes) and QP calcOpticalFlowPyrLK-equivalents in one
launch (pyrlk_track_w21).
A tracked feature is a point that leaves that call with status 1.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Every rank runs the same amount of work on its own sequence (seed + rank): weak scaling, no
data-path collective; per-rank statistics are gathered with one all_gather (RCCL) at the end.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from slam_amd import lib as K  # noqa: E402  (raises if libktlvo_hip.so is missing: no fallback)
from slam_amd import shard, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
HBM_COPY_GBS = 6290.0          # measured float4-copy ceiling
WIN = 21


# the arithmetic the path computes in, by the reading of LK's window sums (ktlvo_lk_params.sums)
DTYPE = {0: "u8 -> i32 fixed point, exact i64-equivalent window sums rounded once to f32",
         2: "u8 -> i32 fixed point, exact i64-equivalent window sums rounded once to f32",
         1: "u8 -> i32 fixed point, f32 window sums in a fixed 64-lane tree"}

STREAM_STEPS = 40   # steps of the host-fed legs' timed regions (amortise the one upload that nothing hides)
PREWARM_STEPS = 40  # untimed steps before the warm-up proper (clock run-up after the host-side set-up)


def lk_bytes_per_feature(levels, win=WIN):
    """Algorithmic bytes of one feature through `levels` pyramid levels: per level one template
    fetch with a 1-px Scharr halo ((w+3)^2) and one search-window fetch ((w+1)^2), plus 29 B of
    point I/O (8 in, 8+1+4 out, rounded up).  Independent of the iteration count."""
    return levels * ((win + 3) ** 2 + (win + 1) ** 2) + 29


def pyramid_bytes(w, h, max_level, win=WIN):
    """Algorithmic bytes of one pyramid build: each level read once, the next written once."""
    total, levels = 0, 1
    for _ in range(max_level):
        nw, nh = (w + 1) // 2, (h + 1) // 2
        if nw <= win or nh <= win:
            break
        total += w * h + nw * nh
        w, h = nw, nh
        levels += 1
    return total, levels


def source_sha1(rel):
    import hashlib
    try:
        return hashlib.sha1(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()
    except OSError:
        return None


def measured_traffic(kernel, waves_per_launch, source="slam_amd/csrc/pyrlk.hip", suffix=""):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/r*_pmc_summary.json, written by profiles/summarize_pmc.py from separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, gfx950 FETCH_SIZE
    doubling applied).  Counters describe the kernel they were collected on: a summary that records the
    SHA-1 of the kernel's source file is only used while that file is unchanged.  None if no summary
    matches this launch shape and source."""
    import glob
    sha = source_sha1(source)
    # (suffix: "_exact_sums" for the counters collected with --lk-sums exact; the plain files are the default reading's)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary%s.json" % suffix)), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        e = doc.get(kernel)
        want = (doc.get("_source_sha1") or {}).get(source)
        if want is not None and want != sha:
            continue   # collected on another version of the kernel
        if e and "hbm_bytes_per_launch" in e and int(e.get("SQ_WAVES", -1)) == int(waves_per_launch):
            return e["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), e
    return None, None, None


def issue_ceiling():
    """Measured issue cost, in shader cycles per wave-instruction per SIMD, of the LK step's instruction mix at
    8 waves per SIMD (tools/probe/issue_rate.hip, result committed under profiles/)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_rate.json")), reverse=True):
        try:
            rows = json.load(open(path))["rows"]
        except (OSError, ValueError, KeyError):
            continue
        pick = {r["kind"]: r for r in rows if r.get("blocks_per_cu") == 8}
        if "lk_step_mix" in pick:
            return {"cycles_per_inst_lk_mix": pick["lk_step_mix"]["cyc_per_inst_simd"],
                    "cycles_per_inst_v_fma_f32": pick.get("v_fma_f32", {}).get("cyc_per_inst_simd"),
                    "cycles_per_inst_v_dot2_i32_i16": pick.get("v_dot2_i32_i16", {}).get("cyc_per_inst_simd"),
                    "clock_ghz_in_probe": pick["lk_step_mix"]["clock_ghz"], "source": os.path.relpath(path, ROOT)}
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=127, help="consecutive frame pairs per sequence and step (<= 127)")
    ap.add_argument("--sequences", type=int, default=2,
                    help="synthetic sequences per rank and step (sequences x pairs <= 256, the capacity of one LK launch)")
    ap.add_argument("--features", type=int, default=None, help="corners per frame (default 2000; batch workload 16384)")
    ap.add_argument("--width", type=int, default=None, help="default 1241; batch workload 3840")
    ap.add_argument("--height", type=int, default=None, help="default 376; batch workload 2160")
    ap.add_argument("--max-level", type=int, default=4)
    ap.add_argument("--min-dist", type=float, default=None, help="default 10; batch workload 8")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline time budget (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="contexts (each with its own stream and slots) the steps alternate between, so one "
                         "step's pyramid build overlaps the previous step's LK tail")
    ap.add_argument("--skip-stream", action="store_true", help="leave out the host-fed (PCIe-inclusive) measurement")
    ap.add_argument("--soak", type=float, default=4.0,
                    help="seconds of extra back-to-back steps after the measurement (a sustained rate, and long enough "
                         "for an external sampler such as rocm-smi to see the GPU busy); reported separately, 0 = none")
    ap.add_argument("--skip-secondary", action="store_true",
                    help="leave out the maxLevel-3 run (profiling passes: one launch shape per kernel)")
    ap.add_argument("--time-kernels", choices=["lk", "all"], default="lk",
                    help="kernels bracketed by HIP events inside the timed region: the dominant one (default; the "
                         "pyramid kernels are then timed in an untimed pass) or all three")
    ap.add_argument("--lk-sums", choices=["default", "exact", "tree"], default="default",
                    help="rounding of LK's window sums (ktlvo_lk_params.sums): the library's default (the float tree), "
                         "exact integer totals rounded once, or the float tree by name")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0002)
    ap.add_argument("--workload", choices=["pairs", "sequence", "ransac", "batch"], default="pairs",
                    help="pairs: the headline frame-pair stream (default); sequence: stereo tracker + pose on a "
                         "synthetic sequence per rank, trajectories gathered (BASELINE configs[0]/[3]); ransac: "
                         "batched hypothesis scoring, 1024 hypotheses x 8000 points (configs[2]); batch: a FIXED "
                         "batch of independent frame pairs sharded round-robin over the ranks (configs[4]: 64 pairs "
                         "of 3840x2160 frames, 16384 corners; strong scaling)")
    ap.add_argument("--batch-pairs", type=int, default=64, help="batch workload: frame pairs in the whole job")
    ap.add_argument("--dump-pairs", default="", help="batch workload: rank 0 saves the per-pair result table here (.npy)")
    ap.add_argument("--frames", type=int, default=20, help="sequence workload: stereo frames per rank")
    ap.add_argument("--rank-offset", type=int, default=0,
                    help="sequence workload: added to the rank when choosing its sequence (lets one GPU replay rank r's input)")
    ap.add_argument("--dump-trajectories", default="", help="sequence workload: rank 0 saves the gathered poses here (.npy)")
    ap.add_argument("--streams", type=int, default=1,
                    help="sequence workload: tracker instances per GPU, one host thread and one HIP stream each")
    a = ap.parse_args()
    # the sequence workload runs the reference's own frame unless told otherwise
    a.seq_width, a.seq_height = a.width or 1242, a.height or 375
    # shapes left open take the workload's BASELINE shape: configs[4] for the fixed batch, configs[1] otherwise
    shape = (3840, 2160, 16384, 8.0) if a.workload == "batch" else (1241, 376, 2000, 10.0)
    if a.width is None: a.width = shape[0]
    if a.height is None: a.height = shape[1]
    if a.features is None: a.features = shape[2]
    if a.min_dist is None: a.min_dist = shape[3]
    # "default" is whatever the library's default configuration says (the float tree)
    a.lk_sums = {"default": K.default_config().lk_sums, "exact": K.LK_SUMS_EXACT, "tree": K.LK_SUMS_TREE}[a.lk_sums]
    return a


def run_sequence(a, rank, world, local, dev):
    """Stereo tracker + RANSAC pose on one synthetic corridor sequence per rank (the reference's 1242x375 frame,
    cameraParameters.h:16-17, unless --width/--height say otherwise: BASELINE configs[0] and [3] name 1241x376);
    every rank runs its own sequence (seed + rank) and the 4x4 poses are gathered with one all_gather."""
    W, H = a.seq_width, a.seq_height
    scene = synth.CorridorScene(W, H, seed=shard.rank_seed(0x5EED0010, rank + a.rank_offset))
    frames = [scene.stereo_pair(f) for f in range(a.frames)]
    # --streams S: S tracker instances (contexts, each with its own HIP stream) driven by S host threads
    # on this rank's GPU, all fed the same sequence; a single instance is latency-bound and leaves most
    # of the GPU idle.  Instance 0 provides the trajectory; the others must reproduce it bit for bit.
    S = max(1, a.streams)
    ctxs = [K.Context(K.default_config(width=W, height=H, cx=scene.cx, cy=scene.cy), device=local) for _ in range(S)]
    all_poses = [np.zeros((a.frames, 16), np.float32) for _ in range(S)]
    stats = [None] * S

    def drive(i):
        ctx, poses = ctxs[i], all_poses[i]
        poses[0] = np.eye(4, dtype=np.float32).reshape(16)
        gt = np.eye(4)
        ate2, tracked, keyframes = 0.0, 0, 0
        for f, (L, R) in enumerate(frames):
            last, cur, kf = ctx.track_stereo_frame(L, R)
            keyframes += int(kf)
            if f == 0:
                continue
            _, _, Tg = ctx.estimate_pose(last, cur)
            poses[f] = Tg.reshape(16)
            tracked += len(last)
            gt = gt @ scene.relative_pose(f - 1, f)
            ate2 += float(np.sum((Tg[:3, 3] - gt[:3, 3]) ** 2))
        ctx.sync()
        stats[i] = (ate2, tracked, keyframes, gt)

    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    if S == 1:
        drive(0)
    else:
        import threading
        threads = [threading.Thread(target=drive, args=(i,)) for i in range(S)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert all(st is not None for st in stats), "a tracker thread failed"
    for i in range(1, S):
        assert np.array_equal(all_poses[i], all_poses[0]) and stats[i][1] == stats[0][1], "instance %d differs" % i
    poses = all_poses[0]
    ate2, tracked, keyframes, gt = stats[0]
    # The reference runs the two halves as two processes (vo.launch:3-5: feature_tracking_node publishes the clouds,
    # visual_odometry_node estimates the pose), so frame N+1 is tracked while frame N's pose is estimated.  The same here
    # with two host threads and a context each (ktlvo_track_stereo_frame on one, ktlvo_estimate_pose on the other, the
    # clouds handed over through a queue); the poses must be those of the run above, bit for bit.
    two_nodes = None
    if S == 1 and world == 1:
        import queue
        import threading
        cfg2 = K.default_config(width=W, height=H, cx=scene.cx, cy=scene.cy)
        ctx_t, ctx_p = K.Context(cfg2, device=local), K.Context(cfg2, device=local)
        poses2 = np.zeros_like(poses)
        poses2[0] = np.eye(4, dtype=np.float32).reshape(16)
        clouds = queue.Queue(maxsize=4)
        failed = []

        def tracking_node():
            try:
                for f, (L, R) in enumerate(frames):
                    last, cur, _ = ctx_t.track_stereo_frame(L, R)
                    if f:
                        clouds.put((f, last, cur))
            except Exception as e:      # the other node must not wait for ever
                failed.append(e)
            clouds.put(None)

        def odometry_node():
            try:
                while True:
                    item = clouds.get()
                    if item is None:
                        return
                    f, last, cur = item
                    poses2[f] = ctx_p.estimate_pose(last, cur)[2].reshape(16)
            except Exception as e:
                failed.append(e)
                while clouds.get() is not None:
                    pass

        ctx_t.sync(); ctx_p.sync()
        t1 = time.perf_counter()
        nodes = [threading.Thread(target=tracking_node), threading.Thread(target=odometry_node)]
        for t in nodes:
            t.start()
        for t in nodes:
            t.join()
        t_two = time.perf_counter() - t1
        assert not failed, failed
        assert np.array_equal(poses2, poses), "the two-node run's poses differ"
        two_nodes = {"value": a.frames / t_two, "unit": "frames/s", "ms_per_frame": t_two / a.frames * 1e3,
                     "note": "tracking and pose estimation on two host threads with a context each, as the reference's two "
                             "nodes run them; poses identical to the single-thread run"}
        ctx_t.close(); ctx_p.close()
    t_max = shard.max_over_ranks(elapsed, dev, world)
    traj = shard.gather_trajectories(torch.from_numpy(poses).to(dev), world)
    rec = torch.tensor([float(a.frames), float(tracked), float(keyframes), np.sqrt(ate2 / max(a.frames - 1, 1)),
                        float(np.abs(gt[:3, 3]).max()), elapsed, 0.0, 0.0], dtype=torch.float64, device=dev)
    recs = shard.gather_records(rec, world)
    if rank == 0 and a.dump_trajectories:
        np.save(a.dump_trajectories, np.asarray(traj))   # [world, frames, 16], for the sharding parity test
    if rank == 0:
        print(json.dumps({
            "metric": "stereo frames/sec, ktl_vo frontend + RANSAC pose (%dx%d, reference parameters)" % (W, H),
            "value": float(recs[:, 0].sum()) * S / t_max, "unit": "frames/s", "n_gpus": world, "scaling": "weak",
            "data": "synthetic", "frames_per_rank": a.frames * S, "tracker_instances_per_gpu": S,
            "ms_per_frame": t_max / a.frames * 1e3,
            "per_rank": [{"tracked_points": r[1], "key_frames": r[2], "ate_rmse_m": r[3], "path_m": r[4],
                          "elapsed_s": r[5]} for r in recs.tolist()],
            "trajectory_shape": list(traj.shape),
            "two_nodes": two_nodes,
            "note": "per frame: 2 host->device uploads (pinned staging), 1 pyramid + Scharr build for the two new "
                    "images, ONE LK launch for the reference's 8 calls (six tracks per point as two chains), one gate kernel, "
                    "RANSAC pose with one scoring launch"}))
    for c in ctxs:
        c.close()


def run_ransac(a, rank, world, local, dev):
    """K9 alone: 1024 pose hypotheses x 8000 points (BASELINE configs[2]), through the host-pointer call
    (copies included) and through the device-pointer call with every array resident in HBM."""
    rng = np.random.default_rng(3 + rank)
    Kc = np.array([1093.5, 1093.5, 960.0, 540.0])
    n, Hh = 8000, 1024
    X = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(4, 60, n)], 1).astype(np.float32)
    uv = np.stack([Kc[0] * X[:, 0] / X[:, 2] + Kc[2], Kc[1] * X[:, 1] / X[:, 2] + Kc[3]], 1).astype(np.float32)
    Rt = np.tile(np.concatenate([np.eye(3).reshape(9), np.zeros(3)]), (Hh, 1))
    Rt[:, 9:] += rng.normal(size=(Hh, 3)) * 1e-3
    ctx = K.Context(K.default_config(width=1920, height=1080, fx=Kc[0], fy=Kc[1], cx=Kc[2], cy=Kc[3]), device=local)
    for _ in range(a.warmup):
        ctx.ransac_score(X, uv, Rt, Kc, np.float32(4.0))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        counts = ctx.ransac_score(X, uv, Rt, Kc, np.float32(4.0))
    elapsed = time.perf_counter() - t0
    # the same with every array resident in HBM (ktlvo_ransac_score_dev), at the reference's 0.1 px
    # threshold (thr^2 = 0.01) and at the looser 2 px one (thr^2 = 4)
    dX, dx = torch.from_numpy(X).to(dev), torch.from_numpy(uv).to(dev)
    dRt = torch.from_numpy(Rt).to(dev)
    dcnt = torch.zeros(Hh, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    algo = 20.0 * n * Hh + 100.0 * Hh
    resident = {}
    for thr in (0.01, 4.0):
        for _ in range(a.warmup):
            ctx.ransac_score_dev(dX.data_ptr(), dx.data_ptr(), n, dRt.data_ptr(), Hh, Kc, thr, dcnt.data_ptr())
        ctx.sync()
        ctx.profile_enable(True)
        ctx.profile_reset()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            ctx.ransac_score_dev(dX.data_ptr(), dx.data_ptr(), n, dRt.data_ptr(), Hh, Kc, thr, dcnt.data_ptr())
        ctx.sync()
        wall = (time.perf_counter() - t1) / a.steps
        ms, nl = ctx.profile_get(K.K_RANSAC)
        ctx.profile_enable(False)
        k_s = ms / max(nl, 1) * 1e-3
        dev_counts = dcnt.cpu().numpy()
        if thr == 4.0:
            assert np.array_equal(dev_counts, counts), "device-pointer and host-pointer scoring disagree"
        resident["thr_sq_%g" % thr] = {
            "scores_per_s": n * Hh / wall, "ms_per_call": wall * 1e3, "kernel_ms": k_s * 1e3,
            "kernel_scores_per_s": n * Hh / k_s if k_s > 0 else None,
            "kernel_GBps": algo / k_s / 1e9 if k_s > 0 else None, "max_count": int(dev_counts.max())}
    if rank == 0:
        r4 = resident["thr_sq_4"]
        print(json.dumps({
            "metric": "RANSAC hypothesis x point scores/sec (1024 hypotheses x 8000 points)",
            "value": r4["scores_per_s"], "unit": "scores/s (arrays resident in HBM, thr^2 = 4)",
            "resident": resident,
            "host_pointer_call": {"scores_per_s": n * Hh * a.steps / elapsed,
                                  "note": "ktlvo_ransac_score: 160 KB + 98 KB host-to-device and 4 KB back per call"},
            "roofline": {"bound": "hbm", "kernel": "ransac_score", "achieved": r4["kernel_GBps"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": r4["kernel_GBps"] / HBM_PEAK_GBS if r4["kernel_GBps"] else None,
                         "algorithmic_bytes": algo},
            "max_count": int(counts.max()), "n_gpus": world, "data": "synthetic"}))
    ctx.close()


def run_batch(a, rank, world, local, dev):
    """BASELINE configs[4]: a fixed batch of independent frame pairs (pair p = frames 0 and 1 of the corridor
    with seed 0x5EED0100 + p), sharded round-robin by pair index; every rank builds the pyramids of its pairs
    and tracks them in one LK launch per step.  Total work is fixed, so this is strong scaling.  The per-pair
    results (tracked count, CRC-32 of the status bytes and of the tracked positions) are gathered with one
    all_gather and are the same table whatever the number of ranks."""
    import zlib
    W, H, NF, NB = a.width, a.height, a.features, a.batch_pairs
    mine = shard.shard_round_robin(NB, rank, world)
    n_loc = len(mine)
    assert n_loc <= 256
    pyr_bytes, levels = pyramid_bytes(W, H, a.max_level)
    # (the renderer runs in child processes; eight pairs are rendered at a time)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as pool:
        rendered = list(pool.map(lambda p: synth.render_corridor_frames(W, H, (0x5EED0100 + p) & 0xFFFFFFFF, 2, workers=2), mine))
    frames = [f for pair in rendered for f in pair]
    del rendered
    cfg = K.default_config(width=W, height=H, num_slots=max(2 * n_loc, 1), max_level=a.max_level,
                           max_points=max(NF, 1024), cx=601.8873 + (W - 1242) * 0.5, cy=183.1104 + (H - 375) * 0.5)
    ctx = K.Context(cfg, device=local)
    for s, f in enumerate(frames):
        ctx.upload_image(s, f)
    ctx.sync()
    slots = list(range(2 * n_loc))
    prev_slots, next_slots = slots[0::2], slots[1::2]
    pts = np.zeros((max(n_loc, 1), NF, 2), np.float32)
    for j, s in enumerate(prev_slots):
        md = a.min_dist
        (xy, _), = ctx.detect_gftt(s, [(0, 0, W, H)], K.gftt_params(NF, 0.01, md, 3, 0, 0.0))
        while len(xy) < NF:   # a frame short of corners at this spacing: tighten it (deterministic per pair)
            md *= 0.9
            if md < 1.0:
                raise SystemExit("pair %d yields %d corners < %d even at minDistance 1" % (mine[j], len(xy), NF))
            (xy, _), = ctx.detect_gftt(s, [(0, 0, W, H)], K.gftt_params(NF, 0.01, md, 3, 0, 0.0))
        pts[j] = xy
    d_prev = torch.from_numpy(pts).to(dev)
    d_next = torch.zeros_like(d_prev)
    d_status = torch.zeros((max(n_loc, 1), NF), dtype=torch.uint8, device=dev)
    d_err = torch.zeros((max(n_loc, 1), NF), dtype=torch.float32, device=dev)
    lk = K.lk_params(WIN, a.max_level, sums=a.lk_sums)

    def step():
        if n_loc:
            ctx.build_pyramids(slots)
            ctx.pyrlk_batch_dev(prev_slots, next_slots, d_prev.data_ptr(), NF, d_next.data_ptr(), d_status.data_ptr(),
                                d_err.data_ptr(), lk)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()

    for _ in range(max(a.warmup, 1)):
        step()
    fence()
    ctx.profile_enable(2 << K.K_PYRLK)
    ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k3_ms, k3_n = ctx.profile_get(K.K_PYRLK)
    ctx.profile_enable(False)
    st, nx = d_status.cpu().numpy(), d_next.cpu().numpy()
    rows = np.zeros((n_loc, len(shard.PAIR_FIELDS)), np.int64)
    for j, p in enumerate(mine):
        ok = st[j] == 1
        rows[j] = (p, NF, int(ok.sum()), zlib.crc32(st[j].tobytes()), zlib.crc32(nx[j][ok].tobytes()))
    table = shard.gather_pair_records(torch.from_numpy(rows).to(dev), NB, world)
    t_max = shard.max_over_ranks(elapsed, dev, world)
    rec = torch.tensor([float(n_loc * NF), float(rows[:, 2].sum()), float(n_loc), k3_ms, float(k3_n), 0.0, 0.0, elapsed],
                       dtype=torch.float64, device=dev)
    recs = shard.gather_records(rec, world)
    if rank == 0:
        if a.dump_pairs:
            np.save(a.dump_pairs, table)
        tracked = int(table[:, 2].sum())
        bpf = lk_bytes_per_feature(levels)
        k3_avg_s = max(float(r[3]) / max(float(r[4]), 1.0) for r in recs.tolist()) * 1e-3   # slowest rank's launch
        print(json.dumps({
            "metric": "tracked features/sec @%dx%d, 4-lvl 21x21 PyrLK, fixed batch of %d frame pairs sharded over 1-8 GPU; "
                      "%% of HBM peak" % (W, H, NB),
            "value": tracked * a.steps / t_max, "unit": "tracked features/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": t_max / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": DTYPE[a.lk_sums], "data": "synthetic",
            "config": {"workload": "%sfixed batch of %d independent %dx%d frame pairs, %d Shi-Tomasi corners each, sharded "
                                   "round-robin by pair index; maxLevel=%d, 21x21 window, 30 it / eps 0.01" % (
                                       "BASELINE configs[4]: " if (W, H, NF, NB) == (3840, 2160, 16384, 64) else "",
                                       NB, W, H, NF, a.max_level),
                       "pairs_total": NB, "pairs_per_rank": [int(r[2]) for r in recs.tolist()],
                       "features_in_per_step": NB * NF, "tracked_per_step": tracked,
                       "pairs_per_s": NB * a.steps / t_max},
            "roofline": {"bound": "hbm", "kernel": "pyrlk_track_w21", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "achieved": bpf * max(int(r[0]) for r in recs.tolist()) / k3_avg_s / 1e9 if k3_avg_s > 0 else None,
                         "frac": bpf * max(int(r[0]) for r in recs.tolist()) / k3_avg_s / 1e9 / HBM_PEAK_GBS if k3_avg_s > 0 else None,
                         "traffic": None, "bytes_model_per_feature": bpf, "avg_launch_ms": k3_avg_s * 1e3,
                         "note": "per GPU: the slowest rank's launch (all ranks hold the same number of pairs when the "
                                 "rank count divides the batch)"},
            "pair_table_crc32": zlib.crc32(table.tobytes()),
            "per_rank": [{"features_in": r[0], "tracked": r[1], "pairs": r[2], "k3_ms": r[3], "elapsed_s": r[7]}
                         for r in recs.tolist()]}))
    ctx.close()


def cpu_baseline(frames, pts, max_level, budget_s):
    """The oracle (a port of the OpenCV path the reference calls) timed on the host cores: per pair both pyramids, the
    previous image's Scharr planes and the LK loop, as one cv::calcOpticalFlowPyrLK call does, every stage cut over
    the threads of a persistent pool as cv::parallel_for_ does it (the reference never calls cv::setNumThreads, so
    all cores is its configuration).  Bounded sample: the first pairs of the same workload, repeated until the time
    budget is spent.  Timed with the serial float accumulator of OpenCV's scalar path and with four accumulators, as an
    SSE2 build of OpenCV sums; on one thread; and with the threads taking whole pairs side by side instead of stripes
    of one pair (`value` is the faster arrangement, named in `arrangement`)."""
    import oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores))))
    n_pairs = min(len(frames) - 1, 4)
    base = O.Baseline()

    def timed(threads, budget, sums):
        tracked, t0, reps = 0, time.perf_counter(), 0
        while True:
            for i in range(n_pairs):
                _, st, _ = base.track_pair(frames[i], frames[i + 1], pts[i], max_level, WIN, threads, sums)
                tracked += int(st.sum())
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget or reps >= 1000:
                return tracked / el, reps, el

    def timed_pairs_in_parallel(budget):
        """`cores` host threads, each tracking whole pairs on its own (single-threaded calls side by side): what a CPU
        does best with a STREAM of independent pairs; no stage waits for the slowest of 16 stripes."""
        import threading
        stop_at = time.perf_counter() + budget
        counts = [0] * cores

        def worker(k):
            i = k % n_pairs
            while time.perf_counter() < stop_at:
                _, st, _ = base.track_pair(frames[i], frames[i + 1], pts[i], max_level, WIN, 1, O.SUMS_SERIAL)
                counts[k] += int(st.sum())
                i = (i + 1) % n_pairs
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        return sum(counts) / el, el

    timed(cores, 0.5, O.SUMS_SERIAL)                      # pool start-up and page faults of the first calls
    v_all, reps, el = timed(cores, budget_s * 0.4, O.SUMS_SERIAL)
    v_four, reps4, el4 = timed(cores, budget_s * 0.15, O.SUMS_FOUR_LANES)
    v_one, reps1, el1 = timed(1, max(1.0, budget_s * 0.2), O.SUMS_SERIAL)
    v_par, el_par = timed_pairs_in_parallel(max(1.0, budget_s * 0.25)) if cores > 1 else (v_one, el1)
    # Gauss-Newton steps per level of the first pair (levels listed from level 0 up)
    it, _ = O.lk_iterations(O.Pyramid(frames[0], max_level, WIN, True), O.Pyramid(frames[1], max_level, WIN, False),
                            pts[0], max_level=max_level)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    best_par = v_par > v_all
    return {"value": max(v_all, v_par), "unit": "tracked features/s", "cores": cores, "kind": "port",
            "arrangement": "pairs_in_parallel" if best_par else "stripes_within_a_pair",
            "sample": "first %d frame pairs x %d features of the same workload, repeated for %.1f s; pyramids + Scharr + LK per "
                      "pair (one calcOpticalFlowPyrLK each); serial float accumulators (OpenCV's scalar path); the faster of two "
                      "arrangements of the %d host threads: %s"
                      % (n_pairs, pts.shape[1], el_par if best_par else el, cores,
                         "whole pairs side by side, a single-threaded call each" if best_par
                         else "one pair at a time, image rows and point stripes taken by the pooled threads"),
            "stripes_within_a_pair": {"value": v_all, "cores": cores,
                                      "sample": "%d repetitions, %.1f s; one pair at a time, every stage cut over %d pooled threads "
                                                "as cv::parallel_for_ cuts it" % (reps, el, cores)},
            "four_accumulators": {"value": v_four, "cores": cores, "sample": "%d repetitions, %.1f s; window sums in four "
                                  "float accumulators, as a 4-lane SIMD build of OpenCV adds them" % (reps4, el4)},
            "single_thread": {"value": v_one, "cores": 1, "sample": "%d repetitions, %.1f s" % (reps1, el1)},
            "pairs_in_parallel": {"value": v_par, "cores": cores, "sample": "%.1f s; %d threads, each tracking whole pairs with "
                                  "single-threaded calls (the arrangement for a stream of independent pairs)" % (el_par, cores)},
            "all_cores_over_one_thread": v_all / v_one if v_one > 0 else None,
            "pairs_in_parallel_over_one_thread": v_par / v_one if v_one > 0 else None,
            "build": "g++ %s -ffp-contract=off" % base.build,
            "cpu_model": model,
            "lk_iterations_per_level_mean": [float(x) for x in it.mean(1)],
            "lk_iterations_per_feature_mean": float(it.sum(0).mean())}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # KTLVO_BENCH_REHEARSAL=1: every rank uses GPU 0 and gloo, to exercise the N > 1 code path on
    # a one-GPU box (RCCL refuses two ranks on one device).  Never set for a measurement.
    rehearsal = os.environ.get("KTLVO_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # Under torchrun (RANK set) the process group is always created, for one rank too: a one-GPU box then runs
    # the same RCCL initialisation, barrier, all_gather and all_reduce as an 8-GPU node.
    grouped = world > 1 or "RANK" in os.environ
    if grouped:
        # RCCL prints a version banner on the process's stdout when its first communicator comes up; stdout
        # carries the one JSON line, so file descriptor 1 points at stderr until that has happened
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    if a.workload != "pairs":
        {"sequence": run_sequence, "ransac": run_ransac, "batch": run_batch}[a.workload](a, rank, world, local, dev)
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return
    W, H, P, NF = a.width, a.height, a.pairs, a.features
    assert 1 <= P <= 127   # the synthetic corridor (120 m, 0.8 m per frame) ends after about 130 frames
    pyr_bytes, levels = pyramid_bytes(W, H, a.max_level)

    # ---- synthetic input: Q sequences per rank, P consecutive pairs of each ---------------------
    # (the corridor ends after about 130 frames and one LK launch takes up to 256 pairs: two sequences
    # fill it, which halves the share of the launch's ramp and drain)
    Q = a.sequences
    assert 1 <= Q and Q * P <= 256
    scene = synth.CorridorScene(W, H, seed=shard.rank_seed(a.seed, rank))
    frames = []
    for q in range(Q):
        frames += synth.render_corridor_frames(W, H, shard.rank_seed(a.seed + 0x1000 * q, rank), P + 1, workers=8)
    NI, NP = Q * (P + 1), Q * P          # images and pairs per step
    cfg = K.default_config(width=W, height=H, num_slots=NI, max_level=a.max_level, max_points=max(NF, 1024),
                           cx=scene.cx, cy=scene.cy)
    ctxs = [K.Context(cfg, device=local) for _ in range(max(1, a.pipeline))]
    ctx = ctxs[0]
    def upload_all(src):
        t = time.perf_counter()
        for s, f in enumerate(src):
            ctx.upload_image(s, f)
        ctx.sync()
        return time.perf_counter() - t

    upload_all(frames)                       # first pass allocates the staging ring
    t_h2d = min(upload_all(frames) for _ in range(3))   # frames in pageable memory: staged through the pinned ring
    pinned = [torch.from_numpy(np.ascontiguousarray(f)).pin_memory() for f in frames] if rank == 0 else []
    src = [t.numpy() for t in pinned]
    t_h2d_pinned = min(upload_all(src) for _ in range(3)) if pinned else None   # read by DMA in place
    del pinned, src
    for c in ctxs[1:]:
        for s, f in enumerate(frames):
            c.upload_image(s, f)
        c.sync()
    # corners of every pair's first frame, from the product detector (Shi-Tomasi, whole frame)
    gp = K.gftt_params(NF, 0.01, a.min_dist, 3, 0, 0.0)
    slots = list(range(NI))
    prev_slots = [q * (P + 1) + i for q in range(Q) for i in range(P)]
    next_slots = [s + 1 for s in prev_slots]
    pts = np.zeros((NP, NF, 2), np.float32)
    min_dist_used = a.min_dist
    for j, i in enumerate(prev_slots):
        md = a.min_dist
        (xy, _), = ctx.detect_gftt(i, [(0, 0, W, H)], gp)
        while len(xy) < NF:
            # a frame short of corners at this spacing: tighten the spacing for that frame until the
            # detector yields the workload's count (deterministic per seed)
            md *= 0.9
            if md < 1.0:
                raise SystemExit("frame %d yields %d corners < %d even at minDistance 1" % (i, len(xy), NF))
            (xy, _), = ctx.detect_gftt(i, [(0, 0, W, H)], K.gftt_params(NF, 0.01, md, 3, 0, 0.0))
        min_dist_used = min(min_dist_used, md)
        pts[j] = xy
    d_prev = torch.from_numpy(pts).to(dev)
    outs = [(torch.zeros_like(d_prev), torch.zeros((NP, NF), dtype=torch.uint8, device=dev),
             torch.zeros((NP, NF), dtype=torch.float32, device=dev)) for _ in ctxs]
    d_next, d_status, d_err = outs[0]
    lk = K.lk_params(WIN, a.max_level, sums=a.lk_sums)
    step_no = [0]

    def step():
        i = step_no[0] % len(ctxs)
        step_no[0] += 1
        c, (o_next, o_status, o_err) = ctxs[i], outs[i]
        c.build_pyramids(slots)
        c.pyrlk_batch_dev(prev_slots, next_slots, d_prev.data_ptr(), NF, o_next.data_ptr(), o_status.data_ptr(),
                          o_err.data_ptr(), lk)

    def fence():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()

    # The GPU idles while the inputs are prepared on the host, and its clocks take some ten launches (about
    # 30 ms) to come back up: the first launches after an idle spell run up to 20 % longer (kernel trace).
    # A fixed untimed run-up before the requested warm-up steps puts the timed region in the steady state.
    for _ in range(PREWARM_STEPS):
        step()
    fence()
    for _ in range(max(a.warmup, len(ctxs))):
        step()
    fence()
    # In the timed region only the dominant kernel is bracketed by events (an event pair costs a few
    # microseconds of stream time; three pairs per step are 3-4 % of this step); the two pyramid kernels
    # are timed in a short untimed pass afterwards.
    for c in ctxs:
        c.profile_enable(True if a.time_kernels == "all" else 2 << K.K_PYRLK)
        c.profile_reset()
    fence()
    step_no[0] = 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k3_ms = k3_n = k1_ms = k1_n = k2_ms = k2_n = 0
    for c in ctxs:
        m, n = c.profile_get(K.K_PYRLK); k3_ms += m; k3_n += n
    if a.time_kernels != "all":
        for c in ctxs:
            c.profile_enable(True)
            c.profile_reset()
        for _ in range(max(3 * len(ctxs), a.steps // 4)):
            step()
        fence()
    for c in ctxs:
        m, n = c.profile_get(K.K_PYRDOWN); k1_ms += m; k1_n += n
        m, n = c.profile_get(K.K_SCHARR); k2_ms += m; k2_n += n
        c.profile_enable(False)

    tracked = int(d_status.sum().item())           # per step on this rank (deterministic input)

    # ---- the detector on the step's first frames, all in one call (reported beside the step, never part of it) ----
    # BASELINE's stream detects its corners on every frame.  One frame at a time the whole-frame detector costs
    # 0.5 ms, forty times the frame's share of the LK launch; ktlvo_detect_gftt_frames_dev runs the batch with
    # the launches of one detection, corners landing where ktlvo_pyrlk_batch_dev reads them.
    detect = None
    if world == 1 and not a.skip_secondary and NP <= 256:   # single-GPU runs only: the other ranks would wait for it
        d_cxy = torch.zeros((NP, NF, 2), device=dev)
        d_cr = torch.zeros((NP, NF), device=dev)
        d_cc = torch.zeros(NP, dtype=torch.int32, device=dev)

        def detect_all():
            ctx.detect_gftt_frames_dev(prev_slots, gp, d_cxy.data_ptr(), d_cr.data_ptr(), d_cc.data_ptr(), NF)
        for _ in range(2):
            detect_all()
        ctx.sync()
        reps = 5
        t = time.perf_counter()
        for _ in range(reps):
            detect_all()
        ctx.sync()
        t_det = (time.perf_counter() - t) / reps
        cc = d_cc.cpu().numpy()
        (xy0, _), = ctx.detect_gftt(prev_slots[0], [(0, 0, W, H)], gp)
        assert int(cc[0]) == len(xy0) and np.array_equal(d_cxy[0, :len(xy0)].cpu().numpy(), xy0)
        # the whole stream step: detect on the first frames, build the pyramids, track the detected corners (frames
        # short of NF corners at this spacing track their list's unused tail as well; only real corners are counted).
        # Two arrangements: the three calls one after the other, and ktlvo_detect_build_frames_dev, which runs the
        # pyramid build on a side stream beside the detection.
        o_next, o_status, o_err = outs[0]

        def full_step(side_by_side):
            if side_by_side == "one_call":
                ctx.detect_build_track_dev(prev_slots, next_slots, gp, d_cxy.data_ptr(), d_cr.data_ptr(), d_cc.data_ptr(), NF, slots,
                                           o_next.data_ptr(), o_status.data_ptr(), o_err.data_ptr(), lk)
                return
            if side_by_side:
                ctx.detect_build_frames_dev(prev_slots, gp, d_cxy.data_ptr(), d_cr.data_ptr(), d_cc.data_ptr(), NF, slots)
            else:
                detect_all()
                ctx.build_pyramids(slots)
            ctx.pyrlk_batch_dev(prev_slots, next_slots, d_cxy.data_ptr(), NF, o_next.data_ptr(), o_status.data_ptr(),
                                o_err.data_ptr(), lk)

        def timed_full(side_by_side, upload=None):
            for _ in range(3):
                if upload is not None:
                    ctx.upload_images(slots, upload)
                full_step(side_by_side)
            ctx.sync()
            # fed from the host the timed region starts with an empty pipeline: its first upload (2.1 ms) runs under
            # nothing, which only many steps amortise (40 steps: 0.05 ms per step; 10 steps would add 0.2 ms to each)
            n = reps if upload is None else max(reps, STREAM_STEPS)
            t = time.perf_counter()
            for _ in range(n):
                if upload is not None:
                    ctx.upload_images(slots, upload)
                full_step(side_by_side)
            ctx.sync()
            return (time.perf_counter() - t) / n

        reps = max(5, a.steps // 2)
        t_full = timed_full(False)
        real = torch.arange(NF, device=dev)[None, :] < d_cc[:, None].to(torch.int64)
        tracked_full = int((o_status.bool() & real).sum().item())
        ref_bits = (o_next.clone(), o_status.clone())
        t_side = timed_full(True)
        assert torch.equal(o_next, ref_bits[0]) and torch.equal(o_status, ref_bits[1]), "side-by-side step differs"
        t_one = timed_full("one_call")
        assert torch.equal(o_next, ref_bits[0]) and torch.equal(o_status, ref_bits[1]), "one-call step differs"
        # ... and fed from the host: the step's frames uploaded from pinned memory every step (one DMA on the copy
        # stream under the previous step's kernels, one ingest launch), then detected, built and tracked: the whole
        # configs[1] stream as one number
        stream_full = None
        if not a.skip_stream:
            host_f = torch.from_numpy(np.stack(frames)).pin_memory()
            t_sf = timed_full("one_call", host_f.numpy())
            assert torch.equal(o_next, ref_bits[0]) and torch.equal(o_status, ref_bits[1]), "host-fed step differs"
            stream_full = {"ms_per_step": t_sf * 1e3, "value": tracked_full / t_sf, "unit": "tracked features/s",
                           "tracked_per_step": tracked_full,
                           "note": "per step: ktlvo_upload_images (all %d frames, %.1f MB, one DMA from pinned memory on the copy "
                                   "stream + one ingest launch), ktlvo_detect_build_track_dev (corners of the %d first frames and "
                                   "all pyramids side by side, the LK launch on the detected corners behind them); one context "
                                   "running ahead of the device, every step's results identical to the resident step's"
                                   % (NI, NI * W * H / 1e6, NP)}
            del host_f
        del ref_bits
        # the detector's three kernels on their own (events around each launch of the batch)
        cand_px = NP * W * H
        detect = {"ms_per_step": t_det * 1e3, "frames": NP, "corners": int(cc.sum()), "us_per_frame": t_det / NP * 1e6,
                  "algorithmic_bytes": cand_px + 12 * int(cc.sum()),
                  "achieved_GBps": (cand_px + 12 * int(cc.sum())) / t_det / 1e9,
                  "detect_build_track": {"ms_per_step": t_full * 1e3, "tracked_per_step": tracked_full,
                                         "value": tracked_full / t_full, "unit": "tracked features/s",
                                         "arrangement": "ktlvo_detect_gftt_frames_dev, ktlvo_build_pyramids, ktlvo_pyrlk_batch_dev in turn"},
                  "detect_build_track_side_by_side": {"ms_per_step": t_side * 1e3, "tracked_per_step": tracked_full,
                                                      "value": tracked_full / t_side, "unit": "tracked features/s",
                                                      "arrangement": "ktlvo_detect_build_frames_dev (pyramid build on a side stream "
                                                                     "beside the detection), ktlvo_pyrlk_batch_dev"},
                  "detect_build_track_one_call": {"ms_per_step": t_one * 1e3, "tracked_per_step": tracked_full,
                                                  "value": tracked_full / t_one, "unit": "tracked features/s",
                                                  "arrangement": "ktlvo_detect_build_track_dev: the side-by-side step with the LK "
                                                                 "launch queued behind the detection before the host waits for "
                                                                 "the detection's flags"},
                  "stream_full": stream_full,
                  "note": "ktlvo_detect_gftt_frames_dev: Shi-Tomasi corners (quality 0.01, minDistance %g, at most %d) of the "
                          "step's %d first frames in one call, output on the device; algorithmic bytes = W H per frame read + "
                          "12 B per corner written" % (a.min_dist, NF, NP)}
        step()                                   # the resident-input step's outputs back in place
        ctx.sync()
        del d_cxy, d_cr, d_cc

    # ---- the same step fed from host memory every time (frame-pair STREAM, never the headline) ------------
    # One batched DMA per step (ktlvo_upload_images from a pinned buffer).  The library issues that DMA on the
    # context's copy stream and unpacks it on the context's stream, so with ONE context the upload of step N+1
    # (called as soon as step N is queued) runs under step N's kernels.  Measured: every step waited for (nothing
    # overlaps), one context running ahead, and three contexts taken in turn (a stream each).
    streamed = None
    if world == 1 and not a.skip_stream:
        host = torch.from_numpy(np.stack(frames)).pin_memory()
        NS = 3
        two = list(ctxs[:NS]) + [K.Context(cfg, device=local) for _ in range(NS - min(NS, len(ctxs)))]
        so = list(outs[:NS]) + [tuple(torch.zeros_like(t) for t in outs[0]) for _ in range(NS - min(NS, len(outs)))]

        def stream_step(k, n_ctx, wait):
            c, (o_next, o_status, o_err) = two[k % n_ctx], so[k % n_ctx]
            c.upload_images(slots, host.numpy())
            c.build_pyramids(slots)
            c.pyrlk_batch_dev(prev_slots, next_slots, d_prev.data_ptr(), NF, o_next.data_ptr(), o_status.data_ptr(),
                              o_err.data_ptr(), lk)
            if wait:
                c.sync()

        def timed_stream(n_ctx, wait, n):
            # the run-up has to bring the clocks back up after the arrangement before it (waiting for every upload leaves
            # the device idle half of the time, and the launches after such a spell run up to 20 % longer for some 30 ms)
            for k in range(4 if wait else PREWARM_STEPS):
                stream_step(k, n_ctx, wait)
            for c in two:
                c.sync()
            t = time.perf_counter()
            for k in range(n):
                stream_step(k, n_ctx, wait)
            for c in two:
                c.sync()
            return (time.perf_counter() - t) / n

        def timed_dma(n):
            two[0].sync()
            t = time.perf_counter()
            for _ in range(n):
                two[0].upload_images(slots, host.numpy())
            two[0].sync()
            return (time.perf_counter() - t) / n

        n_s = max(STREAM_STEPS, a.steps)   # (the timed region's first upload is not hidden: see timed_full)
        t_dma = timed_dma(n_s)
        t_serial = timed_stream(1, True, n_s)
        t_one = timed_stream(1, False, n_s)
        t_three = timed_stream(NS, False, n_s)
        assert all(int(o[1].sum().item()) == tracked for o in so)
        streamed = {"ms_per_step": {"upload_alone": t_dma * 1e3, "upload_then_compute_waited_for": t_serial * 1e3,
                                    "one_context_running_ahead": t_one * 1e3, "three_contexts_in_turn": t_three * 1e3,
                                    "compute_alone": None},
                    "host_link_GBps": NI * W * H / t_dma / 1e9,
                    "value_no_overlap": tracked / t_serial, "value_overlapped": tracked / t_one,
                    "value_three_contexts": tracked / t_three,
                    "note": "rank 0: all %d frames of a step (%.1f MB) uploaded from pinned host memory every step in ONE "
                            "DMA + one ingest launch (ktlvo_upload_images); overlapped = one context, the DMA of step "
                            "N+1 on its copy stream under the kernels of step N" % (NI, NI * W * H / 1e6)}
        for c in two[len(ctxs):]:
            c.close()
        two[0].upload_images(slots, host.numpy())      # leave context 0 as the main run had it
        two[0].build_pyramids(slots)
        two[0].sync()
        del host
    # the same step with maxLevel 3 (4 pyramid images), in case "4-level" is read that way
    ml3 = None
    if world == 1 and a.max_level == 4 and not a.skip_secondary:
        lk3 = K.lk_params(WIN, 3, sums=a.lk_sums)
        n3 = max(3, a.steps // 4)

        def step3():
            ctx.build_pyramids(slots, max_level=3)
            ctx.pyrlk_batch_dev(prev_slots, next_slots, d_prev.data_ptr(), NF, d_next.data_ptr(),
                                d_status.data_ptr(), d_err.data_ptr(), lk3)
        step3()
        fence()
        t3 = time.perf_counter()
        for _ in range(n3):
            step3()
        fence()
        t3 = time.perf_counter() - t3
        tr3 = int(d_status.sum().item())
        ml3 = {"value": tr3 * n3 / t3, "unit": "tracked features/s", "steps": n3, "tracked_per_step": tr3}
        ctx.build_pyramids(slots)   # leave the slots as the main run built them
        fence()
    # the same step with the other reading of the window sums (float tree instead of exact totals, or the reverse)
    other_sums = None
    if world == 1 and not a.skip_secondary:
        alt = K.LK_SUMS_EXACT if a.lk_sums == K.LK_SUMS_TREE else K.LK_SUMS_TREE
        lk_alt = K.lk_params(WIN, a.max_level, sums=alt)
        n_alt = max(5, a.steps // 2)

        def step_alt():
            ctx.build_pyramids(slots)
            ctx.pyrlk_batch_dev(prev_slots, next_slots, d_prev.data_ptr(), NF, d_next.data_ptr(), d_status.data_ptr(),
                                d_err.data_ptr(), lk_alt)
        for _ in range(3):
            step_alt()
        fence()
        ctx.profile_enable(2 << K.K_PYRLK)
        ctx.profile_reset()
        t_alt = time.perf_counter()
        for _ in range(n_alt):
            step_alt()
        fence()
        t_alt = time.perf_counter() - t_alt
        ms_alt, nl_alt = ctx.profile_get(K.K_PYRLK)
        ctx.profile_enable(False)
        tr_alt = int(d_status.sum().item())
        other_sums = {"sums": "exact" if alt == K.LK_SUMS_EXACT else "tree", "value": tr_alt * n_alt / t_alt,
                      "unit": "tracked features/s", "steps": n_alt, "tracked_per_step": tr_alt,
                      "ms_per_step": t_alt / n_alt * 1e3, "lk_launch_ms": ms_alt / max(nl_alt, 1)}
        step()                                   # the main reading's outputs back in place
        fence()
    soak = None
    if a.soak > 0:
        fence()
        t_s, n_s = time.perf_counter(), 0
        while time.perf_counter() - t_s < a.soak:
            for _ in range(20):
                step()
            n_s += 20
            for c in ctxs:
                c.sync()
        t_s = time.perf_counter() - t_s
        soak = {"seconds": t_s, "steps": n_s, "value_this_rank": tracked * n_s / t_s}
    rec = torch.tensor([float(NP * NF), float(tracked), float(NP), k3_ms, float(k3_n), k1_ms, float(k1_n), elapsed],
                       dtype=torch.float64, device=dev)
    t_max = shard.max_over_ranks(elapsed, dev, world)
    recs = shard.gather_records(rec, world)        # the only collective: fixed-size per-rank statistics

    if rank == 0:
        agg = shard.aggregate(recs, a.steps, t_max)
        total_tracked, total_in, value = agg["tracked_per_step"], agg["features_in_per_step"], agg["value"]
        bpf = lk_bytes_per_feature(levels)
        k3_avg_s = (k3_ms / max(k3_n, 1)) * 1e-3
        feats_per_launch = NP * NF
        achieved = bpf * feats_per_launch / k3_avg_s / 1e9 if k3_avg_s > 0 else 0.0
        lv_bytes, w_, h_ = 0, W, H                  # bytes of one image's pyramid
        for _ in range(levels):
            lv_bytes += w_ * h_
            w_, h_ = (w_ + 1) // 2, (h_ + 1) // 2
        traffic, traffic_src, pmc = measured_traffic("pyrlk_track_w21", feats_per_launch,
                                                     suffix="_exact_sums" if a.lk_sums == K.LK_SUMS_EXACT else "")
        # What the kernel is really limited by: vector-ALU issue.  The ceiling is MEASURED (tools/probe/issue_rate.hip):
        # on gfx950 a SIMD issues v_fma_f32 / v_add_f32 every 2.2 cycles but integer, permute, dot2-i16, DPP,
        # convert and packed-f32 instructions every 4.2; the LK step's own mix costs 3.9 cycles per
        # wave-instruction.  Achieved = 1024 SIMDs x shader clock x kernel time / SQ_INSTS_VALU, with the
        # instruction count and the kernel's clock (GRBM_GUI_ACTIVE / 8 / duration under the counters) from
        # the same PMC summary as `traffic`.
        valu = None
        ceil = issue_ceiling()
        if pmc and pmc.get("SQ_INSTS_VALU") and k3_avg_s > 0:
            clock = pmc.get("clock_ghz") or (ceil or {}).get("clock_ghz_in_probe") or 2.4
            # shader cycles of one launch: GRBM_GUI_ACTIVE is summed over the 8 XCDs; without it, clock x live duration
            launch_cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0 if pmc.get("GRBM_GUI_ACTIVE") else clock * 1e9 * k3_avg_s
            cyc_per_inst = 1024.0 * launch_cycles / pmc["SQ_INSTS_VALU"]
            valu = {"insts_per_feature": pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"],
                    "vmem_insts_per_feature": pmc.get("SQ_INSTS_VMEM", 0.0) / pmc["SQ_WAVES"],
                    "shader_clock_ghz": clock, "clock_source": "GRBM_GUI_ACTIVE" if pmc.get("clock_ghz") else "probe",
                    "launch_cycles": launch_cycles,
                    "cycles_per_wave_inst_per_simd": cyc_per_inst,
                    "ceiling": ceil,
                    "issue_frac": (ceil["cycles_per_inst_lk_mix"] / cyc_per_inst) if ceil else None,
                    "note": "1024 SIMDs x shader cycles of a launch / SQ_INSTS_VALU, against the measured issue cost of "
                            "the kernel's instruction mix (a fraction of 1 = every issue slot taken); counters from "
                            + traffic_src}
        out = {
            "metric": "tracked features/sec @1241x376, 4-lvl 21x21 PyrLK, 1-8 GPU; % of HBM peak",
            "value": value,
            "unit": "tracked features/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": t_max / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE[a.lk_sums],
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo; not a measurement)" if rehearsal else ""),
            "config": {
                "workload": "%s%dx%d gray frame-pair stream, %d Shi-Tomasi corners/frame, "
                            "maxLevel=%d (%d pyramid levels), 21x21 window, 30 it / eps 0.01" % (
                                "BASELINE configs[1]: " if (W, H, NF) == (1241, 376, 2000) else
                                "BASELINE configs[2] shape: " if (W, H, NF) == (1920, 1080, 8000) else
                                "BASELINE configs[4] shape: " if (W, H, NF) == (3840, 2160, 16384) else "",
                                W, H, NF, a.max_level, levels),
                "pairs_per_step": NP,
                "lk_window_sums": "tree" if a.lk_sums == K.LK_SUMS_TREE else "exact",
                "prewarm_steps": PREWARM_STEPS,
                "sequences_per_step": Q,
                "pipeline_contexts": len(ctxs),
                "features_per_pair": NF,
                "min_distance_rank0": min_dist_used,
                "per_gpu_sequence": "corridor seeds 0x%X + 0x1000 q + rank, q < %d" % (a.seed, Q),
                "features_in_per_step": total_in,
                "tracked_per_step": total_tracked,
                "pairs_per_s": agg["pairs_per_s"],
            },
            "roofline": {
                # the contract's figure: algorithmic bytes of the gather model over the launch time against HBM
                # peak.  The kernel itself is bound by vector-instruction issue (`limiter`, `valu`), its real
                # HBM traffic is `hbm_measured_GBps`.
                "bound": "hbm",
                "limiter": "valu_issue",
                "kernel": "pyrlk_track_w21",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "hbm_measured_GBps": (traffic / k3_avg_s / 1e9) if (traffic and k3_avg_s > 0) else None,
                "bytes_model_per_feature": bpf,
                # compulsory-unique traffic of one pair: both pyramids read once + point I/O
                "bytes_rocprof_per_feature": (traffic / feats_per_launch) if traffic else None,
                "valu": valu,
                "lds_bank_conflict_rate": 0.0,   # the kernel has no LDS instruction (SQ_INSTS_LDS = 0 in profiles/)
                "bytes_floor_per_feature": (2 * lv_bytes + 29 * NF) / NF,
                "avg_launch_ms": k3_avg_s * 1e3,
                "launches": int(k3_n),
                "features_per_launch": feats_per_launch,
            },
            "detect_frames": detect,
            "kernels": {
                "pyr_down_u8": {"avg_ms_per_step": k1_ms / max(k1_n, 1), "images_per_step": NI,
                                "algorithmic_bytes_per_image": pyr_bytes,
                                "achieved_GBps": pyr_bytes * NI / max(k1_ms / max(k1_n, 1) * 1e-3, 1e-12) / 1e9},
                "scharr_planes": {"avg_ms_per_step": k2_ms / max(k2_n, 1), "images_per_step": NI,
                                  "algorithmic_bytes_per_image": 5 * lv_bytes,   # 1 B read + 4 B written per pixel
                                  "achieved_GBps": 5 * lv_bytes * NI / max(k2_ms / max(k2_n, 1) * 1e-3, 1e-12) / 1e9},
                "pyrlk_track_w21": {"avg_ms_per_step": k3_ms / max(k3_n, 1)},
            },
            "h2d_inclusive": {"streamed": streamed,
                              "value": total_tracked / world / (t_max / a.steps + t_h2d),
                              "value_pinned_source": total_tracked / world / (t_max / a.steps + t_h2d_pinned),
                              "upload_ms": {"pageable_source": t_h2d * 1e3, "pinned_source": t_h2d_pinned * 1e3},
                              "note": "rank-0 rate if the %d frames of a step were uploaded from host memory "
                                      "(pinned staging ring) every step (%.1f ms); never the headline value" % (NI, t_h2d * 1e3)},
            "per_rank": [{"features_in": r[0], "tracked": r[1], "k3_ms": r[3], "k1_ms": r[5], "elapsed_s": r[7]}
                         for r in recs.tolist()],
        }
        if streamed is not None:
            streamed["ms_per_step"]["compute_alone"] = t_max / a.steps * 1e3
        if soak is not None:
            out["soak"] = soak
        if ml3 is not None:
            out["max_level_3"] = ml3
        if other_sums is not None:
            out["other_window_sums"] = other_sums
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(frames[:P + 1], pts[:P], a.max_level, a.cpu_seconds)
        print(json.dumps(out))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
