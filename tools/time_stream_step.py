#!/usr/bin/env python3
"""The frame-pair stream's whole step (detect on the first frames, build all pyramids, track the detected corners) in
its arrangements, on the benchmark's shape with fewer pairs if asked; run on a GPU box, alone for wall times or under
`rocprofv3 --kernel-trace` for the kernels' timeline (tools/trace_timeline.py reads its CSV).
usage: time_stream_step.py [pairs_per_sequence=127] [reps=10] [arrangement: turn|side|one|both|pair] [sums: exact|tree, default the library's]
(one: ktlvo_detect_build_track_dev; both: the three single-context arrangements in turn)
(pair: the side-by-side step on two contexts in turn, each with its own frames and arrays)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import lib as K, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 127
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "both"
W, H, NF, Q = 1241, 376, 2000, 2
frames = []
for q in range(Q):
    frames += synth.render_corridor_frames(W, H, 0x5EED0002 + 0x1000 * q, P + 1, workers=8)
NI, NP = Q * (P + 1), Q * P
scene = synth.CorridorScene(W, H, seed=0x5EED0002)
ctx = K.Context(K.default_config(width=W, height=H, num_slots=NI, max_points=NF, cx=scene.cx, cy=scene.cy))
slots = list(range(NI))
prev = [q * (P + 1) + i for q in range(Q) for i in range(P)]
nxt = [s + 1 for s in prev]
ctx.upload_images(slots, np.stack(frames))
gp = K.gftt_params(NF, 0.01, 10.0, 3, 0, 0.0)
lk = K.lk_params(sums={"exact": K.LK_SUMS_EXACT, "tree": K.LK_SUMS_TREE}.get(sys.argv[4] if len(sys.argv) > 4 else "", K.LK_SUMS_DEFAULT))
dev = torch.device("cuda", 0)
d_xy = torch.zeros((NP, NF, 2), device=dev); d_r = torch.zeros((NP, NF), device=dev)
d_c = torch.zeros(NP, dtype=torch.int32, device=dev)
d_n = torch.zeros((NP, NF, 2), device=dev); d_s = torch.zeros((NP, NF), dtype=torch.uint8, device=dev)
d_e = torch.zeros((NP, NF), device=dev)
torch.cuda.synchronize()


def step(side):
    if side == "one":
        ctx.detect_build_track_dev(prev, nxt, gp, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), NF, slots, d_n.data_ptr(),
                                   d_s.data_ptr(), d_e.data_ptr(), lk)
        return
    if side:
        ctx.detect_build_frames_dev(prev, gp, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), NF, slots)
    else:
        ctx.detect_gftt_frames_dev(prev, gp, d_xy.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), NF)
        ctx.build_pyramids(slots)
    ctx.pyrlk_batch_dev(prev, nxt, d_xy.data_ptr(), NF, d_n.data_ptr(), d_s.data_ptr(), d_e.data_ptr(), lk)


if which == "pair":
    ctx2 = K.Context(K.default_config(width=W, height=H, num_slots=NI, max_points=NF, cx=scene.cx, cy=scene.cy))
    ctx2.upload_images(slots, np.stack(frames))
    ctx2.sync()
    bufs2 = [torch.zeros_like(t) for t in (d_xy, d_r, d_c, d_n, d_s, d_e)]
    both = [(ctx, (d_xy, d_r, d_c, d_n, d_s, d_e)), (ctx2, bufs2)]

    def pair_step(k):
        c, (xy, r, cn, n_, s_, e_) = both[k & 1]
        c.detect_build_frames_dev(prev, gp, xy.data_ptr(), r.data_ptr(), cn.data_ptr(), NF, slots)
        c.pyrlk_batch_dev(prev, nxt, xy.data_ptr(), NF, n_.data_ptr(), s_.data_ptr(), e_.data_ptr(), lk)

    for k in range(6):
        pair_step(k)
    ctx.sync(); ctx2.sync()
    t = time.perf_counter()
    for k in range(2 * reps):
        pair_step(k)
    ctx.sync(); ctx2.sync()
    print("two contexts in turn: %.4f ms per step" % ((time.perf_counter() - t) / (2 * reps) * 1e3), flush=True)
    ctx2.close()
    ctx.close()
    sys.exit(0)
for side in ([False, True, "one"] if which == "both" else ["one"] if which == "one" else [which == "side"]):
    for _ in range(5):
        step(side)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        step(side)
    ctx.sync()
    dt = (time.perf_counter() - t) / reps
    print("%s: %.4f ms per step" % ("one call" if side == "one" else "side by side" if side else "in turn", dt * 1e3), flush=True)
ctx.close()
