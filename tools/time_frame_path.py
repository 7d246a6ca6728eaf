#!/usr/bin/env python3
"""Wall-clock split of the per-frame calls (ktlvo_track_stereo_frame / ktlvo_estimate_pose) on a
synthetic stereo corridor; run on a GPU box.  usage: time_frame_path.py [frames] [device_p3p] [pinned]
(pinned = 1: the frames are handed over in pinned host memory, which the library reads by DMA directly)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import lib as K, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dp3p = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = 1242, 375
scene = synth.CorridorScene(W, H, seed=0x5EED0010)
frames = [scene.stereo_pair(f) for f in range(n)]
if len(sys.argv) > 3 and int(sys.argv[3]):
    import torch
    keep = [(torch.from_numpy(np.ascontiguousarray(L)).pin_memory(), torch.from_numpy(np.ascontiguousarray(R)).pin_memory())
            for L, R in frames]
    frames = [(L.numpy(), R.numpy()) for L, R in keep]
ctx = K.Context(K.default_config(width=W, height=H, device_p3p=dp3p))
tt, tp, kfs, npts = [], [], [], []
for f, (L, R) in enumerate(frames):
    t0 = time.perf_counter()
    last, cur, kf = ctx.track_stereo_frame(L, R)
    t1 = time.perf_counter()
    if f:
        ctx.estimate_pose(last, cur)
    t2 = time.perf_counter()
    tt.append(t1 - t0); tp.append(t2 - t1); kfs.append(int(kf)); npts.append(len(last))
tt, tp, kfs = np.array(tt[2:]) * 1e3, np.array(tp[2:]) * 1e3, np.array(kfs[2:])
print("frames %d  key frames %d  points/frame %.0f" % (len(tt), kfs.sum(), np.mean(npts[2:])))
print("track: median %.3f ms (key frames %.3f, others %.3f)   pose: median %.3f ms" % (
    np.median(tt), np.median(tt[kfs == 1]) if kfs.any() else float("nan"), np.median(tt[kfs == 0]), np.median(tp)))
