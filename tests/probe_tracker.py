"""One-off long comparison of the GPU tracker with the oracle's callbacks over several synthetic sequences
(not collected by pytest); run on a GPU box.  usage: probe_tracker.py [frames] [seed ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle as O  # noqa: E402
from slam_amd import lib as K, synth  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seeds = [int(s, 0) for s in sys.argv[2:]] or [0x5EED0011, 0x5EED0012, 0x5EED0013]
W, H = 1242, 375
for seed in seeds:
    scene = synth.CorridorScene(W, H, seed=seed)
    ctx = K.Context(K.default_config(width=W, height=H))
    otr = O.Tracker(O.default_config(nthreads=16))
    same = worst = 0
    dT = 0.0
    for f in range(nf):
        L, R = scene.stereo_pair(f)
        gl, gc, gk = ctx.track_stereo_frame(L, R)
        ol, oc, ok = otr.track(L, R)
        if f == 0:
            continue
        eq = len(gl) == len(ol) and np.array_equal(gl["ind"], ol["ind"]) and gk == ok
        same += int(eq)
        if eq and len(gl):
            worst = max(worst, float(np.abs(gc["u"] - oc["u"]).max()), float(np.abs(gc["v"] - oc["v"]).max()))
        _, _, gT = ctx.estimate_pose(gl, gc)
        _, _, oT = otr.estimate_pose(ol, oc)
        dT = max(dT, float(np.abs(gT[:3, 3] - oT[:3, 3]).max()))
    print("seed %#x: %d of %d frames keep identical feature ids; worst position difference %.4f px; worst "
          "trajectory difference %.4f m" % (seed, same, nf - 1, worst, dT))
    ctx.close()
