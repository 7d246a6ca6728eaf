#!/usr/bin/env python3
"""Finds the feature on which the GPU frame path and the oracle's callback disagree (one passes a gate, the
other does not) and prints every gate quantity of that feature from both sides; run on a GPU box.
usage: probe_gate_flip.py W H frame"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle as O  # noqa: E402
from slam_amd import lib as K, synth  # noqa: E402

W, H, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = synth.CorridorScene(W, H, seed=0x5EED0001)
cam = dict(width=W, height=H, cx=scene.cx, cy=scene.cy)
ctx = K.Context(K.default_config(**cam))
otr = O.Tracker(O.default_config(nthreads=16, **cam))
prev = None
for f in range(F + 1):
    L, R = scene.stereo_pair(f)
    if f == F:
        break
    gl, gc, _ = ctx.track_stereo_frame(L, R)
    ol, oc, _ = otr.track(L, R)
    assert len(gl) == len(ol)
    prev = (L, R, gc, oc)
LL, LR, gcur, ocur = prev
CL, CR = L, R
print("start lists equal:", np.array_equal(gcur["u"], ocur["u"]) and np.array_equal(gcur["v"], ocur["v"]),
      "max diff", float(max(np.abs(gcur["u"] - ocur["u"]).max(), np.abs(gcur["v"] - ocur["v"]).max())))
# stage by stage from the SAME start list (the oracle's), on both sides
A = np.stack([ocur["u"], ocur["v"]], 1).astype(np.float32)
c2 = K.Context(K.default_config(num_slots=4, **cam))
for s, im in enumerate((LL, LR, CL, CR)):
    c2.upload_image(s, im)
c2.build_pyramids([0, 1, 2, 3])
P = [O.Pyramid(im) for im in (LL, LR, CL, CR)]


def both(a, b, pts):
    g = c2.pyrlk(a, b, pts)[0]
    o = O.pyrlk(P[a], P[b], pts, nthreads=16)[0]
    return g, o


gB, oB = both(0, 1, A)          # LL -> LR
gC, oC = both(0, 2, A)          # LL -> CL
gD, oD = both(1, 3, oB)         # LR -> CR (from the oracle's B on both sides, to isolate one call)
gE, oE = both(2, 3, oC)         # CL -> CR
for name, g, o in (("LL->LR", gB, oB), ("LL->CL", gC, oC), ("LR->CR", gD, oD), ("CL->CR", gE, oE)):
    print("%s max |gpu - oracle| %.5f" % (name, float(np.abs(g - o).max())))
circ_g = (gD[:, 0] - gE[:, 0] < 2) & (gD[:, 1] - gE[:, 1] < 2)
circ_o = (oD[:, 0] - oE[:, 0] < 2) & (oD[:, 1] - oE[:, 1] < 2)
print("circular gate flips:", np.flatnonzero(circ_g != circ_o))
for i in np.flatnonzero(circ_g != circ_o):
    print("  feature", i, "dx", gD[i, 0] - gE[i, 0], oD[i, 0] - oE[i, 0], "dy", gD[i, 1] - gE[i, 1], oD[i, 1] - oE[i, 1])
# forward-backward gates
for (a, b, fw_g, fw_o, yt, nm) in ((0, 1, gB, oB, 1, "stereo"), (0, 2, gC, oC, 5, "temporal")):
    gbk = c2.pyrlk(b, a, fw_o)[0]
    obk = O.pyrlk(P[b], P[a], fw_o, nthreads=16)[0]
    print("%s backward max |gpu - oracle| %.5f" % (nm, float(np.abs(gbk - obk).max())))
    kg = O.fb_gate(A, fw_o, gbk, W, H, 30, 30, 100.0, yt)
    ko = O.fb_gate(A, fw_o, obk, W, H, 30, 30, 100.0, yt)
    for i in np.flatnonzero(kg != ko):
        print("  %s gate flips on feature %d: f1 - chk = gpu (%.6f, %.6f) oracle (%.6f, %.6f), thresholds (5, %d)" % (
            nm, i, A[i, 0] - gbk[i, 0], A[i, 1] - gbk[i, 1], A[i, 0] - obk[i, 0], A[i, 1] - obk[i, 1], yt))
# the calls' worst features: are they also the ones on which the oracle moves when only its summation order changes?
for name, a, b, pts, g, o in (("LL->CL", 0, 2, A, gC, oC), ("CL->CR", 2, 3, oC, gE, oE)):
    O.set_lk_sum_mode(2)
    o2 = O.pyrlk(P[a], P[b], pts, nthreads=16)[0]
    O.set_lk_sum_mode(1)
    o1 = O.pyrlk(P[a], P[b], pts, nthreads=16)[0]
    O.set_lk_sum_mode(0)
    it, _ = O.lk_iterations(P[a], P[b], pts, max_level=4)
    dg = np.abs(g - o).max(1)
    d2 = np.abs(o2 - o).max(1)
    d1 = np.abs(o1 - o).max(1)
    print("%s: points beyond 0.01 px: gpu %d, oracle pairwise %d, oracle 4-lane %d; max %.4f / %.4f / %.4f" % (
        name, (dg > 0.01).sum(), (d2 > 0.01).sum(), (d1 > 0.01).sum(), dg.max(), d2.max(), d1.max()))
    for i in np.argsort(-dg)[:6]:
        print("   feature %4d at %s: gpu-oracle %.4f, pairwise-oracle %.4f, 4lane-oracle %.4f, iterations per level %s" % (
            i, pts[i], dg[i], d2[i], d1[i], it[:, i]))
