#!/usr/bin/env python3
"""How long the LK launch of the benchmark's shape (254 pairs x 2000 features, one wave per feature) takes when every
feature lies far outside its image, so that each wave leaves after a few dozen instructions: the cost of dispatching
508,000 one-wave workgroups, against the 2.4 ms of the real launch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import lib as K  # noqa: E402

W, H, NI, NF = 1241, 376, 256, 2000
ctx = K.Context(K.default_config(width=W, height=H, num_slots=NI, max_level=4, max_points=NF))
rng = np.random.default_rng(0)
ctx.upload_images(list(range(NI)), rng.integers(0, 256, (NI, H, W), dtype=np.uint8))
ctx.build_pyramids(list(range(NI)))
prev = [q * 128 + i for q in range(2) for i in range(127)]
nxt = [s + 1 for s in prev]
NP = len(prev)
lk = K.lk_params(21, 4)
o = (torch.zeros((NP, NF, 2), device="cuda"), torch.zeros((NP, NF), dtype=torch.uint8, device="cuda"), torch.zeros((NP, NF), device="cuda"))
for name, val in (("far outside", -1.0e6), ("inside (noise frames)", None)):
    if val is None:
        pts = np.stack([rng.uniform(20, W - 20, NP * NF), rng.uniform(20, H - 20, NP * NF)], 1).astype(np.float32)
    else:
        pts = np.full((NP * NF, 2), val, np.float32)
    d_prev = torch.from_numpy(pts).cuda()
    for _ in range(5):
        ctx.pyrlk_batch_dev(prev, nxt, d_prev.data_ptr(), NF, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), lk)
    ctx.sync()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        ctx.pyrlk_batch_dev(prev, nxt, d_prev.data_ptr(), NF, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), lk)
    ctx.sync()
    print("%-22s %.4f ms per launch (%d workgroups, %d tracked)" % (name, (time.perf_counter() - t) / n * 1e3, NP * NF, int(o[1].sum())))
