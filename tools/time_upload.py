#!/usr/bin/env python3
"""Time of ktlvo_upload_image per 1241x376 frame from pageable and from pinned host memory; run on a
GPU box.  usage: time_upload.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from slam_amd import lib as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W, H = 1241, 376
rng = np.random.default_rng(1)
frames = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(n)]
pinned = [torch.from_numpy(f).pin_memory() for f in frames]
ctx = K.Context(K.default_config(width=W, height=H, num_slots=n))


def run(src):
    t = time.perf_counter()
    for s, f in enumerate(src):
        ctx.upload_image(s, f)
    t1 = time.perf_counter()
    ctx.sync()
    return (t1 - t) * 1e6 / n, (time.perf_counter() - t) * 1e6 / n


for name, src in (("pageable", frames), ("pinned", [p.numpy() for p in pinned])):
    run(src)
    best = min(run(src) for _ in range(5))
    print("%s: %.1f us per frame to enqueue, %.1f us per frame with the final sync" % (name, best[0], best[1]))
