#!/usr/bin/env python3
"""Per-frame time of the call-by-call C++ node (tests/cpp/frontend_node) against the one-call frame
path, for each library variant given (default: the built one); run on a GPU box.
usage: time_node.py [frames] [variant.so ...]"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from slam_amd import synth  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variants = sys.argv[2:]
W, H = 1242, 375
scene = synth.CorridorScene(W, H, seed=0x5EED0001)
tmp = tempfile.mkdtemp()
raw = os.path.join(tmp, "frames.bin")
with open(raw, "wb") as fh:
    for f in range(nf):
        L, R = scene.stereo_pair(f)
        fh.write(np.ascontiguousarray(L).tobytes())
        fh.write(np.ascontiguousarray(R).tobytes())
node = os.path.join(ROOT, "tests", "cpp", "frontend_node")
lib = os.path.join(ROOT, "slam_amd", "libktlvo_hip.so")
keep = os.path.join(tmp, "keep.so")
shutil.copy(lib, keep)
try:
    for v in variants or [keep]:
        if v != keep:
            shutil.copy(v, lib)
        for _ in range(2):
            r = subprocess.run([node, raw, str(W), str(H), str(nf), os.path.join(tmp, "out.bin"), "--no-oracle"],
                               capture_output=True, text=True, timeout=120)
            print(os.path.basename(v), r.returncode, " | ".join(l for l in r.stdout.splitlines() if l.startswith("per")))
            if os.environ.get("KTLVO_NODE_VERBOSE"):
                print(r.stderr)
finally:
    shutil.copy(keep, lib)
