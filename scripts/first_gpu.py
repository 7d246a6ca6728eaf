"""First GPU check: pyramid + LK parity against the oracle (temporary bring-up script)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from slam_amd import synth
import importlib.util
src = open(os.path.join(os.path.dirname(__file__), "..", "slam_amd", "lib.py")).read()
src = src.replace("fn = getattr(lib, name)  # AttributeError here means the library is stale",
                  "fn = getattr(lib, name, None)\n        if fn is None: continue")
mod = type(sys)("klib"); exec(compile(src.replace("os.path.abspath(__file__)", repr(os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "slam_amd", "lib.py")))), "klib", "exec"), mod.__dict__)
K = mod
W, H = 1241, 376
cfg = K.default_config(width=W, height=H)
ctx = K.Context(cfg)
a, b = synth.texture_pair(W, H, 5, 3.3, -1.7)
ctx.upload_image(0, a); ctx.upload_image(1, b)
ctx.build_pyramids([0, 1])
pa, pb = O.Pyramid(a), O.Pyramid(b)
for l in range(5):
    g = ctx.download_level(0, l); o = pa.level(l)
    print("level", l, g.shape, "pyr equal:", np.array_equal(g, o))
    s = ctx.download_scharr(0, l, o.shape[1], o.shape[0]); so = pa.deriv(l)
    print("   scharr equal:", np.array_equal(s, so))
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(0, W, 4000), rng.uniform(0, H, 4000)], 1).astype(np.float32)
t = time.time(); go, gs, ge = ctx.pyrlk(0, 1, pts); print("gpu lk", time.time() - t)
t = time.time(); oo, os_, oe = O.pyrlk(pa, pb, pts, nthreads=8); print("cpu lk", time.time() - t)
d = np.abs(go - oo).max(1)
both = (gs == 1) & (os_ == 1)
print("status equal frac", (gs == os_).mean(), "tracked", gs.mean(), os_.mean())
print("max diff (both ok)", d[both].max(), "p99", np.percentile(d[both], 99), "frac>0.05", (d[both] > 0.05).mean())
print("err diff", np.abs(ge - oe)[both].max())
print("truth err med", np.median(np.abs(go[both] - pts[both] - np.array([3.3, -1.7], np.float32))))
