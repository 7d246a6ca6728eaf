"""ctypes binding of the CPU oracle (oracle/libktlvo_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, from __graft_entry__.smoke() and from the
cpu_baseline leg of bench.py; nothing in slam_amd/ may import this package.

PARITY UNPINNED: the oracle restates OpenCV 3.x algorithms that are neither vendored in the
reference nor installed here; see oracle/ktlvo_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libktlvo_oracle.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


_lib = _load()

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
        ("x_subregions", C.c_int), ("y_subregions", C.c_int), ("max_per_subregion", C.c_int),
        ("x_boundary", C.c_int), ("y_boundary", C.c_int),
        ("max_track_dis", C.c_double),
        ("win", C.c_int), ("max_level", C.c_int), ("max_count", C.c_int),
        ("epsilon", C.c_double), ("min_eig_thr", C.c_double),
        ("quality", C.c_double), ("min_dist", C.c_double),
        ("use_harris", C.c_int),
        ("harris_k", C.c_double),
        ("anms_keep", C.c_int),
        ("keyframe_min_tracked", C.c_int),
        ("ransac_iters", C.c_int),
        ("ransac_reproj", C.c_float),
        ("ransac_conf", C.c_double),
        ("nthreads", C.c_int),
    ]


SUMS_SERIAL, SUMS_FOUR_LANES, SUMS_PAIRWISE, SUMS_EXACT = 0, 1, 2, 3   # orc_set_lk_sum_mode

IMAGE_POINT = np.dtype([("u", "<f4"), ("v", "<f4"), ("ind", "<i4")])
DEPTH_POINT = np.dtype([("u", "<f4"), ("v", "<f4"), ("depth", "<f4"), ("label", "<i4"), ("ind", "<i4"),
                        ("newKeyFrame", "<i4")])

_lib.orc_pyr_down.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
_lib.orc_pyr_down.restype = None
_lib.orc_scharr.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _i16p]
_lib.orc_scharr.restype = None
_lib.orc_pyramid_create.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
_lib.orc_pyramid_create.restype = C.c_void_p
_lib.orc_pyramid_destroy.argtypes = [C.c_void_p]
_lib.orc_pyramid_destroy.restype = None
_lib.orc_pyramid_levels.argtypes = [C.c_void_p]
_lib.orc_pyramid_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
_lib.orc_pyramid_level_copy.argtypes = [C.c_void_p, C.c_int, _u8p]
_lib.orc_pyramid_deriv_copy.argtypes = [C.c_void_p, C.c_int, _i16p]
_lib.orc_pyrlk.argtypes = [C.c_void_p, C.c_void_p, _f32p, C.c_int, _f32p, _u8p, _f32p, C.c_int, C.c_int, C.c_int,
                           C.c_double, C.c_double, C.c_int]
_lib.orc_set_iter_log.argtypes = [C.c_void_p, C.c_int]
_lib.orc_set_iter_log.restype = None
_lib.orc_corner_response.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_double, _f32p]
_lib.orc_good_features.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_double, C.c_double, C.c_int, C.c_double, _f32p, _f32p, C.c_int]
_lib.orc_anms.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _i32p]
_lib.orc_fb_gate.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                             _u8p]
_lib.orc_circular_gate.argtypes = [_f32p, _f32p, C.c_int, _u8p]
_lib.orc_depth_gate.argtypes = [_f32p, _f32p, C.c_int, C.c_double, C.c_double, _u8p, _f32p]
_lib.orc_rodrigues_v2m.argtypes = [_f64p, _f64p]
_lib.orc_rodrigues_v2m.restype = None
_lib.orc_rodrigues_m2v.argtypes = [_f64p, _f64p]
_lib.orc_rodrigues_m2v.restype = None
_lib.orc_p3p.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p]
_lib.orc_solve_p3p4.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p]
_lib.orc_gao_quartic.argtypes = [C.c_double] * 5 + [_f64p]
_lib.orc_gao_quartic.restype = None
_lib.orc_roots4.argtypes = [_f64p, _f64p]
_lib.orc_set_lk_sum_mode.argtypes = [C.c_int]
_lib.orc_set_lk_sum_mode.restype = None
_lib.orc_set_threads.argtypes = [C.c_int]
_lib.orc_set_threads.restype = None
_lib.orc_get_lk_sum_mode.argtypes = []
_lib.orc_get_lk_sum_mode.restype = C.c_int
_lib.orc_p3p_set_polish.argtypes = [C.c_int]
_lib.orc_p3p_set_polish.restype = None
_lib.orc_project_score.argtypes = [_f32p, _f32p, C.c_int, _f64p, _f64p, C.c_float, C.c_void_p, C.c_void_p]
_lib.orc_solve_pnp_ransac.argtypes = [_f32p, _f32p, C.c_int, _f64p, C.c_int, C.c_float, C.c_double, _f64p, _f64p,
                                      C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
_lib.orc_default_config.argtypes = [C.POINTER(Config)]
_lib.orc_default_config.restype = None
_lib.orc_tracker_create.argtypes = [C.POINTER(Config)]
_lib.orc_tracker_create.restype = C.c_void_p
_lib.orc_tracker_destroy.argtypes = [C.c_void_p]
_lib.orc_tracker_destroy.restype = None
_lib.orc_track_stereo_frame.argtypes = [C.c_void_p, _u8p, _u8p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
_lib.orc_estimate_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _f32p, _f32p]


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def pyr_down(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    _lib.orc_pyr_down(img, w, h, w, out, out.shape[1])
    return out


def scharr(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    _lib.orc_scharr(img, w, h, w, out)
    return out


class Pyramid:
    def __init__(self, img, max_level=4, win=21, with_deriv=True):
        img = _img(img)
        h, w = img.shape
        self._p = _lib.orc_pyramid_create(img, w, h, w, max_level, win, 1 if with_deriv else 0)
        if not self._p:
            raise ValueError("orc_pyramid_create failed")
        self.win = win

    def __del__(self):
        if getattr(self, "_p", None):
            _lib.orc_pyramid_destroy(self._p)
            self._p = None

    @property
    def max_level(self):
        return _lib.orc_pyramid_levels(self._p)

    def level(self, l):
        w, h = C.c_int(), C.c_int()
        if _lib.orc_pyramid_level_size(self._p, l, C.byref(w), C.byref(h)) != 0:
            raise IndexError(l)
        out = np.empty((h.value, w.value), np.uint8)
        _lib.orc_pyramid_level_copy(self._p, l, out)
        return out

    def deriv(self, l):
        w, h = C.c_int(), C.c_int()
        if _lib.orc_pyramid_level_size(self._p, l, C.byref(w), C.byref(h)) != 0:
            raise IndexError(l)
        out = np.empty((h.value, w.value, 2), np.int16)
        if _lib.orc_pyramid_deriv_copy(self._p, l, out) != 0:
            raise ValueError("no derivatives")
        return out


def pyrlk(prev, nxt, pts, win=21, max_level=4, max_count=30, epsilon=0.01, min_eig_thr=1e-4, nthreads=1, sums=None):
    """prev/nxt: Pyramid objects (prev with derivatives) or 2-D uint8 arrays.  sums: the rounding of the window
    sums for this call (see set_lk_sum_mode; None = the mode in force)."""
    if sums is not None:
        with lk_sum_mode(sums):
            return pyrlk(prev, nxt, pts, win, max_level, max_count, epsilon, min_eig_thr, nthreads)
    if not isinstance(prev, Pyramid):
        prev = Pyramid(prev, max_level, win, True)
    if not isinstance(nxt, Pyramid):
        nxt = Pyramid(nxt, max_level, win, False)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros((n, 2), np.float32)
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    rc = _lib.orc_pyrlk(prev._p, nxt._p, pts, n, out, status, err, win, max_level, max_count, epsilon, min_eig_thr,
                        nthreads)
    if rc != 0:
        raise RuntimeError("orc_pyrlk rc=%d" % rc)
    return out, status, err


class Baseline:
    """What the cpu_baseline leg of bench.py times: one cv::calcOpticalFlowPyrLK-equivalent per call (both pyramids,
    the first image's Scharr planes, the LK loop) through raw ctypes calls on a library of its own.  That library is
    this same source built for AVX2-class CPUs (libktlvo_oracle_v3.so) when the host has them, else the checker
    build; -ffp-contract=off in both, so the results are the checker's."""

    def __init__(self):
        flags = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("flags"):
                    flags = line
                    break
        except OSError:
            pass
        v3 = os.path.join(_HERE, "libktlvo_oracle_v3.so")
        ok = all((" " + f + " ") in (flags.replace("\n", " ") + " ") for f in ("avx2", "bmi2", "fma", "movbe", "f16c"))
        self.build = "-O3 -march=x86-64-v3" if ok and os.path.exists(v3) else "-O3 -march=x86-64-v2"
        self.lib = C.CDLL(v3) if ok and os.path.exists(v3) else _lib
        L = self.lib
        L.orc_pyramid_create.argtypes = _lib.orc_pyramid_create.argtypes
        L.orc_pyramid_create.restype = C.c_void_p
        L.orc_pyramid_destroy.argtypes = [C.c_void_p]
        L.orc_pyramid_destroy.restype = None
        L.orc_pyrlk.argtypes = _lib.orc_pyrlk.argtypes
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = None
        L.orc_set_lk_sum_mode.argtypes = [C.c_int]
        L.orc_set_lk_sum_mode.restype = None

    def track_pair(self, a, b, pts, max_level=4, win=21, threads=1, sums=SUMS_SERIAL):
        """Returns (next points, status, err) of one pair; every stage cut over `threads` threads."""
        L = self.lib
        a, b = _img(a), _img(b)
        h, w = a.shape
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        out, status, err = np.zeros((n, 2), np.float32), np.zeros(n, np.uint8), np.zeros(n, np.float32)
        L.orc_set_threads(threads)
        L.orc_set_lk_sum_mode(sums)
        try:
            pa = L.orc_pyramid_create(a, w, h, w, max_level, win, 1)
            pb = L.orc_pyramid_create(b, w, h, w, max_level, win, 0)
            try:
                rc = L.orc_pyrlk(pa, pb, pts, n, out, status, err, win, max_level, 30, 0.01, 1e-4, threads)
            finally:
                L.orc_pyramid_destroy(pa)
                L.orc_pyramid_destroy(pb)
        finally:
            L.orc_set_threads(1)
            L.orc_set_lk_sum_mode(SUMS_SERIAL)
        if rc != 0:
            raise RuntimeError("orc_pyrlk rc=%d" % rc)
        return out, status, err


def lk_iterations(prev, nxt, pts, max_level=4, **kw):
    """Iterations per (level, point) of a single-threaded pyrlk call: uint8 [levels, n]."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    log = np.zeros((max_level + 1, pts.shape[0]), np.uint8)
    _lib.orc_set_iter_log(log.ctypes.data, pts.shape[0])
    try:
        out = pyrlk(prev, nxt, pts, max_level=max_level, nthreads=1, **kw)
    finally:
        _lib.orc_set_iter_log(None, 0)
    return log, out


def corner_response(img, roi=None, use_harris=True, k=0.0):
    img = _img(img)
    h, w = img.shape
    rx, ry, rw, rh = roi if roi is not None else (0, 0, w, h)
    out = np.empty((rh, rw), np.float32)
    rc = _lib.orc_corner_response(img, w, h, w, rx, ry, rw, rh, int(use_harris), k, out)
    if rc != 0:
        raise ValueError("bad roi")
    return out


def good_features(img, roi=None, max_corners=100, quality=0.01, min_dist=10.0, use_harris=True, k=0.0):
    img = _img(img)
    h, w = img.shape
    rx, ry, rw, rh = roi if roi is not None else (0, 0, w, h)
    cap = max_corners if max_corners > 0 else rw * rh
    xy = np.zeros((cap, 2), np.float32)
    resp = np.zeros(cap, np.float32)
    n = _lib.orc_good_features(img, w, h, w, rx, ry, rw, rh, max_corners, quality, min_dist, int(use_harris), k, xy,
                               resp, cap)
    if n < 0:
        raise ValueError("bad roi")
    return xy[:n].copy(), resp[:n].copy()


def anms(xy, resp, num_to_keep):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2).copy()
    resp = np.ascontiguousarray(resp, np.float32).copy()
    n = resp.shape[0]
    perm = np.zeros(max(n, 1), np.int32)
    m = _lib.orc_anms(xy, resp, n, num_to_keep, perm)
    return xy[:m], resp[:m], perm[:m]


def fb_gate(f1, f2, f1chk, img_w, img_h, x_boundary=30, y_boundary=30, max_track_dis=100.0, y_threshold=1):
    f1 = np.ascontiguousarray(f1, np.float32).reshape(-1, 2)
    f2 = np.ascontiguousarray(f2, np.float32).reshape(-1, 2)
    f1chk = np.ascontiguousarray(f1chk, np.float32).reshape(-1, 2)
    keep = np.zeros(f1.shape[0], np.uint8)
    _lib.orc_fb_gate(f1, f2, f1chk, f1.shape[0], img_w, img_h, x_boundary, y_boundary, max_track_dis, y_threshold,
                     keep)
    return keep


def circular_gate(pcr, clr):
    pcr = np.ascontiguousarray(pcr, np.float32).reshape(-1, 2)
    clr = np.ascontiguousarray(clr, np.float32).reshape(-1, 2)
    keep = np.zeros(pcr.shape[0], np.uint8)
    _lib.orc_circular_gate(pcr, clr, pcr.shape[0], keep)
    return keep


def depth_gate(ll, lr, bf, fx):
    ll = np.ascontiguousarray(ll, np.float32).reshape(-1, 2)
    lr = np.ascontiguousarray(lr, np.float32).reshape(-1, 2)
    keep = np.zeros(ll.shape[0], np.uint8)
    depth = np.zeros(ll.shape[0], np.float32)
    _lib.orc_depth_gate(ll, lr, ll.shape[0], bf, fx, keep, depth)
    return keep, depth


def rodrigues_v2m(r):
    R = np.zeros(9, np.float64)
    _lib.orc_rodrigues_v2m(np.ascontiguousarray(r, np.float64), R)
    return R.reshape(3, 3)


def rodrigues_m2v(R):
    r = np.zeros(3, np.float64)
    _lib.orc_rodrigues_m2v(np.ascontiguousarray(R, np.float64).reshape(9), r)
    return r


def p3p(X, uv, K):
    R = np.zeros(36, np.float64)
    t = np.zeros(12, np.float64)
    n = _lib.orc_p3p(np.ascontiguousarray(X, np.float64).reshape(9), np.ascontiguousarray(uv, np.float64).reshape(6),
                     np.ascontiguousarray(K, np.float64), R, t)
    return R.reshape(4, 3, 3)[:n], t.reshape(4, 3)[:n]


def solve_p3p4(X, uv, K):
    R = np.zeros(9, np.float64)
    t = np.zeros(3, np.float64)
    ok = _lib.orc_solve_p3p4(np.ascontiguousarray(X, np.float64).reshape(12),
                             np.ascontiguousarray(uv, np.float64).reshape(8), np.ascontiguousarray(K, np.float64), R,
                             t)
    return bool(ok), R.reshape(3, 3), t


def gao_quartic(a, b, p, q, r):
    c = np.zeros(5, np.float64)
    _lib.orc_gao_quartic(a, b, p, q, r, c)
    return c


def set_lk_sum_mode(mode):
    """Rounding of the LK window sums: 0 serial float accumulator (OpenCV's scalar path; default), 1 four float
    accumulators, 2 pairwise float sum, 3 exact integer total rounded to float once (order-independent)."""
    _lib.orc_set_lk_sum_mode(int(mode))


def set_threads(n):
    """Threads the image-wide stages (pyramid levels, borders, Scharr) are cut over; results do not depend on it."""
    _lib.orc_set_threads(int(n))


def get_lk_sum_mode():
    return int(_lib.orc_get_lk_sum_mode())


class lk_sum_mode:
    """with lk_sum_mode(m): ... runs the block under sum mode m and restores the previous one."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_lk_sum_mode()
        set_lk_sum_mode(self.mode)

    def __exit__(self, *exc):
        set_lk_sum_mode(self.prev)
        return False


def p3p_set_polish(newton_steps):
    _lib.orc_p3p_set_polish(int(newton_steps))


def roots4(coef):
    x = np.zeros(4, np.float64)
    n = _lib.orc_roots4(np.ascontiguousarray(coef, np.float64), x)
    return x[:n]


def project_score(X3, x2, Rt, K, thr_sq):
    X3 = np.ascontiguousarray(X3, np.float32).reshape(-1, 3)
    x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
    n = X3.shape[0]
    mask = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    cnt = _lib.orc_project_score(X3, x2, n, np.ascontiguousarray(Rt, np.float64).reshape(12),
                                 np.ascontiguousarray(K, np.float64), np.float32(thr_sq), mask.ctypes.data,
                                 err.ctypes.data)
    return cnt, mask, err


def solve_pnp_ransac(X3, x2, K, max_iters=300, reproj=0.1, confidence=float(np.float32(0.99))):
    X3 = np.ascontiguousarray(X3, np.float32).reshape(-1, 3)
    x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
    n = X3.shape[0]
    rvec = np.zeros(3, np.float64)
    tvec = np.zeros(3, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ninl, iters = C.c_int(), C.c_int()
    log = np.zeros((max(max_iters, 1), 6), np.int32)
    ok = _lib.orc_solve_pnp_ransac(X3, x2, n, np.ascontiguousarray(K, np.float64), max_iters, np.float32(reproj),
                                   confidence, rvec, tvec, mask.ctypes.data, C.byref(ninl), C.byref(iters),
                                   log.ctypes.data)
    return dict(ok=bool(ok), rvec=rvec, tvec=tvec, mask=mask[:n], n_inliers=ninl.value, iters=iters.value,
                log=log[:iters.value])


def default_config(**kw):
    c = Config()
    _lib.orc_default_config(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


class Tracker:
    """Frame-level oracle: stereoImageCallback + pointCloudCallback."""

    def __init__(self, cfg=None, cap=20000):
        self.cfg = cfg or default_config()
        self._t = _lib.orc_tracker_create(C.byref(self.cfg))
        self.cap = cap

    def __del__(self):
        if getattr(self, "_t", None):
            _lib.orc_tracker_destroy(self._t)
            self._t = None

    def track(self, left, right):
        left, right = _img(left), _img(right)
        assert left.shape == (self.cfg.height, self.cfg.width) and right.shape == left.shape
        last = np.zeros(self.cap, DEPTH_POINT)
        cur = np.zeros(self.cap, IMAGE_POINT)
        kf = C.c_int()
        n = _lib.orc_track_stereo_frame(self._t, left, right, last.ctypes.data, cur.ctypes.data, self.cap,
                                        C.byref(kf))
        if n < 0:
            raise RuntimeError("orc_track_stereo_frame rc=%d" % n)
        return last[:n].copy(), cur[:n].copy(), bool(kf.value)

    def estimate_pose(self, last, cur):
        last = np.ascontiguousarray(last, DEPTH_POINT)
        cur = np.ascontiguousarray(cur, IMAGE_POINT)
        Tl = np.zeros(16, np.float32)
        Tg = np.zeros(16, np.float32)
        n = _lib.orc_estimate_pose(self._t, last.ctypes.data, cur.ctypes.data, last.shape[0], Tl, Tg)
        return n, Tl.reshape(4, 4), Tg.reshape(4, 4)
