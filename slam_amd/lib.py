"""ctypes binding of slam_amd/libktlvo_hip.so (the C ABI declared in include/ktlvo.h).

This is harness plumbing for tests and bench.py; the product is the shared library and the C++
host code above it.  There is no CPU fallback: if the library is missing or cannot be loaded
the import of this module raises.
"""
import ctypes as C
import os

import numpy as np

# PyTorch-ROCm ships its own libamdhip64.so.7.  A process must hold exactly one HIP runtime, and
# the loader only shares it when torch's copy is mapped first (our library then binds to the
# already-loaded soname).  Harnesses that also use torch tensors or torch.distributed therefore
# need torch imported before this library is opened; doing it here makes the order automatic.
try:
    import torch  # noqa: F401
except ImportError:  # a torch-free process simply uses the system ROCm runtime
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libktlvo_hip.so")

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_CAPACITY, ERR_STATE, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
K_PYRDOWN, K_PYRLK, K_CORNER, K_RANSAC, K_SCHARR = 0, 1, 2, 3, 4
LK_SUMS_DEFAULT, LK_SUMS_TREE, LK_SUMS_EXACT = 0, 1, 2


class KtlvoError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s: %s (%d) %s" % (where, strerror(status), status, detail))


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("bf", C.c_double),
        ("num_slots", C.c_int), ("max_level", C.c_int), ("win", C.c_int), ("max_points", C.c_int),
        ("max_batch_pairs", C.c_int),
        ("x_subregions", C.c_int), ("y_subregions", C.c_int), ("max_per_subregion", C.c_int),
        ("x_boundary", C.c_int), ("y_boundary", C.c_int),
        ("max_track_dis", C.c_double),
        ("anms_keep", C.c_int), ("keyframe_min_tracked", C.c_int),
        ("ransac_iters", C.c_int), ("ransac_reproj", C.c_float), ("ransac_conf", C.c_double),
        ("max_hypotheses", C.c_int),
        ("device_p3p", C.c_int),
        ("lk_sums", C.c_int),
        ("p3p_polish", C.c_int),
    ]


class LkParams(C.Structure):
    _fields_ = [("win", C.c_int), ("max_level", C.c_int), ("max_count", C.c_int), ("epsilon", C.c_double),
                ("min_eig_thr", C.c_double), ("sums", C.c_int)]


class GfttParams(C.Structure):
    _fields_ = [("max_corners", C.c_int), ("quality", C.c_double), ("min_dist", C.c_double), ("block_size", C.c_int),
                ("use_harris", C.c_int), ("harris_k", C.c_double)]


class Roi(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class RosHeader(C.Structure):
    _fields_ = [("seq", C.c_uint32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32), ("frame_id", C.c_char_p)]


IMAGE_POINT = np.dtype([("u", "<f4"), ("v", "<f4"), ("ind", "<i4")])
DEPTH_POINT = np.dtype([("u", "<f4"), ("v", "<f4"), ("depth", "<f4"), ("label", "<i4"), ("ind", "<i4"),
                        ("newKeyFrame", "<i4")])

# every symbol include/ktlvo.h declares: name -> (restype, argtypes)
_vp, _i, _d, _f = C.c_void_p, C.c_int, C.c_double, C.c_float
_ip = C.POINTER(C.c_int)
SIGNATURES = {
    "ktlvo_default_config": (_i, [C.POINTER(Config)]),
    "ktlvo_create": (_vp, [C.POINTER(Config), _i]),
    "ktlvo_destroy": (None, [_vp]),
    "ktlvo_set_stream": (_i, [_vp, _vp]),
    "ktlvo_sync": (_i, [_vp]),
    "ktlvo_strerror": (C.c_char_p, [_i]),
    "ktlvo_last_error": (C.c_char_p, [_vp]),
    "ktlvo_upload_image": (_i, [_vp, _i, _vp, _i, _i, _i]),
    "ktlvo_upload_images": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, C.c_size_t]),
    "ktlvo_set_image_dev": (_i, [_vp, _i, _vp, _i, _i, _i]),
    "ktlvo_build_pyramid": (_i, [_vp, _i, _i, _i]),
    "ktlvo_build_pyramids": (_i, [_vp, _ip, _i, _i, _i]),
    "ktlvo_pyramid_levels": (_i, [_vp, _i, _ip]),
    "ktlvo_download_level": (_i, [_vp, _i, _i, _vp, _ip, _ip]),
    "ktlvo_download_scharr": (_i, [_vp, _i, _i, _vp]),
    "ktlvo_pyrlk": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, C.POINTER(LkParams)]),
    "ktlvo_pyrlk_dev": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, C.POINTER(LkParams)]),
    "ktlvo_pyrlk_batch_dev": (_i, [_vp, _ip, _ip, _i, _vp, _i, _vp, _vp, _vp, C.POINTER(LkParams)]),
    "ktlvo_detect_gftt": (_i, [_vp, _i, C.POINTER(Roi), _i, C.POINTER(GfttParams), _vp, _vp, _vp, _i]),
    "ktlvo_detect_gftt_frames": (_i, [_vp, _ip, _i, C.POINTER(Roi), C.POINTER(GfttParams), _vp, _vp, _vp, _i]),
    "ktlvo_detect_gftt_frames_dev": (_i, [_vp, _ip, _i, C.POINTER(Roi), C.POINTER(GfttParams), _vp, _vp, _vp, _i]),
    "ktlvo_detect_build_frames_dev": (_i, [_vp, _ip, _i, C.POINTER(Roi), C.POINTER(GfttParams), _vp, _vp, _vp, _i, _ip, _i, _i, _i]),
    "ktlvo_detect_build_track_dev": (_i, [_vp, _ip, _ip, _i, C.POINTER(Roi), C.POINTER(GfttParams), _vp, _vp, _vp, _i, _ip, _i, _vp, _vp,
                                          _vp, C.POINTER(LkParams)]),
    "ktlvo_detect_gftt_noalloc": (_i, [_vp, _i, C.POINTER(Roi), _i, C.POINTER(GfttParams), _vp, _vp, _vp, _i]),
    "ktlvo_corner_response": (_i, [_vp, _i, C.POINTER(Roi), _i, _d, _vp]),
    "ktlvo_anms": (_i, [_vp, _vp, _vp, _ip, _i]),
    "ktlvo_fb_gate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _ip]),
    "ktlvo_circular_gate": (_i, [_vp, _vp, _vp, _i, _vp, _ip]),
    "ktlvo_get_optical_flow": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _ip]),
    "ktlvo_ransac_score": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _f, _vp, _i, _vp]),
    "ktlvo_ransac_score_dev": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _f, _vp, _i, _vp]),
    "ktlvo_solve_pnp_ransac": (_i, [_vp, _vp, _vp, _i, _i, _f, _d, _vp, _vp, _vp, _ip, _ip]),
    "ktlvo_track_stereo_frame": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _ip, _ip]),
    "ktlvo_estimate_pose": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _ip]),
    "ktlvo_ros_time_from_sec": (_i, [_d, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ktlvo_wire_size_image_points": (C.c_size_t, [_i, C.POINTER(RosHeader)]),
    "ktlvo_wire_size_depth_points": (C.c_size_t, [_i, C.POINTER(RosHeader)]),
    "ktlvo_wire_size_transform": (C.c_size_t, [C.POINTER(RosHeader)]),
    "ktlvo_wire_image_points": (_i, [_vp, _i, C.POINTER(RosHeader), _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ktlvo_wire_depth_points": (_i, [_vp, _i, C.POINTER(RosHeader), _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ktlvo_wire_transform": (_i, [_vp, C.POINTER(RosHeader), _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ktlvo_profile_enable": (_i, [_vp, _i]),
    "ktlvo_profile_reset": (_i, [_vp]),
    "ktlvo_profile_get": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C slam_amd/csrc` (hipcc, gfx950). "
                          "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name) and os.environ.get("KTLVO_AB_OLD_LIBRARY") == "1":
            continue  # A/B timing tools load an older build of the library in place of the current one
        fn = getattr(lib, name)  # AttributeError here means the library is stale
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def strerror(status):
    return lib.ktlvo_strerror(status).decode()


# -- wire format (host-only entry points, no context)
def ros_time_from_sec(t):
    sec, nsec = C.c_uint32(), C.c_uint32()
    rc = lib.ktlvo_ros_time_from_sec(t, C.byref(sec), C.byref(nsec))
    if rc != OK:
        raise KtlvoError(rc, "ktlvo_ros_time_from_sec")
    return sec.value, nsec.value


def _ros_header(seq, stamp, frame_id):
    sec, nsec = ros_time_from_sec(stamp) if not isinstance(stamp, tuple) else stamp
    return RosHeader(seq, sec, nsec, frame_id.encode() if frame_id is not None else None)


def _wire(need, put, where):
    buf = np.zeros(max(need, 1), np.uint8)
    written = C.c_size_t()
    rc = put(_ptr(buf), need, C.byref(written))
    if rc != OK:
        raise KtlvoError(rc, where)
    return buf[:written.value].tobytes()


def wire_image_points(pts, seq=0, stamp=0.0, frame_id=None):
    """ROS 1 body of the sensor_msgs/PointCloud2 the reference publishes on /image_points_cur."""
    pts = np.ascontiguousarray(pts, IMAGE_POINT)
    h = _ros_header(seq, stamp, frame_id)
    return _wire(lib.ktlvo_wire_size_image_points(len(pts), C.byref(h)),
                 lambda out, cap, wr: lib.ktlvo_wire_image_points(_ptr(pts), len(pts), C.byref(h), out, cap, wr),
                 "ktlvo_wire_image_points")


def wire_depth_points(pts, seq=0, stamp=0.0, frame_id=None):
    """ROS 1 body of the sensor_msgs/PointCloud2 the reference publishes on /image_points_last."""
    pts = np.ascontiguousarray(pts, DEPTH_POINT)
    h = _ros_header(seq, stamp, frame_id)
    return _wire(lib.ktlvo_wire_size_depth_points(len(pts), C.byref(h)),
                 lambda out, cap, wr: lib.ktlvo_wire_depth_points(_ptr(pts), len(pts), C.byref(h), out, cap, wr),
                 "ktlvo_wire_depth_points")


def wire_transform(T, seq=0, stamp=0.0, frame_id=None):
    """ROS 1 body of the ktl_vo/transform_msg the reference publishes on /local_transform."""
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    h = _ros_header(seq, stamp, frame_id)
    return _wire(lib.ktlvo_wire_size_transform(C.byref(h)),
                 lambda out, cap, wr: lib.ktlvo_wire_transform(_ptr(T), C.byref(h), out, cap, wr),
                 "ktlvo_wire_transform")


def default_config(**kw):
    c = Config()
    lib.ktlvo_default_config(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def lk_params(win=21, max_level=4, max_count=30, epsilon=0.01, min_eig_thr=1e-4, sums=LK_SUMS_DEFAULT):
    return LkParams(win, max_level, max_count, epsilon, min_eig_thr, sums)


def gftt_params(max_corners=100, quality=0.01, min_dist=10.0, block_size=3, use_harris=1, harris_k=0.0):
    return GfttParams(max_corners, quality, min_dist, block_size, use_harris, harris_k)


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Context:
    """Thin object wrapper over ktlvo_ctx*; numpy arrays in, numpy arrays out."""

    def __init__(self, cfg=None, device=0):
        self.cfg = cfg or default_config()
        self._h = lib.ktlvo_create(C.byref(self.cfg), device)
        if not self._h:
            raise RuntimeError("ktlvo_create failed (no MI355X visible, bad config or out of memory)")

    def close(self):
        if getattr(self, "_h", None) and lib is not None:   # lib is None while the interpreter shuts down
            lib.ktlvo_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, where):
        if rc != OK:
            raise KtlvoError(rc, where, lib.ktlvo_last_error(self._h).decode())

    # -- streams / profiling
    def set_stream(self, stream_handle):
        self._chk(lib.ktlvo_set_stream(self._h, stream_handle), "ktlvo_set_stream")

    def sync(self):
        self._chk(lib.ktlvo_sync(self._h), "ktlvo_sync")

    def profile_enable(self, on=True):
        self._chk(lib.ktlvo_profile_enable(self._h, int(on)), "ktlvo_profile_enable")

    def profile_reset(self):
        self._chk(lib.ktlvo_profile_reset(self._h), "ktlvo_profile_reset")

    def profile_get(self, kid):
        ms, n = C.c_double(), C.c_longlong()
        self._chk(lib.ktlvo_profile_get(self._h, kid, C.byref(ms), C.byref(n)), "ktlvo_profile_get")
        return ms.value, n.value

    # -- images
    def upload_image(self, slot, img):
        """img: uint8 (h, w); rows may be padded (a view with strides (stride, 1)), as in a cv::Mat with step > cols."""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1 or img.strides[0] < img.shape[1]:
            img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        self._chk(lib.ktlvo_upload_image(self._h, slot, _ptr(img), w, h, img.strides[0]), "ktlvo_upload_image")

    def upload_images(self, slots, imgs):
        """imgs: uint8 (n, h, w) contiguous (pageable or pinned); frame i goes into slots[i]: one DMA, one launch."""
        imgs = np.asarray(imgs)
        assert imgs.dtype == np.uint8 and imgs.ndim == 3 and imgs.flags["C_CONTIGUOUS"] and len(slots) == imgs.shape[0]
        n, h, w = imgs.shape
        arr = (C.c_int * max(n, 1))(*slots)
        self._chk(lib.ktlvo_upload_images(self._h, arr, n, _ptr(imgs), w, h, w, w * h), "ktlvo_upload_images")

    def set_image_dev(self, slot, dptr, w, h, stride):
        self._chk(lib.ktlvo_set_image_dev(self._h, slot, dptr, w, h, stride), "ktlvo_set_image_dev")

    def build_pyramid(self, slot, max_level=None, win=None):
        self._chk(lib.ktlvo_build_pyramid(self._h, slot, self.cfg.max_level if max_level is None else max_level,
                                          self.cfg.win if win is None else win), "ktlvo_build_pyramid")

    def build_pyramids(self, slots, max_level=None, win=None):
        arr = (C.c_int * len(slots))(*slots)
        self._chk(lib.ktlvo_build_pyramids(self._h, arr, len(slots),
                                           self.cfg.max_level if max_level is None else max_level,
                                           self.cfg.win if win is None else win), "ktlvo_build_pyramids")

    def pyramid_levels(self, slot):
        n = C.c_int()
        self._chk(lib.ktlvo_pyramid_levels(self._h, slot, C.byref(n)), "ktlvo_pyramid_levels")
        return n.value

    def download_level(self, slot, level):
        buf = np.zeros(self.cfg.width * self.cfg.height, np.uint8)
        w, h = C.c_int(), C.c_int()
        self._chk(lib.ktlvo_download_level(self._h, slot, level, _ptr(buf), C.byref(w), C.byref(h)),
                  "ktlvo_download_level")
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()

    def download_scharr(self, slot, level, w, h):
        buf = np.zeros((h, w, 2), np.int16)
        self._chk(lib.ktlvo_download_scharr(self._h, slot, level, _ptr(buf)), "ktlvo_download_scharr")
        return buf

    # -- LK
    def pyrlk(self, prev_slot, next_slot, pts, params=None):
        params = params or lk_params(self.cfg.win, self.cfg.max_level)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        out = np.zeros((n, 2), np.float32)
        status = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self._chk(lib.ktlvo_pyrlk(self._h, prev_slot, next_slot, _ptr(pts), n, _ptr(out), _ptr(status), _ptr(err),
                                  C.byref(params)), "ktlvo_pyrlk")
        return out, status, err

    def pyrlk_dev(self, prev_slot, next_slot, d_prev, n, d_next, d_status, d_err, params):
        self._chk(lib.ktlvo_pyrlk_dev(self._h, prev_slot, next_slot, d_prev, n, d_next, d_status, d_err,
                                      C.byref(params)), "ktlvo_pyrlk_dev")

    def pyrlk_batch_dev(self, prev_slots, next_slots, d_prev, n_per_pair, d_next, d_status, d_err, params):
        n = len(prev_slots)
        a = (C.c_int * n)(*prev_slots)
        b = (C.c_int * n)(*next_slots)
        self._chk(lib.ktlvo_pyrlk_batch_dev(self._h, a, b, n, d_prev, n_per_pair, d_next, d_status, d_err,
                                            C.byref(params)), "ktlvo_pyrlk_batch_dev")

    # -- corners
    def corner_response(self, slot, roi, use_harris=True, k=0.0):
        r = Roi(*roi)
        out = np.zeros((r.h, r.w), np.float32)
        self._chk(lib.ktlvo_corner_response(self._h, slot, C.byref(r), int(use_harris), k, _ptr(out)),
                  "ktlvo_corner_response")
        return out

    def detect_gftt(self, slot, rois, params=None):
        params = params or gftt_params()
        n = len(rois)
        arr = (Roi * max(n, 1))(*[Roi(*r) for r in rois])
        cap = params.max_corners
        xy = np.zeros((max(n, 1), cap, 2), np.float32)
        resp = np.zeros((max(n, 1), cap), np.float32)
        counts = np.zeros(max(n, 1), np.int32)
        self._chk(lib.ktlvo_detect_gftt(self._h, slot, arr, n, C.byref(params), _ptr(xy), _ptr(resp), _ptr(counts),
                                        cap), "ktlvo_detect_gftt")
        return [(xy[i, :counts[i]].copy(), resp[i, :counts[i]].copy()) for i in range(n)]

    def detect_gftt_frames(self, slots, params, roi=None):
        """The same region (None = whole frame) of several slots in one call; a list of (xy, response) per slot."""
        n = len(slots)
        arr = (C.c_int * max(n, 1))(*slots)
        cap = params.max_corners
        xy = np.zeros((max(n, 1), cap, 2), np.float32)
        resp = np.zeros((max(n, 1), cap), np.float32)
        counts = np.zeros(max(n, 1), np.int32)
        r = C.byref(Roi(*roi)) if roi is not None else None
        self._chk(lib.ktlvo_detect_gftt_frames(self._h, arr, n, r, C.byref(params), _ptr(xy), _ptr(resp), _ptr(counts),
                                               cap), "ktlvo_detect_gftt_frames")
        return [(xy[i, :counts[i]].copy(), resp[i, :counts[i]].copy()) for i in range(n)]

    def detect_gftt_frames_dev(self, slots, params, d_xy, d_resp, d_counts, cap, roi=None):
        n = len(slots)
        arr = (C.c_int * max(n, 1))(*slots)
        r = C.byref(Roi(*roi)) if roi is not None else None
        self._chk(lib.ktlvo_detect_gftt_frames_dev(self._h, arr, n, r, C.byref(params), d_xy, d_resp, d_counts, cap),
                  "ktlvo_detect_gftt_frames_dev")

    def detect_build_frames_dev(self, detect_slots, params, d_xy, d_resp, d_counts, cap, build_slots, roi=None,
                                max_level=None, win=None):
        """Corners of detect_slots and pyramids of build_slots side by side (device outputs as detect_gftt_frames_dev)."""
        n, nb = len(detect_slots), len(build_slots)
        arr = (C.c_int * max(n, 1))(*detect_slots)
        barr = (C.c_int * max(nb, 1))(*build_slots)
        r = C.byref(Roi(*roi)) if roi is not None else None
        self._chk(lib.ktlvo_detect_build_frames_dev(self._h, arr, n, r, C.byref(params), d_xy, d_resp, d_counts, cap, barr, nb,
                                                    self.cfg.max_level if max_level is None else max_level,
                                                    self.cfg.win if win is None else win), "ktlvo_detect_build_frames_dev")

    def detect_build_track_dev(self, prev_slots, next_slots, params, d_xy, d_resp, d_counts, cap, build_slots, d_next, d_status,
                               d_err, lk, roi=None):
        """The frame-pair stream's step in one call: detect_build_frames_dev(prev_slots, ..., build_slots) and
        pyrlk_batch_dev(prev_slots, next_slots, d_xy, cap, ...), the LK launch queued behind the detection."""
        n, nb = len(prev_slots), len(build_slots)
        assert len(next_slots) == n
        pa, na = (C.c_int * max(n, 1))(*prev_slots), (C.c_int * max(n, 1))(*next_slots)
        barr = (C.c_int * max(nb, 1))(*build_slots)
        r = C.byref(Roi(*roi)) if roi is not None else None
        self._chk(lib.ktlvo_detect_build_track_dev(self._h, pa, na, n, r, C.byref(params), d_xy, d_resp, d_counts, cap, barr, nb,
                                                   d_next, d_status, d_err, C.byref(lk)), "ktlvo_detect_build_track_dev")

    def detect_gftt_noalloc(self, slot, rois, params, d_xy, d_resp, d_counts, cap):
        """Device-pointer form that allocates nothing; returns the status code (no exception) so that tests
        can probe its limits.  Outputs are valid after sync()."""
        n = len(rois)
        arr = (Roi * max(n, 1))(*[Roi(*r) for r in rois])
        return lib.ktlvo_detect_gftt_noalloc(self._h, slot, arr, n, C.byref(params), d_xy, d_resp, d_counts, cap)

    def anms(self, xy, resp, num_to_keep):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2).copy()
        resp = np.ascontiguousarray(resp, np.float32).copy()
        n = C.c_int(resp.shape[0])
        self._chk(lib.ktlvo_anms(self._h, _ptr(xy), _ptr(resp), C.byref(n), num_to_keep), "ktlvo_anms")
        return xy[:n.value], resp[:n.value]

    # -- gates
    def fb_gate(self, f1, f2, f1chk, y_threshold):
        f1 = np.ascontiguousarray(f1, np.float32).reshape(-1, 2)
        f2 = np.ascontiguousarray(f2, np.float32).reshape(-1, 2)
        f1chk = np.ascontiguousarray(f1chk, np.float32).reshape(-1, 2)
        n = f1.shape[0]
        idx = np.zeros(max(n, 1), np.int32)
        m = C.c_int()
        self._chk(lib.ktlvo_fb_gate(self._h, _ptr(f1), _ptr(f2), _ptr(f1chk), n, y_threshold, _ptr(idx), C.byref(m)),
                  "ktlvo_fb_gate")
        return idx[:m.value].copy()

    def circular_gate(self, pcr, clr):
        pcr = np.ascontiguousarray(pcr, np.float32).reshape(-1, 2)
        clr = np.ascontiguousarray(clr, np.float32).reshape(-1, 2)
        n = pcr.shape[0]
        idx = np.zeros(max(n, 1), np.int32)
        m = C.c_int()
        self._chk(lib.ktlvo_circular_gate(self._h, _ptr(pcr), _ptr(clr), n, _ptr(idx), C.byref(m)),
                  "ktlvo_circular_gate")
        return idx[:m.value].copy()

    def get_optical_flow(self, slot_1, slot_2, f1, y_threshold):
        f1 = np.ascontiguousarray(f1, np.float32).reshape(-1, 2)
        n = f1.shape[0]
        a = np.zeros((max(n, 1), 2), np.float32)
        b = np.zeros((max(n, 1), 2), np.float32)
        idx = np.zeros(max(n, 1), np.int32)
        m = C.c_int()
        self._chk(lib.ktlvo_get_optical_flow(self._h, slot_1, slot_2, _ptr(f1), n, y_threshold, _ptr(a), _ptr(b),
                                             _ptr(idx), C.byref(m)), "ktlvo_get_optical_flow")
        return a[:m.value].copy(), b[:m.value].copy(), idx[:m.value].copy()

    # -- RANSAC
    def ransac_score(self, X3, x2, Rt, K, thr_sq, mask_hyp=-1):
        X3 = np.ascontiguousarray(X3, np.float32).reshape(-1, 3)
        x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
        Rt = np.ascontiguousarray(Rt, np.float64).reshape(-1, 12)
        K = np.ascontiguousarray(K, np.float64)
        n, H = X3.shape[0], Rt.shape[0]
        counts = np.zeros(max(H, 1), np.int32)
        mask = np.zeros(max(n, 1), np.uint8) if mask_hyp >= 0 else None
        self._chk(lib.ktlvo_ransac_score(self._h, _ptr(X3), _ptr(x2), n, _ptr(Rt), H, _ptr(K), np.float32(thr_sq),
                                         _ptr(counts), mask_hyp, _ptr(mask)), "ktlvo_ransac_score")
        return (counts[:H], mask[:n]) if mask is not None else counts[:H]

    def ransac_score_dev(self, d_X3, d_x2, n, d_Rt, H, K, thr_sq, d_counts, mask_hyp=-1, d_mask=None):
        """Device pointers (ints); asynchronous on the context's stream."""
        K = np.ascontiguousarray(K, np.float64)
        self._chk(lib.ktlvo_ransac_score_dev(self._h, d_X3, d_x2, n, d_Rt, H, _ptr(K), np.float32(thr_sq), d_counts,
                                             mask_hyp, d_mask), "ktlvo_ransac_score_dev")

    def solve_pnp_ransac(self, X3, x2, max_iters=300, reproj=0.1, confidence=float(np.float32(0.99))):
        X3 = np.ascontiguousarray(X3, np.float32).reshape(-1, 3)
        x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
        n = X3.shape[0]
        rvec, tvec = np.zeros(3), np.zeros(3)
        mask = np.zeros(max(n, 1), np.uint8)
        ninl, found = C.c_int(), C.c_int()
        self._chk(lib.ktlvo_solve_pnp_ransac(self._h, _ptr(X3), _ptr(x2), n, max_iters, np.float32(reproj),
                                             confidence, _ptr(rvec), _ptr(tvec), _ptr(mask), C.byref(ninl),
                                             C.byref(found)), "ktlvo_solve_pnp_ransac")
        return dict(ok=bool(found.value), rvec=rvec, tvec=tvec, mask=mask[:n], n_inliers=ninl.value)

    # -- frame level
    def track_stereo_frame(self, left, right, cap=20000):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        assert left.shape == (self.cfg.height, self.cfg.width) and right.shape == left.shape
        last = np.zeros(cap, DEPTH_POINT)
        cur = np.zeros(cap, IMAGE_POINT)
        n, kf = C.c_int(), C.c_int()
        self._chk(lib.ktlvo_track_stereo_frame(self._h, _ptr(left), _ptr(right), _ptr(last), _ptr(cur), cap,
                                               C.byref(n), C.byref(kf)), "ktlvo_track_stereo_frame")
        return last[:n.value].copy(), cur[:n.value].copy(), bool(kf.value)

    def estimate_pose(self, last, cur):
        last = np.ascontiguousarray(last, DEPTH_POINT)
        cur = np.ascontiguousarray(cur, IMAGE_POINT)
        Tl, Tg = np.zeros(16, np.float32), np.zeros(16, np.float32)
        ninl = C.c_int()
        self._chk(lib.ktlvo_estimate_pose(self._h, _ptr(last), _ptr(cur), last.shape[0], _ptr(Tl), _ptr(Tg),
                                          C.byref(ninl)), "ktlvo_estimate_pose")
        return ninl.value, Tl.reshape(4, 4), Tg.reshape(4, 4)
