"""The ROS 1 wire emitters of the C ABI against the independent CPU restatement (oracle/ros_wire.py)
and against hand-assembled known answers.  Host-only entry points: no GPU needed."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import ros_wire as W


def test_known_answer_bytes():
    # one ImagePoint (1.5, -2.0, 7), seq 3, stamp 12.25 s, frame "cam": every byte spelled out
    expect = (struct.pack("<III", 3, 12, 250000000) + struct.pack("<I", 3) + b"cam" +
              struct.pack("<II", 1, 1) + struct.pack("<I", 3) +
              struct.pack("<I", 1) + b"u" + struct.pack("<IBI", 0, 7, 1) +
              struct.pack("<I", 1) + b"v" + struct.pack("<IBI", 4, 7, 1) +
              struct.pack("<I", 3) + b"ind" + struct.pack("<IBI", 8, 5, 1) +
              struct.pack("<BII", 0, 12, 12) + struct.pack("<I", 12) + struct.pack("<ffi", 1.5, -2.0, 7) +
              struct.pack("<B", 1))
    assert W.point_cloud2(W.IMAGE_FIELDS, [(1.5, -2.0, 7)], 3, 12.25, "cam") == expect
    assert len(W.transform_msg(range(16))) == 16 + 64


def test_time_from_sec_oracle():
    assert W.time_from_sec(0.0) == (0, 0)
    assert W.time_from_sec(1317617735.8837) == (1317617735, 883699894)   # a KITTI-like stamp minus 0.1163
    assert W.time_from_sec(5.9999999999) == (6, 0)                        # nsec rounds up to 1e9 and carries


def test_library_matches_oracle(klib):
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 777):
        ip = np.zeros(n, klib.IMAGE_POINT)
        ip["u"], ip["v"] = rng.uniform(0, 1242, n), rng.uniform(0, 375, n)
        ip["ind"] = rng.integers(0, 1 << 20, n)
        dp = np.zeros(n, klib.DEPTH_POINT)
        dp["u"], dp["v"], dp["depth"] = rng.uniform(0, 1242, n), rng.uniform(0, 375, n), rng.uniform(1, 70, n)
        dp["label"], dp["ind"], dp["newKeyFrame"] = 1, rng.integers(0, 1 << 20, n), rng.integers(0, 2, n)
        for seq, stamp, frame in ((0, 0.0, None), (41, 1317617735.8837, "camera"), (2 ** 32 - 1, 99.5, "")):
            got = klib.wire_image_points(ip, seq, stamp, frame)
            assert got == W.point_cloud2(W.IMAGE_FIELDS, [tuple(r) for r in ip.tolist()], seq, stamp, frame or "")
            got = klib.wire_depth_points(dp, seq, stamp, frame)
            assert got == W.point_cloud2(W.DEPTH_FIELDS, [tuple(r) for r in dp.tolist()], seq, stamp, frame or "")
    T = rng.normal(size=16).astype(np.float32)
    assert klib.wire_transform(T, 5, 3.75, "odom") == W.transform_msg(T, 5, 3.75, "odom")
    for t in (0.0, 0.1163, 5.9999999999, 1317617735.8837, 4294967295.25):
        assert klib.ros_time_from_sec(t) == W.time_from_sec(t), t


def test_library_errors(klib):
    h = klib.RosHeader(0, 0, 0, None)
    pts = np.zeros(4, klib.IMAGE_POINT)
    need = klib.lib.ktlvo_wire_size_image_points(4, C.byref(h))
    assert need == len(W.point_cloud2(W.IMAGE_FIELDS, [(0.0, 0.0, 0)] * 4))
    buf = np.full(need, 0xAB, np.uint8)
    wr = C.c_size_t(123)
    rc = klib.lib.ktlvo_wire_image_points(pts.ctypes.data, 4, C.byref(h), buf.ctypes.data, need - 1, C.byref(wr))
    assert rc == klib.ERR_CAPACITY and (buf == 0xAB).all()          # too small: nothing written
    assert klib.lib.ktlvo_wire_image_points(None, 4, C.byref(h), buf.ctypes.data, need, C.byref(wr)) == klib.ERR_INVALID
    assert klib.lib.ktlvo_wire_image_points(pts.ctypes.data, -1, C.byref(h), buf.ctypes.data, need, C.byref(wr)) == klib.ERR_INVALID
    assert klib.lib.ktlvo_wire_size_image_points(-1, C.byref(h)) == 0
    assert klib.lib.ktlvo_wire_size_transform(None) == 0
    with pytest.raises(klib.KtlvoError):
        klib.ros_time_from_sec(-1.0)
    with pytest.raises(klib.KtlvoError):
        klib.ros_time_from_sec(float("nan"))
