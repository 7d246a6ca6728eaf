"""CPU restatement of the ROS 1 wire image of the reference's three output messages.

TEST INFRASTRUCTURE ONLY (see oracle/ktlvo_oracle.h).  PARITY UNPINNED: neither ROS nor PCL is
installed here and the reference holds no recorded messages; this follows the published roscpp
serialisation rules and pcl::toROSMsg / pcl::toPCLPointCloud2 as the reference calls them:
  featureTracking.cpp:656-664   pcl::toROSMsg(*imagePointsLast / *imagePointsCur, msg); msg.header.stamp = ...
  visualOdometry.cpp:316-340    transform_msg.transform[i*4+j] = mTcw_local(i,j); header.stamp; the two clouds again
  pointDefinition.h:18-42       the registered point structs (field names, order, types)
  msg/transform_msg.msg:1-2     Header header; float32[16] transform
Written independently of slam_amd/csrc/wire.hip, with struct.pack.
"""
import math
import struct

INT32, FLOAT32 = 5, 7   # sensor_msgs/PointField datatypes
IMAGE_FIELDS = [("u", FLOAT32, "f"), ("v", FLOAT32, "f"), ("ind", INT32, "i")]
DEPTH_FIELDS = [("u", FLOAT32, "f"), ("v", FLOAT32, "f"), ("depth", FLOAT32, "f"), ("label", INT32, "i"),
                ("ind", INT32, "i"), ("newKeyFrame", INT32, "i")]


def time_from_sec(t):
    """ros::TimeBase::fromSec."""
    sec = math.floor(t)
    nsec = int(round((t - sec) * 1e9))   # boost::math::round: halves away from zero; t >= 0 here
    if (t - sec) * 1e9 - math.floor((t - sec) * 1e9) == 0.5:
        nsec = int(math.floor((t - sec) * 1e9)) + 1
    sec += nsec // 1000000000
    nsec %= 1000000000
    return int(sec), int(nsec)


def _string(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def header(seq, stamp, frame_id=""):
    sec, nsec = stamp if isinstance(stamp, tuple) else time_from_sec(stamp)
    return struct.pack("<III", seq, sec, nsec) + _string(frame_id or "")


def point_cloud2(fields, rows, seq=0, stamp=0.0, frame_id=""):
    """rows: iterable of tuples in field order.  pcl::toPCLPointCloud2 for a cloud filled by push_back:
    width = size, height = 1, packed fields, is_dense = true; little endian."""
    rows = list(rows)
    fmt = "<" + "".join(c for _, _, c in fields)
    step = struct.calcsize(fmt)
    out = header(seq, stamp, frame_id)
    out += struct.pack("<II", 1, len(rows))
    out += struct.pack("<I", len(fields))
    off = 0
    for name, dt, c in fields:
        out += _string(name) + struct.pack("<IBI", off, dt, 1)
        off += struct.calcsize("<" + c)
    data = b"".join(struct.pack(fmt, *r) for r in rows)
    out += struct.pack("<BII", 0, step, step * len(rows))
    out += struct.pack("<I", len(data)) + data
    out += struct.pack("<B", 1)
    return out


def transform_msg(T16, seq=0, stamp=0.0, frame_id=""):
    return header(seq, stamp, frame_id) + struct.pack("<16f", *[float(x) for x in T16])
