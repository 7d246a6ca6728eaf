// ROS 1 wire serialisation of the reference's three output messages (host only).
//   sensor_msgs/PointCloud2 as pcl::toROSMsg fills it for the registered point structs of
//   pointDefinition.h:18-42 (featureTracking.cpp:656-664, visualOdometry.cpp:332-340), and
//   ktl_vo/transform_msg (msg/transform_msg.msg:1-2, filled at visualOdometry.cpp:316-330).
// roscpp serialises field by field in declaration order, little endian: strings and variable-length
// arrays carry a uint32 length, fixed-length arrays and bool/uint8 do not.
#include <cmath>
#include <cstring>

#include "ktlvo.h"

namespace {

struct Field {
    const char *name;
    uint32_t offset;
    uint8_t datatype; // sensor_msgs/PointField: INT32 = 5, FLOAT32 = 7
};
const Field IMAGE_FIELDS[] = {{"u", 0, 7}, {"v", 4, 7}, {"ind", 8, 5}};
const Field DEPTH_FIELDS[] = {{"u", 0, 7}, {"v", 4, 7}, {"depth", 8, 7}, {"label", 12, 5}, {"ind", 16, 5},
                              {"newKeyFrame", 20, 5}};

struct Writer {
    uint8_t *p;
    void u8(uint8_t v) { *p++ = v; }
    void u32(uint32_t v) { // little endian whatever the host is
        p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
        p += 4;
    }
    void bytes(const void *src, size_t n) { if (n) memcpy(p, src, n); p += n; }
    void str(const char *s) { const size_t n = s ? strlen(s) : 0; u32((uint32_t)n); bytes(s, n); }
};

size_t header_size(const ktlvo_ros_header *h) { return 4 + 8 + 4 + (h->frame_id ? strlen(h->frame_id) : 0); }
void put_header(Writer &w, const ktlvo_ros_header *h) {
    w.u32(h->seq); w.u32(h->stamp_sec); w.u32(h->stamp_nsec); w.str(h->frame_id);
}

size_t cloud_size(const Field *f, int nf, size_t point_step, int n, const ktlvo_ros_header *h) {
    if (!h || n < 0) return 0;
    size_t s = header_size(h) + 4 + 4 + 4; // header, height, width, fields length
    for (int i = 0; i < nf; i++) s += 4 + strlen(f[i].name) + 4 + 1 + 4;
    return s + 1 + 4 + 4 + 4 + point_step * (size_t)n + 1; // is_bigendian, point_step, row_step, data length, data, is_dense
}

int put_cloud(const Field *f, int nf, size_t point_step, const void *pts, int n, const ktlvo_ros_header *h,
              uint8_t *out, size_t cap, size_t *written) {
    if (!h || n < 0 || (n > 0 && !pts) || !out || !written) return KTLVO_ERR_INVALID;
    if ((uint64_t)point_step * (uint64_t)n > 0xFFFFFFFFull) return KTLVO_ERR_CAPACITY; // row_step is a uint32
    const size_t need = cloud_size(f, nf, point_step, n, h);
    if (cap < need) return KTLVO_ERR_CAPACITY;
    Writer w{out};
    put_header(w, h);
    w.u32(1);            // height: an unorganised cloud
    w.u32((uint32_t)n);  // width
    w.u32((uint32_t)nf);
    for (int i = 0; i < nf; i++) { w.str(f[i].name); w.u32(f[i].offset); w.u8(f[i].datatype); w.u32(1); }
    w.u8(0);             // is_bigendian
    w.u32((uint32_t)point_step);
    w.u32((uint32_t)(point_step * (size_t)n)); // row_step
    w.u32((uint32_t)(point_step * (size_t)n)); // data length
    w.bytes(pts, point_step * (size_t)n);      // the structs are packed 4-byte members: memory image == wire image on little-endian hosts
    w.u8(1);             // is_dense
    *written = (size_t)(w.p - out);
    return KTLVO_OK;
}

} // namespace

static_assert(sizeof(ktlvo_image_point) == 12 && sizeof(ktlvo_depth_point) == 24, "wire PODs of pointDefinition.h");

extern "C" int ktlvo_ros_time_from_sec(double t, uint32_t *sec, uint32_t *nsec) {
    // ros::TimeBase::fromSec: sec = floor(t), nsec = round((t - sec) * 1e9), then carry
    if (!sec || !nsec || !(t >= 0.0) || t >= 4294967296.0) return KTLVO_ERR_INVALID;
    double s = std::floor(t);
    uint64_t n = (uint64_t)std::llround((t - s) * 1e9);
    uint64_t si = (uint64_t)s + n / 1000000000ull;
    if (si > 0xFFFFFFFFull) return KTLVO_ERR_INVALID;
    *sec = (uint32_t)si;
    *nsec = (uint32_t)(n % 1000000000ull);
    return KTLVO_OK;
}

extern "C" size_t ktlvo_wire_size_image_points(int n, const ktlvo_ros_header *h) {
    return cloud_size(IMAGE_FIELDS, 3, sizeof(ktlvo_image_point), n, h);
}
extern "C" size_t ktlvo_wire_size_depth_points(int n, const ktlvo_ros_header *h) {
    return cloud_size(DEPTH_FIELDS, 6, sizeof(ktlvo_depth_point), n, h);
}
extern "C" size_t ktlvo_wire_size_transform(const ktlvo_ros_header *h) { return h ? header_size(h) + 64 : 0; }

extern "C" int ktlvo_wire_image_points(const ktlvo_image_point *pts, int n, const ktlvo_ros_header *h, uint8_t *out,
                                       size_t cap, size_t *written) {
    return put_cloud(IMAGE_FIELDS, 3, sizeof(ktlvo_image_point), pts, n, h, out, cap, written);
}
extern "C" int ktlvo_wire_depth_points(const ktlvo_depth_point *pts, int n, const ktlvo_ros_header *h, uint8_t *out,
                                       size_t cap, size_t *written) {
    return put_cloud(DEPTH_FIELDS, 6, sizeof(ktlvo_depth_point), pts, n, h, out, cap, written);
}
extern "C" int ktlvo_wire_transform(const float T[16], const ktlvo_ros_header *h, uint8_t *out, size_t cap,
                                    size_t *written) {
    if (!T || !h || !out || !written) return KTLVO_ERR_INVALID;
    if (cap < header_size(h) + 64) return KTLVO_ERR_CAPACITY;
    Writer w{out};
    put_header(w, h);
    for (int i = 0; i < 16; i++) { uint32_t u; memcpy(&u, &T[i], 4); w.u32(u); } // float32[16]: fixed length, no prefix
    *written = (size_t)(w.p - out);
    return KTLVO_OK;
}
