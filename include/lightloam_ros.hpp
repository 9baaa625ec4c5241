/*
 * lightloam_ros.hpp -- sensor_msgs/PointCloud2 <-> packed points, without PCL.
 *
 * The reference nodes move clouds with pcl::fromROSMsg / pcl::toROSMsg (scanRegistration.cpp:105-106, :382-410 of
 * /root/reference).  The node wrappers under ros/ do the same conversions here, so that they depend on roscpp and
 * sensor_msgs only.  Templates over the message type: nothing in this header includes a ROS header, which lets
 * tests/native compile it (and the nodes) against declared test doubles of the message classes.
 *
 * Wire layout written by cloud2_from_points = what pcl::toROSMsg makes of a pcl::PointCloud<pcl::PointXYZI>
 * (PCL 1.10, restated: the source is not in the reference tree): height 1, width n, fields x / y / z / intensity as
 * FLOAT32 at offsets 0 / 4 / 8 / 16, point_step 32 (the SSE padding of PointXYZI travels), little endian, is_dense.
 * Any subscriber that uses pcl::fromROSMsg (laserOdometry.cpp:116-150) maps fields by name and reads it unchanged.
 */
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "lightloam_host.hpp"

namespace lightloam {
namespace ros_io {

enum { FLOAT32 = 7 };                                      /* sensor_msgs::PointField::FLOAT32 */

/* offset of the FLOAT32 field `name`, -1 if the message has none */
template <class Cloud2>
inline int field_offset(const Cloud2 &m, const char *name)
{
    for (const auto &f : m.fields)
        if (f.name == name && f.datatype == FLOAT32 && f.count >= 1) return (int)f.offset;
    return -1;
}

/* pcl::fromROSMsg<pcl::PointXYZ>: x, y, z of every point of the message, 4 floats per point (w = 0), row-major over
 * (height, width).  Returns false when the message has no float x / y / z fields or its data is short. */
template <class Cloud2>
inline bool xyz_from_cloud2(const Cloud2 &m, std::vector<float> &xyz4)
{
    const int ox = field_offset(m, "x"), oy = field_offset(m, "y"), oz = field_offset(m, "z");
    if (ox < 0 || oy < 0 || oz < 0 || m.is_bigendian) return false;
    const size_t w = m.width, h = m.height, step = m.point_step, row = m.row_step ? m.row_step : w * step;
    if (step < 12 || (h && w && m.data.size() < (h - 1) * row + w * step)) return false;
    xyz4.assign(4 * w * h, 0.0f);
    size_t k = 0;
    for (size_t r = 0; r < h; ++r)
        for (size_t c = 0; c < w; ++c, ++k) {
            const uint8_t *p = &m.data[r * row + c * step];
            std::memcpy(&xyz4[4 * k + 0], p + ox, 4);
            std::memcpy(&xyz4[4 * k + 1], p + oy, 4);
            std::memcpy(&xyz4[4 * k + 2], p + oz, 4);
        }
    return true;
}

/* pcl::fromROSMsg<pcl::PointXYZI>: also the intensity (the ring id + relative time the registration node stores there) */
template <class Cloud2>
inline bool points_from_cloud2(const Cloud2 &m, std::vector<PointXYZI> &pts)
{
    std::vector<float> xyz4;
    if (!xyz_from_cloud2(m, xyz4)) return false;
    const int oi = field_offset(m, "intensity");
    const size_t w = m.width, h = m.height, step = m.point_step, row = m.row_step ? m.row_step : w * step;
    pts.resize(w * h);
    size_t k = 0;
    for (size_t r = 0; r < h; ++r)
        for (size_t c = 0; c < w; ++c, ++k) {
            pts[k].x = xyz4[4 * k]; pts[k].y = xyz4[4 * k + 1]; pts[k].z = xyz4[4 * k + 2]; pts[k].intensity = 0.0f;
            if (oi >= 0) std::memcpy(&pts[k].intensity, &m.data[r * row + c * step + oi], 4);
        }
    return true;
}

/* pcl::toROSMsg of a PointXYZI cloud; header (stamp, frame_id) is the caller's, as in the reference (:384-385) */
template <class Cloud2>
inline void cloud2_from_points(const std::vector<PointXYZI> &pts, Cloud2 &m)
{
    typedef typename std::remove_reference<decltype(m.fields)>::type Fields;
    typedef typename Fields::value_type Field;
    static const struct { const char *name; uint32_t offset; } layout[4] = {{"x", 0}, {"y", 4}, {"z", 8}, {"intensity", 16}};
    m.fields.clear();
    for (const auto &l : layout) {
        Field f;
        f.name = l.name; f.offset = l.offset; f.datatype = FLOAT32; f.count = 1;
        m.fields.push_back(f);
    }
    m.height = 1; m.width = (uint32_t)pts.size();
    m.is_bigendian = false; m.is_dense = true;
    m.point_step = 32; m.row_step = 32 * m.width;
    m.data.assign((size_t)m.row_step, 0);
    for (size_t i = 0; i < pts.size(); ++i) {
        uint8_t *p = &m.data[32 * i];
        const float one = 1.0f;                            /* pcl's PointXYZ union: data[3] = 1 */
        std::memcpy(p + 0, &pts[i].x, 4); std::memcpy(p + 4, &pts[i].y, 4); std::memcpy(p + 8, &pts[i].z, 4);
        std::memcpy(p + 12, &one, 4);
        std::memcpy(p + 16, &pts[i].intensity, 4);
    }
}

}  // namespace ros_io
}  // namespace lightloam
