/* Declared test double of the part of tf that ros/lightloam_laser_mapping_node.cpp uses (see ros/ros.h here): the last
 * transform sent is kept in ros::Double-style globals for the test. */
#pragma once
#include <string>
#include "ros/ros.h"

namespace tf {
struct Vector3 { double x, y, z; Vector3(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {} };
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 1;
    void setX(double v) { x = v; } void setY(double v) { y = v; } void setZ(double v) { z = v; } void setW(double v) { w = v; }
};
struct Transform {
    Vector3 origin; Quaternion rotation;
    void setOrigin(const Vector3 &o) { origin = o; }
    void setRotation(const Quaternion &q) { rotation = q; }
};
struct StampedTransform : Transform {
    ros::Time stamp; std::string frame_id, child_frame_id;
    StampedTransform(const Transform &t, const ros::Time &s, const std::string &f, const std::string &c) : Transform(t), stamp(s), frame_id(f), child_frame_id(c) {}
};
struct Sent { int count = 0; std::string frame_id, child_frame_id; double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}; static Sent &get() { static Sent s; return s; } };
class TransformBroadcaster {
public:
    void sendTransform(const StampedTransform &t) {
        auto &S = Sent::get();
        S.count++; S.frame_id = t.frame_id; S.child_frame_id = t.child_frame_id;
        S.t[0] = t.origin.x; S.t[1] = t.origin.y; S.t[2] = t.origin.z;
        S.q[0] = t.rotation.x; S.q[1] = t.rotation.y; S.q[2] = t.rotation.z; S.q[3] = t.rotation.w;
    }
};
}  // namespace tf
