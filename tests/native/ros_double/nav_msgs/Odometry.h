/* Declared test double of nav_msgs/Odometry, nav_msgs/Path, geometry_msgs/PoseStamped: public data members only. */
#pragma once
#include <string>
#include <vector>
#include "sensor_msgs/PointCloud2.h"

namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; double covariance[36] = {0}; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
}  // namespace geometry_msgs

namespace nav_msgs {
struct Odometry {
    std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose;
    typedef std::shared_ptr<const Odometry> ConstPtr;
};
struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; };
}  // namespace nav_msgs
