/*
 * lightloam_scan_registration_node.cpp -- the reference's `ascanRegistration` node with its topic surface unchanged
 * and every scan going through the HIP hot path (lightloam::laserCloudHandler -> ll_upload_scan / ll_extract_batch).
 *
 * Mirrors /root/reference src/scanRegistration.cpp:430-476 (main) and :87-428 (the callback):
 *   parameters   scan_line (16), minimum_range (0.1), lowerBound (-24.9), upBound (2)                     :435-441
 *   subscribes   /rslidar_points   sensor_msgs/PointCloud2, queue 100                                      :453
 *   advertises   /velodyne_cloud_2, /laser_cloud_sharp, /laser_cloud_less_sharp, /laser_cloud_flat,
 *                /laser_cloud_less_flat (queue 100 each), stamped and framed like the input message         :455-463, :382-410
 *                /laser_remove_points is advertised and never published there; here as well                 :465
 *   exits 0 when scan_line is not 16 / 32 / 64                                                             :447-451
 *   warns when a scan takes more than 100 ms                                                               :426-427
 * Not carried over: PUB_EACH_LINE (compiled out in the reference, :54).
 *
 * Build inside the catkin package: roscpp + sensor_msgs, link liblightloam_hip.so.  No PCL.
 * This image has no ROS: tests/native/ros_node_double.cpp compiles this file against declared test doubles of the four
 * roscpp / sensor_msgs classes it uses and drives the callback.
 */
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "lightloam_host.hpp"
#include "lightloam_ros.hpp"

namespace {

const int systemDelay = 0;                                  /* :30 */
int systemInitCount = 0;
bool systemInited = false;

std::unique_ptr<lightloam::Context> g_ll;
ros::Publisher pubLaserCloud, pubCornerPointsSharp, pubCornerPointsLessSharp, pubSurfPointsFlat, pubSurfPointsLessFlat, pubRemovePoints;

void publish_cloud(ros::Publisher &pub, const std::vector<lightloam::PointXYZI> &pts, const sensor_msgs::PointCloud2 &in)
{
    sensor_msgs::PointCloud2 out;
    lightloam::ros_io::cloud2_from_points(pts, out);
    out.header.stamp = in.header.stamp;                     /* :384-385 */
    out.header.frame_id = in.header.frame_id;
    pub.publish(out);
}

}  // namespace

void laserCloudHandler(const sensor_msgs::PointCloud2ConstPtr &laserCloudMsg)
{
    if (!systemInited) {                                    /* :89-98 */
        systemInitCount++;
        if (systemInitCount >= systemDelay) systemInited = true;
        else return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<float> xyz;
    if (!lightloam::ros_io::xyz_from_cloud2(*laserCloudMsg, xyz)) {
        ROS_WARN("scan registration: message without float x / y / z fields");
        return;
    }
    std::vector<lightloam::PointXYZI> laserCloud, sharp, lessSharp, flat, lessFlat;
    try {
        /* NaN removal, the minimum-range filter and everything up to the five clouds happen on the device (:109-376) */
        if (!lightloam::laserCloudHandler(*g_ll, 0, xyz.data(), 4, (int)(xyz.size() / 4), laserCloud, sharp, lessSharp, flat, lessFlat))
            return;                                         /* no point survived the filters: the reference reads points[0] here */
    } catch (const lightloam::Error &e) {
        ROS_ERROR("scan registration: %s (code %d)", e.what(), e.code);
        return;
    }
    publish_cloud(pubLaserCloud, laserCloud, *laserCloudMsg);                 /* :382-386 */
    publish_cloud(pubCornerPointsSharp, sharp, *laserCloudMsg);               /* :388-392 */
    publish_cloud(pubCornerPointsLessSharp, lessSharp, *laserCloudMsg);       /* :394-398 */
    publish_cloud(pubSurfPointsFlat, flat, *laserCloudMsg);                   /* :400-404 */
    publish_cloud(pubSurfPointsLessFlat, lessFlat, *laserCloudMsg);           /* :406-410 */
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 100) ROS_WARN("scan registration process over 100ms");           /* :426-427 */
}

int main(int argc, char **argv)
{
    ros::init(argc, argv, "scanRegistration");
    ros::NodeHandle nh;
    int N_SCANS = 16;
    double MINIMUM_RANGE = 0.1;
    float lowerBound = -24.9f, upBound = 2.0f;
    nh.param<int>("scan_line", N_SCANS, 16);                                  /* :435 */
    nh.param<double>("minimum_range", MINIMUM_RANGE, 0.1);                    /* :438 */
    nh.param<float>("lowerBound", lowerBound, -24.9f);                        /* :439 */
    nh.param<float>("upBound", upBound, 2.0f);                                /* :440 */
    if (N_SCANS != 16 && N_SCANS != 32 && N_SCANS != 64) return 0;            /* :447-451 */
    try {
        g_ll.reset(new lightloam::Context(N_SCANS, /*batch*/ 2, /*device*/ 0, MINIMUM_RANGE, lowerBound, upBound));
    } catch (const lightloam::Error &e) {
        ROS_ERROR("scan registration: no MI355X context: %s (code %d)", e.what(), e.code);   /* there is no CPU fallback */
        return 1;
    }
    ros::Subscriber subLaserCloud = nh.subscribe<sensor_msgs::PointCloud2>("/rslidar_points", 100, laserCloudHandler);   /* :453 */
    pubLaserCloud = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_2", 100);                 /* :455 */
    pubCornerPointsSharp = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 100);         /* :457 */
    pubCornerPointsLessSharp = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 100);/* :459 */
    pubSurfPointsFlat = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_flat", 100);             /* :461 */
    pubSurfPointsLessFlat = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 100);    /* :463 */
    pubRemovePoints = nh.advertise<sensor_msgs::PointCloud2>("/laser_remove_points", 100);            /* :465 */
    ros::spin();                                                              /* :475: single-threaded, one context */
    g_ll.reset();
    return 0;
}
