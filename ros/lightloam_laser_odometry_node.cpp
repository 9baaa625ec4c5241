/*
 * lightloam_laser_odometry_node.cpp -- the reference's `alaserOdometry` node with its topic surface unchanged; the
 * per-frame body (correspondences, graph vote, 3 x ceres::Solve) runs on the device through ll_upload_features +
 * ll_odometry_frames, the last frame's clouds stay there as the next target (ll_set_target_from_slot).
 *
 * Mirrors /root/reference src/laserOdometry.cpp:345-931:
 *   parameters   scan_line (16), mapping_skip_frame (2)                                                    :349-350
 *   subscribes   /laser_cloud_sharp, /laser_cloud_less_sharp, /laser_cloud_flat, /laser_cloud_less_flat,
 *                /velodyne_cloud_2 (queue 100 each) into five queues under one mutex                       :116-150, :354-362
 *   advertises   /laser_cloud_corner_last, /laser_cloud_surf_last, /velodyne_cloud_3 (PointCloud2),
 *                /laser_odom_to_init (nav_msgs/Odometry), /laser_odom_path (nav_msgs/Path), queue 100      :364-372
 *   loop         100 Hz; a frame is taken when all five queues hold a message; their stamps must be equal
 *                or the node breaks (:394-401); the first frame only initialises (:426-430)
 *   per frame    para_q / para_t warm-started from the previous frame (:61-65), vote when now_frame > 5 (:794),
 *                t_w_curr += q_w_curr * t_last_curr; q_w_curr *= q_last_curr (:830-831)
 *   publishes    odometry: frame "rslidar", child "/laser_odom", stamp of the less-flat message (:834-848);
 *                path: that pose appended, frame "rslidar" (:850-856); every mapping_skip_frame-th frame the less-sharp
 *                and less-flat clouds and the full cloud, frame "/camera" (:898-918)
 * Build inside the catkin package: roscpp, sensor_msgs, nav_msgs, geometry_msgs; no PCL, Eigen or Ceres.
 * This image has no ROS: tests/native/ros_odometry_double.cpp compiles this file against declared test doubles.
 */
#include <chrono>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <geometry_msgs/PoseStamped.h>

#include "lightloam_host.hpp"
#include "lightloam_ros.hpp"

namespace {

std::queue<sensor_msgs::PointCloud2ConstPtr> cornerSharpBuf, cornerLessSharpBuf, surfFlatBuf, surfLessFlatBuf, fullPointsBuf;   /* :67-71 */
std::mutex mBuf;                                                                                                               /* :72 */

void laserCloudSharpHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); cornerSharpBuf.push(m); }          /* :116-121 */
void laserCloudLessSharpHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); cornerLessSharpBuf.push(m); }  /* :123-128 */
void laserCloudFlatHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); surfFlatBuf.push(m); }              /* :130-135 */
void laserCloudLessFlatHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); surfLessFlatBuf.push(m); }      /* :137-142 */
void laserCloudFullResHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); fullPointsBuf.push(m); }         /* :145-150 */

void publish_cloud(ros::Publisher &pub, const std::vector<lightloam::PointXYZI> &pts, double stamp)
{
    sensor_msgs::PointCloud2 out;
    lightloam::ros_io::cloud2_from_points(pts, out);
    out.header.stamp = ros::Time().fromSec(stamp);                            /* :902 */
    out.header.frame_id = "/camera";                                          /* :903 */
    pub.publish(out);
}

}  // namespace

int main(int argc, char **argv)
{
    ros::init(argc, argv, "laserOdometry");
    ros::NodeHandle nh;
    int N_SCANS = 16, skipFrameNum = 2;
    nh.param<int>("scan_line", N_SCANS, 16);                                  /* :349 */
    nh.param<int>("mapping_skip_frame", skipFrameNum, 2);                     /* :350 */
    std::unique_ptr<lightloam::Context> ll;
    try {
        ll.reset(new lightloam::Context(N_SCANS, /*batch*/ 2));               /* slot 0 = the current frame, carry = the last one */
    } catch (const lightloam::Error &e) {
        ROS_ERROR("laser odometry: no MI355X context: %s (code %d)", e.what(), e.code);      /* there is no CPU fallback */
        return 1;
    }
    ros::Subscriber subCornerPointsSharp = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 100, laserCloudSharpHandler);
    ros::Subscriber subCornerPointsLessSharp = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 100, laserCloudLessSharpHandler);
    ros::Subscriber subSurfPointsFlat = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_flat", 100, laserCloudFlatHandler);
    ros::Subscriber subSurfPointsLessFlat = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 100, laserCloudLessFlatHandler);
    ros::Subscriber subLaserCloudFullRes = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_cloud_2", 100, laserCloudFullResHandler);
    ros::Publisher pubLaserCloudCornerLast = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 100);
    ros::Publisher pubLaserCloudSurfLast = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 100);
    ros::Publisher pubLaserCloudFullRes = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_3", 100);
    ros::Publisher pubLaserOdometry = nh.advertise<nav_msgs::Odometry>("/laser_odom_to_init", 100);
    ros::Publisher pubLaserPath = nh.advertise<nav_msgs::Path>("/laser_odom_path", 100);

    nav_msgs::Path laserPath;
    lightloam::WorldPose w_curr;                                              /* q_w_curr, t_w_curr (:55-56) */
    double para[7] = {0, 0, 0, 1, 0, 0, 0};                                   /* para_q (x, y, z, w), para_t (:61-62) */
    bool systemInited = false;
    int frameCount = 0, now_frame = 0;
    ros::Rate rate(100);

    while (ros::ok()) {
        ros::spinOnce();
        sensor_msgs::PointCloud2ConstPtr mSharp, mLessSharp, mFlat, mLessFlat, mFull;
        {
            std::lock_guard<std::mutex> l(mBuf);
            if (!cornerSharpBuf.empty() && !cornerLessSharpBuf.empty() && !surfFlatBuf.empty() && !surfLessFlatBuf.empty() && !fullPointsBuf.empty()) {
                mSharp = cornerSharpBuf.front(); mLessSharp = cornerLessSharpBuf.front(); mFlat = surfFlatBuf.front();
                mLessFlat = surfLessFlatBuf.front(); mFull = fullPointsBuf.front();
                const double tFull = mFull->header.stamp.toSec();
                if (mSharp->header.stamp.toSec() != tFull || mLessSharp->header.stamp.toSec() != tFull ||
                    mFlat->header.stamp.toSec() != tFull || mLessFlat->header.stamp.toSec() != tFull) {
                    ROS_BREAK();                                              /* unsync message (:394-401) */
                }
                cornerSharpBuf.pop(); cornerLessSharpBuf.pop(); surfFlatBuf.pop(); surfLessFlatBuf.pop(); fullPointsBuf.pop();
            }
        }
        if (mFull) {
            const auto t0 = std::chrono::steady_clock::now();
            const double timeSurfPointsLessFlat = mLessFlat->header.stamp.toSec();
            std::vector<lightloam::PointXYZI> sharp, lessSharp, flat, lessFlat;
            namespace io = lightloam::ros_io;
            if (!io::points_from_cloud2(*mSharp, sharp) || !io::points_from_cloud2(*mLessSharp, lessSharp) ||
                !io::points_from_cloud2(*mFlat, flat) || !io::points_from_cloud2(*mLessFlat, lessFlat)) {
                ROS_ERROR("laser odometry: feature message without float x / y / z fields");
                continue;
            }
            try {
                ll->check(ll_upload_features(ll->get(), 0, (const ll_point *)sharp.data(), (int)sharp.size(), (const ll_point *)lessSharp.data(),
                                             (int)lessSharp.size(), (const ll_point *)flat.data(), (int)flat.size(),
                                             (const ll_point *)lessFlat.data(), (int)lessFlat.size()));
                if (!systemInited) systemInited = true;                       /* "Initialization finished" (:426-430) */
                else {
                    /* :439-828: 3 x { correspondences, vote when now_frame > 5, LM <= 4 iterations }, warm start in para */
                    ll->check(ll_odometry_frames(ll->get(), 0, 1, para, 3, now_frame, nullptr, para));
                    w_curr.compose(para, para + 4);                           /* :830-831 */
                }
                ll->check(ll_set_target_from_slot(ll->get(), 0));             /* the cloud swap + both kd-tree rebuilds (:882-896) */
            } catch (const lightloam::Error &e) {
                ROS_ERROR("laser odometry: %s (code %d)", e.what(), e.code);
                continue;
            }
            nav_msgs::Odometry laserOdometry;                                 /* :834-848 */
            laserOdometry.header.frame_id = "rslidar";
            laserOdometry.child_frame_id = "/laser_odom";
            laserOdometry.header.stamp = ros::Time().fromSec(timeSurfPointsLessFlat);
            laserOdometry.pose.pose.orientation.x = w_curr.q[0]; laserOdometry.pose.pose.orientation.y = w_curr.q[1];
            laserOdometry.pose.pose.orientation.z = w_curr.q[2]; laserOdometry.pose.pose.orientation.w = w_curr.q[3];
            laserOdometry.pose.pose.position.x = w_curr.t[0]; laserOdometry.pose.pose.position.y = w_curr.t[1];
            laserOdometry.pose.pose.position.z = w_curr.t[2];
            pubLaserOdometry.publish(laserOdometry);
            geometry_msgs::PoseStamped laserPose;                             /* :850-856 */
            laserPose.header = laserOdometry.header;
            laserPose.pose = laserOdometry.pose.pose;
            laserPath.header.stamp = laserOdometry.header.stamp;
            laserPath.poses.push_back(laserPose);
            laserPath.header.frame_id = "rslidar";
            pubLaserPath.publish(laserPath);
            if (frameCount % skipFrameNum == 0) {                             /* :898-918 */
                frameCount = 0;
                publish_cloud(pubLaserCloudCornerLast, lessSharp, timeSurfPointsLessFlat);
                publish_cloud(pubLaserCloudSurfLast, lessFlat, timeSurfPointsLessFlat);
                sensor_msgs::PointCloud2 laserCloudFullRes3 = *mFull;         /* the bytes pcl::toROSMsg(fromROSMsg(.)) would rebuild */
                laserCloudFullRes3.header.stamp = ros::Time().fromSec(timeSurfPointsLessFlat);
                laserCloudFullRes3.header.frame_id = "/camera";
                pubLaserCloudFullRes.publish(laserCloudFullRes3);
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > 100) ROS_WARN("odometry process over 100ms");            /* :921-922 */
            frameCount++;
            now_frame++;
        }
        rate.sleep();
    }
    return 0;
}
