/*
 * lightloam_laser_mapping_node.cpp -- the reference's `alaserMapping` node with its topic surface unchanged; the per-frame
 * body (cube map, scan-to-map optimisation, map update: laserMapping.cpp:1584-2165) runs on the device through
 * lightloam::LaserMapping (ll_cubemap_*).
 *
 * Mirrors /root/reference src/laserMapping.cpp:
 *   parameters   mapping_line_resolution (0.4), mapping_plane_resolution (0.8), RESULT_PATH (" ")           :2361-2365
 *   subscribes   /laser_cloud_corner_last, /laser_cloud_surf_last, /velodyne_cloud_3 (PointCloud2),
 *                /laser_odom_to_init (nav_msgs/Odometry), queue 100 each                                     :2369-2375
 *   advertises   /laser_cloud_surround, /laser_cloud_map, /velodyne_cloud_registered (PointCloud2),
 *                /aft_mapped_to_init, /aft_mapped_to_init_high_frec (nav_msgs/Odometry), /aft_mapped_path    :2377-2387
 *   odometry callback: queue + the high-frequency republish of the odometry pose moved by the last map correction,
 *                with the reference's roll / yaw + pi/2 and permuted quaternion fields                        :168-247
 *   process thread: a frame is taken when all four queues hold a message; odometry / surf / full messages older than
 *                the corner message are dropped, stamps must then be equal, further corner messages are dropped
 *                ("real time")                                                                                :1507-1577
 *   per frame    transformAssociateToMap, the body on the device, transformUpdate                             :1581, :2101
 *   publishes    every 5th frame the clouds of the (up to) 5 x 5 x 3 cubes around the sensor, every 20th frame all
 *                4851 cubes, the full-resolution scan in map coordinates, odomAftMapped + path ("rslidar" ->
 *                "/aft_mapped"), the same as a tf transform, and one line of RESULT_PATH                      :2171-2346
 * Build inside the catkin package: roscpp, sensor_msgs, nav_msgs, geometry_msgs, tf; no PCL, Eigen or Ceres.
 * This image has no ROS: tests/native/ros_mapping_double.cpp compiles this file against declared test doubles.
 */
#include <chrono>
#include <cmath>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <geometry_msgs/PoseStamped.h>
#include <tf/transform_datatypes.h>
#include <tf/transform_broadcaster.h>

#include "lightloam_host.hpp"
#include "lightloam_ros.hpp"

namespace {

const int laserCloudWidth = 21, laserCloudHeight = 21, laserCloudDepth = 11;                       /* :48-50 */
const int laserCloudNum = laserCloudWidth * laserCloudHeight * laserCloudDepth;                     /* :53 */

std::queue<sensor_msgs::PointCloud2ConstPtr> cornerLastBuf, surfLastBuf, fullResBuf;                /* :94-96 */
std::queue<nav_msgs::Odometry::ConstPtr> odometryBuf;                                               /* :97 */
std::mutex mBuf;                                                                                    /* :98 */
std::mutex mPose;                                                                                   /* q_wmap_wodom / t_wmap_wodom between transformUpdate and the odometry callback */
std::unique_ptr<lightloam::Context> g_ll;
std::unique_ptr<lightloam::LaserMapping> g_lm;
std::string RESULT_PATH;
ros::Publisher pubLaserCloudSurround, pubLaserCloudMap, pubLaserCloudFullRes, pubOdomAftMapped, pubOdomAftMappedHighFrec, pubLaserAfterMappedPath;
nav_msgs::Path laserAfterMappedPath;

void qmul(const double a[4], const double b[4], double o[4])
{
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1]; o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3]; o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void qrot(const double q[4], const double v[3], double o[3])           /* Eigen's quaternion * vector */
{
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    double uvx = uy * v[2] - uz * v[1], uvy = uz * v[0] - ux * v[2], uvz = ux * v[1] - uy * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    o[0] = v[0] + w * uvx + (uy * uvz - uz * uvy); o[1] = v[1] + w * uvy + (uz * uvx - ux * uvz); o[2] = v[2] + w * uvz + (ux * uvy - uy * uvx);
}

void laserCloudCornerLastHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); cornerLastBuf.push(m); }   /* :147-152 */
void laserCloudSurfLastHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); surfLastBuf.push(m); }       /* :154-159 */
void laserCloudFullResHandler(const sensor_msgs::PointCloud2ConstPtr &m) { std::lock_guard<std::mutex> l(mBuf); fullResBuf.push(m); }         /* :161-166 */

void laserOdometryHandler(const nav_msgs::Odometry::ConstPtr &laserOdometry)                        /* :168-247 */
{
    double q_wmap_wodom[4], t_wmap_wodom[3];
    { std::lock_guard<std::mutex> l(mBuf); odometryBuf.push(laserOdometry); }
    {
        std::lock_guard<std::mutex> l(mPose);                                                       /* never held across device work */
        for (int k = 0; k < 4; ++k) q_wmap_wodom[k] = g_lm->q_wmap_wodom[k];
        for (int k = 0; k < 3; ++k) t_wmap_wodom[k] = g_lm->t_wmap_wodom[k];
    }
    const auto &P = laserOdometry->pose.pose;
    const double q_wodom_curr[4] = {P.orientation.x, P.orientation.y, P.orientation.z, P.orientation.w};
    const double t_wodom_curr[3] = {P.position.x, P.position.y, P.position.z};
    double q[4], t[3];
    qmul(q_wmap_wodom, q_wodom_curr, q);                                                            /* q_w_curr, x y z w */
    qrot(q_wmap_wodom, t_wodom_curr, t);
    for (int k = 0; k < 3; ++k) t[k] += t_wmap_wodom[k];
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    double roll = std::atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
    const double sinp = 2 * (w * y - z * x);
    const double pitch = (std::fabs(sinp) >= 1) ? std::copysign(M_PI / 2, sinp) : std::asin(sinp);
    double yaw = std::atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
    roll = roll + M_PI / 2;                                                                         /* :211-212 */
    yaw = yaw + M_PI / 2;
    const double cy = std::cos(yaw * 0.5), sy = std::sin(yaw * 0.5), cp = std::cos(pitch * 0.5), sp = std::sin(pitch * 0.5);
    const double cr = std::cos(roll * 0.5), sr = std::sin(roll * 0.5);
    const double aw = cy * cp * cr + sy * sp * sr, ax = cy * cp * sr - sy * sp * cr, ay = sy * cp * sr + cy * sp * cr, az = sy * cp * cr - cy * sp * sr;
    nav_msgs::Odometry odomAftMapped;
    odomAftMapped.header.frame_id = "rslidar";
    odomAftMapped.child_frame_id = "/aft_mapped";
    odomAftMapped.header.stamp = laserOdometry->header.stamp;
    odomAftMapped.pose.pose.orientation.x = ay;                                                     /* the permutation of :237-240 */
    odomAftMapped.pose.pose.orientation.y = -ax;
    odomAftMapped.pose.pose.orientation.z = aw;
    odomAftMapped.pose.pose.orientation.w = -az;
    odomAftMapped.pose.pose.position.x = t[0]; odomAftMapped.pose.pose.position.y = t[1]; odomAftMapped.pose.pose.position.z = t[2];
    pubOdomAftMappedHighFrec.publish(odomAftMapped);
}

/* the clouds of the cubes `ind`, corner then surf per cube, like the `+=` loops of :2173-2178 / :2191-2195 */
void gather_cubes(const std::vector<int> &ind, std::vector<lightloam::PointXYZI> &out)
{
    out.clear();
    std::vector<lightloam::PointXYZI> buf;
    for (int cube : ind)
        for (int surf = 0; surf < 2; ++surf) {
            int n = 0;
            if (ll_cubemap_download_cube(g_lm->get(), surf, cube, nullptr, 1 << 30, &n) != LL_OK || n == 0) continue;
            buf.resize((size_t)n);
            if (ll_cubemap_download_cube(g_lm->get(), surf, cube, (ll_point *)buf.data(), n, &n) != LL_OK) continue;
            out.insert(out.end(), buf.begin(), buf.end());
        }
}

void publish_cloud(ros::Publisher &pub, const std::vector<lightloam::PointXYZI> &pts, double stamp)
{
    sensor_msgs::PointCloud2 out;
    lightloam::ros_io::cloud2_from_points(pts, out);
    out.header.stamp = ros::Time().fromSec(stamp);
    out.header.frame_id = "rslidar";
    pub.publish(out);
}

void process()                                                                                      /* :1502-2354 */
{
    int frameCount = 0;
    std::unique_ptr<lightloam::TrajectoryWriter> traj;
    while (ros::ok()) {
        for (;;) {
            sensor_msgs::PointCloud2ConstPtr mCorner, mSurf, mFull;
            nav_msgs::Odometry::ConstPtr mOdom;
            {
                std::lock_guard<std::mutex> l(mBuf);
                if (cornerLastBuf.empty() || surfLastBuf.empty() || fullResBuf.empty() || odometryBuf.empty()) break;
                const double tc = cornerLastBuf.front()->header.stamp.toSec();
                while (!odometryBuf.empty() && odometryBuf.front()->header.stamp.toSec() < tc) odometryBuf.pop();   /* :1511-1517 */
                if (odometryBuf.empty()) break;
                while (!surfLastBuf.empty() && surfLastBuf.front()->header.stamp.toSec() < tc) surfLastBuf.pop();   /* :1519-1525 */
                if (surfLastBuf.empty()) break;
                while (!fullResBuf.empty() && fullResBuf.front()->header.stamp.toSec() < tc) fullResBuf.pop();      /* :1527-1533 */
                if (fullResBuf.empty()) break;
                const double to = odometryBuf.front()->header.stamp.toSec();
                if (tc != to || surfLastBuf.front()->header.stamp.toSec() != to || fullResBuf.front()->header.stamp.toSec() != to) break;   /* unsync (:1540-1548) */
                mCorner = cornerLastBuf.front(); cornerLastBuf.pop();
                mSurf = surfLastBuf.front(); surfLastBuf.pop();
                mFull = fullResBuf.front(); fullResBuf.pop();
                mOdom = odometryBuf.front(); odometryBuf.pop();
                while (!cornerLastBuf.empty()) cornerLastBuf.pop();                                 /* drop frames for real-time performance (:1572-1576) */
            }
            const double timeLaserOdometry = mOdom->header.stamp.toSec();
            namespace io = lightloam::ros_io;
            std::vector<lightloam::PointXYZI> cornerLast, surfLast, fullRes;
            if (!io::points_from_cloud2(*mCorner, cornerLast) || !io::points_from_cloud2(*mSurf, surfLast) || !io::points_from_cloud2(*mFull, fullRes)) {
                ROS_ERROR("laser mapping: cloud message without float x / y / z fields");
                continue;
            }
            const auto &P = mOdom->pose.pose;
            const double q_wodom_curr[4] = {P.orientation.x, P.orientation.y, P.orientation.z, P.orientation.w};
            const double t_wodom_curr[3] = {P.position.x, P.position.y, P.position.z};
            double guess_t[3];
            int cen[3] = {0, 0, 0};
            try {
                /* mBuf is held around the queue pops only, as in the reference (:1505-1578): the subscriber callbacks -- the
                 * high-frequency republish of laserOdometryHandler among them -- keep running during the device work.
                 * q_wmap_wodom / t_wmap_wodom are written by this thread alone (transformUpdate) and read by that callback:
                 * their own small mutex covers the write */
                g_lm->transformAssociateToMap(q_wodom_curr, t_wodom_curr);                          /* :1581 */
                for (int k = 0; k < 3; ++k) guess_t[k] = g_lm->parameters[4 + k];
                g_lm->process(cornerLast, surfLast);                                                /* :1584-2165 */
                { std::lock_guard<std::mutex> l(mPose); g_lm->transformUpdate(q_wodom_curr, t_wodom_curr); }   /* :2101 */
                if (ll_cubemap_info(g_lm->get(), cen, nullptr) != LL_OK) throw lightloam::Error(LL_ERR_STATE, "cube map info");
            } catch (const lightloam::Error &e) {
                ROS_ERROR("laser mapping: %s (code %d)", e.what(), e.code);
                continue;
            }
            const double *q_w_curr = g_lm->parameters, *t_w_curr = g_lm->parameters + 4;
            if (frameCount % 5 == 0) {                                                              /* :2171-2186 */
                /* laserCloudSurroundInd: the cubes centerCube +-2, +-2, +-1 inside the grid, in the loop order of :1784-1801;
                 * the centre cube comes from the pose guess of :1581, after the shift loops moved it with the grid */
                int c[3];
                for (int k = 0; k < 3; ++k) {
                    c[k] = int((guess_t[k] + 25.0) / 50.0) + cen[k];                                /* :1584-1593 */
                    if (guess_t[k] + 25.0 < 0) c[k]--;
                }
                std::vector<int> ind;
                for (int i = c[0] - 2; i <= c[0] + 2; i++)
                    for (int j = c[1] - 2; j <= c[1] + 2; j++)
                        for (int k = c[2] - 1; k <= c[2] + 1; k++)
                            if (i >= 0 && i < laserCloudWidth && j >= 0 && j < laserCloudHeight && k >= 0 && k < laserCloudDepth)
                                ind.push_back(i + laserCloudWidth * j + laserCloudWidth * laserCloudHeight * k);
                std::vector<lightloam::PointXYZI> surround;
                gather_cubes(ind, surround);
                publish_cloud(pubLaserCloudSurround, surround, timeLaserOdometry);
            }
            if (frameCount % 20 == 0) {                                                             /* :2188-2201 */
                std::vector<int> all((size_t)laserCloudNum);
                for (int i = 0; i < laserCloudNum; ++i) all[(size_t)i] = i;
                std::vector<lightloam::PointXYZI> map;
                gather_cubes(all, map);
                publish_cloud(pubLaserCloudMap, map, timeLaserOdometry);
            }
            for (auto &p : fullRes) {                                                               /* pointAssociateToMap (:125-133), :2203-2207 */
                const double v[3] = {p.x, p.y, p.z};
                double o[3];
                qrot(q_w_curr, v, o);
                p.x = (float)(o[0] + t_w_curr[0]); p.y = (float)(o[1] + t_w_curr[1]); p.z = (float)(o[2] + t_w_curr[2]);
            }
            publish_cloud(pubLaserCloudFullRes, fullRes, timeLaserOdometry);                        /* :2209-2213 */

            nav_msgs::Odometry odomAftMapped;                                                       /* :2226-2238 */
            odomAftMapped.header.frame_id = "rslidar";
            odomAftMapped.child_frame_id = "/aft_mapped";
            odomAftMapped.header.stamp = ros::Time().fromSec(timeLaserOdometry);
            odomAftMapped.pose.pose.orientation.x = q_w_curr[0]; odomAftMapped.pose.pose.orientation.y = q_w_curr[1];
            odomAftMapped.pose.pose.orientation.z = q_w_curr[2]; odomAftMapped.pose.pose.orientation.w = q_w_curr[3];
            odomAftMapped.pose.pose.position.x = t_w_curr[0]; odomAftMapped.pose.pose.position.y = t_w_curr[1];
            odomAftMapped.pose.pose.position.z = t_w_curr[2];
            pubOdomAftMapped.publish(odomAftMapped);
            {                                                                                       /* RESULT_PATH line (:2240-2325) */
                if (!traj) traj.reset(new lightloam::TrajectoryWriter(RESULT_PATH));
                lightloam::WorldPose wp;
                for (int k = 0; k < 4; ++k) wp.q[k] = q_w_curr[k];
                for (int k = 0; k < 3; ++k) wp.t[k] = t_w_curr[k];
                traj->append(wp);
            }
            geometry_msgs::PoseStamped laserAfterMappedPose;                                        /* :2326-2332 */
            laserAfterMappedPose.header = odomAftMapped.header;
            laserAfterMappedPose.pose = odomAftMapped.pose.pose;
            laserAfterMappedPath.header.stamp = odomAftMapped.header.stamp;
            laserAfterMappedPath.header.frame_id = "rslidar";
            laserAfterMappedPath.poses.push_back(laserAfterMappedPose);
            pubLaserAfterMappedPath.publish(laserAfterMappedPath);
            static tf::TransformBroadcaster br;                                                     /* :2334-2346 */
            tf::Transform transform;
            tf::Quaternion q;
            transform.setOrigin(tf::Vector3(t_w_curr[0], t_w_curr[1], t_w_curr[2]));
            q.setW(q_w_curr[3]); q.setX(q_w_curr[0]); q.setY(q_w_curr[1]); q.setZ(q_w_curr[2]);
            transform.setRotation(q);
            br.sendTransform(tf::StampedTransform(transform, odomAftMapped.header.stamp, "rslidar", "/aft_mapped"));
            frameCount++;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(2));                                  /* :2351-2352 */
    }
}

}  // namespace

int main(int argc, char **argv)
{
    ros::init(argc, argv, "laserMapping");
    ros::NodeHandle nh;
    float lineRes = 0, planeRes = 0;
    nh.param<float>("mapping_line_resolution", lineRes, 0.4f);                                      /* :2363 */
    nh.param<float>("mapping_plane_resolution", planeRes, 0.8f);                                    /* :2364 */
    nh.param("RESULT_PATH", RESULT_PATH, std::string(" "));                                         /* :2365 */
    try {
        g_ll.reset(new lightloam::Context(64, /*batch*/ 1));                                        /* the map stage only needs the device + stream */
        g_lm.reset(new lightloam::LaserMapping(*g_ll, lineRes, planeRes));
    } catch (const lightloam::Error &e) {
        ROS_ERROR("laser mapping: no MI355X context: %s (code %d)", e.what(), e.code);              /* there is no CPU fallback */
        return 1;
    }
    ros::Subscriber subLaserCloudCornerLast = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 100, laserCloudCornerLastHandler);
    ros::Subscriber subLaserCloudSurfLast = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 100, laserCloudSurfLastHandler);
    ros::Subscriber subLaserOdometry = nh.subscribe<nav_msgs::Odometry>("/laser_odom_to_init", 100, laserOdometryHandler);
    ros::Subscriber subLaserCloudFullRes = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_cloud_3", 100, laserCloudFullResHandler);
    pubLaserCloudSurround = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surround", 100);
    pubLaserCloudMap = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_map", 100);
    pubLaserCloudFullRes = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_registered", 100);
    pubOdomAftMapped = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init", 100);
    pubOdomAftMappedHighFrec = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init_high_frec", 100);
    pubLaserAfterMappedPath = nh.advertise<nav_msgs::Path>("/aft_mapped_path", 100);
    std::thread mapping_process{process};                                                           /* :2395 */
    ros::spin();                                                                                    /* :2397 */
    mapping_process.join();
    g_lm.reset(); g_ll.reset();
    return 0;
}
