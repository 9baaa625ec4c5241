// Drives ros/lightloam_laser_odometry_node.cpp -- compiled AS IT IS against the declared test doubles in
// tests/native/ros_double (this image has no ROS).  Input: a directory of per-frame feature files written by the test
// (<dir>/<k>.<name>.f4 as raw float4, name in sharp / less_sharp / flat / less_flat / cloud); the double's spinOnce()
// delivers one frame's five messages per turn; every published odometry pose is appended to <dir>/odom.txt, the topic
// surface and publication counts go to stdout.
#define main laser_odometry_main
#include "../../ros/lightloam_laser_odometry_node.cpp"
#undef main
#include <fstream>
#include <iostream>

static std::vector<lightloam::PointXYZI> read_f4(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<lightloam::PointXYZI> v;
    if (!f) return v;
    const size_t bytes = (size_t)f.tellg(); f.seekg(0);
    v.resize(bytes / sizeof(lightloam::PointXYZI)); f.read((char *)v.data(), bytes);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 1;
    const std::string dir = argv[1]; const int nframes = std::atoi(argv[2]);
    auto &D = ros::Double::get();
    D.params_i["scan_line"] = std::atoi(argv[3]);
    int next = 0, odom_seen = 0;
    std::ofstream odom(dir + "/odom.txt");
    odom.precision(17);
    D.on_spin = [&]() {
        // what the previous turn published
        if (D.published["/laser_odom_to_init"] > odom_seen) {
            odom_seen = D.published["/laser_odom_to_init"];
            auto m = std::static_pointer_cast<nav_msgs::Odometry>(D.last["/laser_odom_to_init"]);
            const auto &p = m->pose.pose;
            odom << p.orientation.x << " " << p.orientation.y << " " << p.orientation.z << " " << p.orientation.w << " " << p.position.x << " "
                 << p.position.y << " " << p.position.z << " " << m->header.stamp.sec << " " << m->header.frame_id << " " << m->child_frame_id << "\n";
        }
        if (next >= nframes) return false;                    // ros::ok() turns false: the node's loop ends
        const char *names[5] = {"sharp", "less_sharp", "flat", "less_flat", "cloud"};
        const char *topics[5] = {"/laser_cloud_sharp", "/laser_cloud_less_sharp", "/laser_cloud_flat", "/laser_cloud_less_flat", "/velodyne_cloud_2"};
        for (int k = 0; k < 5; ++k) {
            auto msg = std::make_shared<sensor_msgs::PointCloud2>();
            lightloam::ros_io::cloud2_from_points(read_f4(dir + "/" + std::to_string(next) + "." + names[k] + ".f4"), *msg);
            msg->header.stamp.sec = 100 + next; msg->header.stamp.nsec = 0; msg->header.frame_id = "rslidar";
            D.callbacks[topics[k]](msg);
        }
        ++next;
        return true;
    };
    char *av[] = {argv[0], nullptr};
    const int rc = laser_odometry_main(1, av);
    std::cout << "rc " << rc << "\nsubscribed";
    for (auto &s : D.subscribed) std::cout << " " << s.first << ":" << s.second;
    std::cout << "\nadvertised";
    for (auto &s : D.advertised) std::cout << " " << s.first << ":" << s.second;
    std::cout << "\npublished";
    for (auto &s : D.published) std::cout << " " << s.first << ":" << s.second;
    auto path = std::static_pointer_cast<nav_msgs::Path>(D.last["/laser_odom_path"]);
    auto cl = std::static_pointer_cast<sensor_msgs::PointCloud2>(D.last["/laser_cloud_corner_last"]);
    std::cout << "\npath_poses " << (path ? path->poses.size() : 0) << " path_frame " << (path ? path->header.frame_id : "") << " cloud_frame "
              << (cl ? cl->header.frame_id : "") << " cloud_stamp " << (cl ? cl->header.stamp.sec : 0) << "\n";
    return rc;
}
