// Drives ros/lightloam_laser_mapping_node.cpp -- compiled AS IT IS against the declared test doubles in
// tests/native/ros_double (this image has no ROS).  Input: <dir>/<k>.{less_sharp,less_flat,cloud}.f4 (raw float4) and
// <dir>/<k>.odom.txt (q x y z w, t x y z) per frame; the double's spin() delivers one frame's four messages, waits until the
// node's process thread has published that frame, and goes on.  Every /aft_mapped_to_init pose goes to <dir>/mapped.txt,
// every high-frequency pose to <dir>/high.txt; the topic surface, publication counts and the last tf go to stdout.
#define main laser_mapping_main
#include "../../ros/lightloam_laser_mapping_node.cpp"
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

static void dump(std::ofstream &o, const nav_msgs::Odometry &m)
{
    const auto &p = m.pose.pose;
    o << p.orientation.x << " " << p.orientation.y << " " << p.orientation.z << " " << p.orientation.w << " " << p.position.x << " "
      << p.position.y << " " << p.position.z << " " << m.header.stamp.sec << " " << m.header.frame_id << " " << m.child_frame_id << "\n";
}

int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    const std::string dir = argv[1]; const int nframes = std::atoi(argv[2]);
    auto &D = ros::Double::get();
    D.params_s["RESULT_PATH"] = dir + "/result.txt";
    int next = 0;
    std::ofstream mapped(dir + "/mapped.txt"), high(dir + "/high.txt");
    mapped.precision(17); high.precision(17);
    D.on_spin = [&]() {
        if (next >= nframes) return false;
        const char *names[3] = {"less_sharp", "less_flat", "cloud"};
        const char *topics[3] = {"/laser_cloud_corner_last", "/laser_cloud_surf_last", "/velodyne_cloud_3"};
        for (int k = 0; k < 3; ++k) {
            auto msg = std::make_shared<sensor_msgs::PointCloud2>();
            lightloam::ros_io::cloud2_from_points(read_f4(dir + "/" + std::to_string(next) + "." + names[k] + ".f4"), *msg);
            msg->header.stamp.sec = 100 + next; msg->header.frame_id = "/camera";
            D.callbacks[topics[k]](msg);
        }
        auto od = std::make_shared<nav_msgs::Odometry>();
        std::ifstream f(dir + "/" + std::to_string(next) + ".odom.txt");
        auto &P = od->pose.pose;
        f >> P.orientation.x >> P.orientation.y >> P.orientation.z >> P.orientation.w >> P.position.x >> P.position.y >> P.position.z;
        od->header.stamp.sec = 100 + next; od->header.frame_id = "rslidar"; od->child_frame_id = "/laser_odom";
        D.callbacks["/laser_odom_to_init"](od);
        dump(high, *std::static_pointer_cast<nav_msgs::Odometry>(D.latest("/aft_mapped_to_init_high_frec")));
        ++next;
        for (int spins = 0; D.count("/aft_mapped_to_init") < next && spins < 30000; ++spins) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (D.count("/aft_mapped_to_init") < next) return false;                       // the node did not take the frame
        dump(mapped, *std::static_pointer_cast<nav_msgs::Odometry>(D.latest("/aft_mapped_to_init")));
        return true;
    };
    char *av[] = {argv[0], nullptr};
    const int rc = laser_mapping_main(1, av);
    std::cout << "rc " << rc << " frames " << next << "\nsubscribed";
    for (auto &s : D.subscribed) std::cout << " " << s.first << ":" << s.second;
    std::cout << "\nadvertised";
    for (auto &s : D.advertised) std::cout << " " << s.first << ":" << s.second;
    std::cout << "\npublished";
    for (auto &s : D.published) std::cout << " " << s.first << ":" << s.second;
    auto path = std::static_pointer_cast<nav_msgs::Path>(D.latest("/aft_mapped_path"));
    auto sur = std::static_pointer_cast<sensor_msgs::PointCloud2>(D.latest("/laser_cloud_surround"));
    auto map = std::static_pointer_cast<sensor_msgs::PointCloud2>(D.latest("/laser_cloud_map"));
    auto regd = std::static_pointer_cast<sensor_msgs::PointCloud2>(D.latest("/velodyne_cloud_registered"));
    auto &S = tf::Sent::get();
    std::cout << "\npath_poses " << (path ? path->poses.size() : 0) << " surround_pts " << (sur ? sur->width : 0) << " map_pts " << (map ? map->width : 0)
              << " registered_pts " << (regd ? regd->width : 0) << " registered_frame " << (regd ? regd->header.frame_id : "") << "\ntf " << S.count << " "
              << S.frame_id << " " << S.child_frame_id << "\n";
    std::ofstream tfo(dir + "/tf.txt"); tfo.precision(17);
    tfo << S.q[0] << " " << S.q[1] << " " << S.q[2] << " " << S.q[3] << " " << S.t[0] << " " << S.t[1] << " " << S.t[2] << "\n";
    if (regd) {
        std::vector<lightloam::PointXYZI> pts; lightloam::ros_io::points_from_cloud2(*regd, pts);
        std::ofstream o(dir + "/registered.f4", std::ios::binary); o.write((const char *)pts.data(), sizeof(pts[0]) * pts.size());
    }
    if (sur) {
        std::vector<lightloam::PointXYZI> pts; lightloam::ros_io::points_from_cloud2(*sur, pts);
        std::ofstream o(dir + "/surround.f4", std::ios::binary); o.write((const char *)pts.data(), sizeof(pts[0]) * pts.size());
    }
    return rc;
}
