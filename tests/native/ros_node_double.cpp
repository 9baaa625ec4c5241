// Drives ros/lightloam_scan_registration_node.cpp -- compiled AS IT IS against the declared test doubles of roscpp /
// sensor_msgs in tests/native/ros_double (this image has no ROS) -- and prints what a test checks:
//   ros_node_double layout            no GPU: message packing round trip + the node's topic surface up to the context
//   ros_node_double run <scan.bin>    GPU: one KITTI-format scan through main() + the callback; dumps the five published
//                                     clouds as raw float4 files next to the input (<scan.bin>.<topic>.f4)
#define main scan_registration_main
#include "../../ros/lightloam_scan_registration_node.cpp"
#undef main
#include <fstream>
#include <iostream>

static sensor_msgs::PointCloud2 make_input(const std::vector<float> &xyzi, bool odd_layout)
{
    // a velodyne-style message: x y z intensity ring(uint16) time -- point_step 22 when odd_layout, else 16
    sensor_msgs::PointCloud2 m;
    const uint32_t step = odd_layout ? 22 : 16;
    const char *names[4] = {"x", "y", "z", "intensity"};
    for (int k = 0; k < 4; ++k) { sensor_msgs::PointField f; f.name = names[k]; f.offset = 4 * k; f.datatype = 7; f.count = 1; m.fields.push_back(f); }
    if (odd_layout) { sensor_msgs::PointField f; f.name = "ring"; f.offset = 16; f.datatype = 4; f.count = 1; m.fields.push_back(f); }
    const size_t n = xyzi.size() / 4;
    m.height = 1; m.width = (uint32_t)n; m.point_step = step; m.row_step = step * (uint32_t)n; m.is_dense = 0;
    m.data.assign((size_t)step * n, 0xAB);
    for (size_t i = 0; i < n; ++i) std::memcpy(&m.data[i * step], &xyzi[4 * i], 16);
    m.header.stamp.sec = 1234; m.header.stamp.nsec = 5678; m.header.frame_id = "rslidar";
    return m;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "layout";
    auto &D = ros::Double::get();
    if (mode == "layout") {
        std::vector<lightloam::PointXYZI> pts = {{1.f, 2.f, 3.f, 4.25f}, {-5.f, 6.5f, -7.f, 63.0999f}, {0.f, 0.f, 0.f, 0.f}};
        sensor_msgs::PointCloud2 m;
        lightloam::ros_io::cloud2_from_points(pts, m);
        std::cout << "fields";
        for (auto &f : m.fields) std::cout << " " << f.name << "@" << f.offset << ":" << (int)f.datatype << "x" << f.count;
        std::cout << "\nstep " << m.point_step << " row " << m.row_step << " h " << m.height << " w " << m.width << " dense " << (int)m.is_dense
                  << " be " << (int)m.is_bigendian << " bytes " << m.data.size() << "\n";
        std::vector<lightloam::PointXYZI> back;
        const bool ok = lightloam::ros_io::points_from_cloud2(m, back);
        std::cout << "roundtrip " << (ok && back.size() == pts.size() && std::memcmp(back.data(), pts.data(), sizeof(pts[0]) * pts.size()) == 0) << "\n";
        std::vector<float> in = {1, 2, 3, 9, 4, 5, 6, 9};
        std::vector<float> xyz;
        const bool ok2 = lightloam::ros_io::xyz_from_cloud2(make_input(in, true), xyz);
        std::cout << "odd_layout " << (ok2 && xyz.size() == 8 && xyz[0] == 1 && xyz[2] == 3 && xyz[3] == 0 && xyz[4] == 4 && xyz[6] == 6) << "\n";
        sensor_msgs::PointCloud2 bad = make_input(in, false); bad.fields[1].name = "why";
        std::cout << "missing_field_rejected " << !lightloam::ros_io::xyz_from_cloud2(bad, xyz) << "\n";
        {   // organised cloud: 2 rows x 2 points, rows padded to 40 bytes, fields in the order z y x
            sensor_msgs::PointCloud2 o;
            const char *nm[3] = {"z", "y", "x"};
            for (int k = 0; k < 3; ++k) { sensor_msgs::PointField f; f.name = nm[k]; f.offset = 4 * k; f.datatype = 7; f.count = 1; o.fields.push_back(f); }
            o.height = 2; o.width = 2; o.point_step = 12; o.row_step = 40; o.data.assign(80, 0);
            const float v[4][3] = {{3, 2, 1}, {6, 5, 4}, {9, 8, 7}, {12, 11, 10}};     // stored z y x
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) std::memcpy(&o.data[r * 40 + c * 12], v[r * 2 + c], 12);
            std::vector<float> q;
            const bool ok3 = lightloam::ros_io::xyz_from_cloud2(o, q);
            std::cout << "organised " << (ok3 && q.size() == 16 && q[0] == 1 && q[1] == 2 && q[2] == 3 && q[12] == 10 && q[13] == 11 && q[14] == 12) << "\n";
            o.data.resize(60);                                                           // second row cut short
            std::cout << "short_data_rejected " << !lightloam::ros_io::xyz_from_cloud2(o, q) << "\n";
            o.data.resize(80); o.is_bigendian = 1;
            std::cout << "bigendian_rejected " << !lightloam::ros_io::xyz_from_cloud2(o, q) << "\n";
            sensor_msgs::PointCloud2 e; lightloam::ros_io::cloud2_from_points({}, e);     // an empty cloud is a valid message
            std::vector<lightloam::PointXYZI> ep;
            std::cout << "empty " << (lightloam::ros_io::points_from_cloud2(e, ep) && ep.empty() && e.width == 0 && e.data.empty()) << "\n";
        }
        D.params_i["scan_line"] = 48;                       // not 16 / 32 / 64: main() returns 0 before it needs a device (:447-451)
        int ac = 1; char *av[] = {argv[0], nullptr};
        (void)ac; (void)av;
        std::cout << "bad_scan_line_exit " << scan_registration_main(1, av) << " advertised " << D.advertised.size() << "\n";
        return 0;
    }
    if (mode == "run" && argc > 2) {
        std::ifstream f(argv[2], std::ios::binary);
        std::vector<float> xyzi((std::istreambuf_iterator<char>(f)), {});   // placeholder, replaced below
        f.clear(); f.seekg(0, std::ios::end); const size_t bytes = (size_t)f.tellg(); f.seekg(0);
        xyzi.resize(bytes / 4); f.read((char *)xyzi.data(), bytes);
        D.params_i["scan_line"] = argc > 3 ? std::atoi(argv[3]) : 64;
        D.params_d["minimum_range"] = argc > 4 ? std::atof(argv[4]) : 5.0;
        char *av[] = {argv[0], nullptr};
        const int rc = scan_registration_main(1, av);
        if (rc != 0 || !D.callback) { std::cout << "main rc " << rc << "\n"; return 2; }
        std::cout << "subscribed";
        for (auto &s : D.subscribed) std::cout << " " << s.first << ":" << s.second;
        std::cout << "\nadvertised";
        for (auto &s : D.advertised) std::cout << " " << s.first << ":" << s.second;
        std::cout << "\n";
        // main() returned (spin() of the double returns at once) and released the context: make one for the callback
        g_ll.reset(new lightloam::Context(D.params_i["scan_line"], 2, 0, D.params_d["minimum_range"]));
        auto msg = std::make_shared<sensor_msgs::PointCloud2>(make_input(xyzi, true));
        D.callback(msg);
        for (auto &kv : D.last) {
            auto m = std::static_pointer_cast<sensor_msgs::PointCloud2>(kv.second);
            std::vector<lightloam::PointXYZI> pts;
            lightloam::ros_io::points_from_cloud2(*m, pts);
            std::string name = kv.first; for (auto &c : name) if (c == '/') c = '_';
            std::ofstream o(std::string(argv[2]) + "." + name + ".f4", std::ios::binary);
            o.write((const char *)pts.data(), sizeof(pts[0]) * pts.size());
            std::cout << kv.first << " n " << pts.size() << " published " << D.published[kv.first] << " stamp " << m->header.stamp.sec << "."
                      << m->header.stamp.nsec << " frame " << m->header.frame_id << "\n";
        }
        g_ll.reset();
        return 0;
    }
    return 1;
}
