// Scan-to-scan LiDAR odometry over a directory of KITTI-format velodyne scans (*.bin, float32 x,y,z,reflectance),
// the way the reference's kittiHelper -> scanRegistration -> laserOdometry chain would process them, writing the
// reference's trajectory-file format (laserMapping.cpp:2284-2325: 12 values of H_init^-1 * H per frame).
//
//   ll_odometry_kitti <scan_dir> <result_path> [scan_line = 64] [first-frame forward guess in metres = 0] [mapping = 0] [max_ring_points = 0]
//
// max_ring_points: capacity of one scan line (0 = library default 2304; KITTI / HDL-64E data needs 4608, see lightloam_host.hpp).
//
// mapping = 1 adds the third node: every frame's odometry pose goes through laserMapping's scan-to-map refinement
// (lightloam::LaserMapping, laserMapping.cpp:1581-2165) and the written trajectory is q_w_curr / t_w_curr, which is what
// the reference's result file holds (:2284-2325).  mapping = G >= 2 runs that node with the map split over G ranks
// (lightloam::LaserMapping::process_tile_parallel; G threads stand in for G processes) and writes the same file.
//
// Build:  g++ -O2 -std=c++14 -pthread -I include tools/ll_odometry_kitti.cpp -L light-loam_amd -llightloam_hip -o ll_odometry_kitti
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lightloam_host.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: ll_odometry_kitti <scan_dir> <result_path> [scan_line] [first guess tx] [mapping] [max_ring_points]\n"; return 2; }
    const std::string dir = argv[1], result = argv[2];
    const int scan_line = argc > 3 ? std::atoi(argv[3]) : 64;
    const double tx0 = argc > 4 ? std::atof(argv[4]) : 0.0;
    const bool mapping = argc > 5 && std::atoi(argv[5]) != 0;
    const int max_ring_points = argc > 6 ? std::atoi(argv[6]) : 0;
    std::vector<std::string> files;
    if (DIR *d = opendir(dir.c_str())) {
        while (dirent *e = readdir(d)) {
            const std::string n = e->d_name;
            if (n.size() > 4 && n.substr(n.size() - 4) == ".bin") files.push_back(dir + "/" + n);
        }
        closedir(d);
    }
    std::sort(files.begin(), files.end());
    if (files.size() < 2) { std::cerr << "need at least two .bin scans in " << dir << "\n"; return 2; }
    try {
        using namespace lightloam;
        const int n = (int)files.size();
        /* resident layout: x, y, z packed (12 bytes per point).  A .bin record is (x, y, z, reflectance); the reference drops the reflectance
         * when it converts to PointXYZ (kittiHelper.cpp:128-148 -> scanRegistration.cpp:105-106), and so does ll_upload_scan, on the host:
         * a quarter fewer bytes cross PCIe and the organise stage reads 12 instead of 16 bytes per point */
        Context ctx(scan_line, n, 0, -1.0, -24.9f, 2.0f, 0, max_ring_points, 3);
        for (int k = 0; k < n; ++k) {
            const std::vector<float> pts = read_lidar_data(files[k]);
            ctx.check(ll_upload_scan(ctx.get(), k, pts.data(), 4, (int)(pts.size() / 4)));
        }
        ctx.check(ll_extract_batch(ctx.get(), 0, n));                       // scanRegistration for every scan
        for (int k = 0; k < n; ++k) {                                        // a scan the registration refused must not become a silent hole in the trajectory
            ll_scan_info info;
            ctx.check(ll_get_scan_info(ctx.get(), k, &info));
            if (info.status == LL_ERR_CAPACITY) {
                std::cerr << files[k] << ": a scan line holds " << info.max_ring << " points, beyond the ring capacity " << ctx.params().max_ring_points
                          << " -- pass a larger max_ring_points (KITTI / HDL-64E: 4608)\n";
                return 1;
            }
            if (info.status != LL_OK && info.status != LL_ERR_EMPTY) { std::cerr << files[k] << ": scan registration failed with status " << info.status << "\n"; return 1; }
        }
        ctx.check(ll_set_target_from_slot(ctx.get(), 0));                   // frame 0 only initialises (laserOdometry.cpp:427-431)
        const double pose0[7] = {0, 0, 0, 1, tx0, 0, 0};
        const std::vector<double> rel = odometry_frames(ctx, 1, n - 1, pose0, 1);
        std::remove(result.c_str());
        TrajectoryWriter out(result);
        WorldPose w;
        out.append(w);
        std::unique_ptr<LaserMapping> lm;
        auto map_frame = [&](int k) {                                         // laserMapping's process() for frame k
            lm->transformAssociateToMap(w.q, w.t);                            // :1581
            lm->process_slot(k);                                              // :1584-2165, fed device-to-device from the slot
            lm->transformUpdate(w.q, w.t);                                    // :2101
            WorldPose m;
            for (int i = 0; i < 4; ++i) m.q[i] = lm->parameters[i];
            for (int i = 0; i < 3; ++i) m.t[i] = lm->parameters[4 + i];
            return m;
        };
        const int tiles = argc > 5 ? std::atoi(argv[5]) : 0;
        if (tiles >= 2) {
            // mapping = G >= 2: the map split over G ranks, here G threads of this process, each with its
            // own context and LaserMapping shard; the all-gather is a barrier + memcpy (ncclAllGather between processes).
            // The trajectory written is rank 0's -- and must be the one mapping = 1 writes, byte for byte.
            std::vector<std::vector<PointXYZI>> corner(n), surf(n);
            for (int k = 0; k < n; ++k) {
                ll_scan_info info;
                ctx.check(ll_get_scan_info(ctx.get(), k, &info));
                corner[k].resize(info.n_less_sharp); surf[k].resize(info.n_less_flat);
                ctx.check(ll_download_features(ctx.get(), k, nullptr, 0, (ll_point *)corner[k].data(), info.n_less_sharp, nullptr, 0,
                                               (ll_point *)surf[k].data(), info.n_less_flat));
            }
            std::vector<WorldPose> odo(n);
            for (int k = 1; k < n; ++k) { w.compose(&rel[(size_t)(k - 1) * 7], &rel[(size_t)(k - 1) * 7 + 4]); odo[k] = w; }
            struct Exchange {
                std::mutex mu; std::condition_variable cv; int arrived = 0, generation = 0, world = 0;
                const void *send[64];
                void barrier(std::unique_lock<std::mutex> &lk) {
                    const int gen = generation;
                    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
                    else cv.wait(lk, [&] { return generation != gen; });
                }
                void all_gather(int rank, const void *s, void *recv, size_t bytes) {
                    std::unique_lock<std::mutex> lk(mu);
                    send[rank] = s;
                    barrier(lk);
                    for (int r = 0; r < world; ++r) std::memcpy((char *)recv + (size_t)r * bytes, send[r], bytes);
                    barrier(lk);                                       // nobody reuses its send buffer before all have copied
                }
            } ex;
            ex.world = tiles;
            std::vector<std::string> errors(tiles);
            std::vector<std::vector<WorldPose>> mapped_of(tiles, std::vector<WorldPose>(n));
            std::vector<std::thread> th;
            for (int r = 0; r < tiles; ++r)
                th.emplace_back([&, r] {
                    try {
                        Context c(scan_line, 1, 0, -1.0, -24.9f, 2.0f, 0, max_ring_points);
                        LaserMapping shard(c, 0.4f, 0.8f, scan_line * 120 + 64, 400000, 1 << 22);
                        shard.set_shard(r, tiles);
                        for (int k = 0; k < n; ++k) {
                            shard.transformAssociateToMap(odo[k].q, odo[k].t);
                            shard.process_tile_parallel(corner[k], surf[k], [&](const void *s_, void *d_, size_t b) { ex.all_gather(r, s_, d_, b); });
                            shard.transformUpdate(odo[k].q, odo[k].t);
                            for (int i = 0; i < 4; ++i) mapped_of[r][k].q[i] = shard.parameters[i];
                            for (int i = 0; i < 3; ++i) mapped_of[r][k].t[i] = shard.parameters[4 + i];
                        }
                    } catch (const std::exception &e) { errors[r] = e.what(); std::cerr << "rank " << r << ": " << e.what() << "\n"; std::_Exit(1); }
                });
            for (auto &t : th) t.join();
            for (int r = 1; r < tiles; ++r)
                for (int k = 0; k < n; ++k)
                    if (std::memcmp(&mapped_of[r][k], &mapped_of[0][k], sizeof(WorldPose)) != 0) { std::cerr << "ranks disagree at frame " << k << "\n"; return 1; }
            std::remove(result.c_str());
            TrajectoryWriter mapped(result);
            for (int k = 0; k < n; ++k) mapped.append(mapped_of[0][k]);
            std::cout << "wrote " << n << " mapped poses (map split over " << tiles << " ranks) to " << result << "\n";
            return 0;
        }
        if (mapping) {
            lm.reset(new LaserMapping(ctx, 0.4f, 0.8f, scan_line * 120 + 64, 400000, 1 << 22));
            std::remove(result.c_str());
            TrajectoryWriter mapped(result);
            mapped.append(map_frame(0));
            for (int k = 0; k < n - 1; ++k) { w.compose(&rel[(size_t)k * 7], &rel[(size_t)k * 7 + 4]); mapped.append(map_frame(k + 1)); }
            std::cout << "wrote " << n << " mapped poses to " << result << "; final position " << lm->parameters[4] << " " << lm->parameters[5] << " " << lm->parameters[6] << "\n";
            return 0;
        }
        for (int k = 0; k < n - 1; ++k) { w.compose(&rel[(size_t)k * 7], &rel[(size_t)k * 7 + 4]); out.append(w); }
        std::cout << "wrote " << n << " poses to " << result << "; final position " << w.t[0] << " " << w.t[1] << " " << w.t[2] << "\n";
    } catch (const lightloam::Error &e) {
        std::cerr << "lightloam error " << e.code << ": " << e.what() << "\n";
        return 1;
    }
    return 0;
}
