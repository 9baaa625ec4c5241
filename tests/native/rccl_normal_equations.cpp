// C++ -> RCCL with no torch in the process (include/lightloam_rccl.hpp): the two multi-GPU mapping modes of SURVEY.md section 8e
// at whatever world size the box offers (one rank per visible device, one host thread per rank), against the one-rank calls.
//
//   rccl_normal_equations <dir> <n_frames>
//     <dir>/map_corner.bin map_surf.bin stack_corner.bin stack_surf.bin   float32 x 4 per point
//     <dir>/pose.bin                                                       7 doubles: the guess (q x y z w, t)
//     <dir>/corner_<k>.bin surf_<k>.bin odom_<k>.bin                       per frame: laserCloudCornerLast / SurfLast, odometry pose (7 doubles)
//   writes <dir>/out_row.bin   : world, then per rank 7 doubles, then the one-rank ll_map_optimize pose, then all-reduces per rank
//          <dir>/out_tile.bin  : per frame: per rank 7 doubles, then the one-rank LaserMapping::process pose
//
// Build: g++ -O2 -std=c++14 -pthread -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/native/rccl_normal_equations.cpp
//            -L light-loam_amd -llightloam_hip -L /opt/rocm/lib -lrccl -lamdhip64
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "lightloam_host.hpp"
#include "lightloam_rccl.hpp"

using lightloam::PointXYZI;

static std::vector<PointXYZI> read_points(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const size_t bytes = (size_t)f.tellg();
    std::vector<PointXYZI> p(bytes / sizeof(PointXYZI));
    f.seekg(0); f.read((char *)p.data(), (std::streamsize)(p.size() * sizeof(PointXYZI)));
    return p;
}
static void read_doubles(const std::string &path, double *d, int n)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f.read((char *)d, n * (std::streamsize)sizeof(double));
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: rccl_normal_equations <dir> <n_frames>\n"; return 2; }
    const std::string dir = argv[1];
    const int n_frames = std::atoi(argv[2]);
    try {
        using namespace lightloam;
        static_assert(sizeof(PointXYZI) == 16, "x, y, z, intensity");
        RcclWorld rw;                                                     // ncclCommInitAll over the visible devices
        const int world = rw.size();
        const std::vector<PointXYZI> mc = read_points(dir + "/map_corner.bin"), ms = read_points(dir + "/map_surf.bin");
        const std::vector<PointXYZI> sc = read_points(dir + "/stack_corner.bin"), ss = read_points(dir + "/stack_surf.bin");
        double guess[7]; read_doubles(dir + "/pose.bin", guess, 7);

        // ---- row-parallel Levenberg-Marquardt: the 44-double all-reduce on the library's stream ----
        std::vector<double> pose_of((size_t)world * 7);
        std::vector<long> reduces((size_t)world, 0);
        std::vector<std::string> err((size_t)world);
        {
            std::vector<std::thread> th;
            for (int r = 0; r < world; ++r)
                th.emplace_back([&, r] {
                    try {
                        Context ctx(16, 1, r);
                        MapOptimizer mo(ctx, (int)mc.size() + 16, (int)ms.size() + 16, (int)sc.size() + 16, (int)ss.size() + 16);
                        mo.setInputClouds(mc, ms);
                        auto slice = [&](const std::vector<PointXYZI> &v) {
                            const size_t a = v.size() * (size_t)r / (size_t)world, b = v.size() * (size_t)(r + 1) / (size_t)world;
                            return std::vector<PointXYZI>(v.begin() + (long)a, v.begin() + (long)b);
                        };
                        mo.setScan(slice(sc), slice(ss));
                        RcclRank rk(rw.comm(r), ctx.get(), r, &rw);
                        double p[7]; std::memcpy(p, guess, sizeof(p));
                        map_optimize_row_parallel(mo.get(), rk, p, 2, nullptr);
                        std::memcpy(&pose_of[(size_t)r * 7], p, sizeof(p));
                        reduces[(size_t)r] = rk.n_allreduce;
                    } catch (const std::exception &e) { err[(size_t)r] = e.what(); }
                });
            for (auto &t : th) t.join();
            for (int r = 0; r < world; ++r) if (!err[(size_t)r].empty()) { std::cerr << "row-parallel rank " << r << ": " << err[(size_t)r] << "\n"; return 1; }
        }
        // ---- the helper's exits: a map below laserMapping.cpp:1822's sizes -> false on every rank, no collective issued; a step that
        // fails locally (max_num_iterations < 0 is refused by ll_map_lm_begin_dev) -> every rank still walks the whole sequence of
        // collectives and THEN throws: nobody is left waiting in ncclAllReduce ----
        {
            std::vector<int> small_ok((size_t)world, 0), failed((size_t)world, 0);
            std::vector<std::thread> th;
            for (int r = 0; r < world; ++r)
                th.emplace_back([&, r] {
                    try {
                        Context ctx(16, 1, r);
                        MapOptimizer mo(ctx, (int)mc.size() + 16, (int)ms.size() + 16, (int)sc.size() + 16, (int)ss.size() + 16);
                        RcclRank rk(rw.comm(r), ctx.get(), r, &rw);
                        double p[7]; std::memcpy(p, guess, sizeof(p));
                        mo.setInputClouds(std::vector<PointXYZI>(mc.begin(), mc.begin() + 8), ms);      // 8 corner points: too small
                        mo.setScan(sc, ss);
                        const long before = rk.n_allreduce;
                        const bool ran = map_optimize_row_parallel(mo.get(), rk, p, 2, nullptr);
                        small_ok[(size_t)r] = (!ran && rk.n_allreduce == before && std::memcmp(p, guess, sizeof(p)) == 0) ? 1 : 0;
                        mo.setInputClouds(mc, ms);
                        ll_lm_options bad; ll_lm_default_options(&bad); bad.max_num_iterations = -1;
                        try { (void)map_optimize_row_parallel(mo.get(), rk, p, 2, &bad); }
                        catch (const RcclError &) { failed[(size_t)r] = (rk.n_allreduce == before + 2 && !rk.aborted()) ? 1 : 0; }   // both outer rounds' all-reduces were issued
                    } catch (const std::exception &e) { err[(size_t)r] = e.what(); }
                });
            for (auto &t : th) t.join();
            for (int r = 0; r < world; ++r) {
                if (!err[(size_t)r].empty()) { std::cerr << "row-parallel exits, rank " << r << ": " << err[(size_t)r] << "\n"; return 1; }
                if (!small_ok[(size_t)r]) { std::cerr << "rank " << r << ": a map below the :1822 sizes was optimised against, or a collective was issued\n"; return 1; }
                if (!failed[(size_t)r]) { std::cerr << "rank " << r << ": a failing local step did not end in an exception after the full collective sequence\n"; return 1; }
            }
        }
        double ref[7]; std::memcpy(ref, guess, sizeof(ref));
        {
            Context ctx(16, 1, 0);
            MapOptimizer mo(ctx, (int)mc.size() + 16, (int)ms.size() + 16, (int)sc.size() + 16, (int)ss.size() + 16);
            mo.setInputClouds(mc, ms); mo.setScan(sc, ss);
            if (!mo.optimize(ref, 2)) { std::cerr << "map too small\n"; return 1; }
        }
        {
            std::ofstream f(dir + "/out_row.bin", std::ios::binary);
            const double w = world; f.write((const char *)&w, sizeof(w));
            f.write((const char *)pose_of.data(), (std::streamsize)(pose_of.size() * sizeof(double)));
            f.write((const char *)ref, sizeof(ref));
            for (int r = 0; r < world; ++r) { const double c = (double)reduces[(size_t)r]; f.write((const char *)&c, sizeof(c)); }
        }

        // ---- tile-parallel frames: LaserMapping::process_tile_parallel with RcclRank::all_gather_host as its all_gather ----
        std::vector<std::vector<PointXYZI>> corner((size_t)n_frames), surf((size_t)n_frames);
        std::vector<double> odom((size_t)n_frames * 7);
        for (int k = 0; k < n_frames; ++k) {
            corner[(size_t)k] = read_points(dir + "/corner_" + std::to_string(k) + ".bin");
            surf[(size_t)k] = read_points(dir + "/surf_" + std::to_string(k) + ".bin");
            read_doubles(dir + "/odom_" + std::to_string(k) + ".bin", &odom[(size_t)k * 7], 7);
        }
        std::vector<double> tile((size_t)n_frames * (size_t)(world + 1) * 7);
        {
            std::vector<std::thread> th;
            for (int r = 0; r < world; ++r)
                th.emplace_back([&, r] {
                    try {
                        Context ctx(16, 1, r);
                        LaserMapping shard(ctx, 0.4f, 0.8f, 4096, 32768, 1 << 20);
                        shard.set_shard(r, world);
                        RcclRank rk(rw.comm(r), ctx.get(), r, &rw);
                        for (int k = 0; k < n_frames; ++k) {
                            const double *o = &odom[(size_t)k * 7];
                            shard.transformAssociateToMap(o, o + 4);
                            shard.process_tile_parallel(corner[(size_t)k], surf[(size_t)k], [&](const void *s_, void *d_, size_t b) { rk.all_gather_host(s_, d_, b); });
                            shard.transformUpdate(o, o + 4);
                            std::memcpy(&tile[((size_t)k * (size_t)(world + 1) + (size_t)r) * 7], shard.parameters, 7 * sizeof(double));
                        }
                    } catch (const std::exception &e) { err[(size_t)r] = e.what(); }
                });
            for (auto &t : th) t.join();
            for (int r = 0; r < world; ++r) if (!err[(size_t)r].empty()) { std::cerr << "tile-parallel rank " << r << ": " << err[(size_t)r] << "\n"; return 1; }
        }
        {
            Context ctx(16, 1, 0);
            LaserMapping whole(ctx, 0.4f, 0.8f, 4096, 32768, 1 << 20);
            for (int k = 0; k < n_frames; ++k) {
                const double *o = &odom[(size_t)k * 7];
                whole.transformAssociateToMap(o, o + 4);
                whole.process(corner[(size_t)k], surf[(size_t)k]);
                whole.transformUpdate(o, o + 4);
                std::memcpy(&tile[((size_t)k * (size_t)(world + 1) + (size_t)world) * 7], whole.parameters, 7 * sizeof(double));
            }
        }
        std::ofstream f(dir + "/out_tile.bin", std::ios::binary);
        f.write((const char *)tile.data(), (std::streamsize)(tile.size() * sizeof(double)));
        std::cout << "rccl world " << world << ": row-parallel " << reduces[0] << " all-reduces per rank, tile-parallel " << n_frames << " frames\n";
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
