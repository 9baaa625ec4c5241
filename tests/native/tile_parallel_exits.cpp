// lightloam::LaserMapping::process_tile_parallel must never leave a rank waiting in a collective (include/lightloam_host.hpp: "failure
// discipline").  Three ranks = three host threads on the one device, each with its own context and map shard; the all_gather is a
// barrier + memcpy between the threads -- like ncclAllGather, it only returns when EVERY rank has called it, so a rank that threw
// between two gathers would hang the other two for ever.
//
//   tile_parallel_exits <dir> <n_frames>     (corner_<k>.bin / surf_<k>.bin / odom_<k>.bin as for rccl_normal_equations)
//
// Frame 0 runs on healthy shards (all ranks agree).  Then rank 1's shard is replaced by one whose scan capacity is far too small:
// its ll_cubemap_prepare fails with LL_ERR_CAPACITY on that rank only.  Expected: all three ranks come back from the frame with an
// exception -- rank 1 with its own error, the others with "another rank failed" -- within the watchdog's time; exit code 0.
//
// Build: g++ -O2 -std=c++14 -pthread -I include tests/native/tile_parallel_exits.cpp -L light-loam_amd -llightloam_hip
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lightloam_host.hpp"

using lightloam::PointXYZI;

static std::vector<PointXYZI> read_points(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<PointXYZI> p((size_t)f.tellg() / sizeof(PointXYZI));
    f.seekg(0); f.read((char *)p.data(), (std::streamsize)(p.size() * sizeof(PointXYZI)));
    return p;
}

struct Exchange {
    std::mutex mu; std::condition_variable cv; int arrived = 0, generation = 0, world = 0;
    const void *send[16];
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
        barrier(lk);
    }
};

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: tile_parallel_exits <dir> <n_frames>\n"; return 2; }
    const std::string dir = argv[1];
    const int n_frames = std::atoi(argv[2]), world = 3;
    using namespace lightloam;
    std::vector<std::vector<PointXYZI>> corner((size_t)n_frames), surf((size_t)n_frames);
    std::vector<double> odom((size_t)n_frames * 7);
    for (int k = 0; k < n_frames; ++k) {
        corner[(size_t)k] = read_points(dir + "/corner_" + std::to_string(k) + ".bin");
        surf[(size_t)k] = read_points(dir + "/surf_" + std::to_string(k) + ".bin");
        std::ifstream f(dir + "/odom_" + std::to_string(k) + ".bin", std::ios::binary);
        f.read((char *)&odom[(size_t)k * 7], 7 * sizeof(double));
    }
    Exchange ex; ex.world = world;
    std::vector<std::string> healthy_err((size_t)world), failing_what((size_t)world);
    std::vector<int> threw((size_t)world, 0);
    std::vector<double> pose0((size_t)world * 7);
    std::atomic<int> done{0};
    std::thread watchdog([&] {
        for (int i = 0; i < 600 && done.load() < world; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (done.load() < world) { std::cerr << "a rank is still waiting in a collective after 60 s\n"; std::_Exit(3); }
    });
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            auto gather = [&](const void *s_, void *d_, size_t b) { ex.all_gather(r, s_, d_, b); };
            try {
                Context ctx(16, 1, 0);
                {   // frame 0 on healthy shards
                    LaserMapping shard(ctx, 0.4f, 0.8f, 4096, 32768, 1 << 20);
                    shard.set_shard(r, world);
                    const double *o = &odom[0];
                    shard.transformAssociateToMap(o, o + 4);
                    shard.process_tile_parallel(corner[0], surf[0], gather);
                    std::memcpy(&pose0[(size_t)r * 7], shard.parameters, 7 * sizeof(double));
                }
                {   // a frame in which rank 1 alone cannot even take the scan in
                    LaserMapping shard(ctx, 0.4f, 0.8f, r == 1 ? 8 : 4096, r == 1 ? 8 : 32768, 1 << 20);
                    shard.set_shard(r, world);
                    const double *o = &odom[7];
                    shard.transformAssociateToMap(o, o + 4);
                    try { shard.process_tile_parallel(corner[1], surf[1], gather); }
                    catch (const std::exception &e) { threw[(size_t)r] = 1; failing_what[(size_t)r] = e.what(); }
                }
            } catch (const std::exception &e) { healthy_err[(size_t)r] = e.what(); }
            ++done;
        });
    for (auto &t : th) t.join();
    watchdog.join();
    for (int r = 0; r < world; ++r) {
        if (!healthy_err[(size_t)r].empty()) { std::cerr << "rank " << r << ": " << healthy_err[(size_t)r] << "\n"; return 1; }
        if (!threw[(size_t)r]) { std::cerr << "rank " << r << " came back from the failing frame without an exception\n"; return 1; }
        if (std::memcmp(&pose0[(size_t)r * 7], &pose0[0], 7 * sizeof(double)) != 0) { std::cerr << "ranks disagree on the healthy frame\n"; return 1; }
    }
    if (failing_what[0].find("another rank failed") == std::string::npos || failing_what[2].find("another rank failed") == std::string::npos) {
        std::cerr << "the healthy ranks did not report the peer's failure: '" << failing_what[0] << "' / '" << failing_what[2] << "'\n"; return 1;
    }
    std::cout << "all " << world << " ranks left the failing frame together: rank 1: " << failing_what[1] << "\n";
    return 0;
}
