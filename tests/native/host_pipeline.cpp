// C++ host-side use of include/lightloam_host.hpp, the way the reference nodes would call it:
//   host_pipeline <scan0.bin> <scan1.bin> <out_prefix> <scan_line>
// scan*.bin are KITTI-format clouds (float32 x,y,z,reflectance per point, src/kittiHelper.cpp:22-32).
// Runs scanRegistration on both, makes scan 0 the "last" clouds, runs one odometry iteration for scan 1 with the
// graph vote, then the free-function vote on the same correspondences, and dumps everything for the Python test.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "lightloam_host.hpp"

using namespace lightloam;

static std::vector<float> read_bin(const char *path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "cannot open " << path << "\n"; std::exit(2); }
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<float> v((size_t)bytes / sizeof(float));
    f.read((char *)v.data(), bytes);
    return v;
}

template <typename T>
static void dump(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char **argv)
{
    if (argc < 5) { std::cerr << "usage: host_pipeline scan0.bin scan1.bin out_prefix scan_line\n"; return 2; }
    const std::string out = argv[3];
    try {
        Context ctx(std::atoi(argv[4]), /*batch=*/2);
        std::vector<PointXYZI> cloud[2], sharp[2], lsharp[2], flat[2], lflat[2];
        for (int k = 0; k < 2; ++k) {
            const std::vector<float> pts = read_bin(argv[1 + k]);
            if (!laserCloudHandler(ctx, k, pts.data(), 4, (int)(pts.size() / 4), cloud[k], sharp[k], lsharp[k], flat[k], lflat[k])) {
                std::cerr << "empty scan\n"; return 3;
            }
        }
        dump(out + "_cloud1.bin", cloud[1]); dump(out + "_sharp1.bin", sharp[1]); dump(out + "_lsharp1.bin", lsharp[1]);
        dump(out + "_flat1.bin", flat[1]); dump(out + "_lflat1.bin", lflat[1]);

        OdometryFrame odo(ctx);
        odo.set_last(lsharp[0], lflat[0]);                 // through host clouds, like the topic hand-over of the reference
        double q[4] = {0, 0, 0, 1}, t[3] = {0.9, 0.0, 0.0};
        ll_pair_info info;
        odo.iterate(1, q, t, /*vote=*/true, &info);
        std::vector<double> pose = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
        dump(out + "_pose.bin", pose);

        // the free function on the plane correspondences of that iteration
        std::vector<int> ps(info.n_plane), pa(info.n_plane), pb(info.n_plane), pc(info.n_plane);
        ctx.check(ll_download_plane_corr(ctx.get(), 1, ps.data(), pa.data(), pb.data(), pc.data(), info.n_plane));
        std::vector<Corre_Match> corr(info.n_plane);
        for (int i = 0; i < info.n_plane; ++i) {
            corr[i].index = i; corr[i].src = flat[1][ps[i]]; corr[i].tgt = lflat[0][pa[i]]; corr[i].score = 0; corr[i].s = 1;
        }
        std::vector<Vertex_Vote> selected;
        graph_based_correspondence_vote_simple(ctx, corr, false, selected);
        std::vector<float> sel_flat;
        for (const auto &v : selected) { sel_flat.push_back((float)v.index); sel_flat.push_back(v.score); }
        dump(out + "_selected.bin", sel_flat);
        std::printf("ok n=%zu sharp=%zu flat=%zu edges=%d planes=%d selected=%d/%zu\n", cloud[1].size(), sharp[1].size(), flat[1].size(),
                    info.n_edge, info.n_plane, info.n_plane_selected, selected.size());
        // error behaviour: scan_line 48 must be refused like the node does (scanRegistration.cpp:447-451)
        try { Context bad(48); std::cerr << "scan_line 48 accepted\n"; return 4; } catch (const Error &e) { if (e.code != LL_ERR_BAD_RINGS) return 5; }
    } catch (const Error &e) {
        std::cerr << "lightloam error " << e.code << ": " << e.what() << "\n";
        return 1;
    }
    return 0;
}
