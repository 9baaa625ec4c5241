/*
 * lightloam_host.hpp -- C++ host-side mirror of the reference's own seams over the C ABI (lightloam_hip.h).
 *
 * The reference (BrenYi/Light-LOAM, paths relative to /root/reference/) is three ROS1 nodes; a maintainer replaces
 * the BODY of the functions below and keeps names, argument meaning, output order and error behaviour:
 *   laserCloudHandler                           src/scanRegistration.cpp:87-428
 *   graph_based_correspondence_vote_simple      src/laserOdometry.cpp:165-342 (call :796)
 *   the correspondence loops + ceres::Solve     src/laserOdometry.cpp:439-832  -> OdometryFrame below
 *   Corre_Match / Vertex_Vote / compare_score   include/aloam_velodyne/common.h:20-52
 * Header-only, C++14, no ROS / PCL / Eigen / Ceres needed (this image has none): clouds are std::vector<PointXYZI>
 * with the PCL field names; templates convert from any point type that has x, y, z (and intensity).
 * The optional Ceres adapter at the end compiles only when <ceres/ceres.h> is available and is NOT compiled or
 * tested in this image.
 */
#ifndef LIGHTLOAM_HOST_HPP
#define LIGHTLOAM_HOST_HPP

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lightloam_hip.h"

namespace lightloam {

struct PointXYZI { float x, y, z, intensity; };                      /* pcl::PointXYZI without the SSE padding */
static_assert(sizeof(PointXYZI) == sizeof(ll_point), "layout");
typedef PointXYZI PointType;                                          /* common.h:7 */

typedef struct {                                                      /* common.h:20-31 */
    int index;
    PointXYZI src;
    PointXYZI tgt;
    float score;
    float s;
} Corre_Match;

typedef struct { int index; float score; } Vertex_Vote;               /* common.h:40-43 */
struct compare_score { bool operator()(Vertex_Vote const &a, Vertex_Vote const &b) { return a.score > b.score; } };   /* :50-52 */

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

/* RAII ll_ctx; one per node thread (the reference nodes are single-threaded spinners, scanRegistration.cpp:475) */
class Context {
public:
    /* distortion: the DISTORTION macro of laserOdometry.cpp:23 (0 in the reference's build; 1 = its per-point interpolation path) */
    /* max_ring_points: capacity of one scan line (0: the library default, 2304 = a 2048-column sensor; real HDL-64E / KITTI data
     * under the linear 64-ring model of scanRegistration.cpp:162 needs 4608 -- some bins hold two lasers) */
    explicit Context(int scan_line, int batch = 2, int device = 0, double minimum_range = -1.0,
                     float lowerBound = -24.9f, float upBound = 2.0f, int distortion = 0, int max_ring_points = 0, int input_stride_floats = 4) {
        ll_default_params(&p_, scan_line);
        p_.batch = batch;
        p_.distortion = distortion;
        p_.input_stride_floats = input_stride_floats;                           /* 3: the resident scan keeps x, y, z only, as pcl::fromROSMsg into PointXYZ does (:105-106) */
        if (max_ring_points > 0) p_.max_ring_points = max_ring_points;
        if (minimum_range >= 0) p_.minimum_range = (float)minimum_range;       /* nh.param("minimum_range") :438 */
        p_.lower_bound = lowerBound; p_.up_bound = upBound;                     /* nh.param("lowerBound" / "upBound") :439-440 */
        const int rc = ll_create(device, &p_, &ctx_);
        if (rc != LL_OK) throw Error(rc, ll_last_error(nullptr));               /* LL_ERR_BAD_RINGS == "only support velodyne with 16, 32 or 64 scan line" :447-451 */
    }
    ~Context() { ll_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ll_ctx *get() const { return ctx_; }
    const ll_params &params() const { return p_; }
    void check(int rc) const { if (rc != LL_OK) throw Error(rc, ll_last_error(ctx_)); }
private:
    ll_params p_;
    ll_ctx *ctx_ = nullptr;
};

/* ---- scanRegistration.cpp:87-428.  Input: the points of the PointCloud2 message (x,y,z[,.]) with `stride` floats
 * per point; outputs: what the node publishes on /velodyne_cloud_2, /laser_cloud_sharp, /laser_cloud_less_sharp,
 * /laser_cloud_flat, /laser_cloud_less_flat (:382-410), same point order.  Returns false where the reference would
 * have dereferenced an empty cloud (no point survives the filters). */
inline bool laserCloudHandler(Context &c, int slot, const float *xyz, int stride, int n,
                              std::vector<PointXYZI> &laserCloud, std::vector<PointXYZI> &cornerPointsSharp,
                              std::vector<PointXYZI> &cornerPointsLessSharp, std::vector<PointXYZI> &surfPointsFlat,
                              std::vector<PointXYZI> &surfPointsLessFlat)
{
    c.check(ll_upload_scan(c.get(), slot, xyz, stride, n));
    c.check(ll_extract_batch(c.get(), slot, 1));
    ll_scan_info info;
    c.check(ll_get_scan_info(c.get(), slot, &info));
    if (info.status == LL_ERR_EMPTY) return false;
    if (info.status != LL_OK) throw Error(info.status, "scan registration failed (capacity)");
    laserCloud.resize(info.n); cornerPointsSharp.resize(info.n_sharp); cornerPointsLessSharp.resize(info.n_less_sharp);
    surfPointsFlat.resize(info.n_flat); surfPointsLessFlat.resize(info.n_less_flat);
    c.check(ll_download_cloud(c.get(), slot, (ll_point *)laserCloud.data(), info.n, nullptr, nullptr));
    c.check(ll_download_features(c.get(), slot, (ll_point *)cornerPointsSharp.data(), info.n_sharp,
                                 (ll_point *)cornerPointsLessSharp.data(), info.n_less_sharp,
                                 (ll_point *)surfPointsFlat.data(), info.n_flat, (ll_point *)surfPointsLessFlat.data(), info.n_less_flat));
    return true;
}

/* ---- laserOdometry.cpp:165-342, same signature minus the six unused arguments.  Appends to selected_idx in the
 * reference's order: per region, the vote records sorted with the SAME std::sort / compare_score call on the same
 * initial sequence, then walked from the low-count end (:255, :304-329). */
inline void graph_based_correspondence_vote_simple(Context &c, std::vector<Corre_Match> &correspondences, bool corner_case,
                                                   std::vector<Vertex_Vote> &selected_idx)
{
    const int n = (int)correspondences.size();
    if (n == 0) return;
    std::vector<ll_point> src(n), tgt(n);
    for (int i = 0; i < n; ++i) {
        src[i] = {correspondences[i].src.x, correspondences[i].src.y, correspondences[i].src.z, correspondences[i].src.intensity};
        tgt[i] = {correspondences[i].tgt.x, correspondences[i].tgt.y, correspondences[i].tgt.z, correspondences[i].tgt.intensity};
    }
    std::vector<int> count(n);
    c.check(ll_vote_host(c.get(), src.data(), tgt.data(), n, corner_case ? 1 : 0, count.data(), nullptr, nullptr));
    const int number_of_region = corner_case ? 5 : 10;                                  /* :179-188 */
    for (int num_region = 0; num_region < number_of_region; num_region++) {
        const int initial_pos = n / number_of_region * num_region;                      /* :202 */
        const int end_pos = (num_region == number_of_region - 1) ? n : n / number_of_region * (num_region + 1);
        const int cor_size = end_pos - initial_pos;
        std::vector<Vertex_Vote> vote_record((size_t)std::max(cor_size, 0));
        for (int i = 0; i < cor_size; ++i) { vote_record[i].index = i; vote_record[i].score = (float)count[initial_pos + i]; }
        std::sort(vote_record.begin(), vote_record.end(), compare_score());             /* :255 */
        const float num_selected = 0.90f * cor_size;                                    /* :299-300 */
        for (int i = cor_size - 1; i >= 0; i--) {                                       /* :304-329 */
            Vertex_Vote obj;
            obj.index = correspondences[initial_pos + vote_record[i].index].index;
            if (vote_record[i].score > num_selected) break;
            obj.score = (vote_record[i].score <= 50) ? 5.0f : 1.0f;
            selected_idx.push_back(obj);
        }
    }
}

/* ---- the per-frame body of laserOdometry.cpp:439-832, one outer iteration at a time.
 * slot_curr holds the current scan's features (after laserCloudHandler), the target is what set_last() stored
 * (laserCloudCornerLast / laserCloudSurfLast, :882-896).  q = para_q (x,y,z,w), t = para_t (:61-62). */
class OdometryFrame {
public:
    explicit OdometryFrame(Context &c) : c_(c) {}
    /* kdtreeCornerLast->setInputCloud / kdtreeSurfLast->setInputCloud (:895-896) from host clouds ... */
    void set_last(const std::vector<PointXYZI> &cornerLast, const std::vector<PointXYZI> &surfLast) {
        c_.check(ll_set_target(c_.get(), (const ll_point *)cornerLast.data(), (int)cornerLast.size(),
                               (const ll_point *)surfLast.data(), (int)surfLast.size()));
    }
    /* ... or device-to-device from the slot that was just registered (the pointer swap of :882-888) */
    void set_last_from_slot(int slot) { c_.check(ll_set_target_from_slot(c_.get(), slot)); }

    /* correspondence search (:491-620, :653-793) + vote when now_frame > 5 (:794-810) + residual blocks + one
     * Gauss-Newton iteration of the problem Ceres would be handed (:820-825).  Updates q, t in place. */
    void iterate(int slot_curr, double q[4], double t[3], bool vote, ll_pair_info *info = nullptr) {
        const double pose[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
        c_.check(ll_associate_batch(c_.get(), slot_curr, 1, pose));
        c_.check(ll_vote_batch(c_.get(), slot_curr, 1, vote ? 1 : 0));
        c_.check(ll_normal_equations_batch(c_.get(), slot_curr, 1, nullptr));
        c_.check(ll_gn_step_batch(c_.get(), slot_curr, 1));
        double out[7];
        c_.check(ll_download_pose(c_.get(), slot_curr, out));
        for (int k = 0; k < 4; ++k) q[k] = out[k];
        for (int k = 0; k < 3; ++k) t[k] = out[4 + k];
        if (info) c_.check(ll_get_pair_info(c_.get(), slot_curr, info));
    }
private:
    Context &c_;
};

/* ---- laserMapping's optimisation block (first stage) -------------------------------------------------
 * What laserMapping.cpp:1822-2095 does once per frame, with the names it uses: the clouds gathered from the cube map,
 * the scan's down-sized feature clouds, parameters[7] = (q_w_curr x,y,z,w, t_w_curr).  The cube bookkeeping around it
 * (:1584-1808, :2101-2165) stays in the caller. */
class MapOptimizer {
public:
    MapOptimizer(Context &c, int max_map_corner, int max_map_surf, int max_scan_corner, int max_scan_surf) {
        c.check(ll_map_create(c.get(), max_map_corner, max_map_surf, max_scan_corner, max_scan_surf, &m_));
    }
    ~MapOptimizer() { ll_map_destroy(m_); }
    MapOptimizer(const MapOptimizer &) = delete;
    MapOptimizer &operator=(const MapOptimizer &) = delete;
    /* kdtreeCornerFromMap->setInputCloud(laserCloudCornerFromMap); kdtreeSurfFromMap->setInputCloud(...) (:1826-1827) */
    void setInputClouds(const std::vector<PointXYZI> &laserCloudCornerFromMap, const std::vector<PointXYZI> &laserCloudSurfFromMap) {
        check(ll_map_set_map(m_, (const ll_point *)laserCloudCornerFromMap.data(), (int)laserCloudCornerFromMap.size(),
                             (const ll_point *)laserCloudSurfFromMap.data(), (int)laserCloudSurfFromMap.size()));
    }
    /* laserCloudCornerStack / laserCloudSurfStack (:1813-1821) */
    void setScan(const std::vector<PointXYZI> &laserCloudCornerStack, const std::vector<PointXYZI> &laserCloudSurfStack) {
        check(ll_map_set_scan(m_, (const ll_point *)laserCloudCornerStack.data(), (int)laserCloudCornerStack.size(),
                              (const ll_point *)laserCloudSurfStack.data(), (int)laserCloudSurfStack.size()));
    }
    /* the `if (laserCloudCornerFromMapNum > 10 && laserCloudSurfFromMapNum > 50)` block: iterCount 0..1, data association
     * + ceres::Solve each.  Returns false where the reference prints "time Map corner and surf num are not enough". */
    bool optimize(double parameters[7], int iterations = 2) {
        int ran = 0;
        check(ll_map_optimize(m_, parameters, iterations, nullptr, &ran));
        return ran != 0;
    }
    void counts(int &corner_num, int &surf_num) { check(ll_map_get_counts(m_, &corner_num, &surf_num)); }
    ll_map *get() const { return m_; }
private:
    void check(int rc) { if (rc != LL_OK) throw Error(rc, ll_map_last_error(m_)); }
    ll_map *m_ = nullptr;
};

/* ---- laserMapping's per-frame body with the cube map on the device (second stage) ---------------------
 * process() of laserMapping.cpp between transformAssociateToMap (:1581) and transformUpdate (:2101) plus the map update
 * (:2103-2165): the 21 x 21 x 11 cube arrays, their shifting, laserCloudCornerFromMap / SurfFromMap, the down-sized
 * scan, the optimisation, adding the registered scan and down-sizing the touched cubes.  What is left to the node: the
 * message queues, the two transform helpers below, publishing. */
class LaserMapping {
public:
    LaserMapping(Context &c, float lineRes = 0.4f, float planeRes = 0.8f, int max_scan_corner = 20000, int max_scan_surf = 200000,
                 int pool_points = 1 << 22) {
        c.check(ll_cubemap_create(c.get(), lineRes, planeRes, max_scan_corner, max_scan_surf, pool_points, &cm_));
    }
    ~LaserMapping() { ll_cubemap_destroy(cm_); }
    LaserMapping(const LaserMapping &) = delete;
    LaserMapping &operator=(const LaserMapping &) = delete;

    /* transformAssociateToMap (:113-117): q_w_curr = q_wmap_wodom * q_wodom_curr, t_w_curr = q_wmap_wodom * t_wodom_curr + t_wmap_wodom */
    void transformAssociateToMap(const double q_wodom_curr[4], const double t_wodom_curr[3]) {
        qmul(q_wmap_wodom, q_wodom_curr, parameters);
        double r[3]; qrot(q_wmap_wodom, t_wodom_curr, r);
        for (int k = 0; k < 3; ++k) parameters[4 + k] = r[k] + t_wmap_wodom[k];
    }
    /* transformUpdate (:119-123): q_wmap_wodom = q_w_curr * q_wodom_curr^-1, t_wmap_wodom = t_w_curr - q_wmap_wodom * t_wodom_curr */
    void transformUpdate(const double q_wodom_curr[4], const double t_wodom_curr[3]) {
        const double n2 = q_wodom_curr[0] * q_wodom_curr[0] + q_wodom_curr[1] * q_wodom_curr[1] + q_wodom_curr[2] * q_wodom_curr[2] + q_wodom_curr[3] * q_wodom_curr[3];
        const double inv[4] = {-q_wodom_curr[0] / n2, -q_wodom_curr[1] / n2, -q_wodom_curr[2] / n2, q_wodom_curr[3] / n2};
        qmul(parameters, inv, q_wmap_wodom);
        double r[3]; qrot(q_wmap_wodom, t_wodom_curr, r);
        for (int k = 0; k < 3; ++k) t_wmap_wodom[k] = parameters[4 + k] - r[k];
    }
    /* :1584-2165 for one frame; parameters[] holds the guess on entry and the optimised q_w_curr, t_w_curr on return */
    bool process(const std::vector<PointXYZI> &laserCloudCornerLast, const std::vector<PointXYZI> &laserCloudSurfLast) {
        int ran = 0;
        check(ll_cubemap_process(cm_, parameters, (const ll_point *)laserCloudCornerLast.data(), (int)laserCloudCornerLast.size(),
                                 (const ll_point *)laserCloudSurfLast.data(), (int)laserCloudSurfLast.size(), &ran));
        return ran != 0;
    }
    /* the same for the scan in an extracted slot of the context: nothing crosses the PCIe bus */
    bool process_slot(int slot) {
        int ran = 0;
        check(ll_cubemap_process_slot(cm_, parameters, slot, &ran));
        return ran != 0;
    }
    /* ---- tile-parallel over the GPUs of a node: one LaserMapping per rank, each keeping the cubes it
     * owns.  all_gather(send, recv, bytes): `bytes` from every rank into recv, rank-major (ncclAllGather / MPI_Allgather;
     * the buffers handed over are host memory).  Every rank passes the same scan and ends with the same parameters[]. */
    void set_shard(int rank, int world) { check(ll_cubemap_set_shard(cm_, rank, world)); world_ = world; }
    /* Failure discipline: nothing here throws between two collectives.  A rank whose library call failed remembers it, skips its
     * remaining library calls of the frame but still takes part in every all_gather of the sequence; the ranks exchange a status
     * word with the counts and once per outer iteration, so that ALL of them leave the frame at the same point -- each with an
     * exception (its own error, or "a peer failed") -- instead of one leaving and the others waiting in a gather for ever. */
    template <class AllGather>
    bool process_tile_parallel(const std::vector<PointXYZI> &laserCloudCornerLast, const std::vector<PointXYZI> &laserCloudSurfLast,
                               AllGather &&all_gather, const ll_lm_options *opt = nullptr) {
        std::string local_err; int local_rc = LL_OK;                                      /* first failure on this rank: message + its code */
        /* the message is fetched AFTER the call: as a second argument of step() it would be evaluated in an unspecified order (GCC:
         * right to left, i.e. before the failing call has assigned the error string it points into) */
        auto step = [&](int rc, auto last_err) { if (rc != LL_OK && local_err.empty()) { local_rc = rc; local_err = std::string(last_err()); } return rc == LL_OK; };
        auto cm_err = [&]() { return ll_cubemap_last_error(cm_); };
        auto any_failed = [&]() {                                                         /* one int per rank: a collective of its own */
            int mine = local_err.empty() ? 0 : 1;
            std::vector<int> all((size_t)world_);
            all_gather(&mine, all.data(), sizeof(int));
            for (int v : all) if (v) return true;
            return false;
        };
        auto leave = [&]() { if (local_err.empty()) throw Error(LL_ERR_STATE, std::string("process_tile_parallel: another rank failed")); throw Error(local_rc, local_err); };
        int cnt[4] = {0, 0, 0, 0};
        if (step(ll_cubemap_prepare(cm_, parameters + 4, (const ll_point *)laserCloudCornerLast.data(), (int)laserCloudCornerLast.size(),
                                    (const ll_point *)laserCloudSurfLast.data(), (int)laserCloudSurfLast.size()), cm_err))
            step(ll_cubemap_info(cm_, nullptr, cnt), cm_err);
        int mine[3] = {cnt[0], cnt[1], local_err.empty() ? 0 : 1};
        std::vector<int> all_cnt((size_t)3 * world_);
        all_gather(mine, all_cnt.data(), 3 * sizeof(int));
        long tot[2] = {0, 0}; bool failed = false;
        for (int r = 0; r < world_; ++r) { tot[0] += all_cnt[3 * r]; tot[1] += all_cnt[3 * r + 1]; failed = failed || all_cnt[3 * r + 2] != 0; }
        if (failed) leave();
        const bool ran = tot[0] > 10 && tot[1] > 50;                                      /* :1822 */
        if (ran) {
            ll_map *m = ll_cubemap_map(cm_);
            auto m_err = [&]() { return ll_map_last_error(m); };
            const size_t nc = (size_t)cnt[2] * 5, ns = (size_t)cnt[3] * 5;
            /* ONE all_gather per outer iteration: a rank's status word and its four candidate arrays travel as one packed record
             * [status | corner (x, y, z, d) | corner ids | surf (x, y, z, d) | surf ids] (every rank holds the whole scan, so the
             * records have one size); the receiver lays the parts out rank-major again for ll_map_associate_merged.  (Four gathers
             * and a status gather per iteration were five host-staged collectives, each with its own stream synchronisation.) */
            const size_t o_cn = 4, o_ci = o_cn + nc * 16, o_sn = o_ci + nc * 4, o_si = o_sn + ns * 16, rec = o_si + ns * 4;
            std::vector<unsigned char> pack(rec), all(rec * (size_t)world_);
            std::vector<float> acn(nc * 4 * world_ + 4), asn(ns * 4 * world_ + 4);
            std::vector<int> aci(nc * world_ + 1), asi(ns * world_ + 1);
            for (int it = 0; it < 2; ++it) {                                              /* :1832 */
                if (local_err.empty())
                    step(ll_map_knn_partial(m, parameters, (float *)(pack.data() + o_cn), (int *)(pack.data() + o_ci),
                                            (float *)(pack.data() + o_sn), (int *)(pack.data() + o_si)), m_err);
                const int mine_st = local_err.empty() ? 0 : 1;
                std::memcpy(pack.data(), &mine_st, sizeof(int));
                all_gather(pack.data(), all.data(), rec);
                bool bad = false;
                for (int r = 0; r < world_; ++r) {
                    const unsigned char *q = all.data() + (size_t)r * rec;
                    int st; std::memcpy(&st, q, sizeof(int)); bad = bad || st != 0;
                    std::memcpy((unsigned char *)acn.data() + (size_t)r * nc * 16, q + o_cn, nc * 16);
                    std::memcpy((unsigned char *)aci.data() + (size_t)r * nc * 4, q + o_ci, nc * 4);
                    std::memcpy((unsigned char *)asn.data() + (size_t)r * ns * 16, q + o_sn, ns * 16);
                    std::memcpy((unsigned char *)asi.data() + (size_t)r * ns * 4, q + o_si, ns * 4);
                }
                if (bad) leave();                                                         /* every rank takes this exit in the same iteration */
                if (step(ll_map_associate_merged(m, parameters, world_, acn.data(), aci.data(), asn.data(), asi.data()), m_err))
                    step(ll_map_solve(m, parameters, opt), m_err);
            }
            if (any_failed()) leave();                                                    /* the last iteration's merge / solve */
        }
        check(ll_cubemap_update(cm_, parameters));                                        /* local to the rank: no collective behind it in this frame */
        return ran;
    }
    double parameters[7] = {0, 0, 0, 1, 0, 0, 0};                 /* :81-83 */
    double q_wmap_wodom[4] = {0, 0, 0, 1}, t_wmap_wodom[3] = {0, 0, 0};   /* :88-89 */
    ll_cubemap *get() const { return cm_; }
private:
    static void qmul(const double a[4], const double b[4], double o[4]) {
        const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
        o[0] = aw * bx + ax * bw + ay * bz - az * by; o[1] = aw * by - ax * bz + ay * bw + az * bx;
        o[2] = aw * bz + ax * by - ay * bx + az * bw; o[3] = aw * bw - ax * bx - ay * by - az * bz;
    }
    static void qrot(const double q[4], const double v[3], double o[3]) {
        const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
        double uvx = uy * v[2] - uz * v[1], uvy = uz * v[0] - ux * v[2], uvz = ux * v[1] - uy * v[0];
        uvx += uvx; uvy += uvy; uvz += uvz;
        o[0] = v[0] + w * uvx + (uy * uvz - uz * uvy); o[1] = v[1] + w * uvy + (uz * uvx - ux * uvz); o[2] = v[2] + w * uvz + (ux * uvy - uy * uvx);
    }
    void check(int rc) { if (rc != LL_OK) throw Error(rc, ll_cubemap_last_error(cm_)); }
    static void mcheck(ll_map *m, int rc) { if (rc != LL_OK) throw Error(rc, ll_map_last_error(m)); }
    ll_cubemap *cm_ = nullptr;
    int world_ = 1;
};

/* ---- I/O surface ---------------------------------------------------------------------------------- */

/* KITTI velodyne .bin: float32 (x, y, z, reflectance) per point -- src/kittiHelper.cpp:22-32, same name */
inline std::vector<float> read_lidar_data(const std::string &lidar_data_path)
{
    std::FILE *f = std::fopen(lidar_data_path.c_str(), "rb");
    if (!f) throw Error(LL_ERR_ARG, "cannot open " + lidar_data_path);
    std::fseek(f, 0, SEEK_END);
    const size_t num_elements = (size_t)std::ftell(f) / sizeof(float);
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> lidar_data_buffer(num_elements);
    const size_t got = std::fread(lidar_data_buffer.data(), sizeof(float), num_elements, f);
    std::fclose(f);
    lidar_data_buffer.resize(got);
    return lidar_data_buffer;
}

/* World pose accumulation of laserOdometry (:830-831): t_w = t_w + q_w * t ; q_w = q_w * q   (q = x,y,z,w) */
struct WorldPose {
    double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
    void compose(const double ql[4], const double tl[3]) {
        const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
        double uv[3] = {uy * tl[2] - uz * tl[1], uz * tl[0] - ux * tl[2], ux * tl[1] - uy * tl[0]};
        for (double &v : uv) v += v;
        t[0] += (tl[0] + w * uv[0]) + (uy * uv[2] - uz * uv[1]);
        t[1] += (tl[1] + w * uv[1]) + (uz * uv[0] - ux * uv[2]);
        t[2] += (tl[2] + w * uv[2]) + (ux * uv[1] - uy * uv[0]);
        const double ax = q[0], ay = q[1], az = q[2], aw = q[3], bx = ql[0], by = ql[1], bz = ql[2], bw = ql[3];
        q[3] = aw * bw - ax * bx - ay * by - az * bz;
        q[0] = aw * bx + ax * bw + ay * bz - az * by;
        q[1] = aw * by - ax * bz + ay * bw + az * bx;
        q[2] = aw * bz + ax * by - ay * bx + az * bw;
    }
    void matrix(double H[12]) const {                      /* 3x4 row-major [R | t], Eigen toRotationMatrix() */
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        H[0] = 1 - 2 * (y * y + z * z); H[1] = 2 * (x * y - z * w);     H[2] = 2 * (x * z + y * w);     H[3] = t[0];
        H[4] = 2 * (x * y + z * w);     H[5] = 1 - 2 * (x * x + z * z); H[6] = 2 * (y * z - x * w);     H[7] = t[1];
        H[8] = 2 * (x * z - y * w);     H[9] = 2 * (y * z + x * w);     H[10] = 1 - 2 * (x * x + y * y); H[11] = t[2];
    }
};

/* The evaluation artefact of src/laserMapping.cpp:2284-2325: one line per frame, H_init^-1 * H as 12 values in
 * scientific notation with precision 6, appended to RESULT_PATH. */
class TrajectoryWriter {
public:
    explicit TrajectoryWriter(const std::string &result_path) : path_(result_path) {}
    void append(const WorldPose &p) {
        double H[12]; p.matrix(H);
        if (init_flag_) { for (int i = 0; i < 12; ++i) Hinit_[i] = H[i]; init_flag_ = false; }
        /* H_init^-1 * H for rigid transforms: R0^T R, R0^T (t - t0) */
        double out[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) out[r * 4 + c] = Hinit_[0 * 4 + r] * H[0 * 4 + c] + Hinit_[1 * 4 + r] * H[1 * 4 + c] + Hinit_[2 * 4 + r] * H[2 * 4 + c];
            out[r * 4 + 3] = Hinit_[0 * 4 + r] * (H[3] - Hinit_[3]) + Hinit_[1 * 4 + r] * (H[7] - Hinit_[7]) + Hinit_[2 * 4 + r] * (H[11] - Hinit_[11]);
        }
        std::FILE *f = std::fopen(path_.c_str(), "a");
        if (!f) throw Error(LL_ERR_ARG, "cannot open " + path_);
        for (int i = 0; i < 12; ++i) std::fprintf(f, i == 11 ? "%.6e\n" : "%.6e ", out[i]);
        std::fclose(f);
    }
private:
    std::string path_;
    bool init_flag_ = true;
    double Hinit_[12];
};

/* laserOdometry over a whole sequence of registered scans in slots [first, first+count): the reference's frame loop
 * (3 outer iterations x ceres::Solve, vote from the 6th frame, warm start) -- ll_odometry_frames. */
inline std::vector<double> odometry_frames(Context &c, int first, int count, const double *pose0 = nullptr, int first_frame_index = 1)
{
    std::vector<double> rel((size_t)count * 7);
    c.check(ll_odometry_frames(c.get(), first, count, pose0, 3, first_frame_index, nullptr, rel.data()));
    return rel;
}

}  // namespace lightloam

/* ------------------------------------------------------------------------------------------------------------------
 * Optional Ceres adapter (NOT compiled in this image: Ceres is absent).  Keeps the parameter-block layout (4, 3) of
 * lidarFactor.hpp and lets ceres::Solve drive the device evaluation: ONE cost function for all residual blocks of the
 * frame, whose Evaluate() is ll_residual_jacobian (rows = 3 * edges + selected planes, jacobians[0] rows x 4 over
 * (x,y,z,w), jacobians[1] rows x 3, row-major).  HuberLoss(0.1) is applied per reference residual block, so the batched
 * function must be wrapped with LL huber off and a ceres::LossFunction cannot be used as is -- see INTEGRATION.md.   */
#if defined(LIGHTLOAM_WITH_CERES) && __has_include(<ceres/ceres.h>)
#include <ceres/ceres.h>
namespace lightloam {
class BatchedLidarCost : public ceres::CostFunction {
public:
    BatchedLidarCost(Context &c, int slot, int rows) : c_(c), slot_(slot) {
        set_num_residuals(rows);
        mutable_parameter_block_sizes()->push_back(4);
        mutable_parameter_block_sizes()->push_back(3);
        jq_.resize((size_t)rows * 4); jt_.resize((size_t)rows * 3);
    }
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override {
        const double pose[7] = {parameters[0][0], parameters[0][1], parameters[0][2], parameters[0][3],
                                parameters[1][0], parameters[1][1], parameters[1][2]};
        if (ll_residual_jacobian(c_.get(), slot_, pose, residuals, jq_.data(), jt_.data(), num_residuals()) != LL_OK) return false;
        if (jacobians && jacobians[0]) std::copy(jq_.begin(), jq_.end(), jacobians[0]);
        if (jacobians && jacobians[1]) std::copy(jt_.begin(), jt_.end(), jacobians[1]);
        return true;
    }
private:
    Context &c_; int slot_;
    mutable std::vector<double> jq_, jt_;
};

/* the same for the mapping blocks of a MapOptimizer after ll_map_associate (LidarEdgeFactor + LidarPlaneNormFactor rows,
 * laserMapping.cpp:1918-1919, :2033-2034); parameter blocks (4, 3) = parameters, parameters + 4 (:1871-1872) */
class BatchedMapCost : public ceres::CostFunction {
public:
    BatchedMapCost(MapOptimizer &m, int rows) : m_(m) {
        set_num_residuals(rows);
        mutable_parameter_block_sizes()->push_back(4);
        mutable_parameter_block_sizes()->push_back(3);
        jq_.resize((size_t)rows * 4); jt_.resize((size_t)rows * 3);
    }
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override {
        const double pose[7] = {parameters[0][0], parameters[0][1], parameters[0][2], parameters[0][3],
                                parameters[1][0], parameters[1][1], parameters[1][2]};
        if (ll_map_residual_jacobian(m_.get(), pose, residuals, jq_.data(), jt_.data(), num_residuals()) != LL_OK) return false;
        if (jacobians && jacobians[0]) std::copy(jq_.begin(), jq_.end(), jacobians[0]);
        if (jacobians && jacobians[1]) std::copy(jt_.begin(), jt_.end(), jacobians[1]);
        return true;
    }
private:
    MapOptimizer &m_;
    mutable std::vector<double> jq_, jt_;
};
}  // namespace lightloam
#endif

#endif /* LIGHTLOAM_HOST_HPP */
