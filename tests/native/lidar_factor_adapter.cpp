// The lidarFactor.hpp drop-in (include/lightloam_lidarFactor.hpp) used the way the reference nodes use the original:
//   lidar_factor_adapter <blocks.bin> <out.bin>
// blocks.bin: int32 n_edge, n_plane, n_pnorm, then f64 edge[n][9], plane[n][13], pnorm[n][7], then two poses (7 f64 each).
// Builds the cost functions with LidarEdgeFactor::Create / LidarPlaneFactor_modify::Create / LidarPlaneNormFactor::Create
// in the interleaved order a node would (edge, plane, plane-norm, edge, ...), evaluates every one at pose 0 (with
// Jacobians), again at pose 0 without Jacobians, then at pose 1, and dumps residuals + Jacobians per cost function in
// creation order; the Python test compares them with the oracle's functors.
//
// Ceres is not in this image.  The block below is a TEST DOUBLE of the single Ceres class the adapter derives from --
// just the members lightloam_lidarFactor.hpp touches -- so that the adapter's own logic (batching, caching, row
// bookkeeping, the Create signatures) can be compiled and exercised here.  It is not a Ceres build and nothing of the
// reference is compiled against it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace ceres {
class CostFunction {
public:
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    int num_residuals() const { return num_residuals_; }
    const std::vector<int32_t> &parameter_block_sizes() const { return sizes_; }
protected:
    void set_num_residuals(int n) { num_residuals_ = n; }
    std::vector<int32_t> *mutable_parameter_block_sizes() { return &sizes_; }
private:
    int num_residuals_ = 0;
    std::vector<int32_t> sizes_;
};
}  // namespace ceres

#define LIGHTLOAM_CERES_TEST_DOUBLE
#include "lightloam_lidarFactor.hpp"

struct Vec3 {                                                    // stands in for Eigen::Vector3d: x(), y(), z()
    double v[3];
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
};
static Vec3 v3(const double *p) { return Vec3{{p[0], p[1], p[2]}}; }

static double block_s(bool on, int i) { return on ? 0.25 + 0.5 * ((i % 7) / 7.0) : 1.0; }   // DISTORTION 1: the functors' s_

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const bool with_s = argc > 3 && std::string(argv[3]) == "s";
    std::ifstream f(argv[1], std::ios::binary);
    int32_t n[3];
    f.read((char *)n, sizeof(n));
    std::vector<double> edge((size_t)n[0] * 9), plane((size_t)n[1] * 13), pnorm((size_t)n[2] * 7), poses(14);
    f.read((char *)edge.data(), edge.size() * 8); f.read((char *)plane.data(), plane.size() * 8);
    f.read((char *)pnorm.data(), pnorm.size() * 8); f.read((char *)poses.data(), 14 * 8);
    if (!f) { std::cerr << "short input\n"; return 2; }
    try {
        lightloam::Context ctx(16, 1);
        // without a current batch Create() must refuse
        bool refused = false;
        try { LidarEdgeFactor::Create(v3(&edge[0]), v3(&edge[3]), v3(&edge[6]), 1.0); } catch (const lightloam::Error &e) { refused = e.code == LL_ERR_STATE; }
        if (!refused) { std::cerr << "Create without a batch did not throw\n"; return 1; }
        lightloam::FactorBatch batch(ctx);
        std::vector<std::unique_ptr<ceres::CostFunction>> costs;
        std::vector<int> kind;
        {
            lightloam::FactorBatch::Current use(batch);
            const int most = std::max(n[0], std::max(n[1], n[2]));
            for (int i = 0; i < most; ++i) {                       // interleaved, like a node with several loops would not -- the harder case
                if (i < n[0]) { const double *e = &edge[(size_t)i * 9]; costs.emplace_back(LidarEdgeFactor::Create(v3(e), v3(e + 3), v3(e + 6), block_s(with_s, i))); kind.push_back(0); }
                if (i < n[1]) { const double *p = &plane[(size_t)i * 13]; costs.emplace_back(LidarPlaneFactor_modify::Create(v3(p), v3(p + 3), v3(p + 6), v3(p + 9), block_s(with_s, i), p[12])); kind.push_back(1); }
                if (i < n[2]) { const double *p = &pnorm[(size_t)i * 7]; costs.emplace_back(LidarPlaneNormFactor::Create(v3(p), v3(p + 3), p[6])); kind.push_back(2); }
            }
        }
        if (lightloam::FactorBatch::current() != nullptr) { std::cerr << "Current not restored\n"; return 1; }
        std::vector<double> out;
        for (int pass = 0; pass < 3; ++pass) {
            const double *pose = &poses[(pass == 2) ? 7 : 0];
            const double *params[2] = {pose, pose + 4};
            for (size_t c = 0; c < costs.size(); ++c) {
                const int rows = costs[c]->num_residuals();
                if (rows != (kind[c] == 0 ? 3 : 1) || costs[c]->parameter_block_sizes().size() != 2 ||
                    costs[c]->parameter_block_sizes()[0] != 4 || costs[c]->parameter_block_sizes()[1] != 3) { std::cerr << "bad block shape\n"; return 1; }
                double r[3], jq[12], jt[9];
                double *jac[2] = {jq, jt};
                if (!costs[c]->Evaluate(params, r, pass == 1 ? nullptr : jac)) return 1;
                out.insert(out.end(), r, r + rows);
                if (pass != 1) { out.insert(out.end(), jq, jq + 4 * rows); out.insert(out.end(), jt, jt + 3 * rows); }
            }
            const int want = (pass == 2) ? 2 : 1;                 // one launch per distinct parameter point
            if (batch.evaluations() != want) { std::cerr << "evaluations " << batch.evaluations() << " != " << want << "\n"; return 1; }
        }
        std::ofstream o(argv[2], std::ios::binary);
        o.write((const char *)out.data(), (std::streamsize)(out.size() * 8));
        std::cout << costs.size() << " cost functions, " << batch.evaluations() << " device evaluations\n";
    } catch (const lightloam::Error &e) {
        std::cerr << "lightloam error " << e.code << ": " << e.what() << "\n";
        return 1;
    }
    return 0;
}
