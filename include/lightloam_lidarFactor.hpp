/*
 * lightloam_lidarFactor.hpp -- drop-in for the reference's src/lidarFactor.hpp: the cost functors keep their names and
 * their static Create(...) signatures, so the call sites
 *     ceres::CostFunction *cost_function = LidarEdgeFactor::Create(curr_point, last_point_a, last_point_b, s);      // laserOdometry.cpp:615, laserMapping.cpp:1918
 *     ceres::CostFunction *cost_function = LidarPlaneFactor_modify::Create(curr_point, a, b, c, s, weight);         // laserOdometry.cpp:783, :804
 *     ceres::CostFunction *cost_function = LidarPlaneNormFactor::Create(curr_point, norm, negative_OA_dot_norm);    // laserMapping.cpp:2033
 *     problem.AddResidualBlock(cost_function, loss_function, para_q, para_t);
 * stay as they are -- but nothing is auto-differentiated on the CPU.  Create() files the block with the current
 * lightloam::FactorBatch and returns a thin ceres::CostFunction (3 or 1 residuals, parameter blocks 4 and 3, as the
 * AutoDiffCostFunction<..., 4, 3> it replaces).  The first Evaluate() at a new parameter point evaluates ALL blocks of the
 * batch on the GPU in one launch (ll_factor_blocks_evaluate: residuals + analytic Jacobians w.r.t. q (x, y, z, w) and t);
 * the other blocks' Evaluate() calls at that point copy their rows out of the cached result.  Ceres evaluates every
 * residual block at the same point, so one ceres iteration = one launch.
 *
 * What a maintainer changes in the node (per outer iteration, next to "ceres::Problem problem(problem_options)",
 * laserOdometry.cpp:478, laserMapping.cpp:1869):
 *     lightloam::FactorBatch batch(*g_ll);             // g_ll: the node's lightloam::Context
 *     lightloam::FactorBatch::Current use(batch);      // Create() files blocks here until `use` goes out of scope
 *     ... the correspondence loops, unchanged ...
 *     ceres::Solve(options, &problem, &summary);       // unchanged; `batch` must outlive the problem
 *
 * s_ is honoured per block (1.0 in the reference's build -- DISTORTION 0, laserOdometry.cpp:23, :81-84 -- and then the blocks
 * take the closed-form s = 1 path; any other value goes through Identity.slerp(s, q), s * t like lidarFactor.hpp:25-27).
 * Restriction, checked loudly: all cost functions of a batch must be evaluated with the same parameter blocks (they are:
 * para_q / para_t, or parameters / parameters + 4).
 * Vector arguments: anything with x(), y(), z() (Eigen::Vector3d).
 * Needs <ceres/ceres.h>; this image has no Ceres, so tests compile the header against a declared test double of
 * ceres::CostFunction (tests/native/lidar_factor_adapter.cpp, -DLIGHTLOAM_CERES_TEST_DOUBLE).
 */
#ifndef LIGHTLOAM_LIDARFACTOR_HPP
#define LIGHTLOAM_LIDARFACTOR_HPP

#include <cstring>
#include <vector>

#include "lightloam_host.hpp"

#if !defined(LIGHTLOAM_CERES_TEST_DOUBLE)
#include <ceres/ceres.h>
#endif

namespace lightloam {

class FactorBatch {
public:
    explicit FactorBatch(Context &c) : c_(c) {}
    FactorBatch(const FactorBatch &) = delete;
    FactorBatch &operator=(const FactorBatch &) = delete;

    /* RAII "current batch" of this thread: Create() has no argument to carry it */
    class Current {
    public:
        explicit Current(FactorBatch &b) : prev_(current()) { current() = &b; }
        ~Current() { current() = prev_; }
        Current(const Current &) = delete;
        Current &operator=(const Current &) = delete;
    private:
        FactorBatch *prev_;
    };
    static FactorBatch *&current() { static thread_local FactorBatch *cur = nullptr; return cur; }
    static FactorBatch &require() {
        if (!current()) throw Error(LL_ERR_STATE, "no lightloam::FactorBatch::Current in scope when a cost functor was created");
        return *current();
    }

    enum Kind { EDGE = 0, PLANE = 1, PNORM = 2 };
    /* file one block; returns its index inside its kind */
    int add(Kind k, const double *data, int n, double s = 1.0) {
        std::vector<double> &v = blocks_[k];
        v.insert(v.end(), data, data + n);
        if (k != PNORM) { s_[k].push_back(s); if (s != 1.0) any_s_ = true; }
        uploaded_ = false; have_ = false;
        return count_[k]++;
    }
    int evaluations() const { return evaluations_; }               /* device launches so far (tests) */

    /* rows of block `index` of kind k at (q, t): residuals and, when asked for, the two Jacobian blocks */
    void rows(Kind k, int index, const double *q, const double *t, double *residuals, double *jq, double *jt) {
        if (!have_ || std::memcmp(q, q_, sizeof(q_)) != 0 || std::memcmp(t, t_, sizeof(t_)) != 0) evaluate(q, t);
        const int nrow = (k == EDGE) ? 3 : 1;
        const size_t r0 = (k == EDGE) ? (size_t)3 * index : (k == PLANE) ? (size_t)3 * count_[EDGE] + index
                                                                         : (size_t)3 * count_[EDGE] + count_[PLANE] + index;
        for (int i = 0; i < nrow; ++i) residuals[i] = r_[r0 + i];
        if (jq) std::memcpy(jq, &jq_[(r0) * 4], sizeof(double) * 4 * nrow);
        if (jt) std::memcpy(jt, &jt_[(r0) * 3], sizeof(double) * 3 * nrow);
    }

private:
    void evaluate(const double *q, const double *t) {
        if (!uploaded_) {
            c_.check(ll_factor_blocks_set(c_.get(), count_[EDGE], blocks_[EDGE].data(), count_[PLANE], blocks_[PLANE].data(),
                                          count_[PNORM], blocks_[PNORM].data()));
            if (any_s_) c_.check(ll_factor_blocks_set_s(c_.get(), s_[EDGE].data(), s_[PLANE].data()));   /* the functors' s_ (DISTORTION 1) */
            const size_t rows = (size_t)3 * count_[EDGE] + count_[PLANE] + count_[PNORM];
            r_.resize(rows); jq_.resize(rows * 4); jt_.resize(rows * 3);
            uploaded_ = true;
        }
        c_.check(ll_factor_blocks_evaluate(c_.get(), q, t, r_.data(), jq_.data(), jt_.data(), (int)r_.size()));
        std::memcpy(q_, q, sizeof(q_)); std::memcpy(t_, t, sizeof(t_));
        have_ = true; ++evaluations_;
    }
    Context &c_;
    std::vector<double> blocks_[3];
    std::vector<double> s_[2];                                     /* s_ of every edge / plane block */
    bool any_s_ = false;
    int count_[3] = {0, 0, 0};
    std::vector<double> r_, jq_, jt_;
    double q_[4] = {0, 0, 0, 0}, t_[3] = {0, 0, 0};
    bool uploaded_ = false, have_ = false;
    int evaluations_ = 0;
};

/* the ceres::CostFunction Create() hands out: a view of one block of the batch */
class BatchBlockCost : public ceres::CostFunction {
public:
    BatchBlockCost(FactorBatch &b, FactorBatch::Kind k, int index) : b_(b), k_(k), index_(index) {
        set_num_residuals(k == FactorBatch::EDGE ? 3 : 1);
        mutable_parameter_block_sizes()->push_back(4);
        mutable_parameter_block_sizes()->push_back(3);
    }
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override {
        b_.rows(k_, index_, parameters[0], parameters[1], residuals, jacobians ? jacobians[0] : nullptr, jacobians ? jacobians[1] : nullptr);
        return true;
    }
private:
    FactorBatch &b_;
    FactorBatch::Kind k_;
    int index_;
};

}  // namespace lightloam

/* ---- the reference's names, global namespace like src/lidarFactor.hpp ------------------------------------------- */

struct LidarEdgeFactor {                                              /* lidarFactor.hpp:9-52 */
    template <class V3>
    static ceres::CostFunction *Create(const V3 &curr_point_, const V3 &last_point_a_, const V3 &last_point_b_, const double s_) {
        lightloam::FactorBatch &b = lightloam::FactorBatch::require();
        const double d[9] = {curr_point_.x(), curr_point_.y(), curr_point_.z(), last_point_a_.x(), last_point_a_.y(), last_point_a_.z(),
                             last_point_b_.x(), last_point_b_.y(), last_point_b_.z()};
        return new lightloam::BatchBlockCost(b, lightloam::FactorBatch::EDGE, b.add(lightloam::FactorBatch::EDGE, d, 9, s_));
    }
};

struct LidarPlaneFactor_modify {                                      /* lidarFactor.hpp:203-251 */
    template <class V3>
    static ceres::CostFunction *Create(const V3 &curr_point_, const V3 &last_point_j_, const V3 &last_point_l_, const V3 &last_point_m_,
                                       const double s_, const double weight_) {
        lightloam::FactorBatch &b = lightloam::FactorBatch::require();
        const double d[13] = {curr_point_.x(), curr_point_.y(), curr_point_.z(), last_point_j_.x(), last_point_j_.y(), last_point_j_.z(),
                              last_point_l_.x(), last_point_l_.y(), last_point_l_.z(), last_point_m_.x(), last_point_m_.y(), last_point_m_.z(), weight_};
        return new lightloam::BatchBlockCost(b, lightloam::FactorBatch::PLANE, b.add(lightloam::FactorBatch::PLANE, d, 13, s_));
    }
};

struct LidarPlaneNormFactor {                                         /* lidarFactor.hpp:253-285 */
    template <class V3>
    static ceres::CostFunction *Create(const V3 &curr_point_, const V3 &plane_unit_norm_, const double negative_OA_dot_norm_) {
        lightloam::FactorBatch &b = lightloam::FactorBatch::require();
        const double d[7] = {curr_point_.x(), curr_point_.y(), curr_point_.z(), plane_unit_norm_.x(), plane_unit_norm_.y(), plane_unit_norm_.z(),
                             negative_OA_dot_norm_};
        return new lightloam::BatchBlockCost(b, lightloam::FactorBatch::PNORM, b.add(lightloam::FactorBatch::PNORM, d, 7));
    }
};

#endif /* LIGHTLOAM_LIDARFACTOR_HPP */
