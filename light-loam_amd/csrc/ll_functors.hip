/*
 * ll_functors.hip -- the cost functors of /root/reference/src/lidarFactor.hpp for caller-supplied residual blocks:
 * what ceres::AutoDiffCostFunction<LidarEdgeFactor, 3, 4, 3> (:9-52), <LidarPlaneFactor_modify, 1, 4, 3> (:203-251) and
 * <LidarPlaneNormFactor, 1, 4, 3> (:253-285) return from Evaluate() (at s = 1 unless per-block s values were set) -- residuals and the ambient Jacobians
 * d r / d q (x, y, z, w) and d r / d t -- for ALL blocks of a ceres::Problem in one launch.  This is the device side of
 * include/lightloam_lidarFactor.hpp, which keeps the functors' Create(...) call sites of laserOdometry.cpp:615, :783 and
 * laserMapping.cpp:1918, :2033 unchanged.  One thread per block; block data and results are f64 and coalesced by row.
 */
#include "ll_factor_math.h"

/* edge: curr, a, b (9 doubles); plane: curr, j, l, m, weight (13); plane-norm: curr, unit normal, negative_OA_dot_norm (7).
 * rows: 3 per edge, then 1 per plane, then 1 per plane-norm */
__global__ __launch_bounds__(256) void k_factor_blocks(const double *pose, int n_e, const double *edge, int n_p, const double *plane,
                                                       int n_n, const double *pnorm, const double *s_ep, double *r_out, double *Jq_out, double *Jt_out)
{
    Pose P;
    for (int k = 0; k < 4; ++k) P.q[k] = pose[k];
    for (int k = 0; k < 3; ++k) P.t[k] = pose[4 + k];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_e) {
        const double *e = edge + (size_t)i * 9;
        double r[3], Jq[3][4], Jt[3][3];
        ll_edge_dd(P, e, e + 3, e + 6, r, Jq, Jt, s_ep ? s_ep[i] : 1.0);
        for (int row = 0; row < 3; ++row) {
            const size_t R0 = (size_t)3 * i + row;
            r_out[R0] = r[row];
            for (int k = 0; k < 4; ++k) Jq_out[R0 * 4 + k] = Jq[row][k];
            for (int k = 0; k < 3; ++k) Jt_out[R0 * 3 + k] = Jt[row][k];
        }
    } else if (i < n_e + n_p) {
        const int j = i - n_e;
        const double *p = plane + (size_t)j * 13;
        double r, Jq[4], Jt[3];
        ll_plane_dd(P, p, p + 3, p + 6, p + 9, p[12], r, Jq, Jt, s_ep ? s_ep[i] : 1.0);
        const size_t R0 = (size_t)3 * n_e + j;
        r_out[R0] = r;
        for (int k = 0; k < 4; ++k) Jq_out[R0 * 4 + k] = Jq[k];
        for (int k = 0; k < 3; ++k) Jt_out[R0 * 3 + k] = Jt[k];
    } else if (i < n_e + n_p + n_n) {
        const int j = i - n_e - n_p;
        const double *p = pnorm + (size_t)j * 7;
        double r, Jq[4], Jt[3];
        ll_plane_norm_dd(P, p, p + 3, p[6], r, Jq, Jt);
        const size_t R0 = (size_t)3 * n_e + n_p + j;
        r_out[R0] = r;
        for (int k = 0; k < 4; ++k) Jq_out[R0 * 4 + k] = Jq[k];
        for (int k = 0; k < 3; ++k) Jt_out[R0 * 3 + k] = Jt[k];
    }
}

void ll_launch_factor_blocks(const double *pose, int n_e, const double *edge, int n_p, const double *plane, int n_n, const double *pnorm,
                             const double *s_ep, double *r, double *Jq, double *Jt, hipStream_t st)
{
    const int n = n_e + n_p + n_n;
    if (n > 0) hipLaunchKernelGGL(k_factor_blocks, dim3((n + 255) / 256), dim3(256), 0, st, pose, n_e, edge, n_p, plane, n_n, pnorm, s_ep, r, Jq, Jt);
}
