/*
 * ll_factor_math.h -- device functions shared by the odometry (ll_factors.hip) and mapping (ll_mapping.hip) residual
 * kernels: the pose action and its Jacobian, the cost functors of lidarFactor.hpp in closed form, Ceres' Huber
 * corrector, the 21 + 6 + 1 normal-equation accumulators, the 6 x 6 Cholesky solve and EigenQuaternionManifold::Plus.
 * See ll_factors.hip for the derivations.  f64 throughout.
 */
#pragma once
#include "ll_common.h"

struct Pose { double q[4], t[3]; };

__device__ __forceinline__ void ll_lp_and_jac(const Pose &P, const double v[3], double lp[3], double Jq[3][4])
{
    const double ux = P.q[0], uy = P.q[1], uz = P.q[2], w = P.q[3];
    double uvx = uy * v[2] - uz * v[1], uvy = uz * v[0] - ux * v[2], uvz = ux * v[1] - uy * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    lp[0] = ((v[0] + w * uvx) + (uy * uvz - uz * uvy)) + P.t[0];
    lp[1] = ((v[1] + w * uvy) + (uz * uvx - ux * uvz)) + P.t[1];
    lp[2] = ((v[2] + w * uvz) + (ux * uvy - uy * uvx)) + P.t[2];
    const double udv = ux * v[0] + uy * v[1] + uz * v[2];
    const double u[3] = {ux, uy, uz};
    /* d lp / d u = -2w [v]x + 2((u.v) I + u v^T - 2 v u^T) */
    const double vx[3][3] = {{0.0, -v[2], v[1]}, {v[2], 0.0, -v[0]}, {-v[1], v[0], 0.0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Jq[i][j] = -2.0 * w * vx[i][j] + 2.0 * ((i == j ? udv : 0.0) + u[i] * v[j] - 2.0 * v[i] * u[j]);
    Jq[0][3] = uvx; Jq[1][3] = uvy; Jq[2][3] = uvz;              /* d lp / d w = 2 u x v */
}

/* DISTORTION 1 (laserOdometry.cpp:23 is 0 in the reference's build; its other compile-time path): every point carries an
 * interpolation ratio s = (intensity - int(intensity)) / SCAN_PERIOD (:81-82, :570-571, :740-741; float subtraction, f64
 * division by 0.1) and is transformed by  Identity.slerp(s, q) * p + s * t  (:86-88, lidarFactor.hpp:25-27). */
__device__ __forceinline__ double ll_point_s(int distortion, const float4 c)
{
    if (!distortion) return 1.0;
    return (double)(c.w - (float)(int)c.w) / 0.1;
}

/* Eigen 3.3 QuaternionBase::slerp(s, q) called on Identity (coefficients x, y, z, w), and -- when J is not null -- its
 * Jacobian d qs / d q (4 x 4, q's four coefficients independent, like the Jets of ceres::AutoDiffCostFunction see them):
 *   d = q.w, theta = acos|d|, scale0 = sin((1 - s) theta) / sin theta, scale1 = +-sin(s theta) / sin theta (sign of d),
 *   qs = (scale1 q.xyz, scale0 + scale1 q.w); both scales depend on q.w only.  |d| >= 1 - eps: scale0 = 1 - s, scale1 = +-s. */
__device__ __forceinline__ void ll_slerp_identity(double s, const double q[4], double qs[4], double (*J)[4])
{
    const double one = 1.0 - 2.220446049250313e-16;
    const double d = (0.0 * q[0] + 0.0 * q[1]) + (0.0 * q[2] + 1.0 * q[3]);
    const double absD = fabs(d), sgn = d < 0.0 ? -1.0 : 1.0;
    double scale0, scale1, ds0 = 0.0, ds1 = 0.0;                  /* ds*: derivative with respect to q.w */
    if (absD >= one) { scale0 = 1.0 - s; scale1 = s; }
    else {
        const double theta = acos(absD), sinT = sin(theta), cosT = cos(theta);
        const double a0 = (1.0 - s) * theta, a1 = s * theta;
        scale0 = sin(a0) / sinT; scale1 = sin(a1) / sinT;
        const double dth = -sgn / sqrt(1.0 - absD * absD);      /* d theta / d q.w */
        ds0 = ((1.0 - s) * cos(a0) * sinT - sin(a0) * cosT) / (sinT * sinT) * dth;
        ds1 = (s * cos(a1) * sinT - sin(a1) * cosT) / (sinT * sinT) * dth;
    }
    if (d < 0.0) { scale1 = -scale1; ds1 = -ds1; }
    qs[0] = scale0 * 0.0 + scale1 * q[0]; qs[1] = scale0 * 0.0 + scale1 * q[1];
    qs[2] = scale0 * 0.0 + scale1 * q[2]; qs[3] = scale0 * 1.0 + scale1 * q[3];
    if (J) {
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) J[i][k] = (i == k) ? scale1 : 0.0;
        J[0][3] = ds1 * q[0]; J[1][3] = ds1 * q[1]; J[2][3] = ds1 * q[2];
        J[3][3] = ds0 + scale1 + ds1 * q[3];
    }
}

/* lp = slerp(s, q) * v + s * t and d lp / d q (3 x 4); d lp / d t = s * I.  s == 1 is the reference's own build: slerp(1, q) = +-q
 * and the rotation is the same for q and -q, so that case takes ll_lp_and_jac itself (bit for bit what it always computed). */
__device__ __forceinline__ void ll_lp_and_jac_s(const Pose &P, const double v[3], double s, double lp[3], double Jq[3][4])
{
    if (s == 1.0) { ll_lp_and_jac(P, v, lp, Jq); return; }
    Pose Ps; double D[4][4];
    ll_slerp_identity(s, P.q, Ps.q, D);
    Ps.t[0] = s * P.t[0]; Ps.t[1] = s * P.t[1]; Ps.t[2] = s * P.t[2];
    double A[3][4];
    ll_lp_and_jac(Ps, v, lp, A);                                 /* d lp / d qs */
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 4; ++k) Jq[i][k] = (A[i][0] * D[0][k] + A[i][1] * D[1][k]) + (A[i][2] * D[2][k] + A[i][3] * D[3][k]);
}

/* rows x 4 ambient -> rows x 3 tangent: J * PlusJacobian(q), P rows x:[w,z,-y] y:[-z,w,x] z:[y,-x,w] w:[-x,-y,-z] */
__device__ __forceinline__ void ll_to_local(const Pose &P, const double Ja[4], double Jl[3])
{
    const double x = P.q[0], y = P.q[1], z = P.q[2], w = P.q[3];
    Jl[0] = Ja[0] * w - Ja[1] * z + Ja[2] * y - Ja[3] * x;
    Jl[1] = Ja[0] * z + Ja[1] * w - Ja[2] * x - Ja[3] * y;
    Jl[2] = -Ja[0] * y + Ja[1] * x + Ja[2] * w - Ja[3] * z;
}

/* LidarEdgeFactor (lidarFactor.hpp:9-52): r[3], ambient Jq[3][4], Jt[3][3]; a, b = the two points of the line in f64 */
__device__ __forceinline__ void ll_edge_dd(const Pose &P, const double cp[3], const double a[3], const double b[3],
                                           double r[3], double Jq[3][4], double Jt[3][3], double s = 1.0)
{
    double lp[3], A[3][4];
    ll_lp_and_jac_s(P, cp, s, lp, A);
    const double pa[3] = {lp[0] - a[0], lp[1] - a[1], lp[2] - a[2]}, pb[3] = {lp[0] - b[0], lp[1] - b[1], lp[2] - b[2]};
    const double nu[3] = {pa[1] * pb[2] - pa[2] * pb[1], pa[2] * pb[0] - pa[0] * pb[2], pa[0] * pb[1] - pa[1] * pb[0]};   /* :32 */
    const double de[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};                                                        /* :33 */
    const double n = sqrt(de[0] * de[0] + (de[1] * de[1] + de[2] * de[2]));
    r[0] = nu[0] / n; r[1] = nu[1] / n; r[2] = nu[2] / n;                                                                /* :35-37 */
    /* d r / d lp = [b - a]x / n = -[de]x / n */
    const double D[3][3] = {{0.0, de[2] / n, -de[1] / n}, {-de[2] / n, 0.0, de[0] / n}, {de[1] / n, -de[0] / n, 0.0}};
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) Jq[i][k] = D[i][0] * A[0][k] + D[i][1] * A[1][k] + D[i][2] * A[2][k];
        for (int k = 0; k < 3; ++k) Jt[i][k] = (s == 1.0) ? D[i][k] : D[i][k] * s;    /* t_last_curr = s * t (:27) */
    }
}

__device__ __forceinline__ void ll_edge_d(const Pose &P, const float4 c, const double a[3], const double b[3],
                                          double r[3], double Jq[3][4], double Jt[3][3], double s = 1.0)
{
    const double cp[3] = {c.x, c.y, c.z};
    ll_edge_dd(P, cp, a, b, r, Jq, Jt, s);
}

__device__ __forceinline__ void ll_edge(const Pose &P, const float4 c, const float4 a4, const float4 b4,
                                        double r[3], double Jq[3][4], double Jt[3][3], double s = 1.0)
{
    const double a[3] = {a4.x, a4.y, a4.z}, b[3] = {b4.x, b4.y, b4.z};
    ll_edge_d(P, c, a, b, r, Jq, Jt, s);
}

/* LidarPlaneNormFactor (lidarFactor.hpp:253-285): r = n . (q * cp + t) + d */
__device__ __forceinline__ void ll_plane_norm_dd(const Pose &P, const double cp[3], const double n[3], double d, double &r, double Jq[4], double Jt[3])
{
    double lp[3], A[3][4];
    ll_lp_and_jac(P, cp, lp, A);
    r = (n[0] * lp[0] + (n[1] * lp[1] + n[2] * lp[2])) + d;                                                  /* :270 */
    for (int k = 0; k < 4; ++k) Jq[k] = n[0] * A[0][k] + n[1] * A[1][k] + n[2] * A[2][k];
    for (int k = 0; k < 3; ++k) Jt[k] = n[k];
}

__device__ __forceinline__ void ll_plane_norm(const Pose &P, const float4 c, const double n[3], double d, double &r, double Jq[4], double Jt[3])
{
    const double cp[3] = {c.x, c.y, c.z};
    ll_plane_norm_dd(P, cp, n, d, r, Jq, Jt);
}

/* LidarPlaneFactor_modify (lidarFactor.hpp:203-251) */
__device__ __forceinline__ void ll_plane_dd(const Pose &P, const double cp[3], const double j[3], const double l[3], const double m[3],
                                            double weight, double &r, double Jq[4], double Jt[3], double s = 1.0)
{
    const double a[3] = {j[0] - l[0], j[1] - l[1], j[2] - l[2]};
    const double b[3] = {j[0] - m[0], j[1] - m[1], j[2] - m[2]};
    double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};     /* :210 */
    const double z = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
    if (z > 0.0) { const double nn = sqrt(z); n[0] /= nn; n[1] /= nn; n[2] /= nn; }                      /* :211 normalize() */
    double lp[3], A[3][4];
    ll_lp_and_jac_s(P, cp, s, lp, A);
    const double d[3] = {lp[0] - j[0], lp[1] - j[1], lp[2] - j[2]};
    r = (d[0] * n[0] + (d[1] * n[1] + d[2] * n[2])) * weight;                                              /* :233 */
    for (int k = 0; k < 4; ++k) Jq[k] = (n[0] * A[0][k] + n[1] * A[1][k] + n[2] * A[2][k]) * weight;
    for (int k = 0; k < 3; ++k) Jt[k] = (s == 1.0) ? n[k] * weight : n[k] * s * weight;                   /* t_last_curr = s * t (:221) */
}

__device__ __forceinline__ void ll_plane(const Pose &P, const float4 c, const float4 j4, const float4 l4, const float4 m4,
                                         double weight, double &r, double Jq[4], double Jt[3], double s = 1.0)
{
    const double cp[3] = {c.x, c.y, c.z}, j[3] = {j4.x, j4.y, j4.z}, l[3] = {l4.x, l4.y, l4.z}, m[3] = {m4.x, m4.y, m4.z};
    ll_plane_dd(P, cp, j, l, m, weight, r, Jq, Jt, s);
}

/* ceres HuberLoss(a) + Corrector (rho'' <= 0 branch): scale = sqrt(rho'), cost += rho/2 */
__device__ __forceinline__ double ll_huber_scale(double sq, double a, double &cost)
{
    if (a <= 0.0) { cost += 0.5 * sq; return 1.0; }
    const double b = a * a;
    if (sq > b) {
        const double rn = sqrt(sq);
        cost += 0.5 * (2.0 * a * rn - b);
        double rho1 = a / rn; if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
        return sqrt(rho1);
    }
    cost += 0.5 * sq;
    return 1.0;
}

#define LL_NACC 28   /* 21 upper-triangular H + 6 g + cost */

__device__ __forceinline__ void ll_acc_row(double acc[LL_NACC], const double J[6], double r)
{
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) acc[k++] += J[a] * J[b];
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
}

__device__ __forceinline__ int ll_chol_solve(const double H[36], const double g[6], double d[6])
{
    double Lm[36];
    for (int i = 0; i < 36; ++i) Lm[i] = 0.0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double sacc = H[i * 6 + j];
            for (int k = 0; k < j; ++k) sacc -= Lm[i * 6 + k] * Lm[j * 6 + k];
            if (i == j) { if (!(sacc > 0.0)) return -1; Lm[i * 6 + i] = sqrt(sacc); }
            else Lm[i * 6 + j] = sacc / Lm[j * 6 + j];
        }
    double y[6];
    for (int i = 0; i < 6; ++i) { double sacc = -g[i]; for (int k = 0; k < i; ++k) sacc -= Lm[i * 6 + k] * y[k]; y[i] = sacc / Lm[i * 6 + i]; }
    for (int i = 5; i >= 0; --i) { double sacc = y[i]; for (int k = i + 1; k < 6; ++k) sacc -= Lm[k * 6 + i] * d[k]; d[i] = sacc / Lm[i * 6 + i]; }
    return 0;
}

/* EigenQuaternionManifold::Plus: q+ = [sin|d| d/|d|, cos|d|] (x) q ; t += dt  (laserOdometry.cpp:476-477) */
__device__ __forceinline__ void ll_pose_plus(double *pose, const double d[6])
{
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double ax, ay, az, aw;
    if (n2 != 0.0) { const double n = sqrt(n2), sn = sin(n) / n; aw = cos(n); ax = sn * d[0]; ay = sn * d[1]; az = sn * d[2]; }
    else { aw = 1.0; ax = d[0]; ay = d[1]; az = d[2]; }
    const double bx = pose[0], by = pose[1], bz = pose[2], bw = pose[3];
    pose[3] = aw * bw - ax * bx - ay * by - az * bz;
    pose[0] = aw * bx + ax * bw + ay * bz - az * by;
    pose[1] = aw * by - ax * bz + ay * bw + az * bx;
    pose[2] = aw * bz + ax * by - ay * bx + az * bw;
    pose[4] += d[3]; pose[5] += d[4]; pose[6] += d[5];
}

