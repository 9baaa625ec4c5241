/*
 * ll_oracle.h -- CPU ORACLE for the Light-LOAM per-scan hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (light-loam_amd/) never
 * links, imports or calls anything in oracle/ and has no CPU fallback.
 *
 * PARITY UNPINNED: the reference (BrenYi/Light-LOAM) ships no tests, golden vectors or
 * fixtures, and none of its translation units can be compiled in this image (every TU
 * needs ROS1 + PCL, and the odometry path also Eigen + Ceres; none are installed and no
 * stand-in headers are written for them).  This file is therefore a *restatement* of the
 * reference algorithm, each function citing the reference file:line it follows; the
 * third-party arithmetic on the path (PCL 1.10 VoxelGrid / KdTreeFLANN->FLANN L2_Simple,
 * Eigen 3.3 Quaternion slerp/_transformVector/cross/normalize, Ceres 2.x Jet, HuberLoss,
 * Corrector, EigenQuaternionManifold) is restated from their published algorithms.
 * What IS pinned here: libm.  The oracle calls the host glibc atanf/atan2f/expf/sqrtf
 * exactly where the reference does.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef LL_ORACLE_H
#define LL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::PointXYZI (include/aloam_velodyne/common.h:7) packed to 16 B */
typedef struct { float x, y, z, intensity; } orc_point;

/* scanRegistration.cpp:435-441 node parameters */
typedef struct {
    int    n_scans;        /* "scan_line" */
    int    ring_model;     /* 0 = reference switch (16/32/64 only); 1 = linear model of the 64 branch for any n_scans (extension) */
    double minimum_range;  /* "minimum_range" */
    float  lower_bound;    /* "lowerBound" (-24.9) */
    float  up_bound;       /* "upBound" (2) */
} orc_params;

enum { ORC_OK = 0, ORC_ERR_EMPTY = -1, ORC_ERR_BAD_RINGS = -2, ORC_ERR_CAPACITY = -3 };

/* ---- a1: removeNaN + removeClosedPointCloud + ring/relTime assignment + stable ring bucket ---- */
/* scanRegistration.cpp:58-85, :105-221 */
int orc_organize(const float *xyz, int stride_floats, int n_in, const orc_params *P,
                 orc_point *cloud, int *n_out, int *scan_start, int *scan_end);

/* test helpers: per-point ring ids of :139-168 without the filtering; the host libm over arrays (see ll_oracle.c) */
void orc_scan_ids(const float *xyz, int stride_floats, int n, const orc_params *P, int *ids);
void orc_libm_batch(int op, const float *a, const float *b, const float *c, int n, float *out);

/* ---- a2: curvature, scanRegistration.cpp:225-235 (curv[i] valid for i in [5, n-5)) ---- */
void orc_curvature(const orc_point *cloud, int n, float *curv);

/* ---- a3 + a4: segment sort, greedy pick, less-flat voxel grid, scanRegistration.cpp:246-377 ---- */
/* label is int32 like cloudLabel[]; entries outside [5,n-5) are left untouched (caller zero-fills). */
int orc_pick(const orc_point *cloud, int n, const float *curv,
             const int *scan_start, const int *scan_end, int n_scans, int *label,
             orc_point *sharp, int *n_sharp, orc_point *less_sharp, int *n_less_sharp,
             orc_point *flat, int *n_flat, orc_point *less_flat, int *n_less_flat,
             int *tie_count /* nullable: # of equal-curvature neighbours met in the sorted order */);

/* pcl::VoxelGrid<PointXYZI>::applyFilter restated (PCL 1.10), leaf (l,l,l), downsample_all_data = true */
int orc_voxel_grid(const orc_point *in, int n, float leaf, orc_point *out, int *n_out);

/* whole laserCloudHandler, scanRegistration.cpp:87-428 (without the ROS messages) */
int orc_extract(const float *xyz, int stride_floats, int n_in, const orc_params *P,
                orc_point *cloud, int *n_out, int *scan_start, int *scan_end,
                float *curv, int *label,
                orc_point *sharp, int *n_sharp, orc_point *less_sharp, int *n_less_sharp,
                orc_point *flat, int *n_flat, orc_point *less_flat, int *n_less_flat);

/* DISTORTION (laserOdometry.cpp:23): 0 = the reference's build (default); 1 = its other compile-time path: the interpolation ratio
 * s = (intensity - int(intensity)) / SCAN_PERIOD of every point in TransformToStart (:81-82) and in the two odometry factors (:570, :740) */
void orc_set_distortion(int on);
int orc_get_distortion(void);
double orc_point_s(const orc_point *p);

/* ---- a5: TransformToStart, laserOdometry.cpp:77-95 ---- */
void orc_transform_to_start(const double q[4] /*x,y,z,w*/, const double t[3], const orc_point *pi, orc_point *po);

/* K=1 search strategy of a6/a7: 0 = linear scan (default; the plainest statement of "exact NN"),
 * 1 = uniform grid with exact pruning (same results; used when the oracle is TIMED as the CPU baseline so
 * it has kd-tree-class cost like the reference's PCL KdTreeFLANN). */
void orc_set_nn_mode(int mode);

/* ---- a6: corner association, laserOdometry.cpp:491-620.  One entry per accepted correspondence. ---- */
int orc_associate_corner(const double q[4], const double t[3], const orc_point *sharp, int ns,
                         const orc_point *corner_last, int mc,
                         int *src_idx, int *idx_a, int *idx_b, int *n_e);
/* ---- a7: plane association, laserOdometry.cpp:653-793 ---- */
int orc_associate_plane(const double q[4], const double t[3], const orc_point *flat, int nf,
                        const orc_point *surf_last, int ms,
                        int *src_idx, int *idx_a, int *idx_b, int *idx_c, int *n_p);

/* ---- a8: graph_based_correspondence_vote_simple, laserOdometry.cpp:153-342 ---- */
/* counts[n] = incompatibility count per correspondence; selected (index, weight) in region order then
 * ascending (count, index) -- the reference's order inside equal counts is std::sort-unspecified. */
void orc_vote(const orc_point *src, const orc_point *tgt, int n, int corner_case,
              int *counts, int *sel_idx, float *sel_w, int *n_sel);

/* ---- a9: cost functors evaluated the way ceres::AutoDiffCostFunction does (forward Jets, width 7) ---- */
/* Jq is row-major (rows x 4) w.r.t. q = (x,y,z,w); Jt row-major (rows x 3). */
void orc_edge_factor(const double q[4], const double t[3], const double cp[3], const double a[3],
                     const double b[3], double s, double r[3], double Jq[12], double Jt[9]);      /* lidarFactor.hpp:9-52 */
void orc_plane_factor_modify(const double q[4], const double t[3], const double cp[3], const double j[3],
                     const double l[3], const double m[3], double s, double weight,
                     double r[1], double Jq[4], double Jt[3]);                                    /* lidarFactor.hpp:203-251 */
void orc_plane_norm_factor(const double q[4], const double t[3], const double cp[3], const double n[3],
                     double negative_OA_dot_norm, double r[1], double Jq[4], double Jt[3]);      /* lidarFactor.hpp:253-285 */
/* ceres EigenQuaternionManifold::PlusJacobian (4x3 row-major) and Plus */
void orc_quat_plus_jacobian(const double q[4], double P[12]);
void orc_quat_plus(const double q[4], const double delta[3], double q_out[4]);

/* ---- a10: HuberLoss(0.1) corrector + normal equations + one Gauss-Newton step ---- */
/* laserOdometry.cpp:475-482, :615-616, :797-808, :820-825.  Blocks: n_e edges (3 rows each) then n_p planes.
 * H is 6x6 row-major over (dq[3], dt[3]), g = J^T r, cost = sum 0.5*rho(s). */
void orc_normal_equations(const double q[4], const double t[3],
                          const orc_point *sharp, const int *e_src, const orc_point *corner_last,
                          const int *e_a, const int *e_b, int n_e,
                          const orc_point *flat, const int *p_src, const orc_point *surf_last,
                          const int *p_a, const int *p_b, const int *p_c, const float *p_w, int n_p,
                          double huber_delta /* <=0: no loss */, double H[36], double g[6], double *cost);
int orc_gn_solve(const double H[36], const double g[6], double delta[6]);   /* H delta = -g, Cholesky */
void orc_pose_update(double q[4], double t[3], const double delta[6]);

/* ---- f1: the solver the reference actually runs, ceres::Solve with DENSE_QR, max_num_iterations = 4 and otherwise
 * default options (laserOdometry.cpp:820-825): trust-region minimizer + Levenberg-Marquardt strategy, restated from
 * Ceres 2.x (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc) -- PARITY UNPINNED like the rest.
 * The residual blocks are fixed (one outer iteration's correspondences); q, t are updated in place.
 * summary[0] = initial cost, [1] = final cost, [2] = iterations run, [3] = successful steps. */
typedef struct {
    int    max_num_iterations;        /* 4 (:822) */
    double initial_radius;            /* 1e4  */
    double max_radius, min_radius;    /* 1e16, 1e-32 */
    double min_relative_decrease;     /* 1e-3 */
    double min_lm_diagonal, max_lm_diagonal;   /* 1e-6, 1e32 */
    double function_tolerance, gradient_tolerance, parameter_tolerance;   /* 1e-6, 1e-10, 1e-8 */
    int    jacobi_scaling;            /* 1 */
} orc_lm_options;
void orc_lm_default(orc_lm_options *o);
void orc_lm_solve(double q[4], double t[3],
                  const orc_point *sharp, const int *e_src, const orc_point *corner_last, const int *e_a, const int *e_b, int n_e,
                  const orc_point *flat, const int *p_src, const orc_point *surf_last,
                  const int *p_a, const int *p_b, const int *p_c, const float *p_w, int n_p,
                  double huber_delta, const orc_lm_options *opt, double summary[4]);

/* One frame of laserOdometry (:439-832): n_outer (3) x { correspondence search, vote when `vote`, LM solve }.
 * q, t = para_q / para_t, warm-started by the caller (they persist between frames, :61-65). */
void orc_odometry_frame(double q[4], double t[3], const orc_point *sharp, int ns, const orc_point *flat, int nf,
                        const orc_point *corner_last, int mc, const orc_point *surf_last, int ms,
                        int vote, int n_outer, double huber_delta, const orc_lm_options *opt);

/* ---- f2: laserMapping scan-to-submap optimisation (laserMapping.cpp:1822-2095).  PARITY UNPINNED like the rest:
 * Eigen's SelfAdjointEigenSolver / ColPivHouseholderQR and PCL's kd-tree are restated (Jacobi sweeps, Householder QR with
 * column pivoting, exact 5-NN with (distance, index) order).  Pose = parameters[7] of the reference: q_w_curr, t_w_curr. */
void orc_point_associate_to_map(const double q[4], const double t[3], const orc_point *pi, orc_point *po);   /* :125-134 */
void orc_sym_eig3(const double A[9], double w[3], double V[9]);               /* eigenvalues ascending, eigenvectors in columns */
void orc_qr_solve_5x3(const double A[15], const double b[5], double x[3]);
int orc_map_associate(const double q[4], const double t[3],
                      const orc_point *corner_stack, int n_cs, const orc_point *corner_map, int n_cm,
                      const orc_point *surf_stack, int n_ss, const orc_point *surf_map, int n_sm,
                      int *e_src, double *e_a, double *e_b, int *n_e, int *p_src, double *p_n, double *p_d, int *n_p);
void orc_map_normal_equations(const double q[4], const double t[3],
                              const orc_point *corner_stack, const int *e_src, const double *e_a, const double *e_b, int n_e,
                              const orc_point *surf_stack, const int *p_src, const double *p_n, const double *p_d, int n_p,
                              double huber_delta, double H[36], double g[6], double *cost);
int orc_map_optimize(double q[4], double t[3],
                     const orc_point *corner_stack, int n_cs, const orc_point *corner_map, int n_cm,
                     const orc_point *surf_stack, int n_ss, const orc_point *surf_map, int n_sm,
                     int n_outer, double huber_delta, const orc_lm_options *opt);

/* ---- f2, second stage: the cube map around the optimisation (laserMapping.cpp:1584-1821, :2101-2165) ---- */
typedef struct orc_cubemap orc_cubemap;
orc_cubemap *orc_cubemap_create(float line_res /* 0.4 */, float plane_res /* 0.8 */);
void orc_cubemap_destroy(orc_cubemap *m);
void orc_cubemap_prepare(orc_cubemap *m, const double t_w[3], const orc_point *corner_last, int n_corner, const orc_point *surf_last, int n_surf);
int orc_cubemap_optimize(orc_cubemap *m, double q[4], double t[3], int n_outer, double huber_delta, const orc_lm_options *opt);
void orc_cubemap_update(orc_cubemap *m, const double q[4], const double t[3]);
void orc_cubemap_get(const orc_cubemap *m, int which /* 0 corner from map, 1 surf from map, 2 corner stack, 3 surf stack */, const orc_point **p, int *n);
void orc_cubemap_cube(const orc_cubemap *m, int surf, int cube, const orc_point **p, int *n);
void orc_cubemap_center(const orc_cubemap *m, int cen[3]);

#ifdef __cplusplus
}
#endif
#endif
