/*
 * lightloam_hip.h -- C ABI of the MI355X-native (gfx950) Light-LOAM per-scan hot path.
 *
 * The reference (BrenYi/Light-LOAM, paths relative to /root/reference/) has no FFI/plugin interface; the
 * seams a replacement sits behind are its node callback, one free function and the Ceres cost-functor
 * factories.  Each entry point below names the reference code it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no C++/torch types.  Every function returns LL_OK (0) or a
 *    negative ll_status; nothing throws across the boundary.
 *  - ll_ctx owns all device memory and one HIP stream.  One ctx per host thread; not re-entrant
 *    (neither is the reference: file-scope arrays, scanRegistration.cpp:34-40; single-threaded spinner :475).
 *  - A ctx holds `batch` scan SLOTS resident in HBM.  The *_batch entry points run a stage for a
 *    contiguous range of slots in one set of kernel launches (this is how 256 CUs are filled: one scan
 *    is only 64 ring-sized work items).  The single-scan convenience calls at the end operate on slot 0.
 *  - Points are pcl::PointXYZI packed to 16 B: x, y, z, intensity (= scanID + 0.1*relTime,
 *    scanRegistration.cpp:208).  Quaternions are (x, y, z, w) like para_q (laserOdometry.cpp:61).
 *  - "host" pointers are ordinary host memory; they are copied on the ctx stream and the call returns
 *    after the copy completed.  Device-resident use: upload once, run stages, download what is needed.
 *  - There is no CPU fallback: if no gfx950 device/HIP runtime is usable, ll_create fails with LL_ERR_DEVICE.
 */
#ifndef LIGHTLOAM_HIP_H
#define LIGHTLOAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LL_ABI_VERSION 3          /* 2: ll_params.distortion; 3: ll_params.voxel_sort_ranks, .input_stride_floats */

typedef struct ll_ctx ll_ctx;

typedef struct { float x, y, z, intensity; } ll_point;     /* pcl::PointXYZI, common.h:7 */

typedef enum {
    LL_OK = 0,
    LL_ERR_DEVICE = -1,       /* no usable HIP device / kernel image (product path never falls back to CPU) */
    LL_ERR_ARG = -2,
    LL_ERR_BAD_RINGS = -3,    /* scan_line not in {16,32,64} with ring_model 0: ROS_BREAK / return 0 in the reference
                                 (scanRegistration.cpp:170-174, :447-451) */
    LL_ERR_CAPACITY = -4,     /* more points than max_points (reference: fixed 400000 arrays, :34-40) or a ring
                                 longer than max_ring_points */
    LL_ERR_EMPTY = -5,        /* no point survives the NaN / minimum-range filters (reference dereferences points[0]) */
    LL_ERR_HIP = -6,          /* a HIP runtime call failed; ll_last_error() has the text */
    LL_ERR_STATE = -7         /* stage called before its inputs exist (e.g. associate before set_target) */
} ll_status;

/* Node parameters (scanRegistration.cpp:435-443, laserOdometry.cpp:23-30) + capacities. */
typedef struct {
    int   n_scans;            /* "scan_line": 16 / 32 / 64 */
    int   ring_model;         /* 0: reference switch on n_scans; 1: the linear 64-ring formula (:162) for any n_scans */
    float minimum_range;      /* "minimum_range" (launch: 5 for HDL-64, 0.3 for VLP-16/HDL-32) */
    float lower_bound;        /* "lowerBound" default -24.9 */
    float up_bound;           /* "upBound" default 2 */
    int   max_points;         /* per-scan point capacity (<= 400000 like the reference arrays) */
    int   max_ring_points;    /* per-ring capacity, 32 .. 8192 (default 2304): the feature kernel's LDS staging, and the stride at which
                                 the rings of laserCloud sit in HBM (n_scans x max_ring_points points per slot) */
    int   batch;              /* scan slots resident in HBM */
    /* thresholds; ll_default_params() fills the reference constants */
    float curv_threshold;     /* 0.1  (:266, :321) compared as double like the reference */
    float gap_sq_threshold;   /* 0.05 (:293 ...) */
    float leaf_size;          /* 0.2  (:373) */
    float nn_dist_sq_max;     /* DISTANCE_SQ_THRESHOLD 25 (laserOdometry.cpp:29) */
    float nearby_scan;        /* NEARBY_SCAN 2.5 (laserOdometry.cpp:30) */
    float huber_delta;        /* HuberLoss(0.1) (laserOdometry.cpp:475); <= 0 disables the loss */
    int   write_curvature;    /* also store cloudCurvature[] to HBM (debug / parity output; off on the hot path) */
    int   chunk;              /* ll_hot_path_batch processes the slot range in chunks of this many scans so that a chunk's
                                 intermediates stay in the 256 MiB Infinity Cache between kernels; 0 = whole range */
    int   distortion;         /* DISTORTION of laserOdometry.cpp:23.  0 (default) = the reference's build: s = 1.  1 = its other compile-time
                                 path: every point's interpolation ratio s = (intensity - int(intensity)) / SCAN_PERIOD in TransformToStart
                                 (:81-88) and in LidarEdgeFactor / LidarPlaneFactor_modify (:570-571, :740-741, lidarFactor.hpp:25-27) */
    int   voxel_sort_ranks;   /* the VoxelGrid sort's ranking (ll_features.hip).  0 (default) = auto: ranks from one returning LDS add per row when
                                 the device passes ll_create's lane-order check, else from a match-any; 1 = always the match-any (tests; same clouds
                                 bit for bit, ~0.7 % of that kernel slower) */
    int   input_stride_floats;/* floats per point of the RESIDENT raw scan: 4 (default; KITTI .bin, PointXYZ: x, y, z, one unused float) or 3
                                 (x, y, z packed: scanRegistration.cpp:105-106 keeps nothing else, so a streamed scan crosses PCIe and is read by
                                 the organise stage at 12 instead of 16 bytes per point).  ll_upload_scan repacks any caller stride into it; the
                                 asynchronous uploads take their buffers in exactly this layout */
} ll_params;

/* Per-scan sizes produced by the extract stage. */
typedef struct {
    int status;               /* ll_status of this slot's last extract */
    int n_in;                 /* points uploaded */
    int n;                    /* laserCloud size after filtering + ring rejection (cloudSize, :212) */
    int n_sharp, n_less_sharp, n_flat, n_less_flat;
    int max_ring;             /* longest ring */
} ll_scan_info;

/* Per-pair sizes produced by associate / vote. */
typedef struct {
    int n_edge;               /* corner_correspondence (laserOdometry.cpp:617) */
    int n_plane;              /* plane_correspondence (:790) */
    int n_plane_selected;     /* selected_idx.size() after the vote (:796); == n_plane when vote disabled */
} ll_pair_info;

/* ---------------------------------------------------------------- lifecycle */
void        ll_default_params(ll_params *p, int n_scans);
/* ll_create: LL_ERR_ARG for parameters outside their ranges, LL_ERR_DEVICE when `device` is no usable gfx950 device -- the library has no CPU path
 * --, LL_ERR_HIP when an allocation fails or the one-off lane-order check of the voxel filter's sort could not run (a device that fails the check
 * itself is served by the sort's other ranking, ll_params.voxel_sort_ranks); ll_last_error(NULL) says which. */
int         ll_create(int device, const ll_params *p, ll_ctx **out);
void        ll_destroy(ll_ctx *ctx);
const char *ll_last_error(const ll_ctx *ctx);      /* ctx may be NULL: last create error */
int         ll_abi_version(void);
void       *ll_stream(ll_ctx *ctx);                 /* hipStream_t the ctx launches on (for HIP-event timing) */
int         ll_synchronize(ll_ctx *ctx);

/* ---------------------------------------------------------------- input
 * Replaces pcl::fromROSMsg of the /rslidar_points message (scanRegistration.cpp:105-106, :453).
 * xyz: n points of `stride_floats` floats each (>= 3; 4 = KITTI .bin / PointXYZ padding).              */
int ll_upload_scan(ll_ctx *ctx, int slot, const float *host_xyz, int stride_floats, int n);
/* Streaming input (BASELINE config 5): the same upload, asynchronous on the context's COPY stream.  xyz4: n points in the context's RESIDENT
 * layout -- (x, y, z, .) at a 16-byte stride by default, (x, y, z) packed at 12 bytes with ll_params.input_stride_floats = 3 (nothing is
 * repacked on the way: a quarter fewer bytes cross PCIe) --, ideally page-locked (ll_host_alloc); it must not be modified until the copy has run.  The copy
 * stream (1) and the compute stream (0: every stage call) are ordered by caller-named events 0 .. 7, never by blocking the
 * host: ll_stream_record marks "everything enqueued on that stream so far", ll_stream_wait makes what is enqueued on a stream
 * from now on wait for a mark (a never-recorded event: no wait).  Slots in two halves + two events per half = a double
 * buffer: the upload of one half overlaps the processing of the other (bench.py --stream-input).                         */
#define LL_STREAM_COMPUTE 0
#define LL_STREAM_COPY 1
int   ll_upload_scan_async(ll_ctx *ctx, int slot, const float *xyz4, int n);
int   ll_upload_scans_async(ll_ctx *ctx, int first, int count, const float *const *xyz4, const int *n);   /* slots first .. first+count-1 */
/* the same for a run of slots out of ONE page-locked staging area, scan i at base + i * stride_bytes (stride a multiple of 16; of 4 in the
 * 12-byte layout):
 * two enqueues for the whole run (a per-scan feed is bounded by the host cost of the copy calls, not by PCIe) */
int   ll_upload_scans_async_strided(ll_ctx *ctx, int first, int count, const float *base, size_t stride_bytes, const int *n);
int   ll_stream_record(ll_ctx *ctx, int stream, int event_id);
int   ll_stream_wait(ll_ctx *ctx, int stream, int event_id);
int   ll_synchronize_copy(ll_ctx *ctx);
void *ll_host_alloc(size_t bytes);            /* page-locked host memory (NULL on failure) */
void  ll_host_free(void *p);


/* ---------------------------------------------------------------- a1-a4: laserCloudHandler
 * scanRegistration.cpp:87-428 for slots [first, first+count): removeNaN + removeClosedPointCloud (:58-85,
 * :109-110), ring / relTime assignment and stable ring bucketing (:113-221), curvature (:225-235),
 * per-segment sort + greedy pick (:246-368), per-ring VoxelGrid of the less-flat points (:370-376).
 * Per-slot failures (empty scan, capacity) are reported in ll_scan_info.status; the call itself
 * fails only on argument / runtime errors.                                                            */
int ll_extract_batch(ll_ctx *ctx, int first, int count);
int ll_get_scan_info(ll_ctx *ctx, int slot, ll_scan_info *info);

/* Downloads (what the node publishes, :382-410, plus the file-scope arrays for parity checks).
 * Any pointer may be NULL.  Capacities are in elements; LL_ERR_CAPACITY if too small.                  */
int ll_download_cloud(ll_ctx *ctx, int slot, ll_point *cloud, int cap, int *scan_start, int *scan_end /* n_scans each */);
int ll_download_labels(ll_ctx *ctx, int slot, int8_t *label, float *curvature /* needs write_curvature */, int cap);
int ll_download_features(ll_ctx *ctx, int slot,
                         ll_point *sharp, int cap_sharp, ll_point *less_sharp, int cap_less_sharp,
                         ll_point *flat, int cap_flat, ll_point *less_flat, int cap_less_flat);

/* ---------------------------------------------------------------- a5-a7: data association
 * Replaces the kd-tree rebuild (laserOdometry.cpp:882-896) + the two correspondence loops (:491-620, :653-793)
 * for pairs (slot k, its target).  Targets of slot k are the less-sharp / less-flat clouds of slot k-1;
 * slot `first`'s target is the ctx "carry" target (the previous batch's last scan, or whatever
 * ll_set_target uploaded).  pose_guess: count x 7 doubles (qx,qy,qz,qw,tx,ty,tz) = para_q/para_t at entry
 * (:61-62); NULL = identity.
 * The association FIXES a slot's target: ll_vote_batch, ll_normal_equations_batch, ll_residual_jacobian and the solves read
 * whose points the slot's correspondences name from what ll_associate_batch recorded, so a later call over any sub-range
 * (e.g. ll_vote_batch(k, 1)) refers to the same clouds.                                                                */
int ll_set_target(ll_ctx *ctx, const ll_point *host_corner_last, int m_c, const ll_point *host_surf_last, int m_s);
/* The four feature clouds of a scan from host memory into a slot, in place of ll_extract_batch: what a separate
 * laserOdometry process receives on /laser_cloud_sharp, _less_sharp, _flat, _less_flat (laserOdometry.cpp:116-150,
 * :404-423) when the registration node runs elsewhere.  The slot then serves ll_associate_batch .. ll_odometry_frames and
 * ll_set_target_from_slot like an extracted one (its laserCloud / labels are empty).  intensity must carry the ring id
 * as the registration node stores it (int part = scanID).  LL_ERR_CAPACITY beyond the per-scan feature capacities.     */
int ll_upload_features(ll_ctx *ctx, int slot, const ll_point *host_sharp, int n_sharp, const ll_point *host_less_sharp, int n_less_sharp,
                       const ll_point *host_flat, int n_flat, const ll_point *host_less_flat, int n_less_flat);
int ll_set_target_from_slot(ll_ctx *ctx, int slot);   /* device-to-device: slot's less-sharp/less-flat become the carry */
int ll_associate_batch(ll_ctx *ctx, int first, int count, const double *host_pose_guess);
/* Store the guess in HBM; ll_hot_path_batch(..., NULL, ...) then restarts every slot from it without touching the host. */
int ll_set_pose_guess(ll_ctx *ctx, int first, int count, const double *host_pose_guess);
int ll_get_pair_info(ll_ctx *ctx, int slot, ll_pair_info *info);
/* edge: (src index into sharp, a, b into corner_last); plane: (src into flat, a, b, c into surf_last).  */
int ll_download_edge_corr(ll_ctx *ctx, int slot, int *src, int *a, int *b, int cap);
int ll_download_plane_corr(ll_ctx *ctx, int slot, int *src, int *a, int *b, int *c, int cap);

/* ---------------------------------------------------------------- a8: graph vote
 * graph_based_correspondence_vote_simple (laserOdometry.cpp:165-342, call :796) on the plane
 * correspondences of each pair.  enable = 0 reproduces the now_frame <= 5 branch (:781-787): all kept, weight 1. */
int ll_vote_batch(ll_ctx *ctx, int first, int count, int enable);
/* The free function itself, on caller-supplied host correspondences (Corre_Match.src / .tgt, laserOdometry.cpp:165-172):
 * corner_case selects 5 regions instead of 10 (:179-188).  Outputs are per correspondence, in input order.          */
int ll_vote_host(ll_ctx *ctx, const ll_point *host_src, const ll_point *host_tgt, int n, int corner_case,
                 int *count, uint8_t *selected, float *weight);
/* per plane correspondence (in correspondence order): incompatibility count, selected flag, weight      */
int ll_download_vote(ll_ctx *ctx, int slot, int *count, uint8_t *selected, float *weight, int cap);

/* ---------------------------------------------------------------- a9-a10: residuals, Jacobians, GN
 * Residual blocks in Ceres order: n_edge LidarEdgeFactor blocks (3 rows, lidarFactor.hpp:9-52) then the
 * selected LidarPlaneFactor_modify blocks (1 row, :203-251, weight = vote weight), all with s = 1
 * (DISTORTION 0, laserOdometry.cpp:23).  pose: count x 7 doubles, NULL = the pose the ctx currently holds
 * for the slot (guess, or the result of the last ll_gn_step_batch).                                      */
int ll_normal_equations_batch(ll_ctx *ctx, int first, int count, const double *host_pose);
/* H: 36 doubles row-major over (dtheta[3], dt[3]) in the EigenQuaternionManifold tangent; g = J^T r; cost = sum rho/2 */
int ll_download_normal_equations(ll_ctx *ctx, int slot, double *H36, double *g6, double *cost);
/* Solve H d = -g (Cholesky, f64) on device, q <- Plus(q, d[0:3]), t += d[3:6]; one Gauss-Newton iteration.  */
int ll_gn_step_batch(ll_ctx *ctx, int first, int count);
int ll_download_pose(ll_ctx *ctx, int slot, double *pose7);
/* What ceres::CostFunction::Evaluate would return for the slot's blocks at `pose7`: residuals (rows),
 * jacobians[0] rows x 4 (ambient x,y,z,w) and jacobians[1] rows x 3, row-major; loss NOT applied
 * (Ceres applies it outside Evaluate).  rows = 3*n_edge + n_plane_selected.                             */
int ll_residual_jacobian(ll_ctx *ctx, int slot, const double *pose7, double *r, double *Jq, double *Jt, int cap_rows);

/* ---------------------------------------------------------------- the reference's solver and frame loop
 * ceres::Solve as laserOdometry.cpp:820-825 configures it: trust-region / Levenberg-Marquardt, DENSE_QR,
 * max_num_iterations = 4, every other option at its Ceres default.  ll_lm_default_options fills exactly those.       */
typedef struct {
    int    max_num_iterations;
    double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    int    jacobi_scaling;
} ll_lm_options;
void ll_lm_default_options(ll_lm_options *o);
/* LM solve of the residual blocks the slot currently holds (after associate + vote), starting from the slot's pose;
 * the pose is replaced by the solution.  Device-resident, no host synchronisation; opt NULL = defaults.              */
int ll_lm_solve_batch(ll_ctx *ctx, int first, int count, const ll_lm_options *opt);
/* laserOdometry's per-frame body (:439-832) for the scans in slots [first, first+count), IN SEQUENCE: slot k's target
 * is slot k-1 (the carry for slot `first`), its initial para_q/para_t is the previous slot's result (host_pose0 for the
 * first; NULL = identity) -- the warm start of :61-65.  Per frame: n_outer (3, :439) x { associate, vote when the
 * frame's sequence index first_frame_index + k is > 5 (:794), LM solve }.  host_poses_out: count x 7, may be NULL.
 * The slots must have been extracted (ll_extract_batch).                                                             */
int ll_odometry_frames(ll_ctx *ctx, int first, int count, const double *host_pose0, int n_outer, int first_frame_index,
                       const ll_lm_options *opt, double *host_poses_out);

/* ---------------------------------------------------------------- laserMapping scan-to-submap (first stage)
 * The optimisation laserMapping runs per frame (laserMapping.cpp:1822-2095) for ONE scan against the corner / surf clouds
 * gathered from the cube map.  The cube bookkeeping (:1584-1808, :2101-2165) stays with the caller for now.
 * An ll_map shares the device and stream of the ll_ctx it was created from and must be destroyed before it.            */
typedef struct ll_map ll_map;
int  ll_map_create(ll_ctx *ctx, int max_map_corner, int max_map_surf, int max_scan_corner, int max_scan_surf, ll_map **out);
void ll_map_destroy(ll_map *m);
const char *ll_map_last_error(const ll_map *m);
/* kdtreeCornerFromMap->setInputCloud(laserCloudCornerFromMap) / kdtreeSurfFromMap (:1826-1827): upload + search grid */
int ll_map_set_map(ll_map *m, const ll_point *host_corner_from_map, int n_corner, const ll_point *host_surf_from_map, int n_surf);
/* laserCloudCornerStack / laserCloudSurfStack: the scan's (already down-sized, :1813-1821) feature clouds, sensor frame */
int ll_map_set_scan(ll_map *m, const ll_point *host_corner_stack, int n_corner, const ll_point *host_surf_stack, int n_surf);
/* One data-association pass (:1877-2047) at pose_w = parameters[7] (q_w_curr x,y,z,w, t_w_curr); NULL = the map's
 * current pose.  Produces the residual blocks in stack order: edges (stack index, point_a, point_b), planes (stack
 * index, unit norm, negative_OA_dot_norm).                                                                           */
int ll_map_associate(ll_map *m, const double *pose_w7);
int ll_map_get_counts(ll_map *m, int *n_edge, int *n_plane);
/* laserCloudCornerFromMap->points.size() / laserCloudSurfFromMap->points.size() (:1822: the block runs only above 10 / 50) */
int ll_map_get_map_sizes(ll_map *m, int *n_corner_from_map, int *n_surf_from_map);
int ll_map_download_edges(ll_map *m, int *src, double *a3, double *b3, int cap);
int ll_map_download_planes(ll_map *m, int *src, double *norm3, double *d, int cap);
/* H (6x6 row-major over the manifold tangent + t), g, cost of the current blocks at pose_w (NULL = current pose),
 * HuberLoss(0.1) + EigenQuaternionManifold as :1863-1866                                                            */
int ll_map_normal_equations(ll_map *m, const double *pose_w7, double *H36, double *g6, double *cost);
/* What ceres::CostFunction::Evaluate would return for the current blocks at pose_w (NULL = current pose): residuals,
 * jacobians[0] rows x 4 (ambient x,y,z,w), jacobians[1] rows x 3, row-major, loss NOT applied; rows = 3*n_edge + n_plane,
 * edges first -- for callers that keep ceres::Solve (:2073-2082) and only replace the data association.              */
int ll_map_residual_jacobian(ll_map *m, const double *pose_w7, double *r, double *Jq, double *Jt, int cap_rows);
/* The whole block :1822-2095: if the map holds > 10 corner and > 50 surf points, n_outer (2, :1832) x { associate,
 * ceres::Solve restated (LM, <= 4 iterations; opt NULL = ll_lm_default_options) }.  pose_w7 in/out; *ran = 0 when the
 * map is too small (the reference then keeps the odometry guess, :2096-2100).  LL_ERR_STATE: the solve left an
 * undefined (NaN) pose -- a lost hand-over inside the launch -- or the map holds a row / tile shard.                  */
int ll_map_optimize(ll_map *m, double *pose_w7, int n_outer, const ll_lm_options *opt, int *ran);

/* Row-parallel use over several GPUs (BASELINE config 4): every rank holds the same map and its share of the
 * stack points (ll_map_set_scan with a slice).  The residual blocks are independent, so the normal equations of the
 * whole scan are the SUM over ranks of ll_map_evaluate's 44-double record (H 36, g 6, cost, rows); the caller
 * all-reduces it (RCCL over xGMI: 28 unique doubles matter) and feeds the sum to ll_map_lm_begin / ll_map_lm_accept,
 * so that every rank advances an identical Levenberg-Marquardt state:
 *     associate;  evaluate -> all-reduce -> lm_begin;  repeat max_num_iterations x { lm_propose; evaluate -> all-reduce -> lm_accept }
 * ll_map_optimize is exactly this sequence on one rank without the all-reduce.                                          */
int ll_map_set_pose(ll_map *m, const double *pose_w7);
int ll_map_get_pose(ll_map *m, double *pose_w7);                                        /* LL_ERR_STATE: the pose on the device is undefined (NaN) */
int ll_map_evaluate(ll_map *m, double *neq44);                                           /* at the map's current pose */
int ll_map_lm_begin(ll_map *m, const double *neq44_sum, const ll_lm_options *opt);
int ll_map_lm_propose(ll_map *m, const ll_lm_options *opt);                               /* current pose <- candidate (or unchanged) */
int ll_map_lm_accept(ll_map *m, const double *neq44_sum, const ll_lm_options *opt);       /* current pose <- accepted state */

/* Tile-parallel use over several GPUs: the map -- not the scan -- is split; every rank holds the points
 * of ITS cubes (ll_cubemap_set_shard, or ll_map_set_map + ll_map_set_map_ids for a hand-made split) and the whole scan.
 *     repeat n_outer x { ll_map_knn_partial -> all-gather of the candidates -> ll_map_associate_merged -> ll_map_solve }
 * ll_map_knn_partial: the five nearest of this rank's points per stack point, nn = 5 x (x, y, z, squared distance) floats,
 * id = 5 global ids (INFINITY / INT_MAX in unused slots); buffers of n_stack * 5 entries, HOST OR DEVICE memory (the
 * all-gather is RCCL on device buffers over xGMI, 100 B per stack point and rank).  ll_map_associate_merged: the buffers
 * of all parts back to back ([part][stack point][5], host or device); keeps the five smallest (distance, id) -- the
 * global ids number the points in the order of the unsplit search cloud, so ties fall exactly as in ll_map_associate --
 * and runs the line / plane fit: every rank ends with the residual blocks ll_map_associate would give on the whole map.
 * ll_map_solve: one ceres::Solve restated on those blocks (replicated: <= a few thousand rows; identical on all ranks).
 * ll_map_set_map_ids: ids ascending in the order of the unsplit cloud; both NULL = back to positions.                  */
int ll_map_set_map_ids(ll_map *m, const int *corner_gid, const int *surf_gid);
int ll_map_knn_partial(ll_map *m, const double *pose_w7, float *corner_nn, int *corner_id, float *surf_nn, int *surf_id);
int ll_map_associate_merged(ll_map *m, const double *pose_w7, int n_parts, const float *corner_nn, const int *corner_id,
                            const float *surf_nn, const int *surf_id);
int ll_map_solve(ll_map *m, double *pose_w7, const ll_lm_options *opt);                 /* LL_ERR_STATE: row shard set, or an undefined (NaN) pose came out */
/* Device-resident variants of the same steps for the collectives: every pointer is a DEVICE pointer on the context's GPU, the
 * calls only enqueue on ll_stream(ctx) (no host hop, no synchronisation) and work at the pose already on the device
 * (ll_map_set_pose before, ll_map_get_pose after).  The caller's RCCL all-reduce (44 doubles, of which 28 matter) /
 * all-gather (100 B per stack point and rank) runs on the same buffers, stream-ordered with ll_stream(ctx); this is what
 * lightloam_amd/parallel.py does on the GPU box (laserMapping.cpp:1832-2095 spread over the GPUs of a node).              */
int ll_map_evaluate_dev(ll_map *m, double *neq44_dev);
int ll_map_lm_begin_dev(ll_map *m, const double *neq44_sum_dev, const ll_lm_options *opt);
int ll_map_lm_propose_dev(ll_map *m, const ll_lm_options *opt);
int ll_map_lm_accept_dev(ll_map *m, const double *neq44_sum_dev, const ll_lm_options *opt);
int ll_map_knn_partial_dev(ll_map *m, float *corner_nn_dev, int *corner_id_dev, float *surf_nn_dev, int *surf_id_dev);
int ll_map_associate_merged_dev(ll_map *m, int n_parts, const float *corner_nn_dev, const int *corner_id_dev,
                                const float *surf_nn_dev, const int *surf_id_dev);   /* buffers must stay valid until the stream has run it */
int ll_map_solve_dev(ll_map *m, const ll_lm_options *opt);
/* BASELINE config 4 in full -- tiles for the search AND rows for the solve: after ll_map_associate_merged every rank holds
 * all residual blocks; with a row shard set, ll_map_evaluate / ll_map_normal_equations sum only the blocks i with
 * i % world == rank, and the LM runs through evaluate -> all-reduce(JtJ, Jtr, cost) -> ll_map_lm_begin / _accept as in the
 * row-parallel scheme above (result within f64 summation-order rounding of one GPU; identical on all ranks).  (0, 1) =
 * all blocks.  ll_map_solve / ll_map_optimize refuse while a row shard is set.                                        */
int ll_map_set_row_shard(ll_map *m, int rank, int world);

/* pcl::VoxelGrid<PointType>::filter on a whole cloud of any size (downSizeFilterCorner / downSizeFilterSurf,
 * laserMapping.cpp:1813-1821, :2151-2165): centroids (x, y, z, intensity) per voxel, in voxel-index order.       */
int ll_voxel_grid(ll_ctx *ctx, const ll_point *host_in, int n, float leaf_size, ll_point *host_out, int cap, int *n_out);

/* ---------------------------------------------------------------- lidarFactor.hpp functors on caller-supplied blocks
 * For a node that keeps its own association code and ceres::Problem and only wants the functors evaluated on the device
 * (include/lightloam_lidarFactor.hpp keeps LidarEdgeFactor::Create / LidarPlaneFactor_modify::Create /
 * LidarPlaneNormFactor::Create as they are called at laserOdometry.cpp:615, :783 and laserMapping.cpp:1918, :2033).
 * ll_factor_blocks_set: the blocks of one problem, f64 --
 *     edge9   [n_edge][9]   curr_point, last_point_a, last_point_b                         (lidarFactor.hpp:9-52,  3 rows each)
 *     plane13 [n_plane][13] curr_point, last_point_j, last_point_l, last_point_m, weight   (:203-251, 1 row each)
 *     pnorm7  [n_pnorm][7]  curr_point, plane_unit_norm, negative_OA_dot_norm              (:253-285, 1 row each)
 * all with s = 1 (what the reference's build passes: DISTORTION 0, laserOdometry.cpp:23, :81-84) unless ll_factor_blocks_set_s
 * follows: the functors' s_ of every edge / plane block (lidarFactor.hpp:25-27, :219-221: Identity.slerp(s, q), s * t), f64
 * [n_edge] and [n_plane]; NULL = all ones.  ll_factor_blocks_set resets them to one.
 * ll_factor_blocks_evaluate: residuals [rows], jacobians w.r.t. q (x, y, z, w) [rows][4] and t [rows][3], row-major,
 * rows = 3 n_edge + n_plane + n_pnorm in that order; loss functions are the caller's (Ceres applies them).         */
int ll_factor_blocks_set(ll_ctx *ctx, int n_edge, const double *edge9, int n_plane, const double *plane13, int n_pnorm, const double *pnorm7);
int ll_factor_blocks_set_s(ll_ctx *ctx, const double *edge_s, const double *plane_s);
int ll_factor_blocks_evaluate(ll_ctx *ctx, const double q[4], const double t[3], double *r, double *Jq, double *Jt, int cap_rows);

/* ---------------------------------------------------------------- laserMapping's cube map (second stage)
 * The 21 x 21 x 11 cubes of 50 m that hold the map (laserMapping.cpp:45-53, :74-75), resident in HBM, and the per-frame
 * body around the optimisation: ll_cubemap_prepare = :1584-1821 (centre cube of t_w_curr, the six shift loops, the
 * 5 x 5 x 3 valid cubes gathered into laserCloudCornerFromMap / SurfFromMap, the scan's less-sharp / less-flat clouds
 * down-sized to laserCloudCornerStack / SurfStack), ll_cubemap_optimize = :1822-2100, ll_cubemap_update = :2103-2165
 * (the registered scan's points into their cubes, every valid cube down-sized).  line_res / plane_res =
 * mapping_line_resolution / mapping_plane_resolution (:2363-2364, defaults 0.4 / 0.8).  pool_points: HBM points per
 * cloud type for the cubes (two pools of that size each).                                                          */
typedef struct ll_cubemap ll_cubemap;
int  ll_cubemap_create(ll_ctx *ctx, float line_res, float plane_res, int max_scan_corner, int max_scan_surf, int pool_points, ll_cubemap **out);
void ll_cubemap_destroy(ll_cubemap *cm);
const char *ll_cubemap_last_error(const ll_cubemap *cm);
int ll_cubemap_prepare(ll_cubemap *cm, const double *t_w3, const ll_point *host_corner_last, int n_corner, const ll_point *host_surf_last, int n_surf);
int ll_cubemap_optimize(ll_cubemap *cm, double *pose_w7, int n_outer, const ll_lm_options *opt, int *ran);
int ll_cubemap_update(ll_cubemap *cm, const double *pose_w7);
/* the three calls in sequence with the reference's constants; pose_w7 in: the guess of transformAssociateToMap (:1581) */
int ll_cubemap_process(ll_cubemap *cm, double *pose_w7, const ll_point *host_corner_last, int n_corner, const ll_point *host_surf_last, int n_surf, int *ran);
/* the same for the scan that sits in an extracted slot of the owning ll_ctx: its less-sharp / less-flat clouds go from the
 * slot to the map stage device-to-device (the laser_cloud_corner_last / laser_cloud_surf_last topics, laserOdometry.cpp:898-910) */
int ll_cubemap_process_slot(ll_cubemap *cm, double *pose_w7, int slot, int *ran);
/* Tile shard: this cube map keeps only the cubes owned by `rank` of `world` (ownership is a function of the cube's
 * position in the world, so the shift loops never move a cube between ranks); call before the first scan.  prepare and
 * update work as before on the owned cubes (prepare also numbers the gathered points for ll_map_knn_partial);
 * ll_cubemap_optimize / _process refuse (LL_ERR_STATE): the search goes through ll_cubemap_map() and the calls above.
 * The union of the ranks' cubes is the unsplit cube map, cube by cube and bit for bit.                               */
int ll_cubemap_set_shard(ll_cubemap *cm, int rank, int world);
ll_map *ll_cubemap_map(ll_cubemap *cm);                       /* the inner ll_map (owned by the cube map) */
/* cen3: laserCloudCenWidth / Height / Depth; counts4: corner / surf from map, corner / surf stack */
int ll_cubemap_info(ll_cubemap *cm, int *cen3, int *counts4);
/* which: 0 laserCloudCornerFromMap, 1 laserCloudSurfFromMap, 2 laserCloudCornerStack, 3 laserCloudSurfStack */
int ll_cubemap_download_cloud(ll_cubemap *cm, int which, ll_point *out, int cap, int *n);
int ll_cubemap_download_cube(ll_cubemap *cm, int surf, int cube_index, ll_point *out, int cap, int *n);

/* ---------------------------------------------------------------- whole hot path
 * One pass: extract + associate + vote + normal equations + one GN step for slots [first, first+count),
 * everything device-resident, no host synchronisation inside.  `vote_enable` as above.                   */
int ll_hot_path_batch(ll_ctx *ctx, int first, int count, const double *host_pose_guess, int vote_enable);
/* The same pass continuing a batch that an earlier call opened: the target of slot `first` is slot first - 1, not the carry
 * (a batch processed in pieces -- e.g. the halves of a double-buffered stream -- gives the results of one call over the whole). */
int ll_hot_path_chain(ll_ctx *ctx, int first, int count, int vote_enable);
/* Schedule of the association stage inside ll_hot_path_batch / _chain for calls (chunks) of at least 512 scans.  on = 0 (default): one
 * stream, kernel after kernel.  on = 1: the target grids of one quarter of the range are built while the quarter before it is searched,
 * on two HIP streams ordered by events.  Results are identical either way; on the MI355X boxes of round 6 the second schedule was never
 * faster (profiles/r06_experiments/two_stream_pieces.log), which is why it is not the default.  LIGHTLOAM_TWO_STREAM=1 in the environment
 * makes 1 the default of new contexts. */
int ll_set_two_stream(ll_ctx *ctx, int on);

/* ---------------------------------------------------------------- measurement
 * With profiling on, every kernel launched by the stage calls is bracketed by HIP events on the ctx stream.
 * ll_profile_read synchronises the stream and returns, per kernel, the summed duration and the launch count
 * since the last reset.  names[i] points to a static string.  n is in: capacity / out: kernels returned.    */
int ll_profile_enable(ll_ctx *ctx, int on);
int ll_profile_read(ll_ctx *ctx, int *n, const char **names, double *total_ms, int *launches, int reset);
/* 16 in-kernel phase counters (shader cycles); zero unless the library was built with -DLL_PHASE_TIMING (tools/phase_timing.py) */
int ll_debug_counters(ll_ctx *ctx, unsigned long long *out16, int reset);
/* one float4 streaming copy of `bytes` in + `bytes` out ("k_calib_copy"): the known-byte-count launch that calibrates
 * rocprofv3's FETCH_SIZE / WRITE_SIZE counters (tools/pmc_traffic.py) */
int ll_debug_calibration_copy(ll_ctx *ctx, unsigned long long bytes);
/* One stage's kernel(s) on their own over slots that hold what the stage reads: 0 organise, 1 pick, 2 voxel filter + lists, 3 grid tables,
 * 4 association, 5 vote, 6 normal equations (timing probes only; LL_ERR_ARG for another stage). */
int ll_debug_launch_stage(ll_ctx *ctx, int stage, int first, int count);
/* The DEVICE's evaluation of the arithmetic that must equal the host libm bit for bit (scanRegistration.cpp:139, :177 call
 * atan / atan2 / sqrt of glibc), over host arrays a, b, c (n floats each; unused ones may be NULL), results in out (n x 4 B):
 *   op 0  atanf(a)                    op 1  atan2f(a, b), general path     op 2  atan2f(a, b), k_classify's fast path
 *   op 3  (float)((double)a / M_PI)   op 4  a / sqrtf(b*b + c*c)  (:139's argument, z = a, x = b, y = c)
 *   op 5  ring id (int) of the point (x = b, y = c, z = a) by k_classify's threshold search with this context's parameters
 *   op 6  ring id (int) by evaluating the reference's formula chain directly on the device (:139-168)
 * Test surface (tests/test_gpu_a1_edges.py compares with glibc on the GPU box); not used by the pipeline. */
int ll_debug_exact_math(ll_ctx *ctx, int op, const float *a, const float *b, const float *c, int n, void *out);

/* Algorithmic HBM bytes of the last ll_hot_path_batch / stage calls, summed over the slots they covered,
 * by SURVEY.md section 8d's formula (B_ext, B_assoc, B_vote, B_rj).                                      */
int ll_algorithmic_bytes(ll_ctx *ctx, int first, int count, double *b_ext, double *b_assoc, double *b_vote, double *b_rj);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTLOAM_HIP_H */
