/*
 * ll_common.h -- device-side view of a context's HBM-resident state + helpers shared by the kernels.
 *
 * HBM layout (B = batch slots, NP = max_points rounded up to a whole tile, R = n_scans, T = NP / LL_TILE):
 *   raw        float4 [B][NP]      uploaded points (x, y, z, reflectance/pad) -- KITTI .bin / PointXYZ stride
 *   hdr        ScanHdr[B]          startOri / endOri / halfPassed index / sizes / status
 *   ring_off   int    [B][R+1]     ring r is laserCloud[ring_off[r], ring_off[r+1]) (labels, curvature, C ABI)
 *   cloud      float4 [B][R][ring_cap]  laserCloud (x, y, z, intensity), every ring at a fixed stride (ring_cap = max_ring_points)
 *   label      int8   [B][NP]      cloudLabel   curv float [B][NP] (optional)
 *   ring_rec, ring_cnt             k_ring_pick's lists per ring (local indices) and its three counts
 *   sharp / less_sharp / flat      float4 [B][R*12 / R*120 / R*24], contiguous, in the reference's publication order (ring, segment, pick)
 *   carry_*    target clouds for the first slot of a batch (previous batch's last scan).
 *   corr / vote / neq / pose arrays for the odometry stages.
 *
 * Feature cloud layout -- the less-flat cloud (the only one whose per-ring size is decided by the kernel that writes it):
 *   lflat      float4 [B][LFS >= R * ring_cap]   ring r's VoxelGrid centroids in its own row, ring_nlf[B][R] of them: no ring waits for another.
 *   A point's PLACE is its float4 offset inside the slot (ring * ring_cap + k); its INDEX in the reference's contiguous cloud is
 *   lf_pre[ring] + k, lf_pre[B][R+1] = the exclusive prefix of ring_nlf (k_build_grid writes it with hdr.n_less_flat).  Both orders
 *   agree (ring-major), so everything that only compares -- ties towards the lower index, the ring-window bounds, the visiting order of
 *   the second / third point -- runs on places; the C ABI converts places to indices where correspondences leave the library
 *   (place -> index is a division, index -> place a search: the hot path never needs the latter).  A slot filled by
 *   ll_upload_features, and the carry, hold the caller's contiguous cloud: hdr.lf_strided = 0, place = index.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ll_exact_math.h"

#define LL_TILE 1024          /* points per organise tile: 256 threads x 4 */
#define LL_ORG_SMALL 64       /* calls of at most this many scans take the tile-parallel organise kernels */
#define LL_BLOCK 256
#define LL_MAX_RINGS 128
#define LL_SHARP_PER_SEG 2
#define LL_LSHARP_PER_SEG 20
#define LL_FLAT_PER_SEG 4
#define LL_SEGS 6
/* what k_ring_pick (ll_pick.hip) hands to k_ring_features (ll_features.hip), per ring: the picked points' local indices --
 * sharp [0, 12), less-sharp [12, 132), flat [132, 156), entry = segment * per-segment capacity + pick order -- and the per-segment
 * counts (n_sharp, n_less_sharp, n_flat) x 6 at [156, 174) */
#define LL_REC_U16 176
#ifndef LL_PK_WAVES
#define LL_PK_WAVES 1         /* rings (independent waves) per k_ring_pick workgroup */
#endif
/* nearest-neighbour cell grid over (x, y): 128 x 128 cells of 1 m centred on the sensor; farther points saturate into
 * the border cells, whose rectangles count as unbounded outwards */
#define LL_GRID_G 128
#define LL_GRID_CELL 1.0f
#define LL_GRID_ORG 64.0f
#define LL_GRID_NC (LL_GRID_G * LL_GRID_G)
/* ring tables stored behind the cell starts: first_ge[LL_TAB+1], last_le[LL_TAB+1] (places), ok flag (the tables bound the walks),
 * cloud size, one past the last place */
#define LL_TAB 160
#define LL_TAB_WORDS (2 * (LL_TAB + 1) + 3)
#define LL_GSTRIDE (LL_GRID_NC + 1 + LL_TAB_WORDS)

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#ifdef __HIPCC__
__device__ __forceinline__ int ll_cell_coord(float v)
{
    const int c = (int)floorf((v + LL_GRID_ORG) * (1.0f / LL_GRID_CELL));
    return min(max(c, 0), LL_GRID_G - 1);
}
/* what k_build_grid's histogram sweep needs of a target point: its cell (14 bits) and the walk's scan id int(intensity)
 * (laserOdometry.cpp:500, :664) as a byte, 0xFF when it lies outside the ring tables' range [0, LL_TAB) */
__device__ __forceinline__ unsigned ll_grid_key(const float4 p)
{
    const int c = ll_cell_coord(p.y) * LL_GRID_G + ll_cell_coord(p.x), r = (int)p.w;
    return (unsigned)c | ((r < 0 || r >= LL_TAB) ? 0xFFu : (unsigned)r) << 16;
}
#endif

struct ScanHdr {
    float start_ori, end_ori;
    int first_kept, last_kept;     /* indices into raw[] (INT_MAX / -1 when none) */
    int half_idx;                  /* first valid-ring point whose (ori - startOri) > pi: halfPassed is true for i > half_idx */
    int n;                         /* cloudSize (:212) */
    int status;
    int max_ring;
    int n_sharp, n_less_sharp, n_flat, n_less_flat;
    float so_lo_up, so_hi_dn;      /* ll_f32_ceil(startOri - pi/2), ll_f32_floor(startOri + 3pi/2): the wrap tests of :181-188 in f32 */
    int lf_strided;                /* 1: the slot's less-flat cloud is ring-strided (an extracted scan); 0: contiguous (ll_upload_features) */
};

struct PairHdr {
    int n_edge, n_plane, n_plane_sel;
    int target_slot;               /* -1 = carry */
};

struct LLView {
    /* configuration */
    int B, NP, T, R, ring_model, max_ring, write_curv;
    int distortion;                /* 0: the reference's build (DISTORTION 0); 1: per-point interpolation ratio s in TransformToStart and the factors */
    int sort_match_any;            /* k_ring_features: rank the voxel sort's records by a match-any instead of the returning LDS add's lane order (ll_features.hip) */
    int org_small;                 /* calls of at most this many scans take the tile-parallel organise kernels (LL_ORG_SMALL; test override) */
    float thres, lower_bound, factor;
    const int *ring_thr;           /* [R + 1] ll_ring_thresholds: keys of the smallest t = z / sqrt(x^2 + y^2) of every ring */
    const int *ring_lut; int lut_nb; float lut_t0, lut_scale;   /* ll_ring_lut_build: first guess per t bucket (lut_nb = 0: unused) */
    double curv_thr, gap_thr;      /* 0.1 / 0.05 as double: the reference compares f32 against double literals */
    float curv_gt, curv_lt, gap_gt; /* the same comparisons in f32: c > curv_thr <=> c > curv_gt = ll_f32_floor(curv_thr), c < curv_thr <=>
                                     * c < curv_lt = ll_f32_ceil(curv_thr), g > gap_thr <=> g > gap_gt = ll_f32_floor(gap_thr) (ll_exact_math.h) */
    float leaf, inv_leaf;
    float nn_max;                  /* 25 */
    double nearby;                 /* 2.5 */
    double huber;
    int cap_sharp, cap_lsharp, cap_flat;      /* per-scan capacities R*12, R*120, R*24 */
    /* organise */
    const float4 *raw; const int *n_in;
    int raw_stride;                /* floats per resident raw point: 4, or 3 = x, y, z packed (ll_params.input_stride_floats); a slot's area is NP float4 either way */
    /* scratch of the tile-parallel organise path (calls of at most LL_ORG_SMALL scans), indexed by the position in the launch */
    float *ori; int8_t *ring;
    int *tile_hist; int *tile_base; int *tile_first_p; int *tile_first_kept; int *tile_last_kept;
    ScanHdr *hdr; int *ring_off;
    float4 *cloud; int ring_cap, CS;   /* laserCloud, ring r of slot s at cloud[s * CS + r * ring_cap]; CS = R * ring_cap */
    int8_t *label; float *curv;
    /* features */
    unsigned short *ring_rec;      /* [B][R][LL_REC_U16] k_ring_pick's lists of one ring */
    unsigned *ring_cnt;            /* [B][R] n_sharp | n_less_sharp << 8 | n_flat << 16 | 1 << 31 (the ring has segments): k_ring_pick -> k_ring_features, k_build_grid */
    int *ring_nlf;                 /* [B][R] less-flat points of the ring (k_ring_features) */
    int *tier_cnt; int *tier_list; /* max_ring_points > 2304 only: [4] and [3][B * R] -- the rings longer than 2304 / 3072 / 4608 points of the extract call in
                                    * flight as slot << 8 | ring, per capacity tier 1..3 (the organise stage appends, the tier launches of the ring kernels read) */
    int *lf_pre;                   /* [B][R+1] exclusive prefix of ring_nlf = the contiguous index of every ring's first less-flat point (k_build_grid) */
    float4 *sharp, *lsharp, *flat, *lflat; int LFS;   /* lflat of slot s at lflat + s * LFS, LFS = max(CS, NP): ring-strided rows or a contiguous upload */
    /* targets */
    float4 *carry_corner, *carry_surf; int *carry_cnt;    /* carry_cnt[2] */
    /* NN grids of every slot's less-sharp (0) / less-flat (1) cloud and of the carry: cell starts + cell-ordered (x,y,z,index) */
    int *gstart; float4 *gpts_c, *gpts_s;                 /* [B][2][NC+1], [B][cap_lsharp], [B][NP] */
    int *carry_gstart; float4 *carry_gpts_c, *carry_gpts_s;
    /* association (per query, then compacted) */
    int *eq_a, *eq_b;              /* [B][cap_sharp]  -1 = no correspondence */
    int *pq_a, *pq_b, *pq_c;       /* [B][cap_flat] */
    int *e_src, *e_a, *e_b;        /* compacted */
    int *p_src, *p_a, *p_b, *p_c;
    int *v_count; uint8_t *v_sel; float *v_w;
    PairHdr *pair;
    double *pose;                  /* [B][7] qx qy qz qw tx ty tz */
    double *pose_guess;            /* [B][7] para_q/para_t at entry of the hot path (laserOdometry.cpp:61-62) */
    double *neq;                   /* [B][44]: H[36] row-major, g[6], cost, rows */
    double *lm;                    /* [B][LL_LM_STRIDE] Levenberg-Marquardt state of the slot's current solve */
    int carry_slot;                /* the slot whose target is the carry (first slot of the batch) in the association call being enqueued */
    int *assoc_tgt;                /* [B] the target ll_associate_batch searched for the slot: -1 the carry, >= 0 that slot, < -1 never associated.
                                    * Vote, factors and the solves read the target from here: a later call over a sub-range cannot disagree
                                    * with the association about whose points the correspondences name */
    unsigned long long *dbg;       /* [16] phase-timing counters (only written by -DLL_PHASE_TIMING builds) */
};

#define LL_NEQ_STRIDE 44
#define LL_LM_STRIDE 80

/* ceres::Solver::Options fields the trust-region loop reads (laserOdometry.cpp:820-825 sets two, the rest are defaults) */
struct LLLmOpt {
    int max_num_iterations;
    double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    int jacobi_scaling;
    int nan_poisons_pose;          /* staged mapping solves (ll_map_lm_*): a NaN cost / row count in the summed record poisons the pose (ll_lm_step.h) */
};
void ll_launch_lm_solve(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st);
void ll_launch_lm_begin(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st);
void ll_launch_lm_propose(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st);
void ll_launch_lm_accept(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st);

/* Match-any over a wave: afterwards (mlo, mhi) = the lanes of `among` whose low `bits` bits of v equal this lane's.
 * Per bit one ballot and, per mask half, one three-input bit operation  m & ~(ballot ^ y), y = the lane's bit as 0 / ~0
 * (v_bitop3_b32, truth table 0x90 for inputs m, ballot, y).  Rank inside the group = ll_match_rank, size = ll_match_count. */
__device__ __forceinline__ void ll_match_any(int v, int bits, unsigned long long among, unsigned &mlo, unsigned &mhi)
{
    mlo = (unsigned)among; mhi = (unsigned)(among >> 32);
    for (int b = 0; b < bits; ++b) {
        const int y = __builtin_amdgcn_sbfe(v, (unsigned)b, 1u);           /* v_bfe_i32: the bit as 0 / ~0 in one instruction */
        const unsigned long long s = __ballot(y < 0);
        mlo = __builtin_amdgcn_bitop3_b32(mlo, (unsigned)s, (unsigned)y, 0x90);
        mhi = __builtin_amdgcn_bitop3_b32(mhi, (unsigned)(s >> 32), (unsigned)y, 0x90);
    }
}
__device__ __forceinline__ int ll_match_rank(unsigned mlo, unsigned mhi)     /* set bits below this lane */
{
    return (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
}
__device__ __forceinline__ int ll_match_count(unsigned mlo, unsigned mhi) { return __popc(mlo) + __popc(mhi); }

/* Cross-lane steps as DPP operands of the ALU instruction itself (row shifts inside the 16-lane rows, then the two row
 * broadcasts): ~10 cycles per dependent step against ~100 for a ds_bpermute (__shfl), and the kernels that use them are
 * bound by exactly such dependent chains.  A lane without a source keeps `old` (bound_ctrl off). */
#define LL_DPP_I(old, v, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rmask, 0xf, false)
/* inclusive prefix sum over the 64 lanes */
__device__ __forceinline__ int ll_wave_incl_scan(int v)
{
    v += LL_DPP_I(0, v, 0x111, 0xf);          /* row_shr:1 */
    v += LL_DPP_I(0, v, 0x112, 0xf);          /* row_shr:2 */
    v += LL_DPP_I(0, v, 0x114, 0xf);          /* row_shr:4 */
    v += LL_DPP_I(0, v, 0x118, 0xf);          /* row_shr:8  -> prefix inside every row */
    v += LL_DPP_I(0, v, 0x142, 0xa);          /* row_bcast:15 into rows 1, 3 */
    v += LL_DPP_I(0, v, 0x143, 0xc);          /* row_bcast:31 into rows 2, 3 */
    return v;
}
/* wave-wide reductions, result uniform (read from lane 63): the pattern of ll_wave_max_u32 (ll_features.hip) */
#define LL_WAVE_REDUCE(name, T, OP, IDENT, TOI, FROMI)                                                        \
    __device__ __forceinline__ T name(T v)                                                                    \
    {                                                                                                         \
        v = OP(v, FROMI(LL_DPP_I(TOI(v), TOI(v), 0xb1, 0xf)));      /* quad_perm [1,0,3,2] */                  \
        v = OP(v, FROMI(LL_DPP_I(TOI(v), TOI(v), 0x4e, 0xf)));      /* quad_perm [2,3,0,1] */                  \
        v = OP(v, FROMI(LL_DPP_I(TOI(IDENT), TOI(v), 0x114, 0xf))); /* row_shr:4 */                            \
        v = OP(v, FROMI(LL_DPP_I(TOI(IDENT), TOI(v), 0x118, 0xf))); /* row_shr:8 -> lanes 12..15 hold the row's result */ \
        v = OP(v, FROMI(LL_DPP_I(TOI(IDENT), TOI(v), 0x142, 0xa))); /* row_bcast:15 */                         \
        v = OP(v, FROMI(LL_DPP_I(TOI(IDENT), TOI(v), 0x143, 0xc))); /* row_bcast:31 */                         \
        return FROMI(__builtin_amdgcn_readlane(TOI(v), 63));                                                  \
    }
#define LL_OP_ADD(a, b) ((a) + (b))
#define LL_OP_OR(a, b) ((a) | (b))
#define LL_ID(x) (x)
LL_WAVE_REDUCE(ll_wave_sum_i32, int, LL_OP_ADD, 0, LL_ID, LL_ID)
LL_WAVE_REDUCE(ll_wave_or_u32, unsigned, LL_OP_OR, 0u, (int), (unsigned))
LL_WAVE_REDUCE(ll_wave_min_f32, float, fminf, __builtin_inff(), __float_as_int, __int_as_float)
LL_WAVE_REDUCE(ll_wave_max_f32, float, fmaxf, -__builtin_inff(), __float_as_int, __int_as_float)

/* groups of eight lanes (k_associate: eight lanes per query): the minimum of a 64-bit key over the group in every lane --
 * xor 1 and xor 2 inside the quads, then row_half_mirror (lane i <-> 7 - i: the other quad) -- and the broadcast of lane E's
 * value to the group: quad_perm [E, E, E, E], then the other quad takes it over by a row shift of four with a bank mask. */
__device__ __forceinline__ unsigned long long ll_min8_u64(unsigned long long k)
{
#define LL_MIN8_STEP(ctrl) do { \
        const unsigned lo = (unsigned)LL_DPP_I((int)(unsigned)k, (int)(unsigned)k, ctrl, 0xf), hi = (unsigned)LL_DPP_I((int)(unsigned)(k >> 32), (int)(unsigned)(k >> 32), ctrl, 0xf); \
        const unsigned long long k2 = ((unsigned long long)hi << 32) | lo; k = (k2 < k) ? k2 : k; } while (0)
    LL_MIN8_STEP(0xb1);           /* quad_perm [1,0,3,2] */
    LL_MIN8_STEP(0x4e);           /* quad_perm [2,3,0,1] */
    LL_MIN8_STEP(0x141);          /* row_half_mirror */
#undef LL_MIN8_STEP
    return k;
}
template <int E>
__device__ __forceinline__ int ll_bcast8(int v)
{
    static_assert(E >= 0 && E < 8, "lane of the group");
    constexpr int Q = (E & 3) * 0x55;                                                        /* quad_perm [E, E, E, E] */
    const int t = __builtin_amdgcn_update_dpp(v, v, Q, 0xf, 0xf, false);
    if (E < 4) return __builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xa, false);             /* row_shr:4 into banks 1, 3: the group's upper quad */
    return __builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);                        /* row_shl:4 into banks 0, 2: the group's lower quad */
}

/* exclusive prefix sum of one int per thread over a workgroup of NW waves: 64-lane DPP scan + the wave totals
 * through LDS (sc: >= NW ints).  Returns the exclusive prefix; total = workgroup sum.  Ends with a barrier. */
template <int NW>
__device__ __forceinline__ int ll_block_exscan_n(int v, int *sc, int &total)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int inc = ll_wave_incl_scan(v);
    if (lane == 63) sc[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int t = sc[w]; if (w < wave) base += t; tot += t; }
    total = tot;
    __syncthreads();
    return base + inc - v;
}
__device__ __forceinline__ int ll_block_exscan(int v, int *sc, int &total) { return ll_block_exscan_n<LL_BLOCK / 64>(v, sc, total); }

/* the target the slot's correspondences refer to: -1 = the carry, else a slot (its features).  Fixed by the association (assoc_tgt);
 * before any association: the rule of the call being enqueued (first slot of the range -> carry, else the slot before) */
__device__ __forceinline__ int ll_target_slot(const LLView &V, int s)
{
    const int t0 = V.assoc_tgt[s];
    if (t0 < -1) return (s == V.carry_slot) ? -1 : s - 1;
    return t0;
}
/* target clouds of slot s.  surf is addressed by PLACE (ll_common.h): ring-strided rows of an extracted slot, or contiguous */
__device__ __forceinline__ void ll_targets(const LLView &V, int s, const float4 *&corner, int &mc, const float4 *&surf, int &ms)
{
    const int t = ll_target_slot(V, s);
    if (t < 0) {
        corner = V.carry_corner; mc = V.carry_cnt[0]; surf = V.carry_surf; ms = V.carry_cnt[1];
    } else {
        corner = V.lsharp + (size_t)t * V.cap_lsharp; mc = V.hdr[t].n_less_sharp;
        surf = V.lflat + (size_t)t * V.LFS; ms = V.hdr[t].n_less_flat;
        if (V.hdr[t].status != 0) { mc = 0; ms = 0; }
    }
}

/* per-kernel HIP-event profiler (ll_api.hip); mark(id) = "kernel id starts here, the previous one ended" */
enum { LL_K_CLASSIFY = 0, LL_K_OFFSETS, LL_K_SCATTER, LL_K_RING_FEATURES, LL_K_ORGANIZE, LL_K_ASSOCIATE, LL_K_VOTE,
       LL_K_NORMAL_EQ, LL_K_GN_STEP, LL_K_GRID, LL_K_FIRST, LL_K_PICK,
       LL_K_ASSOC_STAGE,     /* not a kernel: k_build_grid and k_associate of a hot-path call running side by side on two streams (ll_api.hip) */
       LL_K_COUNT, LL_K_END = -1 };
/* ---- mapping stage (ll_mapping.hip): one scan against the clouds gathered from the cube map ---- */
struct LLGrid3 {                       /* dense cell grid over a cloud's bounding box */
    float org[3]; float cell; int dim[3]; int ncell;
    int *start;                        /* [ncell + 1] first cell-ordered point of every cell */
    int *cursor;                       /* [ncell] scatter cursors */
    float4 *pts;                       /* cell-ordered (x, y, z, original index as int bits) */
};
struct LLMapView {
    const float4 *map[2]; int n_map[2];          /* 0: laserCloudCornerFromMap, 1: laserCloudSurfFromMap */
    LLGrid3 grid[2];
    const float4 *stk[2]; int n_stk[2];          /* laserCloudCornerStack / laserCloudSurfStack */
    unsigned char *ok[2];                        /* per stack point: produced a residual block */
    double *qa, *qb, *qn, *qd;                   /* per stack point: line points a, b [n][3]; plane normal [n][3], d [n] */
    int *src[2];                                 /* residual blocks in stack order: stack index ... */
    double *fa, *fb, *fn, *fd;                   /* ... and their line points / plane */
    int *counts;                                 /* [2] edge blocks, plane blocks */
    double *pose, *neq, *lm;                     /* one slot: parameters[7], normal equations [44], LM state */
    double huber;
    const int *gid[2];                           /* tile shard: global id per map point (nullptr: position in the cloud) */
    float4 *nn_pt[2]; int *nn_id[2];             /* tile shard: per stack point 5 x (x, y, z, distance) and 5 global ids */
    int row_rank, row_world;                     /* k_map_normal_eq sums the blocks i with i % row_world == row_rank (0, 1: all) */
    double *neq_part; unsigned *neq_ticket;      /* k_map_normal_eq: per-workgroup partial sums [LL_NEQ_NB][28], arrival counter */
    unsigned *lm_go;                             /* k_map_lm_solve: the round workgroup 0 has released */
    unsigned long long *cpub[2]; int cpub_tag;   /* k_map_compact: launch tag << 32 | valid blocks of workgroup b, per cloud type */
};
#define LL_NEQ_NB 16                             /* workgroups of k_map_normal_eq */
/* device-to-device copy / small constant fill as kernel launches: on this stack an asynchronous copy costs the host ~26 us,
 * a launch ~8 us, and the mapping stage issues dozens of them per frame */
void *ll_pinned_scratch(size_t bytes);                                               /* per host thread, page-locked, >= bytes (NULL on failure) */
int ll_read_back(void *host_dst, const void *dev_src, size_t bytes, hipStream_t st);  /* small device -> host copy + stream sync through it */
void ll_copy_d2d(void *dst, const void *src, size_t bytes, hipStream_t st);
void ll_fill_words(int *dst, int n, int a, int b, int split, hipStream_t st);        /* dst[i] = i < split ? a : b */
void ll_map_launch_bbox(const float4 *pts, int n, int *bbox_dev, hipStream_t st);
void ll_map_bbox_to_grid(const int bbox_host[6], int n, int max_cells, LLGrid3 *G);
void ll_map_launch_build(const LLGrid3 &G, const float4 *pts, int n, int *tile_sum, hipStream_t st);
void ll_map_launch_associate(const LLMapView &M, hipStream_t st);
void ll_map_launch_knn_partial(const LLMapView &M, hipStream_t st);
void ll_map_launch_associate_merged(const LLMapView &M, int n_parts, const float4 *const pt_all[2], const int *const id_all[2], hipStream_t st);
void ll_map_launch_normal_eq(const LLMapView &M, hipStream_t st);
void ll_map_launch_lm_solve(const LLMapView &M, const LLLmOpt &o, hipStream_t st);
void ll_map_launch_rows(const LLMapView &M, double *r, double *Jq, double *Jt, hipStream_t st);

void ll_launch_factor_blocks(const double *pose, int n_e, const double *edge, int n_p, const double *plane, int n_n, const double *pnorm,
                             const double *s_ep, double *r, double *Jq, double *Jt, hipStream_t st);   /* s_ep: [n_e + n_p] or null (all ones) */                        /* ll_functors.hip */

/* ---- whole-cloud VoxelGrid (ll_voxel.hip) ---- */
struct LLVoxSeg;
struct LLVoxWork {
    int cap, max_seg;
    int *segid, *bbox, *seg_off, *flag, *rank, *vals, *tmp_vals, *hist, *tile_sum, *seg_count;
    unsigned long long *keys, *tmp_keys, *or_and;
    unsigned long long *pub;           /* [LL_VX_FUSED_WGS] k_vx_finish: valid bit | heads of workgroup b (zeroed by k_vx_keys) */
    LLVoxSeg *sp;
};
#define LL_VX_FUSED_WGS 256            /* k_vx_finish handles clouds of up to 256 workgroups x 256 points in one launch */
size_t ll_vox_work_bytes(int cap, int max_seg);
void ll_vox_work_carve(void *base, int cap, int max_seg, LLVoxWork *W);
int ll_voxel_grid_segments(const float4 *pts, int n, int nseg, float leaf, const LLVoxWork &W, float4 *out, int *n_out_dev, hipStream_t st,
                           int max_seg_len = 0);   /* 0 = ok; max_seg_len: the longest segment when the host knows it (<= 8192: one sort launch) */
void ll_device_exscan(int *data, int n, int *tile_sum, hipStream_t st);

/* raise a kernel's dynamic-LDS limit once per (kernel, device): the cache is per device, so contexts on several GPUs of
 * one process each get the attribute (a race between host threads only repeats the call) */
#define LL_MAX_DEVICES 64
template <typename K>
static inline void ll_ensure_dynamic_lds(K kernel, size_t bytes, size_t (&cache)[LL_MAX_DEVICES])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LL_MAX_DEVICES) dev = 0;
    if (bytes > cache[dev]) { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); cache[dev] = bytes; }
}

struct LLProfiler;
void ll_prof_mark(LLProfiler *p, int kernel_id, hipStream_t st);

/* launchers implemented in the per-stage .hip files */
void ll_launch_organize(const LLView &V, int first, int count, hipStream_t st, LLProfiler *prof);
void ll_launch_cloud_flatten(const LLView &V, int slot, float4 *dst, hipStream_t st);
void ll_launch_lflat_flatten(const LLView &V, int slot, float4 *dst, hipStream_t st);   /* the slot's less-flat cloud, contiguous (dst: >= NP points) */
void ll_launch_features(const LLView &V, int first, int count, size_t lds_bytes, hipStream_t st, LLProfiler *prof, int what = 3);   /* what: 1 = the pick kernel, 2 = the voxel kernel */
void ll_launch_pick(const LLView &V, int first, int count, hipStream_t st);
void ll_launch_associate(const LLView &V, int first, int count, hipStream_t st, LLProfiler *prof);
void ll_launch_build_grid(const LLView &V, int first, int count, int carry, hipStream_t st, LLProfiler *prof);
void ll_launch_vote(const LLView &V, int first, int count, int enable, hipStream_t st, LLProfiler *prof);
void ll_launch_vote_points(const float4 *src, const float4 *tgt, int n, int regions, int *vc, uint8_t *vs, float *vw, hipStream_t st);
void ll_launch_normal_equations(const LLView &V, int first, int count, int do_step, hipStream_t st, LLProfiler *prof);
void ll_launch_gn_step(const LLView &V, int first, int count, hipStream_t st, LLProfiler *prof);
void ll_launch_rows(const LLView &V, int slot, const double *pose7_dev, double *r, double *Jq, double *Jt, hipStream_t st);
size_t ll_features_lds_bytes(int max_ring);
int ll_lds_atomic_order_ok(hipStream_t st);              /* 1 / 0 / -1 (could not run): the LDS serves the lanes of one returning add in lane order (ll_features.hip: the sort's ranks) */
void ll_launch_debug_exact_math(const LLView &V, int op, const float *a, const float *b, const float *c, int n, float *out, hipStream_t st);
