/*
 * ll_cubemap.hip -- SURVEY 8f #2, second stage: laserMapping's cube map on the device
 * (/root/reference/src/laserMapping.cpp:1584-1821 before the optimisation, :2101-2165 after it).
 *
 * The reference keeps 21 x 21 x 11 cubes of 50 m, each a pcl::PointCloud for corners and one for surfs (:74-75).  Here a
 * cube is an (offset, count) pair into one HBM pool per cloud type; the pair table lives on the host, where the
 * reference's pointer shuffling (the six shift loops, :1595-1778) becomes swapping pairs.  Point data never leaves the
 * device:
 *   prepare   centre cube, shifts, the 5 x 5 x 3 valid cubes; k_cm_copy gathers their clouds, in the reference's loop
 *             order, into the ll_map's search clouds (:1803-1808) and the grids are rebuilt; the scan's less-sharp /
 *             less-flat clouds are down-sized by ll_voxel_grid_segments (:1813-1821).
 *   optimize  ll_map_optimize (:1822-2100).
 *   update    k_cm_assign: pointAssociateToMap + cube index per stack point (:2103-2148, same f32 / f64 mix); a stable
 *             sort by cube keeps the stack order inside a cube; every valid cube's old points followed by its new ones
 *             form one segment of a single ll_voxel_grid_segments call (:2151-2165); the filtered clouds are appended
 *             to the pool and the pairs repointed.  Cubes outside the valid set only get their new points appended.
 * Pool space of replaced clouds is reclaimed by compacting into the second pool when the first is 3/4 full.
 */
#include "ll_internal.h"
#include <algorithm>

#define CM_W 21
#define CM_H 21
#define CM_D 11
#define CM_N (CM_W * CM_H * CM_D)     /* 4851 (:53) */
#define CM_MAX_OPS 16384

struct CmOp { int kind, src, cnt, dst, tag; };   /* kind 0: pool[src + i]; kind 1: points[index[src + i]]; tag + i: global id */

/* tile-parallel mapping: the rank that keeps a cube, from the cube's position in the WORLD (array index
 * minus the running centre, which the shifts preserve) so that ownership never changes when the array shifts */
__host__ __device__ __forceinline__ int cm_owner(int wi, int wj, int wk, int world)
{
    const int h = (wi + 3 * wj + 5 * wk) % world;            /* any distance from the origin: the remainder is folded to 0 .. world - 1 */
    return h < 0 ? h + world : h;
}
#define CM_GID_SHIFT 20                  /* global id = position in the valid list << 20 | position in the cube */

struct ll_cubemap {
    ll_ctx *ctx = nullptr;
    ll_map *map = nullptr;
    float leaf[2] = {0.4f, 0.8f};
    int cen[3] = {10, 10, 5};
    int cap_last[2] = {0, 0};
    size_t cap_pool = 0, top[2] = {0, 0};
    float4 *pool[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int cur[2] = {0, 0};
    std::vector<int> off[2], cnt[2];
    int valid[125]; int n_valid = 0;
    int rank = 0, world = 1;                      /* tile shard: this map keeps the cubes with cm_owner() == rank */
    float4 *d_last = nullptr;
    /* ll_cubemap_update runs both cloud types through every stage before it synchronises: one set of buffers per type */
    float4 *d_tp[2] = {nullptr, nullptr}, *d_work[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr};
    int cap_work = 0;
    CmOp *d_ops = nullptr;
    int *d_addcnt = nullptr, *d_nout = nullptr;   /* [2][CM_N + 1]; [4]: stack sizes of prepare, filtered sizes of update */
    unsigned long long *d_keys[2] = {nullptr, nullptr}; int *d_vals[2] = {nullptr, nullptr};
    void *vox_mem = nullptr; LLVoxWork W;         /* prepare's filters and update's corner filter */
    void *vox_mem2 = nullptr; LLVoxWork W2;       /* update's surface filter */
    void *sort_mem = nullptr; LLVoxWork WS;       /* scratch of the by-cube sort (same layout, stack-sized) */
    std::vector<void *> allocs;
    std::string err;
    bool broken = false;                          /* an update failed half-way: the pair tables no longer describe the pools */
};

#define CM_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) { cm->err = std::string(#call) + ": " + hipGetErrorString(e_); return LL_ERR_HIP; } \
    } while (0)

/* ------------------------------------------------------------------ kernels */
/* the descriptors travel as a kernel ARGUMENT (<= CM_PACK of them per launch): no copy of a host array to wait for */
#define CM_PACK 160
struct CmOpPack { CmOp op[CM_PACK]; };
__global__ __launch_bounds__(256) void k_cm_copy(const float4 *pool, const float4 *points, const int *index, CmOpPack ops, int nops, float4 *dst, int *gid)
{
    const int o = blockIdx.y;
    if (o >= nops) return;
    const CmOp op = ops.op[o];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < op.cnt; i += gridDim.x * 256) {
        dst[op.dst + i] = op.kind == 0 ? pool[op.src + i] : points[index[op.src + i]];
        if (gid) gid[op.dst + i] = op.tag + i;
    }
}

/* pointAssociateToMap (:125-134) + the cube of the result (:2108-2125).  key = cube index, or CM_N when outside the map */
__global__ __launch_bounds__(256) void k_cm_assign(const float4 *stack, int n, const double *pose, int cenx, int ceny, int cenz, int rank, int world,
                                                   float4 *tp, unsigned long long *keys, int *vals, int *addcnt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int cube = -1;                                            /* -1: no point / not counted */
    if (i < n) {
    const float4 po = stack[i];
    const double ux = pose[0], uy = pose[1], uz = pose[2], w = pose[3];
    const double v[3] = {(double)po.x, (double)po.y, (double)po.z};
    double uvx = uy * v[2] - uz * v[1], uvy = uz * v[0] - ux * v[2], uvz = ux * v[1] - uy * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    const float sx = (float)(((v[0] + w * uvx) + (uy * uvz - uz * uvy)) + pose[4]);
    const float sy = (float)(((v[1] + w * uvy) + (uz * uvx - ux * uvz)) + pose[5]);
    const float sz = (float)(((v[2] + w * uvz) + (ux * uvy - uy * uvx)) + pose[6]);
    tp[i] = make_float4(sx, sy, sz, po.w);
    int ci = (int)(((double)sx + 25.0) / 50.0) + cenx, cj = (int)(((double)sy + 25.0) / 50.0) + ceny, ck = (int)(((double)sz + 25.0) / 50.0) + cenz;
    if ((double)sx + 25.0 < 0) ci--;
    if ((double)sy + 25.0 < 0) cj--;
    if ((double)sz + 25.0 < 0) ck--;
    if (ci >= 0 && ci < CM_W && cj >= 0 && cj < CM_H && ck >= 0 && ck < CM_D &&
        (world == 1 || cm_owner(ci - cenx, cj - ceny, ck - cenz, world) == rank))            /* another rank's cube: not kept here */
        cube = ci + CM_W * cj + CM_W * CM_H * ck;
    keys[i] = (unsigned long long)(cube >= 0 ? cube : CM_N);
    vals[i] = i;
    }
    /* the points of a scan fall into a handful of cubes: one add per wave and cube, not one per point on the same address */
    unsigned long long todo = __ballot(cube >= 0);
    while (todo) {
        const int c0 = __shfl(cube, __ffsll((long long)todo) - 1);
        const unsigned long long same = __ballot(cube == c0);
        if ((threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&addcnt[c0], __popcll(same));
        todo &= ~same;
    }
}

int ll_sort_pairs(unsigned long long *keys, int *vals, unsigned long long *tmp_keys, int *tmp_vals, int n, int *hist, int *tile_sum,
                  unsigned long long *or_and_dev, hipStream_t st);

/* ------------------------------------------------------------------ host side */
template <typename T>
static bool cm_alloc(ll_cubemap *cm, T *&ptr, size_t count)
{
    void *p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) { cm->err = "hipMalloc failed (" + std::to_string(bytes) + " bytes)"; return false; }
    cm->allocs.push_back(p);
    ptr = (T *)p;
    return true;
}

extern "C" void ll_cubemap_destroy(ll_cubemap *cm)
{
    if (!cm) return;
    if (cm->ctx) { (void)hipSetDevice(cm->ctx->device); (void)hipStreamSynchronize(cm->ctx->stream); }
    if (cm->map) ll_map_destroy(cm->map);
    for (void *p : cm->allocs) (void)hipFree(p);
    delete cm;
}

extern "C" const char *ll_cubemap_last_error(const ll_cubemap *cm) { return cm ? cm->err.c_str() : "null cube map"; }

extern "C" int ll_cubemap_create(ll_ctx *ctx, float line_res, float plane_res, int max_scan_corner, int max_scan_surf, int pool_points, ll_cubemap **out)
{
    if (!ctx || !out) return LL_ERR_ARG;
    *out = nullptr;
    if (!(line_res > 0.0f) || !(plane_res > 0.0f) || max_scan_corner < 1 || max_scan_surf < 1 || pool_points < 4096 || pool_points > (1 << 26)) {
        ctx->err = "bad cube map parameters"; return LL_ERR_ARG;
    }
    LL_HIP(hipSetDevice(ctx->device));
    ll_cubemap *cm = new ll_cubemap();
    cm->ctx = ctx;
    cm->leaf[0] = line_res; cm->leaf[1] = plane_res;
    cm->cap_last[0] = max_scan_corner; cm->cap_last[1] = max_scan_surf;
    cm->cap_pool = (size_t)pool_points;
    const int cap_from_map = (int)std::min<size_t>(cm->cap_pool, (size_t)1 << 24);
    int rc = ll_map_create(ctx, cap_from_map, cap_from_map, max_scan_corner, max_scan_surf, &cm->map);
    if (rc != LL_OK) { delete cm; return rc; }
    const int mx_last = std::max(max_scan_corner, max_scan_surf);
    if (cap_from_map < mx_last) { ctx->err = "pool_points must be at least the scan capacity"; ll_cubemap_destroy(cm); return LL_ERR_ARG; }
    cm->cap_work = cap_from_map;
    bool ok = true;
    for (int w = 0; w < 2 && ok; ++w) {
        ok = ok && cm_alloc(cm, cm->pool[w][0], cm->cap_pool) && cm_alloc(cm, cm->pool[w][1], cm->cap_pool);
        cm->off[w].assign(CM_N, 0); cm->cnt[w].assign(CM_N, 0);
    }
    ok = ok && cm_alloc(cm, cm->d_last, (size_t)mx_last);
    for (int w = 0; w < 2; ++w) {
        ok = ok && cm_alloc(cm, cm->d_tp[w], (size_t)mx_last) && cm_alloc(cm, cm->d_work[w], (size_t)cm->cap_work) && cm_alloc(cm, cm->d_out[w], (size_t)cm->cap_work);
        ok = ok && cm_alloc(cm, cm->d_keys[w], (size_t)mx_last) && cm_alloc(cm, cm->d_vals[w], (size_t)mx_last);
    }
    ok = ok && cm_alloc(cm, cm->d_ops, CM_MAX_OPS) && cm_alloc(cm, cm->d_addcnt, 2 * (CM_N + 1)) && cm_alloc(cm, cm->d_nout, 4);
    if (ok) {
        const size_t vb = ll_vox_work_bytes(cm->cap_work, 128), sb = ll_vox_work_bytes(mx_last, 1);
        unsigned char *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
        ok = cm_alloc(cm, p1, vb) && cm_alloc(cm, p2, sb) && cm_alloc(cm, p3, vb);
        if (ok) {
            cm->vox_mem = p1; cm->sort_mem = p2; cm->vox_mem2 = p3;
            ll_vox_work_carve(p1, cm->cap_work, 128, &cm->W); ll_vox_work_carve(p2, mx_last, 1, &cm->WS); ll_vox_work_carve(p3, cm->cap_work, 128, &cm->W2);
        }
    }
    if (!ok) { ctx->err = cm->err; ll_cubemap_destroy(cm); return LL_ERR_HIP; }
    *out = cm;
    return LL_OK;
}

static int cm_run_ops(ll_cubemap *cm, const std::vector<CmOp> &ops, const float4 *pool, const float4 *points, const int *index, float4 *dst, int *gid = nullptr)
{
    hipStream_t st = cm->ctx->stream;
    for (size_t o0 = 0; o0 < ops.size(); o0 += CM_PACK) {
        const int n = (int)std::min<size_t>(CM_PACK, ops.size() - o0);
        CmOpPack pk;
        int mx = 1;
        for (int i = 0; i < n; ++i) { pk.op[i] = ops[o0 + i]; mx = std::max(mx, ops[o0 + i].cnt); }
        hipLaunchKernelGGL(k_cm_copy, dim3(std::min(64, (mx + 255) / 256), n), dim3(256), 0, st, pool, points, index, pk, n, dst, gid);
    }
    return LL_OK;
}

/* one step of a shift loop (:1598-1778) on the pair tables */
static void cm_shift(ll_cubemap *cm, int axis, int dir)
{
    const int dim[3] = {CM_W, CM_H, CM_D}, stride[3] = {1, CM_W, CM_W * CM_H};
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    for (int w = 0; w < 2; ++w)
        for (int u = 0; u < dim[a1]; ++u) for (int v = 0; v < dim[a2]; ++v) {
            const int base = u * stride[a1] + v * stride[a2];
            std::vector<int> &off = cm->off[w], &cnt = cm->cnt[w];
            if (dir > 0) {
                for (int i = dim[axis] - 1; i >= 1; --i) { off[base + i * stride[axis]] = off[base + (i - 1) * stride[axis]]; cnt[base + i * stride[axis]] = cnt[base + (i - 1) * stride[axis]]; }
                cnt[base] = 0; off[base] = 0;
            } else {
                for (int i = 0; i < dim[axis] - 1; ++i) { off[base + i * stride[axis]] = off[base + (i + 1) * stride[axis]]; cnt[base + i * stride[axis]] = cnt[base + (i + 1) * stride[axis]]; }
                cnt[base + (dim[axis] - 1) * stride[axis]] = 0; off[base + (dim[axis] - 1) * stride[axis]] = 0;
            }
        }
}

/* live clouds -> the other pool, back to back */
static int cm_compact(ll_cubemap *cm, int w)
{
    std::vector<CmOp> ops;
    size_t top = 0;
    for (int c = 0; c < CM_N; ++c)
        if (cm->cnt[w][c] > 0) { ops.push_back({0, cm->off[w][c], cm->cnt[w][c], (int)top, 0}); top += (size_t)cm->cnt[w][c]; }
    const int other = cm->cur[w] ^ 1;
    int rc = cm_run_ops(cm, ops, cm->pool[w][cm->cur[w]], nullptr, nullptr, cm->pool[w][other]); if (rc) return rc;
    size_t k = 0;
    for (int c = 0; c < CM_N; ++c) if (cm->cnt[w][c] > 0) cm->off[w][c] = ops[k++].dst;
    cm->cur[w] = other; cm->top[w] = top;
    return LL_OK;
}

static int cm_reserve(ll_cubemap *cm, int w, size_t need)
{
    if (cm->top[w] + need > cm->cap_pool * 3 / 4) { int rc = cm_compact(cm, w); if (rc) return rc; }
    if (cm->top[w] + need > cm->cap_pool) { cm->err = "cube map pool exhausted (pool_points too small)"; return LL_ERR_CAPACITY; }
    return LL_OK;
}

static int cm_prepare(ll_cubemap *cm, const double *t_w3, const ll_point *corner_last, int n_corner, const ll_point *surf_last, int n_surf, bool on_device)
{
    if (!cm || !t_w3) return LL_ERR_ARG;
    if (n_corner < 0 || n_surf < 0 || (!corner_last && n_corner > 0) || (!surf_last && n_surf > 0)) { cm->err = "bad scan clouds"; return LL_ERR_ARG; }
    if (n_corner > cm->cap_last[0] || n_surf > cm->cap_last[1]) { cm->err = "scan cloud larger than the capacity given to ll_cubemap_create"; return LL_ERR_CAPACITY; }
    if (cm->broken) { cm->err = "the cube map is unusable: an earlier ll_cubemap_update failed half-way"; return LL_ERR_STATE; }
    CM_HIP(hipSetDevice(cm->ctx->device));
    hipStream_t st = cm->ctx->stream;
    const int dim[3] = {CM_W, CM_H, CM_D};
    int cc[3];
    for (int k = 0; k < 3; ++k) {
        cc[k] = (int)((t_w3[k] + 25.0) / 50.0) + cm->cen[k];                       /* :1584-1586 */
        if (t_w3[k] + 25.0 < 0) cc[k]--;                                           /* :1588-1593 */
    }
    for (int k = 0; k < 3; ++k) {
        while (cc[k] < 3) { cm_shift(cm, k, +1); cc[k]++; cm->cen[k]++; }          /* :1595-1625 and the J, K twins */
        while (cc[k] >= dim[k] - 3) { cm_shift(cm, k, -1); cc[k]--; cm->cen[k]--; }
    }
    cm->n_valid = 0;
    for (int i = cc[0] - 2; i <= cc[0] + 2; i++) for (int j = cc[1] - 2; j <= cc[1] + 2; j++) for (int k = cc[2] - 1; k <= cc[2] + 1; k++)   /* :1783-1801 */
        if (i >= 0 && i < CM_W && j >= 0 && j < CM_H && k >= 0 && k < CM_D) cm->valid[cm->n_valid++] = i + CM_W * j + CM_W * CM_H * k;
    /* laserCloudCornerFromMap / SurfFromMap (:1803-1808) */
    int n_from[2];
    for (int w = 0; w < 2; ++w) {
        std::vector<CmOp> ops; size_t tot = 0;
        for (int v = 0; v < cm->n_valid; ++v) {
            const int c = cm->valid[v];
            if (cm->cnt[w][c] >= (1 << CM_GID_SHIFT) && cm->world > 1) { cm->err = "a cube holds more points than a tile shard can number"; return LL_ERR_CAPACITY; }
            if (cm->cnt[w][c] > 0) { ops.push_back({0, cm->off[w][c], cm->cnt[w][c], (int)tot, v << CM_GID_SHIFT}); tot += (size_t)cm->cnt[w][c]; }
        }
        if (tot > (size_t)cm->map->cap_map[w]) { cm->err = "the valid cubes hold more points than the search cloud capacity"; return LL_ERR_CAPACITY; }
        int rc = cm_run_ops(cm, ops, cm->pool[w][cm->cur[w]], nullptr, nullptr, cm->map->d_map[w], cm->world > 1 ? cm->map->d_gid[w] : nullptr); if (rc) return rc;
        n_from[w] = (int)tot;
    }
    ll_map_rebuild_begin(cm->map, n_from[0], n_from[1]);                           /* the clouds' bounding boxes: enqueued, read below */
    /* laserCloudCornerStack / SurfStack: the scan's clouds down-sized (:1813-1821) -- both filters enqueued back to back */
    const ll_point *src[2] = {corner_last, surf_last}; const int n_in[2] = {n_corner, n_surf};
    for (int w = 0; w < 2; ++w) {
        if (n_in[w] > 0) {
            CM_HIP(hipMemcpyAsync(cm->d_last, src[w], (size_t)n_in[w] * sizeof(ll_point), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
            ll_fill_words(cm->W.seg_off, 2, 0, n_in[w], 1, st);                    /* one segment: {0, n} */
            if (ll_voxel_grid_segments(cm->d_last, n_in[w], 1, cm->leaf[w], cm->W, cm->map->d_stk[w], cm->d_nout + w, st)) { cm->err = "voxel filter: read-back failed"; return LL_ERR_HIP; }
        }
    }
    /* ONE synchronisation for everything the host needs from the above: the two boxes (grid dimensions) and the two sizes */
    {
        int *pin = (int *)ll_pinned_scratch(16 * sizeof(int));
        if (!pin) { cm->err = "no page-locked scratch"; return LL_ERR_HIP; }
        CM_HIP(hipMemcpyAsync(pin, cm->map->d_bbox, 12 * sizeof(int), hipMemcpyDeviceToHost, st));
        CM_HIP(hipMemcpyAsync(pin + 12, cm->d_nout, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        CM_HIP(hipStreamSynchronize(st));
        int bbox[12];
        for (int k = 0; k < 12; ++k) bbox[k] = pin[k];
        for (int w = 0; w < 2; ++w) cm->map->M.n_stk[w] = n_in[w] > 0 ? pin[12 + w] : 0;
        ll_map_rebuild_finish(cm->map, bbox);
    }
    CM_HIP(hipGetLastError());
    return LL_OK;
}

/* keep only the cubes of rank `rank` out of `world` (before the first scan); the searches then go through
 * ll_map_knn_partial / ll_map_associate_merged on ll_cubemap_map() */
extern "C" int ll_cubemap_set_shard(ll_cubemap *cm, int rank, int world)
{
    if (!cm) return LL_ERR_ARG;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) { cm->err = "bad shard"; return LL_ERR_ARG; }
    for (int w = 0; w < 2; ++w) if (cm->top[w] != 0) { cm->err = "the cube map already holds points"; return LL_ERR_STATE; }
    CM_HIP(hipSetDevice(cm->ctx->device));
    const int rc = ll_map_use_ids(cm->map, world > 1);
    if (rc) { cm->err = cm->map->err; return rc; }
    cm->rank = rank; cm->world = world;
    return LL_OK;
}

extern "C" ll_map *ll_cubemap_map(ll_cubemap *cm) { return cm ? cm->map : nullptr; }

extern "C" int ll_cubemap_prepare(ll_cubemap *cm, const double *t_w3, const ll_point *corner_last, int n_corner, const ll_point *surf_last, int n_surf)
{
    return cm_prepare(cm, t_w3, corner_last, n_corner, surf_last, n_surf, false);
}

/* the scan of an extracted slot of the owning context: its less-sharp / less-flat clouds never leave the device
 * (what laserOdometry publishes as laser_cloud_corner_last / laser_cloud_surf_last, laserOdometry.cpp:898-910) */
extern "C" int ll_cubemap_process_slot(ll_cubemap *cm, double *pose_w7, int slot, int *ran)
{
    if (!cm || !pose_w7) return LL_ERR_ARG;
    ll_ctx *ctx = cm->ctx;
    if (slot < 0 || slot >= ctx->p.batch) { cm->err = "slot out of range"; return LL_ERR_ARG; }
    CM_HIP(hipSetDevice(ctx->device));
    ScanHdr h;
    CM_HIP(hipMemcpyAsync(&h, ctx->V.hdr + slot, sizeof(ScanHdr), hipMemcpyDeviceToHost, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    if (h.status != 0) { cm->err = "the slot holds no extracted scan"; return LL_ERR_STATE; }
    const LLView &V = ctx->V;
    /* the less-flat cloud of an extracted slot lies in ring rows (ll_common.h): closed up on the device into the context's staging
     * array, stream-ordered with everything below */
    const float4 *surf_last = V.lflat + (size_t)slot * V.LFS;
    if (h.lf_strided) { ll_launch_lflat_flatten(V, slot, ctx->cloud_flat, ctx->stream); surf_last = ctx->cloud_flat; }
    int rc = cm_prepare(cm, pose_w7 + 4, (const ll_point *)(V.lsharp + (size_t)slot * V.cap_lsharp), h.n_less_sharp,
                        (const ll_point *)surf_last, h.n_less_flat, true);
    if (rc) return rc;
    rc = ll_cubemap_optimize(cm, pose_w7, 2, nullptr, ran); if (rc) return rc;
    return ll_cubemap_update(cm, pose_w7);
}

extern "C" int ll_cubemap_optimize(ll_cubemap *cm, double *pose_w7, int n_outer, const ll_lm_options *opt, int *ran)
{
    if (!cm) return LL_ERR_ARG;
    const int rc = ll_map_optimize(cm->map, pose_w7, n_outer, opt, ran);
    if (rc) cm->err = cm->map->err;
    return rc;
}

extern "C" int ll_cubemap_update(ll_cubemap *cm, const double *pose_w7)
{
    if (!cm || !pose_w7) return LL_ERR_ARG;
    if (cm->broken) { cm->err = "the cube map is unusable: an earlier ll_cubemap_update failed half-way"; return LL_ERR_STATE; }
    CM_HIP(hipSetDevice(cm->ctx->device));
    hipStream_t st = cm->ctx->stream;
    ll_map *m = cm->map;
    CM_HIP(hipMemcpyAsync(m->M.pose, pose_w7, 7 * sizeof(double), hipMemcpyHostToDevice, st));
    /* Nothing above the "commit" line below changes the pair tables; a capacity overflow is detected before it (the map
     * stays as it was).  A HIP failure after it leaves the tables half-updated: the cube map is then marked unusable and
     * every later call returns LL_ERR_STATE instead of mapping on with an emptied neighbourhood. */
    struct Guard { ll_cubemap *cm; bool armed = false; ~Guard() { if (armed) cm->broken = true; } } guard{cm};
    /* Three synchronisations per frame instead of six: both cloud types go through a stage before the host looks at what the
     * stage produced.  Stage A: points -> cubes (k_cm_assign), sorted by cube.  Host: the cubes' new-point counts of both types. */
    const LLVoxWork *VW[2] = {&cm->W, &cm->W2};
    CM_HIP(hipMemsetAsync(cm->d_addcnt, 0, 2 * (CM_N + 1) * sizeof(int), st));
    for (int w = 0; w < 2; ++w) {
        const int ns = m->M.n_stk[w];
        if (ns > 0) {
            hipLaunchKernelGGL(k_cm_assign, dim3((ns + 255) / 256), dim3(256), 0, st, m->d_stk[w], ns, m->M.pose, cm->cen[0], cm->cen[1], cm->cen[2], cm->rank, cm->world,
                               cm->d_tp[w], cm->d_keys[w], cm->d_vals[w], cm->d_addcnt + w * (CM_N + 1));
            if (ll_sort_pairs(cm->d_keys[w], cm->d_vals[w], cm->WS.keys, cm->WS.vals, ns, cm->WS.hist, cm->WS.tile_sum, cm->WS.or_and, st)) { cm->err = "sort by cube: read-back failed"; return LL_ERR_HIP; }   /* by cube, stack order kept */
        }
    }
    std::vector<int> addcnt_all(2 * (CM_N + 1));
    if (ll_read_back(addcnt_all.data(), cm->d_addcnt, addcnt_all.size() * sizeof(int), st)) { cm->err = "read-back failed"; return LL_ERR_HIP; }
    /* Stage B: one voxel-grid segment per valid cube -- its cloud, then its new points (:2119-2125 push_back, :2151-2165 filter) */
    std::vector<int> goff[2], seg_off[2]; std::vector<char> is_valid(CM_N, 0);
    size_t tot[2] = {0, 0};
    for (int v = 0; v < cm->n_valid; ++v) is_valid[cm->valid[v]] = 1;
    for (int w = 0; w < 2; ++w) {
        const int *addcnt = addcnt_all.data() + w * (CM_N + 1);
        goff[w].assign(CM_N + 2, 0);
        for (int c = 0; c < CM_N; ++c) goff[w][c + 1] = goff[w][c] + addcnt[c];        /* the new points of cube c in the sorted order */
        std::vector<CmOp> ops; seg_off[w].assign(cm->n_valid + 1, 0);
        for (int v = 0; v < cm->n_valid; ++v) {
            const int c = cm->valid[v];
            seg_off[w][v] = (int)tot[w];
            if (cm->cnt[w][c] > 0) { ops.push_back({0, cm->off[w][c], cm->cnt[w][c], (int)tot[w], 0}); tot[w] += (size_t)cm->cnt[w][c]; }
            if (addcnt[c] > 0) { ops.push_back({1, goff[w][c], addcnt[c], (int)tot[w], 0}); tot[w] += (size_t)addcnt[c]; }
        }
        seg_off[w][cm->n_valid] = (int)tot[w];
        if (tot[w] > (size_t)cm->cap_work) { cm->err = "the valid cubes hold more points than the filter workspace"; return LL_ERR_CAPACITY; }
        int rc = cm_run_ops(cm, ops, cm->pool[w][cm->cur[w]], cm->d_tp[w], cm->d_vals[w], cm->d_work[w]); if (rc) return rc;
        if (tot[w] > 0) {
            CM_HIP(hipMemcpyAsync(VW[w]->seg_off, seg_off[w].data(), seg_off[w].size() * sizeof(int), hipMemcpyHostToDevice, st));   /* the vector lives until the synchronisation below */
            int max_seg = 0;
            for (int v = 0; v < cm->n_valid; ++v) max_seg = std::max(max_seg, seg_off[w][(size_t)v + 1] - seg_off[w][(size_t)v]);
            if (ll_voxel_grid_segments(cm->d_work[w], (int)tot[w], cm->n_valid, cm->leaf[w], *VW[w], cm->d_out[w], cm->d_nout + 2 + w, st, max_seg)) { cm->err = "voxel filter: read-back failed"; return LL_ERR_HIP; }
        }
    }
    /* Host: both filtered sizes and both sets of per-cube counts in one page-locked read-back */
    std::vector<int> seg_count[2]; int n_out[2] = {0, 0};
    {
        const size_t per = (size_t)cm->n_valid + 1;
        int *pin = (int *)ll_pinned_scratch(2 * per * sizeof(int));
        if (!pin) { cm->err = "no page-locked scratch"; return LL_ERR_HIP; }
        for (int w = 0; w < 2; ++w)
            if (tot[w] > 0) {
                CM_HIP(hipMemcpyAsync(pin + w * per, cm->d_nout + 2 + w, sizeof(int), hipMemcpyDeviceToHost, st));
                CM_HIP(hipMemcpyAsync(pin + w * per + 1, VW[w]->seg_count, (size_t)cm->n_valid * sizeof(int), hipMemcpyDeviceToHost, st));
            }
        CM_HIP(hipStreamSynchronize(st));
        for (int w = 0; w < 2; ++w) {
            seg_count[w].assign(cm->n_valid, 0);
            if (tot[w] > 0) { n_out[w] = pin[w * per]; for (int k = 0; k < cm->n_valid; ++k) seg_count[w][(size_t)k] = pin[w * per + 1 + k]; }
        }
    }
    /* Stage C: the pair tables.  Will it fit?  Decided for BOTH cloud types BEFORE any table changes (a compaction keeps the clouds
     * of the cubes outside the valid set only): a pool that is too small for either type leaves the whole cube map as it was, with
     * nothing of this stage enqueued, and the call can be repeated with a larger pool_points. */
    size_t need_w[2] = {0, 0};
    for (int w = 0; w < 2; ++w) {
        const int *addcnt = addcnt_all.data() + w * (CM_N + 1);
        /* pool space: the filtered valid cubes + the grown clouds of the other cubes that received points */
        size_t need = (size_t)n_out[w];
        for (int c = 0; c < CM_N; ++c) if (!is_valid[c] && addcnt[c] > 0) need += (size_t)cm->cnt[w][c] + (size_t)addcnt[c];
        need_w[w] = need;
        if (cm->top[w] + need > cm->cap_pool * 3 / 4) {
            size_t live = 0;
            for (int c = 0; c < CM_N; ++c) if (!is_valid[c]) live += (size_t)cm->cnt[w][c];
            if (live + need > cm->cap_pool) { cm->err = "cube map pool exhausted (pool_points too small)"; return LL_ERR_CAPACITY; }
        }
    }
    for (int w = 0; w < 2; ++w) {
        const int *addcnt = addcnt_all.data() + w * (CM_N + 1);
        const size_t need = need_w[w];
        /* ---- commit: the valid cubes' old clouds are dead from here on (not carried through a compaction) ---- */
        guard.armed = true;
        for (int v = 0; v < cm->n_valid; ++v) cm->cnt[w][cm->valid[v]] = 0;
        int rc = cm_reserve(cm, w, need); if (rc) return rc;
        float4 *pool = cm->pool[w][cm->cur[w]];
        if (n_out[w] > 0) ll_copy_d2d(pool + cm->top[w], cm->d_out[w], (size_t)n_out[w] * sizeof(float4), st);
        size_t at = cm->top[w];
        for (int v = 0; v < cm->n_valid; ++v) { const int c = cm->valid[v]; cm->off[w][c] = (int)at; cm->cnt[w][c] = seg_count[w][v]; at += (size_t)seg_count[w][v]; }
        std::vector<CmOp> grow;
        for (int c = 0; c < CM_N; ++c)
            if (!is_valid[c] && addcnt[c] > 0) {
                if (cm->cnt[w][c] > 0) grow.push_back({0, cm->off[w][c], cm->cnt[w][c], (int)at, 0});
                grow.push_back({1, goff[w][c], addcnt[c], (int)(at + (size_t)cm->cnt[w][c]), 0});
                cm->off[w][c] = (int)at; cm->cnt[w][c] += addcnt[c]; at += (size_t)cm->cnt[w][c];
            }
        rc = cm_run_ops(cm, grow, pool, cm->d_tp[w], cm->d_vals[w], pool); if (rc) return rc;
        cm->top[w] = at;
    }
    CM_HIP(hipStreamSynchronize(st));
    guard.armed = false;                                             /* both cloud types are consistent again */
    CM_HIP(hipGetLastError());
    return LL_OK;
}

/* :1584-2165 in one call: pose_w7 = parameters[7], in: the guess from transformAssociateToMap (:1581), out: optimised */
extern "C" int ll_cubemap_process(ll_cubemap *cm, double *pose_w7, const ll_point *corner_last, int n_corner, const ll_point *surf_last, int n_surf, int *ran)
{
    if (!cm || !pose_w7) return LL_ERR_ARG;
    int rc = ll_cubemap_prepare(cm, pose_w7 + 4, corner_last, n_corner, surf_last, n_surf); if (rc) return rc;
    rc = ll_cubemap_optimize(cm, pose_w7, 2, nullptr, ran); if (rc) return rc;
    return ll_cubemap_update(cm, pose_w7);
}

extern "C" int ll_cubemap_info(ll_cubemap *cm, int *cen3, int *counts4)
{
    if (!cm) return LL_ERR_ARG;
    if (cen3) for (int k = 0; k < 3; ++k) cen3[k] = cm->cen[k];
    if (counts4) { counts4[0] = cm->map->M.n_map[0]; counts4[1] = cm->map->M.n_map[1]; counts4[2] = cm->map->M.n_stk[0]; counts4[3] = cm->map->M.n_stk[1]; }
    return LL_OK;
}

extern "C" int ll_cubemap_download_cloud(ll_cubemap *cm, int which, ll_point *out, int cap, int *n)
{
    if (!cm || !n || which < 0 || which > 3) return LL_ERR_ARG;
    const int cnt = which < 2 ? cm->map->M.n_map[which] : cm->map->M.n_stk[which - 2];
    const float4 *src = which < 2 ? cm->map->d_map[which] : cm->map->d_stk[which - 2];
    *n = cnt;
    if (cnt > cap) { cm->err = "cloud capacity too small"; return LL_ERR_CAPACITY; }
    CM_HIP(hipSetDevice(cm->ctx->device));
    if (cnt > 0 && out) { CM_HIP(hipMemcpyAsync(out, src, (size_t)cnt * sizeof(ll_point), hipMemcpyDeviceToHost, cm->ctx->stream)); CM_HIP(hipStreamSynchronize(cm->ctx->stream)); }
    return LL_OK;
}

extern "C" int ll_cubemap_download_cube(ll_cubemap *cm, int surf, int cube, ll_point *out, int cap, int *n)
{
    if (!cm || !n || cube < 0 || cube >= CM_N) return LL_ERR_ARG;
    const int w = surf ? 1 : 0, cnt = cm->cnt[w][cube];
    *n = cnt;
    if (cnt > cap) { cm->err = "cube capacity too small"; return LL_ERR_CAPACITY; }
    CM_HIP(hipSetDevice(cm->ctx->device));
    if (cnt > 0 && out) {
        CM_HIP(hipMemcpyAsync(out, cm->pool[w][cm->cur[w]] + cm->off[w][cube], (size_t)cnt * sizeof(ll_point), hipMemcpyDeviceToHost, cm->ctx->stream));
        CM_HIP(hipStreamSynchronize(cm->ctx->stream));
    }
    return LL_OK;
}
