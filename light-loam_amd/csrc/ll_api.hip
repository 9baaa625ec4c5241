/*
 * ll_api.hip -- the C ABI of include/lightloam_hip.h: context, HBM-resident state, stage launches, transfers.
 * No CPU fallback anywhere: every stage is a HIP kernel launch on the ctx stream; if the device or the gfx950
 * code object is unusable, ll_create fails with LL_ERR_DEVICE.
 */
#include "ll_internal.h"
#include <mutex>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>

static const char *const kKernelNames[LL_K_COUNT] = {"k_classify", "k_offsets", "k_scatter", "k_ring_features", "k_organize",
                                                     "k_associate", "k_vote", "k_normal_equations", "k_gn_step", "k_build_grid", "k_first_kept", "k_ring_pick",
                                                     "k_build_grid||k_associate"};

void ll_prof_mark(LLProfiler *p, int kernel_id, hipStream_t st)
{
    if (!p || !p->on || p->n >= (int)p->ev.size()) return;
    /* a closing mark directly followed by an opening mark would cost two events; merge them */
    if (p->n > 0 && p->id[p->n - 1] == LL_K_END && kernel_id != LL_K_END) { /* keep the END event as the start */
        p->id[p->n - 1] = kernel_id | 0x100;   /* 0x100: this event also closed the previous kernel */
        return;
    }
    (void)hipEventRecord(p->ev[p->n], st);
    p->id[p->n] = kernel_id;
    p->n++;
}

static std::string g_create_err;

extern "C" int ll_abi_version(void) { return LL_ABI_VERSION; }

extern "C" void ll_default_params(ll_params *p, int n_scans)
{
    std::memset(p, 0, sizeof(*p));
    p->n_scans = n_scans;
    p->ring_model = 0;
    /* launch/aloam_velodyne_HDL_64.launch:8 (5), VLP_16 / HDL_32 launch files (0.3); node default 0.1 (scanRegistration.cpp:438) */
    p->minimum_range = (n_scans == 64) ? 5.0f : 0.3f;
    p->lower_bound = -24.9f;     /* scanRegistration.cpp:439 */
    p->up_bound = 2.0f;          /* :440 */
    p->max_points = 262144;
    p->max_ring_points = 2304;
    p->batch = 1;
    p->curv_threshold = 0.1f; p->gap_sq_threshold = 0.05f; p->leaf_size = 0.2f;
    p->nn_dist_sq_max = 25.0f; p->nearby_scan = 2.5f; p->huber_delta = 0.1f;
    p->write_curvature = 0;
    p->voxel_sort_ranks = 0;     /* auto */
    p->input_stride_floats = 4;  /* KITTI .bin / PointXYZ */
}

extern "C" const char *ll_last_error(const ll_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" void *ll_stream(ll_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

template <typename T>
static bool dev_alloc(ll_ctx *ctx, T *&ptr, size_t count, bool zero = true)
{
    void *p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) { g_create_err = "hipMalloc failed (" + std::to_string(bytes) + " bytes)"; return false; }
    if (zero && hipMemsetAsync(p, 0, bytes, ctx->stream) != hipSuccess) { g_create_err = "hipMemset failed"; return false; }
    ctx->allocs.push_back(p);
    ptr = (T *)p;
    return true;
}

extern "C" void ll_destroy(ll_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->allocs) (void)hipFree(p);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->ev_ok) for (auto &e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    for (auto &e : ctx->ev_ts) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_x) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->prof.ev) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int ll_create(int device, const ll_params *p, ll_ctx **out)
{
    if (!p || !out) return LL_ERR_ARG;
    *out = nullptr;
    if (p->n_scans < 1 || p->n_scans > LL_MAX_RINGS || p->batch < 1 || p->max_points < 32 || p->max_points > 400000 ||
        p->max_ring_points < 32 || p->max_ring_points > 8192) { g_create_err = "bad parameter"; return LL_ERR_ARG; }
    if (p->voxel_sort_ranks < 0 || p->voxel_sort_ranks > 1) { g_create_err = "voxel_sort_ranks must be 0 (auto) or 1 (match-any)"; return LL_ERR_ARG; }
    if (p->input_stride_floats != 0 && p->input_stride_floats != 3 && p->input_stride_floats != 4) { g_create_err = "input_stride_floats must be 4 (default) or 3"; return LL_ERR_ARG; }
    if (!(p->nn_dist_sq_max > 0.0f) || p->nn_dist_sq_max > 36.0f) {
        /* the cell search is exact for any radius; the bound keeps the worst case (no neighbour: every ring of 1 m cells
         * inside the radius is looked up) at 13 x 13 cells */
        g_create_err = "nn_dist_sq_max must be in (0, 36] (reference: DISTANCE_SQ_THRESHOLD = 25)";
        return LL_ERR_ARG;
    }
    if (p->ring_model == 0 && p->n_scans != 16 && p->n_scans != 32 && p->n_scans != 64) {
        g_create_err = "only support velodyne with 16, 32 or 64 scan line (scanRegistration.cpp:447-451); use ring_model 1 for the linear model";
        return LL_ERR_BAD_RINGS;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_create_err = "no usable HIP device (this library has no CPU fallback)";
        return LL_ERR_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return LL_ERR_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_err = "hipGetDeviceProperties failed"; return LL_ERR_DEVICE; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return LL_ERR_DEVICE;
    }
    ll_ctx *ctx = new ll_ctx();
    ctx->p = *p; ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { g_create_err = "hipStreamCreate failed"; delete ctx; return LL_ERR_DEVICE; }
    for (auto &e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { g_create_err = "hipEventCreate failed"; ll_destroy(ctx); return LL_ERR_DEVICE; }
    /* Once per device and process: does the LDS serve the lanes of one returning add in lane order (the fast ranking of the voxel
     * sort, ll_features.hip)?  A device that does not -- other silicon, a microcode change -- is served by the match-any ranking of
     * rounds 2-5 instead (bit-identical clouds, ~0.7 % of k_ring_features slower on MI355X); it is NOT refused.  A check that could not
     * run (allocation / launch failure) is an LL_ERR_HIP of this call and is not cached.  The cache is guarded: contexts are created from
     * several host threads (tests/native/tile_parallel_exits.cpp); a device index beyond the cache is checked every time. */
    int sort_match_any = (p->voxel_sort_ranks == 1) ? 1 : 0;
    if (!sort_match_any) {
        static std::mutex order_mu;
        static int order_checked[LL_MAX_DEVICES] = {0};        /* 0 not yet, 1 holds, 2 does not */
        std::lock_guard<std::mutex> lock(order_mu);
        int oc = device < LL_MAX_DEVICES ? order_checked[device] : 0;
        if (oc == 0) {
            const int r = ll_lds_atomic_order_ok(ctx->stream);
            if (r < 0) { g_create_err = "the LDS lane-order check could not run (hipMalloc / launch / copy failed)"; ll_destroy(ctx); return LL_ERR_HIP; }
            oc = r ? 1 : 2;
            if (device < LL_MAX_DEVICES) order_checked[device] = oc;
        }
        if (oc == 2) sort_match_any = 1;
    }
    ctx->ev_ok = true;

    LLView &V = ctx->V;
    std::memset(&V, 0, sizeof(V));
    V.sort_match_any = sort_match_any;
    const int B = p->batch, R = p->n_scans;
    const int NP = (p->max_points + LL_TILE - 1) / LL_TILE * LL_TILE;
    V.B = B; V.NP = NP; V.T = NP / LL_TILE; V.R = R; V.ring_model = p->ring_model;
    V.max_ring = p->max_ring_points; V.write_curv = p->write_curvature; V.distortion = p->distortion ? 1 : 0;
    V.thres = p->minimum_range; V.lower_bound = p->lower_bound;
    V.factor = (float)(R - 1) / (p->up_bound - p->lower_bound);                 /* scanRegistration.cpp:441 */
    /* the reference compares f32 values with the double literals 0.1 / 0.05; a float param widened to double is not
     * the same number, so the defaults are mapped back to the exact double literals */
    V.curv_thr = (p->curv_threshold == 0.1f) ? 0.1 : (double)p->curv_threshold;
    V.gap_thr = (p->gap_sq_threshold == 0.05f) ? 0.05 : (double)p->gap_sq_threshold;
    V.curv_gt = ll_f32_floor(V.curv_thr); V.curv_lt = ll_f32_ceil(V.curv_thr); V.gap_gt = ll_f32_floor(V.gap_thr);
    V.leaf = p->leaf_size; V.inv_leaf = 1.0f / p->leaf_size;
    V.nn_max = p->nn_dist_sq_max; V.nearby = (double)p->nearby_scan;
    V.huber = (p->huber_delta == 0.1f) ? 0.1 : (double)p->huber_delta;
    V.cap_sharp = R * LL_SEGS * LL_SHARP_PER_SEG; V.cap_lsharp = R * LL_SEGS * LL_LSHARP_PER_SEG; V.cap_flat = R * LL_SEGS * LL_FLAT_PER_SEG;
    V.carry_slot = 0;

    const size_t BN = (size_t)B * NP;
    bool ok = true;
    float4 *raw = nullptr; int *n_in = nullptr;
    ok = ok && dev_alloc(ctx, raw, BN, false) && dev_alloc(ctx, n_in, B);
    V.raw = raw; V.n_in = n_in;
    V.raw_stride = (p->input_stride_floats == 3) ? 3 : 4;
    V.org_small = LL_ORG_SMALL;
    if (const char *e = std::getenv("LIGHTLOAM_ORG_SMALL")) {                   /* tests: 0 sends every call through k_organize */
        const int v = std::atoi(e);
        if (v >= 0 && v <= LL_ORG_SMALL) V.org_small = v;
    }
    {   /* scratch of the tile-parallel organise path: org_small scans */
        const size_t S = (size_t)std::max(1, std::min(B, V.org_small)), SN = S * NP, ST = S * V.T;
        ok = ok && dev_alloc(ctx, V.ori, SN, false) && dev_alloc(ctx, V.ring, SN, false);
        ok = ok && dev_alloc(ctx, V.tile_hist, ST * R) && dev_alloc(ctx, V.tile_base, ST * R) && dev_alloc(ctx, V.tile_first_p, ST) &&
             dev_alloc(ctx, V.tile_first_kept, ST) && dev_alloc(ctx, V.tile_last_kept, ST);
    }
    ok = ok && dev_alloc(ctx, V.hdr, B) && dev_alloc(ctx, V.ring_off, (size_t)B * (R + 1));
    V.ring_cap = p->max_ring_points < NP ? p->max_ring_points : NP; V.CS = R * V.ring_cap;
    ok = ok && dev_alloc(ctx, V.cloud, (size_t)B * V.CS, false) && dev_alloc(ctx, ctx->cloud_flat, NP, false);
    ok = ok && dev_alloc(ctx, V.label, BN) && dev_alloc(ctx, V.curv, p->write_curvature ? BN : 1);
    ok = ok && dev_alloc(ctx, V.ring_rec, (size_t)B * R * LL_REC_U16) && dev_alloc(ctx, V.ring_cnt, (size_t)B * R);
    ok = ok && dev_alloc(ctx, V.ring_nlf, (size_t)B * R) && dev_alloc(ctx, V.lf_pre, (size_t)B * (R + 1));
    if (p->max_ring_points > 2304) ok = ok && dev_alloc(ctx, V.tier_cnt, 4) && dev_alloc(ctx, V.tier_list, (size_t)3 * B * R, false);   /* else both stay null: no tiers */
    {   /* ring thresholds of this sensor model (ll_exact_math.h), computed once with the same exact arithmetic */
        int *thr_dev = nullptr;
        std::vector<int32_t> thr((size_t)R + 1);
        ll_ring_thresholds(V.ring_model, R, V.lower_bound, V.factor, thr.data());
        ok = ok && dev_alloc(ctx, thr_dev, (size_t)R + 1, false);
        if (ok && (hipMemcpyAsync(thr_dev, thr.data(), ((size_t)R + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                   hipStreamSynchronize(ctx->stream) != hipSuccess)) { g_create_err = "threshold upload failed"; ok = false; }
        V.ring_thr = thr_dev;
        std::vector<int32_t> lut(LL_RING_LUT_MAX);
        V.lut_nb = ll_ring_lut_build(thr.data(), R, lut.data(), &V.lut_t0, &V.lut_scale);
        int *lut_dev = nullptr;
        ok = ok && dev_alloc(ctx, lut_dev, LL_RING_LUT_MAX, false);
        if (ok && (hipMemcpyAsync(lut_dev, lut.data(), LL_RING_LUT_MAX * sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                   hipStreamSynchronize(ctx->stream) != hipSuccess)) { g_create_err = "threshold upload failed"; ok = false; }
        V.ring_lut = lut_dev;
    }
    ok = ok && dev_alloc(ctx, V.sharp, (size_t)B * V.cap_sharp, false) && dev_alloc(ctx, V.lsharp, (size_t)B * V.cap_lsharp, false) &&
         dev_alloc(ctx, V.flat, (size_t)B * V.cap_flat, false);
    V.LFS = V.CS > NP ? V.CS : NP;                                                /* ring-strided rows of an extracted scan, or an uploaded contiguous cloud */
    ok = ok && dev_alloc(ctx, V.lflat, (size_t)B * V.LFS, false);
    ok = ok && dev_alloc(ctx, V.carry_corner, V.cap_lsharp) && dev_alloc(ctx, V.carry_surf, NP) && dev_alloc(ctx, V.carry_cnt, 2);
    ok = ok && dev_alloc(ctx, V.gstart, (size_t)B * 2 * LL_GSTRIDE) && dev_alloc(ctx, V.gpts_c, (size_t)B * V.cap_lsharp, false) && dev_alloc(ctx, V.gpts_s, BN, false);
    ok = ok && dev_alloc(ctx, V.carry_gstart, (size_t)2 * LL_GSTRIDE) && dev_alloc(ctx, V.carry_gpts_c, V.cap_lsharp, false) && dev_alloc(ctx, V.carry_gpts_s, NP, false);
    ok = ok && dev_alloc(ctx, V.eq_a, (size_t)B * V.cap_sharp) && dev_alloc(ctx, V.eq_b, (size_t)B * V.cap_sharp);
    ok = ok && dev_alloc(ctx, V.pq_a, (size_t)B * V.cap_flat) && dev_alloc(ctx, V.pq_b, (size_t)B * V.cap_flat) && dev_alloc(ctx, V.pq_c, (size_t)B * V.cap_flat);
    ok = ok && dev_alloc(ctx, V.e_src, (size_t)B * V.cap_sharp) && dev_alloc(ctx, V.e_a, (size_t)B * V.cap_sharp) && dev_alloc(ctx, V.e_b, (size_t)B * V.cap_sharp);
    ok = ok && dev_alloc(ctx, V.p_src, (size_t)B * V.cap_flat) && dev_alloc(ctx, V.p_a, (size_t)B * V.cap_flat) && dev_alloc(ctx, V.p_b, (size_t)B * V.cap_flat) && dev_alloc(ctx, V.p_c, (size_t)B * V.cap_flat);
    ok = ok && dev_alloc(ctx, V.v_count, (size_t)B * V.cap_flat) && dev_alloc(ctx, V.v_sel, (size_t)B * V.cap_flat) && dev_alloc(ctx, V.v_w, (size_t)B * V.cap_flat);
    ok = ok && dev_alloc(ctx, V.pair, B) && dev_alloc(ctx, V.pose, (size_t)B * 7) && dev_alloc(ctx, V.pose_guess, (size_t)B * 7) && dev_alloc(ctx, V.neq, (size_t)B * LL_NEQ_STRIDE);
    ok = ok && dev_alloc(ctx, ctx->d_tmp_pose, 7) && dev_alloc(ctx, V.dbg, 16) && dev_alloc(ctx, V.lm, (size_t)B * LL_LM_STRIDE);
    ok = ok && dev_alloc(ctx, V.assoc_tgt, B, false);
    if (!ok) { ll_destroy(ctx); return LL_ERR_HIP; }
    if (hipMemset(V.assoc_tgt, 0x80, (size_t)B * sizeof(int)) != hipSuccess) { g_create_err = "hipMemset failed"; ll_destroy(ctx); return LL_ERR_HIP; }   /* < -1: never associated */
    ctx->feat_lds = ll_features_lds_bytes(p->max_ring_points);
    if (ctx->feat_lds > 160 * 1024) { g_create_err = "max_ring_points needs more than 160 KiB of LDS"; ll_destroy(ctx); return LL_ERR_ARG; }
    /* identity pose guesses; status = "nothing extracted yet" */
    std::vector<double> ident((size_t)B * 7, 0.0);
    for (int b = 0; b < B; ++b) ident[(size_t)b * 7 + 3] = 1.0;
    if (hipMemcpyAsync(V.pose, ident.data(), ident.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(V.pose_guess, ident.data(), ident.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { g_create_err = "initial upload failed"; ll_destroy(ctx); return LL_ERR_HIP; }
    ctx->h_stage_pts = NP;
    if (hipHostMalloc((void **)&ctx->h_stage, (size_t)NP * sizeof(float4), hipHostMallocDefault) != hipSuccess) { g_create_err = "hipHostMalloc failed"; ll_destroy(ctx); return LL_ERR_HIP; }
    ctx->n_in_host.assign(B, 0);
    if (const char *e = std::getenv("LIGHTLOAM_TWO_STREAM")) ctx->two_stream = (std::atoi(e) != 0) ? 1 : 0;
    ctx->ts_pieces = LL_TWO_STREAM_PIECES;
    if (const char *e = std::getenv("LIGHTLOAM_TS_PIECES")) { const int v = std::atoi(e); if (v >= 2 && v <= LL_TWO_STREAM_MAX_PIECES) ctx->ts_pieces = v; }   /* A/B runs */
    *out = ctx;
    return LL_OK;
}

extern "C" int ll_synchronize(ll_ctx *ctx)
{
    if (!ctx) return LL_ERR_ARG;
    LL_HIP(hipSetDevice(ctx->device));
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

/* Every entry point that launches, allocates or copies starts here: the calling thread's current device becomes the
 * context's (kernel attributes such as the dynamic-LDS limit, hipMalloc and the launches themselves act on the CURRENT
 * device, not on the stream's), so contexts on several GPUs can be driven from one thread. */
static int ll_enter(ll_ctx *ctx)
{
    if (!ctx) return LL_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return LL_ERR_HIP; }
    return LL_OK;
}

static int check_range(ll_ctx *ctx, int first, int count)
{
    int rc = ll_enter(ctx); if (rc) return rc;
    if (first < 0 || count < 1 || first + count > ctx->p.batch) { ctx->err = "slot range out of bounds"; return LL_ERR_ARG; }
    return LL_OK;
}

extern "C" int ll_upload_scan(ll_ctx *ctx, int slot, const float *xyz, int stride, int n)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    if (n < 0 || stride < 3 || (!xyz && n > 0)) { ctx->err = "bad upload arguments"; return LL_ERR_ARG; }
    if (n > ctx->p.max_points) { ctx->err = "scan larger than max_points"; return LL_ERR_CAPACITY; }
    LLView &V = ctx->V;
    float4 *dst = const_cast<float4 *>(V.raw) + (size_t)slot * V.NP;
    if (n > 0) {
        /* the caller's stride -> the resident one (pcl::fromROSMsg into PointXYZ keeps x, y, z only, :105-106) */
        const int rs = V.raw_stride;
        if (stride == rs) std::memcpy(ctx->h_stage, xyz, (size_t)n * rs * 4);
        else if (rs == 4) for (int i = 0; i < n; ++i) ctx->h_stage[i] = make_float4(xyz[(size_t)i * stride], xyz[(size_t)i * stride + 1], xyz[(size_t)i * stride + 2], 0.0f);
        else { float *d3 = (float *)ctx->h_stage; for (int i = 0; i < n; ++i) { d3[3 * i] = xyz[(size_t)i * stride]; d3[3 * i + 1] = xyz[(size_t)i * stride + 1]; d3[3 * i + 2] = xyz[(size_t)i * stride + 2]; } }
        LL_HIP(hipMemcpyAsync(dst, ctx->h_stage, (size_t)n * rs * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->n_in_host[slot] = n;
    LL_HIP(hipMemcpyAsync(const_cast<int *>(V.n_in) + slot, &ctx->n_in_host[slot], sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

/* ---- streaming input (BASELINE config 5: scans arrive while earlier ones are processed) ----
 * ll_upload_scan_async enqueues the host -> device copy of one scan (x, y, z, . float4 stride, page-locked memory from
 * ll_host_alloc for a truly asynchronous transfer) on the context's COPY stream and returns; the buffer must stay untouched
 * until the copy has run.  The copy stream and the compute stream (every stage call) are ordered by events the caller
 * names (0 .. 7), never by blocking the host: ll_stream_record(ctx, stream, e) marks "everything enqueued on `stream` so far",
 * ll_stream_wait(ctx, stream, e) makes what is enqueued on `stream` from now on wait for that mark (a wait on an event that
 * was never recorded returns at once).  stream: LL_STREAM_COMPUTE 0, LL_STREAM_COPY 1.  With the slots in two halves and
 * two events per half this is a double buffer: the upload of one half overlaps the processing of the other
 * (bench.py --stream-input; ll_hot_path_chain continues a batch across the halves). */
static int ensure_copy_stream(ll_ctx *ctx)
{
    if (ctx->copy_stream) return LL_OK;
    LL_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    return LL_OK;
}

/* The point counts of asynchronously uploaded scans travel BY VALUE, as kernel arguments of a launch on the copy stream: an
 * asynchronous copy reads its host source when it RUNS, and a caller that is several generations ahead (the double buffer never
 * blocks the host) would by then have overwritten a per-slot staging cell with a later generation's count. */
#define LL_COUNTS_PER_LAUNCH 512
struct LLCounts { int v[LL_COUNTS_PER_LAUNCH]; };
__global__ void k_set_counts(int *dst, LLCounts c, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = c.v[i]; }
static void enqueue_counts(ll_ctx *ctx, int first, const int *n, int count)
{
    for (int c0 = 0; c0 < count; c0 += LL_COUNTS_PER_LAUNCH) {
        const int m = count - c0 < LL_COUNTS_PER_LAUNCH ? count - c0 : LL_COUNTS_PER_LAUNCH;
        LLCounts c;
        std::memcpy(c.v, n + c0, (size_t)m * sizeof(int));
        hipLaunchKernelGGL(k_set_counts, dim3((m + 255) / 256), dim3(256), 0, ctx->copy_stream, const_cast<int *>(ctx->V.n_in) + first + c0, c, m);
    }
}

extern "C" void *ll_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void ll_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int ll_upload_scan_async(ll_ctx *ctx, int slot, const float *xyz4, int n)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    if (n < 0 || (!xyz4 && n > 0)) { ctx->err = "bad upload arguments"; return LL_ERR_ARG; }
    if (n > ctx->p.max_points) { ctx->err = "scan larger than max_points"; return LL_ERR_CAPACITY; }
    rc = ensure_copy_stream(ctx); if (rc) return rc;
    LLView &V = ctx->V;
    if (n > 0) LL_HIP(hipMemcpyAsync(const_cast<float4 *>(V.raw) + (size_t)slot * V.NP, xyz4, (size_t)n * V.raw_stride * 4, hipMemcpyHostToDevice, ctx->copy_stream));
    ctx->n_in_host[slot] = n;
    enqueue_counts(ctx, slot, &n, 1);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_upload_scans_async(ll_ctx *ctx, int first, int count, const float *const *xyz4, const int *n)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    if (!xyz4 || !n) return LL_ERR_ARG;
    for (int i = 0; i < count; ++i) { rc = ll_upload_scan_async(ctx, first + i, xyz4[i], n[i]); if (rc) return rc; }
    return LL_OK;
}

/* a run of slots from ONE page-locked staging area (scan i at base + i * stride_bytes): two enqueues for the whole run -- a 2-D
 * copy (rows = scans, row width = the longest scan of the run) and the point counts -- instead of two per scan; on this stack
 * an asynchronous copy costs the host tens of microseconds, which bounds a per-scan feed below the PCIe rate */
extern "C" int ll_upload_scans_async_strided(ll_ctx *ctx, int first, int count, const float *base, size_t stride_bytes, const int *n)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    const size_t pt = (size_t)ctx->V.raw_stride * 4;             /* bytes per resident point: 16, or 12 (ll_params.input_stride_floats = 3) */
    if (!base || !n || (stride_bytes & (pt == 16 ? 15 : 3))) { ctx->err = "bad strided upload arguments"; return LL_ERR_ARG; }
    rc = ensure_copy_stream(ctx); if (rc) return rc;
    int nmax = 0;
    for (int i = 0; i < count; ++i) {
        if (n[i] < 0 || n[i] > ctx->p.max_points || (size_t)n[i] * pt > stride_bytes) { ctx->err = "scan larger than max_points / the stride"; return LL_ERR_CAPACITY; }
        nmax = n[i] > nmax ? n[i] : nmax;
        ctx->n_in_host[first + i] = n[i];
    }
    LLView &V = ctx->V;
    if (nmax > 0)
        LL_HIP(hipMemcpy2DAsync(const_cast<float4 *>(V.raw) + (size_t)first * V.NP, (size_t)V.NP * 16, base, stride_bytes, (size_t)nmax * pt, (size_t)count,
                                hipMemcpyHostToDevice, ctx->copy_stream));
    enqueue_counts(ctx, first, n, count);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

static int stream_event(ll_ctx *ctx, int stream, int ev, hipStream_t *st)
{
    int rc = ll_enter(ctx); if (rc) return rc;
    if ((stream != 0 && stream != 1) || ev < 0 || ev >= 8) { ctx->err = "bad stream / event id"; return LL_ERR_ARG; }
    rc = ensure_copy_stream(ctx); if (rc) return rc;
    *st = stream ? ctx->copy_stream : ctx->stream;
    return LL_OK;
}

extern "C" int ll_stream_record(ll_ctx *ctx, int stream, int ev)
{
    hipStream_t st; int rc = stream_event(ctx, stream, ev, &st); if (rc) return rc;
    if (!ctx->ev_x[ev]) LL_HIP(hipEventCreateWithFlags(&ctx->ev_x[ev], hipEventDisableTiming));
    LL_HIP(hipEventRecord(ctx->ev_x[ev], st));
    return LL_OK;
}

extern "C" int ll_stream_wait(ll_ctx *ctx, int stream, int ev)
{
    hipStream_t st; int rc = stream_event(ctx, stream, ev, &st); if (rc) return rc;
    if (ctx->ev_x[ev]) LL_HIP(hipStreamWaitEvent(st, ctx->ev_x[ev], 0));
    return LL_OK;
}

extern "C" int ll_synchronize_copy(ll_ctx *ctx)
{
    int rc = ll_enter(ctx); if (rc) return rc;
    if (ctx->copy_stream) LL_HIP(hipStreamSynchronize(ctx->copy_stream));
    return LL_OK;
}

/* ------------------------------------------------------------------ stages */
extern "C" int ll_extract_batch(ll_ctx *ctx, int first, int count)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    ll_launch_organize(ctx->V, first, count, ctx->stream, &ctx->prof);
    ll_launch_features(ctx->V, first, count, ctx->feat_lds, ctx->stream, &ctx->prof);
    ll_launch_build_grid(ctx->V, first, count, 0, ctx->stream, &ctx->prof);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

static int upload_poses(ll_ctx *ctx, int first, int count, const double *host_pose)
{
    if (host_pose) {
        LL_HIP(hipMemcpyAsync(ctx->V.pose + (size_t)first * 7, host_pose, (size_t)count * 7 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        LL_HIP(hipMemcpyAsync(ctx->V.pose_guess + (size_t)first * 7, host_pose, (size_t)count * 7 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        LL_HIP(hipStreamSynchronize(ctx->stream));    /* host buffer may be reused by the caller */
    }
    return LL_OK;
}

extern "C" int ll_set_pose_guess(ll_ctx *ctx, int first, int count, const double *host_pose)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    if (!host_pose) return LL_ERR_ARG;
    return upload_poses(ctx, first, count, host_pose);
}

extern "C" int ll_set_target(ll_ctx *ctx, const ll_point *corner, int m_c, const ll_point *surf, int m_s)
{
    int rc0 = ll_enter(ctx); if (rc0) return rc0;
    LLView &V = ctx->V;
    if (m_c < 0 || m_s < 0 || m_c > V.cap_lsharp || m_s > V.NP) { ctx->err = "target larger than capacity"; return LL_ERR_CAPACITY; }
    if (m_c) LL_HIP(hipMemcpyAsync(V.carry_corner, corner, (size_t)m_c * 16, hipMemcpyHostToDevice, ctx->stream));
    if (m_s) LL_HIP(hipMemcpyAsync(V.carry_surf, surf, (size_t)m_s * 16, hipMemcpyHostToDevice, ctx->stream));
    const int cnt[2] = {m_c, m_s};
    LL_HIP(hipMemcpyAsync(V.carry_cnt, cnt, sizeof(cnt), hipMemcpyHostToDevice, ctx->stream));
    ll_launch_build_grid(V, 0, 1, 1, ctx->stream, nullptr);
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

extern "C" int ll_upload_features(ll_ctx *ctx, int slot, const ll_point *sharp, int ns, const ll_point *lsharp, int nls,
                                  const ll_point *flat, int nf, const ll_point *lflat, int nlf)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    LLView &V = ctx->V;
    if (ns < 0 || nls < 0 || nf < 0 || nlf < 0 || (ns && !sharp) || (nls && !lsharp) || (nf && !flat) || (nlf && !lflat)) return LL_ERR_ARG;
    if (ns > V.cap_sharp || nls > V.cap_lsharp || nf > V.cap_flat || nlf > V.NP) { ctx->err = "feature cloud larger than capacity"; return LL_ERR_CAPACITY; }
    if (ns) LL_HIP(hipMemcpyAsync(V.sharp + (size_t)slot * V.cap_sharp, sharp, (size_t)ns * 16, hipMemcpyHostToDevice, ctx->stream));
    if (nls) LL_HIP(hipMemcpyAsync(V.lsharp + (size_t)slot * V.cap_lsharp, lsharp, (size_t)nls * 16, hipMemcpyHostToDevice, ctx->stream));
    if (nf) LL_HIP(hipMemcpyAsync(V.flat + (size_t)slot * V.cap_flat, flat, (size_t)nf * 16, hipMemcpyHostToDevice, ctx->stream));
    if (nlf) LL_HIP(hipMemcpyAsync(V.lflat + (size_t)slot * V.LFS, lflat, (size_t)nlf * 16, hipMemcpyHostToDevice, ctx->stream));
    ScanHdr h;
    std::memset(&h, 0, sizeof(h));
    h.first_kept = 0; h.last_kept = -1; h.half_idx = 0;
    h.n_sharp = ns; h.n_less_sharp = nls; h.n_flat = nf; h.n_less_flat = nlf;
    h.status = 0; h.lf_strided = 0;                         /* the caller's cloud as it is: contiguous, place = index */
    LL_HIP(hipMemcpyAsync(V.hdr + slot, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    /* the slot may become the target of slot + 1 (ll_associate_batch / ll_odometry_frames over a range): its search grid and
     * ring tables are part of "serving like an extracted slot" */
    ll_launch_build_grid(V, slot, 1, 0, ctx->stream, nullptr);
    LL_HIP(hipGetLastError());
    LL_HIP(hipStreamSynchronize(ctx->stream));              /* the host arrays and h may go away */
    if ((size_t)slot < ctx->n_in_host.size()) ctx->n_in_host[slot] = 0;
    return LL_OK;
}

/* the slot's less-sharp / less-flat clouds become the carry target: contiguous copies (the less-flat rows of an extracted slot are
 * closed up through lf_pre, which k_build_grid wrote when the slot was extracted) */
__global__ void k_copy_carry(LLView V, int slot)
{
    const ScanHdr h = V.hdr[slot];
    const int mc = h.status == 0 ? h.n_less_sharp : 0, ms = h.status == 0 ? h.n_less_flat : 0;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gtid; i < mc; i += gsz) V.carry_corner[i] = V.lsharp[(size_t)slot * V.cap_lsharp + i];
    const float4 *lf = V.lflat + (size_t)slot * V.LFS;
    if (!h.lf_strided) { for (int i = gtid; i < ms; i += gsz) V.carry_surf[i] = lf[i]; }
    else if (ms > 0) {                                                   /* (a refused scan: no target, and its prefix table is not this scan's) */
        const int *pre = V.lf_pre + (size_t)slot * (V.R + 1);
        const int wave = gtid >> 6, nwaves = gsz >> 6, lane = gtid & 63;
        for (int r = wave; r < V.R; r += nwaves) {                       /* a wave per ring row */
            const int o = pre[r], n = pre[r + 1] - o;
            for (int i = lane; i < n; i += 64) V.carry_surf[o + i] = lf[(size_t)r * V.ring_cap + i];
        }
    }
    if (gtid == 0) { V.carry_cnt[0] = mc; V.carry_cnt[1] = ms; }
}

extern "C" int ll_set_target_from_slot(ll_ctx *ctx, int slot)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    hipLaunchKernelGGL(k_copy_carry, dim3(64), dim3(256), 0, ctx->stream, ctx->V, slot);
    ll_launch_build_grid(ctx->V, 0, 1, 1, ctx->stream, nullptr);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_associate_batch(ll_ctx *ctx, int first, int count, const double *host_pose_guess)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    rc = upload_poses(ctx, first, count, host_pose_guess); if (rc) return rc;
    ctx->V.carry_slot = first;
    ll_launch_associate(ctx->V, first, count, ctx->stream, &ctx->prof);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_vote_batch(ll_ctx *ctx, int first, int count, int enable)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    ctx->V.carry_slot = first;
    ll_launch_vote(ctx->V, first, count, enable, ctx->stream, &ctx->prof);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

/* graph_based_correspondence_vote_simple on caller-supplied correspondences (laserOdometry.cpp:165-172) */
extern "C" int ll_vote_host(ll_ctx *ctx, const ll_point *src, const ll_point *tgt, int n, int corner_case,
                            int *count, uint8_t *selected, float *weight)
{
    if (!ctx || n < 0 || (n > 0 && (!src || !tgt))) return LL_ERR_ARG;
    if (n == 0) return LL_OK;
    LL_HIP(hipSetDevice(ctx->device));
    if ((size_t)n * 28 + 16 > 160 * 1024) { ctx->err = "ll_vote_host: more than 5850 correspondences do not fit one workgroup's LDS"; return LL_ERR_CAPACITY; }
    /* scratch: src, tgt (float4), count (int), weight (float), selected (u8) */
    const size_t bytes = (size_t)n * (16 + 16 + 4 + 4 + 1) + 64;
    void *d = nullptr;
    LL_HIP(hipMallocAsync(&d, bytes, ctx->stream));
    float4 *dsrc = (float4 *)d, *dtgt = dsrc + n;
    int *dcnt = (int *)(dtgt + n); float *dw = (float *)(dcnt + n); uint8_t *dsel = (uint8_t *)(dw + n);
    int rc = LL_OK;
    do {
        if (hipMemcpyAsync(dsrc, src, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(dtgt, tgt, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = LL_ERR_HIP; break; }
        ll_launch_vote_points(dsrc, dtgt, n, corner_case ? 5 : 10, dcnt, dsel, dw, ctx->stream);   /* :179-188 */
        if (hipGetLastError() != hipSuccess) { rc = LL_ERR_HIP; break; }
        if (count && hipMemcpyAsync(count, dcnt, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = LL_ERR_HIP; break; }
        if (weight && hipMemcpyAsync(weight, dw, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = LL_ERR_HIP; break; }
        if (selected && hipMemcpyAsync(selected, dsel, (size_t)n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = LL_ERR_HIP; break; }
    } while (0);
    (void)hipFreeAsync(d, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = LL_ERR_HIP;
    if (rc != LL_OK) ctx->err = "ll_vote_host: HIP runtime call failed";
    return rc;
}

extern "C" int ll_normal_equations_batch(ll_ctx *ctx, int first, int count, const double *host_pose)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    rc = upload_poses(ctx, first, count, host_pose); if (rc) return rc;
    ctx->V.carry_slot = first;
    ll_launch_normal_equations(ctx->V, first, count, 0, ctx->stream, &ctx->prof);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_gn_step_batch(ll_ctx *ctx, int first, int count)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    ll_launch_gn_step(ctx->V, first, count, ctx->stream, &ctx->prof);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" void ll_lm_default_options(ll_lm_options *o)
{
    o->max_num_iterations = 4;                       /* laserOdometry.cpp:822 */
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->jacobi_scaling = 1;
}

LLLmOpt ll_to_dev_opt(const ll_lm_options *opt)
{
    ll_lm_options d; ll_lm_default_options(&d);
    if (opt) d = *opt;
    LLLmOpt o;
    o.max_num_iterations = d.max_num_iterations; o.initial_radius = d.initial_radius; o.max_radius = d.max_radius;
    o.min_radius = d.min_radius; o.min_relative_decrease = d.min_relative_decrease; o.min_lm_diagonal = d.min_lm_diagonal;
    o.max_lm_diagonal = d.max_lm_diagonal; o.function_tolerance = d.function_tolerance; o.gradient_tolerance = d.gradient_tolerance;
    o.parameter_tolerance = d.parameter_tolerance; o.jacobi_scaling = d.jacobi_scaling;
    o.nan_poisons_pose = 0;
    return o;
}

/* enqueue one LM solve for a slot range (carry_slot must already be set by the caller) */
static void enqueue_lm(ll_ctx *ctx, int first, int count, const LLLmOpt &o)
{
    ll_launch_lm_solve(ctx->V, first, count, o, ctx->stream);      /* evaluate, begin, n x (propose, evaluate, accept) in one launch */
}

extern "C" int ll_lm_solve_batch(ll_ctx *ctx, int first, int count, const ll_lm_options *opt)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    const LLLmOpt o = ll_to_dev_opt(opt);
    if (o.max_num_iterations < 0 || o.max_num_iterations > 64) { ctx->err = "max_num_iterations out of range"; return LL_ERR_ARG; }
    ctx->V.carry_slot = first;
    enqueue_lm(ctx, first, count, o);
    LL_HIP(hipGetLastError());
    return LL_OK;
}

struct LLPose7 { double v[7]; };
__global__ void k_set_pose7(double *dst, LLPose7 p) { if (threadIdx.x < 7) dst[threadIdx.x] = p.v[threadIdx.x]; }

extern "C" int ll_odometry_frames(ll_ctx *ctx, int first, int count, const double *host_pose0, int n_outer, int first_frame_index,
                                  const ll_lm_options *opt, double *host_poses_out)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    if (n_outer < 1 || n_outer > 16) { ctx->err = "n_outer out of range"; return LL_ERR_ARG; }
    const LLLmOpt o = ll_to_dev_opt(opt);
    LLPose7 p0 = {{0, 0, 0, 1, 0, 0, 0}};
    if (host_pose0) std::memcpy(p0.v, host_pose0, sizeof(p0.v));
    hipLaunchKernelGGL(k_set_pose7, dim3(1), dim3(64), 0, ctx->stream, ctx->V.pose + (size_t)first * 7, p0);   /* by value: no copy to wait for */
    ctx->V.carry_slot = first;
    for (int k = first; k < first + count; ++k) {
        if (k > first)      /* para_q / para_t persist from the previous frame (:61-65) */
            LL_HIP(hipMemcpyAsync(ctx->V.pose + (size_t)k * 7, ctx->V.pose + (size_t)(k - 1) * 7, 7 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        const int vote = (first_frame_index + (k - first)) > 5;          /* now_frame > 5 (:794) */
        for (int outer = 0; outer < n_outer; ++outer) {                   /* :439 */
            ll_launch_associate(ctx->V, k, 1, ctx->stream, &ctx->prof);
            ll_launch_vote(ctx->V, k, 1, vote, ctx->stream, &ctx->prof);
            enqueue_lm(ctx, k, 1, o);
        }
    }
    LL_HIP(hipGetLastError());
    if (host_poses_out) {
        if (ll_read_back(host_poses_out, ctx->V.pose + (size_t)first * 7, (size_t)count * 7 * sizeof(double), ctx->stream)) { ctx->err = "read-back failed"; return LL_ERR_HIP; }
    }
    return LL_OK;
}

static int hot_path(ll_ctx *ctx, int first, int count, const double *host_pose_guess, int vote_enable, int chain);

/* The association stage of a hot-path call on TWO streams -- built in round 6, measured, and OFF by default (ll_set_two_stream(ctx, 1) or
 * LIGHTLOAM_TWO_STREAM=1 turns it on).  Round 5's probe had k_build_grid and k_associate finish in 89 % of the sum of their times when both
 * were simply launched side by side, so the slot range is cut into pieces here: the grid of piece i + 1 is built on the context's stream
 * while piece i is searched on a second one -- piece i's targets are its own slots' predecessors, i.e. the grids of pieces <= i, complete
 * when event i fires.  Events only, no host synchronisation; the context's stream waits for the last search before the vote; results do not
 * depend on the schedule (tests/test_gpu_variants.py).  Measured on one box with 2 / 3 / 4 / 6 / 8 pieces of a 16384-scan step
 * (profiles/r06_experiments/two_stream_pieces.log): the stage takes 20.3-21.3 ms on two streams against 20.3-20.7 ms kernel after kernel --
 * never less.  Both kernels ask for a whole CU (k_build_grid: one 1024-thread workgroup with 72 KB of LDS; k_associate: eight workgroups that
 * take all 160 KB), so the dispatcher alternates between them CU by CU instead of interleaving them, and the dependent pieces add their
 * tails.  With the per-kernel profiler on, the two-stream stage is ONE interval ("k_build_grid||k_associate"). */
#define LL_TWO_STREAM_MIN 512        /* scans per hot-path call (chunk) below which the stage stays on one stream: the pieces must still fill the chip */
static int association_two_streams(ll_ctx *ctx, int f, int n)
{
    if (!ctx->stream2) LL_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    const int P = ctx->ts_pieces;
    for (int i = 0; i <= P; ++i)
        if (!ctx->ev_ts[i]) LL_HIP(hipEventCreateWithFlags(&ctx->ev_ts[i], hipEventDisableTiming));
    hipStream_t s1 = ctx->stream, s2 = ctx->stream2;
    /* pieces of whole groups of eight scans (the kernels deal eight scans to the eight XCDs) */
    int lo[LL_TWO_STREAM_MAX_PIECES + 1];
    for (int i = 0; i <= P; ++i) { lo[i] = (int)((long long)n * i / P) & ~7; }
    lo[0] = 0; lo[P] = n;
    ll_prof_mark(&ctx->prof, LL_K_ASSOC_STAGE, s1);
    ll_launch_build_grid(ctx->V, f + lo[0], lo[1] - lo[0], 0, s1, nullptr);
    LL_HIP(hipEventRecord(ctx->ev_ts[0], s1));
    for (int i = 0; i < P; ++i) {
        if (i + 1 < P && lo[i + 2] > lo[i + 1]) {
            ll_launch_build_grid(ctx->V, f + lo[i + 1], lo[i + 2] - lo[i + 1], 0, s1, nullptr);
            LL_HIP(hipEventRecord(ctx->ev_ts[i + 1], s1));
        }
        if (lo[i + 1] > lo[i]) {
            LL_HIP(hipStreamWaitEvent(s2, ctx->ev_ts[i], 0));
            ll_launch_associate(ctx->V, f + lo[i], lo[i + 1] - lo[i], s2, nullptr);
        }
    }
    LL_HIP(hipEventRecord(ctx->ev_ts[P], s2));
    LL_HIP(hipStreamWaitEvent(s1, ctx->ev_ts[P], 0));
    ll_prof_mark(&ctx->prof, LL_K_END, s1);
    return LL_OK;
}

extern "C" int ll_hot_path_batch(ll_ctx *ctx, int first, int count, const double *host_pose_guess, int vote_enable)
{
    return hot_path(ctx, first, count, host_pose_guess, vote_enable, 0);
}
/* the same pass, continuing a batch: the target of slot `first` is slot first - 1 (extracted by an earlier call), not the carry */
extern "C" int ll_hot_path_chain(ll_ctx *ctx, int first, int count, int vote_enable)
{
    if (ctx && first < 1) { ctx->err = "ll_hot_path_chain needs a previous slot"; return LL_ERR_ARG; }
    return hot_path(ctx, first, count, nullptr, vote_enable, 1);
}
static int hot_path(ll_ctx *ctx, int first, int count, const double *host_pose_guess, int vote_enable, int chain)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    rc = upload_poses(ctx, first, count, host_pose_guess); if (rc) return rc;
    /* NULL guess: restart from the stored guess (device-to-device, no host synchronisation) */
    if (!host_pose_guess)
        LL_HIP(hipMemcpyAsync(ctx->V.pose + (size_t)first * 7, ctx->V.pose_guess + (size_t)first * 7, (size_t)count * 7 * sizeof(double),
                              hipMemcpyDeviceToDevice, ctx->stream));
    if (!chain) ctx->V.carry_slot = first;                      /* chain: the carry belongs to the slot that opened the batch */
    /* chunks keep a chunk's intermediates (ori/ring, laserCloud, feature slots) inside the Infinity Cache between
     * the producing and the consuming kernel; the target of a chunk's first slot is the previous chunk's last slot */
    const int chunk = (ctx->p.chunk > 0) ? ctx->p.chunk : count;
    for (int c0 = 0; c0 < count; c0 += chunk) {
        const int f = first + c0, n = (count - c0 < chunk) ? count - c0 : chunk;
        ll_launch_organize(ctx->V, f, n, ctx->stream, &ctx->prof);
        ll_launch_features(ctx->V, f, n, ctx->feat_lds, ctx->stream, &ctx->prof);
        if (!ctx->two_stream || n < LL_TWO_STREAM_MIN) {
            ll_launch_build_grid(ctx->V, f, n, 0, ctx->stream, &ctx->prof);
            ll_launch_associate(ctx->V, f, n, ctx->stream, &ctx->prof);
        } else {
            rc = association_two_streams(ctx, f, n); if (rc) return rc;
        }
        ll_launch_vote(ctx->V, f, n, vote_enable, ctx->stream, &ctx->prof);
        ll_launch_normal_equations(ctx->V, f, n, 1, ctx->stream, &ctx->prof);
    }
    LL_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_set_two_stream(ll_ctx *ctx, int on)
{
    if (!ctx) return LL_ERR_ARG;
    ctx->two_stream = on ? 1 : 0;
    return LL_OK;
}

/* ------------------------------------------------------------------ lidarFactor.hpp functors on caller-supplied blocks (ll_functors.hip) */
extern "C" int ll_factor_blocks_set(ll_ctx *ctx, int n_edge, const double *edge9, int n_plane, const double *plane13, int n_pnorm, const double *pnorm7)
{
    if (!ctx) return LL_ERR_ARG;
    if (n_edge < 0 || n_plane < 0 || n_pnorm < 0 || (n_edge > 0 && !edge9) || (n_plane > 0 && !plane13) || (n_pnorm > 0 && !pnorm7) ||
        (long long)n_edge + n_plane + n_pnorm > (1 << 24)) { ctx->err = "bad residual blocks"; return LL_ERR_ARG; }
    LL_HIP(hipSetDevice(ctx->device));
    const size_t need = (size_t)n_edge * 9 + (size_t)n_plane * 13 + (size_t)n_pnorm * 7;
    if (need > ctx->fb_cap) {
        double *p = nullptr;
        if (!dev_alloc(ctx, p, need + need / 2 + 64, false)) return LL_ERR_HIP;      /* the smaller buffer stays in allocs until destroy */
        ctx->d_fb = p; ctx->fb_cap = need + need / 2 + 64;
    }
    hipStream_t st = ctx->stream;
    double *d = ctx->d_fb;
    if (n_edge) LL_HIP(hipMemcpyAsync(d, edge9, (size_t)n_edge * 9 * sizeof(double), hipMemcpyHostToDevice, st));
    d += (size_t)n_edge * 9;
    if (n_plane) LL_HIP(hipMemcpyAsync(d, plane13, (size_t)n_plane * 13 * sizeof(double), hipMemcpyHostToDevice, st));
    d += (size_t)n_plane * 13;
    if (n_pnorm) LL_HIP(hipMemcpyAsync(d, pnorm7, (size_t)n_pnorm * 7 * sizeof(double), hipMemcpyHostToDevice, st));
    LL_HIP(hipStreamSynchronize(st));
    ctx->fb_n[0] = n_edge; ctx->fb_n[1] = n_plane; ctx->fb_n[2] = n_pnorm;
    ctx->fb_has_s = false;
    return LL_OK;
}

extern "C" int ll_factor_blocks_set_s(ll_ctx *ctx, const double *edge_s, const double *plane_s)
{
    if (!ctx) return LL_ERR_ARG;
    const int ne = ctx->fb_n[0], np = ctx->fb_n[1];
    if (!edge_s && !plane_s) { ctx->fb_has_s = false; return LL_OK; }
    LL_HIP(hipSetDevice(ctx->device));
    const size_t need = (size_t)ne + np;
    if (need > ctx->fb_s_cap) {
        double *p = nullptr;
        if (!dev_alloc(ctx, p, need + need / 2 + 64, false)) return LL_ERR_HIP;
        ctx->d_fb_s = p; ctx->fb_s_cap = need + need / 2 + 64;
    }
    std::vector<double> h(need, 1.0);
    if (edge_s) std::copy(edge_s, edge_s + ne, h.begin());
    if (plane_s) std::copy(plane_s, plane_s + np, h.begin() + ne);
    if (need) {
        LL_HIP(hipMemcpyAsync(ctx->d_fb_s, h.data(), need * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        LL_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx->fb_has_s = true;
    return LL_OK;
}

extern "C" int ll_factor_blocks_evaluate(ll_ctx *ctx, const double *q4, const double *t3, double *r, double *Jq, double *Jt, int cap_rows)
{
    if (!ctx || !q4 || !t3) return LL_ERR_ARG;
    const int ne = ctx->fb_n[0], np = ctx->fb_n[1], nn = ctx->fb_n[2];
    const size_t rows = (size_t)3 * ne + np + nn;
    if ((size_t)cap_rows < rows) { ctx->err = "row capacity too small"; return LL_ERR_CAPACITY; }
    if (rows == 0) return LL_OK;
    LL_HIP(hipSetDevice(ctx->device));
    if (rows * 8 > ctx->fb_out_cap) {
        double *p = nullptr;
        if (!dev_alloc(ctx, p, rows * 8 + rows * 4 + 64, false)) return LL_ERR_HIP;
        ctx->d_fb_out = p; ctx->fb_out_cap = rows * 8 + rows * 4 + 64;
    }
    hipStream_t st = ctx->stream;
    const double pose[7] = {q4[0], q4[1], q4[2], q4[3], t3[0], t3[1], t3[2]};
    LL_HIP(hipMemcpyAsync(ctx->d_tmp_pose, pose, sizeof(pose), hipMemcpyHostToDevice, st));
    double *dr = ctx->d_fb_out, *dJq = dr + rows, *dJt = dJq + rows * 4;
    const double *edge = ctx->d_fb, *plane = edge + (size_t)ne * 9, *pnorm = plane + (size_t)np * 13;
    ll_launch_factor_blocks(ctx->d_tmp_pose, ne, edge, np, plane, nn, pnorm, ctx->fb_has_s ? ctx->d_fb_s : nullptr, dr, dJq, dJt, st);
    LL_HIP(hipGetLastError());
    if (r) LL_HIP(hipMemcpyAsync(r, dr, rows * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Jq) LL_HIP(hipMemcpyAsync(Jq, dJq, rows * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Jt) LL_HIP(hipMemcpyAsync(Jt, dJt, rows * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    LL_HIP(hipStreamSynchronize(st));
    return LL_OK;
}

/* ------------------------------------------------------------------ mapping stage (ll_mapping.hip) */
extern "C" void ll_map_destroy(ll_map *m)
{
    if (!m) return;
    if (m->ctx) { (void)hipSetDevice(m->ctx->device); (void)hipStreamSynchronize(m->ctx->stream); }
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
}

extern "C" const char *ll_map_last_error(const ll_map *m) { return m ? m->err.c_str() : "null map"; }

extern "C" int ll_map_create(ll_ctx *ctx, int max_map_corner, int max_map_surf, int max_scan_corner, int max_scan_surf, ll_map **out)
{
    if (!ctx || !out) return LL_ERR_ARG;
    *out = nullptr;
    if (max_map_corner < 1 || max_map_surf < 1 || max_scan_corner < 1 || max_scan_surf < 1 ||
        max_map_corner > (1 << 24) || max_map_surf > (1 << 24)) { ctx->err = "bad mapping capacities"; return LL_ERR_ARG; }
    LL_HIP(hipSetDevice(ctx->device));
    ll_map *m = new ll_map();
    m->ctx = ctx;
    m->cap_map[0] = max_map_corner; m->cap_map[1] = max_map_surf; m->cap_stk[0] = max_scan_corner; m->cap_stk[1] = max_scan_surf;
    m->max_cells = 1 << 23;                                     /* 32 MB of cell starts per cloud at most */
    LLMapView &M = m->M;
    std::memset(&M, 0, sizeof(M));
    bool ok = true;
    for (int w = 0; w < 2 && ok; ++w) {
        ok = ok && map_alloc(m, m->d_map[w], (size_t)m->cap_map[w]) && map_alloc(m, m->d_stk[w], (size_t)m->cap_stk[w]);
        ok = ok && map_alloc(m, M.grid[w].start, (size_t)m->max_cells + 1) && map_alloc(m, M.grid[w].cursor, (size_t)m->max_cells) &&
             map_alloc(m, M.grid[w].pts, (size_t)m->cap_map[w]);
        ok = ok && map_alloc(m, M.ok[w], (size_t)m->cap_stk[w]) && map_alloc(m, M.src[w], (size_t)m->cap_stk[w]);
        M.map[w] = m->d_map[w]; M.stk[w] = m->d_stk[w];
    }
    ok = ok && map_alloc(m, M.qa, (size_t)m->cap_stk[0] * 3) && map_alloc(m, M.qb, (size_t)m->cap_stk[0] * 3) &&
         map_alloc(m, M.fa, (size_t)m->cap_stk[0] * 3) && map_alloc(m, M.fb, (size_t)m->cap_stk[0] * 3);
    ok = ok && map_alloc(m, M.qn, (size_t)m->cap_stk[1] * 3) && map_alloc(m, M.qd, (size_t)m->cap_stk[1]) &&
         map_alloc(m, M.fn, (size_t)m->cap_stk[1] * 3) && map_alloc(m, M.fd, (size_t)m->cap_stk[1]);
    for (int w = 0; w < 2 && ok; ++w)
        ok = ok && map_alloc(m, M.nn_pt[w], (size_t)m->cap_stk[w] * 5) && map_alloc(m, M.nn_id[w], (size_t)m->cap_stk[w] * 5);
    ok = ok && map_alloc(m, M.counts, 2) && map_alloc(m, M.pose, 7) && map_alloc(m, M.neq, LL_NEQ_STRIDE) && map_alloc(m, M.lm, LL_LM_STRIDE);
    ok = ok && map_alloc(m, M.neq_part, (size_t)LL_NEQ_NB * 28) && map_alloc(m, M.neq_ticket, 1) && map_alloc(m, M.lm_go, 1);
    for (int w = 0; w < 2; ++w) ok = ok && map_alloc(m, M.cpub[w], (size_t)m->cap_stk[w] / 256 + 2);
    ok = ok && map_alloc(m, m->d_bbox, 12) && map_alloc(m, m->d_tile, (size_t)(m->max_cells + 1 + 4095) / 4096 + 1);
    if (!ok) { ctx->err = m->err; ll_map_destroy(m); return LL_ERR_HIP; }
    M.huber = ctx->V.huber;
    const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
    if (hipMemcpy(M.pose, ident, sizeof(ident), hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "initial upload failed"; ll_map_destroy(m); return LL_ERR_HIP; }
    *out = m;
    return LL_OK;
}

static int map_upload(ll_map *m, float4 *dst, const ll_point *src, int n)
{
    if (n > 0) LLM_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(ll_point), hipMemcpyHostToDevice, m->ctx->stream));
    return LL_OK;
}

/* search grids over the clouds that already sit in d_map[0 / 1] (uploaded by ll_map_set_map, gathered by the cube map) */
/* the search grids over the clouds already in d_map[], in two halves so that a caller with more to read back can share the
 * one synchronisation: _begin enqueues the bounding boxes (m->d_bbox, 12 ints), _finish takes them on the host (the grid
 * dimensions are launch parameters) and enqueues the builds */
void ll_map_rebuild_begin(ll_map *m, int n_corner, int n_surf)
{
    const int n[2] = {n_corner, n_surf};
    for (int w = 0; w < 2; ++w) {
        m->M.n_map[w] = n[w];
        ll_map_launch_bbox(m->d_map[w], n[w], m->d_bbox + 6 * w, m->ctx->stream);
    }
}
void ll_map_rebuild_finish(ll_map *m, const int bbox_host[12])
{
    for (int w = 0; w < 2; ++w) {
        ll_map_bbox_to_grid(bbox_host + 6 * w, m->M.n_map[w], m->max_cells, &m->M.grid[w]);
        ll_map_launch_build(m->M.grid[w], m->d_map[w], m->M.n_map[w], m->d_tile, m->ctx->stream);
    }
}
int ll_map_rebuild(ll_map *m, int n_corner, int n_surf)
{
    hipStream_t st = m->ctx->stream;
    int bbox[12];
    ll_map_rebuild_begin(m, n_corner, n_surf);
    if (ll_read_back(bbox, m->d_bbox, sizeof(bbox), st)) { m->err = "read-back failed"; return LL_ERR_HIP; }   /* the grid dimensions are launch parameters */
    ll_map_rebuild_finish(m, bbox);
    LLM_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_map_set_map(ll_map *m, const ll_point *corner, int n_corner, const ll_point *surf, int n_surf)
{
    if (!m) return LL_ERR_ARG;
    if (n_corner < 0 || n_surf < 0 || (!corner && n_corner > 0) || (!surf && n_surf > 0)) { m->err = "bad map clouds"; return LL_ERR_ARG; }
    if (n_corner > m->cap_map[0] || n_surf > m->cap_map[1]) { m->err = "map cloud larger than the capacity given to ll_map_create"; return LL_ERR_CAPACITY; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    int rc = map_upload(m, m->d_map[0], corner, n_corner); if (rc) return rc;
    rc = map_upload(m, m->d_map[1], surf, n_surf); if (rc) return rc;
    rc = ll_map_rebuild(m, n_corner, n_surf); if (rc) return rc;
    LLM_HIP(hipStreamSynchronize(m->ctx->stream));              /* host buffers may be reused by the caller */
    return LL_OK;
}

extern "C" int ll_map_set_scan(ll_map *m, const ll_point *corner, int n_corner, const ll_point *surf, int n_surf)
{
    if (!m) return LL_ERR_ARG;
    if (n_corner < 0 || n_surf < 0 || (!corner && n_corner > 0) || (!surf && n_surf > 0)) { m->err = "bad scan clouds"; return LL_ERR_ARG; }
    if (n_corner > m->cap_stk[0] || n_surf > m->cap_stk[1]) { m->err = "scan cloud larger than the capacity given to ll_map_create"; return LL_ERR_CAPACITY; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    int rc = map_upload(m, m->d_stk[0], corner, n_corner); if (rc) return rc;
    rc = map_upload(m, m->d_stk[1], surf, n_surf); if (rc) return rc;
    m->M.n_stk[0] = n_corner; m->M.n_stk[1] = n_surf;
    LLM_HIP(hipStreamSynchronize(m->ctx->stream));
    return LL_OK;
}

static int map_set_pose(ll_map *m, const double *pose_w7)
{
    if (pose_w7) {
        LLM_HIP(hipMemcpyAsync(m->M.pose, pose_w7, 7 * sizeof(double), hipMemcpyHostToDevice, m->ctx->stream));
        LLM_HIP(hipStreamSynchronize(m->ctx->stream));
    }
    return LL_OK;
}

extern "C" int ll_map_associate(ll_map *m, const double *pose_w7)
{
    if (!m) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    int rc = map_set_pose(m, pose_w7); if (rc) return rc;
    ll_map_launch_associate(m->M, m->ctx->stream);
    LLM_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_map_get_counts(ll_map *m, int *n_edge, int *n_plane)
{
    if (!m) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    int c[2];
    if (ll_read_back(c, m->M.counts, sizeof(c), m->ctx->stream)) { m->err = "read-back failed"; return LL_ERR_HIP; }
    if (n_edge) *n_edge = c[0];
    if (n_plane) *n_plane = c[1];
    return LL_OK;
}

/* the sizes of laserCloudCornerFromMap / laserCloudSurfFromMap as the map holds them: what laserMapping.cpp:1822 tests before it
 * optimises at all (host state: no device round trip) */
extern "C" int ll_map_get_map_sizes(ll_map *m, int *n_corner_from_map, int *n_surf_from_map)
{
    if (!m) return LL_ERR_ARG;
    if (n_corner_from_map) *n_corner_from_map = m->M.n_map[0];
    if (n_surf_from_map) *n_surf_from_map = m->M.n_map[1];
    return LL_OK;
}

extern "C" int ll_map_download_edges(ll_map *m, int *src, double *a3, double *b3, int cap)
{
    int ne = 0; int rc = ll_map_get_counts(m, &ne, nullptr); if (rc) return rc;
    if (cap < ne) { m->err = "edge capacity too small"; return LL_ERR_CAPACITY; }
    hipStream_t st = m->ctx->stream;
    if (ne > 0) {
        if (src) LLM_HIP(hipMemcpyAsync(src, m->M.src[0], (size_t)ne * sizeof(int), hipMemcpyDeviceToHost, st));
        if (a3) LLM_HIP(hipMemcpyAsync(a3, m->M.fa, (size_t)ne * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        if (b3) LLM_HIP(hipMemcpyAsync(b3, m->M.fb, (size_t)ne * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    LLM_HIP(hipStreamSynchronize(st));
    return LL_OK;
}

extern "C" int ll_map_download_planes(ll_map *m, int *src, double *norm3, double *d, int cap)
{
    int np = 0; int rc = ll_map_get_counts(m, nullptr, &np); if (rc) return rc;
    if (cap < np) { m->err = "plane capacity too small"; return LL_ERR_CAPACITY; }
    hipStream_t st = m->ctx->stream;
    if (np > 0) {
        if (src) LLM_HIP(hipMemcpyAsync(src, m->M.src[1], (size_t)np * sizeof(int), hipMemcpyDeviceToHost, st));
        if (norm3) LLM_HIP(hipMemcpyAsync(norm3, m->M.fn, (size_t)np * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        if (d) LLM_HIP(hipMemcpyAsync(d, m->M.fd, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    LLM_HIP(hipStreamSynchronize(st));
    return LL_OK;
}

extern "C" int ll_map_normal_equations(ll_map *m, const double *pose_w7, double *H36, double *g6, double *cost)
{
    if (!m) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    int rc = map_set_pose(m, pose_w7); if (rc) return rc;
    ll_map_launch_normal_eq(m->M, m->ctx->stream);
    double out[LL_NEQ_STRIDE];
    LLM_HIP(hipMemcpyAsync(out, m->M.neq, sizeof(out), hipMemcpyDeviceToHost, m->ctx->stream));
    LLM_HIP(hipStreamSynchronize(m->ctx->stream));
    if (H36) std::memcpy(H36, out, 36 * sizeof(double));
    if (g6) std::memcpy(g6, out + 36, 6 * sizeof(double));
    if (cost) *cost = out[42];
    return LL_OK;
}

/* ---- tile-parallel mapping: the map's cubes are spread over the ranks; every rank searches its own
 * points, the candidates are all-gathered by the caller (RCCL on device buffers, gloo on host buffers -- the copies
 * below take either), and every rank forms the same residual blocks from the merged five nearest. ---- */
int ll_map_use_ids(ll_map *m, bool on)
{
    for (int w = 0; w < 2; ++w) {
        if (on && !m->d_gid[w] && !map_alloc(m, m->d_gid[w], (size_t)m->cap_map[w])) return LL_ERR_HIP;
        m->M.gid[w] = on ? m->d_gid[w] : nullptr;
    }
    return LL_OK;
}

extern "C" int ll_map_set_map_ids(ll_map *m, const int *corner_gid, const int *surf_gid)
{
    if (!m) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    if (!corner_gid && !surf_gid) return ll_map_use_ids(m, false);
    if (!corner_gid || !surf_gid) { m->err = "give the ids of both clouds or of neither"; return LL_ERR_ARG; }
    int rc = ll_map_use_ids(m, true); if (rc) return rc;
    const int *src[2] = {corner_gid, surf_gid};
    for (int w = 0; w < 2; ++w)
        if (m->M.n_map[w] > 0) LLM_HIP(hipMemcpyAsync(m->d_gid[w], src[w], (size_t)m->M.n_map[w] * sizeof(int), hipMemcpyDefault, m->ctx->stream));
    LLM_HIP(hipStreamSynchronize(m->ctx->stream));
    return LL_OK;
}

extern "C" int ll_map_knn_partial(ll_map *m, const double *pose_w7, float *corner_nn, int *corner_id, float *surf_nn, int *surf_id)
{
    if (!m) return LL_ERR_ARG;
    if ((m->M.n_stk[0] > 0 && (!corner_nn || !corner_id)) || (m->M.n_stk[1] > 0 && (!surf_nn || !surf_id))) { m->err = "null candidate buffers"; return LL_ERR_ARG; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    int rc = map_set_pose(m, pose_w7); if (rc) return rc;
    hipStream_t st = m->ctx->stream;
    ll_map_launch_knn_partial(m->M, st);
    LLM_HIP(hipGetLastError());
    float *nn[2] = {corner_nn, surf_nn}; int *id[2] = {corner_id, surf_id};
    for (int w = 0; w < 2; ++w) {
        const size_t n = (size_t)m->M.n_stk[w] * 5;
        if (n == 0) continue;
        LLM_HIP(hipMemcpyAsync(nn[w], m->M.nn_pt[w], n * sizeof(float4), hipMemcpyDefault, st));
        LLM_HIP(hipMemcpyAsync(id[w], m->M.nn_id[w], n * sizeof(int), hipMemcpyDefault, st));
    }
    LLM_HIP(hipStreamSynchronize(st));
    return LL_OK;
}

extern "C" int ll_map_associate_merged(ll_map *m, const double *pose_w7, int n_parts, const float *corner_nn, const int *corner_id,
                                       const float *surf_nn, const int *surf_id)
{
    if (!m) return LL_ERR_ARG;
    if (n_parts < 1 || n_parts > 64) { m->err = "n_parts out of range"; return LL_ERR_ARG; }
    if ((m->M.n_stk[0] > 0 && (!corner_nn || !corner_id)) || (m->M.n_stk[1] > 0 && (!surf_nn || !surf_id))) { m->err = "null candidate buffers"; return LL_ERR_ARG; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    if (n_parts > m->cap_parts) {
        for (int w = 0; w < 2; ++w) {
            m->d_all_pt[w] = nullptr; m->d_all_id[w] = nullptr;   /* the smaller buffers stay in allocs until destroy */
            if (!map_alloc(m, m->d_all_pt[w], (size_t)n_parts * m->cap_stk[w] * 5) || !map_alloc(m, m->d_all_id[w], (size_t)n_parts * m->cap_stk[w] * 5)) return LL_ERR_HIP;
        }
        m->cap_parts = n_parts;
    }
    int rc = map_set_pose(m, pose_w7); if (rc) return rc;
    hipStream_t st = m->ctx->stream;
    const float *nn[2] = {corner_nn, surf_nn}; const int *id[2] = {corner_id, surf_id};
    for (int w = 0; w < 2; ++w) {
        const size_t n = (size_t)n_parts * m->M.n_stk[w] * 5;
        if (n == 0) continue;
        LLM_HIP(hipMemcpyAsync(m->d_all_pt[w], nn[w], n * sizeof(float4), hipMemcpyDefault, st));
        LLM_HIP(hipMemcpyAsync(m->d_all_id[w], id[w], n * sizeof(int), hipMemcpyDefault, st));
    }
    const float4 *pt_all[2] = {m->d_all_pt[0], m->d_all_pt[1]}; const int *id_all[2] = {m->d_all_id[0], m->d_all_id[1]};
    ll_map_launch_associate_merged(m->M, n_parts, pt_all, id_all, st);
    LLM_HIP(hipGetLastError());
    LLM_HIP(hipStreamSynchronize(st));                          /* the caller's buffers may be reused */
    return LL_OK;
}

/* one ceres::Solve (:2072-2082) on the residual blocks the last association left behind */
static void map_lm_solve(ll_map *m, const LLLmOpt &o)
{
    ll_map_launch_lm_solve(m->M, o, m->ctx->stream);          /* evaluate, begin, n x (propose, evaluate, accept): one launch */
}
/* A solve whose workgroups timed out on one another returns a NaN pose and leaves the round word at "everybody out"; stragglers may
 * also have bumped the arrival counter after workgroup 0 reset it.  Left alone, the NEXT solve's workgroups would pass their
 * waits at once and sum stale partials -- a wrong pose without an error.  Called where a solve's pose has just been read back
 * (the stream is drained: no straggler is left): on NaN both words are zeroed again. */
/* Returns true when the pose was NaN -- a solve that timed out, or k_map_compact's lost-predecessor path, which poisons the pose -- after
 * the repair; the callers then return LL_ERR_STATE instead of handing a NaN pose back with LL_OK (round-3 advice). */
static bool map_lm_repair(ll_map *m, const double *pose7)
{
    bool bad = false;
    for (int k = 0; k < 7; ++k) bad = bad || std::isnan(pose7[k]);
    if (!bad) return false;
    ll_fill_words((int *)m->M.lm_go, 1, 0, 0, 1, m->ctx->stream);
    ll_fill_words((int *)m->M.neq_ticket, 1, 0, 0, 1, m->ctx->stream);
    m->err = "the mapping solve returned an undefined pose (a workgroup never heard from its predecessors, or the input pose was NaN): set a pose and solve again";
    return true;
}

extern "C" int ll_map_set_row_shard(ll_map *m, int rank, int world)
{
    if (!m) return LL_ERR_ARG;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) { m->err = "bad row shard"; return LL_ERR_ARG; }
    m->M.row_rank = rank; m->M.row_world = world;
    return LL_OK;
}

extern "C" int ll_map_solve(ll_map *m, double *pose_w7, const ll_lm_options *opt)
{
    if (!m || !pose_w7) return LL_ERR_ARG;
    const LLLmOpt o = ll_to_dev_opt(opt);
    if (o.max_num_iterations < 0 || o.max_num_iterations > 64) { m->err = "max_num_iterations out of range"; return LL_ERR_ARG; }
    if (m->M.row_world > 1) { m->err = "the map sums a row shard: step with ll_map_evaluate + all-reduce + ll_map_lm_*"; return LL_ERR_STATE; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    int rc = map_set_pose(m, pose_w7); if (rc) return rc;
    map_lm_solve(m, o);
    LLM_HIP(hipGetLastError());
    if (ll_read_back(pose_w7, m->M.pose, 7 * sizeof(double), m->ctx->stream)) { m->err = "read-back failed"; return LL_ERR_HIP; }
    if (map_lm_repair(m, pose_w7)) return LL_ERR_STATE;
    return LL_OK;
}

extern "C" int ll_map_optimize(ll_map *m, double *pose_w7, int n_outer, const ll_lm_options *opt, int *ran)
{
    if (!m || !pose_w7) return LL_ERR_ARG;
    if (n_outer < 1 || n_outer > 16) { m->err = "n_outer out of range"; return LL_ERR_ARG; }
    const LLLmOpt o = ll_to_dev_opt(opt);
    if (o.max_num_iterations < 0 || o.max_num_iterations > 64) { m->err = "max_num_iterations out of range"; return LL_ERR_ARG; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    if (ran) *ran = 0;
    if (!(m->M.n_map[0] > 10 && m->M.n_map[1] > 50)) return LL_OK;           /* :1822 */
    int rc = map_set_pose(m, pose_w7); if (rc) return rc;
    hipStream_t st = m->ctx->stream;
    if (m->M.row_world > 1) { m->err = "the map sums a row shard: step with ll_map_evaluate + all-reduce + ll_map_lm_*"; return LL_ERR_STATE; }
    if (m->M.gid[0]) { m->err = "the map holds a tile shard: search with ll_map_knn_partial / ll_map_associate_merged"; return LL_ERR_STATE; }
    for (int it = 0; it < n_outer; ++it) {                                     /* :1832 */
        ll_map_launch_associate(m->M, st);
        map_lm_solve(m, o);
    }
    LLM_HIP(hipGetLastError());
    if (ll_read_back(pose_w7, m->M.pose, 7 * sizeof(double), st)) { m->err = "read-back failed"; return LL_ERR_HIP; }
    if (map_lm_repair(m, pose_w7)) return LL_ERR_STATE;
    if (ran) *ran = 1;
    return LL_OK;
}

extern "C" int ll_map_residual_jacobian(ll_map *m, const double *pose_w7, double *r, double *Jq, double *Jt, int cap_rows)
{
    if (!m) return LL_ERR_ARG;
    int ne = 0, np = 0;
    int rc = ll_map_get_counts(m, &ne, &np); if (rc) return rc;
    const size_t rows = (size_t)3 * ne + np;
    if ((size_t)cap_rows < rows) { m->err = "row capacity too small"; return LL_ERR_CAPACITY; }
    if (rows == 0) return LL_OK;
    rc = map_set_pose(m, pose_w7); if (rc) return rc;
    hipStream_t st = m->ctx->stream;
    double *d = nullptr;
    LLM_HIP(hipMalloc((void **)&d, rows * 8 * sizeof(double)));
    double *dr = d, *dJq = d + rows, *dJt = dJq + rows * 4;
    ll_map_launch_rows(m->M, dr, dJq, dJt, st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && r) e = hipMemcpyAsync(r, dr, rows * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && Jq) e = hipMemcpyAsync(Jq, dJq, rows * 4 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && Jt) e = hipMemcpyAsync(Jt, dJt, rows * 3 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) { m->err = std::string("ll_map_residual_jacobian: ") + hipGetErrorString(e); return LL_ERR_HIP; }
    return LL_OK;
}

extern "C" int ll_map_set_pose(ll_map *m, const double *pose_w7)
{
    if (!m || !pose_w7) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    return map_set_pose(m, pose_w7);
}

extern "C" int ll_map_get_pose(ll_map *m, double *pose_w7)
{
    if (!m || !pose_w7) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    if (ll_read_back(pose_w7, m->M.pose, 7 * sizeof(double), m->ctx->stream)) { m->err = "read-back failed"; return LL_ERR_HIP; }
    if (map_lm_repair(m, pose_w7)) return LL_ERR_STATE;          /* this is where a caller of ll_map_solve_dev sees the result */
    return LL_OK;
}

extern "C" int ll_map_evaluate(ll_map *m, double *neq44)
{
    if (!m || !neq44) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    ll_map_launch_normal_eq(m->M, m->ctx->stream);
    LLM_HIP(hipMemcpyAsync(neq44, m->M.neq, LL_NEQ_STRIDE * sizeof(double), hipMemcpyDeviceToHost, m->ctx->stream));
    LLM_HIP(hipStreamSynchronize(m->ctx->stream));
    return LL_OK;
}

static int map_lm_stage(ll_map *m, int stage, const double *neq44_sum, const ll_lm_options *opt)
{
    if (!m || (stage != 1 && !neq44_sum)) return LL_ERR_ARG;
    LLLmOpt o = ll_to_dev_opt(opt);
    o.nan_poisons_pose = 1;                                     /* a failing rank's NaN record must reach every rank's pose (lightloam_rccl.hpp) */
    if (o.max_num_iterations < 0 || o.max_num_iterations > 64) { m->err = "max_num_iterations out of range"; return LL_ERR_ARG; }   /* as ll_map_optimize */
    LLM_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    if (neq44_sum) LLM_HIP(hipMemcpyAsync(m->M.neq, neq44_sum, LL_NEQ_STRIDE * sizeof(double), hipMemcpyHostToDevice, st));
    LLView Vm = m->ctx->V;
    Vm.pose = m->M.pose; Vm.neq = m->M.neq; Vm.lm = m->M.lm;
    if (stage == 0) ll_launch_lm_begin(Vm, 0, 1, o, st);
    else if (stage == 1) ll_launch_lm_propose(Vm, 0, 1, o, st);
    else ll_launch_lm_accept(Vm, 0, 1, o, st);
    LLM_HIP(hipGetLastError());
    LLM_HIP(hipStreamSynchronize(st));                          /* neq44_sum may be reused by the caller */
    return LL_OK;
}

extern "C" int ll_map_lm_begin(ll_map *m, const double *neq44_sum, const ll_lm_options *opt) { return map_lm_stage(m, 0, neq44_sum, opt); }
extern "C" int ll_map_lm_propose(ll_map *m, const ll_lm_options *opt) { return map_lm_stage(m, 1, nullptr, opt); }
extern "C" int ll_map_lm_accept(ll_map *m, const double *neq44_sum, const ll_lm_options *opt) { return map_lm_stage(m, 2, neq44_sum, opt); }

/* ---- device-resident variants for the collectives of SURVEY 8e: every pointer is a DEVICE pointer on the context's GPU, every
 * call only enqueues on ll_stream(ctx) and returns (no host hop, no stream synchronisation).  The caller runs its collective
 * (RCCL all-reduce / all-gather on the same buffers) stream-ordered with ll_stream(ctx): the 224 B all-reduce of the normal
 * equations and the K-NN candidates never touch the host. ---- */
extern "C" int ll_map_evaluate_dev(ll_map *m, double *neq44_dev)
{
    if (!m || !neq44_dev) return LL_ERR_ARG;
    LLM_HIP(hipSetDevice(m->ctx->device));
    ll_map_launch_normal_eq(m->M, m->ctx->stream);
    ll_copy_d2d(neq44_dev, m->M.neq, LL_NEQ_STRIDE * sizeof(double), m->ctx->stream);
    LLM_HIP(hipGetLastError());
    return LL_OK;
}

static int map_lm_stage_dev(ll_map *m, int stage, const double *neq44_sum_dev, const ll_lm_options *opt)
{
    if (!m || (stage != 1 && !neq44_sum_dev)) return LL_ERR_ARG;
    LLLmOpt o = ll_to_dev_opt(opt);
    o.nan_poisons_pose = 1;
    if (o.max_num_iterations < 0 || o.max_num_iterations > 64) { m->err = "max_num_iterations out of range"; return LL_ERR_ARG; }   /* as ll_map_optimize */
    LLM_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    if (neq44_sum_dev) ll_copy_d2d(m->M.neq, neq44_sum_dev, LL_NEQ_STRIDE * sizeof(double), st);
    LLView Vm = m->ctx->V;
    Vm.pose = m->M.pose; Vm.neq = m->M.neq; Vm.lm = m->M.lm;
    if (stage == 0) ll_launch_lm_begin(Vm, 0, 1, o, st);
    else if (stage == 1) ll_launch_lm_propose(Vm, 0, 1, o, st);
    else ll_launch_lm_accept(Vm, 0, 1, o, st);
    LLM_HIP(hipGetLastError());
    return LL_OK;
}
extern "C" int ll_map_lm_begin_dev(ll_map *m, const double *neq44_sum_dev, const ll_lm_options *opt) { return map_lm_stage_dev(m, 0, neq44_sum_dev, opt); }
extern "C" int ll_map_lm_propose_dev(ll_map *m, const ll_lm_options *opt) { return map_lm_stage_dev(m, 1, nullptr, opt); }
extern "C" int ll_map_lm_accept_dev(ll_map *m, const double *neq44_sum_dev, const ll_lm_options *opt) { return map_lm_stage_dev(m, 2, neq44_sum_dev, opt); }

extern "C" int ll_map_knn_partial_dev(ll_map *m, float *corner_nn_dev, int *corner_id_dev, float *surf_nn_dev, int *surf_id_dev)
{
    if (!m) return LL_ERR_ARG;
    if ((m->M.n_stk[0] > 0 && (!corner_nn_dev || !corner_id_dev)) || (m->M.n_stk[1] > 0 && (!surf_nn_dev || !surf_id_dev))) { m->err = "null candidate buffers"; return LL_ERR_ARG; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    ll_map_launch_knn_partial(m->M, st);                        /* at the pose already on the device */
    float *nn[2] = {corner_nn_dev, surf_nn_dev}; int *id[2] = {corner_id_dev, surf_id_dev};
    for (int w = 0; w < 2; ++w) {
        const size_t n = (size_t)m->M.n_stk[w] * 5;
        if (n == 0) continue;
        ll_copy_d2d(nn[w], m->M.nn_pt[w], n * sizeof(float4), st);
        ll_copy_d2d(id[w], m->M.nn_id[w], n * sizeof(int), st);
    }
    LLM_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_map_associate_merged_dev(ll_map *m, int n_parts, const float *corner_nn_dev, const int *corner_id_dev, const float *surf_nn_dev, const int *surf_id_dev)
{
    if (!m) return LL_ERR_ARG;
    if (n_parts < 1 || n_parts > 64) { m->err = "n_parts out of range"; return LL_ERR_ARG; }
    if ((m->M.n_stk[0] > 0 && (!corner_nn_dev || !corner_id_dev)) || (m->M.n_stk[1] > 0 && (!surf_nn_dev || !surf_id_dev))) { m->err = "null candidate buffers"; return LL_ERR_ARG; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    /* the gathered candidates are read in place: no copy into the map's own buffers */
    const float4 *pt_all[2] = {(const float4 *)corner_nn_dev, (const float4 *)surf_nn_dev}; const int *id_all[2] = {corner_id_dev, surf_id_dev};
    ll_map_launch_associate_merged(m->M, n_parts, pt_all, id_all, m->ctx->stream);
    LLM_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_map_solve_dev(ll_map *m, const ll_lm_options *opt)
{
    if (!m) return LL_ERR_ARG;
    const LLLmOpt o = ll_to_dev_opt(opt);
    if (o.max_num_iterations < 0 || o.max_num_iterations > 64) { m->err = "max_num_iterations out of range"; return LL_ERR_ARG; }
    if (m->M.row_world > 1) { m->err = "the map sums a row shard: step with ll_map_evaluate_dev + all-reduce + ll_map_lm_*_dev"; return LL_ERR_STATE; }
    LLM_HIP(hipSetDevice(m->ctx->device));
    /* no read-back here to notice a timed-out predecessor by: the hand-over words are zeroed on the stream before every solve */
    ll_fill_words((int *)m->M.lm_go, 1, 0, 0, 1, m->ctx->stream);
    ll_fill_words((int *)m->M.neq_ticket, 1, 0, 0, 1, m->ctx->stream);
    map_lm_solve(m, o);                                          /* from the pose on the device, result left there (ll_map_get_pose) */
    LLM_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_voxel_grid(ll_ctx *ctx, const ll_point *host_in, int n, float leaf_size, ll_point *host_out, int cap, int *n_out)
{
    if (!ctx || !n_out || n < 0 || (n > 0 && !host_in) || !(leaf_size > 0.0f)) { if (ctx) ctx->err = "bad voxel grid arguments"; return LL_ERR_ARG; }
    *n_out = 0;
    if (n == 0) return LL_OK;
    LL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *work = nullptr; float4 *d_in = nullptr, *d_out = nullptr; int *d_n = nullptr;
    int rc = LL_OK;
    const int seg_off[2] = {0, n};
    LLVoxWork W;
    auto fail = [&](const char *what) { ctx->err = what; rc = LL_ERR_HIP; };
    if (hipMalloc(&work, ll_vox_work_bytes(n, 1)) != hipSuccess || hipMalloc((void **)&d_in, (size_t)n * 16) != hipSuccess ||
        hipMalloc((void **)&d_out, (size_t)n * 16) != hipSuccess || hipMalloc((void **)&d_n, sizeof(int)) != hipSuccess) fail("hipMalloc failed");
    if (rc == LL_OK) {
        ll_vox_work_carve(work, n, 1, &W);
        if (hipMemcpyAsync(d_in, host_in, (size_t)n * 16, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(W.seg_off, seg_off, sizeof(seg_off), hipMemcpyHostToDevice, st) != hipSuccess) fail("upload failed");
    }
    if (rc == LL_OK) {
        const int vrc = ll_voxel_grid_segments(d_in, n, 1, leaf_size, W, d_out, d_n, st);
        int m = 0;
        if (vrc) fail("voxel grid: read-back failed");
        else if (hipMemcpyAsync(&m, d_n, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) fail("voxel grid failed");
        else if (m > cap) { ctx->err = "voxel grid output capacity too small"; rc = LL_ERR_CAPACITY; *n_out = m; }
        else {
            *n_out = m;
            if (m > 0 && host_out && (hipMemcpyAsync(host_out, d_out, (size_t)m * 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) fail("download failed");
        }
    }
    (void)hipFree(d_n); (void)hipFree(d_out); (void)hipFree(d_in); (void)hipFree(work);
    return rc;
}

/* ------------------------------------------------------------------ downloads */
static int dl(ll_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!dst || !bytes) return LL_OK;
    LL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return LL_OK;
}

static int fetch_hdr(ll_ctx *ctx, int slot, ScanHdr *h)
{
    if (ll_read_back(h, ctx->V.hdr + slot, sizeof(ScanHdr), ctx->stream)) { ctx->err = "read-back failed"; return LL_ERR_HIP; }
    return LL_OK;
}

extern "C" int ll_get_scan_info(ll_ctx *ctx, int slot, ll_scan_info *info)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    if (!info) return LL_ERR_ARG;
    ScanHdr h; rc = fetch_hdr(ctx, slot, &h); if (rc) return rc;
    info->status = h.status; info->n_in = ctx->n_in_host[slot]; info->n = h.n;
    info->n_sharp = h.n_sharp; info->n_less_sharp = h.n_less_sharp; info->n_flat = h.n_flat; info->n_less_flat = h.n_less_flat;
    info->max_ring = h.max_ring;
    return LL_OK;
}

extern "C" int ll_download_cloud(ll_ctx *ctx, int slot, ll_point *cloud, int cap, int *scan_start, int *scan_end)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    ScanHdr h; rc = fetch_hdr(ctx, slot, &h); if (rc) return rc;
    LLView &V = ctx->V;
    /* a slot the registration refused (a ring beyond max_ring_points: LL_ERR_CAPACITY) holds a truncated laserCloud whose gaps
     * would come back as stale points: hand the status back instead of LL_OK */
    if (h.status != 0 && h.status != LL_ERR_EMPTY) { ctx->err = "the slot's scan was refused (status " + std::to_string(h.status) + "): nothing to download"; return h.status; }
    if (cloud) {
        if (cap < h.n) { ctx->err = "cloud capacity too small"; return LL_ERR_CAPACITY; }
        ll_launch_cloud_flatten(V, slot, ctx->cloud_flat, ctx->stream);          /* the rings sit at a fixed stride on the device */
        rc = dl(ctx, cloud, ctx->cloud_flat, (size_t)h.n * 16); if (rc) return rc;
    }
    std::vector<int> off(V.R + 1);
    rc = dl(ctx, off.data(), V.ring_off + (size_t)slot * (V.R + 1), (size_t)(V.R + 1) * sizeof(int)); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < V.R; ++r) {
        if (scan_start) scan_start[r] = off[r] + 5;          /* scanRegistration.cpp:218 */
        if (scan_end) scan_end[r] = off[r + 1] - 6;          /* :220 */
    }
    return LL_OK;
}

extern "C" int ll_download_labels(ll_ctx *ctx, int slot, int8_t *label, float *curvature, int cap)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    ScanHdr h; rc = fetch_hdr(ctx, slot, &h); if (rc) return rc;
    LLView &V = ctx->V;
    if (h.status != 0 && h.status != LL_ERR_EMPTY) { ctx->err = "the slot's scan was refused (status " + std::to_string(h.status) + "): nothing to download"; return h.status; }
    if (cap < h.n) { ctx->err = "label capacity too small"; return LL_ERR_CAPACITY; }
    if (curvature && !V.write_curv) { ctx->err = "curvature requested but write_curvature = 0"; return LL_ERR_STATE; }
    rc = dl(ctx, label, V.label + (size_t)slot * V.NP, (size_t)h.n); if (rc) return rc;
    rc = dl(ctx, curvature, V.curv + (size_t)slot * V.NP, (size_t)h.n * 4); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

extern "C" int ll_download_features(ll_ctx *ctx, int slot, ll_point *sharp, int cap_sharp, ll_point *less_sharp, int cap_ls,
                                    ll_point *flat, int cap_flat, ll_point *less_flat, int cap_lf)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    ScanHdr h; rc = fetch_hdr(ctx, slot, &h); if (rc) return rc;
    LLView &V = ctx->V;
    if ((sharp && cap_sharp < h.n_sharp) || (less_sharp && cap_ls < h.n_less_sharp) || (flat && cap_flat < h.n_flat) ||
        (less_flat && cap_lf < h.n_less_flat)) { ctx->err = "feature capacity too small"; return LL_ERR_CAPACITY; }
    rc = dl(ctx, sharp, V.sharp + (size_t)slot * V.cap_sharp, (size_t)h.n_sharp * 16); if (rc) return rc;
    rc = dl(ctx, less_sharp, V.lsharp + (size_t)slot * V.cap_lsharp, (size_t)h.n_less_sharp * 16); if (rc) return rc;
    rc = dl(ctx, flat, V.flat + (size_t)slot * V.cap_flat, (size_t)h.n_flat * 16); if (rc) return rc;
    if (less_flat && h.n_less_flat > 0) {
        if (h.lf_strided) {                                              /* ring rows -> the reference's contiguous cloud, on the device */
            ll_launch_lflat_flatten(V, slot, ctx->cloud_flat, ctx->stream);
            rc = dl(ctx, less_flat, ctx->cloud_flat, (size_t)h.n_less_flat * 16); if (rc) return rc;
        } else { rc = dl(ctx, less_flat, V.lflat + (size_t)slot * V.LFS, (size_t)h.n_less_flat * 16); if (rc) return rc; }
    }
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

static int fetch_pair(ll_ctx *ctx, int slot, PairHdr *p)
{
    if (ll_read_back(p, ctx->V.pair + slot, sizeof(PairHdr), ctx->stream)) { ctx->err = "read-back failed"; return LL_ERR_HIP; }
    return LL_OK;
}

extern "C" int ll_get_pair_info(ll_ctx *ctx, int slot, ll_pair_info *info)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    if (!info) return LL_ERR_ARG;
    PairHdr p; rc = fetch_pair(ctx, slot, &p); if (rc) return rc;
    info->n_edge = p.n_edge; info->n_plane = p.n_plane; info->n_plane_selected = p.n_plane_sel;
    return LL_OK;
}

extern "C" int ll_download_edge_corr(ll_ctx *ctx, int slot, int *src, int *a, int *b, int cap)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    PairHdr p; rc = fetch_pair(ctx, slot, &p); if (rc) return rc;
    if (cap < p.n_edge) { ctx->err = "edge capacity too small"; return LL_ERR_CAPACITY; }
    LLView &V = ctx->V; const size_t o = (size_t)slot * V.cap_sharp, by = (size_t)p.n_edge * sizeof(int);
    rc = dl(ctx, src, V.e_src + o, by); if (rc) return rc;
    rc = dl(ctx, a, V.e_a + o, by); if (rc) return rc;
    rc = dl(ctx, b, V.e_b + o, by); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

extern "C" int ll_download_plane_corr(ll_ctx *ctx, int slot, int *src, int *a, int *b, int *c, int cap)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    PairHdr p; rc = fetch_pair(ctx, slot, &p); if (rc) return rc;
    if (cap < p.n_plane) { ctx->err = "plane capacity too small"; return LL_ERR_CAPACITY; }
    LLView &V = ctx->V; const size_t o = (size_t)slot * V.cap_flat, by = (size_t)p.n_plane * sizeof(int);
    rc = dl(ctx, src, V.p_src + o, by); if (rc) return rc;
    rc = dl(ctx, a, V.p_a + o, by); if (rc) return rc;
    rc = dl(ctx, b, V.p_b + o, by); if (rc) return rc;
    rc = dl(ctx, c, V.p_c + o, by); if (rc) return rc;
    /* inside the library a less-flat target point is named by its PLACE in the target slot (ll_common.h); the reference's
     * closestPointInd / minPointInd2 / minPointInd3 (laserOdometry.cpp:661-721) are indices into the contiguous cloud */
    std::vector<int> pre;
    int stride = 0;
    if (p.target_slot >= 0 && p.n_plane > 0) {
        ScanHdr th; rc = fetch_hdr(ctx, p.target_slot, &th); if (rc) return rc;
        if (th.lf_strided) {
            pre.resize((size_t)V.R + 1); stride = V.ring_cap;
            rc = dl(ctx, pre.data(), V.lf_pre + (size_t)p.target_slot * (V.R + 1), pre.size() * sizeof(int)); if (rc) return rc;
        }
    }
    LL_HIP(hipStreamSynchronize(ctx->stream));
    if (stride) {
        int *arr[3] = {a, b, c};
        for (int *x : arr) {
            if (!x) continue;
            for (int i = 0; i < p.n_plane; ++i) if (x[i] >= 0) { const int r = x[i] / stride; x[i] = pre[(size_t)r] + (x[i] - r * stride); }
        }
    }
    return LL_OK;
}

extern "C" int ll_download_vote(ll_ctx *ctx, int slot, int *count, uint8_t *selected, float *weight, int cap)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    PairHdr p; rc = fetch_pair(ctx, slot, &p); if (rc) return rc;
    if (cap < p.n_plane) { ctx->err = "vote capacity too small"; return LL_ERR_CAPACITY; }
    LLView &V = ctx->V; const size_t o = (size_t)slot * V.cap_flat;
    rc = dl(ctx, count, V.v_count + o, (size_t)p.n_plane * sizeof(int)); if (rc) return rc;
    rc = dl(ctx, selected, V.v_sel + o, (size_t)p.n_plane); if (rc) return rc;
    rc = dl(ctx, weight, V.v_w + o, (size_t)p.n_plane * sizeof(float)); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

extern "C" int ll_download_normal_equations(ll_ctx *ctx, int slot, double *H36, double *g6, double *cost)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    double buf[LL_NEQ_STRIDE];
    rc = dl(ctx, buf, ctx->V.neq + (size_t)slot * LL_NEQ_STRIDE, sizeof(buf)); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    if (H36) std::memcpy(H36, buf, 36 * sizeof(double));
    if (g6) std::memcpy(g6, buf + 36, 6 * sizeof(double));
    if (cost) *cost = buf[42];
    return LL_OK;
}

extern "C" int ll_download_pose(ll_ctx *ctx, int slot, double *pose7)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    rc = dl(ctx, pose7, ctx->V.pose + (size_t)slot * 7, 7 * sizeof(double)); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

extern "C" int ll_residual_jacobian(ll_ctx *ctx, int slot, const double *pose7, double *r, double *Jq, double *Jt, int cap_rows)
{
    int rc = check_range(ctx, slot, 1); if (rc) return rc;
    PairHdr p; rc = fetch_pair(ctx, slot, &p); if (rc) return rc;
    const size_t rows = (size_t)3 * p.n_edge + p.n_plane_sel;
    if ((size_t)cap_rows < rows) { ctx->err = "row capacity too small"; return LL_ERR_CAPACITY; }
    if (rows == 0) return LL_OK;
    if (ctx->rows_cap < rows) {
        void *q = nullptr;
        LL_HIP(hipMalloc(&q, rows * 8 * sizeof(double)));
        ctx->allocs.push_back(q); ctx->d_rows = (double *)q; ctx->rows_cap = rows;
    }
    const double *dpose = ctx->V.pose + (size_t)slot * 7;
    if (pose7) {
        LL_HIP(hipMemcpyAsync(ctx->d_tmp_pose, pose7, 7 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dpose = ctx->d_tmp_pose;
    }
    double *dr = ctx->d_rows, *dJq = dr + rows, *dJt = dJq + rows * 4;
    ctx->V.carry_slot = ctx->V.carry_slot;   /* unchanged: rows use the same target selection as the last stage call */
    ll_launch_rows(ctx->V, slot, dpose, dr, dJq, dJt, ctx->stream);
    LL_HIP(hipGetLastError());
    rc = dl(ctx, r, dr, rows * sizeof(double)); if (rc) return rc;
    rc = dl(ctx, Jq, dJq, rows * 4 * sizeof(double)); if (rc) return rc;
    rc = dl(ctx, Jt, dJt, rows * 3 * sizeof(double)); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

/* debug: the 16 phase-timing counters (all zero unless built with -DLL_PHASE_TIMING); reset = zero them afterwards */
/* PMC calibration: a float4 streaming copy of exactly `bytes` read + `bytes` written (k_calib_copy in the kernel trace).
 * rocprofv3's FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section);
 * tools/pmc_traffic.py divides the known byte count by the counter value of this launch to get the correction. */
__global__ void k_calib_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

/* One stage's launch on its own (tools/experiments/overlap_probe.py: which kernels gain from running beside which): 0 organise, 1 pick,
 * 2 voxel filter + lists, 3 grid tables, 4 association, 5 vote, 6 normal equations + step -- the launches of ll_hot_path_batch, one stage per call.
 * The slots must hold what the stage reads. */
extern "C" int ll_debug_launch_stage(ll_ctx *ctx, int stage, int first, int count)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    LL_HIP(hipSetDevice(ctx->device));
    switch (stage) {
    case 0: ll_launch_organize(ctx->V, first, count, ctx->stream, &ctx->prof); break;
    case 1: ll_launch_features(ctx->V, first, count, ctx->feat_lds, ctx->stream, &ctx->prof, 1); break;
    case 2: ll_launch_features(ctx->V, first, count, ctx->feat_lds, ctx->stream, &ctx->prof, 2); break;
    case 3: ll_launch_build_grid(ctx->V, first, count, 0, ctx->stream, &ctx->prof); break;
    case 4: ll_launch_associate(ctx->V, first, count, ctx->stream, &ctx->prof); break;
    case 5: ll_launch_vote(ctx->V, first, count, 1, ctx->stream, &ctx->prof); break;
    case 6: ll_launch_normal_equations(ctx->V, first, count, 1, ctx->stream, &ctx->prof); break;
    default: ctx->err = "no such stage"; return LL_ERR_ARG;
    }
    LL_HIP(hipGetLastError());
    return LL_OK;
}

extern "C" int ll_debug_calibration_copy(ll_ctx *ctx, unsigned long long bytes)
{
    int rc0 = ll_enter(ctx); if (rc0) return rc0;
    const size_t n = (size_t)(bytes / 16);
    void *a = nullptr, *b = nullptr;
    LL_HIP(hipMalloc(&a, n * 16)); LL_HIP(hipMalloc(&b, n * 16));
    LL_HIP(hipMemsetAsync(a, 1, n * 16, ctx->stream));
    hipLaunchKernelGGL(k_calib_copy, dim3(2048), dim3(256), 0, ctx->stream, (const float4 *)a, (float4 *)b, n);
    LL_HIP(hipGetLastError());
    LL_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(a); (void)hipFree(b);
    return LL_OK;
}

extern "C" int ll_debug_exact_math(ll_ctx *ctx, int op, const float *a, const float *b, const float *c, int n, void *out)
{
    int rc = ll_enter(ctx); if (rc) return rc;
    if (op < 0 || op > 6 || n < 0 || (n > 0 && (!a || !out)) || (n > 0 && (op == 1 || op == 2 || op >= 4) && !b) || (n > 0 && op >= 4 && !c)) {
        ctx->err = "bad ll_debug_exact_math arguments"; return LL_ERR_ARG;
    }
    if (n == 0) return LL_OK;
    float *d = nullptr;
    LL_HIP(hipMalloc((void **)&d, (size_t)n * 4 * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d, a, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && b) e = hipMemcpyAsync(d + n, b, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && c) e = hipMemcpyAsync(d + 2 * (size_t)n, c, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        ll_launch_debug_exact_math(ctx->V, op, d, b ? d + n : nullptr, c ? d + 2 * (size_t)n : nullptr, n, d + 3 * (size_t)n, ctx->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + 3 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { ctx->err = std::string("ll_debug_exact_math: ") + hipGetErrorString(e); return LL_ERR_HIP; }
    return LL_OK;
}

extern "C" int ll_debug_counters(ll_ctx *ctx, unsigned long long *out16, int reset)
{
    if (!ctx || !out16) return LL_ERR_ARG;
    LL_HIP(hipSetDevice(ctx->device));
    LL_HIP(hipMemcpyAsync(out16, ctx->V.dbg, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    if (reset) LL_HIP(hipMemsetAsync(ctx->V.dbg, 0, 16 * sizeof(unsigned long long), ctx->stream));
    LL_HIP(hipStreamSynchronize(ctx->stream));
    return LL_OK;
}

extern "C" int ll_profile_enable(ll_ctx *ctx, int on)
{
    int rc0 = ll_enter(ctx); if (rc0) return rc0;
    LLProfiler &P = ctx->prof;
    if (on && P.ev.empty()) {
        P.ev.resize(LL_PROF_EVENTS); P.id.assign(LL_PROF_EVENTS, LL_K_END);
        for (auto &e : P.ev) LL_HIP(hipEventCreate(&e));
    }
    P.on = on != 0;
    return LL_OK;
}

extern "C" int ll_profile_read(ll_ctx *ctx, int *n, const char **names, double *total_ms, int *launches, int reset)
{
    if (!ctx || !n) return LL_ERR_ARG;
    LLProfiler &P = ctx->prof;
    LL_HIP(hipSetDevice(ctx->device));
    LL_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i + 1 < P.n; ++i) {
        const int k = P.id[i] & 0xff;
        if (P.id[i] == LL_K_END || k >= LL_K_COUNT) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, P.ev[i], P.ev[i + 1]) == hipSuccess) { P.total_ms[k] += ms; P.launches[k]++; }
    }
    P.n = 0;
    const int m = (*n < LL_K_COUNT) ? *n : LL_K_COUNT;
    for (int k = 0; k < m; ++k) {
        if (names) names[k] = kKernelNames[k];
        if (total_ms) total_ms[k] = P.total_ms[k];
        if (launches) launches[k] = P.launches[k];
    }
    *n = m;
    if (reset) for (int k = 0; k < LL_K_COUNT; ++k) { P.total_ms[k] = 0; P.launches[k] = 0; }
    return LL_OK;
}

extern "C" int ll_algorithmic_bytes(ll_ctx *ctx, int first, int count, double *b_ext, double *b_assoc, double *b_vote, double *b_rj)
{
    int rc = check_range(ctx, first, count); if (rc) return rc;
    LLView &V = ctx->V;
    std::vector<ScanHdr> h(count); std::vector<PairHdr> p(count);
    int carry[2] = {0, 0};
    rc = dl(ctx, h.data(), V.hdr + first, sizeof(ScanHdr) * count); if (rc) return rc;
    rc = dl(ctx, p.data(), V.pair + first, sizeof(PairHdr) * count); if (rc) return rc;
    rc = dl(ctx, carry, V.carry_cnt, sizeof(carry)); if (rc) return rc;
    LL_HIP(hipStreamSynchronize(ctx->stream));
    double be = 0, ba = 0, bv = 0, br = 0;
    for (int i = 0; i < count; ++i) {
        const ScanHdr &s = h[i];
        if (s.status != 0) { be += 4.0 * ctx->V.raw_stride * ctx->n_in_host[first + i]; continue; }
        /* SURVEY.md section 8d */
        be += 4.0 * ctx->V.raw_stride * ctx->n_in_host[first + i] + 16.0 * s.n + 16.0 * (s.n_sharp + s.n_less_sharp + s.n_flat + s.n_less_flat) + 1.0 * s.n;
        const int mc = (i == 0) ? carry[0] : h[i - 1].n_less_sharp, ms = (i == 0) ? carry[1] : h[i - 1].n_less_flat;
        ba += 16.0 * (s.n_sharp + s.n_flat) + 16.0 * (mc + ms) + 8.0 * p[i].n_edge + 12.0 * p[i].n_plane;
        bv += 32.0 * p[i].n_plane + 8.0 * p[i].n_plane_sel;
        br += 48.0 * p[i].n_edge + 64.0 * p[i].n_plane_sel + 216.0;
    }
    if (b_ext) *b_ext = be; if (b_assoc) *b_assoc = ba; if (b_vote) *b_vote = bv; if (b_rj) *b_rj = br;
    return LL_OK;
}
