/*
 * ll_internal.h -- what the C-ABI translation units (ll_api.hip, ll_cubemap.hip) share: the context structs behind the
 * opaque handles of include/lightloam_hip.h and the error-return macros.
 */
#pragma once
#include "lightloam_hip.h"
#include "ll_common.h"
#include <string>
#include <vector>
#include <cstring>

#define LL_PROF_EVENTS 8192
#define LL_TWO_STREAM_PIECES 4
#define LL_TWO_STREAM_MAX_PIECES 8
struct LLProfiler {
    bool on = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> id;          /* kernel that STARTS at event i, LL_K_END for a closing mark */
    int n = 0;
    double total_ms[LL_K_COUNT] = {0};
    int launches[LL_K_COUNT] = {0};
};
struct ll_ctx {
    LLProfiler prof;
    ll_params p;
    LLView V;
    hipStream_t stream = nullptr;
    int device = 0;
    std::vector<void *> allocs;
    std::string err;
    size_t feat_lds = 0;
    float4 *h_stage = nullptr;      /* pinned staging for uploads */
    float4 *cloud_flat = nullptr;   /* [NP] device staging of ll_download_cloud: one slot's laserCloud without the ring stride */
    size_t h_stage_pts = 0;
    double *d_tmp_pose = nullptr, *d_rows = nullptr;
    size_t rows_cap = 0;
    std::vector<int> n_in_host;
    hipEvent_t ev[16];
    bool ev_ok = false;
    /* caller-supplied residual blocks (ll_factor_blocks_set / _evaluate) */
    double *d_fb = nullptr, *d_fb_out = nullptr;
    size_t fb_cap = 0, fb_out_cap = 0;
    int fb_n[3] = {0, 0, 0};
    double *d_fb_s = nullptr; size_t fb_s_cap = 0; bool fb_has_s = false;   /* per-block s of the edge / plane blocks (ll_factor_blocks_set_s) */
    /* streaming input (ll_upload_scan_async): a second stream for host -> device copies, ordered against the compute stream by
     * ll_stream_record / ll_stream_wait; the point counts go down by value (k_set_counts), not from a staging cell */
    /* the association stage of ll_hot_path_batch on two streams: k_build_grid of piece i + 1 beside k_associate of piece i */
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_ts[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int ts_pieces = 4;              /* pieces the slot range of a call is cut into (LIGHTLOAM_TS_PIECES: A/B runs) */
    int two_stream = 0;             /* ll_set_two_stream; LIGHTLOAM_TWO_STREAM=1 in the environment turns it on at ll_create (measured: no gain, ll_api.hip) */
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_x[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   /* ll_stream_record / ll_stream_wait */
};


#define LL_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return LL_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

struct ll_map {
    ll_ctx *ctx = nullptr;
    LLMapView M;
    int cap_map[2] = {0, 0}, cap_stk[2] = {0, 0};
    int max_cells = 0;
    float4 *d_map[2] = {nullptr, nullptr}, *d_stk[2] = {nullptr, nullptr};
    int *d_bbox = nullptr, *d_tile = nullptr;
    /* tile-parallel search: global ids of a shard's map points, this rank's candidates, everybody's candidates */
    int *d_gid[2] = {nullptr, nullptr};
    float4 *d_all_pt[2] = {nullptr, nullptr}; int *d_all_id[2] = {nullptr, nullptr};
    int cap_parts = 0;
    std::vector<void *> allocs;
    std::string err;
};

#define LLM_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) { m->err = std::string(#call) + ": " + hipGetErrorString(e_); return LL_ERR_HIP; } \
    } while (0)

template <typename T>
static inline bool map_alloc(ll_map *m, T *&ptr, size_t count)
{
    void *p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) { m->err = "hipMalloc failed (" + std::to_string(bytes) + " bytes)"; return false; }
    if (hipMemset(p, 0, bytes) != hipSuccess) { m->err = "hipMemset failed"; (void)hipFree(p); return false; }
    m->allocs.push_back(p);
    ptr = (T *)p;
    return true;
}


LLLmOpt ll_to_dev_opt(const ll_lm_options *opt);
int ll_map_rebuild(ll_map *m, int n_corner, int n_surf);      /* grids over the clouds already in d_map[] */
void ll_map_rebuild_begin(ll_map *m, int n_corner, int n_surf);       /* ... in two halves around the caller's own read-back */
void ll_map_rebuild_finish(ll_map *m, const int bbox_host[12]);
int ll_map_use_ids(ll_map *m, bool on);                       /* search ties by d_gid[] (a tile shard) or by position */
