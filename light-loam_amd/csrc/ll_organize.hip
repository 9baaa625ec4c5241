/*
 * ll_organize.hip -- a1: NaN / minimum-range filter, ring + azimuth assignment, stable ring bucketing.
 * Replaces scanRegistration.cpp:58-85 (removeClosedPointCloud), :105-221 of /root/reference.
 *
 * The reference loop is sequential (one push_back per point, one halfPassed flag).  k_organize keeps the order of the
 * walk -- one workgroup per scan, tile after tile -- and makes every tile data-parallel; see the kernel's comment.
 */
#include "ll_common.h"
#include <limits.h>

__device__ __forceinline__ bool ll_xcd_map(int id, int per_scan, int count, int &scan, int &item)
{
    const int xcd = id & 7, j = id >> 3;
    scan = (j / per_scan) * 8 + xcd;
    item = j % per_scan;
    return scan < count;
}

/* The resident raw scan of a slot (ll_params.input_stride_floats): 4 floats per point (KITTI .bin / PointXYZ: x, y, z and one float the
 * reference never reads, scanRegistration.cpp:105-106) or 3 (x, y, z packed -- a quarter fewer bytes over PCIe and out of HBM).  A slot's
 * area starts at the same 16 NP-byte stride either way; raw[i] is the point as (x, y, z, 0).  S3 is a template argument where the
 * register allocation matters (k_organize) and a run-time flag on the node-style path. */
struct __attribute__((packed, aligned(4))) LLXyz { float x, y, z; };
template <bool S3>
struct LLRaw {
    const float *b;
    __device__ __forceinline__ float4 operator[](int i) const {
        if (S3) { const LLXyz q = ((const LLXyz *)b)[i]; return make_float4(q.x, q.y, q.z, 0.0f); }     /* one global_load_dwordx3 */
        return ((const float4 *)b)[i];
    }
};
struct LLRawAny {
    const float *b; bool s3;
    __device__ __forceinline__ float4 operator[](int i) const {
        if (s3) { const LLXyz q = ((const LLXyz *)b)[i]; return make_float4(q.x, q.y, q.z, 0.0f); }
        return ((const float4 *)b)[i];
    }
};
__device__ __forceinline__ const float *ll_raw_base(const LLView &V, int s) { return (const float *)(V.raw + (size_t)s * V.NP); }

__device__ __forceinline__ bool ll_keep(const float4 p, float thres)
{
    /* pcl::removeNaNFromPointCloud (:109) then x*x + y*y + z*z < thres*thres -> drop (:72), all f32 */
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return false;
    return !(p.x * p.x + p.y * p.y + p.z * p.z < thres * thres);
}

/* scanID of a point (:139-168): the chain atan -> degrees -> ring formula -> int() is monotone in t = z / sqrt(x^2 + y^2), so
 * the ring is the number of precomputed thresholds (ll_ring_thresholds, exact for this context's parameters) not above t,
 * minus one.  thr: R + 2 keys in LDS (the last one INT_MAX).  With a bucket table (ll_ring_lut_build) the count starts at
 * the bucket's first ring and needs two comparisons; without one it is a binary search over the R + 1 keys. */
template <bool LUT>
__device__ __forceinline__ int ll_scan_id(const int *thr, const int *lut, int nb, float t0, float scale, int R, const float4 p)
{
    const float t = p.z / sqrtf(p.x * p.x + p.y * p.y);
    if (t != t) return -1;                                  /* 0 / 0: the reference's int(NaN) is INT_MIN -> rejected */
    const int key = ll_float_key(t);
    int id;
    if (LUT) {
        const int g = lut[ll_ring_bucket(t, t0, scale, nb)];
        id = g + (key >= thr[g + 1] ? 1 : 0) + (key >= thr[g + 2] ? 1 : 0);
        if (key < thr[0]) id = -1;
    } else {
        int lo = 0, hi = R + 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (thr[mid] <= key) lo = mid + 1; else hi = mid; }
        id = lo - 1;
    }
    return (id > R - 1 || id < 0) ? -1 : id;
}

/* a ring longer than the common capacity goes on the work list of its tier (ll_launch_pick / ll_launch_features: 2304 < n <= 3072,
 * <= 4608, beyond): the order of the entries is whatever the atomics make it -- no ring's result depends on another's */
__device__ __forceinline__ void ll_tier_append(const LLView &V, int s, int r, int n)
{
    if (n <= 2304 || !V.tier_list) return;
    const int t = n <= 3072 ? 1 : n <= 4608 ? 2 : 3;
    const int pos = atomicAdd(&V.tier_cnt[t], 1);
    V.tier_list[(size_t)(t - 1) * V.B * V.R + pos] = (s << 8) | r;
}

/* One workgroup per scan walks its input in order, tile after tile (LL_TILE points: lanes hold consecutive indices, so the
 * first / last set lane of a ballot is the smallest / largest index):
 *   search   the first and the last point that survive the filters (:109, :72) -> startOri, endOri (:114-126); nothing
 *            before the first or after the last one is kept, so the walk covers only the tiles between them
 *   per tile keep test, scanID, ori = -atan2f(y, x) (:177), the state-free predicate P(i) = "(ori_i adjusted against
 *            startOri) - startOri > pi" whose first hit over the scan is where halfPassed flips (:189-192), relTime /
 *            intensity (:194-208), and a stable multi-split by ring: rank inside the wave from a match-any of the ring id,
 *            (sub-tile, wave, ring) counters in LDS, the rings' running totals carried from tile to tile in LDS.
 * The sequential walk is what makes one pass enough: a point's position in its ring needs the ring's count over all
 * earlier tiles, which a tile-parallel launch only has after a histogram pass over the whole scan (the round-1 design:
 * k_classify + k_offsets + k_scatter read the scan twice and kept ori / ring per point in HBM).  Ring r of slot s lands at
 * cloud[s * CS + r * ring_cap ...]: rings at a fixed stride, so no offset has to be known before the scan is through;
 * ring_off keeps the contiguous laserCloud offsets (:218-220) for the labels, the curvature and the C ABI.
 * HBM traffic per input point: 16 B read, 16 B written per kept point. */
#ifndef LL_OWAVES
#define LL_OWAVES 6       /* workgroups per CU (= waves per SIMD) of k_organize: 4 / 5 / 6 / 7 measured 7.88 / 7.31 / 7.00 / 7.00 ms per 8192 scans */
#endif
template <bool LUT, bool S3>
__global__ __launch_bounds__(LL_BLOCK, LL_OWAVES) void k_organize(LLView V, int first, int count)
{
    if ((int)blockIdx.x >= count) return;
    const int s = first + blockIdx.x;
    const int n_in = V.n_in[s];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LLRaw<S3> raw = {ll_raw_base(V, s)};
    constexpr int NK = LL_TILE / LL_BLOCK, NW = LL_BLOCK / 64;

    __shared__ int thr[LL_MAX_RINGS + 2];
    __shared__ int lut[LUT ? LL_RING_LUT_MAX : 1];
    __shared__ int cnt[2][NK * NW][LL_MAX_RINGS];           /* double-buffered: tile t counts into one while tile t - 1's bases are still read */
    __shared__ int ringtot[LL_MAX_RINGS];
    __shared__ int sh_fk, sh_lk, sh_half;
    if (tid <= V.R) thr[tid] = V.ring_thr[tid];
    if (tid == 0) { sh_fk = INT_MAX; sh_lk = -1; sh_half = INT_MAX; thr[V.R + 1] = INT_MAX; }
    if (tid < LL_MAX_RINGS) ringtot[tid] = 0;
    if (LUT) for (int i = tid; i < V.lut_nb; i += LL_BLOCK) lut[i] = V.ring_lut[i];
    for (int i = tid; i < 2 * NK * NW * LL_MAX_RINGS; i += LL_BLOCK) (&cnt[0][0][0])[i] = 0;
    __syncthreads();

    /* ---- the first and the last kept point ---- */
    for (int c = 0; c < n_in; c += LL_TILE) {
        int mine = INT_MAX;
#pragma unroll
        for (int k = NK - 1; k >= 0; --k) {
            const int i = c + k * LL_BLOCK + tid;
            if (i < n_in && ll_keep(raw[i], V.thres)) mine = i;
        }
        if (__syncthreads_or(mine != INT_MAX ? 1 : 0)) {
            if (mine != INT_MAX) atomicMin(&sh_fk, mine);
            break;
        }
    }
    __syncthreads();
    const int fk = sh_fk;
    ScanHdr h;
    h.start_ori = 0.0f; h.end_ori = 0.0f; h.first_kept = fk; h.last_kept = -1; h.half_idx = INT_MAX; h.n = 0; h.status = 0; h.max_ring = 0;
    h.n_sharp = h.n_less_sharp = h.n_flat = h.n_less_flat = 0; h.so_lo_up = 0.0f; h.so_hi_dn = 0.0f;
    h.lf_strided = 1;                                                                 /* an extracted scan: ring-strided less-flat cloud (ll_common.h) */
    int *ring_off = V.ring_off + (size_t)s * (V.R + 1);
    if (fk == INT_MAX) {                                                              /* nothing survives: LL_ERR_EMPTY */
        if (tid <= V.R) ring_off[tid] = 0;
        if (tid == 0) { h.status = -5; V.hdr[s] = h; }
        return;
    }
    const int t0 = fk / LL_TILE;
    for (int c = (n_in - 1) / LL_TILE * LL_TILE; c >= t0 * LL_TILE; c -= LL_TILE) {
        int mine = -1;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i = c + k * LL_BLOCK + tid;
            if (i < n_in && ll_keep(raw[i], V.thres)) mine = i;
        }
        if (__syncthreads_or(mine >= 0 ? 1 : 0)) {
            if (mine >= 0) atomicMax(&sh_lk, mine);
            break;
        }
    }
    __syncthreads();
    const int lk = sh_lk, t1 = lk / LL_TILE;
    /* startOri, endOri (:114-126): uniform, every thread computes them from the two points */
    float start_ori, end_ori, so_lo_up, so_hi_dn;
    {
        const float4 p0 = raw[fk], p1 = raw[lk];
        start_ori = -ll_atan2f(p0.y, p0.x);
        so_lo_up = ll_f32_ceil((double)start_ori - M_PI / 2);
        so_hi_dn = ll_f32_floor((double)start_ori + M_PI * 3 / 2);
        float eo = (float)((double)(-ll_atan2f_finite(p1.y, p1.x)) + 2 * M_PI);       /* -atan2f(last kept) + 2*pi in f64, stored f32 */
        if ((double)(eo - start_ori) > 3 * M_PI)     eo = (float)((double)eo - 2 * M_PI);
        else if ((double)(eo - start_ori) < M_PI)    eo = (float)((double)eo + 2 * M_PI);
        end_ori = eo;
    }

    /* ---- the walk ---- */
    float4 *cloud = V.cloud + (size_t)s * V.CS;
    const int cap = V.ring_cap;
    int bits = 0; while ((1 << bits) < V.R) ++bits;
    float4 p[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i = t0 * LL_TILE + k * LL_BLOCK + tid;
        p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n_in) p[k] = raw[i];
    }
    /* The first tile's points have landed before the walk starts, and every later tile's before the stores of the tile in front of it
     * (the wait below): inside the loop the compiler then knows p[] to be complete and leaves the NEXT tile's loads in flight
     * through the classification math.  (Without the two explicit waits it cannot tell how many of the predicated loads are
     * outstanding and waits for vmcnt(0) -- i.e. also for the loads it has just issued -- at the first use of a point.) */
    __builtin_amdgcn_s_waitcnt(0x0f70);                                                /* vmcnt(0) */
    int half = INT_MAX;                                                                /* half_idx once it is known (uniform) */
    for (int t = t0; t <= t1; ++t) {
        const int base = t * LL_TILE, buf = (t - t0) & 1;
        float4 pn[NK];                                                                 /* the next tile's points, in flight during this tile */
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i = base + LL_TILE + k * LL_BLOCK + tid;
            pn[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < t1 && i < n_in) pn[k] = raw[i];
        }
        int my_ring[NK], my_rank[NK]; float my_o[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i = base + k * LL_BLOCK + tid;
            const bool kept = i < n_in && ll_keep(p[k], V.thres);
            int id = -1; float o = 0.0f; bool firstp = false;
            if (kept) {
                o = -ll_atan2f_finite(p[k].y, p[k].x);                                    /* :177 */
                id = ll_scan_id<LUT>(thr, lut, V.lut_nb, V.lut_t0, V.lut_scale, V.R, p[k]);
                if (id >= 0 && half == INT_MAX) {
                    /* the !halfPassed branch (:180-192) evaluated as if the flag were still false; the float-against-double
                     * comparisons in f32 against bounds rounded outward (ll_f32_ceil / ll_f32_floor) */
                    float a = o;
                    const bool below = a < so_lo_up, above = a > so_hi_dn;
                    if (below || above) a = (float)((double)a + (below ? 2 * M_PI : -(2 * M_PI)));
                    firstp = (a - start_ori) >= 3.14159274101257324f;                 /* (double)(a - startOri) > M_PI */
                }
            }
            my_o[k] = o; my_ring[k] = id;
            const int i0 = base + k * LL_BLOCK + (tid & ~63);
            if (half == INT_MAX) {
                const unsigned long long pm = __ballot(firstp);
                if (pm && lane == 0) atomicMin(&sh_half, i0 + __ffsll((long long)pm) - 1);
            }
            /* rank among the wave's points of the same ring, in index order */
            const unsigned long long m = __ballot(id >= 0);
            const int id0 = __builtin_amdgcn_readfirstlane(id);
            int rank = 0;
            if (__ballot(id != id0) == 0ull) {                                         /* ring-major input: one ring per wave */
                rank = lane;
                if (id0 >= 0 && lane == 0) cnt[buf][k * NW + wave][id0] = 64;
            } else if (m) {
                unsigned mlo, mhi;
                ll_match_any(id, bits, m, mlo, mhi);
                rank = ll_match_rank(mlo, mhi);
                if (id >= 0 && rank == 0) cnt[buf][k * NW + wave][id] = ll_match_count(mlo, mhi);
            }
            my_rank[k] = rank;
        }
        __syncthreads();
        /* bases of the (sub-tile, wave) groups of every ring = the ring's running total + the groups before */
        if (tid < V.R) {
            int run = ringtot[tid];
#pragma unroll
            for (int kw = 0; kw < NK * NW; ++kw) { const int c = cnt[buf][kw][tid]; cnt[buf][kw][tid] = run; run += c; }
            ringtot[tid] = run;
        }
        for (int i = tid; i < (NK * NW) << bits; i += LL_BLOCK) cnt[buf ^ 1][i >> bits][i & ((1 << bits) - 1)] = 0;    /* the next tile's counters */
        if (half == INT_MAX) half = sh_half;
        __syncthreads();
        /* One unconditional wait for the next tile's points (in flight since the top of this iteration) in front of the stores.  Without it
         * the compiler -- which waited for this tile's points only inside the `kept` blocks above -- still counts their registers as in
         * flight here and puts an s_waitcnt vmcnt(0) in front of EVERY store below: each of a thread's four stores then waits for the
         * one before it to be acknowledged. */
        __builtin_amdgcn_s_waitcnt(0x0f70);                                            /* vmcnt(0) */
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int r = my_ring[k];
            if (r < 0) continue;
            const int i = base + k * LL_BLOCK + tid;
            float o = my_o[k];
            if (i <= half) {                                                          /* !halfPassed (:178-193) */
                if ((double)o < (double)start_ori - M_PI / 2)            o = (float)((double)o + 2 * M_PI);
                else if ((double)o > (double)start_ori + M_PI * 3 / 2)   o = (float)((double)o - 2 * M_PI);
            } else {                                                                  /* :194-205 */
                o = (float)((double)o + 2 * M_PI);
                if ((double)o < (double)end_ori - M_PI * 3 / 2)          o = (float)((double)o + 2 * M_PI);
                else if ((double)o > (double)end_ori + M_PI / 2)         o = (float)((double)o - 2 * M_PI);
            }
            const float rel = (o - start_ori) / (end_ori - start_ori);               /* :207 */
            const int pos = cnt[buf][k * NW + wave][r] + my_rank[k];
            if (pos < cap) cloud[(size_t)r * cap + pos] = make_float4(p[k].x, p[k].y, p[k].z, (float)((double)r + 0.1 * (double)rel));   /* :208 */
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) p[k] = pn[k];
    }
    __syncthreads();
    /* contiguous laserCloud offsets, cloudSize (:212), status */
    if (tid == 0) {
        int run = 0, mx = 0;
        for (int r = 0; r < V.R; ++r) { const int c = ringtot[r]; ring_off[r] = run; run += c; mx = max(mx, c); }
        ring_off[V.R] = run;
        if (mx > 2304 && mx <= V.max_ring) for (int r = 0; r < V.R; ++r) ll_tier_append(V, s, r, ringtot[r]);
        h.start_ori = start_ori; h.end_ori = end_ori; h.so_lo_up = so_lo_up; h.so_hi_dn = so_hi_dn;
        h.last_kept = lk; h.half_idx = half; h.n = run; h.max_ring = mx;
        if (mx > V.max_ring) h.status = -4;                                            /* LL_ERR_CAPACITY: the ring's tail was not stored */
        V.hdr[s] = h;
    }
}

/* ---- the tile-parallel path for calls of at most LL_ORG_SMALL scans (one scan at a time, the way the ROS nodes drive the
 * library): a lone workgroup walking a whole scan is latency-bound (~0.75 ms per 64-ring scan), so few scans are split
 * over (tile, scan) workgroups instead, at the price of a second pass over the input:
 *   k_first_kept  first kept point -> startOri
 *   k_classify    per tile: keep test, scanID, ori, first-P, ring histogram; ori / ring per point to scratch
 *   k_offsets     per scan: scan of the tile histograms -> position of every (tile, ring) bucket inside its ring, endOri
 *   k_scatter     per tile: relTime / intensity, stable multi-split by ring
 * Same arithmetic, same laserCloud layout (rings at a fixed stride) and the same ScanHdr as k_organize; the scratch arrays
 * (ori, ring, tile tables) hold LL_ORG_SMALL scans and are indexed by the scan's position in the launch. ---- */
/* first kept point of each scan -> startOri (:114).  One workgroup per scan, 4 points per thread per round, stops at
 * the first round that keeps anything.  (With minimum_range 5 the first rings of a ring-major scan are dropped
 * entirely: the first kept point can be tens of thousands of points in.) */
/* This path serves calls of a few scans, where the chain of dependent rounds is what costs: 1024 threads x 16 points per round
 * (with minimum_range 5 the first kept point of a 64-ring scan is ~20 k points in: two rounds instead of twenty) */
#define LL_FK_THREADS 1024
#define LL_FK_PER 16
__global__ __launch_bounds__(LL_FK_THREADS) void k_first_kept(LLView V, int first, int count)
{
    if ((int)blockIdx.x >= count) return;
    const int s = first + blockIdx.x;
    const int n_in = V.n_in[s];
    const int tid = threadIdx.x;
    const LLRawAny raw = {ll_raw_base(V, s), V.raw_stride == 3};
    __shared__ int sh_first;
    if (tid == 0) sh_first = INT_MAX;
    __syncthreads();
    for (int c = 0; c < n_in; c += LL_FK_THREADS * LL_FK_PER) {
        float4 p[LL_FK_PER];
#pragma unroll
        for (int k = 0; k < LL_FK_PER; ++k) { const int i = c + k * LL_FK_THREADS + tid; if (i < n_in) p[k] = raw[i]; }
        int mine = INT_MAX;
#pragma unroll
        for (int k = LL_FK_PER - 1; k >= 0; --k) {
            const int i = c + k * LL_FK_THREADS + tid;
            if (i < n_in && ll_keep(p[k], V.thres)) mine = i;
        }
        if (__syncthreads_or(mine != INT_MAX ? 1 : 0)) {
            if (mine != INT_MAX) atomicMin(&sh_first, mine);
            break;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int fk = sh_first;
        float start_ori = 0.0f;
        if (fk != INT_MAX) { const float4 p0 = raw[fk]; start_ori = -ll_atan2f(p0.y, p0.x); }
        V.hdr[s].start_ori = start_ori; V.hdr[s].first_kept = fk;
        V.hdr[s].so_lo_up = ll_f32_ceil((double)start_ori - M_PI / 2);
        V.hdr[s].so_hi_dn = ll_f32_floor((double)start_ori + M_PI * 3 / 2);
    }
}

template <bool LUT>
__global__ __launch_bounds__(LL_BLOCK) void k_classify(LLView V, int first, int count)
{
    int sl, tile;
    if (!ll_xcd_map(blockIdx.x, V.T, count, sl, tile)) return;
    const int s = first + sl;
    const int n_in = V.n_in[s];
    const int base = tile * LL_TILE;
    if (base >= n_in) return;
    const int tid = threadIdx.x;
    const LLRawAny raw = {ll_raw_base(V, s), V.raw_stride == 3};

    __shared__ int sh_first_p, sh_fk, sh_lk;
    __shared__ int hist[LL_MAX_RINGS];
    __shared__ int thr[LL_MAX_RINGS + 2];
    __shared__ int lut[LUT ? LL_RING_LUT_MAX : 1];
    if (tid <= V.R) thr[tid] = V.ring_thr[tid];
    if (tid == 0) { sh_first_p = INT_MAX; sh_fk = INT_MAX; sh_lk = -1; thr[V.R + 1] = INT_MAX; }
    if (tid < LL_MAX_RINGS) hist[tid] = 0;
    if (LUT) for (int i = tid; i < V.lut_nb; i += LL_BLOCK) lut[i] = V.ring_lut[i];
    __syncthreads();
    const ScanHdr h0 = V.hdr[s];                                      /* k_first_kept */
    const float start_ori = h0.start_ori, so_lo_up = h0.so_lo_up, so_hi_dn = h0.so_hi_dn;

    float *ori = V.ori + (size_t)sl * V.NP;                            /* scratch: by position in the launch */
    int8_t *ring = V.ring + (size_t)sl * V.NP;
    const int lane = tid & 63;
    int bits = 0; while ((1 << bits) < V.R) ++bits;
#pragma unroll
    for (int k = 0; k < LL_TILE / LL_BLOCK; ++k) {
        const int i = base + k * LL_BLOCK + tid;
        const bool in = i < n_in;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) p = raw[i];
        const bool kept = in && ll_keep(p, V.thres);
        int id = -1; float o = 0.0f; bool firstp = false;
        if (kept) {
            o = -ll_atan2f_finite(p.y, p.x);                                          /* :177 (also start/endOri source) */
            id = ll_scan_id<LUT>(thr, lut, V.lut_nb, V.lut_t0, V.lut_scale, V.R, p);
            if (id >= 0) {
                /* the !halfPassed branch (:180-192) evaluated as if the flag were still false; the float-against-double
                 * comparisons in f32 against the rounded-outward bounds of k_first_kept (ll_f32_ceil / ll_f32_floor) */
                float a = o;
                const bool below = a < so_lo_up, above = a > so_hi_dn;
                if (below || above) a = (float)((double)a + (below ? 2 * M_PI : -(2 * M_PI)));
                firstp = (a - start_ori) >= 3.14159274101257324f;                     /* (double)(a - startOri) > M_PI */
            }
        }
        if (in) { ori[i] = o; ring[i] = (int8_t)id; }
        /* wave-level reductions by ballot: lanes hold consecutive indices, so first/last set lane = min/max index.
         * (per-lane LDS atomics on one address serialise 64-way; ring-major input puts a whole wave on one bin) */
        const int i0 = base + k * LL_BLOCK + (tid & ~63);
        const unsigned long long km = __ballot(kept);
        if (km && lane == 0) { atomicMin(&sh_fk, i0 + __ffsll((long long)km) - 1); atomicMax(&sh_lk, i0 + 63 - __clzll((long long)km)); }
        const unsigned long long pm = __ballot(firstp);
        if (pm && lane == 0) atomicMin(&sh_first_p, i0 + __ffsll((long long)pm) - 1);
        const unsigned long long m = __ballot(id >= 0);
        const int id0 = __builtin_amdgcn_readfirstlane(id);
        if (__ballot(id != id0) == 0ull) {                                             /* ring-major input: one ring per wave */
            if (id0 >= 0 && lane == 0) atomicAdd(&hist[id0], __popcll(m));
        } else if (m) {
            unsigned mlo, mhi;
            ll_match_any(id, bits, m, mlo, mhi);
            if (id >= 0 && ll_match_rank(mlo, mhi) == 0) atomicAdd(&hist[id], ll_match_count(mlo, mhi));   /* one add per distinct ring */
        }
    }
    __syncthreads();
    const size_t tb = ((size_t)sl * V.T + tile);
    if (tid < V.R) V.tile_hist[tb * V.R + tid] = hist[tid];
    if (tid == 0) {
        V.tile_first_p[tb] = sh_first_p;
        V.tile_first_kept[tb] = sh_fk;
        V.tile_last_kept[tb] = sh_lk;
    }
}

__global__ __launch_bounds__(LL_BLOCK) void k_offsets(LLView V, int first, int count)
{
    const int s = first + blockIdx.x;
    if (blockIdx.x >= count) return;
    const int tid = threadIdx.x;
    const int n_in = V.n_in[s];
    const int nt = (n_in + LL_TILE - 1) / LL_TILE;
    __shared__ int ring_cnt[LL_MAX_RINGS + 1];
    __shared__ int ring_off[LL_MAX_RINGS + 1];
    __shared__ int sh_first_p, sh_lk;
    if (tid == 0) { sh_first_p = INT_MAX; sh_lk = -1; }
    __syncthreads();
    const size_t tb = (size_t)blockIdx.x * V.T;
    constexpr int TU = 8;                                   /* tile counters in flight per thread: the loops are latency chains */
    if (tid < V.R) {
        int run = 0;
        for (int t = 0; t < nt; t += TU) {
            int hh[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) hh[u] = (t + u < nt) ? V.tile_hist[(tb + t + u) * V.R + tid] : 0;
#pragma unroll
            for (int u = 0; u < TU; ++u) run += hh[u];
        }
        ring_cnt[tid] = run;
    }
    int fp = INT_MAX, lk = -1;
    for (int t = tid; t < nt; t += LL_BLOCK) {
        fp = min(fp, V.tile_first_p[tb + t]);
        lk = max(lk, V.tile_last_kept[tb + t]);
    }
    if (fp != INT_MAX) atomicMin(&sh_first_p, fp);
    if (lk >= 0) atomicMax(&sh_lk, lk);
    __syncthreads();
    if (tid == 0) {
        int run = 0, mx = 0;
        for (int r = 0; r < V.R; ++r) { ring_off[r] = run; run += ring_cnt[r]; mx = max(mx, ring_cnt[r]); }
        ring_off[V.R] = run;
        if (mx > 2304 && mx <= V.max_ring && n_in > 0 && sh_lk >= 0) for (int r = 0; r < V.R; ++r) ll_tier_append(V, s, r, ring_cnt[r]);
        ScanHdr h = V.hdr[s];
        h.n = run; h.max_ring = mx; h.half_idx = sh_first_p; h.last_kept = sh_lk;
        h.n_sharp = h.n_less_sharp = h.n_flat = h.n_less_flat = 0;
        h.status = 0; h.lf_strided = 1;
        if (n_in <= 0 || sh_lk < 0) { h.status = -5; h.n = 0; }                    /* LL_ERR_EMPTY */
        else {
            /* endOri (:115-126): -atan2f(last kept) + 2*pi in f64, stored f32, then the 3*pi / pi adjustment */
            const float so = h.start_ori;
            float eo = (float)((double)V.ori[(size_t)blockIdx.x * V.NP + sh_lk] + 2 * M_PI);
            if ((double)(eo - so) > 3 * M_PI)     eo = (float)((double)eo - 2 * M_PI);
            else if ((double)(eo - so) < M_PI)    eo = (float)((double)eo + 2 * M_PI);
            h.end_ori = eo;
            if (mx > V.max_ring) h.status = -4;                                      /* LL_ERR_CAPACITY */
        }
        V.hdr[s] = h;
    }
    __syncthreads();
    if (tid <= V.R) V.ring_off[(size_t)s * (V.R + 1) + tid] = ring_off[tid];
    if (tid < V.R) {
        int run = 0;                                            /* positions inside the ring: the rings sit at a fixed stride */
        for (int t = 0; t < nt; t += TU) {
            int hh[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) hh[u] = (t + u < nt) ? V.tile_hist[(tb + t + u) * V.R + tid] : 0;
#pragma unroll
            for (int u = 0; u < TU; ++u)
                if (t + u < nt) { V.tile_base[(tb + t + u) * V.R + tid] = run; run += hh[u]; }
        }
    }
}

__global__ __launch_bounds__(LL_BLOCK) void k_scatter(LLView V, int first, int count)
{
    int sl, tile;
    if (!ll_xcd_map(blockIdx.x, V.T, count, sl, tile)) return;
    const int s = first + sl;
    const int n_in = V.n_in[s];
    const int base = tile * LL_TILE;
    if (base >= n_in) return;
    const ScanHdr h = V.hdr[s];
    if (h.status == -5) return;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NK = LL_TILE / LL_BLOCK, NW = LL_BLOCK / 64;
    __shared__ int cnt[NK * NW][LL_MAX_RINGS];
    for (int i = tid; i < NK * NW * LL_MAX_RINGS; i += LL_BLOCK) (&cnt[0][0])[i] = 0;
    __syncthreads();

    const LLRawAny raw = {ll_raw_base(V, s), V.raw_stride == 3};
    const float *ori = V.ori + (size_t)sl * V.NP;
    const int8_t *ring = V.ring + (size_t)sl * V.NP;
    int bits = 0; while ((1 << bits) < V.R) ++bits;

    int my_ring[NK], my_rank[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i = base + k * LL_BLOCK + tid;
        const int r = (i < n_in) ? (int)ring[i] : -1;
        unsigned mlo, mhi;
        ll_match_any(r, bits, __ballot(r >= 0), mlo, mhi);
        my_ring[k] = r;
        my_rank[k] = ll_match_rank(mlo, mhi);
        if (r >= 0 && my_rank[k] == 0) cnt[k * NW + wave][r] = ll_match_count(mlo, mhi);
    }
    __syncthreads();
    if (tid < V.R) {
        int run = V.tile_base[((size_t)sl * V.T + tile) * V.R + tid];
        for (int kw = 0; kw < NK * NW; ++kw) { const int c = cnt[kw][tid]; cnt[kw][tid] = run; run += c; }
    }
    __syncthreads();
    float4 *cloud = V.cloud + (size_t)s * V.CS;
    const int cap = V.ring_cap;
    const float so = h.start_ori, eo = h.end_ori;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int r = my_ring[k];
        if (r < 0) continue;
        const int i = base + k * LL_BLOCK + tid;
        float o = ori[i];
        if (i <= h.half_idx) {                                                          /* !halfPassed (:178-193) */
            if ((double)o < (double)so - M_PI / 2)            o = (float)((double)o + 2 * M_PI);
            else if ((double)o > (double)so + M_PI * 3 / 2)   o = (float)((double)o - 2 * M_PI);
        } else {                                                                        /* :194-205 */
            o = (float)((double)o + 2 * M_PI);
            if ((double)o < (double)eo - M_PI * 3 / 2)        o = (float)((double)o + 2 * M_PI);
            else if ((double)o > (double)eo + M_PI / 2)       o = (float)((double)o - 2 * M_PI);
        }
        const float rel = (o - so) / (eo - so);                                         /* :207 */
        const float4 p = raw[i];
        const int pos = cnt[k * NW + wave][r] + my_rank[k];
        if (pos < cap) cloud[(size_t)r * cap + pos] = make_float4(p.x, p.y, p.z, (float)((double)r + 0.1 * (double)rel)); /* :208 */
    }
}

/* laserCloud as the reference holds it -- ring after ring without gaps -- for the C ABI (ll_download_cloud) */
__global__ __launch_bounds__(LL_BLOCK) void k_cloud_flatten(LLView V, int slot, float4 *dst)
{
    const int r = blockIdx.x;
    const int *ring_off = V.ring_off + (size_t)slot * (V.R + 1);
    const int off = ring_off[r], n = min(ring_off[r + 1] - off, V.ring_cap);
    const float4 *src = V.cloud + (size_t)slot * V.CS + (size_t)r * V.ring_cap;
    for (int i = threadIdx.x; i < n; i += LL_BLOCK) dst[off + i] = src[i];
}

void ll_launch_cloud_flatten(const LLView &V, int slot, float4 *dst, hipStream_t st)
{
    hipLaunchKernelGGL(k_cloud_flatten, dim3(V.R), dim3(LL_BLOCK), 0, st, V, slot, dst);
}

/* the less-flat cloud as the reference publishes it -- ring after ring without gaps -- for the C ABI (ll_download_features) and the
 * mapping stage (ll_cubemap_process_slot): the rows of an extracted slot closed up through lf_pre; an uploaded slot is contiguous already */
__global__ __launch_bounds__(LL_BLOCK) void k_lflat_flatten(LLView V, int slot, float4 *dst)
{
    const int r = blockIdx.x;
    const float4 *lf = V.lflat + (size_t)slot * V.LFS;
    if (!V.hdr[slot].lf_strided) {
        const int n = V.hdr[slot].n_less_flat;
        for (int i = blockIdx.x * LL_BLOCK + threadIdx.x; i < n; i += gridDim.x * LL_BLOCK) dst[i] = lf[i];
        return;
    }
    const int *pre = V.lf_pre + (size_t)slot * (V.R + 1);
    const int o = pre[r], n = pre[r + 1] - o;
    for (int i = threadIdx.x; i < n; i += LL_BLOCK) dst[o + i] = lf[(size_t)r * V.ring_cap + i];
}

void ll_launch_lflat_flatten(const LLView &V, int slot, float4 *dst, hipStream_t st)
{
    hipLaunchKernelGGL(k_lflat_flatten, dim3(V.R), dim3(LL_BLOCK), 0, st, V, slot, dst);
}

/* ll_debug_exact_math: the device arithmetic of a1 on caller-supplied operands (see include/lightloam_hip.h) */
__global__ __launch_bounds__(LL_BLOCK) void k_debug_exact_math(LLView V, int op, const float *a, const float *b, const float *c, int n, float *out)
{
    __shared__ int thr[LL_MAX_RINGS + 2];
    __shared__ int lut[LL_RING_LUT_MAX];
    const int tid = threadIdx.x;
    if (tid <= V.R) thr[tid] = V.ring_thr[tid];
    if (tid == 0) thr[V.R + 1] = INT_MAX;
    for (int i = tid; i < V.lut_nb; i += LL_BLOCK) lut[i] = V.ring_lut[i];
    __syncthreads();
    for (int i = blockIdx.x * LL_BLOCK + tid; i < n; i += gridDim.x * LL_BLOCK) {
        const float x = a[i], y = b ? b[i] : 0.0f, z = c ? c[i] : 0.0f;
        float r = 0.0f;
        switch (op) {
        case 0: r = ll_atanf(x); break;
        case 1: r = ll_atan2f(x, y); break;
        case 2: r = ll_atan2f_finite(x, y); break;
        case 3: r = ll_div_pi_f32(x); break;
        case 4: r = x / sqrtf(y * y + z * z); break;
        case 5: {
            const float4 p = make_float4(y, z, x, 0.0f);
            const int id = V.lut_nb > 0 ? ll_scan_id<true>(thr, lut, V.lut_nb, V.lut_t0, V.lut_scale, V.R, p)
                                        : ll_scan_id<false>(thr, lut, 0, 0.0f, 0.0f, V.R, p);
            r = __int_as_float(id); break;
        }
        default: {
            const float t = x / sqrtf(y * y + z * z);
            int id = ll_ring_of_t(t, V.ring_model, V.R, V.lower_bound, V.factor);
            if (id > V.R - 1 || id < 0) id = -1;
            r = __int_as_float(id); break;
        }
        }
        out[i] = r;
    }
}

void ll_launch_debug_exact_math(const LLView &V, int op, const float *a, const float *b, const float *c, int n, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_debug_exact_math, dim3(min(2048, (n + LL_BLOCK - 1) / LL_BLOCK)), dim3(LL_BLOCK), 0, st, V, op, a, b, c, n, out);
}

void ll_launch_organize(const LLView &V, int first, int count, hipStream_t st, LLProfiler *prof)
{
    if (V.tier_list) (void)hipMemsetAsync(V.tier_cnt, 0, 4 * sizeof(int), st);       /* the work lists of this extract call's long rings start empty */
    if (count > V.org_small) {
        ll_prof_mark(prof, LL_K_ORGANIZE, st);
        const bool s3 = V.raw_stride == 3;
        if (V.lut_nb > 0) { if (s3) hipLaunchKernelGGL((k_organize<true, true>), dim3(count), dim3(LL_BLOCK), 0, st, V, first, count);
                            else hipLaunchKernelGGL((k_organize<true, false>), dim3(count), dim3(LL_BLOCK), 0, st, V, first, count); }
        else { if (s3) hipLaunchKernelGGL((k_organize<false, true>), dim3(count), dim3(LL_BLOCK), 0, st, V, first, count);
               else hipLaunchKernelGGL((k_organize<false, false>), dim3(count), dim3(LL_BLOCK), 0, st, V, first, count); }
        ll_prof_mark(prof, LL_K_END, st);
        return;
    }
    const int groups = (count + 7) / 8;
    const int grid = 8 * V.T * groups;
    ll_prof_mark(prof, LL_K_FIRST, st);
    hipLaunchKernelGGL(k_first_kept, dim3(count), dim3(LL_FK_THREADS), 0, st, V, first, count);
    ll_prof_mark(prof, LL_K_CLASSIFY, st);
    if (V.lut_nb > 0) hipLaunchKernelGGL(k_classify<true>, dim3(grid), dim3(LL_BLOCK), 0, st, V, first, count);
    else hipLaunchKernelGGL(k_classify<false>, dim3(grid), dim3(LL_BLOCK), 0, st, V, first, count);
    ll_prof_mark(prof, LL_K_OFFSETS, st);
    hipLaunchKernelGGL(k_offsets, dim3(count), dim3(LL_BLOCK), 0, st, V, first, count);
    ll_prof_mark(prof, LL_K_SCATTER, st);
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(LL_BLOCK), 0, st, V, first, count);
    ll_prof_mark(prof, LL_K_END, st);
}
