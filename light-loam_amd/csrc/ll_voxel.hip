/*
 * ll_voxel.hip -- pcl::VoxelGrid<PointXYZI>::filter for whole clouds of any size, many clouds per call.
 * Replaces downSizeFilterCorner / downSizeFilterSurf .filter() of /root/reference/src/laserMapping.cpp (:1813-1821 on
 * the scan, :2151-2165 on every cube of the 5 x 5 x 3 neighbourhood); the per-ring filter of scanRegistration stays in
 * k_ring_features.
 *
 * The clouds ("segments") lie back to back in one array.  Per segment PCL computes the bounding box, the voxel index
 * of every point ((floor(x/leaf) - min_b) . (1, div_x, div_x div_y)), sorts by index, and emits one centroid (x, y, z,
 * intensity: f32 sums divided by the count) per distinct index, in index order.  Here:
 *   k_vx_segid / k_vx_bbox / k_vx_params / k_vx_keys   segment of every point, bounding boxes by ordered-int atomics,
 *                       PCL's box arithmetic per segment, 64-bit keys (segment << 32 | voxel index)
 *   ll_sort_pairs       device-wide stable LSD radix sort (8-bit digits; tile histograms -> scan -> stable scatter with
 *                       wave match-any ranking); digits that are equal for all keys are skipped
 *   k_vx_heads / scan / k_vx_centroid   run heads -> output rank -> sums in INPUT order (the sort is stable; PCL's
 *                       std::sort leaves the order inside a voxel unspecified, the oracle defines input order too)
 * "Leaf size is too small" (more than INT_MAX voxels): the segment is passed through unchanged, as in PCL.
 */
#include "ll_common.h"
#include <limits.h>
#include <string.h>
#include <algorithm>
#include <stdint.h>

#define LL_VB 256

__global__ __launch_bounds__(256) void k_copy_words(unsigned *dst, const unsigned *src, size_t nwords)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy_quads(uint4 *dst, const uint4 *src, size_t nquads)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void k_fill_words(int *dst, int n, int a, int b, int split)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i < split ? a : b;
}

/* Counts read back to size the next launch land in page-locked memory: a copy into pageable memory is staged and costs
 * the host ~26 us before the synchronisation even starts. */
void *ll_pinned_scratch(size_t bytes)
{
    static thread_local void *buf = nullptr;
    static thread_local size_t cap = 0;
    if (bytes > cap) {
        if (buf) (void)hipHostFree(buf);
        buf = nullptr; cap = 0;
        const size_t want = bytes < 32768 ? 32768 : bytes;
        if (hipHostMalloc(&buf, want, hipHostMallocPortable) != hipSuccess) { buf = nullptr; return nullptr; }
        cap = want;
    }
    return buf;
}

int ll_read_back(void *host_dst, const void *dev_src, size_t bytes, hipStream_t st)
{
    void *pin = ll_pinned_scratch(bytes);
    hipError_t e = hipMemcpyAsync(pin ? pin : host_dst, dev_src, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && pin) memcpy(host_dst, pin, bytes);
    return e == hipSuccess ? 0 : -1;
}

void ll_copy_d2d(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return;
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0) {
        const size_t nq = bytes / 16;
        hipLaunchKernelGGL(k_copy_quads, dim3((unsigned)std::min<size_t>(2048, (nq + 255) / 256)), dim3(256), 0, st, (uint4 *)dst, (const uint4 *)src, nq);
    } else if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 3) == 0) {
        const size_t nw = bytes / 4;
        hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<size_t>(2048, (nw + 255) / 256)), dim3(256), 0, st, (unsigned *)dst, (const unsigned *)src, nw);
    } else (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
}

void ll_fill_words(int *dst, int n, int a, int b, int split, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_fill_words, dim3((n + 63) / 64), dim3(64), 0, st, dst, n, a, b, split);
}

__device__ __forceinline__ int ll_vx_f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ll_vx_ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

/* seg_off[nseg + 1]: ascending input offsets; segid[i] = the segment that holds point i */
__global__ __launch_bounds__(LL_VB) void k_vx_segid(const int *seg_off, int nseg, int n, int *segid, int *bbox)
{
    const int i = blockIdx.x * LL_VB + threadIdx.x;
    if (i < nseg * 6) bbox[i] = (i % 6 < 3) ? INT_MAX : INT_MIN;
    if (i >= n) return;
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid - 1; }
    segid[i] = lo;
}

__global__ __launch_bounds__(LL_VB) void k_vx_bbox(const float4 *pts, const int *segid, int n, int *bbox)
{
    const int i = blockIdx.x * LL_VB + threadIdx.x;
    const bool in = i < n;
    const float4 p = in ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int s = in ? segid[i] : -1;
    /* most waves hold one segment: reduce in the wave; most workgroups do too: then one set of atomics per workgroup
     * (a segment's box is six words that every one of its points would otherwise hit) */
    const int wave = threadIdx.x >> 6;
    const int s0 = __builtin_amdgcn_readfirstlane(s);
    const bool uniform = __ballot(s != s0 && in) == 0ull && s0 >= 0;
    __shared__ float red[LL_VB / 64][6];
    __shared__ int wseg[LL_VB / 64];
    float mn[3] = {in ? p.x : INFINITY, in ? p.y : INFINITY, in ? p.z : INFINITY};
    float mx[3] = {in ? p.x : -INFINITY, in ? p.y : -INFINITY, in ? p.z : -INFINITY};
    if (uniform) {
        for (int o = 32; o > 0; o >>= 1)
            for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; ++k) { red[wave][k] = mn[k]; red[wave][3 + k] = mx[k]; }
    } else if (in) {
        atomicMin(&bbox[s * 6 + 0], ll_vx_f2ord(p.x)); atomicMin(&bbox[s * 6 + 1], ll_vx_f2ord(p.y)); atomicMin(&bbox[s * 6 + 2], ll_vx_f2ord(p.z));
        atomicMax(&bbox[s * 6 + 3], ll_vx_f2ord(p.x)); atomicMax(&bbox[s * 6 + 4], ll_vx_f2ord(p.y)); atomicMax(&bbox[s * 6 + 5], ll_vx_f2ord(p.z));
    }
    if ((threadIdx.x & 63) == 0) wseg[wave] = uniform ? s0 : -1;      /* -1: nothing left to merge for this wave */
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        /* merge the waves that hold the same segment as an earlier wave into that wave's slot, then one atomic per distinct segment */
        for (int w = 0; w < LL_VB / 64; ++w) {
            const int sw = wseg[w];
            if (sw < 0) continue;
            bool first = true;
            for (int u = 0; u < w; ++u) if (wseg[u] == sw) first = false;
            if (!first) continue;
            float v = red[w][k];
            for (int u = w + 1; u < LL_VB / 64; ++u) if (wseg[u] == sw) v = k < 3 ? fminf(v, red[u][k]) : fmaxf(v, red[u][k]);
            if (k < 3) atomicMin(&bbox[sw * 6 + k], ll_vx_f2ord(v)); else atomicMax(&bbox[sw * 6 + k], ll_vx_f2ord(v));
        }
    }
}

/* per segment: min_b[3] as floats, mul1, mul2, too_small  (voxel_grid.hpp applyFilter, PCL 1.10) */
struct LLVoxSeg { float fb[3]; int mul1, mul2, too_small; };

__global__ void k_vx_params(const int *bbox, const int *seg_off, int nseg, float inv, LLVoxSeg *sp)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    LLVoxSeg o;
    o.fb[0] = o.fb[1] = o.fb[2] = 0.0f; o.mul1 = 1; o.mul2 = 1; o.too_small = 0;
    if (seg_off[s + 1] > seg_off[s]) {
        long long d[3]; int min_b[3], div_b[3];
        for (int c = 0; c < 3; ++c) {
            const float mn = ll_vx_ord2f(bbox[s * 6 + c]), mx = ll_vx_ord2f(bbox[s * 6 + 3 + c]);
            d[c] = (long long)((mx - mn) * inv) + 1;
            min_b[c] = (int)floorf(mn * inv);
            div_b[c] = (int)floorf(mx * inv) - min_b[c] + 1;
        }
        /* ... or where PCL's int voxel index would wrap (undefined in the reference): the same exit in ll_features.hip and in the CPU checker under tests */
        o.too_small = d[0] * d[1] * d[2] > (long long)INT_MAX || (long long)div_b[0] * div_b[1] * div_b[2] > 0xffffffffLL;
        o.mul1 = div_b[0]; o.mul2 = div_b[0] * div_b[1];
        for (int c = 0; c < 3; ++c) o.fb[c] = (float)min_b[c];
    }
    sp[s] = o;
}

__global__ __launch_bounds__(LL_VB) void k_vx_keys(const float4 *pts, const int *segid, const int *seg_off, const LLVoxSeg *sp, int n, float inv,
                                                  unsigned long long *keys, int *vals, unsigned long long *pub, int *seg_count, int nseg)
{
    const int i = blockIdx.x * LL_VB + threadIdx.x;
    if (i < LL_VX_FUSED_WGS) pub[i] = 0ull;                          /* k_vx_finish's hand-over words: nothing published yet */
    for (int j = i; j < nseg; j += gridDim.x * LL_VB) seg_count[j] = 0;   /* ... and its per-segment counters */
    if (i >= n) return;
    const int s = segid[i];
    const LLVoxSeg P = sp[s];
    const float4 p = pts[i];
    unsigned idx;
    if (P.too_small) idx = (unsigned)(i - seg_off[s]);
    else {
        const int i0 = (int)(floorf(p.x * inv) - P.fb[0]);
        const int i1 = (int)(floorf(p.y * inv) - P.fb[1]);
        const int i2 = (int)(floorf(p.z * inv) - P.fb[2]);
        idx = (unsigned)(i0 + i1 * P.mul1 + i2 * P.mul2);
    }
    keys[i] = ((unsigned long long)(unsigned)s << 32) | idx;
    vals[i] = i;
}

/* ------------------------------------------------------------------ device-wide stable radix sort of (u64 key, i32 value) */
#define LL_RS_TILE 4096           /* keys per workgroup: 4 rows of 1024 */
#define LL_RS_ROWS 4

__global__ __launch_bounds__(LL_VB) void k_rs_or_and(const unsigned long long *keys, int n, unsigned long long *or_and)
{
    unsigned long long o = 0ull, a = ~0ull;
    for (int i = blockIdx.x * LL_VB + threadIdx.x; i < n; i += gridDim.x * LL_VB) { const unsigned long long k = keys[i]; o |= k; a &= k; }
    for (int s = 32; s > 0; s >>= 1) { o |= __shfl_xor(o, s); a &= __shfl_xor(a, s); }
    if ((threadIdx.x & 63) == 0) { atomicOr(&or_and[0], o); atomicAnd(&or_and[1], a); }
}

/* 8-bit digits (a pass is five launches: histogram, three for the scan, scatter).  A tile is 16 waves x 256
 * consecutive keys; a wave ranks its four rows one after the other into its own 256 counters (its LDS operations execute in
 * order), the rank inside a row is a wave match-any, and one scan of the digit-major (digit, wave) table turns the counters
 * into offsets inside the tile's share of every digit -- the scheme of k_ring_features' voxel sort. */
__global__ __launch_bounds__(1024) void k_rs_hist8(const unsigned long long *keys, int n, int shift, int nblk, int *hist /* [256][nblk] */)
{
    __shared__ int h[256];
    const int tid = threadIdx.x;
    if (tid < 256) h[tid] = 0;
    __syncthreads();
    const int base = blockIdx.x * LL_RS_TILE;
#pragma unroll
    for (int r = 0; r < LL_RS_ROWS; ++r) {
        const int i = base + r * 1024 + tid;
        if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1);
    }
    __syncthreads();
    if (tid < 256) hist[tid * nblk + blockIdx.x] = h[tid];
}

__global__ __launch_bounds__(1024) void k_rs_scatter8(const unsigned long long *keys, const int *vals, int n, int shift, int nblk,
                                                      const int *hist_scanned, unsigned long long *keys_out, int *vals_out)
{
    __shared__ int cnt[16][256];                     /* [wave][digit] */
    __shared__ int sc[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16 * 256; i += 1024) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * LL_RS_TILE + wave * (LL_RS_ROWS * 64);       /* the wave's 256 consecutive keys */
    unsigned long long k[LL_RS_ROWS]; int v[LL_RS_ROWS], rnk[LL_RS_ROWS];
    int *wc = cnt[wave];
#pragma unroll
    for (int r = 0; r < LL_RS_ROWS; ++r) {
        const int i = base + r * 64 + lane;
        const bool in = i < n;
        k[r] = in ? keys[i] : ~0ull; v[r] = in ? vals[i] : 0;
        const int d = (int)((k[r] >> shift) & 255ull);
        unsigned mlo, mhi;
        ll_match_any(d, 8, __ballot(in), mlo, mhi);
        const int rk = ll_match_rank(mlo, mhi);
        const int pre = in ? wc[d] : 0;                                          /* earlier rows of this wave with digit d */
        rnk[r] = pre + rk;
        if (in && rk == 0) wc[d] = pre + ll_match_count(mlo, mhi);
    }
    __syncthreads();
    /* exclusive scan of the table in digit-major order: thread t owns entries 4t .. 4t+3 of [digit][wave] */
    int e[4]; int sum = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int idx = 4 * tid + u; e[u] = cnt[idx & 15][idx >> 4]; sum += e[u]; }
    int total;
    int run = ll_block_exscan_n<16>(sum, sc, total);
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int idx = 4 * tid + u; cnt[idx & 15][idx >> 4] = run; run += e[u]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LL_RS_ROWS; ++r) {
        const int i = base + r * 64 + lane;
        if (i >= n) continue;
        const int d = (int)((k[r] >> shift) & 255ull);
        const int within = cnt[wave][d] - cnt[0][d] + rnk[r];                   /* earlier waves' keys of this digit, then this wave's */
        const int pos = hist_scanned[d * nblk + blockIdx.x] + within;
        keys_out[pos] = k[r]; vals_out[pos] = v[r];
    }
}

/* the same sort for n <= 8192 pairs in ONE launch: one 1024-thread workgroup, keys and values in LDS between the
 * passes, digits that do not vary found on the device (no host read-back).  112 KB of dynamic LDS. */
#define LL_RSS_ROWS 8
#define LL_RSS_MAX (LL_RSS_ROWS * 1024)
extern __shared__ __attribute__((aligned(16))) unsigned char ll_rss_smem[];
/* seg_off != null: workgroup b sorts the pairs [seg_off[b], seg_off[b + 1]) on their own -- keys whose high part is the
 * segment number are then in global order too (the voxel filter of all valid cubes: 75 small sorts in one launch instead of the
 * device-wide sort's five passes of five launches) */
__global__ __launch_bounds__(1024) void k_rs_small(unsigned long long *keys, int *vals, int n, const int *seg_off, int chunk)
{
    if (seg_off) { const int base = seg_off[blockIdx.x]; n = seg_off[blockIdx.x + 1] - base; keys += base; vals += base; }
    else if (chunk > 0) { const int base = blockIdx.x * chunk; n = min(chunk, n - base); keys += base; vals += base; }   /* workgroup b: the b-th chunk on its own */
    unsigned long long *lk = (unsigned long long *)ll_rss_smem;           /* [LL_RSS_MAX] */
    int *lv = (int *)(lk + LL_RSS_MAX);                                     /* [LL_RSS_MAX] */
    int *cnt = lv + LL_RSS_MAX;                                             /* [16 waves][256 digits] */
    __shared__ int sc[16];
    __shared__ unsigned long long oa[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    /* Record g lives in wave g / (64 nrows), row (g / 64) % nrows, lane g % 64: (wave, row, lane) is the input order.  A wave ranks
     * its rows one after the other into its OWN 256 counters (its LDS operations execute in order, so a row reads what the rows
     * before it added), the rank inside a row is a wave match-any, and one scan of the digit-major (digit, wave) table turns
     * the counters into bases -- the scheme of k_ring_features' voxel sort: 8-bit digits, a third of the passes of a 4-bit
     * sort with per-row counters. */
    const int nrows = (n + 1023) >> 10;
    const int wbase = wave * nrows * 64;
    unsigned long long k[LL_RSS_ROWS]; int v[LL_RSS_ROWS];
    unsigned long long o = 0ull, a = ~0ull;
#pragma unroll
    for (int r = 0; r < LL_RSS_ROWS; ++r) {
        const int i = wbase + r * 64 + lane;
        k[r] = ~0ull; v[r] = 0;
        if (r < nrows && i < n) { k[r] = keys[i]; v[r] = vals[i]; o |= k[r]; a &= k[r]; }
    }
    if (tid == 0) { oa[0] = 0ull; oa[1] = ~0ull; }
    __syncthreads();
    for (int s_ = 32; s_ > 0; s_ >>= 1) { o |= __shfl_xor(o, s_); a &= __shfl_xor(a, s_); }
    if (lane == 0) { atomicOr(&oa[0], o); atomicAnd(&oa[1], a); }
    __syncthreads();
    const unsigned long long vary = oa[0] ^ oa[1];
    int *wc = cnt + wave * 256;
    for (int shift = 0; shift < 64; shift += 8) {
        if (((vary >> shift) & 255ull) == 0ull) continue;
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt[u * 1024 + tid] = 0;
        __syncthreads();
        int rnk[LL_RSS_ROWS];
#pragma unroll
        for (int r = 0; r < LL_RSS_ROWS; ++r) {
            rnk[r] = 0;
            if (r < nrows) {                                               /* the padding (all-ones keys) takes part and stays last */
                const int d = (int)((k[r] >> shift) & 255ull);
                unsigned mlo, mhi;
                ll_match_any(d, 8, ~0ull, mlo, mhi);
                const int rk = ll_match_rank(mlo, mhi);
                /* same digit, earlier rows of this wave: the read must see the adds of the rows before it -- an atomic load, so that
                 * neither the language's memory model nor a merged / hoisted load of a later row stands between the two */
                const int pre = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (rk == 0) atomicAdd(&wc[d], ll_match_count(mlo, mhi));
                rnk[r] = pre + rk;
            }
        }
        __syncthreads();
        {   /* exclusive scan of the (digit, wave) table in digit-major order: thread d < 256 owns digit d */
            int vv[16]; int s_ = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { vv[w] = (tid < 256) ? cnt[w * 256 + tid] : 0; s_ += vv[w]; }
            int total;
            int run = ll_block_exscan_n<16>(s_, sc, total);
            if (tid < 256) {
#pragma unroll
                for (int w = 0; w < 16; ++w) { cnt[w * 256 + tid] = run; run += vv[w]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < LL_RSS_ROWS; ++r)
            if (r < nrows) {
                const int d = (int)((k[r] >> shift) & 255ull);
                const int pos = wc[d] + rnk[r];
                lk[pos] = k[r]; lv[pos] = v[r];
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < LL_RSS_ROWS; ++r)
            if (r < nrows) { k[r] = lk[wbase + r * 64 + lane]; v[r] = lv[wbase + r * 64 + lane]; }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < LL_RSS_ROWS; ++r) {
        const int i = wbase + r * 64 + lane;
        if (r < nrows && i < n) { keys[i] = k[r]; vals[i] = v[r]; }
    }
}

/* chunks of `chunk` pairs, each sorted: pair i goes to its rank among all of them (see ll_sort_pairs).  The searches in the
 * other chunks advance together, one probe of every chunk per step: ~14 dependent round trips, not 14 per chunk. */
#define LL_RSM_MAXC 8
__global__ __launch_bounds__(256) void k_rs_merge(const unsigned long long *__restrict__ keys, const int *__restrict__ vals, int n, int chunk,
                                                  unsigned long long *__restrict__ out_keys, int *__restrict__ out_vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long K = keys[i];
    const int c = i / chunk;
    int lo[LL_RSM_MAXC], hi[LL_RSM_MAXC];
#pragma unroll
    for (int cc = 0; cc < LL_RSM_MAXC; ++cc) {
        lo[cc] = min(n, cc * chunk); hi[cc] = (cc == c) ? lo[cc] : min(n, lo[cc] + chunk);
    }
    for (int step = 0; step < 32; ++step) {
        bool open = false;
#pragma unroll
        for (int cc = 0; cc < LL_RSM_MAXC; ++cc) open = open || lo[cc] < hi[cc];
        if (!open) break;
        unsigned long long km[LL_RSM_MAXC];
#pragma unroll
        for (int cc = 0; cc < LL_RSM_MAXC; ++cc) if (lo[cc] < hi[cc]) km[cc] = keys[(lo[cc] + hi[cc]) >> 1];
#pragma unroll
        for (int cc = 0; cc < LL_RSM_MAXC; ++cc)
            if (lo[cc] < hi[cc]) {
                const int mid = (lo[cc] + hi[cc]) >> 1;
                const bool before = (cc < c) ? (km[cc] <= K) : (km[cc] < K);     /* does the probed pair go before pair i? */
                if (before) lo[cc] = mid + 1; else hi[cc] = mid;
            }
    }
    int rank = i - c * chunk;
#pragma unroll
    for (int cc = 0; cc < LL_RSM_MAXC; ++cc) if (cc != c) rank += lo[cc] - min(n, cc * chunk);
    out_keys[rank] = K; out_vals[rank] = vals[i];
}

__global__ __launch_bounds__(256) void k_rs_copy_pairs(const unsigned long long *__restrict__ sk, const int *__restrict__ sv, int n,
                                                       unsigned long long *__restrict__ dk, int *__restrict__ dv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { dk[i] = sk[i]; dv[i] = sv[i]; }
}

/* every segment [seg_off[b], seg_off[b + 1]) of (keys, vals) sorted on its own, nseg <= 65535 segments of at most LL_RSS_MAX pairs */
void ll_sort_pairs_segments(unsigned long long *keys, int *vals, const int *seg_off_dev, int nseg, hipStream_t st)
{
    if (nseg <= 0) return;
    const size_t lds = (size_t)LL_RSS_MAX * 12 + 16 * 256 * sizeof(int);
    static size_t attr_bytes[LL_MAX_DEVICES] = {0};
    ll_ensure_dynamic_lds(k_rs_small, lds, attr_bytes);
    hipLaunchKernelGGL(k_rs_small, dim3(nseg), dim3(1024), lds, st, keys, vals, 0, seg_off_dev, 0);
}

/* sorts (keys, vals) by keys, stable; the result is in (keys, vals) again.  tmp_* hold n elements, hist 16 * ceil(n/4096)
 * ints, tile_sum as for ll_device_exscan of that, or_and_host is pinned host memory for the varying-bit mask */
int ll_sort_pairs(unsigned long long *keys, int *vals, unsigned long long *tmp_keys, int *tmp_vals, int n, int *hist, int *tile_sum,
                  unsigned long long *or_and_dev, hipStream_t st)
{
    if (n <= 1) return 0;
    if (n <= LL_RSS_MAX) {
        const size_t lds = (size_t)LL_RSS_MAX * 12 + 16 * 256 * sizeof(int);
        static size_t attr_bytes[LL_MAX_DEVICES] = {0};
        ll_ensure_dynamic_lds(k_rs_small, lds, attr_bytes);
        hipLaunchKernelGGL(k_rs_small, dim3(1), dim3(1024), lds, st, keys, vals, n, (const int *)nullptr, 0);
        return 0;
    }
    if (n <= LL_RSM_MAXC * LL_RSS_MAX) {
        /* up to 64 k pairs (a scan's less-flat cloud is ~33 k): every chunk of 8192 sorted in LDS by its own workgroup, then
         * every pair placed by its rank among ALL chunks -- its place in its chunk plus, by binary search, the number of
         * pairs of the other chunks that go before it (<= K in the chunks before its own, < K in those behind: stable).
         * Three launches and no host read-back, against ~16 launches and one read-back for the device-wide passes. */
        const size_t lds = (size_t)LL_RSS_MAX * 12 + 16 * 256 * sizeof(int);
        static size_t attr_bytes[LL_MAX_DEVICES] = {0};
        ll_ensure_dynamic_lds(k_rs_small, lds, attr_bytes);
        const int nch = (n + LL_RSS_MAX - 1) / LL_RSS_MAX;
        hipLaunchKernelGGL(k_rs_small, dim3(nch), dim3(1024), lds, st, keys, vals, n, (const int *)nullptr, LL_RSS_MAX);
        hipLaunchKernelGGL(k_rs_merge, dim3((n + 255) / 256), dim3(256), 0, st, (const unsigned long long *)keys, (const int *)vals, n, LL_RSS_MAX, tmp_keys, tmp_vals);
        hipLaunchKernelGGL(k_rs_copy_pairs, dim3((n + 255) / 256), dim3(256), 0, st, (const unsigned long long *)tmp_keys, (const int *)tmp_vals, n, keys, vals);
        return 0;
    }
    unsigned long long oa[2];
    ll_fill_words((int *)or_and_dev, 4, 0, -1, 2, st);              /* {0, ~0}: the OR and the AND of all keys start here */
    hipLaunchKernelGGL(k_rs_or_and, dim3(min(1024, (n + LL_VB - 1) / LL_VB)), dim3(LL_VB), 0, st, keys, n, or_and_dev);
    /* a failed read-back must not leave the pass mask to chance: report it (the callers fail loudly) and sort on every digit */
    const int rb = ll_read_back(oa, or_and_dev, sizeof(oa), st);
    if (rb) { oa[0] = ~0ull; oa[1] = 0ull; }
    const unsigned long long vary = oa[0] ^ oa[1];                /* bits that differ between some two keys */
    const int nblk = (n + LL_RS_TILE - 1) / LL_RS_TILE;
    unsigned long long *ki = keys, *ko = tmp_keys; int *vi = vals, *vo = tmp_vals;
    for (int shift = 0; shift < 64; shift += 8) {
        if (((vary >> shift) & 255ull) == 0ull) continue;
        hipLaunchKernelGGL(k_rs_hist8, dim3(nblk), dim3(1024), 0, st, ki, n, shift, nblk, hist);
        ll_device_exscan(hist, 256 * nblk, tile_sum, st);
        hipLaunchKernelGGL(k_rs_scatter8, dim3(nblk), dim3(1024), 0, st, ki, vi, n, shift, nblk, hist, ko, vo);
        unsigned long long *tk = ki; ki = ko; ko = tk;
        int *tv = vi; vi = vo; vo = tv;
    }
    if (ki != keys) {
        ll_copy_d2d(keys, ki, (size_t)n * sizeof(unsigned long long), st);
        ll_copy_d2d(vals, vi, (size_t)n * sizeof(int), st);
    }
    return rb;
}

/* ------------------------------------------------------------------ runs -> centroids */
__global__ __launch_bounds__(LL_VB) void k_vx_heads(const unsigned long long *keys, int n, int nseg, int *flag, int *seg_count)
{
    const int i = blockIdx.x * LL_VB + threadIdx.x;
    if (i < nseg) seg_count[i] = 0;
    if (i >= n) return;
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(LL_VB) void k_vx_centroid(const float4 *pts, const unsigned long long *keys, const int *vals, const int *flag,
                                                      const int *rank, int n, float4 *out, int *seg_count)
{
    const int i = blockIdx.x * LL_VB + threadIdx.x;
    const bool head = i < n && flag[i];
    int seg = -1;
    if (head) {
        const unsigned long long key = keys[i];
        seg = (int)(key >> 32);
        /* the run ends at the next head: one dependent load per step (flag) instead of three (key, index, point) */
        int e = i + 1;
        while (e < n && !flag[e]) ++e;
        /* CentroidPoint<PointXYZI>: f32 sums from zero in input order, divided by float(n); four independent gathers in flight */
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
        for (int j = i; j < e; j += 4) {
            int v[4]; float4 p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j + k < e) v[k] = vals[j + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j + k < e) p[k] = pts[v[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j + k < e) { sx += p[k].x; sy += p[k].y; sz += p[k].z; si += p[k].w; }
        }
        const float fn = (float)(e - i);
        out[rank[i]] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
    }
    /* voxels per segment: the keys are sorted by segment, so a wave holds one or two of them -- one atomic per wave and
     * segment instead of one per voxel (tens of thousands of adds on a few dozen addresses serialise in the L2) */
    unsigned long long todo = __ballot(head);
    while (todo) {
        const int s0 = __shfl(seg, __ffsll((long long)todo) - 1);
        const unsigned long long same = __ballot(head && seg == s0);
        if ((threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&seg_count[s0], __popcll(same));
        todo &= ~same;
    }
}

/* k_vx_heads + the rank scan + k_vx_centroid in ONE launch for clouds of up to LL_VX_FUSED_WGS x LL_VB points (every cloud of the
 * mapping stage's frame): a workgroup counts the run heads of its 256 sorted keys, publishes the count (valid bit | count, one
 * 8-byte agent-scope atomic -- the data is the flag), sums the counts of the workgroups before it as they appear (they are
 * all resident: at most 256 workgroups) and writes its centroids at their ranks.  seg_count must be zero on entry. */
__global__ __launch_bounds__(LL_VB) void k_vx_finish(const float4 *pts, const unsigned long long *keys, const int *vals, int n, float4 *out,
                                                    int *seg_count, int *n_out_dev, unsigned long long *pub)
{
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const int i = b * LL_VB + tid;
    unsigned long long key = ~0ull;
    bool head = false;
    if (i < n) { key = keys[i]; head = (i == 0) || keys[i - 1] != key; }
    __shared__ int sc[LL_VB / 64];
    __shared__ int sbase;
    int total;
    const int lrank = ll_block_exscan_n<LL_VB / 64>(head ? 1 : 0, sc, total);
    if (tid == 0) __hip_atomic_store(&pub[b], (1ull << 63) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int before = 0;
    for (int q = tid; q < b; q += LL_VB) {
        unsigned long long w; int spins = 0;
        while (!((w = __hip_atomic_load(&pub[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63) && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(8);
        if (!(w >> 63)) __builtin_trap();                             /* a workgroup before this one never published: abort the launch (a HIP error at the next synchronisation), never a silently wrong cloud */
        before += (int)(w & 0xffffffffull);
    }
    before = ll_wave_sum_i32(before);
    if (tid == 0) sbase = 0;
    __syncthreads();
    if (lane == 0 && before) atomicAdd(&sbase, before);
    __syncthreads();
    const int base = sbase;
    int seg = -1;
    if (head) {
        seg = (int)(key >> 32);
        int e = i + 1;
        while (e < n && keys[e] == key) ++e;                          /* the run ends at the next head */
        /* CentroidPoint<PointXYZI>: f32 sums from zero in input order, divided by float(n); four independent gathers in flight */
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
        for (int j = i; j < e; j += 4) {
            int v[4]; float4 p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j + k < e) v[k] = vals[j + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j + k < e) p[k] = pts[v[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j + k < e) { sx += p[k].x; sy += p[k].y; sz += p[k].z; si += p[k].w; }
        }
        const float fn = (float)(e - i);
        out[base + lrank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
    }
    /* voxels per segment: one atomic per wave and segment (the keys are sorted by segment) */
    unsigned long long todo = __ballot(head);
    while (todo) {
        const int s0 = __shfl(seg, __ffsll((long long)todo) - 1);
        const unsigned long long same = __ballot(head && seg == s0);
        if (lane == __ffsll((long long)same) - 1) atomicAdd(&seg_count[s0], __popcll(same));
        todo &= ~same;
    }
    if (b == (int)gridDim.x - 1 && tid == 0) *n_out_dev = base + total;
}

/* workspace for up to cap points / max_seg segments: one device allocation carved into LLVoxWork (ll_common.h) */
size_t ll_vox_work_bytes(int cap, int max_seg)
{
    const size_t nblk = ((size_t)cap + LL_RS_TILE - 1) / LL_RS_TILE + 1;
    size_t b = 0;
    b += (size_t)cap * sizeof(int) * 5;                             /* segid, flag, rank, vals, tmp_vals */
    b += (size_t)cap * sizeof(unsigned long long) * 2;              /* keys, tmp_keys */
    b += (size_t)max_seg * (6 + 1 + 1) * sizeof(int) + sizeof(int); /* bbox, seg_off (+1), seg_count */
    b += (size_t)max_seg * sizeof(LLVoxSeg);
    b += 256 * nblk * sizeof(int) + (256 * nblk / 4096 + 2) * sizeof(int) + ((size_t)cap / 4096 + 2) * sizeof(int);
    b += 2 * sizeof(unsigned long long) + LL_VX_FUSED_WGS * sizeof(unsigned long long);
    return b + 8192;
}

void ll_vox_work_carve(void *base, int cap, int max_seg, LLVoxWork *W)
{
    unsigned char *p = (unsigned char *)base;
    auto take = [&](size_t bytes) { void *r = p; p += (bytes + 255) / 256 * 256; return r; };
    const size_t nblk = ((size_t)cap + LL_RS_TILE - 1) / LL_RS_TILE + 1;
    W->cap = cap; W->max_seg = max_seg;
    W->keys = (unsigned long long *)take((size_t)cap * 8); W->tmp_keys = (unsigned long long *)take((size_t)cap * 8);
    W->or_and = (unsigned long long *)take(16);
    W->pub = (unsigned long long *)take(LL_VX_FUSED_WGS * 8);
    W->segid = (int *)take((size_t)cap * 4); W->flag = (int *)take((size_t)cap * 4); W->rank = (int *)take((size_t)cap * 4 + 4);
    W->vals = (int *)take((size_t)cap * 4); W->tmp_vals = (int *)take((size_t)cap * 4);
    W->bbox = (int *)take((size_t)max_seg * 24); W->seg_off = (int *)take((size_t)(max_seg + 1) * 4); W->seg_count = (int *)take((size_t)max_seg * 4);
    W->sp = (LLVoxSeg *)take((size_t)max_seg * sizeof(LLVoxSeg));
    W->hist = (int *)take(256 * nblk * 4);
    W->tile_sum = (int *)take((256 * nblk / 4096 + 2 + (size_t)cap / 4096 + 2) * 4);
}

/* pts[0..n): nseg clouds back to back, seg_off (device, nseg + 1 ascending offsets, seg_off[nseg] = n).
 * out: the filtered clouds back to back in segment order; seg_count (W.seg_count, device): points per filtered cloud;
 * *n_out_dev (device int, may alias nothing else): total.  Everything is enqueued on st except the sort's one
 * host read-back of the varying key bits.  Returns non-zero when that read-back failed (a HIP error the caller reports). */
int ll_voxel_grid_segments(const float4 *pts, int n, int nseg, float leaf, const LLVoxWork &W, float4 *out, int *n_out_dev, hipStream_t st, int max_seg_len)
{
    const float inv = 1.0f / leaf;                                   /* inverse_leaf_size_ = Array4f::Ones() / leaf_size_ */
    const int nb = (max(n, nseg * 6) + LL_VB - 1) / LL_VB;
    hipLaunchKernelGGL(k_vx_segid, dim3(max(nb, 1)), dim3(LL_VB), 0, st, W.seg_off, nseg, n, W.segid, W.bbox);
    if (n <= 0) { (void)hipMemsetAsync(W.seg_count, 0, (size_t)nseg * sizeof(int), st); (void)hipMemsetAsync(n_out_dev, 0, sizeof(int), st); return 0; }
    hipLaunchKernelGGL(k_vx_bbox, dim3((n + LL_VB - 1) / LL_VB), dim3(LL_VB), 0, st, pts, W.segid, n, W.bbox);
    hipLaunchKernelGGL(k_vx_params, dim3((nseg + 63) / 64), dim3(64), 0, st, W.bbox, W.seg_off, nseg, inv, W.sp);
    hipLaunchKernelGGL(k_vx_keys, dim3((n + LL_VB - 1) / LL_VB), dim3(LL_VB), 0, st, pts, W.segid, W.seg_off, W.sp, n, inv, W.keys, W.vals, W.pub, W.seg_count, nseg);
    int rc = 0;
    if (nseg > 1 && max_seg_len > 0 && max_seg_len <= LL_RSS_MAX) ll_sort_pairs_segments(W.keys, W.vals, W.seg_off, nseg, st);   /* the keys lead with the segment */
    else rc = ll_sort_pairs(W.keys, W.vals, W.tmp_keys, W.tmp_vals, n, W.hist, W.tile_sum, W.or_and, st);
    if (n <= LL_VX_FUSED_WGS * LL_VB) {
        hipLaunchKernelGGL(k_vx_finish, dim3((n + LL_VB - 1) / LL_VB), dim3(LL_VB), 0, st, pts, (const unsigned long long *)W.keys, (const int *)W.vals, n, out,
                           W.seg_count, n_out_dev, W.pub);
        return rc;
    }
    hipLaunchKernelGGL(k_vx_heads, dim3((max(n, nseg) + LL_VB - 1) / LL_VB), dim3(LL_VB), 0, st, W.keys, n, nseg, W.flag, W.seg_count);
    ll_copy_d2d(W.rank, W.flag, (size_t)n * sizeof(int), st);
    ll_fill_words(W.rank + n, 1, 0, 0, 1, st);
    ll_device_exscan(W.rank, n + 1, W.tile_sum, st);                 /* rank[n] = number of voxels */
    ll_copy_d2d(n_out_dev, W.rank + n, sizeof(int), st);
    hipLaunchKernelGGL(k_vx_centroid, dim3((n + LL_VB - 1) / LL_VB), dim3(LL_VB), 0, st, pts, W.keys, W.vals, W.flag, W.rank, n, out, W.seg_count);
    return rc;                                                       /* non-zero: the sort's read-back failed (the output is still sorted) */
}
