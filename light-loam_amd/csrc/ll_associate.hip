/*
 * ll_associate.hip -- a5 + a6 + a7: TransformToStart, K=1 nearest neighbour, ring-window second / third point.
 * Replaces laserOdometry.cpp:77-95, :491-620 (corners), :653-793 (planes) and the kd-tree rebuild :895-896
 * of /root/reference.
 *
 * Targets of slot s are the less-sharp / less-flat clouds of slot s-1, still resident in HBM from the
 * extract stage (no publish -> subscribe -> fromROSMsg -> kd-tree build).  The kd-tree's exact K=1 search
 * is restated as an exact scan: 256 queries per workgroup (one per lane), target tiles of 1024 points staged
 * in LDS and broadcast-read, FLANN's L2_Simple accumulation order ((dx*dx + dy*dy) + dz*dz, f32, no FMA),
 * equal distances resolved to the lowest index.  The ring walks (:504-553, :668-721) are sequential loops
 * with early breaks in the reference; here a wave scans the window 64 targets at a time for one query,
 * the break becomes a ballot ("first lane whose ring is out of the +-2.5 window"), and the running
 * minimum with strict '<' becomes a lexicographic (distance, visiting order) min-reduction.
 */
#include "ll_common.h"
#include <limits.h>

#define LL_NN_TILE 1024

__device__ __forceinline__ void ll_rotate(const double q[4], const double v[3], double out[3])
{
    /* Eigen _transformVector: uv = u x v; uv += uv; v + w*uv + u x uv */
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    double uvx = uy * v[2] - uz * v[1], uvy = uz * v[0] - ux * v[2], uvz = ux * v[1] - uy * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    out[0] = (v[0] + w * uvx) + (uy * uvz - uz * uvy);
    out[1] = (v[1] + w * uvy) + (uz * uvx - ux * uvz);
    out[2] = (v[2] + w * uvz) + (ux * uvy - uy * uvx);
}

struct Best { float d; int ord; int j; };

__device__ __forceinline__ void ll_best_reduce(Best &b)
{
    for (int o = 32; o > 0; o >>= 1) {
        const float d2 = __shfl_xor(b.d, o); const int o2 = __shfl_xor(b.ord, o); const int j2 = __shfl_xor(b.j, o);
        if (d2 < b.d || (d2 == b.d && o2 < b.ord)) { b.d = d2; b.ord = o2; b.j = j2; }
    }
}

__device__ __forceinline__ float ll_walk_d2(const float4 p, float sx, float sy, float sz)
{
    /* (p.x - sel.x)*(p.x - sel.x) + (p.y - sel.y)*(...) + (p.z - sel.z)*(...)  f32 (:514-519) */
    return (p.x - sx) * (p.x - sx) + (p.y - sy) * (p.y - sy) + (p.z - sz) * (p.z - sz);
}

template <bool PLANE>
__device__ __forceinline__ void ll_associate_block(const LLView &V, int s, int qblock,
                                                   const float4 *queries, int nq, const float4 *tgt, int M,
                                                   int *out_a, int *out_b, int *out_c, float4 *tile)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int qi = qblock * LL_BLOCK + tid;
    const bool have = qi < nq;
    /* a5: TransformToStart, s = 1 (DISTORTION 0): f64 rotate + translate, f32 store */
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (have) {
        const double *pose = V.pose + (size_t)s * 7;
        const float4 p = queries[qi];
        const double v[3] = {(double)p.x, (double)p.y, (double)p.z};
        double rr[3];
        ll_rotate(pose, v, rr);
        sx = (float)(rr[0] + pose[4]); sy = (float)(rr[1] + pose[5]); sz = (float)(rr[2] + pose[6]);
    }
    /* exact K=1 NN */
    float bestd = INFINITY; int besti = -1;
    for (int t0 = 0; t0 < M; t0 += LL_NN_TILE) {
        __syncthreads();
        for (int k = tid; k < LL_NN_TILE; k += LL_BLOCK) if (t0 + k < M) tile[k] = tgt[t0 + k];
        __syncthreads();
        const int cnt = min(LL_NN_TILE, M - t0);
        for (int j = 0; j < cnt; ++j) {
            const float4 p = tile[j];
            float diff = sx - p.x; float d = diff * diff;
            diff = sy - p.y; d += diff * diff;
            diff = sz - p.z; d += diff * diff;
            if (d < bestd) { bestd = d; besti = t0 + j; }
        }
    }
    int closest = (have && besti >= 0 && bestd < V.nn_max) ? besti : -1;      /* :497 / :659 */

    /* ring-window walks, one query at a time per wave */
    int res_b = -1, res_c = -1;
    for (int qq = 0; qq < 64; ++qq) {
        const int c = __shfl(closest, qq);
        if (c < 0) continue;
        const float qx = __shfl(sx, qq), qy = __shfl(sy, qq), qz = __shfl(sz, qq);
        const int rc = (int)tgt[c].w;                                           /* closestPointScanID (:500, :664) */
        Best b2 = {V.nn_max, INT_MAX, -1}, b3 = {V.nn_max, INT_MAX, -1};
        /* increasing scan line (:504-527 / :668-693) */
        for (int j0 = c + 1; j0 < M; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < M;
            const float4 p = in ? tgt[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int rj = (int)p.w;
            const bool stop = in && ((double)rj > (double)rc + V.nearby);
            const unsigned long long sm = __ballot(stop);
            const bool ok = in && (sm == 0ull || lane < __ffsll((long long)sm) - 1);
            if (ok) {
                const float d = ll_walk_d2(p, qx, qy, qz);
                const int ord = j - c - 1;
                if (PLANE) {
                    if (rj <= rc) { if (d < b2.d) { b2.d = d; b2.ord = ord; b2.j = j; } }
                    else          { if (d < b3.d) { b3.d = d; b3.ord = ord; b3.j = j; } }
                } else {
                    if (rj > rc && d < b2.d) { b2.d = d; b2.ord = ord; b2.j = j; }
                }
            }
            if (sm) break;
        }
        /* decreasing scan line (:530-553 / :696-721) */
        for (int j0 = c - 1; j0 >= 0; j0 -= 64) {
            const int j = j0 - lane;
            const bool in = j >= 0;
            const float4 p = in ? tgt[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int rj = (int)p.w;
            const bool stop = in && ((double)rj < (double)rc - V.nearby);
            const unsigned long long sm = __ballot(stop);
            const bool ok = in && (sm == 0ull || lane < __ffsll((long long)sm) - 1);
            if (ok) {
                const float d = ll_walk_d2(p, qx, qy, qz);
                const int ord = M + (c - 1 - j);
                if (PLANE) {
                    if (rj >= rc) { if (d < b2.d) { b2.d = d; b2.ord = ord; b2.j = j; } }
                    else          { if (d < b3.d) { b3.d = d; b3.ord = ord; b3.j = j; } }
                } else {
                    if (rj < rc && d < b2.d) { b2.d = d; b2.ord = ord; b2.j = j; }
                }
            }
            if (sm) break;
        }
        ll_best_reduce(b2);
        if (PLANE) ll_best_reduce(b3);
        if (lane == qq) { res_b = b2.j; res_c = b3.j; }
    }
    if (have) {
        bool valid = closest >= 0 && res_b >= 0 && (!PLANE || res_c >= 0);       /* :556 / :723 */
        out_a[qi] = valid ? closest : -1;
        out_b[qi] = valid ? res_b : -1;
        if (PLANE) out_c[qi] = valid ? res_c : -1;
    }
}

__global__ __launch_bounds__(LL_BLOCK) void k_associate(LLView V, int first, int count, int qb_corner, int qb_plane)
{
    const int per = qb_corner + qb_plane;
    const int sl = blockIdx.x / per, item = blockIdx.x % per;
    if (sl >= count) return;
    const int s = first + sl;
    __shared__ float4 tile[LL_NN_TILE];
    const ScanHdr h = V.hdr[s];
    const float4 *corner, *surf; int mc, ms;
    ll_targets(V, s, corner, mc, surf, ms);
    const bool ok = h.status == 0;
    if (item < qb_corner) {
        const int nq = ok ? h.n_sharp : 0;
        if (item * LL_BLOCK >= nq) return;
        ll_associate_block<false>(V, s, item, V.sharp + (size_t)s * V.cap_sharp, nq, corner, mc,
                                  V.eq_a + (size_t)s * V.cap_sharp, V.eq_b + (size_t)s * V.cap_sharp, nullptr, tile);
    } else {
        const int qb = item - qb_corner;
        const int nq = ok ? h.n_flat : 0;
        if (qb * LL_BLOCK >= nq) return;
        ll_associate_block<true>(V, s, qb, V.flat + (size_t)s * V.cap_flat, nq, surf, ms,
                                 V.pq_a + (size_t)s * V.cap_flat, V.pq_b + (size_t)s * V.cap_flat,
                                 V.pq_c + (size_t)s * V.cap_flat, tile);
    }
}

void ll_launch_associate(const LLView &V, int first, int count, hipStream_t st, LLProfiler *prof)
{
    const int qbc = (V.cap_sharp + LL_BLOCK - 1) / LL_BLOCK, qbp = (V.cap_flat + LL_BLOCK - 1) / LL_BLOCK;
    ll_prof_mark(prof, LL_K_ASSOCIATE, st);
    hipLaunchKernelGGL(k_associate, dim3(count * (qbc + qbp)), dim3(LL_BLOCK), 0, st, V, first, count, qbc, qbp);
    ll_prof_mark(prof, LL_K_END, st);
}
