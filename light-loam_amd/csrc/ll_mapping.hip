/*
 * ll_mapping.hip -- SURVEY 8f #2, first stage: the scan-to-submap optimisation of laserMapping
 * (/root/reference/src/laserMapping.cpp:1822-2095) for one scan against the corner / surf clouds gathered from the
 * cube map (laserCloudCornerFromMap / laserCloudSurfFromMap, :1803-1808).
 *
 *   k_map_bbox, k_map_count, scan, k_map_scatter   what replaces kdtree->setInputCloud (:1826-1827): a dense 3-D cell grid
 *                       over the cloud's bounding box, cells >= 1.01 m.  The search below only has to be exact when the
 *                       fifth neighbour is closer than 1 m (:1884, :1952), and then all five lie in the 3 x 3 x 3 cells
 *                       around the query's cell (the 1 % margin absorbs the f32 rounding of the cell coordinate).
 *   k_map_knn           one thread per stack point (both stacks in one launch of 64-thread workgroups): pointAssociateToMap (:125-134, f64 rotate + translate, f32 store),
 *                       exact K = 5 by (FLANN L2_Simple f32 distance, index) -- PCL's kd-tree leaves equal distances
 *                       to traversal order, here and in the oracle the lower index wins -- then
 *                         corners (:1886-1921): mean, 3 x 3 covariance, symmetric eigen-decomposition (cyclic Jacobi in
 *                           f64 for Eigen's SelfAdjointEigenSolver), line if l2 > 3 l1, a / b = mean +- 0.1 direction;
 *                         surfs (:1954-2035): 5 x 3 least squares n.x = -1 by Householder QR with column pivoting
 *                           (Eigen's colPivHouseholderQr), unit normal + negative_OA_dot_norm, valid if all five points
 *                           are within 0.2 of the plane.
 *   k_map_compact       residual blocks in stack order (the order AddResidualBlock sees them).
 *   k_map_normal_eq     LidarEdgeFactor + LidarPlaneNormFactor rows (lidarFactor.hpp:9-52, :253-285), HuberLoss(0.1),
 *                       EigenQuaternionManifold, 28 f64 accumulators per thread -> shuffle tree -> LDS.
 * The LM iterations reuse k_lm_begin / k_lm_propose / k_lm_accept of ll_factors.hip on a one-slot view.
 * Everything after the f32 distances is f64; parity with the oracle is to rounding (1e-9), the selected blocks exact.
 */
#include "ll_factor_math.h"
#include "ll_lm_step.h"
#include <limits.h>
#include <string.h>
#include <atomic>

#define LL_MAPB 256

/* order-preserving float <-> int for atomicMin / atomicMax */
__device__ __forceinline__ int ll_f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
static inline float ll_ord2f_host(int i) { const int j = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &j, 4); return f; }

__global__ __launch_bounds__(LL_MAPB) void k_map_bbox(const float4 *pts, int n, int *bbox /* min xyz, max xyz as ordered ints */)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * LL_MAPB + threadIdx.x; i < n; i += gridDim.x * LL_MAPB) {
        const float4 p = pts[i];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
    for (int o = 32; o > 0; o >>= 1)
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    /* one set of global atomics per workgroup (they all hit the same six words) */
    __shared__ float red[LL_MAPB / 64][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; ++k) { red[wave][k] = mn[k]; red[wave][3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < LL_MAPB / 64; ++w) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bbox[threadIdx.x], ll_f2ord(v)); else atomicMax(&bbox[threadIdx.x], ll_f2ord(v));
    }
}

__device__ __forceinline__ int ll_cell3(const LLGrid3 &G, float x, float y, float z)
{
    int cx = (int)floorf((x - G.org[0]) / G.cell), cy = (int)floorf((y - G.org[1]) / G.cell), cz = (int)floorf((z - G.org[2]) / G.cell);
    cx = min(max(cx, 0), G.dim[0] - 1); cy = min(max(cy, 0), G.dim[1] - 1); cz = min(max(cz, 0), G.dim[2] - 1);
    return (cz * G.dim[1] + cy) * G.dim[0] + cx;
}

__global__ __launch_bounds__(LL_MAPB) void k_map_count(LLGrid3 G, const float4 *pts, int n)
{
    const int i = blockIdx.x * LL_MAPB + threadIdx.x;
    if (i < n) { const float4 p = pts[i]; atomicAdd(&G.start[ll_cell3(G, p.x, p.y, p.z)], 1); }
}

/* exclusive scan of a device int array in three launches: 4096-element tiles, the tile totals, the offsets */
#define LL_SCAN_TILE 4096
__global__ __launch_bounds__(1024) void k_scan_tiles(int *data, int n, int *tile_sum)
{
    __shared__ int sc[16];
    const int base = blockIdx.x * LL_SCAN_TILE + threadIdx.x * 4;
    int v[4], s = 0;
    for (int u = 0; u < 4; ++u) { v[u] = (base + u < n) ? data[base + u] : 0; s += v[u]; }
    int total;
    int run = ll_block_exscan_n<16>(s, sc, total);
    for (int u = 0; u < 4; ++u) { if (base + u < n) data[base + u] = run; run += v[u]; }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_scan_totals(int *tile_sum, int nt)
{
    __shared__ int sc[16];
    const int per = (nt + 1023) / 1024;
    const int a0 = min(nt, (int)threadIdx.x * per), a1 = min(nt, a0 + per);
    int s = 0;
    for (int i = a0; i < a1; ++i) s += tile_sum[i];
    int total;
    int run = ll_block_exscan_n<16>(s, sc, total);
    for (int i = a0; i < a1; ++i) { const int v = tile_sum[i]; tile_sum[i] = run; run += v; }
}
__global__ __launch_bounds__(1024) void k_scan_add(int *data, int n, const int *tile_sum)
{
    const int base = blockIdx.x * LL_SCAN_TILE + threadIdx.x * 4;
    const int o = tile_sum[blockIdx.x];
    for (int u = 0; u < 4; ++u) if (base + u < n) data[base + u] += o;
}

__global__ __launch_bounds__(LL_MAPB) void k_map_scatter(LLGrid3 G, const float4 *pts, int n)
{
    const int i = blockIdx.x * LL_MAPB + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int pos = atomicAdd(&G.cursor[ll_cell3(G, p.x, p.y, p.z)], 1);
    G.pts[pos] = make_float4(p.x, p.y, p.z, __int_as_float(i));
}

/* ---- 3 x 3 symmetric eigen-decomposition, cyclic Jacobi ---- */
__device__ __forceinline__ void ll_sym_eig3(const double Ain[3][3], double w[3], double V[3][3])
{
    double A[3][3], Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = Ain[i][j];
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double qkp = Q[k][p], qkq = Q[k][q]; Q[k][p] = c * qkp - sn * qkq; Q[k][q] = sn * qkp + c * qkq; }
            }
    }
    /* ascending eigenvalues: bubble the three diagonal entries, columns follow */
    double d[3] = {A[0][0], A[1][1], A[2][2]};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2 - i; ++j)
            if (d[j] > d[j + 1]) {
                const double td = d[j]; d[j] = d[j + 1]; d[j + 1] = td;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double tq = Q[k][j]; Q[k][j] = Q[k][j + 1]; Q[k][j + 1] = tq; }
            }
    for (int k = 0; k < 3; ++k) { w[k] = d[k]; for (int i = 0; i < 3; ++i) V[i][k] = Q[i][k]; }
}

/* ---- 5 x 3 least squares by Householder QR with column pivoting ---- */
__device__ __forceinline__ void ll_qr_solve_5x3(double A[5][3], double b[5], double x[3])
{
    int perm[3] = {0, 1, 2};
    double maxn2 = 0.0;
    for (int j = 0; j < 3; ++j) { double n2 = 0.0; for (int i = 0; i < 5; ++i) n2 += A[i][j] * A[i][j]; if (n2 > maxn2) maxn2 = n2; }
    const double thr = maxn2 * (2.220446049250313e-16 * 2.220446049250313e-16) / 5.0;
    int rank = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k >= rank) break;
        int piv = k; double best = -1.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < k) continue;
            double n2 = 0.0;
            for (int i = k; i < 5; ++i) n2 += A[i][j] * A[i][j];
            if (n2 > best) { best = n2; piv = j; }
        }
        if (best < thr) { rank = k; break; }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j == piv && piv != k) {
                for (int i = 0; i < 5; ++i) { const double tmp = A[i][k]; A[i][k] = A[i][j]; A[i][j] = tmp; }
                const int tp = perm[k]; perm[k] = perm[j]; perm[j] = tp;
            }
        double tail = 0.0;
        for (int i = k + 1; i < 5; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double tau, beta, v[5];
        if (tail <= 2.2250738585072014e-308) { tau = 0.0; beta = c0; for (int i = 0; i < 5; ++i) v[i] = 0.0; }
        else {
            beta = sqrt(c0 * c0 + tail); if (c0 >= 0.0) beta = -beta;
            for (int i = 0; i < 5; ++i) v[i] = (i > k) ? A[i][k] / (c0 - beta) : 0.0;
            tau = (beta - c0) / beta;
        }
        v[k] = 1.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j <= k) continue;
            double d = 0.0;
            for (int i = k; i < 5; ++i) d += v[i] * A[i][j];
            d *= tau;
            for (int i = k; i < 5; ++i) A[i][j] -= d * v[i];
        }
        { double d = 0.0; for (int i = k; i < 5; ++i) d += v[i] * b[i]; d *= tau; for (int i = k; i < 5; ++i) b[i] -= d * v[i]; }
        A[k][k] = beta;
        for (int i = k + 1; i < 5; ++i) A[i][k] = 0.0;
    }
    double z[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i >= rank) continue;
        double sacc = b[i];
        for (int j = i + 1; j < 3; ++j) if (j < rank) sacc -= A[i][j] * z[j];
        z[i] = sacc / A[i][i];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) if (perm[k] == c) x[c] = z[k];
}

/* pointAssociateToMap (:125-134) of stack point po */
__device__ __forceinline__ void ll_map_to_world(const double *pose, const float4 po, float &sx, float &sy, float &sz)
{
    const double ux = pose[0], uy = pose[1], uz = pose[2], w = pose[3];
    const double v[3] = {(double)po.x, (double)po.y, (double)po.z};
    double uvx = uy * v[2] - uz * v[1], uvy = uz * v[0] - ux * v[2], uvz = ux * v[1] - uy * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    sx = (float)(((v[0] + w * uvx) + (uy * uvz - uz * uvy)) + pose[4]);
    sy = (float)(((v[1] + w * uvy) + (uz * uvx - ux * uvz)) + pose[5]);
    sz = (float)(((v[2] + w * uvz) + (ux * uvy - uy * uvx)) + pose[6]);
}

/* insertion of (d, j[, p]) into five slots kept ascending by (distance, index) */
template <bool WITH_PT>
__device__ __forceinline__ void ll_five_insert(float bd[5], int bi[5], float4 bp[5], int &nb, float d, int j, const float4 &p)
{
    if (d < bd[4] || (d == bd[4] && j < bi[4])) {
        bd[4] = d; bi[4] = j;
        if (WITH_PT) bp[4] = p;
#pragma unroll
        for (int s = 4; s > 0; --s)
            if (bd[s] < bd[s - 1] || (bd[s] == bd[s - 1] && bi[s] < bi[s - 1])) {
                const float td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
                const int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
                if (WITH_PT) { const float4 tp = bp[s]; bp[s] = bp[s - 1]; bp[s - 1] = tp; }
            }
        if (nb < 5) ++nb;
    }
}

/* exact K = 5 over the 27 cells around the query, ascending (distance, index).  gid == nullptr: index = position in the
 * search cloud; otherwise the caller's global id of the point (a tile shard of a larger cloud, same tie order). */
template <bool WITH_PT>
__device__ __forceinline__ void ll_map_search5(const LLGrid3 &G, int n_map, const int *gid, float sx, float sy, float sz,
                                               float bd[5], int bi[5], float4 bp[5], int &nb)
{
    nb = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { bd[k] = INFINITY; bi[k] = INT_MAX; if (WITH_PT) bp[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (n_map <= 0) return;
    int cx = (int)floorf((sx - G.org[0]) / G.cell), cy = (int)floorf((sy - G.org[1]) / G.cell), cz = (int)floorf((sz - G.org[2]) / G.cell);
    cx = min(max(cx, 0), G.dim[0] - 1); cy = min(max(cy, 0), G.dim[1] - 1); cz = min(max(cz, 0), G.dim[2] - 1);
    /* the nine (z, y) rows of the 27 cells: the three x-cells of a row are contiguous, so a row is one range.  All eighteen
     * bounds first (independent loads): a thread's search is a chain of dependent round trips, and a frame's association is
     * ~11 k of them on a chip that holds 65 k threads. */
    int st[9], en[9];
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.dim[0] - 1);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
        st[r] = 0; en[r] = 0;
        if (yy >= 0 && zz >= 0 && yy < G.dim[1] && zz < G.dim[2]) {
            const int row = (zz * G.dim[1] + yy) * G.dim[0];
            st[r] = G.start[row + x0]; en[r] = G.start[row + x1 + 1];
        }
    }
    /* all nine ranges advance together, two points of each per round: the five best are a total order on (distance, index), so
     * the order of insertion does not matter, and a search is 2-3 round trips of up to eighteen loads instead of nine chains */
    for (;;) {
        float4 p[9][2];
        bool any = false;
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int u = 0; u < 2; ++u) if (st[r] + u < en[r]) { p[r][u] = G.pts[st[r] + u]; any = true; }
        if (!any) break;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (st[r] + u >= en[r]) continue;
                float diff = sx - p[r][u].x; float d = diff * diff;            /* FLANN L2_Simple: a = query, b = data */
                diff = sy - p[r][u].y; d += diff * diff;
                diff = sz - p[r][u].z; d += diff * diff;
                int j = __float_as_int(p[r][u].w);
                if (WITH_PT && gid) j = gid[j];
                ll_five_insert<WITH_PT>(bd, bi, bp, nb, d, j, p[r][u]);
            }
            st[r] += 2;
        }
    }
}

/* the five neighbours of stack point i -> line (eigen test, :1888-1930) or plane (QR fit, :1960-2000) */
template <bool CORNER>
__device__ __forceinline__ void ll_map_fit(const LLMapView &M, int i, const double P5[5][3])
{
    const int which = CORNER ? 0 : 1;
    unsigned char ok = 0;
    if (CORNER) {
        double c[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 5; ++j) for (int k = 0; k < 3; ++k) c[k] = c[k] + P5[j][k];              /* :1888-1895 */
        for (int k = 0; k < 3; ++k) c[k] = c[k] / 5.0;
        double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int j = 0; j < 5; ++j) {                                                                  /* :1898-1903 */
            const double z[3] = {P5[j][0] - c[0], P5[j][1] - c[1], P5[j][2] - c[2]};
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a][b] = cov[a][b] + z[a] * z[b];
        }
        double w[3], V[3][3];
        ll_sym_eig3(cov, w, V);
        if (w[2] > 3 * w[1]) {                                                                         /* :1911 */
            ok = 1;
            for (int k = 0; k < 3; ++k) { M.qa[(size_t)i * 3 + k] = 0.1 * V[k][2] + c[k]; M.qb[(size_t)i * 3 + k] = -0.1 * V[k][2] + c[k]; }
        }
    } else {
        double A[5][3], b[5] = {-1.0, -1.0, -1.0, -1.0, -1.0}, nrm[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 5; ++j) for (int k = 0; k < 3; ++k) A[j][k] = P5[j][k];
        ll_qr_solve_5x3(A, b, nrm);                                                                     /* :1972 */
        const double len = sqrt((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
        const double nd = 1 / len;                                                                     /* :1973 */
        if (len * len > 0.0) { nrm[0] /= len; nrm[1] /= len; nrm[2] /= len; }                        /* :1974 */
        bool valid = true;
#pragma unroll
        for (int j = 0; j < 5; ++j)                                                                    /* :1980-1990 */
            if (fabs(nrm[0] * P5[j][0] + nrm[1] * P5[j][1] + nrm[2] * P5[j][2] + nd) > 0.2) valid = false;
        if (valid) {
            ok = 1;
            for (int k = 0; k < 3; ++k) M.qn[(size_t)i * 3 + k] = nrm[k];
            M.qd[i] = nd;
        }
    }
    M.ok[which][i] = ok;
}

#define LL_KNNB 64            /* threads per workgroup of the searches: ~11 k stack points should spread over the chip, not fill 45 workgroups */
template <bool CORNER>
__device__ __forceinline__ void ll_map_knn_one(const LLMapView &M, int i)
{
    const int which = CORNER ? 0 : 1;
    if (i >= M.n_stk[which]) return;
    float sx, sy, sz;
    ll_map_to_world(M.pose, M.stk[which][i], sx, sy, sz);
    float bd[5]; int bi[5]; int nb; float4 unused[1];
    ll_map_search5<false>(M.grid[which], M.n_map[which], nullptr, sx, sy, sz, bd, bi, unused, nb);
    if (nb == 5 && bd[4] < 1.0f) {                                                                     /* :1885, :1958 */
        const float4 *cloud = M.map[which];
        double P5[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) { const float4 p = cloud[bi[j]]; P5[j][0] = p.x; P5[j][1] = p.y; P5[j][2] = p.z; }
        ll_map_fit<CORNER>(M, i, P5);
    } else M.ok[which][i] = 0;
}
/* both stacks in ONE launch: the first blocks_corner workgroups take the corner points, the rest the surface points */
__global__ __launch_bounds__(LL_KNNB) void k_map_knn(LLMapView M, int blocks_corner)
{
    if ((int)blockIdx.x < blocks_corner) ll_map_knn_one<true>(M, blockIdx.x * LL_KNNB + threadIdx.x);
    else ll_map_knn_one<false>(M, ((int)blockIdx.x - blocks_corner) * LL_KNNB + threadIdx.x);
}

/* tile-parallel mapping: this rank's search clouds are the points of ITS cubes; per stack point the five
 * nearest of them go out as (x, y, z, distance) + global id, INFINITY / INT_MAX in the unused slots */
template <bool CORNER>
__device__ __forceinline__ void ll_map_knn_partial_one(const LLMapView &M, int i)
{
    const int which = CORNER ? 0 : 1;
    if (i >= M.n_stk[which]) return;
    float sx, sy, sz;
    ll_map_to_world(M.pose, M.stk[which][i], sx, sy, sz);
    float bd[5]; int bi[5]; float4 bp[5]; int nb;
    ll_map_search5<true>(M.grid[which], M.n_map[which], M.gid[which], sx, sy, sz, bd, bi, bp, nb);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        M.nn_pt[which][(size_t)i * 5 + j] = make_float4(bp[j].x, bp[j].y, bp[j].z, bd[j]);
        M.nn_id[which][(size_t)i * 5 + j] = bi[j];
    }
}
__global__ __launch_bounds__(LL_KNNB) void k_map_knn_partial(LLMapView M, int blocks_corner)
{
    if ((int)blockIdx.x < blocks_corner) ll_map_knn_partial_one<true>(M, blockIdx.x * LL_KNNB + threadIdx.x);
    else ll_map_knn_partial_one<false>(M, ((int)blockIdx.x - blocks_corner) * LL_KNNB + threadIdx.x);
}

/* the candidates of all ranks ([part][stack point][5]) -> the five nearest overall -> the same fit as k_map_knn */
template <bool CORNER>
__global__ __launch_bounds__(LL_MAPB) void k_map_fit_merged(LLMapView M, int n_parts, const float4 *pt_all, const int *id_all)
{
    const int which = CORNER ? 0 : 1;
    const int n = M.n_stk[which];
    const int i = blockIdx.x * LL_MAPB + threadIdx.x;
    if (i >= n) return;
    float bd[5]; int bi[5]; float4 bp[5]; int nb = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { bd[k] = INFINITY; bi[k] = INT_MAX; bp[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
    for (int part = 0; part < n_parts; ++part)
        for (int j = 0; j < 5; ++j) {
            const size_t at = ((size_t)part * n + i) * 5 + j;
            const float4 p = pt_all[at];
            const int id = id_all[at];
            if (id == INT_MAX) break;                                   /* a part's slots are filled front to back */
            ll_five_insert<true>(bd, bi, bp, nb, p.w, id, p);
        }
    if (nb == 5 && bd[4] < 1.0f) {
        double P5[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) { P5[j][0] = bp[j].x; P5[j][1] = bp[j].y; P5[j][2] = bp[j].z; }
        ll_map_fit<CORNER>(M, i, P5);
    } else M.ok[which][i] = 0;
}

/* per-query results -> residual blocks in stack order; one 1024-thread workgroup */
/* stack points that produced a residual block -> the block lists, in stack order.  One thread per stack point; a workgroup
 * counts its valid points, publishes the count (launch tag << 32 | count: one 8-byte agent-scope atomic) and takes its offset
 * from the counts of the workgroups before it as they appear (all of a frame's workgroups are resident).  The last workgroup
 * of a cloud type writes the total. */
#define LL_CMPB 256
__global__ __launch_bounds__(LL_CMPB) void k_map_compact(LLMapView M, int blocks_corner)
{
    __shared__ int sc[LL_CMPB / 64];
    __shared__ int sbase;
    const int tid = threadIdx.x, lane = tid & 63;
    const int which = (int)blockIdx.x < blocks_corner ? 0 : 1;
    const int b = which ? (int)blockIdx.x - blocks_corner : (int)blockIdx.x;
    const int nb = which ? (int)gridDim.x - blocks_corner : blocks_corner;
    const int n = M.n_stk[which];
    const int i = b * LL_CMPB + tid;
    const bool ok = i < n && M.ok[which][i];
    /* the block's data, fetched before the wait */
    double v[7];
    if (ok) {
        if (which == 0) for (int k = 0; k < 3; ++k) { v[k] = M.qa[(size_t)i * 3 + k]; v[3 + k] = M.qb[(size_t)i * 3 + k]; }
        else { for (int k = 0; k < 3; ++k) v[k] = M.qn[(size_t)i * 3 + k]; v[3] = M.qd[i]; }
    }
    int total;
    const int lpos = ll_block_exscan_n<LL_CMPB / 64>(ok ? 1 : 0, sc, total);
    unsigned long long *pub = M.cpub[which];
    const unsigned long long tag = (unsigned long long)(unsigned)M.cpub_tag << 32;
    if (tid == 0) __hip_atomic_store(&pub[b], tag | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ int slost;
    int before = 0;
    bool lost = false;
    for (int q = tid; q < b; q += LL_CMPB) {
        unsigned long long w; int spins = 0;
        while (((w = __hip_atomic_load(&pub[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (unsigned)M.cpub_tag && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(8);
        if ((w >> 32) != (unsigned)M.cpub_tag) lost = true;
        before += (int)(w & 0xffffffffull);
    }
    before = ll_wave_sum_i32(before);
    if (tid == 0) { sbase = 0; slost = 0; }
    __syncthreads();
    if (lane == 0 && before) atomicAdd(&sbase, before);
    if (lost) slost = 1;
    __syncthreads();
    if (slost) {
        /* an earlier block never published (a logic error, not a race: they are dispatched first): never a silently wrong block list --
         * this block places nothing and poisons the pose, so that the solve that follows hands back NaN (what ll_map_solve's callers
         * already have to check for, map_lm_repair) */
        if (tid == 0) { M.pose[0] = __builtin_nan(""); if (b == nb - 1) M.counts[which] = 0; }
        return;
    }
    const int pos = sbase + lpos;
    if (ok) {
        M.src[which][pos] = i;
        if (which == 0) for (int k = 0; k < 3; ++k) { M.fa[(size_t)pos * 3 + k] = v[k]; M.fb[(size_t)pos * 3 + k] = v[3 + k]; }
        else { for (int k = 0; k < 3; ++k) M.fn[(size_t)pos * 3 + k] = v[k]; M.fd[pos] = v[3]; }
    }
    if (b == nb - 1 && tid == 0) M.counts[which] = sbase + total;
}

__global__ __launch_bounds__(256) void k_map_normal_eq(LLMapView M)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gtid = blockIdx.x * 256 + tid, gsz = LL_NEQ_NB * 256;
    Pose P;
    for (int k = 0; k < 4; ++k) P.q[k] = M.pose[k];
    for (int k = 0; k < 3; ++k) P.t[k] = M.pose[4 + k];
    const int n_e = M.counts[0], n_p = M.counts[1];
    double acc[LL_NACC];
#pragma unroll
    for (int k = 0; k < LL_NACC; ++k) acc[k] = 0.0;
    const int rw = M.row_world > 1 ? M.row_world : 1, rr = M.row_world > 1 ? M.row_rank : 0;   /* this rank's share of the blocks */
    for (int i = rr + rw * gtid; i < n_e; i += gsz * rw) {
        double r[3], Jq[3][4], Jt[3][3];
        ll_edge_d(P, M.stk[0][M.src[0][i]], &M.fa[(size_t)i * 3], &M.fb[(size_t)i * 3], r, Jq, Jt);
        const double sc = ll_huber_scale(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], M.huber, acc[27]);
        for (int row = 0; row < 3; ++row) {
            double J[6];
            ll_to_local(P, Jq[row], J);
            J[3] = Jt[row][0]; J[4] = Jt[row][1]; J[5] = Jt[row][2];
            for (int k = 0; k < 6; ++k) J[k] *= sc;
            ll_acc_row(acc, J, r[row] * sc);
        }
    }
    for (int i = rr + rw * gtid; i < n_p; i += gsz * rw) {
        double r, Jq[4], Jt[3], J[6];
        ll_plane_norm(P, M.stk[1][M.src[1][i]], &M.fn[(size_t)i * 3], M.fd[i], r, Jq, Jt);
        const double sc = ll_huber_scale(r * r, M.huber, acc[27]);
        ll_to_local(P, Jq, J);
        J[3] = Jt[0]; J[4] = Jt[1]; J[5] = Jt[2];
        for (int k = 0; k < 6; ++k) J[k] *= sc;
        ll_acc_row(acc, J, r * sc);
    }
    __shared__ double red[4][LL_NACC];
    __shared__ double tot[LL_NACC];
    __shared__ int s_last;
#pragma unroll
    for (int k = 0; k < LL_NACC; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < LL_NACC) {
        const double v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        __hip_atomic_store(&M.neq_part[blockIdx.x * LL_NACC + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(M.neq_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == LL_NEQ_NB - 1);
    __syncthreads();
    if (!s_last) return;
    __shared__ double spart[LL_NEQ_NB * LL_NACC];
    for (int e = tid; e < LL_NEQ_NB * LL_NACC; e += 256)
        spart[e] = __hip_atomic_load(&M.neq_part[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid < LL_NACC) {
        double v = 0.0;
        for (int b = 0; b < LL_NEQ_NB; ++b) v += spart[b * LL_NACC + tid];
        tot[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double *out = M.neq;
        int k = 0;
        for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { out[a * 6 + b] = tot[k]; out[b * 6 + a] = tot[k]; ++k; }
        for (int a = 0; a < 6; ++a) out[36 + a] = tot[21 + a];
        out[42] = tot[27];
        out[43] = (double)(3 * ((n_e - rr + rw - 1) / rw) + (n_p - rr + rw - 1) / rw);   /* rows summed here */
        __hip_atomic_store(M.neq_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* for the next launch */
    }
}

/* One ceres::Solve (:2072-2082) in ONE launch: evaluate, begin, n x (propose, evaluate, accept) on the LL_NEQ_NB workgroups of
 * k_map_normal_eq, which stay resident for the whole solve.  Every round all workgroups sum their rows at the current pose
 * (the same partition and the same fixed order of the partial sums as k_map_normal_eq: bit-identical normal equations);
 * workgroup 0 waits for the LL_NEQ_NB partial sums, takes the single-thread trust-region step (ll_lm_step.h, state in its LDS),
 * publishes the next pose and releases the others.  Partial sums, pose, arrival counter and round counter cross the
 * workgroups (and the XCDs) as agent-scope atomics; every wait is bounded and a time-out leaves a NaN pose (fails loudly).
 * 16 workgroups are always co-resident on a 256-CU chip.  Replaces 2 + 3 n dependent launches (the launch-per-step kernels
 * stay for the row-parallel mode, where an all-reduce sits between evaluate and accept). */
__global__ __launch_bounds__(256) void k_map_lm_solve(LLMapView M, LLLmOpt o)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int gtid = wg * 256 + tid, gsz = LL_NEQ_NB * 256;
    const int n_e = M.counts[0], n_p = M.counts[1];
    __shared__ double red[4][LL_NACC];
    __shared__ double sp[7], sneq[LL_NEQ_STRIDE], sL[LL_LM_STRIDE], stot[LL_NACC], spart[LL_NEQ_NB * LL_NACC];
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;
    for (int k = tid; k < LL_LM_STRIDE; k += 256) sL[k] = 0.0;
    constexpr int SPINS = 1 << 22;
    for (int round = 0; round <= o.max_num_iterations; ++round) {
        if (round == 0) { if (tid < 7) sp[tid] = M.pose[tid]; }
        else if (wg != 0) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(M.lm_go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)round && ++spins < SPINS) __builtin_amdgcn_s_sleep(4);
                if (spins >= SPINS) s_fail = 1;
            }
            __syncthreads();
            if (s_fail) return;                                   /* workgroup 0 timed out or died: give up, it reports */
            if (tid < 7) sp[tid] = __hip_atomic_load(&M.pose[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        Pose P;
        for (int k = 0; k < 4; ++k) P.q[k] = sp[k];
        for (int k = 0; k < 3; ++k) P.t[k] = sp[4 + k];
        double acc[LL_NACC];
#pragma unroll
        for (int k = 0; k < LL_NACC; ++k) acc[k] = 0.0;
        for (int i = gtid; i < n_e; i += gsz) {
            double r[3], Jq[3][4], Jt[3][3];
            ll_edge_d(P, M.stk[0][M.src[0][i]], &M.fa[(size_t)i * 3], &M.fb[(size_t)i * 3], r, Jq, Jt);
            const double sc = ll_huber_scale(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], M.huber, acc[27]);
            for (int row = 0; row < 3; ++row) {
                double J[6];
                ll_to_local(P, Jq[row], J);
                J[3] = Jt[row][0]; J[4] = Jt[row][1]; J[5] = Jt[row][2];
                for (int k = 0; k < 6; ++k) J[k] *= sc;
                ll_acc_row(acc, J, r[row] * sc);
            }
        }
        for (int i = gtid; i < n_p; i += gsz) {
            double r, Jq[4], Jt[3], J[6];
            ll_plane_norm(P, M.stk[1][M.src[1][i]], &M.fn[(size_t)i * 3], M.fd[i], r, Jq, Jt);
            const double sc = ll_huber_scale(r * r, M.huber, acc[27]);
            ll_to_local(P, Jq, J);
            J[3] = Jt[0]; J[4] = Jt[1]; J[5] = Jt[2];
            for (int k = 0; k < 6; ++k) J[k] *= sc;
            ll_acc_row(acc, J, r * sc);
        }
#pragma unroll
        for (int k = 0; k < LL_NACC; ++k) {
            double v = acc[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (tid < LL_NACC) {
            const double v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
            __hip_atomic_store(&M.neq_part[wg * LL_NACC + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(M.neq_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (wg != 0) { if (round == o.max_num_iterations) return; continue; }
        /* ---- workgroup 0: all partial sums of this round, then the step ---- */
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(M.neq_ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)((round + 1) * LL_NEQ_NB) && ++spins < SPINS) __builtin_amdgcn_s_sleep(4);
            if (spins >= SPINS) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) break;
        for (int e = tid; e < LL_NEQ_NB * LL_NACC; e += 256)      /* all loads in flight at once, the sums in workgroup order */
            spart[e] = __hip_atomic_load(&M.neq_part[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid < LL_NACC) {
            double v = 0.0;
            for (int b = 0; b < LL_NEQ_NB; ++b) v += spart[b * LL_NACC + tid];
            stot[tid] = v;
        }
        __syncthreads();
        if (tid == 0) {
            int k = 0;
            for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { sneq[a * 6 + b] = stot[k]; sneq[b * 6 + a] = stot[k]; ++k; }
            for (int a = 0; a < 6; ++a) sneq[36 + a] = stot[21 + a];
            sneq[42] = stot[27];
            sneq[43] = (double)(3 * n_e + n_p);
            if (round == 0) ll_lm_begin_one(sL, sneq, sp, o); else ll_lm_accept_one(sL, sneq, sp, o);
            if (round < o.max_num_iterations) {
                ll_lm_propose_one(sL, sp, o);
                for (int q = 0; q < 7; ++q) __hip_atomic_store(&M.pose[q], sp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(M.lm_go, (unsigned)(round + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    /* workgroup 0 only: the result, and the counters back to zero for the next launch (nobody waits on them any more, except
     * after a time-out -- then the round counter is pushed past every round so that stragglers leave) */
    if (s_fail) { if (tid < 7) sp[tid] = __builtin_nan(""); }
    __syncthreads();
    if (tid < 7) M.pose[tid] = sp[tid];
    for (int k = tid; k < LL_NEQ_STRIDE; k += 256) M.neq[k] = sneq[k];
    for (int k = tid; k < LL_LM_STRIDE; k += 256) M.lm[k] = sL[k];
    if (tid == 0) {
        __hip_atomic_store(M.neq_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(M.lm_go, s_fail ? 0x7fffffffu : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* what ceres::CostFunction::Evaluate returns for the blocks (loss not applied): edges first (3 rows each), then planes */
__global__ __launch_bounds__(LL_MAPB) void k_map_rows(LLMapView M, double *r_out, double *Jq_out, double *Jt_out)
{
    Pose P;
    for (int k = 0; k < 4; ++k) P.q[k] = M.pose[k];
    for (int k = 0; k < 3; ++k) P.t[k] = M.pose[4 + k];
    const int n_e = M.counts[0], n_p = M.counts[1];
    const int gtid = blockIdx.x * LL_MAPB + threadIdx.x, gsz = gridDim.x * LL_MAPB;
    for (int i = gtid; i < n_e; i += gsz) {
        double r[3], Jq[3][4], Jt[3][3];
        ll_edge_d(P, M.stk[0][M.src[0][i]], &M.fa[(size_t)i * 3], &M.fb[(size_t)i * 3], r, Jq, Jt);
        for (int row = 0; row < 3; ++row) {
            const size_t R0 = (size_t)3 * i + row;
            r_out[R0] = r[row];
            for (int k = 0; k < 4; ++k) Jq_out[R0 * 4 + k] = Jq[row][k];
            for (int k = 0; k < 3; ++k) Jt_out[R0 * 3 + k] = Jt[row][k];
        }
    }
    for (int i = gtid; i < n_p; i += gsz) {
        double r, Jq[4], Jt[3];
        ll_plane_norm(P, M.stk[1][M.src[1][i]], &M.fn[(size_t)i * 3], M.fd[i], r, Jq, Jt);
        const size_t R0 = (size_t)3 * n_e + i;
        r_out[R0] = r;
        for (int k = 0; k < 4; ++k) Jq_out[R0 * 4 + k] = Jq[k];
        for (int k = 0; k < 3; ++k) Jt_out[R0 * 3 + k] = Jt[k];
    }
}

/* ------------------------------------------------------------------ launchers */
/* in-place exclusive scan of n ints (n <= 4096 * 4096); tile_sum: ceil(n / 4096) ints of scratch */
void ll_device_exscan(int *data, int n, int *tile_sum, hipStream_t st)
{
    if (n <= 0) return;
    const int nt = (n + LL_SCAN_TILE - 1) / LL_SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(1024), 0, st, data, n, tile_sum);
    if (nt == 1) return;                                         /* one tile: already the whole scan */
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, st, tile_sum, nt);
    hipLaunchKernelGGL(k_scan_add, dim3(nt), dim3(1024), 0, st, data, n, tile_sum);
}

void ll_map_launch_bbox(const float4 *pts, int n, int *bbox_dev, hipStream_t st)
{
    ll_fill_words(bbox_dev, 6, INT_MAX, INT_MIN, 3, st);
    if (n > 0) hipLaunchKernelGGL(k_map_bbox, dim3(min(1024, (n + LL_MAPB - 1) / LL_MAPB)), dim3(LL_MAPB), 0, st, pts, n, bbox_dev);
}

void ll_map_bbox_to_grid(const int bbox_host[6], int n, int max_cells, LLGrid3 *G)
{
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = n > 0 ? ll_ord2f_host(bbox_host[k]) : 0.0f; mx[k] = n > 0 ? ll_ord2f_host(bbox_host[3 + k]) : 0.0f; }
    float cell = 1.01f;                      /* >= the 1 m acceptance radius + rounding margin */
    for (;;) {
        double nc = 1.0;
        for (int k = 0; k < 3; ++k) { G->dim[k] = (int)floorf((mx[k] - mn[k]) / cell) + 1; nc *= G->dim[k]; }
        if (nc <= (double)max_cells) break;
        cell *= 1.5f;
    }
    for (int k = 0; k < 3; ++k) G->org[k] = mn[k];
    G->cell = cell;
    G->ncell = G->dim[0] * G->dim[1] * G->dim[2];
}

void ll_map_launch_build(const LLGrid3 &G, const float4 *pts, int n, int *tile_sum, hipStream_t st)
{
    const int ns = G.ncell + 1;
    (void)hipMemsetAsync(G.start, 0, (size_t)ns * sizeof(int), st);
    if (n > 0) hipLaunchKernelGGL(k_map_count, dim3((n + LL_MAPB - 1) / LL_MAPB), dim3(LL_MAPB), 0, st, G, pts, n);
    ll_device_exscan(G.start, ns, tile_sum, st);
    ll_copy_d2d(G.cursor, G.start, (size_t)G.ncell * sizeof(int), st);
    if (n > 0) hipLaunchKernelGGL(k_map_scatter, dim3((n + LL_MAPB - 1) / LL_MAPB), dim3(LL_MAPB), 0, st, G, pts, n);
}

static void ll_map_launch_compact(const LLMapView &M, hipStream_t st)
{
    static std::atomic<unsigned> tag{0};                 /* distinguishes this launch's words from the previous launch's on the same map (never 0) */
    LLMapView V = M;
    V.cpub_tag = (int)(tag.fetch_add(1u) % 0x3fffffffu) + 1;
    const int bc = max(1, (M.n_stk[0] + LL_CMPB - 1) / LL_CMPB), bs = max(1, (M.n_stk[1] + LL_CMPB - 1) / LL_CMPB);   /* >= 1: an empty stack still writes its count */
    hipLaunchKernelGGL(k_map_compact, dim3(bc + bs), dim3(LL_CMPB), 0, st, V, bc);
}

void ll_map_launch_associate(const LLMapView &M, hipStream_t st)
{
    const int bc = (M.n_stk[0] + LL_KNNB - 1) / LL_KNNB, bs = (M.n_stk[1] + LL_KNNB - 1) / LL_KNNB;
    if (bc + bs > 0) hipLaunchKernelGGL(k_map_knn, dim3(bc + bs), dim3(LL_KNNB), 0, st, M, bc);
    ll_map_launch_compact(M, st);
}

void ll_map_launch_knn_partial(const LLMapView &M, hipStream_t st)
{
    const int bc = (M.n_stk[0] + LL_KNNB - 1) / LL_KNNB, bs = (M.n_stk[1] + LL_KNNB - 1) / LL_KNNB;
    if (bc + bs > 0) hipLaunchKernelGGL(k_map_knn_partial, dim3(bc + bs), dim3(LL_KNNB), 0, st, M, bc);
}

void ll_map_launch_associate_merged(const LLMapView &M, int n_parts, const float4 *const pt_all[2], const int *const id_all[2], hipStream_t st)
{
    if (M.n_stk[0] > 0) hipLaunchKernelGGL(k_map_fit_merged<true>, dim3((M.n_stk[0] + LL_MAPB - 1) / LL_MAPB), dim3(LL_MAPB), 0, st, M, n_parts, pt_all[0], id_all[0]);
    if (M.n_stk[1] > 0) hipLaunchKernelGGL(k_map_fit_merged<false>, dim3((M.n_stk[1] + LL_MAPB - 1) / LL_MAPB), dim3(LL_MAPB), 0, st, M, n_parts, pt_all[1], id_all[1]);
    ll_map_launch_compact(M, st);
}

void ll_map_launch_normal_eq(const LLMapView &M, hipStream_t st)
{
    hipLaunchKernelGGL(k_map_normal_eq, dim3(LL_NEQ_NB), dim3(256), 0, st, M);
}

void ll_map_launch_lm_solve(const LLMapView &M, const LLLmOpt &o, hipStream_t st)
{
    hipLaunchKernelGGL(k_map_lm_solve, dim3(LL_NEQ_NB), dim3(256), 0, st, M, o);
}

void ll_map_launch_rows(const LLMapView &M, double *r, double *Jq, double *Jt, hipStream_t st)
{
    hipLaunchKernelGGL(k_map_rows, dim3(64), dim3(LL_MAPB), 0, st, M, r, Jq, Jt);
}
