/*
 * ll_oracle.c -- CPU ORACLE (test infrastructure; see ll_oracle.h header comment).
 * PARITY UNPINNED: restatement of /root/reference sources + published third-party algorithms;
 * no reference-held golden vectors exist and the reference cannot be built here.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (baseline x86-64: f32 stays f32, no FMA),
 * matching the reference build (CMakeLists.txt:4-6: -O3, no -march).
 *
 * Float/double promotion in every expression below follows the C++ source of the reference:
 * comments show the original expression where the promotion is not obvious.
 */
#include "ll_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* x86-64 cvttsd2si: NaN / out-of-range -> "integer indefinite" 0x80000000 */
static int trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

/* scanRegistration.cpp:139-162: the ring number of one point before the range check of :145-149 / :164-168.
 * float angle = atan(point.z / sqrt(point.x*point.x + point.y*point.y)) * 180 / M_PI;   (:139)
 * float overloads of sqrt/atan (libstdc++ <math.h> wrapper is in the TU via ros/tf headers):
 * atanf(...) * 180 is an f32 product, "/ M_PI" promotes to f64, the result is stored as f32. */
static int scan_id_of(float x, float y, float z, const orc_params *P, float factor)
{
    const float angle = (float)((double)(atanf(z / sqrtf(x * x + y * y)) * 180.0f) / M_PI);
    if (P->ring_model == 0 && P->n_scans == 16)
        return trunc_to_int((double)((angle + 15.0f) / 2.0f) + 0.5);                        /* :144 */
    if (P->ring_model == 0 && P->n_scans == 32)
        return trunc_to_int(((double)angle + 92.0 / 3.0) * 3.0 / 4.0);                      /* :153 */
    return trunc_to_int((double)((angle - P->lower_bound) * factor) + 0.5);                 /* :162 */
}

/* test helper: the ring of every input point as the loop of :130-168 computes it (-1 = rejected), no filtering */
void orc_scan_ids(const float *xyz, int stride, int n, const orc_params *P, int *ids)
{
    const float factor = (float)(P->n_scans - 1) / (P->up_bound - P->lower_bound);           /* :441 */
    for (int i = 0; i < n; ++i) {
        const int id = scan_id_of(xyz[(size_t)i * stride], xyz[(size_t)i * stride + 1], xyz[(size_t)i * stride + 2], P, factor);
        ids[i] = (id > P->n_scans - 1 || id < 0) ? -1 : id;
    }
}

/* test helper: the HOST libm (what the reference links) over arrays, for the device-vs-glibc check of the restated
 * atanf / atan2f (tests/test_gpu_a1_edges.py).  op 0: atanf(a); 1: atan2f(a, b); 2: (float)((double)a / M_PI);
 * 3: a / sqrtf(b * b + c * c) (the argument of :139); 4: expf(a) */
void orc_libm_batch(int op, const float *a, const float *b, const float *c, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        switch (op) {
        case 0: out[i] = atanf(a[i]); break;
        case 1: out[i] = atan2f(a[i], b[i]); break;
        case 2: out[i] = (float)((double)a[i] / M_PI); break;
        case 3: out[i] = a[i] / sqrtf(b[i] * b[i] + c[i] * c[i]); break;
        default: out[i] = expf(a[i]); break;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a1  scanRegistration.cpp:58-85 (removeClosedPointCloud), :105-221                           */
/* ------------------------------------------------------------------------------------------ */
int orc_organize(const float *xyz, int stride, int n_in, const orc_params *P,
                 orc_point *cloud, int *n_out, int *scan_start, int *scan_end)
{
    const int N_SCANS = P->n_scans;
    if (P->ring_model == 0 && N_SCANS != 16 && N_SCANS != 32 && N_SCANS != 64)
        return ORC_ERR_BAD_RINGS;                       /* :447-451, :170-174 */
    if (N_SCANS < 1) return ORC_ERR_BAD_RINGS;

    float *kept = (float *)malloc((size_t)(n_in > 0 ? n_in : 1) * 3 * sizeof(float));
    int cloudSize = 0;
    /* pcl::removeNaNFromPointCloud (:109) then removeClosedPointCloud(thres = MINIMUM_RANGE) (:110) */
    const float thres = (float)P->minimum_range;        /* double -> float parameter conversion (:60, :110) */
    for (int i = 0; i < n_in; ++i) {
        const float x = xyz[(size_t)i * stride], y = xyz[(size_t)i * stride + 1], z = xyz[(size_t)i * stride + 2];
        if (!isfinite(x) || !isfinite(y) || !isfinite(z)) continue;
        if (x * x + y * y + z * z < thres * thres) continue;   /* :72, all f32 */
        kept[3 * cloudSize] = x; kept[3 * cloudSize + 1] = y; kept[3 * cloudSize + 2] = z;
        cloudSize++;
    }
    if (cloudSize == 0) { free(kept); *n_out = 0; return ORC_ERR_EMPTY; }   /* reference would read points[0] of an empty cloud */

    /* :114-126 */
    float startOri = -atan2f(kept[1], kept[0]);
    float endOri = (float)((double)(-atan2f(kept[3 * (cloudSize - 1) + 1], kept[3 * (cloudSize - 1)])) + 2 * M_PI);
    if ((double)(endOri - startOri) > 3 * M_PI)      endOri = (float)((double)endOri - 2 * M_PI);
    else if ((double)(endOri - startOri) < M_PI)     endOri = (float)((double)endOri + 2 * M_PI);

    /* _factor = (N_SCANS-1) / (upBound - lowerBound)   (:441, int / float -> float) */
    const float factor = (float)(N_SCANS - 1) / (P->up_bound - P->lower_bound);

    int *ring = (int *)malloc((size_t)cloudSize * sizeof(int));
    float *inten = (float *)malloc((size_t)cloudSize * sizeof(float));
    int *cnt = (int *)calloc((size_t)N_SCANS + 1, sizeof(int));
    int halfPassed = 0;
    int count = cloudSize;
    for (int i = 0; i < cloudSize; ++i) {
        const float x = kept[3 * i], y = kept[3 * i + 1], z = kept[3 * i + 2];
        const int scanID = scan_id_of(x, y, z, P, factor);
        if (scanID > (N_SCANS - 1) || scanID < 0) { ring[i] = -1; count--; continue; }      /* :145-149, :164-168 */

        float ori = -atan2f(y, x);                                                            /* :177 */
        if (!halfPassed) {
            if ((double)ori < (double)startOri - M_PI / 2)            ori = (float)((double)ori + 2 * M_PI);
            else if ((double)ori > (double)startOri + M_PI * 3 / 2)   ori = (float)((double)ori - 2 * M_PI);
            if ((double)(ori - startOri) > M_PI) halfPassed = 1;
        } else {
            ori = (float)((double)ori + 2 * M_PI);
            if ((double)ori < (double)endOri - M_PI * 3 / 2)          ori = (float)((double)ori + 2 * M_PI);
            else if ((double)ori > (double)endOri + M_PI / 2)         ori = (float)((double)ori - 2 * M_PI);
        }
        const float relTime = (ori - startOri) / (endOri - startOri);                         /* :207 f32 */
        inten[i] = (float)((double)scanID + 0.1 * (double)relTime);                           /* :208 scanPeriod = 0.1 (double) */
        ring[i] = scanID;
        cnt[scanID + 1]++;
    }
    /* stable bucket by ring + concat (:209, :215-221) */
    for (int r = 0; r < N_SCANS; ++r) cnt[r + 1] += cnt[r];
    for (int r = 0; r < N_SCANS; ++r) {
        scan_start[r] = cnt[r] + 5;
        scan_end[r] = cnt[r + 1] - 6;
    }
    int *cur = (int *)malloc((size_t)N_SCANS * sizeof(int));
    memcpy(cur, cnt, (size_t)N_SCANS * sizeof(int));
    for (int i = 0; i < cloudSize; ++i) {
        if (ring[i] < 0) continue;
        orc_point *p = &cloud[cur[ring[i]]++];
        p->x = kept[3 * i]; p->y = kept[3 * i + 1]; p->z = kept[3 * i + 2]; p->intensity = inten[i];
    }
    *n_out = count;
    free(cur); free(cnt); free(inten); free(ring); free(kept);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a2  scanRegistration.cpp:225-235                                                            */
/* ------------------------------------------------------------------------------------------ */
void orc_curvature(const orc_point *c, int n, float *curv)
{
    for (int i = 5; i < n - 5; i++) {
        /* strictly left-to-right f32, "10 * p.x" is int->float then f32 product (:228-230) */
        float diffX = c[i - 5].x + c[i - 4].x + c[i - 3].x + c[i - 2].x + c[i - 1].x - 10 * c[i].x + c[i + 1].x + c[i + 2].x + c[i + 3].x + c[i + 4].x + c[i + 5].x;
        float diffY = c[i - 5].y + c[i - 4].y + c[i - 3].y + c[i - 2].y + c[i - 1].y - 10 * c[i].y + c[i + 1].y + c[i + 2].y + c[i + 3].y + c[i + 4].y + c[i + 5].y;
        float diffZ = c[i - 5].z + c[i - 4].z + c[i - 3].z + c[i - 2].z + c[i - 1].z - 10 * c[i].z + c[i + 1].z + c[i + 2].z + c[i + 3].z + c[i + 4].z + c[i + 5].z;
        curv[i] = diffX * diffX + diffY * diffY + diffZ * diffZ;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a4  pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.10 voxel_grid.hpp), restated              */
/*     call site scanRegistration.cpp:370-376, leaf (0.2,0.2,0.2), downsample_all_data_ = true */
/* ------------------------------------------------------------------------------------------ */
typedef struct { unsigned int idx; int pos; } vox_key;
static int vox_cmp(const void *a, const void *b)
{
    const vox_key *ka = (const vox_key *)a, *kb = (const vox_key *)b;
    if (ka->idx != kb->idx) return ka->idx < kb->idx ? -1 : 1;
    /* PCL uses an unstable std::sort on idx only; the order inside a voxel (hence the f32 summation
     * order of the centroid) is unspecified there.  Oracle and HIP path define it: input order. */
    return (ka->pos > kb->pos) - (ka->pos < kb->pos);
}

int orc_voxel_grid(const orc_point *in, int n, float leaf, orc_point *out, int *n_out)
{
    if (n <= 0) { *n_out = 0; return ORC_OK; }
    const float inv = 1.0f / leaf;                       /* inverse_leaf_size_ = Array4f::Ones() / leaf_size_ */
    float mn[3] = {in[0].x, in[0].y, in[0].z}, mx[3] = {in[0].x, in[0].y, in[0].z};
    for (int i = 1; i < n; ++i) {                        /* getMinMax3D */
        const float p[3] = {in[i].x, in[i].y, in[i].z};
        for (int k = 0; k < 3; ++k) { if (p[k] < mn[k]) mn[k] = p[k]; if (p[k] > mx[k]) mx[k] = p[k]; }
    }
    int64_t d[3];
    for (int k = 0; k < 3; ++k) d[k] = (int64_t)((mx[k] - mn[k]) * inv) + 1;
    if (d[0] * d[1] * d[2] > (int64_t)INT_MAX) {         /* "Leaf size is too small": output = *input_ */
        memcpy(out, in, (size_t)n * sizeof(orc_point)); *n_out = n; return ORC_OK;
    }
    int min_b[3], max_b[3], div_b[3], mul[3];
    for (int k = 0; k < 3; ++k) {
        min_b[k] = (int)floorf(mn[k] * inv);
        max_b[k] = (int)floorf(mx[k] * inv);
        div_b[k] = max_b[k] - min_b[k] + 1;
    }
    if ((int64_t)div_b[0] * div_b[1] * div_b[2] > (int64_t)0xffffffffLL) {   /* PCL's int index would wrap here (undefined in the reference;
                                                                              * coordinates ~1e8 m apart): defined as the exit above, like the HIP path */
        memcpy(out, in, (size_t)n * sizeof(orc_point)); *n_out = n; return ORC_OK;
    }
    mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];
    vox_key *keys = (vox_key *)malloc((size_t)n * sizeof(vox_key));
    for (int i = 0; i < n; ++i) {
        const int ijk0 = (int)(floorf(in[i].x * inv) - (float)min_b[0]);
        const int ijk1 = (int)(floorf(in[i].y * inv) - (float)min_b[1]);
        const int ijk2 = (int)(floorf(in[i].z * inv) - (float)min_b[2]);
        keys[i].idx = (unsigned int)(ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2]);
        keys[i].pos = i;
    }
    qsort(keys, (size_t)n, sizeof(vox_key), vox_cmp);
    int m = 0;
    for (int first = 0; first < n;) {
        int last = first + 1;
        while (last < n && keys[last].idx == keys[first].idx) ++last;
        /* CentroidPoint<PointXYZI>: AccumulatorXYZ (Vector3f sum, get = sum / n) + AccumulatorIntensity */
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
        for (int k = first; k < last; ++k) {
            const orc_point *p = &in[keys[k].pos];
            sx += p->x; sy += p->y; sz += p->z; si += p->intensity;
        }
        const float cnt = (float)(last - first);
        out[m].x = sx / cnt; out[m].y = sy / cnt; out[m].z = sz / cnt; out[m].intensity = si / cnt;
        ++m; first = last;
    }
    free(keys);
    *n_out = m;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a3  scanRegistration.cpp:42 (comp), :246-377                                                */
/* ------------------------------------------------------------------------------------------ */
static const float *g_curv_for_sort;
static int curv_cmp(const void *a, const void *b)
{
    const int ia = *(const int *)a, ib = *(const int *)b;
    const float ca = g_curv_for_sort[ia], cb = g_curv_for_sort[ib];
    if (ca < cb) return -1;
    if (ca > cb) return 1;
    /* comp() (:42) orders by curvature only and std::sort is unstable: order of equal curvatures is
     * unspecified in the reference.  Oracle and HIP path define it: ascending point index. */
    return (ia > ib) - (ia < ib);
}

static float gap2(const orc_point *c, int a, int b)
{
    const float dx = c[a].x - c[b].x, dy = c[a].y - c[b].y, dz = c[a].z - c[b].z;
    return dx * dx + dy * dy + dz * dz;
}

static void mark_neighbours(const orc_point *c, int ind, int *picked)
{
    for (int l = 1; l <= 5; l++) {                               /* :288-299 */
        if ((double)gap2(c, ind + l, ind + l - 1) > 0.05) break;
        picked[ind + l] = 1;
    }
    for (int l = -1; l >= -5; l--) {                             /* :300-311 */
        if ((double)gap2(c, ind + l, ind + l + 1) > 0.05) break;
        picked[ind + l] = 1;
    }
}

int orc_pick(const orc_point *cloud, int n, const float *curv,
             const int *scan_start, const int *scan_end, int n_scans, int *label,
             orc_point *sharp, int *n_sharp, orc_point *less_sharp, int *n_less_sharp,
             orc_point *flat, int *n_flat, orc_point *less_flat, int *n_less_flat, int *tie_count)
{
    int *sortInd = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    int *picked = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
    orc_point *lf_scan = (orc_point *)malloc((size_t)(n > 0 ? n : 1) * sizeof(orc_point));
    for (int i = 0; i < n; ++i) sortInd[i] = i;
    for (int i = 5; i < n - 5; ++i) label[i] = 0;                /* :232-234 */
    int ns = 0, nls = 0, nf = 0, nlf = 0, ties = 0;

    for (int i = 0; i < n_scans; i++) {
        if (scan_end[i] - scan_start[i] < 6) continue;           /* :248 */
        int n_lf_scan = 0;
        for (int j = 0; j < 6; j++) {
            const int sp = scan_start[i] + (scan_end[i] - scan_start[i]) * j / 6;           /* :253 */
            const int ep = scan_start[i] + (scan_end[i] - scan_start[i]) * (j + 1) / 6 - 1; /* :254 */
            g_curv_for_sort = curv;
            qsort(sortInd + sp, (size_t)(ep - sp + 1), sizeof(int), curv_cmp);               /* :257 */
            for (int k = sp; k < ep; ++k) if (curv[sortInd[k]] == curv[sortInd[k + 1]]) ties++;

            int largestPickedNum = 0;
            for (int k = ep; k >= sp; k--) {                     /* :261-313 */
                const int ind = sortInd[k];
                if (picked[ind] == 0 && (double)curv[ind] > 0.1) {
                    largestPickedNum++;
                    if (largestPickedNum <= 2) {
                        label[ind] = 2;
                        sharp[ns++] = cloud[ind];
                        less_sharp[nls++] = cloud[ind];
                    } else if (largestPickedNum <= 20) {
                        label[ind] = 1;
                        less_sharp[nls++] = cloud[ind];
                    } else {
                        break;
                    }
                    picked[ind] = 1;
                    mark_neighbours(cloud, ind, picked);
                }
            }
            int smallestPickedNum = 0;
            for (int k = sp; k <= ep; k++) {                     /* :316-359 */
                const int ind = sortInd[k];
                if (picked[ind] == 0 && (double)curv[ind] < 0.1) {
                    label[ind] = -1;
                    flat[nf++] = cloud[ind];
                    smallestPickedNum++;
                    if (smallestPickedNum >= 4) break;           /* :328-331: breaks BEFORE marking */
                    picked[ind] = 1;
                    mark_neighbours(cloud, ind, picked);
                }
            }
            for (int k = sp; k <= ep; k++)                       /* :361-367 */
                if (label[k] <= 0) lf_scan[n_lf_scan++] = cloud[k];
        }
        int m = 0;
        orc_voxel_grid(lf_scan, n_lf_scan, 0.2f, less_flat + nlf, &m);   /* :370-376 */
        nlf += m;
    }
    *n_sharp = ns; *n_less_sharp = nls; *n_flat = nf; *n_less_flat = nlf;
    if (tie_count) *tie_count = ties;
    free(lf_scan); free(picked); free(sortInd);
    return ORC_OK;
}

int orc_extract(const float *xyz, int stride, int n_in, const orc_params *P,
                orc_point *cloud, int *n_out, int *scan_start, int *scan_end,
                float *curv, int *label,
                orc_point *sharp, int *n_sharp, orc_point *less_sharp, int *n_less_sharp,
                orc_point *flat, int *n_flat, orc_point *less_flat, int *n_less_flat)
{
    int rc = orc_organize(xyz, stride, n_in, P, cloud, n_out, scan_start, scan_end);
    if (rc != ORC_OK) return rc;
    orc_curvature(cloud, *n_out, curv);
    return orc_pick(cloud, *n_out, curv, scan_start, scan_end, P->n_scans, label,
                    sharp, n_sharp, less_sharp, n_less_sharp, flat, n_flat, less_flat, n_less_flat, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* a5  laserOdometry.cpp:77-95 with DISTORTION 0 (:23)                                         */
/* ------------------------------------------------------------------------------------------ */
/* Eigen::QuaternionBase::_transformVector: uv = u x v; uv += uv; v + w*uv + u x uv */
static void quat_rotate(const double q[4], const double v[3], double out[3])
{
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    double uvx = uy * v[2] - uz * v[1];
    double uvy = uz * v[0] - ux * v[2];
    double uvz = ux * v[1] - uy * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    out[0] = (v[0] + w * uvx) + (uy * uvz - uz * uvy);
    out[1] = (v[1] + w * uvy) + (uz * uvx - ux * uvz);
    out[2] = (v[2] + w * uvz) + (ux * uvy - uy * uvx);
}

/* DISTORTION (laserOdometry.cpp:23): the reference is built with 0; 1 is its other compile-time path (:81-82, :570-571, :740-741) */
static int g_distortion = 0;
void orc_set_distortion(int on) { g_distortion = on ? 1 : 0; }
int orc_get_distortion(void) { return g_distortion; }

/* the interpolation ratio of a point (:81-84): float intensity - int(intensity) in f32, / SCAN_PERIOD (0.1) in f64 */
double orc_point_s(const orc_point *p)
{
    if (!g_distortion) return 1.0;
    return (double)(p->intensity - (float)(int)p->intensity) / 0.1;
}

/* Eigen 3.3 QuaternionBase::slerp(t, other) called on Identity, plain doubles (the Jet form is slerp_from_identity below) */
static void slerp_from_identity_d(double t, const double o[4] /* x y z w */, double out[4])
{
    const double one = 1.0 - 2.220446049250313e-16;
    const double d = (0.0 * o[0] + 0.0 * o[1]) + (0.0 * o[2] + 1.0 * o[3]);
    const double absD = fabs(d);
    double scale0, scale1;
    if (absD >= one) { scale0 = 1.0 - t; scale1 = t; }
    else {
        const double theta = acos(absD), sinTheta = sin(theta);
        scale0 = sin((1.0 - t) * theta) / sinTheta;
        scale1 = sin(t * theta) / sinTheta;
    }
    if (d < 0.0) scale1 = -scale1;
    out[0] = scale0 * 0.0 + scale1 * o[0]; out[1] = scale0 * 0.0 + scale1 * o[1];
    out[2] = scale0 * 0.0 + scale1 * o[2]; out[3] = scale0 * 1.0 + scale1 * o[3];
}

void orc_transform_to_start(const double q[4], const double t[3], const orc_point *pi, orc_point *po)
{
    const double v[3] = {(double)pi->x, (double)pi->y, (double)pi->z};
    double r[3];
    if (!g_distortion) {
        /* s = 1.0 (:84).  Identity.slerp(1, q) (Eigen 3.3): scale0 = 0 and scale1 = +-1 exactly in both
         * branches (1-t = 0; sin(theta)/sin(theta) = 1), so q_point_last = +-q and the rotation formula,
         * even in (u,w), gives bit-identical results for q and -q.  t_point_last = 1.0 * t = t. */
        quat_rotate(q, v, r);
        po->x = (float)(r[0] + t[0]);
        po->y = (float)(r[1] + t[1]);
        po->z = (float)(r[2] + t[2]);
    } else {
        const double s = orc_point_s(pi);                          /* :82 */
        double qs[4];
        slerp_from_identity_d(s, q, qs);                            /* :86 */
        quat_rotate(qs, v, r);
        po->x = (float)(r[0] + s * t[0]);                           /* :87, :89 */
        po->y = (float)(r[1] + s * t[1]);
        po->z = (float)(r[2] + s * t[2]);
    }
    po->intensity = pi->intensity;
}

/* FLANN L2_Simple<float> (PCL KdTreeFLANN distance): result = 0; for d: diff = a[d]-b[d]; result += diff*diff */
static float flann_l2(const orc_point *a, const orc_point *b)
{
    float result = 0.0f, diff;
    diff = a->x - b->x; result += diff * diff;
    diff = a->y - b->y; result += diff * diff;
    diff = a->z - b->z; result += diff * diff;
    return result;
}

/* exact K=1 nearest neighbour (kd-tree search is exact; equal-distance winner is traversal-dependent
 * in FLANN, defined here and in the HIP path as the lowest index). */
static int nn1_brute(const orc_point *q, const orc_point *cloud, int m, float *dist)
{
    int best = -1; float bd = INFINITY;
    for (int j = 0; j < m; ++j) {
        const float d = flann_l2(q, &cloud[j]);
        if (d < bd) { bd = d; best = j; }
    }
    *dist = bd;
    return best;
}

/* Same answer for every query whose nearest neighbour is closer than sqrt(r2max) -- the only case the
 * callers use (:497 / :659 reject d >= 25) -- from a uniform grid instead of a linear scan, so that the timed
 * CPU baseline has kd-tree-class cost like the reference (PCL KdTreeFLANN).  Candidates are compared with the
 * same f32 flann_l2 value and the same lowest-index tie rule; cells are pruned only when every point in them
 * is provably farther (in double, with margin) than the current best. */
typedef struct { float cell; float mn[3]; int dim[3]; int *start; int *idx; int m; const orc_point *cloud; } nn_grid;
static int g_nn_mode = 0;               /* 0 = linear scan, 1 = grid */
void orc_set_nn_mode(int mode) { g_nn_mode = mode; }

static void grid_cell_of(const nn_grid *g, const orc_point *p, int c[3])
{
    const float v[3] = {p->x, p->y, p->z};
    for (int k = 0; k < 3; ++k) {
        int i = (int)floorf((v[k] - g->mn[k]) / g->cell);
        if (i < 0) i = 0;
        if (i >= g->dim[k]) i = g->dim[k] - 1;
        c[k] = i;
    }
}

static nn_grid *grid_build(const orc_point *cloud, int m, float cell)
{
    nn_grid *g = (nn_grid *)calloc(1, sizeof(nn_grid));
    g->cell = cell; g->m = m; g->cloud = cloud;
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    g->mn[0] = g->mn[1] = g->mn[2] = INFINITY;
    for (int i = 0; i < m; ++i) {
        const float v[3] = {cloud[i].x, cloud[i].y, cloud[i].z};
        for (int k = 0; k < 3; ++k) { if (v[k] < g->mn[k]) g->mn[k] = v[k]; if (v[k] > mx[k]) mx[k] = v[k]; }
    }
    for (int k = 0; k < 3; ++k) {
        g->dim[k] = (m > 0) ? (int)floorf((mx[k] - g->mn[k]) / cell) + 1 : 1;
        if (g->dim[k] > 512) { g->dim[k] = 512; }
    }
    /* if an axis was clamped the cell index saturates; correctness is kept by the bounds-based pruning below */
    const size_t nc = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (int *)calloc(nc + 1, sizeof(int));
    g->idx = (int *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int));
    int *cellof = (int *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int));
    for (int i = 0; i < m; ++i) {
        int c[3]; grid_cell_of(g, &cloud[i], c);
        cellof[i] = (c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0];
        g->start[cellof[i] + 1]++;
    }
    for (size_t c = 0; c < nc; ++c) g->start[c + 1] += g->start[c];
    int *cur = (int *)malloc(nc * sizeof(int));
    memcpy(cur, g->start, nc * sizeof(int));
    for (int i = 0; i < m; ++i) g->idx[cur[cellof[i]]++] = i;      /* ascending index inside a cell */
    free(cur); free(cellof);
    return g;
}

static void grid_free(nn_grid *g) { if (g) { free(g->start); free(g->idx); free(g); } }

static int nn1_grid(const nn_grid *g, const orc_point *q, float r2max, float *dist)
{
    int best = -1; float bd = INFINITY;
    if (g->m == 0) { *dist = bd; return -1; }
    int c0[3]; grid_cell_of(g, q, c0);
    const int rad = (int)ceilf(sqrtf(r2max) / g->cell) + 1;
    const double qv[3] = {q->x, q->y, q->z};
    for (int ring = 0; ring <= rad; ++ring) {
        /* every cell of Chebyshev ring `ring` is at least (ring-1)*cell away; stop once that exceeds the best */
        if (ring >= 2) { const double lb = (double)(ring - 1) * g->cell; if (lb * lb * 0.999 > (double)(bd < r2max ? bd : r2max)) break; }
        for (int dz = -ring; dz <= ring; ++dz) for (int dy = -ring; dy <= ring; ++dy) for (int dx = -ring; dx <= ring; ++dx) {
            const int ad = (abs(dx) > abs(dy) ? abs(dx) : abs(dy)); const int cheb = ad > abs(dz) ? ad : abs(dz);
            if (cheb != ring) continue;
            const int cx = c0[0] + dx, cy = c0[1] + dy, cz = c0[2] + dz;
            if (cx < 0 || cy < 0 || cz < 0 || cx >= g->dim[0] || cy >= g->dim[1] || cz >= g->dim[2]) continue;
            const size_t c = ((size_t)cz * g->dim[1] + cy) * g->dim[0] + cx;
            for (int k = g->start[c]; k < g->start[c + 1]; ++k) {
                const int j = g->idx[k];
                const float d = flann_l2(q, &g->cloud[j]);
                if (d < bd || (d == bd && j < best)) { bd = d; best = j; }
            }
        }
    }
    (void)qv;
    /* saturated (clamped) border cells can hold far points only; a query outside the grid by more than the
     * search radius simply finds nothing closer than r2max, which the callers treat as "no match" */
    *dist = bd;
    return best;
}

static int nn1(const orc_point *q, const orc_point *cloud, int m, const nn_grid *g, float *dist)
{
    if (g) {
        int j = nn1_grid(g, q, 25.0f, dist);
        if (j >= 0 && *dist < 25.0f) return j;
        *dist = INFINITY;            /* nothing within the acceptance radius: callers reject either way */
        return -1;
    }
    return nn1_brute(q, cloud, m, dist);
}

/* (points[j].x - pointSel.x) * (...) + ... : all f32, widened to double on assignment (:514-519) */
static double walk_d2(const orc_point *p, const orc_point *sel)
{
    return (double)((p->x - sel->x) * (p->x - sel->x) + (p->y - sel->y) * (p->y - sel->y) + (p->z - sel->z) * (p->z - sel->z));
}

/* ------------------------------------------------------------------------------------------ */
/* a6  laserOdometry.cpp:491-620                                                               */
/* ------------------------------------------------------------------------------------------ */
int orc_associate_corner(const double q[4], const double t[3], const orc_point *sharp, int ns,
                         const orc_point *last, int mc, int *src_idx, int *idx_a, int *idx_b, int *n_e)
{
    int ne = 0;
    nn_grid *grid = (g_nn_mode && mc > 0) ? grid_build(last, mc, 1.0f) : NULL;
    for (int i = 0; i < ns && mc > 0; ++i) {
        orc_point sel;
        orc_transform_to_start(q, t, &sharp[i], &sel);
        float d0;
        const int nn = nn1(&sel, last, mc, grid, &d0);
        int closestPointInd = -1, minPointInd2 = -1;
        if ((double)d0 < 25.0) {                                  /* DISTANCE_SQ_THRESHOLD (:29, :497) */
            closestPointInd = nn;
            const int closestPointScanID = (int)last[closestPointInd].intensity;
            double minPointSqDis2 = 25.0;
            for (int j = closestPointInd + 1; j < mc; ++j) {      /* :504-527 */
                if ((int)last[j].intensity <= closestPointScanID) continue;
                if ((double)(int)last[j].intensity > (closestPointScanID + 2.5)) break;
                const double d = walk_d2(&last[j], &sel);
                if (d < minPointSqDis2) { minPointSqDis2 = d; minPointInd2 = j; }
            }
            for (int j = closestPointInd - 1; j >= 0; --j) {      /* :530-553 */
                if ((int)last[j].intensity >= closestPointScanID) continue;
                if ((double)(int)last[j].intensity < (closestPointScanID - 2.5)) break;
                const double d = walk_d2(&last[j], &sel);
                if (d < minPointSqDis2) { minPointSqDis2 = d; minPointInd2 = j; }
            }
        }
        if (minPointInd2 >= 0) {                                  /* :556 */
            src_idx[ne] = i; idx_a[ne] = closestPointInd; idx_b[ne] = minPointInd2; ne++;
        }
    }
    *n_e = ne;
    grid_free(grid);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a7  laserOdometry.cpp:653-793                                                               */
/* ------------------------------------------------------------------------------------------ */
int orc_associate_plane(const double q[4], const double t[3], const orc_point *flat, int nf,
                        const orc_point *last, int ms, int *src_idx, int *idx_a, int *idx_b, int *idx_c, int *n_p)
{
    int np = 0;
    nn_grid *grid = (g_nn_mode && ms > 0) ? grid_build(last, ms, 1.0f) : NULL;
    for (int i = 0; i < nf && ms > 0; ++i) {
        orc_point sel;
        orc_transform_to_start(q, t, &flat[i], &sel);
        float d0;
        const int nn = nn1(&sel, last, ms, grid, &d0);
        int closestPointInd = -1, minPointInd2 = -1, minPointInd3 = -1;
        if ((double)d0 < 25.0) {
            closestPointInd = nn;
            const int closestPointScanID = (int)last[closestPointInd].intensity;
            double minPointSqDis2 = 25.0, minPointSqDis3 = 25.0;
            for (int j = closestPointInd + 1; j < ms; ++j) {      /* :668-693 */
                if ((double)(int)last[j].intensity > (closestPointScanID + 2.5)) break;
                const double d = walk_d2(&last[j], &sel);
                if ((int)last[j].intensity <= closestPointScanID && d < minPointSqDis2) { minPointSqDis2 = d; minPointInd2 = j; }
                else if ((int)last[j].intensity > closestPointScanID && d < minPointSqDis3) { minPointSqDis3 = d; minPointInd3 = j; }
            }
            for (int j = closestPointInd - 1; j >= 0; --j) {      /* :696-721 */
                if ((double)(int)last[j].intensity < (closestPointScanID - 2.5)) break;
                const double d = walk_d2(&last[j], &sel);
                if ((int)last[j].intensity >= closestPointScanID && d < minPointSqDis2) { minPointSqDis2 = d; minPointInd2 = j; }
                else if ((int)last[j].intensity < closestPointScanID && d < minPointSqDis3) { minPointSqDis3 = d; minPointInd3 = j; }
            }
            if (minPointInd2 >= 0 && minPointInd3 >= 0) {         /* :723 */
                src_idx[np] = i; idx_a[np] = closestPointInd; idx_b[np] = minPointInd2; idx_c[np] = minPointInd3; np++;
            }
        }
    }
    *n_p = np;
    grid_free(grid);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a8  laserOdometry.cpp:153-342                                                               */
/* ------------------------------------------------------------------------------------------ */
static float Distance(const orc_point *a, const orc_point *b)      /* :153-162 */
{
    const float dx = a->x - b->x, dy = a->y - b->y, dz = a->z - b->z;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

typedef struct { int index; float score; } vertex_vote;           /* common.h:40-43 */
static int vote_cmp_canonical(const void *a, const void *b)
{
    const vertex_vote *va = (const vertex_vote *)a, *vb = (const vertex_vote *)b;
    if (va->score != vb->score) return va->score < vb->score ? -1 : 1;
    return (va->index > vb->index) - (va->index < vb->index);
}

void orc_vote(const orc_point *src, const orc_point *tgt, int n, int corner_case,
              int *counts, int *sel_idx, float *sel_w, int *n_sel)
{
    const int number_of_region = corner_case ? 5 : 10;            /* :179-188 */
    const float score_threshold = 0.96f;
    int nsel = 0;
    for (int num_region = 0; num_region < number_of_region; num_region++) {
        const int initial_pos = n / number_of_region * num_region;                        /* :202 */
        const int end_pos = (num_region == number_of_region - 1) ? n : n / number_of_region * (num_region + 1);
        const int cor_size = end_pos - initial_pos;
        if (cor_size <= 0) continue;
        vertex_vote *vote_record = (vertex_vote *)malloc((size_t)cor_size * sizeof(vertex_vote));
        for (int i = 0; i < cor_size; ++i) { vote_record[i].index = i; vote_record[i].score = 0.0f; }
        for (int i = 0; i < cor_size; i++) {                      /* :228-252 */
            for (int j = i + 1; j < cor_size; j++) {
                const float s1 = Distance(&src[initial_pos + i], &src[initial_pos + j]);
                const float s2 = Distance(&tgt[initial_pos + i], &tgt[initial_pos + j]);
                const float dis_gap = fabsf(s1 - s2);
                const float score = expf(-(dis_gap * dis_gap) / (1.0f * 1.0f));          /* std::exp(float) */
                if (score < score_threshold) { vote_record[j].score += 1; vote_record[i].score += 1; }
            }
        }
        for (int i = 0; i < cor_size; ++i) counts[initial_pos + i] = (int)vote_record[i].score;
        /* reference: std::sort descending (:255) then walk from the low-count end (:304-329).
         * canonical order here: ascending (count, index). */
        qsort(vote_record, (size_t)cor_size, sizeof(vertex_vote), vote_cmp_canonical);
        const float num_selected = 0.90f * (float)cor_size;       /* :299-300 */
        for (int i = 0; i < cor_size; ++i) {
            if (vote_record[i].score > num_selected) break;       /* :312-316 */
            sel_idx[nsel] = initial_pos + vote_record[i].index;
            sel_w[nsel] = (vote_record[i].score <= 50.0f) ? 5.0f : 1.0f;                  /* :317-322 */
            nsel++;
        }
        free(vote_record);
    }
    *n_sel = nsel;
}

/* ------------------------------------------------------------------------------------------ */
/* a9  lidarFactor.hpp under ceres::AutoDiffCostFunction<., R, 4, 3>: forward Jets of width 7   */
/* ------------------------------------------------------------------------------------------ */
#define JN 7
typedef struct { double a; double v[JN]; } jet;

static jet J_const(double a) { jet r; r.a = a; for (int k = 0; k < JN; ++k) r.v[k] = 0.0; return r; }
static jet J_var(double a, int k) { jet r = J_const(a); r.v[k] = 1.0; return r; }
static jet J_add(jet f, jet g) { jet r; r.a = f.a + g.a; for (int k = 0; k < JN; ++k) r.v[k] = f.v[k] + g.v[k]; return r; }
static jet J_sub(jet f, jet g) { jet r; r.a = f.a - g.a; for (int k = 0; k < JN; ++k) r.v[k] = f.v[k] - g.v[k]; return r; }
static jet J_neg(jet f) { jet r; r.a = -f.a; for (int k = 0; k < JN; ++k) r.v[k] = -f.v[k]; return r; }
/* ceres/jet.h: f*g = (f.a*g.a, f.a*g.v + f.v*g.a) */
static jet J_mul(jet f, jet g) { jet r; r.a = f.a * g.a; for (int k = 0; k < JN; ++k) r.v[k] = f.a * g.v[k] + f.v[k] * g.a; return r; }
/* ceres/jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a*g_a_inverse; (f_a_by_g_a, (f.v - f_a_by_g_a*g.v)*g_a_inverse) */
static jet J_div(jet f, jet g)
{
    jet r; const double gi = 1.0 / g.a; const double fg = f.a * gi;
    r.a = fg; for (int k = 0; k < JN; ++k) r.v[k] = (f.v[k] - fg * g.v[k]) * gi; return r;
}
static jet J_sqrt(jet f)
{
    jet r; const double tmp = sqrt(f.a); const double two_a_inverse = 1.0 / (2.0 * tmp);
    r.a = tmp; for (int k = 0; k < JN; ++k) r.v[k] = f.v[k] * two_a_inverse; return r;
}
static jet J_acos(jet f)
{
    jet r; const double tmp = -1.0 / sqrt(1.0 - f.a * f.a);
    r.a = acos(f.a); for (int k = 0; k < JN; ++k) r.v[k] = tmp * f.v[k]; return r;
}
static jet J_sin(jet f)
{
    jet r; const double c = cos(f.a);
    r.a = sin(f.a); for (int k = 0; k < JN; ++k) r.v[k] = c * f.v[k]; return r;
}

typedef struct { jet x, y, z, w; } jquat;      /* Eigen coeffs order x,y,z,w */
typedef struct { jet x, y, z; } jvec;

/* Eigen 3.3 QuaternionBase::slerp(t, other) called on Identity (lidarFactor.hpp:25-26) */
static jquat slerp_from_identity(jet t, jquat o)
{
    const double one = 1.0 - 2.220446049250313e-16;
    const jet I0 = J_const(0.0), I1 = J_const(1.0);
    /* d = coeffs().dot(other.coeffs()), 4-term novec tree: (x*ox + y*oy) + (z*oz + w*ow) */
    jet d = J_add(J_add(J_mul(I0, o.x), J_mul(I0, o.y)), J_add(J_mul(I0, o.z), J_mul(I1, o.w)));
    jet absD = d.a < 0.0 ? J_neg(d) : d;
    jet scale0, scale1;
    if (absD.a >= one) {
        scale0 = J_sub(J_const(1.0), t);
        scale1 = t;
    } else {
        jet theta = J_acos(absD);
        jet sinTheta = J_sin(theta);
        scale0 = J_div(J_sin(J_mul(J_sub(J_const(1.0), t), theta)), sinTheta);
        scale1 = J_div(J_sin(J_mul(t, theta)), sinTheta);
    }
    if (d.a < 0.0) scale1 = J_neg(scale1);
    jquat r;
    r.x = J_add(J_mul(scale0, I0), J_mul(scale1, o.x));
    r.y = J_add(J_mul(scale0, I0), J_mul(scale1, o.y));
    r.z = J_add(J_mul(scale0, I0), J_mul(scale1, o.z));
    r.w = J_add(J_mul(scale0, I1), J_mul(scale1, o.w));
    return r;
}

static jvec jcross(jvec a, jvec b)
{
    jvec r;
    r.x = J_sub(J_mul(a.y, b.z), J_mul(a.z, b.y));
    r.y = J_sub(J_mul(a.z, b.x), J_mul(a.x, b.z));
    r.z = J_sub(J_mul(a.x, b.y), J_mul(a.y, b.x));
    return r;
}
static jvec jvsub(jvec a, jvec b) { jvec r = {J_sub(a.x, b.x), J_sub(a.y, b.y), J_sub(a.z, b.z)}; return r; }
static jvec jvadd(jvec a, jvec b) { jvec r = {J_add(a.x, b.x), J_add(a.y, b.y), J_add(a.z, b.z)}; return r; }
static jvec jvconst(const double p[3]) { jvec r = {J_const(p[0]), J_const(p[1]), J_const(p[2])}; return r; }
/* Eigen novec redux tree for 3 terms: e0 + (e1 + e2) */
static jet jnorm(jvec a) { return J_sqrt(J_add(J_mul(a.x, a.x), J_add(J_mul(a.y, a.y), J_mul(a.z, a.z)))); }
static jet jdot(jvec a, jvec b) { return J_add(J_mul(a.x, b.x), J_add(J_mul(a.y, b.y), J_mul(a.z, b.z))); }

/* q * v: uv = u x v; uv += uv; v + w*uv + u x uv */
static jvec jrotate(jquat q, jvec v)
{
    jvec u = {q.x, q.y, q.z};
    jvec uv = jcross(u, v);
    uv.x = J_add(uv.x, uv.x); uv.y = J_add(uv.y, uv.y); uv.z = J_add(uv.z, uv.z);
    jvec wuv = {J_mul(q.w, uv.x), J_mul(q.w, uv.y), J_mul(q.w, uv.z)};
    return jvadd(jvadd(v, wuv), jcross(u, uv));
}

static jvec transformed_point(const double q[4], const double t[3], const double cp[3], double s)
{
    /* Quaternion<T> q_last_curr{q[3], q[0], q[1], q[2]}  (w,x,y,z ctor; lidarFactor.hpp:24) */
    jquat ql = {J_var(q[0], 0), J_var(q[1], 1), J_var(q[2], 2), J_var(q[3], 3)};
    ql = slerp_from_identity(J_const(s), ql);
    jet S = J_const(s);
    jvec tl = {J_mul(S, J_var(t[0], 4)), J_mul(S, J_var(t[1], 5)), J_mul(S, J_var(t[2], 6))};
    return jvadd(jrotate(ql, jvconst(cp)), tl);
}

static void unpack(jet r, double *res, double *Jq, double *Jt)
{
    *res = r.a;
    for (int k = 0; k < 4; ++k) Jq[k] = r.v[k];
    for (int k = 0; k < 3; ++k) Jt[k] = r.v[4 + k];
}

void orc_edge_factor(const double q[4], const double t[3], const double cp[3], const double a[3],
                     const double b[3], double s, double r[3], double Jq[12], double Jt[9])
{
    jvec lp = transformed_point(q, t, cp, s);
    jvec lpa = jvconst(a), lpb = jvconst(b);
    jvec nu = jcross(jvsub(lp, lpa), jvsub(lp, lpb));             /* lidarFactor.hpp:32 */
    jvec de = jvsub(lpa, lpb);                                    /* :33 */
    jet n = jnorm(de);
    unpack(J_div(nu.x, n), &r[0], &Jq[0], &Jt[0]);                /* :35-37 */
    unpack(J_div(nu.y, n), &r[1], &Jq[4], &Jt[3]);
    unpack(J_div(nu.z, n), &r[2], &Jq[8], &Jt[6]);
}

void orc_plane_factor_modify(const double q[4], const double t[3], const double cp[3], const double j[3],
                     const double l[3], const double m[3], double s, double weight,
                     double r[1], double Jq[4], double Jt[3])
{
    /* ctor (:210-211): ljm_norm = (j - l).cross(j - m); ljm_norm.normalize();  (doubles) */
    const double a[3] = {j[0] - l[0], j[1] - l[1], j[2] - l[2]}, b[3] = {j[0] - m[0], j[1] - m[1], j[2] - m[2]};
    double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double z = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];   /* Vector3d squaredNorm, SSE2 linear-vectorised redux */
    if (z > 0.0) { const double nn = sqrt(z); n[0] /= nn; n[1] /= nn; n[2] /= nn; }
    jvec lp = transformed_point(q, t, cp, s);
    jet res = J_mul(jdot(jvsub(lp, jvconst(j)), jvconst(n)), J_const(weight));               /* :233 */
    unpack(res, r, Jq, Jt);
}

void orc_plane_norm_factor(const double q[4], const double t[3], const double cp[3], const double n[3],
                     double negative_OA_dot_norm, double r[1], double Jq[4], double Jt[3])
{
    jquat qw = {J_var(q[0], 0), J_var(q[1], 1), J_var(q[2], 2), J_var(q[3], 3)};          /* :263 */
    jvec tw = {J_var(t[0], 4), J_var(t[1], 5), J_var(t[2], 6)};
    jvec pw = jvadd(jrotate(qw, jvconst(cp)), tw);                                          /* :267 */
    jet res = J_add(jdot(jvconst(n), pw), J_const(negative_OA_dot_norm));                   /* :270 */
    unpack(res, r, Jq, Jt);
}

/* ceres::EigenQuaternionManifold::PlusJacobian, 4x3 row-major, rows in storage order x,y,z,w */
void orc_quat_plus_jacobian(const double x[4], double P[12])
{
    const double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
    P[0] =  qw; P[1]  =  qz; P[2]  = -qy;     /* row x */
    P[3] = -qz; P[4]  =  qw; P[5]  =  qx;     /* row y */
    P[6] =  qy; P[7]  = -qx; P[8]  =  qw;     /* row z */
    P[9] = -qx; P[10] = -qy; P[11] = -qz;     /* row w */
}

/* ceres QuaternionPlus: q+ = [sin(|d|) d/|d|, cos|d|] (x) q */
void orc_quat_plus(const double x[4], const double d[3], double out[4])
{
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double qd[4];
    if (n2 != 0.0) {
        const double n = sqrt(n2); const double sn = sin(n) / n;
        qd[3] = cos(n); qd[0] = sn * d[0]; qd[1] = sn * d[1]; qd[2] = sn * d[2];
    } else { qd[3] = 1.0; qd[0] = d[0]; qd[1] = d[1]; qd[2] = d[2]; }
    const double ax = qd[0], ay = qd[1], az = qd[2], aw = qd[3], bx = x[0], by = x[1], bz = x[2], bw = x[3];
    out[3] = aw * bw - ax * bx - ay * by - az * bz;
    out[0] = aw * bx + ax * bw + ay * bz - az * by;
    out[1] = aw * by - ax * bz + ay * bw + az * bx;
    out[2] = aw * bz + ax * by - ay * bx + az * bw;
}

/* ------------------------------------------------------------------------------------------ */
/* a10  residual blocks -> HuberLoss(0.1) corrector -> normal equations                         */
/* ------------------------------------------------------------------------------------------ */
static void accumulate_block(int rows, const double *r, const double *Jq, const double *Jt, const double P[12],
                             double huber_delta, double H[36], double g[6], double *cost)
{
    double Jl[3][6], rr[3];
    for (int i = 0; i < rows; ++i) {
        for (int c = 0; c < 3; ++c) {            /* local = ambient(rows x 4) * PlusJacobian(4 x 3) */
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += Jq[i * 4 + k] * P[k * 3 + c];
            Jl[i][c] = acc;
        }
        for (int c = 0; c < 3; ++c) Jl[i][3 + c] = Jt[i * 3 + c];
        rr[i] = r[i];
    }
    double sq = 0.0;
    for (int i = 0; i < rows; ++i) sq += rr[i] * rr[i];
    double rho0 = sq, rho1 = 1.0;
    if (huber_delta > 0.0) {
        /* ceres::HuberLoss::Evaluate, a = delta, b = a^2 */
        const double b = huber_delta * huber_delta;
        if (sq > b) {
            const double rnorm = sqrt(sq);
            rho0 = 2.0 * huber_delta * rnorm - b;
            rho1 = huber_delta / rnorm; if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
        }
        /* Corrector: rho'' <= 0 for Huber, so residual_scaling_ = sqrt(rho'), alpha_sq_norm_ = 0 */
        const double sr = sqrt(rho1);
        for (int i = 0; i < rows; ++i) { rr[i] *= sr; for (int c = 0; c < 6; ++c) Jl[i][c] *= sr; }
    }
    *cost += 0.5 * rho0;
    for (int i = 0; i < rows; ++i)
        for (int a = 0; a < 6; ++a) {
            g[a] += Jl[i][a] * rr[i];
            for (int c = 0; c < 6; ++c) H[a * 6 + c] += Jl[i][a] * Jl[i][c];
        }
}

void orc_normal_equations(const double q[4], const double t[3],
                          const orc_point *sharp, const int *e_src, const orc_point *corner_last,
                          const int *e_a, const int *e_b, int n_e,
                          const orc_point *flat, const int *p_src, const orc_point *surf_last,
                          const int *p_a, const int *p_b, const int *p_c, const float *p_w, int n_p,
                          double huber_delta, double H[36], double g[6], double *cost)
{
    double P[12];
    orc_quat_plus_jacobian(q, P);
    memset(H, 0, 36 * sizeof(double)); memset(g, 0, 6 * sizeof(double)); *cost = 0.0;
    for (int i = 0; i < n_e; ++i) {
        const orc_point *c = &sharp[e_src[i]], *a = &corner_last[e_a[i]], *b = &corner_last[e_b[i]];
        const double cp[3] = {c->x, c->y, c->z}, pa[3] = {a->x, a->y, a->z}, pb[3] = {b->x, b->y, b->z};
        double r[3], Jq[12], Jt[9];
        orc_edge_factor(q, t, cp, pa, pb, orc_point_s(c), r, Jq, Jt);          /* s: :568-571 */
        accumulate_block(3, r, Jq, Jt, P, huber_delta, H, g, cost);
    }
    for (int i = 0; i < n_p; ++i) {
        const orc_point *c = &flat[p_src[i]], *a = &surf_last[p_a[i]], *b = &surf_last[p_b[i]], *d = &surf_last[p_c[i]];
        const double cp[3] = {c->x, c->y, c->z}, pa[3] = {a->x, a->y, a->z}, pb[3] = {b->x, b->y, b->z}, pc[3] = {d->x, d->y, d->z};
        double r[1], Jq[4], Jt[3];
        orc_plane_factor_modify(q, t, cp, pa, pb, pc, orc_point_s(c), p_w ? (double)p_w[i] : 1.0, r, Jq, Jt);   /* s: :738-741 */
        accumulate_block(1, r, Jq, Jt, P, huber_delta, H, g, cost);
    }
}

int orc_gn_solve(const double H[36], const double g[6], double delta[6])
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = H[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0.0)) return -1; L[i * 6 + i] = sqrt(s); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = -g[i]; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * delta[k]; delta[i] = s / L[i * 6 + i]; }
    return 0;
}

void orc_pose_update(double q[4], double t[3], const double delta[6])
{
    double qn[4];
    orc_quat_plus(q, delta, qn);
    memcpy(q, qn, sizeof(qn));
    t[0] += delta[3]; t[1] += delta[4]; t[2] += delta[5];
}

/* ------------------------------------------------------------------------------------------ */
/* f1  ceres::Solve as the reference configures it (laserOdometry.cpp:820-825), restated        */
/* ------------------------------------------------------------------------------------------ */
void orc_lm_default(orc_lm_options *o)
{
    o->max_num_iterations = 4;            /* :822 */
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->jacobi_scaling = 1;
}

/* (S H S + D^2) y = -(S g) by Cholesky; DENSE_QR solves the same augmented least-squares problem */
static int lm_step(const double H[36], const double g[6], const double scale[6], double radius,
                   double min_diag, double max_diag, double step_scaled[6])
{
    double A[36], b[6];
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) A[i * 6 + j] = H[i * 6 + j] * scale[i] * scale[j];
        b[i] = g[i] * scale[i];
    }
    for (int i = 0; i < 6; ++i) {
        double d = A[i * 6 + i];                         /* diag(J^T J) of the SCALED Jacobian */
        if (d < min_diag) d = min_diag;
        if (d > max_diag) d = max_diag;
        A[i * 6 + i] += d / radius;                      /* lm_diagonal^2 = clamp(diag) / radius */
    }
    return orc_gn_solve(A, b, step_scaled);
}

typedef void (*lm_eval_fn)(void *ctx, const double q[4], const double t[3], double H[36], double g[6], double *cost);

static void lm_core(double q[4], double t[3], lm_eval_fn eval, void *ctx, const orc_lm_options *opt, double summary[4])
{
    double H[36], g[6], cost;
    eval(ctx, q, t, H, g, &cost);
    summary[0] = cost; summary[1] = cost; summary[2] = 0; summary[3] = 0;
    double scale[6];
    for (int i = 0; i < 6; ++i) scale[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[i * 6 + i])) : 1.0;
    double radius = opt->initial_radius, decrease_factor = 2.0;
    int iter = 0;
    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (iter >= opt->max_num_iterations) break;
        {   /* gradient tolerance: max-norm of Plus(x, -gradient) - x in the ambient space */
            double q2[4], t2[3], ng[6];
            for (int i = 0; i < 6; ++i) ng[i] = -g[i];
            memcpy(q2, q, sizeof(q2)); memcpy(t2, t, sizeof(t2));
            orc_pose_update(q2, t2, ng);
            double m = 0.0;
            for (int i = 0; i < 4; ++i) m = fmax(m, fabs(q2[i] - q[i]));
            for (int i = 0; i < 3; ++i) m = fmax(m, fabs(t2[i] - t[i]));
            if (m <= opt->gradient_tolerance) break;
        }
        if (radius < opt->min_radius) break;
        iter++;
        double ys[6], delta[6];
        int ok = lm_step(H, g, scale, radius, opt->min_lm_diagonal, opt->max_lm_diagonal, ys) == 0;
        double model_cost_change = 0.0;
        if (ok) {
            /* -(g_s . y + y^T H_s y / 2) in the scaled space == -(g . d + d^T H d / 2) with d = S y */
            for (int i = 0; i < 6; ++i) delta[i] = ys[i] * scale[i];
            double gd = 0.0, dHd = 0.0;
            for (int i = 0; i < 6; ++i) { gd += g[i] * delta[i]; for (int j = 0; j < 6; ++j) dHd += delta[i] * H[i * 6 + j] * delta[j]; }
            model_cost_change = -(gd + 0.5 * dHd);
        }
        if (!ok || !(model_cost_change > 0.0)) { radius *= 0.5; continue; }     /* invalid step: LM StepIsInvalid */
        double qc[4], tc[3], Hc[36], gc[6], cc;
        memcpy(qc, q, sizeof(qc)); memcpy(tc, t, sizeof(tc));
        orc_pose_update(qc, tc, delta);
        eval(ctx, qc, tc, Hc, gc, &cc);
        /* ParameterToleranceReached / FunctionToleranceReached: both return BEFORE the candidate is taken */
        double step_norm = 0.0, x_norm = 0.0;
        for (int i = 0; i < 4; ++i) { step_norm += (qc[i] - q[i]) * (qc[i] - q[i]); x_norm += q[i] * q[i]; }
        for (int i = 0; i < 3; ++i) { step_norm += (tc[i] - t[i]) * (tc[i] - t[i]); x_norm += t[i] * t[i]; }
        step_norm = sqrt(step_norm); x_norm = sqrt(x_norm);
        if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) break;
        if (fabs(cost - cc) <= opt->function_tolerance * cost) break;
        const double relative_decrease = (cost - cc) / model_cost_change;
        if (relative_decrease > opt->min_relative_decrease) {               /* HandleSuccessfulStep + StepAccepted */
            memcpy(q, qc, sizeof(qc)); memcpy(t, tc, sizeof(tc));
            memcpy(H, Hc, sizeof(Hc)); memcpy(g, gc, sizeof(gc)); cost = cc;
            const double f = 1.0 - pow(2.0 * relative_decrease - 1.0, 3.0);
            radius = radius / fmax(1.0 / 3.0, f);
            if (radius > opt->max_radius) radius = opt->max_radius;
            decrease_factor = 2.0;
            summary[3] += 1;
        } else {                                                             /* StepRejected */
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
        }
    }
    summary[1] = cost; summary[2] = iter;
}

typedef struct {
    const orc_point *sharp; const int *e_src; const orc_point *corner_last; const int *e_a, *e_b; int n_e;
    const orc_point *flat; const int *p_src; const orc_point *surf_last; const int *p_a, *p_b, *p_c; const float *p_w; int n_p;
    double huber_delta;
} odo_eval_ctx;

static void odo_eval(void *vc, const double q[4], const double t[3], double H[36], double g[6], double *cost)
{
    const odo_eval_ctx *c = (const odo_eval_ctx *)vc;
    orc_normal_equations(q, t, c->sharp, c->e_src, c->corner_last, c->e_a, c->e_b, c->n_e, c->flat, c->p_src, c->surf_last,
                         c->p_a, c->p_b, c->p_c, c->p_w, c->n_p, c->huber_delta, H, g, cost);
}

void orc_lm_solve(double q[4], double t[3],
                  const orc_point *sharp, const int *e_src, const orc_point *corner_last, const int *e_a, const int *e_b, int n_e,
                  const orc_point *flat, const int *p_src, const orc_point *surf_last,
                  const int *p_a, const int *p_b, const int *p_c, const float *p_w, int n_p,
                  double huber_delta, const orc_lm_options *opt, double summary[4])
{
    odo_eval_ctx c = {sharp, e_src, corner_last, e_a, e_b, n_e, flat, p_src, surf_last, p_a, p_b, p_c, p_w, n_p, huber_delta};
    lm_core(q, t, odo_eval, &c, opt, summary);
}

void orc_odometry_frame(double q[4], double t[3], const orc_point *sharp, int ns, const orc_point *flat, int nf,
                        const orc_point *corner_last, int mc, const orc_point *surf_last, int ms,
                        int vote, int n_outer, double huber_delta, const orc_lm_options *opt)
{
    int *es = (int *)malloc(sizeof(int) * (size_t)(ns + 1) * 3), *ps = (int *)malloc(sizeof(int) * (size_t)(nf + 1) * 4);
    int *ea = es + ns + 1, *eb = ea + ns + 1, *pa = ps + nf + 1, *pb = pa + nf + 1, *pc = pb + nf + 1;
    int *cnt = (int *)malloc(sizeof(int) * (size_t)(nf + 1) * 2), *sidx = cnt + nf + 1;
    float *sw = (float *)malloc(sizeof(float) * (size_t)(nf + 1) * 2), *w = sw + nf + 1;
    orc_point *src = (orc_point *)malloc(sizeof(orc_point) * (size_t)(nf + 1) * 2), *tgt = src + nf + 1;
    int *qs = (int *)malloc(sizeof(int) * (size_t)(nf + 1) * 4), *qa = qs + nf + 1, *qb = qa + nf + 1, *qc = qb + nf + 1;
    for (int outer = 0; outer < n_outer; ++outer) {                          /* :439 */
        int ne = 0, np = 0;
        orc_associate_corner(q, t, sharp, ns, corner_last, mc, es, ea, eb, &ne);
        orc_associate_plane(q, t, flat, nf, surf_last, ms, ps, pa, pb, pc, &np);
        int nsel = np;
        if (vote) {                                                          /* :794-810 */
            for (int i = 0; i < np; ++i) { src[i] = flat[ps[i]]; tgt[i] = surf_last[pa[i]]; }
            orc_vote(src, tgt, np, 0, cnt, sidx, sw, &nsel);
            /* residual blocks in correspondence order (the order only permutes the blocks) */
            char *keep = (char *)calloc((size_t)np + 1, 1);
            for (int i = 0; i < np; ++i) w[i] = 1.0f;
            for (int i = 0; i < nsel; ++i) { keep[sidx[i]] = 1; w[sidx[i]] = sw[i]; }
            int m = 0;
            for (int i = 0; i < np; ++i) if (keep[i]) { qs[m] = ps[i]; qa[m] = pa[i]; qb[m] = pb[i]; qc[m] = pc[i]; w[m] = w[i]; m++; }
            free(keep);
            nsel = m;
        } else {
            for (int i = 0; i < np; ++i) { qs[i] = ps[i]; qa[i] = pa[i]; qb[i] = pb[i]; qc[i] = pc[i]; w[i] = 1.0f; }   /* :781-787 */
        }
        double summary[4];
        orc_lm_solve(q, t, sharp, es, corner_last, ea, eb, ne, flat, qs, surf_last, qa, qb, qc, w, nsel, huber_delta, opt, summary);
    }
    free(qs); free(src); free(sw); free(cnt); free(ps); free(es);
}

/* ------------------------------------------------------------------------------------------ */
/* f2  laserMapping scan-to-submap (laserMapping.cpp:1826-2095)                                 */
/* ------------------------------------------------------------------------------------------ */
/* pointAssociateToMap (:125-134): Eigen q_w_curr * p + t_w_curr in f64, stored to the float PointType */
void orc_point_associate_to_map(const double q[4], const double t[3], const orc_point *pi, orc_point *po)
{
    const double v[3] = {pi->x, pi->y, pi->z};
    double r[3];
    quat_rotate(q, v, r);
    po->x = (float)(r[0] + t[0]); po->y = (float)(r[1] + t[1]); po->z = (float)(r[2] + t[2]);
    po->intensity = pi->intensity;
}

/* kdtree->nearestKSearch(pointSel, 5, ...) (:1882, :1948): the five smallest (FLANN L2_Simple f32 distance, index)
 * pairs in ascending order.  The callers use the result only when the fifth distance is < 1.0, so a grid with
 * cells >= 1 m searched one cell around the query gives the same answer in every case that matters; otherwise
 * found < 5 or d[4] >= 1 and the point is rejected either way. */
static int knn5(const orc_point *q, const orc_point *cloud, int m, const nn_grid *g, int idx[5], float d[5])
{
    int n = 0;
#define KNN_TAKE(j) do { const float dj = flann_l2(q, &cloud[j]); int pos = n < 5 ? n : 5;                                    \
        while (pos > 0 && (dj < d[pos - 1] || (dj == d[pos - 1] && (j) < idx[pos - 1]))) --pos;                                    \
        if (pos < 5) { for (int k = (n < 5 ? n : 4); k > pos; --k) { d[k] = d[k - 1]; idx[k] = idx[k - 1]; }                      \
                       d[pos] = dj; idx[pos] = (j); if (n < 5) ++n; } } while (0)
    if (!g) { for (int j = 0; j < m; ++j) KNN_TAKE(j); return n; }
    int c0[3]; grid_cell_of(g, q, c0);
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        const int cx = c0[0] + dx, cy = c0[1] + dy, cz = c0[2] + dz;
        if (cx < 0 || cy < 0 || cz < 0 || cx >= g->dim[0] || cy >= g->dim[1] || cz >= g->dim[2]) continue;
        const size_t c = ((size_t)cz * g->dim[1] + cy) * g->dim[0] + cx;
        for (int k = g->start[c]; k < g->start[c + 1]; ++k) KNN_TAKE(g->idx[k]);
    }
#undef KNN_TAKE
    return n;
}

/* Eigen::SelfAdjointEigenSolver<Matrix3d> (:1905) restated as cyclic Jacobi in f64 (Eigen tridiagonalises and runs
 * implicit QL; the results agree to rounding, the decomposition is only used through eigenvalue ratios and the
 * direction of the largest eigenvector, whose sign cancels in the edge factor).  Eigenvalues ascending. */
void orc_sym_eig3(const double A_in[9], double w[3], double V[9])
{
    double A[3][3], Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = A_in[i * 3 + j];
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (A[p][q] == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
            for (int k = 0; k < 3; ++k) { const double qkp = Q[k][p], qkq = Q[k][q]; Q[k][p] = c * qkp - sn * qkq; Q[k][q] = sn * qkp + c * qkq; }
        }
    }
    int o[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j) if (A[o[j]][o[j]] > A[o[j + 1]][o[j + 1]]) { const int tmp = o[j]; o[j] = o[j + 1]; o[j + 1] = tmp; }
    for (int k = 0; k < 3; ++k) { w[k] = A[o[k]][o[k]]; for (int i = 0; i < 3; ++i) V[i * 3 + k] = Q[i][o[k]]; }
}

/* matA0.colPivHouseholderQr().solve(matB0) (:1972) for the 5 x 3 system A x = b: Householder QR with column pivoting
 * on the largest remaining column norm, least-squares solution (Eigen 3.3 ColPivHouseholderQR, restated). */
void orc_qr_solve_5x3(const double A_in[15], const double b_in[5], double x[3])
{
    double A[5][3], b[5]; int perm[3] = {0, 1, 2};
    for (int i = 0; i < 5; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = A_in[i * 3 + j]; b[i] = b_in[i]; }
    double maxn2 = 0.0;
    for (int j = 0; j < 3; ++j) { double n2 = 0.0; for (int i = 0; i < 5; ++i) n2 += A[i][j] * A[i][j]; if (n2 > maxn2) maxn2 = n2; }
    const double thr = maxn2 * (2.220446049250313e-16 * 2.220446049250313e-16) / 5.0;     /* threshold_helper */
    int rank = 3;
    for (int k = 0; k < 3; ++k) {
        int piv = k; double best = -1.0;
        for (int j = k; j < 3; ++j) { double n2 = 0.0; for (int i = k; i < 5; ++i) n2 += A[i][j] * A[i][j]; if (n2 > best) { best = n2; piv = j; } }
        if (best < thr) { rank = k; break; }
        if (piv != k) { for (int i = 0; i < 5; ++i) { const double tmp = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = tmp; } const int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp; }
        /* makeHouseholderInPlace */
        double tail = 0.0; for (int i = k + 1; i < 5; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double tau, beta, v[5];
        if (tail <= 2.2250738585072014e-308) { tau = 0.0; beta = c0; for (int i = k + 1; i < 5; ++i) v[i] = 0.0; }
        else {
            beta = sqrt(c0 * c0 + tail); if (c0 >= 0.0) beta = -beta;
            for (int i = k + 1; i < 5; ++i) v[i] = A[i][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        v[k] = 1.0;
        for (int j = k + 1; j < 3; ++j) { double d = 0.0; for (int i = k; i < 5; ++i) d += v[i] * A[i][j]; d *= tau; for (int i = k; i < 5; ++i) A[i][j] -= d * v[i]; }
        { double d = 0.0; for (int i = k; i < 5; ++i) d += v[i] * b[i]; d *= tau; for (int i = k; i < 5; ++i) b[i] -= d * v[i]; }
        A[k][k] = beta; for (int i = k + 1; i < 5; ++i) A[i][k] = 0.0;
    }
    double z[3] = {0, 0, 0};
    for (int i = rank - 1; i >= 0; --i) { double sacc = b[i]; for (int j = i + 1; j < rank; ++j) sacc -= A[i][j] * z[j]; z[i] = sacc / A[i][i]; }
    for (int k = 0; k < 3; ++k) x[perm[k]] = z[k];
}

/* one data-association pass (:1877-2047).  Edge blocks: (stack index, a, b); plane blocks: (stack index, unit normal, d) */
int orc_map_associate(const double q[4], const double t[3],
                      const orc_point *corner_stack, int n_cs, const orc_point *corner_map, int n_cm,
                      const orc_point *surf_stack, int n_ss, const orc_point *surf_map, int n_sm,
                      int *e_src, double *e_a, double *e_b, int *n_e, int *p_src, double *p_n, double *p_d, int *n_p)
{
    /* cells a little over the 1 m acceptance radius: the f32 rounding of the cell coordinate cannot push a neighbour
     * that is closer than 1 m two cells away */
    nn_grid *gc = (g_nn_mode && n_cm > 0) ? grid_build(corner_map, n_cm, 1.01f) : NULL;
    nn_grid *gs = (g_nn_mode && n_sm > 0) ? grid_build(surf_map, n_sm, 1.01f) : NULL;
    int ne = 0, np = 0;
    for (int i = 0; i < n_cs; ++i) {
        orc_point sel; int idx[5]; float d[5];
        orc_point_associate_to_map(q, t, &corner_stack[i], &sel);
        if (knn5(&sel, corner_map, n_cm, gc, idx, d) < 5 || !(d[4] < 1.0)) continue;                 /* :1884 */
        double c[3] = {0, 0, 0}, pts[5][3];
        for (int j = 0; j < 5; ++j) {                                                                  /* :1888-1895 */
            pts[j][0] = corner_map[idx[j]].x; pts[j][1] = corner_map[idx[j]].y; pts[j][2] = corner_map[idx[j]].z;
            for (int k = 0; k < 3; ++k) c[k] = c[k] + pts[j][k];
        }
        for (int k = 0; k < 3; ++k) c[k] = c[k] / 5.0;
        double cov[9] = {0};
        for (int j = 0; j < 5; ++j) {                                                                  /* :1898-1903 */
            const double z[3] = {pts[j][0] - c[0], pts[j][1] - c[1], pts[j][2] - c[2]};
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a * 3 + b] = cov[a * 3 + b] + z[a] * z[b];
        }
        double w[3], V[9];
        orc_sym_eig3(cov, w, V);
        if (!(w[2] > 3 * w[1])) continue;                                                              /* :1911 */
        e_src[ne] = i;
        for (int k = 0; k < 3; ++k) { e_a[ne * 3 + k] = 0.1 * V[k * 3 + 2] + c[k]; e_b[ne * 3 + k] = -0.1 * V[k * 3 + 2] + c[k]; }   /* :1915-1916 */
        ++ne;
    }
    for (int i = 0; i < n_ss; ++i) {
        orc_point sel; int idx[5]; float d[5];
        orc_point_associate_to_map(q, t, &surf_stack[i], &sel);
        if (knn5(&sel, surf_map, n_sm, gs, idx, d) < 5 || !(d[4] < 1.0)) continue;                   /* :1952 */
        double A[15], b[5] = {-1, -1, -1, -1, -1}, nrm[3];
        for (int j = 0; j < 5; ++j) { A[j * 3] = surf_map[idx[j]].x; A[j * 3 + 1] = surf_map[idx[j]].y; A[j * 3 + 2] = surf_map[idx[j]].z; }
        orc_qr_solve_5x3(A, b, nrm);                                                                    /* :1972 */
        const double len = sqrt((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
        const double negative_OA_dot_norm = 1 / len;                                                   /* :1973 */
        if (len * len > 0.0) { nrm[0] /= len; nrm[1] /= len; nrm[2] /= len; }                        /* :1974 normalize() */
        int valid = 1;
        for (int j = 0; j < 5; ++j)                                                                    /* :1980-1990 */
            if (fabs(nrm[0] * A[j * 3] + nrm[1] * A[j * 3 + 1] + nrm[2] * A[j * 3 + 2] + negative_OA_dot_norm) > 0.2) { valid = 0; break; }
        if (!valid) continue;
        p_src[np] = i; p_n[np * 3] = nrm[0]; p_n[np * 3 + 1] = nrm[1]; p_n[np * 3 + 2] = nrm[2]; p_d[np] = negative_OA_dot_norm;
        ++np;
    }
    grid_free(gc); grid_free(gs);
    *n_e = ne; *n_p = np;
    return 0;
}

/* the residual blocks of one pass (:1918-1919, :2033-2034) under HuberLoss(0.1) + EigenQuaternionManifold */
void orc_map_normal_equations(const double q[4], const double t[3],
                              const orc_point *corner_stack, const int *e_src, const double *e_a, const double *e_b, int n_e,
                              const orc_point *surf_stack, const int *p_src, const double *p_n, const double *p_d, int n_p,
                              double huber_delta, double H[36], double g[6], double *cost)
{
    double P[12];
    orc_quat_plus_jacobian(q, P);
    memset(H, 0, 36 * sizeof(double)); memset(g, 0, 6 * sizeof(double)); *cost = 0.0;
    for (int i = 0; i < n_e; ++i) {
        const orc_point *c = &corner_stack[e_src[i]];
        const double cp[3] = {c->x, c->y, c->z};
        double r[3], Jq[12], Jt[9];
        orc_edge_factor(q, t, cp, &e_a[i * 3], &e_b[i * 3], 1.0, r, Jq, Jt);
        accumulate_block(3, r, Jq, Jt, P, huber_delta, H, g, cost);
    }
    for (int i = 0; i < n_p; ++i) {
        const orc_point *c = &surf_stack[p_src[i]];
        const double cp[3] = {c->x, c->y, c->z};
        double r[1], Jq[4], Jt[3];
        orc_plane_norm_factor(q, t, cp, &p_n[i * 3], p_d[i], r, Jq, Jt);
        accumulate_block(1, r, Jq, Jt, P, huber_delta, H, g, cost);
    }
}

typedef struct {
    const orc_point *corner_stack; const int *e_src; const double *e_a, *e_b; int n_e;
    const orc_point *surf_stack; const int *p_src; const double *p_n, *p_d; int n_p; double huber_delta;
} map_eval_ctx;

static void map_eval(void *vc, const double q[4], const double t[3], double H[36], double g[6], double *cost)
{
    const map_eval_ctx *c = (const map_eval_ctx *)vc;
    orc_map_normal_equations(q, t, c->corner_stack, c->e_src, c->e_a, c->e_b, c->n_e, c->surf_stack, c->p_src, c->p_n, c->p_d, c->n_p,
                             c->huber_delta, H, g, cost);
}

/* the optimisation of one mapping frame (:1822-2095): if the map holds > 10 corner and > 50 surf points, n_outer (2)
 * times { data association at the current pose, ceres::Solve (<= 4 LM iterations) }.  Returns 1 when it ran. */
int orc_map_optimize(double q[4], double t[3],
                     const orc_point *corner_stack, int n_cs, const orc_point *corner_map, int n_cm,
                     const orc_point *surf_stack, int n_ss, const orc_point *surf_map, int n_sm,
                     int n_outer, double huber_delta, const orc_lm_options *opt)
{
    if (!(n_cm > 10 && n_sm > 50)) return 0;                                                           /* :1822 */
    int *e_src = (int *)malloc(sizeof(int) * (size_t)(n_cs + 1)), *p_src = (int *)malloc(sizeof(int) * (size_t)(n_ss + 1));
    double *e_a = (double *)malloc(sizeof(double) * 6 * (size_t)(n_cs + 1)), *e_b = e_a + 3 * (size_t)(n_cs + 1);
    double *p_n = (double *)malloc(sizeof(double) * 4 * (size_t)(n_ss + 1)), *p_d = p_n + 3 * (size_t)(n_ss + 1);
    for (int it = 0; it < n_outer; ++it) {                                                             /* :1832 */
        int ne = 0, np = 0;
        orc_map_associate(q, t, corner_stack, n_cs, corner_map, n_cm, surf_stack, n_ss, surf_map, n_sm, e_src, e_a, e_b, &ne, p_src, p_n, p_d, &np);
        map_eval_ctx c = {corner_stack, e_src, e_a, e_b, ne, surf_stack, p_src, p_n, p_d, np, huber_delta};
        double summary[4];
        lm_core(q, t, map_eval, &c, opt, summary);
    }
    free(p_n); free(e_a); free(p_src); free(e_src);
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* f2 (second stage)  the cube map around the optimisation: laserMapping.cpp:1584-1821, :2101-2165 */
/* ------------------------------------------------------------------------------------------ */
#define CM_W 21
#define CM_H 21
#define CM_D 11
#define CM_N (CM_W * CM_H * CM_D)        /* 4851, :53 */
typedef struct { orc_point *p; int n, cap; } cm_cloud;
struct orc_cubemap {
    cm_cloud corner[CM_N], surf[CM_N];   /* laserCloudCornerArray / laserCloudSurfArray (:74-75) */
    int cen[3];                          /* laserCloudCenWidth / Height / Depth (:45-47) */
    float line_res, plane_res;           /* :2363-2364 */
    int valid[125], n_valid;             /* laserCloudValidInd (:56) */
    cm_cloud from_map[2], stack[2];      /* laserCloudCornerFromMap / SurfFromMap, laserCloudCornerStack / SurfStack */
};

static void cm_push(cm_cloud *c, const orc_point *p, int n)
{
    if (c->n + n > c->cap) { c->cap = (c->n + n) * 2 + 16; c->p = (orc_point *)realloc(c->p, (size_t)c->cap * sizeof(orc_point)); }
    if (n > 0) memcpy(c->p + c->n, p, (size_t)n * sizeof(orc_point));
    c->n += n;
}

orc_cubemap *orc_cubemap_create(float line_res, float plane_res)
{
    orc_cubemap *m = (orc_cubemap *)calloc(1, sizeof(orc_cubemap));
    m->cen[0] = 10; m->cen[1] = 10; m->cen[2] = 5;
    m->line_res = line_res; m->plane_res = plane_res;
    return m;
}

void orc_cubemap_destroy(orc_cubemap *m)
{
    if (!m) return;
    for (int i = 0; i < CM_N; ++i) { free(m->corner[i].p); free(m->surf[i].p); }
    for (int w = 0; w < 2; ++w) { free(m->from_map[w].p); free(m->stack[w].p); }
    free(m);
}

/* one step of a shift loop (:1598-1778): along `axis`, every line of cubes moves by one towards +end (dir = +1: the
 * cube at the far end wraps to index 0 and is cleared) or towards index 0 (dir = -1) */
static void cm_shift(orc_cubemap *m, int axis, int dir)
{
    const int dim[3] = {CM_W, CM_H, CM_D}, stride[3] = {1, CM_W, CM_W * CM_H};
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    for (int u = 0; u < dim[a1]; ++u) for (int v = 0; v < dim[a2]; ++v) {
        const int base = u * stride[a1] + v * stride[a2];
        if (dir > 0) {
            cm_cloud c = m->corner[base + (dim[axis] - 1) * stride[axis]], sf = m->surf[base + (dim[axis] - 1) * stride[axis]];
            for (int i = dim[axis] - 1; i >= 1; --i) {
                m->corner[base + i * stride[axis]] = m->corner[base + (i - 1) * stride[axis]];
                m->surf[base + i * stride[axis]] = m->surf[base + (i - 1) * stride[axis]];
            }
            c.n = 0; sf.n = 0;
            m->corner[base] = c; m->surf[base] = sf;
        } else {
            cm_cloud c = m->corner[base], sf = m->surf[base];
            for (int i = 0; i < dim[axis] - 1; ++i) {
                m->corner[base + i * stride[axis]] = m->corner[base + (i + 1) * stride[axis]];
                m->surf[base + i * stride[axis]] = m->surf[base + (i + 1) * stride[axis]];
            }
            c.n = 0; sf.n = 0;
            m->corner[base + (dim[axis] - 1) * stride[axis]] = c; m->surf[base + (dim[axis] - 1) * stride[axis]] = sf;
        }
    }
}

/* :1584-1821: centre cube of t_w_curr, shifts, the 5 x 5 x 3 valid cubes, the clouds gathered from them, the scan's
 * feature clouds down-sized */
void orc_cubemap_prepare(orc_cubemap *m, const double t_w[3], const orc_point *corner_last, int n_corner, const orc_point *surf_last, int n_surf)
{
    const int dim[3] = {CM_W, CM_H, CM_D};
    int cc[3];
    for (int k = 0; k < 3; ++k) {
        cc[k] = (int)((t_w[k] + 25.0) / 50.0) + m->cen[k];                                  /* :1584-1586 */
        if (t_w[k] + 25.0 < 0) cc[k]--;                                                     /* :1588-1593 */
    }
    for (int k = 0; k < 3; ++k) {
        while (cc[k] < 3) { cm_shift(m, k, +1); cc[k]++; m->cen[k]++; }                     /* :1595-1625 and the J, K twins */
        while (cc[k] >= dim[k] - 3) { cm_shift(m, k, -1); cc[k]--; m->cen[k]--; }           /* :1627-1657 */
    }
    m->n_valid = 0;
    for (int i = cc[0] - 2; i <= cc[0] + 2; i++) for (int j = cc[1] - 2; j <= cc[1] + 2; j++) for (int k = cc[2] - 1; k <= cc[2] + 1; k++)   /* :1783-1801 */
        if (i >= 0 && i < CM_W && j >= 0 && j < CM_H && k >= 0 && k < CM_D) m->valid[m->n_valid++] = i + CM_W * j + CM_W * CM_H * k;
    m->from_map[0].n = 0; m->from_map[1].n = 0;
    for (int i = 0; i < m->n_valid; ++i) {                                                  /* :1803-1808 */
        cm_push(&m->from_map[0], m->corner[m->valid[i]].p, m->corner[m->valid[i]].n);
        cm_push(&m->from_map[1], m->surf[m->valid[i]].p, m->surf[m->valid[i]].n);
    }
    const orc_point *in[2] = {corner_last, surf_last}; const int n_in[2] = {n_corner, n_surf};
    const float leaf[2] = {m->line_res, m->plane_res};
    for (int w = 0; w < 2; ++w) {                                                           /* :1813-1821 */
        m->stack[w].n = 0;
        cm_push(&m->stack[w], NULL, 0);
        if (m->stack[w].cap < n_in[w] + 1) { m->stack[w].cap = n_in[w] + 16; m->stack[w].p = (orc_point *)realloc(m->stack[w].p, (size_t)m->stack[w].cap * sizeof(orc_point)); }
        int n_out = 0;
        orc_voxel_grid(in[w], n_in[w], leaf[w], m->stack[w].p, &n_out);
        m->stack[w].n = n_out;
    }
}

int orc_cubemap_optimize(orc_cubemap *m, double q[4], double t[3], int n_outer, double huber_delta, const orc_lm_options *opt)
{
    return orc_map_optimize(q, t, m->stack[0].p, m->stack[0].n, m->from_map[0].p, m->from_map[0].n,
                            m->stack[1].p, m->stack[1].n, m->from_map[1].p, m->from_map[1].n, n_outer, huber_delta, opt);
}

/* :2103-2165: the registered scan's points into their cubes, then every valid cube down-sized */
void orc_cubemap_update(orc_cubemap *m, const double q[4], const double t[3])
{
    for (int w = 0; w < 2; ++w) {
        cm_cloud *arr = w ? m->surf : m->corner;
        for (int i = 0; i < m->stack[w].n; ++i) {
            orc_point sel;
            orc_point_associate_to_map(q, t, &m->stack[w].p[i], &sel);
            const float v[3] = {sel.x, sel.y, sel.z};
            int c[3], ok = 1;
            const int dim[3] = {CM_W, CM_H, CM_D};
            for (int k = 0; k < 3; ++k) {
                c[k] = (int)((v[k] + 25.0) / 50.0) + m->cen[k];                             /* :2108-2110 */
                if (v[k] + 25.0 < 0) c[k]--;                                                /* :2112-2117 */
                if (c[k] < 0 || c[k] >= dim[k]) ok = 0;
            }
            if (ok) cm_push(&arr[c[0] + CM_W * c[1] + CM_W * CM_H * c[2]], &sel, 1);        /* :2119-2125 */
        }
        const float leaf = w ? m->plane_res : m->line_res;
        for (int i = 0; i < m->n_valid; ++i) {                                              /* :2151-2165 */
            cm_cloud *c = &arr[m->valid[i]];
            if (c->n == 0) continue;
            orc_point *tmp = (orc_point *)malloc((size_t)c->n * sizeof(orc_point));
            int n_out = 0;
            orc_voxel_grid(c->p, c->n, leaf, tmp, &n_out);
            memcpy(c->p, tmp, (size_t)n_out * sizeof(orc_point)); c->n = n_out;
            free(tmp);
        }
    }
}

void orc_cubemap_get(const orc_cubemap *m, int which, const orc_point **p, int *n)
{
    const cm_cloud *c = which < 2 ? &m->from_map[which] : &m->stack[which - 2];
    *p = c->p; *n = c->n;
}
void orc_cubemap_cube(const orc_cubemap *m, int surf, int cube, const orc_point **p, int *n)
{
    const cm_cloud *c = surf ? &m->surf[cube] : &m->corner[cube];
    *p = c->p; *n = c->n;
}
void orc_cubemap_center(const orc_cubemap *m, int cen[3]) { cen[0] = m->cen[0]; cen[1] = m->cen[1]; cen[2] = m->cen[2]; }
