/*
 * ll_synth.c -- deterministic synthetic LiDAR scans (host utility; input generator only).
 *
 * Analytic ray-cast of a procedural street scene: ground plane at z = -1.73 m
 * below the sensor, axis-aligned boxes (buildings / cars) and vertical cylinders (poles) laid out on a
 * hashed 15 m lattice, sensor driving a circular arc (v = 10 m/s, yaw rate 0.1 rad/s).  Ring elevations
 * sit on the bin centres of the reference's ring model (scanRegistration.cpp:144, :162), so the
 * ring assignment of a synthetic point never sits on a bin edge.  Output is KITTI .bin shaped:
 * float4 (x, y, z, reflectance) per return, in the SENSOR frame.
 *
 * Every random number is a counter-based hash of (seed, scan index, ray index): results do not depend
 * on thread count or evaluation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    int    rings;          /* 16, 32, 64, 128 ... */
    int    azimuths;       /* columns per revolution */
    double elev_lo_deg;    /* elevation of ring 0 */
    double elev_hi_deg;    /* elevation of the last ring */
    double range_sigma;    /* gaussian range noise, metres */
    double max_range;      /* returns beyond are dropped */
    double az_jitter_deg;  /* per-ring fixed azimuth offset amplitude (real sensors fire lasers at offsets) */
    int    order;          /* 0 = ring-major (KITTI .bin style), 1 = azimuth-major (firing order) */
    uint32_t seed;
    double speed;          /* m/s */
    double yaw_rate;       /* rad/s */
    double period;         /* s per scan */
    double drop_prob;      /* probability a return is missing (emitted as NaN when emit_nan, else skipped) */
    int    emit_nan;
} ll_synth_cfg;

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static uint32_t hash3(uint32_t a, uint32_t b, uint32_t c)
{
    return hash32(a ^ hash32(b + 0x9e3779b9U + hash32(c + 0x85ebca6bU)));
}
static double u01(uint32_t h) { return ((double)h + 0.5) / 4294967296.0; }

typedef struct { int type; double cx, cy, hx, hy, z0, z1, r; } obj_t;   /* type 0 box, 1 cylinder */

#define CELL 15.0

/* object of lattice cell (ix, iy); returns 0 when the cell is empty or inside the road corridor */
static int cell_object(uint32_t seed, int ix, int iy, double arc_R, obj_t *o)
{
    const uint32_t h = hash3(seed, (uint32_t)(ix * 73856093), (uint32_t)(iy * 19349663));
    const double kind = u01(hash32(h + 1));
    const double cx = (ix + 0.2 + 0.6 * u01(hash32(h + 2))) * CELL;
    const double cy = (iy + 0.2 + 0.6 * u01(hash32(h + 3))) * CELL;
    /* road corridor: keep 7 m either side of the circular path centred at (0, arc_R) */
    const double dr = sqrt(cx * cx + (cy - arc_R) * (cy - arc_R)) - fabs(arc_R);
    o->cx = cx; o->cy = cy; o->z0 = -1.73;
    if (kind < 0.45) {          /* building */
        o->type = 0; o->hx = 2.0 + 4.0 * u01(hash32(h + 4)); o->hy = 2.0 + 4.0 * u01(hash32(h + 5));
        o->z1 = o->z0 + 3.0 + 12.0 * u01(hash32(h + 6));
        if (fabs(dr) < 7.0 + sqrt(o->hx * o->hx + o->hy * o->hy)) return 0;
    } else if (kind < 0.65) {   /* parked car */
        o->type = 0; o->hx = 2.2; o->hy = 0.9; o->z1 = o->z0 + 1.5;
        if (u01(hash32(h + 7)) < 0.5) { o->hx = 0.9; o->hy = 2.2; }
        if (fabs(dr) < 4.5) return 0;
    } else if (kind < 0.90) {   /* pole */
        o->type = 1; o->r = 0.15; o->z1 = o->z0 + 4.0 + 4.0 * u01(hash32(h + 8));
        if (fabs(dr) < 3.5) return 0;
    } else return 0;
    return 1;
}

void ll_synth_default(ll_synth_cfg *c, int rings)
{
    memset(c, 0, sizeof(*c));
    c->rings = rings;
    c->azimuths = (rings == 16) ? 1800 : 2048;
    if (rings == 16) { c->elev_lo_deg = -15.0; c->elev_hi_deg = 15.0; }
    else if (rings == 32) { c->elev_lo_deg = -30.0; c->elev_hi_deg = 11.0 + 1.0 / 3.0; }   /* bin centres of :153: angle = (id+0.5)*4/3 - 92/3 */
    else if (rings == 64) { c->elev_lo_deg = -24.9; c->elev_hi_deg = 2.0; }
    else { c->elev_lo_deg = -25.0; c->elev_hi_deg = 15.0; }
    c->range_sigma = 0.02; c->max_range = 120.0; c->az_jitter_deg = 0.0; c->order = 0;
    c->seed = 0x5EED0000u; c->speed = 10.0; c->yaw_rate = 0.1; c->period = 0.1;
    c->drop_prob = 0.0; c->emit_nan = 0;
}

/* pose of scan k: x, y, yaw (world) */
void ll_synth_pose(const ll_synth_cfg *c, int k, double pose[3])
{
    const double th = c->yaw_rate * c->period * k;
    if (fabs(c->yaw_rate) < 1e-12) { pose[0] = c->speed * c->period * k; pose[1] = 0.0; pose[2] = 0.0; return; }
    const double R = c->speed / c->yaw_rate;
    pose[0] = R * sin(th); pose[1] = R * (1.0 - cos(th)); pose[2] = th;
}

/* returns number of float4 written to out (capacity rings*azimuths) */
int ll_synth_scan(const ll_synth_cfg *c, int k, float *out)
{
    const int W = c->azimuths, Rn = c->rings;
    double pose[3];
    ll_synth_pose(c, k, pose);
    const double arc_R = (fabs(c->yaw_rate) < 1e-12) ? 1e12 : c->speed / c->yaw_rate;
    const double cy_ = cos(pose[2]), sy_ = sin(pose[2]);

    /* candidate objects within reach */
    const int span = (int)(c->max_range / CELL) + 2;
    const int cix = (int)floor(pose[0] / CELL), ciy = (int)floor(pose[1] / CELL);
    obj_t *objs = (obj_t *)malloc(sizeof(obj_t) * (size_t)(2 * span + 1) * (2 * span + 1));
    int nobj = 0;
    for (int iy = ciy - span; iy <= ciy + span; ++iy)
        for (int ix = cix - span; ix <= cix + span; ++ix) {
            obj_t o;
            if (!cell_object(c->seed, ix, iy, arc_R, &o)) continue;
            const double dx = o.cx - pose[0], dy = o.cy - pose[1];
            if (sqrt(dx * dx + dy * dy) > c->max_range + 12.0) continue;
            objs[nobj++] = o;
        }

    float *tmp = (float *)malloc(sizeof(float) * 4 * (size_t)W * Rn);   /* slot per (ring, column); x = NaN marks "no return" */
#pragma omp parallel for schedule(dynamic, 16)
    for (int j = 0; j < W; ++j) {
        for (int r = 0; r < Rn; ++r) {
            const double el = (Rn > 1 ? c->elev_lo_deg + (c->elev_hi_deg - c->elev_lo_deg) * r / (Rn - 1) : c->elev_lo_deg) * M_PI / 180.0;
            const double jit = c->az_jitter_deg * (2.0 * u01(hash3(c->seed, 0xA11CEu, (uint32_t)r)) - 1.0) * M_PI / 180.0;
            const double az = -2.0 * M_PI * j / W + jit;                  /* clockwise sweep: -atan2(y,x) grows with j */
            const double ce = cos(el), se = sin(el);
            const double dsx = ce * cos(az), dsy = ce * sin(az), dsz = se; /* sensor frame */
            const double dwx = cy_ * dsx - sy_ * dsy, dwy = sy_ * dsx + cy_ * dsy, dwz = dsz;
            double best = 1e30;
            if (dwz < -1e-9) { const double tg = -1.73 / dwz; if (tg < best) best = tg; }
            for (int i = 0; i < nobj; ++i) {
                const obj_t *o = &objs[i];
                const double ox = pose[0] - o->cx, oy = pose[1] - o->cy;
                if (o->type == 0) {       /* slab test */
                    double t0 = 0.0, t1 = best;
                    const double lo[3] = {-o->hx, -o->hy, o->z0}, hi[3] = {o->hx, o->hy, o->z1};
                    const double org[3] = {ox, oy, 0.0}, dir[3] = {dwx, dwy, dwz};
                    int miss = 0;
                    for (int a = 0; a < 3 && !miss; ++a) {
                        if (fabs(dir[a]) < 1e-12) { if (org[a] < lo[a] || org[a] > hi[a]) miss = 1; }
                        else {
                            double ta = (lo[a] - org[a]) / dir[a], tb = (hi[a] - org[a]) / dir[a];
                            if (ta > tb) { const double s = ta; ta = tb; tb = s; }
                            if (ta > t0) t0 = ta;
                            if (tb < t1) t1 = tb;
                            if (t0 > t1) miss = 1;
                        }
                    }
                    if (!miss && t0 > 1e-6 && t0 < best) best = t0;
                } else {                  /* infinite cylinder clipped in z */
                    const double a = dwx * dwx + dwy * dwy, b = ox * dwx + oy * dwy, cc = ox * ox + oy * oy - o->r * o->r;
                    const double disc = b * b - a * cc;
                    if (a > 1e-12 && disc >= 0.0) {
                        const double tt = (-b - sqrt(disc)) / a;
                        const double z = tt * dwz;
                        if (tt > 1e-6 && tt < best && z >= o->z0 && z <= o->z1) best = tt;
                    }
                }
            }
            float *p = &tmp[4 * ((size_t)r * W + j)];
            const uint32_t hk = hash3(c->seed + (uint32_t)k, (uint32_t)r, (uint32_t)j);
            int have = best <= c->max_range;
            if (have && c->drop_prob > 0.0 && u01(hash32(hk + 11)) < c->drop_prob) have = 0;
            if (!have) { p[0] = p[1] = p[2] = NAN; p[3] = 0.0f; continue; }
            const double u1 = u01(hash32(hk + 1)), u2 = u01(hash32(hk + 2));
            const double gn = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
            const double rng = best + c->range_sigma * gn;
            p[0] = (float)(rng * dsx); p[1] = (float)(rng * dsy); p[2] = (float)(rng * dsz);
            p[3] = (float)u01(hash32(hk + 3));
        }
    }
    int n = 0;
    if (c->order == 0) {
        for (int r = 0; r < Rn; ++r) for (int j = 0; j < W; ++j) {
            const float *p = &tmp[4 * ((size_t)r * W + j)];
            if (p[0] != p[0] && !c->emit_nan) continue;
            memcpy(&out[4 * (size_t)n], p, 16); n++;
        }
    } else {
        for (int j = 0; j < W; ++j) for (int r = 0; r < Rn; ++r) {
            const float *p = &tmp[4 * ((size_t)r * W + j)];
            if (p[0] != p[0] && !c->emit_nan) continue;
            memcpy(&out[4 * (size_t)n], p, 16); n++;
        }
    }
    free(tmp); free(objs);
    return n;
}
