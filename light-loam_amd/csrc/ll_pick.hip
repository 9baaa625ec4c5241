/*
 * ll_pick.hip -- a2 + a3: curvature and the per-segment greedy feature pick, ONE WAVE PER RING, no barriers.
 * Replaces scanRegistration.cpp:225-235 and :246-359 of /root/reference (the less-flat VoxelGrid, :361-376, is the next
 * kernel: ll_features.hip).
 *
 * Why a wave and not a workgroup: the reference's unit of sequential dependence is the ring (cloudNeighborPicked marks leak
 * from segment j into segment j + 1 of the same ring, never across rings), and inside a ring the pick is a chain of ~150
 * dependent arg-max steps.  Rounds 1-3 spread a ring over four waves that met at ~60 barriers; the waves spent 58 % of their
 * life parked and no unit of the CU was busy more than half the time.  Here a wave walks its ring segment by segment:
 *   - the segment's points (+5 halo either side) come into a wave-private LDS tile by LDS-DMA (global_load_lds_dwordx4: no
 *     staging registers, the next segment's first tile is in flight during the pick),
 *   - the eleven taps are ds_read_b128 of consecutive float4 (one instruction per tap, 4 LDS cycles), the sums strictly left
 *     to right in f32 (:228-230), the consecutive-point gap test (:290-293) is a by-product of two taps,
 *   - the gap flags never touch memory: one ballot per row, the suppression extents of a point (how far the :288-311 walks
 *     get) are ten bits of the ballot words picked out with v_alignbit,
 *   - the pick itself is the sort-free arg-max loop of rounds 1-3 (lane-local max, DPP wave max, ballot; see k_ring_features'
 *     history in DESIGN.md), with the earlier segments' forward marks imported eagerly -- they are complete, the wave made them,
 *   - labels, the picked points' local indices (sharp / less-sharp / flat, per segment) and the ring's counts go to HBM; the
 *     next kernel turns indices into published clouds.
 * No __syncthreads, no cross-wave traffic: a workgroup is LL_PK_WAVES independent rings, 8 waves per SIMD.
 */
#include "ll_common.h"
#include <limits.h>
#include <type_traits>

typedef float ll_f4 __attribute__((ext_vector_type(4)));
typedef float ll_f2 __attribute__((ext_vector_type(2)));
/* rows of 64 segment points per curvature tile (+ 5 halo points either side): three -- 4.6 KB of LDS per wave, eight waves per SIMD;
 * a segment's first tile comes in during the previous segment's pick, its second is waited for in the open.  Whole-segment tiles
 * (LL_PK_TILE_ROWS >= 6: one wait per segment, 7.7 KB, five waves per SIMD) and double-buffered tiles measured slower. */
#ifndef LL_PK_TILE_ROWS
#define LL_PK_TILE_ROWS 3
#endif
#define LL_PK_TR (SR < LL_PK_TILE_ROWS ? SR : LL_PK_TILE_ROWS)
#define LL_PK_TILE (LL_PK_TR * 64 + 10)
#ifndef LL_PK_CROWS
#define LL_PK_CROWS 3                            /* rows of compacted corner candidates the pick knows how to scan */
#endif
#define LL_PK_COMPACT (64 * LL_PK_CROWS)                       /* corner candidates compacted to the front rows when at most this many */

typedef __attribute__((address_space(3))) void ll_lds_void;
typedef const __attribute__((address_space(1))) void ll_glb_void;

template <int SR>
struct PickLds {
    union {
        ll_f4 tile[LL_PK_TILE + 6];             /* 16-byte records: ds_read_b128 taps */
        struct { unsigned wkey[LL_PK_COMPACT], wli[LL_PK_COMPACT]; } c;   /* compacted corner candidates (between the segment's last tap and the
                                                                         * next segment's first tile): curvature bits; local index | extents << 16 */
    };
    unsigned picked[(SR * 64 * 6 + 16 + 31) / 32 + 2];   /* cloudNeighborPicked over the ring's local indices */
    unsigned lab2[(SR * 64 * 6 + 16 + 15) / 16 + 2];     /* cloudLabel, two bits per local index: 0, 1, 2, 3 = -1 */
#ifdef LL_PK_LDS_PAD
    unsigned char pad[LL_PK_LDS_PAD];           /* timing builds: fewer waves per SIMD */
#endif
    unsigned short rec[LL_REC_U16];             /* the ring's lists (ring_rec layout); stored when the ring is done: no store inside the
                                                 * segment loop, so every vmcnt wait there is a wait for a tile and nothing else */
    unsigned gw[2 * (SR + 2) + 2];              /* gap flags of the segment: 64-bit word k + 1 = bit l <-> local index sp + k * 64 + l + 5, k = -1 .. SR */
};

/* wave-wide max of a u32, result uniform: quad swaps, row shifts, row broadcasts (DPP), then lane 63 */
__device__ __forceinline__ unsigned ll_pk_wave_max_u32(unsigned v)
{
#define LL_DPP_MAX(ctrl, rmask) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false))
    LL_DPP_MAX(0xb1, 0xf);        /* quad_perm [1,0,3,2] */
    LL_DPP_MAX(0x4e, 0xf);        /* quad_perm [2,3,0,1] */
    LL_DPP_MAX(0x114, 0xf);       /* row_shr:4 */
    LL_DPP_MAX(0x118, 0xf);       /* row_shr:8  -> lanes 12..15 of a row hold the row max */
    LL_DPP_MAX(0x142, 0xa);       /* row_bcast:15 into rows 1, 3 */
    LL_DPP_MAX(0x143, 0xc);       /* row_bcast:31 into rows 2, 3 */
#undef LL_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ bool ll_pk_bit(const unsigned *bm, int i) { return (bm[i >> 5] >> (i & 31)) & 1u; }

/* the ring r of slot s, if ring_lo < its length <= ring_hi (ring_hi <= 384 SR + 6 keeps a segment within SR rows): one wave */
template <int SR>
__device__ __forceinline__ void ll_ring_pick_ring(const LLView &V, int s, int r, int ring_lo, int ring_hi)
{
    static_assert(SR <= 32, "row bitmasks are 32 bits wide");
    static_assert(2 * (SR + 2) + 2 <= 64, "the segment's gap words (L.gw) are cleared by one lane each");
    __shared__ __attribute__((aligned(16))) PickLds<SR> lds_all[LL_PK_WAVES];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    /* status, cloudSize and the ring's two offsets in one round trip, the exits one test (separate tests make each load wait for the
     * branch in front of it) */
    const int status = V.hdr[s].status, N = V.hdr[s].n;
    const int off = V.ring_off[(size_t)s * (V.R + 1) + r];
    const int nr = V.ring_off[(size_t)s * (V.R + 1) + r + 1] - off;
    if ((status != 0) | (nr <= ring_lo) | (nr > ring_hi)) return;    /* a refused scan | another tier's ring */
    PickLds<SR> &L = lds_all[wave];
    unsigned short *rec_g = V.ring_rec + ((size_t)s * V.R + r) * LL_REC_U16;
    unsigned *rcnt = V.ring_cnt + (size_t)s * V.R + r;
    int8_t *label = V.label + (size_t)s * V.NP + off;
    const int S = off + 5, E = off + nr - 6;                          /* scanStartInd / scanEndInd (:218-220) */
    const bool active = (nr > 0) && (E - S >= 6);                     /* :248 */
    const int Lseg = active ? (E - S) : 0;
    const float4 *slot_cloud = V.cloud + (size_t)s * V.CS;
    const float4 *ring = slot_cloud + (size_t)r * V.ring_cap;         /* ring[li] = the ring's point li = laserCloud[off + li] */

    if (V.write_curv && nr > 0) {
        /* cloudCurvature of the ring's first five and last six points reaches into the adjacent rings (the reference runs over
         * the flat array, :225-235); they are in no segment, so only the optional curvature output wants them */
        auto cloud_at = [&](int g) -> float4 {
            int l = g - off;
            if (l >= 0 && l < nr) return ring[l];
            const int *ro = V.ring_off + (size_t)s * (V.R + 1);
            int q = r;
            if (l < 0) { do { --q; l += ro[q + 1] - ro[q]; } while (l < 0); }
            else { int c = nr; do { l -= c; ++q; c = ro[q + 1] - ro[q]; } while (l >= c); }
            return slot_cloud[(size_t)q * V.ring_cap + l];
        };
        /* lanes 0..4: li = lane; lanes 5..10: li = nr - 6 + (lane - 5); a ring shorter than 17 points: every point */
        int li = -1;
        if (!active) { if (lane < nr) li = lane; }
        else if (lane < 5) li = lane;
        else if (lane < 11) li = nr - 11 + lane;
        const int g = off + li;
        if (li >= 0 && g >= 5 && g < N - 5) {
            float dX, dY, dZ;
            {
                float4 p = cloud_at(g - 5); dX = p.x; dY = p.y; dZ = p.z;
                for (int d = -4; d <= 5; ++d) {
                    p = cloud_at(g + d);
                    if (d == 0) { dX = dX - 10 * p.x; dY = dY - 10 * p.y; dZ = dZ - 10 * p.z; }
                    else { dX = dX + p.x; dY = dY + p.y; dZ = dZ + p.z; }
                }
            }
            V.curv[(size_t)s * V.NP + g] = dX * dX + dY * dY + dZ * dZ;
        }
    }
    if (!active) {                                                    /* no segments: every label 0, no features (also nr <= 0) */
        for (int i = lane; i < nr; i += 64) label[i] = 0;
        if (lane < 3 * LL_SEGS) rec_g[156 + lane] = 0;
        if (lane == 0) *rcnt = 0u;
        return;
    }
    {
        const int nwords = (nr + 31) / 32 + 1;
        for (int i = lane; i < nwords; i += 64) L.picked[i] = 0;
        for (int i = lane; i < 2 * nwords; i += 64) L.lab2[i] = 0;   /* the points outside the segments keep label 0 */
    }
    unsigned short *rec = L.rec;

    /* the tile of rows [k0, k0 + TR) of segment [sp, sp + len): tile[t] = ring[sp + k0 * 64 + t], t < TR * 64 + 10, as far as the
     * segment + halo reaches (sp + len + 9 <= nr - 2) */
    auto tile_dma = [&](int sp, int len, int k0) __attribute__((always_inline)) {
        const int lim = len + 10 - k0 * 64;                           /* tile slots that exist */
        const float4 *src = ring + sp + k0 * 64;
#pragma unroll
        for (int u = 0; u <= LL_PK_TR; ++u) {
            const int t = u * 64 + lane;
#ifndef LL_PK_TIMING_NODMA
            if (u * 64 < lim && t < LL_PK_TILE && t < lim)
#else
            if (false)
#endif
                __builtin_amdgcn_global_load_lds((ll_glb_void *)(src + t), (ll_lds_void *)(L.tile + u * 64), 16, 0, 0);
        }
    };
    auto tile_wait = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0f70);                           /* vmcnt(0): the LDS-DMA has landed */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto gap_at = [&](int t) -> bool {                                /* (:290-293) squared distance of tile point t to its predecessor > 0.05 */
        const ll_f4 a = L.tile[t], b = L.tile[t - 1];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        return dx * dx + dy * dy + dz * dz > V.gap_gt;
    };

    unsigned segc = 0;                                                /* scalar: n_sharp, n_lsharp, n_flat totals, 8 bits each */
    /* lane constants of the extents window: bits [60 + lane, 70 + lane) of the 192-bit string word[k-1] : word[k] : word[k+1] */
    const int wsh = (60 + lane) & 31, wwi = (60 + lane) >> 5;         /* first 32-bit word of the window: 1, 2 or 3 */

    tile_dma(0, Lseg * 1 / 6, 0);                                     /* segment 0: sp = 0, len = ep + 1 */
    for (int j = 0; j < LL_SEGS; ++j) {
        const int sp = Lseg * j / 6, ep = Lseg * (j + 1) / 6 - 1;     /* record slots; = (:253-254) - S */
        const int len = ep - sp + 1;
        const int nrows = (len + 63) >> 6;                            /* <= SR, uniform */
        const int li0 = sp + 5 + lane;                                /* local index of this lane's row-0 point */
        unsigned cb[SR];                                              /* curvature bits per row */
        auto gw_put = [&](int word, unsigned long long w) __attribute__((always_inline)) {    /* one lane stores the ballot */
            if (lane == 0) { L.gw[2 * word] = (unsigned)w; L.gw[2 * word + 1] = (unsigned)(w >> 32); }
        };
        if (lane < 2 * (SR + 2) + 2) L.gw[lane] = 0u;
        /* ---------------- curvature + gap flags, tile by tile ---------------- */
#pragma unroll
        for (int k0 = 0; k0 < SR; k0 += LL_PK_TR) {
            if (k0 < nrows) {
                if (k0 > 0) tile_dma(sp, len, k0);
                tile_wait();
                if (k0 == 0) {                                        /* the four flags below the first centre: local index sp + 1 .. sp + 4 */
                    const bool g = (lane >= 1 && lane < 5) ? gap_at(lane) : false;
                    gw_put(0, __ballot(g) << 59);                     /* lane 1 -> bit 60 (= centre -4) */
                }
#pragma unroll
                for (int kk = 0; kk < LL_PK_TR; ++kk) {
                    const int k = k0 + kk;
                    if (k < SR && k < nrows) {
                        const int q = k * 64 + lane;
                        const ll_f4 *C = L.tile + kk * 64 + lane + 5;
                        /* :225-235, strict left to right.  Every tap is one ds_read_b128 (4 LDS cycles; the 12-byte read the compiler
                         * picks when w is unused takes 8) and two packed adds -- (x, y) and (z, w): the w lane rides along and is
                         * "used" once at the end so that the reads stay 16 bytes wide.  Six taps in flight, then five. */
                        ll_f2 xy, zw;
                        bool gf;
                        {
                            const ll_f4 t0 = C[-5], t1 = C[-4], t2 = C[-3], t3 = C[-2], t4 = C[-1], t5 = C[0];
                            xy = t0.xy + t1.xy; zw = t0.zw + t1.zw;
                            xy = xy + t2.xy; zw = zw + t2.zw;
                            xy = xy + t3.xy; zw = zw + t3.zw;
                            xy = xy + t4.xy; zw = zw + t4.zw;
                            xy = xy - 10.0f * t5.xy; zw = zw - 10.0f * t5.zw;
                            const float gx = t5.x - t4.x, gy = t5.y - t4.y, gz = t5.z - t4.z;
                            gf = (q < len + 5) && (gx * gx + gy * gy + gz * gz > V.gap_gt);   /* centres up to local index ep + 10 */
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        {
                            const ll_f4 t6 = C[1], t7 = C[2], t8 = C[3], t9 = C[4], t10 = C[5];
                            xy = xy + t6.xy; zw = zw + t6.zw;
                            xy = xy + t7.xy; zw = zw + t7.zw;
                            xy = xy + t8.xy; zw = zw + t8.zw;
                            xy = xy + t9.xy; zw = zw + t9.zw;
                            xy = xy + t10.xy; zw = zw + t10.zw;
                        }
                        asm volatile("" :: "v"(zw.y));
                        const float cv = xy.x * xy.x + xy.y * xy.y + zw.x * zw.x;
                        gw_put(k + 1, __ballot(gf));
                        __builtin_amdgcn_sched_barrier(0);
                        cb[k] = (q < len) ? ll_f2u(cv) : 0u;
                        if (V.write_curv && q < len) V.curv[(size_t)s * V.NP + S + sp + q] = cv;
                    } else if (k < SR) cb[k] = 0u;
                }
                if (k0 + LL_PK_TR >= nrows) {                         /* the last tile: flags beyond the last row's centres, local index sp + nrows * 64 + 5 .. ep + 10 */
                    const int b = nrows * 64 + 5 + lane;              /* relative to sp */
                    const bool g = (lane < 5 && b <= len + 9) ? gap_at(b - k0 * 64) : false;
                    gw_put(nrows + 1, __ballot(g));
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < LL_PK_TR; ++kk) if (k0 + kk < SR) cb[k0 + kk] = 0u;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                           /* lgkmcnt(0): every tap has been read, the tile is dead */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        /* ---------------- suppression extents (:288-311) ----------------
         * A pick at local index li marks li - bn .. li + fn: fn / bn = how far the forward / backward walk gets before a
         * consecutive-point gap above the threshold stops it: ten flags li - 4 .. li + 5 = bits (k * 64 + lane) - 4 .. + 5 of W' */
        constexpr int EW = (SR + 3) / 4;
        unsigned exw[EW];                                             /* extents bn | fn << 4, four rows per word */
        unsigned candc = 0, candf = 0, sup = 0;                       /* row bitmasks: corner / flat candidate, suppressed */
#pragma unroll
        for (int w = 0; w < EW; ++w) exw[w] = 0u;
#pragma unroll
        for (int k = 0; k < SR; ++k) {
            if (k < nrows) {
                const unsigned lo = L.gw[2 * k + wwi], hi = L.gw[2 * k + wwi + 1];           /* one ds_read2_b32 */
                const unsigned bits = __builtin_amdgcn_alignbit(hi, lo, (unsigned)wsh) & 0x3ffu;   /* bit t = gap flag of li - 4 + t */
                const unsigned fwd = bits >> 5;                                              /* l = 1..5  -> flag[li + l] */
                const int fn = fwd ? (__ffs(fwd) - 1) : 5;
                const unsigned bwd = bits & 0x1fu;                                           /* l = -1..-5 -> flag[li + l + 1] = bits 4..0 */
                const int bn = bwd ? 4 - (31 - __clz((int)bwd)) : 5;
                const int q = k * 64 + lane;
                if (q < len) {
                    exw[k >> 2] |= (unsigned)(bn | (fn << 4)) << ((k & 3) * 8);
                    const float cv = ll_u2f(cb[k]);                   /* f32 curvature against the double literal 0.1, in f32 */
                    if (cv > V.curv_gt) candc |= 1u << k;             /* :266 */
                    if (cv < V.curv_lt) candf |= 1u << k;             /* :321 */
                }
            }
        }
        /* the earlier segments' forward marks reach at most this segment's first five points; they are all made (this wave made
         * them): a marked point is no corner candidate, and the flat pass reads the bitmap anyway */
        if (j > 0 && lane < 5 && lane < len && ll_pk_bit(L.picked, li0)) candc &= ~1u;

        /* ---------------- the greedy pick (:251-359), no sort ----------------
         * Visiting a segment in descending (curvature, index) order and taking every candidate that is not yet suppressed is
         * the same as repeatedly taking the arg-max over the still-eligible candidates, because suppression only grows;
         * likewise arg-min for the flats.  std::sort leaves equal curvatures unspecified; this path and the oracle define
         * ascending index. */
        /* the corner candidates (usually a small part of the segment) compacted to the front rows, ascending index */
        int nc = 0;
#pragma unroll
        for (int k = 0; k < SR; ++k) {
            if (k < nrows) {
                const bool c = (candc >> k) & 1u;
                const unsigned long long m = __ballot(c);
                const int pos = nc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (c && pos < LL_PK_COMPACT) {
                    L.c.wkey[pos] = cb[k];
                    L.c.wli[pos] = (unsigned)(li0 + k * 64) | (((exw[k >> 2] >> ((k & 3) * 8)) & 0xffu) << 16);
                }
                nc += __popcll(m);
            }
        }
        if (lane < 4 && nc + lane < LL_PK_COMPACT) L.c.wkey[nc + lane] = 0u;        /* the rank loop below reads four keys at a time: zeros behind the last */
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const bool compact = nc <= LL_PK_COMPACT;                     /* else: a segment full of corners, its rows as they are */
        const int ncr = (nc + 63) >> 6;
        unsigned ck[LL_PK_CROWS]; int cli[LL_PK_CROWS];               /* masked key (0 = not eligible); local index | suppression extents << 16 */
#pragma unroll
        for (int rr = 0; rr < LL_PK_CROWS; ++rr) {
            ck[rr] = 0u; cli[rr] = 0;
            if (compact && rr < ncr && rr * 64 + lane < nc) { ck[rr] = L.c.wkey[rr * 64 + lane]; cli[rr] = (int)L.c.wli[rr * 64 + lane]; }
        }
        /* rank of every candidate of a one-row corner pass (descending curvature; sorted_walk below), from the keys while they still lie in LDS:
         * one broadcast read brings four keys to every lane -- two vector instructions per key; fetched lane by lane with v_readlane (round 4)
         * it was three and the scalar-register hazards between them: 0.35 ms of the kernel's 12.3 */
        int rank_pre = 0;
        if (compact && ncr <= 1 && nc > 0) {
            const unsigned key0 = ck[0];
            for (int t = 0; t < nc; t += 4) {
                const uint4 k4 = *(const uint4 *)&L.c.wkey[t];
                rank_pre += (k4.x > key0 ? 1 : 0) + (k4.y > key0 ? 1 : 0) + (k4.z > key0 ? 1 : 0) + (k4.w > key0 ? 1 : 0);
            }
        }
        /* the next segment's first tile travels during the pick (the candidates are in registers, their LDS rows are dead) */
        __builtin_amdgcn_s_waitcnt(0xc07f);                           /* lgkmcnt(0) */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (j + 1 < LL_SEGS) { const int sp1 = ep + 1, ep1 = Lseg * (j + 2) / 6 - 1; tile_dma(sp1, ep1 - sp1 + 1, 0); }
        int nrec[2] = {0, 0};
#ifdef LL_PK_TIMING_NOPICK
        for (int pass = 0; pass < 0; ++pass) {
#elif defined(LL_PK_TIMING_NOFLAT)
        for (int pass = 0; pass < 1; ++pass) {
#else
        for (int pass = 0; pass < 2; ++pass) {
#endif
            int npick = 0;
            unsigned myrec = 0;                                       /* lane n: pick n + 1 as li | extents << 16 */
            /* non-negative float bits order like the floats; the flats maximise the complement */
            unsigned mk[SR];
            if (pass == 1) {
                /* what the corner picks of this segment (and the earlier segments) marked */
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int k = 0; k < SR; ++k) if (k < nrows && k * 64 + lane < len && ll_pk_bit(L.picked, li0 + k * 64)) sup |= 1u << k;
            }
            /* one pick loop, instantiated for the rows it scans: COMPACT -- the corner pass over the compacted candidates (NR = 1 or 2
             * rows of ck / cli); otherwise the segment's own rows (mk): the flat pass, or the corner pass of a segment with more than
             * LL_PK_COMPACT candidates */
            auto pick_loop = [&](auto nr_tag, auto corner_tag, auto compact_tag, auto &key, int (&cli)[LL_PK_CROWS]) __attribute__((always_inline)) {
                constexpr int NR = decltype(nr_tag)::value;
                constexpr bool CORNER = decltype(corner_tag)::value;
                constexpr bool COMPACT = decltype(compact_tag)::value;
                for (;;) {
                    /* lane-local best; rows ascend in index, so on equal keys ">=" keeps the larger index (descending
                     * visit order of the corners) and ">" the smaller (ascending order of the flats) */
                    unsigned best = key[0]; int row_l = 0;
#pragma unroll
                    for (int rr = 1; rr < NR; ++rr) {
                        const bool t = CORNER ? key[rr] >= best : key[rr] > best;
                        best = t ? key[rr] : best; row_l = t ? rr : row_l;
                    }
                    const unsigned kmax = ll_pk_wave_max_u32(best);
                    if (kmax == 0u) break;                            /* nothing eligible is left */
                    const unsigned long long bal = __ballot(best == kmax);
                    int selp;                                         /* row * 64 + lane of the choice */
                    if (COMPACT && NR == 1) {
                        /* one row of candidates in ascending index: among equal curvatures the highest lane is the largest
                         * index, which the descending walk of the corners meets first, the lowest lane the smallest, which the
                         * ascending walk of the flats meets first -- no tie path */
                        selp = CORNER ? 63 - __builtin_clzll(bal) : __ffsll((long long)bal) - 1;
                    } else if (__popcll(bal) == 1) {
                        const int f = __ffsll((long long)bal) - 1;
                        selp = (NR > 1 ? __builtin_amdgcn_readlane(row_l, f) * 64 : 0) + f;
                    } else {                                          /* equal curvatures in several lanes: index decides */
                        const int myp = row_l * 64 + lane;
                        const unsigned t = (best == kmax) ? (CORNER ? (unsigned)(myp + 1) : (unsigned)(0x10000 - myp)) : 0u;
                        const unsigned tm = ll_pk_wave_max_u32(t);
                        selp = CORNER ? (int)tm - 1 : 0x10000 - (int)tm;
                    }
                    int sel;                                          /* local index of the choice */
                    int e = 0;                                        /* its suppression extents: bn | fn << 4 */
                    if (COMPACT) {                                    /* compacted layout -> the element's own index, its extents with it */
                        int slv = __builtin_amdgcn_readlane(cli[0], selp & 63);
#pragma unroll
                        for (int rr = 1; rr < NR; ++rr) if ((selp >> 6) == rr) slv = __builtin_amdgcn_readlane(cli[rr], selp & 63);
                        sel = slv & 0xffff;
                        e = slv >> 16;
                    } else sel = sp + selp + 5;
                    npick++;
                    if (CORNER && npick > LL_LSHARP_PER_SEG) break;   /* :281-284 */
                    if (!COMPACT) {                                   /* the owner's extents: uniform row, lane */
                        unsigned ew = 0u;
#pragma unroll
                        for (int w = 0; w < EW; ++w) if ((selp >> 8) == w) ew = (unsigned)__builtin_amdgcn_readlane((int)exw[w], selp & 63);
                        e = (int)((ew >> (((selp >> 6) & 3) * 8)) & 0xffu);
                    }
                    {   /* lane n: pick n + 1 (value and lane come out of scalar instructions: no read / write-lane hazard) */
                        const unsigned recv = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)sel | ((unsigned)e << 16)));
                        const int ln = __builtin_amdgcn_readfirstlane(npick - 1);
                        unsigned m0_keep;                               /* v_writelane takes one scalar operand + m0; m0 is the compiler's: put it back */
                        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                                     : "+v"(myrec), "=&s"(m0_keep) : "s"(recv), "s"(ln));
                    }
                    if (!CORNER && npick >= LL_FLAT_PER_SEG) break;   /* :328-331: labelled, but no marking */
                    const int slo = sel - (e & 15), shi = sel + (e >> 4);
                    if (COMPACT) {                                    /* the marks themselves go to the bitmap after the pass */
#pragma unroll
                        for (int rr = 0; rr < NR; ++rr) key[rr] = ((unsigned)((cli[rr] & 0xffff) - slo) <= (unsigned)(shi - slo)) ? 0u : key[rr];
                    } else {
                        /* rows are 64 apart and a pick marks at most 11 consecutive indices: per lane at most one row is hit */
                        const int dd = shi - li0;
                        const int kl = (dd >= 0 && (dd & 63) <= shi - slo) ? (dd >> 6) : -1;
#pragma unroll
                        for (int k = 0; k < NR; ++k) key[k] = (kl == k) ? 0u : key[k];
                    }
                }
            };
            /* The corner pass over at most 64 candidates WITHOUT the arg-max chain: rank them once (descending curvature; the
             * candidates sit in ascending index, so "larger index first" among equal curvatures would be the higher lane --
             * equal curvatures are left to the arg-max loop instead), bring candidate number i of that order to lane i, and
             * walk a scalar eligibility mask: the next pick is its lowest set bit, a pick clears the lanes whose index lies in
             * its suppression range.  Per pick two v_readlane, one range compare and a handful of scalar instructions instead of
             * six dependent DPP steps with their wait states.  Returns false (nothing done) when two candidates tie. */
            auto sorted_walk = [&]() __attribute__((always_inline)) -> bool {
                int rank = rank_pre;                                  /* candidates with a larger curvature (equal ones: left to the arg-max loop) */
                rank = (lane < nc) ? rank : lane;                     /* the idle lanes keep their places */
                const int sA = __builtin_amdgcn_ds_permute(rank * 4, cli[0]);   /* lane rank <- (local index | extents << 16); 0 where nobody wrote */
                const unsigned long long got = __ballot(sA != 0);
                if (__popcll(got) != nc) return false;                /* two lanes with one rank: equal curvatures */
                const int selv = sA & 0xffff;
                const int sB = (selv - ((sA >> 16) & 15)) | ((((sA >> 16) & 15) + (sA >> 20)) << 16);   /* first marked index | marks beyond it << 16 */
                unsigned long long elig = got;
                while (elig) {
                    const int i = __ffsll((long long)elig) - 1;
                    const int a_p = __builtin_amdgcn_readlane(sA, i), b_p = __builtin_amdgcn_readlane(sB, i);
                    npick++;
                    if (npick > LL_LSHARP_PER_SEG) break;             /* :281-284 */
                    myrec = (lane == npick - 1) ? (unsigned)a_p : myrec;
                    const unsigned long long hit = __ballot((unsigned)(selv - (b_p & 0xffff)) <= (unsigned)(b_p >> 16));
                    elig &= ~hit;                                     /* the pick itself is in its range */
                }
                return true;
            };
            using std::integral_constant;
            using std::true_type; using std::false_type;
            if (pass == 1) {
                /* the flat pass over the segment's own rows (two threshold-compacted variants -- the four picks lie among the 34 smallest
                 * eligible curvatures -- measured slower: DESIGN.md 13.3 liii) */
                const unsigned fm = candf & ~sup;
                int nf = 0;
#pragma unroll
                for (int k = 0; k < SR; ++k) if (k < nrows) nf += __popcll(__ballot((fm >> k) & 1u));
                if (nf > 0) {
#pragma unroll
                    for (int k = 0; k < SR; ++k) mk[k] = ((fm >> k) & 1u) ? ~cb[k] : 0u;
                    pick_loop(integral_constant<int, SR>{}, false_type{}, false_type{}, mk, cli);
                }
            } else if (!compact) {
#pragma unroll
                for (int k = 0; k < SR; ++k) mk[k] = ((candc >> k) & 1u) ? cb[k] : 0u;
                pick_loop(integral_constant<int, SR>{}, true_type{}, false_type{}, mk, cli);
            } else if (ncr <= 1) { if (nc > 0 && !sorted_walk()) pick_loop(integral_constant<int, 1>{}, true_type{}, true_type{}, ck, cli); }
            else if (ncr <= 2) pick_loop(integral_constant<int, 2>{}, true_type{}, true_type{}, ck, cli);
            else pick_loop(integral_constant<int, LL_PK_CROWS>{}, true_type{}, true_type{}, ck, cli);
            /* the picked records, lane-parallel: labels, list entries, marks */
            const int nr_ = pass == 0 ? min(npick, LL_LSHARP_PER_SEG) : npick;
            nrec[pass] = nr_;
            if (lane < nr_) {
                const int sel = (int)(myrec & 0xffffu), e = (int)(myrec >> 16);
                unsigned code;                                        /* cloudLabel (:271, :276, :323) */
                if (pass == 0) {
                    if (lane < LL_SHARP_PER_SEG) { code = 2u; rec[j * LL_SHARP_PER_SEG + lane] = (unsigned short)sel; }
                    else code = 1u;
                    rec[12 + j * LL_LSHARP_PER_SEG + lane] = (unsigned short)sel;
                } else { code = 3u; rec[132 + j * LL_FLAT_PER_SEG + lane] = (unsigned short)sel; }
                atomicOr(&L.lab2[sel >> 4], code << ((sel & 15) * 2));
                /* marks into the bitmap: a corner pick's whole range from this segment's first index on (the flat pass reads it
                 * back), a flat pick's only beyond this segment (inside it the flat pass keeps them in registers) */
                const int shi = sel + (e >> 4), f0 = max(sel - (e & 15), pass == 0 ? sp + 5 : ep + 6);
                if ((pass == 0 || lane < LL_FLAT_PER_SEG - 1) && shi >= f0) {
                    const unsigned long long bits = ((1ull << (shi - f0 + 1)) - 1ull) << (f0 & 31);
                    atomicOr(&L.picked[f0 >> 5], (unsigned)bits);
                    if (bits >> 32) atomicOr(&L.picked[(f0 >> 5) + 1], (unsigned)(bits >> 32));
                }
            }
        }
        if (lane < 3) rec[156 + j * 3 + lane] = (unsigned short)(lane == 0 ? min(nrec[0], LL_SHARP_PER_SEG) : lane == 1 ? nrec[0] : nrec[1]);
        segc += (unsigned)min(nrec[0], LL_SHARP_PER_SEG) | ((unsigned)nrec[0] << 8) | ((unsigned)nrec[1] << 16);
    }
    /* the ring is done: labels, lists, counts */
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int li = lane; li < nr; li += 64) {
        const unsigned c = (L.lab2[li >> 4] >> ((li & 15) * 2)) & 3u;
        label[li] = (int8_t)(c == 3u ? -1 : (int)c);
    }
    if (lane < LL_REC_U16 / 2) ((unsigned *)rec_g)[lane] = ((const unsigned *)L.rec)[lane];
    if (lane + 64 < LL_REC_U16 / 2) ((unsigned *)rec_g)[lane + 64] = ((const unsigned *)L.rec)[lane + 64];
    if (lane == 0) *rcnt = segc | 0x80000000u;                        /* the ring's counts */
}

/* The launch of the common capacity (rings of at most 2304 points): one workgroup per (scan, ring group).  The tiers of longer rings
 * (max_ring_points > 2304: a real HDL-64E under the linear 64-ring model) run over the WORK LIST k_organize filled for them
 * (tier_list: slot << 8 | ring, tier_cnt entries): a fixed grid of waves takes entries in turn, so that a launch costs what its rings
 * cost and not 64 x count workgroups that fetch a header and two offsets to find out they are not wanted. */
template <int SR>
__device__ __forceinline__ void ll_ring_pick_grid(const LLView &V, int first, int count, int ring_lo, int ring_hi)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int groups = (V.R + LL_PK_WAVES - 1) / LL_PK_WAVES;
    int sl, grp;
    {   /* block -> (scan, ring group): the rings of a scan on one XCD (its second kernel reads their lists from that L2) */
        const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
        sl = (jj / groups) * 8 + xcd; grp = jj % groups;
        if (sl >= count) return;
    }
    const int r = grp * LL_PK_WAVES + wave;
    if (r >= V.R) return;
    ll_ring_pick_ring<SR>(V, first + sl, r, ring_lo, ring_hi);
}
template <int SR>
__device__ __forceinline__ void ll_ring_pick_list(const LLView &V, int ring_lo, int ring_hi, const int *list, const int *list_n)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int n = *list_n;
    for (int i = blockIdx.x * LL_PK_WAVES + wave; i < n; i += gridDim.x * LL_PK_WAVES) {
        const int e = list[i];
        ll_ring_pick_ring<SR>(V, e >> 8, e & 0xFF, ring_lo, ring_hi);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       /* the next ring starts on a quiet tile */
    }
}

/* one kernel per row count: the register cap is a property of the kernel (8 / 6 / 4 / 2 waves per SIMD).  The six-row kernel is the
 * grid launch of the common capacity (its signature and body as in round 4: two more kernel arguments cost it 0.2 ms per 16384 scans
 * through the register allocation); the others are the tiers, over their lists */
#define LL_PICK_KERNEL(SR, WAVES, NVGPR)                                                                              \
    __global__ __launch_bounds__(64 * LL_PK_WAVES, WAVES) __attribute__((amdgpu_num_vgpr(NVGPR)))                      \
    void k_ring_pick##SR(LLView V, int first, int count, int ring_lo, int ring_hi) { ll_ring_pick_grid<SR>(V, first, count, ring_lo, ring_hi); }
#define LL_PICK_TIER_KERNEL(SR, WAVES, NVGPR)                                                                         \
    __global__ __launch_bounds__(64 * LL_PK_WAVES, WAVES) __attribute__((amdgpu_num_vgpr(NVGPR)))                      \
    void k_ring_pick##SR(LLView V, int ring_lo, int ring_hi, const int *list, const int *list_n) { ll_ring_pick_list<SR>(V, ring_lo, ring_hi, list, list_n); }
#if LL_PK_TILE_ROWS >= 6
LL_PICK_KERNEL(6, 5, 96)         /* 7.7 KB of LDS per wave: five waves per SIMD */
LL_PICK_TIER_KERNEL(8, 4, 128)
LL_PICK_TIER_KERNEL(12, 2, 128)
LL_PICK_TIER_KERNEL(22, 1, 256)
#else
LL_PICK_KERNEL(6, 8, 64)
LL_PICK_TIER_KERNEL(8, 6, 80)
LL_PICK_TIER_KERNEL(12, 4, 128)
LL_PICK_TIER_KERNEL(22, 2, 256)
#endif

/* a tier of long rings: resident waves over the work list the organise stage filled */
template <typename K>
static void ll_launch_ring_pick_tier(K kernel, const LLView &V, int count, int ring_lo, int ring_hi, int tier, int waves_per_simd, hipStream_t st)
{
    const int groups = (V.R + LL_PK_WAVES - 1) / LL_PK_WAVES;
    int grid = 8 * groups * ((count + 7) / 8);
    const int resident = 256 * 4 * waves_per_simd / LL_PK_WAVES;
    if (grid > resident) grid = resident;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * LL_PK_WAVES), 0, st, V, ring_lo, ring_hi, V.tier_list + (size_t)(tier - 1) * V.B * V.R, V.tier_cnt + tier);
}

/* the tiers of ll_launch_features: rings of at most 2304 points in the six-row instantiation at eight waves per SIMD */
void ll_launch_pick(const LLView &V, int first, int count, hipStream_t st)
{
    const int cap = (V.max_ring + 255) / 256 * 256;
    if (cap > 4608) ll_launch_ring_pick_tier(k_ring_pick22, V, count, 4608, cap, 3, 2, st);
    if (cap > 3072) ll_launch_ring_pick_tier(k_ring_pick12, V, count, 3072, cap < 4608 ? cap : 4608, 2, 4, st);
    if (cap > 2304) ll_launch_ring_pick_tier(k_ring_pick8, V, count, 2304, cap < 3072 ? cap : 3072, 1, 6, st);
    const int groups = (V.R + LL_PK_WAVES - 1) / LL_PK_WAVES;
    hipLaunchKernelGGL(k_ring_pick6, dim3(8 * groups * ((count + 7) / 8)), dim3(64 * LL_PK_WAVES), 0, st, V, first, count, INT_MIN, cap < 2304 ? cap : 2304);
}
