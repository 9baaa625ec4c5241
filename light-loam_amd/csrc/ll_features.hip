/*
 * ll_features.hip -- a2 + a3 + a4: curvature, per-segment sort, greedy feature pick, less-flat VoxelGrid.
 * Replaces scanRegistration.cpp:225-235, :246-368, :370-376 of /root/reference.
 *
 * One 256-thread workgroup per (scan, ring): the ring is the reference's unit of sequential dependence
 * (cloudNeighborPicked marks leak from segment j into segment j+1 of the same ring, never across rings,
 * because scanStartInd/EndInd keep a 5-point margin).  Everything between reading laserCloud and writing
 * labels + feature points stays on chip:
 *   phase 1  512-point tiles of x/y/z (+5 halo, the flat array is used like the reference: curvature
 *            crosses ring boundaries) staged in LDS -> 11-tap curvature (strict left-to-right f32, no FMA),
 *            consecutive-point gap flags (1 bit/point) and sort records (u32 curvature bits, u16 local index)
 *   phase 2  stable LSD radix sort (4-bit digits, ballot ranking) by curvature, then by segment: the six
 *            segments end up sorted in place by (curvature, index) -- std::sort leaves equal curvatures
 *            unspecified, this path and the oracle define ascending index
 *   phase 3  wave 0 replays the greedy pick: 64 candidates per step, eligibility and each candidate's suppression
 *            range in registers, a pick = ballot -> readlane -> range compare
 *   phase 4  less-flat points (label <= 0) compacted in index order, voxel index per PCL's formula, radix-sorted by
 *            (voxel, input order), one thread per voxel run sums in input order
 *   phase 5  labels + feature slots out; k_compact turns per-ring slots into the published clouds.
 * The kernel is latency-bound (dependent LDS round trips, one serial phase), so LDS is kept at ~26 KB for a
 * 2304-point ring capacity: six workgroups per CU.  ROWS = sort records per thread (capacity 256 * ROWS).
 * HBM traffic per ring point: 16 B read (+ L2-hot re-reads for the centroid gather), 1 B label, features.
 */
#include "ll_common.h"
#include <limits.h>

__device__ __forceinline__ bool ll_xcd_map2(int id, int per_scan, int count, int &scan, int &item)
{
    const int xcd = id & 7, j = id >> 3;
    scan = (j / per_scan) * 8 + xcd;
    item = j % per_scan;
    return scan < count;
}

#define LL_FTILE 512      /* points per curvature tile */
#define LL_NLIST 160      /* sharp[12] lsharp[120] flat[24] + 3 counters */

struct FeatLds {
    unsigned *k32;             /* [mr] sort key: curvature bits / voxel index */
    unsigned short *k16;       /* [mr] payload: local index / input order */
    float *tx, *ty, *tz;       /* [LL_FTILE + 16] phase 1; afterwards tx.. is reused as lf_list (u16 [mr]) */
    unsigned short *lf_list;   /* aliases the tile */
    unsigned *picked, *gapf;   /* bitmaps over local index */
    int8_t *lab;               /* [mr] */
    int *lists;                /* [LL_NLIST] */
    int *cnt;                  /* [16 * ROWS * 4 + 1] radix counters */
    int *sc;                   /* [64] scan scratch, bounds, segment table */
};

static size_t ll_feat_tile_bytes(size_t mr) { const size_t t = 3 * 4 * (size_t)(LL_FTILE + 16); return t > 2 * mr ? t : 2 * mr; }

size_t ll_features_lds_bytes(int max_ring)
{
    const size_t mr = (size_t)((max_ring + 255) / 256 * 256);
    const size_t rows = (mr / 256 <= 9) ? 9 : (mr / 256 <= 18) ? 18 : 36;     /* the ROWS instantiation that will run */
    size_t b = 4 * mr + 2 * mr;              /* k32 + k16 */
    b += ll_feat_tile_bytes(mr);             /* tile / lf_list */
    b += 2 * 4 * (mr / 32 + 2);              /* bitmaps */
    b += mr;                                 /* labels */
    b += 4 * LL_NLIST;
    b += 4 * (16 * rows * 4 + 4);            /* radix counters */
    b += 4 * 64;
    return (b + 15) / 16 * 16 + 64;
}

__device__ __forceinline__ FeatLds ll_carve(unsigned char *base, int max_ring)
{
    const size_t mr = (size_t)((max_ring + 255) / 256 * 256);
    const size_t rows = (mr / 256 <= 9) ? 9 : (mr / 256 <= 18) ? 18 : 36;     /* the ROWS instantiation that will run */
    const size_t tile = 3 * 4 * (size_t)(LL_FTILE + 16);
    const size_t tile_bytes = tile > 2 * mr ? tile : 2 * mr;
    FeatLds L;
    unsigned char *p = base;
    L.k32 = (unsigned *)p; p += 4 * mr;
    L.tx = (float *)p; L.ty = L.tx + (LL_FTILE + 16); L.tz = L.ty + (LL_FTILE + 16);
    L.lf_list = (unsigned short *)p; p += tile_bytes;
    L.picked = (unsigned *)p; p += 4 * (mr / 32 + 2);
    L.gapf = (unsigned *)p; p += 4 * (mr / 32 + 2);
    L.lists = (int *)p; p += 4 * LL_NLIST;
    L.cnt = (int *)p; p += 4 * (16 * rows * 4 + 4);
    L.sc = (int *)p; p += 4 * 64;
    L.k16 = (unsigned short *)p; p += 2 * mr;
    L.lab = (int8_t *)p;
    return L;
}

/* Stable least-significant-digit radix sort of the records (k32[g], k16[g]), g < n <= 256*ROWS, by k32; one workgroup.
 * 4-bit digits.  Record g lives in row g / 256 of thread g % 256 (registers); its destination is
 *   #(records with a smaller digit) + #(same digit, earlier (row, wave)) + rank inside its wave,
 * the first two from one workgroup exclusive scan over the digit-major table cnt[digit][row*4 + wave], the last from a
 * ballot match (5 ballots).  Digits that are equal for every key are skipped (their pass would be the identity).
 * Stability makes the result ordered by (k32, original position): exactly the (curvature, index) / (voxel, input
 * order) orders the oracle defines.  SEG_PASS appends a pass on the segment of the local index in k16, so the six
 * curvature segments end up sorted in place in their own ranges. */
template <int ROWS, bool SEG_PASS>
__device__ __forceinline__ void ll_radix_sort(unsigned *k32, unsigned short *k16, int n, int *cnt, int *sc, const int *segb, int tid)
{
    constexpr int SLOTS = ROWS * (LL_BLOCK / 64);               /* (row, wave) pairs */
    constexpr int NCNT = 16 * SLOTS;
    constexpr int PER = (NCNT + LL_BLOCK - 1) / LL_BLOCK;       /* counters per thread in the scan */
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned e32[ROWS]; unsigned short e16[ROWS];
    unsigned vary = 0;
    const unsigned hi0 = (n > 0) ? k32[0] : 0u;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int g = k * LL_BLOCK + tid;
        e32[k] = (g < n) ? k32[g] : 0xffffffffu;
        e16[k] = (g < n) ? k16[g] : (unsigned short)0;
        if (g < n) vary |= e32[k] ^ hi0;
    }
    for (int o = 32; o > 0; o >>= 1) vary |= __shfl_xor(vary, o);
    if (tid == 0) cnt[NCNT] = 0;
    __syncthreads();
    if (lane == 0 && vary) atomicOr((unsigned *)&cnt[NCNT], vary);
    __syncthreads();
    vary = (unsigned)cnt[NCNT];
    int sb1 = 0, sb2 = 0, sb3 = 0, sb4 = 0, sb5 = 0;
    if (SEG_PASS) { sb1 = segb[1]; sb2 = segb[2]; sb3 = segb[3]; sb4 = segb[4]; sb5 = segb[5]; }
    for (int sh = 0; sh < (SEG_PASS ? 36 : 32); sh += 4) {
        const bool segpass = sh >= 32;
        if (!segpass && ((vary >> sh) & 15u) == 0u) continue;
        for (int i = tid; i < NCNT; i += LL_BLOCK) cnt[i] = 0;
        __syncthreads();
        int dig[ROWS], rnk[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const int g = k * LL_BLOCK + tid;
            const bool valid = g < n;
            int d;
            if (segpass) { const int q = (int)e16[k] - 5; d = (q >= sb1) + (q >= sb2) + (q >= sb3) + (q >= sb4) + (q >= sb5); }
            else d = (int)((e32[k] >> sh) & 15u);
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool bit = (d >> b) & 1;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            dig[k] = d; rnk[k] = __popcll(m & lt);
            if (valid && rnk[k] == 0) cnt[d * SLOTS + k * (LL_BLOCK / 64) + wave] = __popcll(m);
        }
        __syncthreads();
        {   /* exclusive scan of the digit-major table: PER consecutive counters per thread */
            const int i0 = tid * PER;
            int v[PER]; int s = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u) { v[u] = (i0 + u < NCNT) ? cnt[i0 + u] : 0; s += v[u]; }
            int total;
            int run = ll_block_exscan(s, sc, total);
#pragma unroll
            for (int u = 0; u < PER; ++u) { if (i0 + u < NCNT) cnt[i0 + u] = run; run += v[u]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const int g = k * LL_BLOCK + tid;
            if (g < n) { const int pos = cnt[dig[k] * SLOTS + k * (LL_BLOCK / 64) + wave] + rnk[k]; k32[pos] = e32[k]; k16[pos] = e16[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const int g = k * LL_BLOCK + tid;
            e32[k] = (g < n) ? k32[g] : 0xffffffffu;
            e16[k] = (g < n) ? k16[g] : (unsigned short)0;
        }
    }
}

__device__ __forceinline__ bool ll_bit(const unsigned *bm, int i) { return (bm[i >> 5] >> (i & 31)) & 1u; }

extern __shared__ __attribute__((aligned(16))) unsigned char ll_smem[];

/* opt-in phase timing (tools/phase_timing.py builds with -DLL_PHASE_TIMING): thread 0 of every workgroup adds the
 * s_memtime cycles spent in each phase to V.dbg[phase]; V.dbg[15] counts workgroups */
#ifdef LL_PHASE_TIMING
#define LL_PHASE_BEGIN() long long ll_t0 = (tid == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0
#define LL_PHASE(i) do { __syncthreads(); if (tid == 0) { const long long t1 = (long long)__builtin_amdgcn_s_memtime(); \
    atomicAdd(&V.dbg[i], (unsigned long long)(t1 - ll_t0)); ll_t0 = t1; } } while (0)
#else
#define LL_PHASE_BEGIN() do {} while (0)
#define LL_PHASE(i) do {} while (0)
#endif

/* 2nd launch bound = waves per SIMD: six 256-thread workgroups per CU for the common 2304-point capacity (<= 80 VGPRs) */
template <int ROWS>
__global__ __launch_bounds__(LL_BLOCK, (ROWS <= 9 ? 6 : 1)) void k_ring_features(LLView V, int first, int count)
{
    int sl, r;
    if (!ll_xcd_map2(blockIdx.x, V.R, count, sl, r)) return;
    const int s = first + sl;
    const int tid = threadIdx.x, lane = tid & 63;
    const ScanHdr h = V.hdr[s];
    int *fcnt = V.ring_feat_cnt + ((size_t)s * V.R + r) * 4;
    if (h.status != 0) { if (tid < 4) fcnt[tid] = 0; return; }
    const int N = h.n;
    const int off = V.ring_off[(size_t)s * (V.R + 1) + r];
    const int nr = V.ring_off[(size_t)s * (V.R + 1) + r + 1] - off;
    if (nr <= 0) { if (tid < 4) fcnt[tid] = 0; return; }
    const int S = off + 5, E = off + nr - 6;                          /* scanStartInd / scanEndInd (:218-220) */
    const bool active = (E - S >= 6);                                 /* :248 */
    const int Lseg = active ? (E - S) : 0;                            /* indices S .. E-1 are in segments */
    const float4 *cloud = V.cloud + (size_t)s * V.NP;
    FeatLds L = ll_carve(ll_smem, V.max_ring);
    int *segb = L.sc + 16;                                            /* segment bounds (slots) for sort + pick */
    float *fs = (float *)(L.sc + 32);                                 /* 24 floats: per-wave bounds */

    const int nwords = (nr + 31) / 32 + 1;
    for (int i = tid; i < nwords; i += LL_BLOCK) { L.picked[i] = 0; L.gapf[i] = 0; }
    for (int i = tid; i < nr; i += LL_BLOCK) L.lab[i] = 0;
    if (tid < 3) L.lists[156 + tid] = 0;                              /* n_sharp, n_lsharp, n_flat */
    if (tid <= LL_SEGS) segb[tid] = Lseg * tid / LL_SEGS;             /* sp_j - S, int math of :253-254 */
    __syncthreads();

    LL_PHASE_BEGIN();
    /* ---------------- phase 1: curvature + gap flags + sort records ---------------- */
    for (int c0 = 0; c0 < nr; c0 += LL_FTILE) {
        const int g0 = off + c0;                                      /* global index of tile slot 5 */
        for (int t = tid; t < LL_FTILE + 10; t += LL_BLOCK) {
            const int g = g0 - 5 + t;
            if (g >= 0 && g < N) { const float4 p = cloud[g]; L.tx[t] = p.x; L.ty[t] = p.y; L.tz[t] = p.z; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LL_FTILE / LL_BLOCK; ++k) {
            const int li = c0 + k * LL_BLOCK + tid;
            if (li >= nr) continue;
            const int g = off + li, t = li - c0 + 5;
            if (g >= 1) {                                             /* gap to the previous point (:290-293) */
                const float dx = L.tx[t] - L.tx[t - 1], dy = L.ty[t] - L.ty[t - 1], dz = L.tz[t] - L.tz[t - 1];
                if ((double)(dx * dx + dy * dy + dz * dz) > V.gap_thr) atomicOr(&L.gapf[li >> 5], 1u << (li & 31));
            }
            if (g >= 5 && g < N - 5) {                                /* :225-235, strict left-to-right */
                const float *X = L.tx + t, *Y = L.ty + t, *Z = L.tz + t;
                const float dX = X[-5] + X[-4] + X[-3] + X[-2] + X[-1] - 10 * X[0] + X[1] + X[2] + X[3] + X[4] + X[5];
                const float dY = Y[-5] + Y[-4] + Y[-3] + Y[-2] + Y[-1] - 10 * Y[0] + Y[1] + Y[2] + Y[3] + Y[4] + Y[5];
                const float dZ = Z[-5] + Z[-4] + Z[-3] + Z[-2] + Z[-1] - 10 * Z[0] + Z[1] + Z[2] + Z[3] + Z[4] + Z[5];
                const float cv = dX * dX + dY * dY + dZ * dZ;
                if (V.write_curv) V.curv[(size_t)s * V.NP + g] = cv;
                if (active && g >= S && g < E) { L.k32[g - S] = ll_f2u(cv); L.k16[g - S] = (unsigned short)li; }
            }
        }
        __syncthreads();
    }

    LL_PHASE(0);
    /* ---------------- phase 2: sort the six segments (:251-257) ---------------- */
    if (active) ll_radix_sort<ROWS, true>(L.k32, L.k16, Lseg, L.cnt, L.sc, segb, tid);
    __syncthreads();

    LL_PHASE(1);
    /* ---------------- phase 3: greedy pick, wave 0 ---------------- */
    if (active && tid < 64) {
        /* the pick is one long dependent instruction chain on a single wave while the other waves of the workgroup
         * wait at the barrier: let it win issue arbitration against the co-resident workgroups' bulk phases */
        __builtin_amdgcn_s_setprio(3);
        int ns = 0, nls = 0, nf = 0;
        for (int j = 0; j < LL_SEGS; ++j) {
            const int sp = Lseg * j / 6, ep = Lseg * (j + 1) / 6 - 1;     /* record slots; = (:253-254) - S */
            const int len = ep - sp + 1;
            /* pass 0: corners, descending curvature (:261-313); pass 1: flats, ascending (:316-359) */
            for (int pass = 0; pass < 2; ++pass) {
                int npick = 0; bool done = false;
                for (int c0 = 0; c0 < len && !done; c0 += 64) {
                    const bool have = c0 + lane < len;
                    const int slot = pass == 0 ? ep - (c0 + lane) : sp + c0 + lane;
                    const int li = have ? (int)L.k16[slot] : 0;
                    const double cv = have ? (double)ll_u2f(L.k32[slot]) : 0.0;
                    const bool cand = have && (pass == 0 ? cv > V.curv_thr : cv < V.curv_thr);
                    if (__ballot(cand) == 0ull) break;
                    /* per candidate, once per chunk: is it already suppressed, and which index range [lo, hi] would its
                     * own pick suppress (:288-311): bits gapf[li-4 .. li+5] straight from the LDS bitmap */
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          /* marks of earlier chunks / segments */
                    bool elig = false; int lo = 0, hi = 0;
                    if (cand) {
                        elig = !ll_bit(L.picked, li);
                        const int b0 = li - 4;
                        const unsigned long long w = ((unsigned long long)L.gapf[(b0 >> 5) + 1] << 32) | L.gapf[b0 >> 5];
                        const unsigned bits = (unsigned)(w >> (b0 & 31)) & 0x3ffu;   /* bit t = gapf[li - 4 + t] */
                        const unsigned fwd = bits >> 5;                              /* l = 1..5  -> gapf[li + l] */
                        const int fn = fwd ? (__ffs(fwd) - 1) : 5;
                        int bn = 5;                                                  /* l = -1..-5 -> gapf[li + l + 1] */
#pragma unroll
                        for (int mm = 4; mm >= 0; --mm) if ((bits >> (4 - mm)) & 1u) bn = mm;
                        lo = li - bn; hi = li + fn;
                    }
                    /* the serial part only decides: which lane is picked next and which lanes that suppresses.  Labels,
                     * list entries and the cloudNeighborPicked marks are written by the picked lanes themselves after the
                     * chunk (nothing inside the chunk reads them: in-chunk suppression is the register compare). */
                    /* picks go in ascending lane order (= sorted order), so the set of picked lanes is all the epilogue needs */
                    unsigned long long em = __ballot(elig), pm = 0ull;                   /* uniform: eligible / picked lanes */
                    const int base = npick;
                    while (em) {
                        const int f = __ffsll((long long)em) - 1;
                        npick++;
                        if (pass == 0 && npick > LL_LSHARP_PER_SEG) { done = true; break; }     /* :281-284 */
                        pm |= 1ull << f;
                        if (pass == 1 && npick >= LL_FLAT_PER_SEG) { done = true; break; }      /* :328-331: before marking */
                        const int slo = __builtin_amdgcn_readlane(lo, f), shi = __builtin_amdgcn_readlane(hi, f);
                        em &= ~__ballot(li >= slo && li <= shi);
                    }
                    const int myord = ((pm >> lane) & 1ull) ? base + __popcll(pm & ((1ull << lane) - 1ull)) + 1 : 0;
                    if (myord) {
                        if (pass == 0) {
                            if (myord <= LL_SHARP_PER_SEG) { L.lab[li] = 2; L.lists[ns + myord - 1] = li; }
                            else L.lab[li] = 1;
                            L.lists[12 + nls + myord - 1] = li;
                        } else { L.lab[li] = -1; L.lists[132 + nf + myord - 1] = li; }
                        if (pass == 0 || myord < LL_FLAT_PER_SEG) {                      /* cloudNeighborPicked[lo..hi] = 1 */
                            const unsigned long long bits = ((1ull << (hi - lo + 1)) - 1ull) << (lo & 31);
                            atomicOr(&L.picked[lo >> 5], (unsigned)bits);
                            if (bits >> 32) atomicOr(&L.picked[(lo >> 5) + 1], (unsigned)(bits >> 32));
                        }
                    }
                    if (__ballot(have && !cand) != 0ull) break;                         /* the rest is beyond the threshold */
                }
                if (pass == 0) { ns += min(npick, LL_SHARP_PER_SEG); nls += min(npick, LL_LSHARP_PER_SEG); }
                else nf += npick;
            }
        }
        if (lane == 0) { L.lists[156] = ns; L.lists[157] = nls; L.lists[158] = nf; }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();

    LL_PHASE(2);
    /* ---------------- phase 4: less-flat compaction + VoxelGrid (:361-376) ----------------
     * Blocked layout: thread t owns the `per` consecutive slots [t*per, (t+1)*per), so a workgroup exclusive scan of the
     * per-thread less-flat counts compacts in input order.  The points are read once (independent loads) and stay in
     * registers for the bounding box and for PCL's voxel index; after the stable sort by voxel the same blocked layout
     * prefetches every thread's points before the left-to-right f32 centroid sums. */
    int n_lf_out = 0;
    if (active) {
        const int per = (Lseg + LL_BLOCK - 1) / LL_BLOCK;                        /* <= ROWS */
        const int a0 = min(Lseg, tid * per), a1 = min(Lseg, a0 + per);          /* slots -> local index slot + 5 */
        float px[ROWS], py[ROWS], pz[ROWS];
        unsigned lfm = 0;                                                        /* bit u: slot a0 + u is less-flat */
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            const int q = a0 + u;
            if (q < a1) {
                const float4 p = cloud[off + q + 5];
                px[u] = p.x; py[u] = p.y; pz[u] = p.z;
                if (L.lab[q + 5] <= 0) lfm |= 1u << u;
            }
        }
        float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY, mxz = -INFINITY;
#pragma unroll
        for (int u = 0; u < ROWS; ++u)
            if ((lfm >> u) & 1u) {
                mnx = fminf(mnx, px[u]); mny = fminf(mny, py[u]); mnz = fminf(mnz, pz[u]);
                mxx = fmaxf(mxx, px[u]); mxy = fmaxf(mxy, py[u]); mxz = fmaxf(mxz, pz[u]);
            }
        /* wave reduce min/max, then across the 4 waves through LDS */
        for (int o = 32; o > 0; o >>= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, o)); mny = fminf(mny, __shfl_xor(mny, o)); mnz = fminf(mnz, __shfl_xor(mnz, o));
            mxx = fmaxf(mxx, __shfl_xor(mxx, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o));
        }
        if (lane == 0) { float *w = fs + (tid >> 6) * 6; w[0] = mnx; w[1] = mny; w[2] = mnz; w[3] = mxx; w[4] = mxy; w[5] = mxz; }
        int m = 0;
        int pos = ll_block_exscan(__popc(lfm), L.sc, m);             /* barriers inside also publish fs[] */
        float mn[3] = {fs[0], fs[1], fs[2]}, mx[3] = {fs[3], fs[4], fs[5]};
        for (int w = 1; w < LL_BLOCK / 64; ++w)
            for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], fs[w * 6 + c]); mx[c] = fmaxf(mx[c], fs[w * 6 + 3 + c]); }
        if (m > 0) {
            /* pcl::VoxelGrid::applyFilter (PCL 1.10), restated */
            const float inv = V.inv_leaf;
            long long d[3]; int min_b[3], div_b[3];
            for (int c = 0; c < 3; ++c) {
                d[c] = (long long)((mx[c] - mn[c]) * inv) + 1;
                min_b[c] = (int)floorf(mn[c] * inv);
                div_b[c] = (int)floorf(mx[c] * inv) - min_b[c] + 1;
            }
            const bool too_small = d[0] * d[1] * d[2] > (long long)INT_MAX;        /* "leaf size too small": output = input */
            const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
            const float fb0 = (float)min_b[0], fb1 = (float)min_b[1], fb2 = (float)min_b[2];
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                if ((lfm >> u) & 1u) {
                    unsigned idx;
                    if (too_small) idx = (unsigned)pos;
                    else {
                        const int i0 = (int)(floorf(px[u] * inv) - fb0);
                        const int i1 = (int)(floorf(py[u] * inv) - fb1);
                        const int i2 = (int)(floorf(pz[u] * inv) - fb2);
                        idx = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
                    }
                    L.k32[pos] = idx; L.k16[pos] = (unsigned short)(a0 + u + 5);       /* payload: local index */
                    ++pos;
                }
            __syncthreads();
            LL_PHASE(3);
            ll_radix_sort<ROWS, false>(L.k32, L.k16, m, L.cnt, L.sc, segb, tid);
            LL_PHASE(4);
            __syncthreads();
            /* voxel runs -> centroids.  Thread t owns sorted positions [t*perm, (t+1)*perm): its points are fetched up
             * front, a run is summed by the thread that owns its head and may continue into the following threads' range */
            const int perm = (m + LL_BLOCK - 1) / LL_BLOCK;                          /* <= ROWS */
            const int b0 = min(m, tid * perm), b1 = min(m, b0 + perm);
            float4 pt[ROWS]; unsigned vk[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                if (b0 + u < b1) { vk[u] = L.k32[b0 + u]; pt[u] = cloud[off + L.k16[b0 + u]]; }
            unsigned headm = 0;
            {
                unsigned prev = 0; bool has_prev = false;
                if (b0 > 0 && b0 < b1) { prev = L.k32[b0 - 1]; has_prev = true; }
#pragma unroll
                for (int u = 0; u < ROWS; ++u)
                    if (b0 + u < b1) {
                        if (!has_prev || vk[u] != prev) headm |= 1u << u;
                        prev = vk[u]; has_prev = true;
                    }
            }
            int o = ll_block_exscan(__popc(headm), L.sc, n_lf_out);
            float4 *out = V.lflat_slot + (size_t)s * V.NP + off;
            /* CentroidPoint<PointXYZI>: f32 sums from zero in input order, divided by float(n) */
            float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f; int cn = 0;
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                if (b0 + u < b1) {
                    if ((headm >> u) & 1u) {
                        if (cn) { const float fn = (float)cn; out[o++] = make_float4(sx / fn, sy / fn, sz / fn, si / fn); }
                        sx = 0.0f; sy = 0.0f; sz = 0.0f; si = 0.0f; cn = 0;
                    } else if (!cn) continue;                                        /* tail of an earlier thread's run */
                    sx += pt[u].x; sy += pt[u].y; sz += pt[u].z; si += pt[u].w; ++cn;
                }
            if (cn) {
                const unsigned vid = L.k32[b1 - 1];
                for (int e = b1; e < m && L.k32[e] == vid; ++e) {
                    const float4 q = cloud[off + L.k16[e]];
                    sx += q.x; sy += q.y; sz += q.z; si += q.w; ++cn;
                }
                const float fn = (float)cn; out[o++] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
            }
        }
    }

    LL_PHASE(5);
    /* ---------------- phase 5: labels + feature slots ---------------- */
    int8_t *label = V.label + (size_t)s * V.NP + off;
    for (int i = tid; i < nr; i += LL_BLOCK) label[i] = L.lab[i];
    const int ns = L.lists[156], nls = L.lists[157], nf = L.lists[158];
    const size_t ring_id = (size_t)s * V.R + r;
    if (tid < ns) V.sharp_slot[ring_id * 12 + tid] = cloud[off + L.lists[tid]];
    if (tid < nls) V.lsharp_slot[ring_id * 120 + tid] = cloud[off + L.lists[12 + tid]];
    if (tid < nf) V.flat_slot[ring_id * 24 + tid] = cloud[off + L.lists[132 + tid]];
    if (tid == 0) { fcnt[0] = ns; fcnt[1] = nls; fcnt[2] = nf; fcnt[3] = n_lf_out; }
    LL_PHASE(6);
#ifdef LL_PHASE_TIMING
    if (tid == 0) atomicAdd(&V.dbg[15], 1ull);
#endif
}

/* per-ring slots -> clouds in publication order (ring, segment, pick order; :273-279, :325, :376) */
__global__ __launch_bounds__(128) void k_compact(LLView V, int first, int count)
{
    int sl, r;
    if (!ll_xcd_map2(blockIdx.x, V.R, count, sl, r)) return;
    const int s = first + sl;
    const int tid = threadIdx.x;
    __shared__ int pre[4], mine[4];
    const int *fc = V.ring_feat_cnt + (size_t)s * V.R * 4;
    if (tid < 4) {
        int run = 0;
        for (int q = 0; q < r; ++q) run += fc[q * 4 + tid];
        pre[tid] = run; mine[tid] = fc[r * 4 + tid];
    }
    __syncthreads();
    const size_t ring_id = (size_t)s * V.R + r;
    for (int i = tid; i < mine[0]; i += 128) V.sharp[(size_t)s * V.cap_sharp + pre[0] + i] = V.sharp_slot[ring_id * 12 + i];
    for (int i = tid; i < mine[1]; i += 128) V.lsharp[(size_t)s * V.cap_lsharp + pre[1] + i] = V.lsharp_slot[ring_id * 120 + i];
    for (int i = tid; i < mine[2]; i += 128) V.flat[(size_t)s * V.cap_flat + pre[2] + i] = V.flat_slot[ring_id * 24 + i];
    const int off = V.ring_off[(size_t)s * (V.R + 1) + r];
    for (int i = tid; i < mine[3]; i += 128) V.lflat[(size_t)s * V.NP + pre[3] + i] = V.lflat_slot[(size_t)s * V.NP + off + i];
    if (r == V.R - 1 && tid == 0) {
        ScanHdr *h = &V.hdr[s];
        h->n_sharp = pre[0] + mine[0]; h->n_less_sharp = pre[1] + mine[1];
        h->n_flat = pre[2] + mine[2]; h->n_less_flat = pre[3] + mine[3];
    }
}

template <int ROWS>
static void ll_launch_ring_features(const LLView &V, int first, int count, int grid, size_t lds_bytes, hipStream_t st)
{
    static size_t attr_bytes = 0;
    if (lds_bytes > attr_bytes) {
        (void)hipFuncSetAttribute((const void *)k_ring_features<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL(k_ring_features<ROWS>, dim3(grid), dim3(LL_BLOCK), lds_bytes, st, V, first, count);
}

void ll_launch_features(const LLView &V, int first, int count, size_t lds_bytes, hipStream_t st, LLProfiler *prof)
{
    const int groups = (count + 7) / 8;
    const int grid = 8 * V.R * groups;
    ll_prof_mark(prof, LL_K_RING_FEATURES, st);
    const int rows = (V.max_ring + 255) / 256;                 /* sort records per thread */
    if (rows <= 9) ll_launch_ring_features<9>(V, first, count, grid, lds_bytes, st);
    else if (rows <= 18) ll_launch_ring_features<18>(V, first, count, grid, lds_bytes, st);
    else ll_launch_ring_features<36>(V, first, count, grid, lds_bytes, st);
    ll_prof_mark(prof, LL_K_COMPACT, st);
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(128), 0, st, V, first, count);
    ll_prof_mark(prof, LL_K_END, st);
}
