/*
 * ll_associate.hip -- a5 + a6 + a7: TransformToStart, K=1 nearest neighbour, ring-window second / third point.
 * Replaces laserOdometry.cpp:77-95, :491-620 (corners), :653-793 (planes) and the kd-tree rebuild :895-896
 * of /root/reference.
 *
 * Targets of slot s are the less-sharp / less-flat clouds of slot s-1, still resident in HBM from the
 * extract stage (no publish -> subscribe -> fromROSMsg -> kd-tree build).
 *
 * k_build_grid (one workgroup per (scan, cloud)) builds what replaces the kd-tree:
 *   - a uniform 2-D (x, y) cell grid: LDS histogram over 128 x 128 cells of 1 m, exclusive scan, scatter of
 *     (x, y, z, place << 8 | ring) into cell order -- a point's PLACE is its float4 offset in the slot's cloud: its index for a
 *     contiguous cloud, ring * ring_cap + k for the ring rows of an extracted less-flat cloud (ll_common.h); places order like
 *     the reference's indices.  Points outside +-64 m saturate into the border cells, whose rectangles are treated as
 *     unbounded outwards;
 *   - for an extracted slot, first: the rings' counts -> the prefix table lf_pre (place -> index) and the header's four totals;
 *   - ring tables first_ge[v] = min{j : ring_j >= v}, last_le[v] = max{j : ring_j <= v} (ring_j = int(intensity_j))
 *     and a flag saying whether they reproduce the reference's sequential walk bounds for EVERY start index
 *     (true whenever ring ids never run more than NEARBY_SCAN ahead/behind of their position, as in any cloud
 *     the extract stage produces; arbitrary user targets may clear it).
 * k_associate, 8 lanes per query (one 128-byte line of cell-ordered points per step):
 *   K=1 search   cells visited in Chebyshev rings around the query's cell, a cell skipped only when its rectangle
 *                is provably farther than the best so far, the ring loop stopped when the whole next ring is.
 *                Distance = FLANN L2_Simple in f32 ((dx*dx + dy*dy) + dz*dz, no FMA); equal distances -> lowest
 *                place = lowest original index (traversal-dependent in the kd-tree; defined here and in the oracle).  Only
 *                neighbours closer than DISTANCE_SQ_THRESHOLD are ever used (:497 / :659): at most 7 rings.
 *   ring walks   (:504-553, :668-721) are sequential loops over a contiguous index window with early breaks and a
 *                running minimum under strict '<'.  With valid tables the window is (last_le[..], first_ge[..]),
 *                and the result is the lexicographic minimum of (distance, visiting order) over the window points
 *                of the right ring class -- found with the same pruned cell search instead of scanning ~2500
 *                points per query.  Without valid tables the wave-cooperative sequential scan below is used
 *                (break = ballot of "first lane whose ring leaves the +-2.5 window").  Both are exact.
 */
#include "ll_common.h"
#include "ll_factor_math.h"
#include <limits.h>
#include <type_traits>

/* walk bounds of a start ring rc: the up-walk stops at the first ring > rc + NEARBY_SCAN, the down-walk at the
 * first ring < rc - NEARBY_SCAN (compared in double like the reference); for integer rings that is > hi / < lo */
__device__ __forceinline__ int ll_ring_hi(int rc, double nearby) { return (int)floor((double)rc + nearby); }
__device__ __forceinline__ int ll_ring_lo(int rc, double nearby) { return (int)ceil((double)rc - nearby); }

extern __shared__ __attribute__((aligned(16))) unsigned char ll_gsm[];

#ifdef LL_PHASE_TIMING   /* tools/phase_timing.py: V.dbg[8..11] = k_build_grid phases, V.dbg[14] = its workgroups */
#define LL_GPHASE_BEGIN() long long ll_t0 = (tid == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0
#define LL_GPHASE(i) do { __syncthreads(); if (tid == 0) { const long long t1 = (long long)__builtin_amdgcn_s_memtime(); \
    atomicAdd(&V.dbg[i], (unsigned long long)(t1 - ll_t0)); ll_t0 = t1; if ((i) == 11) atomicAdd(&V.dbg[14], 1ull); } } while (0)
#else
#define LL_GPHASE_BEGIN() do {} while (0)
#define LL_GPHASE(i) do {} while (0)
#endif

/* grid + ring tables of a target cloud; carry != 0: the carry clouds, else slot first + blockIdx/2.
 *
 * The cloud is read as CHUNKS of 64 consecutive points, one chunk per wave and load: a contiguous cloud (less-sharp; the carry; an
 * uploaded less-flat cloud) is chunk c = points [64 c, 64 c + 64); the ring-strided less-flat cloud of an extracted slot
 * (ll_common.h) is, ring after ring, the ceil(ring_nlf[r] / 64) chunks of ring r's row; the chunks are dealt to the sixteen waves in
 * equal blocks, so all of them stay busy whatever the rings' lengths.  A point is named by its PLACE (float4 offset inside the slot's
 * array; = index for a contiguous cloud): place order = index order, and every consumer of the grid only compares.
 * For an extracted slot this kernel is also where the scan's totals become known: the workgroup of the less-flat cloud scans the
 * per-ring counts into lf_pre and writes the four totals of the header (no ring of k_ring_features waited for another to learn them). */
#define LL_GB 1024      /* threads: the kernel is a chain of latency-bound sweeps and LDS (68 KB) allows two workgroups per CU */
/* Workgroups per CU.  Two 1024-thread workgroups are 8 waves per SIMD, and the hardware admits the eighth wave only below 81 SGPRs
 * (MI355X_MICROARCH.md, residency: floor(800 / (ceil(sgpr / 16) * 16 + 16)) waves per SIMD; the occupancy query says 8 up to 96):
 * rounds 1-4 compiled this kernel to 82 SGPRs, so ONE workgroup per CU was resident whatever the occupancy report said.  Round 5
 * measured both on one box (profiles/r05_experiments/ab_build_grid_*.log): bound to 8 waves (78 SGPRs, two resident workgroups, 4 loads
 * in flight per lane) 9.4-9.7 ms per 16384 scans, left at one workgroup with 8 loads in flight 8.9-9.2 -- every phase of the kernel
 * takes 1.8-2.7x as long with a neighbour on the CU (instrumented build): it is bound by what a CU's LDS and memory pipes deliver,
 * not by latency a second workgroup could hide. */
#ifndef LL_GRID_WAVES
#define LL_GRID_WAVES 4
#endif
static size_t ll_grid_lds_bytes(const LLView &) { return (LL_GRID_NC + LL_GRID_NC / 16) * sizeof(int); }
__global__ __launch_bounds__(LL_GB, LL_GRID_WAVES) void k_build_grid(LLView V, int first, int count, int carry)
{
    const int which = blockIdx.x & 1, sl = blockIdx.x >> 1;
    if (sl >= count) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = LL_GB / 64;
    /* cell c lives at hist[c + (c >> 4)]: the per-thread scan below walks 16 consecutive cells per lane, and the
     * one-word skew per 16 cells spreads the lanes over all LDS banks instead of putting them on two */
#define LL_HI(c) ((c) + ((c) >> 4))
    int *hist = (int *)ll_gsm;                      /* [LL_GRID_NC + LL_GRID_NC / 16] */
    __shared__ int sc[LL_GB / 64];
    __shared__ int feq[LL_TAB + 1], leq[LL_TAB + 1];
    __shared__ int okflag;
    __shared__ int rcnt[LL_MAX_RINGS], pre[LL_MAX_RINGS + 1], cpre[LL_MAX_RINGS + 1];   /* per ring: points, index of the first, chunk of the first */
    __shared__ int tot3[3];
    for (int i = tid; i < LL_GRID_NC + LL_GRID_NC / 16; i += LL_GB) hist[i] = 0;
    for (int i = tid; i <= LL_TAB; i += LL_GB) { feq[i] = INT_MAX; leq[i] = -1; }
    if (tid == 0) okflag = (V.nearby >= 0.0) ? 1 : 0;
    if (tid < 3) tot3[tid] = 0;

    const float4 *pts; int m = 0; int *gstart; float4 *gpts;
    bool extracted = false;                                /* an extracted slot: the counts are per ring, the totals not yet known */
    int s = 0;
    /* the per-ring counts are requested together with the header, before it is known whether they will be wanted (an extracted slot):
     * one round trip at the head of the workgroup instead of two */
    unsigned pc_spec = 0u; int nlf_spec = 0;
    if (!carry && tid < V.R) { const size_t o = (size_t)(first + sl) * V.R + tid; pc_spec = V.ring_cnt[o]; nlf_spec = V.ring_nlf[o]; }
    if (carry) {
        pts = which ? V.carry_surf : V.carry_corner; m = V.carry_cnt[which];
        gstart = V.carry_gstart + (size_t)which * LL_GSTRIDE;
        gpts = which ? V.carry_gpts_s : V.carry_gpts_c;
    } else {
        s = first + sl;
        const ScanHdr h = V.hdr[s];
        pts = which ? V.lflat + (size_t)s * V.LFS : V.lsharp + (size_t)s * V.cap_lsharp;
        gstart = V.gstart + ((size_t)s * 2 + which) * LL_GSTRIDE;
        gpts = which ? V.gpts_s + (size_t)s * V.NP : V.gpts_c + (size_t)s * V.cap_lsharp;
        if (h.status != 0) m = 0;
        else if (!h.lf_strided) m = which ? h.n_less_flat : h.n_less_sharp;         /* ll_upload_features: the caller's counts */
        else extracted = true;
    }
    __syncthreads();
    LL_GPHASE_BEGIN();
    if (extracted) {
        /* the rings' counts: less-flat points (k_ring_features) or less-sharp picks (k_ring_pick) of every ring; the workgroup of
         * the less-flat cloud also sums the three small clouds' counts for the header */
        const int R = V.R;
        if (tid < LL_MAX_RINGS) {
            const unsigned pc = pc_spec;
            rcnt[tid] = which ? nlf_spec : (int)((pc >> 8) & 0xffu);
            if (which) {
                const int a = ll_wave_sum_i32((int)(pc & 0xffu)), b = ll_wave_sum_i32((int)((pc >> 8) & 0xffu)), c = ll_wave_sum_i32((int)((pc >> 16) & 0xffu));
                if (lane == 0) { atomicAdd(&tot3[0], a); atomicAdd(&tot3[1], b); atomicAdd(&tot3[2], c); }
            }
        }
        __syncthreads();
        if (tid < 64) {                                    /* wave 0: exclusive prefixes of the points and of the 64-point chunks, two halves */
            int carry_p = 0, carry_c = 0;
            for (int half = 0; half * 64 < R; ++half) {                /* rings beyond R hold nothing: their prefixes are the totals (below) */
                const int q = half * 64 + lane;
                const int v = rcnt[q], c = (v + 63) >> 6;
                const int ip = ll_wave_incl_scan(v), ic = ll_wave_incl_scan(c);
                pre[q] = carry_p + ip - v; cpre[q] = carry_c + ic - c;
                carry_p += __builtin_amdgcn_readlane(ip, 63); carry_c += __builtin_amdgcn_readlane(ic, 63);
            }
            for (int q = (R + 63) / 64 * 64 + lane; q <= LL_MAX_RINGS; q += 64) { pre[q] = carry_p; cpre[q] = carry_c; }   /* incl. [LL_MAX_RINGS]: the totals */
        }
        __syncthreads();
        m = pre[LL_MAX_RINGS];
        if (which) {
            /* rings beyond R hold no point: pre[R] = the total.  lf_pre: what turns a place into the reference's index (C ABI, carry copy) */
            if (tid <= R) V.lf_pre[(size_t)s * (R + 1) + tid] = pre[tid];
            if (tid == 0) { ScanHdr *hh = &V.hdr[s]; hh->n_sharp = tot3[0]; hh->n_less_sharp = tot3[1]; hh->n_flat = tot3[2]; hh->n_less_flat = m; }
        }
        __syncthreads();
    }
    const bool strided = extracted && which != 0;
    const int stride = V.ring_cap;
    const int pend = strided ? V.R * stride : m;           /* one past the last place */
    const int nch = strided ? cpre[LL_MAX_RINGS] : (m + 63) >> 6;
    /* Every wave takes a BLOCK of consecutive chunks and walks it with a cursor (ring q, chunk k of that ring, the ring's length): the
     * next chunk is k + 1 or the first chunk of the next ring that holds a point -- scalar arithmetic, one LDS read per ring crossed,
     * nothing per chunk (a per-chunk lookup table cost the sweeps 15 % of their time). */
    const int per_wave = (nch + NW - 1) / NW;
    const int cw0 = min(nch, wave * per_wave), cw1 = min(nch, cw0 + per_wave);
    struct Cursor { int q, k, len, next_len; };     /* next_len: the following ring's length, read ahead (a VGPR holding a uniform value:
                                                      * the LDS read is waited for at the next crossing, not where it is issued) */
    auto ring_len = [&](int q) -> int { return q < LL_MAX_RINGS ? rcnt[q] : 1; };    /* beyond the table: a stopper for the skip loop */
    auto cursor_at = [&](int c) __attribute__((always_inline)) -> Cursor {       /* chunk c of the cloud (c < nch) */
        Cursor cu; cu.q = 0; cu.k = c; cu.len = m; cu.next_len = 0;
        if (strided) {
            int lo = 0, hi = V.R;                                                 /* the last ring whose first chunk is <= c (it holds a chunk: cpre[q + 1] > c) */
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (__builtin_amdgcn_readfirstlane(cpre[mid]) <= c) lo = mid; else hi = mid; }
            cu.q = lo; cu.k = c - __builtin_amdgcn_readfirstlane(cpre[lo]); cu.len = __builtin_amdgcn_readfirstlane(rcnt[lo]);
            cu.next_len = ring_len(lo + 1);
        }
        return cu;
    };
    /* the cursor's chunk -> its first place and the points it holds; then on to the next chunk */
    auto take = [&](Cursor &cu, int &place0, int &nv) __attribute__((always_inline)) {
        if (strided) {
            place0 = cu.q * stride + cu.k * 64; nv = cu.len - cu.k * 64;
            if (++cu.k * 64 >= cu.len) {
                cu.k = 0;
                do { ++cu.q; cu.len = __builtin_amdgcn_readfirstlane(cu.next_len); cu.next_len = ring_len(cu.q + 1); } while (cu.len == 0);
            }
        } else { place0 = cu.k * 64; nv = m - cu.k * 64; ++cu.k; }
    };
#ifndef LL_GRID_UN
#define LL_GRID_UN 8
#endif
    constexpr int UN = LL_GRID_UN;                   /* independent loads in flight per lane */
/* One UNCONDITIONAL wait behind a batch of predicated loads.  Left to itself the compiler waits inside each `if (lane < nv)` block that
 * uses a point; on the path around the block nothing was waited for, so at the join the destination registers still count as "in
 * flight", and the next batch's address arithmetic -- which reuses them -- gets an s_waitcnt vmcnt(0) in front of EVERY load: one load
 * in flight instead of UN (the sweeps ran 2.2x slower than the flat loops they replaced until this line went in). */
#define LL_LOADS_LANDED() __builtin_amdgcn_s_waitcnt(0x0f70)                              /* vmcnt(0) */
#define LL_WSHR1(x) __builtin_amdgcn_update_dpp(0, (int)(x), 0x138, 0xf, 0xf, false)      /* wave_shr:1: lane i <- lane i - 1 */
#define LL_WSHL1(x) __builtin_amdgcn_update_dpp(0, (int)(x), 0x130, 0xf, 0xf, false)      /* wave_shl:1: lane i <- lane i + 1 */
    bool bad = false;
    /* histogram + ring tables from a chunk's keys (cell | ring value << 16) */
    auto count_chunk = [&](unsigned kv, int place0, int nv) __attribute__((always_inline)) {
        const bool in = lane < nv;
        if (in) atomicAdd(&hist[LL_HI((int)(kv & 0xFFFFu))], 1);
        /* ring tables, same sweep: first / last place of every ring value (only run boundaries touch LDS) */
        const int r = in ? (int)(kv >> 16) : 0;
        const bool oob = in && r >= LL_TAB;                                   /* 0xFF: int(intensity) outside [0, LL_TAB) */
        if (oob) bad = true;
        const int rprev = LL_WSHR1(r), rnext = LL_WSHL1(r);
        if (in && !oob) {
            const int place = place0 + lane;
            if (lane == 0 || r != rprev) atomicMin(&feq[r], place);
            if (lane == 63 || lane == nv - 1 || r != rnext) atomicMax(&leq[r], place);
        }
    };
    if (cw0 < cw1) {
        Cursor cu = cursor_at(cw0);
        for (int c = cw0; c < cw1; c += UN) {
            float4 p[UN]; int pl[UN], nv[UN];
            /* the cursor first, for all UN chunks, THEN the loads back to back: with the cursor's loop between two loads the compiler
             * waits for the first (s_waitcnt vmcnt(0) at the loop's head) before it issues the second -- one load in flight, not UN */
#pragma unroll
            for (int u = 0; u < UN; ++u) { pl[u] = 0; nv[u] = 0; if (c + u < cw1) take(cu, pl[u], nv[u]); }
#pragma unroll
            for (int u = 0; u < UN; ++u) if (lane < nv[u]) p[u] = pts[pl[u] + lane];
            LL_LOADS_LANDED();
#pragma unroll
            for (int u = 0; u < UN; ++u) count_chunk(lane < nv[u] ? ll_grid_key(p[u]) : 0u, pl[u], nv[u]);
        }
    }
    if (bad) okflag = 0;
    __syncthreads();
    LL_GPHASE(8);
    constexpr int PER = LL_GRID_NC / LL_GB;         /* 16 */
    int sum = 0;
    for (int k = 0; k < PER; ++k) sum += hist[tid * (PER + 1) + k];
    int total = 0;
    int run = ll_block_exscan_n<LL_GB / 64>(sum, sc, total);
    for (int k = 0; k < PER; ++k) { const int c = hist[tid * (PER + 1) + k]; hist[tid * (PER + 1) + k] = run; gstart[tid * PER + k] = run; run += c; }
    if (tid == LL_GB - 1) gstart[LL_GRID_NC] = total;
    __syncthreads();
    LL_GPHASE(9);
    if (cw0 < cw1) {
        Cursor cu = cursor_at(cw0);
        for (int c = cw0; c < cw1; c += UN) {
            float4 p[UN]; int pl[UN], nv[UN], pos[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) { pl[u] = 0; nv[u] = 0; if (c + u < cw1) take(cu, pl[u], nv[u]); }
#pragma unroll
            for (int u = 0; u < UN; ++u) if (lane < nv[u]) p[u] = pts[pl[u] + lane];
            LL_LOADS_LANDED();
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (lane < nv[u]) { const int cc = ll_cell_coord(p[u].y) * LL_GRID_G + ll_cell_coord(p[u].x); pos[u] = atomicAdd(&hist[LL_HI(cc)], 1); }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (lane < nv[u]) {
                    const int r = (int)p[u].w;                                   /* int(intensity): the walk's scan id */
                    gpts[pos[u]] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float(((pl[u] + lane) << 8) | (r & 0xFF)));   /* place < 2^24 */
                }
        }
    }
#undef LL_HI
#undef LL_LOADS_LANDED
#undef LL_WSHR1
#undef LL_WSHL1

    LL_GPHASE(10);
    /* ---- ring tables + validity.  The tables first_ge / last_le (in places) reproduce the reference's sequential walk bounds for
     * EVERY start point iff no point lies before a point whose up-window it exceeds and none after a point whose
     * down-window it undercuts:
     *   exists j < c with ring_j > hi(ring_c)   <=>   exists value b:  first_ge[hi(b) + 1] < last_eq[b]
     *   exists j > c with ring_j < lo(ring_c)   <=>   exists value b:  last_le[lo(b) - 1]  > first_eq[b]
     * so validity is a check on the two 160-entry tables, no per-point scan. ---- */
    LL_GPHASE(11);
    __syncthreads();
    int *tab = gstart + LL_GRID_NC + 1;              /* first_ge[LL_TAB+1], last_le[LL_TAB+1], ok flag, m, one past the last place */
    /* first_ge = suffix minimum of the first places, last_le = prefix maximum of the last ones: wave 0, three
     * 64-entry chunks with shuffle scans, the running value carried between chunks */
    __shared__ int fge[LL_TAB + 1], lle[LL_TAB + 1];
    if (tid < 64) {
        constexpr int NCH = (LL_TAB + 1 + 63) / 64;
        int carry_min = pend;
        for (int ch = NCH - 1; ch >= 0; --ch) {
            const int v = ch * 64 + lane;
            int x = (v <= LL_TAB && feq[v] != INT_MAX) ? feq[v] : pend;
            x = min(x, pend);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(x, o); if (lane + o < 64) x = min(x, t); }
            x = min(x, carry_min);
            if (v <= LL_TAB) { fge[v] = x; tab[v] = x; }
            carry_min = __shfl(x, 0);
        }
        int carry_max = -1;
        for (int ch = 0; ch < NCH; ++ch) {
            const int v = ch * 64 + lane;
            int x = (v <= LL_TAB) ? leq[v] : -1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(x, o); if (lane >= o) x = max(x, t); }
            x = max(x, carry_max);
            if (v <= LL_TAB) { lle[v] = x; tab[LL_TAB + 1 + v] = x; }
            carry_max = __shfl(x, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        bool viol = false, nonmono = false;
        for (int ch = 0; ch < NCH; ++ch) {
            const int bv = ch * 64 + lane;
            if (bv < LL_TAB && feq[bv] != INT_MAX) {
                const int hi = ll_ring_hi(bv, V.nearby), lo = ll_ring_lo(bv, V.nearby);
                const int fg = (hi + 1 > LL_TAB) ? pend : fge[max(hi + 1, 0)];
                const int ll = (lo - 1 < 0) ? -1 : lle[min(lo - 1, LL_TAB)];
                if (fg < leq[bv] || ll > feq[bv]) viol = true;
                if (fge[bv + 1] < leq[bv]) nonmono = true;           /* a point of a higher ring in front of a point of ring bv */
            }
        }
        const bool any_viol = __ballot(viol) != 0ull, any_nonmono = __ballot(nonmono) != 0ull;
        if (lane == 0) {
            /* bit 0: the tables bound the walks; bit 1: the ring values never decrease along the cloud (every cloud the extract stage
             * produces) -- then "same scan line" is rj == rc and the place window is the ring window, which is what lets k_associate
             * keep per-ring minima while it looks for the nearest neighbour */
            tab[2 * (LL_TAB + 1)] = ((okflag && !any_viol) ? 1 : 0) | ((okflag && !any_viol && !any_nonmono) ? 2 : 0);
            tab[2 * (LL_TAB + 1) + 1] = m;
            tab[2 * (LL_TAB + 1) + 2] = pend;
        }
    }
}

struct TargetRef { const float4 *pts; int m; const int *gstart; const float4 *gpts; const int *tab;
                   const int *pre; int stride; };   /* pre != null: pts is ring-strided (ll_common.h), index = pre[ring] + place - ring * stride */

__device__ __forceinline__ TargetRef ll_target(const LLView &V, int s, int which)
{
    TargetRef T;
    if (s == V.carry_slot) {
        T.pts = which ? V.carry_surf : V.carry_corner; T.m = V.carry_cnt[which];
        T.gstart = V.carry_gstart + (size_t)which * LL_GSTRIDE;
        T.gpts = which ? V.carry_gpts_s : V.carry_gpts_c;
        T.pre = nullptr; T.stride = 0;
    } else {
        const int t = s - 1;
        const ScanHdr h = V.hdr[t];
        T.pts = which ? V.lflat + (size_t)t * V.LFS : V.lsharp + (size_t)t * V.cap_lsharp;
        T.m = (h.status != 0) ? 0 : (which ? h.n_less_flat : h.n_less_sharp);
        const bool strided = which != 0 && h.lf_strided != 0;
        T.pre = strided ? V.lf_pre + (size_t)t * (V.R + 1) : nullptr; T.stride = strided ? V.ring_cap : 0;
        T.gstart = V.gstart + ((size_t)t * 2 + which) * LL_GSTRIDE;
        T.gpts = which ? V.gpts_s + (size_t)t * V.NP : V.gpts_c + (size_t)t * V.cap_lsharp;
    }
    T.tab = T.gstart + LL_GRID_NC + 1;
    return T;
}

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

__device__ __forceinline__ void ll_best_take(Best &b, float d, int ord, int j, float dmax)
{
    /* sequential semantics "if (d < min) take" over the visiting order == lexicographic min of (d, ord), d < dmax */
    if (d < dmax && (d < b.d || (d == b.d && ord < b.ord))) { b.d = d; b.ord = ord; b.j = j; }
}

template <int WIDTH>
__device__ __forceinline__ void ll_best_reduce(Best &b)
{
    for (int o = WIDTH / 2; o > 0; o >>= 1) {
        const float d2 = __shfl_xor(b.d, o); const int o2 = __shfl_xor(b.ord, o); const int j2 = __shfl_xor(b.j, o);
        if (d2 < b.d || (d2 == b.d && o2 < b.ord)) { b.d = d2; b.ord = o2; b.j = j2; }
    }
}

__device__ __forceinline__ float ll_walk_d2(const float4 p, float sx, float sy, float sz)
{
    /* (p.x - sel.x)*(p.x - sel.x) + (p.y - sel.y)*(...) + (p.z - sel.z)*(...)  f32 (:514-519) */
    return (p.x - sx) * (p.x - sx) + (p.y - sy) * (p.y - sy) + (p.z - sz) * (p.z - sz);
}

/* e-th entry (0 <= e < 4*ring, or the centre cell for ring 0) of the Chebyshev ring `ring`: the top row, the bottom row
 * (2*ring + 1 cells each, CONTIGUOUS in the cell-ordered point array, so a row is one range), then the two side cells of
 * every inner row.  Returns the cell range [x0, x1] of row yy. */
__device__ __forceinline__ void ll_ring_entry(int ring, int e, int cx, int cy, int &x0, int &x1, int &yy)
{
    if (ring == 0) { x0 = cx; x1 = cx; yy = cy; return; }
    if (e < 2) { x0 = cx - ring; x1 = cx + ring; yy = e ? cy + ring : cy - ring; return; }
    const int t = e - 2;
    yy = cy - ring + 1 + (t >> 1);
    x0 = x1 = (t & 1) ? cx + ring : cx - ring;
}

#ifndef LL_SCAN_UN
#define LL_SCAN_UN 2              /* point loads a lane keeps in flight while scanning a cell */
#endif
#define LL_RING_CELLS 18          /* entries (rows / side cells) whose bounds are fetched per round -- the sweep beyond ring 2 at the default radius is 18 row
                                   * ranges; wider sweeps take several rounds */

/* squared distance from (qx, qy) to the rectangle of cells [xa, xb] of row yy, shrunk by a 1 mm margin so that float rounding
 * in ll_cell_coord can never make it an over-estimate; border cells are unbounded outwards */
__device__ __forceinline__ float ll_range_lb2(float qx, float qy, int xa, int xb, int yy)
{
    const float lox = (float)xa * LL_GRID_CELL - LL_GRID_ORG, hix = (float)(xb + 1) * LL_GRID_CELL - LL_GRID_ORG;
    const float loy = (float)yy * LL_GRID_CELL - LL_GRID_ORG;
    float dx = 0.0f, dy = 0.0f;
    if (qx < lox && xa > 0) dx = lox - qx;
    else if (qx > hix && xb < LL_GRID_G - 1) dx = qx - hix;
    if (qy < loy && yy > 0) dy = loy - qy;
    else if (qy > loy + LL_GRID_CELL && yy < LL_GRID_G - 1) dy = qy - (loy + LL_GRID_CELL);
    dx = fmaxf(dx - 1e-3f, 0.0f); dy = fmaxf(dy - 1e-3f, 0.0f);
    return dx * dx + dy * dy;
}

/* Entry e of what lies between Chebyshev ring LL_NEAR_RINGS and ring rmax, as row ranges ordered from near to far: the rows
 * that cross the inner block contribute a left and a right part (e < 2 * (2 * LL_NEAR_RINGS + 1)), every other row is one
 * range of 2 * rmax + 1 cells.  2 * rmax + 2 * LL_NEAR_RINGS + 2 entries instead of the 4 * ring entries of every ring: a
 * query whose partner is far away (or missing: ground points far out have no neighbour on the next ring within the
 * 5 m limit) walks 18 ranges, not 72 entries. */
#define LL_NEAR_RINGS 2
__device__ __forceinline__ void ll_annulus_entry(int rmax, int e, int cx, int cy, int &x0, int &x1, int &yy)
{
    constexpr int NP = 2 * (2 * LL_NEAR_RINGS + 1);
    if (e < NP) {
        const int r = e >> 1, k = (r + 1) >> 1;                       /* rows 0, -1, +1, -2, +2 */
        yy = cy + ((r & 1) ? -k : k);
        if (e & 1) { x0 = cx + LL_NEAR_RINGS + 1; x1 = cx + rmax; } else { x0 = cx - rmax; x1 = cx - LL_NEAR_RINGS - 1; }
        return;
    }
    const int t = e - NP, k = LL_NEAR_RINGS + 1 + (t >> 1);
    yy = cy + ((t & 1) ? k : -k);
    x0 = cx - rmax; x1 = cx + rmax;
}

/* visit the cells around (qx, qy) from near to far; scan(start, end) scans one contiguous range of the cell-ordered points,
 * bound() is the current pruning radius^2 (shrinks as candidates are found), sync() shares the best inside the 8-lane group.
 * Chebyshev rings 0 .. LL_NEAR_RINGS one by one (most searches end there), then the rest as row ranges in one sweep.
 * The bounds of ALL entries of a round are fetched at once -- lane `sub` of the group takes entries sub, sub+8, sub+16 -- and
 * exchanged through a per-group LDS table: one parallel round of loads instead of one dependent load per entry. */
template <typename Scan, typename Bound, typename Sync>
__device__ __forceinline__ void ll_grid_search(const int *gstart, float qx, float qy, int rmax, int *cellb, int sub, int ring_from,
                                               Scan scan, Bound bound, Sync sync)
{
    const int cx = ll_cell_coord(qx), cy = ll_cell_coord(qy);
    auto sweep = [&](int nent, auto entry) __attribute__((always_inline)) {
        for (int e0 = 0; e0 < nent; e0 += LL_RING_CELLS) {            /* more entries than the table holds: several rounds */
            const int ne = min(LL_RING_CELLS, nent - e0);
            const float bnd = bound();
            for (int e = sub; e < ne; e += 8) {
                int x0, x1, yy; entry(e0 + e, x0, x1, yy);
                const int xa = max(x0, 0), xb = min(x1, LL_GRID_G - 1);
                int st = 0, en = 0; float lb = 0.0f;
                if (xa <= xb && yy >= 0 && yy < LL_GRID_G) {
                    lb = ll_range_lb2(qx, qy, xa, xb, yy);
                    if (!(lb > bnd)) { st = gstart[yy * LL_GRID_G + xa]; en = gstart[yy * LL_GRID_G + xb + 1]; }
                }
                cellb[3 * e] = st; cellb[3 * e + 1] = en; cellb[3 * e + 2] = __float_as_int(lb);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (int e = 0; e < ne; ++e) {
                const int st = cellb[3 * e], en = cellb[3 * e + 1];
                if (st >= en) continue;
                if (__int_as_float(cellb[3 * e + 2]) > bound()) continue;   /* the bound may have shrunk since the fetch */
                scan(st, en);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            sync();
        }
    };
    const int near = min(rmax, LL_NEAR_RINGS);
    for (int ring = ring_from; ring <= near; ++ring) {
        if (ring >= 2) { const float lbr = (float)(ring - 1) * LL_GRID_CELL - 1e-3f; if (lbr * lbr > bound()) return; }
        sweep(ring == 0 ? 1 : 4 * ring, [&](int e, int &x0, int &x1, int &yy) { ll_ring_entry(ring, e, cx, cy, x0, x1, yy); });
    }
    if (rmax > LL_NEAR_RINGS) {
        const float lbr = (float)LL_NEAR_RINGS * LL_GRID_CELL - 1e-3f;
        if (lbr * lbr > bound()) return;
        sweep(2 * rmax + 2 * LL_NEAR_RINGS + 2, [&](int e, int &x0, int &x1, int &yy) { ll_annulus_entry(rmax, e, cx, cy, x0, x1, yy); });
    }
}

/* ---- k_associate: ONE traversal for the nearest neighbour and its ring-window partners (round 6) ----
 * Rounds 1-5 searched twice: K=1, then -- the nearest point c and its ring rc known -- a second pruned search over the SAME near cells
 * for the second / third point (2.5 of the kernel's 6 ms re-scanned cells the first search had just read).  For a target whose ring
 * values never decrease along the cloud (bit 1 of the table flags: every extracted cloud) the partners are "the nearest other point on
 * ring rc" and "the nearest point on the rings rc +- 1, rc +- 2" (:504-553, :668-721), so the first traversal can collect them before rc
 * is known: every candidate it computes a distance for also goes, by its ring, into a small per-query table in LDS --
 *   TMIN[v] = min (distance bits << 32 | place << 8 | ring)  over the candidates of ring v EXCEPT the nearest neighbour itself
 *   TMAX[v] = max (~distance bits << 32 | place << 8 | ring) over the same candidates: the same distance with the HIGHEST place
 * -- for LL_ATAB_W ring values around the ring the query's own elevation predicts (the neighbour lies within its distance of the query,
 * so within a few rings of that).  "Except the nearest neighbour" without knowing it: a lane withholds the candidate that is its running
 * minimum and hands it to the table only when a better one replaces it (or when the group's shared minimum turns out to be another
 * lane's); the one key never handed over is the final minimum c.  Both updates are NON-returning LDS atomics: nothing in the scan loop
 * waits for them.  When c and rc are known, TMIN[rc] is the nearest OTHER point of ring rc and TMIN[rc +- 1], TMIN[rc +- 2] are those
 * rings' nearest points -- exactly the minima, over everything the first traversal visited, that the second search used to recompute;
 * they become its starting bounds, the near entries the first traversal already scanned are skipped (a bit per lane and entry), and
 * the SAME frontier goes on outward only while an unvisited rectangle can still beat them.  Exactness: equal distances inside one ring
 * are the only case in which the table's order (lowest place) and the walks' visiting order can disagree; they show as TMIN and TMAX
 * naming different places, and such a query -- like one whose window rc +- NEARBY_SCAN leaves the table, or a target that is not
 * monotone -- simply starts the second search from nothing with no entry marked visited: the search of rounds 1-5, same code. */
#ifndef LL_ASSOC_CORNER_ROWS
#define LL_ASSOC_CORNER_ROWS 1          /* corner queries take the 3 x 3 cells around them as three rows */
#endif
#ifndef LL_ASSOC_PLANE_ROWS_BELOW
#define LL_ASSOC_PLANE_ROWS_BELOW 64    /* ... plane queries too when their own cell holds fewer points than this */
#endif
#ifndef LL_ATAB_W
#define LL_ATAB_W 14              /* ring values per query table: 32 groups x 2 tables x 14 x 8 B = 7 KB of the workgroup's 20 KB */
#endif
#ifndef LL_ATAB_BACK
#define LL_ATAB_BACK (LL_ATAB_W / 2)   /* the table covers rings [predicted - 7, predicted + 6] */
#endif

template <bool PLANE>
__device__ __forceinline__ void ll_associate_block(const LLView &V, int s, int qblock, int qpb,
                                                   const float4 *queries, int nq, const TargetRef T,
                                                   int *out_a, int *out_b, int *out_c, float4 *qs, unsigned long long *rtab_all, int *cellb_all, int *tab, unsigned char *perm, int *hist)
{
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LL_TAB_WORDS; i += LL_BLOCK) tab[i] = T.tab[i];             /* the ring tables: read by every query */
    const int qi = qblock * qpb + tid;                        /* qpb queries per workgroup: 256 in a batch, 32 (one pass) when few scans must fill the chip */
    const bool have = tid < qpb && qi < nq;
    /* a5: TransformToStart (s = 1 with DISTORTION 0, the reference's build): f64 rotate + translate, f32 store */
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int pred = 0;
    if (have) {
        const double *pose = V.pose + (size_t)s * 7;
        const float4 p = queries[qi];
        const double v[3] = {(double)p.x, (double)p.y, (double)p.z};
        double rr[3];
        if (!V.distortion) {
            ll_rotate(pose, v, rr);
            sx = (float)(rr[0] + pose[4]); sy = (float)(rr[1] + pose[5]); sz = (float)(rr[2] + pose[6]);
        } else {                                              /* DISTORTION 1 (:81-88): q_point_last = Identity.slerp(s, q), t_point_last = s * t */
            const double sr = ll_point_s(1, p);
            double qs4[4];
            ll_slerp_identity(sr, pose, qs4, nullptr);
            ll_rotate(qs4, v, rr);
            sx = (float)(rr[0] + sr * pose[4]); sy = (float)(rr[1] + sr * pose[5]); sz = (float)(rr[2] + sr * pose[6]);
        }
        /* where the per-ring table is centred: the ring the query's own elevation falls into in the target's frame (the sensor model's
         * formula, scanRegistration.cpp:139-162) -- a guess that only decides how often the table is used, never a result */
        const float t_el = sz / sqrtf(sx * sx + sy * sy);
        pred = (t_el == t_el) ? ll_ring_of_t(t_el, V.ring_model, V.R, V.lower_bound, V.factor) : 0;
        pred = min(max(pred, 0), 250);
    }
    if (tid < 64) hist[tid] = 0;
    /* The eight queries of a wave advance in lockstep, so they should cost about the same: the block's queries are dealt to
     * the (pass, wave, group) slots in the order of the population of their own cell -- what both searches scan first and the
     * best predictor of their length.  Counting sort over 32 population classes; any order inside a class (no result depends
     * on it).  Slot p of the order is worked on in pass p / 32 by wave (p / 8) % 4: every wave gets every fourth octet. */
    {
        int cls = 31;                                                       /* queries beyond nq / without targets: one class, skipped together */
        int rows3 = 0;
        if (have && T.m > 0) {
            const int cell = ll_cell_coord(sy) * LL_GRID_G + ll_cell_coord(sx);
            const int cnt = T.gstart[cell + 1] - T.gstart[cell];
            cls = min(30, cnt >> 3);
            /* how the 3 x 3 cells around the query are taken (below): as three whole rows where the target is sparse around it -- every corner
             * query (the less-sharp cloud holds 0.4 points per cell), a plane query whose own cell holds few points -- else as five entries,
             * own cell first, so that a dense cell's neighbours can still be pruned by what the own cell gave */
            rows3 = (PLANE ? cnt < LL_ASSOC_PLANE_ROWS_BELOW : LL_ASSOC_CORNER_ROWS) ? 1 : 0;
        }
        qs[tid] = make_float4(sx, sy, sz, __int_as_float(have ? ((pred + 1) | (rows3 << 16)) : 0));
        __syncthreads();
        const bool dealt = tid < qpb;                                       /* one-pass workgroups order their 32 queries only */
        const int rank = dealt ? atomicAdd(&hist[cls], 1) : 0;
        __syncthreads();
        if (tid < 64) {                                                     /* exclusive scan of the 32 class counts (one wave) */
            const int v = tid < 32 ? hist[tid] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { const int t_ = __shfl_up(inc, o); if (lane >= o) inc += t_; }
            if (tid < 32) hist[32 + tid] = inc - v;
        }
        __syncthreads();
        if (dealt) perm[hist[32 + cls] + rank] = (unsigned char)tid;
        __syncthreads();
    }

    const int g = tid >> 3, sub = tid & 7;
    int *cellb = cellb_all + g * (3 * LL_RING_CELLS);
    unsigned long long *rtab = rtab_all + g * (2 * LL_ATAB_W);   /* T1 [0, W), T2 [W, 2 W) of this group's current query */
    const int rmax = (int)ceilf(sqrtf(V.nn_max) / LL_GRID_CELL) + 1;
    const float4 *gpts = T.gpts; const int *gstart = T.gstart;
    const float4 *tgt = T.pts; const int M = T.m;
    const int tflags = tab[2 * (LL_TAB + 1)];
    const bool tab_ok = (tflags & 1) != 0 && tab[2 * (LL_TAB + 1) + 1] == M;
    const bool use_table = tab_ok && (tflags & 2) != 0;          /* monotone rings: classes by ring value (workgroup-uniform) */
    const int PEND = tab[2 * (LL_TAB + 1) + 2];                  /* one past the last place of the target cloud (= M when it is contiguous) */
    const float dmax = V.nn_max;
#ifdef LL_ASSOC_STATS   /* tools/assoc_stats.py: candidates scanned by the two searches, sweep rounds, queries -> V.dbg[0..3] corners, [4..7] planes; [12], [13]: queries that did not use their table */
    unsigned long long st_nn = 0, st_w = 0, st_sync = 0, st_q = 0, st_fb = 0;
#define LL_STAT(x) (++(x))
#else
#define LL_STAT(x) do {} while (0)
#endif
    for (int pass = 0; pass < qpb / 32; ++pass) {
        const int ql = perm[((pass * (LL_BLOCK / 64) + (g >> 3)) << 3) + (g & 7)];
        const float4 q = qs[ql];
        const int qtag = __float_as_int(q.w);
        int closest = -1, res_b = -1, res_c = -1;
        if (qtag != 0 && M > 0) {
            /* ---- exact K=1 NN within nn_max ----
             * One 64-bit key per candidate: the distance's bits (non-negative floats order like their bits) above the packed
             * word place << 8 | ring, so "closer, or as close with the lower index" is one unsigned minimum, and the winner's
             * ring (closestPointScanID, :500 / :664) comes with it. */
            /* "none yet" = (dmax, 0): only a candidate with d < dmax undercuts it -- one AT the limit has the same upper word and a
             * lower word >= 0, so it does not (:497 / :659 accept d < DISTANCE_SQ_THRESHOLD only) */
            const unsigned long long knone = (unsigned long long)__float_as_uint(dmax) << 32;
            unsigned long long kb = knone;
            const unsigned rq0 = (unsigned)((qtag & 0xFFFF) - 1 - LL_ATAB_BACK);          /* first ring value of the table (may be "negative": unsigned compare below) */
            if (use_table) {
                for (int e = sub; e < 2 * LL_ATAB_W; e += 8) rtab[e] = (e < LL_ATAB_W) ? knone : 0ull;   /* TMIN [0, W), TMAX [W, 2 W) */
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            /* hand a key to the table (a candidate that is not, or no longer, this lane's running minimum) */
            auto table_put = [&](unsigned long long k) __attribute__((always_inline)) {
                const unsigned rel = ((unsigned)k & 0xFFu) - rq0;
                if (rel < (unsigned)LL_ATAB_W) {
                    (void)__hip_atomic_fetch_min(&rtab[rel], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_max(&rtab[LL_ATAB_W + rel], k ^ 0xFFFFFFFF00000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            unsigned vis = 0u;                                                  /* bit e: THIS lane scanned its share of near entry e in the first traversal */
            /* The 3 x 3 cells around the query serve BOTH searches: lane e of the group fetches entry e's bounds once and keeps them in its
             * registers, the scans get them by a lane broadcast.  Two layouts (round 6; the kernel's cost is per visited range, not per
             * candidate).  Dense surroundings -- a plane query whose own cell holds >= LL_ASSOC_PLANE_ROWS_BELOW points: FIVE entries, the own
             * cell, then the row below, the row above, the left and the right cell; own cell, share the best, the ring-1 entries the bound
             * still admits, share again.  Sparse surroundings -- every corner query (the less-sharp cloud holds 0.4 points per cell), the
             * other plane queries: THREE entries, the own ROW [cx - 1, cx + 1] (one contiguous range of the cell-ordered array), then the row
             * below and the row above: two dependent scans fewer per search, a handful of candidates more.  Only a search that is still open
             * after that (its bound reaches beyond the 3 x 3 cells) goes on ring by ring. */
            int a_st = 0, a_en = 0; float a_lb = 0.0f;
            constexpr bool CROWS = !PLANE && LL_ASSOC_CORNER_ROWS;             /* every corner query: entries 3, 4 do not exist in the code */
            const bool rows3 = CROWS || (qtag >> 16) != 0;                     /* uniform in the group: the query's own choice (set-up above) */
            if (sub < 5) {
                const int cx = ll_cell_coord(q.x), cy = ll_cell_coord(q.y);
                int x0, x1, yy;
                if (rows3) { x0 = cx - 1; x1 = sub < 3 ? cx + 1 : cx - 2; yy = sub == 0 ? cy : (sub == 1 ? cy - 1 : cy + 1); }   /* own row, row below, row above; entries 3, 4 empty */
                else ll_ring_entry(sub ? 1 : 0, sub ? sub - 1 : 0, cx, cy, x0, x1, yy);
                const int xa = max(x0, 0), xb = min(x1, LL_GRID_G - 1);
                if (xa <= xb && yy >= 0 && yy < LL_GRID_G) {
                    a_lb = ll_range_lb2(q.x, q.y, xa, xb, yy);
                    if (!(a_lb > dmax)) { a_st = gstart[yy * LL_GRID_G + xa]; a_en = gstart[yy * LL_GRID_G + xb + 1]; }
                }
            }
            /* near entries: first traversal marks what it scans, the second one skips what is marked */
            auto scan_near = [&](auto scan, auto bound, auto sync, auto first_tag) __attribute__((always_inline)) {
                constexpr bool FIRST = decltype(first_tag)::value;
                auto entry = [&](auto e_tag) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_tag)::value;
                    const int st = ll_bcast8<e>(a_st), en = ll_bcast8<e>(a_en);
                    const float lb = __int_as_float(ll_bcast8<e>(__float_as_int(a_lb)));
                    if (FIRST) { if (st < en && !(lb > bound())) { scan(st, en); vis |= 1u << e; } }
                    else if (!((vis >> e) & 1u) && st < en && !(lb > bound())) scan(st, en);
                };
                entry(std::integral_constant<int, 0>{}); sync();
                entry(std::integral_constant<int, 1>{}); entry(std::integral_constant<int, 2>{});
                if constexpr (!CROWS) {
                    /* entries 3, 4 are empty for a query that takes rows; the queries of a wave are dealt by the population of their own cell,
                     * so a wave is usually all rows or all entries: skip the two when no lane of the wave has them */
                    if (__ballot(!rows3) != 0ull) { entry(std::integral_constant<int, 3>{}); entry(std::integral_constant<int, 4>{}); }
                }
                sync();
            };
            auto nn_scan = [&](int st, int en) {
                    for (int k0 = st + sub; k0 < en; k0 += 8 * LL_SCAN_UN) {      /* LL_SCAN_UN loads in flight per lane */
                        float4 pp[LL_SCAN_UN];
#pragma unroll
                        for (int u = 0; u < LL_SCAN_UN; ++u) if (k0 + 8 * u < en) pp[u] = gpts[k0 + 8 * u];
#pragma unroll
                        for (int u = 0; u < LL_SCAN_UN; ++u) {
                            if (k0 + 8 * u >= en) break;
                            const float4 p = pp[u];
                            LL_STAT(st_nn);
                            float diff = q.x - p.x; float d = diff * diff;    /* FLANN L2_Simple: a = query, b = data */
                            diff = q.y - p.y; d += diff * diff;
                            diff = q.z - p.z; d += diff * diff;
                            const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
                            const bool better = k < kb;                       /* d < dmax is implied: kb starts at (dmax, 0) */
                            if (use_table) table_put(better ? kb : k);        /* the displaced minimum, or the candidate itself (knone lands nowhere: min with itself) */
                            kb = better ? k : kb;
                        }
                    }
                };
            auto nn_bound = [&]() { return __uint_as_float((unsigned)(kb >> 32)); };
            /* share the minimum; a lane whose own minimum lost hands it to the table */
            auto nn_sync = [&]() {
                const unsigned long long gmin = ll_min8_u64(kb);
                if (use_table && kb != gmin) table_put(kb);
                kb = gmin;
                if (sub == 0) LL_STAT(st_sync);
            };
            if (sub == 0) LL_STAT(st_q);
            scan_near(nn_scan, nn_bound, nn_sync, std::true_type{});
            ll_grid_search(gstart, q.x, q.y, rmax, cellb, sub, 2, nn_scan, nn_bound, nn_sync);
            int rc = 0;
            if (kb != knone) { closest = (int)((unsigned)kb >> 8); rc = (int)((unsigned)kb & 0xFFu); }   /* :497 / :659: d < DISTANCE_SQ_THRESHOLD */

            /* ---- second / third point inside the ring window ----
             * The walks' "if (d < min) take" over their visiting order is the lexicographic minimum of (d, visiting order):
             * again one 64-bit key, distance bits above the order; the index is recovered from the order at the end. */
            if (closest >= 0 && tab_ok) {
                const int c = closest;
                const int hi = ll_ring_hi(rc, V.nearby), lo = ll_ring_lo(rc, V.nearby);
                const int jhi = (hi + 1 > LL_TAB) ? PEND : tab[max(hi + 1, 0)];                /* first j with ring > hi */
                const int jlo = (lo - 1 < 0) ? -1 : tab[LL_TAB + 1 + min(lo - 1, LL_TAB)];     /* last j with ring < lo */
                const int c1 = c + 1, mc = PEND + c - 1;                                       /* c, j, jlo, jhi: PLACES in the target cloud (their order is the index order) */
                const unsigned long long wnone = (unsigned long long)__float_as_uint(dmax) << 32;           /* (dmax, 0): strict d < dmax as above (:512, :520, :677 ...) */
                unsigned long long k2 = wnone, k3 = wnone;
                if (use_table) {
                    /* what the first traversal left in the table.  Usable when the window [lo, hi] lies inside the table and no ring of it holds
                     * two candidates at its minimal distance; else the search below starts from nothing (vis = 0) */
                    const unsigned long long gm = 0xFFull << (lane & 56);
                    const bool inside = lo >= (int)rq0 && hi < (int)rq0 + LL_ATAB_W && hi - lo < 8;   /* rq0 may be negative: ring - rq0 wraps correctly */
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int v = lo + sub;                                                     /* lane `sub` reads ring lo + sub (the window holds <= 8 values: `inside`) */
                    unsigned long long kk = knone; bool tie = false;
                    if (inside && v <= hi) {
                        kk = rtab[v - (int)rq0];
                        const unsigned long long kx = rtab[LL_ATAB_W + v - (int)rq0];
                        tie = kk < knone && (unsigned)kk != (unsigned)kx;                     /* two places at the ring's minimal distance */
                    }
                    if (inside && (__ballot(tie) & gm) == 0ull) {
                        if (kk < knone) {
                            const int j = (int)((unsigned)kk >> 8);
                            const int ord = (j > c) ? j - c1 : mc - j;
                            const unsigned long long key = (kk & 0xFFFFFFFF00000000ull) | (unsigned)ord;
                            if (PLANE) { if (v == rc) k2 = key; else k3 = key; }
                            else if (v != rc) k2 = key;
                        }
                    } else { vis = 0u; LL_STAT(st_fb); }
                } else vis = 0u;
                auto w_scan = [&](int st, int en) {
                        for (int k0 = st + sub; k0 < en; k0 += 8 * LL_SCAN_UN) {
                          float4 pp[LL_SCAN_UN];
#pragma unroll
                          for (int u = 0; u < LL_SCAN_UN; ++u) if (k0 + 8 * u < en) pp[u] = gpts[k0 + 8 * u];
#pragma unroll
                          for (int u = 0; u < LL_SCAN_UN; ++u) {
                            if (k0 + 8 * u >= en) break;
                            const float4 p = pp[u];
                            LL_STAT(st_w);
                            const unsigned w = (unsigned)__float_as_int(p.w);
                            const int j = (int)(w >> 8), rj = (int)(w & 0xFFu);
                            const bool in = j > jlo && j < jhi && j != c;
                            const float d = ll_walk_d2(p, q.x, q.y, q.z);
                            const bool up = j > c;                                /* increasing scan line (:504-527 / :668-693), else decreasing (:530-553 / :696-721) */
                            const int ord = up ? j - c1 : mc - j;
                            const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)ord;
                            if (PLANE) {
                                const bool same = up ? rj <= rc : rj >= rc;
                                const unsigned long long ka = (in && same) ? k : wnone, kc = (in && !same) ? k : wnone;
                                k2 = (ka < k2) ? ka : k2; k3 = (kc < k3) ? kc : k3;
                            } else {
                                const bool other = up ? rj > rc : rj < rc;
                                const unsigned long long ka = (in && other) ? k : wnone;
                                k2 = (ka < k2) ? ka : k2;
                            }
                          }
                        }
                    };
                auto w_bound = [&]() { const unsigned h2 = (unsigned)(k2 >> 32), h3 = (unsigned)(k3 >> 32); return __uint_as_float(PLANE ? max(h2, h3) : h2); };
                auto w_sync = [&]() { k2 = ll_min8_u64(k2); if (PLANE) k3 = ll_min8_u64(k3); if (sub == 0) LL_STAT(st_sync); };
                auto index_of = [&](unsigned long long k) { const int ord = (int)(unsigned)k; return (k == wnone) ? -1 : (ord < PEND ? c1 + ord : mc - ord); };   /* up: ord = j - c1 < PEND, down: ord = mc - j >= PEND */
                if (use_table) w_sync();                                         /* the table's minima, shared: the bound every lane starts from */
                scan_near(w_scan, w_bound, w_sync, std::false_type{});
                ll_grid_search(gstart, q.x, q.y, rmax, cellb, sub, 2, w_scan, w_bound, w_sync);
                res_b = index_of(k2); res_c = PLANE ? index_of(k3) : -1;
            }
        }
        if (sub == 0 && ql < qpb) {
            const int qo = qblock * qpb + ql;
            if (qo < nq) {
                /* :556 / :723; a target whose tables do not bound the walks: the nearest point now, the partners from the walks below */
                const bool valid = closest >= 0 && res_b >= 0 && (!PLANE || res_c >= 0);
                out_a[qo] = (valid || !tab_ok) ? closest : -1;
                out_b[qo] = valid ? res_b : -1;
                if (PLANE) out_c[qo] = valid ? res_c : -1;
            }
        }
    }
#ifdef LL_ASSOC_STATS
    { unsigned long long *d = V.dbg + (PLANE ? 4 : 0); atomicAdd(d, st_nn); atomicAdd(d + 1, st_w); atomicAdd(d + 2, st_sync); atomicAdd(d + 3, st_q);
      atomicAdd(V.dbg + (PLANE ? 13 : 12), st_fb); }
#endif
#undef LL_STAT
    if (tab_ok) return;                                                          /* workgroup-uniform */

    /* fallback: the reference's sequential walks, one query at a time per wave.  They run over INDICES; a ring-strided target
     * (an extracted slot -- whose tables can only fail on degenerate intensities) is addressed through its prefix table:
     * place -> index by a division, index -> place by a search.  Rare by construction, exact always.  The nearest points come back
     * from the output array the pass loop parked them in (other lanes of this workgroup wrote them: L1-bypassing loads behind a fence). */
    __threadfence();
    __syncthreads();
    const int closest = have ? __hip_atomic_load(&out_a[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
    int res_b = -1, res_c = -1;
    {
        const int *tpre = T.pre; const int tstride = T.stride, RR = V.R;
        auto to_index = [&](int place) -> int { if (!tpre) return place; const int q = place / tstride; return tpre[q] + (place - q * tstride); };
        auto to_place = [&](int idx) -> int {
            if (!tpre) return idx;
            int lo = 0, hi = RR;                                       /* the last ring q with pre[q] <= idx that holds a point */
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tpre[mid] <= idx) lo = mid; else hi = mid; }
            return lo * tstride + (idx - tpre[lo]);
        };
        for (int qq = 0; qq < 64; ++qq) {
            const int cp = __shfl(closest, qq);
            if (cp < 0) continue;
            const int c = to_index(cp);
            const float qx = __shfl(sx, qq), qy = __shfl(sy, qq), qz = __shfl(sz, qq);
            const int rc = (int)tgt[cp].w;
            Best b2 = {dmax, INT_MAX, -1}, b3 = {dmax, INT_MAX, -1};
            for (int j0 = c + 1; j0 < M; j0 += 64) {
                const int j = j0 + lane;
                const bool in = j < M;
                const int jp = in ? to_place(j) : 0;
                const float4 p = in ? tgt[jp] : make_float4(0.f, 0.f, 0.f, 0.f);
                const int rj = (int)p.w;
                const bool stop = in && ((double)rj > (double)rc + V.nearby);
                const unsigned long long sm = __ballot(stop);
                const bool ok = in && (sm == 0ull || lane < __ffsll((long long)sm) - 1);
                if (ok) {
                    const float d = ll_walk_d2(p, qx, qy, qz);
                    const int ord = j - c - 1;
                    if (PLANE) { if (rj <= rc) ll_best_take(b2, d, ord, jp, dmax); else ll_best_take(b3, d, ord, jp, dmax); }
                    else if (rj > rc) ll_best_take(b2, d, ord, jp, dmax);
                }
                if (sm) break;
            }
            for (int j0 = c - 1; j0 >= 0; j0 -= 64) {
                const int j = j0 - lane;
                const bool in = j >= 0;
                const int jp = in ? to_place(j) : 0;
                const float4 p = in ? tgt[jp] : make_float4(0.f, 0.f, 0.f, 0.f);
                const int rj = (int)p.w;
                const bool stop = in && ((double)rj < (double)rc - V.nearby);
                const unsigned long long sm = __ballot(stop);
                const bool ok = in && (sm == 0ull || lane < __ffsll((long long)sm) - 1);
                if (ok) {
                    const float d = ll_walk_d2(p, qx, qy, qz);
                    const int ord = M + (c - 1 - j);
                    if (PLANE) { if (rj >= rc) ll_best_take(b2, d, ord, jp, dmax); else ll_best_take(b3, d, ord, jp, dmax); }
                    else if (rj < rc) ll_best_take(b2, d, ord, jp, dmax);
                }
                if (sm) break;
            }
            ll_best_reduce<64>(b2);
            if (PLANE) ll_best_reduce<64>(b3);
            if (lane == qq) { res_b = b2.j; res_c = b3.j; }
        }
    }
    if (have) {
        bool valid = closest >= 0 && res_b >= 0 && (!PLANE || res_c >= 0);       /* :556 / :723 */
        out_a[qi] = valid ? closest : -1;
        out_b[qi] = valid ? res_b : -1;
        if (PLANE) out_c[qi] = valid ? res_c : -1;
    }
}

/* 2nd launch bound = waves per SIMD.  Left alone the compiler spends 106 SGPRs and lands on 7; asked for 8 it fits in 78
 * with the same 60 VGPRs and no scratch -- the kernel is latency-bound, one more wave per SIMD is worth 17 % (A/B, one box).
 * LDS: 4 KB queries + 6.8 KB entry bounds + 7 KB ring tables + 1.3 KB walk tables + 0.5 KB = 19.6 of the 20 KB that eight
 * workgroups per CU leave each. */
#ifndef LL_ASSOC_WAVES
#define LL_ASSOC_WAVES 8          /* waves per SIMD = workgroups per CU the kernel is compiled for (A/B: 4 with more loads in flight per lane) */
#endif
__global__ __launch_bounds__(LL_BLOCK, LL_ASSOC_WAVES) void k_associate(LLView V, int first, int count, int qb_corner, int qb_plane, int qpb)
{
    /* All query blocks of a scan on ONE XCD (workgroup b runs on XCD b % 8): they search the same target grid, and an XCD's 4 MB L2 is
     * private -- dealt round-robin over the XCDs, every scan's target would be fetched into all eight of them. */
    const int per = qb_corner + qb_plane;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int sl = (jb / per) * 8 + xcd, item = jb % per;
    if (sl >= count) return;
    const int s = first + sl;
    if (item == 0 && threadIdx.x == 0) V.assoc_tgt[s] = (s == V.carry_slot) ? -1 : s - 1;   /* whose points this slot's correspondences name from now on */
    __shared__ float4 qs[LL_BLOCK];
    __shared__ unsigned long long rtab[(LL_BLOCK / 8) * 2 * LL_ATAB_W];
    __shared__ int cellb[(LL_BLOCK / 8) * 3 * LL_RING_CELLS];
    __shared__ int tab[LL_TAB_WORDS];
    __shared__ unsigned char perm[LL_BLOCK];
    __shared__ int hist[64];
    const ScanHdr h = V.hdr[s];
    const bool ok = h.status == 0;
    if (item < qb_corner) {
        const int nq = ok ? h.n_sharp : 0;
        if (item * qpb >= nq) return;
        ll_associate_block<false>(V, s, item, qpb, V.sharp + (size_t)s * V.cap_sharp, nq, ll_target(V, s, 0),
                                  V.eq_a + (size_t)s * V.cap_sharp, V.eq_b + (size_t)s * V.cap_sharp, nullptr, qs, rtab, cellb, tab, perm, hist);
    } else {
        const int qb = item - qb_corner;
        const int nq = ok ? h.n_flat : 0;
        if (qb * qpb >= nq) return;
        ll_associate_block<true>(V, s, qb, qpb, V.flat + (size_t)s * V.cap_flat, nq, ll_target(V, s, 1),
                                       V.pq_a + (size_t)s * V.cap_flat, V.pq_b + (size_t)s * V.cap_flat,
                                       V.pq_c + (size_t)s * V.cap_flat, qs, rtab, cellb, tab, perm, hist);
    }
}

void ll_launch_build_grid(const LLView &V, int first, int count, int carry, hipStream_t st, LLProfiler *prof)
{
    static size_t attr_bytes[LL_MAX_DEVICES] = {0};   /* 68 KiB histogram + the chunk table + static LDS exceed the default dynamic-LDS limit */
    const size_t lds = ll_grid_lds_bytes(V);
    ll_ensure_dynamic_lds(k_build_grid, lds, attr_bytes);
    ll_prof_mark(prof, LL_K_GRID, st);
    hipLaunchKernelGGL(k_build_grid, dim3(2 * (carry ? 1 : count)), dim3(LL_GB), lds, st, V, first, carry ? 1 : count, carry);
    ll_prof_mark(prof, LL_K_END, st);
}

void ll_launch_associate(const LLView &V, int first, int count, hipStream_t st, LLProfiler *prof)
{
    /* a node-style call (one scan pair) would put 9 workgroups of 8 sequential passes on a 256-CU chip: 217 us of dependent
     * loads; 32 queries per workgroup (one pass) spread the same work over 72 workgroups */
#ifdef LL_ASSOC_QPB
    const int qpb = (count <= 16) ? 32 : LL_ASSOC_QPB;         /* A/B builds (tools/make_ab_variant.sh) */
#else
    const int qpb = (count <= 16) ? 32 : LL_BLOCK;
#endif
    const int qbc = (V.cap_sharp + qpb - 1) / qpb, qbp = (V.cap_flat + qpb - 1) / qpb;
    ll_prof_mark(prof, LL_K_ASSOCIATE, st);
    hipLaunchKernelGGL(k_associate, dim3(8 * (qbc + qbp) * ((count + 7) / 8)), dim3(LL_BLOCK), 0, st, V, first, count, qbc, qbp, qpb);
    ll_prof_mark(prof, LL_K_END, st);
}
