/*
 * ll_features.hip -- a2 + a3 + a4: curvature, per-segment greedy feature pick, less-flat VoxelGrid.
 * Replaces scanRegistration.cpp:225-235, :246-368, :370-376 of /root/reference.
 *
 * One 256-thread workgroup per (scan, ring): the ring is the reference's unit of sequential dependence
 * (cloudNeighborPicked marks leak from segment j into segment j+1 of the same ring, never across rings,
 * because scanStartInd/EndInd keep a 5-point margin).  Everything between reading laserCloud and writing
 * labels + feature points stays on chip:
 *   phase 1  512-point tiles of x/y/z (+5 halo, the flat array is used like the reference: curvature
 *            crosses ring boundaries) staged in LDS, the next tile's loads in flight during the math ->
 *            11-tap curvature (strict left-to-right f32, no FMA), consecutive-point gap flags (1 bit/point)
 *   phase 2  per slot: how far a pick's cloudNeighborPicked marks reach forwards / backwards (:288-311)
 *   phase 3  the pick without a sort: one wave per segment keeps the segment in registers as masked keys; a pick is
 *            lane-local max -> DPP wave max -> ballot (ties: second reduction on the index) -> one-row range
 *            compare.  std::sort leaves equal curvatures unspecified; this path and the oracle define ascending
 *            index.  Segments run concurrently; a wave imports the earlier segments' forward marks only when it is
 *            about to pick one of its first five points (the only place they can matter)
 *   phase 4  less-flat points (label <= 0) compacted in index order with their coordinates in registers, voxel index
 *            per PCL's formula, stable LSD radix sort (4-bit digits, match-any ranking) by (voxel, input order),
 *            voxel runs summed left to right in f32 by the thread that owns the run head
 *   phase 5  labels out; feature points straight into the published clouds: the ring's offsets come from a decoupled
 *            look-back over the earlier rings' counts (no separate compaction pass).
 * Capacity tiers (ll_launch_features): rings of at most 2304 points -- every ring of a 2048-column sensor -- run the 9-row
 * instantiation at six workgroups per CU whatever max_ring_points is.  Longer rings (a real HDL-64E under the linear 64-ring
 * model puts two lasers into some bins) are worked on FIRST, by the 18- / 32-row instantiations, which stage their lists and
 * publish their counts without waiting for anybody; in the main launch the workgroup of such a ring only places the staged
 * lists at the look-back offsets.
 * The kernel is VALU-bound (~80 % VALU busy at six workgroups per CU): LDS is kept at ~26 KB for a 2304-point
 * ring capacity, VGPRs at 80.  ROWS = sort records per thread (capacity 256 * ROWS).
 * HBM traffic per ring point: 16 B read (+ L2-hot re-reads for voxel keys and centroids), 1 B label, features.
 */
#include "ll_common.h"
#include <limits.h>
#include <type_traits>

__device__ __forceinline__ bool ll_xcd_map2(int id, int per_scan, int count, int &scan, int &item)
{
    const int xcd = id & 7, j = id >> 3;
    scan = (j / per_scan) * 8 + xcd;
    item = j % per_scan;
    return scan < count;
}

#ifndef LL_FTILE
#define LL_FTILE 512      /* points per curvature tile */
#endif
#ifndef LL_FWAVES
#define LL_FWAVES 6
#endif
#ifndef LL_FWAVES_SPLIT
#define LL_FWAVES_SPLIT 6     /* the voxel-only kernel of the split pipeline */
#endif
#ifdef LL_PHASE_STOP
#define LL_LOOKBACK_SPINS 1            /* instruction-count builds return early and never publish: do not wait for them */
#else
#define LL_LOOKBACK_SPINS (1 << 24)    /* bounded so that a logic error ends in wrong results, not in a hung GPU */
#endif
#ifndef LL_PICK_PRIO
#define LL_PICK_PRIO 3     /* wave priority during the pick (the workgroup's longest serial stretch) */
#endif
#ifndef LL_TAIL_PRIO
#define LL_TAIL_PRIO 0     /* ... and after it */
#endif
#define LL_NLIST 176      /* per segment slots: sharp[6][2] lsharp[6][20] flat[6][4] + counters[6][3] */

struct FeatLds {
    unsigned *k32;             /* [mr] phases 1-3: curvature bits per slot; phase 4: sort key (voxel index) */
    unsigned short *k16;       /* [mr] phases 2-3: suppression extents per slot; phase 4: sort payload (local index) */
    float *tx, *ty, *tz;       /* [LL_FTILE + 16] phase 1; afterwards the region is the pick's per-wave scratch, then the radix counters */
    unsigned *picked, *gapf;   /* bitmaps over local index */
    int8_t *lab;               /* [mr] */
    int *lists;                /* [LL_NLIST] */
    int *cnt;                  /* [32 * ROWS * 4 + 1] radix counters (= the tile region) */
    int *sc;                   /* [64] scan scratch [0..15], per-wave bounds [32..55], finished-segment mask [60] */
};

/* the curvature tile; afterwards four per-wave scratch rows of 64 * SR u16 (SR = ceil(ROWS * 256 / 384)) for the pick */
static size_t ll_feat_tile_bytes(size_t rows)
{
    const size_t t = 3 * 4 * (size_t)(LL_FTILE + 16), w = 4 * 64 * ((rows * 256 + 383) / 384) * 2;
    const size_t c = 4 * (32 * rows * 4 + 4);                   /* phase 4: the radix counters (5-bit digits) live here too */
    const size_t m = t > w ? t : w;
    return m > c ? m : c;
}

size_t ll_features_lds_bytes(int max_ring, int split)      /* max_ring: the ring capacity of the LAUNCH (a tier of ll_launch_features) */
{
    const size_t mr = (size_t)((max_ring + 255) / 256 * 256);
    const size_t rows = (mr / 256 <= 9) ? 9 : (mr / 256 <= 12) ? 12 : (mr / 256 <= 18) ? 18 : 32;     /* the ROWS instantiation that will run */
    size_t b = 4 * mr + 2 * mr;              /* k32 + k16 */
    b += split ? 4 * (32 * rows * 4 + 4) : ll_feat_tile_bytes(rows);   /* radix counters | tile / pick scratch / radix counters */
    b += (split ? 1 : 2) * 4 * (mr / 32 + 2);   /* bitmaps */
    if (!split) b += mr;                     /* labels */
    b += 4 * LL_NLIST;
    b += 4 * 64;
    return (b + 15) / 16 * 16 + 64;
}
size_t ll_features_lds_bytes(int max_ring) { return ll_features_lds_bytes(max_ring, 0); }   /* the larger of the two: ll_create's capacity check */

template <bool SPLIT>
__device__ __forceinline__ FeatLds ll_carve(unsigned char *base, int max_ring)
{
    const size_t mr = (size_t)((max_ring + 255) / 256 * 256);
    const size_t rows = (mr / 256 <= 9) ? 9 : (mr / 256 <= 12) ? 12 : (mr / 256 <= 18) ? 18 : 32;     /* the ROWS instantiation that will run */
    const size_t tile = 3 * 4 * (size_t)(LL_FTILE + 16), wscr = 4 * 64 * ((rows * 256 + 383) / 384) * 2, cbytes = 4 * (32 * rows * 4 + 4);
    const size_t tile_bytes = SPLIT ? cbytes : ((tile > wscr ? tile : wscr) > cbytes ? (tile > wscr ? tile : wscr) : cbytes);
    FeatLds L;
    unsigned char *p = base;
    L.k32 = (unsigned *)p; p += 4 * mr;
    L.tx = (float *)p; L.ty = L.tx + (LL_FTILE + 16); L.tz = L.ty + (LL_FTILE + 16);
    p += tile_bytes;
    L.picked = (unsigned *)p; p += 4 * (mr / 32 + 2);
    L.gapf = (unsigned *)p; if (!SPLIT) p += 4 * (mr / 32 + 2);       /* split: never touched */
    L.lists = (int *)p; p += 4 * LL_NLIST;
    L.cnt = (int *)L.tx;                      /* radix counters of phase 4 share the tile region */
    L.sc = (int *)p; p += 4 * 64;
    L.k16 = (unsigned short *)p; p += 2 * mr;
    L.lab = (int8_t *)p;
    return L;
}

/* Stable least-significant-digit radix sort of the records (k32[g], k16[g]), g < n <= 256*ROWS, by k32; one workgroup.
 * Record g lives in wave g / (64 * nrows), row (g / 64) % nrows, lane g % 64 (nrows = rows per wave), so that the order
 * (wave, row, lane) is the input order.  Its destination is
 *   #(records with a smaller digit) + #(same digit, earlier wave) + #(same digit, same wave, earlier row) + rank in its row:
 * every wave counts into its own 2^BITS counters, one row after the other (a wave's LDS operations execute in order, so
 * the counter a row reads already holds the rows before it), the rank inside the row comes from a wave match-any, and one
 * workgroup exclusive scan over the digit-major (digit, wave) table turns the counters into bases.  Wave-private counters
 * make the table small enough for digits of up to 8 bits: the keys are below 2^key_bits (the caller knows: the voxel grid's
 * dimensions), which takes ceil(key_bits / 8) passes of 5..8-bit digits (three for the usual 17..24 bits of a ring's voxel indices).
 * The last row is padded with all-ones keys that take part like records (they stay at the end, no per-record guards);
 * rows beyond it are skipped.  Keys must be < 0xffffffff.  Stability makes the result ordered by (k32, original
 * position): exactly the (voxel, input order) order the oracle defines. */
template <int ROWS, bool PRELOADED = false>
__device__ __forceinline__ void ll_radix_sort(unsigned *k32, unsigned short *k16, int n, int key_bits, int *cnt, int *sc, int tid,
                                              const unsigned *pre32 = nullptr, const unsigned short *pre16 = nullptr)
{
    constexpr int NW = LL_BLOCK / 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      /* uniform: scalar registers / branches */
    const int nrows = (n + LL_BLOCK - 1) / LL_BLOCK;            /* rows per wave, uniform */
    const int wbase = wave * nrows * 64;                        /* the wave's first record */
    const int myrows = max(0, min(nrows, (n - wbase + 63) >> 6));   /* rows of this wave that hold a record */
    unsigned e32[ROWS]; unsigned short e16[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int g = wbase + k * 64 + lane;
        e32[k] = 0xffffffffu; e16[k] = 0;
        if (PRELOADED) { if (k < myrows && g < n) { e32[k] = pre32[k]; e16[k] = pre16[k]; } }     /* the caller's registers, this layout */
        else if (k < myrows && g < n) { e32[k] = k32[g]; e16[k] = k16[g]; }
    }
    if (key_bits <= 0) return;                                  /* one voxel (or none): already in order */
    const int lo = 0, hi = key_bits;                            /* the caller's bound on the keys: no reduction over them to find the varying bits */
    auto pass = [&](auto bits_tag, int sh) __attribute__((always_inline)) {
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr int ND = 1 << BITS;
        constexpr unsigned DM = (unsigned)ND - 1u;
        for (int i = tid; i < NW * ND; i += LL_BLOCK) cnt[i] = 0;
        __syncthreads();                                        /* also: every thread holds its records in registers */
        int *wc = cnt + wave * ND;                              /* this wave's counters */
        int rnk[ROWS], pre[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            rnk[k] = 0; pre[k] = 0;
            if (k < myrows) {
                const int d = (int)((e32[k] >> sh) & DM);
                unsigned mlo, mhi;
                ll_match_any(d, BITS, ~0ull, mlo, mhi);
                rnk[k] = ll_match_rank(mlo, mhi);
                pre[k] = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   /* records of the earlier rows of this wave with digit d ... (an
                                                                                 * atomic load: it must see the rows' adds, whatever the compiler would like to merge) */
                if (rnk[k] == 0) atomicAdd(&wc[d], ll_match_count(mlo, mhi));   /* ... bumped by an LDS add that does not wait for the read:
                                                                                 * the wave's LDS operations execute in order, so the next row's read sees it */
            }
        }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) rnk[k] += pre[k];
        __syncthreads();
        {   /* exclusive scan of the (digit, wave) table in digit-major order: thread d owns digit d */
            int v[NW]; int s = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { v[w] = (tid < ND) ? cnt[w * ND + tid] : 0; s += v[w]; }
            int total;
            int run = ll_block_exscan(s, sc, total);
            if (tid < ND) {
#pragma unroll
                for (int w = 0; w < NW; ++w) { cnt[w * ND + tid] = run; run += v[w]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            if (k < myrows) {
                const int d = (int)((e32[k] >> sh) & DM);
                const int pos = wc[d] + rnk[k];
                k32[pos] = e32[k]; k16[pos] = e16[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            if (k < myrows) { const int g = wbase + k * 64 + lane; e32[k] = k32[g]; e16[k] = k16[g]; }
        }
    };
    int sh = lo, rem = hi - lo;
    for (int left = (rem + 7) / 8; left > 0; --left) {
        const int bits = max(5, (rem + left - 1) / left);       /* spread the varying bits evenly over the passes */
        if (bits <= 5) pass(std::integral_constant<int, 5>{}, sh);
        else if (bits == 6) pass(std::integral_constant<int, 6>{}, sh);
        else if (bits == 7) pass(std::integral_constant<int, 7>{}, sh);
        else pass(std::integral_constant<int, 8>{}, sh);
        sh += bits; rem -= bits;
    }
}

/* wave-wide max of a u32, result uniform: quad swaps, row shifts, row broadcasts (DPP), then lane 63 */
__device__ __forceinline__ unsigned ll_wave_max_u32(unsigned v)
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

__device__ __forceinline__ bool ll_bit(const unsigned *bm, int i) { return (bm[i >> 5] >> (i & 31)) & 1u; }

extern __shared__ __attribute__((aligned(16))) unsigned char ll_smem[];

/* opt-in phase timing (tools/phase_timing.py builds with -DLL_PHASE_TIMING): thread 0 of every workgroup adds the
 * s_memtime cycles spent in each phase to V.dbg[phase]; V.dbg[15] counts workgroups */
#ifdef LL_PHASE_TIMING
#define LL_PHASE_BEGIN() long long ll_t0 = (tid == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0
#define LL_PHASE(i) do { __syncthreads(); if (tid == 0) { const long long t1 = (long long)__builtin_amdgcn_s_memtime(); \
    atomicAdd(&V.dbg[i], (unsigned long long)(t1 - ll_t0)); ll_t0 = t1; } } while (0)
#define LL_WAIT_BEGIN() const long long ll_w0 = (tid == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0
#define LL_WAIT_END() do { if (tid == 0) atomicAdd(&V.dbg[7], (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - ll_w0)); } while (0)
#elif defined(LL_PHASE_STOP)   /* tools/phase_valu.py: the kernel returns after phase LL_PHASE_STOP (instruction counts per phase by difference) */
#define LL_PHASE_BEGIN() do {} while (0)
#define LL_PHASE(i) do { if ((i) == LL_PHASE_STOP) return; } while (0)
#define LL_WAIT_BEGIN() do {} while (0)
#define LL_WAIT_END() do {} while (0)
#else
#define LL_PHASE_BEGIN() do {} while (0)
#define LL_PHASE(i) do {} while (0)
#define LL_WAIT_BEGIN() do {} while (0)
#define LL_WAIT_END() do {} while (0)
#endif

/* 2nd launch bound = waves per SIMD: six 256-thread workgroups per CU for the common 2304-point capacity (<= 80 VGPRs) */
/* ring_lo < ring length <= ring_hi: the rings this launch extracts (ring_hi, a multiple of 256 <= 256 ROWS, also sizes its LDS).
 * stage_only (= more rows than the main launch's nine): a tier of long rings, launched before the main one -- lists to the staging rows,
 * counts published, no look-back.  The main launch (ring_lo = INT_MIN) extracts its rings and places the staged lists of the longer ones. */
/* SPLIT: phases 1-3 ran in k_ring_pick (ll_pick.hip): this launch starts from its lists (ring_rec) and does phases 4-5 only. */
template <int ROWS, bool SPLIT>
__global__ __launch_bounds__(LL_BLOCK, (ROWS <= 9 ? (SPLIT ? LL_FWAVES_SPLIT : LL_FWAVES) : ROWS <= 12 ? 5 : ROWS <= 18 ? 3 : 1)) void k_ring_features(LLView V, int first, int count, int ring_lo, int ring_hi)
{
    constexpr int stage_only = ROWS > 9 ? 1 : 0;                      /* the tiers of long rings are the instantiations with more rows than the main launch's */
    static_assert(ROWS <= 32, "lfm / headm / endm hold one bit per row of a thread");
    int sl, r;
    if (!ll_xcd_map2(blockIdx.x, V.R, count, sl, r)) return;
    const int s = first + sl;
    const int tid = threadIdx.x, lane = tid & 63;
    /* header and offsets fetched together -- the early exit below would otherwise put a memory round trip between them */
    const ScanHdr h = V.hdr[s];
    const int off = V.ring_off[(size_t)s * (V.R + 1) + r];
    const int nr = V.ring_off[(size_t)s * (V.R + 1) + r + 1] - off;
    if (h.status != 0) return;                                        /* every ring of the scan takes this exit: nobody waits */
    if (nr <= ring_lo || nr > ring_hi) return;                        /* another tier's ring (a longer ring that an earlier tier has staged: k_ring_place moves it) */
    const int N = h.n;
    /* One 64-bit word per ring carries the four counts AND the launch tag, written and polled with relaxed atomics: no
     * release / acquire fence is needed (at agent scope those write back / invalidate the whole L2 on this chip). */
    unsigned long long *ring_pub = V.ring_pub + (size_t)s * V.R;
    const unsigned long long tag = (unsigned long long)V.epoch << 40;
    auto ll_pub = [&](int c0, int c1, int c2, int c3) {
        __hip_atomic_store(&ring_pub[r], tag | ((unsigned long long)c3 << 16) | ((unsigned long long)c1 << 9) | ((unsigned long long)c2 << 4) | (unsigned long long)c0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto ll_poll = [&](int q, int v[4]) {
        unsigned long long w; int spins = 0;
        while (((w = __hip_atomic_load(&ring_pub[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 40) != (unsigned long long)V.epoch && ++spins < LL_LOOKBACK_SPINS)
            __builtin_amdgcn_s_sleep(16);
#ifndef LL_PHASE_STOP
        if (spins >= LL_LOOKBACK_SPINS) V.hdr[s].status = -7;     /* LL_ERR_STATE: an earlier ring never published -- fail loudly, not wrongly */
#endif
        v[0] += (int)(w & 15u); v[2] += (int)((w >> 4) & 31u); v[1] += (int)((w >> 9) & 127u); v[3] += (int)((w >> 16) & 0xffffffu);
    };
    if (nr <= 0) {                                                    /* empty ring: publish zero counts for the rings behind it */
        if (tid == 0) {
            ll_pub(0, 0, 0, 0);
            if (r == V.R - 1) {                                       /* the scan's totals are the last ring's offsets */
                int tot[4] = {0, 0, 0, 0};
                for (int q = 0; q < r; ++q) ll_poll(q, tot);
                ScanHdr *hh = &V.hdr[s];
                hh->n_sharp = tot[0]; hh->n_less_sharp = tot[1]; hh->n_flat = tot[2]; hh->n_less_flat = tot[3];
            }
        }
        return;
    }
    const int S = off + 5, E = off + nr - 6;                          /* scanStartInd / scanEndInd (:218-220) */
    const bool active = (E - S >= 6);                                 /* :248 */
    const int Lseg = active ? (E - S) : 0;                            /* indices S .. E-1 are in segments */
    /* the ring's points by local index; laserCloud index g of the reference = off + local index.  Neighbours beyond the
     * ring's ends (the curvature of a ring's first / last five points reaches into the adjacent rings, :225-235) are looked
     * up through the offsets table: the rings sit at a fixed stride (ll_organize.hip) */
    const float4 *slot_cloud = V.cloud + (size_t)s * V.CS;
    const float4 *cloud = slot_cloud + (size_t)r * V.ring_cap - off;          /* cloud[off + li] = the ring's point li */
    auto cloud_at = [&](int g) -> float4 {                                      /* any laserCloud index 0 <= g < N */
        int l = g - off;
        if (l >= 0 && l < nr) return cloud[g];
        const int *ro = V.ring_off + (size_t)s * (V.R + 1);
        int q = r;
        if (l < 0) { do { --q; l += ro[q + 1] - ro[q]; } while (l < 0); }
        else { int c = nr; do { l -= c; ++q; c = ro[q + 1] - ro[q]; } while (l >= c); }
        return slot_cloud[(size_t)q * V.ring_cap + l];
    };
    FeatLds L = ll_carve<SPLIT>(ll_smem, ring_hi);
    float *fs = (float *)(L.sc + 32);                                 /* 24 floats: per-wave bounds */

    /* Decoupled look-back over the rings of the scan (they run on one XCD, dispatched in ring order): publish() this
     * ring's four feature counts as soon as they are known, go on with everything that does not need the offsets, then
     * prefix() waits for every earlier ring's counts and returns their sums = this ring's offsets in the four published
     * clouds (ring, segment, pick order; scanRegistration.cpp:273-279, :325, :376).  Each is called exactly once. */
    auto publish = [&](int c0, int c1, int c2, int c3) __attribute__((always_inline)) { if (tid == 0) ll_pub(c0, c1, c2, c3); };
    auto prefix = [&](int *outp) __attribute__((always_inline)) {
        int v[4] = {0, 0, 0, 0};
        for (int q = tid; q < r; q += LL_BLOCK) ll_poll(q, v);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = ll_wave_sum_i32(v[c]);
        __syncthreads();
        if (lane == 0) for (int c = 0; c < 4; ++c) L.sc[32 + (tid >> 6) * 4 + c] = v[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) { int t = 0; for (int w = 0; w < LL_BLOCK / 64; ++w) t += L.sc[32 + w * 4 + c]; outp[c] = t; }
        __syncthreads();
    };
    int roff[4] = {0, 0, 0, 0}; bool looked_back = false;           /* this ring's offsets in sharp / less-sharp / flat / less-flat */
    /* staging rows of the long rings: the slot's own search-grid arrays, dead between this kernel and k_build_grid (which
     * rewrites them), and V.stage_sf; plain stores -- the reader is a later launch */
    float4 *stage_lf = V.gpts_s + (size_t)s * V.NP + off;                                  /* less-flat: at the ring's laserCloud offset */
    float4 *stage_ls = V.gpts_c + (size_t)s * V.cap_lsharp + (size_t)r * (LL_SEGS * LL_LSHARP_PER_SEG);
    float4 *stage_sf = V.stage_sf + ((size_t)s * V.R + r) * LL_STAGE_SF;                   /* sharp [12], flat [24] */
    const int nwords = (nr + 31) / 32 + 1;
    if constexpr (SPLIT) { for (int i = tid; i < nwords; i += LL_BLOCK) L.picked[i] = 0; }
    else {
        for (int i = tid; i < nwords; i += LL_BLOCK) { L.picked[i] = 0; L.gapf[i] = 0; }
        for (int i = tid; i < (nr + 3) / 4; i += LL_BLOCK) ((unsigned *)L.lab)[i] = 0u;   /* labels, four at a time (the array is 4-byte aligned and padded) */
    }
    if (tid < 3 * LL_SEGS) L.lists[156 + tid] = 0;                    /* per segment: n_sharp, n_lsharp, n_flat */
    if (tid == 0) L.sc[60] = 0;                                       /* segments finished (bit j) */
    __syncthreads();

    /* the picked points of the per-segment lists: thread t holds the local index of entry t of each list and its position
     * (segment, pick order) among the ring's sharp / less-sharp / flat points, -1 = no entry */
    /* (only where the points ARE goes across the wait: the three points themselves, fetched ahead of it, used to live in
     * scratch memory for its whole length -- 12 KB written and read back per ring, more than half of the kernel's write traffic) */
    int fsrc0 = 0, fsrc1 = 0, fsrc2 = 0; int fpos[3] = {-1, -1, -1};
    auto gather_lists = [&]() __attribute__((always_inline)) {
        const int js = tid / LL_SHARP_PER_SEG, jl = tid / LL_LSHARP_PER_SEG, jf = tid / LL_FLAT_PER_SEG;
        int os = 0, ol = 0, of = 0;
        for (int j = 0; j < LL_SEGS; ++j) {
            if (j < js) os += L.lists[156 + j * 3];
            if (j < jl) ol += L.lists[157 + j * 3];
            if (j < jf) of += L.lists[158 + j * 3];
        }
        if (js < LL_SEGS && tid % LL_SHARP_PER_SEG < L.lists[156 + js * 3]) { fpos[0] = os + tid % LL_SHARP_PER_SEG; fsrc0 = L.lists[tid]; }
        if (jl < LL_SEGS && tid % LL_LSHARP_PER_SEG < L.lists[157 + jl * 3]) { fpos[1] = ol + tid % LL_LSHARP_PER_SEG; fsrc1 = L.lists[12 + tid]; }
        if (jf < LL_SEGS && tid % LL_FLAT_PER_SEG < L.lists[158 + jf * 3]) { fpos[2] = of + tid % LL_FLAT_PER_SEG; fsrc2 = L.lists[132 + tid]; }
    };
    auto write_labels = [&]() __attribute__((always_inline)) {
        if (SPLIT) return;                                            /* k_ring_pick wrote cloudLabel */
        int8_t *label = V.label + (size_t)s * V.NP + off;
        for (int i = tid; i < nr; i += LL_BLOCK) label[i] = L.lab[i];
    };
    auto seg_totals = [&](int &ns_, int &nls_, int &nf_) __attribute__((always_inline)) {
        ns_ = 0; nls_ = 0; nf_ = 0;
        for (int j = 0; j < LL_SEGS; ++j) { ns_ += L.lists[156 + j * 3]; nls_ += L.lists[157 + j * 3]; nf_ += L.lists[158 + j * 3]; }
    };

    LL_PHASE_BEGIN();
    if constexpr (SPLIT) {
        /* the pick's lists of this ring: local indices + per-segment counts -> L.lists; the less-sharp picks (label 1 / 2) -> bitmap:
         * less-flat = every segment point that is not one of them (:361-367) */
        const unsigned short *rec = V.ring_rec + ((size_t)s * V.R + r) * LL_REC_U16;
        if (tid < 174) L.lists[tid] = (int)rec[tid];
        __syncthreads();
        if (tid < LL_SEGS * LL_LSHARP_PER_SEG && tid % LL_LSHARP_PER_SEG < L.lists[157 + (tid / LL_LSHARP_PER_SEG) * 3]) {
            const int li = L.lists[12 + tid];
            atomicOr(&L.picked[li >> 5], 1u << (li & 31));
        }
        __syncthreads();
    } else {
    /* ---------------- phase 1: curvature + gap flags + sort records ---------------- */
    constexpr int TLOADS = (LL_FTILE + 10 + LL_BLOCK - 1) / LL_BLOCK;
    /* TWO tiles' points in flight during the math (preA: even tiles, preB: odd tiles): one tile ahead leaves a workgroup
     * with 8 KB outstanding -- 48 KB per CU -- which is what bounds this phase (a latency-bound stream), not its arithmetic */
    float4 preA[TLOADS], preB[TLOADS];
    auto tile_fetch = [&](float4 (&buf)[TLOADS], int c0) __attribute__((always_inline)) {
        if (c0 < nr) {
#pragma unroll
            for (int u = 0; u < TLOADS; ++u) {
                const int t = u * LL_BLOCK + tid, g = off + c0 - 5 + t;
                if (t < LL_FTILE + 10 && g >= 0 && g < N) buf[u] = cloud_at(g);
            }
        }
    };
    auto tile_math = [&](float4 (&buf)[TLOADS], int c0) __attribute__((always_inline)) {
        const int g0 = off + c0;                                      /* global index of tile slot 5 */
#pragma unroll
        for (int u = 0; u < TLOADS; ++u) {
            const int t = u * LL_BLOCK + tid, g = g0 - 5 + t;
            if (t < LL_FTILE + 10 && g >= 0 && g < N) { L.tx[t] = buf[u].x; L.ty[t] = buf[u].y; L.tz[t] = buf[u].z; }
        }
        __syncthreads();
        tile_fetch(buf, c0 + 2 * LL_FTILE);                           /* this buffer's next tile */
        /* The arithmetic runs for every lane (the tile arrays are valid for every slot a lane can address; what lies beyond
         * the ring is stale but harmless) and only the stores are predicated: no divergent regions around the math. */
#pragma unroll
        for (int k = 0; k < LL_FTILE / LL_BLOCK; ++k) {
            const int li = c0 + k * LL_BLOCK + tid;
            const int g = off + li, t = li - c0 + 5;
            const float *X = L.tx + t, *Y = L.ty + t, *Z = L.tz + t;
            {                                                         /* gap to the previous point (:290-293) */
                const float dx = X[0] - X[-1], dy = Y[0] - Y[-1], dz = Z[0] - Z[-1];
                if (li < nr && g >= 1 && dx * dx + dy * dy + dz * dz > V.gap_gt) atomicOr(&L.gapf[li >> 5], 1u << (li & 31));   /* (double)g > 0.05 */
            }
            /* :225-235, strict left-to-right */
            const float dX = X[-5] + X[-4] + X[-3] + X[-2] + X[-1] - 10 * X[0] + X[1] + X[2] + X[3] + X[4] + X[5];
            const float dY = Y[-5] + Y[-4] + Y[-3] + Y[-2] + Y[-1] - 10 * Y[0] + Y[1] + Y[2] + Y[3] + Y[4] + Y[5];
            const float dZ = Z[-5] + Z[-4] + Z[-3] + Z[-2] + Z[-1] - 10 * Z[0] + Z[1] + Z[2] + Z[3] + Z[4] + Z[5];
            const float cv = dX * dX + dY * dY + dZ * dZ;
            if (V.write_curv) { if (li < nr && g >= 5 && g < N - 5) V.curv[(size_t)s * V.NP + g] = cv; }
            if (active && g >= S && g < E) L.k32[g - S] = ll_f2u(cv);  /* S >= 5, E <= N - 5, E - off < nr: implies the bounds above */
        }
        __syncthreads();
    };
    tile_fetch(preA, 0);
    tile_fetch(preB, LL_FTILE);
    for (int c0 = 0; c0 < nr; c0 += 2 * LL_FTILE) {
        tile_math(preA, c0);
        if (c0 + LL_FTILE < nr) tile_math(preB, c0 + LL_FTILE);
    }

    LL_PHASE(0);
    /* ---------------- phase 2: suppression extents (:288-311) per slot, all threads ----------------
     * A pick at local index li marks li-bn .. li+fn: fn / bn = how far the forward / backward walk gets before a
     * consecutive-point gap above the threshold stops it.  Bits gapf[li-4 .. li+5] straight from the LDS bitmap. */
    for (int q = tid; q < Lseg; q += LL_BLOCK) {
        const int b0 = q + 1;                                                        /* li - 4 */
        const unsigned long long w = ((unsigned long long)L.gapf[(b0 >> 5) + 1] << 32) | L.gapf[b0 >> 5];
        const unsigned bits = (unsigned)(w >> (b0 & 31)) & 0x3ffu;                   /* bit t = gapf[li - 4 + t] */
        const unsigned fwd = bits >> 5;                                              /* l = 1..5  -> gapf[li + l] */
        const int fn = fwd ? (__ffs(fwd) - 1) : 5;
        const unsigned bwd = bits & 0x1fu;                                           /* l = -1..-5 -> gapf[li + l + 1] = bits 4..0 */
        const int bn = bwd ? 4 - (31 - __clz((int)bwd)) : 5;                         /* stops at the first set bit walking down from bit 4 */
        L.k16[q] = (unsigned short)(bn | (fn << 4));
    }
    __syncthreads();

    LL_PHASE(1);
    /* ---------------- phase 3: the greedy pick (:251-359), one wave per segment, no sort ----------------
     * Visiting a segment in descending (curvature, index) order and taking every candidate that is not yet suppressed is
     * the same as repeatedly taking the arg-max over the still-eligible candidates, because suppression only grows;
     * likewise arg-min for the flats.  A wave keeps its segment (<= 64 * SR records) in registers as masked keys (0 =
     * not eligible): a pick is a lane-local max, one DPP wave max, a ballot for the owner, and a range compare.
     * Segments run concurrently on the four waves.  The only coupling the reference has between them is forward:
     * cloudNeighborPicked marks of segment j reach at most 5 points into the following segments, and they matter only
     * if one of those points is about to be picked.  So a wave exports its forward marks to the LDS bitmap when its
     * segment is finished, and a wave that is about to pick one of its first five points first waits for all earlier
     * segments and imports their marks (at most once per segment). */
    if (active) {
        constexpr int SR = (ROWS * LL_BLOCK + 383) / 384;
        int *donemask = L.sc + 60;
        __builtin_amdgcn_s_setprio(LL_PICK_PRIO);
        /* the wave index is uniform; saying so keeps the segment bounds and every branch on them scalar */
        for (int j = __builtin_amdgcn_readfirstlane(tid >> 6); j < LL_SEGS; j += LL_BLOCK / 64) {
            const int sp = Lseg * j / 6, ep = Lseg * (j + 1) / 6 - 1;     /* record slots; = (:253-254) - S */
            const int len = ep - sp + 1;
            const int li0 = sp + 5 + lane;                                /* local index of this lane's row-0 record */
            constexpr int EW = (SR + 3) / 4;
            unsigned cb[SR], exw[EW];                                     /* curvature bits; suppression extents, 4 rows per word */
            unsigned candc = 0, candf = 0, sup = 0;                       /* row bitmasks: corner / flat candidate, suppressed */
#pragma unroll
            for (int w = 0; w < EW; ++w) exw[w] = 0u;
#pragma unroll
            for (int k = 0; k < SR; ++k) {
                const int q = k * 64 + lane;
                cb[k] = 0u;
                if (q < len) {
                    cb[k] = L.k32[sp + q];
                    exw[k >> 2] |= (unsigned)L.k16[sp + q] << ((k & 3) * 8);
                    const float cv = ll_u2f(cb[k]);                       /* f32 curvature against the double literal 0.1, in f32 */
                    if (cv > V.curv_gt) candc |= 1u << k;                 /* :266 */
                    if (cv < V.curv_lt) candf |= 1u << k;                 /* :321 */
                }
            }
            int imp_below = (j == 0) ? 0 : 5;                             /* a choice below this slot needs the earlier segments' marks first (0: imported) */
            int nrec[2] = {0, 0};
            /* the corner candidates (usually a small part of the segment) compacted to the front rows, ascending index:
             * the corner pass then scans ceil(nc / 64) rows per pick instead of SR */
            unsigned short *wbuf = (unsigned short *)L.tx + (size_t)j % (LL_BLOCK / 64) * (64 * SR);   /* this wave's scratch */
            int nc = 0;
#pragma unroll
            for (int k = 0; k < SR; ++k) {
                const bool c = (candc >> k) & 1u;
                const unsigned long long m = __ballot(c);
                if (c) wbuf[nc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)(k * 64 + lane);
                nc += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const int ncr = (nc + 63) >> 6;
            unsigned ck[SR]; int cli[SR];                                  /* masked key (0 = not eligible); local index | suppression extents << 16 */
#pragma unroll
            for (int r = 0; r < SR; ++r) {
                ck[r] = 0u; cli[r] = 0;
                if (r < ncr && r * 64 + lane < nc) { const int q = (int)wbuf[r * 64 + lane]; ck[r] = L.k32[sp + q]; cli[r] = (sp + 5 + q) | ((int)L.k16[sp + q] << 16); }
            }
            for (int pass = 0; pass < 2; ++pass) {
                int npick = 0;
                unsigned myrec = 0;                                       /* lane n: pick n+1 as li | extents << 16 */
                /* non-negative float bits order like the floats; the flats maximise the complement */
                unsigned mk[SR];
                if (pass == 1) {
                    /* what the corner picks of this segment marked (exported to the bitmap below, together with any
                     * forward marks of earlier segments that are already there -- those are the reference's too) */
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                    for (int k = 0; k < SR; ++k) if (k * 64 + lane < len && ll_bit(L.picked, li0 + k * 64)) sup |= 1u << k;
                }
#pragma unroll
                for (int k = 0; k < SR; ++k) mk[k] = (pass == 1 && ((candf & ~sup) >> k) & 1u) ? ~cb[k] : 0u;
                /* one pick loop, instantiated for the row count it scans (NR rows of `key`): the corner pass over the
                 * compacted candidates (1, 2 or SR rows), the flat pass over the whole segment */
                auto pick_loop = [&](auto nr_tag, auto corner_tag, unsigned (&key)[SR]) __attribute__((always_inline)) {
                    constexpr int NR = decltype(nr_tag)::value;
                    constexpr bool CORNER = decltype(corner_tag)::value;
                    for (;;) {
                        /* lane-local best; rows ascend in index, so on equal keys ">=" keeps the larger index (descending
                         * visit order of the corners) and ">" the smaller (ascending order of the flats) */
                        unsigned best = key[0]; int row_l = 0;
#pragma unroll
                        for (int r = 1; r < NR; ++r) {
                            const bool t = CORNER ? key[r] >= best : key[r] > best;
                            best = t ? key[r] : best; row_l = t ? r : row_l;
                        }
                        const unsigned kmax = ll_wave_max_u32(best);
                        if (kmax == 0u) break;                            /* nothing eligible is left */
                        const unsigned long long bal = __ballot(best == kmax);
                        int selp;                                         /* row * 64 + lane of the choice */
                        if (CORNER && NR == 1) {
                            /* one row of candidates in ascending index: among equal curvatures the highest lane is the largest
                             * index, which the descending walk meets first -- no tie path */
                            selp = 63 - __builtin_clzll(bal);
                        } else if (__popcll(bal) == 1) {
                            const int f = __ffsll((long long)bal) - 1;
                            selp = (NR > 1 ? __builtin_amdgcn_readlane(row_l, f) * 64 : 0) + f;
                        } else {                                          /* equal curvatures in several lanes: index decides */
                            const int myp = row_l * 64 + lane;
                            const unsigned t = (best == kmax) ? (CORNER ? (unsigned)(myp + 1) : (unsigned)(0x10000 - myp)) : 0u;
                            const unsigned tm = ll_wave_max_u32(t);
                            selp = CORNER ? (int)tm - 1 : 0x10000 - (int)tm;
                        }
                        int selq = selp;                                  /* slot inside the segment */
                        int e = 0;                                        /* the choice's suppression extents: bn | fn << 4 */
                        if (CORNER) {                                     /* compacted layout -> the element's own slot, its extents with it */
                            int sl = __builtin_amdgcn_readlane(cli[0], selp & 63);
#pragma unroll
                            for (int r = 1; r < NR; ++r) if ((selp >> 6) == r) sl = __builtin_amdgcn_readlane(cli[r], selp & 63);
                            selq = (sl & 0xffff) - sp - 5;
                            e = sl >> 16;
                        }
                        if (selq < imp_below) {                           /* one of the segment's first five points, marks not imported yet */
                            const int need = (1 << j) - 1;
                            while ((__hip_atomic_load(donemask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & need) != need)
                                __builtin_amdgcn_s_sleep(2);
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            if (lane < 5 && lane < len && ll_bit(L.picked, li0)) { mk[0] = 0u; sup |= 1u; }
                            if ((cli[0] & 0xffff) - sp - 5 < 5 && ck[0] != 0u && ll_bit(L.picked, cli[0] & 0xffff)) ck[0] = 0u;   /* ascending order: row 0 */
                            imp_below = 0;
                            continue;                                     /* select again: the choice may be gone */
                        }
                        npick++;
                        if (CORNER && npick > LL_LSHARP_PER_SEG) break;   /* :281-284 */
                        const int sel = sp + selq + 5;
                        if (!CORNER) {                                    /* the owner's extents: uniform row, lane */
                            unsigned ew = 0u;
#pragma unroll
                            for (int w = 0; w < EW; ++w) if ((selq >> 8) == w) ew = (unsigned)__builtin_amdgcn_readlane((int)exw[w], selq & 63);
                            e = (int)((ew >> (((selq >> 6) & 3) * 8)) & 0xffu);
                        }
                        {   /* lane n: pick n + 1 (value and lane come out of scalar instructions: no read / write-lane hazard) */
                            const unsigned rec = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)sel | ((unsigned)e << 16)));
                            const int ln = __builtin_amdgcn_readfirstlane(npick - 1);
                            unsigned m0_keep;                               /* v_writelane takes one scalar operand + m0; m0 is the compiler's: put it back */
                            asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                                         : "+v"(myrec), "=&s"(m0_keep) : "s"(rec), "s"(ln));
                        }
                        if (!CORNER && npick >= LL_FLAT_PER_SEG) break;   /* :328-331: labelled, but no marking */
                        const int slo = sel - (e & 15), shi = sel + (e >> 4);
                        if (CORNER) {                                     /* the marks themselves go to the bitmap after the pass */
#pragma unroll
                            for (int r = 0; r < NR; ++r) key[r] = ((unsigned)((cli[r] & 0xffff) - slo) <= (unsigned)(shi - slo)) ? 0u : key[r];
                        } else {
                            /* rows are 64 apart and a pick marks at most 11 consecutive indices: per lane at most one row is hit */
                            const int dd = shi - li0;
                            const int kl = (dd >= 0 && (dd & 63) <= shi - slo) ? (dd >> 6) : -1;
#pragma unroll
                            for (int k = 0; k < NR; ++k) key[k] = (kl == k) ? 0u : key[k];
                        }
                    }
                };
                using std::integral_constant;
                if (pass == 1) pick_loop(integral_constant<int, SR>{}, integral_constant<bool, false>{}, mk);
                else if (ncr <= 1) pick_loop(integral_constant<int, 1>{}, integral_constant<bool, true>{}, ck);
                else if (ncr <= 2) pick_loop(integral_constant<int, 2>{}, integral_constant<bool, true>{}, ck);
                else pick_loop(integral_constant<int, SR>{}, integral_constant<bool, true>{}, ck);
                /* the picked records, lane-parallel: labels, list entries, forward marks */
                const int nr_ = pass == 0 ? min(npick, LL_LSHARP_PER_SEG) : npick;
                nrec[pass] = nr_;
                if (lane < nr_) {
                    const int sel = (int)(myrec & 0xffffu), e = (int)(myrec >> 16);
                    if (pass == 0) {
                        if (lane < LL_SHARP_PER_SEG) { L.lab[sel] = 2; L.lists[j * LL_SHARP_PER_SEG + lane] = sel; }
                        else L.lab[sel] = 1;
                        L.lists[12 + j * LL_LSHARP_PER_SEG + lane] = sel;
                    } else { L.lab[sel] = -1; L.lists[132 + j * LL_FLAT_PER_SEG + lane] = sel; }
                    /* marks into the bitmap: a corner pick's whole range from this segment's first index on (the flat pass
                     * reads it back; never below sp + 5 -- an earlier segment may still be running and must not see marks the
                     * reference makes after it), a flat pick's only beyond this segment */
                    const int shi = sel + (e >> 4), f0 = max(sel - (e & 15), pass == 0 ? sp + 5 : ep + 6);
                    if ((pass == 0 || lane < LL_FLAT_PER_SEG - 1) && shi >= f0) {
                        const unsigned long long bits = ((1ull << (shi - f0 + 1)) - 1ull) << (f0 & 31);
                        atomicOr(&L.picked[f0 >> 5], (unsigned)bits);
                        if (bits >> 32) atomicOr(&L.picked[(f0 >> 5) + 1], (unsigned)(bits >> 32));
                    }
                }
            }
            if (lane == 0) {
                L.lists[156 + j * 3] = min(nrec[0], LL_SHARP_PER_SEG); L.lists[157 + j * 3] = nrec[0]; L.lists[158 + j * 3] = nrec[1];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicOr(donemask, 1 << j);
        }
        __builtin_amdgcn_s_setprio(LL_TAIL_PRIO);
    }
    __syncthreads();
    }   /* !SPLIT */

    LL_PHASE(2);
    /* ---------------- phase 4: less-flat compaction + VoxelGrid (:361-376) ----------------
     * Blocked layout: thread t owns the `per` consecutive slots [t*per, (t+1)*per), so a workgroup exclusive scan of the
     * per-thread less-flat counts compacts in input order.  The points are read once (independent loads) and stay in
     * registers for the bounding box and for PCL's voxel index; after the stable sort by voxel the same blocked layout
     * prefetches every thread's points before the left-to-right f32 centroid sums. */
    int n_lf_out = 0;
    if (active) {
        int m = 0;
        bool sorted_ok = false;
        if constexpr (SPLIT) {
            /* The segment points go to the sort as they lie, in the sort's own (wave, row, lane) order -- coalesced rows, no compaction
             * scan, no trip through LDS: a less-sharp pick (not less-flat, :361-367) just carries the all-ones key that the padding
             * of the last row carries and ends up behind the m real records. */
            const int nrows_w = (Lseg + LL_BLOCK - 1) / LL_BLOCK;                  /* rows per wave, uniform */
            const int wbase = __builtin_amdgcn_readfirstlane(tid >> 6) * nrows_w * 64;
            float px[ROWS], py[ROWS], pz[ROWS];
            unsigned mem = 0;                                                        /* bit k: record wbase + k * 64 + lane is less-flat */
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int g = wbase + k * 64 + lane;
                if (k < nrows_w && g < Lseg) {
                    const float4 p = cloud[off + g + 5];
                    px[k] = p.x; py[k] = p.y; pz[k] = p.z;
                    if (!ll_bit(L.picked, g + 5)) mem |= 1u << k;
                }
            }
            float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY, mxz = -INFINITY;
#pragma unroll
            for (int k = 0; k < ROWS; ++k)
                if (k < nrows_w && ((mem >> k) & 1u)) {
                    mnx = fminf(mnx, px[k]); mny = fminf(mny, py[k]); mnz = fminf(mnz, pz[k]);
                    mxx = fmaxf(mxx, px[k]); mxy = fmaxf(mxy, py[k]); mxz = fmaxf(mxz, pz[k]);
                }
            mnx = ll_wave_min_f32(mnx); mny = ll_wave_min_f32(mny); mnz = ll_wave_min_f32(mnz);
            mxx = ll_wave_max_f32(mxx); mxy = ll_wave_max_f32(mxy); mxz = ll_wave_max_f32(mxz);
            if (lane == 0) { float *w = fs + (tid >> 6) * 6; w[0] = mnx; w[1] = mny; w[2] = mnz; w[3] = mxx; w[4] = mxy; w[5] = mxz; }
            { int nls_ = 0; for (int j = 0; j < LL_SEGS; ++j) nls_ += L.lists[157 + j * 3]; m = Lseg - nls_; }   /* every pick is a segment point */
            __syncthreads();
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
                /* "leaf size too small": output = input.  Also taken where PCL's int voxel index would wrap (div_b is up to d + 1 per axis, so
                 * the product of the three can pass 2^32 - 1 while d's stays below INT_MAX -- undefined in the reference; not reachable for a
                 * ring of one laser, whose points share an elevation): a wrapped key could equal the all-ones padding key and sort among the
                 * padding.  oracle/ll_oracle.c takes the same exit. */
                const bool too_small = d[0] * d[1] * d[2] > (long long)INT_MAX || (long long)div_b[0] * div_b[1] * div_b[2] > 0xffffffffLL;
                /* every real key is below key_end; the digit range covers key_end itself so that all-ones in it is above every real key */
                const long long key_end = too_small ? (long long)m : (long long)div_b[0] * div_b[1] * div_b[2];
                const int key_bits = min(32, 64 - __clzll(key_end));
                const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
                const float fb0 = (float)min_b[0], fb1 = (float)min_b[1], fb2 = (float)min_b[2];
                int rank_base = 0;                                                   /* too_small only: less-flat points before this wave's rows */
                if (too_small) {                                                     /* uniform; the key is the point's place among the less-flat points */
                    int wtot = 0;
#pragma unroll
                    for (int k = 0; k < ROWS; ++k) if (k < nrows_w) wtot += __popcll(__ballot((mem >> k) & 1u));
                    __syncthreads();
                    if (lane == 0) L.sc[tid >> 6] = wtot;
                    __syncthreads();
                    for (int w = 0; w < (tid >> 6); ++w) rank_base += L.sc[w];
                }
                unsigned e32[ROWS]; unsigned short e16[ROWS];
#pragma unroll
                for (int k = 0; k < ROWS; ++k) {
                    e32[k] = 0xffffffffu; e16[k] = 0;
                    if (k < nrows_w) {
                        const bool mb = (mem >> k) & 1u;
                        if (too_small) {
                            const unsigned long long bal = __ballot(mb);
                            if (mb) e32[k] = (unsigned)(rank_base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)));
                            rank_base += __popcll(bal);
                        } else if (mb) {
                            const int i0 = (int)(floorf(px[k] * inv) - fb0);
                            const int i1 = (int)(floorf(py[k] * inv) - fb1);
                            const int i2 = (int)(floorf(pz[k] * inv) - fb2);
                            e32[k] = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
                        }
                        e16[k] = (unsigned short)(wbase + k * 64 + lane + 5);       /* payload: local index */
                    }
                }
                LL_PHASE(3);
                ll_radix_sort<ROWS, true>(L.k32, L.k16, Lseg, key_bits, L.cnt, L.sc, tid, e32, e16);
                LL_PHASE(4);
                __syncthreads();
                sorted_ok = true;
            }
        } else {
        const int per = (Lseg + LL_BLOCK - 1) / LL_BLOCK;                        /* <= ROWS; uniform: the row loops below skip the rows beyond it by scalar branches */
        const int a0 = min(Lseg, tid * per), a1 = min(Lseg, a0 + per);          /* slots -> local index slot + 5 */
        float px[ROWS], py[ROWS], pz[ROWS];
        unsigned lfm = 0;                                                        /* bit u: slot a0 + u is less-flat */
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            const int q = a0 + u;
            if (u < per && q < a1) {
                const float4 p = cloud[off + q + 5];
                px[u] = p.x; py[u] = p.y; pz[u] = p.z;
                if (SPLIT ? !ll_bit(L.picked, q + 5) : (L.lab[q + 5] <= 0)) lfm |= 1u << u;
            }
        }
        float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY, mxz = -INFINITY;
#pragma unroll
        for (int u = 0; u < ROWS; ++u)
            if (u < per && ((lfm >> u) & 1u)) {
                mnx = fminf(mnx, px[u]); mny = fminf(mny, py[u]); mnz = fminf(mnz, pz[u]);
                mxx = fmaxf(mxx, px[u]); mxy = fmaxf(mxy, py[u]); mxz = fmaxf(mxz, pz[u]);
            }
        /* wave reduce min/max, then across the 4 waves through LDS */
        mnx = ll_wave_min_f32(mnx); mny = ll_wave_min_f32(mny); mnz = ll_wave_min_f32(mnz);
        mxx = ll_wave_max_f32(mxx); mxy = ll_wave_max_f32(mxy); mxz = ll_wave_max_f32(mxz);
        if (lane == 0) { float *w = fs + (tid >> 6) * 6; w[0] = mnx; w[1] = mny; w[2] = mnz; w[3] = mxx; w[4] = mxy; w[5] = mxz; }
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
            /* every key is below the number of voxels of the bounding box (below m when the filter degenerates): the sort's digit range */
            const long long key_end = too_small ? (long long)m : (long long)div_b[0] * div_b[1] * div_b[2];
            const int key_bits = key_end <= 1 ? 0 : min(32, 64 - __clzll(key_end - 1));
            const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
            const float fb0 = (float)min_b[0], fb1 = (float)min_b[1], fb2 = (float)min_b[2];
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                if (u < per && ((lfm >> u) & 1u)) {
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
            ll_radix_sort<ROWS>(L.k32, L.k16, m, key_bits, L.cnt, L.sc, tid);
            LL_PHASE(4);
            __syncthreads();
            sorted_ok = true;
        }
        }
        if (sorted_ok) {
            /* voxel runs -> centroids.  Thread t owns sorted positions [t*perm, (t+1)*perm): its points are fetched up
             * front, a run is summed by the thread that owns its head and may continue into the following threads' range */
            const int perm = (m + LL_BLOCK - 1) / LL_BLOCK;                          /* <= ROWS */
            const int b0 = min(m, tid * perm), b1 = min(m, b0 + perm);
            float4 pt[ROWS]; unsigned vk[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                if (u < perm && b0 + u < b1) { vk[u] = L.k32[b0 + u]; pt[u] = cloud[off + L.k16[b0 + u]]; }
            unsigned headm = 0;
            {
                unsigned prev = 0; bool has_prev = false;
                if (b0 > 0 && b0 < b1) { prev = L.k32[b0 - 1]; has_prev = true; }
#pragma unroll
                for (int u = 0; u < ROWS; ++u)
                    if (u < perm && b0 + u < b1) {
                        if (!has_prev || vk[u] != prev) headm |= 1u << u;
                        prev = vk[u]; has_prev = true;
                    }
            }
            int o = ll_block_exscan(__popc(headm), L.sc, n_lf_out);
            LL_PHASE(12);
            { int a_, b_, c_; seg_totals(a_, b_, c_); publish(a_, b_, c_, n_lf_out); }   /* the counts go out now, the wait comes last */
            /* CentroidPoint<PointXYZI>: f32 sums from zero in input order, divided by float(n).  A run that ends inside the
             * thread's range leaves its centroid in the registers of its last point (bit u of endm); nothing is stored until
             * the offsets are known */
            float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f; int cn = 0;
            unsigned endm = 0;
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                if (u < perm && b0 + u < b1) {
                    if ((headm >> u) & 1u) {
                        if (u > 0 && cn) {
                            const float fn = (float)cn;
                            pt[u > 0 ? u - 1 : 0] = make_float4(sx / fn, sy / fn, sz / fn, si / fn); endm |= 1u << (u > 0 ? u - 1 : 0);
                        }
                        sx = 0.0f; sy = 0.0f; sz = 0.0f; si = 0.0f; cn = 0;
                    } else if (!cn) continue;                                        /* tail of an earlier thread's run */
                    sx += pt[u].x; sy += pt[u].y; sz += pt[u].z; si += pt[u].w; ++cn;
                }
            LL_PHASE(13);
            /* The run that is open at the end of the range goes on with the next thread's points up to that thread's first
             * head: they sit in the next lane's registers (never overwritten above: a thread's leading points belong to
             * no run of its own) and come over by a one-lane wave shift.  Whatever lies beyond -- a next thread without a
             * head, or the next wave -- is fetched from the cloud. */
            {
#define LL_SHL1(x) __builtin_amdgcn_update_dpp(0, (int)(x), 0x130, 0xf, 0xf, false)      /* wave_shl:1: lane i <- lane i + 1 */
                const unsigned hm_n = (unsigned)LL_SHL1(headm);
                const int cnt_n = LL_SHL1(b1 - b0);
                const int c_n = (lane == 63) ? 0 : (hm_n ? __ffs((int)hm_n) - 1 : cnt_n);   /* leading non-head points of the next lane */
#pragma unroll
                for (int u = 0; u < ROWS; ++u)
                    if (u < perm) {
                        const float qx = __int_as_float(LL_SHL1(__float_as_int(pt[u].x))), qy = __int_as_float(LL_SHL1(__float_as_int(pt[u].y)));
                        const float qz = __int_as_float(LL_SHL1(__float_as_int(pt[u].z))), qw = __int_as_float(LL_SHL1(__float_as_int(pt[u].w)));
                        if (cn && u < c_n) { sx += qx; sy += qy; sz += qz; si += qw; ++cn; }
                    }
#undef LL_SHL1
                float4 last = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (cn) {
                    const unsigned vid = L.k32[b1 - 1];
                    for (int e = b1 + c_n; e < m && L.k32[e] == vid; ++e) {
                        const float4 q = cloud[off + L.k16[e]];
                        sx += q.x; sy += q.y; sz += q.z; si += q.w; ++cn;
                    }
                    const float fn = (float)cn; last = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
                }
                LL_PHASE(5);
                write_labels(); gather_lists();
                if (!stage_only) prefix(roff);
                looked_back = true;
                LL_PHASE(7);
                float4 *out = stage_only ? stage_lf : V.lflat + (size_t)s * V.NP + roff[3];
#pragma unroll
                for (int u = 0; u < ROWS; ++u) if (u < perm && ((endm >> u) & 1u)) out[o++] = pt[u];
                if (cn) out[o] = last;
            }
        }
    }

    /* ---------------- phase 5: labels + feature slots ---------------- */
    /* per-segment lists -> the published clouds at this ring's offsets, in (segment, pick order) */
    int ns = 0, nls = 0, nf = 0;
    seg_totals(ns, nls, nf);
    if (!looked_back) { publish(ns, nls, nf, n_lf_out); write_labels(); gather_lists(); if (!stage_only) prefix(roff); }   /* rings without a less-flat point */
    if (stage_only) {
        if (fpos[0] >= 0) stage_sf[fpos[0]] = cloud[off + fsrc0];
        if (fpos[1] >= 0) stage_ls[fpos[1]] = cloud[off + fsrc1];
        if (fpos[2] >= 0) stage_sf[LL_SEGS * LL_SHARP_PER_SEG + fpos[2]] = cloud[off + fsrc2];
        return;
    }
    {   /* the three loads together, then the three stores */
        float4 p0, p1, p2;
        if (fpos[0] >= 0) p0 = cloud[off + fsrc0];
        if (fpos[1] >= 0) p1 = cloud[off + fsrc1];
        if (fpos[2] >= 0) p2 = cloud[off + fsrc2];
        if (fpos[0] >= 0) V.sharp[(size_t)s * V.cap_sharp + roff[0] + fpos[0]] = p0;
        if (fpos[1] >= 0) V.lsharp[(size_t)s * V.cap_lsharp + roff[1] + fpos[1]] = p1;
        if (fpos[2] >= 0) V.flat[(size_t)s * V.cap_flat + roff[2] + fpos[2]] = p2;
    }
    if (r == V.R - 1 && tid == 0) {                                            /* the scan's totals */
        ScanHdr *hh = &V.hdr[s];
        hh->n_sharp = roff[0] + ns; hh->n_less_sharp = roff[1] + nls; hh->n_flat = roff[2] + nf; hh->n_less_flat = roff[3] + n_lf_out;
    }
    LL_PHASE(6);
#ifdef LL_PHASE_TIMING
    if (tid == 0) atomicAdd(&V.dbg[15], 1ull);
#endif
}

template <int ROWS, bool SPLIT>
static void ll_launch_ring_features(const LLView &V, int first, int count, int grid, int ring_lo, int ring_hi, hipStream_t st)
{
    static size_t attr_bytes[LL_MAX_DEVICES] = {0};
    const size_t lds_bytes = ll_features_lds_bytes(ring_hi, SPLIT ? 1 : 0);
    ll_ensure_dynamic_lds(k_ring_features<ROWS, SPLIT>, lds_bytes, attr_bytes);
    hipLaunchKernelGGL((k_ring_features<ROWS, SPLIT>), dim3(grid), dim3(LL_BLOCK), lds_bytes, st, V, first, count, ring_lo, ring_hi);
}

/* The long rings' lists were staged and their counts published by the tier launches; the main launch's rings have looked back at those
 * counts and stored around them.  One workgroup per scan -- on the XCD its rings ran on (scan sl <-> XCD sl % 8, ll_xcd_map2) -- reads all
 * ring words (every ring has published: the launches before this one are complete), scans the four counts over the rings and moves the
 * staged lists to their places.  (Until round 3 the long rings sat in the main launch as workgroups that did nothing but wait for their
 * predecessors' counts -- a fifth of an HDL-64E scan's workgroups holding a slot of the CU for most of a ring's run time.) */
__global__ __launch_bounds__(LL_BLOCK) void k_ring_place(LLView V, int first, int count, int ring_hi)
{
    const int sl = blockIdx.x;
    if (sl >= count) return;
    const int s = first + sl, tid = threadIdx.x, R = V.R;
    ScanHdr *hh = &V.hdr[s];
    if (hh->status != 0) return;
    __shared__ int cnt[4][LL_MAX_RINGS], pre[4][LL_MAX_RINGS + 1];
    __shared__ int lost;
    const int *ro = V.ring_off + (size_t)s * (R + 1);
    if (tid == 0) lost = 0;
    __syncthreads();
    if (tid < R) {
        const unsigned long long w = __hip_atomic_load(&V.ring_pub[(size_t)s * R + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((w >> 40) != (unsigned long long)V.epoch) lost = 1;                           /* a ring that never published: fail loudly (LL_ERR_STATE) */
        cnt[0][tid] = (int)(w & 15u); cnt[2][tid] = (int)((w >> 4) & 31u); cnt[1][tid] = (int)((w >> 9) & 127u); cnt[3][tid] = (int)((w >> 16) & 0xffffffu);
    }
    __syncthreads();
    if (lost) { if (tid == 0) hh->status = -7; return; }
    if (tid < 4) { int run = 0; for (int r = 0; r < R; ++r) { pre[tid][r] = run; run += cnt[tid][r]; } pre[tid][R] = run; }
    __syncthreads();
    for (int r = 0; r < R; ++r) {
        const int off = ro[r], nr = ro[r + 1] - off;
        if (nr <= ring_hi) continue;
        const float4 *stage_lf = V.gpts_s + (size_t)s * V.NP + off;
        const float4 *stage_ls = V.gpts_c + (size_t)s * V.cap_lsharp + (size_t)r * (LL_SEGS * LL_LSHARP_PER_SEG);
        const float4 *stage_sf = V.stage_sf + ((size_t)s * R + r) * LL_STAGE_SF;
        if (tid < cnt[0][r]) V.sharp[(size_t)s * V.cap_sharp + pre[0][r] + tid] = stage_sf[tid];
        if (tid < cnt[1][r]) V.lsharp[(size_t)s * V.cap_lsharp + pre[1][r] + tid] = stage_ls[tid];
        if (tid < cnt[2][r]) V.flat[(size_t)s * V.cap_flat + pre[2][r] + tid] = stage_sf[LL_SEGS * LL_SHARP_PER_SEG + tid];
        float4 *out = V.lflat + (size_t)s * V.NP + pre[3][r];
        for (int i = tid; i < cnt[3][r]; i += LL_BLOCK) out[i] = stage_lf[i];
    }
    /* the scan's totals are the last ring's offsets + counts: written by that ring in the main launch unless it is a long one */
    if (tid == 0 && ro[R] - ro[R - 1] > ring_hi) { hh->n_sharp = pre[0][R]; hh->n_less_sharp = pre[1][R]; hh->n_flat = pre[2][R]; hh->n_less_flat = pre[3][R]; }
}

void ll_launch_features(const LLView &V, int first, int count, size_t /* LDS of the largest tier: checked by ll_create */, hipStream_t st, LLProfiler *prof)
{
    const int groups = (count + 7) / 8;
    const int grid = 8 * V.R * groups;
    const int cap = (V.max_ring + 255) / 256 * 256;
    if (V.ring_split) {
        ll_prof_mark(prof, LL_K_PICK, st);
        ll_launch_pick(V, first, count, st);
        ll_prof_mark(prof, LL_K_RING_FEATURES, st);            /* V.epoch: the caller's per-context tag of this extract call (ll_next_epoch) */
        if (cap > 4608) ll_launch_ring_features<32, true>(V, first, count, grid, 4608, cap, st);
        if (cap > 3072) ll_launch_ring_features<18, true>(V, first, count, grid, 3072, cap < 4608 ? cap : 4608, st);
        if (cap > 2304) ll_launch_ring_features<12, true>(V, first, count, grid, 2304, cap < 3072 ? cap : 3072, st);
        ll_launch_ring_features<9, true>(V, first, count, grid, INT_MIN, cap < 2304 ? cap : 2304, st);
    } else {
    ll_prof_mark(prof, LL_K_RING_FEATURES, st);
    /* the tiers of long rings first (they wait for nobody), the main launch last: its look-back finds their counts published */
    if (cap > 4608) ll_launch_ring_features<32, false>(V, first, count, grid, 4608, cap, st);   /* <= 8192 points: the per-thread row masks are 32 bits wide */
    if (cap > 3072) ll_launch_ring_features<18, false>(V, first, count, grid, 3072, cap < 4608 ? cap : 4608, st);
    if (cap > 2304) ll_launch_ring_features<12, false>(V, first, count, grid, 2304, cap < 3072 ? cap : 3072, st);   /* two lasers of a 64-beam sensor in one bin: five workgroups per CU */
    ll_launch_ring_features<9, false>(V, first, count, grid, INT_MIN, cap < 2304 ? cap : 2304, st);
    }
    if (cap > 2304) hipLaunchKernelGGL(k_ring_place, dim3(count), dim3(LL_BLOCK), 0, st, V, first, count, 2304);
    ll_prof_mark(prof, LL_K_END, st);
}
