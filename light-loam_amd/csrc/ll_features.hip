/*
 * ll_features.hip -- a4: the less-flat VoxelGrid of a ring and the placement of the ring's four feature lists.
 * Replaces scanRegistration.cpp:361-376 of /root/reference (curvature + the greedy pick, :225-359, are the kernel before this one:
 * ll_pick.hip hands over the picked points' local indices, 352 bytes per ring).
 *
 * One 256-thread workgroup per (scan, ring):
 *   lists    the pick's lists of this ring -> LDS; the less-sharp picks (label 1 / 2) -> a bitmap: less-flat = every segment point that
 *            is not one of them (:361-367)
 *   keys     the ring's points are loaded in the sort's own (wave, row, lane) order, record g = point g: every load instruction covers
 *            eight whole 128-byte lines, no compaction scan; the picked points of the three small lists are fetched right behind them
 *            and parked in LDS; bounding box, PCL's voxel index (a pick, and the five points in front of the first segment, carry the
 *            all-ones key of the last row's padding and sort behind the real records)
 *   sort     stable LSD radix sort by (voxel, input order): ceil(bits / 8) passes of 5..8-bit digits; wave-private counters; a record's
 *            rank = what one returning LDS add per row hands back (lane order of such an add: checked on the device by ll_create);
 *            every wave scans the bases of its own counters; three barriers per pass
 *   sums     voxel runs summed left to right in f32 by the thread that owns the run head (PCL sorts by voxel only; oracle and HIP path
 *            define input order inside a voxel), continued from the next lane by a one-lane wave shift and beyond it from the cloud,
 *            four points at a time, / float(n)
 *   outputs  NO hand-over between the rings of a scan (rounds 1-4 passed the rings' counts forward through a decoupled look-back:
 *            polling, an in-order-dispatch assumption, a time-out status).  The less-flat centroids go to the ring's OWN row of the
 *            ring-strided cloud -- lflat[slot][ring][ring_cap] -- with the count in ring_nlf[slot][ring]; the three small lists go to
 *            their place in the contiguous published clouds, whose offsets are sums over the earlier rings of ring_cnt: written by
 *            k_ring_pick, i.e. complete before this launch starts.  k_build_grid, the first reader, turns the 64 less-flat counts
 *            into the prefix table lf_pre and the scan's totals; readers of the less-flat cloud address it as (ring, place in the
 *            ring) -- see ll_common.h "feature cloud layout".
 * Capacity tiers (ll_launch_features): rings of at most 2304 points -- every ring of a 2048-column sensor -- run the 9-row instantiation
 * at seven workgroups per CU whatever max_ring_points is; longer rings (a real HDL-64E under the linear 64-ring model puts two lasers
 * into some bins) run in the 12- / 18- / 32-row instantiations.  With no hand-over the tiers are independent launches in any order.
 * HBM traffic per ring point: 16 B read (+ the gather's second read behind the sort), 16 B per feature written.
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

#ifndef LL_FWAVES_SPLIT
#define LL_FWAVES_SPLIT 7     /* workgroups per CU (= waves per SIMD) of the 9-row instantiation: 66 registers, no scratch, 19-22 KB of LDS.  Six are
                               * 4 % slower, eight (64 registers, fits at the S64 capacity) 3 % slower: DESIGN.md 14.7 */
#endif
#define LL_NLIST 176      /* per segment slots: sharp[6][2] lsharp[6][20] flat[6][4] + counters[6][3] */

struct FeatLds {
    unsigned *k32;             /* [mr] sort key (voxel index) */
    unsigned short *k16;       /* [mr] sort payload (local index) */
    unsigned *picked;          /* bitmap over local index: the less-sharp picks */
    int *lists;                /* [LL_NLIST] */
    int *cnt;                  /* [4 waves][256] radix counters */
    int *sc;                   /* [64] scan scratch [0..15], the waves' run-head totals [20..23], per-wave bounds [32..55] */
    float4 *stash;             /* [LL_NSTASH] the picked points of the three small lists, entry t of each list in its owner thread's slot */
};
#define LL_NSTASH (LL_SEGS * (LL_SHARP_PER_SEG + LL_LSHARP_PER_SEG + LL_FLAT_PER_SEG))     /* 156 */


size_t ll_features_lds_bytes(int max_ring)      /* max_ring: the ring capacity of the LAUNCH (a tier of ll_launch_features) */
{
    const size_t mr = (size_t)((max_ring + 255) / 256 * 256);
    size_t b = 4 * mr + 2 * mr;              /* k32 + k16 */
    b += 4 * (LL_BLOCK / 64 * 256 + 4);      /* radix counters */
    b += 4 * (mr / 32 + 2);                  /* bitmap */
    b += 4 * LL_NLIST;
    b += 4 * 64;
    b += 16 * LL_NSTASH;
    return (b + 15) / 16 * 16 + 64;
}

__device__ __forceinline__ FeatLds ll_carve(unsigned char *base, int max_ring)
{
    const size_t mr = (size_t)((max_ring + 255) / 256 * 256);
    FeatLds L;
    unsigned char *p = base;
    L.stash = (float4 *)p; p += 16 * LL_NSTASH;
    L.k32 = (unsigned *)p; p += 4 * mr;
    L.cnt = (int *)p; p += 4 * (LL_BLOCK / 64 * 256 + 4);
    L.picked = (unsigned *)p; p += 4 * (mr / 32 + 2);
    L.lists = (int *)p; p += 4 * LL_NLIST;
    L.sc = (int *)p; p += 4 * 64;
    L.k16 = (unsigned short *)p;
    return L;
}

/* Stable least-significant-digit radix sort of the records (k32[g], k16[g]), g < n <= 256*ROWS, by k32; one workgroup.
 * Record g lives in wave g / (64 * nrows), row (g / 64) % nrows, lane g % 64 (nrows = rows per wave), so that the order
 * (wave, row, lane) is the input order.  Its destination is
 *   #(records with a smaller digit) + #(same digit, earlier wave) + #(same digit, same wave, earlier row) + rank in its row:
 * every wave counts into its own 2^BITS counters with ONE returning LDS add per row (a wave's LDS operations execute in order, so
 * the value a row gets back holds the rows before it, and the lanes of one add that hit the same counter are served in lane order:
 * the value is also the rank inside the row), and every wave turns the counters of the digit-major (digit, wave) table into the
 * bases of its OWN counters by an exclusive scan it does alone.  Wave-private counters (at a fixed stride of 256, cleared by their
 * wave) make the table small enough for digits of up to 8 bits: the keys are below 2^key_bits (the caller knows: the voxel grid's
 * dimensions), which takes ceil(key_bits / 8) passes of 5..8-bit digits (three for the usual 17..24 bits of a ring's voxel indices).
 * The last row is padded with all-ones keys that take part like records (they stay at the end, no per-record guards);
 * rows beyond it are skipped.  Keys must be < 0xffffffff.  Stability makes the result ordered by (k32, original
 * position): exactly the (voxel, input order) order the oracle defines. */
template <int ROWS, bool PRELOADED = false, bool MATCH = false>
__device__ __forceinline__ void ll_radix_sort(unsigned *k32, unsigned short *k16, int n, int key_bits, int *cnt, int tid,
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
    /* Wave-private counters at a fixed stride of 256, zero whenever a pass begins: a wave clears its own table (before the first pass, and behind
     * its own scatter in every pass -- its LDS instructions execute in order, nobody else touches the table then). */
    int *wc = cnt + wave * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) wc[j * 64 + lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    auto pass = [&](auto bits_tag, int sh) __attribute__((always_inline)) {
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr int ND = 1 << BITS;
        constexpr unsigned DM = (unsigned)ND - 1u;
        constexpr int NJ = (ND + 63) / 64;                      /* digits per lane */
        int rnk[ROWS];
        /* ONE returning LDS add per row: the value that comes back is (records of this wave's earlier rows with digit d) + (lower lanes of
         * this row with digit d) -- a wave's LDS instructions execute in order, and the lanes of one ds_add_rtn that hit the same counter
         * are served in ascending lane order.  The ISA manual does not promise the second; ll_create checks it on the device
         * (ll_lds_atomic_order_ok, tools/ubench/lds_atomic_order.hip: 9.4e8 lanes, none out of order) and selects the match-any
         * ranking below (MATCH) for a device where it does not hold.  The nine adds of a thread go out back to back and are waited for once, behind the counter scan.
         * (Rounds 2-5 took the rank from a match-any, four vector instructions per digit bit and row: a third of the kernel's vector work.) */
        if constexpr (!MATCH) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                rnk[k] = 0;
                if (k < myrows) rnk[k] = __hip_atomic_fetch_add(&wc[(int)((e32[k] >> sh) & DM)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            /* The ranking that needs no promise from the LDS arbiter (the sort of rounds 2-5; ll_create selects it for a device that
             * fails the lane-order check, ll_params.voxel_sort_ranks = 1 forces it): the rank inside the row from a match-any of the
             * digit (four vector instructions per digit bit), the earlier rows' count by a read of the wave's counter, which the
             * first lane of every digit group then bumps by the group's size -- a wave's LDS operations execute in order, so the next
             * row's read sees it.  Same destinations, bit for bit; ~0.7 % of the kernel's time slower on MI355X (round 5, A/B). */
            int pre[ROWS];
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                rnk[k] = 0; pre[k] = 0;
                if (k < myrows) {
                    const int d = (int)((e32[k] >> sh) & DM);
                    unsigned mlo, mhi;
                    ll_match_any(d, BITS, ~0ull, mlo, mhi);
                    rnk[k] = ll_match_rank(mlo, mhi);
                    pre[k] = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   /* an atomic load: it must see the rows' adds */
                    if (rnk[k] == 0) atomicAdd(&wc[d], ll_match_count(mlo, mhi));
                }
            }
#pragma unroll
            for (int k = 0; k < ROWS; ++k) rnk[k] += pre[k];
        }
        __syncthreads();                                        /* every wave's counts are complete */
        /* Every wave works out the bases of ITS OWN counters by itself -- exclusive scan over the (digit, wave) table in digit-major order;
         * lane l holds the digits l, l + 64, ... -- instead of one workgroup-wide scan (three barriers, two trips through LDS). */
        int own[NJ];
        {
            int carry = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = j * 64 + lane;
                int tot = 0, below = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) { const int c = (d < ND) ? cnt[w * 256 + d] : 0; tot += c; below += (w < wave) ? c : 0; }
                const int inc = ll_wave_incl_scan(tot);
                own[j] = carry + inc - tot + below;
                carry += __builtin_amdgcn_readlane(inc, 63);
            }
        }
        __syncthreads();                                        /* every wave has read the counts: the bases may overwrite them */
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (j * 64 + lane < ND) wc[j * 64 + lane] = own[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            if (k < myrows) {
                const int d = (int)((e32[k] >> sh) & DM);
                const int pos = wc[d] + rnk[k];
                k32[pos] = e32[k]; k16[pos] = e16[k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (j * 64 + lane < ND) wc[j * 64 + lane] = 0;       /* ready for the next pass */
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

/* The property the sort's ranks rest on, checked once per device by ll_create: within one returning LDS add, lanes that hit the same counter get
 * their values in ascending lane order (and a later instruction of the wave sees the earlier one's adds).  Digit patterns: uniform over 2^b
 * values, b = 0..8, as they lie and scattered over the lanes.  out[0] += lanes that disagreed with a match-any count. */
__global__ __launch_bounds__(256, LL_FWAVES_SPLIT) void k_lds_atomic_order(int iters, unsigned *out)
{
    __shared__ int cnt[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int b = it % 9;
        int d = (int)(rnd() & ((1u << b) - 1u));
        if (it & 1) d = __shfl(d, (lane * 7 + it) & 63);
        const int d2 = (int)(rnd() & ((1u << b) - 1u));
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int got = __hip_atomic_fetch_add(&cnt[wave][d], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int got2 = __hip_atomic_fetch_add(&cnt[wave][d2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned mlo, mhi;
        ll_match_any(d, 8, ~0ull, mlo, mhi);
        const int want = ll_match_rank(mlo, mhi);
        int want2 = 0;
        for (int l = 0; l < 64; ++l) { const int dl = __shfl(d, l), dl2 = __shfl(d2, l); want2 += (dl == d2) + (l < lane && dl2 == d2); }
        bad += (got != want) + (got2 != want2);
    }
    if (bad) atomicAdd(out, bad);
}
/* 1: the property holds, 0: it does not, -1: the check could not run (allocation / launch failure: nothing learnt about the device).
 * Launched at the sort kernel's own residency and beyond -- eight 256-thread workgroups per CU, every wave of the chip adding into
 * its private table at once, as k_ring_features<9> runs at seven -- so that the arbiter is asked under the contention it works under. */
int ll_lds_atomic_order_ok(hipStream_t st)
{
    unsigned *d = nullptr, h = 1;
    if (hipMalloc(&d, sizeof(unsigned)) != hipSuccess) return -1;
    bool ran = hipMemsetAsync(d, 0, sizeof(unsigned), st) == hipSuccess;
    if (ran) {
        hipLaunchKernelGGL(k_lds_atomic_order, dim3(2048), dim3(256), 0, st, 45, d);
        ran = hipGetLastError() == hipSuccess && hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipStreamSynchronize(st) == hipSuccess;
    }
    (void)hipFree(d);
    return ran ? (h == 0 ? 1 : 0) : -1;
}

__device__ __forceinline__ bool ll_bit(const unsigned *bm, int i) { return (bm[i >> 5] >> (i & 31)) & 1u; }

extern __shared__ __attribute__((aligned(16))) unsigned char ll_smem[];

/* opt-in phase timing (tools/phase_timing.py builds with -DLL_PHASE_TIMING): thread 0 of every workgroup adds the
 * s_memtime cycles spent in each phase to V.dbg[phase]; V.dbg[15] counts workgroups */
#ifdef LL_PHASE_TIMING
#define LL_PHASE_BEGIN() long long ll_t0 = (tid == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0
#define LL_PHASE(i) do { __syncthreads(); if (tid == 0) { const long long t1 = (long long)__builtin_amdgcn_s_memtime(); \
    atomicAdd(&V.dbg[i], (unsigned long long)(t1 - ll_t0)); ll_t0 = t1; } } while (0)
#elif defined(LL_PHASE_STOP)   /* tools/phase_valu.py: the kernel returns after phase LL_PHASE_STOP (instruction counts per phase by difference) */
#define LL_PHASE_BEGIN() do {} while (0)
#define LL_PHASE(i) do { if ((i) == LL_PHASE_STOP) return; } while (0)
#else
#define LL_PHASE_BEGIN() do {} while (0)
#define LL_PHASE(i) do {} while (0)
#endif

/* One ring.  (The kernel's 2nd launch bound = waves per SIMD: seven 256-thread workgroups per CU for the common 2304-point capacity, <= 72 VGPRs.)
 * ring_lo < ring length <= ring_hi: the rings this launch works on (ring_hi, a multiple of 256 <= 256 ROWS, also sizes its LDS). */
template <int ROWS, bool MATCH>
__device__ __forceinline__ void ll_ring_features_ring(const LLView &V, int s, int r, int ring_lo, int ring_hi)
{
    static_assert(ROWS <= 32, "headm / endm hold one bit per row of a thread");
    const int tid = threadIdx.x, lane = tid & 63;
    /* Everything the set-up needs from memory depends on (slot, ring) only: the earlier rings' pick counts (k_ring_pick's ring_cnt:
     * complete, that launch is over -- in every wave lane q holds ring q's; their sums over q < r are this ring's offsets in the three contiguous
     * small clouds: scanRegistration.cpp:273-279, :325), this ring's lists, the header's status and the ring's two offsets.  All of it
     * is requested before anything is waited for, and the three exits are ONE test -- as separate tests each exit put a memory round
     * trip of its own in front of the next load (four dependent trips at the head of a 23 us workgroup). */
    const unsigned cnt_raw0 = lane < r ? V.ring_cnt[(size_t)s * V.R + lane] : 0u;          /* every wave holds all the rings' counts: lane q has ring q's */
    const unsigned cnt_raw1 = lane + 64 < r ? V.ring_cnt[(size_t)s * V.R + lane + 64] : 0u;  /* ... and ring q + 64's (LL_MAX_RINGS = 128) */
    const unsigned short *rec = V.ring_rec + ((size_t)s * V.R + r) * LL_REC_U16;
    const unsigned short rec_v = tid < 174 ? rec[tid] : (unsigned short)0;
    const int status = V.hdr[s].status;
    const int off = V.ring_off[(size_t)s * (V.R + 1) + r];
    const int nr = V.ring_off[(size_t)s * (V.R + 1) + r + 1] - off;
    if ((status != 0) | (nr <= ring_lo) | (nr > ring_hi)) return;    /* a refused scan | another tier's ring */
    int *ring_nlf = V.ring_nlf + (size_t)s * V.R;
    if (nr <= 0) { if (tid == 0) ring_nlf[r] = 0; return; }          /* an empty ring: no segment, no list (k_ring_pick wrote ring_cnt = 0) */
    const int S = off + 5, E = off + nr - 6;                          /* scanStartInd / scanEndInd (:218-220) */
    const bool active = (E - S >= 6);                                 /* :248 */
    const int Lseg = active ? (E - S) : 0;                            /* indices S .. E-1 are in segments */
    /* the ring's points by local index; laserCloud index g of the reference = off + local index */
    const float4 *cloud = V.cloud + (size_t)s * V.CS + (size_t)r * V.ring_cap - off;          /* cloud[off + li] = the ring's point li */
    FeatLds L = ll_carve(ll_smem, ring_hi);
    float *fs = (float *)(L.sc + 32);                                 /* 24 floats: per-wave bounds */
    LL_PHASE_BEGIN();
    /* the pick's lists of this ring: local indices + per-segment counts -> L.lists; the less-sharp picks (label 1 / 2) -> bitmap:
     * less-flat = every segment point that is not one of them (:361-367) */
    if (tid < 174) L.lists[tid] = (int)rec_v;
    { const int nwords = (nr + 31) / 32 + 1; for (int i = tid; i < nwords; i += LL_BLOCK) L.picked[i] = 0; }
    __syncthreads();
    if (tid < LL_SEGS * LL_LSHARP_PER_SEG && tid % LL_LSHARP_PER_SEG < L.lists[157 + (tid / LL_LSHARP_PER_SEG) * 3]) {
        const int li = L.lists[12 + tid];
        atomicOr(&L.picked[li >> 5], 1u << (li & 31));
    }
    __syncthreads();

    /* the picked points of the per-segment lists: thread t owns entry t of each list; its position
     * (segment, pick order) among the ring's sharp / less-sharp / flat points, -1 = no entry */
    int fpos[3] = {-1, -1, -1};
    auto gather_lists = [&]() __attribute__((always_inline)) {
        const int js = tid / LL_SHARP_PER_SEG, jl = tid / LL_LSHARP_PER_SEG, jf = tid / LL_FLAT_PER_SEG;
        int os = 0, ol = 0, of = 0;
        for (int j = 0; j < LL_SEGS; ++j) {
            if (j < js) os += L.lists[156 + j * 3];
            if (j < jl) ol += L.lists[157 + j * 3];
            if (j < jf) of += L.lists[158 + j * 3];
        }
        if (js < LL_SEGS && tid % LL_SHARP_PER_SEG < L.lists[156 + js * 3]) fpos[0] = os + tid % LL_SHARP_PER_SEG;
        if (jl < LL_SEGS && tid % LL_LSHARP_PER_SEG < L.lists[157 + jl * 3]) fpos[1] = ol + tid % LL_LSHARP_PER_SEG;
        if (jf < LL_SEGS && tid % LL_FLAT_PER_SEG < L.lists[158 + jf * 3]) fpos[2] = of + tid % LL_FLAT_PER_SEG;
    };
    /* this ring's offsets in sharp / less-sharp / flat: the sums of the earlier rings' counts (no waiting: the counts came in with the header) */
    int roff[3] = {0, 0, 0};
    auto small_offsets = [&]() __attribute__((always_inline)) {
        static_assert(LL_MAX_RINGS <= 128, "two counts per lane");
        int v[3] = {(int)(cnt_raw0 & 0xffu) + (int)(cnt_raw1 & 0xffu), (int)((cnt_raw0 >> 8) & 0xffu) + (int)((cnt_raw1 >> 8) & 0xffu),
                    (int)((cnt_raw0 >> 16) & 0xffu) + (int)((cnt_raw1 >> 16) & 0xffu)};
#pragma unroll
        for (int c = 0; c < 3; ++c) roff[c] = ll_wave_sum_i32(v[c]);      /* a wave-wide sum in every wave: no barrier, no LDS */
    };

    /* ---------------- VoxelGrid of the less-flat points (:361-376) ---------------- */
    int n_lf_out = 0;
    float4 *out = V.lflat + (size_t)s * V.LFS + (size_t)r * V.ring_cap;          /* this ring's row of the ring-strided less-flat cloud */
    bool offsets_done = false;
    if (active) {
        int m = 0;
        bool sorted_ok = false;
        {
            /* The segment points go to the sort as they lie, in the sort's own (wave, row, lane) order -- coalesced rows, no compaction
             * scan, no trip through LDS: a less-sharp pick (not less-flat, :361-367) just carries the all-ones key that the padding
             * of the last row carries and ends up behind the m real records. */
            /* Record g IS the ring's point g (local index): the five points in front of the first segment are records that never become
             * less-flat.  A wave's load instruction then covers 64 points from a multiple of 64 = eight whole 128-byte lines; with the
             * records starting at the first segment point (80 bytes into a line) every instruction straddled a ninth line that the next
             * one asked for again -- and this chip's L2 sends EVERY request for a line that is still on its way across the fabric
             * (profiles/r05_fetch_granule.txt): 14 % more lines fetched than the ring has. */
            const int nrec = Lseg + 5;
            const int nrows_w = (nrec + LL_BLOCK - 1) / LL_BLOCK;                   /* rows per wave, uniform */
            const int wbase = __builtin_amdgcn_readfirstlane(tid >> 6) * nrows_w * 64;
            float px[ROWS], py[ROWS], pz[ROWS];
            unsigned mem = 0;                                                        /* bit k: record wbase + k * 64 + lane is less-flat */
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int g = wbase + k * 64 + lane;
                if (k < nrows_w && g >= 5 && g < nrec) {
                    const float4 p = cloud[off + g];
                    px[k] = p.x; py[k] = p.y; pz[k] = p.z;
                    if (!ll_bit(L.picked, g)) mem |= 1u << k;
                }
            }
            /* The picked points of the three small lists are fetched HERE, behind the row's own loads: their lines are on their way into the
             * L2 (every pick is a segment point) -- at the end of the workgroup, ~20 us later, half of them had left it again and came
             * over the fabric a third time.  Entry t of each list belongs to thread t from here to its store: parked in LDS meanwhile. */
            const bool hv0 = tid < LL_SEGS * LL_SHARP_PER_SEG && tid % LL_SHARP_PER_SEG < L.lists[156 + (tid / LL_SHARP_PER_SEG) * 3];
            const bool hv1 = tid < LL_SEGS * LL_LSHARP_PER_SEG && tid % LL_LSHARP_PER_SEG < L.lists[157 + (tid / LL_LSHARP_PER_SEG) * 3];
            const bool hv2 = tid < LL_SEGS * LL_FLAT_PER_SEG && tid % LL_FLAT_PER_SEG < L.lists[158 + (tid / LL_FLAT_PER_SEG) * 3];
            float4 q0, q1, q2;
            if (hv0) q0 = cloud[off + L.lists[tid]];
            if (hv1) q1 = cloud[off + L.lists[12 + tid]];
            if (hv2) q2 = cloud[off + L.lists[132 + tid]];
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
            if (hv0) L.stash[tid] = q0;
            if (hv1) L.stash[12 + tid] = q1;
            if (hv2) L.stash[132 + tid] = q2;
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
                 * padding.  The CPU checker takes the same exit. */
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
                        e16[k] = (unsigned short)(wbase + k * 64 + lane);           /* payload: local index */
                    }
                }
                LL_PHASE(3);
                ll_radix_sort<ROWS, true, MATCH>(L.k32, L.k16, nrec, key_bits, L.cnt, tid, e32, e16);
                LL_PHASE(4);
                __syncthreads();
#ifdef LL_SORT_CHECK    /* -DLL_SORT_CHECK (tools/soak_extract*.py with LL_SORT_CHECK_LIB): the sort's result IS ordered by (voxel, input order) -- the property
                         * the ranking from returning LDS adds rests on, asserted on every ring of a soak instead of inferred from equal clouds.  V.dbg[9] counts
                         * adjacent pairs out of order, V.dbg[10] the pairs looked at. */
                {
                    unsigned bad = 0, seen = 0;
                    for (int i = tid; i + 1 < m; i += LL_BLOCK) {
                        const unsigned ka = L.k32[i], kb = L.k32[i + 1];
                        bad += (ka > kb || (ka == kb && L.k16[i] >= L.k16[i + 1])) ? 1u : 0u; ++seen;
                    }
                    if (bad) atomicAdd(&V.dbg[9], (unsigned long long)bad);
                    atomicAdd(&V.dbg[10], (unsigned long long)seen);
                }
#endif
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
            /* places of the thread's centroids in the ring's row: prefix over the run heads (wave scan + the earlier waves' totals through slots
             * of their own, written here and nowhere else: one barrier, none behind the reads) */
            int o;
            {
                const int nh = __popc(headm), inc = ll_wave_incl_scan(nh);
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                if (lane == 63) L.sc[20 + wv] = inc;
                __syncthreads();
                int before = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < LL_BLOCK / 64; ++w) { const int t = L.sc[20 + w]; before += (w < wv) ? t : 0; tot += t; }
                o = before + inc - nh; n_lf_out = tot;
            }
            LL_PHASE(12);
            /* CentroidPoint<PointXYZI>: f32 sums from zero in input order, divided by float(n).  A run that ends inside the
             * thread's range leaves its centroid in the registers of its last point (bit u of endm) */
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
                /* the centroids of the runs that ended inside the range leave now, in voxel order, for the ring's own row (:376: lessFlatScanDS
                 * appended ring after ring): their registers are free for the points the open run still has to fetch */
#pragma unroll
                for (int u = 0; u < ROWS; ++u) if (u < perm && ((endm >> u) & 1u)) out[o++] = pt[u];
                float4 last = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (cn) {
                    /* four at a time: the keys and indices of the next four sorted places (LDS), the points of those that still belong to
                     * the run requested TOGETHER, added in order.  (One by one, each point was a dependent LDS -> memory round trip; a ring
                     * that sweeps the ground close to the sensor puts ten points into a voxel, its runs span several threads, and the
                     * workgroup waited at the next barrier for the longest of these chains: a fifth of its time.) */
                    const unsigned vid = L.k32[b1 - 1];
                    for (int e = b1 + c_n; e < m; e += 4) {
                        unsigned kk[4]; int ii[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { kk[u] = (e + u < m) ? L.k32[e + u] : ~vid; ii[u] = (e + u < m) ? (int)L.k16[e + u] : 0; }
                        const int n = (kk[0] != vid) ? 0 : (kk[1] != vid) ? 1 : (kk[2] != vid) ? 2 : (kk[3] != vid) ? 3 : 4;
                        float4 q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (u < n) q[u] = cloud[off + ii[u]];
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (u < n) { sx += q[u].x; sy += q[u].y; sz += q[u].z; si += q[u].w; ++cn; }
                        if (n < 4) break;
                    }
                    const float fn = (float)cn; last = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
                }
                LL_PHASE(5);
                if (cn) out[o] = last;
                gather_lists(); small_offsets();
                offsets_done = true;
                LL_PHASE(7);
            }
        }
    }
    if (tid == 0) ring_nlf[r] = n_lf_out;

    /* ---------------- the three small lists -> the published clouds at this ring's offsets, in (segment, pick order) ---------------- */
    if (!offsets_done) { gather_lists(); small_offsets(); }              /* rings without a less-flat point */
    {   /* the points were parked in LDS by this thread when the ring was read (a list has entries only where the ring has a segment) */
        if (fpos[0] >= 0) V.sharp[(size_t)s * V.cap_sharp + roff[0] + fpos[0]] = L.stash[tid];
        if (fpos[1] >= 0) V.lsharp[(size_t)s * V.cap_lsharp + roff[1] + fpos[1]] = L.stash[12 + tid];
        if (fpos[2] >= 0) V.flat[(size_t)s * V.cap_flat + roff[2] + fpos[2]] = L.stash[132 + tid];
    }
    LL_PHASE(6);
#ifdef LL_PHASE_TIMING
    if (tid == 0) atomicAdd(&V.dbg[15], 1ull);
#endif
}

/* The launch of the common capacity: one workgroup per (scan, ring).  A tier of longer rings runs over the work list k_organize filled
 * for it (list != null: slot << 8 | ring, *list_n entries) with a fixed grid of resident workgroups taking entries in turn. */
template <int ROWS, bool MATCH>
__global__ __launch_bounds__(LL_BLOCK, (ROWS <= 9 ? LL_FWAVES_SPLIT : ROWS <= 12 ? 4 : ROWS <= 18 ? 3 : 1)) void k_ring_features(LLView V, int first, int count, int ring_lo, int ring_hi,
                                                                                                                              const int *list, const int *list_n)
{
    if constexpr (ROWS > 9) {                                         /* a tier: always over its list (one body per kernel: the register allocation of
                                                                       * the main instantiation must not pay for a loop it never runs) */
        const int n = *list_n;
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
            const int e = list[i];
            ll_ring_features_ring<ROWS, MATCH>(V, e >> 8, e & 0xFF, ring_lo, ring_hi);
            __syncthreads();                                          /* the next ring's set-up overwrites what this one's last reads used */
        }
    } else {
        int sl, r;
        if (!ll_xcd_map2(blockIdx.x, V.R, count, sl, r)) return;
        ll_ring_features_ring<ROWS, MATCH>(V, first + sl, r, ring_lo, ring_hi);
    }
}

template <int ROWS, bool MATCH>
static void ll_launch_ring_features_as(const LLView &V, int first, int count, int grid, int ring_lo, int ring_hi, int tier, int wg_per_cu, hipStream_t st)
{
    static size_t attr_bytes[LL_MAX_DEVICES] = {0};
    const size_t lds_bytes = ll_features_lds_bytes(ring_hi);
    ll_ensure_dynamic_lds(k_ring_features<ROWS, MATCH>, lds_bytes, attr_bytes);
    const int *list = nullptr, *list_n = nullptr;
    if (tier > 0) {
        list = V.tier_list + (size_t)(tier - 1) * V.B * V.R; list_n = V.tier_cnt + tier;
        if (grid > 256 * wg_per_cu) grid = 256 * wg_per_cu;
    }
    hipLaunchKernelGGL((k_ring_features<ROWS, MATCH>), dim3(grid), dim3(LL_BLOCK), lds_bytes, st, V, first, count, ring_lo, ring_hi, list, list_n);
}
/* V.sort_match_any: the sort's ranking that does not lean on the LDS arbiter's lane order (ll_create: the device failed the check, or
 * ll_params.voxel_sort_ranks asked for it) */
template <int ROWS>
static void ll_launch_ring_features(const LLView &V, int first, int count, int grid, int ring_lo, int ring_hi, int tier, int wg_per_cu, hipStream_t st)
{
    if (V.sort_match_any) ll_launch_ring_features_as<ROWS, true>(V, first, count, grid, ring_lo, ring_hi, tier, wg_per_cu, st);
    else ll_launch_ring_features_as<ROWS, false>(V, first, count, grid, ring_lo, ring_hi, tier, wg_per_cu, st);
}

void ll_launch_features(const LLView &V, int first, int count, size_t /* LDS of the largest tier: checked by ll_create */, hipStream_t st, LLProfiler *prof, int what)
{
    const int groups = (count + 7) / 8;
    const int grid = 8 * V.R * groups;
    const int cap = (V.max_ring + 255) / 256 * 256;
    if (what & 1) {
        ll_prof_mark(prof, LL_K_PICK, st);
        ll_launch_pick(V, first, count, st);
    }
    if (!(what & 2)) { ll_prof_mark(prof, LL_K_END, st); return; }
    ll_prof_mark(prof, LL_K_RING_FEATURES, st);
    /* one launch per capacity tier; every ring is worked on by exactly one of them, none waits for another */
    if (cap > 4608) ll_launch_ring_features<32>(V, first, count, grid, 4608, cap, 3, 1, st);   /* <= 8192 points: the per-thread row masks are 32 bits wide */
    if (cap > 3072) ll_launch_ring_features<18>(V, first, count, grid, 3072, cap < 4608 ? cap : 4608, 2, 3, st);
    if (cap > 2304) ll_launch_ring_features<12>(V, first, count, grid, 2304, cap < 3072 ? cap : 3072, 1, 4, st);   /* two lasers of a 64-beam sensor in one bin: four resident workgroups per CU (no scratch at 128 registers) */
    ll_launch_ring_features<9>(V, first, count, grid, INT_MIN, cap < 2304 ? cap : 2304, 0, LL_FWAVES_SPLIT, st);
    ll_prof_mark(prof, LL_K_END, st);
}
