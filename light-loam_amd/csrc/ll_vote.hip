/*
 * ll_vote.hip -- correspondence compaction + a8: graph_based_correspondence_vote_simple.
 * Replaces laserOdometry.cpp:153-342 (call site :796) of /root/reference, plus the push_back order of the
 * correspondence lists (:574-585, :745-757).
 *
 * One workgroup per scan pair.  Per-query association results are compacted in query order (the order the
 * reference appends them).  The vote is the reference's dense all-pairs test inside each of 10 (planes) or 5
 * (corners) contiguous regions: every pair (i, j) of a region with | |src_i-src_j| - |tgt_i-tgt_j| |^2 >= T counts for both,
 * where "std::exp(-gap^2) < 0.96f" is replaced by the bit-exact threshold of ll_exact_math.h.  src/tgt points are
 * staged once in LDS and broadcast-read.  Selection: count <= 0.9f*m, weight 5 if count <= 50 else 1 (:299-322).
 * Output is per correspondence (count, selected, weight); the reference's output ORDER (ascending count, std::sort
 * ties unspecified) only permutes residual blocks and is reproduced on the host side where a caller needs it
 * (include/lightloam_host.hpp runs the same std::sort on the counts).
 */
#include "ll_common.h"

extern __shared__ __attribute__((aligned(16))) unsigned char ll_vsm[];

/* stable compaction of valid_src[0..n) >= 0 by one workgroup; emit(i, pos) for every kept i; returns the total */
template <int NT, typename F>
__device__ __forceinline__ int ll_block_compact(int n, int *sc, F emit, const int *valid_src)
{
    const int tid = threadIdx.x;
    const int per = (n + NT - 1) / NT;
    const int a0 = min(n, tid * per), a1 = min(n, a0 + per);
    int c = 0;
    for (int i = a0; i < a1; ++i) c += valid_src[i] >= 0;
    int total = 0;
    int pos = ll_block_exscan_n<NT / 64>(c, sc, total);
    for (int i = a0; i < a1; ++i) if (valid_src[i] >= 0) emit(i, pos++);
    __syncthreads();
    return total;
}

/* the vote proper on LDS-staged records S3[6 * i .. 6 * i + 5] = (src xyz, tgt xyz) of correspondence i (T3 = the end of the
 * records); returns this thread's number of selected entries.
 * Every unordered pair of a region is evaluated ONCE, like the reference's i < j loop (:228-252): entry i of a region of m
 * takes the partners at circular offsets 1 .. (m - 1) / 2 (and, for even m, the lower half takes offset m / 2), which
 * gives every lane the same trip count; an incompatible pair bumps the entry's own register count and the partner's
 * count in LDS (cntL, n_p ints, zeroed here).  Counts are integers, so the summation order does not matter. */
template <int NT>
__device__ __forceinline__ int ll_vote_core(const float *S3, const float *T3, int n_p, int number_of_region, int enable,
                                            int *vc, uint8_t *vs, float *vw, int *cntL)
{
    const int chunk = n_p / number_of_region;                     /* cor_size_all / number_of_region (:202) */
    for (int i = threadIdx.x; i < n_p; i += NT) cntL[i] = 0;
    __syncthreads();
    if (enable) {
        for (int i = threadIdx.x; i < n_p; i += NT) {
            const int rg = (chunk > 0) ? min(i / chunk, number_of_region - 1) : number_of_region - 1;
            const int b0 = chunk * rg, b1 = (rg == number_of_region - 1) ? n_p : chunk * (rg + 1);
            const int m = b1 - b0;
            const float ax = S3[6 * i], ay = S3[6 * i + 1], az = S3[6 * i + 2];
            const float bx = S3[6 * i + 3], by = S3[6 * i + 4], bz = S3[6 * i + 5];
            const int half = (m - 1) / 2;
            const int nd = half + (((m & 1) == 0 && (i - b0) < m / 2) ? 1 : 0);        /* even m: offset m / 2 once per pair */
            int cnt = 0;
            int j = i;
            for (int d = 1; d <= nd; ++d) {
                ++j; if (j >= b1) j -= m;
                /* Distance() (:153-162): f32 sqrt of dx*dx + dy*dy + dz*dz; the squares make the operand order irrelevant
                 * bit-for-bit, so (i, j) and (j, i) are the same test */
                const float2 *pj = (const float2 *)(S3 + 6 * j);                  /* one record: three 8-byte reads */
                const float2 p0 = pj[0], p1 = pj[1], p2 = pj[2];
                float dx = ax - p0.x, dy = ay - p0.y, dz = az - p1.x;
                const float a2 = dx * dx + dy * dy + dz * dz;
                dx = bx - p1.y; dy = by - p2.x; dz = bz - p2.y;
                const float b2 = dx * dx + dy * dy + dz * dz;
                /* The reference's test is gap >= g_T on correctly rounded square roots (ll_vote_incompatible_gap).  The
                 * hardware square root (v_sqrt_f32, ~1 ulp) decides it unless the approximate gap lies within
                 * (s1 + s2) * 2^-20 of g_T -- more than three times what two approximate roots (a few ulp each) and the
                 * roundings of the two subtractions can move it -- and only those pairs take the exact roots. */
                const float s1a = __builtin_amdgcn_sqrtf(a2), s2a = __builtin_amdgcn_sqrtf(b2);
                const float ga = fabsf(s1a - s2a), e2 = (s1a + s2a) * 9.5367431640625e-07f;
                const float gT = ll_u2f(LL_VOTE_GAP_BITS);
                bool inc = ga >= gT + e2;
                if (!inc && !(ga <= gT - e2)) {                                    /* too close to call (or not finite) */
                    const float s1 = sqrtf(a2), s2 = sqrtf(b2);
                    inc = ll_vote_incompatible_gap(fabsf(s1 - s2));
                }
                if (inc) { ++cnt; atomicAdd(&cntL[j], 1); }
            }
            if (cnt) atomicAdd(&cntL[i], cnt);
        }
    }
    __syncthreads();
    int my_sel = 0;
    for (int i = threadIdx.x; i < n_p; i += NT) {
        int cnt = 0, sel = 1; float w = 1.0f;
        if (enable) {
            const int rg = (chunk > 0) ? min(i / chunk, number_of_region - 1) : number_of_region - 1;
            const int b0 = chunk * rg, b1 = (rg == number_of_region - 1) ? n_p : chunk * (rg + 1);
            cnt = cntL[i];
            const float num_selected = 0.90f * (float)(b1 - b0);                 /* :299-300 */
            sel = !((float)cnt > num_selected);                                   /* :312 */
            w = ((float)cnt <= 50.0f) ? 5.0f : 1.0f;                              /* :317-322 */
        }
        vc[i] = cnt; vs[i] = (uint8_t)sel; vw[i] = w;
        my_sel += sel;
    }
    return my_sel;
}

/* LL_VT threads: the all-pairs loop is the serial work of a thread (72 k pair tests per scan pair: ~280 each with 256 threads);
 * 256 / 512 / 1024 threads: 0.80 / 0.63 / 1.14 ms per 8192 scan pairs */
#ifndef LL_VT
#define LL_VT 512
#endif
__global__ __launch_bounds__(LL_VT) void k_vote(LLView V, int first, int count, int enable)
{
    if ((int)blockIdx.x >= count) return;
    const int s = first + blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ int sc[LL_VT / 64 + 1];
    __shared__ int nsel_sh;
    const ScanHdr h = V.hdr[s];
    const bool ok = h.status == 0;
    const int ns = ok ? h.n_sharp : 0, nf = ok ? h.n_flat : 0;
    const float4 *corner, *surf; int mc, ms;
    ll_targets(V, s, corner, mc, surf, ms);

    /* edges (:574-617) */
    const int *eqa = V.eq_a + (size_t)s * V.cap_sharp, *eqb = V.eq_b + (size_t)s * V.cap_sharp;
    int *es = V.e_src + (size_t)s * V.cap_sharp, *ea = V.e_a + (size_t)s * V.cap_sharp, *eb = V.e_b + (size_t)s * V.cap_sharp;
    const int n_e = ll_block_compact<LL_VT>(ns, sc, [&](int i, int pos) { es[pos] = i; ea[pos] = eqa[i]; eb[pos] = eqb[i]; }, eqa);
    /* planes (:745-790) */
    const int *pqa = V.pq_a + (size_t)s * V.cap_flat, *pqb = V.pq_b + (size_t)s * V.cap_flat, *pqc = V.pq_c + (size_t)s * V.cap_flat;
    int *ps = V.p_src + (size_t)s * V.cap_flat, *pa = V.p_a + (size_t)s * V.cap_flat, *pb = V.p_b + (size_t)s * V.cap_flat, *pc = V.p_c + (size_t)s * V.cap_flat;
    const int n_p = ll_block_compact<LL_VT>(nf, sc, [&](int i, int pos) { ps[pos] = i; pa[pos] = pqa[i]; pb[pos] = pqb[i]; pc[pos] = pqc[i]; }, pqa);

    /* stage Corre_Match.src (raw current point, :753) and .tgt (closest target point, :754) */
    float *S3 = (float *)ll_vsm;                 /* [n_p][6]: src xyz, tgt xyz */
    float *T3 = S3 + 3 * (size_t)V.cap_flat;     /* S3 + 6 * cap_flat = T3 + 3 * cap_flat: the counts */
    const float4 *flat = V.flat + (size_t)s * V.cap_flat;
    if (tid == 0) nsel_sh = 0;
    __syncthreads();
    for (int i = tid; i < n_p; i += LL_VT) {
        const float4 a = flat[ps[i]], b = surf[pa[i]];
        S3[6 * i] = a.x; S3[6 * i + 1] = a.y; S3[6 * i + 2] = a.z;
        S3[6 * i + 3] = b.x; S3[6 * i + 4] = b.y; S3[6 * i + 5] = b.z;
    }
    __syncthreads();
    const int my_sel = ll_vote_core<LL_VT>(S3, T3, n_p, 10 /* plane case (:186-187) */, enable,
                                    V.v_count + (size_t)s * V.cap_flat, V.v_sel + (size_t)s * V.cap_flat, V.v_w + (size_t)s * V.cap_flat,
                                    (int *)(T3 + 3 * (size_t)V.cap_flat));
    if (my_sel) atomicAdd(&nsel_sh, my_sel);
    __syncthreads();
    if (tid == 0) {
        PairHdr p; p.n_edge = n_e; p.n_plane = n_p; p.n_plane_sel = nsel_sh; p.target_slot = ll_target_slot(V, s);
        V.pair[s] = p;
    }
}

/* the free function's own signature: caller-supplied correspondences (src / tgt points), 5 or 10 regions */
__global__ __launch_bounds__(LL_BLOCK) void k_vote_points(const float4 *src, const float4 *tgt, int n, int regions,
                                                          int *vc, uint8_t *vs, float *vw)
{
    float *S3 = (float *)ll_vsm, *T3 = S3 + 3 * (size_t)n;
    for (int i = threadIdx.x; i < n; i += LL_BLOCK) {
        const float4 a = src[i], b = tgt[i];
        S3[6 * i] = a.x; S3[6 * i + 1] = a.y; S3[6 * i + 2] = a.z;
        S3[6 * i + 3] = b.x; S3[6 * i + 4] = b.y; S3[6 * i + 5] = b.z;
    }
    __syncthreads();
    (void)ll_vote_core<LL_BLOCK>(S3, T3, n, regions, 1, vc, vs, vw, (int *)(T3 + 3 * (size_t)n));
}

void ll_launch_vote(const LLView &V, int first, int count, int enable, hipStream_t st, LLProfiler *prof)
{
    const size_t lds = (size_t)V.cap_flat * 28;                   /* src + tgt triples + one count per correspondence */
    static size_t attr_bytes[LL_MAX_DEVICES] = {0};
    ll_ensure_dynamic_lds(k_vote, lds, attr_bytes);
    ll_prof_mark(prof, LL_K_VOTE, st);
    hipLaunchKernelGGL(k_vote, dim3(count), dim3(LL_VT), lds, st, V, first, count, enable);
    ll_prof_mark(prof, LL_K_END, st);
}

void ll_launch_vote_points(const float4 *src, const float4 *tgt, int n, int regions, int *vc, uint8_t *vs, float *vw, hipStream_t st)
{
    const size_t lds = (size_t)n * 28 + 16;
    static size_t attr_bytes[LL_MAX_DEVICES] = {0};
    ll_ensure_dynamic_lds(k_vote_points, lds, attr_bytes);
    hipLaunchKernelGGL(k_vote_points, dim3(1), dim3(LL_BLOCK), lds, st, src, tgt, n, regions, vc, vs, vw);
}
