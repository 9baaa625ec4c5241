/*
 * ll_factors.hip -- a9 + a10: point-to-line / point-to-plane residuals, Jacobians, Huber, normal equations,
 * one Gauss-Newton step.
 * Replaces, of /root/reference: lidarFactor.hpp:9-52 (LidarEdgeFactor), :203-251 (LidarPlaneFactor_modify),
 * the per-block loss + manifold handling Ceres applies around them (laserOdometry.cpp:475-482, :615-616,
 * :797-808) and, for the "one GN iteration" unit of work, the linear solve inside ceres::Solve (:820-825).
 *
 * The reference evaluates Jacobians with forward-mode Jets (AutoDiffCostFunction<., R, 4, 3>).  Here they are
 * closed-form, at s = 1 (DISTORTION 0, laserOdometry.cpp:23): with lp = q*cp + t by Eigen's
 * _transformVector formula f(u,w) = v + w*(2 u x v) + u x (2 u x v),
 *     d lp / d w = 2 u x v,   d lp / d u = -2w [v]x + 2((u.v) I + u v^T - 2 v u^T),   d lp / d t = I,
 * (the derivative of that exact formula, valid for non-unit q as autodiff is), edge: d r / d lp = [b-a]x / |a-b|,
 * plane: d r / d lp = weight * n^T.  Local (tangent) Jacobian = ambient * EigenQuaternionManifold::PlusJacobian.
 * Everything is f64.  One workgroup per scan pair; each thread walks residual blocks, accumulates the 21 + 6 + 1
 * unique normal-equation terms in registers, then a 64-lane shuffle tree + LDS combine.  No atomics.
 */
#include "ll_common.h"

#include "ll_factor_math.h"
#include "ll_lm_step.h"

/* the normal equations of slot s at the pose pose_in[7] (global or LDS), by the whole workgroup; thread 0 leaves the 44-double
 * record in out[] (global or LDS).  Ends with every thread past the last barrier but WITHOUT a barrier after thread 0's
 * stores: the caller synchronises before anybody else reads out[]. */
/* DIST: DISTORTION 1 (ll_params.distortion): a second instantiation of the kernels, so that the reference's own build (s = 1 everywhere) pays
 * neither a branch nor a register for it */
template <int NT, bool DIST>
__device__ __forceinline__ void ll_neq_eval(const LLView &V, int s, const double *pose_in, double *out, double (*red)[LL_NACC])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const PairHdr ph = V.pair[s];
    Pose P;
    for (int k = 0; k < 4; ++k) P.q[k] = pose_in[k];
    for (int k = 0; k < 3; ++k) P.t[k] = pose_in[4 + k];
    const float4 *corner, *surf; int mc, ms;
    ll_targets(V, s, corner, mc, surf, ms);
    const float4 *sharp = V.sharp + (size_t)s * V.cap_sharp, *flat = V.flat + (size_t)s * V.cap_flat;
    const int *es = V.e_src + (size_t)s * V.cap_sharp, *ea = V.e_a + (size_t)s * V.cap_sharp, *eb = V.e_b + (size_t)s * V.cap_sharp;
    const int *ps = V.p_src + (size_t)s * V.cap_flat, *pa = V.p_a + (size_t)s * V.cap_flat, *pb = V.p_b + (size_t)s * V.cap_flat, *pc = V.p_c + (size_t)s * V.cap_flat;
    const uint8_t *vs = V.v_sel + (size_t)s * V.cap_flat; const float *vw = V.v_w + (size_t)s * V.cap_flat;

    double acc[LL_NACC];
#pragma unroll
    for (int k = 0; k < LL_NACC; ++k) acc[k] = 0.0;
    for (int i = tid; i < ph.n_edge; i += NT) {
        double r[3], Jq[3][4], Jt[3][3];
        ll_edge(P, sharp[es[i]], corner[ea[i]], corner[eb[i]], r, Jq, Jt, DIST ? ll_point_s(1, sharp[es[i]]) : 1.0);
        const double sc = ll_huber_scale(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], V.huber, acc[27]);
        for (int row = 0; row < 3; ++row) {
            double J[6];
            ll_to_local(P, Jq[row], J);
            J[3] = Jt[row][0]; J[4] = Jt[row][1]; J[5] = Jt[row][2];
            for (int k = 0; k < 6; ++k) J[k] *= sc;
            ll_acc_row(acc, J, r[row] * sc);
        }
    }
    for (int i = tid; i < ph.n_plane; i += NT) {
        if (!vs[i]) continue;
        double r, Jq[4], Jt[3], J[6];
        ll_plane(P, flat[ps[i]], surf[pa[i]], surf[pb[i]], surf[pc[i]], (double)vw[i], r, Jq, Jt, DIST ? ll_point_s(1, flat[ps[i]]) : 1.0);
        const double sc = ll_huber_scale(r * r, V.huber, acc[27]);
        ll_to_local(P, Jq, J);
        J[3] = Jt[0]; J[4] = Jt[1]; J[5] = Jt[2];
        for (int k = 0; k < 6; ++k) J[k] *= sc;
        ll_acc_row(acc, J, r * sc);
    }
#pragma unroll
    for (int k = 0; k < LL_NACC; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double tot[LL_NACC];
        for (int k = 0; k < LL_NACC; ++k) { double v = 0.0; for (int w = 0; w < NT / 64; ++w) v += red[w][k]; tot[k] = v; }
        int k = 0;
        for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { out[a * 6 + b] = tot[k]; out[b * 6 + a] = tot[k]; ++k; }
        for (int a = 0; a < 6; ++a) out[36 + a] = tot[21 + a];
        out[42] = tot[27];
        out[43] = (double)(3 * ph.n_edge + ph.n_plane_sel);
    }
}

template <bool DIST>
__global__ __launch_bounds__(LL_BLOCK) void k_normal_equations(LLView V, int first, int count, int do_step)
{
    if ((int)blockIdx.x >= count) return;
    const int s = first + blockIdx.x;
    __shared__ double red[LL_BLOCK / 64][LL_NACC];
    double *pose = V.pose + (size_t)s * 7, *out = V.neq + (size_t)s * LL_NEQ_STRIDE;
    ll_neq_eval<LL_BLOCK, DIST>(V, s, pose, out, red);
    if (threadIdx.x == 0 && do_step) {
        double H[36], g[6], d[6];
        for (int i = 0; i < 36; ++i) H[i] = out[i];
        for (int i = 0; i < 6; ++i) g[i] = out[36 + i];
        if (ll_chol_solve(H, g, d) == 0) ll_pose_plus(pose, d);
    }
}

__global__ void k_gn_step(LLView V, int first, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int s = first + i;
    const double *in = V.neq + (size_t)s * LL_NEQ_STRIDE;
    double H[36], g[6], d[6];
    for (int k = 0; k < 36; ++k) H[k] = in[k];
    for (int k = 0; k < 6; ++k) g[k] = in[36 + k];
    if (ll_chol_solve(H, g, d) == 0) ll_pose_plus(V.pose + (size_t)s * 7, d);
}

/* What CostFunction::Evaluate returns per block (no loss): rows of r, Jq (x4 ambient), Jt (x3). */
__global__ __launch_bounds__(LL_BLOCK) void k_rows(LLView V, int s, const double *pose7, double *r_out, double *Jq_out, double *Jt_out)
{
    const PairHdr ph = V.pair[s];
    Pose P;
    for (int k = 0; k < 4; ++k) P.q[k] = pose7[k];
    for (int k = 0; k < 3; ++k) P.t[k] = pose7[4 + k];
    const float4 *corner, *surf; int mc, ms;
    ll_targets(V, s, corner, mc, surf, ms);
    const float4 *sharp = V.sharp + (size_t)s * V.cap_sharp, *flat = V.flat + (size_t)s * V.cap_flat;
    const int *es = V.e_src + (size_t)s * V.cap_sharp, *ea = V.e_a + (size_t)s * V.cap_sharp, *eb = V.e_b + (size_t)s * V.cap_sharp;
    const int *ps = V.p_src + (size_t)s * V.cap_flat, *pa = V.p_a + (size_t)s * V.cap_flat, *pb = V.p_b + (size_t)s * V.cap_flat, *pc = V.p_c + (size_t)s * V.cap_flat;
    const uint8_t *vs = V.v_sel + (size_t)s * V.cap_flat; const float *vw = V.v_w + (size_t)s * V.cap_flat;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gtid; i < ph.n_edge; i += gsz) {
        double r[3], Jq[3][4], Jt[3][3];
        ll_edge(P, sharp[es[i]], corner[ea[i]], corner[eb[i]], r, Jq, Jt, ll_point_s(V.distortion, sharp[es[i]]));
        for (int row = 0; row < 3; ++row) {
            const size_t R0 = (size_t)3 * i + row;
            r_out[R0] = r[row];
            for (int k = 0; k < 4; ++k) Jq_out[R0 * 4 + k] = Jq[row][k];
            for (int k = 0; k < 3; ++k) Jt_out[R0 * 3 + k] = Jt[row][k];
        }
    }
    /* selected planes keep correspondence order: row index = 3*n_edge + rank among selected */
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        size_t R0 = (size_t)3 * ph.n_edge;
        for (int i = 0; i < ph.n_plane; ++i) {
            if (!vs[i]) continue;
            double r, Jq[4], Jt[3];
            ll_plane(P, flat[ps[i]], surf[pa[i]], surf[pb[i]], surf[pc[i]], (double)vw[i], r, Jq, Jt, ll_point_s(V.distortion, flat[ps[i]]));
            r_out[R0] = r;
            for (int k = 0; k < 4; ++k) Jq_out[R0 * 4 + k] = Jq[k];
            for (int k = 0; k < 3; ++k) Jt_out[R0 * 3 + k] = Jt[k];
            ++R0;
        }
    }
}

void ll_launch_normal_equations(const LLView &V, int first, int count, int do_step, hipStream_t st, LLProfiler *prof)
{
    ll_prof_mark(prof, LL_K_NORMAL_EQ, st);
    if (V.distortion) hipLaunchKernelGGL(k_normal_equations<true>, dim3(count), dim3(LL_BLOCK), 0, st, V, first, count, do_step);
    else hipLaunchKernelGGL(k_normal_equations<false>, dim3(count), dim3(LL_BLOCK), 0, st, V, first, count, do_step);
    ll_prof_mark(prof, LL_K_END, st);
}
/* ------------------------------------------------------------------------------------------------------------------
 * f1: ceres::Solve as laserOdometry.cpp:820-825 configures it (DENSE_QR, max_num_iterations 4, defaults otherwise):
 * trust-region minimizer + Levenberg-Marquardt strategy (Ceres 2.x trust_region_minimizer.cc,
 * levenberg_marquardt_strategy.cc), one thread per scan pair; the residual evaluation between propose and accept is
 * k_normal_equations at the candidate pose.  State layout (doubles): 0-6 x, 7 cost, 8-43 H, 44-49 g, 50-55 jacobi scale,
 * 56 radius, 57 decrease_factor, 58 iteration, 59 done, 60-66 candidate, 67 model_cost_change, 68 pending, 69 successes,
 * 70 initial cost.
 * ------------------------------------------------------------------------------------------------------------------ */
__global__ void k_lm_begin(LLView V, int first, int count, LLLmOpt o)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int s = first + i;
    ll_lm_begin_one(V.lm + (size_t)s * LL_LM_STRIDE, V.neq + (size_t)s * LL_NEQ_STRIDE, V.pose + (size_t)s * 7, o);
}
__global__ void k_lm_propose(LLView V, int first, int count, LLLmOpt o)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int s = first + i;
    ll_lm_propose_one(V.lm + (size_t)s * LL_LM_STRIDE, V.pose + (size_t)s * 7, o);
}
__global__ void k_lm_accept(LLView V, int first, int count, LLLmOpt o)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int s = first + i;
    ll_lm_accept_one(V.lm + (size_t)s * LL_LM_STRIDE, V.neq + (size_t)s * LL_NEQ_STRIDE, V.pose + (size_t)s * 7, o);
}

/* The whole solve of a slot in ONE launch: evaluate, begin, max_num_iterations x (propose, evaluate, accept) -- one workgroup per
 * slot, the steps of the single-thread trust-region logic on thread 0 between the workgroup-wide evaluations, the pose and the
 * normal equations handed over in LDS.  The same trust-region logic as the launch-per-step sequence above (kept for the
 * row-parallel mode, where an all-reduce sits between evaluate and accept), but NOT bit-identical to it: the evaluation here
 * sums the rows over 512 threads (ll_neq_eval<512>, stride-512 partition, eight waves), k_normal_equations over 256, so the f64
 * sums differ in their last bits and the two paths agree to rounding only (tests compare them within tolerance).  The
 * node-style odometry frame drops from 48 dependent launches to 9.  The state and the last normal equations are left in V.lm / V.neq. */
#define LL_LM_THREADS 512     /* the solve is a chain of 1 + max_num_iterations evaluations on ONE workgroup: twice the threads, half the chain */
template <bool DIST>
__global__ __launch_bounds__(LL_LM_THREADS) void k_lm_solve(LLView V, int first, int count, LLLmOpt o)
{
    if ((int)blockIdx.x >= count) return;
    const int s = first + blockIdx.x;
    __shared__ double red[LL_LM_THREADS / 64][LL_NACC];
    __shared__ double sp[7], sneq[LL_NEQ_STRIDE], sL[LL_LM_STRIDE];
    const int tid = threadIdx.x;
    if (tid < 7) sp[tid] = V.pose[(size_t)s * 7 + tid];
    for (int k = tid; k < LL_LM_STRIDE; k += LL_LM_THREADS) sL[k] = 0.0;
    __syncthreads();
    ll_neq_eval<LL_LM_THREADS, DIST>(V, s, sp, sneq, red);
    if (tid == 0) ll_lm_begin_one(sL, sneq, sp, o);
    for (int it = 0; it < o.max_num_iterations; ++it) {
        if (tid == 0) ll_lm_propose_one(sL, sp, o);
        __syncthreads();
        ll_neq_eval<LL_LM_THREADS, DIST>(V, s, sp, sneq, red);
        if (tid == 0) ll_lm_accept_one(sL, sneq, sp, o);
    }
    __syncthreads();
    if (tid < 7) V.pose[(size_t)s * 7 + tid] = sp[tid];
    for (int k = tid; k < LL_NEQ_STRIDE; k += LL_LM_THREADS) V.neq[(size_t)s * LL_NEQ_STRIDE + k] = sneq[k];
    for (int k = tid; k < LL_LM_STRIDE; k += LL_LM_THREADS) V.lm[(size_t)s * LL_LM_STRIDE + k] = sL[k];
}

void ll_launch_lm_solve(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st)
{
    if (V.distortion) hipLaunchKernelGGL(k_lm_solve<true>, dim3(count), dim3(LL_LM_THREADS), 0, st, V, first, count, o);
    else hipLaunchKernelGGL(k_lm_solve<false>, dim3(count), dim3(LL_LM_THREADS), 0, st, V, first, count, o);
}
void ll_launch_lm_begin(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st)
{
    hipLaunchKernelGGL(k_lm_begin, dim3((count + 63) / 64), dim3(64), 0, st, V, first, count, o);
}
void ll_launch_lm_propose(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st)
{
    hipLaunchKernelGGL(k_lm_propose, dim3((count + 63) / 64), dim3(64), 0, st, V, first, count, o);
}
void ll_launch_lm_accept(const LLView &V, int first, int count, const LLLmOpt &o, hipStream_t st)
{
    hipLaunchKernelGGL(k_lm_accept, dim3((count + 63) / 64), dim3(64), 0, st, V, first, count, o);
}

void ll_launch_gn_step(const LLView &V, int first, int count, hipStream_t st, LLProfiler *prof)
{
    ll_prof_mark(prof, LL_K_GN_STEP, st);
    hipLaunchKernelGGL(k_gn_step, dim3((count + 63) / 64), dim3(64), 0, st, V, first, count);
    ll_prof_mark(prof, LL_K_END, st);
}
void ll_launch_rows(const LLView &V, int slot, const double *pose7_dev, double *r, double *Jq, double *Jt, hipStream_t st)
{
    hipLaunchKernelGGL(k_rows, dim3(8), dim3(LL_BLOCK), 0, st, V, slot, pose7_dev, r, Jq, Jt);
}
