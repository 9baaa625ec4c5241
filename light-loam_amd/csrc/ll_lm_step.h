/*
 * ll_lm_step.h -- the single-thread steps of the trust-region / Levenberg-Marquardt loop (Ceres 2.x trust_region_minimizer.cc,
 * levenberg_marquardt_strategy.cc as laserOdometry.cpp:820-825 / laserMapping.cpp:2072-2082 configure it), shared by the
 * odometry solve (ll_factors.hip) and the mapping solve (ll_mapping.hip).  State layout (doubles): 0-6 x, 7 cost, 8-43 H,
 * 44-49 g, 50-55 jacobi scale, 56 radius, 57 decrease_factor, 58 iteration, 59 done, 60-66 candidate, 67 model_cost_change,
 * 68 pending, 69 successes, 70 initial cost, 71 poisoned (LLLmOpt::nan_poisons_pose: the summed normal equations carried a NaN
 * cost or row count -- the record a failing rank of a row-parallel solve contributes, lightloam_rccl.hpp -- and every later step
 * leaves a NaN pose, so that the healthy ranks' ll_map_get_pose reports LL_ERR_STATE instead of a finite, un-optimised pose).
 */
#pragma once
#include "ll_common.h"
#include "ll_factor_math.h"

/* the three steps on one solve: L = its state (LL_LM_STRIDE doubles), in = the normal equations at `pose`, pose = where the next
 * evaluation happens (global or LDS) */
__device__ __forceinline__ bool ll_lm_record_is_nan(const double *in) { return in[42] != in[42] || in[43] != in[43]; }
__device__ __forceinline__ void ll_lm_poison(double *pose) { for (int k = 0; k < 7; ++k) pose[k] = __builtin_nan(""); }

__device__ __forceinline__ void ll_lm_begin_one(double *L, const double *in, double *pose, const LLLmOpt &o)
{
    for (int k = 0; k < 7; ++k) L[k] = pose[k];
    L[71] = (o.nan_poisons_pose && ll_lm_record_is_nan(in)) ? 1.0 : 0.0;
    if (L[71] != 0.0) ll_lm_poison(pose);
    L[7] = in[42];
    for (int k = 0; k < 36; ++k) L[8 + k] = in[k];
    for (int k = 0; k < 6; ++k) L[44 + k] = in[36 + k];
    for (int k = 0; k < 6; ++k) L[50 + k] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(in[k * 6 + k])) : 1.0;
    L[56] = o.initial_radius; L[57] = 2.0; L[58] = 0.0; L[59] = 0.0; L[68] = 0.0; L[69] = 0.0; L[70] = in[42];
}

__device__ __forceinline__ void ll_lm_propose_one(double *L, double *pose, const LLLmOpt &o)
{
    L[68] = 0.0;
    if (L[71] != 0.0) { ll_lm_poison(pose); return; }
    if (L[59] == 0.0) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        bool stop = (int)L[58] >= o.max_num_iterations || L[56] < o.min_radius;
        if (!stop) {   /* gradient tolerance: max-norm of Plus(x, -gradient) - x */
            double x2[7], ng[6];
            for (int k = 0; k < 7; ++k) x2[k] = L[k];
            for (int k = 0; k < 6; ++k) ng[k] = -L[44 + k];
            ll_pose_plus(x2, ng);
            double m = 0.0;
            for (int k = 0; k < 7; ++k) m = fmax(m, fabs(x2[k] - L[k]));
            stop = m <= o.gradient_tolerance;
        }
        if (stop) L[59] = 1.0;
    }
    if (L[59] != 0.0) { for (int k = 0; k < 7; ++k) pose[k] = L[k]; return; }
    L[58] += 1.0;
    double A[36], b[6], ys[6], d[6];
    for (int a = 0; a < 6; ++a) {
        for (int c = 0; c < 6; ++c) A[a * 6 + c] = L[8 + a * 6 + c] * L[50 + a] * L[50 + c];
        b[a] = L[44 + a] * L[50 + a];
    }
    for (int a = 0; a < 6; ++a) {
        double dg = A[a * 6 + a];
        dg = fmin(fmax(dg, o.min_lm_diagonal), o.max_lm_diagonal);
        A[a * 6 + a] += dg / L[56];
    }
    bool ok = ll_chol_solve(A, b, ys) == 0;
    double mcc = 0.0;
    if (ok) {
        double gd = 0.0, dHd = 0.0;
        for (int a = 0; a < 6; ++a) d[a] = ys[a] * L[50 + a];
        for (int a = 0; a < 6; ++a) { gd += L[44 + a] * d[a]; for (int c = 0; c < 6; ++c) dHd += d[a] * L[8 + a * 6 + c] * d[c]; }
        mcc = -(gd + 0.5 * dHd);
    }
    if (!ok || !(mcc > 0.0)) { L[56] *= 0.5; for (int k = 0; k < 7; ++k) pose[k] = L[k]; return; }   /* invalid step */
    double c7[7];
    for (int k = 0; k < 7; ++k) c7[k] = L[k];
    ll_pose_plus(c7, d);
    for (int k = 0; k < 7; ++k) { L[60 + k] = c7[k]; pose[k] = c7[k]; }
    L[67] = mcc; L[68] = 1.0;
}

__device__ __forceinline__ void ll_lm_accept_one(double *L, const double *in, double *pose, const LLLmOpt &o)
{
    if (o.nan_poisons_pose && ll_lm_record_is_nan(in)) L[71] = 1.0;
    if (L[71] != 0.0) { L[68] = 0.0; ll_lm_poison(pose); return; }
    if (L[59] == 0.0 && L[68] != 0.0) {
        const double cc = in[42], cost = L[7];
        double step_norm = 0.0, x_norm = 0.0;
        for (int k = 0; k < 7; ++k) { const double e = L[60 + k] - L[k]; step_norm += e * e; x_norm += L[k] * L[k]; }
        step_norm = sqrt(step_norm); x_norm = sqrt(x_norm);
        if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) L[59] = 1.0;     /* before the step is taken */
        else if (fabs(cost - cc) <= o.function_tolerance * cost) L[59] = 1.0;
        else {
            const double rho = (cost - cc) / L[67];
            if (rho > o.min_relative_decrease) {
                for (int k = 0; k < 7; ++k) L[k] = L[60 + k];
                L[7] = cc;
                for (int k = 0; k < 36; ++k) L[8 + k] = in[k];
                for (int k = 0; k < 6; ++k) L[44 + k] = in[36 + k];
                const double f = 1.0 - pow(2.0 * rho - 1.0, 3.0);
                L[56] = fmin(L[56] / fmax(1.0 / 3.0, f), o.max_radius);
                L[57] = 2.0; L[69] += 1.0;
            } else {
                L[56] = L[56] / L[57];
                L[57] *= 2.0;
            }
        }
    }
    L[68] = 0.0;
    for (int k = 0; k < 7; ++k) pose[k] = L[k];
}

