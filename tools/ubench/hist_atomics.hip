// What would a per-scan x/y-cell histogram cost if the ring kernel's workgroups bumped it with global atomics while they write
// their less-flat points (so that k_build_grid could drop its first pass)?  64 ring workgroups per scan, all of a scan on one XCD
// (the ring kernel's mapping), ~510 points per ring, neighbouring points mostly in the same cell.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/hist_atomics.hip -o gpurun_out/hist_atomics && gpurun_out/hist_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SCOPE>
__global__ __launch_bounds__(256) void k_hist(int *hist, int nscan, int per_ring)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int scan = (j / 64) * 8 + xcd, ring = j % 64;
    if (scan >= nscan) return;
    int *h = hist + (size_t)scan * 16384;
    for (int i = threadIdx.x; i < per_ring; i += 256) {
        // a ring at ~10 m: 510 points around the circle, ~4 per cell
        const float a = 6.2831853f * (float)i / (float)per_ring, rho = 6.0f + 0.5f * ring;
        const int cx = min(max((int)floorf(rho * cosf(a) + 64.0f), 0), 127), cy = min(max((int)floorf(rho * sinf(a) + 64.0f), 0), 127);
        __hip_atomic_fetch_add(&h[cy * 128 + cx], 1, __ATOMIC_RELAXED, SCOPE);
    }
}

int main()
{
    const int nscan = 8192, per_ring = 510;
    int *hist; CK(hipMalloc(&hist, (size_t)nscan * 16384 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int scope = 0; scope < 2; ++scope) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(hist, 0, (size_t)nscan * 16384 * 4));
            CK(hipEventRecord(a));
            if (scope == 0) hipLaunchKernelGGL(k_hist<__HIP_MEMORY_SCOPE_AGENT>, dim3(64 * nscan), dim3(256), 0, 0, hist, nscan, per_ring);
            else hipLaunchKernelGGL(k_hist<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(64 * nscan), dim3(256), 0, 0, hist, nscan, per_ring);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("%s scope: %d scans x 64 rings x %d atomics: %.3f ms (%.1f G atomics/s)\n", scope ? "workgroup" : "agent", nscan, per_ring, ms,
                   (double)nscan * 64 * per_ring / ms * 1e-6);
        }
    }
    CK(hipEventRecord(a)); CK(hipMemsetAsync(hist, 0, (size_t)nscan * 16384 * 4)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("memset of the %d histograms: %.3f ms\n", nscan, ms);
    return 0;
}
