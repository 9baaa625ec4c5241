// What a 16-byte load that touches a 128-byte line ONCE costs across the L2 <-> fabric boundary, and what rocprofv3's FETCH_SIZE says about it:
// every lane reads 16 B at a stride of S bytes (S = 16: a wide coalesced read, every byte used; 32, 64, 128, 256: one 16-byte piece per S bytes), over a
// buffer far beyond the L2 and the Infinity Cache, 8 loads in flight per lane.  Prints the lines / sectors touched per second; under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace      (and TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum, TCC_BUBBLE_sum where the counters exist)
// the per-dispatch counter / known request count tells which granule the counter tallies for which pattern (tools/fetch_granule.sh).
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_granule fetch_granule.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int STRIDE16>      // stride in units of 16 bytes
__global__ __launch_bounds__(256) void k_touch(const float4 *__restrict__ src, size_t n_loads, float *sink)
{
    float acc = 0.0f;
    const size_t per = 256 * 8;
    for (size_t i0 = (size_t)blockIdx.x * per + threadIdx.x; i0 < n_loads; i0 += (size_t)gridDim.x * per) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t i = i0 + (size_t)u * 256; v[u] = (i < n_loads) ? src[i * STRIDE16] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// the same line asked for by the FOUR waves of a workgroup at about the same time (one 16-byte piece each, a different piece per wave): does the L2
// merge requests for a line that is already on its way, or does each of them cross the fabric?
__global__ __launch_bounds__(256) void k_touch_dup(const float4 *__restrict__ src, size_t n_lines, float *sink)
{
    float acc = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per = 64 * 8;
    for (size_t l0 = (size_t)blockIdx.x * per + lane; l0 < n_lines; l0 += (size_t)gridDim.x * per) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t l = l0 + (size_t)u * 64; v[u] = (l < n_lines) ? src[l * 8 + wave * 2] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
static void run_dup(const float4 *buf, size_t bytes, float *sink, int reps)
{
    const size_t n_lines = bytes / 128;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_touch_dup, dim3(256 * 8), dim3(256), 0, 0, buf, n_lines, sink);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_touch_dup, dim3(256 * 8), dim3(256), 0, 0, buf, n_lines, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("{\"pattern\": \"each line by 4 waves of a workgroup\", \"lines\": %zu, \"loads\": %zu, \"ms\": %.4f, \"GBps_if_every_line_once\": %.0f, \"GBps_if_every_load_a_line\": %.0f}\n",
           n_lines, n_lines * 4, ms, n_lines * 128.0 / ms * 1e-6, n_lines * 4 * 128.0 / ms * 1e-6);
}

template <int S>
static void run(const float4 *buf, size_t bytes, float *sink, int reps)
{
    const size_t n_loads = bytes / (16 * (size_t)S);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_touch<S>, dim3(256 * 8), dim3(256), 0, 0, buf, n_loads, sink);
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_touch<S>, dim3(256 * 8), dim3(256), 0, 0, buf, n_loads, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("{\"stride_bytes\": %d, \"loads\": %zu, \"ms\": %.4f, \"G_loads_per_s\": %.2f, \"GBps_if_16B\": %.0f, \"GBps_if_32B\": %.0f, \"GBps_if_64B\": %.0f, \"GBps_if_128B\": %.0f}\n",
           S * 16, n_loads, ms, n_loads / ms * 1e-6, n_loads * 16.0 / ms * 1e-6, n_loads * 32.0 / ms * 1e-6, n_loads * 64.0 / ms * 1e-6, n_loads * 128.0 / ms * 1e-6);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const size_t bytes = (size_t)8 << 30;
    float4 *buf; float *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 1, bytes));
    run<1>(buf, bytes, sink, reps); run<2>(buf, bytes, sink, reps); run<4>(buf, bytes, sink, reps); run<8>(buf, bytes, sink, reps); run<16>(buf, bytes, sink, reps);
    run_dup(buf, bytes, sink, reps);
    return 0;
}
