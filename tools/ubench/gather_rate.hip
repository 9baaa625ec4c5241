// How many DIVERGENT 16-byte accesses (every lane its own 128-byte line) a CU's vector memory path takes per cycle, with the data in the L2 (a 2 MB
// buffer: the fabric is out of the picture) -- loads and stores, against the same accesses coalesced (a wave's 64 lanes on 8 consecutive lines).
// k_ring_features does ~2 500 such lane-accesses per ring (the gather behind its sort, its centroid stores), k_build_grid one per point (its scatter).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>    // 0 divergent load, 1 coalesced load, 2 divergent store, 3 coalesced store, 4 divergent load inside a 26 KB window per workgroup (a ring row)
__global__ __launch_bounds__(256) void k_g(float4 *buf, unsigned mask_lines, int iters, float *sink)
{
    float acc = 0.0f;
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    const unsigned lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        float4 v[8]; unsigned idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            unsigned line;
            if (MODE == 0 || MODE == 2) line = s & mask_lines;
            else if (MODE == 4) line = ((blockIdx.x * 208u) + (s % 208u)) & mask_lines;
            else line = ((s & mask_lines) & ~7u) + (lane >> 3);                          // the wave's 64 lanes on 8 consecutive lines ...
            if (MODE == 1 || MODE == 3) { line = (__builtin_amdgcn_readfirstlane(s) & mask_lines & ~7u) + (lane >> 3); idx[u] = line * 8 + (lane & 7); }   // ... lane l on piece l % 8: 1 KB contiguous
            else idx[u] = line * 8 + (s >> 29);
        }
        if (MODE == 0 || MODE == 1 || MODE == 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = buf[idx[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u].x;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) buf[idx[u]] = make_float4(1.f, 2.f, 3.f, (float)it);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
static void run(const char *name, float4 *buf, unsigned lines, float *sink)
{
    const int iters = 200, grid = 256 * 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_g<MODE>, dim3(grid), dim3(256), 0, 0, buf, lines - 1, iters, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_g<MODE>, dim3(grid), dim3(256), 0, 0, buf, lines - 1, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)grid * 256 * iters * 8;
    printf("{\"pattern\": \"%s\", \"buffer_MB\": %.1f, \"lane_accesses\": %.0f, \"ms\": %.3f, \"G_lane_accesses_per_s\": %.1f, \"per_CU_per_cycle_at_2.4GHz\": %.3f, \"cycles_per_wave_instruction\": %.1f}\n",
           name, lines * 128.0 / 1e6, n, ms, n / ms * 1e-6, n / (ms * 1e-3) / 256.0 / 2.4e9, 64.0 / (n / (ms * 1e-3) / 256.0 / 2.4e9));
}

int main()
{
    float4 *buf; float *sink;
    const unsigned big = 1u << 24;      // 2 GB of lines
    CK(hipMalloc(&buf, (size_t)big * 128)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 0, (size_t)big * 128));
    for (unsigned lines : {1u << 14, 1u << 18, 1u << 24}) {       // 2 MB (every XCD's L2 holds it), 32 MB (the L2s together), 2 GB
        run<0>("divergent 16-byte loads", buf, lines, sink);
        run<1>("coalesced 16-byte loads (1 KB per wave instruction)", buf, lines, sink);
        run<2>("divergent 16-byte stores", buf, lines, sink);
        run<3>("coalesced 16-byte stores", buf, lines, sink);
    }
    run<4>("divergent 16-byte loads inside a 26 KB window per workgroup", buf, 1u << 20, sink);
    return 0;
}
