// What one MI355X streams through the L2 <-> fabric boundary when the kernel does nothing else: read-only, write-only and
// copy kernels with U independent 16-byte accesses in flight per lane (U = 1, 2, 4, 8), an LDS-DMA read (global_load_lds,
// no staging registers) and the two address patterns the library uses:
//   flat     one contiguous range per workgroup trip (raw[], the published clouds)
//   rows     rows of `row` float4 at a stride of `cap` float4 (laserCloud: ring_cap = 2304, rings ~2000 points long),
//            one row per workgroup trip
// Persistent grid: (workgroups per CU) x 256 CUs, every workgroup walks its share in trips.  Prints GB/s (bytes that
// MUST move: read + written once) and, from Little's law at the latency given on the command line, the bytes in flight
// per CU that rate implies.  Run under  rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --kernel-trace
// to get the measured latency and requests in flight next to it (tools/stream_rate.sh).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_rate stream_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// row / cap in float4; flat = (row == cap).  n_rows rows in all; workgroup b takes rows b, b + grid, ...
template <int U, int MODE>      // MODE 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ src, float4 *__restrict__ dst, int n_rows, int row, int cap, float *sink)
{
    float acc = 0.0f;
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const float4 *s = src + (size_t)r * cap;
        float4 *d = dst + (size_t)r * cap;
        for (int i0 = threadIdx.x; i0 < row; i0 += 256 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256;
                v[u] = make_float4(1.0f, 2.0f, 3.0f, (float)i);
                if (MODE != 1 && i < row) v[u] = s[i];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256;
                if (MODE == 0) acc += v[u].x + v[u].y + v[u].z + v[u].w;
                else if (i < row) d[i] = v[u];
            }
        }
    }
    if (MODE == 0 && acc == 12345.678f) sink[0] = acc;      // never true: keeps the loads
}

// read through LDS-DMA: T tiles of 256 x 16 B per wave-group in flight, consumed by a ds_read per lane
template <int T>
__global__ __launch_bounds__(256) void k_stream_dma(const float4 *__restrict__ src, int n_rows, int row, int cap, float *sink)
{
    __shared__ __attribute__((aligned(16))) float4 tile[T][256];
    float acc = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const float4 *s = src + (size_t)r * cap;
        for (int i0 = 0; i0 < row; i0 += 256 * T) {
#pragma unroll
            for (int u = 0; u < T; ++u) {
                const int i = i0 + u * 256 + wave * 64 + lane;
                if (i < row) __builtin_amdgcn_global_load_lds((glb_void *)(s + i), (lds_void *)(&tile[u][wave * 64]), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = 0; u < T; ++u) {
                const int i = i0 + u * 256 + wave * 64 + lane;
                if (i < row) { const float4 v = tile[u][wave * 64 + lane]; acc += v.x + v.y + v.z + v.w; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

struct Res { const char *name; int u; int wgs; double gbps; };

template <typename L>
static double time_ms(L launch, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    // defaults: the headline's laserCloud -- 64 rings x 2304 capacity, ~1650 points used per ring (105 k of 147 k slots), 8192 scans
    int cap = 2304, row = 1648, scans = 8192, rings = 64, reps = 5; double lat_cycles = 1100.0; int only_wgs = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--cap")) cap = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--row")) row = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--scans")) scans = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--latency")) lat_cycles = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--wgs")) only_wgs = atoi(argv[i + 1]);
    }
    const size_t n_rows = (size_t)scans * rings, total = n_rows * cap;
    float4 *a, *b; float *sink;
    CK(hipMalloc(&a, total * 16)); CK(hipMalloc(&b, total * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, total * 16)); CK(hipMemset(b, 0, total * 16));
    printf("{\"stream_rate\": {\"cap\": %d, \"row\": %d, \"rows\": %zu, \"GB_per_array\": %.2f, \"assumed_latency_cycles\": %.0f, \"results\": [\n", cap, row, n_rows, total * 16 / 1e9, lat_cycles);
    bool firstline = true;
    auto emit = [&](const char *pattern, const char *mode, int u, int wgs, double bytes, double ms) {
        const double gbps = bytes / (ms * 1e-3) / 1e9;
        const double inflight_per_cu = gbps * 1e9 / 256.0 * (lat_cycles / 2.4e9);      // Little: bytes in flight per CU at that latency
        printf("%s  {\"pattern\": \"%s\", \"mode\": \"%s\", \"in_flight_per_lane\": %d, \"wg_per_cu\": %d, \"ms\": %.3f, \"GBps\": %.0f, \"little_KB_in_flight_per_cu\": %.1f}",
               firstline ? "" : ",\n", pattern, mode, u, wgs, ms, gbps, inflight_per_cu / 1024.0);
        firstline = false; fflush(stdout);
    };
    for (int pat = 0; pat < 2; ++pat) {
        // flat: the same bytes as one contiguous range (rows of `cap` at stride `cap`, fewer of them)
        const int prow = pat == 0 ? cap : row, pcap = cap;
        const size_t prow_n = pat == 0 ? (size_t)((double)n_rows * row / cap) : n_rows;
        const double rbytes = (double)prow_n * prow * 16;
        const char *pname = pat == 0 ? "flat" : "rows";
        for (int wgs : {2, 4, 6, 8}) {
            if (only_wgs && wgs != only_wgs) continue;
            const int grid = 256 * wgs;
#define RUN(U, MODE, mname, mult) emit(pname, mname, U, wgs, rbytes * mult, time_ms([&] { hipLaunchKernelGGL((k_stream<U, MODE>), dim3(grid), dim3(256), 0, 0, a, b, (int)prow_n, prow, pcap, sink); }, reps))
            RUN(1, 0, "read", 1); RUN(2, 0, "read", 1); RUN(4, 0, "read", 1); RUN(8, 0, "read", 1);
            RUN(1, 1, "write", 1); RUN(4, 1, "write", 1);
            RUN(1, 2, "copy", 2); RUN(2, 2, "copy", 2); RUN(4, 2, "copy", 2); RUN(8, 2, "copy", 2);
#undef RUN
#define RUND(T) emit(pname, "read_lds_dma", T, wgs, rbytes, time_ms([&] { hipLaunchKernelGGL((k_stream_dma<T>), dim3(grid), dim3(256), 0, 0, a, (int)prow_n, prow, pcap, sink); }, reps))
            RUND(1); RUND(2); RUND(4);
#undef RUND
        }
    }
    printf("\n]}}\n");
    return 0;
}
