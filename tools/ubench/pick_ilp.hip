// Does a second, independent arg-max chain in the same wave's instruction stream come (almost) for free?
// One "pick step" = DPP wave max -> readlane -> ballot -> highest lane -> readlane of a payload -> range clear, on one or on two
// independent key sets (branch-free, one basic block).  Reports cycles per loop iteration for 1 wave and 6 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned wave_max(unsigned v)
{
#define M(ctrl, rm) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rm, 0xf, false))
    M(0xb1, 0xf); M(0x4e, 0xf); M(0x114, 0xf); M(0x118, 0xf); M(0x142, 0xa); M(0x143, 0xc);
#undef M
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ void step(unsigned &key, int cli, unsigned &rec, int &npick)
{
    const unsigned kmax = wave_max(key);
    const unsigned long long bal = __ballot(key == kmax);
    const int f = 63 - __builtin_clzll(bal | 1ull);
    const int sl = __builtin_amdgcn_readlane(cli, f);
    const int sel = sl & 0xffff, e = sl >> 16;
    const int slo = sel - (e & 15), shi = sel + (e >> 4);
    key = ((unsigned)((cli & 0xffff) - slo) <= (unsigned)(shi - slo)) ? 0u : key;
    rec += (unsigned)sl; npick += (kmax != 0u);
}
template <int CH>
__global__ void k(unsigned *out, int iters, long long *cyc)
{
    unsigned keyA = threadIdx.x * 2654435761u | 1u, keyB = threadIdx.x * 40503u + 77u | 1u;
    const int cli = (int)(threadIdx.x & 63) * 7 + 100 + (0x33 << 16);
    unsigned recA = 0, recB = 0; int nA = 0, nB = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        step(keyA, cli, recA, nA);
        if (CH == 2) step(keyB, cli, recB, nB);
        if ((it & 15) == 15) { keyA = (threadIdx.x + it) * 2654435761u | 1u; keyB = (threadIdx.x + it) * 40503u | 1u; }   // refill
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = recA + recB + nA + nB;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int CH> void run(int wps, unsigned *d, long long *dc)
{
    const int iters = 20000, blocks = 256 * 4 * wps;
    hipLaunchKernelGGL((k<CH>), dim3(blocks), dim3(64), 0, 0, d, 100, dc); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<CH>), dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("chains %d  waves/SIMD %d : %.3f ms  wave 0: %.1f s_memtime ticks per iteration; %.1f ns per iteration per wave\n", CH, wps, ms, (double)c / iters, ms * 1e6 / iters);
}
int main()
{
    unsigned *d; long long *dc; hipMalloc(&d, 256 * 4 * 8 * 64 * 4); hipMalloc(&dc, 8);
    for (int w : {1, 6}) { run<1>(w, d, dc); run<2>(w, d, dc); }
    return 0;
}
