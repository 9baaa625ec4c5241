// VALU issue rate of one SIMD of gfx950: W waves per SIMD, each a loop of independent (ILP 8) or dependent (ILP 1) v_fma_f32 / v_add_u32.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, bool INTOP>
__global__ void k(float *out, int iters, float a, float b)
{
    float x[ILP]; int y[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / ILP; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (INTOP) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[i]) : "v"(it));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0; for (int i = 0; i < ILP; ++i) s += x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP, bool INTOP>
void run(int wps, float *d)
{
    const int iters = 20000, blocks = 256 * 4 * wps;     // one-wave blocks: wps per SIMD if spread evenly
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ILP, INTOP>), dim3(blocks), dim3(64), 0, 0, d, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ILP, INTOP>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * wps;       // wave-instructions issued on one SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%s ILP %d  waves/SIMD %d : %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", INTOP ? "v_add_u32" : "v_fma_f32", ILP, wps, ms, cyc / instr_per_simd);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 4, 8}) { run<8, false>(w, d); run<1, false>(w, d); run<8, true>(w, d); run<1, true>(w, d); }
    return 0;
}
