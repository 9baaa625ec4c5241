// Does the LDS serve the lanes of ONE returning atomic instruction that hit the same address in ascending lane order?  (The ISA manual promises no
// order.  A stable counting sort that takes its rank from ds_add_rtn_u32 needs exactly this: lane i < lane j, same counter -> i gets the smaller value.)
// Every wave draws random digit patterns (uniform over 2^b values for b = 0..8, and runs of equal digits), does one ds_add_rtn per pattern on zeroed
// wave-private counters and compares the returned value with the number of lower lanes holding the same digit (match-any by ballots).
// Prints the number of patterns tried and the number of lanes that disagreed.    Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_order lds_atomic_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned rnd(unsigned &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ __launch_bounds__(256) void k_order(int iters, unsigned long long *out)
{
    __shared__ int cnt[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    unsigned long long bad = 0, tried = 0;
    for (int it = 0; it < iters; ++it) {
        const int b = it % 9;
        int d = (int)(rnd(s) & ((1u << b) - 1u));
        if ((it / 9) % 3 == 1) d = __shfl(d, lane & ~((1 << (it % 5)) - 1));          // runs of 1, 2, 4, 8, 16 equal digits
        if ((it / 9) % 3 == 2) d = __shfl(d, (lane * 7 + it) & 63);                    // the same multiset, scattered over the lanes
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int got = atomicAdd(&cnt[wave][d], 1);
        int want = 0;
        for (int l = 0; l < 64; ++l) { const int dl = __shfl(d, l); if (l < lane && dl == d) ++want; }
        bad += (got != want);
        // the same on counters that are NOT zero, two instructions back to back (row k + 1 must see row k's adds)
        const int d2 = (int)(rnd(s) & ((1u << b) - 1u));
        const int got2 = atomicAdd(&cnt[wave][d2], 1);
        int want2 = 0;
        for (int l = 0; l < 64; ++l) { const int dl = __shfl(d, l), dl2 = __shfl(d2, l); if (dl == d2) ++want2; if (l < lane && dl2 == d2) ++want2; }
        bad += (got2 != want2);
        tried += 2;
    }
    atomicAdd(&out[0], tried); atomicAdd(&out[1], bad);
}

int main()
{
    unsigned long long *out, h[2];
    CK(hipMalloc(&out, 16)); CK(hipMemset(out, 0, 16));
    hipLaunchKernelGGL(k_order, dim3(256 * 8), dim3(256), 0, 0, 900, out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("{\"lane_instructions_checked\": %llu, \"lanes_out_of_order\": %llu}\n", h[0], h[1]);
    return h[1] != 0;
}
