// Probe: how many cycles does ISSUING a vector memory instruction cost one wave on gfx950
// (s_memtime around the instruction, no wait on its result)?  One wave per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 256
__global__ __launch_bounds__(64) void k(const double* tab, double* out, long long* cyc, int4* tags)
{
    const int lane = threadIdx.x;
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    double sum = 0;
    uint32_t idx = (lane * 2654435761u + blockIdx.x * 97u) & 0xFFFFF;
    int4 tg = make_int4(0, 0, 0, 0);
    for (int it = 0; it < REP; ++it) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        double a = tab[idx];                                  // 64-lane gather, issue only
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double b = tab[(idx * 7u + 3u) & 0xFFFFF];            // second gather back to back
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (lane < 5) tg.x = reinterpret_cast<int*>(tags)[(((idx + it) & 0xFFFF) << 2) + lane];   // 5-lane dword load
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        if (lane < 5) reinterpret_cast<int*>(tags)[(((idx * 3u + it) & 0xFFFF) << 2) + lane] = it;   // 5-lane dword store
        unsigned long long t4 = __builtin_amdgcn_s_memtime();
        sum += a + b + tg.x;                                  // wait for everything
        unsigned long long t5 = __builtin_amdgcn_s_memtime();
        idx = (idx * 1664525u + 1013904223u + (uint32_t)(sum != 1.5)) & 0xFFFFF;
        for (int w = 0; w < 40; ++w) asm volatile("s_nop 15");   // let the memory pipes go idle
        unsigned long long t6 = __builtin_amdgcn_s_memtime();
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t6 - t5;
    }
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) for (int q = 0; q < 6; ++q) cyc[blockIdx.x * 6 + q] = (long long)acc[q];
}
int main()
{
    double *tab, *out; long long* cyc; int4* tags;
    hipMalloc(&tab, (1u << 20) * 8); hipMemset(tab, 0, (1u << 20) * 8);
    hipMalloc(&out, 256 * 64 * 8); hipMalloc(&cyc, 256 * 6 * 8);
    hipMalloc(&tags, 65536 * 16); hipMemset(tags, 0, 65536 * 16);
    for (int blocks : {256, 2304}) {
        hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 6 * 8);
        k<<<blocks, 64>>>(tab, out, cyc, tags);
        k<<<blocks, 64>>>(tab, out, cyc, tags);
        long long* h = (long long*)malloc(blocks * 6 * 8);
        if (hipMemcpy(h, cyc, blocks * 6 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        double m[6] = {0};
        for (int b = 0; b < blocks; ++b) for (int q = 0; q < 6; ++q) m[q] += (double)h[b * 6 + q] / REP / blocks;
        printf("blocks %4d: gather#1 %.0f  gather#2 %.0f  5-lane dword load %.0f  5-lane dword store %.0f  wait-all %.0f  idle-loop %.0f cycles\n",
               blocks, m[0], m[1], m[2], m[3], m[4], m[5]);
    }
    return 0;
}
