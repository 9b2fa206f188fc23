// Probe: what can a kernel of the rate fill's stream shape (1 B read + 4 B written per site) reach at the fill's sizes,
// timed the way the library times its kernels (a pair of events round one launch)?  An empty launch gives the floor of
// that timing; the copy kernels move the same bytes with perfectly coalesced 16-byte accesses and do nothing else.
// Build: hipcc -O3 --offload-arch=gfx950 -o stream_floor_probe stream_floor_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 0; }

// thread = 16 sites: one 16-byte load, four 16-byte streaming stores
__global__ __launch_bounds__(256) void shape_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n16; t += (size_t)gridDim.x * 256) {
        const uint4 v = in[t];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint4 o;
            o.x = w[k] & 0xFF; o.y = (w[k] >> 8) & 0xFF; o.z = (w[k] >> 16) & 0xFF; o.w = w[k] >> 24;
            __builtin_nontemporal_store(o.x, &out[t * 4 + k].x);
            __builtin_nontemporal_store(o.y, &out[t * 4 + k].y);
            __builtin_nontemporal_store(o.z, &out[t * 4 + k].z);
            __builtin_nontemporal_store(o.w, &out[t * 4 + k].w);
        }
    }
}

// the same with one dword store per lane and site (what a wave-instruction of the fill writes: 256 contiguous bytes)
__global__ __launch_bounds__(256) void shape_copy_dw(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, size_t n)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store((uint32_t)in[t], &out[t]);
}

int main()
{
    const size_t sites[] = {4096ull * 900, 4096ull * 1600, 64ull * 65536, 1ull << 20, 4096ull * 4096, 16384ull * 4096};
    const char* names[] = {"4096 x 30x30", "4096 x 40x40", "64 x 256x256", "1 x 1024x1024", "4096 x 64x64", "16384 x 64x64"};
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    uint8_t* in; uint32_t* out;
    const size_t nmax = 16384ull * 4096;
    hipMalloc(&in, nmax); hipMalloc(&out, nmax * 4);
    hipMemset(in, 1, nmax);
    auto timeit = [&](auto&& launch) {
        std::vector<float> ms;
        for (int k = 0; k < 24; ++k) {
            hipEventRecord(e0, st);
            launch();
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float m; hipEventElapsedTime(&m, e0, e1);
            if (k >= 4) ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        return std::make_pair(ms[0], ms[ms.size() / 2]);
    };
    auto a = timeit([&] { empty_kernel<<<1, 64, 0, st>>>(nullptr); });
    printf("empty launch, 1 x 64:      min %.2f us median %.2f us\n", a.first * 1e3, a.second * 1e3);
    a = timeit([&] { empty_kernel<<<4096, 64, 0, st>>>(nullptr); });
    printf("empty launch, 4096 x 64:   min %.2f us median %.2f us\n", a.first * 1e3, a.second * 1e3);
    a = timeit([&] { empty_kernel<<<1024, 256, 0, st>>>(nullptr); });
    printf("empty launch, 1024 x 256:  min %.2f us median %.2f us\n", a.first * 1e3, a.second * 1e3);
    for (int c = 0; c < 6; ++c) {
        const size_t n = sites[c];
        const double by = 5.0 * (double)n;
        for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
            const unsigned grid = (unsigned)std::min<size_t>((n / 16 + 255) / 256, (size_t)256 * blocks_per_cu);
            a = timeit([&] { shape_copy<<<grid, 256, 0, st>>>((const uint4*)in, (uint4*)out, n / 16); });
            printf("%-14s copy16 grid %5u: min %6.2f us median %6.2f us  %.2f TB/s  %.3f of 8 TB/s\n", names[c], grid, a.first * 1e3,
                   a.second * 1e3, by / (a.second * 1e-3) / 1e12, by / (a.second * 1e-3) / 8e12);
        }
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)256 * 8);
        a = timeit([&] { shape_copy_dw<<<grid, 256, 0, st>>>(in, out, n); });
        printf("%-14s copy-dw grid %5u: min %6.2f us median %6.2f us  %.2f TB/s  %.3f of 8 TB/s\n", names[c], grid, a.first * 1e3,
               a.second * 1e3, by / (a.second * 1e-3) / 1e12, by / (a.second * 1e-3) / 8e12);
    }
    return 0;
}
