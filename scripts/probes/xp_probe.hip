// Probe: run exact_select_par alone on synthetic keys / table and watch its progress from the host.
#define KMC_XP_TRACE
#include "../../kmc_amd/csrc/kmc_exact_par.h"
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>
#include <math.h>
using namespace kmc;
__global__ __launch_bounds__(256) void k(const uint32_t* keys, const double* table, int n, double rho, long long* out, volatile long long* trace)
{
    __shared__ ExactParShared sh;
    if (threadIdx.x == 0) g_xp_trace = trace;
    __syncthreads();
    exact_select_par(keys, table, n, rho, &sh);
    if (threadIdx.x == 0) { out[0] = sh.idx; out[1] = __double_as_longlong(sh.gamma); out[2] = __double_as_longlong(sh.c_found); out[3] = 777; }
}
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 70 * 66;
    const double rho = argc > 2 ? atof(argv[2]) : 0.37;
    uint32_t* hk = (uint32_t*)malloc(n * 4);
    double* ht = (double*)calloc((size_t)(1 << 19) * 6, 8);
    srand(5);
    for (int r = 0; r < (1 << 18); ++r) for (int d = 0; d < 6; ++d) ht[(size_t)r * 6 + d] = (rand() % 3 == 0) ? 0.0 : exp(-10.0 + 30.0 * rand() / RAND_MAX);
    for (int s = 0; s < n; ++s) hk[s] = (rand() % 2) ? 0xFFFFFFFFu : (uint32_t)(rand() % (1 << 18));
    // CPU replay
    double gamma_seq = 0; { for (int s = 0; s < n; ++s) if (hk[s] != 0xFFFFFFFFu) for (int d = 0; d < 6; ++d) gamma_seq += ht[(size_t)hk[s] * 6 + d]; }
    uint32_t* dk; double* dt; long long* dout; long long* trace;
    hipMalloc(&dk, n * 4); hipMalloc(&dt, (size_t)(1 << 19) * 48); hipMalloc(&dout, 64);
    hipHostMalloc((void**)&trace, 64, hipHostMallocMapped); memset((void*)trace, 0, 64);
    hipMemcpy(dk, hk, n * 4, hipMemcpyHostToDevice); hipMemcpy(dt, ht, (size_t)(1 << 19) * 48, hipMemcpyHostToDevice);
    hipMemset(dout, 0, 64);
    k<<<1, 256>>>(dk, dt, n, rho, dout, trace);
    for (int it = 0; it < 50; ++it) {
        usleep(100000);
        if (hipStreamQuery(0) == hipSuccess) break;
        if (it % 10 == 9) printf("t=%.1fs stage %lld tile0 %lld pos %lld c %g iters %lld\n", it * 0.1, trace[0], trace[1], trace[2], *(double*)&trace[3], trace[4]);
    }
    if (hipStreamQuery(0) != hipSuccess) { printf("STUCK: stage %lld tile0 %lld pos %lld c %g iters %lld\n", trace[0], trace[1], trace[2], *(double*)&trace[3], trace[4]); fflush(stdout); _exit(3); }
    long long ho[8]; hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
    double g = *(double*)&ho[1];
    // sequential reference
    double c = 0; long idx = 6L * n;
    for (long e = 0; e < 6L * n; ++e) { uint32_t kk = hk[e / 6]; double pv = kk == 0xFFFFFFFFu ? 0.0 : ht[(size_t)kk * 6 + e % 6]; c += pv; if (c / g >= rho) { idx = e; break; } }
    printf("cycles: np.sum phase %lld, total %lld\n", trace[5], trace[6]);
    printf("gpu idx %lld gamma %.17g c %.17g | cpu idx %ld c %.17g (sequential gamma %.17g) iters %lld %s\n", ho[0], g, *(double*)&ho[2], idx, c, gamma_seq, trace[4],
           (ho[0] == idx && *(double*)&ho[2] == c) ? "MATCH" : "MISMATCH");
    return 0;
}
