// Probe: does wave_shr:1 (DPP ctrl 0x138) shift across the 16-lane rows on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) {
    int x = threadIdx.x + 100;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(-7, x, 0x138, 0xF, 0xF, true);
}
int main() {
    int* d; int h[64];
    if (hipMalloc(&d, 256) != hipSuccess) return 2;
    k<<<1, 64>>>(d);
    if (hipMemcpy(h, d, 256, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    int bad = 0;
    for (int l = 0; l < 64; ++l) { int want = l == 0 ? 0 : l + 99; if (h[l] != want) { ++bad; printf("lane %d got %d want %d\n", l, h[l], want); } }
    printf("wave_shr:1 %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
