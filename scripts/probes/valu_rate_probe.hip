// Probe: what does one wave64 vector instruction cost a SIMD, by kind, at 1, 2 and 4 waves per SIMD?
// Every block is one wave; W waves per SIMD = 256 CUs x 4 SIMDs x W blocks (amdgpu_waves_per_eu pins W by
// launch size only: blocks are distributed round-robin).  Each kernel runs ITER x 64 independent instructions of
// one kind between two s_memtime stamps; cycles per instruction per SIMD = (t1 - t0) * ... / (ITER * 64 * W).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define ITER 2000
#define REP64(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x

#define KERNEL(name, body, setup)                                                                 \
__global__ __launch_bounds__(64) void name(unsigned long long* out, int n)                        \
{                                                                                                 \
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;               \
    setup;                                                                                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
    for (int it = 0; it < n; ++it) {                                                              \
        asm volatile(REP64(body)                                                                  \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),     \
              "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) :: "vcc", "scc", "s20", "s21"); \
    }                                                                                             \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                              \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345 && d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 1.5) out[0] = 0; \
}

// 8 independent chains, so that dependent-issue latency does not show (each line = 8 instructions; REP64 / 8 ...)
// -> body below is ONE instruction; REP64 makes 64 of them on rotating registers by listing 8 different ones 8 times
#define B8(i0, i1, i2, i3, i4, i5, i6, i7) i0 i1 i2 i3 i4 i5 i6 i7
#undef REP64
#define REP64(x) x x x x x x x x

KERNEL(k_add_u32, B8("v_add_u32 %0, %0, %1\n", "v_add_u32 %1, %1, %2\n", "v_add_u32 %2, %2, %3\n", "v_add_u32 %3, %3, %4\n",
                     "v_add_u32 %4, %4, %5\n", "v_add_u32 %5, %5, %6\n", "v_add_u32 %6, %6, %7\n", "v_add_u32 %7, %7, %0\n"), )
KERNEL(k_mov_dpp, B8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n",
                     "v_mov_b32_dpp %2, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "v_mov_b32_dpp %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n",
                     "v_mov_b32_dpp %4, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "v_mov_b32_dpp %5, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n",
                     "v_mov_b32_dpp %6, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n", "v_mov_b32_dpp %7, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"), )
KERNEL(k_add_f64, B8("v_add_f64 %8, %8, %9\n", "v_add_f64 %9, %9, %10\n", "v_add_f64 %10, %10, %11\n", "v_add_f64 %11, %11, %12\n",
                     "v_add_f64 %12, %12, %13\n", "v_add_f64 %13, %13, %14\n", "v_add_f64 %14, %14, %15\n", "v_add_f64 %15, %15, %8\n"), )
KERNEL(k_cmp_f64, B8("v_cmp_le_f64 vcc, %8, %9\n", "v_cmp_le_f64 vcc, %9, %10\n", "v_cmp_le_f64 vcc, %10, %11\n", "v_cmp_le_f64 vcc, %11, %12\n",
                     "v_cmp_le_f64 vcc, %12, %13\n", "v_cmp_le_f64 vcc, %13, %14\n", "v_cmp_le_f64 vcc, %14, %15\n", "v_cmp_le_f64 vcc, %15, %8\n"), )
KERNEL(k_readlane, B8("v_readlane_b32 s20, %0, 5\n", "v_readlane_b32 s21, %1, 6\n", "v_readlane_b32 s20, %2, 7\n", "v_readlane_b32 s21, %3, 8\n",
                      "v_readlane_b32 s20, %4, 9\n", "v_readlane_b32 s21, %5, 10\n", "v_readlane_b32 s20, %6, 11\n", "v_readlane_b32 s21, %7, 12\n"), )
KERNEL(k_bfe, B8("v_bfe_u32 %0, %1, %2, 5\n", "v_bfe_u32 %1, %2, %3, 5\n", "v_bfe_u32 %2, %3, %4, 5\n", "v_bfe_u32 %3, %4, %5, 5\n",
                 "v_bfe_u32 %4, %5, %6, 5\n", "v_bfe_u32 %5, %6, %7, 5\n", "v_bfe_u32 %6, %7, %0, 5\n", "v_bfe_u32 %7, %0, %1, 5\n"), )
KERNEL(k_cndmask, B8("v_cndmask_b32 %0, %0, %1, vcc\n", "v_cndmask_b32 %1, %1, %2, vcc\n", "v_cndmask_b32 %2, %2, %3, vcc\n", "v_cndmask_b32 %3, %3, %4, vcc\n",
                     "v_cndmask_b32 %4, %4, %5, vcc\n", "v_cndmask_b32 %5, %5, %6, vcc\n", "v_cndmask_b32 %6, %6, %7, vcc\n", "v_cndmask_b32 %7, %7, %0, vcc\n"), )
KERNEL(k_mul_f64, B8("v_mul_f64 %8, %8, %9\n", "v_mul_f64 %9, %9, %10\n", "v_mul_f64 %10, %10, %11\n", "v_mul_f64 %11, %11, %12\n",
                     "v_mul_f64 %12, %12, %13\n", "v_mul_f64 %13, %13, %14\n", "v_mul_f64 %14, %14, %15\n", "v_mul_f64 %15, %15, %8\n"), )
KERNEL(k_salu, B8("s_add_u32 s20, s20, 1\n", "s_add_u32 s21, s21, 1\n", "s_add_u32 s20, s20, 1\n", "s_add_u32 s21, s21, 1\n",
                  "s_add_u32 s20, s20, 1\n", "s_add_u32 s21, s21, 1\n", "s_add_u32 s20, s20, 1\n", "s_add_u32 s21, s21, 1\n"), )
// a scan step as the hop kernels issue it: two DPP moves feeding one fp64 add (dependent)
KERNEL(k_scanstep, B8("v_mov_b32_dpp %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "v_mov_b32_dpp %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n",
                      "v_add_f64 %8, %8, %9\n", "s_nop 1\n", "v_mov_b32_dpp %4, %6 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n",
                      "v_mov_b32_dpp %5, %7 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "v_add_f64 %10, %10, %11\n", "s_nop 1\n"), )
// LDS cross-lane moves instead (no memory access): 8 swizzles
KERNEL(k_swizzle, B8("ds_swizzle_b32 %0, %1 offset:0x041F\n", "ds_swizzle_b32 %1, %2 offset:0x041F\n", "ds_swizzle_b32 %2, %3 offset:0x041F\n", "ds_swizzle_b32 %3, %4 offset:0x041F\n",
                     "ds_swizzle_b32 %4, %5 offset:0x041F\n", "ds_swizzle_b32 %5, %6 offset:0x041F\n", "ds_swizzle_b32 %6, %7 offset:0x041F\n", "ds_swizzle_b32 %7, %0 offset:0x041F\ns_waitcnt lgkmcnt(0)\n"), )

// do vector and scalar instructions of ONE wave / of different waves share the SIMD's issue slots?
KERNEL(k_mix_vs, B8("v_bfe_u32 %0, %1, %2, 5\n", "s_add_u32 s20, s20, 1\n", "v_bfe_u32 %1, %2, %3, 5\n", "s_add_u32 s21, s21, 1\n",
                    "v_bfe_u32 %2, %3, %4, 5\n", "s_add_u32 s20, s20, 1\n", "v_bfe_u32 %3, %4, %5, 5\n", "s_add_u32 s21, s21, 1\n"), )
KERNEL(k_snop, B8("s_nop 1\n", "s_nop 1\n", "s_nop 1\n", "s_nop 1\n", "s_nop 1\n", "s_nop 1\n", "s_nop 1\n", "s_nop 1\n"), )
KERNEL(k_snop0, B8("s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n"), )
KERNEL(k_waitcnt, B8("s_waitcnt lgkmcnt(0)\n", "s_waitcnt vmcnt(0)\n", "s_waitcnt lgkmcnt(0)\n", "s_waitcnt vmcnt(0)\n",
                     "s_waitcnt lgkmcnt(0)\n", "s_waitcnt vmcnt(0)\n", "s_waitcnt lgkmcnt(0)\n", "s_waitcnt vmcnt(0)\n"), )
// branches: never taken / always taken (to the next instruction)
KERNEL(k_br_nt, B8("s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc0 1f\n1:\n", "s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc0 1f\n1:\n",
                   "s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc0 1f\n1:\n", "s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc0 1f\n1:\n"), )
KERNEL(k_br_t, B8("s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc1 1f\n1:\n", "s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc1 1f\n1:\n",
                  "s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc1 1f\n1:\n", "s_cmp_eq_u32 s20, s20\n", "s_cbranch_scc1 1f\n1:\n"), )
KERNEL(k_vop2_mix, B8("v_add_u32 %0, %0, %1\n", "v_and_b32 %1, %1, %2\n", "v_or_b32 %2, %2, %3\n", "v_lshlrev_b32 %3, 1, %4\n",
                      "v_xor_b32 %4, %4, %5\n", "v_sub_u32 %5, %5, %6\n", "v_mov_b32 %6, %7\n", "v_add_u32 %7, %7, %0\n"), )
KERNEL(k_vop3_int, B8("v_lshl_or_b32 %0, %0, 3, %1\n", "v_and_or_b32 %1, %1, 7, %2\n", "v_lshl_add_u32 %2, %2, 1, %3\n", "v_alignbit_b32 %3, %3, %4, %5\n",
                      "v_add3_u32 %4, %4, %5, %6\n", "v_or3_b32 %5, %5, %6, %7\n", "v_mad_u32_u24 %6, %6, 12, %7\n", "v_bfe_u32 %7, %0, 2, 5\n"), )
KERNEL(k_writelane, B8("v_writelane_b32 %0, s20, 5\n", "v_writelane_b32 %1, s21, 6\n", "v_writelane_b32 %2, s20, 7\n", "v_writelane_b32 %3, s21, 8\n",
                       "v_writelane_b32 %4, s20, 9\n", "v_writelane_b32 %5, s21, 10\n", "v_writelane_b32 %6, s20, 11\n", "v_writelane_b32 %7, s21, 12\n"), )
KERNEL(k_cmp_u32_s, B8("v_cmp_eq_u32 s[20:21], %0, %1\n", "v_cmp_eq_u32 s[20:21], %1, %2\n", "v_cmp_eq_u32 s[20:21], %2, %3\n", "v_cmp_eq_u32 s[20:21], %3, %4\n",
                       "v_cmp_eq_u32 s[20:21], %4, %5\n", "v_cmp_eq_u32 s[20:21], %5, %6\n", "v_cmp_eq_u32 s[20:21], %6, %7\n", "v_cmp_eq_u32 s[20:21], %7, %0\n"), )
KERNEL(k_bpermute, B8("ds_bpermute_b32 %0, %1, %2\n", "ds_bpermute_b32 %1, %2, %3\n", "ds_bpermute_b32 %2, %3, %4\n", "ds_bpermute_b32 %3, %4, %5\n",
                      "ds_bpermute_b32 %4, %5, %6\n", "ds_bpermute_b32 %5, %6, %7\n", "ds_bpermute_b32 %6, %7, %0\n", "ds_bpermute_b32 %7, %0, %1\ns_waitcnt lgkmcnt(0)\n"), )

typedef void (*kern_t)(unsigned long long*, int);
int main()
{
    struct { const char* name; kern_t k; int per_body; } ks[] = {
        {"v_add_u32", k_add_u32, 64}, {"v_mov_b32_dpp", k_mov_dpp, 64}, {"v_add_f64", k_add_f64, 64}, {"v_cmp_le_f64", k_cmp_f64, 64},
        {"v_readlane_b32", k_readlane, 64}, {"v_bfe_u32", k_bfe, 64}, {"v_cndmask_b32", k_cndmask, 64}, {"v_mul_f64", k_mul_f64, 64},
        {"s_add_u32", k_salu, 64}, {"scan step (2 dpp + add_f64 + s_nop 1) x2", k_scanstep, 16}, {"ds_swizzle_b32", k_swizzle, 64},
        {"v_bfe_u32 + s_add_u32 alternating (per PAIR)", k_mix_vs, 32}, {"s_nop 1", k_snop, 64}, {"s_nop 0", k_snop0, 64}, {"s_waitcnt (nothing pending)", k_waitcnt, 64},
        {"s_cmp + s_cbranch not taken (per pair)", k_br_nt, 32}, {"s_cmp + s_cbranch taken (per pair)", k_br_t, 32},
        {"VOP2 integer mix", k_vop2_mix, 64}, {"VOP3 integer mix", k_vop3_int, 64}, {"v_writelane_b32", k_writelane, 64},
        {"v_cmp_eq_u32 -> sgpr pair", k_cmp_u32_s, 64}, {"ds_bpermute_b32", k_bpermute, 64}};
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long* d;
    const int maxb = 256 * 4 * 8;
    if (hipMalloc(&d, maxb * 8) != hipSuccess) return 2;
    unsigned long long* h = (unsigned long long*)malloc(maxb * 8);
    printf("%-44s %10s %10s %10s %10s   (shader cycles per instruction per SIMD = wave cycles / (instructions x waves per SIMD); last: 1 wave on the chip)\n", "instruction", "W=1", "W=2", "W=4", "alone");
    for (auto& e : ks) {
        printf("%-44s", e.name);
        for (int w : {1, 2, 4, 0}) {
            const int blocks = w ? 256 * 4 * w : 1;
            e.k<<<blocks, 64>>>(d, 16);                       // warm-up
            e.k<<<blocks, 64>>>(d, ITER);
            if (hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
            double s = 0;
            for (int b = 0; b < blocks; ++b) s += (double)h[b];
            // wave cycles per instruction; with w waves sharing a SIMD each instruction costs the SIMD 1/w of that
            printf(" %10.2f", s / blocks / ((double)ITER * e.per_body) / (w ? w : 1));
        }
        printf("\n");
    }
    return 0;
}
