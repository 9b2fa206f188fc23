// kmc_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Geometry tables restate the reference's conventions (file:line in the
// reference tree):
//   18-site key order     hex_LiCoO2_kmc.py:423-424
//   hop directions        hex_LiCoO2_kmc.py:503-504
//   end-point key slots   hex_dict_builder.py:41
//   12 rotated views      hex_dict_builder.py:56-134
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KMC_VACANT 0xFFFFFFFFu
#define KMC_WAVE 64

namespace kmc {

// dir -> (di, dj)
__device__ __constant__ const int8_t kDirI[6] = {1, 0, -1, -1, 0, 1};
__device__ __constant__ const int8_t kDirJ[6] = {0, 1, 1, 0, -1, -1};
// key char p -> (di, dj)
__device__ __constant__ const int8_t kNbI[18] = {0, -1, -2, 1, 0, -1, -2, 2, 1, -1, -2, 2, 1, 0, -1, 2, 1, 0};
__device__ __constant__ const int8_t kNbJ[18] = {2, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, -1, -1, -1, -1, -2, -2, -2};
__device__ __constant__ const int8_t kEndp[6] = {8, 4, 5, 9, 13, 12};
__device__ __constant__ const int8_t kViews[6][2][9] = {
    {{5, 9, 4, 13, 3, 16, 7, 11, 15}, {13, 9, 12, 5, 11, 1, 7, 3, 0}},
    {{9, 13, 5, 12, 1, 11, 0, 3, 7}, {12, 13, 8, 9, 3, 6, 0, 1, 2}},
    {{13, 12, 9, 8, 6, 3, 2, 1, 0}, {8, 12, 4, 13, 1, 14, 2, 6, 10}},
    {{12, 8, 13, 4, 14, 1, 10, 6, 2}, {4, 8, 5, 12, 6, 16, 10, 14, 17}},
    {{8, 4, 12, 5, 16, 6, 17, 14, 10}, {5, 4, 9, 8, 14, 11, 17, 16, 15}},
    {{4, 5, 8, 9, 11, 14, 15, 16, 17}, {9, 5, 13, 4, 16, 3, 15, 11, 7}}};

// ---------------------------------------------------------------- wave64 ops
// DPP controls (GFX9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Inclusive prefix sum over the 64 lanes, fixed association (deterministic).
// Lanes without a source read +0.0, which is exact for the non-negative sums
// this engine scans.
__device__ __forceinline__ double wave_scan_incl(double x)
{
    x += dpp_f64<0x111, 0xF>(x);
    x += dpp_f64<0x112, 0xF>(x);
    x += dpp_f64<0x114, 0xF>(x);
    x += dpp_f64<0x118, 0xF>(x);
    x += dpp_f64<0x142, 0xA>(x);
    x += dpp_f64<0x143, 0xC>(x);
    return x;
}

// Inclusive prefix sum valid in lanes 0..7 only (lanes 8..15 also see lanes below them).
__device__ __forceinline__ double scan8_incl(double x)
{
    x += dpp_f64<0x111, 0xF>(x);
    x += dpp_f64<0x112, 0xF>(x);
    x += dpp_f64<0x114, 0xF>(x);
    return x;
}

// Sum over each aligned group of 8 lanes, association ((x0+x1)+(x2+x3))+((x4+x5)+(x6+x7));
// every lane of the group receives the same bits (fp add is commutative).
__device__ __forceinline__ double group8_sum(double x)
{
    x += dpp_f64<0xB1, 0xF>(x);      // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E, 0xF>(x);      // quad_perm [2,3,0,1]
    x += dpp_f64<0x141, 0xF>(x);     // row_half_mirror
    return x;
}

// Sum of 8 consecutive doubles (a row's block sums), the same tree as group8_sum.
__device__ __forceinline__ double row_sum8(const double* b)
{
    return ((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]));
}

__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double x) { return readlane_f64(wave_scan_incl(x), 63); }

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------- Philox4x32-10
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

#define KMC_STREAM_HOP  0x4B4D4300u   // "KMC\0": per-hop draws
#define KMC_STREAM_FILL 0x4B4D4301u   // random fill keys

// 53-bit uniform in [0, 1) from two words (same bit recipe as numpy's legacy
// random_sample: (a >> 5) * 2^26 + (b >> 6))
__host__ __device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}
// clock draw in (0, 1]: log() stays finite
__host__ __device__ __forceinline__ double u53_open0(uint32_t a, uint32_t b)
{
    return (double)((((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) + 1) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------- table fold
// One direction of one 18-bit key: hex_dict_builder.py:45-139.
__device__ __forceinline__ double fold_rate(uint32_t row, int e, const double* base512, double dumbbell)
{
    auto bit = [&](int p) -> uint32_t { return (row >> (17 - p)) & 1u; };
    const uint32_t end = bit(kEndp[e]);
    const uint32_t prev = bit(kEndp[(e + 5) % 6]);
    const uint32_t next = bit(kEndp[(e + 1) % 6]);
    if (end) return 0.0;                                   // :138-139
    if (prev && next) return dumbbell;                     // :52-53
    double r1 = 0.0, r2 = 0.0;                             // sentinel '3' -> 0 (:21, :46-47)
    if (!prev) {
        uint32_t ix = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) ix |= bit(kViews[e][0][q]) << (8 - q);
        r1 = base512[ix];
    }
    if (!next) {
        uint32_t ix = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) ix |= bit(kViews[e][1][q]) << (8 - q);
        r2 = base512[ix];
    }
    return r1 + r2;                                        // :136-137
}

// ---------------------------------------------------------------- compact per-direction table
// The folded rate of direction e depends on 13 of the key's 18 bits only: the end point, its two
// side sites and the 2 x 9 sites of the two rotated views (hex_dict_builder.py:45-139).  So the
// dense table T[2^18][6] (12.6 MB, larger than an XCD's 4 MB L2) factors into Tc[6][2^13]
// (393 KB, L2 / L1 resident).  compact_mask(e) = the key bits (bit 17 - p for key char p)
// direction e reads; the index is those bits packed together in ascending order.
__host__ __device__ inline uint32_t compact_mask(int e)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int8_t* endp = kEndp;
    const int8_t (*views)[2][9] = kViews;
#else
    static const int8_t endp[6] = {8, 4, 5, 9, 13, 12};
    static const int8_t views[6][2][9] = {
        {{5, 9, 4, 13, 3, 16, 7, 11, 15}, {13, 9, 12, 5, 11, 1, 7, 3, 0}},
        {{9, 13, 5, 12, 1, 11, 0, 3, 7}, {12, 13, 8, 9, 3, 6, 0, 1, 2}},
        {{13, 12, 9, 8, 6, 3, 2, 1, 0}, {8, 12, 4, 13, 1, 14, 2, 6, 10}},
        {{12, 8, 13, 4, 14, 1, 10, 6, 2}, {4, 8, 5, 12, 6, 16, 10, 14, 17}},
        {{8, 4, 12, 5, 16, 6, 17, 14, 10}, {5, 4, 9, 8, 14, 11, 17, 16, 15}},
        {{4, 5, 8, 9, 11, 14, 15, 16, 17}, {9, 5, 13, 4, 16, 3, 15, 11, 7}}};
#endif
    uint32_t m = (1u << (17 - endp[e])) | (1u << (17 - endp[(e + 5) % 6])) | (1u << (17 - endp[(e + 1) % 6]));
    for (int v = 0; v < 2; ++v)
        for (int q = 0; q < 9; ++q) m |= 1u << (17 - views[e][v][q]);
    return m;
}
#define KMC_COMPACT_BITS 13
#define KMC_COMPACT_ROWS (1u << KMC_COMPACT_BITS)
// Pair tables for hop_q64_kernel, whose lanes read the two consecutive entries 2m, 2m+1 of a site with
// one 16-byte gather: directions 2m and 2m+1 together read 15 key bits.  Pair 1's bits come in four runs;
// adding key bit 2 (which its rates ignore) joins two of them, so that every pair's index is at most THREE
// bit-field extracts, at the price of a 16-bit index: Tc2[3][2^16] x {rate 2m, rate 2m+1} = 3 MB allocated,
// 2 MB touched (pairs 0 and 2 use 2^15 rows each), instead of the 12.6 MB dense table.
__host__ __device__ inline uint32_t compact2_mask(int m)
{
    return compact_mask(2 * m) | compact_mask(2 * m + 1) | (m == 1 ? 4u : 0u);
}
#define KMC_COMPACT2_BITS 16
#define KMC_COMPACT2_ROWS (1u << KMC_COMPACT2_BITS)

__host__ __device__ inline uint32_t pext18(uint32_t key, uint32_t mask)     // slow, table builders only
{
    uint32_t out = 0, o = 0;
    for (int b = 0; b < 18; ++b)
        if ((mask >> b) & 1u) { out |= ((key >> b) & 1u) << o; ++o; }
    return out;
}
__host__ __device__ inline uint32_t pdep18(uint32_t idx, uint32_t mask)
{
    uint32_t out = 0, o = 0;
    for (int b = 0; b < 18; ++b)
        if ((mask >> b) & 1u) { out |= ((idx >> o) & 1u) << b; ++o; }
    return out;
}

// Sum of the six rates of a table row in selection order; the hop kernels keep
// this per site, so it must be THE association used everywhere.
__device__ __forceinline__ double row_sum6(const double* r)
{
    return ((((r[0] + r[1]) + r[2]) + r[3]) + r[4]) + r[5];
}

// ---------------------------------------------------------------- keys
__device__ __forceinline__ int wrap1(int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); }

// 18-bit key from byte occupancy (any nx, ny >= 3): hex_LiCoO2_kmc.py:422-427
__device__ __forceinline__ uint32_t key_from_bytes(const uint8_t* occ, int nx, int ny, int i, int j)
{
    uint32_t key = 0;
#pragma unroll
    for (int p = 0; p < 18; ++p) {
        int ii = (i + kNbI[p] + nx) % nx, jj = (j + kNbJ[p] + ny) % ny;
        key = (key << 1) | (occ[ii * ny + jj] & 1u);
    }
    return key;
}

// 18-bit key from column bitmaps (bit i of col[j] = occupancy of (i, j)); needs
// 4 <= nx <= 64.  The key's chars run column by column (dj = +2 .. -2) with di
// descending inside each column, so each column contributes one contiguous bit
// field of a 5-row window starting at row i-2.
template <class ColFn>
__device__ __forceinline__ uint32_t key_from_cols(ColFn col, int nx, int ny, int i, int j)
{
    int st = i - 2;
    if (st < 0) st += nx;
    const int sl = (nx - st) & 63;
    auto win = [&](int dj) -> uint32_t {
        const uint64_t c = col(wrap1(j + dj, ny));
        return (uint32_t)((c >> st) | (c << sl)) & 31u;
    };
    const uint32_t w2 = win(2), w1 = win(1), w0 = win(0), m1 = win(-1), m2 = win(-2);
    return ((w2 & 7u) << 15) | ((w1 & 15u) << 11) | (((w0 >> 3) & 3u) << 9) | ((w0 & 3u) << 7) |
           (((m1 >> 1) & 15u) << 3) | ((m2 >> 2) & 7u);
}

// ---------------------------------------------------------------- np.sum model
// numpy's float64 pairwise sum (numpy/_core/src/umath/loops_utils.h.src) as
// np.sum applies it to a contiguous vector: pairwise inside each 8192-element
// buffer, buffers accumulated left to right (pinned against
// np.sum in tests/test_oracle_c.py).  `entry(e)` returns element e.
struct PwFrame { long lo, n; double left; int stage; };

template <class EntryFn>
__device__ double np_pairwise(long lo0, long n0, EntryFn entry, PwFrame* st)
{
    int sp = 0;
    st[0].lo = lo0; st[0].n = n0; st[0].stage = 0; st[0].left = 0;
    double ret = 0;
    while (sp >= 0) {
        const long lo = st[sp].lo, n = st[sp].n;
        const int stage = st[sp].stage;
        if (stage == 0) {
            if (n < 8) {
                double res = 0.;
                for (long i = 0; i < n; ++i) res += entry(lo + i);
                ret = res; --sp;
            } else if (n <= 128) {
                double r0 = entry(lo), r1 = entry(lo + 1), r2 = entry(lo + 2), r3 = entry(lo + 3),
                       r4 = entry(lo + 4), r5 = entry(lo + 5), r6 = entry(lo + 6), r7 = entry(lo + 7);
                long i;
                for (i = 8; i < n - (n % 8); i += 8) {
                    r0 += entry(lo + i); r1 += entry(lo + i + 1); r2 += entry(lo + i + 2); r3 += entry(lo + i + 3);
                    r4 += entry(lo + i + 4); r5 += entry(lo + i + 5); r6 += entry(lo + i + 6); r7 += entry(lo + i + 7);
                }
                double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                for (; i < n; ++i) res += entry(lo + i);
                ret = res; --sp;
            } else {
                long n2 = n / 2; n2 -= n2 % 8;
                st[sp].stage = 1;
                ++sp; st[sp].lo = lo; st[sp].n = n2; st[sp].stage = 0;
            }
        } else if (stage == 1) {
            long n2 = n / 2; n2 -= n2 % 8;
            st[sp].left = ret; st[sp].stage = 2;
            ++sp; st[sp].lo = lo + n2; st[sp].n = n - n2; st[sp].stage = 0;
        } else {
            ret = st[sp].left + ret; --sp;
        }
    }
    return ret;
}

#define KMC_PW_DEPTH 24

template <class EntryFn>
__device__ double np_sum_model(long n, EntryFn entry, PwFrame* st)
{
    double res = 0.;
    for (long lo = 0; lo < n; lo += 8192) {
        const long m = (n - lo) < 8192 ? (n - lo) : 8192;
        res = res + np_pairwise(lo, m, entry, st);
    }
    return res;
}

}  // namespace kmc
