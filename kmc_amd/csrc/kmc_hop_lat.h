// kmc_hop_lat.h -- the fused hop loop for FEW replicas per GPU: one wave = one replica, ONE replica per CU.
//
// hop_q64_kernel (kmc_hop_q64.h) is built for 16 replicas per CU: 10 KB of LDS and 128 registers each, two L2
// gathers per hop (the pair table for the entry pick, W for the re-sum) that the other fifteen waves hide.  With
// 256 replicas on 256 CUs nothing hides them, a lone wave issues one instruction every ~5 cycles whatever its
// kind, and ~110 KB of the CU's LDS and three of its four SIMDs idle.  This kernel spends the idle LDS on
// everything a hop's dependent chain reads, so that ONE L2 gather per hop is left on it (W and the rates of the
// <= 36 refreshed sites, a function of their new keys) and the chain is ~40 % fewer instructions:
//   WR   16 B [64][72]  per site {W, RP}: W = f64 sum of the six rates (row_sum6 of the site's stale table row), RP = the
//                       rates as six 10-bit indices into a palette of the table's distinct values (the folded tables
//                       hold ~800 distinct rates: tables with more than 1024 keep q64); one 16-byte read per lane
//                       fetches a whole row of both
//   PAL  f64 [1024]     the palette
//   FR   u32 [64][72]   the site's FRESH 18-bit key (bit 17 - p = occupancy of neighbour p, hex_LiCoO2_kmc.py:423-424)
//                       and its own occupancy in bit 18, maintained by two LDS atomics per hop: the 18 sites that see
//                       the start point clear its bit, the 18 that see the end point set it.  A refreshed site's key
//                       is then ONE LDS read instead of five column reads and ~25 instructions of assembly
//   ST   u32 [64][72]   the stale key of each site (its key at its last refresh: the state the reference's window
//                       semantics need); only written in the loop, read by the exact path and the write-back
// (row stride 72: the <= 7 window rows fall on different banks).  Selection: row (ballot over the per-lane row
// prefixes) -> site (ONE read of the row's 64 W, wave scan) -> direction (the site's six palette entries, 3-step
// scan on six lanes), certified as in the other kernels (kmc_hop_lds.h) or replayed exactly.  Re-sum: a row's sum
// is the balanced tree over its 64 W (stride-8 leaves per lane, then the 8-lane tree), a pure function of the keys
// with a fixed association like every sum of the other kernels: no drift, split runs are bit-identical.  The
// association differs from q64's quarter sums: selections, lattices, keys, tags, histograms are identical, gamma
// and the clocks agree to rounding.
#pragma once
#include "kmc_device.h"
#include "kmc_hop_lds.h"
#include "kmc_hop_q64.h"
#include "kmc_exact_par.h"

namespace kmc {

#define LAT_PAL_MAX 1024
#define LAT_STRIDE 72
// FR, ST and PAL first: every address the hop loop computes in a register then takes its array's base as the 16-bit
// immediate offset of the LDS instruction
#define LAT_FR 0
#define LAT_ST (LAT_FR + 64 * LAT_STRIDE * 4)
#define LAT_PAL (LAT_ST + 64 * LAT_STRIDE * 4)
#define LAT_WR (LAT_PAL + 1024 * 8)                    // {W, RP} interleaved, 16 B per site
#define LAT_MISC (LAT_WR + 64 * LAT_STRIDE * 16)
#define LAT_IDLE (LAT_MISC + 96)                       // u32 [64]: where the idle lanes of the fresh-key atomics point
#define LAT_LDS_BYTES (LAT_IDLE + 256)
#define LAT_OCC_BIT 0x40000u

static_assert(LAT_LDS_BYTES <= 160 * 1024, "one replica per CU");
static_assert(LAT_WR + 8 < 65536, "immediate offsets of the LDS instructions");

// PK (two replicas per CU, for 257 ... 512 replicas per GPU): one 128-thread workgroup = two waves = two replicas that share
// the palette; each wave's planes are squeezed to 76 KB: row stride 64, the fresh keys as a u16 plane (key bits 0-15) and
// a 4-bit plane (key bits 16, 17 and the occupancy bit; eight sites per dword, the LDS atomics take any bit of a dword),
// and the stale keys not in LDS at all -- a refreshed site's key is stored straight to the replica's key array in global
// memory (one store per hop, issued behind the gather so that the wait for the gather does not wait for it).
#define LAT2_STRIDE 64
#define LAT2_PAL 0
#define LAT2_REGION0 (LAT_PAL_MAX * 8)
#define LAT2_A16 0
#define LAT2_B4 8192
#define LAT2_MISC (LAT2_B4 + 2048)
#define LAT2_IDLE (LAT2_MISC + 96)
#define LAT2_WR (LAT2_IDLE + 256)
#define LAT2_REGION (LAT2_WR + 64 * LAT2_STRIDE * 16)
#define LAT2_LDS_BYTES (LAT2_REGION0 + 2 * LAT2_REGION)

struct __attribute__((aligned(16))) LatWP {
    double w;                    // row_sum6 of the table row
    unsigned long long rp;       // the row's six rates as 10-bit palette indices, direction d at bits 10 d
};

struct LatParams {
    HopParams h;
    const LatWP* wp;             // [2^18]
    const double* pal;           // [LAT_PAL_MAX], pal[0] = 0.0
    int n_rep;                   // replicas of this launch (PK: two per workgroup, the last one may hold one)
};
static_assert(LAT2_WR % 16 == 0 && LAT2_REGION % 16 == 0 && LAT2_LDS_BYTES <= 160 * 1024, "two replicas per CU");

__device__ __noinline__ void exact_select_lat(const unsigned char* smem, const double* table, int ny, int n, double rho,
                                              ExactParShared* sh)
{
    const uint32_t* st = reinterpret_cast<const uint32_t*>(smem + LAT_ST);
    exact_select_par_impl<1, 2, true>([=](long s) -> uint32_t {
        const int i = (int)s / ny, j = (int)s - i * ny;
        return st[i * LAT_STRIDE + j];
    }, table, n, rho, sh);
}

// PK: the stale keys live in the replica's key array in global memory, where the hop loop stores them as it refreshes
// sites; read past the vector L1 (agent scope), which the wave's own stores do not update.
__device__ __noinline__ void exact_select_lat_g(const uint32_t* keys, const double* table, int n, double rho, ExactParShared* sh)
{
    exact_select_par_impl<1, 2, true>([=](long s) -> uint32_t {
        return __hip_atomic_load(keys + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }, table, n, rho, sh);
}

// first set bit of a wave mask, 63 if there is none (s_ff1_i32_b64 returns -1 for zero; __ffsll() - 1 costs a compare and
// a select more, three times a hop)
__device__ __forceinline__ int ff1_63(unsigned long long m)
{
    int r;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return r & 63;
}

// ANY = false: 64 x 64.  ANY = true: any 8 <= nx, ny <= 64 in the same 64 x 72 planes (sites beyond the lattice hold W = 0
// and are never selected or refreshed); site numbers in memory are row * ny + column.
// CLS: the refresh maps the window's columns to lanes by their class (column mod 8), see the hop loop; possible when the
// <= 6 columns of a window always fall in distinct classes: ny a multiple of 8, or ny mod 8 >= the widest window of the
// window mode (a window that wraps round the lattice's edge then still meets each class once).  Chosen by the host
// (lat_cls_ok); otherwise lane c takes window column c and the window's rows are re-summed from LDS.
__host__ __device__ inline bool lat_cls_ok(int ny, int window_mode)
{
    const int m = ny & 7, widest = window_mode == 0 ? 5 : 6;
    return m == 0 || m >= widest;
}

template <bool ANY, bool CLS, bool PK>
__global__ __launch_bounds__(PK ? 128 : 64)
void hop_lat_kernel(LatParams lp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const HopParams& p = lp.h;
    const int lane = threadIdx.x & 63;
    // (the wave's number as a scalar: to the compiler threadIdx.x >> 6 is a per-lane value, and with it every loop bound below)
    const int wave = PK ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int slot = PK ? (int)blockIdx.x * 2 + wave : (int)blockIdx.x;     // replica slot of this wave
    constexpr int S = PK ? LAT2_STRIDE : LAT_STRIDE;   // row stride of the planes, in sites
    double* PAL_l = reinterpret_cast<double*>(smem_all + (PK ? LAT2_PAL : LAT_PAL));
    if (PK) {
        // the palette is the one thing the two waves share: filled by both, then each wave is on its own (no barrier after)
        for (int k = threadIdx.x; k < LAT_PAL_MAX; k += 128) PAL_l[k] = lp.pal[k];
        __syncthreads();
        if (slot >= lp.n_rep) return;
    }
    unsigned char* const smem = smem_all + (PK ? LAT2_REGION0 + wave * LAT2_REGION : 0);
    const int r = p.order ? p.order[slot + p.r_base] : slot + p.r_base;
    const int nx = ANY ? p.nx : 64, ny = ANY ? p.ny : 64;
    const int n = ANY ? p.n : 4096;
    LatWP* WR_l = reinterpret_cast<LatWP*>(smem + (PK ? LAT2_WR : LAT_WR));
    uint32_t* FR_l = reinterpret_cast<uint32_t*>(smem + (PK ? LAT2_A16 : LAT_FR));    // PK: two u16 per dword
    uint32_t* FB_l = reinterpret_cast<uint32_t*>(smem + LAT2_B4);                      // PK only: eight nibbles per dword
    uint32_t* ST_l = reinterpret_cast<uint32_t*>(smem + LAT_ST);                       // not PK only
    // the 19-bit fresh word of site s (key | occupancy << 18), and the dword / bit a key bit lives at
    auto fr_word = [&](int s) -> uint32_t {
        if (!PK) return FR_l[s];
        const uint32_t a = reinterpret_cast<const uint16_t*>(FR_l)[s];
        return a | (((FB_l[s >> 3] >> ((s & 7) * 4)) & 7u) << 16);
    };
    auto fr_or = [&](int s, int bit) {                 // prologue: set bit `bit` (0 .. 18) of site s
        if (!PK) atomicOr(&FR_l[s], 1u << bit);
        else if (bit < 16) atomicOr(&FR_l[s >> 1], 1u << ((s & 1) * 16 + bit));
        else atomicOr(&FB_l[s >> 3], 1u << ((s & 7) * 4 + (bit - 16)));
    };
    ExactParShared* const xsh = reinterpret_cast<ExactParShared*>(reinterpret_cast<unsigned char*>(p.xscratch) + (size_t)r * Q64_XSCRATCH);
    uint8_t* const occ_g = p.occ + (size_t)r * n;
    uint32_t* const keys_g = p.keys + (size_t)r * n;
    int4* const tags_g = p.tags + (size_t)r * n;
    auto ext = [&](int so) -> int { const int i = so / S, j = so - i * S; return i * ny + j; };
    auto wrap_i = [&](int a) -> int { return ANY ? (a < 0 ? a + nx : (a >= nx ? a - nx : a)) : (a & 63); };
    auto wrap_j = [&](int a) -> int { return ANY ? (a < 0 ? a + ny : (a >= ny ? a - ny : a)) : (a & 63); };

    // ---- load the replica ------------------------------------------------
    if (!PK) {
        for (int k = lane; k < LAT_PAL_MAX; k += KMC_WAVE) PAL_l[k] = lp.pal[k];
        for (int k = lane; k < 64 * S; k += KMC_WAVE) { FR_l[k] = 0u; ST_l[k] = Q64_VAC; }
    } else {
        for (int k = lane; k < (8192 + 2048) / 4; k += KMC_WAVE) FR_l[k] = 0u;          // both fresh-key planes
    }
    for (int k = lane; k < 64 * S; k += KMC_WAVE) { WR_l[k].w = 0.0; WR_l[k].rp = 0ull; }
    wave_fence();
    // fresh keys: every occupied site sets its bit in the 18 sites that see it, and its own occupancy bit
    if (!ANY || lane < ny) {
        for (int i = 0; i < nx; ++i) {
            if (occ_g[i * ny + lane] & 1u) {
                fr_or(i * S + lane, 18);
#pragma unroll
                for (int q = 0; q < 18; ++q)
                    fr_or(wrap_i(i - kNbI[q]) * S + wrap_j(lane - kNbJ[q]), 17 - q);
            }
        }
    }
    // stale keys as they were left (kmc_rates_full or the last run), W and the palette words they imply
    if (!ANY || lane < ny) {
        for (int i0 = 0; i0 < nx; i0 += 8) {
            uint32_t k[8];
            LatWP v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q < nx ? i0 + q : nx - 1;
                k[q] = keys_g[i * ny + lane];
                k[q] = k[q] > 0x3FFFFu ? Q64_VAC : k[q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = lp.wp[k[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (i0 + q < nx) {
                    const int so = (i0 + q) * S + lane;
                    if (!PK) ST_l[so] = k[q];
                    WR_l[so] = v[q];
                }
            }
        }
    }
    wave_fence();

    const int grp_a = lane >> 3, grp_c = lane & 7;     // lane = 8 * a + c
    // the sum of row `row` (all 8 lanes of group a receive it): leaves c, c + 8, ... c + 56 per lane as a balanced
    // tree, then the 8-lane tree (group8_sum).  A pure function of the row's W with one fixed association.
    auto row_tree = [&](int row) -> double {
        const LatWP* w = WR_l + row * S + grp_c;
        const double l0 = w[0].w, l1 = w[8].w, l2 = w[16].w, l3 = w[24].w, l4 = w[32].w, l5 = w[40].w, l6 = w[48].w, l7 = w[56].w;
        return group8_sum(((l0 + l1) + (l2 + l3)) + ((l4 + l5) + (l6 + l7)));
    };
    double rowsum = 0.0;                               // lane i: the sum of row i
    for (int row0 = 0; row0 < 64; row0 += 8) {
        const double s = row_tree(row0 + grp_a);
        const int src = ((lane - row0) & 7) << 5;      // byte address of lane 8 * (lane - row0)
        const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(s));
        const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(s));
        if ((lane >> 3) == (row0 >> 3)) rowsum = __hiloint2double(hi, lo);
    }
    ScanZero sz = {0, 0, 0, 0};
    double rowpre = wave_scan_incl_z(rowsum, sz);
    double rowex = dpp_f64<0x138, 0xF>(rowpre);        // exclusive prefix (wave_shr:1, lane 0 <- 0)
    double gamma = readlane_f64(rowpre, 63);

    int status = 0;
    long long n_exact = 0, hops = 0;
    long long* hist_l = reinterpret_cast<long long*>(smem + (PK ? LAT2_MISC : LAT_MISC));            // [6]
    double* time_l = reinterpret_cast<double*>(smem + (PK ? LAT2_MISC : LAT_MISC) + 48);              // launch clock
    double* bstat_l = reinterpret_cast<double*>(smem + (PK ? LAT2_MISC : LAT_MISC) + 56);             // last, sum, min, max, count
    if (lane < 6) hist_l[lane] = 0;
    if (lane == 0) time_l[0] = 0.0;
    if (lane < 5) bstat_l[lane] = lane == 0 ? p.barrier[r * 5 + 0] : (lane == 2 ? 1e300 : (lane == 3 ? -1e300 : 0.0));
    const long long hop0 = p.hops_done[r];

    // displacement tags: tracked in registers per 64-hop chunk as in hop_q64_kernel (one atom per lane), but the
    // sites are LDS offsets (row * 72 + column): tg_sv = where the atom is now | the site it entered the chunk from << 16
    int tg_sv = -1, tg_ow = 0;

    // ---- per-lane constants of the hop loop --------------------------------------
    // fresh-key atomics: lanes 0..18 = the 18 sites that see the START point (key char q = lane) and the start point
    // itself (lane 18, the occupancy bit); lanes 32..50 the same for the END point
    const int at_q = lane & 31;
    const bool at_on = at_q <= 18;
    const int at_di = at_on && at_q < 18 ? -kNbI[at_q < 18 ? at_q : 0] : 0;
    const int at_dj = at_on && at_q < 18 ? -kNbJ[at_q < 18 ? at_q : 0] : 0;
    const uint32_t at_bit = at_on ? (at_q < 18 ? 1u << (17 - at_q) : LAT_OCC_BIT) : 0u;
    const int at_end = lane >= 32 ? -1 : 0;            // all ones: this lane works on the end point
    uint32_t* const at_idle = reinterpret_cast<uint32_t*>(smem + (PK ? LAT2_IDLE : LAT_IDLE)) + lane;
    const uint32_t at_and = (lane < 32) ? ~at_bit : ~0u;
    const uint32_t at_or = (lane >= 32) ? at_bit : 0u;
    // PK: this lane's key bit lives in the u16 plane (bits 0-15: dword s >> 1, shift 16 (s & 1) + bit) or in the nibble
    // plane (bits 16-18: dword s >> 3, shift 4 (s & 7) + bit - 16): which one is a constant of the lane
    const int at_b = at_on ? (at_q < 18 ? 17 - at_q : 18) : 0;
    const bool at_hi = at_b >= 16;
    const uint32_t pk_dsh = at_hi ? 3u : 1u, pk_m = at_hi ? 7u : 1u, pk_k = at_hi ? 2u : 4u, pk_pb = at_hi ? (uint32_t)at_b - 16u : (uint32_t)at_b;
    uint32_t* const pk_base = at_hi ? FB_l : FR_l;
    const uint32_t pk_src = (at_on && lane < 32) ? ~0u : 0u, pk_dst = (at_on && lane >= 32) ? ~0u : 0u;
    const uint32_t rp_sh = lane < 6 ? 10u * (uint32_t)lane : 60u;       // lanes >= 6 read index 0 (rate 0)
    // the geometry of a hop in direction d (:503-504) and of its refresh window (:444-453; the three window modes as in
    // the other kernels: i_lo = i - 2 + ext * min(di, 0), n_i = ni0 + ext * |di|), packed in lane d: one lane read and
    // six scalar bit-field extracts per hop instead of ~25 scalar instructions
    int geo;
    {
        const int wm_ext = p.window_mode == 1 ? 0 : -1;    // all-ones mask: the window extends along the hop
        const int wm_ni0 = p.window_mode == 0 ? 4 : 5;
        const int wm_nj0 = p.window_mode == 0 ? 4 : (p.window_mode == 1 ? 6 : 5);
        const int d = lane < 6 ? lane : 0;
        const int di = ((2310 >> (2 * d)) & 3) - 1;          // {1, 0, -1, -1, 0, 1}
        const int dj = ((105 >> (2 * d)) & 3) - 1;           // {0, 1, 1, 0, -1, -1}
        const int min_di = di < 0 ? di : 0, min_dj = dj < 0 ? dj : 0;
        const int ilo = -2 + (wm_ext & min_di), n_i = wm_ni0 + (wm_ext & di & 1);
        const int jlo = -2 + (wm_ext & min_dj), n_j = wm_nj0 + (wm_ext & dj & 1);
        geo = (di & 3) | ((dj & 3) << 2) | ((ilo & 7) << 4) | ((jlo & 7) << 7) | (n_i << 10) | (n_j << 13);
    }
    const double mfac = p.mfac;
    const unsigned long long ok_mask = p.force_exact != 0 ? 0ull : ~0ull;
    const bool presim = p.presim != 0;
    const long long my_hops = p.n_hops_r ? p.n_hops_r[r] : p.n_hops;
    const unsigned char* __restrict__ wp_b = reinterpret_cast<const unsigned char*>(lp.wp);

#ifdef KMC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    bool need_exact = false;
    int done = 0;
    for (long long h0 = 0; h0 < my_hops && status == 0; h0 += done) {
        const int cnt = (int)((my_hops - h0) < KMC_WAVE ? (my_hops - h0) : KMC_WAVE);
        auto draw_pair = [&](double& rho, double& u) {
            rho = 0.5; u = 1.0;
            if (lane < cnt) {
                if (p.draws) {
                    const double2 dv = reinterpret_cast<const double2*>(p.draws)[(size_t)r * p.n_hops + h0 + lane];
                    rho = dv.x; u = dv.y;
                } else {
                    const unsigned long long hh = (unsigned long long)(hop0 + h0 + lane);
                    u32x4 c; c.x = (uint32_t)hh; c.y = (uint32_t)(hh >> 32); c.z = (uint32_t)r + p.rid_base; c.w = KMC_STREAM_HOP;
                    const u32x4 o = philox4x32_10(c, p.seed_lo, p.seed_hi);
                    rho = u53(o.x, o.y); u = u53_open0(o.z, o.w);
                }
            }
        };
        double rho_v, u_v;
        draw_pair(rho_v, u_v);
        int gam_lo = __double2loint(1.0), gam_hi = __double2hiint(1.0), pn_lo = gam_lo, pn_hi = gam_hi;
        int idx_v = -1;                    // bit 30 set: pn is already divided (exact path)

        int hl = 0;
        int M_lo = 0, M_hi = 0, m_hi_limit = 0;
        bool ex_have = false;
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the draws are in
        int run = cnt;
        if (need_exact) {
            // ---- exact path (rare): the reference's selection replayed literally for hop h0 -----
            need_exact = false;
            wave_fence();
            if (PK) {
                __threadfence();                // the key stores of the hops so far have reached the L2
                exact_select_lat_g(keys_g, p.table, n, readlane_f64(rho_v, 0), xsh);
            } else {
                exact_select_lat(smem, p.table, ny, n, readlane_f64(rho_v, 0), xsh);
            }
            wave_fence();
            ++n_exact;
            if (!(xsh->gamma > 0.0)) { status = 2; run = 0; }
            else if (xsh->idx >= 6L * (long)n) { status = 1; run = 0; }
            else {
                ex_have = true;
                if (xsh->idx == 0) run = 1;     // rho <= 0 selects entry 0 whatever its rate (B4): a chunk of its own
            }
        }
        KMC_STAMP(0);
        for (; hl < run; ++hl) {
            const double t = readlane_f64(rho_v * gamma, hl);
            if (__double2hiint(gamma) >= m_hi_limit) {
                asm volatile("");
                const double m2 = (2.0 * gamma) * mfac;
                M_lo = __builtin_amdgcn_readfirstlane(__double2loint(m2));
                M_hi = __builtin_amdgcn_readfirstlane(__double2hiint(m2));
                m_hi_limit = __double2hiint(gamma) + 0x00100000;
            }
            const double M = __hiloint2double(M_hi, M_lo);
            // ---- row ------------------------------------------------------
            const int i = ff1_63(ballot64(rowpre >= t));
            const double base_i = readlane_f64(rowex, i);
            // ---- site: the row's 64 W (and palette words, used after the pick), one scan -----------
            const LatWP wr_row = WR_l[i * S + lane];
            const double wv = wr_row.w;
            const unsigned long long rp_row = wr_row.rp;
            const double wsc = wave_scan_incl_z(wv, sz);
            const int js = ff1_63(ballot64(base_i + wsc >= t));       // no site reaches t: 63, and nothing certifies
            const double lo_site = readlane_f64(base_i + dpp_f64<0x138, 0xF>(wsc), js);
            KMC_STAMP(1);
            // ---- direction: the site's six palette entries on lanes 0..5 -------------
            const uint32_t rp_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rp_row, js);
            const uint32_t rp_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rp_row >> 32), js);
            const unsigned long long rpw = ((unsigned long long)rp_hi << 32) | rp_lo;
            const uint32_t pi = (uint32_t)(rpw >> rp_sh) & 1023u;
            const double rate = PAL_l[pi];
            double rs = rate + dpp_f64<0x111, 0xF>(rate);
            rs += dpp_f64<0x112, 0xF>(rs);
            rs += dpp_f64<0x114, 0xF>(rs);
            KMC_STAMP(7);
            const double ap = lo_site + rs;
            const double ape = lo_site + dpp_f64<0x111, 0xF>(rs);
            const unsigned long long md = ballot64(ap >= t) & 0x3Full;
            const int dl = ff1_63(md);
            const unsigned long long mok = ballot64(ap - t >= M) & ballot64(t - ape >= M) & ballot64(rate > 0.0);
            unsigned long long cert = md & (0ull - md) & mok & ok_mask;         // the lowest set bit of md, certified
            asm("" : "+s"(cert));
            int pnum_lo = __builtin_amdgcn_readlane(__double2loint(rate), dl);
            int pnum_hi = __builtin_amdgcn_readlane(__double2hiint(rate), dl);
            int j = js, d = dl & 7, i_sel = i;
            int exact_flag = 0;
            if (cert == 0ull && !ex_have) { need_exact = true; break; }      // the chunk ends here
            if (ex_have) {
                asm volatile("");
                ex_have = false;
                const long ex_idx = xsh->idx;
                const double g = xsh->gamma;
                const double pn = xsh->c_found / g - xsh->cprev_found / g;      // the reference's own expression (:497)
                d = (int)(ex_idx % 6);
                i_sel = (int)(ex_idx / 6) / ny;
                j = (int)(ex_idx / 6) - i_sel * ny;
                pnum_lo = __builtin_amdgcn_readfirstlane(__double2loint(pn));
                pnum_hi = __builtin_amdgcn_readfirstlane(__double2hiint(pn));
                exact_flag = 1 << 30;
            }
            KMC_STAMP(2);
            const int ii = i_sel;
            const int idx = (ii * ny + j) * 6 + d;
            // ---- update (:518-535): the hop's geometry, packed per direction in lane d of `geo` ---------
            const int gw = __builtin_amdgcn_readlane(geo, d);
            const int di = __builtin_amdgcn_sbfe(gw, 0, 2), dj = __builtin_amdgcn_sbfe(gw, 2, 2);
            const int i_lo = ii + __builtin_amdgcn_sbfe(gw, 4, 3), j_lo = j + __builtin_amdgcn_sbfe(gw, 7, 3);
            const int n_i = (int)__builtin_amdgcn_ubfe((uint32_t)gw, 10, 3), n_j = (int)__builtin_amdgcn_ubfe((uint32_t)gw, 13, 3);
            const int ie = ii + di, je = j + dj;                 // unwrapped
            const int iw = wrap_i(ie), jw = wrap_j(je);
            const int s_src = ii * S + j, s_dst = iw * S + jw;
            // fresh keys: clear the start point's bit in the sites that see it, set the end point's (AND / OR, not
            // flips: with rho == 0 the reference "hops" entry 0 whatever its rate, B4)
            {
                const int ti = wrap_i(ii + at_di + (di & at_end));
                const int tj = wrap_j(j + at_dj + (dj & at_end));
                // (idle lanes AND ~0 / OR 0 into a word of their own: thirteen lanes on ONE address would be serialised)
                if (!PK) {
                    uint32_t* fa = at_on ? FR_l + (ti * S + tj) : at_idle;
                    __hip_atomic_fetch_and(fa, at_and, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_or(fa, at_or, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                } else {
                    const uint32_t ts = (uint32_t)(ti * S + tj);
                    uint32_t* fa = at_on ? pk_base + (ts >> pk_dsh) : at_idle;
                    const uint32_t mk = 1u << (((ts & pk_m) << pk_k) + pk_pb);
                    __hip_atomic_fetch_and(fa, ~(mk & pk_src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_or(fa, mk & pk_dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            // ---- windowed refresh (:444-465): lane (a, c) = window row a and the window column = c (mod 8), so that
            // the lane's site is a leaf of ITS class in the row's tree (columns c, c + 8, ... c + 56).  That needs the
            // window's columns in distinct classes (CLS) -- else lane c takes window column c and the rows are
            // re-summed from LDS.
            const int i_lo0 = wrap_i(i_lo), j_lo0 = wrap_j(j_lo);
            constexpr bool classes = CLS;
            int si = i_lo0 + grp_a, sj;
            bool inw = grp_a < n_i;
            if (!ANY) {
                const int wcol = (grp_c - j_lo0) & 7;
                si &= 63; sj = (j_lo0 + wcol) & 63;
                inw &= wcol < n_j;
            } else if (classes) {
                // lane c owns class c: the window column of that class, if there is one -- in the part of the window
                // before the lattice's edge (j_lo0 + k, k = (c - j_lo0) mod 8) or in the part wrapped to columns 0 .. b - 1
                // (column c itself) -- else column c, an unchanged leaf of the class
                si = si >= nx ? si - nx : si;
                const int k1 = (grp_c - j_lo0) & 7, col1 = j_lo0 + k1;
                const bool v1 = (k1 < n_j) & (col1 < ny);
                const int b = j_lo0 + n_j - ny;                  // uniform; <= 0: the window does not wrap
                sj = v1 ? col1 : grp_c;
                inw &= v1 | (grp_c < b);
            } else {
                si = si >= nx ? si - nx : si;
                sj = j_lo0 + grp_c; sj = sj >= ny ? sj - ny : sj;
                inw &= grp_c < n_j;
            }
            const int srow = si * S;
            const int so = srow + sj;
            wave_fence();
            uint32_t knew = fr_word(so);
            uint32_t kglob;                      // what the key array in global memory holds for the site: 0xFFFFFFFF if vacant
            {
                // vacant site -> 0x3FFFF: the sign-extended occupancy bit selects (no compare, no VCC)
                const uint32_t om = (uint32_t)__builtin_amdgcn_sbfe((int)knew, 18u, 1u);
                kglob = (knew & 0x3FFFFu) | ~om;
                knew = kglob & 0x3FFFFu;
            }
            LatWP wn;
            asm("" : "=v"(wn.w), "=v"(wn.rp));
            if (inw) {
                if (!PK) ST_l[so] = knew;
                uint32_t ko = knew << 4;
                asm("" : "+v"(ko));              // one 32-bit value: an SGPR base + VGPR offset load
                wn = *reinterpret_cast<const LatWP*>(wp_b + ko);
                if (PK) keys_g[ANY ? si * ny + sj : so] = kglob;      // behind the gather: its wait leaves this store in flight
            }
            KMC_STAMP(3);
            // ---- while the gather is in flight: what nothing in this hop waits for, and the unchanged part of the
            // window rows' sums.  A row's sum is the balanced tree over its 64 W; the lane's own leaf (column oc, the
            // refreshed site or, for lanes without one, column c) meets its sibling leaf, then the sibling pair, then
            // the sibling quad of its class: all old values, read now.
            double sib1 = 0.0, sib2 = 0.0, sib4 = 0.0, leaf = 0.0;
            if (classes) {
                const int oc = sj;               // in class c whether refreshed or not
                const LatWP* wr = WR_l + srow;
                const int c2 = (oc ^ 16) & ~8, c4 = (oc ^ 32) & ~24;
                leaf = wr[oc].w;
                sib1 = wr[oc ^ 8].w;
                sib2 = wr[c2].w + wr[c2 + 8].w;
                sib4 = (wr[c4].w + wr[c4 + 8].w) + (wr[c4 + 16].w + wr[c4 + 24].w);
            }
            writelane5(gam_lo, gam_hi, pn_lo, pn_hi, idx_v, __double2loint(gamma), __double2hiint(gamma),
                       pnum_lo, pnum_hi, idx | exact_flag, hl);
            ++hops;
            if (!presim) {
                const unsigned long long m = ballot64((tg_sv & 0xFFFF) == s_src);
                const int wrap = ANY ? ((ie < 0 ? -1 : (ie >= nx ? 1 : 0)) << 8) + ((je < 0 ? -1 : (je >= ny ? 1 : 0)) << 16)
                                     : ((ie >> 6) << 8) + ((je >> 6) << 16);
                const bool seen = m != 0ull;
                const int lm = seen ? (int)__builtin_ctzll(m) : hl;
                const int osv = __builtin_amdgcn_readlane(tg_sv, lm), oow = __builtin_amdgcn_readlane(tg_ow, lm);
                const int nsv = (seen ? (osv & (int)0xFFFF0000) : (s_src << 16)) | s_dst;
                const int now_ = (seen ? oow : (64 << 8) + (64 << 16)) + wrap;
                writelane2(tg_sv, tg_ow, nsv, now_, lm);
            }
            KMC_STAMP(4);
            // ---- re-sum the window's rows, re-scan ------------------------------------------------
            double s;
            if (classes) {
                leaf = inw ? wn.w : leaf;
                s = group8_sum(((leaf + sib1) + sib2) + sib4);
                if (inw) WR_l[so] = wn;
            } else {
                if (inw) WR_l[so] = wn;
                wave_fence();
                s = row_tree(si);
            }
            {
                int rel;
                if (ANY) { rel = lane - i_lo0; rel = rel < 0 ? rel + nx : rel; }
                else rel = (lane - i_lo) & 63;
                const int src = (rel & 7) << 5;                  // byte address of lane 8 * rel
                const bool take = ANY ? (rel < n_i) & (lane < nx) : rel < n_i;
                const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(s));
                const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(s));
                rowsum = take ? __hiloint2double(hi, lo) : rowsum;
            }
            KMC_STAMP(5);
            wave_fence();
            rowpre = wave_scan_incl_z(rowsum, sz);
            rowex = dpp_f64<0x138, 0xF>(rowpre);
            gamma = readlane_f64(rowpre, 63);
            KMC_STAMP(6);
        }
        // ---- flush the chunk: clock, trace, barrier diagnostics ----------------------
        done = hl;
        {
            const int dlv = lane < done ? (idx_v & 0x3FFFFFFF) % 6 : -1;
            int mine = 0;
#pragma unroll
            for (int dd = 0; dd < 6; ++dd) {
                const int c = __popcll(ballot64(dlv == dd));
                mine = lane == dd ? c : mine;
            }
            if (lane < 6) hist_l[lane] += (long long)mine;
        }
        const unsigned long long tg_used = presim ? 0ull : ballot64((tg_sv & 0xFFFF) != 0xFFFF);
        if (tg_used) {
            // ---- move the chunk's tags: memory still holds the state before the chunk --------
            const int tg_site = tg_sv & 0xFFFF, tg_vac = (int)((unsigned)tg_sv >> 16);
            const bool used = (tg_used >> lane) & 1ull;
            int4 tv = make_int4(0, 0, 0, 0);
            if (used) tv = tags_g[ext(tg_vac)];
            // the site an atom left is vacant now unless a moved atom sits on it, i.e. unless it is occupied
            const bool home_empty = used && (fr_word(used ? tg_vac : 0) & LAT_OCC_BIT) == 0u;
            tv.z += ((tg_ow >> 8) & 255) - 64;
            tv.w += ((tg_ow >> 16) & 255) - 64;
            if (used) tags_g[ext(tg_site)] = tv;
            if (home_empty) tags_g[ext(tg_vac)] = make_int4(-1, 0, 0, 0);
            tg_sv = -1;
        }
        const double gam_v = __hiloint2double(gam_hi, gam_lo), pn_v = __hiloint2double(pn_hi, pn_lo);
        if (!presim) {
            const double dt = lane < done ? (-1.0 / gam_v) * log(u_v) : 0.0;      // :530
            const double tsum = wave_sum(dt);
            if (lane == 0) time_l[0] += tsum;
        }
        if (p.sel && lane < (status != 0 ? cnt : done)) p.sel[(size_t)r * p.n_hops + h0 + lane] = lane < done ? (idx_v & 0x3FFFFFFF) : -1;
        {
            const bool has = lane < done && (idx_v & 0x3FFFFFFF) != 0;           // :495
            const double pd_v = (idx_v & (1 << 30)) ? 1.0 : gam_v;
            const double bar = has ? -log((pn_v / pd_v) / 1e13) * 0.025 : 0.0;  // :496-498
            const unsigned long long m = __ballot(has);
            if (m) {
                const double last = readlane_f64(bar, 63 - __builtin_clzll(m));
                const double sum = wave_sum(bar);
                double mn = has ? bar : 1e300, mx = has ? bar : -1e300;
                for (int off = 32; off > 0; off >>= 1) {
                    mn = fmin(mn, __shfl_xor(mn, off, 64));
                    mx = fmax(mx, __shfl_xor(mx, off, 64));
                }
                if (lane < 5) {
                    const double b_stat = bstat_l[lane];
                    bstat_l[lane] = lane == 0 ? last : (lane == 1 ? b_stat + sum : (lane == 2 ? fmin(b_stat, mn) :
                                    (lane == 3 ? fmax(b_stat, mx) : b_stat + (double)__popcll(m))));
                }
            }
        }
        if (status != 0 && p.sel) {
            for (long long hh = h0 + KMC_WAVE + lane; hh < p.n_hops; hh += KMC_WAVE)
                p.sel[(size_t)r * p.n_hops + hh] = -1;
        }
    }
    // ---- store the replica --------------------------------------------------------------
    wave_fence();
    if (!ANY || lane < ny) {
        for (int i = 0; i < nx; ++i) {
            const uint32_t f = fr_word(i * S + lane);
            const bool o = (f & LAT_OCC_BIT) != 0u;
            occ_g[i * ny + lane] = (uint8_t)o;
            if (!PK) keys_g[i * ny + lane] = o ? ST_l[i * S + lane] : KMC_VACANT;      // (PK: stored as the sites were refreshed)
        }
    }
    if (lane < 6) p.hist[r * 6 + lane] += hist_l[lane];
    const double b_stat = lane < 5 ? bstat_l[lane] : 0.0;
    if (lane == 0) {
        const double time_acc = time_l[0];
        p.time_total[r] += time_acc;
        p.step_time[r] = time_acc;
        p.gamma[r] = gamma;
        p.hops_done[r] = hop0 + hops;
        p.n_exact[r] += n_exact;
        p.status[r] |= status;
#ifdef KMC_STAMPS
        if (p.stamps)
            for (int k = 0; k < 8; ++k) p.stamps[r * 8 + k] = (long long)st_acc[k];
#endif
    }
    if (lane == 0) p.barrier[r * 5 + 0] = b_stat;
    if (lane == 1 || lane == 4) p.barrier[r * 5 + lane] += b_stat;
    if (lane == 2) p.barrier[r * 5 + 2] = fmin(p.barrier[r * 5 + 2], b_stat);
    if (lane == 3) p.barrier[r * 5 + 3] = fmax(p.barrier[r * 5 + 3], b_stat);
}

}  // namespace kmc
