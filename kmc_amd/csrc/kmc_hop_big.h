// kmc_hop_big.h -- the hop loop for lattices larger than 64 x 64 whose occupancy bitmaps still
// fit one CU's LDS (up to ~1.1 M sites: BASELINE config 4, one 1024 x 1024 lattice).
//
// Same algorithm as kmc_hop_lds64.h -- hierarchical certified selection, window refresh from
// bitmaps, drift-free re-sum of what the window touched -- with a deeper hierarchy, because a
// row is now up to 4096 sites and there are up to 4096 rows.  All sums are trees of 8
// (group8_sum / row_sum8 association), so every level is a pure function of the level below:
//
//   W[key] per site (gathered, never stored)
//   L1[row][b]   8 sites        L2[row][b]  8 L1        L3[row][8]  8 L2       (global, L2-cache resident)
//   rowsum[row]  8 L3           R1[g]       8 rows                              (LDS)
//   R2[l]        8 R1, one per lane of wave 0, in a register, scanned every hop
//
// Workgroup = 256 threads = one lattice.  Wave 0 runs the dependent chain of a hop alone (no
// barrier inside it): descend R2 -> 64 rows -> the row's <= 64 L2 nodes -> 8 L1 -> 48 entries (the R1
// and L3 levels exist for the climb only: a wave-wide scan skips them on the way down), certify, flip two
// bits, recompute the <= 36 window keys from LDS bitmaps, re-sum the touched blocks from W and
// climb the tree in registers (children prefetched before the update, DPP trees, ds_bpermute
// between the 8-lane groups).  The other waves sleep-poll a sequence number in LDS and join only for
// the exact path (kmc_exact_par.h), which at 1024 x 1024 is taken by ~0.5 % of the hops: the
// common hop contains no barrier and no forced wait on its own stores.
//
// LDS: column bitmaps with a 2-row / 2-column periodic halo, (ny + 4) x (ceil((nx + 4) / 32) + 1)
// words (139 808 B at 1024 x 1024); the 5 x 5 neighbourhood of a site is five 2-word reads and
// five v_alignbit, the periodic wrap being in the halo.
#pragma once
#include "kmc_device.h"
#include "kmc_hop_lds.h"
#include "kmc_hop_lds64.h"
#include "kmc_exact_par.h"
#include "kmc_hop_lat.h"                     // LatWP, LAT_PAL_MAX: the palette form of the rate table

namespace kmc {

// waves per workgroup: wave 0 runs the hop loop, all of them the prologue, the write-back and the exact replay
// (round 3: 8 -- two waves per SIMD bring a replay at 1024 x 1024 from 0.97 to 0.72 ms; 16, at half the registers each, 2.7 ms)
#ifndef KMC_BIG_WAVES
#define KMC_BIG_WAVES 8
#endif
#ifndef KMC_BIG_XP_SITES
#define KMC_BIG_XP_SITES 4                   // sites per thread and tile of the replay
#endif
#define KMC_BIG_THREADS (64 * KMC_BIG_WAVES)
#define KMC_BIG_VAC 0x7FFFFu                 // table row of a vacant site (all zeros), see K64_VAC

typedef ExactParSharedT<KMC_BIG_WAVES> BigXpShared;

struct BigGeom {
    int nx, ny, n;
    int nb1, S1, S2;                         // L1 blocks per row; row strides of L1 (= 8 * nb2) and L2 (= 8 * nb3)
    int nr1, nr2;                            // R1 groups (ceil(nx / 8)), R2 groups (ceil(nr1 / 8) <= 64)
    int cw;                                  // words per bitmap column
    long long offL1, offL2, offL3, offRow, offR1, hier_doubles;   // offsets (in doubles) inside a replica's slab
    int lds_bytes;
};

// host + device: layout of a lattice; returns false if this kernel cannot take it
__host__ __device__ inline bool big_geometry(int nx, int ny, BigGeom* g)
{
    g->nx = nx; g->ny = ny; g->n = nx * ny;
    g->nb1 = (ny + 7) >> 3;
    const int nb2 = (g->nb1 + 7) >> 3, nb3 = (nb2 + 7) >> 3;
    g->S1 = nb2 * 8; g->S2 = nb3 * 8;
    g->nr1 = (nx + 7) >> 3; g->nr2 = (g->nr1 + 7) >> 3;
    g->cw = ((nx + 4 + 31) >> 5) + 1;
    long long o = 0;
    g->offL1 = o; o += (long long)nx * g->S1;
    g->offL2 = o; o += (long long)nx * g->S2;
    g->offL3 = o; o += (long long)nx * 8;
    g->offRow = o; o += (long long)g->nr1 * 8;
    g->offR1 = o; o += (long long)g->nr2 * 8;
    g->hier_doubles = o;
    const long long lds = ((((long long)(ny + 4) * g->cw * 4) + 15) & ~15LL) + 640 + (((long long)sizeof(BigXpShared) + 15) & ~15LL) +
                          ((long long)g->nr1 * 8 + (long long)g->nr2 * 8) * 8 + 64;
    g->lds_bytes = (int)lds;
    return nx >= 5 && ny >= 5 && nb3 <= 8 && g->nr2 <= 64 && lds <= 160 * 1024;
}

struct HopBigParams {
    BigGeom g;
    int window_mode;
    long long n_hops;
    const long long* n_hops_r;
    int presim, force_exact;
    double mfac;                 // unused here (the LDS kernels' flat margin factor)
    double mu;                   // 2^-53 * 1.001 * margin_scale: unit of the position-dependent margin
    const double* table;
    const double* wsum;
    const double* tcompact;
    uint8_t* occ;
    uint32_t* keys;
    double* hier;                // [R][g.hier_doubles], zero-initialised once (the pads stay zero)
    int4* tags;
    const double* draws;
    int32_t* sel;
    uint32_t seed_lo, seed_hi;
    double* time_total; double* step_time; double* gamma; long long* hist; long long* hops_done;
    long long* n_exact; int* status; double* barrier;
    long long* stamps;           // [R][8] cycles per segment (-DKMC_STAMPS builds) or null
    uint32_t rid_base;           // global id of replica 0 (see HopParams)
    // SITEW instances (round 3): the palette form of the table (hop_lat_kernel's: WP[key] = {W, six 10-bit palette indices},
    // the palette itself in LDS) and the per-site copy SW[site] = WP[key of the site], which the hop loop keeps current
    const LatWP* wp;
    const double* pal;           // [LAT_PAL_MAX]
    LatWP* sw;                   // [R][n]
};

struct BigCtrl {                 // wave 0 -> the helper waves
    int seq, stop;               // seq is bumped for every request; stop = leave
    int n_li;                    // Li atoms of the lattice (conserved)
    double rho, gam;             // the draw and the tree's gamma of the hop to replay
    uint32_t kscr[6 * 8];        // new keys of the window, [window row][window column]
    double wscr[6 * 8];          // SITEW: their W instead
};
#define KMC_BIG_CTRL_BYTES 640
// SITEW: the palette behind everything else
__host__ __device__ inline int big_sitew_lds_bytes(const BigGeom& g) { return g.lds_bytes + LAT_PAL_MAX * 8; }
__host__ __device__ inline bool big_sitew_ok(const BigGeom& g) { return g.S1 <= 128 && big_sitew_lds_bytes(g) <= 160 * 1024; }

// Parameters used at chunk boundaries and at the ends of the kernel only are re-read from the kernel-argument
// segment there (scalar loads through the constant cache) instead of occupying scalar registers for the whole hop
// loop, which at ~100 live scalars spills them through vector lanes (163 spill slots, ~160 v_readlane / v_writelane
// per hop before this); see q64_cold_params (kmc_hop_q64.h).
typedef const __attribute__((address_space(4))) HopBigParams* big_params_ptr;
__device__ __forceinline__ big_params_ptr big_cold_params()
{
    big_params_ptr kp = (big_params_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}
#define BP (big_cold_params())

// The exact path of this kernel (all four waves).  gamma_hint > 0: the short form of kmc_exact_par.h, which skips the
// np.sum pass (two thirds of a replay at 1024 x 1024) unless an entry falls inside the band that np.sum decides.
__device__ __noinline__ void exact_select_big(const uint32_t* __restrict__ keys, const double* __restrict__ table,
                                              int n, double rho, BigXpShared* sh, double gamma_hint, double delta)
{
#ifdef KMC_BIG_XP_OWNER         // scripts/variants_big.py: A/B against the owner layout only
    exact_select_par_impl<KMC_BIG_WAVES, KMC_BIG_XP_SITES, false, false>([&](long s) -> uint32_t { return keys[s]; }, table, n, rho, sh, gamma_hint, delta);
#else
    exact_select_par_impl<KMC_BIG_WAVES, KMC_BIG_XP_SITES, false, true>([&](long s) -> uint32_t { return keys[s]; }, table, n, rho, sh, gamma_hint, delta);
#endif
}

// The same replay from the per-site words and the palette in LDS (SITEW instances)
__device__ __noinline__ void exact_select_big_sw(const LatWP* __restrict__ sw, const double* pal, int n, double rho, BigXpShared* sh,
                                                 double gamma_hint, double delta)
{
    static_assert(sizeof(XpSiteWord) == sizeof(LatWP), "same words");
    XpRowsPalette rows = {reinterpret_cast<const XpSiteWord*>(sw), (const __attribute__((address_space(3))) double*)pal};
    exact_select_rows_impl<KMC_BIG_WAVES, KMC_BIG_XP_SITES, false, true>(rows, n, rho, sh, gamma_hint, delta);
}

// NSLOTS = L1 blocks a window row can touch: 2 when ny is a multiple of 8 (the third slot's code is then gone, with the
// uniform branches round it -- a taken branch costs this kernel's lone wave ~27 cycles), else 3
//
// SITEW (round 3; rows of up to 1024 sites, tables of up to 1024 distinct rates): the selection makes TWO dependent trips to
// global memory instead of four.  (1) The row's <= 128 L1 sums are read at once, two per lane, and one wave scan picks the
// block: the L2 level is kept for the climb only.  (2) Every site carries SW[site] = {W, the six rates of its (stale) key as
// 10-bit indices into a palette of the table's distinct values}, the palette (8 KB) lives in LDS: the entry pick reads the
// block's eight words and looks the 48 rates up in LDS, where it used to read the keys and then gather table rows.  The window
// refresh gathers WP[new key] (16 B, the trip the W gather made anyway) and stores it to SW; the sites of a touched block
// outside the window bring their W from SW, not from a second gather behind their keys.
template <bool COMPACT, int NSLOTS, bool SITEW = false>
__global__ __launch_bounds__(KMC_BIG_THREADS) void hop_big_kernel(HopBigParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const BigGeom g = p.g;
    const int nx = g.nx, ny = g.ny, n = g.n, cw = g.cw;
    uint32_t* bm = reinterpret_cast<uint32_t*>(smem);                      // [(ny + 4)][cw]
    const size_t bm_bytes = (((size_t)(ny + 4) * cw * 4) + 15) & ~(size_t)15;
    BigCtrl* ctl = reinterpret_cast<BigCtrl*>(smem + bm_bytes);
    BigXpShared* xp = reinterpret_cast<BigXpShared*>(smem + bm_bytes + KMC_BIG_CTRL_BYTES);
    static_assert(sizeof(BigCtrl) <= KMC_BIG_CTRL_BYTES, "ctrl block");
    double* rowsum = reinterpret_cast<double*>(smem + bm_bytes + KMC_BIG_CTRL_BYTES + ((sizeof(BigXpShared) + 15) & ~(size_t)15));   // [nr1 * 8]
    double* R1 = rowsum + (size_t)g.nr1 * 8;                                                                             // [nr2 * 8]
    double* PAL = R1 + (size_t)g.nr2 * 8 + 8;           // SITEW: [LAT_PAL_MAX], behind the 64 bytes of flush-only accumulators

    // (readfirstlane: the wave index is wave-uniform, which the compiler cannot see in tid >> 6 -- with it the hop loop's
    // control flow is scalar branches and its uniform values stay scalar)
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* occ_g = p.occ + (size_t)r * n;
    uint32_t* keys_g = p.keys + (size_t)r * n;
    int4* tags_g = p.tags + (size_t)r * n;
    double* hier = p.hier + (size_t)r * g.hier_doubles;
    double* L1 = hier + g.offL1;
    double* L2 = hier + g.offL2;
    double* L3 = hier + g.offL3;
    const double* __restrict__ wsum = p.wsum;
    const unsigned char* __restrict__ table_b = reinterpret_cast<const unsigned char*>(p.table);
    const unsigned char* __restrict__ tc_b = reinterpret_cast<const unsigned char*>(p.tcompact);
    auto wkey = [](uint32_t k) -> uint32_t { return k > 0x3FFFFu ? KMC_BIG_VAC : k; };   // KMC_VACANT -> zero row
    // Hop-loop accesses: ONE 32-bit element index off a base pointer (scalar base + 32-bit vector offset, the cheapest global
    // address form) instead of 64-bit per-lane pointer arithmetic; a replica's arrays are far below 4 GB.  hidx = index into
    // the replica's slab of sums (L1, L2, L3 at g.offL1 / offL2 / offL3).
    const unsigned char* hier_b = reinterpret_cast<const unsigned char*>(hier);
    const unsigned char* keys_b = reinterpret_cast<const unsigned char*>(keys_g);
    const uint32_t oL1 = (uint32_t)g.offL1, oL2 = (uint32_t)g.offL2, oL3 = (uint32_t)g.offL3;
    auto ld_h = [&](uint32_t hidx) -> double { asm("" : "+v"(hidx)); return *reinterpret_cast<const double*>(hier_b + (hidx << 3)); };
    const unsigned char* wsum_b = reinterpret_cast<const unsigned char*>(wsum);
    auto ld_w = [&](uint32_t k) -> double { asm("" : "+v"(k)); return *reinterpret_cast<const double*>(wsum_b + (k << 3)); };
    auto ld_k = [&](uint32_t s) -> uint32_t { asm("" : "+v"(s)); return *reinterpret_cast<const uint32_t*>(keys_b + (s << 2)); };
    LatWP* sw_g = SITEW ? p.sw + (size_t)r * n : nullptr;
    const unsigned char* sw_b = reinterpret_cast<const unsigned char*>(sw_g);
    const unsigned char* wp_b = reinterpret_cast<const unsigned char*>(p.wp);
    auto ld_sw_w = [&](uint32_t s) -> double { asm("" : "+v"(s)); return *reinterpret_cast<const double*>(sw_b + (s << 4)); };
    auto ld_sw_rp = [&](uint32_t s) -> uint2 { asm("" : "+v"(s)); return *reinterpret_cast<const uint2*>(sw_b + (s << 4) + 8); };
    // WP[key], zeros for a vacant site (the palette table has the 2^18 rows of real keys only)
    auto ld_wp = [&](uint32_t k) -> LatWP {
        uint32_t kc = k > 0x3FFFFu ? 0u : k;
        asm("" : "+v"(kc));
        LatWP v = *reinterpret_cast<const LatWP*>(wp_b + (kc << 4));
        if (k > 0x3FFFFu) { v.w = 0.0; v.rp = 0ull; }
        return v;
    };

    // bit (row rr, column cc) with rr in [-2, nx + 1], cc in [-2, ny + 1]
    auto bm_word = [&](int rr, int cc) -> int { return (cc + 2) * cw + ((rr + 2) >> 5); };
    auto bm_flip = [&](int row, int col, bool set) {
        // the site and its periodic images inside the halo
        for (int a = 0; a < 2; ++a) {
            const int rr = a == 0 ? row : (row < 2 ? row + nx : (row >= nx - 2 ? row - nx : -99));
            if (rr == -99) continue;
            for (int b = 0; b < 2; ++b) {
                const int cc = b == 0 ? col : (col < 2 ? col + ny : (col >= ny - 2 ? col - ny : -99));
                if (cc == -99) continue;
                const uint32_t bit = 1u << ((rr + 2) & 31);
                if (set) atomicOr(&bm[bm_word(rr, cc)], bit); else atomicAnd(&bm[bm_word(rr, cc)], ~bit);
            }
        }
    };

    // ---- load: bitmaps from the byte occupancy -------------------------------------------
    for (int w = tid; w < (ny + 4) * cw; w += KMC_BIG_THREADS) bm[w] = 0u;
    if (tid == 0) { ctl->seq = 0; ctl->stop = 0; ctl->n_li = 0; }
    __syncthreads();
    int my_li = 0;
    for (int s = tid; s < n; s += KMC_BIG_THREADS)
        if (occ_g[s] & 1u) { bm_flip(s / ny, s % ny, true); ++my_li; }
    atomicAdd(&ctl->n_li, my_li);
    // ---- the tree, level by level (each parent = row_sum8 of its 8 children) --------------
    for (long b = tid; b < (long)nx * g.nb1; b += KMC_BIG_THREADS) {
        const int row = (int)(b / g.nb1), blk = (int)(b % g.nb1);
        double w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = blk * 8 + e;
            if (SITEW) {
                LatWP v; v.w = 0.0; v.rp = 0ull;
                if (j < ny) { v = ld_wp(keys_g[(size_t)row * ny + j]); sw_g[(size_t)row * ny + j] = v; }
                w[e] = v.w;
            } else {
                w[e] = j < ny ? wsum[wkey(keys_g[(size_t)row * ny + j])] : 0.0;
            }
        }
        L1[(size_t)row * g.S1 + blk] = row_sum8(w);
    }
    if (SITEW)
        for (int k = tid; k < LAT_PAL_MAX; k += KMC_BIG_THREADS) PAL[k] = p.pal[k];
    __syncthreads();
    for (long q = tid; q < (long)nx * (g.S1 >> 3); q += KMC_BIG_THREADS) {
        const int row = (int)(q / (g.S1 >> 3)), b = (int)(q % (g.S1 >> 3));
        L2[(size_t)row * g.S2 + b] = row_sum8(L1 + (size_t)row * g.S1 + b * 8);
    }
    __syncthreads();
    for (long q = tid; q < (long)nx * (g.S2 >> 3); q += KMC_BIG_THREADS) {
        const int row = (int)(q / (g.S2 >> 3)), b = (int)(q % (g.S2 >> 3));
        L3[(size_t)row * 8 + b] = row_sum8(L2 + (size_t)row * g.S2 + b * 8);
    }
    __syncthreads();
    for (int row = tid; row < g.nr1 * 8; row += KMC_BIG_THREADS) rowsum[row] = row < nx ? row_sum8(L3 + (size_t)row * 8) : 0.0;
    __syncthreads();
    for (int q = tid; q < g.nr2 * 8; q += KMC_BIG_THREADS) R1[q] = q < g.nr1 ? row_sum8(rowsum + q * 8) : 0.0;
    __syncthreads();

    const double six_nli = 6.0 * (double)ctl->n_li;
    // ---- wave 0 state ------------------------------------------------------------------------
    double r2 = (wv == 0 && lane < g.nr2) ? row_sum8(R1 + lane * 8) : 0.0;
    double r2pre = wave_scan_incl(r2);
    double r2ex = dpp_f64<0x138, 0xF>(r2pre);
    double gamma = readlane_f64(r2pre, 63);

    int status = 0;
    int n_exact = 0;
    int hist_c = 0;
    long long hist_v = 0;
    // flush-only accumulators (clock, barrier statistics: last, sum, min, max, count) live in the 64 spare bytes behind R1,
    // not in scalar registers that the hop loop would spill (see kmc_hop_q64.h)
    double* acc_l = R1 + (size_t)g.nr2 * 8;
    if (tid < 6) acc_l[tid] = tid == 1 ? p.barrier[r * 5 + 0] : (tid == 3 ? 1e300 : (tid == 4 ? -1e300 : 0.0));
    const long long hop0 = p.hops_done[r];
    const long long my_hops = p.n_hops_r ? p.n_hops_r[r] : p.n_hops;

    const int grp_a = lane >> 3, grp_c = lane & 7;
    const int ent_site = lane < 48 ? (lane * 43) >> 8 : 0;
    const int ent_dir = lane < 48 ? lane - 6 * ent_site : 0;
    const uint32_t ent_off = COMPACT ? (uint32_t)ent_dir * (KMC_COMPACT_ROWS * 8u) : (uint32_t)ent_dir * 8u;
    // SITEW: direction d's palette index is bits 10 d .. 10 d + 9 of the 60-bit word: from the low dword (d = 3 straddles: a
    // funnel shift) or from the high one
    const bool ent_hi = ent_dir >= 4;
    const uint32_t ent_sh = ent_hi ? 10u * (uint32_t)ent_dir - 32u : 10u * (uint32_t)ent_dir;
    uint32_t cf_st[5], cf_ln[5], cf_sh[5];
    {
        uint32_t m = compact_mask(ent_dir), o = 0;
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const uint32_t st = m ? (uint32_t)__builtin_ctz(m) : 0u;
            const uint32_t inv = ~(m >> st);
            const uint32_t ln = m ? (inv ? (uint32_t)__builtin_ctz(inv) : 32u) : 0u;
            cf_st[f] = st; cf_ln[f] = ln; cf_sh[f] = o;
            o += ln;
            m = ln >= 32u ? 0u : (m & ~(((1u << ln) - 1u) << st));
        }
    }
    const int wm_ext = p.window_mode == 1 ? 0 : 1;
    const int wm_ni0 = p.window_mode == 0 ? 4 : 5;
    const int wm_nj0 = p.window_mode == 0 ? 4 : (p.window_mode == 1 ? 6 : 5);
    constexpr int nslots = NSLOTS;                      // L1 blocks a window row can touch: (ny & 7) ? 3 : 2
    const double mu = p.mu;
    const double xtra = 6.0 * (double)n / 8192.0 + 512.0;
    const bool force_exact = p.force_exact != 0;
    // kmc_set_debug(force_exact = 1): every hop through the exact path in its long form (np.sum always); 2: every hop
    // through it, short form allowed -- what a hop that cannot be certified gets
    // (read where they are used, from the kernel-argument segment: the hop loop has no scalar registers to spare)
#define BIG_XP_SHORT (BP->force_exact != 1)
#define BIG_XP_DELTA (BP->mu * (6.0 * (double)BP->g.n / 8192.0 + 512.0) + 8.0 * 1.1102230246251565e-16)
    const bool presim = p.presim != 0;

    // one level of the descent: 8 children at ch[0..7], running base; returns the child picked
    auto descend = [&](uint32_t ch, double t, double& base, bool& ok) -> int {     // ch = slab index of child 0
        const double sc = scan8_incl(ld_h(ch + (uint32_t)grp_c));
        const double bp = base + sc;
        const double bpe = base + dpp_f64<0x111, 0xF>(sc);
        const unsigned long long mb = ballot64(bp >= t) & 0xFFull;
        ok = ok && mb != 0ull;
        const int q = (__ffsll((long long)mb) - 1) & 7;
        base = readlane_f64(bpe, q);
        return q;
    };

    if (wv != 0) {
        // ---- helper waves: wait for exact-path requests -------------------------------------------
        volatile int* vseq = &ctl->seq;
        volatile int* vstop = &ctl->stop;
        int seen = 0;
        for (;;) {
            int sq;
            while ((sq = *vseq) == seen) __builtin_amdgcn_s_sleep(16);
            seen = sq;
            if (*vstop) break;
            if (SITEW)
                exact_select_big_sw(sw_g, PAL, n, *(volatile double*)&ctl->rho, xp,
                                    BIG_XP_SHORT ? *(volatile double*)&ctl->gam : 0.0, BIG_XP_DELTA);
            else
                exact_select_big(keys_g, BP->table, n, *(volatile double*)&ctl->rho, xp,
                                 BIG_XP_SHORT ? *(volatile double*)&ctl->gam : 0.0, BIG_XP_DELTA);
        }
    }
    double rho_v = 0.5, u_v = 1.0, gam_v = 1.0, pn_v = 1.0;
    int idx_v = -1;
    int seq_sent = 0;
#ifdef KMC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    // The exact path is a function call; inside the hop loop it makes the compiler spill scalar registers
    // throughout the loop.  A hop that is not certified leaves the loop, the call is made in the outer
    // loop, and the hop loop is re-entered at the same hop with the answer waiting in xp (ex_have).
    bool ex_have = false;
    long long h = 0;
    for (;;) {
    for (; wv == 0 && h < my_hops && status == 0; ++h) {
        const int hl = (int)(h & 63);
        int i = 0, j = 0, d = 0;
        double pnum = 1.0;
        bool ok = false;
        if (hl == 0) {
            // ---- uniforms for the next 64 hops, one hop per lane ---------------------------
            const int cnt = (int)((my_hops - h) < 64 ? (my_hops - h) : 64);
            rho_v = 0.5; u_v = 1.0; gam_v = 1.0; pn_v = 1.0; idx_v = -1;
            const big_params_ptr bp = BP;               // taken here, in wave-uniform control flow
            if (lane < cnt) {
                if (bp->draws) {
                    const double2 dv = reinterpret_cast<const double2*>(bp->draws)[(size_t)r * bp->n_hops + h + lane];
                    rho_v = dv.x; u_v = dv.y;
                } else {
                    const unsigned long long hh = (unsigned long long)(hop0 + h + lane);
                    u32x4 c; c.x = (uint32_t)hh; c.y = (uint32_t)(hh >> 32); c.z = (uint32_t)r + bp->rid_base; c.w = KMC_STREAM_HOP;
                    const u32x4 o = philox4x32_10(c, bp->seed_lo, bp->seed_hi);
                    rho_v = u53(o.x, o.y); u_v = u53_open0(o.z, o.w);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the draws are in; nothing else in the hop waits on them
        }
        const double rho = readlane_f64(rho_v, hl);
        KMC_STAMP(0);
        {
            // ---- descent ----------------------------------------------------------------------
            const double t = rho * gamma;
            ok = gamma > 0.0 && !force_exact;
            const unsigned long long m2 = ballot64(r2pre >= t);
            const int g2 = (__ffsll((long long)m2) - 1) & 63;
            double base = readlane_f64(r2ex, g2);
            // 64 rows of group g2 at once (LDS), then the row's L2 nodes at once (<= 64, one global read)
            {
                const int ri = g2 * 64 + lane;
                // (loads that sat under a lane condition are read at a clamped, valid address and selected afterwards: a
                // guarded load costs an exec save / restore and a skip branch the compiler keeps round memory operations)
                const int rmax = g.nr1 * 8 - 1;
                const double rs_v = rowsum[ri < rmax ? ri : rmax];
                const double sc = wave_scan_incl(ri <= rmax ? rs_v : 0.0);
                const double bp = base + sc;
                const double bpe = base + dpp_f64<0x138, 0xF>(sc);
                const unsigned long long mb = ballot64(bp >= t);
                ok = ok && mb != 0ull;
                const int q = (__ffsll((long long)mb) - 1) & 63;
                base = readlane_f64(bpe, q);
                i = g2 * 64 + q;
                if (i >= nx) { ok = false; i = 0; }
            }
            int b1;
            if (SITEW) {
                // the row's L1 sums, two per lane (S1 <= 128, its pads are zeros): one trip, one scan, and the pair's own pick
                const int half = g.S1 >> 1;
                uint32_t ho = (oL1 + (uint32_t)i * (uint32_t)g.S1 + 2u * (uint32_t)(lane < half ? lane : half - 1)) << 3;
                asm("" : "+v"(ho));
                double2 lp = *reinterpret_cast<const double2*>(hier_b + ho);
                if (lane >= half) lp = make_double2(0.0, 0.0);
                const double sc = wave_scan_incl(lp.x + lp.y);
                const double bp = base + sc;
                const double bpe = base + dpp_f64<0x138, 0xF>(sc);
                const double a0 = bpe + lp.x;
                const bool first = a0 >= t;
                const double base_l = first ? bpe : a0;
                const unsigned long long mb = ballot64(bp >= t);
                ok = ok && mb != 0ull;
                const int q = (__ffsll((long long)mb) - 1) & 63;
                b1 = 2 * q + (int)(((ballot64(first) >> q) & 1ull) ^ 1ull);
                base = readlane_f64(base_l, q);
            } else {
                int b2;
                {
                    const double l2_v = ld_h(oL2 + (uint32_t)i * (uint32_t)g.S2 + (uint32_t)(lane < g.S2 ? lane : g.S2 - 1));
                    const double sc = wave_scan_incl(lane < g.S2 ? l2_v : 0.0);
                    const double bp = base + sc;
                    const double bpe = base + dpp_f64<0x138, 0xF>(sc);
                    const unsigned long long mb = ballot64(bp >= t);
                    ok = ok && mb != 0ull;
                    b2 = (__ffsll((long long)mb) - 1) & 63;
                    base = readlane_f64(bpe, b2);
                    if (b2 >= (g.S1 >> 3)) { ok = false; b2 = 0; }
                }
                b1 = b2 * 8 + descend(oL1 + (uint32_t)i * (uint32_t)g.S1 + (uint32_t)b2 * 8u, t, base, ok);
            }
            if (b1 >= g.nb1) { ok = false; b1 = 0; }
            KMC_STAMP(1);
            // ---- entry: 8 sites x 6 directions on 48 lanes ---------------------------------
            const int ej = b1 * 8 + ent_site;
            double rate;
            if (SITEW) {
                const uint2 rp = ld_sw_rp((uint32_t)i * (uint32_t)ny + (uint32_t)(ej < ny ? ej : ny - 1));
                const uint32_t pi = __builtin_amdgcn_alignbit(ent_hi ? 0u : rp.y, ent_hi ? rp.y : rp.x, ent_sh) & 1023u;
                rate = PAL[(ej < ny && lane < 48) ? pi : 0u];             // palette[0] = 0.0
            } else {
                const uint32_t ek_v = wkey(ld_k((uint32_t)i * (uint32_t)ny + (uint32_t)(ej < ny ? ej : ny - 1)));
                const uint32_t ek = ej < ny ? ek_v : KMC_BIG_VAC;
                if (COMPACT) {
                    uint32_t ci = __builtin_amdgcn_ubfe(ek, cf_st[0], cf_ln[0]);
#pragma unroll
                    for (int f = 1; f < 5; ++f) ci |= __builtin_amdgcn_ubfe(ek, cf_st[f], cf_ln[f]) << cf_sh[f];
                    uint32_t ro = (ci << 3) + ent_off;
                    asm("" : "+v"(ro));
                    rate = *reinterpret_cast<const double*>(tc_b + ro);
                } else {
                    uint32_t ro = ek * 48u + ent_off;
                    asm("" : "+v"(ro));
                    rate = *reinterpret_cast<const double*>(table_b + ro);
                }
            }
            const double esc = wave_scan_incl(rate);
            const double ap = base + esc;
            const double ape = base + dpp_f64<0x138, 0xF>(esc);
            const unsigned long long me = ballot64(ap >= t) & 0xFFFFFFFFFFFFull;
            const int el = (__ffsll((long long)me) - 1) & 63;
            // Margin of THIS entry (kmc_exact_par.h has the flat version's derivation): the reference's
            // sequential sum has rounded at most m = min(6 (site + 1), 6 N_Li) times when it reaches the
            // entry (adding a vacant site's zeros is exact), each time by <= u relative to a partial sum
            // <= ap: m u ap.  Our tree sums (<= 70 roundings), np.sum against our gamma (buffers added left
            // to right: 6n / 8192 + 30), fl(rho gamma) and the 2u of the quotient are all relative to t <= ap.
            const double m_add = fmin(6.0 * (double)(i * ny + ej + 1), six_nli);
            const double M = ap * (mu * (m_add + xtra));
            const unsigned long long mok = ballot64((ap - t >= M) & (t - ape >= M) & (rate > 0.0));
            ok = ok && ((me & mok & (0ull - me)) != 0ull);
            pnum = readlane_f64(rate, el);
            const int es = (el * 43) >> 8;
            j = b1 * 8 + es;
            d = el - 6 * es;
        }
        KMC_STAMP(2);
        if (!ok && !ex_have) break;                          // to the exact path in the outer loop
        int exact_flag = 0;
        if (ex_have) {
            asm volatile("");
            ex_have = false;
            const long long idx = xp->idx;
            const double gx = xp->gamma;
            d = (int)(idx % 6);
            j = (int)((idx / 6) % ny);
            i = (int)((idx / 6) / ny);
            pnum = xp->c_found / gx - xp->cprev_found / gx;       // the reference's own expression (:497)
            exact_flag = 1 << 30;
        }
        // ---- update (:518-535) -----------------------------------------------------------------
        const int s_src = i * ny + j;
        const int idx = s_src * 6 + d;
        const int di = ((2310 >> (2 * d)) & 3) - 1;          // {1, 0, -1, -1, 0, 1}
        const int dj = ((105 >> (2 * d)) & 3) - 1;           // {0, 1, 1, 0, -1, -1}
        const int ie = i + di, je = j + dj;                  // unwrapped
        const int iw = wrap1(ie, nx), jw = wrap1(je, ny);
        const int s_dst = iw * ny + jw;
        // tag move (:521-528): the tag is requested here and stored behind the W gathers -- its 16 MB array is read from
        // HBM, and a store issued here would make the wave wait out that latency on the spot
        int4 tg = make_int4(0, 0, 0, 0);
        if (!presim) {                                       // every lane, one address: one request
            uint32_t to = (uint32_t)s_src << 4;
            asm("" : "+v"(to));
            tg = *reinterpret_cast<const int4*>(reinterpret_cast<const unsigned char*>(tags_g) + to);
        }
        // window geometry (:444-453)
        const int min_di = di < 0 ? di : 0, min_dj = dj < 0 ? dj : 0;
        const int i_lo = wrap1(i - 2 + wm_ext * min_di, nx), j_lo = wrap1(j - 2 + wm_ext * min_dj, ny);
        int n_i = wm_ni0 + wm_ext * (di != 0), n_j = wm_nj0 + wm_ext * (dj != 0);
        const bool act = grp_a < n_i;
        const int si = wrap1(i_lo + (act ? grp_a : 0), nx);                // this group's lattice row
        // the <= 3 L1 blocks the window's columns fall in, cyclically from the block of column j_lo
        int bs[3];
        bs[0] = j_lo >> 3;
        bs[1] = bs[0] + 1 >= g.nb1 ? 0 : bs[0] + 1;
        bs[2] = bs[1] + 1 >= g.nb1 ? 0 : bs[1] + 1;
        // ---- reads that do not depend on the update: old keys of the blocks' sites, and the
        // children of every tree node the window will change
        uint32_t kold[3];
        double wold[3];                                      // SITEW: the W of the blocks' sites as they stand
        double c1[3], c2[3];
        // this lane's row in each array (24-bit multiplies: rows and strides are far below 2^24)
        const uint32_t si_ny = __umul24((uint32_t)si, (uint32_t)ny);
        const uint32_t si_L1 = oL1 + __umul24((uint32_t)si, (uint32_t)g.S1), si_L2 = oL2 + __umul24((uint32_t)si, (uint32_t)g.S2),
                       si_L3 = oL3 + (uint32_t)si * 8u;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int cj = bs[q] * 8 + grp_c;
            wold[q] = 0.0;
            if (q < nslots && SITEW) {
                const double wv_ = ld_sw_w(si_ny + (uint32_t)(cj < ny ? cj : ny - 1));
                wold[q] = cj < ny ? wv_ : 0.0;
                kold[q] = KMC_VACANT;
            } else if (q < nslots) {
                const uint32_t kv = ld_k(si_ny + (uint32_t)(cj < ny ? cj : ny - 1));
                kold[q] = cj < ny ? kv : KMC_VACANT;
            } else kold[q] = KMC_VACANT;
            c1[q] = q < nslots ? ld_h(si_L1 + (uint32_t)((bs[q] >> 3) * 8 + grp_c)) : 0.0;      // siblings under its L2 parent
            c2[q] = q < nslots ? ld_h(si_L2 + (uint32_t)((bs[q] >> 6) * 8 + grp_c)) : 0.0;      // siblings under its L3 parent
        }
        double c3 = ld_h(si_L3 + (uint32_t)grp_c);
        // R1 / R2 nodes above the window rows: up to three distinct R1 parents, group k takes the k-th
        // (in order of first appearance along the window: the parent of row i_lo, that of the last row before the lattice's
        // edge or of the window's last row, and R1[0] when the window wraps -- fewer than 8 rows span at most two parents on
        // either side of the edge; a loop over the rows costs ~20 taken branches a hop)
        int rp[3], nrp = 1;
        {
            const int last_row = i_lo + n_i - 1;
            const bool wraps = last_row >= nx;
            const int pA = i_lo >> 3, pB = (wraps ? nx - 1 : last_row) >> 3;
            rp[0] = pA; rp[1] = pA; rp[2] = pA;
            if (pB != pA) { rp[1] = pB; nrp = 2; }
            if (wraps && pA != 0 && pB != 0) { if (nrp == 2) rp[2] = 0; else rp[1] = 0; ++nrp; }
        }
        const int my_rp = rp[grp_a < 3 ? grp_a : 0];
        const double crow = rowsum[my_rp * 8 + grp_c];                      // the 8 rows under R1[my_rp]
        const double cr1 = R1[(my_rp >> 3) * 8 + grp_c];                    // the 8 R1 under R2[my_rp >> 3]
        // ---- flip the two sites in the bitmaps ---------------------------------------------------
        {
            // lanes 0-3 clear the source bit and its periodic images in the halo, lanes 4-7 set the
            // destination's: lane & 1 = the row image, lane & 2 = the column image
            const bool isdst = (lane & 4) != 0;
            const int row = isdst ? iw : i, col = isdst ? jw : j;
            const int mr = row < 2 ? row + nx : (row >= nx - 2 ? row - nx : row);
            const int mc = col < 2 ? col + ny : (col >= ny - 2 ? col - ny : col);
            const int rr = (lane & 1) ? mr : row, cc = (lane & 2) ? mc : col;
            const bool valid = (lane < 8) & (!(lane & 1) | (mr != row)) & (!(lane & 2) | (mc != col));
            const uint32_t bit = 1u << ((rr + 2) & 31);
            uint32_t* wp = &bm[(cc + 2) * cw + ((rr + 2) >> 5)];
            if (valid & !isdst) atomicAnd(wp, ~bit);
            if (valid & isdst) atomicOr(wp, bit);
        }
        hist_c += (lane == d);
        if (lane == hl) { gam_v = gamma; pn_v = pnum; idx_v = idx | exact_flag; }
        wave_fence();
        KMC_STAMP(3);
        // ---- windowed refresh (:453-465): lane = 8 * (window row) + (window column) ----------
        if (act && grp_c < n_j) {
            const int sj = wrap1(j_lo + grp_c, ny);
            // rows si-2 .. si+2 are bits si .. si+4 of a column; columns sj-2 .. sj+2
            const uint32_t sh = (uint32_t)si & 31u;
            const uint32_t* cwp = bm + (size_t)sj * cw + (si >> 5);         // column sj-2 is halo column sj
            uint32_t win[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) win[q] = __builtin_amdgcn_alignbit(cwp[q * cw + 1], cwp[q * cw], sh);
            const uint32_t m2 = win[0], m1 = win[1], w0 = win[2], w1 = win[3], w2 = win[4];
            uint32_t knew = ((w2 & 7u) << 15) | ((w1 & 15u) << 11) | (((w0 >> 3) & 3u) << 9) | ((w0 & 3u) << 7) |
                            (((m1 >> 1) & 15u) << 3) | ((m2 >> 2) & 7u);
            knew = (w0 & 4u) ? knew : KMC_VACANT;
            {
                uint32_t ks = si_ny + (uint32_t)sj;
                asm("" : "+v"(ks));
                *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(keys_g) + (ks << 2)) = knew;
                if (SITEW) {
                    const LatWP v = ld_wp(knew);
                    *reinterpret_cast<LatWP*>(reinterpret_cast<unsigned char*>(sw_g) + (ks << 4)) = v;
                    ctl->wscr[grp_a * 8 + grp_c] = v.w;
                }
            }
            if (!SITEW) ctl->kscr[grp_a * 8 + grp_c] = knew;
        }
        wave_fence();
        KMC_STAMP(4);
        // ---- re-sum the touched L1 blocks from W, 8 lanes per block ------------------------------
        double s1[3] = {0.0, 0.0, 0.0}, wq[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 3; ++q) {                                         // all gathers in flight together
            if (q < nslots) {
                const int cj = bs[q] * 8 + grp_c;
                int wc = cj - j_lo;
                if (wc < 0) wc += ny;
                double wv_;
                if (SITEW) {
                    const double ww = ctl->wscr[(act ? grp_a : 0) * 8 + (wc & 7)];
                    wv_ = (cj < ny && wc < n_j) ? ww : wold[q];
                } else {
                    const uint32_t kw = ctl->kscr[(act ? grp_a : 0) * 8 + (wc & 7)];
                    const uint32_t k = (cj < ny && wc < n_j) ? kw : kold[q];
                    wv_ = ld_w(wkey(k));                     // a lane of an idle group reads row i_lo's keys: valid, unused
                }
                wq[q] = act ? wv_ : 0.0;
            }
        }
        s1[0] = group8_sum(wq[0]);
        s1[1] = group8_sum(wq[1]);
        if (nslots > 2) s1[2] = group8_sum(wq[2]);
        if (!presim && lane == 0) {
            tg.z += ie < 0 ? -1 : (ie >= nx ? 1 : 0);
            tg.w += je < 0 ? -1 : (je >= ny ? 1 : 0);
            tags_g[s_dst] = tg;
            tags_g[s_src] = make_int4(-1, 0, 0, 0);
        }
        KMC_STAMP(5);
        // ---- climb: L2, L3, row (each node = tree of its 8 children, the changed ones replaced) ---
        double s2[3] = {0.0, 0.0, 0.0}, s3[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < nslots) {
                double v = c1[q];
                const int first = (bs[q] >> 3) * 8 + grp_c;                  // L1 index of this lane's child
#pragma unroll
                for (int h2 = 0; h2 < 3; ++h2) if (h2 < nslots && first == bs[h2]) v = s1[h2];
                s2[q] = group8_sum(v);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < nslots) {
                double v = c2[q];
                const int first = (bs[q] >> 6) * 8 + grp_c;                  // L2 index of this lane's child
#pragma unroll
                for (int h2 = 0; h2 < 3; ++h2) if (h2 < nslots && first == (bs[h2] >> 3)) v = s2[h2];
                s3[q] = group8_sum(v);
            }
        }
        {
            // one store instruction for the three levels: lanes 0-2 of a group write L1, 3-5 L2, 6-7 L3
            // (issuing a vector store costs the wave ~120 cycles whatever its width)
            const int q = grp_c < 3 ? grp_c : (grp_c < 6 ? grp_c - 3 : grp_c - 6);
            const int bq = q == 0 ? bs[0] : (q == 1 ? bs[1] : bs[2]);
            const double v1 = q == 0 ? s1[0] : (q == 1 ? s1[1] : s1[2]);
            const double v2 = q == 0 ? s2[0] : (q == 1 ? s2[1] : s2[2]);
            const double v3 = q == 0 ? s3[0] : s3[1];
            uint32_t hs = grp_c < 3 ? si_L1 + (uint32_t)bq : (grp_c < 6 ? si_L2 + (uint32_t)(bq >> 3) : si_L3 + (uint32_t)(bq >> 6));
            asm("" : "+v"(hs));
            unsigned char* hier_w = reinterpret_cast<unsigned char*>(hier);
            if (act && q < nslots) *reinterpret_cast<double*>(hier_w + (hs << 3)) = grp_c < 3 ? v1 : (grp_c < 6 ? v2 : v3);
            if (nslots > 2 && act && grp_c == 0) {
                uint32_t h3 = si_L3 + (uint32_t)(bs[2] >> 6);
                asm("" : "+v"(h3));
                *reinterpret_cast<double*>(hier_w + (h3 << 3)) = s3[2];
            }
        }
        {
#pragma unroll
            for (int h2 = 0; h2 < 3; ++h2) if (h2 < nslots && grp_c == (bs[h2] >> 6)) c3 = s3[h2];
        }
        const double rnew = group8_sum(c3);                                  // new sum of row si (groups a < n_i)
        if (act && grp_c == 0) rowsum[si] = rnew;
        KMC_STAMP(6);
        // ---- R1: group k < nrp re-sums R1[rp[k]] from its 8 rows ---------------------------------
        double r1new;
        {
            const int rc = my_rp * 8 + grp_c;                                // lattice row of this lane's child
            int rel = rc - i_lo;
            if (rel < 0) rel += nx;
            const int src = (rel & 7) << 5;                                  // lane 8 * rel holds that row's new sum
            const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(rnew));
            const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(rnew));
            const double v = (rc < nx && rel < n_i) ? __hiloint2double(hi, lo) : crow;
            r1new = group8_sum(v);
            if (grp_a < nrp && grp_c == 0) R1[my_rp] = r1new;
        }
        // ---- R2: the lanes of wave 0 that own a changed R2 node ------------------------------------
        {
            double v = cr1;
            const int first = (my_rp >> 3) * 8 + grp_c;                      // R1 index of this lane's child
            // (all three groups, no loop over nrp: a group beyond nrp repeats group 0's parent and so its value)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double rk = readlane_f64(r1new, k * 8);
                if (first == rp[k]) v = rk;
            }
            const double r2new = group8_sum(v);                              // group k: new R2[rp[k] >> 3]
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double rk = readlane_f64(r2new, k * 8);
                if (lane == (rp[k] >> 3)) r2 = rk;
            }
        }
        wave_fence();
        r2pre = wave_scan_incl(r2);
        r2ex = dpp_f64<0x138, 0xF>(r2pre);
        gamma = readlane_f64(r2pre, 63);
        KMC_STAMP(7);

        // ---- flush the chunk: clock, trace, barrier diagnostics ------------------------------------
        if (hl == 63 || h + 1 == my_hops) {
            const int done = hl + 1;
            const long long h0 = h - hl;
            hist_v += hist_c; hist_c = 0;
            if (!presim) {
                const double dt = lane < done ? (-1.0 / gam_v) * log(u_v) : 0.0;      // :530
                const double ts = wave_sum(dt);
                if (lane == 0) acc_l[0] += ts;
            }
            const big_params_ptr bp = BP;
            if (bp->sel && lane < done) bp->sel[(size_t)r * bp->n_hops + h0 + lane] = idx_v & 0x3FFFFFFF;
            const bool has = lane < done && (idx_v & 0x3FFFFFFF) != 0;               // :495
            const double pd_v = (idx_v & (1 << 30)) ? 1.0 : gam_v;
            const double bar = has ? -log((pn_v / pd_v) / 1e13) * 0.025 : 0.0;      // :496-498
            const unsigned long long m = __ballot(has);
            if (m) {
                const double last = readlane_f64(bar, 63 - __builtin_clzll(m));
                const double sum = wave_sum(bar);
                double mn = has ? bar : 1e300, mx = has ? bar : -1e300;
                for (int off = 32; off > 0; off >>= 1) {
                    mn = fmin(mn, __shfl_xor(mn, off, 64));
                    mx = fmax(mx, __shfl_xor(mx, off, 64));
                }
                if (lane >= 1 && lane < 6) {
                    const double o = acc_l[lane];
                    acc_l[lane] = lane == 1 ? last : (lane == 2 ? o + sum : (lane == 3 ? fmin(o, mn) : (lane == 4 ? fmax(o, mx) : o + (double)__popcll(m))));
                }
            }
        }
    }
    if (!(wv == 0 && h < my_hops && status == 0)) break;
    {
        // ---- exact path for hop h: publish the keys this wave stored, wake the helpers, replay the
        // reference together
        const double rho = readlane_f64(rho_v, (int)(h & 63));
        __threadfence_block();
        if (lane == 0) { *(volatile double*)&ctl->rho = rho; *(volatile double*)&ctl->gam = gamma; *(volatile int*)&ctl->seq = ++seq_sent; }
        else ++seq_sent;
        if (SITEW) exact_select_big_sw(sw_g, PAL, n, rho, xp, BIG_XP_SHORT ? gamma : 0.0, BIG_XP_DELTA);
        else exact_select_big(keys_g, BP->table, n, rho, xp, BIG_XP_SHORT ? gamma : 0.0, BIG_XP_DELTA);
        ++n_exact;
        if (!(xp->gamma > 0.0)) { status = 2; break; }
        if (xp->idx >= 6LL * n) { status = 1; break; }
        ex_have = true;
    }
    }
    if (tid == 0) { *(volatile int*)&ctl->stop = 1; *(volatile int*)&ctl->seq = seq_sent + 1; }
    __syncthreads();
    // ---- store the lattice (every thread), scalars (wave 0) ------------------------------------------
    for (int s = tid; s < n; s += KMC_BIG_THREADS) {
        const int row = s / ny, col = s % ny;
        occ_g[s] = (uint8_t)((bm[bm_word(row, col)] >> ((row + 2) & 31)) & 1u);
    }
    if (wv == 0) {
        const big_params_ptr bp = BP;
        const long long hops = h;                           // hops done: h counts completed hops only
        if (status != 0 && hops < my_hops) {
            // a replica that stopped: records of the unfinished chunk, then the trace of what never ran
            const int done = (int)(hops & 63);
            const long long h0 = hops - done;
            hist_v += hist_c;
            if (done > 0) {
                if (!presim) {
                    const double dt = lane < done ? (-1.0 / gam_v) * log(u_v) : 0.0;
                    const double ts = wave_sum(dt);
                    if (lane == 0) acc_l[0] += ts;
                }
                if (bp->sel && lane < done) bp->sel[(size_t)r * bp->n_hops + h0 + lane] = idx_v & 0x3FFFFFFF;
                const bool has = lane < done && (idx_v & 0x3FFFFFFF) != 0;
                const double pd_v = (idx_v & (1 << 30)) ? 1.0 : gam_v;
                const double bar = has ? -log((pn_v / pd_v) / 1e13) * 0.025 : 0.0;
                const unsigned long long m = __ballot(has);
                if (m) {
                    const double last = readlane_f64(bar, 63 - __builtin_clzll(m));
                    const double sum = wave_sum(bar);
                    double mn = has ? bar : 1e300, mx = has ? bar : -1e300;
                    for (int off = 32; off > 0; off >>= 1) {
                        mn = fmin(mn, __shfl_xor(mn, off, 64));
                        mx = fmax(mx, __shfl_xor(mx, off, 64));
                    }
                    if (lane >= 1 && lane < 6) {
                        const double o = acc_l[lane];
                        acc_l[lane] = lane == 1 ? last : (lane == 2 ? o + sum : (lane == 3 ? fmin(o, mn) : (lane == 4 ? fmax(o, mx) : o + (double)__popcll(m))));
                    }
                }
            }
        }
        if (bp->sel)
            for (long long hh = hops + lane; hh < bp->n_hops; hh += 64) bp->sel[(size_t)r * bp->n_hops + hh] = -1;
        if (lane < 6) bp->hist[r * 6 + lane] += hist_v;
        wave_fence();
        const double time_acc = acc_l[0], b_last = acc_l[1], b_sum = acc_l[2], b_min = acc_l[3], b_max = acc_l[4], b_cnt = acc_l[5];
        if (lane == 0) {
            bp->time_total[r] += time_acc;
            bp->step_time[r] = time_acc;
            bp->gamma[r] = gamma;
            bp->hops_done[r] = hop0 + hops;
            bp->n_exact[r] += n_exact;
            bp->status[r] |= status;
            bp->barrier[r * 5 + 0] = b_last;
            bp->barrier[r * 5 + 1] += b_sum;
            bp->barrier[r * 5 + 2] = fmin(bp->barrier[r * 5 + 2], b_min);
            bp->barrier[r * 5 + 3] = fmax(bp->barrier[r * 5 + 3], b_max);
            bp->barrier[r * 5 + 4] += b_cnt;
#ifdef KMC_STAMPS
            if (bp->stamps)
                for (int k = 0; k < 8; ++k) bp->stamps[r * 8 + k] = (long long)st_acc[k];
#endif
        }
    }
}

}  // namespace kmc
