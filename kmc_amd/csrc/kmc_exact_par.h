// kmc_exact_par.h -- the reference's selection, hex_LiCoO2_kmc.py:486-492, replayed EXACTLY by a
// whole workgroup: 256 threads in the large-lattice kernels, where 6n is millions, and the single
// wave of hop_q64_kernel, where a literal one-lane replay of 24 576 entries would hold the launch
// up for milliseconds (template parameter NW = waves; KeyFn(s) returns site s's key).
//
//     gamma = np.sum(P);  c_k = fl(c_{k-1} + P[k]);  idx = first k with fl(c_k / gamma) >= rho
//
// np.sum: numpy's pairwise tree inside 8192-element buffers, one wave per buffer, buffers added
// left to right (np_sum_model's association).
//
// The sequentially rounded cumsum looks inherently serial, but between two "events" it is an
// integer prefix sum.  While c stays inside one binade [2^e, 2^(e+1)) it is a multiple A*U of the
// binade's ulp U = 2^(e-52), and for a non-negative term p
//     fl(A*U + p) = (A + RN(p / U)) * U      unless p / U is exactly half way between two integers
//                                            (the tie is broken by A's parity) or the sum leaves
//                                            the binade (the rounding unit changes).
// p / U is an exact scaling, so q = RN(p / U) is an exact int64 and c_k = (A + q_1 + ... + q_k) * U
// with integer, associative sums: a block scan.  The events -- first term that ties, first term
// that reaches 2^(e+1), c = 0 -- are found by the same scan; the one addition at an event is done
// as the real fp64 add the reference does, the binade is re-read, and the scan resumes behind it.
// A run over all 6n entries has ~ log2(gamma / p_min) binade crossings, each re-scans one tile.
// The stopping rule fl(c / gamma) >= rho is monotone in c, so it is the comparison c >= c*, with
// c* the smallest double that satisfies it (found once, by stepping ulps around rho * gamma).
#pragma once
#include "kmc_device.h"

namespace kmc {

#define KMC_XP_WAVES 4                       // of the common instance (the large-lattice kernel may use more)
#define KMC_XP_NOEV 0xFFFFFFFFFFFFFFFFull

template <int W>
struct ExactParSharedT {
    double chunk_sum[W];
    long long wave_tot[2][W];                // double-buffered: one barrier per event-free tile
    int wave_flag[2][W];
    unsigned long long ev;                   // earliest event of the pass: entry index << 1 | type
    double c;                                // exact cumsum in front of entry `pos`
    long long pos;
    double gamma, cstar, c_found, cprev_found;
    long long idx;                           // 6n = not found
    int done;
    PwFrame frames[KMC_PW_DEPTH];
};
typedef ExactParSharedT<KMC_XP_WAVES> ExactParShared;

#ifdef KMC_XP_TRACE
__device__ volatile long long* g_xp_trace = nullptr;      // scripts/probes/xp_probe.hip: progress in pinned host memory
#define XP_TRACE(slot, val) do { if (threadIdx.x == 0 && g_xp_trace) g_xp_trace[slot] = (long long)(val); } while (0)
#else
#define XP_TRACE(slot, val) do {} while (0)
#endif

__device__ __forceinline__ double ulp_step(double x, int dir)    // x > 0
{
    return __longlong_as_double(__double_as_longlong(x) + dir);
}

__device__ __forceinline__ uint32_t xp_row(uint32_t k) { return k > 0x3FFFFu ? 0x7FFFFu : k; }   // vacant -> zero row

// numpy's 128-element leaf (8 interleaved accumulators) over entries [lo, lo + 128) of the flattened
// rate matrix, lo % 6 == O.  Everything is unrolled with compile-time positions so that the 22 rows
// the leaf covers are loaded as independent 16-byte loads and stay in registers.
template <int O, class KeyFn>
__device__ __forceinline__ double xp_leaf128(KeyFn keys, const double* __restrict__ table, long lo)
{
    const long s0 = (lo - O) / 6;
    uint32_t k[22];
#pragma unroll
    for (int q = 0; q < 22; ++q) k[q] = xp_row(keys(s0 + q));          // 22 loads in flight
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool first[8] = {true, true, true, true, true, true, true, true};
#pragma unroll
    for (int b = 0; b < 22; b += 8) {                     // batches of 8 rows, 24 independent 16-byte loads
        double v[8][6];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (b + q < 22) {
                double2 x = make_double2(0, 0), y = x, z = x;
                if (k[b + q] != 0x7FFFFu) {                // a vacant site's row is all zeros: no request
                    const double2* t = reinterpret_cast<const double2*>(table + (size_t)k[b + q] * 6);
                    x = t[0]; y = t[1]; z = t[2];
                }
                v[q][0] = x.x; v[q][1] = x.y; v[q][2] = y.x; v[q][3] = y.y; v[q][4] = z.x; v[q][5] = z.y;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const int e = (b + q) * 6 + d - O;        // position inside the leaf
                if (b + q < 22 && e >= 0 && e < 128) {
                    if (first[e & 7]) { r[e & 7] = v[q][d]; first[e & 7] = false; }
                    else r[e & 7] += v[q][d];
                }
            }
        }
    }
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// The same leaf with run-time loops and 8 loads in flight: ~40 registers instead of ~170, for callers
// that must stay inside a small register budget (hop_q64_kernel: 128).
template <class EntryFn>
__device__ __forceinline__ double xp_leaf128_lean(EntryFn entry, long lo)
{
    double r0 = entry(lo), r1 = entry(lo + 1), r2 = entry(lo + 2), r3 = entry(lo + 3),
           r4 = entry(lo + 4), r5 = entry(lo + 5), r6 = entry(lo + 6), r7 = entry(lo + 7);
#pragma unroll 1
    for (long i = lo + 8; i < lo + 128; i += 8) {
        const double a0 = entry(i), a1 = entry(i + 1), a2 = entry(i + 2), a3 = entry(i + 3),
                     a4 = entry(i + 4), a5 = entry(i + 5), a6 = entry(i + 6), a7 = entry(i + 7);
        r0 += a0; r1 += a1; r2 += a2; r3 += a3; r4 += a4; r5 += a5; r6 += a6; r7 += a7;
    }
    return ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
}

// inclusive prefix sum of a 64-bit integer over the wave (DPP, same steps as wave_scan_incl)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_i64(long long x)
{
    int lo = (int)x, hi = (int)(x >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long wave_scan_incl_i64(long long x)
{
    x += dpp_i64<0x111, 0xF>(x);
    x += dpp_i64<0x112, 0xF>(x);
    x += dpp_i64<0x114, 0xF>(x);
    x += dpp_i64<0x118, 0xF>(x);
    x += dpp_i64<0x142, 0xA>(x);
    x += dpp_i64<0x143, 0xC>(x);
    return x;
}

// Result in sh->idx, sh->gamma, sh->c_found, sh->cprev_found (every thread may read them after the
// call; the call ends with a barrier).  keys: KMC_VACANT = vacant; table must have the all-zero row
// 0x7FFFF (kmc_create allocates 2^19 rows for that).
//
// gamma_hint > 0 (the large-lattice kernel): skip the np.sum pass when the answer does not depend on it.  The caller's
// own sum gamma_hint and G = np.sum(P) both lie within delta / 2 (relative) of the real sum -- delta is the caller's bound,
// the "6n / 8192 + 512" term of its certification margin -- so with t = rho * gamma_hint
//     c >= t (1 + delta)  implies  c / G >= rho                  hence fl(c / G) >= rho, and
//     c <  t (1 - delta)  implies  c / G < rho (1 - u)           hence fl(c / G) <  rho  (delta carries 8 u of slack).
// The sequential cumsum is replayed exactly as always and stopped at the first k with c_k >= T_hi = t (1 + delta); if
// c_(k-1) < T_lo = t (1 - delta), every earlier entry fails the reference's test whatever G is, and k is the reference's
// index: two thirds of the work (all of np.sum's pass, which cannot stop early) are saved.  Otherwise (an entry
// inside the 2 delta band: ~1e-7 of the calls at 1024 x 1024) the function starts over with np.sum.  In the short
// form sh->gamma is gamma_hint, not np.sum's bits: the caller's barrier diagnostic (S5) of such a hop is good to ~1e-9.
//
// STRIPED (the large-lattice kernel): a tile without an event needs only the TOTAL of its integer quanta, whatever thread
// holds which entry.  So the tile is first read striped -- consecutive lanes take consecutive 16-byte pieces (two entries) of
// the rate rows, three lanes to a site: the three pieces of a row arrive with one or two cache-line look-ups instead of
// three, and the replay is bound by exactly those (one CU takes ~0.5 scattered 16-byte requests per cycle) -- and summed;
// only a tile that holds an event (a binade crossing, a tie, the stop: ~60 of the 1000 tiles of a 1024 x 1024 replay) is
// read again in the owner layout below, where every thread holds consecutive entries and the event's place can be found.
//
// Where the rates come from is a policy (Rows): XpRowsTable gathers the 48-byte table row behind a site's 18-bit key (every
// kernel's state); XpRowsPalette reads the site's six 10-bit palette indices from a per-site array and looks the rates up in
// a palette in LDS (hop_big_kernel's SITEW instances: a replay then STREAMS 8 of every 16 contiguous bytes instead of
// gathering a scattered table row per site, which is what bounded it at 1024 x 1024).
template <class KeyFn> struct XpRowsTable {
    KeyFn keys;
    const double* __restrict__ table;
    typedef uint32_t key_t;
    static constexpr bool WHOLE = false;                  // striped reading: three lanes share a row's three 16-byte pieces
    static constexpr bool HAS_TABLE = true;
    __device__ __forceinline__ key_t none() const { return KMC_VACANT; }
    __device__ __forceinline__ key_t key(long s) const { return keys(s); }
    __device__ __forceinline__ bool live(key_t k) const { return k <= 0x3FFFFu; }
    __device__ __forceinline__ double2 piece(key_t k, int part) const
    {
        return *reinterpret_cast<const double2*>(table + (size_t)k * 6 + 2 * part);
    }
    __device__ __forceinline__ double entry(long e) const
    {
        const int s = (int)(e / 6);
        const int d = (int)(e - 6 * (long)s);
        return table[(size_t)xp_row(keys(s)) * 6 + d];
    }
};

struct XpSiteWord { double w; unsigned long long rp; };   // = LatWP (kmc_hop_lat.h): {W, six 10-bit palette indices}
struct XpRowsPalette {
    const XpSiteWord* __restrict__ sw;                    // per site
    const __attribute__((address_space(3))) double* pal;  // in LDS (typed so: a generic pointer makes every look-up a flat load); pal[0] = 0.0
    typedef unsigned long long key_t;
    static constexpr bool WHOLE = true;                   // a lane reads whole rows: consecutive lanes, consecutive sites
    static constexpr bool HAS_TABLE = false;
    __device__ __forceinline__ key_t none() const { return 0ull; }
    __device__ __forceinline__ key_t key(long s) const { return sw[s].rp; }
    __device__ __forceinline__ bool live(key_t k) const { return k != 0ull; }      // all six rates zero otherwise
    __device__ __forceinline__ double2 piece(key_t k, int part) const
    {
        return make_double2(pal[(uint32_t)(k >> (20 * part)) & 1023u], pal[(uint32_t)(k >> (20 * part + 10)) & 1023u]);
    }
    __device__ __forceinline__ double entry(long e) const
    {
        const long s = e / 6;
        const int d = (int)(e - 6 * s);
        return pal[(uint32_t)(sw[s].rp >> (10 * d)) & 1023u];
    }
};

#define XP_SYNC() do { if (NW == 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); else __syncthreads(); } while (0)
template <int NW, int KMC_XP_SITES, bool LEAN, bool STRIPED = false, class Rows, class Shared>      // KMC_XP_SITES = sites per thread and tile
__device__ __forceinline__ void exact_select_rows_impl(Rows rows, int n, double rho, Shared* sh,
                                                       double gamma_hint = 0.0, double delta = 0.0)
{
    typedef typename Rows::key_t key_t;
    constexpr int XP_THREADS = 64 * NW;
    constexpr long XP_TILE = (long)XP_THREADS * KMC_XP_SITES * 6;
    // NW == 1: the caller is ONE wave, possibly of a workgroup that holds other waves doing something else (hop_lat_kernel's
    // two-replica workgroups): the thread number is the lane and a hand-off needs the memory fence only, not the barrier
    const int tid = NW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef KMC_XP_TRACE
    const unsigned long long xp_t0 = __builtin_amdgcn_s_memtime();
#endif
    auto entry = [&](long e) -> double { return rows.entry(e); };
    // ---- np.sum: full 8192-element buffers, one wave each (lane = 128-element leaf)
    const long n6 = 6L * n;
    const long nfull = n6 / 8192;
    double total = 0.0;                                   // meaningful in thread 0
    bool fast = gamma_hint > 0.0 && rho > 0.0;            // uniform
restart:
    for (long c0 = 0; c0 < nfull && !fast; c0 += NW) {
        const long c = c0 + wv;
        if (c < nfull) {
            const long lo = c * 8192 + (long)lane * 128;
            const int o = (int)(lo % 6);                  // 0, 2 or 4: 8192 and 128 are both 2 (mod 6)
            double res;
            if constexpr (LEAN || !Rows::HAS_TABLE) res = xp_leaf128_lean(entry, lo);
            else res = o == 0 ? xp_leaf128<0>(rows.keys, rows.table, lo) : (o == 2 ? xp_leaf128<2>(rows.keys, rows.table, lo) : xp_leaf128<4>(rows.keys, rows.table, lo));
            // pairwise tree over the 64 leaves: lower half + upper half at every level
            for (int b = 1; b < 64; b <<= 1) {
                const double other = __shfl_xor(res, b, 64);
                res = (lane & b) ? other + res : res + other;
            }
            if (lane == 0) sh->chunk_sum[wv] = res;
        }
        XP_SYNC();
        if (tid == 0)
            for (int q = 0; q < NW && c0 + q < nfull; ++q) total = total + sh->chunk_sum[q];
        XP_SYNC();
    }
    double t_lo = 0.0;                                    // fast form: the lower edge of the band (thread 0)
    if (tid == 0) {
        const long tail = n6 - nfull * 8192;
        if (fast) total = gamma_hint;
        else if (tail > 0) total = total + np_pairwise(nfull * 8192, tail, entry, sh->frames);
        sh->gamma = total;
        sh->done = 0; sh->idx = n6; sh->c = 0.0; sh->pos = 0; sh->c_found = 0.0; sh->cprev_found = 0.0;
        sh->cstar = 0.0;
        if (fast) {
            const double t = rho * total;
            sh->cstar = ulp_step(t + t * delta, +1);
            t_lo = ulp_step(t - t * delta, -1);
            if (!(t_lo > 0.0)) sh->done = 3;               // rho * gamma underflows: the long form decides
        } else if (!(total > 0.0)) {
            sh->done = 1;                                  // gamma == 0 (or NaN): the caller stops the replica
        } else if (0.0 / total >= rho) {
            sh->done = 1; sh->idx = 0;                     // searchsorted(left) with rho <= 0
        } else {
            // smallest double x with fl(x / gamma) >= rho (the quotient is monotone in x)
            double x = rho * total;
            if (!(x > 0.0)) x = 4.9406564584124654e-324;
            while (x / total < rho) x = ulp_step(x, +1);
            for (;;) {
                const double xm = ulp_step(x, -1);
                if (xm > 0.0 && xm / total >= rho) x = xm; else break;
            }
            sh->cstar = x;
        }
    }
    XP_SYNC();
    XP_TRACE(0, 1); XP_TRACE(5, __builtin_amdgcn_s_memtime() - xp_t0);
    if (fast && sh->done == 3) { XP_SYNC(); fast = false; total = 0.0; goto restart; }
    if (sh->done) return;
    const double cstar = sh->cstar;
    const long long two53 = 1LL << 53, two54 = 1LL << 54;
    const double magic = 4503599627370496.0;              // 2^52

    key_t knext[KMC_XP_SITES];
#pragma unroll
    for (int q = 0; q < KMC_XP_SITES; ++q) {
        const long s = (long)tid * KMC_XP_SITES + q;
        knext[q] = (!STRIPED && s < n) ? rows.key(s) : rows.none();
    }
    // striped reading: piece tid + XP_THREADS * m of the tile = part (piece % 3) of site (piece / 3)
    constexpr int NP = KMC_XP_SITES * 3;                  // 16-byte pieces per thread and tile
    key_t kstr[STRIPED ? NP : 1];
    const int tid_part = tid % 3, tid_site = tid / 3;
    // (Rows::WHOLE: pieces 3 q .. 3 q + 2 are the three parts of the thread's q-th site, site number q * XP_THREADS + tid)
    auto stripe_part = [&](int m) -> int { return Rows::WHOLE ? m % 3 : (tid_part + m * (XP_THREADS % 3)) % 3; };
    auto stripe_site = [&](int m) -> long { return Rows::WHOLE ? (long)((m / 3) * XP_THREADS + tid) : (long)(m * XP_THREADS + tid) / 3; };
    auto stripe_keys = [&](long t0) {
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const long s = t0 / 6 + stripe_site(m);
            if (Rows::WHOLE && m % 3) kstr[m] = kstr[m - 1];
            else kstr[m] = s < n ? rows.key(s) : rows.none();
        }
    };
    (void)tid_site;
    if (STRIPED) stripe_keys(0);
    // the striped pieces of the tile at hand, read while the tile before it was summed (vvalid), and then kstr holds the keys
    // of the tile after it: two tiles of requests are in flight, a tile's sum waits for neither
    double2 v[STRIPED ? NP : 1];
    bool vvalid = false;
    auto stripe_rows = [&](double2* dst) {
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            dst[m] = make_double2(0.0, 0.0);               // vacant, or beyond the lattice: zeros, no request
            if (rows.live(kstr[m])) dst[m] = rows.piece(kstr[m], stripe_part(m));
        }
    };
    // The running state (c, pos) is uniform and lives in registers: every thread derives it from the
    // same shared totals.  An event-free tile costs one barrier (wave totals are double-buffered).
    double c = 0.0;
    long pos = 0;
    int parity = 0;
    bool done = false;
    const int NE = KMC_XP_SITES * 6;
    for (long tile0 = 0; tile0 < n6 && !done; tile0 += XP_TILE) {
        const long tile_end = (tile0 + XP_TILE) < n6 ? (tile0 + XP_TILE) : n6;
        const long e0 = tile0 + (long)tid * NE;
        if (STRIPED) {
            if (c > 0.0) {                                 // (pos == tile0 here: every tile is left at its end)
                if (!vvalid) { stripe_rows(v); stripe_keys(tile0 + XP_TILE); }
                double2 vn[NP];
                stripe_rows(vn);                           // the next tile's rows and the keys of the one after it travel
                stripe_keys(tile0 + 2 * XP_TILE);          // while this one is summed
                const int ex = ilogb(c);
                const long long A = (long long)ldexp(c, 52 - ex);
                const double scale = ldexp(1.0, 52 - ex);
                long long run = 0;
                bool flag = false;
#pragma unroll
                for (int m = 0; m < 2 * NP; ++m) {
                    const double pv = (m & 1) ? v[m >> 1].y : v[m >> 1].x;
                    const double y = pv * scale, z = y + magic, qd = z - magic;
                    flag = flag | (!(y < magic - 1.0)) | (fabs(qd - y) == 0.5);
                    run += __double_as_longlong(z) & ((1LL << 52) - 1);
                }
                run = (flag || run > two54) ? two54 : run;
                const long long incl = wave_scan_incl_i64(run);
                const bool wflag = __builtin_amdgcn_ballot_w64(flag) != 0ull;
                if (lane == 63) { sh->wave_tot[parity][wv] = incl; sh->wave_flag[parity][wv] = wflag ? 1 : 0; }
                XP_SYNC();
                long long tot = 0;
                int anyflag = 0;
#pragma unroll
                for (int q = 0; q < NW; ++q) { tot += sh->wave_tot[parity][q]; anyflag |= sh->wave_flag[parity][q]; }
                parity ^= 1;
                const double c_end = ldexp((double)(A + tot), ex - 52);      // exact if no event
#pragma unroll
                for (int m = 0; m < NP; ++m) v[m] = vn[m];
                vvalid = true;
                if (!(anyflag != 0 || A + tot >= two53 || c_end >= cstar)) { c = c_end; pos = tile_end; continue; }
            } else {
                stripe_keys(tile0 + XP_TILE);
                vvalid = false;
            }
            // an event in this tile (or the sum still at zero): the owner layout finds it
#pragma unroll
            for (int q = 0; q < KMC_XP_SITES; ++q) {
                const long s = e0 / 6 + q;
                knext[q] = s < n ? rows.key(s) : rows.none();
            }
        }
        double p[KMC_XP_SITES * 6];
#pragma unroll
        for (int q = 0; q < KMC_XP_SITES; ++q) {
            double2 a = make_double2(0, 0), b = a, cc = a;
            if (rows.live(knext[q])) {                         // vacant: zeros, no request
                a = rows.piece(knext[q], 0); b = rows.piece(knext[q], 1); cc = rows.piece(knext[q], 2);
            }
            p[q * 6 + 0] = a.x; p[q * 6 + 1] = a.y; p[q * 6 + 2] = b.x; p[q * 6 + 3] = b.y; p[q * 6 + 4] = cc.x; p[q * 6 + 5] = cc.y;
        }
#pragma unroll
        for (int q = 0; q < KMC_XP_SITES; ++q) {          // the next tile's keys travel while this one is summed
            const long s = (e0 + XP_TILE) / 6 + q;
            knext[q] = (!STRIPED && s < n) ? rows.key(s) : rows.none();
        }
        const long hi_l = tile_end - e0;
        const int hi_m = hi_l < 0 ? 0 : (hi_l > NE ? NE : (int)hi_l);      // my entries [lo_m, hi_m) are live
        while (pos < tile_end && !done) {                 // segments of the tile between events
            XP_TRACE(1, tile0); XP_TRACE(2, pos); XP_TRACE(3, __double_as_longlong(c)); XP_TRACE(4, g_xp_trace[4] + 1);
            const long lo_l = pos - e0;
            const int lo_m = lo_l < 0 ? 0 : (lo_l > NE ? NE : (int)lo_l);
            int ex = 0;
            long long A = 0;
            double scale = 0.0;
            if (c > 0.0) { ex = ilogb(c); A = (long long)ldexp(c, 52 - ex); scale = ldexp(1.0, 52 - ex); }
            // RN(pv / U) for pv / U < 2^52 - 1 through the 2^52 trick; anything larger, and exact ties,
            // are left to the reference's own addition (an event)
            auto quantum = [&](double pv, bool* special) -> long long {
                const double y = pv * scale;              // exact: scale is a power of two
                const double z = y + magic;
                const double qd = z - magic;
                *special = (!(y < magic - 1.0)) | (fabs(qd - y) == 0.5);
                return __double_as_longlong(z) & ((1LL << 52) - 1);
            };
            // pass 1 (branch free): this thread's integer total and whether it holds a special entry
            long long run = 0;
            bool flag = false;
            if (c > 0.0) {
#pragma unroll
                for (int m = 0; m < NE; ++m) {
                    const bool live = (m >= lo_m) & (m < hi_m);
                    bool sp;
                    const long long q = quantum(p[m], &sp);
                    flag = flag | (sp & live);
                    run += live ? q : 0LL;
                }
                run = (flag || run > two54) ? two54 : run;        // saturate: everything behind an event is void (and no
                                                                  // sum of a tile's quanta can overflow)
            } else {
#pragma unroll
                for (int m = 0; m < NE; ++m) flag = flag | ((m >= lo_m) & (m < hi_m) & (p[m] > 0.0));
            }
            const long long incl = wave_scan_incl_i64(run);
            const bool wflag = __builtin_amdgcn_ballot_w64(flag) != 0ull;
            if (lane == 63) { sh->wave_tot[parity][wv] = incl; sh->wave_flag[parity][wv] = wflag ? 1 : 0; }
            XP_SYNC();
            long long base = A + (incl - run), tot = 0;
            int anyflag = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const long long wt = sh->wave_tot[parity][q];
                if (q < wv) base += wt;
                tot += wt;
                anyflag |= sh->wave_flag[parity][q];
            }
            parity ^= 1;
            const double c_end = c > 0.0 ? ldexp((double)(A + tot), ex - 52) : 0.0;      // exact if no event
            const bool any = anyflag != 0 || (c > 0.0 && (A + tot >= two53 || c_end >= cstar));
            if (!any) { c = c_end; pos = tile_end; continue; }           // the common tile
            // ---- an event somewhere in the tile: find the first one ----------------------------
            if (tid == 0) sh->ev = KMC_XP_NOEV;
            XP_SYNC();
            unsigned long long myev = KMC_XP_NOEV;
            double ev_before = 0.0, ev_after = 0.0, ev_p = 0.0;
            const bool look = flag || (c > 0.0 && (base + run >= two53 || ldexp((double)(base + run), ex - 52) >= cstar));
            if (look) {
#pragma unroll
                for (int m = 0; m < NE; ++m) {
                    if (myev == KMC_XP_NOEV && m >= lo_m && m < hi_m) {
                        const long em = e0 + m;
                        if (c > 0.0) {
                            bool sp;
                            const long long q = quantum(p[m], &sp);
                            const long long after = base + q;
                            if (sp || after >= two53) {           // the reference's own add decides
                                myev = ((unsigned long long)em << 1) | 0ull;
                                ev_before = ldexp((double)base, ex - 52); ev_p = p[m];
                            } else {
                                const double ca = ldexp((double)after, ex - 52);
                                if (ca >= cstar) {
                                    myev = ((unsigned long long)em << 1) | 1ull;
                                    ev_before = ldexp((double)base, ex - 52); ev_after = ca;
                                }
                            }
                            base = after;
                        } else if (p[m] > 0.0) {                  // c == 0: fl(0 + p) = p
                            myev = ((unsigned long long)em << 1) | 0ull;
                            ev_before = 0.0; ev_p = p[m];
                        }
                    }
                }
            }
            if (myev != KMC_XP_NOEV) atomicMin(&sh->ev, myev);
            XP_SYNC();
            const unsigned long long ev = sh->ev;
            if (ev != KMC_XP_NOEV && ev == myev) {
                const long k = (long)(ev >> 1);
                if (ev & 1ull) {
                    sh->idx = k; sh->c_found = ev_after; sh->cprev_found = ev_before; sh->done = 1;
                } else {
                    const double cn = ev_before + ev_p;       // the reference's addition, as is
                    if (cn >= cstar) { sh->idx = k; sh->c_found = cn; sh->cprev_found = ev_before; sh->done = 1; }
                    else { sh->c = cn; sh->pos = k + 1; }
                }
            }
            XP_SYNC();
            if (ev == KMC_XP_NOEV) { c = c_end; pos = tile_end; }
            else { c = sh->c; pos = sh->pos; done = sh->done != 0; }
        }
    }
    XP_SYNC();
    if (fast) {
        // the short form stands if an entry reached T_hi and the one before it is below T_lo
        if (tid == 0) sh->done = (sh->idx < n6 && sh->cprev_found < t_lo) ? 1 : 3;
        XP_SYNC();
        if (sh->done == 3) { XP_SYNC(); fast = false; total = 0.0; goto restart; }
    }
    XP_TRACE(6, __builtin_amdgcn_s_memtime() - xp_t0);
}

// the form every kernel but the SITEW instances of hop_big_kernel uses: rows gathered from the dense table by key
template <int NW, int KMC_XP_SITES, bool LEAN, bool STRIPED = false, class KeyFn, class Shared>
__device__ __forceinline__ void exact_select_par_impl(KeyFn keys, const double* __restrict__ table,
                                                      int n, double rho, Shared* sh,
                                                      double gamma_hint = 0.0, double delta = 0.0)
{
    XpRowsTable<KeyFn> rows = {keys, table};
    exact_select_rows_impl<NW, KMC_XP_SITES, LEAN, STRIPED>(rows, n, rho, sh, gamma_hint, delta);
}

// the 256-thread instance of hop_global_kernel (striped tiles, short form on request); keys in global memory,
// KMC_VACANT = vacant
__device__ __noinline__ void exact_select_par(const uint32_t* __restrict__ keys, const double* __restrict__ table,
                                              int n, double rho, ExactParShared* sh, double gamma_hint, double delta)
{
    exact_select_par_impl<4, 4, false, true>([&](long s) -> uint32_t { return keys[s]; }, table, n, rho, sh, gamma_hint, delta);
}

}  // namespace kmc
