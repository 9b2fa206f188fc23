// kmc_exact_par.h -- the reference's selection, hex_LiCoO2_kmc.py:486-492, replayed EXACTLY by a
// whole 256-thread workgroup (the rare path of the large-lattice kernels, where 6n is millions).
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
#include "kmc_hop_lds.h"

namespace kmc {

#define KMC_XP_THREADS 256
#define KMC_XP_WAVES 4
#define KMC_XP_SITES 4                       // sites per thread and tile
#define KMC_XP_TILE (KMC_XP_THREADS * KMC_XP_SITES * 6)
#define KMC_XP_NOEV 0xFFFFFFFFFFFFFFFFull

struct ExactParShared {
    double chunk_sum[KMC_XP_WAVES];
    long long wave_tot[KMC_XP_WAVES];
    unsigned long long ev;                   // earliest event of the pass: entry index << 1 | type
    int any;                                 // some thread of the pass has to look closer
    double c;                                // exact cumsum in front of entry `pos`
    long long pos;
    double gamma, cstar, c_found, cprev_found;
    long long idx;                           // 6n = not found
    int done;
    PwFrame frames[KMC_PW_DEPTH];
};

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
template <int O>
__device__ __forceinline__ double xp_leaf128(const uint32_t* __restrict__ keys, const double* __restrict__ table, long lo)
{
    const long s0 = (lo - O) / 6;
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool first[8] = {true, true, true, true, true, true, true, true};
#pragma unroll
    for (int b = 0; b < 22; b += 4) {                     // batches of 4 sites
        double v[4][6];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (b + q < 22) {
                const double2* t = reinterpret_cast<const double2*>(table + (size_t)xp_row(keys[s0 + b + q]) * 6);
                const double2 x = t[0], y = t[1], z = t[2];
                v[q][0] = x.x; v[q][1] = x.y; v[q][2] = y.x; v[q][3] = y.y; v[q][4] = z.x; v[q][5] = z.y;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
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

// Result in sh->idx, sh->gamma, sh->c_found, sh->cprev_found (every thread may read them after the
// call; the call ends with a barrier).  keys: KMC_VACANT = vacant; table must have the all-zero row
// 0x7FFFF (kmc_create allocates 2^19 rows for that).
__device__ __noinline__ void exact_select_par(const uint32_t* __restrict__ keys, const double* __restrict__ table,
                                              int n, double rho, ExactParShared* sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto entry = [&](long e) -> double {
        const int s = (int)(e / 6);
        const int d = (int)(e - 6 * (long)s);
        return table[(size_t)xp_row(keys[s]) * 6 + d];
    };
    // ---- np.sum: full 8192-element buffers, one wave each (lane = 128-element leaf)
    const long n6 = 6L * n;
    const long nfull = n6 / 8192;
    double total = 0.0;                                   // meaningful in thread 0
    for (long c0 = 0; c0 < nfull; c0 += KMC_XP_WAVES) {
        const long c = c0 + wv;
        if (c < nfull) {
            const long lo = c * 8192 + (long)lane * 128;
            const int o = (int)(lo % 6);                  // 0, 2 or 4: 8192 and 128 are both 2 (mod 6)
            double res = o == 0 ? xp_leaf128<0>(keys, table, lo) : (o == 2 ? xp_leaf128<2>(keys, table, lo) : xp_leaf128<4>(keys, table, lo));
            // pairwise tree over the 64 leaves: lower half + upper half at every level
            for (int b = 1; b < 64; b <<= 1) {
                const double other = __shfl_xor(res, b, 64);
                res = (lane & b) ? other + res : res + other;
            }
            if (lane == 0) sh->chunk_sum[wv] = res;
        }
        __syncthreads();
        if (tid == 0)
            for (int q = 0; q < KMC_XP_WAVES && c0 + q < nfull; ++q) total = total + sh->chunk_sum[q];
        __syncthreads();
    }
    if (tid == 0) {
        const long tail = n6 - nfull * 8192;
        if (tail > 0) total = total + np_pairwise(nfull * 8192, tail, entry, sh->frames);
        sh->gamma = total;
        sh->done = 0; sh->idx = n6; sh->c = 0.0; sh->pos = 0; sh->c_found = 0.0; sh->cprev_found = 0.0;
        sh->cstar = 0.0;
        if (!(total > 0.0)) {
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
    __syncthreads();
    XP_TRACE(0, 1);
    if (sh->done) return;
    const double cstar = sh->cstar;
    const long long two53 = 1LL << 53, two54 = 1LL << 54;
    const double magic = 4503599627370496.0;              // 2^52

    uint32_t knext[KMC_XP_SITES];
#pragma unroll
    for (int q = 0; q < KMC_XP_SITES; ++q) {
        const long s = (long)tid * KMC_XP_SITES + q;
        knext[q] = s < n ? keys[s] : KMC_VACANT;
    }
    for (long tile0 = 0; tile0 < n6; tile0 += KMC_XP_TILE) {
        const long tile_end = (tile0 + KMC_XP_TILE) < n6 ? (tile0 + KMC_XP_TILE) : n6;
        const long e0 = tile0 + (long)tid * (KMC_XP_SITES * 6);
        double p[KMC_XP_SITES * 6];
#pragma unroll
        for (int q = 0; q < KMC_XP_SITES; ++q) {
            const double2* t = reinterpret_cast<const double2*>(table + (size_t)xp_row(knext[q]) * 6);
            const double2 a = t[0], b = t[1], c = t[2];
            p[q * 6 + 0] = a.x; p[q * 6 + 1] = a.y; p[q * 6 + 2] = b.x; p[q * 6 + 3] = b.y; p[q * 6 + 4] = c.x; p[q * 6 + 5] = c.y;
        }
#pragma unroll
        for (int q = 0; q < KMC_XP_SITES; ++q) {          // the next tile's keys travel while this one is summed
            const long s = (e0 + KMC_XP_TILE) / 6 + q;
            knext[q] = s < n ? keys[s] : KMC_VACANT;
        }
        for (;;) {                                        // segments of the tile between events
            if (tid == 0) { sh->ev = KMC_XP_NOEV; sh->any = 0; }
            __syncthreads();
            const double c = sh->c;
            const long pos = sh->pos;
            XP_TRACE(1, tile0); XP_TRACE(2, pos); XP_TRACE(3, __double_as_longlong(c)); XP_TRACE(4, g_xp_trace[4] + 1);
            if (sh->done || pos >= tile_end) break;
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
                *special = !(y < magic - 1.0) || fabs(qd - y) == 0.5;
                return __double_as_longlong(z) & ((1LL << 52) - 1);
            };
            // pass 1: this thread's integer total and whether it needs a closer look
            long long run = 0;
            bool flag = false;
            if (c > 0.0) {
#pragma unroll
                for (int m = 0; m < KMC_XP_SITES * 6; ++m) {
                    const long em = e0 + m;
                    if (em >= pos && em < tile_end) {
                        bool sp;
                        const long long q = quantum(p[m], &sp);
                        flag = flag || sp;
                        run += q;
                    }
                }
                if (flag) run = two54;                    // saturate: everything behind an event is void
            } else {
#pragma unroll
                for (int m = 0; m < KMC_XP_SITES * 6; ++m) {
                    const long em = e0 + m;
                    flag = flag || (em >= pos && em < tile_end && p[m] > 0.0);
                }
            }
            long long incl = run;
            for (int off = 1; off < 64; off <<= 1) {
                const long long o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (lane == 63) sh->wave_tot[wv] = incl;
            __syncthreads();
            long long base = A + (incl - run), tot = 0;
#pragma unroll
            for (int q = 0; q < KMC_XP_WAVES; ++q) {
                if (q < wv) base += sh->wave_tot[q];
                tot += sh->wave_tot[q];
            }
            if (c > 0.0) {
                const long long endv = base + run;
                flag = flag || endv >= two53 || ldexp((double)endv, ex - 52) >= cstar;
            }
            if (flag) sh->any = 1;
            __syncthreads();
            const bool any = sh->any != 0;                // false on the common tile: no event, the sum just moves on
            // pass 2: flagged threads look for their first event
            unsigned long long myev = KMC_XP_NOEV;
            double ev_before = 0.0, ev_after = 0.0, ev_p = 0.0;
            if (any && flag) {
#pragma unroll
                for (int m = 0; m < KMC_XP_SITES * 6; ++m) {
                    const long em = e0 + m;
                    if (myev == KMC_XP_NOEV && em >= pos && em < tile_end) {
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
            __syncthreads();
            const unsigned long long ev = sh->ev;
            if (ev == KMC_XP_NOEV) {
                if (tid == 0) {
                    if (c > 0.0) sh->c = ldexp((double)(A + tot), ex - 52);
                    sh->pos = tile_end;
                }
            } else if (ev == myev) {
                const long k = (long)(ev >> 1);
                if (ev & 1ull) {
                    sh->idx = k; sh->c_found = ev_after; sh->cprev_found = ev_before; sh->done = 1;
                } else {
                    const double cn = ev_before + ev_p;       // the reference's addition, as is
                    if (cn >= cstar) { sh->idx = k; sh->c_found = cn; sh->cprev_found = ev_before; sh->done = 1; }
                    else { sh->c = cn; sh->pos = k + 1; }
                }
            }
            __syncthreads();
        }
        if (sh->done) break;
        __syncthreads();
    }
    __syncthreads();
}

}  // namespace kmc
