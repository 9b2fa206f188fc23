// kmc_hop_lds.h -- the fused, LDS-resident hop loop (rows S1-S5, U1-U4, R3, K).
//
// One wave64 = one replica.  A replica's whole selection state lives in LDS:
//   w[n]    f64  sum of the six rates of each site (at its last refresh)
//   key[n]  u32  18-bit key of each site at its last refresh (stale keys are
//                state: the reference's window leaves 6-7 affected sites
//                un-refreshed per hop)
//   col[ny] u64  occupancy, one bitmap per lattice column (bit i = row i)
// and in registers: lane i holds the sum of row i and its inclusive prefix.
//
// Per hop (hex_LiCoO2_kmc.py:484-535):
//   t = rho * gamma                                   (:486-490)
//   row    = first lane with rowprefix >= t           ballot on registers
//   site   = first lane with rowbase + scan(w[row]) >= t   one LDS row + DPP scan
//   dir    = first of the six table rates crossing t  scalar loads of T[key]
//   certify: the reference picks the first idx with fl(cumsum_seq[idx]/np.sum) >= rho;
//            our sums associate differently, so the pick is accepted only if t is
//            further than a rigorous rounding bound M from both neighbouring
//            boundaries; otherwise (about 3e-8 of hops at 64x64) the hop is
//            re-evaluated by exact_select(), a literal sequential restatement.
//   update bitmaps / tags / clock / histogram         (:518-535)
//   refresh the <= 36 window sites (keys from bitmaps, gather Wsum[key]), re-sum
//   the <= 6 touched rows, re-scan the row prefix     (:444-465)
#pragma once
#include "kmc_device.h"

namespace kmc {

struct HopParams {
    int nx, ny, n, window_mode;
    long long n_hops;            // hops per replica (stride of draws / sel)
    const long long* n_hops_r;   // [R] per-replica hop counts (<= n_hops) or null
    int presim, force_exact;
    double mfac;                 // certification margin = gamma * mfac
    const double* table;         // [2^18][6]
    const double* wsum;          // [2^18]
    uint8_t* occ;                // [R][n]
    uint32_t* keys;              // [R][n]
    int4* tags;                  // [R][n]  i0, j0, wi, wj
    const double* draws;         // [R][n_hops][2] or null
    int32_t* sel;                // [R][n_hops] or null
    uint32_t seed_lo, seed_hi;
    double* time_total;          // [R]
    double* step_time;           // [R]
    double* gamma;               // [R]
    long long* hist;             // [R][6]
    long long* hops_done;        // [R]
    long long* n_exact;          // [R]
    int* status;                 // [R]
    double* barrier;             // [R][5] last, sum, min, max, count
};

typedef const __attribute__((address_space(4))) double* cdouble_ptr;

// The workgroup is ONE wave: its LDS instructions execute in issue order, so a
// cross-lane LDS hand-off needs no s_barrier and no s_waitcnt, only a point the
// compiler may not move memory operations across.  (__syncthreads() would also
// drain vmcnt and stall on the deferred tag load.)
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

struct ExactPick { long idx; double gamma, c, cprev; };

// Literal restatement of hex_LiCoO2_kmc.py:486-492 on the stale-key state:
// gamma with np.sum's association, strictly sequential cumsum, first index whose
// normalised value reaches rho.  Every lane runs it redundantly (uniform).
__device__ __noinline__ void exact_select(const uint32_t* key, const double* table, int n, double rho,
                                          PwFrame* frames, ExactPick* out)
{
    auto entry = [&](long e) -> double {
        const int s = (int)(e / 6);
        const int d = (int)(e - 6 * (long)s);
        const uint32_t k = key[s];
        return k == KMC_VACANT ? 0.0 : table[(size_t)k * 6 + d];
    };
    const long n6 = 6L * n;
    const double gamma = np_sum_model(n6, entry, frames);
    double c = 0., cprev = 0.;
    long idx = n6;
    if (0.0 / gamma >= rho) {
        idx = 0;                                   // searchsorted(left) with rho <= 0
    } else {
        for (int s = 0; s < n && idx == n6; ++s) {
            const uint32_t k = key[s];
            if (k == KMC_VACANT) continue;         // + 0.0 changes nothing
            for (int d = 0; d < 6; ++d) {
                const double pv = table[(size_t)k * 6 + d];
                if (pv == 0.0) continue;
                cprev = c;
                c += pv;
                if (c / gamma >= rho) { idx = (long)s * 6 + d; break; }
            }
        }
    }
    out->idx = idx; out->gamma = gamma; out->c = c; out->cprev = cprev;
}

__global__ __launch_bounds__(64) void hop_lds_kernel(HopParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = blockIdx.x, lane = threadIdx.x;
    const int nx = p.nx, ny = p.ny, n = p.n;
    double* w = reinterpret_cast<double*>(smem);
    uint64_t* col = reinterpret_cast<uint64_t*>(smem + (size_t)8 * n);
    uint32_t* key = reinterpret_cast<uint32_t*>(smem + (size_t)8 * n + 512);
    PwFrame* frames = reinterpret_cast<PwFrame*>(smem + (size_t)8 * n + 512 + (((size_t)4 * n + 15) & ~(size_t)15));
    ExactPick* pick = reinterpret_cast<ExactPick*>(frames + KMC_PW_DEPTH);

    uint8_t* occ_g = p.occ + (size_t)r * n;
    uint32_t* keys_g = p.keys + (size_t)r * n;
    int4* tags_g = p.tags + (size_t)r * n;

    // ---- load the replica ------------------------------------------------
    uint64_t cb = 0;                       // lane j owns column j's bitmap
    if (lane < ny)
        for (int i = 0; i < nx; ++i) cb |= (uint64_t)(occ_g[i * ny + lane] & 1u) << i;
    col[lane] = cb;
    for (int s = lane; s < n; s += KMC_WAVE) {
        const uint32_t k = keys_g[s];
        key[s] = k;
        w[s] = (k == KMC_VACANT) ? 0.0 : p.wsum[k];
    }
    __syncthreads();
    double rsum = 0.0;                     // lane i owns the sum of row i
    for (int row = 0; row < nx; ++row) {
        const double v = lane < ny ? w[row * ny + lane] : 0.0;
        const double tot = wave_sum(v);
        if (lane == row) rsum = tot;
    }
    double rowpre = wave_scan_incl(rsum);
    double gamma = readlane_f64(rowpre, 63);

    int status = 0;
    long long n_exact = 0, hops = 0;
    long long hist_v = 0;                  // lane d < 6 counts direction d
    double time_acc = 0.0;
    double b_last = p.barrier[r * 5 + 0], b_sum = 0.0, b_min = 1e300, b_max = -1e300, b_cnt = 0.0;
    const long long hop0 = p.hops_done[r];
    const cdouble_ptr ctable = (cdouble_ptr)(uintptr_t)p.table;

    // deferred tag move (the load of one hop is consumed during the next)
    bool pend = false;
    int pend_src = 0, pend_dst = 0, pend_dwi = 0, pend_dwj = 0;
    int4 pend_tag = make_int4(0, 0, 0, 0);

    const long long my_hops = p.n_hops_r ? p.n_hops_r[r] : p.n_hops;
    for (long long h0 = 0; h0 < my_hops && status == 0; h0 += KMC_WAVE) {
        const int cnt = (int)((my_hops - h0) < KMC_WAVE ? (my_hops - h0) : KMC_WAVE);
        // ---- uniforms for the next 64 hops, one hop per lane --------------
        double rho_v = 0.5, u_v = 1.0;
        if (lane < cnt) {
            if (p.draws) {
                const double2 dv = reinterpret_cast<const double2*>(p.draws)[(size_t)r * p.n_hops + h0 + lane];
                rho_v = dv.x; u_v = dv.y;
            } else {
                const unsigned long long hh = (unsigned long long)(hop0 + h0 + lane);
                u32x4 c; c.x = (uint32_t)hh; c.y = (uint32_t)(hh >> 32); c.z = (uint32_t)r; c.w = KMC_STREAM_HOP;
                const u32x4 o = philox4x32_10(c, p.seed_lo, p.seed_hi);
                rho_v = u53(o.x, o.y); u_v = u53_open0(o.z, o.w);
            }
        }
        double gam_v = 1.0, pn_v = 1.0, pd_v = 1.0;    // per-lane record of hop (h0 + lane)
        int idx_v = -1;

        int hl = 0;
        for (; hl < cnt; ++hl) {
            const double rho = readlane_f64(rho_v, hl);
            const double t = rho * gamma;
            const double M = gamma * p.mfac;
            int i = 0, j = 0, d = 0;
            double lo = 0.0, hi = 0.0, pnum = 1.0, pden = 1.0;   // chosen rate / gamma, divided at the flush
            bool ok = (gamma > 0.0) && !p.force_exact;
            // ---- row ------------------------------------------------------
            if (ok) {
                const unsigned long long m = __ballot(rowpre >= t && lane < nx);
                ok = m != 0ull;
                i = ok ? (__ffsll((long long)m) - 1) : 0;
            }
            double rowbase = 0.0, sp = 0.0;
            if (ok) {
                rowbase = i > 0 ? readlane_f64(rowpre, i - 1) : 0.0;
                // ---- site ---------------------------------------------------
                const double v = lane < ny ? w[i * ny + lane] : 0.0;
                sp = wave_scan_incl(v);
                const unsigned long long m = __ballot(rowbase + sp >= t && lane < ny);
                ok = m != 0ull;
                j = ok ? (__ffsll((long long)m) - 1) : 0;
            }
            if (ok) {
                // ---- direction ----------------------------------------------
                const double sitebase = rowbase + (j > 0 ? readlane_f64(sp, j - 1) : 0.0);
                const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)key[i * ny + j]);
                ok = k != KMC_VACANT;
                if (ok) {
                    const cdouble_ptr tr = ctable + (size_t)k * 6;
                    const double r0 = tr[0], r1 = tr[1], r2 = tr[2], r3 = tr[3], r4 = tr[4], r5 = tr[5];
                    const double c0 = sitebase + r0, c1 = c0 + r1, c2 = c1 + r2, c3 = c2 + r3, c4 = c3 + r4,
                                 c5 = c4 + r5;
                    // running sums are non-decreasing: the first one reaching t is counted by the misses
                    d = (c0 < t) + (c1 < t) + (c2 < t) + (c3 < t) + (c4 < t) + (c5 < t);
                    ok = d < 6;
                    lo = d == 0 ? sitebase : d == 1 ? c0 : d == 2 ? c1 : d == 3 ? c2 : d == 4 ? c3 : c4;
                    hi = d == 0 ? c0 : d == 1 ? c1 : d == 2 ? c2 : d == 3 ? c3 : d == 4 ? c4 : c5;
                    pnum = d == 0 ? r0 : d == 1 ? r1 : d == 2 ? r2 : d == 3 ? r3 : d == 4 ? r4 : r5;
                    pden = gamma;
                    ok = ok && (hi - t >= M) && (t - lo >= M);
                }
            }
            if (!ok) {
                // ---- exact sequential path (rare) --------------------------------
                wave_fence();
                exact_select(key, p.table, n, rho, frames, pick);
                wave_fence();
                ++n_exact;
                const long idx = pick->idx;
                const double g = pick->gamma;
                if (!(g > 0.0)) { status = 2; break; }
                if (idx >= 6L * n) { status = 1; break; }
                d = (int)(idx % 6);
                j = (int)((idx / 6) % ny);
                i = (int)((idx / 6) / ny);
                pnum = pick->c / g - pick->cprev / g;      // the reference's own expression (:497)
                pden = 1.0;
            }
            const int s_src = i * ny + j;
            const int idx = s_src * 6 + d;
            // ---- update (:518-535) -------------------------------------------
            const int di = (d == 0 || d == 5) - (d == 2 || d == 3);
            const int dj = (d == 1 || d == 2) - (d == 4 || d == 5);
            const int ie = i + di, je = j + dj;                 // unwrapped
            const int iw = wrap1(ie, nx), jw = wrap1(je, ny);
            const int s_dst = iw * ny + jw;
            if (lane == j) cb &= ~(1ull << i);
            if (lane == jw) cb |= (1ull << iw);
            if (lane == j || lane == jw) col[lane] = cb;
            if (lane == d) ++hist_v;
            if (lane == hl) { gam_v = gamma; pn_v = pnum; pd_v = pden; idx_v = idx; }
            ++hops;
            wave_fence();
            // ---- windowed refresh (:444-465) --------------------------------------
            int i_lo, n_i, j_lo, n_j;
            if (p.window_mode == 0) {
                i_lo = (di < 0 ? ie : i) - 2; n_i = 4 + (di != 0);
                j_lo = (dj < 0 ? je : j) - 2; n_j = 4 + (dj != 0);
            } else if (p.window_mode == 1) {
                i_lo = i - 2; n_i = 5; j_lo = j - 2; n_j = 6;
            } else {
                i_lo = (di < 0 ? ie : i) - 2; n_i = 5 + (di != 0);
                j_lo = (dj < 0 ? je : j) - 2; n_j = 5 + (dj != 0);
            }
            {
                // lane = 8 * (window row) + (window column): no division by n_j
                const int a = lane >> 3, b = lane & 7;
                if (a < n_i && b < n_j) {
                    const int si = wrap1(i_lo + a, nx), sj = wrap1(j_lo + b, ny);
                    const int s = si * ny + sj;
                    uint32_t knew = KMC_VACANT;
                    double wnew = 0.0;
                    if ((col[sj] >> si) & 1ull) {
                        knew = key_from_cols([&](int c) { return col[c]; }, nx, ny, si, sj);
                        wnew = p.wsum[knew];
                    }
                    key[s] = knew;
                    w[s] = wnew;
                }
            }
            // ---- tag move, deferred by one hop (:521-528) ------------------------------
            if (!p.presim) {
                if (lane == 0) {
                    if (pend) {
                        int4 tg = pend_tag;
                        tg.z += pend_dwi; tg.w += pend_dwj;
                        tags_g[pend_dst] = tg;
                        tags_g[pend_src] = make_int4(-1, 0, 0, 0);
                    }
                    pend_tag = tags_g[s_src];
                }
                pend = true; pend_src = s_src; pend_dst = s_dst;
                pend_dwi = ie < 0 ? -1 : (ie >= nx ? 1 : 0);
                pend_dwj = je < 0 ? -1 : (je >= ny ? 1 : 0);
            }
            wave_fence();
            // ---- re-sum touched rows, re-scan -----------------------------------------
            {
                // lane = 8 * (window row a) + c: sum every 8th site of that lattice row starting
                // at c, then combine the eight partial sums inside the 8-lane group
                const int a = lane >> 3, c = lane & 7;
                const int row = wrap1(i_lo + (a < n_i ? a : 0), nx);
                const int per = (ny + 7) >> 3;
                const double* wr = w + row * ny;
                double part = 0.0;
                for (int q = 0; q < per; ++q) {
                    const int jj = q * 8 + c;
                    part += (jj < ny) ? wr[jj] : 0.0;
                }
                part += dpp_f64<0xB1, 0xF>(part);      // quad_perm [1,0,3,2]: + neighbour
                part += dpp_f64<0x4E, 0xF>(part);      // quad_perm [2,3,0,1]: + other pair
                part += dpp_f64<0x141, 0xF>(part);     // row_half_mirror: + other quad of the 8
                for (int aa = 0; aa < n_i; ++aa) {
                    const double tot = readlane_f64(part, aa * 8);
                    if (lane == wrap1(i_lo + aa, nx)) rsum = tot;
                }
            }
            rowpre = wave_scan_incl(rsum);
            gamma = readlane_f64(rowpre, 63);
        }
        // ---- flush the chunk: clock, trace, barrier diagnostics ----------------------
        const int done = hl;                // hops of this chunk that executed
        if (!p.presim) {
            const double dt = lane < done ? (-1.0 / gam_v) * log(u_v) : 0.0;      // :530
            time_acc += wave_sum(dt);
        }
        if (p.sel && lane < cnt) p.sel[(size_t)r * p.n_hops + h0 + lane] = lane < done ? idx_v : -1;
        {
            const bool has = lane < done && idx_v != 0;                          // :495
            const double bar = has ? -log((pn_v / pd_v) / 1e13) * 0.025 : 0.0;  // :496-498
            const unsigned long long m = __ballot(has);
            if (m) {
                b_last = readlane_f64(bar, 63 - __builtin_clzll(m));
                b_sum += wave_sum(bar);
                double mn = has ? bar : 1e300, mx = has ? bar : -1e300;
                for (int off = 32; off > 0; off >>= 1) {
                    mn = fmin(mn, __shfl_xor(mn, off, 64));
                    mx = fmax(mx, __shfl_xor(mx, off, 64));
                }
                b_min = fmin(b_min, mn); b_max = fmax(b_max, mx);
                b_cnt += (double)__popcll(m);
            }
        }
        if (status != 0 && p.sel) {
            // mark the hops never attempted in later chunks
            for (long long hh = h0 + KMC_WAVE + lane; hh < p.n_hops; hh += KMC_WAVE)
                p.sel[(size_t)r * p.n_hops + hh] = -1;
        }
    }
    // ---- finish the pending tag move, store the replica -------------------------------
    if (!p.presim && pend && lane == 0) {
        int4 tg = pend_tag;
        tg.z += pend_dwi; tg.w += pend_dwj;
        tags_g[pend_dst] = tg;
        tags_g[pend_src] = make_int4(-1, 0, 0, 0);
    }
    wave_fence();
    if (lane < ny)
        for (int i = 0; i < nx; ++i) occ_g[i * ny + lane] = (uint8_t)((cb >> i) & 1ull);
    for (int s = lane; s < n; s += KMC_WAVE) keys_g[s] = key[s];
    if (lane < 6) p.hist[r * 6 + lane] += hist_v;
    if (lane == 0) {
        p.time_total[r] += time_acc;
        p.step_time[r] = time_acc;
        p.gamma[r] = gamma;
        p.hops_done[r] = hop0 + hops;
        p.n_exact[r] += n_exact;
        p.status[r] |= status;
        p.barrier[r * 5 + 0] = b_last;
        p.barrier[r * 5 + 1] += b_sum;
        p.barrier[r * 5 + 2] = fmin(p.barrier[r * 5 + 2], b_min);
        p.barrier[r * 5 + 3] = fmax(p.barrier[r * 5 + 3], b_max);
        p.barrier[r * 5 + 4] += b_cnt;
    }
}

}  // namespace kmc
