// kmc_hop_global.h -- the hop loop for lattices that do not fit the LDS-resident
// kernel (nx or ny > 64, e.g. BASELINE config 4: one 1024 x 1024 lattice).
//
// One 256-thread workgroup = one replica.  Per-site state (stale keys, per-site
// rate sums w, byte occupancy, tags) stays in global memory (L2 / Infinity Cache);
// only the nx row sums live in LDS.  The algorithm is the same certified
// hierarchical selection as kmc_hop_lds.h, with block-wide scans:
//   rows  -> 256 chunk sums + block scan -> first chunk reaching t -> row
//   sites -> the same over w[row][0..ny)
//   dir   -> six table rates
// followed by the window refresh (keys from byte occupancy), the re-sum of the
// touched rows and, when the certification margin is not met, the exact
// block-parallel replay of the reference's np.sum / sequential cumsum (kmc_exact_par.h).
// Per hop it reads O(nx + 6*ny) values instead of the reference's 6*nx*ny.
#pragma once
#include "kmc_device.h"
#include "kmc_hop_lds.h"
#include "kmc_exact_par.h"

namespace kmc {

#define KMC_GB_THREADS 256
#define KMC_GB_WAVES 4

struct HopGlobalParams {
    int nx, ny, n, window_mode;
    long long n_hops;
    const long long* n_hops_r;
    int presim, force_exact;
    double mfac;                 // the flat margin factor of the LDS kernels (unused here)
    double mu;                   // 2^-53 * 1.001 * margin_scale: unit of the position-dependent margin (kmc_hop_big.h)
    const double* table;
    const double* wsum;
    uint8_t* occ;
    uint32_t* keys;
    double* w;                   // [R][n] per-site rate sums (scratch state, rebuilt on entry)
    int4* tags;
    const double* draws;
    int32_t* sel;
    uint32_t seed_lo, seed_hi;
    double* time_total; double* step_time; double* gamma; long long* hist; long long* hops_done;
    long long* n_exact; int* status; double* barrier;
    uint32_t rid_base;           // global id of replica 0 (see HopParams)
};

struct GbShared {
    double wave_tot[KMC_GB_WAVES];
    int first_thread;
    int i, j, d, ok;
    double rowbase, sitebase, lo, hi, pnorm;
    ExactParShared xp;           // exact path (kmc_exact_par.h)
};

// inclusive block scan (fixed association): wave scan, then the totals of the
// preceding waves added left to right
__device__ __forceinline__ double block_scan_incl(double x, GbShared* sh, double* total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s = wave_scan_incl(x);
    if (lane == 63) sh->wave_tot[wv] = s;
    __syncthreads();
    double pre = 0.0, tot = 0.0;
#pragma unroll
    for (int q = 0; q < KMC_GB_WAVES; ++q) {
        if (q < wv) pre += sh->wave_tot[q];
        tot += sh->wave_tot[q];
    }
    __syncthreads();
    *total = tot;
    return pre + s;
}

// index of the first thread whose flag is set (KMC_GB_THREADS if none)
__device__ __forceinline__ int block_first(bool flag, GbShared* sh)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) sh->first_thread = KMC_GB_THREADS;
    __syncthreads();
    const unsigned long long m = __ballot(flag);
    if (lane == 0 && m) atomicMin(&sh->first_thread, wv * 64 + (__ffsll((long long)m) - 1));
    __syncthreads();
    const int f = sh->first_thread;
    __syncthreads();
    return f;
}

__global__ __launch_bounds__(KMC_GB_THREADS) void hop_global_kernel(HopGlobalParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GbShared* sh = reinterpret_cast<GbShared*>(smem);
    double* rowsum = reinterpret_cast<double*>(smem + ((sizeof(GbShared) + 15) & ~(size_t)15));
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int nx = p.nx, ny = p.ny, n = p.n;
    uint8_t* occ = p.occ + (size_t)r * n;
    uint32_t* keys = p.keys + (size_t)r * n;
    double* w = p.w + (size_t)r * n;
    int4* tags = p.tags + (size_t)r * n;
    const int cr = (nx + KMC_GB_THREADS - 1) / KMC_GB_THREADS;      // rows per thread
    const int cs = (ny + KMC_GB_THREADS - 1) / KMC_GB_THREADS;      // sites per thread (within a row)

    // deterministic row sum: per-thread chunks of the row, then the block scan's total
    auto row_total = [&](int row) -> double {
        double loc = 0.0;
        for (int q = 0; q < cs; ++q) {
            const int j = tid * cs + q;
            if (j < ny) loc += w[(size_t)row * ny + j];
        }
        double tot;
        block_scan_incl(loc, sh, &tot);
        return tot;
    };

    // ---- load: w from the stale keys, row sums
    for (int s = tid; s < n; s += KMC_GB_THREADS) {
        const uint32_t k = keys[s];
        w[s] = (k == KMC_VACANT) ? 0.0 : p.wsum[k];
    }
    __syncthreads();
    for (int row = 0; row < nx; ++row) {
        const double tot = row_total(row);
        if (tid == 0) rowsum[row] = tot;
    }
    __syncthreads();

    int status = 0;
    long long n_exact = 0, hops = 0;
    long long hist[6] = {0, 0, 0, 0, 0, 0};                // thread 0
    double time_acc = 0.0, gamma = 0.0;
    double b_last = p.barrier[r * 5 + 0], b_sum = 0.0, b_min = 1e300, b_max = -1e300, b_cnt = 0.0;
    const long long hop0 = p.hops_done[r];
    const long long my_hops = p.n_hops_r ? p.n_hops_r[r] : p.n_hops;

    for (long long h = 0; h < my_hops; ++h) {
        // ---- draws (uniform across the block)
        double rho, u;
        if (p.draws) {
            const double2 dv = reinterpret_cast<const double2*>(p.draws)[(size_t)r * p.n_hops + h];
            rho = dv.x; u = dv.y;
        } else {
            const unsigned long long hh = (unsigned long long)(hop0 + h);
            u32x4 c; c.x = (uint32_t)hh; c.y = (uint32_t)(hh >> 32); c.z = (uint32_t)r + p.rid_base; c.w = KMC_STREAM_HOP;
            const u32x4 o = philox4x32_10(c, p.seed_lo, p.seed_hi);
            rho = u53(o.x, o.y); u = u53_open0(o.z, o.w);
        }
        // ---- rows
        double loc = 0.0;
        for (int q = 0; q < cr; ++q) {
            const int row = tid * cr + q;
            if (row < nx) loc += rowsum[row];
        }
        const double incl = block_scan_incl(loc, sh, &gamma);
        const double t = rho * gamma;
        // Margin of the chosen entry, as in kmc_hop_big.h: the reference's sequential sum has rounded at most 6 (site + 1)
        // times when it reaches the entry, each time by <= u relative to a partial sum <= hi; this kernel's own sums (< 70
        // roundings), np.sum's 6n / 8192 + 30 and the quotient's are in xtra.  A third of the flat gamma * 6n * u on average:
        // at 2048 x 2048, 1.4 % of the hops were replayed with the flat margin.
        const double xtra = 6.0 * (double)n / 8192.0 + 512.0;
        bool ok = (gamma > 0.0) && !p.force_exact;
        int first = block_first(ok && incl >= t, sh);
        if (first == KMC_GB_THREADS) ok = false;
        if (ok && tid == first) {
            double c = incl - loc;
            int found = 0;
            for (int q = 0; q < cr; ++q) {
                const int row = tid * cr + q;
                if (row >= nx) break;
                const double cn = c + rowsum[row];
                if (cn >= t) { sh->i = row; sh->rowbase = c; found = 1; break; }
                c = cn;
            }
            sh->ok = found;
        }
        if (!ok && tid == 0) sh->ok = 0;
        __syncthreads();
        ok = sh->ok != 0;
        int i = sh->i;
        // ---- sites of row i
        if (ok) {
            const double rowbase = sh->rowbase;
            double sl = 0.0;
            for (int q = 0; q < cs; ++q) {
                const int j = tid * cs + q;
                if (j < ny) sl += w[(size_t)i * ny + j];
            }
            double rtot;
            const double sincl = block_scan_incl(sl, sh, &rtot);
            first = block_first(rowbase + sincl >= t, sh);
            if (first == KMC_GB_THREADS) {
                ok = false;
            } else if (tid == first) {
                double c = rowbase + (sincl - sl);
                int found = 0;
                for (int q = 0; q < cs; ++q) {
                    const int j = tid * cs + q;
                    if (j >= ny) break;
                    const double cn = c + w[(size_t)i * ny + j];
                    if (cn >= t) { sh->j = j; sh->sitebase = c; found = 1; break; }
                    c = cn;
                }
                if (found) {
                    // ---- direction + certification
                    const int j = sh->j;
                    const uint32_t k = keys[i * ny + j];
                    found = 0;
                    if (k != KMC_VACANT) {
                        double cc = sh->sitebase;
                        for (int dd = 0; dd < 6; ++dd) {
                            const double rate = p.table[(size_t)k * 6 + dd];
                            const double cn = cc + rate;
                            if (cn >= t) { found = 1; sh->d = dd; sh->lo = cc; sh->hi = cn; sh->pnorm = rate / gamma; break; }
                            cc = cn;
                        }
                        if (found) {
                            const double M = sh->hi * (p.mu * (6.0 * (double)(i * ny + j + 1) + xtra));
                            if (!((sh->hi - t >= M) && (t - sh->lo >= M))) found = 0;
                        }
                    }
                }
                sh->ok = found;
            }
            __syncthreads();
            if (ok) ok = sh->ok != 0;
        }
        int j, d;
        double pnorm;
        if (!ok) {
            __syncthreads();
            // (force_exact = 1: always the replay's long form; otherwise its short form where np.sum cannot matter)
            exact_select_par(keys, p.table, n, rho, &sh->xp, p.force_exact == 1 ? 0.0 : gamma,
                             p.mu * xtra + 8.0 * 1.1102230246251565e-16);
            ++n_exact;
            const long long idx = sh->xp.idx;
            const double g = sh->xp.gamma;
            if (!(g > 0.0)) { status = 2; break; }
            if (idx >= 6LL * n) { status = 1; break; }
            d = (int)(idx % 6);
            j = (int)((idx / 6) % ny);
            i = (int)((idx / 6) / ny);
            pnorm = sh->xp.c_found / g - sh->xp.cprev_found / g;
        } else {
            j = sh->j; d = sh->d; pnorm = sh->pnorm;
        }
        __syncthreads();
        const int s_src = i * ny + j;
        const int idx = s_src * 6 + d;
        const int di = (d == 0 || d == 5) - (d == 2 || d == 3);
        const int dj = (d == 1 || d == 2) - (d == 4 || d == 5);
        const int ie = i + di, je = j + dj;
        const int iw = wrap1(ie, nx), jw = wrap1(je, ny);
        const int s_dst = iw * ny + jw;
        // ---- update (:518-535)
        if (tid == 0) {
            occ[s_src] = 0;
            occ[s_dst] = 1;
            ++hist[d];
            if (!p.presim) {
                int4 tg = tags[s_src];
                tg.z += ie < 0 ? -1 : (ie >= nx ? 1 : 0);
                tg.w += je < 0 ? -1 : (je >= ny ? 1 : 0);
                tags[s_dst] = tg;
                tags[s_src] = make_int4(-1, 0, 0, 0);
                time_acc += (-1.0 / gamma) * log(u);                              // :530-532
            }
            if (p.sel) p.sel[(size_t)r * p.n_hops + h] = idx;
            if (idx != 0) {                                                       // :495-498
                const double bar = -log(pnorm / 1e13) * 0.025;
                b_last = bar; b_sum += bar; b_min = fmin(b_min, bar); b_max = fmax(b_max, bar); b_cnt += 1.0;
            }
        }
        ++hops;
        __syncthreads();
        // ---- windowed refresh (:444-465)
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
        if (tid < n_i * n_j) {
            const int a = tid / n_j, b = tid - a * n_j;
            int si = (i_lo + a) % nx; if (si < 0) si += nx;
            int sj = (j_lo + b) % ny; if (sj < 0) sj += ny;
            const int s = si * ny + sj;
            uint32_t knew = KMC_VACANT;
            double wnew = 0.0;
            if (occ[s]) {
                knew = key_from_bytes(occ, nx, ny, si, sj);
                wnew = p.wsum[knew];
            }
            keys[s] = knew;
            w[s] = wnew;
        }
        __syncthreads();
        for (int a = 0; a < n_i; ++a) {
            int row = (i_lo + a) % nx; if (row < 0) row += nx;
            const double tot = row_total(row);
            if (tid == 0) rowsum[row] = tot;
        }
        __syncthreads();
    }
    // ---- trace entries never reached, final gamma, scalars
    if (p.sel)
        for (long long hh = hops + tid; hh < p.n_hops; hh += KMC_GB_THREADS) p.sel[(size_t)r * p.n_hops + hh] = -1;
    {
        double loc = 0.0;
        for (int q = 0; q < cr; ++q) {
            const int row = tid * cr + q;
            if (row < nx) loc += rowsum[row];
        }
        block_scan_incl(loc, sh, &gamma);
    }
    if (tid == 0) {
        for (int q = 0; q < 6; ++q) p.hist[r * 6 + q] += hist[q];
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
    (void)lane;
}

}  // namespace kmc
