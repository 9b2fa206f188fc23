/* CPU oracle, C twin of oracle/kmc_oracle.py -- TEST INFRASTRUCTURE ONLY.
 *
 * A single-thread restatement of the reference's per-hop algorithm
 * (quentinstoyel/KMC, hex_LiCoO2_kmc.py:418-542) with numpy's floating-point
 * association spelled out, so that parity tests at N = 64 and bench.py's
 * cpu_baseline leg finish in seconds.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline may load this; the product never does.
 *
 * Parity status: PINNED through tests/test_oracle_c.py, which checks it hop for
 * hop against oracle/kmc_oracle.py (itself pinned on the reference-generated
 * golden vectors) and its np.sum model against np.sum.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KMCO_VACANT 0xFFFFFFFFu

/* key char p -> (di, dj)                         hex_LiCoO2_kmc.py:423-424 */
static const int NB[18][2] = {
    {0, 2}, {-1, 2}, {-2, 2}, {1, 1}, {0, 1}, {-1, 1}, {-2, 1}, {2, 0}, {1, 0},
    {-1, 0}, {-2, 0}, {2, -1}, {1, -1}, {0, -1}, {-1, -1}, {2, -2}, {1, -2}, {0, -2}};
/* dir -> (di, dj)                                hex_LiCoO2_kmc.py:503-504 */
static const int DIRS[6][2] = {{1, 0}, {0, 1}, {-1, 1}, {-1, 0}, {0, -1}, {1, -1}};
/* hex_dict_builder.py:41 */
static const int ENDP[6] = {8, 4, 5, 9, 13, 12};
/* hex_dict_builder.py:56-134, free slots of the 16-char key in order */
static const int VIEWS[6][2][9] = {
    {{5, 9, 4, 13, 3, 16, 7, 11, 15}, {13, 9, 12, 5, 11, 1, 7, 3, 0}},
    {{9, 13, 5, 12, 1, 11, 0, 3, 7}, {12, 13, 8, 9, 3, 6, 0, 1, 2}},
    {{13, 12, 9, 8, 6, 3, 2, 1, 0}, {8, 12, 4, 13, 1, 14, 2, 6, 10}},
    {{12, 8, 13, 4, 14, 1, 10, 6, 2}, {4, 8, 5, 12, 6, 16, 10, 14, 17}},
    {{8, 4, 12, 5, 16, 6, 17, 14, 10}, {5, 4, 9, 8, 14, 11, 17, 16, 15}},
    {{4, 5, 8, 9, 11, 14, 15, 16, 17}, {9, 5, 13, 4, 16, 3, 15, 11, 7}}};

static inline int pmod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }
static inline int fdiv(int a, int n) { return (a - pmod(a, n)) / n; }   /* python // */

/* hex_dict_builder.py:38-142 */
void kmco_build_table(const double *base512, double dumbbell, double *table)
{
    for (uint32_t row = 0; row < (1u << 18); ++row) {
        for (int e = 0; e < 6; ++e) {
#define BIT(p) ((row >> (17 - (p))) & 1u)
            double rate;
            uint32_t end = BIT(ENDP[e]), prev = BIT(ENDP[(e + 5) % 6]), next = BIT(ENDP[(e + 1) % 6]);
            if (end) rate = 0.0;
            else if (prev && next) rate = dumbbell;
            else {
                double r1 = 0.0, r2 = 0.0;
                if (!prev) { uint32_t ix = 0; for (int q = 0; q < 9; ++q) ix |= BIT(VIEWS[e][0][q]) << (8 - q); r1 = base512[ix]; }
                if (!next) { uint32_t ix = 0; for (int q = 0; q < 9; ++q) ix |= BIT(VIEWS[e][1][q]) << (8 - q); r2 = base512[ix]; }
                rate = r1 + r2;
            }
#undef BIT
            table[(size_t)row * 6 + e] = rate;
        }
    }
}

/* hex_LiCoO2_kmc.py:422-427 */
static uint32_t site_key(const uint8_t *occ, int nx, int ny, int i, int j)
{
    uint32_t key = 0;
    for (int p = 0; p < 18; ++p)
        key = (key << 1) | occ[pmod(i + NB[p][0], nx) * ny + pmod(j + NB[p][1], ny)];
    return key;
}

static void refresh_site(const uint8_t *occ, int nx, int ny, const double *table,
                         double *P, uint32_t *keys, int i, int j)
{
    size_t s = (size_t)i * ny + j;
    if (occ[s]) {
        uint32_t k = site_key(occ, nx, ny, i, j);
        memcpy(P + s * 6, table + (size_t)k * 6, 6 * sizeof(double));
        if (keys) keys[s] = k;
    } else {
        memset(P + s * 6, 0, 6 * sizeof(double));
        if (keys) keys[s] = KMCO_VACANT;
    }
}

/* kmc_evolve, initialise mode: hex_LiCoO2_kmc.py:435-443, :453-465 */
void kmco_rates_full(const uint8_t *occ, int nx, int ny, const double *table,
                     double *P, uint32_t *keys)
{
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j) refresh_site(occ, nx, ny, table, P, keys, i, j);
}

/* numpy DOUBLE pairwise sum (numpy/_core/src/umath/loops_utils.h.src), as np.sum
 * applies it: per 8192-element buffer, buffers accumulated left to right. */
static double pairwise(const double *a, long n)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        long i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise(a, n2) + pairwise(a + n2, n - n2);
    }
}

double kmco_numpy_sum(const double *a, long n)
{
    double res = 0.;
    for (long lo = 0; lo < n; lo += 8192) res = res + pairwise(a + lo, (n - lo) < 8192 ? (n - lo) : 8192);
    return res;
}

/* window_mode: 0 hex_ref (hex_LiCoO2_kmc.py:446-449), 1 obj_ref (Mc_Lattice.py:178),
 * 2 full (inclusive bounding box of both radius-2 hexagons) */
static void window(int mode, int is, int js, int ie, int je, int *ilo, int *ihi, int *jlo, int *jhi)
{
    int imin = is < ie ? is : ie, imax = is < ie ? ie : is;
    int jmin = js < je ? js : je, jmax = js < je ? je : js;
    if (mode == 0) { *ilo = imin - 2; *ihi = imax + 2; *jlo = jmin - 2; *jhi = jmax + 2; }
    else if (mode == 1) { *ilo = is - 2; *ihi = is + 3; *jlo = js - 2; *jhi = js + 4; }
    else { *ilo = imin - 2; *ihi = imax + 3; *jlo = jmin - 2; *jhi = jmax + 3; }
}

/* The hop loop, hex_LiCoO2_kmc.py:484-535, for one lattice.
 *   occ   [nx*ny] 1 = Li;  P [nx*ny*6];  keys [nx*ny] or NULL (stale key per site)
 *   tags  [nx*ny*3] int64 = i0, j0, 100*wi+wj  (ignored when presim)
 *   draws [n_hops*2]: rho, clock draw (second one unused when presim)
 *   sel_out [n_hops] or NULL, gamma_out [n_hops] or NULL
 *   hist [6] accumulated; *time accumulated (sum of dt of this call, in hop order)
 *   *barrier: last chosen_barrier (:495-498), untouched if every idx == 0
 * Returns the number of hops executed; stops early (returns h) on selection
 * overflow (idx == 6*n, the reference's IndexError) or gamma == 0. */
long kmco_run(uint8_t *occ, int nx, int ny, const double *table, double *P, uint32_t *keys,
              int64_t *tags, long n_hops, int presim, int window_mode, const double *draws,
              int32_t *sel_out, double *gamma_out, double *hist, double *time, double *barrier)
{
    const long n6 = (long)nx * ny * 6;
    double step_time = 0;
    for (long h = 0; h < n_hops; ++h) {
        const double rho = draws[2 * h];
        const double gamma = kmco_numpy_sum(P, n6);                 /* :486 */
        /* :488-492 first idx with fl(cumsum[idx] / gamma) >= rho */
        double c = 0., cprev = 0.;
        long idx = n6;
        for (long e = 0; e < n6; ++e) {
            cprev = c;
            c += P[e];
            if (c / gamma >= rho) { idx = e; break; }
        }
        if (idx == n6 || !(gamma > 0)) { *time += step_time; return h; }
        if (gamma_out) gamma_out[h] = gamma;
        if (sel_out) sel_out[h] = (int32_t)idx;
        const int d = (int)(idx % 6);                               /* :494 */
        if (idx != 0)                                               /* :495-498 */
            *barrier = -log((c / gamma - cprev / gamma) / 1e13) * 0.025;
        const int js = (int)((idx / 6) % ny), is = (int)((idx / 6) / ny);   /* :500-501 */
        const int ie = is + DIRS[d][0], je = js + DIRS[d][1];       /* unwrapped */
        const int iw = pmod(ie, nx), jw = pmod(je, ny);
        occ[is * ny + js] = 0;                                      /* :518 */
        occ[iw * ny + jw] = 1;                                      /* :519-520 */
        if (!presim) {                                              /* :521-532 */
            int64_t *ts = tags + ((size_t)is * ny + js) * 3, *te = tags + ((size_t)iw * ny + jw) * 3;
            te[0] = ts[0]; te[1] = ts[1];
            te[2] = ts[2] + 100 * fdiv(ie, nx) + fdiv(je, ny);
            ts[0] = -1; ts[1] = 0; ts[2] = 0;
            step_time = step_time + (-1. / gamma) * log(draws[2 * h + 1]);
        }
        int ilo, ihi, jlo, jhi;                                     /* :533-534, :444-465 */
        window(window_mode, is, js, ie, je, &ilo, &ihi, &jlo, &jhi);
        for (int i = ilo; i < ihi; ++i)
            for (int j = jlo; j < jhi; ++j)
                refresh_site(occ, nx, ny, table, P, keys, pmod(i, nx), pmod(j, ny));
        hist[d] += 1;                                               /* :535 */
    }
    *time += step_time;
    return n_hops;
}

/* get_distance + get_diffusion_coefficient, hex_LiCoO2_kmc.py:558-603.
 * out[0] = D*, out[1] = D_J, out[2] = sum r^2 (lattice units), out[3] = |sum r|^2 */
void kmco_diffusion(const int64_t *tags, int nx, int ny, double time_diffusing,
                    double lattice_constant, double conc, double *out)
{
    double si = 0, sj = 0, sumsq = 0;
    long cnt = 0;
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j) {
            const int64_t *t = tags + ((size_t)i * ny + j) * 3;
            if (t[0] < 0) continue;
            double wi = nearbyint((double)t[2] / 100);              /* py3 round() */
            double ii = i + nx * wi, jj = j + ny * ((double)t[2] - wi * 100);
            double di = (double)t[0] - ii, dj = (double)t[1] - jj;
            double r = sqrt(fabs(di * di + dj * dj + di * dj)) * lattice_constant;
            sumsq += r * r;
            si += di; sj += dj; ++cnt;
        }
    double n_atoms = (double)nx * ny * 1 * conc;
    double dj_len = sqrt(fabs(si * si + sj * sj + si * sj)) * lattice_constant;
    out[0] = 1 / (2 * 2 * time_diffusing / n_atoms) * (sumsq / cnt);
    out[1] = 1 / (2 * 2 * time_diffusing / n_atoms) * (dj_len * dj_len) / n_atoms;
    out[2] = sumsq / (lattice_constant * lattice_constant);
    out[3] = si * si + sj * sj + si * sj;
}
