// kmc_hop_lds.h -- the fused, LDS-resident hop loop (rows S1-S5, U1-U4, R3, K).
//
// One wave64 = one replica = one workgroup.  Per replica in LDS (64 x 64: 21 KB,
// so 7 replicas per CU instead of the 3 a per-site fp64 array would allow):
//   key[n]   u32  18-bit key of each site at its last refresh (stale keys are
//                 state: the reference's window leaves 6-7 affected sites
//                 un-refreshed per hop)
//   blk[nx*8] f64 sum of the rates of each block of 8 consecutive sites of a row
//   col[ny]  u64  occupancy, one bitmap per lattice column (bit i = row i)
// and in registers: lane i holds the sum of row i and its inclusive prefix.
// Per-site rate sums are NOT stored: W[key] (2 MB, L2 resident) is gathered when
// a site's weight is needed, T[key] (12.6 MB) when its six rates are.
//
// Per hop (hex_LiCoO2_kmc.py:484-535):
//   t = rho * gamma                                            (:486-490)
//   row    first lane with rowprefix >= t                      ballot on registers
//   block  first of the row's 8 block sums reaching t          8 LDS reads + DPP scan
//   entry  the block's 8 sites x 6 directions = 48 lanes gather T[key][dir]
//          (one L2 round trip), DPP scan, first lane reaching t -> site AND direction
//   certify (below) or exact_select()
//   update bitmaps / tags / clock / histogram                  (:518-535)
//   refresh the <= 36 window sites: new keys from the bitmaps   (:444-465)
//   re-sum the <= 18 blocks the window touches from W[key] (8 lanes per block, fixed
//   tree), re-sum the <= 6 touched rows from their blocks, re-scan the row prefix.
//   Every stored sum is a pure function of the keys: nothing is updated by
//   differences, so nothing drifts.
//
// Exactness.  The reference picks the first idx with fl(cumsum_seq[idx]/np.sum) >= rho.
// Our running sums associate differently, so the pick is accepted only if t is
// further than a rigorous bound M from both neighbouring boundaries (lo, hi):
//     M = gamma * (6n (1 + 1/8192) + 512) * 2^-53 * 1.001         (HopParams::mfac, set by the host)
//   6n        rounding of the reference's sequential sum (k <= 6n terms)
//   6n/8192   np.sum adds its 8192-element buffers left to right
//   512       our own <= 40 roundings (5 per W, 3 per block, 7 per row, 6 + 3 + 6 in the scans), the
//             ~30 roundings inside one np.sum buffer, fl(rho*gamma), the 2u of fl(c/gamma) >= rho
// Otherwise (about 3e-8 of hops at 64x64) exact_select() replays the reference.
#pragma once
#include "kmc_device.h"
#include "kmc_exact_par.h"

namespace kmc {

struct HopParams {
    int nx, ny, n, window_mode;
    long long n_hops;            // hops per replica (stride of draws / sel)
    const long long* n_hops_r;   // [R] per-replica hop counts (<= n_hops) or null
    int presim, force_exact;
    double mfac;                 // (6n (1 + 1/8192) + 512) * 2^-53 * 1.001 * margin_scale
    const double* table;         // [2^18][6]
    const double* wsum;          // [2^18]
    const double* tcompact;      // [6][2^13] per-direction compact table (hop_lds64_kernel<true>) or null
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
    long long* stamps;           // [R][8] cycles per segment (-DKMC_STAMPS profiling builds only) or null
    void* xscratch;              // [R][Q64_XSCRATCH] exact-path scratch of hop_q64_kernel
    const double* tcompact2;     // [3][2^15][2] pair tables (hop_q64_kernel<true>) or null
    int r_base;                  // replica of block 0 (a run may be launched as several groups)
    const int* order;            // [R] or null: block b runs replica order[b] (kmc_run_counts: longest runs first, so that
                                 // the replicas with few hops fill the tail of the launch instead of the other way round)
    uint32_t rid_base;           // global id of the context's replica 0 (kmc_set_replica_offset): the Philox counter
                                 // word is rid_base + r, so an ensemble sharded over contexts / GPUs draws what one context would
};

// Profiling builds (-DKMC_STAMPS, scripts/stamps.py) time the segments of a hop with s_memtime.
// -DKMC_STAMPS=2 (hop_q64_kernel): no stamps inside the loop, only each wave's shader-clock and real-time
// (100 MHz) counters at its start and end: the clock the chip held and when each wave of a launch ended.
#if defined(KMC_STAMPS) && KMC_STAMPS != 2
#define KMC_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                          st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define KMC_STAMP(k) do {} while (0)
#endif

// The SIMD's arbiter serves its oldest wave first: left alone, the waves of a SIMD finish one after the other
// (hop_q64_kernel: 2.9 ... 4.4 ms for the same 2048 hops, scripts/wave_clocks.py) and a launch waits for the
// last one, which ends up alone on its SIMD.  With priorities that rotate, a new one every 64-hop chunk, they
// advance together (3.5 ... 3.85 ms): -12 % per launch at four waves per SIMD.  The priority is
// (slot on the SIMD - chunks done) mod 4: distinct for waves in step, and a wave that gets a chunk ahead drops
// below its neighbours (rotating the other way, + chunks done, is 3 % slower on long launches; cycling through
// all 24 orders of the four priorities or rotating every 16, 4 or 1 hops is no better).
__device__ __forceinline__ int kmc_wave_slot()
{
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hwid));          // WAVE_ID: slot on the SIMD
    return (int)hwid;
}
__device__ __forceinline__ void kmc_rotate_priority(int k)
{
    switch (k & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// The workgroup is ONE wave: its LDS instructions execute in issue order, so a
// cross-lane LDS hand-off needs no s_barrier and no s_waitcnt, only a point the
// compiler may not move memory operations across.  (__syncthreads() would also
// drain vmcnt and stall on the deferred tag load.)
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }


// Literal restatement of hex_LiCoO2_kmc.py:486-492 on the stale-key state:
// gamma with np.sum's association, strictly sequential cumsum, first index whose
// normalised value reaches rho.  Every lane runs it redundantly (uniform).
// The reference's selection replayed exactly by the replica's one wave (kmc_exact_par.h: np.sum's tree,
// the sequential cumsum as binade-wise integer prefix sums).  A literal one-lane replay of the 6n entries
// takes milliseconds at 64 x 64, and a launch ends when its slowest replica does.
__device__ __noinline__ void exact_select(const uint32_t* key, const double* table, int n, double rho, ExactParShared* sh)
{
    exact_select_par_impl<1, 2, true>([=](long s) -> uint32_t { return key[s]; }, table, n, rho, sh);
}


__host__ __device__ inline size_t hop_lds_layout(int nx, int n, size_t* off_blk, size_t* off_col, size_t* off_frames)
{
    size_t o = ((size_t)4 * n + 15) & ~(size_t)15;           // key
    *off_blk = o; o += (size_t)nx * 8 * 8;                   // blk
    *off_col = o; o += 64 * 8;                               // col
    *off_frames = o; o += (sizeof(ExactParShared) + 15) & ~(size_t)15;
    return o;
}

// Two builds of one body.  hop_lds_kernel: at least 3 waves per SIMD (<= 168 registers; the budget also binds the
// exact path the kernel calls, which would take 226 registers left alone), 12 replicas per CU.  hop_lds_kernel_w4:
// 4 waves per SIMD (128 registers, a few spilled), for lattices whose LDS footprint lets 16 replicas share a CU
// (<= 10 KB: up to 38 x 38) when there are more replicas than 12 per CU: +12 ... +26 % on ensembles of 10 x 10 ...
// 32 x 32, while at 40 x 40 and beyond (15 or fewer per CU anyway) the spills cost 8-14 %, and a single lattice
// (latency-bound) loses 4 %: there the 3-wave build stays.
static __device__ __forceinline__ void hop_lds_body(const HopParams& p, unsigned char* smem);

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void hop_lds_kernel(HopParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    hop_lds_body(p, smem);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void hop_lds_kernel_w4(HopParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    hop_lds_body(p, smem);
}

static __device__ __forceinline__ void hop_lds_body(const HopParams& p, unsigned char* smem)
{
    const int r = p.order ? p.order[blockIdx.x + p.r_base] : (int)blockIdx.x + p.r_base, lane = threadIdx.x;
    const int nx = p.nx, ny = p.ny, n = p.n;
    size_t off_blk, off_col, off_frames;
    hop_lds_layout(nx, n, &off_blk, &off_col, &off_frames);
    uint32_t* key = reinterpret_cast<uint32_t*>(smem);
    double* blk = reinterpret_cast<double*>(smem + off_blk);
    uint64_t* col = reinterpret_cast<uint64_t*>(smem + off_col);
    ExactParShared* xsh = reinterpret_cast<ExactParShared*>(smem + off_frames);

    uint8_t* occ_g = p.occ + (size_t)r * n;
    uint32_t* keys_g = p.keys + (size_t)r * n;
    int4* tags_g = p.tags + (size_t)r * n;
    const double* __restrict__ wsum = p.wsum;
    const double* __restrict__ table = p.table;

    // ---- load the replica ------------------------------------------------
    uint64_t cb = 0;                       // lane j owns column j's bitmap
    if (lane < ny)
        for (int i = 0; i < nx; ++i) cb |= (uint64_t)(occ_g[i * ny + lane] & 1u) << i;
    col[lane] = cb;
    for (int s = lane; s < n; s += KMC_WAVE) key[s] = keys_g[s];
    wave_fence();

    // per-lane constants: lane = 8 * a + c (8-lane groups)
    const int grp_a = lane >> 3, grp_c = lane & 7;
    const int nb = (ny + 7) >> 3;                      // blocks of 8 sites per row
    const int nslots = nb < ((ny & 7) ? 3 : 2) ? nb : ((ny & 7) ? 3 : 2);   // blocks a window row can touch

    // Every block sum is the fixed tree ((w0+w1)+(w2+w3))+((w4+w5)+(w6+w7)) over 8 lanes,
    // every row sum the left-to-right sum of its 8 block sums.
    for (int row = 0; row < nx; ++row) {
        const uint32_t k = lane < ny ? key[row * ny + lane] : KMC_VACANT;
        const double bs = group8_sum(k == KMC_VACANT ? 0.0 : wsum[k]);
        if (grp_c == 0) blk[row * 8 + grp_a] = bs;
    }
    wave_fence();
    double rsum = lane < nx ? row_sum8(blk + lane * 8) : 0.0;   // lane i owns the sum of row i
    double rowpre = wave_scan_incl(rsum);
    double gamma = readlane_f64(rowpre, 63);

    int status = 0;
    long long n_exact = 0, hops = 0;
    long long hist_v = 0;                  // lane d < 6 counts direction d
    double time_acc = 0.0;
    double b_last = p.barrier[r * 5 + 0], b_sum = 0.0, b_min = 1e300, b_max = -1e300, b_cnt = 0.0;
    const long long hop0 = p.hops_done[r];

    // Displacement tags (:521-528) move once per 64-hop chunk (issuing a vector store costs a wave ~120
    // cycles, and the tag values never feed back into the dynamics).  Lane = one atom the chunk has moved:
    // tg_sv = where it is now | the site it entered the chunk from << 16 (0xFFFF: lane unused), tg_ow = the
    // wraps gathered since, (wi + 64) << 8 | (wj + 64) << 16.  An atom that moves again keeps its lane; a
    // new atom takes lane hl (its hop's number within the chunk, free by construction).
    int tg_sv = -1, tg_ow = 0;

    // per-lane constants of the 48-lane entry pick: lane = 6 * (site in block) + direction
    const int ent_site = (lane * 43) >> 8;             // lane / 6 for lane < 48
    const int ent_dir = lane - 6 * ent_site;

    // Move the tracked tags to memory.  Called at every chunk's end, and around a hop with rho <= 0: that hop
    // takes entry 0 whatever its rate, so it may leave a vacant site or land on an occupied one,
    // which the tracking does not model -- it is made a tracking period of its own.
    auto flush_tags = [&]() {
        const unsigned long long tg_used = p.presim ? 0ull : __ballot((tg_sv & 0xFFFF) != 0xFFFF);
        if (tg_used) {
            // memory still holds the tags as they were before the chunk: every moved atom loads its own
            // from the site it left, adds its wraps and stores it where it is now; a site that was left
            // is vacant now unless it is occupied (then a moved atom sits on it and stores its tag there)
            const bool used = (tg_used >> lane) & 1ull;
            const int tg_site = tg_sv & 0xFFFF, tg_vac = used ? (int)((unsigned)tg_sv >> 16) : 0;
            int4 tv = make_int4(0, 0, 0, 0);
            if (used) tv = tags_g[tg_vac];
            const int vi = tg_vac / ny, vj = tg_vac - vi * ny;
            const bool home_empty = ((col[vj] >> vi) & 1ull) == 0ull;
            tv.z += ((tg_ow >> 8) & 255) - 64;
            tv.w += ((tg_ow >> 16) & 255) - 64;
            if (used) tags_g[tg_site] = tv;
            if (used && home_empty) tags_g[tg_vac] = make_int4(-1, 0, 0, 0);
            tg_sv = -1;
        }
    };
    bool tg_flush_after = false;
    const long long my_hops = p.n_hops_r ? p.n_hops_r[r] : p.n_hops;
#ifdef KMC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    const int wave_slot = kmc_wave_slot();
    for (long long h0 = 0; h0 < my_hops && status == 0; h0 += KMC_WAVE) {
        const int cnt = (int)((my_hops - h0) < KMC_WAVE ? (my_hops - h0) : KMC_WAVE);
        kmc_rotate_priority(wave_slot - (int)(h0 >> 6));       // the waves of a SIMD advance together (kmc_hop_lds.h)
        // ---- uniforms for the next 64 hops, one hop per lane --------------
        double rho_v = 0.5, u_v = 1.0;
        if (lane < cnt) {
            if (p.draws) {
                const double2 dv = reinterpret_cast<const double2*>(p.draws)[(size_t)r * p.n_hops + h0 + lane];
                rho_v = dv.x; u_v = dv.y;
            } else {
                const unsigned long long hh = (unsigned long long)(hop0 + h0 + lane);
                u32x4 c; c.x = (uint32_t)hh; c.y = (uint32_t)(hh >> 32); c.z = (uint32_t)r + p.rid_base; c.w = KMC_STREAM_HOP;
                const u32x4 o = philox4x32_10(c, p.seed_lo, p.seed_hi);
                rho_v = u53(o.x, o.y); u_v = u53_open0(o.z, o.w);
            }
        }
        double gam_v = 1.0, pn_v = 1.0, pd_v = 1.0;    // per-lane record of hop (h0 + lane)
        int idx_v = -1;

        int hl = 0;
        KMC_STAMP(0);
        for (; hl < cnt; ++hl) {
            const double rho = readlane_f64(rho_v, hl);
            const double t = rho * gamma;
            const double M = gamma * p.mfac;
            int i = 0, j = 0, d = 0;
            double pnum = 1.0, pden = 1.0;          // chosen rate / gamma, divided at the flush
            bool ok = (gamma > 0.0) && !p.force_exact;
            // ---- row ------------------------------------------------------
            if (ok) {
                const unsigned long long m = __ballot(rowpre >= t && lane < nx);
                ok = m != 0ull;
                i = ok ? (__ffsll((long long)m) - 1) : 0;
            }
            if (ok) {
                const double rowbase = i > 0 ? readlane_f64(rowpre, i - 1) : 0.0;
                // ---- block: lanes 0..7 hold the row's block sums ----------------------------
                const double bv = lane < 8 ? blk[i * 8 + lane] : 0.0;
                const double bp = rowbase + scan8_incl(bv);           // only lanes 0..7 are read
                const unsigned long long mb = __ballot(bp >= t && lane < 8);
                ok = mb != 0ull;
                if (ok) {
                    const int q = __ffsll((long long)mb) - 1;
                    const double blockbase = q > 0 ? readlane_f64(bp, q - 1) : rowbase;
                    KMC_STAMP(1);
                    // ---- entry: 8 sites x 6 directions on 48 lanes -------------------------
                    const int js = q * 8 + ent_site;
                    double rate = 0.0;
                    if (lane < 48 && js < ny) {
                        const uint32_t k = key[i * ny + js];
                        if (k != KMC_VACANT) rate = table[(size_t)k * 6 + ent_dir];
                    }
                    const double ap = blockbase + wave_scan_incl(rate);
                    const unsigned long long me = __ballot(ap >= t && lane < 48);
                    ok = me != 0ull;
                    if (ok) {
                        const int el = __ffsll((long long)me) - 1;
                        const double hi = readlane_f64(ap, el);
                        const double lo = el > 0 ? readlane_f64(ap, el - 1) : blockbase;
                        pnum = readlane_f64(rate, el);
                        pden = gamma;
                        const int es = (el * 43) >> 8;
                        j = q * 8 + es;
                        d = el - 6 * es;
                        ok = (pnum > 0.0) && (hi - t >= M) && (t - lo >= M);
                    }
                }
            }
            if (!ok) {
                // ---- exact path (rare): the reference replayed by the whole wave -----
                wave_fence();
                exact_select(key, p.table, n, rho, xsh);
                wave_fence();
                ++n_exact;
                const long idx = xsh->idx;
                const double g = xsh->gamma;
                if (!(g > 0.0)) { status = 2; break; }
                if (idx >= 6L * n) { status = 1; break; }
                d = (int)(idx % 6);
                j = (int)((idx / 6) % ny);
                i = (int)((idx / 6) / ny);
                pnum = xsh->c_found / g - xsh->cprev_found / g;      // the reference's own expression (:497)
                pden = 1.0;
                if (idx == 0) { flush_tags(); tg_flush_after = true; }     // rho <= 0: see flush_tags
            }
            KMC_STAMP(2);
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
            if (n_i > nx) n_i = nx;
            if (n_j > ny) n_j = ny;
            // lane = 8 * (window row) + (window column)
            const int si = wrap1(i_lo + (grp_a < n_i ? grp_a : 0), nx);      // this group's lattice row
            if (grp_a < n_i && grp_c < n_j) {
                const int sj = wrap1(j_lo + grp_c, ny);
                uint32_t knew = KMC_VACANT;
                if ((col[sj] >> si) & 1ull) knew = key_from_cols([&](int c) { return col[c]; }, nx, ny, si, sj);
                key[si * ny + sj] = knew;
            }
            wave_fence();
            KMC_STAMP(3);
            // ---- tag move (:521-528): registers only, memory is touched at the chunk flush -----
            if (!p.presim) {
                const unsigned long long m = __ballot((tg_sv & 0xFFFF) == s_src);
                const int wrap = ((ie < 0 ? -1 : (ie >= nx ? 1 : 0)) << 8) + ((je < 0 ? -1 : (je >= ny ? 1 : 0)) << 16);
                if (m) {                                 // the atom came with an earlier hop of this chunk
                    const int lm = __ffsll((long long)m) - 1;
                    const int osv = __builtin_amdgcn_readlane(tg_sv, lm), oow = __builtin_amdgcn_readlane(tg_ow, lm);
                    if (lane == lm) { tg_sv = (osv & (int)0xFFFF0000) | s_dst; tg_ow = oow + wrap; }
                } else if (lane == hl) {                 // its tag is still in memory, at s_src
                    tg_sv = s_dst | (s_src << 16); tg_ow = (64 << 8) + (64 << 16) + wrap;
                }
                if (tg_flush_after) { tg_flush_after = false; flush_tags(); }
            }
            // ---- re-sum the touched blocks and rows -------------------------------------
            {
                // a window row's <= 6 columns lie in at most two consecutive blocks of 8 sites
                // (three when the row's last block is partial and the window wraps past it);
                // group a re-sums those blocks of its row, lane c taking site c of each
                const int b0 = wrap1(j_lo, ny) >> 3;                   // block of window column 0
                const bool act = grp_a < n_i;
                const uint32_t* krow = key + si * ny;
                auto site_w = [&](int bg) -> double {
                    const int cj = bg * 8 + grp_c;
                    const uint32_t k = (act && cj < ny) ? krow[cj] : KMC_VACANT;
                    return k == KMC_VACANT ? 0.0 : wsum[k];
                };
                int b1 = b0 + 1; if (b1 >= nb) b1 -= nb;
                int b2 = b1 + 1; if (b2 >= nb) b2 -= nb;
                const double w0 = site_w(b0);
                const double w1 = nslots > 1 ? site_w(b1) : 0.0;
                const double w2 = nslots > 2 ? site_w(b2) : 0.0;
                const double s0 = group8_sum(w0);
                if (act && grp_c == 0) blk[si * 8 + b0] = s0;
                if (nslots > 1) {
                    const double s1 = group8_sum(w1);
                    if (act && grp_c == 1) blk[si * 8 + b1] = s1;
                }
                if (nslots > 2) {
                    const double s2 = group8_sum(w2);
                    if (act && grp_c == 2) blk[si * 8 + b2] = s2;
                }
                wave_fence();
                KMC_STAMP(4);
                int rel = lane - wrap1(i_lo, nx);
                if (rel < 0) rel += nx;
                if (rel < n_i && lane < nx) rsum = row_sum8(blk + lane * 8);
            }
            rowpre = wave_scan_incl(rsum);
            gamma = readlane_f64(rowpre, 63);
            KMC_STAMP(5);
        }
        // ---- flush the chunk: tags, clock, trace, barrier diagnostics ----------------
        const int done = hl;                // hops of this chunk that executed
        flush_tags();
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
    // ---- store the replica ----------------------------------------------------------------
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
#ifdef KMC_STAMPS
        KMC_STAMP(6);
        if (p.stamps)
            for (int k = 0; k < 8; ++k) p.stamps[r * 8 + k] = (long long)st_acc[k];
#endif
    }
}

}  // namespace kmc
