// kmc_hop_lds64.h -- the fused hop loop of kmc_hop_lds.h, specialised for 64 x 64 replicas
// (BASELINE.json's ensemble size): same algorithm, same sums with the same association
// (so gamma, the picks and the certification are bit-identical to the generic kernel), but
// laid out so that one hop is the shortest dependent instruction chain I could find.  The
// kernel is latency bound: one wave alone needs ~4600 cycles per hop in the generic form,
// of which ~600 are memory waits (scripts/stamps.py), so every instruction on the chain
// counts and the number of replicas resident per CU (LDS bytes) sets the throughput.
//
// LDS per replica, 17 760 B -> 9 replicas per CU (the generic layout allows 7):
//   klo  u16[4096]   low 16 bits of each site's key at its last refresh
//   khi  u8 [4096]   bits 16..18: bit 18 set = vacant.  A vacant site is key 2^18, and row
//                    2^18 of T and W is all zeros, so no path tests for vacancy
//   blk  f64[512]    sums of 8 consecutive sites of a row
//   col  16 B x 68   column bitmaps {lo, hi, lo, -} for columns -2 .. 65 (two halo columns on
//                    each side mirror the periodic wrap).  A read at +0 sees rows 0..63, a
//                    read at +4 sees the bitmap rotated by 32 rows, so the 5 rows around any
//                    site are ONE v_alignbit_b32 per column, wrap included
//   frames, pick     exact path only
//
// Differences from the generic kernel that shorten the chain:
//   * straight-line selection: row, block and entry picks always run, one `ok` at the end
//   * exclusive prefixes come from DPP shifts, the certification is evaluated on all lanes
//     and read as one bit: no conditional read-lanes
//   * window keys: one address per lane, five ds_read2_b32, five v_alignbit; occupancy of
//     the site itself is bit 2 of its own column window
//   * the tag load / stores are issued after the W gathers, so nothing on the chain waits
//     for them
#pragma once
#include "kmc_device.h"
#include "kmc_hop_lds.h"

namespace kmc {

#define K64_VAC 0x40000u          // key of a vacant site inside this kernel (table row 2^18 = zeros)
#define K64_KLO 0
#define K64_KHI 8192
#define K64_BLK 12288
#define K64_COL 16384
#define K64_FRAMES (K64_COL + 68 * 16)
#define K64_NFRAMES 8             // np.sum's pairwise recursion over 8192-element buffers is 7 deep
#define K64_PICK (K64_FRAMES + K64_NFRAMES * 32)
#define K64_LDS_BYTES (K64_PICK + 32)

static_assert(sizeof(PwFrame) == 32, "frame size");
static_assert(sizeof(ExactPick) == 32, "pick size");

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// literal restatement of hex_LiCoO2_kmc.py:486-492, see exact_select() in kmc_hop_lds.h
__device__ __noinline__ void exact_select64(const unsigned char* smem, const double* table, double rho,
                                            PwFrame* frames, ExactPick* out)
{
    const uint16_t* klo = reinterpret_cast<const uint16_t*>(smem + K64_KLO);
    const uint8_t* khi = smem + K64_KHI;
    auto entry = [&](long e) -> double {
        const int s = (int)(e / 6);
        const int d = (int)(e - 6 * (long)s);
        const uint32_t k = (uint32_t)klo[s] | ((uint32_t)khi[s] << 16);
        return table[(size_t)k * 6 + d];
    };
    const long n6 = 6L * 4096;
    const double gamma = np_sum_model(n6, entry, frames);
    double c = 0., cprev = 0.;
    long idx = n6;
    if (0.0 / gamma >= rho) {
        idx = 0;
    } else {
        for (int s = 0; s < 4096 && idx == n6; ++s) {
            const uint32_t k = (uint32_t)klo[s] | ((uint32_t)khi[s] << 16);
            if (k == K64_VAC) continue;
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

__global__ __launch_bounds__(64) void hop_lds64_kernel(HopParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = blockIdx.x, lane = threadIdx.x;
    uint16_t* klo = reinterpret_cast<uint16_t*>(smem + K64_KLO);
    uint8_t* khi = smem + K64_KHI;
    double* blk = reinterpret_cast<double*>(smem + K64_BLK);
    uint32_t* colw = reinterpret_cast<uint32_t*>(smem + K64_COL);
    PwFrame* frames = reinterpret_cast<PwFrame*>(smem + K64_FRAMES);
    ExactPick* pick = reinterpret_cast<ExactPick*>(smem + K64_PICK);

    uint8_t* occ_g = p.occ + (size_t)r * 4096;
    uint32_t* keys_g = p.keys + (size_t)r * 4096;
    int4* tags_g = p.tags + (size_t)r * 4096;
    const unsigned char* __restrict__ wsum_b = reinterpret_cast<const unsigned char*>(p.wsum);
    const unsigned char* __restrict__ table_b = reinterpret_cast<const unsigned char*>(p.table);
    auto load_w = [&](uint32_t k) -> double { return *reinterpret_cast<const double*>(wsum_b + (k << 3)); };
    auto load_key = [&](int s) -> uint32_t { return (uint32_t)klo[s] | ((uint32_t)khi[s] << 16); };

    // ---- load the replica ------------------------------------------------
    uint64_t cb = 0;                       // lane j owns column j's bitmap
    for (int i = 0; i < 64; ++i) cb |= (uint64_t)(occ_g[i * 64 + lane] & 1u) << i;
    // column c lives at dword 4 * (c + 2); columns 0, 1 are mirrored at 64, 65 and 62, 63 at -2, -1
    const int col_dw = 4 * (lane + 2);
    const int col_mirror_dw = lane < 2 ? 4 * (lane + 66) : (lane >= 62 ? 4 * (lane - 62) : -1);
    auto write_col = [&]() {
        const uint32_t lo = (uint32_t)cb, hi = (uint32_t)(cb >> 32);
        *reinterpret_cast<uint4*>(colw + col_dw) = make_uint4(lo, hi, lo, 0u);
        if (col_mirror_dw >= 0) *reinterpret_cast<uint4*>(colw + col_mirror_dw) = make_uint4(lo, hi, lo, 0u);
    };
    write_col();
    for (int s = lane; s < 4096; s += KMC_WAVE) {
        uint32_t k = keys_g[s];
        k = k > K64_VAC ? K64_VAC : k;     // KMC_VACANT (0xFFFFFFFF) -> row 2^18
        klo[s] = (uint16_t)k;
        khi[s] = (uint8_t)(k >> 16);
    }
    wave_fence();

    const int grp_a = lane >> 3, grp_c = lane & 7;     // lane = 8 * a + c
    for (int row = 0; row < 64; ++row) {
        const double bs = group8_sum(load_w(load_key(row * 64 + lane)));
        if (grp_c == 0) blk[row * 8 + grp_a] = bs;
    }
    wave_fence();
    double rsum = row_sum8(blk + lane * 8);            // lane i owns the sum of row i
    double rowpre = wave_scan_incl(rsum);
    double rowex = dpp_f64<0x138, 0xF>(rowpre);        // exclusive prefix (wave_shr:1, lane 0 <- 0)
    double gamma = readlane_f64(rowpre, 63);

    int status = 0;
    long long n_exact = 0, hops = 0;
    int hist_c = 0;                        // lane d < 6 counts direction d within the launch chunk
    long long hist_v = 0;
    double time_acc = 0.0;
    double b_last = p.barrier[r * 5 + 0], b_sum = 0.0, b_min = 1e300, b_max = -1e300, b_cnt = 0.0;
    const long long hop0 = p.hops_done[r];

    bool pend = false;                     // deferred tag move (the load of one hop is consumed in the next)
    int pend_src = 0, pend_dst = 0, pend_dwi = 0, pend_dwj = 0;
    int4 pend_tag = make_int4(0, 0, 0, 0);

    // entry pick: lane = 6 * (site in block) + direction, lanes >= 48 repeat site 0 (masked out)
    const int ent_site = lane < 48 ? (lane * 43) >> 8 : 0;
    const int ent_dir = lane < 48 ? lane - 6 * ent_site : 0;
    const uint32_t ent_off = (uint32_t)ent_dir * 8u;
    // window geometry by mode: i_lo = i - 2 + ext * min(di, 0), n_i = ni0 + ext * |di| (same for j)
    const int wm_ext = p.window_mode == 1 ? 0 : 1;
    const int wm_ni0 = p.window_mode == 0 ? 4 : 5;
    const int wm_nj0 = p.window_mode == 0 ? 4 : (p.window_mode == 1 ? 6 : 5);
    const double mfac = p.mfac;
    const bool force_exact = p.force_exact != 0;
    const bool presim = p.presim != 0;

    const long long my_hops = p.n_hops_r ? p.n_hops_r[r] : p.n_hops;
#ifdef KMC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
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
        double gam_v = 1.0, pn_v = 1.0;    // per-lane record of hop (h0 + lane)
        int idx_v = -1;                    // bit 30 set: pn_v is already divided (exact path)

        int hl = 0;
        KMC_STAMP(0);
        for (; hl < cnt; ++hl) {
            const double t = readlane_f64(rho_v, hl) * gamma;
            const double M = gamma * mfac;
            // ---- row ------------------------------------------------------
            const unsigned long long mr = ballot64(rowpre >= t);
            const int i = (__ffsll((long long)mr) - 1) & 63;
            const double rowbase = readlane_f64(rowex, i);
            // ---- block: lanes 0..7 scan the row's block sums ---------------
            const double bsc = scan8_incl(blk[i * 8 + grp_c]);
            const double bp = rowbase + bsc;
            const double bpe = rowbase + dpp_f64<0x111, 0xF>(bsc);
            const unsigned long long mb = ballot64(bp >= t) & 0xFFull;
            const int q = (__ffsll((long long)mb) - 1) & 7;
            const double blockbase = readlane_f64(bpe, q);
            KMC_STAMP(1);
            // ---- entry: 8 sites x 6 directions on 48 lanes ------------------
            const uint32_t ek = load_key(i * 64 + q * 8 + ent_site);
            const double rate = *reinterpret_cast<const double*>(table_b + (ek * 48u + ent_off));
            const double esc = wave_scan_incl(rate);
            const double ap = blockbase + esc;
            const double ape = blockbase + dpp_f64<0x138, 0xF>(esc);
            const unsigned long long me = ballot64(ap >= t) & 0xFFFFFFFFFFFFull;
            const int el = (__ffsll((long long)me) - 1) & 63;
            const unsigned long long mok = ballot64((ap - t >= M) && (t - ape >= M) && (rate > 0.0));
            bool ok = (mb != 0ull) && ((me & mok & (0ull - me)) != 0ull) && !force_exact;   // lowest set bit of me certified
            int j, d, i_sel = i;
            double pnum = readlane_f64(rate, el);
            int exact_flag = 0;
            {
                const int es = (el * 43) >> 8;
                j = q * 8 + es;
                d = el - 6 * es;
            }
            if (!ok) {
                // ---- exact sequential path (rare) --------------------------------
                wave_fence();
                exact_select64(smem, p.table, readlane_f64(rho_v, hl), frames, pick);
                wave_fence();
                ++n_exact;
                const long idx = pick->idx;
                const double g = pick->gamma;
                if (!(g > 0.0)) { status = 2; break; }
                if (idx >= 6L * 4096) { status = 1; break; }
                d = (int)(idx % 6);
                j = (int)((idx / 6) & 63);
                i_sel = (int)((idx / 6) >> 6);
                pnum = pick->c / g - pick->cprev / g;      // the reference's own expression (:497)
                exact_flag = 1 << 30;
            }
            KMC_STAMP(2);
            const int ii = i_sel;
            const int s_src = ii * 64 + j;
            const int idx = s_src * 6 + d;
            // ---- update (:518-535) -------------------------------------------
            const int di = ((2310 >> (2 * d)) & 3) - 1;          // {1, 0, -1, -1, 0, 1}
            const int dj = ((105 >> (2 * d)) & 3) - 1;           // {0, 1, 1, 0, -1, -1}
            const int ie = ii + di, je = j + dj;                 // unwrapped
            const int iw = ie & 63, jw = je & 63;
            const int s_dst = iw * 64 + jw;
            if (lane == j) cb &= ~(1ull << ii);
            if (lane == jw) cb |= (1ull << iw);
            if (lane == j || lane == jw) write_col();
            hist_c += (lane == d);
            if (lane == hl) { gam_v = gamma; pn_v = pnum; idx_v = idx | exact_flag; }
            ++hops;
            wave_fence();
            // ---- windowed refresh (:444-465): lane = 8 * (window row) + (window column) ----
            const int min_di = di < 0 ? di : 0, min_dj = dj < 0 ? dj : 0;
            const int i_lo = ii - 2 + wm_ext * min_di, n_i = wm_ni0 + wm_ext * (di != 0);
            const int j_lo = j - 2 + wm_ext * min_dj, n_j = wm_nj0 + wm_ext * (dj != 0);
            const int si = (i_lo + grp_a) & 63;
            const bool act = grp_a < n_i;
            {
                const int sj = (j_lo + grp_c) & 63;
                const int st = (si - 2) & 63;
                // columns sj-2 .. sj+2 are halo columns sj .. sj+4; +1 dword selects the rotated view
                const uint32_t* cw = colw + (sj * 4 + (st >> 5));
                const uint32_t sh = (uint32_t)st & 31u;
                const uint32_t m2 = __builtin_amdgcn_alignbit(cw[1], cw[0], sh);
                const uint32_t m1 = __builtin_amdgcn_alignbit(cw[5], cw[4], sh);
                const uint32_t w0 = __builtin_amdgcn_alignbit(cw[9], cw[8], sh);
                const uint32_t w1 = __builtin_amdgcn_alignbit(cw[13], cw[12], sh);
                const uint32_t w2 = __builtin_amdgcn_alignbit(cw[17], cw[16], sh);
                uint32_t knew = ((w2 & 7u) << 15) | ((w1 & 15u) << 11) | (((w0 >> 3) & 3u) << 9) | ((w0 & 3u) << 7) |
                                (((m1 >> 1) & 15u) << 3) | ((m2 >> 2) & 7u);
                knew = (w0 & 4u) ? knew : K64_VAC;
                if (act && grp_c < n_j) {
                    const int s = si * 64 + sj;
                    klo[s] = (uint16_t)knew;
                    khi[s] = (uint8_t)(knew >> 16);
                }
            }
            wave_fence();
            KMC_STAMP(3);
            // ---- re-sum the touched blocks (8 lanes per block) ---------------------------
            const int b0 = (j_lo & 63) >> 3, b1 = (b0 + 1) & 7;
            double w0 = 0.0, w1 = 0.0;
            if (act) {
                const int sb = si * 64 + grp_c;
                w0 = load_w(load_key(sb + b0 * 8));
                w1 = load_w(load_key(sb + b1 * 8));
            }
            // ---- tag move, deferred by one hop (:521-528); after the gathers were issued ----
            if (!presim) {
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
                pend_dwi = ie >> 6;                              // -1, 0, +1: floor(ie / 64)
                pend_dwj = je >> 6;
            }
            const double s0 = group8_sum(w0);
            const double s1 = group8_sum(w1);
            if (act && grp_c == 0) blk[si * 8 + b0] = s0;
            if (act && grp_c == 1) blk[si * 8 + b1] = s1;
            wave_fence();
            KMC_STAMP(4);
            if (((lane - i_lo) & 63) < n_i) rsum = row_sum8(blk + lane * 8);
            rowpre = wave_scan_incl(rsum);
            rowex = dpp_f64<0x138, 0xF>(rowpre);
            gamma = readlane_f64(rowpre, 63);
            KMC_STAMP(5);
        }
        // ---- flush the chunk: clock, trace, barrier diagnostics ----------------------
        const int done = hl;                // hops of this chunk that executed
        hist_v += hist_c; hist_c = 0;
        if (!presim) {
            const double dt = lane < done ? (-1.0 / gam_v) * log(u_v) : 0.0;      // :530
            time_acc += wave_sum(dt);
        }
        if (p.sel && lane < cnt) p.sel[(size_t)r * p.n_hops + h0 + lane] = lane < done ? (idx_v & 0x3FFFFFFF) : -1;
        {
            const bool has = lane < done && (idx_v & 0x3FFFFFFF) != 0;           // :495
            const double pd_v = (idx_v & (1 << 30)) ? 1.0 : gam_v;
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
            for (long long hh = h0 + KMC_WAVE + lane; hh < p.n_hops; hh += KMC_WAVE)
                p.sel[(size_t)r * p.n_hops + hh] = -1;
        }
    }
    // ---- finish the pending tag move, store the replica -------------------------------
    if (!presim && pend && lane == 0) {
        int4 tg = pend_tag;
        tg.z += pend_dwi; tg.w += pend_dwj;
        tags_g[pend_dst] = tg;
        tags_g[pend_src] = make_int4(-1, 0, 0, 0);
    }
    wave_fence();
    for (int i = 0; i < 64; ++i) occ_g[i * 64 + lane] = (uint8_t)((cb >> i) & 1ull);
    for (int s = lane; s < 4096; s += KMC_WAVE) {
        const uint32_t k = load_key(s);
        keys_g[s] = k == K64_VAC ? KMC_VACANT : k;
    }
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
