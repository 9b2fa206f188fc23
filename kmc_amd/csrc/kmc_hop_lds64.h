// kmc_hop_lds64.h -- the fused hop loop of kmc_hop_lds.h, specialised for 64 x 64 replicas
// (BASELINE.json's ensemble size): same algorithm, same sums with the same association
// (so gamma, the picks and the certification are bit-identical to the generic kernel), but
// laid out so that one hop is the shortest dependent instruction chain I could find.  The
// kernel is latency bound: one wave alone needs ~4600 cycles per hop in the generic form,
// of which ~600 are memory waits (scripts/stamps.py), so every instruction on the chain
// counts and the number of replicas resident per CU (LDS bytes) sets the throughput.
//
// LDS per replica, 18 496 B (8 replicas per CU, see K64_WAVES_PER_SIMD; the
// generic layout allows 7):
//   klo  u16[4096]   low 16 bits of each site's key at its last refresh
//   khi  u8 [4096]   bits 16..18: bit 18 set = vacant.  A vacant site is key 0x7FFFF, whose rates
//                    are all zero in T, Tc and W, so no path tests for vacancy
//   blk  f64[512]    sums of 8 consecutive sites of a row
//   col  16 B x 68   column bitmaps {lo, hi, lo, -} for columns -2 .. 65 (two halo columns on
//                    each side mirror the periodic wrap).  A read at +0 sees rows 0..63, a
//                    read at +4 sees the bitmap rotated by 32 rows, so the 5 rows around any
//                    site are ONE v_alignbit_b32 per column, wrap included
//   xsh              exact path only (ExactParShared, 1 KB)
//
// Differences from the generic kernel that shorten the chain:
//   * the six rates of a site come from the compact per-direction table Tc[6][2^13] (393 KB, L2
//     resident; kmc_device.h) through a 5-field bit extract with per-lane constants, instead of
//     the 12.6 MB dense table whose gathers mostly miss the 4 MB L2 (template parameter COMPACT;
//     a user table that does not factor keeps the dense gather)
//   * straight-line selection: row, block and entry picks always run, one `ok` at the end
//   * exclusive prefixes come from DPP shifts, the certification is evaluated on all lanes
//     and read as one bit: no conditional read-lanes
//   * window keys: one address per lane, five ds_read2_b32, five v_alignbit; occupancy of
//     the site itself is bit 2 of its own column window
//   * each lane of a window row re-reads the old keys of its two sites before the update and
//     computes the new key of the one window column it owns itself: no key round trip
//     through LDS between the refresh and the re-sum
//   * row sums come from the block sums in registers (tree over 8 lanes + ds_bpermute to the
//     owning lane); the LDS copies of the block sums are written off the critical path
//   * displacement tags move once per 64-hop chunk (one load per atom entering the chunk, two
//     store instructions per chunk) instead of one load + two stores per hop
#pragma once
#include "kmc_device.h"
#include "kmc_hop_lds.h"

namespace kmc {

#define K64_VAC 0x7FFFFu          // key of a vacant site inside this kernel: bit 18 + all 18 neighbours Li.
                                  // Row 0x7FFFF of T and W is zero, and so is the compact table's entry
                                  // (every end point occupied)
#define K64_KLO 0
#define K64_KHI 8192
#define K64_BLK 12288
#define K64_COL 16384
#define K64_XSH (K64_COL + 68 * 16)   // ExactParShared: the exact path's state
#define K64_LDS_BYTES (K64_XSH + 1024)
// Exactly two waves per SIMD (8 replicas per CU), enforced through the register allocation
// (amdgpu_waves_per_eu): the kernel is latency bound, a round of replicas lasts as long as its
// slowest wave, and a SIMD that is given a third wave slows all three.  Measured on 4096 replicas
// (2 rounds of 256 CUs x 8): 6.0 ms per 2048-hop launch against 6.5 ms when LDS alone limits the
// CU to 9 workgroups, and 6.9 ms when LDS limits it to 8 but a SIMD may still take three.
#define K64_WAVES_PER_SIMD 2

static_assert(sizeof(ExactParShared) <= 1024, "exact-path state");

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// the reference's selection replayed exactly by the wave, see exact_select() in kmc_hop_lds.h
__device__ __noinline__ void exact_select64(const unsigned char* smem, const double* table, double rho, ExactParShared* sh)
{
    const uint16_t* klo = reinterpret_cast<const uint16_t*>(smem + K64_KLO);
    const uint8_t* khi = smem + K64_KHI;
    // a vacant site is 0x7FFFF here: xp_row() maps everything above 0x3FFFF to the zero row
    exact_select_par_impl<1, 2, true>([=](long s) -> uint32_t { return (uint32_t)klo[s] | ((uint32_t)khi[s] << 16); },
                                      table, 4096, rho, sh);
}

template <bool COMPACT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K64_WAVES_PER_SIMD, K64_WAVES_PER_SIMD)))
void hop_lds64_kernel(HopParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = p.order ? p.order[blockIdx.x + p.r_base] : (int)blockIdx.x + p.r_base, lane = threadIdx.x;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint16_t* klo = reinterpret_cast<uint16_t*>(smem + K64_KLO);
    uint8_t* khi = smem + K64_KHI;
    double* blk = reinterpret_cast<double*>(smem + K64_BLK);
    uint32_t* colw = reinterpret_cast<uint32_t*>(smem + K64_COL);
    ExactParShared* xsh = reinterpret_cast<ExactParShared*>(smem + K64_XSH);

    uint8_t* occ_g = p.occ + (size_t)r * 4096;
    uint32_t* keys_g = p.keys + (size_t)r * 4096;
    int4* tags_g = p.tags + (size_t)r * 4096;
    const unsigned char* __restrict__ wsum_b = reinterpret_cast<const unsigned char*>(p.wsum);
    const unsigned char* __restrict__ table_b = reinterpret_cast<const unsigned char*>(p.table);
    auto load_w = [&](uint32_t k) -> double {
        asm("" : "+v"(k));                 // one 32-bit value: keeps k << 3 a 32-bit offset (SGPR base + VGPR offset load)
        return *reinterpret_cast<const double*>(wsum_b + (k << 3));
    };
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
        k = k > 0x3FFFFu ? K64_VAC : k;    // KMC_VACANT (0xFFFFFFFF) -> 0x7FFFF
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

    // Displacement tags (:521-528) are moved once per 64-hop chunk, not once per hop: issuing a
    // vector store costs a wave ~120 cycles (scripts/probes/vmem_issue_probe.hip) and the tag values
    // never feed back into the dynamics.  Lane k of the chunk tracks the atom that hop k moved:
    //   tg_site  where that atom is now
    //   tg_ow    lane whose load holds its tag (bits 0-5) and the wrap counts gathered since (8-15, 16-23)
    //   tg_vac   (root lanes) the site it entered the chunk from
    // `live` = lanes whose atom has not been moved again by a later hop of the chunk, `root` =
    // lanes whose atom entered the chunk from memory.  Memory is not touched until the flush at
    // the chunk's end: the roots load their tags (still the pre-chunk state), the live atoms'
    // tags are stored and the roots' home sites that nothing re-occupied are vacated.
    int tg_site = -1, tg_ow = 0, tg_vac = -1;
    unsigned long long tg_live = 0ull, tg_root = 0ull;

    // entry pick: lane = 6 * (site in block) + direction, lanes >= 48 repeat site 0 (masked out)
    const int ent_site = lane < 48 ? (lane * 43) >> 8 : 0;
    const int ent_dir = lane < 48 ? lane - 6 * ent_site : 0;
    const uint32_t ent_off = COMPACT ? (uint32_t)ent_dir * (KMC_COMPACT_ROWS * 8u) : (uint32_t)ent_dir * 8u;
    // compact index of this lane's direction: up to 5 runs of consecutive key bits, packed
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
    const unsigned char* __restrict__ tc_b = reinterpret_cast<const unsigned char*>(p.tcompact);
    // window geometry by mode: i_lo = i - 2 + ext * min(di, 0), n_i = ni0 + ext * |di| (same for j)
    const int wm_ext = p.window_mode == 1 ? 0 : 1;
    const int wm_ni0 = p.window_mode == 0 ? 4 : 5;
    const int wm_nj0 = p.window_mode == 0 ? 4 : (p.window_mode == 1 ? 6 : 5);
    const double mfac = p.mfac;
    const bool force_exact = p.force_exact != 0;
    const bool presim = p.presim != 0;

    // Move the tracked tags to memory (memory holds the tags as they were when tracking started).  Called at
    // every chunk's end, and around a hop with rho <= 0: that hop takes entry 0 whatever its rate,
    // so it may leave a vacant site or land on an occupied one, which the tracking does not model -- it is
    // made a tracking period of its own.
    auto flush_tags = [&]() {
        if (!presim && tg_root) {
            // ---- move the chunk's tags: memory still holds the state before the chunk --------
            int4 tg_val = make_int4(0, 0, 0, 0);
            if ((tg_root >> lane) & 1ull) tg_val = tags_g[tg_vac];
            const bool is_live = (tg_live >> lane) & 1ull;
            const int site_l = is_live ? tg_site : -1;
            // a root's home site is vacant now unless a live atom sits on it
            unsigned long long vac = 0ull;
            for (unsigned long long rm = tg_root; rm; rm &= rm - 1ull) {
                const int rl = __ffsll((long long)rm) - 1;
                const int home = __builtin_amdgcn_readlane(tg_vac, rl);
                if (ballot64(site_l == home) == 0ull) vac |= 1ull << rl;
            }
            const int src = (tg_ow & 63) << 2;
            int4 tv;
            tv.x = __builtin_amdgcn_ds_bpermute(src, tg_val.x);
            tv.y = __builtin_amdgcn_ds_bpermute(src, tg_val.y);
            tv.z = __builtin_amdgcn_ds_bpermute(src, tg_val.z) + (((tg_ow >> 8) & 255) - 64);
            tv.w = __builtin_amdgcn_ds_bpermute(src, tg_val.w) + (((tg_ow >> 16) & 255) - 64);
            if (is_live) tags_g[tg_site] = tv;
            if ((vac >> lane) & 1ull) tags_g[tg_vac] = make_int4(-1, 0, 0, 0);
            tg_live = 0ull; tg_root = 0ull; tg_site = -1;
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
        double gam_v = 1.0, pn_v = 1.0;    // per-lane record of hop (h0 + lane)
        int idx_v = -1;                    // bit 30 set: pn_v is already divided (exact path)

        int hl = 0;
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the draws are in; nothing in the hop loop waits on them
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
            double rate;
            if (COMPACT) {
                uint32_t ci = __builtin_amdgcn_ubfe(ek, cf_st[0], cf_ln[0]);
#pragma unroll
                for (int f = 1; f < 5; ++f) ci |= __builtin_amdgcn_ubfe(ek, cf_st[f], cf_ln[f]) << cf_sh[f];
                rate = *reinterpret_cast<const double*>(tc_b + ((ci << 3) + ent_off));
            } else {
                rate = *reinterpret_cast<const double*>(table_b + (ek * 48u + ent_off));
            }
            const double esc = wave_scan_incl(rate);
            const double ap = blockbase + esc;
            const double ape = blockbase + dpp_f64<0x138, 0xF>(esc);
            const unsigned long long me = ballot64(ap >= t) & 0xFFFFFFFFFFFFull;
            const int el = (__ffsll((long long)me) - 1) & 63;
            const unsigned long long mok = ballot64((ap - t >= M) & (t - ape >= M) & (rate > 0.0));
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
                // ---- exact path (rare): the reference replayed by the whole wave -----
                wave_fence();
                exact_select64(smem, p.table, readlane_f64(rho_v, hl), xsh);
                wave_fence();
                ++n_exact;
                const long idx = xsh->idx;
                const double g = xsh->gamma;
                if (!(g > 0.0)) { status = 2; break; }
                if (idx >= 6L * 4096) { status = 1; break; }
                d = (int)(idx % 6);
                j = (int)((idx / 6) & 63);
                i_sel = (int)((idx / 6) >> 6);
                pnum = xsh->c_found / g - xsh->cprev_found / g;      // the reference's own expression (:497)
                exact_flag = 1 << 30;
                if (idx == 0) { flush_tags(); tg_flush_after = true; }     // rho <= 0: see flush_tags
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
            KMC_STAMP(6);
            // ---- window geometry (:444-453): group a = window row a, its 8 lanes = the 8 sites
            // (c, c + 8 apart) of the two blocks b0, b1 the window's columns fall in
            const int min_di = di < 0 ? di : 0, min_dj = dj < 0 ? dj : 0;
            const int i_lo = ii - 2 + wm_ext * min_di, n_i = wm_ni0 + wm_ext * (di != 0);
            const int j_lo = j - 2 + wm_ext * min_dj, n_j = wm_nj0 + wm_ext * (dj != 0);
            const int si = (i_lo + grp_a) & 63;
            const bool act = grp_a < n_i;
            const int b0 = (j_lo & 63) >> 3, b1 = (b0 + 1) & 7;
            // reads that do not depend on the update go first: the keys as they are now (the
            // window site among them is replaced below) and the row's block sums
            const int sb = si * 64 + grp_c;
            const uint32_t kold0 = load_key(sb + b0 * 8), kold1 = load_key(sb + b1 * 8);
            double bq = blk[si * 8 + grp_c];
            if (lane == j) cb &= ~(1ull << ii);
            if (lane == jw) cb |= (1ull << iw);
            if (lane == j || lane == jw) write_col();
            wave_fence();
            KMC_STAMP(7);
            // ---- windowed refresh (:453-465): this lane's window column is the one = c (mod 8)
            const int wcol = (grp_c - j_lo) & 7;
            const int sj = (j_lo + wcol) & 63;
            const bool inw = act & (wcol < n_j);
            uint32_t knew;
            {
                const int st = (si - 2) & 63;
                // columns sj-2 .. sj+2 are halo columns sj .. sj+4; +1 dword selects the rotated view
                // Five ds_read2_b32 issued back to back and awaited once.  (Left to the compiler the
                // same loads were, in some builds, each followed by its own wait: +300 cycles per hop.)
                const uint32_t ca = lds_base + (uint32_t)(K64_COL + sj * 16 + (st >> 5) * 4);   // LDS byte address
                const uint32_t sh = (uint32_t)st & 31u;
                uint64_t c0, c1, c2, c3, c4;
                asm volatile("ds_read2_b32 %0, %5 offset1:1\n\t"
                             "ds_read2_b32 %1, %5 offset0:4 offset1:5\n\t"
                             "ds_read2_b32 %2, %5 offset0:8 offset1:9\n\t"
                             "ds_read2_b32 %3, %5 offset0:12 offset1:13\n\t"
                             "ds_read2_b32 %4, %5 offset0:16 offset1:17\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)
                             : "v"(ca)
                             : "memory");
                const uint32_t m2 = __builtin_amdgcn_alignbit((uint32_t)(c0 >> 32), (uint32_t)c0, sh);
                const uint32_t m1 = __builtin_amdgcn_alignbit((uint32_t)(c1 >> 32), (uint32_t)c1, sh);
                const uint32_t w0 = __builtin_amdgcn_alignbit((uint32_t)(c2 >> 32), (uint32_t)c2, sh);
                const uint32_t w1 = __builtin_amdgcn_alignbit((uint32_t)(c3 >> 32), (uint32_t)c3, sh);
                const uint32_t w2 = __builtin_amdgcn_alignbit((uint32_t)(c4 >> 32), (uint32_t)c4, sh);
                knew = ((w2 & 7u) << 15) | ((w1 & 15u) << 11) | (((w0 >> 3) & 3u) << 9) | ((w0 & 3u) << 7) |
                       (((m1 >> 1) & 15u) << 3) | ((m2 >> 2) & 7u);
                knew = (w0 & 4u) ? knew : K64_VAC;
            }
            const bool in1 = (sj >> 3) != b0;                    // the window site lies in block b1
            const uint32_t k0 = (inw & !in1) ? knew : kold0;
            const uint32_t k1 = (inw & in1) ? knew : kold1;
            if (inw) {
                const int s = si * 64 + sj;
                klo[s] = (uint16_t)knew;
                khi[s] = (uint8_t)(knew >> 16);
            }
            KMC_STAMP(3);
            // ---- re-sum the touched blocks (8 lanes per block) and rows -------------------
            double w0 = 0.0, w1 = 0.0;
            if (act) { w0 = load_w(k0); w1 = load_w(k1); }
            // ---- book-keeping that nothing in this hop waits for runs while the gathers are in flight
            hist_c += (lane == d);
            if (lane == hl) { gam_v = gamma; pn_v = pnum; idx_v = idx | exact_flag; }
            ++hops;
            // tag move (:521-528): registers only, memory is touched at the chunk flush
            if (!presim) {
                const unsigned long long m = ballot64(tg_site == s_src) & tg_live;
                // tg_ow = origin lane | (wi + 64) << 8 | (wj + 64) << 16; |w| <= 64 inside a chunk
                int ow = hl | (64 << 8) | (64 << 16);
                if (m) {
                    // the atom came with an earlier hop of this chunk: inherit its origin and wraps
                    ow = __builtin_amdgcn_readlane(tg_ow, __ffsll((long long)m) - 1);
                    tg_live &= ~m;
                } else {
                    tg_root |= 1ull << hl;                       // its tag is still in memory, at s_src
                }
                ow += ((ie >> 6) << 8) + ((je >> 6) << 16);      // wrap: floor(ie / 64) = -1, 0, +1
                if (lane == hl) { tg_site = s_dst; tg_ow = ow; tg_vac = s_src; }
                tg_live |= 1ull << hl;
                if (tg_flush_after) { tg_flush_after = false; flush_tags(); }
            }
            const double s0 = group8_sum(w0);
            const double s1 = group8_sum(w1);
            if (act && grp_c == 0) blk[si * 8 + b0] = s0;
            if (act && grp_c == 1) blk[si * 8 + b1] = s1;
            // the row's new sum from its block sums in registers (same tree as row_sum8), then
            // from group a to the lane that owns the row
            bq = grp_c == b0 ? s0 : (grp_c == b1 ? s1 : bq);
            const double rnew = group8_sum(bq);
            KMC_STAMP(4);
            {
                const int rel = (lane - i_lo) & 63;
                const int src = (rel & 7) << 5;                  // byte address of lane 8 * rel
                const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(rnew));
                const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(rnew));
                if (rel < n_i) rsum = __hiloint2double(hi, lo);
            }
            wave_fence();
            rowpre = wave_scan_incl(rsum);
            rowex = dpp_f64<0x138, 0xF>(rowpre);
            gamma = readlane_f64(rowpre, 63);
            KMC_STAMP(5);
        }
        // ---- flush the chunk: clock, trace, barrier diagnostics ----------------------
        const int done = hl;                // hops of this chunk that executed
        hist_v += hist_c; hist_c = 0;
        flush_tags();
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
    // ---- store the replica --------------------------------------------------------------
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
        if (p.stamps)
            for (int k = 0; k < 8; ++k) p.stamps[r * 8 + k] = (long long)st_acc[k];
#endif
    }
}

}  // namespace kmc
