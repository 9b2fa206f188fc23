// kmc_hop_q64.h -- the fused hop loop for 64 x 64 replicas at SIXTEEN replicas per CU.
//
// hop_lds64_kernel (kmc_hop_lds64.h) is bound by the latency of one wave's dependent chain, and its
// 17.8 KB of LDS per replica let a CU carry 8 replicas: the vector ALUs are busy 55 % of the time
// (profiles/r1c_pmc.json) and 4096 replicas per GPU need two rounds.  This kernel runs the same
// algorithm in 10 032 B of LDS and <= 128 VGPRs, so 16 replicas share a CU (4 waves per SIMD) and
// the 4096 replicas of the headline configuration are ONE round.  What had to go:
//   * the block sums (4 KB): lane i keeps the four QUARTER sums of lattice row i (16 sites each) in
//     registers instead.  quarter = tree8 over c of (W[site c] + W[site c + 8]), row = (Q0+Q1)+(Q2+Q3):
//     a fixed association (every stored sum is a pure function of the keys: no drift, split runs are
//     bit-identical), but not the block-sum association of the other LDS kernels: gamma and the clocks
//     agree with theirs to rounding (~1e-16), the selections exactly
//   * the byte plane of key bits 16-18 (4 KB): bits 16, 17 are packed 2 per site (1 KB, updated with
//     ds_mskor_b32), and a vacant site is simply key 0x3FFFF (all neighbours Li: every end point
//     occupied, so its rates are zero in T and W).  Whether a site IS vacant is read from the column
//     bitmaps when the replica is stored; the hop loop never asks
//   * the column bitmaps' padding: {lo, hi, lo} = 12 B per column (816 B with the halo columns)
//   * the exact path's state: in global memory (HopParams::xscratch), it runs once in ~1e8 hops
//   * registers: the clock's draw is regenerated at the chunk flush, flush-only accumulators live in 96 B
//     of LDS, tag registers are packed, loop-invariant addresses are recomputed
// Selection is three levels: row (ballot over the per-lane row prefixes), quarter (three compares
// in the lane that owns the row), entry (16 sites x 6 directions = 2 entries per lane on 48 lanes:
// ONE 16-byte gather per lane from the pair table Tc2 -- or the dense table if the user's table does
// not factor, template parameter COMPACT -- and a wave scan of the pair sums).
// At 4 waves per SIMD the kernel is bound by vector-instruction issue rather than latency: uniform
// decisions are branches or scalar code instead of per-lane selects, one-lane writes are v_writelane,
// and a round trip through LDS is preferred to a handful of vector instructions.
#pragma once
#include "kmc_device.h"
#include "kmc_hop_lds.h"
#include "kmc_hop_lds64.h"
#include "kmc_exact_par.h"

namespace kmc {

#define Q64_VAC 0x3FFFFu
#define Q64_KLO 0
#define Q64_KHP 8192              // u32[256]: row r = dwords 4r .. 4r+3, site column c = bits 2(c&15) of dword c>>4
#define Q64_COL 9216              // 12 B x 68: {lo, hi, lo} for columns -2 .. 65
#define Q64_MISC (Q64_COL + 68 * 12)   // i64 hist[6], f64 time, f64 barrier stats[5]: touched at the chunk flush only,
                                      // kept here rather than in registers that would be spilled around the hop loop
#define Q64_LDS_BYTES (Q64_MISC + 96)
#define Q64_XSCRATCH 1024         // bytes of global scratch per replica (exact path: ExactParShared)
#define Q64_WAVES_PER_SIMD 4

static_assert(Q64_LDS_BYTES <= 10240, "16 replicas per CU need <= 10 KB each");
static_assert(Q64_KHP == 8192, "the ds_mskor_b32 in the hop loop carries this offset as an immediate");

__device__ __forceinline__ uint32_t q64_key(const uint16_t* klo, const uint32_t* khp, int s)
{
    return (uint32_t)klo[s] | (((khp[s >> 4] >> ((s & 15) * 2)) & 3u) << 16);
}

// One lane of a VGPR <- a wave-uniform value (v_writelane_b32; this clang has no builtin for it).  The
// kernel is bound by vector-ALU issue at 4 waves per SIMD, and "if (lane == k) x = uniform" costs a
// move and a select per dword.  M0 carries the lane number; nothing else in this file uses M0.
__device__ __forceinline__ void writelane2(int& a, int& b, int va, int vb, int lane)
{
    asm("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
        : "+v"(a), "+v"(b) : "s"(va), "s"(vb), "s"(lane) : "m0");
}
__device__ __forceinline__ void writelane5(int& a, int& b, int& c, int& d, int& e, int va, int vb, int vc, int vd, int ve,
                                           int lane)
{
    asm("s_mov_b32 m0, %10\n\ts_nop 0\n\tv_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\t"
        "v_writelane_b32 %2, %7, m0\n\tv_writelane_b32 %3, %8, m0\n\tv_writelane_b32 %4, %9, m0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "s"(va), "s"(vb), "s"(vc), "s"(vd), "s"(ve), "s"(lane) : "m0");
}

// wave_scan_incl (kmc_device.h) with the two cross-row steps reading into register pairs whose masked
// rows are zero and stay zero (the DPP move never writes them): saves re-zeroing them, 4 moves a scan.
// z15 / z31 must be zero in rows {0, 2} / {0, 1} on entry; the other rows are scratch.
struct ScanZero { int a_lo, a_hi, b_lo, b_hi; };
__device__ __forceinline__ double wave_scan_incl_z(double x, ScanZero& z)
{
    x += dpp_f64<0x111, 0xF>(x);
    x += dpp_f64<0x112, 0xF>(x);
    x += dpp_f64<0x114, 0xF>(x);
    x += dpp_f64<0x118, 0xF>(x);
    z.a_lo = __builtin_amdgcn_update_dpp(z.a_lo, __double2loint(x), 0x142, 0xA, 0xF, false);
    z.a_hi = __builtin_amdgcn_update_dpp(z.a_hi, __double2hiint(x), 0x142, 0xA, 0xF, false);
    x += __hiloint2double(z.a_hi, z.a_lo);
    z.b_lo = __builtin_amdgcn_update_dpp(z.b_lo, __double2loint(x), 0x143, 0xC, 0xF, false);
    z.b_hi = __builtin_amdgcn_update_dpp(z.b_hi, __double2hiint(x), 0x143, 0xC, 0xF, false);
    x += __hiloint2double(z.b_hi, z.b_lo);
    return x;
}

static_assert(sizeof(ExactParShared) <= Q64_XSCRATCH, "exact-path scratch");

// first set bit of a wave mask, 63 if there is none (s_ff1_i32_b64 returns -1 for zero: no compare and select)
__device__ __forceinline__ int q64_ff1_63(unsigned long long m)
{
    int r;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return r & 63;
}

// The kernel's parameters, re-read from the kernel-argument segment (scalar loads through the constant
// cache) where they are used outside the hop loop.  Read once at the top they occupy ~50 scalar registers
// for the whole kernel, and the hop loop then spills and refills its own scalars through vector lanes
// (v_readlane / v_writelane: vector instructions, which is what the loop is short of).
typedef const __attribute__((address_space(4))) HopParams* q64_params_ptr;
__device__ __forceinline__ q64_params_ptr q64_cold_params()
{
    q64_params_ptr kp = (q64_params_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));           // opaque: the loads through it stay where they are written
    return kp;
}
#define QP (q64_cold_params())

// The reference's selection replayed exactly (hex_LiCoO2_kmc.py:486-492) by the replica's one wave:
// kmc_exact_par.h with the keys read from the packed LDS planes.  ~50 us instead of the ~2.5 ms a
// one-lane replay of the 24 576 entries takes -- which matters although only one hop in ~1e8 comes here:
// a launch of 8.4e6 hops ends when its slowest replica does.  The lean instance (2 sites per lane and
// tile, looped leaves) keeps the function inside the kernel's 128-register budget: a callee's registers
// count towards the kernel's.
// (Built with -mllvm -enable-ipra=0, kmc_amd/build.py: with inter-procedural register allocation the
// kernel sees this function clobber ~100 scalar registers and spills its own all over the hop loop;
// under the plain calling convention the callee saves what it uses.  Inlined it is worse still.)
__device__ __noinline__ void exact_select_q64(const unsigned char* smem, const double* table, double rho, ExactParShared* sh)
{
    const uint16_t* klo = reinterpret_cast<const uint16_t*>(smem + Q64_KLO);
    const uint32_t* khp = reinterpret_cast<const uint32_t*>(smem + Q64_KHP);
    exact_select_par_impl<1, 2, true>([=](long s) -> uint32_t { return q64_key(klo, khp, (int)s); }, table, 4096, rho, sh);
}

// The same for an nx x ny lattice kept in the 64-column layout (ANY, below): the reference's entry order is
// row-major over nx x ny, the planes are indexed by row * 64 + column.
__device__ __noinline__ void exact_select_qany(const unsigned char* smem, const double* table, int ny, int n, double rho,
                                               ExactParShared* sh)
{
    const uint16_t* klo = reinterpret_cast<const uint16_t*>(smem + Q64_KLO);
    const uint32_t* khp = reinterpret_cast<const uint32_t*>(smem + Q64_KHP);
    exact_select_par_impl<1, 2, true>([=](long s) -> uint32_t {
        const int i = (int)s / ny, j = (int)s - i * ny;
        return q64_key(klo, khp, i * 64 + j);
    }, table, n, rho, sh);
}

// ANY = false: 64 x 64, every wrap is a mask, the instantiation the headline number is measured on.
// ANY = true: any 8 <= nx, ny <= 64 in the SAME 10 KB layout (rows and columns beyond the lattice hold the all-Li
// key, whose rates are zero, and are never selected), so that ensembles of small lattices -- the reference's
// D-against-size study, Mc_Lattice.py:243-258 -- also run 16 to a CU with two-entries-per-lane pair gathers.
// What changes: wraps are compare-and-add; the halo columns mirror columns 0, 1 at ny, ny + 1 and ny - 2, ny - 1
// at -2, -1; a column's bitmap carries rows 0..3 again at bits nx .. nx + 3 (what the five-row read at
// (row - 2) mod nx reaches); a window that wraps round a lattice whose width is not a multiple of 16 can touch a
// third quarter; site numbers in memory (trace, tags, keys) are row * ny + column.
template <bool COMPACT, bool ANY = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(Q64_WAVES_PER_SIMD, Q64_WAVES_PER_SIMD)))
void hop_q64_kernel(HopParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = p.order ? p.order[blockIdx.x + p.r_base] : (int)blockIdx.x + p.r_base, lane = threadIdx.x;
    const int nx = ANY ? p.nx : 64, ny = ANY ? p.ny : 64;
    // site number in memory of the site at (row s >> 6, column s & 63) of the 64-column layout
    auto ext = [&](int s) -> int { return ANY ? (s >> 6) * ny + (s & 63) : s; };
#if defined(KMC_STAMPS) && KMC_STAMPS == 2
    const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint16_t* klo = reinterpret_cast<uint16_t*>(smem + Q64_KLO);
    uint32_t* khp = reinterpret_cast<uint32_t*>(smem + Q64_KHP);
    uint32_t* colw = reinterpret_cast<uint32_t*>(smem + Q64_COL);
#define Q64_XSH (reinterpret_cast<ExactParShared*>(reinterpret_cast<unsigned char*>(QP->xscratch) + (size_t)r * Q64_XSCRATCH))

#define Q64_N_SITES (ANY ? (size_t)QP->n : (size_t)4096)
#define Q64_OCC_G (QP->occ + (size_t)r * Q64_N_SITES)
#define Q64_KEYS_G (QP->keys + (size_t)r * Q64_N_SITES)
#define Q64_TAGS_G (QP->tags + (size_t)r * Q64_N_SITES)
    const unsigned char* __restrict__ wsum_b = reinterpret_cast<const unsigned char*>(p.wsum);
    const unsigned char* __restrict__ table_b = reinterpret_cast<const unsigned char*>(p.table);
    auto load_w = [&](uint32_t k) -> double {
        asm("" : "+v"(k));                 // one 32-bit value: keeps k << 3 a 32-bit offset (SGPR base + VGPR offset load)
        return *reinterpret_cast<const double*>(wsum_b + (k << 3));
    };

    // ---- load the replica ------------------------------------------------
    uint64_t cb = 0;                       // lane j owns column j's bitmap
    {
        const uint8_t* occ_g = Q64_OCC_G;
        if (ANY) {
            if (lane < ny)
                for (int i = 0; i < nx; ++i) cb |= (uint64_t)(occ_g[i * ny + lane] & 1u) << i;
            if (nx < 64) cb |= (cb & 15ull) << nx;          // rows 0..3 again above row nx - 1
        } else {
            for (int i = 0; i < 64; ++i) cb |= (uint64_t)(occ_g[i * 64 + lane] & 1u) << i;
        }
    }
    // column c lives at dword 3 * (c + 2); columns 0, 1 are mirrored at ny, ny + 1 and ny - 2, ny - 1 at -2, -1
    const bool col_mirrored = ANY ? (lane < 2 || (lane >= ny - 2 && lane < ny)) : (lane < 2 || lane >= 62);
    auto write_col = [&]() {
        const uint32_t lo = (uint32_t)cb, hi = (uint32_t)(cb >> 32);
        // third dword = bits 64 .. 95 of the periodic bitmap: rows 64 - nx .. (read only when nx + 3 >= 64)
        const uint32_t lo3 = ANY ? lo >> ((64 - nx) & 31) : lo;
        // the addresses are recomputed from the lane number on every call (one multiply-add): hoisted
        // out of the hop loop they are spilled to scratch memory at the 128-register limit
        int ln = lane;
        asm volatile("" : "+v"(ln));
        typedef __attribute__((address_space(3))) uint32_t lds_u32;
        lds_u32* cp = (lds_u32*)(uintptr_t)(lds_base + (uint32_t)Q64_COL + 12u * (uint32_t)(ln + 2));
        cp[0] = lo; cp[1] = hi; cp[2] = lo3;
        if (col_mirrored) {
            // halo columns 66, 67 (lanes 0, 1) and 0, 1 (lanes 62, 63)
            const uint32_t mc = ANY ? (uint32_t)(ln < 2 ? ln + ny + 2 : ln - ny + 2) : (uint32_t)((ln + 66) & 127);
            lds_u32* mp = (lds_u32*)(uintptr_t)(lds_base + (uint32_t)Q64_COL + 12u * mc);
            mp[0] = lo; mp[1] = hi; mp[2] = lo3;
        }
    };
    write_col();
    for (int s = lane; s < 256; s += KMC_WAVE) khp[s] = 0u;
    wave_fence();
    const uint32_t* keys_in = Q64_KEYS_G;
    for (int s = lane; s < 4096; s += KMC_WAVE) {
        uint32_t k;
        if (ANY) k = ((s >> 6) < nx && lane < ny) ? keys_in[(s >> 6) * ny + lane] : Q64_VAC;     // s & 63 == lane
        else k = keys_in[s];
        k = k > 0x3FFFFu ? Q64_VAC : k;    // KMC_VACANT (0xFFFFFFFF) -> all neighbours Li (zero row)
        klo[s] = (uint16_t)k;
        atomicOr(&khp[s >> 4], (k >> 16) << ((s & 15) * 2));
    }
    wave_fence();

    const int grp_a = lane >> 3, grp_c = lane & 7;     // lane = 8 * a + c
    double Q0 = 0.0, Q1 = 0.0, Q2 = 0.0, Q3 = 0.0;     // lane i: the quarter sums of row i
    for (int row0 = 0; row0 < 64; row0 += 8) {
        // quarter = tree8 over c of (W[site c] + W[site c + 8]): lanes 8..15 of each 16-lane row.
        // Eight rows' gathers are issued together (one round trip of latency, not eight).
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = load_w(q64_key(klo, khp, (row0 + k) * 64 + lane));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double bs = group8_sum(x[k] + dpp_f64<0x118, 0xF>(x[k]));      // row_shr:8
            const double q0 = readlane_f64(bs, 8), q1 = readlane_f64(bs, 24), q2 = readlane_f64(bs, 40), q3 = readlane_f64(bs, 56);
            if (lane == row0 + k) { Q0 = q0; Q1 = q1; Q2 = q2; Q3 = q3; }
        }
    }
    ScanZero sz = {0, 0, 0, 0};
    double rowpre = wave_scan_incl_z((Q0 + Q1) + (Q2 + Q3), sz);
    double rowex = dpp_f64<0x138, 0xF>(rowpre);        // exclusive prefix (wave_shr:1, lane 0 <- 0)
    double gamma = readlane_f64(rowpre, 63);

    int status = 0;
    long long n_exact = 0, hops = 0;
    long long* hist_l = reinterpret_cast<long long*>(smem + Q64_MISC);            // [6]
    double* time_l = reinterpret_cast<double*>(smem + Q64_MISC + 48);              // launch clock
    double* bstat_l = reinterpret_cast<double*>(smem + Q64_MISC + 56);             // last, sum, min, max, count
    if (lane < 6) hist_l[lane] = 0;
    if (lane == 0) time_l[0] = 0.0;
    if (lane < 5) bstat_l[lane] = lane == 0 ? QP->barrier[r * 5 + 0] : (lane == 2 ? 1e300 : (lane == 3 ? -1e300 : 0.0));
    const long long hop0 = QP->hops_done[r];

    // displacement tags: see kmc_hop_lds64.h (moved once per 64-hop chunk)
    // here tg_site and tg_vac share one register: site | vac << 16 (registers are the scarce resource at 4 waves per SIMD)
    // Lane = one atom the chunk has moved: tg_sv = where it is now | the site it entered the chunk from << 16
    // (0xFFFF: lane unused), tg_ow = the wraps gathered since, (wi + 64) << 8 | (wj + 64) << 16.  An atom
    // that moves again keeps its lane; a new atom takes lane hl (its hop's number, free by construction).
    // No loop-carried scalar masks: at 4 waves per SIMD the compiler spills those through vector lanes.
    int tg_sv = -1, tg_ow = 0;

    // entry pick: lane = 3 * (site in quarter) + m, entries 2m and 2m+1 of that site; lanes >= 48 masked out
    const int ent_site = lane < 48 ? lane / 3 : 0;
    const int ent_m = lane < 48 ? lane - 3 * ent_site : 0;
    const uint32_t ent_off = COMPACT ? (uint32_t)ent_m * (KMC_COMPACT2_ROWS * 16u) : (uint32_t)ent_m * 16u;
    // compact index of this lane's direction pair: up to 3 runs of consecutive key bits, packed
    uint32_t cf_st[3], cf_ln[3], cf_sh[3];
    {
        uint32_t m = compact2_mask(ent_m), o = 0;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const uint32_t st = m ? (uint32_t)__builtin_ctz(m) : 0u;
            const uint32_t inv = ~(m >> st);
            const uint32_t ln = m ? (inv ? (uint32_t)__builtin_ctz(inv) : 32u) : 0u;
            cf_st[f] = st; cf_ln[f] = ln; cf_sh[f] = o;
            o += ln;
            m = ln >= 32u ? 0u : (m & ~(((1u << ln) - 1u) << st));
        }
    }
    const unsigned char* __restrict__ tc2_b = reinterpret_cast<const unsigned char*>(p.tcompact2);
    const uint32_t ent_sh = (uint32_t)ent_site * 2u;
    // window geometry by mode: i_lo = i - 2 + ext * min(di, 0), n_i = ni0 + ext * |di| (same for j)
    const int wm_ext = p.window_mode == 1 ? 0 : -1;    // all-ones mask: the window extends along the hop
    const int wm_ni0 = p.window_mode == 0 ? 4 : 5;
    const int wm_nj0 = p.window_mode == 0 ? 4 : (p.window_mode == 1 ? 6 : 5);
    // per direction d, in lane d: di, dj, the window's origin relative to the start point and its size, for the context's window
    // mode (i_lo = i - 2 + ext * min(di, 0), n_i = ni0 + ext * |di|, same for j).  Round 3: -25 scalar instructions per hop,
    // -3 % at 4096 replicas -- at four waves per SIMD scalar instructions take issue slots too (profiles/r3_var_geo.txt)
    int geo;
    {
        const int dd = lane < 6 ? lane : 0;
        const int gdi = ((2310 >> (2 * dd)) & 3) - 1, gdj = ((105 >> (2 * dd)) & 3) - 1;
        const int mdi = gdi < 0 ? gdi : 0, mdj = gdj < 0 ? gdj : 0;
        const int ilo = -2 + (wm_ext & mdi), gni = wm_ni0 + (wm_ext & gdi & 1);
        const int jlo = -2 + (wm_ext & mdj), gnj = wm_nj0 + (wm_ext & gdj & 1);
        geo = (gdi & 3) | ((gdj & 3) << 2) | ((ilo & 7) << 4) | ((jlo & 7) << 7) | (gni << 10) | (gnj << 13);
    }
    const double mfac = p.mfac;
    const unsigned long long ok_mask = p.force_exact != 0 ? 0ull : ~0ull;     // kmc_set_debug: every hop through the exact path
    const bool presim = p.presim != 0;

    const long long my_hops = QP->n_hops_r ? QP->n_hops_r[r] : QP->n_hops;
#ifdef KMC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#if KMC_STAMPS == 2
    st_acc[6] = __builtin_amdgcn_s_memrealtime() - ck_rt0;            // the prologue (100 MHz ticks)
#endif
#endif
    // Hops run in chunks of up to 64 (one hop's draws per lane).  A hop that cannot be certified ENDS its
    // chunk early: the chunk is flushed with the hops done so far and the next chunk starts with that hop,
    // whose exact selection is computed first (need_exact).  The exact path is a function call, and a call
    // anywhere inside the hop loop's nest makes the compiler spill scalar registers all over the loop
    // (+75 vector instructions per hop); up here hardly anything is live.
    bool need_exact = false;
    int done = 0;
    const int wave_slot = kmc_wave_slot();
    // Per-replica hop counts (kmc_run_counts: the concentration sweep, every replica its own int(n x) hops a step): the launch
    // lasts as long as its longest replica, so a wave's issue priority is the quarter of the longest run it still has to do
    // -- the long replicas are served first while the short ones, which have the time, wait: BASELINE config 3 (9 x 512 replicas,
    // x = 0.1 ... 0.9) 5.81 -> 5.45 ms per step (profiles/r3k_c3_priority.txt; classes by the replica's whole run instead of what
    // is left of it measure the same)
    const bool ragged = QP->n_hops_r != nullptr;
    const long long run_quarter = (QP->n_hops + 3) >> 2;
    for (long long h0 = 0; h0 < my_hops && status == 0; h0 += done) {
        const int cnt = (int)((my_hops - h0) < KMC_WAVE ? (my_hops - h0) : KMC_WAVE);
        // priority = (slot on the SIMD - chunks done) mod 4: see kmc_rotate_priority (kmc_hop_lds.h)
        if (ragged) {
            const long long rem = my_hops - h0;
            kmc_rotate_priority((rem > run_quarter) + (rem > 2 * run_quarter) + (rem > 3 * run_quarter));
        } else kmc_rotate_priority(wave_slot - (int)(h0 >> 6));
        // ---- uniforms for the next 64 hops, one hop per lane --------------
        // (the clock's draw u is needed only at the flush and is produced again there: two registers
        // fewer across the hop loop, which runs at the 128-register limit of 4 waves per SIMD)
        auto draw_pair = [&](double& rho, double& u) {
            rho = 0.5; u = 1.0;
            if (lane < cnt) {
                if (QP->draws) {
                    const double2 dv = reinterpret_cast<const double2*>(QP->draws)[(size_t)r * QP->n_hops + h0 + lane];
                    rho = dv.x; u = dv.y;
                } else {
                    const unsigned long long hh = (unsigned long long)(hop0 + h0 + lane);
                    u32x4 c; c.x = (uint32_t)hh; c.y = (uint32_t)(hh >> 32); c.z = (uint32_t)r + QP->rid_base; c.w = KMC_STREAM_HOP;
                    const u32x4 o = philox4x32_10(c, QP->seed_lo, QP->seed_hi);
                    rho = u53(o.x, o.y); u = u53_open0(o.z, o.w);
                }
            }
        };
        double rho_v;
        {
            double u_unused;
            draw_pair(rho_v, u_unused);
        }
        // per-lane record of hop (h0 + lane): gamma, chosen rate (as dwords, written with v_writelane), index
        int gam_lo = __double2loint(1.0), gam_hi = __double2hiint(1.0), pn_lo = gam_lo, pn_hi = gam_hi;
        int idx_v = -1;                    // bit 30 set: pn is already divided (exact path)

        int hl = 0;
        int M_lo = 0, M_hi = 0, m_hi_limit = 0;      // margin in scalar registers; 0 limit: the first hop computes it
        bool ex_have = false;                   // hop 0 of the chunk takes the exact answer waiting in xsh
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the draws are in; nothing in the hop loop waits on them
        int run = cnt;                          // hops this chunk may run (0 if the exact path stops the replica)
        if (need_exact) {
            // ---- exact path (rare): the reference's selection replayed literally for hop h0 -----
            need_exact = false;
            wave_fence();
            ExactParShared* xsh = Q64_XSH;
            if (ANY) exact_select_qany(smem, QP->table, ny, QP->n, readlane_f64(rho_v, 0), xsh);
            else exact_select_q64(smem, QP->table, readlane_f64(rho_v, 0), xsh);
            wave_fence();
            ++n_exact;
            if (!(xsh->gamma > 0.0)) { status = 2; run = 0; }
            else if (xsh->idx >= 6L * (long)Q64_N_SITES) { status = 1; run = 0; }
            else {
                ex_have = true;
                // rho <= 0 selects entry 0 whatever its rate (searchsorted left, B4): such a hop may leave a
                // vacant site or land on an occupied one, which the chunk's tag tracking does not model.
                // It runs as a chunk of its own: memory holds the true tags before and after it.
                if (xsh->idx == 0) run = 1;
            }
        }
        KMC_STAMP(0);
        for (; hl < run; ++hl) {
            const double t = readlane_f64(rho_v * gamma, hl);    // one multiply on all lanes: no scalar -> vector moves
            // margin: any M >= gamma * mfac certifies.  It is recomputed only when gamma has grown past the
            // value it was computed from (scalar compare of the high words: gamma > 0), so most hops skip
            // the multiply; it is never more than 2 * gamma * mfac... unless gamma shrank, which only makes
            // the test more conservative (more exact-path hops, never a wrong pick) -- refreshed every chunk
            if (__double2hiint(gamma) >= m_hi_limit) {
                asm volatile("");
                const double m2 = (2.0 * gamma) * mfac;
                M_lo = __builtin_amdgcn_readfirstlane(__double2loint(m2));
                M_hi = __builtin_amdgcn_readfirstlane(__double2hiint(m2));
                m_hi_limit = __double2hiint(gamma) + 0x00100000;   // next binade: gamma < 2 * (gamma at refresh)
            }
            const double M = __hiloint2double(M_hi, M_lo);
            // ---- row ------------------------------------------------------
            const unsigned long long mr = ballot64(rowpre >= t);
            const int i = q64_ff1_63(mr);
            // ---- quarter: every lane decides for its own row, lane i's answer counts ----
            const double p2 = Q0 + Q1;
            const double c1 = rowex + Q0, c2 = rowex + p2, c3 = rowex + (p2 + Q2);
#ifndef Q64_QPICK_SCALAR
            // per-lane answer of every row (selects), lane i's read with three lane reads: no ballots handed to the
            // scalar unit, no branch tree.  A wave's hop is a chain of vector -> scalar -> vector hand-offs and their
            // latencies, not its instruction slots, set its length: this and the entry pick below, both +9 vector
            // instructions, are -1.4 % at 4096 replicas and -4.4 % at 256 (scripts/variants.py; the scalar versions are
            // kept under Q64_QPICK_SCALAR / Q64_EPICK_SCALAR)
            const bool l1 = c1 < t, l2 = c2 < t, l3 = c3 < t;
            const double qb_l = l3 ? c3 : (l2 ? c2 : (l1 ? c1 : rowex));
            const int q_l = (int)l1 + (int)l2 + (int)l3;
            const int q = __builtin_amdgcn_readlane(q_l, i);
            const double qbase = readlane_f64(qb_l, i);
#else
            const unsigned long long b1 = ballot64(c1 < t), b2 = ballot64(c2 < t), b3 = ballot64(c3 < t);
            const int q = (int)((b1 >> i) & 1ull) + (int)((b2 >> i) & 1ull) + (int)((b3 >> i) & 1ull);
            double qbase;                  // q is uniform: a branch and one read-lane instead of six selects
            switch (q) {
            case 0: asm volatile(""); qbase = readlane_f64(rowex, i); break;
            case 1: asm volatile(""); qbase = readlane_f64(c1, i); break;
            case 2: asm volatile(""); qbase = readlane_f64(c2, i); break;
            default: asm volatile(""); qbase = readlane_f64(c3, i); break;
            }
#endif
            KMC_STAMP(1);
            // ---- entry: 16 sites x 6 directions, two entries per lane on 48 lanes ----------
            const uint32_t ek = (uint32_t)klo[i * 64 + q * 16 + ent_site] |
                                (__builtin_amdgcn_ubfe(khp[i * 4 + q], ent_sh, 2u) << 16);
            double2 rr;
            if (COMPACT) {
                uint32_t ci = __builtin_amdgcn_ubfe(ek, cf_st[0], cf_ln[0]);
#pragma unroll
                for (int f = 1; f < 3; ++f) ci |= __builtin_amdgcn_ubfe(ek, cf_st[f], cf_ln[f]) << cf_sh[f];
                rr = *reinterpret_cast<const double2*>(tc2_b + ((ci << 4) + ent_off));
            } else {
                rr = *reinterpret_cast<const double2*>(table_b + (ek * 48u + ent_off));
            }
            const double esc = wave_scan_incl_z(rr.x + rr.y, sz);
            const double ap = qbase + esc;
            const double ape = qbase + dpp_f64<0x138, 0xF>(esc);
            const double a0 = ape + rr.x;
#ifndef Q64_EPICK_SCALAR
            // every lane settles its own candidate (first or second entry), one ballot picks the lane
            const bool first = a0 >= t;
            const double lo_l = first ? ape : a0, hi_l = first ? a0 : ap, rate_l = first ? rr.x : rr.y;
            const bool ok_l = (hi_l - t >= M) & (t - lo_l >= M) & (rate_l > 0.0);
            const unsigned long long me = ballot64(ap >= t) & 0xFFFFFFFFFFFFull;       // a0 >= t implies ap >= t
            const int el = q64_ff1_63(me);
            const unsigned long long mok = ballot64(ok_l);
            const int snd = (int)((ballot64(first) >> el) & 1ull) ^ 1;
            int pnum_lo = __builtin_amdgcn_readlane(__double2loint(rate_l), el);
            int pnum_hi = __builtin_amdgcn_readlane(__double2hiint(rate_l), el);
#else
            const bool first = a0 >= t;
            const unsigned long long mfirst = ballot64(first);
            const unsigned long long me = (mfirst | ballot64(ap >= t)) & 0xFFFFFFFFFFFFull;
            const int el = (__ffsll((long long)me) - 1) & 63;
            // which of lane el's two entries: uniform, so a branch instead of six selects.  Only bit el
            // of the certification and lane el's rate are used
            const int snd = (int)((mfirst >> el) & 1ull) ^ 1;        // 0: lane el's first entry, 1: its second (scalar)
            const bool fst = snd == 0;
            unsigned long long mok;
            int pnum_lo, pnum_hi;          // the chosen rate as two scalar dwords (v_writelane takes scalar sources only)
            if (fst) {
                asm volatile("");
                mok = ballot64(a0 - t >= M) & ballot64(t - ape >= M) & ballot64(rr.x > 0.0);
                pnum_lo = __builtin_amdgcn_readlane(__double2loint(rr.x), el);
                pnum_hi = __builtin_amdgcn_readlane(__double2hiint(rr.x), el);
            } else {
                asm volatile("");
                mok = ballot64(ap - t >= M) & ballot64(t - a0 >= M) & ballot64(rr.y > 0.0);
                pnum_lo = __builtin_amdgcn_readlane(__double2loint(rr.y), el);
                pnum_hi = __builtin_amdgcn_readlane(__double2hiint(rr.y), el);
            }
#endif
            // lowest set bit of me certified?  (scalar integer logic: as a bool the compiler rebuilds it per lane)
            unsigned long long cert = me & mok & (0ull - me) & ok_mask;
            asm("" : "+s"(cert));
            int j, d, i_sel = i;
            int exact_flag = 0;
            {
                const int es = el / 3;
                j = q * 16 + es;
                d = 2 * (el - 3 * es) + snd;
            }
            if (cert == 0ull && !ex_have) { need_exact = true; break; }      // the chunk ends here
            if (ex_have) {
                asm volatile("");
                ex_have = false;
                const ExactParShared* xsh = Q64_XSH;
                const long ex_idx = xsh->idx;
                const double g = xsh->gamma;
                const double pn = xsh->c_found / g - xsh->cprev_found / g;      // the reference's own expression (:497)
                d = (int)(ex_idx % 6);
                if (ANY) {
                    i_sel = (int)(ex_idx / 6) / ny;
                    j = (int)(ex_idx / 6) - i_sel * ny;
                } else {
                    j = (int)((ex_idx / 6) & 63);
                    i_sel = (int)((ex_idx / 6) >> 6);
                }
                pnum_lo = __builtin_amdgcn_readfirstlane(__double2loint(pn));
                pnum_hi = __builtin_amdgcn_readfirstlane(__double2hiint(pn));
                exact_flag = 1 << 30;
            }
            KMC_STAMP(2);
            const int ii = i_sel;
            const int s_src = ii * 64 + j;
            const int idx = (ANY ? ii * ny + j : s_src) * 6 + d;
            // ---- update (:518-535) -------------------------------------------
            // the hop's geometry (:503-504, :444-453), packed per direction in lane d of `geo` (as in kmc_hop_lat.h): one
            // lane read and six scalar bit-field extracts instead of ~25 scalar instructions
            const int gw = __builtin_amdgcn_readlane(geo, d);
            const int di = __builtin_amdgcn_sbfe(gw, 0, 2), dj = __builtin_amdgcn_sbfe(gw, 2, 2);      // {1, 0, -1, -1, 0, 1}, {0, 1, 1, 0, -1, -1}
            const int ie = ii + di, je = j + dj;                 // unwrapped
            const int iw = ANY ? (ie < 0 ? ie + nx : (ie >= nx ? ie - nx : ie)) : (ie & 63);
            const int jw = ANY ? (je < 0 ? je + ny : (je >= ny ? je - ny : je)) : (je & 63);
            const int s_dst = iw * 64 + jw;
            KMC_STAMP(6);
            // ---- window geometry (:444-453): group a = window row a; its lane c holds sites c and
            // c + 8 of the one or two quarters the window's columns fall in
            const int i_lo = ii + __builtin_amdgcn_sbfe(gw, 4, 3), j_lo = j + __builtin_amdgcn_sbfe(gw, 7, 3);
            const int n_i = (int)__builtin_amdgcn_ubfe((uint32_t)gw, 10, 3), n_j = (int)__builtin_amdgcn_ubfe((uint32_t)gw, 13, 3);
            int si;                                              // this group's lattice row
            // (ANY: the window's first row and column are wrapped once, as scalars; a lane's offset then needs one compare)
            const int i_lo0 = ANY ? (i_lo < 0 ? i_lo + nx : i_lo) : 0, j_lo0 = ANY ? (j_lo < 0 ? j_lo + ny : j_lo) : 0;
            if (ANY) { si = i_lo0 + grp_a; si = si >= nx ? si - nx : si; }
            else si = (i_lo + grp_a) & 63;
            const bool act = grp_a < n_i;
            int qa, qb, qc = 0;
            bool three = false;                                  // ANY: a window that wraps past a partial last quarter
            if (ANY) {
                const int ja = j_lo < 0 ? j_lo + ny : j_lo, je_ = j_lo + n_j - 1, jb = je_ >= ny ? je_ - ny : je_;
                qa = ja >> 4; qb = jb >> 4; qc = (ny - 1) >> 4;
                three = ja > jb && qc != qa;                     // then qb = 0 and the quarters are qa < qc and 0
            } else {
                qa = (j_lo & 63) >> 4; qb = ((j_lo + n_j - 1) & 63) >> 4;
            }
            const bool two = qa != qb;                           // uniform
            // two single-lane updates under exec masks (the empty asm keeps them branches: as selects they
            // cost four moves and four selects more)
            // (clear / set, not flips: with rho == 0 the reference "hops" entry 0 whatever its rate, B4, so the
            // source may be vacant already and the destination occupied)
            unsigned long long bit_src = 1ull << ii, bit_dst = 1ull << iw;
            if (ANY) {                                           // rows 0..3 again at bits nx .. nx + 3
                if (ii < 4 && ii + nx < 64) bit_src |= 1ull << (ii + nx);
                if (iw < 4 && iw + nx < 64) bit_dst |= 1ull << (iw + nx);
            }
            if (lane == j) { asm volatile(""); cb &= ~bit_src; }
            if (lane == jw) { asm volatile(""); cb |= bit_dst; }
            if (lane == j || lane == jw) write_col();
            wave_fence();
            KMC_STAMP(7);
            // ---- windowed refresh (:453-465): this lane's window column is the one = c (mod 8)
            const int wcol = ANY ? grp_c : (grp_c - j_lo) & 7;
            int sj;
            if (ANY) { sj = j_lo0 + wcol; sj = sj >= ny ? sj - ny : sj; }
            else sj = (j_lo + wcol) & 63;
            const bool inw = act & (wcol < n_j);
            uint32_t knew;
            {
                const int st = ANY ? (si < 2 ? si - 2 + nx : si - 2) : (si - 2) & 63;
                // columns sj-2 .. sj+2 are halo columns sj .. sj+4; +1 dword selects the rotated view
                const uint32_t ca = lds_base + (uint32_t)(Q64_COL + sj * 12 + (st >> 5) * 4);   // LDS byte address
                const uint32_t sh = (uint32_t)st & 31u;
                uint64_t c0, c1, c2, c3, c4;
                asm volatile("ds_read2_b32 %0, %5 offset1:1\n\t"
                             "ds_read2_b32 %1, %5 offset0:3 offset1:4\n\t"
                             "ds_read2_b32 %2, %5 offset0:6 offset1:7\n\t"
                             "ds_read2_b32 %3, %5 offset0:9 offset1:10\n\t"
                             "ds_read2_b32 %4, %5 offset0:12 offset1:13\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)
                             : "v"(ca)
                             : "memory");
                const uint32_t m2 = __builtin_amdgcn_alignbit((uint32_t)(c0 >> 32), (uint32_t)c0, sh);
                const uint32_t m1 = __builtin_amdgcn_alignbit((uint32_t)(c1 >> 32), (uint32_t)c1, sh);
                const uint32_t w0 = __builtin_amdgcn_alignbit((uint32_t)(c2 >> 32), (uint32_t)c2, sh);
                const uint32_t w1 = __builtin_amdgcn_alignbit((uint32_t)(c3 >> 32), (uint32_t)c3, sh);
                const uint32_t w2 = __builtin_amdgcn_alignbit((uint32_t)(c4 >> 32), (uint32_t)c4, sh);
                knew = __builtin_amdgcn_ubfe(m2, 2u, 3u);
                knew |= __builtin_amdgcn_ubfe(m1, 1u, 4u) << 3;
                knew |= ((w0 & 3u) | ((w0 >> 1) & 12u)) << 7;        // window bits 4, 3, 1, 0 (bit 2 is the site itself)
                knew |= (w1 & 15u) << 11;
                knew |= (w2 & 7u) << 15;
                // vacant site -> 0x3FFFF: sign-extended occupancy bit as the select mask of one v_bfi
                const uint32_t om = (uint32_t)__builtin_amdgcn_sbfe((int)w0, 2u, 1u);
                knew = (knew & om) | (Q64_VAC & ~om);
            }
            if (inw) {
                const int s = si * 64 + sj;
                klo[s] = (uint16_t)knew;
                const uint32_t fsh = (uint32_t)(sj & 15) * 2u;
                const uint32_t haddr = lds_base + (uint32_t)((s >> 4) * 4);
                // MEM = (MEM & ~mask) | value: two key bits of one site inside the quarter's dword
                asm volatile("ds_mskor_b32 %0, %1, %2 offset:8192" :: "v"(haddr), "v"(3u << fsh), "v"((knew >> 16) << fsh) : "memory");
            }
            KMC_STAMP(3);
            // ---- re-sum the touched quarters (8 lanes per half quarter) and rows -------------
            // The lane's sites are read back from LDS after the window's stores (a wave's LDS operations
            // execute in order): a round trip of latency that the other three waves of the SIMD hide,
            // instead of ~10 vector instructions deciding which of the lane's keys the window replaced.
            // Lanes of rows outside the window keep whatever their registers hold: their sums are not used.
            wave_fence();
            double wA0, wA1, wB0, wB1, wC0, wC1;
            asm("" : "=v"(wA0), "=v"(wA1), "=v"(wB0), "=v"(wB1));
            if (ANY) asm("" : "=v"(wC0), "=v"(wC1));
            if (act) {
                const int sA = si * 64 + qa * 16 + grp_c;
                const uint32_t hwA = khp[si * 4 + qa];
                wA0 = load_w((uint32_t)klo[sA] | (__builtin_amdgcn_ubfe(hwA, 2u * grp_c, 2u) << 16));
                wA1 = load_w((uint32_t)klo[sA + 8] | (__builtin_amdgcn_ubfe(hwA, 2u * grp_c + 16u, 2u) << 16));
            }
            if (two) {
                if (act) {
                    const int sB = si * 64 + qb * 16 + grp_c;
                    const uint32_t hwB = khp[si * 4 + qb];
                    wB0 = load_w((uint32_t)klo[sB] | (__builtin_amdgcn_ubfe(hwB, 2u * grp_c, 2u) << 16));
                    wB1 = load_w((uint32_t)klo[sB + 8] | (__builtin_amdgcn_ubfe(hwB, 2u * grp_c + 16u, 2u) << 16));
                }
            }
            if (ANY && three) {
                if (act) {
                    const int sC = si * 64 + qc * 16 + grp_c;
                    const uint32_t hwC = khp[si * 4 + qc];
                    wC0 = load_w((uint32_t)klo[sC] | (__builtin_amdgcn_ubfe(hwC, 2u * grp_c, 2u) << 16));
                    wC1 = load_w((uint32_t)klo[sC + 8] | (__builtin_amdgcn_ubfe(hwC, 2u * grp_c + 16u, 2u) << 16));
                }
            }
            // ---- book-keeping that nothing in this hop waits for runs while the gathers are in flight
            writelane5(gam_lo, gam_hi, pn_lo, pn_hi, idx_v, __double2loint(gamma), __double2hiint(gamma),
                       pnum_lo, pnum_hi, idx | exact_flag, hl);
            ++hops;
            // tag move (:521-528): registers only, memory is touched at the chunk flush
            if (!presim) {
                const unsigned long long m = ballot64((tg_sv & 0xFFFF) == s_src);
                const int wrap = ANY ? ((ie < 0 ? -1 : (ie >= nx ? 1 : 0)) << 8) + ((je < 0 ? -1 : (je >= ny ? 1 : 0)) << 16)
                                     : ((ie >> 6) << 8) + ((je >> 6) << 16);      // floor(ie / 64) = -1, 0, +1
                // One straight line for both cases (a taken branch costs a wave ~27 cycles, a not-taken one ~11:
                // scripts/probes/valu_rate_probe.hip): the atom either came with an earlier hop of this chunk (its
                // lane lm keeps the site it entered the chunk from and gathers the wrap) or its tag is still in
                // memory at s_src (it takes lane hl, free by construction).  Scalar selects, two lane reads, two writes.
                const bool seen = m != 0ull;
                const int lm = seen ? (int)__builtin_ctzll(m) : hl;
                const int osv = __builtin_amdgcn_readlane(tg_sv, lm), oow = __builtin_amdgcn_readlane(tg_ow, lm);
                const int nsv = (seen ? (osv & (int)0xFFFF0000) : (s_src << 16)) | s_dst;
                const int now_ = (seen ? oow : (64 << 8) + (64 << 16)) + wrap;
                writelane2(tg_sv, tg_ow, nsv, now_, lm);
            }
            const double sumA = group8_sum(wA0 + wA1);
            KMC_STAMP(4);
            {
                // from group a to the lane that owns row i_lo + a
                int rel;
                if (ANY) { rel = lane - i_lo0; rel = rel < 0 ? rel + nx : rel; }      // lanes >= nx are masked below
                else rel = (lane - i_lo) & 63;
                const int src = (rel & 7) << 5;                  // byte address of lane 8 * rel
                const bool take = ANY ? (rel < n_i) & (lane < nx) : rel < n_i;
                const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(sumA));
                const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(sumA));
                const double nA = __hiloint2double(hi, lo);
                if (qa == 0) { asm volatile(""); Q0 = take ? nA : Q0; }
                if (qa == 1) { asm volatile(""); Q1 = take ? nA : Q1; }
                if (qa == 2) { asm volatile(""); Q2 = take ? nA : Q2; }
                if (qa == 3) { asm volatile(""); Q3 = take ? nA : Q3; }
                if (two) {
                    const double sumB = group8_sum(wB0 + wB1);
                    const int lob = __builtin_amdgcn_ds_bpermute(src, __double2loint(sumB));
                    const int hib = __builtin_amdgcn_ds_bpermute(src, __double2hiint(sumB));
                    const double nB = __hiloint2double(hib, lob);
                    if (qb == 0) { asm volatile(""); Q0 = take ? nB : Q0; }
                    if (qb == 1) { asm volatile(""); Q1 = take ? nB : Q1; }
                    if (qb == 2) { asm volatile(""); Q2 = take ? nB : Q2; }
                    if (qb == 3) { asm volatile(""); Q3 = take ? nB : Q3; }
                }
                if (ANY && three) {
                    const double sumC = group8_sum(wC0 + wC1);
                    const int loc = __builtin_amdgcn_ds_bpermute(src, __double2loint(sumC));
                    const int hic = __builtin_amdgcn_ds_bpermute(src, __double2hiint(sumC));
                    const double nC = __hiloint2double(hic, loc);
                    if (qc == 1) { asm volatile(""); Q1 = take ? nC : Q1; }     // qa < qc, so qc is 1, 2 or 3
                    if (qc == 2) { asm volatile(""); Q2 = take ? nC : Q2; }
                    if (qc == 3) { asm volatile(""); Q3 = take ? nC : Q3; }
                }
            }
            wave_fence();
            rowpre = wave_scan_incl_z((Q0 + Q1) + (Q2 + Q3), sz);
            rowex = dpp_f64<0x138, 0xF>(rowpre);
            gamma = readlane_f64(rowpre, 63);
            KMC_STAMP(5);
        }
        // ---- flush the chunk: clock, trace, barrier diagnostics ----------------------
        done = hl;                          // hops of this chunk that executed
        {
            // direction histogram (:535) from the recorded indices, six ballots per chunk
            const int dl = lane < done ? (idx_v & 0x3FFFFFFF) % 6 : -1;
            int mine = 0;
#pragma unroll
            for (int dd = 0; dd < 6; ++dd) {
                const int c = __popcll(ballot64(dl == dd));
                mine = lane == dd ? c : mine;
            }
            if (lane < 6) hist_l[lane] += (long long)mine;
        }
        const unsigned long long tg_used = presim ? 0ull : ballot64((tg_sv & 0xFFFF) != 0xFFFF);
        if (tg_used) {
            // ---- move the chunk's tags: memory still holds the state before the chunk --------
            const int tg_site = tg_sv & 0xFFFF, tg_vac = (int)((unsigned)tg_sv >> 16);
            int4* tags_g = Q64_TAGS_G;
            const bool used = (tg_used >> lane) & 1ull;
            int4 tv = make_int4(0, 0, 0, 0);
            if (used) tv = tags_g[ext(tg_vac)];
            // the site an atom left is vacant now unless a moved atom sits on it, i.e. unless it is occupied:
            // one read of the column bitmaps
            const uint32_t hw = colw[3 * ((tg_vac & 63) + 2) + ((tg_vac >> 11) & 1)];
            const bool home_empty = ((hw >> ((tg_vac >> 6) & 31)) & 1u) == 0u;
            tv.z += ((tg_ow >> 8) & 255) - 64;
            tv.w += ((tg_ow >> 16) & 255) - 64;
            if (used) tags_g[ext(tg_site)] = tv;
            if (used && home_empty) tags_g[ext(tg_vac)] = make_int4(-1, 0, 0, 0);
            tg_sv = -1;
        }
        const double gam_v = __hiloint2double(gam_hi, gam_lo), pn_v = __hiloint2double(pn_hi, pn_lo);
        if (!presim) {
            double rho_unused, u_v;
            draw_pair(rho_unused, u_v);
            const double dt = lane < done ? (-1.0 / gam_v) * log(u_v) : 0.0;      // :530
            const double tsum = wave_sum(dt);
            if (lane == 0) time_l[0] += tsum;
        }
        // a stopped replica fills the rest of its trace with -1; otherwise later chunks write the hops not done here
        if (QP->sel && lane < (status != 0 ? cnt : done)) QP->sel[(size_t)r * QP->n_hops + h0 + lane] = lane < done ? (idx_v & 0x3FFFFFFF) : -1;
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
        if (status != 0 && QP->sel) {
            for (long long hh = h0 + KMC_WAVE + lane; hh < QP->n_hops; hh += KMC_WAVE)
                QP->sel[(size_t)r * QP->n_hops + hh] = -1;
        }
    }
    // ---- store the replica --------------------------------------------------------------
    wave_fence();
#if defined(KMC_STAMPS) && KMC_STAMPS == 2
    st_acc[7] = __builtin_amdgcn_s_memrealtime();                     // the write-back starts
#endif
    uint8_t* occ_g = Q64_OCC_G;
    uint32_t* keys_g = Q64_KEYS_G;
    if (ANY) {
        if (lane < ny)
            for (int i = 0; i < nx; ++i) {
                occ_g[i * ny + lane] = (uint8_t)((cb >> i) & 1ull);
                keys_g[i * ny + lane] = ((cb >> i) & 1ull) ? q64_key(klo, khp, i * 64 + lane) : KMC_VACANT;
            }
    } else {
        for (int i = 0; i < 64; ++i) occ_g[i * 64 + lane] = (uint8_t)((cb >> i) & 1ull);
        for (int s = lane; s < 4096; s += KMC_WAVE) {
            // site s = (row s >> 6, column lane): its occupancy is a bit of this lane's own bitmap
            keys_g[s] = ((cb >> (s >> 6)) & 1ull) ? q64_key(klo, khp, s) : KMC_VACANT;
        }
    }
    if (lane < 6) QP->hist[r * 6 + lane] += hist_l[lane];
    const double b_stat = lane < 5 ? bstat_l[lane] : 0.0;
    if (lane == 0) {
        const double time_acc = time_l[0];
        QP->time_total[r] += time_acc;
        QP->step_time[r] = time_acc;
        QP->gamma[r] = gamma;
        QP->hops_done[r] = hop0 + hops;
        QP->n_exact[r] += n_exact;
        QP->status[r] |= status;
#if defined(KMC_STAMPS) && KMC_STAMPS == 2
        st_acc[0] = __builtin_amdgcn_s_memtime() - ck_t0;
        __builtin_amdgcn_s_waitcnt(0);                                 // the stores of the write-back have left
        st_acc[3] = __builtin_amdgcn_s_memrealtime();
        st_acc[7] = st_acc[3] - st_acc[7];
        st_acc[1] = st_acc[3] - ck_rt0;
        st_acc[2] = ck_rt0;
        {
            unsigned hwid;                                   // HW_ID: wave, SIMD, CU, SH, SE, ... (which slot ran it)
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            st_acc[4] = hwid; st_acc[5] = xcc;
        }
#endif
#ifdef KMC_STAMPS
        if (QP->stamps)
            for (int k = 0; k < 8; ++k) QP->stamps[r * 8 + k] = (long long)st_acc[k];
#endif
    }
    if (lane == 0) QP->barrier[r * 5 + 0] = b_stat;
    if (lane == 1 || lane == 4) QP->barrier[r * 5 + lane] += b_stat;
    if (lane == 2) QP->barrier[r * 5 + 2] = fmin(QP->barrier[r * 5 + 2], b_stat);
    if (lane == 3) QP->barrier[r * 5 + 3] = fmax(QP->barrier[r * 5 + 3], b_stat);
}

}  // namespace kmc
