// kmc_hip.hip -- C-ABI library of the MI355X (gfx950) hex-lattice KMC engine.
//
// Hand-written HIP for CDNA4 only: wave64, DPP scans, LDS-resident replicas.
// No CPU backend, no MFMA (the path is gather / scan / latency bound).
// Entry points are declared in include/kmc_hip.h with the reference lines each
// one replaces.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: the library itself is opened at run time (rccl_api)
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/kmc_hip.h"
#include "kmc_device.h"
#include "kmc_hop_lds.h"
#include "kmc_hop_lds64.h"
#include "kmc_hop_q64.h"
#include "kmc_hop_lat.h"
#include "kmc_hop_global.h"
#include "kmc_hop_big.h"

using namespace kmc;

#define KMC_TABLE_ROWS (1u << 18)
#define KMC_EV_RING 64
#define KMC_LAT_ONE_PER_CU 256    // hop_lat_kernel: one replica per CU up to here, two (its PK layout) beyond
#define KMC_LAT_AUTO_MAX 512      // hop_lat_kernel is the default up to two replicas per CU

// ======================================================================
// context
// ======================================================================
#define KMC_GROUPS 2
struct kmc_ctx {
    int device, R, nx, ny, n, window_mode;
    bool lds_path;
    bool lds64;            // 64 x 64 replicas: the specialised kernels of kmc_hop_lds64.h / kmc_hop_q64.h
    bool q64;              // ... the 16-replicas-per-CU kernel (kmc_hop_q64.h) rather than the 8-per-CU one
    bool q64_auto;         // what kmc_set_hop_kernel(ctx, 0) picks
    bool qany_ok, qany;    // 8 <= nx, ny <= 64, not 64 x 64: hop_q64_kernel<., true> (the 16-per-CU layout at any size)
    void* xscratch;        // [R][Q64_XSCRATCH] exact-path scratch of hop_q64_kernel
    // hop_lat_kernel (kmc_hop_lat.h): one replica per CU, everything a hop reads in LDS.  lat_mode: 0 = automatic
    // (taken when the replicas fit one per CU and the table's distinct rates fit the palette), 1 = forced
    // (kmc_set_hop_kernel(ctx, 5)), -1 = never.  pal_state: 0 = palette not built for the current table, 1 = built,
    // -1 = the table holds more than LAT_PAL_MAX distinct rates.
    bool lat_ok, lat_pk; int lat_mode, pal_state, n_pal;
    int which_req;         // what kmc_set_hop_kernel asked for
    bool vac_row_ok;       // the table's all-Li row 0x3FFFF is zero (true of every table that describes hops): see apply_hop_kernel_choice
    LatWP* wp;             // [2^18] {W, palette indices} (lazy)
    double* pal;           // [LAT_PAL_MAX]
    bool big_ok, use_big;  // larger lattices whose bitmaps fit LDS: kmc_hop_big.h (else kmc_hop_global.h)
    BigGeom geom;
    double* hier;          // [R][geom.hier_doubles] sum tree of hop_big_kernel (lazy, zeroed once)
    LatWP* sw;             // [R][n] per-site {W, palette indices} of hop_big_kernel's SITEW instances (lazy)
    bool big_sitew_ok, big_tree;   // SITEW possible for this lattice; kmc_set_hop_kernel(2): the round-2 instances anyway
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    bool timed;
    // A run of an LDS hop kernel over more replicas than 3/4 of one resident round is launched as
    // KMC_GROUPS kernels over disjoint replica ranges, each on a stream of its own.  Consecutive runs are NOT
    // joined: group g's next launch waits for group g's last one only, so the groups drift apart and one's
    // prologue / epilogue / tail overlaps the other's hop loop (+8 % at 4096 replicas, +35 % at 4608).  Every
    // other entry point joins the groups into `stream` first (CTXCHK), the next run forks from it again.
    hipStream_t gstream[KMC_GROUPS];
    hipEvent_t gdone[KMC_GROUPS], fork_ev;
    bool groups_busy, need_fork;
    int timed_slot;                      // run_ev slot if the last timed launch was a run in groups, else -1
    int split;                           // kmc_set_launch_groups: 0 = automatic, 1 = one launch, 2 = groups
    hipEvent_t run_ev[KMC_EV_RING][KMC_GROUPS][2];   // HIP events around the last hop-kernel launches
    unsigned char run_groups[KMC_EV_RING];
    long long run_launches;
    // device state
    uint8_t* occ;
    uint32_t* keys;
    int4* tags;
    double* table;
    double* wsum;
    double* tcompact;      // [6][2^13] per-direction compact table, valid if have_compact
    double* tcompact2;     // [3][2^16][2] pair tables (kmc_device.h), valid if have_compact
    bool have_compact;
    double* time_total;
    double* step_time;
    double* gamma;
    long long* hist;
    long long* hops_done;
    long long* n_exact;
    int* status;
    double* barrier;
    double* scratch_d;     // [R*8] small per-replica scratch
    long long* noalloc_scratch;   // this device's address of kmc_noalloc_scratch (kmc_diffusion_noalloc)
    double* w_g;           // [R][n] per-site rate sums, global-memory hop kernel only (lazy)
    // per-call staging
    double* draws; size_t draws_cap;
    int32_t* sel; size_t sel_cap;
    bool have_table, have_keys;
    int force_exact;
    double margin_scale;
    uint32_t rid_base;     // kmc_set_replica_offset: global id of replica 0 (Philox counter word)
    double* coll_buf; size_t coll_cap;   // device staging of kmc_allreduce_stats (pooled)
    void* pool; size_t pool_cap;         // device scratch of the per-call entry points (grown on demand, never shrunk):
                                         // the drop-in kmc_step / kmc_evolve path calls them once per reference call
    void* hpool; size_t hpool_cap;       // the same on the host
    char err[512];
};

static char g_create_err[512] = "";

static int fail(kmc_ctx* ctx, int code, const char* fmt, ...)
{
    char* dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(ctx, KMC_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// the launch groups of the last runs -> the context's stream (device-side: nothing blocks the host)
static hipError_t join_groups(kmc_ctx* c)
{
    c->need_fork = true;
    if (!c->groups_busy) return hipSuccess;
    for (int g = 0; g < KMC_GROUPS; ++g) {
        hipError_t e = hipEventRecord(c->gdone[g], c->gstream[g]);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->gdone[g], 0);
        if (e != hipSuccess) return e;
    }
    c->groups_busy = false;
    return hipSuccess;
}

#define CTXCHK_RUN(ctx)                                                \
    do {                                                               \
        if (!(ctx)) return fail(nullptr, KMC_EINVAL, "null context"); \
        HIPCHK(ctx, hipSetDevice((ctx)->device));                      \
    } while (0)

#define CTXCHK(ctx)                                                    \
    do {                                                               \
        CTXCHK_RUN(ctx);                                               \
        HIPCHK(ctx, join_groups(ctx));                                 \
    } while (0)

static void time_begin(kmc_ctx* c) { hipEventRecord(c->ev0, c->stream); c->timed_slot = -1; }
static void time_end(kmc_ctx* c) { hipEventRecord(c->ev1, c->stream); c->timed = true; }

// ======================================================================
// kernels: tables
// ======================================================================
// Row T.  One thread per 18-bit key; base table staged in LDS (4 KB).
// Writes 48 B of rates + 8 B row sum per thread: 14.7 MB, write-bound.
__global__ __launch_bounds__(256) void fold_table_kernel(const double* __restrict__ base512, double dumbbell,
                                                          double* __restrict__ table, double* __restrict__ wsum)
{
    __shared__ double sb[512];
    for (int i = threadIdx.x; i < 512; i += 256) sb[i] = base512[i];
    __syncthreads();
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    double r[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) r[e] = fold_rate(row, e, sb, dumbbell);
    double2* out = reinterpret_cast<double2*>(table + (size_t)row * 6);
    out[0] = make_double2(r[0], r[1]);
    out[1] = make_double2(r[2], r[3]);
    out[2] = make_double2(r[4], r[5]);
    wsum[row] = row_sum6(r);
}

__global__ __launch_bounds__(256) void wsum_kernel(const double* __restrict__ table, double* __restrict__ wsum)
{
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    double r[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) r[e] = table[(size_t)row * 6 + e];
    wsum[row] = row_sum6(r);
}

// Compact table Tc[e][idx13] = T[pdep(idx13, mask_e)][e] and its verification against every entry
// of the dense table (a user-supplied dense table need not factor; then the dense gather is kept).
__global__ __launch_bounds__(256) void compact_build_kernel(const double* __restrict__ table, double* __restrict__ tc)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;          // 6 * 8192 threads
    const int e = (int)(t >> KMC_COMPACT_BITS);
    const uint32_t idx = t & (KMC_COMPACT_ROWS - 1);
    tc[t] = table[(size_t)pdep18(idx, compact_mask(e)) * 6 + e];
}

__global__ __launch_bounds__(256) void compact2_build_kernel(const double* __restrict__ table, double* __restrict__ tc2)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;            // [3][2^15]
    const int m = (int)(t >> KMC_COMPACT2_BITS);
    const uint32_t idx = t & (KMC_COMPACT2_ROWS - 1);
    const size_t row = (size_t)pdep18(idx, compact2_mask(m)) * 6;
    tc2[2 * (size_t)t] = table[row + 2 * m];
    tc2[2 * (size_t)t + 1] = table[row + 2 * m + 1];
}

__global__ __launch_bounds__(256) void compact_verify_kernel(const double* __restrict__ table, const double* __restrict__ tc,
                                                             int* __restrict__ mismatches)
{
    const uint32_t key = blockIdx.x * 256 + threadIdx.x;
    int bad = 0;
    for (int e = 0; e < 6; ++e) {
        const double a = table[(size_t)key * 6 + e], b = tc[((size_t)e << KMC_COMPACT_BITS) + pext18(key, compact_mask(e))];
        bad += __double_as_longlong(a) != __double_as_longlong(b);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Row G.  hex_LiCoO2_kmc.py:103-199 + :402-403, one thread per base key.
// Site q (1..9) of base index t: bit (9 - q) ... the nine free chars MSB first feed
// key positions 0,1,4,6,8,11,13,14,15 and the cluster sites read positions
// 0,4,8,1,6,11,13,14,15 (:131) -> free-slot order 0,2,4,1,3,5,6,7,8.
__device__ __constant__ const int8_t kLceSlot[10] = {-1, 0, 2, 4, 1, 3, 5, 6, 7, 8};
__device__ __constant__ const int8_t kLceClusters[45][5] = {
    {1,0,0,0,0},{3,0,0,0,0},{4,0,0,0,0},{7,0,0,0,0},{5,0,0,0,0},{8,0,0,0,0},{6,0,0,0,0},{9,0,0,0,0},{2,0,0,0,0},
    {6,9,0,0,0},{4,1,0,0,0},{3,7,0,0,0},{4,5,6,0,0},{7,8,9,0,0},{1,2,3,0,0},{5,2,8,0,0},
    {5,9,0,0,0},{6,8,0,0,0},{5,1,0,0,0},{3,8,0,0,0},{2,4,0,0,0},{2,7,0,0,0},{5,6,0,0,0},{8,9,0,0,0},{5,4,0,0,0},
    {7,8,0,0,0},{1,2,0,0,0},{2,3,0,0,0},{2,1,4,0,0},{2,3,7,0,0},{5,4,1,0,0},{8,7,3,0,0},
    {5,6,9,0,0},{8,9,6,0,0},{1,3,0,0,0},{4,6,0,0,0},{7,9,0,0,0},{1,4,5,6,0},{3,7,8,9,0},
    {5,1,2,3,6},{2,8,0,0,0},{5,8,0,0,0},{5,4,1,2,0},{5,6,9,8,0},{2,3,7,8,0}};

__global__ __launch_bounds__(512) void lce_rates_kernel(const double* __restrict__ eci, double prefactor, double kb_t,
                                                         double* __restrict__ base512, double* __restrict__ energies)
{
    const int t = threadIdx.x;
    double s[10];
    s[0] = 0.0;
#pragma unroll
    for (int q = 1; q < 10; ++q) s[q] = ((t >> (8 - kLceSlot[q])) & 1) ? -1.0 : 1.0;   // Li -> -1, Vac -> +1 (:130-132)
    double tot = 0.0;
#pragma unroll
    for (int q = 1; q < 10; ++q) tot = tot + s[q];
    double e = 0.0;
    e = e + eci[0];
    e = e + eci[1] * tot;
    for (int c = 0; c < 45; ++c) {
        double v = s[kLceClusters[c][0]];
        for (int q = 1; q < 5 && kLceClusters[c][q]; ++q) v = v * s[kLceClusters[c][q]];
        e = e + eci[c + 2] * v;
    }
    base512[t] = prefactor * exp(-(e) / kb_t);
    if (energies) energies[t] = e;
}

// ======================================================================
// kernels: full rate fill (rows R1, R2)
// ======================================================================
// rates_full_kernel: the first version, kept for lattices under 5 rows or columns (the register kernels below need a 5-row
// window that does not wrap onto itself).  One block = a TI x TJ tile of one replica.  The (TI+4) x (TJ+4) occupancy
// halo (periodic wrap, true modulo) is staged into LDS with coalesced byte-row reads, then each thread builds the 18-bit
// keys of its sites from LDS.
#define RF_TI 16
#define RF_TJ 64
__global__ __launch_bounds__(256) void rates_full_kernel(const uint8_t* __restrict__ occ, int nx, int ny, int tiles_i,
                                                          int tiles_j, uint32_t* __restrict__ keys)
{
    __shared__ uint8_t tile[(RF_TI + 4) * (RF_TJ + 4)];
    const int n = nx * ny;
    int b = blockIdx.x;
    const int tj = b % tiles_j; b /= tiles_j;
    const int ti = b % tiles_i; b /= tiles_i;
    const int r = b;
    const uint8_t* o = occ + (size_t)r * n;
    const int i0 = ti * RF_TI, j0 = tj * RF_TJ;
    for (int e = threadIdx.x; e < (RF_TI + 4) * (RF_TJ + 4); e += 256) {
        const int a = e / (RF_TJ + 4), c = e - a * (RF_TJ + 4);
        int gi = (i0 + a - 2) % nx; if (gi < 0) gi += nx;
        int gj = (j0 + c - 2) % ny; if (gj < 0) gj += ny;
        tile[e] = o[gi * ny + gj] & 1u;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RF_TI * RF_TJ; e += 256) {
        const int a = e / RF_TJ, c = e - a * RF_TJ;
        const int gi = i0 + a, gj = j0 + c;
        if (gi >= nx || gj >= ny) continue;
        const size_t s = (size_t)r * n + (size_t)gi * ny + gj;
        uint32_t key = KMC_VACANT;
        if (tile[(a + 2) * (RF_TJ + 4) + (c + 2)]) {
            key = 0;
#pragma unroll
            for (int p = 0; p < 18; ++p)
                key = (key << 1) | tile[(a + 2 + kNbI[p]) * (RF_TJ + 4) + (c + 2 + kNbJ[p])];
        }
        keys[s] = key;
    }
}


// Rows R1, R2 for lattices up to 64 x 64 (every LDS-path lattice; the headline's 4096 x 64 x 64 = 16.8 M sites):
// one wave per replica, lane j = lattice column j.  The kernel is HBM-bound by design -- 1 B read + 4 B written per
// site -- so the work per byte is kept low: each lane packs its column into a 64-bit bitmap straight from coalesced
// byte loads (a wave-instruction reads one 64-byte lattice row), fetches the four neighbouring columns' bitmaps once
// (ds_bpermute: no LDS memory, no barrier) and then every key is five 5-row windows of those registers: a funnel
// shift and a bit-field extract per column (the key's chars run column by column, kmc_device.h key_from_cols).
// Stores are one dword per lane = 256 contiguous bytes per wave-instruction.  NX = 64: the row loop is unrolled and
// every shift is an immediate; NX = 0: any 5 <= nx, ny <= 64 (run-time rotates).
// Measured (scripts/time_rates_full.py, profiles/r2_rates_full.txt): 4096 x 64 x 64 in 18.0 us = 4.7 TB/s, 16384
// replicas at 5.7 TB/s (0.58 / 0.71 of the 8 TB/s peak; the stream is 80 % writes); two waves per replica (32 rows
// each) were 7 % slower.  rates_full_kernel below (byte tiles in LDS, 18 LDS byte reads per key; any lattice size)
// needs 49 us for the same, 0.21 of the peak.
template <int NX>
__global__ __launch_bounds__(64) void rates_cols_kernel(const uint8_t* __restrict__ occ, int R, int nx_rt, int ny,
                                                         uint32_t* __restrict__ keys)
{
    const int nx = NX ? NX : nx_rt;
    // lattices of up to 32 (16) columns: two (four) replicas per wave, each in its own group of lanes
    const int gw = NX ? 64 : (ny <= 16 ? 16 : (ny <= 32 ? 32 : 64));
    const int lane_w = threadIdx.x, lane = lane_w & (gw - 1), lane0 = lane_w - lane;
    const int r = (int)blockIdx.x * (64 / gw) + lane_w / gw, n = nx * ny;
    const bool live = lane < ny && r < R;
    const uint8_t* o = occ + (live ? (size_t)r * n + lane : 0);
    uint64_t c0 = 0;
    if (NX) {
#pragma unroll
        for (int i0 = 0; i0 < NX; i0 += 16) {
            uint32_t b[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = o[(i0 + k) * ny];
#pragma unroll
            for (int k = 0; k < 16; ++k) c0 |= (uint64_t)(b[k] & 1u) << (i0 + k);
        }
    } else {
        // every row's load in flight at once, issued in uniform groups of 16 (rows beyond nx are not read): as one loop over
        // the rows the loads ran one after the other, and the fill of 4096 x 30x30 took as long as that of 4096 x 64x64
        uint32_t b[64];
#pragma unroll
        for (int i0 = 0; i0 < 64; i0 += 16) {
            if (i0 < nx) {                                   // uniform
#pragma unroll
                for (int k = 0; k < 16; ++k) b[i0 + k] = o[(i0 + k < nx ? i0 + k : nx - 1) * ny];
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) b[i0 + k] = 0u;
            }
        }
        uint32_t half[2] = {0u, 0u};
#pragma unroll
        for (int k = 0; k < 64; ++k) half[k >> 5] |= (b[k] & 1u) << (k & 31);
        c0 = ((uint64_t)half[1] << 32) | half[0];
        if (nx < 64) c0 &= (1ull << nx) - 1ull;
    }
    if (!live) c0 = 0;
    // neighbouring columns j +- 1, j +- 2 (periodic in ny)
    auto from = [&](int dj) -> uint64_t {
        int src = lane + dj;
        src = lane0 + (src < 0 ? src + ny : (src >= ny ? src - ny : src));
        const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)c0);
        const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)(c0 >> 32));
        return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    };
    uint64_t p2 = 0, p1 = 0, m1 = 0, m2 = 0;
    if (NX) { p2 = from(2); p1 = from(1); m1 = from(-1); m2 = from(-2); }
    uint32_t* out = keys + (live ? (size_t)r * n + lane : 0);
    // Window of column c at row i: bits (i-2 .. i+2) mod nx.  The key has six fields, each a run of window bits of one column
    // (the key's chars run column by column, kmc_device.h key_from_cols):
    //   column j+2 bits 0-2 -> key 15-17 | j+1 bits 0-3 -> 11-14 | j bits 3-4 -> 9-10 | j bits 0-1 -> 7-8 | j-1 bits 1-4 -> 3-6 |
    //   j-2 bits 2-4 -> 0-2;  the site itself is bit 2 of column j's window.
    // Every field gets a copy of its column rotated so that bit i + P of the copy is row i - 2, P = the field's place in the key
    // minus its first window bit: one v_alignbit by the row number then drops the field IN PLACE, and a key is 6 alignbits, 6
    // and / and-or, and 2 instructions for the vacancy (round 3: 15 vector instructions per key instead of 22; the kernel is
    // bound by their issue, not by HBM: profiles/r3_stream_floor.txt has what a plain copy of the same bytes reaches).
    constexpr int FP[6] = {15, 11, 6, 7, 2, -2};
    constexpr uint32_t FM[6] = {0x38000u, 0x7800u, 0x600u, 0x180u, 0x78u, 0x7u};
    auto rotl = [&](uint64_t c, int k) -> uint64_t {        // rotate left by k within nx bits, 0 < k < nx
        const uint64_t m = nx == 64 ? ~0ull : ((1ull << nx) - 1ull);
        return ((c << k) | (c >> (nx - k))) & m;
    };
    auto emit = [&](int i, const uint32_t x[6]) {
        uint32_t key = x[0] & FM[0];
#pragma unroll
        for (int f = 1; f < 6; ++f) key |= x[f] & FM[f];
        // vacant site -> KMC_VACANT (all ones): the site's bit sits at bit 9 of field 3's word
        key |= ~(uint32_t)__builtin_amdgcn_sbfe((int)x[3], 9u, 1u);
        // (streaming stores: the fill writes 4 B per site that nothing reads before the whole array is written; without
        // the hint the stream ran at 0.59 of the HBM peak, with it at 0.69, 0.82 on 16 384 replicas)
        if (live) __builtin_nontemporal_store(key, &out[i * ny]);
    };
    if (NX == 64) {
        const uint64_t col[6] = {p2, p1, c0, c0, m1, m2};
        uint32_t lo[6], hi[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const uint64_t a = FP[f] + 2 ? rotl(col[f], FP[f] + 2) : col[f];
            lo[f] = (uint32_t)a; hi[f] = (uint32_t)(a >> 32);
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            uint32_t x[6];
#pragma unroll
            for (int f = 0; f < 6; ++f)       // rows >= 32 take the window from the copy rotated by 32 more: the dwords swapped
                x[f] = i < 32 ? __builtin_amdgcn_alignbit(hi[f], lo[f], (uint32_t)i) : __builtin_amdgcn_alignbit(lo[f], hi[f], (uint32_t)(i - 32));
            emit(i, x);
        }
    } else {
        // any nx <= 64: the column rotated left by 2 is continued periodically over 96 bits (three dwords), so that the window
        // of row i is bits i .. i + 4 whatever nx is.  The lane continues its own column and fetches the neighbours' three
        // dwords (the 64-bit shifts of the continuation are slow instructions); the six placed copies are 96-bit shifts of those.
        // (Round 3: two 64-bit shifts per column and row stood here, and the rows' loads ran one after the other:
        // 4096 x 30x30 took 13.3 us, as long as 4096 x 64x64.)
        uint32_t e[5][3];
        {
            const uint64_t a = rotl(c0, 2);
            uint64_t lo = 0, hi = 0;                         // bits 0 .. 63 and 64 .. 127 of the periodic continuation
            for (int sh = 0; sh < 96; sh += nx) {
                if (sh < 64) { lo |= a << sh; if (sh) hi |= a >> (64 - sh); }
                else hi |= a << (sh - 64);
            }
            e[2][0] = (uint32_t)lo; e[2][1] = (uint32_t)(lo >> 32); e[2][2] = (uint32_t)hi;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (k == 2) continue;
            int src = lane + (2 - k);                        // k = 0 .. 4: columns j + 2, j + 1, j, j - 1, j - 2
            src = lane0 + (src < 0 ? src + ny : (src >= ny ? src - ny : src));
#pragma unroll
            for (int w = 0; w < 3; ++w) e[k][w] = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)e[2][w]);
        }
        constexpr int FC[6] = {0, 1, 2, 2, 3, 4};           // the field's column in e[]
        uint32_t g[6][3];
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const uint32_t* c = e[FC[f]];
            if (FP[f] > 0) {
                g[f][0] = c[0] << FP[f];
                g[f][1] = __builtin_amdgcn_alignbit(c[1], c[0], 32u - (uint32_t)FP[f]);
                g[f][2] = __builtin_amdgcn_alignbit(c[2], c[1], 32u - (uint32_t)FP[f]);
            } else {
                g[f][0] = __builtin_amdgcn_alignbit(c[1], c[0], (uint32_t)-FP[f]);
                g[f][1] = __builtin_amdgcn_alignbit(c[2], c[1], (uint32_t)-FP[f]);
                g[f][2] = c[2] >> -FP[f];
            }
        }
        const int n_lo = nx < 32 ? nx : 32;
        for (int i = 0; i < n_lo; ++i) {
            uint32_t x[6];
#pragma unroll
            for (int f = 0; f < 6; ++f) x[f] = __builtin_amdgcn_alignbit(g[f][1], g[f][0], (uint32_t)i);
            emit(i, x);
        }
        for (int i = 32; i < nx; ++i) {
            uint32_t x[6];
#pragma unroll
            for (int f = 0; f < 6; ++f) x[f] = __builtin_amdgcn_alignbit(g[f][2], g[f][1], (uint32_t)(i - 32));
            emit(i, x);
        }
    }
}

// Rows R1, R2 for lattices beyond 64 x 64 (config 4's 1024 x 1024, the 256 x 256 ensembles): one wave = a tile of TI rows x 64
// columns of one replica, in registers like rates_cols_kernel.  Lane = lattice column j0 + lane: bit k of its bitmap is row
// i0 - 2 + k (TI + 4 <= 32 rows: one dword), built from coalesced byte loads (a wave-instruction reads 64 contiguous bytes of
// a lattice row); lanes 0 .. 3 also build the four halo columns j0 - 2, j0 - 1, j0 + 64, j0 + 65.  A key is then six bit
// fields at immediate positions of five registers, and a wave-instruction stores 256 contiguous bytes.  TI is picked by the
// host so that the launch has enough waves for the chip (a 1024 x 1024 lattice is 1376 tiles of 12 rows).
// (Round 3: rates_full_kernel -- byte tiles in LDS, 18 one-byte LDS reads per key -- took 18.6 us for 64 x 256 x 256 and 8.6 us
// for one 1024 x 1024 where a plain copy of the same bytes takes 6.7 / 6.4, 5.9 of which is the empty launch:
// profiles/r3_stream_floor.txt.)
template <int TI>
__global__ __launch_bounds__(64) void rates_tile_kernel(const uint8_t* __restrict__ occ, int nx, int ny, int tiles_i, int tiles_j,
                                                         uint32_t* __restrict__ keys)
{
    static_assert(TI + 4 <= 32, "a tile's rows and its halo are one dword per column");
    int b = blockIdx.x;
    const int tj = b % tiles_j; b /= tiles_j;
    const int ti = b % tiles_i;
    const int r = b / tiles_i;
    const int lane = threadIdx.x, i0 = ti * TI, j0 = tj * 64;
    const size_t n = (size_t)nx * ny;
    const uint8_t* o = occ + (size_t)r * n;
    const int gj = j0 + lane;
    const int gjc = gj < ny ? gj : gj % ny;                 // lanes beyond the lattice hold the wrapped columns (ny < 64 included)
    int hj = lane < 2 ? j0 - 2 + lane : j0 + 62 + lane;     // lanes 0 .. 3: columns j0 - 2, j0 - 1, j0 + 64, j0 + 65
    hj = ((hj % ny) + ny) % ny;
    const bool halo = lane < 4;
    uint32_t c = 0, h = 0;
    {
        int gi = i0 - 2;
        if (gi < 0) gi += nx;                               // i0 - 2 >= -2 and nx >= 5
        uint32_t bc[TI + 4], bh[TI + 4];
#pragma unroll
        for (int k = 0; k < TI + 4; ++k) {
            const uint8_t* rowp = o + (size_t)gi * ny;      // uniform
            bc[k] = rowp[gjc];
            bh[k] = halo ? rowp[hj] : 0u;
            ++gi;
            if (gi == nx) gi = 0;
        }
#pragma unroll
        for (int k = 0; k < TI + 4; ++k) { c |= (bc[k] & 1u) << k; h |= (bh[k] & 1u) << k; }
    }
    const uint32_t h0 = __builtin_amdgcn_readlane(h, 0), h1 = __builtin_amdgcn_readlane(h, 1),
                   h2 = __builtin_amdgcn_readlane(h, 2), h3 = __builtin_amdgcn_readlane(h, 3);
    auto from = [&](int dj) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + dj) & 63) << 2, (int)c); };
    uint32_t m1 = from(-1), m2 = from(-2), p1 = from(1), p2 = from(2);
    m1 = lane == 0 ? h1 : m1;
    m2 = lane == 0 ? h0 : (lane == 1 ? h1 : m2);
    p1 = lane == 63 ? h2 : p1;
    p2 = lane == 62 ? h2 : (lane == 63 ? h3 : p2);
    const bool live = gj < ny;
    uint32_t* out = keys + (size_t)r * n + (size_t)i0 * ny + (live ? gj : 0);
#pragma unroll
    for (int a = 0; a < TI; ++a) {
        if (i0 + a < nx) {                                  // uniform
            uint32_t key = (((p2 >> a) & 7u) << 15) | (((p1 >> a) & 15u) << 11) | (((c >> (a + 3)) & 3u) << 9) | (((c >> a) & 3u) << 7) |
                           (((m1 >> (a + 1)) & 15u) << 3) | ((m2 >> (a + 2)) & 7u);
            key = ((c >> (a + 2)) & 1u) ? key : KMC_VACANT;
            if (live) __builtin_nontemporal_store(key, out + (size_t)a * ny);
        }
    }
}

// P[s] = T[key[s]] (or zeros): the rate matrix of the stale-key state
__global__ __launch_bounds__(256) void materialize_rates_kernel(const uint32_t* __restrict__ keys, size_t total,
                                                                 const double* __restrict__ table, double* __restrict__ P)
{
    for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < total; s += (size_t)gridDim.x * 256) {
        const uint32_t key = keys[s];
        double2 v0 = make_double2(0., 0.), v1 = v0, v2 = v0;
        if (key != KMC_VACANT) {
            const double2* t = reinterpret_cast<const double2*>(table + (size_t)key * 6);
            v0 = t[0]; v1 = t[1]; v2 = t[2];
        }
        double2* out = reinterpret_cast<double2*>(P + s * 6);
        out[0] = v0; out[1] = v1; out[2] = v2;
    }
}

// Row R3 as a stand-alone call: one window per replica (one wave each).
__global__ __launch_bounds__(64) void refresh_window_kernel(const uint8_t* __restrict__ occ, int nx, int ny, int mode,
                                                             const int* __restrict__ se, uint32_t* __restrict__ keys)
{
    const int r = blockIdx.x, lane = threadIdx.x, n = nx * ny;
    const int is = se[r * 4 + 0], js = se[r * 4 + 1], ie = se[r * 4 + 2], je = se[r * 4 + 3];
    if (is < 0) return;
    const int imin = is < ie ? is : ie, imax = is < ie ? ie : is;
    const int jmin = js < je ? js : je, jmax = js < je ? je : js;
    int i_lo, n_i, j_lo, n_j;
    if (mode == 0) { i_lo = imin - 2; n_i = imax - imin + 4; j_lo = jmin - 2; n_j = jmax - jmin + 4; }
    else if (mode == 1) { i_lo = is - 2; n_i = 5; j_lo = js - 2; n_j = 6; }
    else { i_lo = imin - 2; n_i = imax - imin + 5; j_lo = jmin - 2; n_j = jmax - jmin + 5; }
    const uint8_t* o = occ + (size_t)r * n;
    for (int e = lane; e < n_i * n_j; e += 64) {
        const int a = e / n_j, b = e - a * n_j;
        int si = (i_lo + a) % nx; if (si < 0) si += nx;
        int sj = (j_lo + b) % ny; if (sj < 0) sj += ny;
        keys[(size_t)r * n + si * ny + sj] = o[si * ny + sj] ? key_from_bytes(o, nx, ny, si, sj) : KMC_VACANT;
    }
}

__global__ void site_probability_kernel(const uint8_t* __restrict__ occ, int nx, int ny, int i, int j,
                                        const double* __restrict__ table, uint32_t* key_out, double* rates_out)
{
    if (threadIdx.x != 0) return;
    const uint32_t k = key_from_bytes(occ, nx, ny, i, j);
    *key_out = k;
    for (int d = 0; d < 6; ++d) rates_out[d] = table[(size_t)k * 6 + d];
}

// ======================================================================
// kernels: lattice fill, tags, observables, exact gamma
// ======================================================================
// Exactly n_li sites per replica, uniformly without replacement: every site
// draws a 64-bit Philox key; the n_li smallest keys (ties by site index) get Li.
// One wave per replica; the threshold is found by bisection on the key value.
__global__ __launch_bounds__(64) void fill_random_kernel(uint8_t* __restrict__ occ, int n, const int* __restrict__ n_li,
                                                          uint32_t seed_lo, uint32_t seed_hi, uint32_t rid_base)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    const int want = n_li[r];
    auto site_key64 = [&](int s) -> unsigned long long {
        u32x4 c; c.x = (uint32_t)s; c.y = 0; c.z = (uint32_t)r + rid_base; c.w = KMC_STREAM_FILL;
        const u32x4 o = philox4x32_10(c, seed_lo, seed_hi);
        return ((unsigned long long)o.x << 32) | o.y;
    };
    // smallest T with count(key <= T) >= want
    unsigned long long lo = 0, hi = ~0ull;
    long long below = 0;
    if (n <= 4096) {
        // the LDS-size lattices of an ensemble: a lane's <= 64 keys are drawn once and stay in registers for the 64
        // bisection steps (2.1 ms -> 0.1 ms for 4096 x 64 x 64; the same T, the same lattice)
        unsigned long long kreg[64];
#pragma unroll
        for (int q = 0; q < 64; ++q) kreg[q] = q * 64 + lane < n ? site_key64(q * 64 + lane) : 0ull;
        while (lo < hi) {
            const unsigned long long mid = lo + ((hi - lo) >> 1);
            long long c = 0;
#pragma unroll
            for (int q = 0; q < 64; ++q) c += (q * 64 + lane < n) & (kreg[q] <= mid);
            c = wave_sum_i64(c);
            if (c >= want) hi = mid; else lo = mid + 1;
        }
#pragma unroll
        for (int q = 0; q < 64; ++q) below += (q * 64 + lane < n) & (kreg[q] < lo);
    } else {
        while (lo < hi) {
            const unsigned long long mid = lo + ((hi - lo) >> 1);
            long long c = 0;
            for (int s = lane; s < n; s += 64) c += site_key64(s) <= mid;
            c = wave_sum_i64(c);
            if (c >= want) hi = mid; else lo = mid + 1;
        }
        for (int s = lane; s < n; s += 64) below += site_key64(s) < lo;
    }
    const unsigned long long T = lo;
    below = wave_sum_i64(below);
    int ties_left = want - (int)below;         // ties at T, lowest site index first
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        const unsigned long long k = s < n ? site_key64(s) : ~0ull;
        const bool tie = s < n && k == T && want > 0;
        const unsigned long long m = __ballot(tie);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        const bool li = s < n && want > 0 && (k < T || (tie && rank < ties_left));
        if (s < n) occ[(size_t)r * n + s] = li ? 1 : 0;
        ties_left -= __popcll(m);
        if (ties_left < 0) ties_left = 0;
    }
}

// hex_LiCoO2_kmc.py:734-740
__global__ __launch_bounds__(256) void init_tags_kernel(const uint8_t* __restrict__ occ, int ny, int n, size_t total,
                                                         int4* __restrict__ tags)
{
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int s = (int)(g % n);
        tags[g] = occ[g] ? make_int4(s / ny, s % ny, 0, 0) : make_int4(-1, 0, 0, 0);
    }
}

// Rows D1/D2 on the device: the hex metric r^2 = di^2 + dj^2 + di*dj is an
// integer, so the per-replica sums are exact and order independent.
// out[r] = { sum r^2, sum di, sum dj, count }            hex_LiCoO2_kmc.py:567-585
__global__ __launch_bounds__(256) void msd_kernel(const int4* __restrict__ tags, int nx, int ny, long long* __restrict__ out)
{
    const int r = blockIdx.x, n = nx * ny;
    long long sq = 0, sdi = 0, sdj = 0, cnt = 0;
    for (int s = threadIdx.x; s < n; s += 256) {
        const int4 t = tags[(size_t)r * n + s];
        if (t.x < 0) continue;
        const int i = s / ny, j = s - i * ny;
        const long long di = (long long)t.x - ((long long)i + (long long)nx * t.z);
        const long long dj = (long long)t.y - ((long long)j + (long long)ny * t.w);
        const long long q = di * di + dj * dj + di * dj;
        sq += q < 0 ? -q : q; sdi += di; sdj += dj; ++cnt;
    }
    __shared__ long long red[4][4];
    sq = wave_sum_i64(sq); sdi = wave_sum_i64(sdi); sdj = wave_sum_i64(sdj); cnt = wave_sum_i64(cnt);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = sq; red[wv][1] = sdi; red[wv][2] = sdj; red[wv][3] = cnt; }
    __syncthreads();
    if (threadIdx.x < 4)
        out[r * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Row D1, per site: sqrt(|di^2 + dj^2 + di*dj|) * a (0 where vacant)   hex_LiCoO2_kmc.py:570-578
__global__ __launch_bounds__(256) void distance_kernel(const int4* __restrict__ tags, int nx, int ny, size_t total, double a,
                                                        double* __restrict__ out)
{
    const int n = nx * ny;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int4 t = tags[g];
        double v = 0.0;
        if (t.x >= 0) {
            const int s = (int)(g % n);
            const int i = s / ny, j = s - i * ny;
            const long long di = (long long)t.x - ((long long)i + (long long)nx * t.z);
            const long long dj = (long long)t.y - ((long long)j + (long long)ny * t.w);
            const long long q = di * di + dj * dj + di * dj;
            v = sqrt(fabs((double)q)) * a;
        }
        out[g] = v;
    }
}

// Row S1 with numpy's association: one lane per replica (every replica walks the
// same pairwise tree, so the lanes of a wave stay converged).
__global__ __launch_bounds__(64) void gamma_exact_kernel(const uint32_t* __restrict__ keys, int R, int n,
                                                          const double* __restrict__ table, double* __restrict__ gamma)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    PwFrame st[KMC_PW_DEPTH];
    const uint32_t* k = keys + (size_t)r * n;
    auto entry = [&](long e) -> double {
        const int s = (int)(e / 6);
        const int d = (int)(e - 6 * (long)s);
        const uint32_t kk = k[s];
        return kk == KMC_VACANT ? 0.0 : table[(size_t)kk * 6 + d];
    };
    gamma[r] = np_sum_model(6L * n, entry, st);
}

// Finishing step of kmc_diffusion_noalloc: the host arithmetic of kmc_diffusion (hex_LiCoO2_kmc.py:591-602),
// one lane per replica, from the exact integer sums msd_kernel left in `part`.
__global__ __launch_bounds__(256) void diffusion_finalize_kernel(const long long* __restrict__ part,
                                                                  const double* __restrict__ time_total,
                                                                  const double* __restrict__ conc, int R, int nx, int ny,
                                                                  double a, double* __restrict__ dstar, double* __restrict__ dj,
                                                                  double* __restrict__ msd_out)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const double n_atoms = (double)nx * ny * 1 * conc[r];                  // a float in the reference
    const double sq = (double)part[r * 4], si = (double)part[r * 4 + 1], sj = (double)part[r * 4 + 2], cnt = (double)part[r * 4 + 3];
    if (msd_out) { msd_out[r * 4] = sq; msd_out[r * 4 + 1] = si; msd_out[r * 4 + 2] = sj; msd_out[r * 4 + 3] = cnt; }
    if (n_atoms == 0) {                                                     // "concentration is zero, no atoms to move!"
        dstar[r] = __longlong_as_double(0x7FF8000000000000LL);
        dj[r] = __longlong_as_double(0x7FF8000000000000LL);
        return;
    }
    const double mean_sq = (a * a) * sq / cnt;
    const double dj2 = (a * a) * fabs(si * si + sj * sj + si * sj);
    const double pref = 1 / (2 * 2 * time_total[r] / n_atoms);
    dstar[r] = pref * mean_sq;
    dj[r] = pref * dj2 / n_atoms;
}

// WP[key] = {W[key], the six rates of row `key` as indices into the sorted palette} for hop_lat_kernel.
__global__ __launch_bounds__(256) void lat_wp_build_kernel(const double* __restrict__ table, const double* __restrict__ wsum,
                                                           const double* __restrict__ pal, int n_pal, LatWP* __restrict__ wp,
                                                           int* __restrict__ bad)
{
    const uint32_t key = blockIdx.x * 256 + threadIdx.x;
    unsigned long long rp = 0;
    for (int d = 0; d < 6; ++d) {
        const double v = table[(size_t)key * 6 + d];
        int idx = 0;
        if (v != 0.0) {
            int lo = 1, hi = n_pal - 1;                 // pal[1 .. n_pal) ascending
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (pal[mid] < v) lo = mid + 1; else hi = mid; }
            idx = lo;
            if (!(pal[idx] == v)) atomicAdd(bad, 1);
        }
        rp |= (unsigned long long)idx << (10 * d);
    }
    LatWP o; o.w = wsum[key]; o.rp = rp;
    wp[key] = o;
}

// Scratch of kmc_diffusion_noalloc: the per-replica integer sums of msd_kernel ([R][4] i64), which kmc_diffusion
// keeps in a buffer it allocates per call.  One array per GPU the library runs on (2 MiB), held until the process
// ends.  Host code never uses this NAME as an address: kmc_create asks hipGetSymbolAddress for the array's address
// on the context's device and kernels / memsets take that pointer.
#define KMC_NOALLOC_MAX_REPLICAS 65536
static __device__ __align__(256) long long kmc_noalloc_scratch[(size_t)KMC_NOALLOC_MAX_REPLICAS * 4];

// ======================================================================
// host API
// ======================================================================
// Device / host scratch for one entry point at a time (a context is single-owner and its calls are stream-ordered;
// every user synchronises the stream before it returns).
static int dev_pool(kmc_ctx* c, size_t bytes, void** out)
{
    if (c->pool_cap < bytes) {
        if (c->pool) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->pool)); }
        c->pool = nullptr; c->pool_cap = 0;
        const size_t cap = bytes < 65536 ? 65536 : bytes;
        HIPCHK(c, hipMalloc(&c->pool, cap + 256));
        c->pool_cap = cap;
    }
    *out = c->pool;
    return KMC_OK;
}
// The bulk getters (kmc_get_rates: a 201 MB staging chunk, kmc_get_distances: 134 MB at 4096 x 64 x 64) give their
// staging buffer back: the pool keeps at most KMC_POOL_KEEP bytes between calls.  Called after the stream is idle.
#define KMC_POOL_KEEP ((size_t)8 << 20)
static void dev_pool_trim(kmc_ctx* c)
{
    if (c->pool && c->pool_cap > KMC_POOL_KEEP) {
        hipFree(c->pool);
        c->pool = nullptr; c->pool_cap = 0;
    }
}
static void* host_pool(kmc_ctx* c, size_t bytes)
{
    if (c->hpool_cap < bytes) {
        free(c->hpool);
        c->hpool = malloc(bytes < 65536 ? 65536 : bytes);
        c->hpool_cap = c->hpool ? (bytes < 65536 ? 65536 : bytes) : 0;
    }
    return c->hpool;
}

static size_t host_hop_lds_bytes(int nx, int n)
{
    size_t a, b, c;
    return hop_lds_layout(nx, n, &a, &b, &c);
}

extern "C" {

const char* kmc_last_error(kmc_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

int kmc_create(kmc_ctx** out, int device, int n_replicas, int nx, int ny, int window_mode)
{
    if (!out) return fail(nullptr, KMC_EINVAL, "kmc_create: out is null");
    *out = nullptr;
    if (n_replicas < 1 || nx < 3 || ny < 3 || (long long)nx * ny > (1ll << 24))
        return fail(nullptr, KMC_EINVAL, "kmc_create: need n_replicas >= 1, nx, ny >= 3, nx*ny <= 2^24 (got %d, %d, %d)",
                    n_replicas, nx, ny);
    if (window_mode < 0 || window_mode > 2) return fail(nullptr, KMC_EINVAL, "kmc_create: window_mode must be 0, 1 or 2");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, KMC_EHIP, "kmc_create: no HIP device (%s); this engine has no CPU backend",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, KMC_EINVAL, "kmc_create: device %d out of range [0, %d)", device, ndev);
    kmc_ctx* c = (kmc_ctx*)calloc(1, sizeof(kmc_ctx));
    if (!c) return fail(nullptr, KMC_EINVAL, "kmc_create: out of host memory");
    c->device = device; c->R = n_replicas; c->nx = nx; c->ny = ny; c->n = nx * ny; c->window_mode = window_mode;
    c->lds_path = nx >= 5 && ny >= 5 && nx <= 64 && ny <= 64;
    c->margin_scale = 1.0;
#define CREATE_CHK(call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fail(nullptr, KMC_EHIP, "kmc_create: %s failed: %s", #call, hipGetErrorString(e_)); \
            kmc_destroy(c);                                                                \
            return KMC_EHIP;                                                               \
        }                                                                                  \
    } while (0)
    CREATE_CHK(hipSetDevice(device));
    CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreate(&c->ev0));
    CREATE_CHK(hipEventCreate(&c->ev1));
    for (int k = 0; k < KMC_EV_RING; ++k)
        for (int g = 0; g < KMC_GROUPS; ++g) { CREATE_CHK(hipEventCreate(&c->run_ev[k][g][0])); CREATE_CHK(hipEventCreate(&c->run_ev[k][g][1])); }
    for (int g = 0; g < KMC_GROUPS; ++g) {
        CREATE_CHK(hipStreamCreateWithFlags(&c->gstream[g], hipStreamNonBlocking));
        CREATE_CHK(hipEventCreateWithFlags(&c->gdone[g], hipEventDisableTiming));
    }
    CREATE_CHK(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
    c->need_fork = true;
    const size_t Rn = (size_t)c->R * c->n;
    CREATE_CHK(hipMalloc((void**)&c->occ, Rn + 16));
    CREATE_CHK(hipMalloc((void**)&c->keys, Rn * 4 + 16));
    CREATE_CHK(hipMalloc((void**)&c->tags, Rn * 16 + 16));
    // 2^19 rows: hop_lds64_kernel keys a vacant site as 0x7FFFF (bit 18 + all neighbours Li), whose
    // row is all zeros; rows 2^18 .. 2^19-2 are never touched
    CREATE_CHK(hipMalloc((void**)&c->table, (size_t)(2 * KMC_TABLE_ROWS) * 6 * 8));
    CREATE_CHK(hipMalloc((void**)&c->wsum, (size_t)(2 * KMC_TABLE_ROWS) * 8));
    CREATE_CHK(hipMemsetAsync(c->table + (size_t)(2 * KMC_TABLE_ROWS - 1) * 6, 0, 48, c->stream));
    CREATE_CHK(hipMemsetAsync(c->wsum + (2 * KMC_TABLE_ROWS - 1), 0, 8, c->stream));
    CREATE_CHK(hipMalloc((void**)&c->tcompact, (size_t)6 * KMC_COMPACT_ROWS * 8));
    CREATE_CHK(hipMalloc((void**)&c->tcompact2, (size_t)3 * KMC_COMPACT2_ROWS * 16));
    CREATE_CHK(hipMalloc((void**)&c->time_total, c->R * 8));
    CREATE_CHK(hipMalloc((void**)&c->step_time, c->R * 8));
    CREATE_CHK(hipMalloc((void**)&c->gamma, c->R * 8));
    CREATE_CHK(hipMalloc((void**)&c->hist, c->R * 6 * 8));
    CREATE_CHK(hipMalloc((void**)&c->hops_done, c->R * 8));
    CREATE_CHK(hipMalloc((void**)&c->n_exact, c->R * 8));
    CREATE_CHK(hipMalloc((void**)&c->status, c->R * 4));
    CREATE_CHK(hipMalloc((void**)&c->barrier, c->R * 5 * 8));
    CREATE_CHK(hipMalloc((void**)&c->scratch_d, c->R * 8 * 8));
    {
        // each GPU holds its own copy of the array at its own address: asked for with THIS device current
        void* sp = nullptr;
        CREATE_CHK(hipGetSymbolAddress(&sp, HIP_SYMBOL(kmc_noalloc_scratch)));
        c->noalloc_scratch = static_cast<long long*>(sp);
    }
    CREATE_CHK(hipMemsetAsync(c->occ, 0, Rn, c->stream));
    CREATE_CHK(hipMemsetAsync(c->keys, 0xFF, Rn * 4, c->stream));
    CREATE_CHK(hipMemsetAsync(c->tags, 0, Rn * 16, c->stream));
    CREATE_CHK(hipMemsetAsync(c->time_total, 0, c->R * 8, c->stream));
    CREATE_CHK(hipMemsetAsync(c->step_time, 0, c->R * 8, c->stream));
    CREATE_CHK(hipMemsetAsync(c->gamma, 0, c->R * 8, c->stream));
    CREATE_CHK(hipMemsetAsync(c->hist, 0, c->R * 6 * 8, c->stream));
    CREATE_CHK(hipMemsetAsync(c->hops_done, 0, c->R * 8, c->stream));
    CREATE_CHK(hipMemsetAsync(c->n_exact, 0, c->R * 8, c->stream));
    CREATE_CHK(hipMemsetAsync(c->status, 0, c->R * 4, c->stream));
    c->lds64 = c->lds_path && nx == 64 && ny == 64;
    // The 16-per-CU kernel is the default at every replica count: up to 8 replicas per CU the two kernels are
    // within 2 % of each other (scripts/kernel_crossover.py), beyond it wins by up to 1.5x.
    c->q64_auto = true;
    c->q64 = c->lds64 && c->q64_auto;
    c->qany_ok = c->lds_path && !c->lds64 && nx >= 8 && ny >= 8;
    c->qany = c->qany_ok;
    if (c->lds64 || c->qany_ok) CREATE_CHK(hipMalloc(&c->xscratch, (size_t)c->R * Q64_XSCRATCH));
    c->lat_ok = c->lds64 || c->qany_ok;
    c->vac_row_ok = true;
    if (c->lat_ok) {
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lat_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LAT_LDS_BYTES));
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lat_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LAT_LDS_BYTES));
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lat_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LAT_LDS_BYTES));
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lat_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LAT2_LDS_BYTES));
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lat_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LAT2_LDS_BYTES));
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lat_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LAT2_LDS_BYTES));
    }
    if (c->lds_path) {
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lds64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, K64_LDS_BYTES));
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lds64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, K64_LDS_BYTES));
        const size_t lds = host_hop_lds_bytes(c->nx, c->n);
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CREATE_CHK(hipFuncSetAttribute((const void*)hop_lds_kernel_w4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else {
        if (nx <= 4096) {
            const size_t lds = ((sizeof(GbShared) + 15) & ~(size_t)15) + (size_t)nx * 8;
            CREATE_CHK(hipFuncSetAttribute((const void*)hop_global_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        c->big_ok = big_geometry(nx, ny, &c->geom);
        c->use_big = c->big_ok;
        if (c->big_ok) {
            CREATE_CHK(hipFuncSetAttribute((const void*)hop_big_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, c->geom.lds_bytes));
            CREATE_CHK(hipFuncSetAttribute((const void*)hop_big_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, c->geom.lds_bytes));
            CREATE_CHK(hipFuncSetAttribute((const void*)hop_big_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, c->geom.lds_bytes));
            CREATE_CHK(hipFuncSetAttribute((const void*)hop_big_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, c->geom.lds_bytes));
            c->big_sitew_ok = big_sitew_ok(c->geom);
            if (c->big_sitew_ok) {
                CREATE_CHK(hipFuncSetAttribute((const void*)hop_big_kernel<false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big_sitew_lds_bytes(c->geom)));
                CREATE_CHK(hipFuncSetAttribute((const void*)hop_big_kernel<false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big_sitew_lds_bytes(c->geom)));
            }
        }
    }
    CREATE_CHK(hipStreamSynchronize(c->stream));
#undef CREATE_CHK
    *out = c;
    // barrier stats start "empty"
    kmc_init_displacements(c);
    return KMC_OK;
}

int kmc_destroy(kmc_ctx* c)
{
    if (!c) return KMC_OK;
    hipSetDevice(c->device);
    for (int g = 0; g < KMC_GROUPS; ++g) if (c->gstream[g]) hipStreamSynchronize(c->gstream[g]);
    if (c->stream) hipStreamSynchronize(c->stream);
    void* bufs[] = {c->occ, c->keys, c->tags, c->table, c->wsum, c->time_total, c->step_time, c->gamma, c->hist,
                    c->hops_done, c->n_exact, c->status, c->barrier, c->scratch_d, c->draws, c->sel, c->w_g, c->tcompact, c->tcompact2, c->hier, c->xscratch,
                    c->coll_buf, c->pool, c->wp, c->pal, c->sw};
    for (void* b : bufs) if (b) hipFree(b);
    free(c->hpool);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    for (int k = 0; k < KMC_EV_RING; ++k) for (int g = 0; g < KMC_GROUPS; ++g) for (int q = 0; q < 2; ++q)
        if (c->run_ev[k][g][q]) hipEventDestroy(c->run_ev[k][g][q]);
    for (int g = 0; g < KMC_GROUPS; ++g) {
        if (c->gdone[g]) hipEventDestroy(c->gdone[g]);
        if (c->gstream[g]) hipStreamDestroy(c->gstream[g]);
    }
    if (c->fork_ev) hipEventDestroy(c->fork_ev);
    if (c->stream) hipStreamDestroy(c->stream);
    free(c);
    return KMC_OK;
}

int kmc_sync(kmc_ctx* c)
{
    CTXCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_last_kernel_ms(kmc_ctx* c, float* ms)
{
    CTXCHK(c);                                  // (timed_slot is reset by the next timed launch, time_begin)
    if (!ms) return fail(c, KMC_EINVAL, "kmc_last_kernel_ms: null pointer");
    if (!c->timed) return fail(c, KMC_ESTATE, "kmc_last_kernel_ms: nothing timed yet");
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    if (c->timed_slot >= 0) {                   // a run in launch groups: first start to last end
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // (either group may get the chip first: take the earliest start against the latest end)
        const int sl = c->timed_slot;
        for (int gf = 0; gf < c->run_groups[sl]; ++gf)
            for (int gl = 0; gl < c->run_groups[sl]; ++gl) {
                float m2 = 0.f;
                HIPCHK(c, hipEventElapsedTime(&m2, c->run_ev[sl][gf][0], c->run_ev[sl][gl][1]));
                if (m2 > *ms) *ms = m2;
            }
    }
    return KMC_OK;
}

int kmc_run_kernel_times(kmc_ctx* c, float* ms, int max_n, int* n_out)
{
    CTXCHK(c);
    if (!ms || !n_out || max_n < 0) return fail(c, KMC_EINVAL, "kmc_run_kernel_times: bad arguments");
    long long n = c->run_launches < KMC_EV_RING ? c->run_launches : KMC_EV_RING;
    if (n > max_n) n = max_n;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (long long k = 0; k < n; ++k) {          // oldest first
        const int slot = (int)((c->run_launches - n + k) % KMC_EV_RING);
        HIPCHK(c, hipEventElapsedTime(&ms[k], c->run_ev[slot][0][0], c->run_ev[slot][0][1]));
    }
    *n_out = (int)n;
    return KMC_OK;
}

// GPU time from the start of the n-th last run's first kernel to the end of the last run's last kernel (HIP
// events), and how many kernels each of those runs launched (1, or KMC_GROUPS concurrent ones).
int kmc_run_span_ms(kmc_ctx* c, int n_runs, float* span_ms, int* groups_out)
{
    CTXCHK(c);
    if (!span_ms || n_runs < 1 || n_runs > KMC_EV_RING || n_runs > c->run_launches)
        return fail(c, KMC_EINVAL, "kmc_run_span_ms: need 1 <= n_runs <= min(%d, runs so far)", KMC_EV_RING);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int first = (int)((c->run_launches - n_runs) % KMC_EV_RING), last = (int)((c->run_launches - 1) % KMC_EV_RING);
    float best = 0.f;
    for (int gf = 0; gf < c->run_groups[first]; ++gf)
        for (int gl = 0; gl < c->run_groups[last]; ++gl) {
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->run_ev[first][gf][0], c->run_ev[last][gl][1]));
            if (ms > best) best = ms;
        }
    *span_ms = best;
    if (groups_out) *groups_out = c->run_groups[last];
    return KMC_OK;
}

int kmc_set_launch_groups(kmc_ctx* c, int mode)
{
    CTXCHK(c);
    if (mode < 0 || mode > 2) return fail(c, KMC_EINVAL, "kmc_set_launch_groups: mode must be 0 (automatic), 1 or 2");
    c->split = mode;
    return KMC_OK;
}

int kmc_uses_lds_path(kmc_ctx* c) { return c && c->lds_path ? 1 : 0; }
int kmc_lds_bytes(kmc_ctx* c)
{
    if (!c || !c->lds_path) return 0;
    if (c->qany) return Q64_LDS_BYTES;
    return c->lds64 ? (c->q64 ? Q64_LDS_BYTES : K64_LDS_BYTES) : (int)host_hop_lds_bytes(c->nx, c->n);
}

// which = 0: the default kernel for the lattice size and replica count; 1: the generic kernel (any size);
// 2 / 3: for 64 x 64 replicas, the 8-replicas-per-CU kernel (kmc_hop_lds64.h) / the 16-per-CU one (kmc_hop_q64.h);
// 4: the 16-per-CU layout for lattices of 8 .. 64 rows and columns other than 64 x 64 (the default there)
// (hop_q64_kernel and hop_lat_kernel store a vacant site as the all-Li key 0x3FFFF, whose table row is zero for every table
// that describes hops -- all six end points are occupied.  A table whose row 0x3FFFF is NOT zero, vac_row_ok == false, takes
// the kernels that keep the explicit zero row 0x7FFFF instead; asking for the others by number is then an error.)
static void apply_hop_kernel_choice(kmc_ctx* c)
{
    int which = c->which_req;
    c->lat_mode = which >= 5 ? 1 : (which == 0 ? 0 : -1);
    c->lat_pk = which == 6;                    // its two-per-CU layout whatever the replica count (tests)
    if (which >= 5) which = 0;
    c->lds64 = c->lds_path && c->nx == 64 && c->ny == 64 && which != 1 && which != 4;
    c->qany = c->qany_ok && (which == 4 || which == 0);
    c->q64 = c->lds64 && (which == 3 || (which == 0 && c->q64_auto));
    c->use_big = c->big_ok && which != 1;
    c->big_tree = which == 2;                  // lattices beyond 64 x 64: hop_big_kernel without the per-site words
    if (!c->vac_row_ok) { c->qany = false; c->q64 = false; c->lat_mode = -1; }
}

int kmc_set_hop_kernel(kmc_ctx* c, int which)
{
    CTXCHK(c);
    if (which < 0 || which > 6) return fail(c, KMC_EINVAL, "kmc_set_hop_kernel: which must be 0 ... 6");
    if (which >= 5 && !c->lat_ok)
        return fail(c, KMC_EINVAL, "kmc_set_hop_kernel: the one-replica-per-CU kernel takes lattices of 8 ... 64 rows and columns");
    if (which >= 3 && c->lds_path && !c->vac_row_ok)
        return fail(c, KMC_ESTATE, "kmc_set_hop_kernel: the rate table's all-Li row (key 0x3FFFF) is not zero, which the kernels %d "
                    "stands for take for granted (they store a vacant site as that key)", which);
    c->which_req = which;
    apply_hop_kernel_choice(c);
    return KMC_OK;
}

// after a table has been set: is its all-Li row zero?
static int check_vacant_row(kmc_ctx* c)
{
    double row[6];
    HIPCHK(c, hipMemcpyAsync(row, c->table + (size_t)0x3FFFF * 6, sizeof(row), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    bool ok = true;
    for (int d = 0; d < 6; ++d) ok = ok && row[d] == 0.0;
    c->vac_row_ok = ok;
    if (!ok && c->which_req >= 3) c->which_req = 0;      // the kernel asked for by number no longer applies
    apply_hop_kernel_choice(c);
    return KMC_OK;
}

const char* kmc_hop_kernel_name(kmc_ctx* c)
{
    if (!c) return "none";
    if (!c->lds_path)
        return c->use_big ? (c->big_sitew_ok && !c->big_tree && c->pal_state >= 0 ? "hop_big_kernel<site words>" : "hop_big_kernel") : "hop_global_kernel";
    if (c->lat_ok && c->pal_state >= 0 && (c->lat_mode == 1 || (c->lat_mode == 0 && c->R <= KMC_LAT_AUTO_MAX && c->split != 2)))
        return c->R > KMC_LAT_ONE_PER_CU || c->lat_pk ? (c->lds64 ? "hop_lat_kernel<2 per CU>" : "hop_lat_kernel<any, 2 per CU>")
                                         : (c->lds64 ? "hop_lat_kernel" : "hop_lat_kernel<any>");
    if (c->qany) return "hop_q64_kernel<any>";
    if (c->lds64) return c->q64 ? "hop_q64_kernel" : "hop_lds64_kernel";
    return "hop_lds_kernel";
}

// The Philox streams (random fill, per-hop draws) are keyed by (seed, global replica id, counter): a context
// whose replica 0 is replica `offset` of a larger ensemble draws exactly what that ensemble's one context would.
int kmc_set_replica_offset(kmc_ctx* c, uint64_t offset)
{
    CTXCHK(c);
    if (offset > 0xFFFFFFFFull - (uint64_t)c->R)           // (not offset + R > ...: that sum wraps for offsets near 2^64)
        return fail(c, KMC_EINVAL, "kmc_set_replica_offset: offset + replicas must fit 32 bits (the Philox counter word)");
    c->rid_base = (uint32_t)offset;
    return KMC_OK;
}

// ---------------------------------------------------------------- tables
static int build_compact(kmc_ctx* c)
{
    int* flag = reinterpret_cast<int*>(c->scratch_d);
    HIPCHK(c, hipMemsetAsync(flag, 0, 4, c->stream));
    compact_build_kernel<<<6 * KMC_COMPACT_ROWS / 256, 256, 0, c->stream>>>(c->table, c->tcompact);
    // the pair tables need no check of their own: a rate that reads 13 key bits reads the pair's 15
    compact2_build_kernel<<<3 * KMC_COMPACT2_ROWS / 256, 256, 0, c->stream>>>(c->table, c->tcompact2);
    compact_verify_kernel<<<KMC_TABLE_ROWS / 256, 256, 0, c->stream>>>(c->table, c->tcompact, flag);
    HIPCHK(c, hipGetLastError());
    int bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_compact = bad == 0 && c->vac_row_ok;        // (the compact paths read a vacant site's rates at the all-Li entry)
    return KMC_OK;
}

int kmc_table_is_compact(kmc_ctx* c) { return c && c->have_compact ? 1 : 0; }

// The tables of hop_lat_kernel: the palette of the rate table's distinct values (<= LAT_PAL_MAX, 0.0 first) and
// WP[key] = {W[key], the row's six palette indices}.  Built on first use for the current table: the table comes
// back to the host once (12.6 MB), its values go through a small open-addressing set.
static int ensure_lat_tables(kmc_ctx* c)
{
    if (c->pal_state != 0) return KMC_OK;
    const size_t n_ent = (size_t)KMC_TABLE_ROWS * 6;
    uint64_t* h = (uint64_t*)host_pool(c, n_ent * 8);
    if (!h) return fail(c, KMC_EINVAL, "kmc_run: out of host memory (palette)");
    HIPCHK(c, hipMemcpyAsync(h, c->table, n_ent * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint32_t SLOTS = 8192;
    static const uint64_t EMPTY = ~0ull;               // a NaN pattern: never a rate
    uint64_t set[SLOTS];
    for (uint32_t k = 0; k < SLOTS; ++k) set[k] = EMPTY;
    double pal[LAT_PAL_MAX];
    int n_pal = 0;
    pal[n_pal++] = 0.0;                                // index 0 = rate 0: what lanes without an entry read
    set[0] = 0;                                        // (hash of +0.0 below is 0)
    bool ok = true;
    for (size_t e = 0; e < n_ent && ok; ++e) {
        const uint64_t b = h[e];
        uint32_t k = (uint32_t)((b * 0x9E3779B97F4A7C15ull) >> 51);          // 13 bits
        while (set[k] != b) {
            if (set[k] == EMPTY) {
                if (n_pal == LAT_PAL_MAX) { ok = false; break; }
                set[k] = b;
                memcpy(&pal[n_pal++], &b, 8);
                break;
            }
            k = (k + 1) & (SLOTS - 1);
        }
    }
    if (!ok) { c->pal_state = -1; return KMC_OK; }
    std::sort(pal + 1, pal + n_pal);                   // a deterministic order (the indices never leave the device)
    for (int k = n_pal; k < LAT_PAL_MAX; ++k) pal[k] = 0.0;
    if (!c->pal) HIPCHK(c, hipMalloc((void**)&c->pal, LAT_PAL_MAX * 8));
    if (!c->wp) HIPCHK(c, hipMalloc((void**)&c->wp, (size_t)KMC_TABLE_ROWS * sizeof(LatWP)));
    HIPCHK(c, hipMemcpyAsync(c->pal, pal, LAT_PAL_MAX * 8, hipMemcpyHostToDevice, c->stream));
    int* flag = reinterpret_cast<int*>(c->scratch_d);
    HIPCHK(c, hipMemsetAsync(flag, 0, 4, c->stream));
    lat_wp_build_kernel<<<KMC_TABLE_ROWS / 256, 256, 0, c->stream>>>(c->table, c->wsum, c->pal, n_pal, c->wp, flag);
    HIPCHK(c, hipGetLastError());
    int bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad) return fail(c, KMC_ESTATE, "palette build: %d table entries not found in the palette", bad);
    c->n_pal = n_pal;
    c->pal_state = 1;
    return KMC_OK;
}

int kmc_set_base_table(kmc_ctx* c, const double* base512, double dumbbell)
{
    CTXCHK(c);
    if (!base512) return fail(c, KMC_EINVAL, "kmc_set_base_table: null table");
    double* dbase = nullptr;
    if (int rc = dev_pool(c, 512 * 8, (void**)&dbase)) return rc;
    HIPCHK(c, hipMemcpyAsync(dbase, base512, 512 * 8, hipMemcpyHostToDevice, c->stream));
    time_begin(c);
    fold_table_kernel<<<KMC_TABLE_ROWS / 256, 256, 0, c->stream>>>(dbase, dumbbell, c->table, c->wsum);
    time_end(c);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_table = true;
    c->pal_state = 0;
    if (int rc = check_vacant_row(c)) return rc;
    return build_compact(c);
}

int kmc_set_hex_table(kmc_ctx* c, const double* table)
{
    CTXCHK(c);
    if (!table) return fail(c, KMC_EINVAL, "kmc_set_hex_table: null table");
    HIPCHK(c, hipMemcpyAsync(c->table, table, (size_t)KMC_TABLE_ROWS * 48, hipMemcpyHostToDevice, c->stream));
    wsum_kernel<<<KMC_TABLE_ROWS / 256, 256, 0, c->stream>>>(c->table, c->wsum);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_table = true;
    c->pal_state = 0;
    if (int rc = check_vacant_row(c)) return rc;
    return build_compact(c);
}

int kmc_get_hex_table(kmc_ctx* c, double* table)
{
    CTXCHK(c);
    if (!c->have_table) return fail(c, KMC_ESTATE, "kmc_get_hex_table: no table set");
    HIPCHK(c, hipMemcpyAsync(table, c->table, (size_t)KMC_TABLE_ROWS * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_generate_base_table(kmc_ctx* c, const double* eci47, double prefactor, double kb_t, double* base512_out,
                            double* energies_out)
{
    CTXCHK(c);
    if (!eci47 || !base512_out) return fail(c, KMC_EINVAL, "kmc_generate_base_table: null pointer");
    double* d = nullptr;
    if (int rc = dev_pool(c, (47 + 512 + 512) * 8 + 64, (void**)&d)) return rc;
    HIPCHK(c, hipMemcpyAsync(d, eci47, 47 * 8, hipMemcpyHostToDevice, c->stream));
    lce_rates_kernel<<<1, 512, 0, c->stream>>>(d, prefactor, kb_t, d + 48, d + 48 + 512);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(base512_out, d + 48, 512 * 8, hipMemcpyDeviceToHost, c->stream));
    if (energies_out) HIPCHK(c, hipMemcpyAsync(energies_out, d + 48 + 512, 512 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

// ---------------------------------------------------------------- lattice
int kmc_set_lattice(kmc_ctx* c, const int8_t* occ)
{
    CTXCHK(c);
    if (!occ) return fail(c, KMC_EINVAL, "kmc_set_lattice: null pointer");
    const size_t Rn = (size_t)c->R * c->n;
    for (size_t s = 0; s < Rn; ++s)
        if (occ[s] != 0 && occ[s] != 1) return fail(c, KMC_EINVAL, "kmc_set_lattice: occupancy must be 0 or 1 (site %zu holds %d)", s, (int)occ[s]);
    HIPCHK(c, hipMemcpyAsync(c->occ, occ, Rn, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_keys = false;
    return KMC_OK;
}

int kmc_get_lattice(kmc_ctx* c, int8_t* occ)
{
    CTXCHK(c);
    if (!occ) return fail(c, KMC_EINVAL, "kmc_get_lattice: null pointer");
    HIPCHK(c, hipMemcpyAsync(occ, c->occ, (size_t)c->R * c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_fill_random(kmc_ctx* c, const double* conc, uint64_t seed)
{
    CTXCHK(c);
    if (!conc) return fail(c, KMC_EINVAL, "kmc_fill_random: null pointer");
    int* h = (int*)malloc(sizeof(int) * c->R);
    for (int r = 0; r < c->R; ++r) {
        if (!(conc[r] >= 0.0 && conc[r] <= 1.0)) { free(h); return fail(c, KMC_EINVAL, "kmc_fill_random: concentration %g of replica %d not in [0, 1]", conc[r], r); }
        h[r] = (int)((double)c->nx * c->ny * 1 * conc[r]);          // int(N*N*1*x), hex_LiCoO2_kmc.py:484
    }
    int* d = reinterpret_cast<int*>(c->scratch_d);
    hipError_t e = hipMemcpyAsync(d, h, sizeof(int) * c->R, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free(h);
    HIPCHK(c, e);
    fill_random_kernel<<<c->R, 64, 0, c->stream>>>(c->occ, c->n, d, (uint32_t)seed, (uint32_t)(seed >> 32), c->rid_base);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_keys = false;
    return KMC_OK;
}

// ---------------------------------------------------------------- rates
static int launch_rates_full(kmc_ctx* c)
{
    if (c->nx >= 5 && c->ny >= 5 && c->nx <= 64 && c->ny <= 64) {
        if (c->nx == 64 && c->ny == 64)
            rates_cols_kernel<64><<<c->R, 64, 0, c->stream>>>(c->occ, c->R, c->nx, c->ny, c->keys);
        else {
            const int per_wave = c->ny <= 16 ? 4 : (c->ny <= 32 ? 2 : 1);
            // (two or four waves sharing a lattice's rows, for more waves on the chip, are slower: 4096 x 30x30 10.8 us with one,
            // 11.2 with two, 13.2 with four)
            rates_cols_kernel<0><<<(c->R + per_wave - 1) / per_wave, 64, 0, c->stream>>>(c->occ, c->R, c->nx, c->ny, c->keys);
        }
        HIPCHK(c, hipGetLastError());
        return KMC_OK;
    }
    if (c->nx >= 5 && c->ny >= 5) {
        // tiles of 28 rows when that still gives the chip 32 waves per CU, else of 12
        const int tj = (c->ny + 63) / 64;
        const long long b28 = (long long)c->R * ((c->nx + 27) / 28) * tj, b12 = (long long)c->R * ((c->nx + 11) / 12) * tj;
        if (b12 > 0x7fffffffLL) return fail(c, KMC_EINVAL, "rates_full: grid too large");
        if (b12 < 4096) {
            const long long b4 = (long long)c->R * ((c->nx + 3) / 4) * tj;
            rates_tile_kernel<4><<<(unsigned)b4, 64, 0, c->stream>>>(c->occ, c->nx, c->ny, (c->nx + 3) / 4, tj, c->keys);
        } else if (b28 >= 8192) rates_tile_kernel<28><<<(unsigned)b28, 64, 0, c->stream>>>(c->occ, c->nx, c->ny, (c->nx + 27) / 28, tj, c->keys);
        else rates_tile_kernel<12><<<(unsigned)b12, 64, 0, c->stream>>>(c->occ, c->nx, c->ny, (c->nx + 11) / 12, tj, c->keys);
        HIPCHK(c, hipGetLastError());
        return KMC_OK;
    }
    const int ti = (c->nx + RF_TI - 1) / RF_TI, tj = (c->ny + RF_TJ - 1) / RF_TJ;
    const long long blocks = (long long)c->R * ti * tj;
    if (blocks > 0x7fffffffLL) return fail(c, KMC_EINVAL, "rates_full: grid too large");
    rates_full_kernel<<<(unsigned)blocks, 256, 0, c->stream>>>(c->occ, c->nx, c->ny, ti, tj, c->keys);
    HIPCHK(c, hipGetLastError());
    return KMC_OK;
}

int kmc_rates_full(kmc_ctx* c)
{
    CTXCHK(c);
    if (!c->have_table) return fail(c, KMC_ESTATE, "kmc_rates_full: set a rate table first");
    time_begin(c);
    int rc = launch_rates_full(c);
    time_end(c);
    if (rc) return rc;
    c->have_keys = true;
    return KMC_OK;
}

int kmc_get_keys(kmc_ctx* c, uint32_t* keys)
{
    CTXCHK(c);
    if (!keys) return fail(c, KMC_EINVAL, "kmc_get_keys: null pointer");
    if (!c->have_keys) return fail(c, KMC_ESTATE, "kmc_get_keys: run kmc_rates_full first");
    HIPCHK(c, hipMemcpyAsync(keys, c->keys, (size_t)c->R * c->n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_set_keys(kmc_ctx* c, const uint32_t* keys)
{
    CTXCHK(c);
    if (!keys) return fail(c, KMC_EINVAL, "kmc_set_keys: null pointer");
    const size_t Rn = (size_t)c->R * c->n;
    for (size_t s = 0; s < Rn; ++s)
        if (keys[s] != KMC_VACANT_KEY && keys[s] >= KMC_TABLE_ROWS)
            return fail(c, KMC_EINVAL, "kmc_set_keys: key %u at %zu is not an 18-bit key", keys[s], s);
    HIPCHK(c, hipMemcpyAsync(c->keys, keys, Rn * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_keys = true;
    return KMC_OK;
}

int kmc_get_rates(kmc_ctx* c, double* P)
{
    CTXCHK(c);
    if (!P) return fail(c, KMC_EINVAL, "kmc_get_rates: null pointer");
    if (!c->have_keys || !c->have_table) return fail(c, KMC_ESTATE, "kmc_get_rates: needs a table and kmc_rates_full");
    const size_t Rn = (size_t)c->R * c->n;
    // staged through a bounded device buffer so that 4096 x 64 x 64 (805 MB) needs no 805 MB of HBM
    const size_t chunk = Rn < ((size_t)1 << 22) ? Rn : ((size_t)1 << 22);
    double* d = nullptr;
    if (int rc = dev_pool(c, chunk * 48, (void**)&d)) return rc;
    for (size_t s0 = 0; s0 < Rn; s0 += chunk) {
        const size_t m = Rn - s0 < chunk ? Rn - s0 : chunk;
        materialize_rates_kernel<<<2048, 256, 0, c->stream>>>(c->keys + s0, m, c->table, d);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(P + s0 * 6, d, m * 48, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    dev_pool_trim(c);
    return KMC_OK;
}

int kmc_refresh_window(kmc_ctx* c, const int32_t* se)
{
    CTXCHK(c);
    if (!se) return fail(c, KMC_EINVAL, "kmc_refresh_window: null pointer");
    if (!c->have_keys) return fail(c, KMC_ESTATE, "kmc_refresh_window: run kmc_rates_full (or kmc_set_keys) first");
    for (int r = 0; r < c->R; ++r) {
        const int32_t* q = se + 4 * r;
        if (q[0] < 0) continue;
        if (q[0] >= c->nx || q[1] < 0 || q[1] >= c->ny || abs(q[2] - q[0]) > 2 || abs(q[3] - q[1]) > 2)
            return fail(c, KMC_EINVAL, "kmc_refresh_window: replica %d: start (%d,%d) / end (%d,%d) out of range", r, q[0], q[1], q[2], q[3]);
    }
    int* d = nullptr;
    if (int rc = dev_pool(c, (size_t)c->R * 16, (void**)&d)) return rc;
    HIPCHK(c, hipMemcpyAsync(d, se, (size_t)c->R * 16, hipMemcpyHostToDevice, c->stream));
    refresh_window_kernel<<<c->R, 64, 0, c->stream>>>(c->occ, c->nx, c->ny, c->window_mode, d, c->keys);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_site_probability(kmc_ctx* c, int replica, int i, int j, uint32_t* key, double* rates6)
{
    CTXCHK(c);
    if (!c->have_table) return fail(c, KMC_ESTATE, "kmc_site_probability: set a rate table first");
    if (replica < 0 || replica >= c->R || i < 0 || i >= c->nx || j < 0 || j >= c->ny)
        return fail(c, KMC_EINVAL, "kmc_site_probability: replica/site out of range");
    uint32_t* dk = reinterpret_cast<uint32_t*>(c->scratch_d);
    double* dr = c->scratch_d + 1;
    site_probability_kernel<<<1, 64, 0, c->stream>>>(c->occ + (size_t)replica * c->n, c->nx, c->ny, i, j, c->table, dk, dr);
    HIPCHK(c, hipGetLastError());
    uint32_t hk; double hr[6];
    HIPCHK(c, hipMemcpyAsync(&hk, dk, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hr, dr, 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (key) *key = hk;
    if (rates6) memcpy(rates6, hr, 48);
    return KMC_OK;
}

// ---------------------------------------------------------------- tags
int kmc_init_displacements(kmc_ctx* c)
{
    CTXCHK(c);
    const size_t Rn = (size_t)c->R * c->n;
    init_tags_kernel<<<1024, 256, 0, c->stream>>>(c->occ, c->ny, c->n, Rn, c->tags);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemsetAsync(c->time_total, 0, c->R * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->step_time, 0, c->R * 8, c->stream));
    double* h = (double*)malloc((size_t)c->R * 5 * 8);
    for (int r = 0; r < c->R; ++r) { h[r * 5] = NAN; h[r * 5 + 1] = 0; h[r * 5 + 2] = 1e300; h[r * 5 + 3] = -1e300; h[r * 5 + 4] = 0; }
    hipError_t e = hipMemcpyAsync(c->barrier, h, (size_t)c->R * 5 * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free(h);
    HIPCHK(c, e);
    return KMC_OK;
}

int kmc_set_displacements(kmc_ctx* c, const int32_t* tags)
{
    CTXCHK(c);
    if (!tags) return fail(c, KMC_EINVAL, "kmc_set_displacements: null pointer");
    HIPCHK(c, hipMemcpyAsync(c->tags, tags, (size_t)c->R * c->n * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_get_displacements(kmc_ctx* c, int32_t* tags)
{
    CTXCHK(c);
    if (!tags) return fail(c, KMC_EINVAL, "kmc_get_displacements: null pointer");
    HIPCHK(c, hipMemcpyAsync(tags, c->tags, (size_t)c->R * c->n * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

// ---------------------------------------------------------------- hop loop
int kmc_set_debug(kmc_ctx* c, int force_exact, double margin_scale)
{
    CTXCHK(c);
    if (!(margin_scale >= 1.0)) return fail(c, KMC_EINVAL, "kmc_set_debug: margin_scale must be >= 1");
    c->force_exact = force_exact == 2 ? 2 : (force_exact != 0);
    c->margin_scale = margin_scale;
    return KMC_OK;
}

static int run_impl(kmc_ctx* c, int64_t n_hops, const int64_t* counts, int presim, const double* draws, uint64_t seed,
                    int32_t* sel_out)
{
    if (n_hops < 0) return fail(c, KMC_EINVAL, "kmc_run: n_hops < 0");
    if (!c->have_table) return fail(c, KMC_ESTATE, "kmc_run: set a rate table first");
    if (!c->have_keys) return fail(c, KMC_ESTATE, "kmc_run: run kmc_rates_full (or kmc_set_keys) first");
    if (!c->lds_path && !c->use_big && c->nx > 4096)
        return fail(c, KMC_EINVAL, "kmc_run: nx = %d exceeds the 4096 rows the global-memory hop kernel keeps in LDS", c->nx);
    if (n_hops == 0) return KMC_OK;
    // (decided before anything is staged: building the palette uses the context's small scratch buffer)
    // (an explicit request for launch groups, kmc_set_launch_groups(ctx, 2), keeps the kernels that have them)
    bool use_lat = c->lat_ok && c->lat_mode >= 0 && (c->lat_mode == 1 || (c->R <= KMC_LAT_AUTO_MAX && c->split != 2));
    if (use_lat) {
        if (int rc = ensure_lat_tables(c)) return rc;
        if (c->pal_state != 1) {
            if (c->lat_mode == 1)
                return fail(c, KMC_ESTATE, "kmc_run: the rate table holds more than %d distinct rates: the one-replica-per-CU "
                            "kernel (kmc_set_hop_kernel 5) cannot hold its palette in LDS", LAT_PAL_MAX);
            use_lat = false;
        }
    }
    bool use_sitew = !c->lds_path && c->use_big && c->big_sitew_ok && !c->big_tree;
    if (use_sitew) {
        if (int rc = ensure_lat_tables(c)) return rc;
        use_sitew = c->pal_state == 1;          // more than 1024 distinct rates: the round-2 instances
    }
    const size_t Rh = (size_t)c->R * (size_t)n_hops;
    if (draws) {
        if (c->draws_cap < Rh * 2) {
            if (c->draws) HIPCHK(c, hipFree(c->draws));
            c->draws = nullptr; c->draws_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->draws, Rh * 16));
            c->draws_cap = Rh * 2;
        }
        HIPCHK(c, hipMemcpyAsync(c->draws, draws, Rh * 16, hipMemcpyHostToDevice, c->stream));
    }
    if (sel_out) {
        if (c->sel_cap < Rh) {
            if (c->sel) HIPCHK(c, hipFree(c->sel));
            c->sel = nullptr; c->sel_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->sel, Rh * 4));
            c->sel_cap = Rh;
        }
    }
    long long* dcounts = nullptr;
    if (counts) {
        dcounts = reinterpret_cast<long long*>(c->scratch_d);
        HIPCHK(c, hipMemcpyAsync(dcounts, counts, (size_t)c->R * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    // 6n: the reference's sequential cumsum; 6n / 8192: np.sum adds its buffers left to right; 512: the
    // roundings of the kernels' own hierarchical sums (<= 70), np.sum inside a buffer (<= 30), fl(rho * gamma),
    // the 2u of fl(c / gamma) >= rho
    const double mfac = (6.0 * c->n * (1.0 + 1.0 / 8192.0) + 512.0) * ldexp(1.0, -53) * 1.001 * c->margin_scale;
    if (!c->lds_path && c->use_big) {
        if (!c->hier) {
            const size_t bytes = (size_t)c->R * c->geom.hier_doubles * 8;
            HIPCHK(c, hipMalloc((void**)&c->hier, bytes));
            HIPCHK(c, hipMemsetAsync(c->hier, 0, bytes, c->stream));
        }
        HopBigParams g;
        memset(&g, 0, sizeof(g));
        g.g = c->geom; g.window_mode = c->window_mode;
        g.n_hops = n_hops; g.n_hops_r = dcounts; g.presim = presim != 0; g.force_exact = c->force_exact;
        g.mfac = mfac; g.mu = ldexp(1.0, -53) * 1.001 * c->margin_scale;
        g.table = c->table; g.wsum = c->wsum; g.tcompact = c->tcompact; g.occ = c->occ; g.keys = c->keys; g.hier = c->hier;
        g.tags = c->tags;
        if (use_sitew) {
            if (!c->sw) HIPCHK(c, hipMalloc((void**)&c->sw, (size_t)c->R * c->n * sizeof(LatWP)));
            g.wp = c->wp; g.pal = c->pal; g.sw = c->sw;
        }
        g.draws = draws ? c->draws : nullptr; g.sel = sel_out ? c->sel : nullptr;
        g.seed_lo = (uint32_t)seed; g.seed_hi = (uint32_t)(seed >> 32); g.rid_base = c->rid_base;
        g.time_total = c->time_total; g.step_time = c->step_time; g.gamma = c->gamma; g.hist = c->hist;
        g.hops_done = c->hops_done; g.n_exact = c->n_exact; g.status = c->status; g.barrier = c->barrier;
#ifdef KMC_STAMPS
        g.stamps = dcounts ? nullptr : reinterpret_cast<long long*>(c->scratch_d);
#endif
        const int slot = (int)(c->run_launches % KMC_EV_RING);
        time_begin(c);
        c->run_groups[slot] = 1;
        HIPCHK(c, hipEventRecord(c->run_ev[slot][0][0], c->stream));
        const bool three = (c->ny & 7) != 0;                 // L1 blocks a window row can touch
        if (use_sitew && three)
            hop_big_kernel<false, 3, true><<<c->R, KMC_BIG_THREADS, big_sitew_lds_bytes(c->geom), c->stream>>>(g);
        else if (use_sitew)
            hop_big_kernel<false, 2, true><<<c->R, KMC_BIG_THREADS, big_sitew_lds_bytes(c->geom), c->stream>>>(g);
        else if (c->have_compact && three)
            hop_big_kernel<true, 3><<<c->R, KMC_BIG_THREADS, c->geom.lds_bytes, c->stream>>>(g);
        else if (c->have_compact)
            hop_big_kernel<true, 2><<<c->R, KMC_BIG_THREADS, c->geom.lds_bytes, c->stream>>>(g);
        else if (three)
            hop_big_kernel<false, 3><<<c->R, KMC_BIG_THREADS, c->geom.lds_bytes, c->stream>>>(g);
        else
            hop_big_kernel<false, 2><<<c->R, KMC_BIG_THREADS, c->geom.lds_bytes, c->stream>>>(g);
        HIPCHK(c, hipEventRecord(c->run_ev[slot][0][1], c->stream));
        time_end(c);
        ++c->run_launches;
        HIPCHK(c, hipGetLastError());
        if (sel_out) {
            HIPCHK(c, hipMemcpyAsync(sel_out, c->sel, Rh * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (draws) HIPCHK(c, hipStreamSynchronize(c->stream));
        return KMC_OK;
    }
    if (!c->lds_path) {
        if (!c->w_g) HIPCHK(c, hipMalloc((void**)&c->w_g, (size_t)c->R * c->n * 8));
        HopGlobalParams g;
        memset(&g, 0, sizeof(g));
        g.nx = c->nx; g.ny = c->ny; g.n = c->n; g.window_mode = c->window_mode;
        g.n_hops = n_hops; g.n_hops_r = dcounts; g.presim = presim != 0; g.force_exact = c->force_exact;
        g.mfac = mfac; g.mu = ldexp(1.0, -53) * 1.001 * c->margin_scale;
        g.table = c->table; g.wsum = c->wsum; g.occ = c->occ; g.keys = c->keys; g.w = c->w_g; g.tags = c->tags;
        g.draws = draws ? c->draws : nullptr; g.sel = sel_out ? c->sel : nullptr;
        g.seed_lo = (uint32_t)seed; g.seed_hi = (uint32_t)(seed >> 32); g.rid_base = c->rid_base;
        g.time_total = c->time_total; g.step_time = c->step_time; g.gamma = c->gamma; g.hist = c->hist;
        g.hops_done = c->hops_done; g.n_exact = c->n_exact; g.status = c->status; g.barrier = c->barrier;
        const size_t lds = ((sizeof(GbShared) + 15) & ~(size_t)15) + (size_t)c->nx * 8;
        const int slot = (int)(c->run_launches % KMC_EV_RING);
        time_begin(c);
        c->run_groups[slot] = 1;
        HIPCHK(c, hipEventRecord(c->run_ev[slot][0][0], c->stream));
        hop_global_kernel<<<c->R, KMC_GB_THREADS, lds, c->stream>>>(g);
        HIPCHK(c, hipEventRecord(c->run_ev[slot][0][1], c->stream));
        time_end(c);
        ++c->run_launches;
        HIPCHK(c, hipGetLastError());
        if (sel_out) {
            HIPCHK(c, hipMemcpyAsync(sel_out, c->sel, Rh * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (draws) HIPCHK(c, hipStreamSynchronize(c->stream));
        return KMC_OK;
    }
    HopParams p;
    memset(&p, 0, sizeof(p));
    p.nx = c->nx; p.ny = c->ny; p.n = c->n; p.window_mode = c->window_mode;
    p.n_hops = n_hops; p.n_hops_r = dcounts; p.presim = presim != 0; p.force_exact = c->force_exact;
    // rigorous rounding bound of the certified selection, see kmc_hop_lds.h / DESIGN.md
    p.mfac = mfac;
    p.table = c->table; p.wsum = c->wsum; p.occ = c->occ; p.keys = c->keys; p.tags = c->tags;
    p.draws = draws ? c->draws : nullptr; p.sel = sel_out ? c->sel : nullptr;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32); p.rid_base = c->rid_base;
    p.time_total = c->time_total; p.step_time = c->step_time; p.gamma = c->gamma; p.hist = c->hist;
    p.hops_done = c->hops_done; p.n_exact = c->n_exact; p.status = c->status; p.barrier = c->barrier;
#ifdef KMC_STAMPS
    p.stamps = dcounts ? nullptr : reinterpret_cast<long long*>(c->scratch_d);
#endif
    const int slot = (int)(c->run_launches % KMC_EV_RING);
    p.tcompact = c->tcompact;
    p.xscratch = c->xscratch;
    p.tcompact2 = c->tcompact2;
    // More replicas than 3/4 of the workgroups the chip holds at once: two launch groups (see kmc_ctx).  Resident
    // workgroups per CU: 16 (hop_q64_kernel), 8 (hop_lds64_kernel); the any-size kernel runs 3 waves per SIMD.
    const size_t gen_lds = host_hop_lds_bytes(c->nx, c->n);
    // 16 replicas fit a CU and there are more than the 3-wave build holds at once (12 per CU): the 4-waves-per-SIMD
    // build; below that its spills only cost (a single 30 x 30 lattice: -4 %)
    const bool gen_w4 = gen_lds * 16 <= 160 * 1024 && c->R > 256 * 12;
    const int per_cu = c->qany ? 16 : c->lds64 ? (c->q64 ? 16 : 8) : (gen_w4 ? 16 : (gen_lds * 12 <= 160 * 1024 ? 12 : (int)((160 * 1024) / gen_lds)));
    const bool split = c->R >= 2 * 8 && (c->split == 2 || (c->split == 0 && c->R > 192 * per_cu));
    const int r_half = ((c->R / 2 + 7) / 8) * 8;             // multiples of 8: blocks go round the 8 XCDs
    if (counts && c->R > 1) {
        // Per-replica hop counts (a sweep: int(n x) hops per step, a factor 9 between x = 0.1 and 0.9): workgroups are
        // dispatched in block order and a launch lasts as long as its last replica, so the longest runs go first and
        // the short ones fill the tail.  With two launch groups each group orders ITS OWN replicas: a replica must stay
        // on its group's stream, because consecutive runs of different groups are not ordered against each other.
        int* h = (int*)host_pool(c, (size_t)c->R * sizeof(int));
        if (!h) return fail(c, KMC_EINVAL, "kmc_run_counts: out of host memory");
        for (int r = 0; r < c->R; ++r) h[r] = r;
        auto longer = [&](int a, int b) { return counts[a] > counts[b]; };
        if (split) {
            std::stable_sort(h, h + r_half, longer);
            std::stable_sort(h + r_half, h + c->R, longer);
        } else {
            std::stable_sort(h, h + c->R, longer);
        }
        int* dorder = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(c->scratch_d) + (size_t)c->R * 8);
        HIPCHK(c, hipMemcpyAsync(dorder, h, (size_t)c->R * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        p.order = dorder;
    }
    if (use_lat) {
        LatParams lp;
        lp.h = p; lp.wp = c->wp; lp.pal = c->pal;
        c->run_groups[slot] = 1;
        time_begin(c);
        HIPCHK(c, hipEventRecord(c->run_ev[slot][0][0], c->stream));
        lp.n_rep = c->R;
        const bool cls = lat_cls_ok(c->ny, c->window_mode);
        if (c->R > KMC_LAT_ONE_PER_CU || c->lat_pk) {
            // more replicas than CUs: two to a CU, one 2-wave workgroup each (kmc_hop_lat.h, PK)
            const int nb = (c->R + 1) / 2;
            if (c->lds64) hop_lat_kernel<false, true, true><<<nb, 128, LAT2_LDS_BYTES, c->stream>>>(lp);
            else if (cls) hop_lat_kernel<true, true, true><<<nb, 128, LAT2_LDS_BYTES, c->stream>>>(lp);
            else hop_lat_kernel<true, false, true><<<nb, 128, LAT2_LDS_BYTES, c->stream>>>(lp);
        } else {
            if (c->lds64) hop_lat_kernel<false, true, false><<<c->R, 64, LAT_LDS_BYTES, c->stream>>>(lp);
            else if (cls) hop_lat_kernel<true, true, false><<<c->R, 64, LAT_LDS_BYTES, c->stream>>>(lp);
            else hop_lat_kernel<true, false, false><<<c->R, 64, LAT_LDS_BYTES, c->stream>>>(lp);
        }
        HIPCHK(c, hipEventRecord(c->run_ev[slot][0][1], c->stream));
        time_end(c);
        ++c->run_launches;
        HIPCHK(c, hipGetLastError());
        if (sel_out) {
            HIPCHK(c, hipMemcpyAsync(sel_out, c->sel, Rh * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (draws) HIPCHK(c, hipStreamSynchronize(c->stream));
        return KMC_OK;
    }
    if (split) {
        if (c->need_fork) {
            HIPCHK(c, hipEventRecord(c->fork_ev, c->stream));
            for (int g = 0; g < KMC_GROUPS; ++g) HIPCHK(c, hipStreamWaitEvent(c->gstream[g], c->fork_ev, 0));
            c->need_fork = false;
        }
        c->run_groups[slot] = KMC_GROUPS;
        for (int g = 0; g < KMC_GROUPS; ++g) {
            p.r_base = g == 0 ? 0 : r_half;
            const int rg = g == 0 ? r_half : c->R - r_half;
            hipStream_t gs = c->gstream[g];
            if (g == 0) hipEventRecord(c->ev0, gs);
            HIPCHK(c, hipEventRecord(c->run_ev[slot][g][0], gs));
            if (c->qany && c->have_compact)
                hop_q64_kernel<true, true><<<rg, 64, Q64_LDS_BYTES, gs>>>(p);
            else if (c->qany)
                hop_q64_kernel<false, true><<<rg, 64, Q64_LDS_BYTES, gs>>>(p);
            else if (c->lds64 && c->q64 && c->have_compact)
                hop_q64_kernel<true><<<rg, 64, Q64_LDS_BYTES, gs>>>(p);
            else if (c->lds64 && c->q64)
                hop_q64_kernel<false><<<rg, 64, Q64_LDS_BYTES, gs>>>(p);
            else if (c->lds64 && c->have_compact)
                hop_lds64_kernel<true><<<rg, 64, K64_LDS_BYTES, gs>>>(p);
            else if (c->lds64)
                hop_lds64_kernel<false><<<rg, 64, K64_LDS_BYTES, gs>>>(p);
            else if (gen_w4)
                hop_lds_kernel_w4<<<rg, 64, gen_lds, gs>>>(p);
            else
                hop_lds_kernel<<<rg, 64, gen_lds, gs>>>(p);
            HIPCHK(c, hipEventRecord(c->run_ev[slot][g][1], gs));
            if (g == 0) { hipEventRecord(c->ev1, gs); c->timed = true; c->timed_slot = slot; }
        }
        c->groups_busy = true;
        ++c->run_launches;
        HIPCHK(c, hipGetLastError());
        if (sel_out || draws) {
            HIPCHK(c, join_groups(c));
            if (sel_out) HIPCHK(c, hipMemcpyAsync(sel_out, c->sel, Rh * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        return KMC_OK;
    }
    c->run_groups[slot] = 1;
    time_begin(c);
    HIPCHK(c, hipEventRecord(c->run_ev[slot][0][0], c->stream));
#ifdef Q64_FORCE_DENSE          // scripts/variants.py: A/B of the pair tables against the dense table
    const bool q_compact = false;
#else
    const bool q_compact = c->have_compact;
#endif
    if (c->qany && q_compact)
        hop_q64_kernel<true, true><<<c->R, 64, Q64_LDS_BYTES, c->stream>>>(p);
    else if (c->qany)
        hop_q64_kernel<false, true><<<c->R, 64, Q64_LDS_BYTES, c->stream>>>(p);
    else if (c->lds64 && c->q64 && q_compact)
        hop_q64_kernel<true><<<c->R, 64, Q64_LDS_BYTES, c->stream>>>(p);
    else if (c->lds64 && c->q64)
        hop_q64_kernel<false><<<c->R, 64, Q64_LDS_BYTES, c->stream>>>(p);
    else if (c->lds64 && c->have_compact)
        hop_lds64_kernel<true><<<c->R, 64, K64_LDS_BYTES, c->stream>>>(p);
    else if (c->lds64)
        hop_lds64_kernel<false><<<c->R, 64, K64_LDS_BYTES, c->stream>>>(p);
    else if (gen_w4)
        hop_lds_kernel_w4<<<c->R, 64, gen_lds, c->stream>>>(p);
    else
        hop_lds_kernel<<<c->R, 64, gen_lds, c->stream>>>(p);
    HIPCHK(c, hipEventRecord(c->run_ev[slot][0][1], c->stream));
    time_end(c);
    ++c->run_launches;
    HIPCHK(c, hipGetLastError());
    if (sel_out) {
        HIPCHK(c, hipMemcpyAsync(sel_out, c->sel, Rh * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (draws) HIPCHK(c, hipStreamSynchronize(c->stream));   // the caller may reuse its buffer
    return KMC_OK;
}

int kmc_debug_stamps(kmc_ctx* c, int64_t* out)
{
    CTXCHK(c);
#ifdef KMC_STAMPS
    HIPCHK(c, hipMemcpyAsync(out, c->scratch_d, (size_t)c->R * 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
#else
    (void)out;
    return fail(c, KMC_ESTATE, "kmc_debug_stamps: this library was not built with -DKMC_STAMPS");
#endif
}

int kmc_run(kmc_ctx* c, int64_t n_hops, int presim, const double* draws, uint64_t seed, int32_t* sel_out)
{
    CTXCHK_RUN(c);
    if (draws || sel_out) HIPCHK(c, join_groups(c));         // staging copies go through the context's stream
    return run_impl(c, n_hops, nullptr, presim, draws, seed, sel_out);
}

int kmc_run_counts(kmc_ctx* c, const int64_t* n_hops, int presim, uint64_t seed)
{
    CTXCHK(c);
    if (!n_hops) return fail(c, KMC_EINVAL, "kmc_run_counts: null pointer");
    int64_t mx = 0;
    for (int r = 0; r < c->R; ++r) {
        if (n_hops[r] < 0) return fail(c, KMC_EINVAL, "kmc_run_counts: negative hop count for replica %d", r);
        if (n_hops[r] > mx) mx = n_hops[r];
    }
    return run_impl(c, mx, n_hops, presim, nullptr, seed, nullptr);
}

int kmc_get_scalars(kmc_ctx* c, double* time, double* step_time, double* gamma, int64_t* hist)
{
    CTXCHK(c);
    if (time) HIPCHK(c, hipMemcpyAsync(time, c->time_total, c->R * 8, hipMemcpyDeviceToHost, c->stream));
    if (step_time) HIPCHK(c, hipMemcpyAsync(step_time, c->step_time, c->R * 8, hipMemcpyDeviceToHost, c->stream));
    if (gamma) HIPCHK(c, hipMemcpyAsync(gamma, c->gamma, c->R * 8, hipMemcpyDeviceToHost, c->stream));
    if (hist) HIPCHK(c, hipMemcpyAsync(hist, c->hist, c->R * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_set_time(kmc_ctx* c, const double* time)
{
    CTXCHK(c);
    if (!time) return fail(c, KMC_EINVAL, "kmc_set_time: null pointer");
    HIPCHK(c, hipMemcpyAsync(c->time_total, time, c->R * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

// Resuming a checkpoint: the hops each replica has done (= its position in the Philox stream of kmc_run without
// injected draws) and its direction histogram.  Either pointer may be null.
int kmc_set_counters(kmc_ctx* c, const int64_t* hops_done, const int64_t* hist)
{
    CTXCHK(c);
    if (hops_done) {
        for (int r = 0; r < c->R; ++r)
            if (hops_done[r] < 0) return fail(c, KMC_EINVAL, "kmc_set_counters: negative hop count for replica %d", r);
        HIPCHK(c, hipMemcpyAsync(c->hops_done, hops_done, c->R * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (hist) HIPCHK(c, hipMemcpyAsync(c->hist, hist, c->R * 48, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_get_barrier_stats(kmc_ctx* c, double* out)
{
    CTXCHK(c);
    if (!out) return fail(c, KMC_EINVAL, "kmc_get_barrier_stats: null pointer");
    HIPCHK(c, hipMemcpyAsync(out, c->barrier, (size_t)c->R * 40, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_get_status(kmc_ctx* c, int32_t* status, int64_t* hops_done, int64_t* n_exact)
{
    CTXCHK(c);
    if (status) HIPCHK(c, hipMemcpyAsync(status, c->status, c->R * 4, hipMemcpyDeviceToHost, c->stream));
    if (hops_done) HIPCHK(c, hipMemcpyAsync(hops_done, c->hops_done, c->R * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_exact) HIPCHK(c, hipMemcpyAsync(n_exact, c->n_exact, c->R * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_gamma_exact(kmc_ctx* c, double* gamma)
{
    CTXCHK(c);
    if (!gamma) return fail(c, KMC_EINVAL, "kmc_gamma_exact: null pointer");
    if (!c->have_keys || !c->have_table) return fail(c, KMC_ESTATE, "kmc_gamma_exact: needs a table and kmc_rates_full");
    double* d = nullptr;
    if (int rc = dev_pool(c, (size_t)c->R * 8, (void**)&d)) return rc;
    gamma_exact_kernel<<<(c->R + 63) / 64, 64, 0, c->stream>>>(c->keys, c->R, c->n, c->table, d);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(gamma, d, (size_t)c->R * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

// ---------------------------------------------------------------- observables
int kmc_diffusion(kmc_ctx* c, const double* conc, double a, double* dstar, double* dj, double* msd_out)
{
    CTXCHK(c);
    if (!conc || !dstar || !dj) return fail(c, KMC_EINVAL, "kmc_diffusion: null pointer");
    long long* d = nullptr;
    if (int rc0 = dev_pool(c, (size_t)c->R * 32, (void**)&d)) return rc0;
    msd_kernel<<<c->R, 256, 0, c->stream>>>(c->tags, c->nx, c->ny, d);
    long long* h = (long long*)host_pool(c, (size_t)c->R * 40);
    if (!h) return fail(c, KMC_EINVAL, "kmc_diffusion: out of host memory");
    double* t = (double*)(h + (size_t)c->R * 4);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, (size_t)c->R * 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t, c->time_total, (size_t)c->R * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    int rc = KMC_OK;
    if (e != hipSuccess) rc = fail(c, KMC_EHIP, "kmc_diffusion: %s", hipGetErrorString(e));
    for (int r = 0; rc == KMC_OK && r < c->R; ++r) {
        // hex_LiCoO2_kmc.py:591-602 (number_of_atoms is a float)
        const double n_atoms = (double)c->nx * c->ny * 1 * conc[r];
        if (n_atoms == 0) { rc = fail(c, KMC_EINVAL, "concentration is zero, no atoms to move!"); break; }
        const double sq = (double)h[r * 4], si = (double)h[r * 4 + 1], sj = (double)h[r * 4 + 2], cnt = (double)h[r * 4 + 3];
        const double mean_sq = (a * a) * sq / cnt;                         // np.average(np.square(list))
        const double dj2 = (a * a) * fabs(si * si + sj * sj + si * sj);    // d_j_distance**2
        const double pref = 1 / (2 * 2 * t[r] / n_atoms);
        dstar[r] = pref * mean_sq;
        dj[r] = pref * dj2 / n_atoms;
        if (msd_out) { msd_out[r * 4] = sq; msd_out[r * 4 + 1] = si; msd_out[r * 4 + 2] = sj; msd_out[r * 4 + 3] = cnt; }
    }
    return rc;
}

// kmc_diffusion for callers that cannot allocate (code under HIP graph capture, tight serving loops): every
// pointer is DEVICE memory on the context's GPU, the per-replica sums go through the library's static scratch
// array, everything is enqueued on the context's stream and nothing is allocated, freed or waited for.
int kmc_diffusion_noalloc(kmc_ctx* c, const double* conc_dev, double a, double* dstar_dev, double* dj_dev, double* msd_dev)
{
    CTXCHK(c);
    if (!conc_dev || !dstar_dev || !dj_dev) return fail(c, KMC_EINVAL, "kmc_diffusion_noalloc: null pointer");
    if (c->R > KMC_NOALLOC_MAX_REPLICAS)
        return fail(c, KMC_EINVAL, "kmc_diffusion_noalloc: %d replicas exceed the %d the library's static scratch holds; use kmc_diffusion",
                    c->R, KMC_NOALLOC_MAX_REPLICAS);
    long long* part = c->noalloc_scratch;
    if (!part) return fail(c, KMC_ESTATE, "kmc_diffusion_noalloc: the scratch array's device address is unknown");
    // every call on this GPU shares the array and finds the last call's sums in it: msd_kernel overwrites all of
    // [0, R) x 4, the part this call reads, so nothing needs clearing
    msd_kernel<<<c->R, 256, 0, c->stream>>>(c->tags, c->nx, c->ny, part);
    diffusion_finalize_kernel<<<(c->R + 255) / 256, 256, 0, c->stream>>>(part, c->time_total, conc_dev, c->R, c->nx, c->ny, a,
                                                                         dstar_dev, dj_dev, msd_dev);
    HIPCHK(c, hipGetLastError());
    return KMC_OK;
}

// ---------------------------------------------------------------- the one collective
// RCCL is opened at run time, so that a process that never reduces across GPUs (one GPU, or torch.distributed as
// the transport) does not load the 570 MB library.  It must be the copy built against THIS library's HIP runtime
// (the engine's stream is handed to it): the librccl.so.1 that lies beside the libamdhip64 we are linked to --
// not whichever copy the process may hold already (torch ships its own RCCL over its own HIP runtime) -- and it is
// opened RTLD_LOCAL, so that the two copies never see each other's symbols.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    char err[256];
    bool tried, ok;
};
static RcclApi g_rccl;

static const RcclApi* rccl_api()
{
    RcclApi& a = g_rccl;
    if (a.tried) return a.ok ? &a : nullptr;
    a.tried = true;
    void* h = nullptr;
    Dl_info info;
    char path[4096];
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
        const char* slash = strrchr(info.dli_fname, '/');
        if (slash && (size_t)(slash - info.dli_fname) + 32 < sizeof path) {
            const size_t dir = (size_t)(slash - info.dli_fname) + 1;
            memcpy(path, info.dli_fname, dir);
            strcpy(path + dir, "librccl.so.1");
            h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        }
    }
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) { snprintf(a.err, sizeof a.err, "cannot open the librccl.so.1 of this library's ROCm: %s", dlerror()); return nullptr; }
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
    if (!a.ok) snprintf(a.err, sizeof a.err, "librccl lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce");
    return a.ok ? &a : nullptr;
}

#define RCCLCHK(ctx, api, call)                                                                 \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) return fail(ctx, KMC_EHIP, "%s failed: %s", #call, (api)->GetErrorString(r_)); \
    } while (0)

int kmc_comm_unique_id(void* id128)
{
    if (!id128) return fail(nullptr, KMC_EINVAL, "kmc_comm_unique_id: null pointer");
    const RcclApi* a = rccl_api();
    if (!a) return fail(nullptr, KMC_EHIP, "kmc_comm_unique_id: %s", g_rccl.err);
    ncclUniqueId id;
    RCCLCHK(nullptr, a, a->GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return KMC_OK;
}

int kmc_comm_create(kmc_ctx* c, const void* id128, int nranks, int rank, void** comm_out)
{
    CTXCHK(c);
    if (!id128 || !comm_out || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(c, KMC_EINVAL, "kmc_comm_create: need an id, 0 <= rank < nranks");
    const RcclApi* a = rccl_api();
    if (!a) return fail(c, KMC_EHIP, "kmc_comm_create: %s", g_rccl.err);
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCLCHK(c, a, a->CommInitRank(&comm, nranks, id, rank));      // binds the communicator to the context's device (current)
    *comm_out = comm;
    return KMC_OK;
}

int kmc_comm_destroy(kmc_ctx* c, void* comm)
{
    CTXCHK(c);
    if (!comm) return KMC_OK;
    const RcclApi* a = rccl_api();
    if (!a) return fail(c, KMC_EHIP, "kmc_comm_destroy: %s", g_rccl.err);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    RCCLCHK(c, a, a->CommDestroy((ncclComm_t)comm));
    return KMC_OK;
}

// Sum of acc[0 .. n) over the ranks of rccl_comm, in place: the mean / np.std accumulators of the ensemble loop
// (hex_LiCoO2_kmc.py:783-786).  One ncclAllReduce(sum, f64) on the context's stream; ~1 KB, latency bound.
int kmc_allreduce_stats(kmc_ctx* c, void* rccl_comm, double* acc, int n)
{
    CTXCHK(c);
    if (!rccl_comm || !acc || n < 1 || n > (1 << 24)) return fail(c, KMC_EINVAL, "kmc_allreduce_stats: need a communicator, a buffer and 1 <= n <= 2^24");
    const RcclApi* a = rccl_api();
    if (!a) return fail(c, KMC_EHIP, "kmc_allreduce_stats: %s", g_rccl.err);
    if (c->coll_cap < (size_t)n) {
        if (c->coll_buf) HIPCHK(c, hipFree(c->coll_buf));
        c->coll_buf = nullptr; c->coll_cap = 0;
        const size_t cap = (size_t)n < 4096 ? 4096 : (size_t)n;
        HIPCHK(c, hipMalloc((void**)&c->coll_buf, cap * 8));
        c->coll_cap = cap;
    }
    HIPCHK(c, hipMemcpyAsync(c->coll_buf, acc, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(c, a, a->AllReduce(c->coll_buf, c->coll_buf, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)rccl_comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(acc, c->coll_buf, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

int kmc_get_distances(kmc_ctx* c, double a, double* matrix)
{
    CTXCHK(c);
    if (!matrix) return fail(c, KMC_EINVAL, "kmc_get_distances: null pointer");
    const size_t Rn = (size_t)c->R * c->n;
    double* d = nullptr;
    if (int rc = dev_pool(c, Rn * 8, (void**)&d)) return rc;
    distance_kernel<<<1024, 256, 0, c->stream>>>(c->tags, c->nx, c->ny, Rn, a, d);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(matrix, d, Rn * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dev_pool_trim(c);
    return KMC_OK;
}

}  // extern "C"
