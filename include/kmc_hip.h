/* kmc_hip.h -- C ABI of the MI355X (gfx950) hex-lattice KMC engine.
 *
 * The reference (quentinstoyel/KMC) has no FFI: its boundary is a set of Python
 * functions over module globals.  Each entry point below names the reference
 * function / lines it replaces (file:line in the reference tree).  The Python
 * host (kmc_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - One opaque context per GPU; a context owns all device memory, is bound to
 *     one device and one HIP stream, and is single-owner (not thread-safe).
 *   - The caller owns every host buffer (C-contiguous, exact dtypes below).
 *   - Every function returns 0 on success or a negative KMC_E* code; the text of
 *     the last error is kmc_last_error(ctx).  Nothing throws across the ABI.
 *   - There is NO CPU backend: kmc_create fails if the device is not usable.
 *   - Replica r, site (i, j): flat site index s = i*ny + j; flat selection index
 *     idx = s*6 + dir (hex_LiCoO2_kmc.py:494, :500-501).
 *   - Occupancy 1 = Li, 0 = vacancy (reference: -1.0 / +1.0, hex_LiCoO2_kmc.py:25).
 */
#ifndef KMC_HIP_H
#define KMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kmc_ctx kmc_ctx;

#define KMC_OK            0
#define KMC_EINVAL       -1   /* bad argument / shape */
#define KMC_EHIP         -2   /* a HIP runtime call failed */
#define KMC_ESTATE       -3   /* call order (e.g. run before a table was set) */

#define KMC_VACANT_KEY   0xFFFFFFFFu

/* refresh window after a hop */
#define KMC_WINDOW_HEX_REF 0  /* hex_LiCoO2_kmc.py:446-449 (half-open, leaves stale rates) */
#define KMC_WINDOW_OBJ_REF 1  /* Mc_Lattice.py:178 (5x6 sites around the start point)       */
#define KMC_WINDOW_FULL    2  /* every affected site (no stale rates)                        */

/* per-replica status bits (kmc_get_status) */
#define KMC_ST_OVERFLOW   1   /* rho above cumsum[-1]/gamma: reference raises IndexError     */
#define KMC_ST_ZERO_GAMMA 2   /* no hop possible (np.sum(P) == 0)                            */

/* ---- life cycle ------------------------------------------------------- */
/* n_replicas independent nx*ny lattices (the loops hex_LiCoO2_kmc.py:679-693). */
int kmc_create(kmc_ctx** out, int device, int n_replicas, int nx, int ny, int window_mode);
int kmc_destroy(kmc_ctx* ctx);
const char* kmc_last_error(kmc_ctx* ctx);       /* ctx may be NULL: last create error */
int kmc_sync(kmc_ctx* ctx);
/* device time (ms, HIP events on the context's stream) of the last kmc_run /
 * kmc_rates_full / kmc_set_base_table launch */
int kmc_last_kernel_ms(kmc_ctx* ctx, float* ms);
/* device time (ms) of each of the last (up to 64) hop-kernel launches of kmc_run /
 * kmc_run_counts, oldest first: HIP events recorded on the context's stream
 * around every launch.  Synchronises the stream. */
int kmc_run_kernel_times(kmc_ctx* ctx, float* ms, int max_n, int* n_out);
/* device time (ms, HIP events) from the start of the n_runs-th last run's first kernel to the end of the
 * last run's last kernel, and in *groups_out (may be NULL) how many concurrent kernels the last run launched
 * (see kmc_set_launch_groups); n_runs <= 64.  With launch groups kmc_run_kernel_times reports the first
 * group's kernel of each run.  No counterpart in the reference. */
int kmc_run_span_ms(kmc_ctx* ctx, int n_runs, float* span_ms, int* groups_out);

/* ---- rate tables (row T) --------------------------------------------- */
/* Folds a 512-entry base table into the 2^18 x 6 device table on the GPU:
 * replaces hex_dict_builder.py:38-142.  base512 index = the nine free chars of
 * the 16-char key (positions 0,1,4,6,8,11,13,14,15) MSB first. */
int kmc_set_base_table(kmc_ctx* ctx, const double* base512, double dumbbell_rate);
/* A prebuilt dense table (hex_dict.p contents), row = int(key, 2). */
int kmc_set_hex_table(kmc_ctx* ctx, const double* table /* [262144][6] */);
int kmc_get_hex_table(kmc_ctx* ctx, double* table /* [262144][6] */);
/* Row G: base table from the local cluster expansion,
 * hex_LiCoO2_kmc.py:103-199 + :402-403 for all 512 keys (dictionary_builder :345-365). */
int kmc_generate_base_table(kmc_ctx* ctx, const double* eci47, double prefactor, double kb_t,
                            double* base512_out, double* energies_out /* or NULL */);

/* ---- lattice state ----------------------------------------------------- */
int kmc_set_lattice(kmc_ctx* ctx, const int8_t* occ /* [R][nx][ny] */);
int kmc_get_lattice(kmc_ctx* ctx, int8_t* occ);
/* exactly int(nx*ny*conc[r]) Li per replica, uniformly without replacement,
 * Philox4x32-10 keyed by (seed, replica): the fill loop hex_LiCoO2_kmc.py:703-714 */
int kmc_fill_random(kmc_ctx* ctx, const double* conc /* [R] */, uint64_t seed);

/* The ensemble loops hex_LiCoO2_kmc.py:679-693 sharded over several contexts (GPUs): this context's replica 0 is
 * replica `offset` of the whole ensemble.  kmc_fill_random and the Philox draws of kmc_run / kmc_run_counts are keyed
 * by (seed, offset + r, counter), so with ONE seed the ensemble is the same at any GPU count (two contexts of R/2 with
 * offsets 0 and R/2 reproduce one context of R bit for bit).  Default 0; offset + R must fit 32 bits. */
int kmc_set_replica_offset(kmc_ctx* ctx, uint64_t offset);

/* ---- rates (rows R1-R3) ------------------------------------------------ */
/* kmc_evolve(..., use_initialize_mode=True), hex_LiCoO2_kmc.py:435-443, :453-465,
 * for every replica: 18-bit keys (get_hex_probability :418-428) + rate gather. */
int kmc_rates_full(kmc_ctx* ctx);
/* key at last refresh per site, KMC_VACANT_KEY where the site held no Li then */
int kmc_get_keys(kmc_ctx* ctx, uint32_t* keys /* [R][nx][ny] */);
/* restore (possibly stale) rate state, e.g. from a checkpoint */
int kmc_set_keys(kmc_ctx* ctx, const uint32_t* keys);
/* the rate matrix hop_probability_matrix, hex_LiCoO2_kmc.py:695-696 */
int kmc_get_rates(kmc_ctx* ctx, double* P /* [R][nx][ny][6] */);
/* kmc_evolve(..., use_initialize_mode=False), hex_LiCoO2_kmc.py:444-465: one
 * window per replica; se = i_s, j_s, i_e, j_e with the end point unwrapped.
 * A replica whose i_s is negative is skipped. */
int kmc_refresh_window(kmc_ctx* ctx, const int32_t* se /* [R][4] */);
/* get_hex_probability([i, j, 0]) on the current lattice of one replica
 * (hex_LiCoO2_kmc.py:418-428): fresh key and its six rates, whatever the site holds. */
int kmc_site_probability(kmc_ctx* ctx, int replica, int i, int j, uint32_t* key, double* rates6);

/* ---- displacement tags (row U2) --------------------------------------- */
/* distance_lattice initialisation, hex_LiCoO2_kmc.py:734-740; also zeroes the clocks */
int kmc_init_displacements(kmc_ctx* ctx);
/* tags = i0, j0, wi, wj per site; vacant = -1,0,0,0.  The reference packs
 * w = 100*wi + wj (hex_LiCoO2_kmc.py:525-526); the Python shim does the packing. */
int kmc_set_displacements(kmc_ctx* ctx, const int32_t* tags /* [R][nx][ny][4] */);
int kmc_get_displacements(kmc_ctx* ctx, int32_t* tags);

/* ---- the hop loop (rows S1-S5, U1-U4, K) ------------------------------ */
/* n_hops iterations of hex_LiCoO2_kmc.py:484-535 in every replica.
 *   presim  != 0: no tag / clock update (use_presimulation_mode, :521)
 *   draws   [R][n_hops][2] = rho, clock draw (injected uniforms: parity mode), or
 *           NULL: counter-based Philox4x32-10 keyed by (seed, replica, hop number)
 *   sel_out [R][n_hops] hop_master_index trace (:491), or NULL; -1 past a stop
 * The launch is asynchronous (unless draws / sel_out force a copy).  A replica that
 * hits the reference's IndexError (rho above cumsum[-1]/gamma) or has no possible
 * hop stops there with its state consistent; see kmc_get_status. */
int kmc_run(kmc_ctx* ctx, int64_t n_hops, int presim, const double* draws, uint64_t seed,
            int32_t* sel_out);
/* Same loop with a hop count per replica (Philox stream only): one reference
 * "KMC step" is int(nx*ny*x) hops (:484), which differs between the
 * concentrations of a sweep (hex_LiCoO2_kmc.py:686). */
int kmc_run_counts(kmc_ctx* ctx, const int64_t* n_hops /* [R] */, int presim, uint64_t seed);
/* time [R]: total_time (:754); step_time [R]: clock advance of the last kmc_run
 * (time_step_per_kmcstep :531); gamma [R]: rate sum after the last hop;
 * hist [R][6]: hop_histogram (:535).  Any pointer may be NULL. */
int kmc_get_scalars(kmc_ctx* ctx, double* time, double* step_time, double* gamma, int64_t* hist);
int kmc_set_time(kmc_ctx* ctx, const double* time /* [R] */);
/* Resuming a saved state (the reference's state is the arrays kmc_step takes and returns, :470, :539: lattice,
 * rate matrix = keys here, distance_lattice, total_time, hop_histogram): hops_done [R] = each replica's position in
 * the Philox stream of kmc_run without injected draws, hist [R][6].  Either pointer may be NULL.  With
 * kmc_set_lattice, kmc_set_keys, kmc_set_displacements and kmc_set_time a run continues bit for bit. */
int kmc_set_counters(kmc_ctx* ctx, const int64_t* hops_done, const int64_t* hist);
/* chosen_barrier (:495-498): last, and sum / min / max / count over the hops of
 * all kmc_run calls since kmc_init_displacements; out = [R][5] */
int kmc_get_barrier_stats(kmc_ctx* ctx, double* out);
/* status bits, hops executed so far, number of hops resolved by the exact
 * sequential path; any pointer may be NULL */
int kmc_get_status(kmc_ctx* ctx, int32_t* status, int64_t* hops_done, int64_t* n_exact);
/* np.sum(hop_probability_matrix) with numpy's own association (:486), per replica */
int kmc_gamma_exact(kmc_ctx* ctx, double* gamma /* [R] */);
/* testing knobs: force_exact != 0 resolves every hop by the exact sequential
 * path (2: the large-lattice kernel may use the replay's short form, which skips
 * the np.sum pass when the index does not depend on it -- what it does for the
 * hops it cannot certify; 1: always the long form); margin_scale multiplies the
 * certification margin (default 1). */
int kmc_set_debug(kmc_ctx* ctx, int force_exact, double margin_scale);

/* ---- observables (rows D1, D2) ----------------------------------------- */
/* get_diffusion_coefficient, hex_LiCoO2_kmc.py:589-603 (with get_distance :558-586)
 * for every replica, using the context's clocks.  msd_out [R][4] (or NULL) =
 * sum r^2, sum di, sum dj, n_atoms in lattice units (exact integers). */
int kmc_diffusion(kmc_ctx* ctx, const double* conc /* [R] */, double lattice_constant,
                  double* dstar /* [R] */, double* dj /* [R] */, double* msd_out);

/* kmc_diffusion for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code under HIP
 * graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same result as kmc_diffusion, but
 *   - conc_dev, dstar_dev, dj_dev, msd_dev (or NULL) are DEVICE pointers on the context's GPU ([R], [R], [R], [R][4]);
 *   - the per-replica sums go through a static device array of the library (2 MiB per GPU, held until the
 *     process ends) instead of a buffer allocated per call: at most 65536 replicas, more is KMC_EINVAL;
 *   - the call only enqueues two kernels on the context's stream: no allocation, no free, no copy, no wait
 *     (kmc_sync, or any stream-ordered reader, before the results are used);
 *   - a replica of concentration 0 gets NaN (kmc_diffusion returns the reference's sys.exit text instead).
 * Every call on one GPU shares that GPU's array: calls from different contexts that overlap on one GPU race. */
int kmc_diffusion_noalloc(kmc_ctx* ctx, const double* conc_dev, double lattice_constant,
                          double* dstar_dev, double* dj_dev, double* msd_dev);

/* ---- the one collective (multi-GPU ensembles) --------------------------------------------------------
 * Replicas (the loops hex_LiCoO2_kmc.py:679-693) are sharded over one context per GPU, one process per GPU; the
 * only exchange is the sum of the per-concentration accumulators behind the reference's mean and np.std
 * (:783-786).  kmc_allreduce_stats sums acc[0 .. n) (host, in place) over the ranks of an RCCL communicator with ONE
 * ncclAllReduce(sum, f64) on the context's stream.  The communicator is the caller's (any ncclComm_t whose device is
 * the context's), or is made here: rank 0 calls kmc_comm_unique_id, the 128 bytes reach the other ranks by the
 * host's own means (torch.distributed broadcast, MPI, a file), every rank calls kmc_comm_create (collective: it
 * returns when all nranks have called).  RCCL is opened with dlopen on first use: the librccl.so.1 of the ROCm this
 * library is linked to (the stream handed to it is this library's HIP runtime's), so a communicator made elsewhere
 * must come from that same copy -- torch's own communicators (torch ships its own RCCL and HIP runtime) are not
 * usable here.  No counterpart in the reference, which is single-process. */
int kmc_comm_unique_id(void* id128 /* 128 bytes out */);
int kmc_comm_create(kmc_ctx* ctx, const void* id128, int nranks, int rank, void** comm_out);
int kmc_comm_destroy(kmc_ctx* ctx, void* comm);
int kmc_allreduce_stats(kmc_ctx* ctx, void* rccl_comm, double* acc, int n);

/* get_distance, hex_LiCoO2_kmc.py:558-586: distances_travelled_matrix * a per site
 * (0 where vacant); the list and the vector sum follow from it and msd_out. */
int kmc_get_distances(kmc_ctx* ctx, double lattice_constant, double* matrix /* [R][nx][ny] */);

/* ---- introspection ------------------------------------------------------ */
/* 1 if the LDS-resident one-wave-per-replica hop kernel serves this context,
 * 0 if the global-memory workgroup kernel does (nx or ny > 64) */
int kmc_uses_lds_path(kmc_ctx* ctx);
/* dynamic LDS bytes per replica of the hop kernel */
int kmc_lds_bytes(kmc_ctx* ctx);
/* 1 if the current rate table factors into the per-direction compact table Tc[6][2^13] (each folded rate
 * reads 13 of the 18 key bits, hex_dict_builder.py:45-139) that the 64 x 64 kernel gathers from; a dense
 * table given to kmc_set_hex_table that does not factor returns 0 and keeps the dense gather.
 * No counterpart in the reference. */
int kmc_table_is_compact(kmc_ctx* ctx);
/* Choice of hop kernel (tests compare them): which = 0 the default for the lattice size and replica count;
 * 1 the generic kernel (any-size LDS kernel, or the global-memory kernel for large lattices); for 64 x 64
 * replicas 2 = the 8-replicas-per-CU kernel (kmc_hop_lds64.h), 3 = the 16-per-CU kernel (kmc_hop_q64.h);
 * the default is 3; 4 = that layout for lattices of 8 ... 64 rows and columns other than 64 x 64 (the default there);
 * 5 = the one-replica-per-CU kernel (kmc_hop_lat.h: rates, weights and fresh keys of every site in LDS), which is
 * also what 0 picks when the context holds at most 512 replicas of 8 ... 64 rows and columns (one replica per CU up to
 * 256, two beyond) and the rate table's distinct values fit its 1024-entry palette; 6 = its two-per-CU layout at any
 * replica count.  Lattices beyond 64 x 64: 2 = the large-lattice kernel (kmc_hop_big.h) without the per-site rate words
 * it uses by default where a row has at most 1024 sites and the table at most 1024 distinct rates.
 * Kernels 3 ... 6 store a vacant site as the all-Li key 0x3FFFF, whose row is zero in every table that describes hops (all
 * six end points occupied); a table whose row 0x3FFFF is not zero runs on the kernels that keep an explicit zero row (0 picks
 * them by itself) and asking for 3 ... 6 returns KMC_ESTATE.  No counterpart in the reference. */
int kmc_set_hop_kernel(kmc_ctx* ctx, int which);
/* Name of the hop kernel the next kmc_run would launch ("hop_q64_kernel", "hop_lat_kernel<any>", ...): what
 * bench.py reports beside its timings.  A static string; never NULL.  No counterpart in the reference. */
const char* kmc_hop_kernel_name(kmc_ctx* ctx);
/* Launch groups of the LDS-resident hop kernels (lattices up to 64 x 64).  mode 2: every run is launched as two
 * kernels over the two halves of the replicas, each on a stream of its own, and consecutive runs are not joined
 * (a group's next launch waits for that group's last one only), so one group's start-up, write-back and tail
 * overlap the other's hop loop; every other entry point joins the groups first, results are those of one
 * launch.  mode 1: one launch.  mode 0 (default): two groups when the replicas exceed 3/4 of the workgroups the
 * chip holds at once (3072 for the default 64 x 64 kernel).  No counterpart in the reference. */
int kmc_set_launch_groups(kmc_ctx* ctx, int mode);
/* profiling builds only (-DKMC_STAMPS): cycles [R][8] the last kmc_run spent per hop segment, measured
 * with s_memtime; any other build returns KMC_ESTATE.  No counterpart in the reference. */
int kmc_debug_stamps(kmc_ctx* ctx, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* KMC_HIP_H */
