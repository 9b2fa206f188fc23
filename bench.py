#!/usr/bin/env python3
"""bench.py -- hops/s of the hex-lattice KMC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts N rank
processes (fresh children, before anything here has touched a GPU), waits for them and exits non-zero if
any of them fails.  Under torch.distributed.run the ranks are already there.

Workload (BASELINE.json configs[4], the one the metric is quoted on): an ensemble of independent 64 x 64 hex
lattices at x = 0.5, random occupancy, rate table = 2^18 x 6 fold of the normcons base table, Philox draws
keyed by the ensemble-wide replica id (one seed for all ranks: the ensemble does not depend on N).
  --scaling weak   (default) 4096 replicas PER GPU: rank r owns replicas [4096 r, 4096 (r + 1))
  --scaling strong 4096 replicas IN TOTAL (BASELINE config 5 as written): rank r owns [4096 r / N, 4096 (r + 1) / N)
At N > 1 the line of the chosen scaling also carries the other one's numbers (`other_scaling`).
One "step" = one reference KMC step = int(64*64*0.5) = 2048 hops in every replica (hex_LiCoO2_kmc.py:484).
Nothing happens on the host between the steps of a run (the reference's loop, :745-758, adds the step's clock to a
total), so the K timed steps are executed in launches of up to 32 steps of the fused hop kernel, as the production
driver (ensemble.run_ensemble) runs them; --launch-steps 1 gives one launch per step, the granularity of the drop-in
kmc_step loop, which the line also reports (`per_step_launch_rank0`).  The results do not depend on the split.
Replicas are independent, ranks share nothing on the data path;
the single exchange is the all-reduce of the D* / D_J accumulators after the timed steps (inside the timed
region), loops hex_LiCoO2_kmc.py:679-693.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     -- the resource that binds hop_q64_kernel: vector-ALU busy cycles.  achieved = SQ_ACTIVE_INST_VALU
                  (quad-cycles, all waves) of one step / the step's GPU time measured live with HIP events on the
                  engine's streams; peak = 1024 SIMDs x 2.4 GHz / 4.  The counter comes from the committed PMC pass
                  profiles/pmc_current.json, which records the hash of the sources it was taken on: a pass taken on
                  other code is refused (frac = null, "stale").  `hbm` = measured HBM bytes (same pass) against 8 TB/s,
                  `traffic_avoided_vs_reference_algorithm` = SURVEY 8d's algorithmic bytes of the REFERENCE algorithm
                  (48 N^2 + 114 W + 50 per hop) / the bytes actually moved
  cpu_baseline -- the numpy oracle (a port of the reference loop) timed on one host core on a bounded sample of the
                  same workload, its C twin farmed over the box's host cores (count printed), and D* of sampled
                  GPU replicas against the C oracle on the same Philox draws
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = 'KMC hops/sec (whole node), 4096×64×64 hex replicas; D* match vs CPU ref'   # BASELINE.json, verbatim
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
N_SIMD = 256 * 4                 # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                 # peak engine clock (stamped in-kernel clock under this load: 2.36-2.39 GHz)
N_SIDE = 64
CONC = 0.5
ENSEMBLE = 4096                  # replicas per GPU (weak) / in total (strong)
WINDOW_SITES = 21.7              # mean sites refreshed per hop, hex_ref window
SEED = 20240607
REFERENCE_TEXT_HOPS_PER_S = 2.0e3   # the reference's own function text under python3, 1 core, N = 64 (SURVEY 6 / 8d,
                                    # measured in the build container: the text cannot travel to the GPU box)


def algorithmic_bytes_per_hop(n_side, window_sites=WINDOW_SITES):
    """one read of the fp64 rate matrix (sum + scan + search) + per refreshed site 18 B occupancy,
    48 B table row, 48 B rate write + 50 B of lattice / tag / draw / clock traffic."""
    return 48.0 * n_side * n_side + 114.0 * window_sites + 50.0


# ---------------------------------------------------------------------------------------------- launcher
class Watchdog:
    """A rank that does not get through a rendezvous (init_process_group, the RCCL communicator) must not sit in it for
    the process group's 20 minutes while its peers have died: `with Watchdog(what, seconds):` ends the process with exit
    code 3 and a line on stderr if the block has not finished in time.  The process is ended, never re-executed."""
    def __init__(self, what, seconds):
        import threading
        self.what, self.seconds = what, seconds
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True

    def fire(self):
        sys.stderr.write('bench.py: rank %s: %s did not complete in %d s; exiting\n' % (os.environ.get('RANK', '0'), self.what, self.seconds))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()


RENDEZVOUS_TIMEOUT_S = int(os.environ.get('KMC_BENCH_RENDEZVOUS_TIMEOUT', '180'))


def launch_ranks(n, argv):
    """Parent of a self-launched N-rank run.  Touches no GPU and imports no torch."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write('bench.py: rank %d exited with %d; stopping the others\n' % (procs.index(p), code))
                for q in live:
                    q.terminate()
    return rc


# ---------------------------------------------------------------------------------------------- CPU side
def cpu_baseline(np, base512, gpu_check, n_threads):
    """The oracle's numpy restatement of the per-hop loop (np.sum + np.cumsum + np.searchsorted + Python window
    refresh) on 1 core; its C twin farmed over n_threads host threads (ctypes releases the GIL), one independent
    64 x 64 lattice each; and the D* of sampled GPU replicas recomputed by the C oracle from the same lattices
    and the same Philox draws."""
    import threading
    from oracle import c_oracle as CO
    from oracle import kmc_oracle as O
    from oracle import philox as PH
    table = O.build_hex_table(base512)
    rs = np.random.RandomState(1)
    occ = np.zeros(N_SIDE * N_SIDE, dtype=np.uint8)
    occ[rs.permutation(N_SIDE * N_SIDE)[:int(N_SIDE * N_SIDE * CONC)]] = 1
    occ = occ.reshape(N_SIDE, N_SIDE)
    P = O.rates_full(occ, table)
    tags = O.init_tags(occ)
    n_hops = 81920                                           # 40 reference KMC steps, ~9 s
    draws = iter(rs.rand(2 * n_hops))
    t0 = time.perf_counter()
    O.kmc_step(occ.copy(), tags, P, table, n_hops, draws, False)
    dt = time.perf_counter() - t0
    out = dict(value=n_hops / dt, unit='hops/s', cores=1, kind='port',
               sample='1 lattice 64x64 x=0.5, %d hops (40 KMC steps), numpy oracle (np.sum + np.cumsum + np.searchsorted per hop)' % n_hops,
               seconds=dt, reference_text_value=REFERENCE_TEXT_HOPS_PER_S,
               reference_text_note='the reference\'s own function text (hex_LiCoO2_kmc.py:418-542) under python3 on one core of the build '
                                   'container, N = 64; it builds 18-character key strings per site, the port indexes a dense array')
    # C twin, one core and all cores
    farm_hops = 250000

    def one(k, res):
        r = np.random.RandomState(100 + k)
        o = np.zeros(N_SIDE * N_SIDE, dtype=np.uint8)
        o[r.permutation(N_SIDE * N_SIDE)[:int(N_SIDE * N_SIDE * CONC)]] = 1
        rep = CO.Replica(o.reshape(N_SIDE, N_SIDE), table)
        rep.init_tags()
        d = r.rand(farm_hops, 2)
        t1 = time.perf_counter()
        done, _, _ = rep.run(d, False, want_trace=False)
        res[k] = (done, time.perf_counter() - t1)
    res = {}
    one(0, res)
    out['c_port_value'] = res[0][0] / res[0][1]
    out['c_port_sample'] = '%d hops, oracle/kmc_oracle_c.c, 1 core' % res[0][0]
    res = {}
    th = [threading.Thread(target=one, args=(k, res)) for k in range(n_threads)]
    t2 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t2
    out['farm'] = dict(value=sum(v[0] for v in res.values()) / wall, unit='hops/s', cores=n_threads, kind='port',
                       sample='%d independent 64x64 lattices x %d hops, oracle/kmc_oracle_c.c, one lattice per host thread' % (n_threads, farm_hops),
                       seconds=wall)
    # D* match vs CPU ref: the GPU's replicas replayed by the C oracle (same lattice, same Philox stream)
    if gpu_check is not None:
        worst_d, worst_t = 0.0, 0.0
        same_lattice = True
        for k, rid in enumerate(gpu_check['replica_ids']):
            d = PH.hop_draws(SEED, int(rid), 0, gpu_check['presim_hops'] + gpu_check['sim_hops'])
            rep = CO.Replica(gpu_check['occ0'][k], table)
            rep.run(d[:gpu_check['presim_hops']], True, want_trace=False)
            rep.init_tags()
            rep.run(d[gpu_check['presim_hops']:], False, want_trace=False)
            ref = rep.diffusion(CONC)
            worst_d = max(worst_d, abs(gpu_check['d_star'][k] - ref[0]) / ref[0])
            worst_t = max(worst_t, abs(gpu_check['time'][k] - rep.time.value) / rep.time.value)
            same_lattice = same_lattice and bool(np.array_equal(gpu_check['occ1'][k], rep.occ))
        out['d_star_match'] = dict(replicas_checked=len(gpu_check['replica_ids']), hops_each=gpu_check['presim_hops'] + gpu_check['sim_hops'],
                                   final_lattice_identical=same_lattice, max_rel_err_d_star=worst_d, max_rel_err_clock=worst_t,
                                   tolerance=1e-12, ok=bool(same_lattice and worst_d <= 1e-12 and worst_t <= 1e-12))
    return out


def host_threads():
    """Host cores this process may use: the scheduler affinity, capped by the box's share (OMP_NUM_THREADS is set
    to it on the GPU box, where os.cpu_count() shows the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        n = min(n, int(os.environ.get('OMP_NUM_THREADS', n)))
    except ValueError:
        pass
    return max(1, min(n, 16))


# ---------------------------------------------------------------------------------------------- counters
def load_pmc(config, kernel, replicas_per_launch=None):
    """The entry of profiles/pmc_current.json for this BASELINE config (scripts/pmc_config.sh + scripts/summarize_profile.py)
    if it was taken on the sources this library is built from, on this kernel and launch shape; else (None, reason)."""
    from kmc_amd import build as kb
    path = os.path.join(ROOT, 'profiles', 'pmc_current.json')
    try:
        with open(path) as f:
            pj = json.load(f).get('configs', {}).get(str(config))
    except (OSError, ValueError):
        return None, 'profiles/pmc_current.json is missing'
    if pj is None:
        return None, 'profiles/pmc_current.json holds no counter pass for config %d' % config
    here = kb.source_sha256()
    if pj.get('source_sha256') != here:
        return None, 'stale: the counter pass %s was taken on sources %s..., this library is built from %s...' % (
            pj.get('tag'), str(pj.get('source_sha256'))[:12], here[:12])
    if kernel.split('<')[0] not in pj.get('kernel', ''):
        return None, 'the counter pass %s is of %s, this run launches %s' % (pj.get('tag'), pj.get('kernel'), kernel)
    if replicas_per_launch is not None and pj.get('replicas_per_launch') != replicas_per_launch:
        return None, 'the counter pass %s is of launches of %s replicas, this run launches %s' % (
            pj.get('tag'), pj.get('replicas_per_launch'), replicas_per_launch)
    return pj, None


LDS_PEAK_B_PER_CLK_CU = 256.0    # MI355X_MICROARCH.md (LDS): 64 banks x 4 B per clock per CU
N_CU = 256
# profiles/r2_issue_cost_probe.txt (scripts/probes/valu_rate_probe.hip): shader cycles per instruction per SIMD, from the
# cheapest class (VOP2 / SALU) to the dearest common one (VOP3, DPP, fp64, lane reads), by waves per SIMD
ISSUE_CYCLES = {1: (4.69, 5.70), 2: (2.35, 3.74), 3: (1.9, 3.2), 4: (1.67, 2.86)}


def build_roofline(config, kernel, k_ms, step_hops, groups, replicas_per_launch, waves_per_simd, simds_used, launch_ms=None, b_hop_ref=None):
    """The roofline object of the JSON line.  Every fraction is (per-hop counter of the committed PMC pass) x (hops per
    second measured live with HIP events) / (the resource's peak): VALU busy cycles, LDS bytes and LDS-array cycles,
    instruction issue, HBM bytes.  The counters are per launch; hops per launch = replicas per launch x hops per replica."""
    roof = {'bound': 'valu_busy', 'achieved': None, 'peak': simds_used * CLOCK_HZ / 4 / 1e9,
            'unit': 'G VALU-busy quad-cycles/s (over the %d SIMDs that hold a wave)' % simds_used,
            'frac': None, 'traffic': None, 'kernel': kernel, 'kernel_ms': k_ms, 'launches_per_step': groups,
            'hops_per_step_rank0': step_hops, 'waves_per_simd': waves_per_simd, 'simds_used': simds_used,
            'formula': 'achieved = SQ_ACTIVE_INST_VALU per hop (PMC, quad-cycles summed over waves) x hops per step / kernel_ms; '
                       'peak = SIMDs holding a wave x 2.4 GHz / 4; kernel_ms = HIP-event span of the timed launches / steps in them (live); '
                       'the counters are per hop, from a pass of one-step launches (a fused launch spares a prologue and a '
                       'write-back per step: ~1 % fewer instructions per hop than counted)'}
    if launch_ms is not None:
        roof['launch_ms'] = launch_ms
    pj, why = load_pmc(config, kernel, replicas_per_launch)
    roof['pmc'] = why
    if pj is None:
        return roof
    hops_launch = step_hops / groups                          # the pass ran this workload: same launches
    hops_s = step_hops / (k_ms * 1e-3)
    busy_q = pj['active_inst_valu_per_launch'] / hops_launch
    ach = busy_q * hops_s
    hbm_step = pj['hbm_bytes_per_launch'] / hops_launch * step_hops
    insts = {k: pj.get(k + '_insts_per_launch', 0.0) / hops_launch for k in ('valu', 'salu', 'lds', 'vmem_rd', 'vmem_wr', 'smem')}
    n_inst = sum(insts.values())
    lo, hi = ISSUE_CYCLES[max(1, min(4, int(round(waves_per_simd))))]
    per_simd = n_inst * hops_s / simds_used                   # instructions per second per SIMD
    roof.update(achieved=ach / 1e9, frac=ach / (simds_used * CLOCK_HZ / 4), traffic=int(hbm_step),
                pmc={'file': 'profiles/pmc_current.json', 'tag': pj.get('tag'), 'source_sha256': pj['source_sha256'],
                     'replicas_per_launch': pj['replicas_per_launch'], 'hops_per_launch': hops_launch,
                     'SQ_ACTIVE_INST_VALU_per_hop_quadcycles': busy_q,
                     'insts_per_hop': insts, 'SQ_WAVE_CYCLES_per_hop_quadcycles': pj.get('wave_cycles_per_launch', 0) / hops_launch,
                     'SQ_WAIT_ANY_per_hop_quadcycles': pj.get('wait_any_per_launch', 0) / hops_launch,
                     # share of a SIMD's cycles with the VALU busy while its waves are resident
                     'valu_busy_while_resident': (waves_per_simd * pj['active_inst_valu_per_launch'] / pj['wave_cycles_per_launch']
                                                  if pj.get('wave_cycles_per_launch') else None),
                     'lds_bank_conflict_frac': pj.get('lds_bank_conflict_frac'),
                     'l2_hit_rate': (pj['l2_hits_per_launch'] / (pj['l2_hits_per_launch'] + pj['l2_misses_per_launch'])
                                     if pj.get('l2_hits_per_launch') else None),
                     'pmc_kernel_avg_ms': pj.get('kernel_avg_ms'),
                     # the clock the chip held DURING the counter passes (GRBM_GUI_ACTIVE / the dispatch's duration in that
                     # pass) and the duration of a launch in the SQ pass: what reconciles SQ_WAVE_CYCLES with the 2.4 GHz peak
                     'pmc_clock_ghz': pj.get('pmc_clock_ghz'), 'pmc_launch_ms': pj.get('sq_pass_launch_ms')},
                issue={'insts_per_hop': n_inst, 'insts_per_s_per_simd': per_simd,
                       'probe_cycles_per_inst': [lo, hi], 'probe': 'profiles/r2_issue_cost_probe.txt, %d wave(s) per SIMD' % waves_per_simd,
                       'frac_range': [per_simd * lo / CLOCK_HZ, per_simd * hi / CLOCK_HZ],
                       'note': 'share of a SIMD\'s cycles its waves need just to ISSUE the hop\'s instructions, between the cheapest '
                               '(VOP2, SALU) and the dearest common class (VOP3, DPP, fp64, lane reads) of the issue-rate probe'},
                hbm={'bytes_per_step': int(hbm_step), 'bytes_per_hop': pj['hbm_bytes_per_launch'] / hops_launch,
                     'GB/s': hbm_step / (k_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'frac': hbm_step / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'note': 'FETCH_SIZE x 2 (gfx950 correction, upper bound for 16-B gathers) + WRITE_SIZE, separate --pmc passes'})
    try:
        with open(os.path.join(ROOT, 'profiles', 'lds_bytes_static.json')) as f:
            lb = json.load(f)
        from kmc_amd import build as kb
        ent = lb['kernels'].get(kernel)
        if ent and lb.get('source_sha256') == kb.source_sha256():
            b = ent['lds_array_bytes_per_hop_upper']
            peak = LDS_PEAK_B_PER_CLK_CU * N_CU * CLOCK_HZ
            roof['lds'] = {'bytes_per_hop_static_upper': b, 'TB/s': b * hops_s / 1e12, 'peak_TB/s': peak / 1e12, 'frac': b * hops_s / peak,
                           'array_busy_frac': (pj['lds_idx_active_per_launch'] / hops_launch * hops_s / (N_CU * CLOCK_HZ)
                                               if pj.get('lds_idx_active_per_launch') else None),
                           'note': 'bytes: every DS instruction of the hop loop x 64 lanes (scripts/lds_bytes.py, an upper bound); peak = 256 B/clk/CU '
                                   'x 256 CUs x 2.4 GHz; array_busy_frac = SQ_LDS_IDX_ACTIVE (LDS-array cycles, measured) / (256 CUs x clock)'}
    except (OSError, ValueError, KeyError):
        pass
    if b_hop_ref is not None:
        roof['traffic_avoided_vs_reference_algorithm'] = b_hop_ref * hops_launch / pj['hbm_bytes_per_launch']
    # which resource is nearest its peak, and how a wave's resident time splits (SQ_WAVE_CYCLES = issuing + parked on a
    # wait + stalled at issue): a kernel none of whose resources is near 1 is bound by its waves' dependent chains
    shares = {'valu_busy': roof['frac'], 'issue': 0.5 * sum(roof['issue']['frac_range']), 'hbm': roof['hbm']['frac']}
    if roof.get('lds') and roof['lds'].get('array_busy_frac') is not None:
        shares['lds_array'] = roof['lds']['array_busy_frac']
    roof['nearest_peak'] = max(shares, key=lambda k: shares[k])
    roof['shares_of_peak'] = shares
    if pj.get('wave_cycles_per_launch'):
        wc = pj['wave_cycles_per_launch']
        roof['wave_time'] = {'issuing': pj.get('active_inst_any_per_launch', 0) / wc, 'parked_on_a_wait': pj.get('wait_any_per_launch', 0) / wc,
                             'note': 'SQ_ACTIVE_INST_ANY and SQ_WAIT_ANY over SQ_WAVE_CYCLES (the rest: stalled at issue)'}
    return roof


# ---------------------------------------------------------------------------------------------- one measurement
# The other BASELINE.json configs (parity-test cases, not the metric's workload) as `--config N` lines of the same
# shape: n = lattice side, R = replicas (in total), conc = concentrations (cycled over the replicas, iteration-major as
# kmc_amd.ensemble shards them), table, hops of one "step" per replica (None: one reference KMC step, int(n*n*x)).
OTHER_CONFIGS = {
    1: dict(n=30, R=1, conc=[0.5], table='master', step_hops=None,
            what='config 1: single 30x30 hex lattice, x=0.5, master_dictionary.p table (its 512-key restriction folded)'),
    2: dict(n=64, R=256, conc=[0.5], table='normcons', step_hops=None, what='config 2: 256 independent 64x64 replicas at x=0.5'),
    3: dict(n=64, R=9 * 512, conc=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9], table='normcons', step_hops=None,
            what='config 3: concentration sweep x=0.1..0.9, 9 points x 512 replicas of 64x64; a step = one reference KMC step = int(4096 x) hops per replica'),
    4: dict(n=1024, R=1, conc=[0.5], table='normcons', step_hops=16384,
            what='config 4: single 1024x1024 lattice, x=0.5; a step here = 16384 hops (a reference KMC step would be 524288)'),
}


def measure(kmc_amd, ensemble, np, args, R, offset, local_rank, dist, torch, red_dev, keep_check, cfg=None):
    """W warm-up + K timed steps on this rank's R replicas (global ids offset .. offset + R).  Returns a dict;
    `elapsed` is the max over ranks."""
    n_side = cfg['n'] if cfg else N_SIDE
    tabs = np.load(os.path.join(ROOT, 'tests', 'golden', 'tables.npz'))
    base512 = tabs['base512_' + (cfg['table'] if cfg else 'normcons')]
    if cfg:
        conc = np.asarray(cfg['conc'], dtype=np.float64)[(offset + np.arange(R)) % len(cfg['conc'])]
        hops_vec = (np.full(R, cfg['step_hops']) if cfg['step_hops'] else (n_side * n_side * 1 * conc)).astype(np.int64)
        return measure_other(kmc_amd, ensemble, np, args, R, offset, local_rank, dist, torch, red_dev, cfg, conc, hops_vec, base512)
    hops_per_step = int(N_SIDE * N_SIDE * 1 * CONC)
    eng = kmc_amd.KMCEngine(R, N_SIDE, N_SIDE, window_mode='hex_ref', device=local_rank, replica_offset=offset)
    eng.set_base_table(base512, kmc_amd.tables.dumbbell_rate())
    fold_ms = eng.last_kernel_ms()
    eng.fill_random(CONC, SEED)                              # ONE seed: streams are keyed by the global replica id
    eng.rates_full()
    eng.sync()
    rates_ms = eng.last_kernel_ms()
    # equilibrate a little (presimulation mode), then switch the observables on
    eng.run(hops_per_step, presim=True, seed=SEED)
    eng.init_displacements()
    eng.run(hops_per_step, presim=False, seed=SEED)
    eng.diffusion(CONC)              # warm-up of the exchange step too (first call loads its kernel)
    rccl = None
    if args.collective == 'rccl' and dist is not None:
        # the engine's own communicator: rank 0 makes the id, torch.distributed only carries its 128 bytes
        with Watchdog('kmc_comm_create', RENDEZVOUS_TIMEOUT_S):
            box = [kmc_amd.KMCEngine.comm_unique_id() if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0, device=red_dev)
            rccl = (eng, eng.comm_create(box[0], dist.get_world_size(), dist.get_rank()))
    ensemble.all_reduce_acc(np.zeros((1, len(ensemble.ACC_FIELDS))), red_dev, rccl)   # ... and of the collective (RCCL set-up)
    S = launch_steps(args)

    def run_steps(k):
        """k KMC steps (2048 hops per replica each) in launches of up to S steps: nothing on the host happens between the
        steps of a run (the reference's loop, hex_LiCoO2_kmc.py:745-758, only adds the step's clock to a total), so the
        production driver, ensemble.run_ensemble, puts them into one launch; the results do not depend on the split."""
        sizes = []
        while k > 0:
            m = min(S, k)
            eng.run(hops_per_step * m, presim=False, seed=SEED)
            sizes.append(m)
            k -= m
        return sizes
    # the W warm-up steps come LAST, straight in front of the timed ones (no idle GPU between them and the timed region).
    # A run of few steps still reads lower than a long one -- 20 steps 2.33e9, 2000 steps 2.41e9 hops/s -- because the first
    # launch's prologue and the last one's tail are not hidden under a neighbour (one step alone takes 4.1 ms, back to back
    # 3.48) and the exchange step (0.5 ms, scripts/time_exchange.py) is shared by fewer steps.
    run_steps(args.warmup)
    eng.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            if red_dev is not None:
                torch.cuda.synchronize()
        eng.sync()

    barrier()
    t0 = time.perf_counter()
    sizes = run_steps(args.steps)
    # the one exchange step: D* / D_J accumulators of this rank's replicas
    ds, dj, _ = eng.diffusion(CONC)
    sc = eng.scalars()
    st = eng.status()
    acc = ensemble.accumulate(np.zeros(R, dtype=np.int64), 1, ds, dj, sc['time'],
                              st['hops_done'].astype(np.float64), sc['hist'].astype(np.float64))
    acc = ensemble.all_reduce_acc(acc, red_dev, rccl)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev if red_dev is not None else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # Duration of the dominant kernel: HIP events the engine recorded on its own streams around every hop-kernel
    # launch.  A step may be several concurrent launches over disjoint replicas (kmc_set_launch_groups) and the
    # launches of consecutive steps overlap, so the GPU time per step is the span from the first kernel start of
    # the timed region (of its last 64 steps) to the last kernel end, divided by the steps in it.
    n_span = min(len(sizes), 64)                             # launches (of S steps each, the last one maybe shorter)
    ev = eng.run_kernel_times(64)[-n_span:]
    span_ms, groups = eng.run_span_ms(n_span)
    out = dict(elapsed=elapsed, k_ms=span_ms / sum(sizes[-n_span:]), launch_ms=float(np.mean(ev)), groups=groups, R=R, launch_steps=S,
               n_exact=int(st['n_exact'].sum()), stopped=int((st['status'] != 0).sum()),
               stats=ensemble.finalize(acc, [CONC])[0], fold_ms=fold_ms, rates_ms=rates_ms,
               hops_per_step=hops_per_step, base512=base512, kernel=eng.hop_kernel_name)
    # For information (not `value`): the same steps as ONE LAUNCH PER KMC STEP, the granularity of the drop-in kmc_step loop
    # (every launch pays a prologue, a write-back and the skew of its waves' ends: the slots of a launch group's early
    # finishers idle until its last wave is done).  After everything that is reported.
    # (--no-long-launch: the profile passes skip it)
    out['per_step_launch'] = None
    if not args.no_long_launch and S > 1:
        n1 = 40
        for _ in range(4):
            eng.run(hops_per_step, presim=False, seed=SEED)
        eng.sync()
        t1 = time.perf_counter()
        for _ in range(n1):
            eng.run(hops_per_step, presim=False, seed=SEED)
        eng.sync()
        w1 = time.perf_counter() - t1
        span1, _ = eng.run_span_ms(n1)
        out['per_step_launch'] = dict(kmc_steps=n1, ms_per_step=w1 / n1 * 1e3, kernel_ms_per_step=span1 / n1,
                                      hops_per_s_this_rank=R * hops_per_step * n1 / w1)
    if rccl is not None:
        eng.comm_destroy(rccl[1])
    eng.close()
    return out


def measure_other(kmc_amd, ensemble, np, args, R, offset, local_rank, dist, torch, red_dev, cfg, conc, hops_vec, base512):
    """measure() for the --config lines: per-replica concentrations and hop counts (kmc_run_counts)."""
    n = cfg['n']
    eng = kmc_amd.KMCEngine(R, n, n, window_mode='hex_ref', device=local_rank, replica_offset=offset)
    eng.set_base_table(base512, kmc_amd.tables.dumbbell_rate())
    eng.fill_random(conc, SEED)
    eng.rates_full()
    eng.run_counts(hops_vec, presim=True, seed=SEED)
    eng.init_displacements()
    eng.run_counts(hops_vec, presim=False, seed=SEED)
    eng.diffusion(conc)
    ensemble.all_reduce_acc(np.zeros((1, len(ensemble.ACC_FIELDS))), red_dev)
    S = launch_steps(args)

    def run_steps(k):                    # k steps in launches of up to S (see measure())
        sizes = []
        while k > 0:
            m = min(S, k)
            eng.run_counts(hops_vec * m, presim=False, seed=SEED)
            sizes.append(m)
            k -= m
        return sizes
    run_steps(args.warmup)               # last, straight in front of the timed steps (see measure())
    eng.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            if red_dev is not None:
                torch.cuda.synchronize()
        eng.sync()

    barrier()
    t0 = time.perf_counter()
    sizes = run_steps(args.steps)
    ds, dj, _ = eng.diffusion(conc)
    sc, st = eng.scalars(), eng.status()
    point = ((offset + np.arange(R)) % len(cfg['conc'])).astype(np.int64)
    acc = ensemble.accumulate(point, len(cfg['conc']), ds, dj, sc['time'], st['hops_done'].astype(np.float64), sc['hist'].astype(np.float64))
    acc = ensemble.all_reduce_acc(acc, red_dev)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev if red_dev is not None else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_span = min(len(sizes), 64)
    span_ms, groups = eng.run_span_ms(n_span)
    out = dict(elapsed=elapsed, k_ms=span_ms / sum(sizes[-n_span:]), groups=groups, R=R, launch_steps=S, n_exact=int(st['n_exact'].sum()),
               stopped=int((st['status'] != 0).sum()), stats=ensemble.finalize(acc, cfg['conc']), step_hops_rank=int(hops_vec.sum()),
               lds_bytes=eng.lds_bytes, lds_path=eng.uses_lds_path, kernel=eng.hop_kernel_name,
               uniform_hops=bool((hops_vec == hops_vec[0]).all()))
    eng.close()
    return out


def dropin_rate(np, cfg, n_sim=40):
    """hops/s of the DROP-IN path itself for config 1: the reference's driver loop (hex_LiCoO2_kmc.py:716-758) written
    against kmc_amd.hex_LiCoO2_kmc -- kmc_evolve(initialise), presimulation kmc_steps, displacement tags, simulation
    kmc_steps, one call per reference KMC step, numpy's global RandomState for the draws.  Each call moves the draws up
    and the lattice, the rate matrix (48 B a site) and the tags back, as the signature demands; the state itself stays on
    the device between calls."""
    import kmc_amd.hex_LiCoO2_kmc as K
    n, x = cfg['n'], cfg['conc'][0]
    tabs = np.load(os.path.join(ROOT, 'tests', 'golden', 'tables.npz'))
    K.window_mode = 'hex_ref'
    K.set_rate_table(tabs['base512_' + cfg['table']])
    K.mc_lattice_size = [n, n, 1]
    K.Li_concentration = x
    np.random.seed(SEED % (2 ** 32))
    occ = np.zeros(n * n, dtype=np.uint8)
    occ[np.random.permutation(n * n)[:int(n * n * x)]] = 1
    lat = np.where(occ.reshape(n, n, 1) == 1, -1.0, 1.0)
    K.mc_lattice = lat
    P = K.kmc_evolve(lat, np.zeros((n, n, 1, 6)), 0, 0, True)
    for _ in range(3):
        (lat, P) = K.kmc_step(lat, None, P, True)
    dist = np.tile([-1, 0, 0], (n, n, 1, 1))
    ii, jj = np.nonzero(lat[:, :, 0] == -1)
    dist[ii, jj, 0, 0], dist[ii, jj, 0, 1] = ii, jj
    (lat, dist, _, P, _, _) = K.kmc_step(lat, dist, P, False)          # first simulation call: the tags go up
    t0 = time.perf_counter()
    total = 0.0
    for _ in range(n_sim):
        (lat, dist, dt, P, _, _) = K.kmc_step(lat, dist, P, False)
        total += dt
    wall = time.perf_counter() - t0
    hops = n_sim * int(n * n * 1 * x)
    (d_star, _, d_j) = K.get_diffusion_coefficient(dist, [n, n, 1], total, K.lattice_constant, x)
    K.close()
    return dict(hops_per_s=hops / wall, kmc_step_calls=n_sim, hops_per_call=int(n * n * 1 * x), ms_per_call=wall / n_sim * 1e3,
                d_star=float(d_star), what='kmc_amd.hex_LiCoO2_kmc.kmc_step in the reference\'s own loop shape (hex_LiCoO2_kmc.py:745-758), '
                                           'one call per KMC step, state resident on the device between calls')


def d_star_check_run(kmc_amd, np, local_rank, base512, n_replicas=4, sim_steps=40):
    """"D* match vs CPU ref": the ensemble's first replicas (global ids 0 ..) run again in a small engine of their own --
    the Philox streams are keyed by the global replica id, so these ARE the timed run's replicas over its first steps --
    for 1 presimulation + sim_steps simulation steps; cpu_baseline() replays them with the C oracle.  (Kept apart from
    the timed run so that its length does not grow with --steps: the oracle does 5e4 hops/s.)"""
    hops_per_step = int(N_SIDE * N_SIDE * 1 * CONC)
    with kmc_amd.KMCEngine(n_replicas, N_SIDE, N_SIDE, window_mode='hex_ref', device=local_rank) as eng:
        eng.set_base_table(base512, kmc_amd.tables.dumbbell_rate())
        eng.fill_random(CONC, SEED)
        occ0 = eng.get_lattice()
        eng.rates_full()
        eng.run(hops_per_step, presim=True, seed=SEED)
        eng.init_displacements()
        for _ in range(sim_steps):
            eng.run(hops_per_step, presim=False, seed=SEED)
        ds, _, _ = eng.diffusion(CONC)
        return dict(replica_ids=list(range(n_replicas)), occ0=occ0, occ1=eng.get_lattice(), d_star=ds, time=eng.scalars()['time'],
                    presim_hops=hops_per_step, sim_hops=hops_per_step * sim_steps)


def launch_steps(args):
    """KMC steps per launch of the timed run: --launch-steps, else as many as fit (up to 32)."""
    return max(1, args.launch_steps if args.launch_steps else min(args.steps, 32))


def occupancy_of(kernel, R):
    """(waves per SIMD, SIMDs that hold a wave) of a launch of R one-wave replicas of `kernel` on the 256 CUs."""
    if kernel.startswith('hop_big') or kernel.startswith('hop_global'):
        return 1, min(R, N_CU) * 4                            # one workgroup per lattice, on all four SIMDs of its CU (wave 0 runs
                                                              # the hop loop, the others sleep outside the exact replay)
    per_cu = 1 if kernel.startswith('hop_lat') else 16
    waves_cu = max(1, min(per_cu, -(-R // N_CU)))
    simds = min(R, N_CU * 4) if waves_cu >= 4 else min(R, N_CU * waves_cu)
    return max(1, waves_cu // 4), simds


def shard(total, rank, world):
    lo, hi = total * rank // world, total * (rank + 1) // world
    return hi - lo, lo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # default: 2000 steps = 7 s of timed region at 3.6 ms per step, long enough for a 5-second utilisation sampler to see the
    # GPU busy (config 4's step is 55 ms: 200)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--replicas', type=int, default=ENSEMBLE, help='replicas per GPU (weak) / in total (strong)')
    ap.add_argument('--collective', choices=('torch', 'rccl'), default='torch',
                    help='transport of the one all-reduce: torch.distributed (backend nccl = RCCL), or kmc_allreduce_stats '
                         '(an RCCL communicator of the engine\'s own, behind the C ABI)')
    ap.add_argument('--config', type=int, default=5, choices=(1, 2, 3, 4, 5),
                    help='BASELINE.json config to run: 5 (default) is the metric\'s workload; 1-4 are its parity-test cases')
    ap.add_argument('--launch-steps', type=int, default=None,
                    help='KMC steps per kernel launch (default: up to 32, as ensemble.run_ensemble runs them; 1 = one launch per '
                         'step, the granularity of the drop-in kmc_step loop and of the counter passes)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-long-launch', action='store_true', help='skip the informational one-launch-per-step leg after the timed region')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.config == 4 else 2000

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # Only the JSON line may reach stdout: libraries under us write there too (gloo's "[Gloo] Rank 0 is connected ..."),
    # so file descriptor 1 is pointed at stderr for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + '\n').encode())

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    torch = None
    backend = os.environ.get('KMC_BENCH_BACKEND', 'nccl')        # 'gloo': rehearsal of the N > 1 path on a 1-GPU box
    red_dev = None
    if world > 1:
        import datetime
        import torch
        import torch.distributed as dist
        if backend == 'nccl':
            if torch.cuda.device_count() < world:
                sys.exit('bench.py: %d ranks but %d GPU(s) visible (KMC_BENCH_BACKEND=gloo rehearses the N > 1 path on fewer)'
                         % (world, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            red_dev = torch.device('cuda', local_rank)
            with Watchdog('init_process_group(nccl)', RENDEZVOUS_TIMEOUT_S):
                dist.init_process_group('nccl', device_id=red_dev, timeout=datetime.timedelta(minutes=20))
                dist.barrier()                           # the communicator itself is made on first use: make it here
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            with Watchdog('init_process_group(%s)' % backend, RENDEZVOUS_TIMEOUT_S):
                dist.init_process_group(backend, timeout=datetime.timedelta(minutes=20))
                dist.barrier()

    import numpy as np
    import kmc_amd
    from kmc_amd import ensemble

    def replicas_of(scaling):
        if scaling == 'weak':
            return args.replicas, rank * args.replicas
        return shard(args.replicas, rank, world)

    if args.config != 5:
        cfg = OTHER_CONFIGS[args.config]
        # every rank runs the whole configuration's share: replicas sharded strong (a fixed ensemble), a single
        # lattice does not shard (replicas only) -- then every rank runs its own
        Rc, offc = shard(cfg['R'], rank, world) if cfg['R'] >= world else (cfg['R'], rank * cfg['R'])
        mo = measure(kmc_amd, ensemble, np, args, Rc, offc, local_rank, dist, torch, red_dev, 0, cfg)
        hops_all = mo['step_hops_rank']
        if dist is not None:
            tt = torch.tensor([float(hops_all)], dtype=torch.float64, device=red_dev if red_dev is not None else 'cpu')
            dist.all_reduce(tt)
            hops_all = float(tt.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            dropin = dropin_rate(np, cfg) if args.config == 1 else None
            kern = mo['kernel']
            wps, simds = occupancy_of(kern, mo['R'])
            if mo['uniform_hops']:
                roof = build_roofline(args.config, kern, mo['k_ms'], mo['step_hops_rank'], mo['groups'], mo['R'] // mo['groups'], wps, simds)
            else:
                roof = {'bound': 'valu_busy', 'achieved': None, 'peak': None, 'unit': None, 'frac': None, 'traffic': None, 'kernel': kern,
                        'kernel_ms': mo['k_ms'], 'launches_per_step': mo['groups'],
                        'pmc': 'replicas of this configuration run different hop counts per step: per-hop counters of a launch do not '
                               'normalise; the kernel is the headline line\'s (--config 5), which carries the roofline'}
            emit({
                'metric': 'KMC hops/sec, BASELINE config %d (a parity-test case, not the headline workload)' % args.config,
                'value': hops_all * args.steps / mo['elapsed'], 'unit': 'hops/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': mo['elapsed'] / args.steps * 1e3, 'higher_is_better': True,
                'scaling': 'strong' if cfg['R'] >= world else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                'config': {'workload': cfg['what'], 'replicas_total': cfg['R'] if cfg['R'] >= world else cfg['R'] * world,
                           'lattice': [cfg['n'], cfg['n']], 'concentrations': cfg['conc'], 'hops_per_step_all_replicas': hops_all, 'seed': SEED},
                'roofline': roof, 'dropin': dropin, 'launch_steps': mo['launch_steps'],
                'kernel_hops_per_s': mo['step_hops_rank'] / (mo['k_ms'] * 1e-3), 'exact_path_hops': mo['n_exact'],
                'stopped_replicas': mo['stopped'], 'd_star_by_concentration': [[r['concentration'], r['d_star'], r['error_star']] for r in mo['stats']]})
        return

    R, offset = replicas_of(args.scaling)
    m = measure(kmc_amd, ensemble, np, args, R, offset, local_rank, dist, torch, red_dev, keep_check=0)
    other = None
    if world > 1:
        o_scaling = 'strong' if args.scaling == 'weak' else 'weak'
        Ro, offo = replicas_of(o_scaling)
        mo = measure(kmc_amd, ensemble, np, args, Ro, offo, local_rank, dist, torch, red_dev, keep_check=0)
        total_o = args.replicas * (world if o_scaling == 'weak' else 1)
        other = dict(scaling=o_scaling, value=total_o * mo['hops_per_step'] * args.steps / mo['elapsed'], unit='hops/s',
                     replicas_total=total_o, replicas_rank0=Ro, ms_per_step=mo['elapsed'] / args.steps * 1e3,
                     kernel_ms=mo['k_ms'], launches_per_step=mo['groups'], d_star_mean=mo['stats']['d_star'])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    hops_per_step = m['hops_per_step']
    total = args.replicas * (world if args.scaling == 'weak' else 1)
    value = float(total) * hops_per_step * args.steps / m['elapsed']
    k_ms = m['k_ms']
    step_hops = R * hops_per_step                            # rank 0's hops per step
    b_hop = algorithmic_bytes_per_hop(N_SIDE)
    wps, simds = occupancy_of(m['kernel'], R)
    roof = build_roofline(5, m['kernel'], k_ms, step_hops, m['groups'], R // m['groups'], wps, simds, launch_ms=m['launch_ms'], b_hop_ref=b_hop)
    roof['reference_algorithm_bytes_per_hop'] = b_hop
    out = {
        'metric': METRIC,
        'value': value, 'unit': 'hops/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': m['elapsed'] / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%d x 64x64 hex replicas %s (%s scaling; %d in total on %d GPU(s)), x=0.5, hex_ref window, 2048 hops/step '
                               '(1 reference KMC step; up to %d steps per kernel launch), Philox draws keyed by global replica id, normcons table'
                               % (args.replicas, 'per GPU' if args.scaling == 'weak' else 'in total', args.scaling, total, world, m['launch_steps']),
                   'replicas_total': total, 'replicas_rank0': R, 'lattice': [N_SIDE, N_SIDE], 'concentration': CONC,
                   'hops_per_step': hops_per_step, 'seed': SEED,
                   'parallelism': 'replicas sharded over %d GPU(s), no data-path collective, one all-reduce of D* accumulators (%s)'
                                  % (world, 'torch.distributed' if args.collective == 'torch' or world == 1 else 'kmc_allreduce_stats: RCCL behind the C ABI')},
        'roofline': roof,
        'kernel_hops_per_s': step_hops / (k_ms * 1e-3),
        'exact_path_hops': m['n_exact'], 'stopped_replicas': m['stopped'],
        'd_star_mean': m['stats']['d_star'], 'd_star_std': m['stats']['error_star'], 'd_j_mean': m['stats']['d_j'],
        'stage_ms': {'table_fold': m['fold_ms'], 'rates_full': m['rates_ms']},
        'launch_steps': m['launch_steps'], 'per_step_launch_rank0': m['per_step_launch'],
    }
    if other is not None:
        out['other_scaling'] = other
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(np, m['base512'], d_star_check_run(kmc_amd, np, local_rank, m['base512']), host_threads())
    emit(out)


if __name__ == '__main__':
    main()
