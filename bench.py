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
One "step" = one reference KMC step = int(64*64*0.5) = 2048 hops in every replica (hex_LiCoO2_kmc.py:484),
executed by one run of the fused hop kernel.  Replicas are independent, ranks share nothing on the data path;
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
def load_pmc(np, R_local, hops_per_step):
    """profiles/pmc_current.json (scripts/profile_bench.sh + scripts/summarize_profile.py) if it was taken on the
    sources this library is built from and on this workload; else (None, reason)."""
    from kmc_amd import build as kb
    path = os.path.join(ROOT, 'profiles', 'pmc_current.json')
    try:
        with open(path) as f:
            pj = json.load(f)
    except OSError:
        return None, 'profiles/pmc_current.json is missing'
    here = kb.source_sha256()
    if pj.get('source_sha256') != here:
        return None, 'stale: profiles/pmc_current.json (%s) was taken on sources %s..., this library is built from %s...' % (
            pj.get('tag'), str(pj.get('source_sha256'))[:12], here[:12])
    if pj.get('hops_per_launch') != hops_per_step:
        return None, 'profiles/pmc_current.json is for %s hops per launch' % pj.get('hops_per_launch')
    return pj, None


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
    for _ in range(args.warmup):
        eng.run(hops_per_step, presim=False, seed=SEED)
    eng.diffusion(CONC)              # warm-up of the exchange step too (first call loads its kernel)
    rccl = None
    if args.collective == 'rccl' and dist is not None:
        # the engine's own communicator: rank 0 makes the id, torch.distributed only carries its 128 bytes
        box = [kmc_amd.KMCEngine.comm_unique_id() if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0, device=red_dev)
        rccl = (eng, eng.comm_create(box[0], dist.get_world_size(), dist.get_rank()))
    ensemble.all_reduce_acc(np.zeros((1, len(ensemble.ACC_FIELDS))), red_dev, rccl)   # ... and of the collective (RCCL set-up)
    eng.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            if red_dev is not None:
                torch.cuda.synchronize()
        eng.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run(hops_per_step, presim=False, seed=SEED)
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
    n_span = min(args.steps, 64)
    ev = eng.run_kernel_times(64)[-n_span:]
    span_ms, groups = eng.run_span_ms(n_span)
    out = dict(elapsed=elapsed, k_ms=span_ms / n_span, launch_ms=float(np.mean(ev)), groups=groups, R=R,
               n_exact=int(st['n_exact'].sum()), stopped=int((st['status'] != 0).sum()),
               stats=ensemble.finalize(acc, [CONC])[0], fold_ms=fold_ms, rates_ms=rates_ms,
               hops_per_step=hops_per_step, base512=base512)
    # For information (not `value`): the same kernel when a run is ONE launch of many KMC steps, as ensemble.run_ensemble
    # and the sweep driver make them -- no prologue / write-back / tail per 2048 hops.  After everything that is reported.
    # (--no-long-launch: the profile passes average their counters over the launches of ONE size)
    out['long_launch'] = None
    if not args.no_long_launch:
        long_steps = 32
        eng.run(hops_per_step * long_steps, presim=False, seed=SEED)
        eng.sync()
        out['long_launch'] = dict(kmc_steps_per_launch=long_steps, kernel_ms=eng.last_kernel_ms(),
                                  hops_per_s_this_rank=R * hops_per_step * long_steps / (eng.last_kernel_ms() * 1e-3))
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
    for _ in range(args.warmup):
        eng.run_counts(hops_vec, presim=False, seed=SEED)
    eng.diffusion(conc)
    ensemble.all_reduce_acc(np.zeros((1, len(ensemble.ACC_FIELDS))), red_dev)
    eng.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            if red_dev is not None:
                torch.cuda.synchronize()
        eng.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run_counts(hops_vec, presim=False, seed=SEED)
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
    n_span = min(args.steps, 64)
    span_ms, groups = eng.run_span_ms(n_span)
    out = dict(elapsed=elapsed, k_ms=span_ms / n_span, groups=groups, R=R, n_exact=int(st['n_exact'].sum()),
               stopped=int((st['status'] != 0).sum()), stats=ensemble.finalize(acc, cfg['conc']), step_hops_rank=int(hops_vec.sum()),
               lds_bytes=eng.lds_bytes, lds_path=eng.uses_lds_path)
    eng.close()
    return out


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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-long-launch', action='store_true', help='skip the informational 32-step launch after the timed region')
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
            dist.init_process_group('nccl', device_id=red_dev, timeout=datetime.timedelta(minutes=20))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=20))

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
            kern = ('hop_q64_kernel' if cfg['n'] == 64 else 'hop_lds_kernel') if mo['lds_path'] else 'hop_big_kernel'
            emit({
                'metric': 'KMC hops/sec, BASELINE config %d (a parity-test case, not the headline workload)' % args.config,
                'value': hops_all * args.steps / mo['elapsed'], 'unit': 'hops/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': mo['elapsed'] / args.steps * 1e3, 'higher_is_better': True,
                'scaling': 'strong' if cfg['R'] >= world else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                'config': {'workload': cfg['what'], 'replicas_total': cfg['R'] if cfg['R'] >= world else cfg['R'] * world,
                           'lattice': [cfg['n'], cfg['n']], 'concentrations': cfg['conc'], 'hops_per_step_all_replicas': hops_all, 'seed': SEED},
                'roofline': {'bound': 'latency (one wave\'s dependent chain)' if cfg['R'] < 3072 else 'valu_busy', 'achieved': None, 'peak': None, 'unit': None,
                             'frac': None, 'traffic': None, 'kernel': kern, 'kernel_ms': mo['k_ms'], 'launches_per_step': mo['groups'],
                             'note': 'no PMC pass is kept for this configuration; the headline line (--config 5) carries the roofline'},
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
    pj, why = load_pmc(np, R, hops_per_step)
    roof = {'bound': 'valu_busy', 'achieved': None, 'peak': N_SIMD * CLOCK_HZ / 4 / 1e9, 'unit': 'G VALU-busy quad-cycles/s (all SIMDs)',
            'frac': None, 'traffic': None, 'kernel': 'hop_q64_kernel', 'kernel_ms': k_ms, 'launches_per_step': m['groups'],
            'launch_ms': m['launch_ms'], 'hops_per_step_rank0': step_hops,
            'formula': 'achieved = SQ_ACTIVE_INST_VALU per hop (PMC, quad-cycles summed over waves) x hops per step / kernel_ms; '
                       'peak = 1024 SIMDs x 2.4 GHz / 4; kernel_ms = HIP-event span of the timed launches / steps (live); '
                       'launch_ms = mean duration of one launch (what rocprofv3 lists)',
            'pmc': why}
    if pj is not None:
        per_hop = pj['hops_per_launch'] * pj['replicas_per_launch']
        busy_q = pj['active_inst_valu_per_launch'] / per_hop                     # quad-cycles per hop
        ach = busy_q * step_hops / (k_ms * 1e-3)
        hbm_step = pj['hbm_bytes_per_launch'] / per_hop * step_hops
        roof.update(achieved=ach / 1e9, frac=ach / (N_SIMD * CLOCK_HZ / 4), traffic=int(hbm_step),
                    pmc={'file': 'profiles/pmc_current.json', 'tag': pj.get('tag'), 'source_sha256': pj['source_sha256'],
                         'replicas_per_launch': pj['replicas_per_launch'],
                         'SQ_ACTIVE_INST_VALU_per_hop_quadcycles': busy_q,
                         'SQ_INSTS_VALU_per_hop': pj.get('valu_insts_per_launch', 0) / per_hop,
                         'SQ_INSTS_SALU_per_hop': pj.get('salu_insts_per_launch', 0) / per_hop,
                         'valu_busy_while_resident': pj.get('valu_busy_while_resident'),
                         'lds_bank_conflict_frac': pj.get('lds_bank_conflict_frac'),
                         'pmc_kernel_avg_ms': pj.get('kernel_avg_ms')},
                    hbm={'bytes_per_step': int(hbm_step), 'bytes_per_hop': pj['hbm_bytes_per_launch'] / per_hop,
                         'GB/s': hbm_step / (k_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'frac': hbm_step / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'note': 'FETCH_SIZE x 2 (gfx950 correction, upper bound for 16-B gathers) + WRITE_SIZE, separate --pmc passes'},
                    traffic_avoided_vs_reference_algorithm=b_hop * per_hop / pj['hbm_bytes_per_launch'])
    roof['reference_algorithm_bytes_per_hop'] = b_hop
    out = {
        'metric': METRIC,
        'value': value, 'unit': 'hops/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': m['elapsed'] / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%d x 64x64 hex replicas %s (%s scaling; %d in total on %d GPU(s)), x=0.5, hex_ref window, 2048 hops/step '
                               '(1 reference KMC step), Philox draws keyed by global replica id, normcons table'
                               % (args.replicas, 'per GPU' if args.scaling == 'weak' else 'in total', args.scaling, total, world),
                   'replicas_total': total, 'replicas_rank0': R, 'lattice': [N_SIDE, N_SIDE], 'concentration': CONC,
                   'hops_per_step': hops_per_step, 'seed': SEED,
                   'parallelism': 'replicas sharded over %d GPU(s), no data-path collective, one all-reduce of D* accumulators (%s)'
                                  % (world, 'torch.distributed' if args.collective == 'torch' or world == 1 else 'kmc_allreduce_stats: RCCL behind the C ABI')},
        'roofline': roof,
        'kernel_hops_per_s': step_hops / (k_ms * 1e-3),
        'exact_path_hops': m['n_exact'], 'stopped_replicas': m['stopped'],
        'd_star_mean': m['stats']['d_star'], 'd_star_std': m['stats']['error_star'], 'd_j_mean': m['stats']['d_j'],
        'stage_ms': {'table_fold': m['fold_ms'], 'rates_full': m['rates_ms']},
        'long_launch_rank0': m['long_launch'],
    }
    if other is not None:
        out['other_scaling'] = other
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(np, m['base512'], d_star_check_run(kmc_amd, np, local_rank, m['base512']), host_threads())
    emit(out)


if __name__ == '__main__':
    main()
