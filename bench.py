#!/usr/bin/env python3
"""bench.py -- hops/s of the hex-lattice KMC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[4], the one the metric is quoted on): an ensemble
of 4096 independent 64 x 64 hex lattices at x = 0.5 PER GPU, random occupancy,
rate table = 2^18 x 6 fold of the normcons base table, Philox draws.  One
"step" = one reference KMC step = int(64*64*0.5) = 2048 hops in every replica
(hex_LiCoO2_kmc.py:484), executed by one launch of the fused hop kernel.  Replicas
are independent, so ranks share nothing on the data path (scaling = weak); the
single exchange is the all-reduce of the D* / D_J accumulators after the timed
steps (inside the timed region).

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     -- algorithmic bytes of the REFERENCE algorithm per launch
                  (B_hop = 48*N^2 + 114*W + 50 bytes) / kernel time
                  measured with HIP events on the engine's stream, vs 8 TB/s HBM
  cpu_baseline -- the numpy oracle (a port of the reference loop) timed on one
                  host core on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIDE = 64
CONC = 0.5
REPLICAS_PER_GPU = 4096
WINDOW_SITES = 21.7              # mean sites refreshed per hop, hex_ref window


def algorithmic_bytes_per_hop(n_side, window_sites=WINDOW_SITES):
    """one read of the fp64 rate matrix (sum + scan + search) +
    per refreshed site 18 B occupancy, 48 B table row, 48 B rate write + 50 B of
    lattice / tag / draw / clock traffic."""
    return 48.0 * n_side * n_side + 114.0 * window_sites + 50.0


def cpu_baseline(np, base512, seed=1):
    """The oracle's numpy restatement of the per-hop loop (np.sum + np.cumsum +
    np.searchsorted + Python window refresh), 1 core, one 64x64 x=0.5 lattice."""
    from oracle import kmc_oracle as O
    from oracle import c_oracle as CO
    table = O.build_hex_table(base512)
    rs = np.random.RandomState(seed)
    occ = np.zeros(N_SIDE * N_SIDE, dtype=np.uint8)
    occ[rs.permutation(N_SIDE * N_SIDE)[:int(N_SIDE * N_SIDE * CONC)]] = 1
    occ = occ.reshape(N_SIDE, N_SIDE)
    P = O.rates_full(occ, table)
    tags = O.init_tags(occ)
    n_hops = 122880                                          # 60 reference KMC steps, ~14 s
    draws = iter(rs.rand(2 * n_hops))
    t0 = time.perf_counter()
    O.kmc_step(occ, tags, P, table, n_hops, draws, False)
    dt = time.perf_counter() - t0
    # the C twin of the same algorithm, for scale (still one core)
    rep = CO.Replica(occ, table)
    rep.init_tags()
    d = rs.rand(400000, 2)
    t1 = time.perf_counter()
    done, _, _ = rep.run(d, False, want_trace=False)
    dt_c = time.perf_counter() - t1
    return dict(value=n_hops / dt, unit='hops/s', cores=1, kind='port',
                sample='1 lattice 64x64 x=0.5, %d hops (60 KMC steps), numpy oracle (np.sum+np.cumsum+np.searchsorted per hop)' % n_hops,
                seconds=dt, c_port_value=done / dt_c, c_port_sample='%d hops, oracle/kmc_oracle_c.c, 1 core' % done)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--replicas', type=int, default=REPLICAS_PER_GPU, help='replicas per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    torch = None
    backend = os.environ.get('KMC_BENCH_BACKEND', 'nccl')        # 'gloo': rehearsal of the N > 1 path on a 1-GPU box
    red_dev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            red_dev = torch.device('cuda', local_rank)
            dist.init_process_group('nccl', device_id=red_dev)
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend)

    import numpy as np
    import kmc_amd
    from kmc_amd import ensemble

    t = np.load(os.path.join(ROOT, 'tests', 'golden', 'tables.npz'))
    base512 = t['base512_normcons']
    hops_per_step = int(N_SIDE * N_SIDE * 1 * CONC)
    R = args.replicas
    seed = 20240607

    eng = kmc_amd.KMCEngine(R, N_SIDE, N_SIDE, window_mode='hex_ref', device=local_rank)
    eng.set_base_table(base512, kmc_amd.tables.dumbbell_rate())
    fold_ms = eng.last_kernel_ms()
    eng.fill_random(CONC, seed + rank)
    eng.rates_full()
    eng.sync()
    rates_ms = eng.last_kernel_ms()
    # equilibrate a little (presimulation mode), then switch the observables on
    eng.run(hops_per_step, presim=True, seed=seed + rank)
    eng.init_displacements()
    for _ in range(args.warmup):
        eng.run(hops_per_step, presim=False, seed=seed + rank)
    eng.diffusion(CONC)              # warm-up of the exchange step too (first call loads its kernel)
    ensemble.all_reduce_acc(np.zeros((1, len(ensemble.ACC_FIELDS))), red_dev)   # ... and of the collective (RCCL set-up)
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
        eng.run(hops_per_step, presim=False, seed=seed + rank)
    # the one exchange step: D* / D_J accumulators of this rank's replicas
    ds, dj, _ = eng.diffusion(CONC)
    sc = eng.scalars()
    st = eng.status()
    acc = ensemble.accumulate(np.zeros(R, dtype=np.int64), 1, ds, dj, sc['time'],
                              st['hops_done'].astype(np.float64), sc['hist'].astype(np.float64))
    acc = ensemble.all_reduce_acc(acc, red_dev)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev if red_dev is not None else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # duration of the dominant kernel: HIP events the engine recorded on its own
    # stream around each hop-kernel launch of the timed region
    # A step may be several concurrent launches over disjoint replicas (kmc_set_launch_groups), and the launches
    # of consecutive steps overlap, so the rate comes from the span: first kernel start of the timed region (of
    # its last 64 steps) to last kernel end, divided by the steps in it.
    n_span = min(args.steps, 64)
    ev = eng.run_kernel_times(64)[-n_span:]
    launch_ms = float(np.mean(ev))                           # one launch, as rocprofv3's kernel trace shows it
    span_ms, groups = eng.run_span_ms(n_span)
    k_ms = span_ms / n_span                                  # GPU time per step
    status = eng.status()
    stats = ensemble.finalize(acc, [CONC])[0]

    if rank == 0:
        hops_total = float(world) * R * hops_per_step * args.steps
        value = hops_total / elapsed
        b_hop = algorithmic_bytes_per_hop(N_SIDE)
        bytes_per_launch = b_hop * R * hops_per_step
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9       # bytes_per_launch: of one step (all its launches)
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as f:
                tj = json.load(f)
            if tj.get('replicas') == R and tj.get('hops_per_launch') == hops_per_step:
                # measured with rocprofv3 --pmc, not live; per launch -> per step
                traffic = int(tj['hbm_bytes_per_launch'] * (R / tj.get('replicas_per_launch', R)))
        except OSError:
            pass
        # what really bounds the kernel: vector-ALU issue.  VALU instructions per launch from the committed PMC
        # pass (profiles/, SQ_INSTS_VALU, same command) / the live kernel time, against one wave64 VALU
        # instruction per SIMD per 4 cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 4.
        valu = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as f:
                tj = json.load(f)
            if tj.get('replicas') == R and tj.get('hops_per_launch') == hops_per_step and 'valu_insts_per_launch' in tj:
                peak = 256 * 4 * 2.4e9 / 4
                per_step = tj['valu_insts_per_launch'] * (R / tj.get('replicas_per_launch', R))
                ach = per_step / (k_ms * 1e-3)
                valu = {'achieved': ach / 1e9, 'peak': peak / 1e9, 'unit': 'G wave-instructions/s', 'frac': ach / peak,
                        'insts_per_hop': per_step / (R * hops_per_step),
                        'source': 'SQ_INSTS_VALU of profiles/%s_pmc.json (not live) / live kernel time' % tj.get('tag')}
        except OSError:
            pass
        out = {
            'metric': 'KMC hops/sec (whole node), 4096x64x64 hex replicas per GPU; D* match vs CPU ref',
            'value': value, 'unit': 'hops/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%d x 64x64 hex replicas per GPU, x=0.5, hex_ref window, 2048 hops/step (1 reference KMC step), Philox draws, normcons table' % R,
                       'replicas_per_gpu': R, 'lattice': [N_SIDE, N_SIDE], 'concentration': CONC,
                       'hops_per_step': hops_per_step, 'parallelism': 'replicas sharded over %d GPU(s), no data-path collective, one all-reduce of D* accumulators' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'hop_q64_kernel', 'kernel_ms': k_ms, 'launches_per_step': groups,
                         'launch_ms': launch_ms, 'algorithmic_bytes_per_hop': b_hop,
                         'hops_per_launch': R * hops_per_step // groups,
                         'span': 'a step is %d concurrent launch(es) over disjoint replicas and steps are not joined on the device; achieved and traffic are per step: the step\'s algorithmic bytes / kernel_ms, kernel_ms = HIP-event span of the timed launches / steps; launch_ms = mean duration of one launch (what rocprofv3 lists)' % groups,
                         'note': 'algorithmic bytes are those of the REFERENCE algorithm (full rate-matrix scan per hop); the kernel keeps replica state in LDS and scans hierarchically, so frac > 1 means traffic avoided, not HBM over peak'},
            'valu_issue': valu,
            'kernel_hops_per_s': R * hops_per_step / (k_ms * 1e-3),
            'exact_path_hops': int(status['n_exact'].sum()), 'stopped_replicas': int((status['status'] != 0).sum()),
            'd_star_mean': stats['d_star'], 'd_star_std': stats['error_star'], 'd_j_mean': stats['d_j'],
            'stage_ms': {'table_fold': fold_ms, 'rates_full': rates_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(np, base512)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
