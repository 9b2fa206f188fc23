"""N > 1 path on CPU: two gloo ranks shard a concentration sweep and all-reduce the
accumulators (the single exchange step of the ensemble)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    from kmc_amd import ensemble
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    points = [0.2, 0.5, 0.8]
    conc, idx, _ = ensemble.shard_replicas(points, 8, rank, world)
    rs = np.random.RandomState(100 + rank)            # stands in for this rank's device results
    ds, dj, t = rs.rand(conc.size), rs.rand(conc.size), rs.rand(conc.size)
    acc = ensemble.accumulate(idx, 3, ds, dj, t, np.full(conc.size, 10.0), np.ones((conc.size, 6)))
    tot = ensemble.all_reduce_acc(acc)
    if rank == 0:
        print('RESULT ' + json.dumps(dict(acc=tot.tolist(), fin=ensemble.finalize(tot, points))))
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_two_rank_all_reduce(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29617', str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('RESULT ')][0]
    res = json.loads(line[7:])
    # recompute both ranks' contributions here
    from kmc_amd import ensemble
    want = 0
    all_ds = {0: [], 1: [], 2: []}
    for rank in range(2):
        conc, idx, _ = ensemble.shard_replicas([0.2, 0.5, 0.8], 8, rank, 2)
        rs = np.random.RandomState(100 + rank)
        ds, dj, t = rs.rand(conc.size), rs.rand(conc.size), rs.rand(conc.size)
        want = want + ensemble.accumulate(idx, 3, ds, dj, t, np.full(conc.size, 10.0), np.ones((conc.size, 6)))
        for p in range(3):
            all_ds[p] += ds[idx == p].tolist()
    np.testing.assert_allclose(np.array(res['acc']), want, rtol=1e-14)
    for p in range(3):
        assert res['fin'][p]['replicas'] == 8
        np.testing.assert_allclose(res['fin'][p]['d_star'], np.mean(all_ds[p]), rtol=1e-13)
        np.testing.assert_allclose(res['fin'][p]['error_star'], np.std(all_ds[p]), rtol=1e-9)


SWEEP_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from kmc_amd import sweep

    class FakeEngine:
        """Stands in for the device: deterministic numbers from (fill seed, replica, concentration)."""
        def __init__(self, R, nx, ny, wm, dev):
            self.R, self.n, self.hops = R, nx * ny, np.zeros(R, dtype=np.int64)
        def set_base_table(self, base, dumbbell): pass
        def set_replica_offset(self, offset): self.offset = offset
        def fill_random(self, conc, seed): self.conc, self.seed = np.asarray(conc, dtype=float), seed
        def rates_full(self): pass
        def init_displacements(self): pass
        def run_counts(self, n, presim=False, seed=0): self.hops += np.asarray(n)
        def status(self): return dict(status=np.zeros(self.R, dtype=np.int32), hops_done=self.hops, n_exact=self.hops * 0)
        def _v(self): return (1 + self.offset + np.arange(self.R)) * self.conc * (1 + (self.seed %% 97))
        def diffusion(self, conc): return self._v() * 1e-10, self._v() * 3e-10, np.zeros((self.R, 4))
        def scalars(self): return dict(time=self._v() * 1e-6, hist=np.tile(np.arange(6), (self.R, 1)) + self.hops[:, None])
        def barrier_stats(self): return dict(sum=self._v(), count=np.ones(self.R), min=self._v() * 0.5, max=self._v() * 2, last=self._v())
        def close(self): pass

    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    res = sweep.run_sweep(('base', np.ones(512), 1.0), [8, 12], [0.25, 0.5], 4, 2, 3, seed=5, out_dir=sys.argv[1],
                          rank=rank, world=world, engine_factory=FakeEngine, echo=False)
    if rank == 0:
        print('RESULT ' + json.dumps({str(k): v for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_two_rank_sweep_writes_files(tmp_path):
    """The driver script's loop (kmc_amd.sweep) over two gloo ranks with a stand-in engine: every rank takes half
    of each concentration's replicas, the means are over both, rank 0 writes the reference's result files."""
    import json
    from kmc_amd import ensemble, report
    script = tmp_path / 'sweep_worker.py'
    script.write_text(SWEEP_WORKER)
    out_dir = tmp_path / 'out'
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29619', str(script), str(out_dir)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('RESULT ')][0][7:])
    dat = report.read_data_out(str(out_dir / 'LiCoO2_data.out'))
    kout = report.read_kmc_out(str(out_dir / 'LiCoO2_kmc.out'))
    assert kout['size'] == [8, 8, 1] and kout['presim'] == 2 and kout['sim'] == 3
    row = 0
    for n in (8, 12):
        for k, x in enumerate((0.25, 0.5)):
            # what the stand-in engines of both ranks produce for this point: its 4 replicas are the global
            # ids a * 2 + k (iteration-major), two per rank, ONE seed for both ranks
            fill_seed = ensemble.derive_seed(5, n)
            vals = ((1 + np.arange(4) * 2 + k) * x * (1 + fill_seed % 97)).tolist()
            r = res[str(n)][k]
            assert r['replicas'] == 4 and r['hops'] == 4 * 5 * int(n * n * x)
            np.testing.assert_allclose(r['d_star'], np.mean(vals) * 1e-10, rtol=1e-13)
            np.testing.assert_allclose(r['error_j'], np.std(vals) * 3e-10, rtol=1e-8)
            assert dat['Dimension'][row] == n and dat['Concentration'][row] == x and dat['D_star'][row] == r['d_star']
            np.testing.assert_allclose(kout['table'][row], [r['d_star'], x, r['error_star'], r['d_j'], r['error_j']], rtol=2e-8)
            row += 1


def test_shard_replicas_is_a_partition_at_any_world_size():
    """Global replica ids: iteration-major, contiguous per rank, the same ensemble for every number of ranks,
    also when the replicas per concentration do not divide by it."""
    from kmc_amd import ensemble
    points = [0.1, 0.4, 0.7]
    whole_c, whole_i, off = ensemble.shard_replicas(points, 10, 0, 1)
    assert off == 0 and whole_c.size == 30 and np.array_equal(whole_i, np.tile(np.arange(3), 10))
    for world in (2, 3, 4, 7, 8):
        got_c, got_i, nxt = [], [], 0
        for rank in range(world):
            c, i, o = ensemble.shard_replicas(points, 10, rank, world)
            assert o == nxt and c.size % 3 == 0            # contiguous, whole iterations
            nxt = o + c.size
            got_c.append(c)
            got_i.append(i)
        assert np.array_equal(np.concatenate(got_c), whole_c) and np.array_equal(np.concatenate(got_i), whole_i)
    import pytest
    with pytest.raises(ValueError):
        ensemble.shard_replicas(points, 2, 2, 4)           # a rank without replicas
    # ... and EVERY rank raises it, also those whose own share is not empty (7 replicas over 8 ranks leave only rank 0
    # without one): otherwise the others would go on into the all-reduce and wait for the process-group time-out
    for rank in range(8):
        with pytest.raises(ValueError):
            ensemble.shard_replicas(points, 7, rank, 8)
    # seeds of neighbouring lattice sizes are unrelated (seed + dimension would collide with seed + 1, dimension - 1)
    assert ensemble.derive_seed(5, 11) != ensemble.derive_seed(6, 10)
    assert len({ensemble.derive_seed(s, d) for s in range(20) for d in range(20)}) == 400


def test_bench_self_launch_reports_a_failing_rank():
    """`python bench.py --gpus 2` with no WORLD_SIZE spawns two rank processes before touching a GPU and exits
    non-zero when one fails.  Here (no GPU) every rank fails at kmc_create -- the engine has no CPU backend -- so the
    launcher must say which rank and return its code, with nothing on stdout; on a GPU box the same command prints
    the one JSON line with n_gpus = 2."""
    import json
    env = dict(os.environ, KMC_BENCH_BACKEND='gloo', OMP_NUM_THREADS='1')
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=300)
    if out.returncode == 0:                       # a GPU is present
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2
    else:
        assert out.stdout.strip() == ''
        assert 'bench.py: rank' in out.stderr and 'no CPU backend' in out.stderr


def test_rank_whose_peer_never_arrives_exits_in_time():
    """bench.py's rendezvous watchdog: rank 0 of a two-rank job whose rank 1 never starts must not wait out the process
    group's 20 minutes -- it exits with code 3 and says what did not complete."""
    import socket
    import time
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, KMC_BENCH_BACKEND='gloo', KMC_BENCH_RENDEZVOUS_TIMEOUT='6', RANK='0', LOCAL_RANK='0', WORLD_SIZE='2',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1')
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])
    assert 'did not complete' in out.stderr and 'init_process_group' in out.stderr
    assert time.time() - t0 < 200
