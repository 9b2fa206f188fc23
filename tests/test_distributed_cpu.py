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
    conc, idx = ensemble.shard_replicas(points, 8, rank, world)
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
        conc, idx = ensemble.shard_replicas([0.2, 0.5, 0.8], 8, rank, 2)
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
