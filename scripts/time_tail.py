"""Host-side cost of the exchange step that closes bench.py's timed region."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
from kmc_amd import ensemble
t = np.load('tests/golden/tables.npz')
R = 4096
eng = kmc_amd.KMCEngine(R, 64)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 1); eng.rates_full(); eng.run(2048, presim=True, seed=3); eng.init_displacements()
eng.run(2048, presim=False, seed=3); eng.sync()
for rep in range(3):
    t0 = time.perf_counter(); ds, dj, _ = eng.diffusion(0.5)
    t1 = time.perf_counter(); sc = eng.scalars()
    t2 = time.perf_counter(); st = eng.status()
    t3 = time.perf_counter()
    acc = ensemble.accumulate(np.zeros(R, dtype=np.int64), 1, ds, dj, sc['time'], st['hops_done'].astype(np.float64), sc['hist'].astype(np.float64))
    t4 = time.perf_counter(); acc = ensemble.all_reduce_acc(acc, None)
    t5 = time.perf_counter()
    print('diffusion %.3f  scalars %.3f  status %.3f  accumulate %.3f  all_reduce %.3f ms' % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
