"""Host time of the exchange step's calls after a 4096-replica step (what a bench.py run of few steps amortises).
    python scripts/time_exchange.py      (GPU box)"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, kmc_amd
t = np.load(os.path.join(ROOT, 'tests/golden/tables.npz'))
eng = kmc_amd.KMCEngine(4096, 64)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 1); eng.rates_full(); eng.init_displacements()
eng.run(2048, presim=False, seed=1); eng.sync()
for rep in range(3):
    eng.run(2048, presim=False, seed=1)
    t0 = time.perf_counter(); eng.sync(); t1 = time.perf_counter()
    ds = eng.diffusion(0.5); t2 = time.perf_counter()
    sc = eng.scalars(); t3 = time.perf_counter()
    st = eng.status(); t4 = time.perf_counter()
    print('sync %.3f diffusion %.3f scalars %.3f status %.3f ms' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
