"""One 1024 x 1024 lattice (BASELINE config 4), three 20000-hop launches of hop_big_kernel (for rocprofv3 passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
eng = kmc_amd.KMCEngine(1, 1024)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 11); eng.rates_full()
eng.run(2000, presim=True, seed=3); eng.init_displacements(); eng.sync()
for _ in range(3):
    eng.run(20000, presim=False, seed=3); eng.sync()
    print('kernel %.3f ms' % eng.last_kernel_ms(), flush=True)
print('n_exact', int(eng.status()['n_exact'].sum()))
