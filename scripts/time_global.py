import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
for nx, generic in ((2048, False), (1024, True)):
    eng = kmc_amd.KMCEngine(1, nx)
    if generic: eng.set_hop_kernel(generic=True)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 11); eng.rates_full()
    eng.run(500, presim=True, seed=3); eng.init_displacements(); eng.sync()
    eng.run(5000, presim=False, seed=3); eng.sync()
    ms = eng.last_kernel_ms(); st = eng.status()
    print('n %d  global-memory kernel: %.1f ms per 5000 hops  %.3e hops/s  n_exact %d' % (nx, ms, 5000 / (ms * 1e-3), int(st['n_exact'].sum())), flush=True)
    eng.close()
