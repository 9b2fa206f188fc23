"""Cost of one exact-path hop (forced) and of certified hops on a single large lattice."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load('tests/golden/tables.npz')
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = kmc_amd.KMCEngine(1, nx)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 11); eng.rates_full(); eng.init_displacements()
eng.run(64, presim=False, seed=3); eng.sync()
for hops in (1, 64, 2048):
    eng.run(hops, presim=False, seed=3); eng.sync(); print('certified: %5d hops  %.3f ms  n_exact so far %d' % (hops, eng.last_kernel_ms(), eng.status()['n_exact'][0]))
eng.set_debug(force_exact=True)
for hops in (1, 8):
    eng.run(hops, presim=False, seed=3); eng.sync(); print('forced exact: %d hops  %.3f ms' % (hops, eng.last_kernel_ms()))
