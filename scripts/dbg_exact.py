import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import kmc_amd
t = np.load('tests/golden/tables.npz')
for generic in (True, False):
    eng = kmc_amd.KMCEngine(1, 70, 66)
    eng.set_hop_kernel(generic=generic)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 11); eng.rates_full(); eng.init_displacements()
    eng.set_debug(force_exact=True)
    for hops in (1, 1, 2, 8):
        print('generic', generic, 'hops', hops, flush=True)
        tr = eng.run(hops, presim=False, seed=3, want_trace=True); eng.sync()
        print('  trace', tr[0], 'status', eng.status(), flush=True)
    eng.close()
