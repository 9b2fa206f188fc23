"""Any-size LDS kernel on ensembles of small lattices (the sizes of the reference's D-against-size study)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
for n, R in ((10, 8192), (20, 8192), (30, 4096), (32, 4096), (40, 4096), (50, 2048), (50, 4096)):
    eng = kmc_amd.KMCEngine(R, n)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 1); eng.rates_full()
    H = 2048
    eng.run(H, presim=True, seed=3); eng.init_displacements(); eng.sync()
    for _ in range(2): eng.run(H, presim=False, seed=3)
    reps = 6
    for _ in range(reps): eng.run(H, presim=False, seed=3)
    span, g = eng.run_span_ms(reps)
    print('N %3d  R %5d  LDS %6d B/replica  %.3f ms per %d-hop run  %.3e hops/s' % (n, R, eng.lds_bytes, span / reps, H, R * H * reps / (span * 1e-3)), flush=True)
    eng.close()
