"""Ensembles of small lattices (the sizes of the reference's D-against-size study): the any-size kernel with block sums in
LDS (kmc_set_hop_kernel 1) against the 16-per-CU layout at any size (4, the default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
for n, R, which in [(n, R, w) for n, R in ((10, 8192), (20, 8192), (30, 4096), (32, 4096), (40, 4096), (50, 2048), (50, 4096), (30, 1), (30, 256))
                    for w in (1, 4)]:
    eng = kmc_amd.KMCEngine(R, n)
    eng.set_hop_kernel(which=which)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 1); eng.rates_full()
    H = 2048
    eng.run(H, presim=True, seed=3); eng.init_displacements(); eng.sync()
    for _ in range(2): eng.run(H, presim=False, seed=3)
    reps = 6
    for _ in range(reps): eng.run(H, presim=False, seed=3)
    span, g = eng.run_span_ms(reps)
    print('kernel %d  N %3d  R %5d  LDS %6d B/replica  %.3f ms per %d-hop run  %.3e hops/s' % (which, n, R, eng.lds_bytes, span / reps, H, R * H * reps / (span * 1e-3)), flush=True)
    eng.close()
