"""GPU time per hop-round against hops per run, one launch per run vs two launch groups (kmc_set_launch_groups)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for mode in (1, 2):
    eng = kmc_amd.KMCEngine(R, 64)
    eng.set_launch_groups(mode)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 1); eng.rates_full()
    eng.run(4096, presim=True, seed=3); eng.init_displacements(); eng.sync()
    for H, reps in ((512, 40), (2048, 30), (8192, 12), (32768, 4)):
        for _ in range(3): eng.run(H, presim=False, seed=3)
        eng.sync()
        for _ in range(reps): eng.run(H, presim=False, seed=3)
        span, g = eng.run_span_ms(reps)
        print('R %d groups %d  hops/run %6d  %8.3f ms per run  %7.1f ns per hop-round  %.3e hops/s'
              % (R, g, H, span / reps, span / reps * 1e6 / H, R * H * reps / (span * 1e-3)), flush=True)
    eng.close()
