"""Rate of hop_q64_kernel on long launches (4096 x 64x64, the launch groups as the engine picks them):
    python scripts/long_launch.py        (GPU box; KMC_AMD_LIB selects a variant build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
eng = kmc_amd.KMCEngine(4096, 64)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 1); eng.rates_full()
eng.run(4096, presim=True, seed=3); eng.init_displacements(); eng.sync()
for H in (2048, 32768, 262144):
    ms = []
    for k in range(3):
        eng.run(H, presim=False, seed=3); eng.sync(); ms.append(eng.last_kernel_ms())
    m = min(ms)
    print('hops/launch %7d  kernel %9.3f ms  %.3e hops/s' % (H, m, 4096 * H / (m * 1e-3)), flush=True)
