"""Kernel time against hops per launch and against replica count near one resident round (GPU box):
separates the per-launch cost (state in/out of LDS, building the sums) and the tail of a launch from the
per-hop cost.   python scripts/launch_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
def run(R, hops_list, reps=6):
    eng = kmc_amd.KMCEngine(R, 64)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 1); eng.rates_full()
    eng.run(4096, presim=True, seed=3); eng.init_displacements(); eng.sync()
    for H in hops_list:
        ms = []
        for k in range(reps):
            eng.run(H, presim=False, seed=3); eng.sync(); ms.append(eng.last_kernel_ms())
        m = float(np.median(ms[1:]))
        print('R %5d  hops/launch %6d  kernel %8.3f ms  %7.1f ns/hop/replica-round  %.3e hops/s'
              % (R, H, m, m * 1e6 / H, R * H / (m * 1e-3)), flush=True)
    eng.close()
run(4096, [64, 256, 1024, 2048, 4096, 8192, 16384])
for R in (2048, 3072, 3584, 3840, 4096):
    run(R, [2048])
