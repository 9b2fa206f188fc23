"""Two engines (two contexts, two streams) in one process, launches overlapping: does a launch lose time to
its tail?   python scripts/corun.py R_each n_engines"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
def go(R, E, H=2048, steps=20):
    engs = []
    for e in range(E):
        eng = kmc_amd.KMCEngine(R, 64)
        eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
        eng.fill_random(0.5, 1 + e); eng.rates_full()
        eng.run(H, presim=True, seed=3); eng.init_displacements(); eng.sync()
        engs.append(eng)
    for _ in range(3):
        for eng in engs: eng.run(H, presim=False, seed=3)
    for eng in engs: eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for eng in engs: eng.run(H, presim=False, seed=3)
    for eng in engs: eng.sync()
    dt = time.perf_counter() - t0
    print('%d engine(s) x %5d replicas: %.3f ms per step of all engines, %.3e hops/s, kernel ms %s'
          % (E, R, dt / steps * 1e3, E * R * H * steps / dt, ['%.2f' % engs[e].last_kernel_ms() for e in range(E)]), flush=True)
    for eng in engs: eng.close()
if len(sys.argv) > 1:
    for a in sys.argv[1:]:
        R, E = a.split('x'); go(int(R), int(E))
else:
    go(4096, 1); go(4096, 2); go(2048, 2); go(1024, 4); go(4096, 3); go(512, 8)
