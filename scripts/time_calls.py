import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
eng = kmc_amd.KMCEngine(R, 64)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 1); eng.rates_full(); eng.sync()
eng.run(H, presim=True, seed=3); eng.sync()
eng.init_displacements()
for k in range(6):
    t0 = time.perf_counter(); eng.run(H, presim=(k % 2 == 0), seed=3); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    print('presim' if k % 2 == 0 else 'sim   ', 'launch %.3f ms  sync %.3f ms  kernel %.3f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, eng.last_kernel_ms()))
print(eng.status()['n_exact'].sum(), eng.run_kernel_times())
