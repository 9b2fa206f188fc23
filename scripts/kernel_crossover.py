"""64x64: the 8-per-CU kernel (kmc_hop_lds64.h, shorter chain) against the 16-per-CU one (kmc_hop_q64.h) by replica count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmc_amd
t = np.load(os.path.join(ROOT, 'tests/golden/tables.npz'))
H = 2048
for R in (256, 512, 1024, 1536, 2048, 2560, 3072, 4096, 6144, 8192):
    row = []
    for which in (3, 2):
        eng = kmc_amd.KMCEngine(R, 64)
        eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
        eng.set_hop_kernel(which=which)
        eng.fill_random(0.5, 1); eng.rates_full(); eng.init_displacements(); eng.sync()
        eng.run(H, presim=False, seed=3); eng.sync()
        ms = []
        for k in range(3):
            eng.run(H, presim=False, seed=4 + k); eng.sync(); ms.append(eng.last_kernel_ms())
        row.append(min(ms))
        eng.close()
    print('R = %5d   q64 %.3f ms   lds64 %.3f ms   (%.2e / %.2e hops/s)' % (R, row[0], row[1], R * H / row[0] * 1e3, R * H / row[1] * 1e3), flush=True)
