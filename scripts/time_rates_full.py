"""Rate fill (rows R1, R2) timing: kmc_rates_full on 4096 replicas; bytes = 1 B/site read + 4 B/site key write.
    python scripts/time_rates_full.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
for R, n in ((4096, 64), (16384, 64), (4096, 30), (4096, 40), (64, 256), (1, 1024)):
    eng = kmc_amd.KMCEngine(R, n)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 1)
    ms = []
    for _ in range(8):
        eng.rates_full(); eng.sync(); ms.append(eng.last_kernel_ms())
    best, med = min(ms), float(np.median(ms))
    by = 5.0 * R * n * n
    print('%5d x %4dx%-4d  rates_full min %.4f ms median %.4f ms  %.1f MB  %.2f TB/s (median)  %.3f of 8 TB/s' % (
        R, n, n, best, med, by / 1e6, by / (med * 1e-3) / 1e12, by / (med * 1e-3) / 8e12), flush=True)
    eng.close()
