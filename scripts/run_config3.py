"""BASELINE config 3 on one GPU: the concentration sweep x = 0.1 ... 0.9, 512 replicas of 64 x 64 each, 250
presimulation + 750 simulation KMC steps (hex_LiCoO2_kmc.py:656-657), result files as the reference writes them.
    python scripts/run_config3.py [out_dir] [replicas_per_concentration]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kmc_amd import report, sweep, tables
out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'config3')
per = int(sys.argv[2]) if len(sys.argv) > 2 else 512
base = np.load(os.path.join(ROOT, 'tests', 'golden', 'tables.npz'))['base512_normcons']
xs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
t0 = time.time()
res = sweep.run_sweep(('base', base, tables.dumbbell_rate()), [64], xs, per, 250, 750, seed=2024, out_dir=out_dir)[64]
dt = time.time() - t0
hops = sum(r['hops'] for r in res)
print(json.dumps(dict(config='config 3: 9 concentrations x %d replicas of 64x64, 250 + 750 steps, 1 GPU' % per,
                      hops=hops, seconds=dt, hops_per_s=hops / dt)))
for r in res:
    print('x %.1f  D* %.4e +- %.2e  D_J %.4e +- %.2e  t %.3e s  offset_score %.3f' % (
        r['concentration'], r['d_star'], r['error_star'], r['d_j'], r['error_j'], r['time'],
        report.offset_score(np.array(r['hop_histogram']))))
