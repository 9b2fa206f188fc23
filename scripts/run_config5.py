"""BASELINE config 5 on one GPU: 4096 replicas of 64 x 64 at x = 0.5, 10^6 hops per replica (488 reference KMC steps
of 2048 hops: 122 presimulation + 366 simulation), D* and D_J over the ensemble; then the D_J-against-size study of
the reference's dimension_check (Mc_Lattice.py:243-258) with 256 lattices per size instead of 10.
    python scripts/run_config5.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmc_amd
from kmc_amd import diagnostics, ensemble, tables
base = np.load(os.path.join(ROOT, 'tests', 'golden', 'tables.npz'))['base512_normcons']
table = ('base', base, tables.dumbbell_rate())
t0 = time.time()
res, _ = ensemble.run_ensemble(lambda R, nx, ny, wm, dev: kmc_amd.KMCEngine(R, nx, ny, wm, dev), 64, 64, [0.5], 4096,
                               122, 366, seed=5, table=table)
dt = time.time() - t0
r = res[0]
print(json.dumps(dict(config='config 5: 4096 x 64x64, x = 0.5, 10^6 hops per replica (122 + 366 KMC steps), 1 GPU',
                      hops=r['hops'], seconds=dt, hops_per_s=r['hops'] / dt, d_star=r['d_star'], d_star_std=r['error_star'],
                      d_j=r['d_j'], d_j_std=r['error_j'], mean_time=r['time'], hop_histogram=r['hop_histogram'])), flush=True)
t0 = time.time()
sizes = [10, 20, 30, 40, 50, 64]
dj = diagnostics.dimension_check(table, dimensions=sizes, averaging=256, seed=9)
print('D_J against size (x = 0.2, 200 + 500 steps, 256 lattices per size; class normalisation), %.1f s' % (time.time() - t0))
for n in sizes:
    print('N %3d  D_J %.4e +- %.2e (std of one lattice), standard error %.1e' % (n, dj[n].mean(), dj[n].std(), dj[n].std() / 16))
