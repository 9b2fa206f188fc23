import os, sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np
import kmc_amd
t = np.load('tests/golden/tables.npz')
def run(R, nx, hops, conc=0.5, reps=2, label=''):
    eng = kmc_amd.KMCEngine(R, nx)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(conc, 11); eng.rates_full()
    eng.run(hops, presim=True, seed=3); eng.init_displacements(); eng.sync()
    ms = []
    for _ in range(reps):
        eng.run(hops, presim=False, seed=3); eng.sync(); ms.append(eng.last_kernel_ms())
    st = eng.status()
    print(json.dumps(dict(config=label, kernel_ms=float(np.median(ms)), hops_per_s=R * hops / (np.median(ms) * 1e-3), n_exact=int(st['n_exact'].sum()), hops_done=int(st['hops_done'].sum()))), flush=True)
    eng.close()
run(1, 1024, 5000, label='config4: single 1024x1024')
run(64, 256, 5000, label='64 x 256x256')
