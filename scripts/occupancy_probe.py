"""How does hop throughput scale with waves per CU?  Same kernel, smaller lattices
(less LDS per replica -> more resident replicas per CU)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
t = np.load('tests/golden/tables.npz')
for nx, R in ((64, 3072), (56, 4096), (48, 4096), (40, 8192), (32, 8192), (24, 16384), (16, 16384)):
    eng = kmc_amd.KMCEngine(R, nx)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 11); eng.rates_full()
    eng.run(1024, presim=True, seed=3); eng.init_displacements(); eng.sync()
    ms = []
    for _ in range(3):
        eng.run(2048, presim=False, seed=3); eng.sync(); ms.append(eng.last_kernel_ms())
    lds = eng.lds_bytes
    per_cu = min(32, 163840 // lds)
    print(json.dumps(dict(n=nx, R=R, lds=lds, waves_per_cu=per_cu, ms=float(np.median(ms)),
                          hops_per_s=R * 2048 / (np.median(ms) * 1e-3),
                          us_per_hop_per_wave=np.median(ms) * 1e3 / 2048 / max(1.0, R / (per_cu * 256.0)))), flush=True)
    eng.close()
