"""A/B of variant builds (scripts/variants.py --build) on one 1024 x 1024 lattice (BASELINE config 4): same seed, so the same
hops take the exact path in every variant.
    python scripts/variants_big.py name1 name2 ...          (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import kmc_amd
t = np.load(%r + '/tests/golden/tables.npz')
eng = kmc_amd.KMCEngine(1, 1024)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 11); eng.rates_full()
eng.run(2000, presim=True, seed=3); eng.init_displacements(); eng.sync()
ms = []
for _ in range(3):
    eng.run(20000, presim=False, seed=3); eng.sync(); ms.append(eng.last_kernel_ms())
st = eng.status()
print('kernel ms', ' '.join('%%.2f' %% m for m in ms), ' hops/s %%.4g' %% (20000 / (min(ms) * 1e-3)), ' n_exact', int(st['n_exact'].sum()), flush=True)
''' % (ROOT, ROOT)
for name in sys.argv[1:]:
    env = dict(os.environ, KMC_AMD_LIB=os.path.join(ROOT, 'kmc_amd', 'libkmc_hip_v_%s.so' % name))
    o = subprocess.run([sys.executable, '-c', CODE], env=env, capture_output=True, text=True, timeout=200)
    print('%-14s %s' % (name, (o.stdout.strip() or o.stderr.strip()[-400:])), flush=True)
