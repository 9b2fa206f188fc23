"""Time of one exact replay at 1024 x 1024 (hop_big_kernel): every hop forced through it, long form (np.sum + cumsum) and
short form (cumsum only, kmc_set_debug 2).      python scripts/time_exact_big.py [variant names]      (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import kmc_amd
t = np.load(%r + '/tests/golden/tables.npz')
eng = kmc_amd.KMCEngine(1, 1024)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 11); eng.rates_full(); eng.init_displacements()
eng.run(64, presim=False, seed=3); eng.sync()
eng.run(32, presim=False, seed=3); eng.sync(); base = eng.last_kernel_ms()
out = []
for mode in (1, 2):
    eng.set_debug(force_exact=mode)
    eng.run(32, presim=False, seed=3); eng.sync()
    out.append((eng.last_kernel_ms() - base) / 32)
print('ms per replay: long form %%.3f  short form %%.3f   (32 hops each; rho uniform: the short form reads half the entries on average)' %% tuple(out))
''' % (ROOT, ROOT)
for name in (sys.argv[1:] or ['']):
    env = dict(os.environ)
    if name: env['KMC_AMD_LIB'] = os.path.join(ROOT, 'kmc_amd', 'libkmc_hip_v_%s.so' % name)
    o = subprocess.run([sys.executable, '-c', CODE], env=env, capture_output=True, text=True, timeout=200)
    print('%-10s %s' % (name or 'library', (o.stdout.strip() or o.stderr.strip()[-400:])), flush=True)
