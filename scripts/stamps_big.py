"""Per-segment cycles of hop_big_kernel (run with KMC_AMD_LIB=kmc_amd/libkmc_hip_stamps.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd
from kmc_amd import _lib
t = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/tables.npz'))
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng = kmc_amd.KMCEngine(1, nx)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 11); eng.rates_full(); eng.init_displacements()
eng.run(64, presim=False, seed=3); eng.sync()
names = ['chunk setup / flush', 'descent to the L1 block', 'entry: keys, Tc, scan, certify', 'tags, prefetches, bitmap flip',
         'window keys', 'W gathers + L1 sums', 'climb L2, L3, row + store', 'R1, R2, scan']
for presim in (True, False):
    eng.run(H, presim=presim, seed=3); eng.sync()
    st = np.zeros((1, 8), dtype=np.int64)
    assert _lib.load().kmc_debug_stamps(eng._ctx, st.ctypes.data_as(C.POINTER(C.c_int64))) == 0
    print('presim' if presim else 'sim', 'kernel %.3f ms, n_exact %d' % (eng.last_kernel_ms(), eng.status()['n_exact'][0]))
    for k in range(8):
        print('  %-34s %9.1f cycles/hop' % (names[k], st[0, k] / H))
    print('  total %.1f' % (st[0].sum() / H))
