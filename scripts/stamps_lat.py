"""Per-segment cycle breakdown (s_memtime stamps, -DKMC_STAMPS build) of the hop loop at ONE replica per CU: hop_q64_kernel
(what 256 replicas ran before round 3) against hop_lat_kernel (what they run now).
    python scripts/stamps.py --build                                                 (build container)
    KMC_AMD_LIB=kmc_amd/libkmc_hip_stamps.so python scripts/stamps_lat.py [R] [hops] [kernels, e.g. 3 or 3,5]   (GPU box)
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmc_amd
from kmc_amd import _lib
t = np.load(os.path.join(ROOT, 'tests/golden/tables.npz'))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
NAMES = {3: ['chunk setup + flush (draws, clock, tags)', 'row + quarter pick', 'entry: keys, pair-table gather (L2), scan, certify',
             'window keys (column reads, key assembly)', 'W gathers (L2) + book-keeping + quarter sums', 'bpermute + row scan',
             'update scalars (after pick)', 'bitmaps'],
         5: ['chunk setup + flush (draws, clock, tags)', 'row pick, row read (LDS), site scan + pick',
             'direction: palette read (LDS), 6-lane scan, certify', 'geometry, fresh-key atomics, window key read, gather issue',
             'old leaves (LDS) + book-keeping in the gather\'s shadow', 'wait for the gather (L2), 3 adds, 8-lane tree, bpermute',
             'row scan, gamma', '(of the direction stage: palette words, palette read, 6-lane scan)']}
ONLY = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [3, 5]
for which, label in ((3, 'hop_q64_kernel'), (5, 'hop_lat_kernel')):
    if which not in ONLY:
        continue
    eng = kmc_amd.KMCEngine(R, 64)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.set_hop_kernel(which=which)
    eng.fill_random(0.5, 1); eng.rates_full(); eng.sync()
    eng.run(H, presim=True, seed=3); eng.sync()
    eng.init_displacements()
    for presim in (True, False):
        eng.run(H, presim=presim, seed=3); eng.sync()
        st = np.zeros((R, 8), dtype=np.int64)
        rc = _lib.load().kmc_debug_stamps(eng._ctx, st.ctypes.data_as(C.POINTER(C.c_int64)))
        assert rc == 0, rc
        print('%s, %d replicas, %s: kernel %.3f ms (stamped build)' % (label, R, 'presimulation' if presim else 'simulation', eng.last_kernel_ms()))
        tot = st.sum(axis=1).mean()
        for k in range(8):
            if st[:, k].any():
                print('  %-62s %8.1f cycles/hop %5.1f %%' % (NAMES[which][k], st[:, k].mean() / H, 100 * st[:, k].mean() / tot))
        print('  %-62s %8.1f cycles/hop' % ('total', tot / H))
    eng.close()
