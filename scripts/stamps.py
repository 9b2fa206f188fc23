"""Per-segment cycle breakdown of hop_lds_kernel (s_memtime stamps).

Build the instrumented library here (CPU box), run on the GPU box:
    python scripts/stamps.py --build          # hipcc -DKMC_STAMPS -> gpurun_out/libkmc_hip_stamps.so is NOT used;
                                              # the .so is written to kmc_amd/libkmc_hip_stamps.so
    KMC_AMD_LIB=kmc_amd/libkmc_hip_stamps.so python scripts/stamps.py [R] [hops]
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if '--build' in sys.argv:
    from kmc_amd import build
    out = os.path.join(ROOT, 'kmc_amd', 'libkmc_hip_stamps.so')
    subprocess.check_call([build.hipcc()] + build.FLAGS + ['-DKMC_STAMPS', '-o', out, build.SRC])
    print(out)
    sys.exit(0)
import ctypes as C
import numpy as np
import kmc_amd
from kmc_amd import _lib
t = np.load(os.path.join(ROOT, 'tests/golden/tables.npz'))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
eng = kmc_amd.KMCEngine(R, 64)
eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
eng.fill_random(0.5, 1); eng.rates_full(); eng.sync()
eng.run(H, presim=True, seed=3); eng.sync()
eng.init_displacements()
names = ['chunk setup + flush (draws, clock, tags)', 'row + block pick', 'entry: key, Tc gather, scan, certify',
         'window keys (column reads, key assembly)', 'W gathers + book-keeping + block / row sums', 'bpermute + row scan',
         'update scalars (after pick)', 'early reads + bitmaps']
for presim in (True, False):
    eng.run(H, presim=presim, seed=3); eng.sync()
    st = np.zeros((R, 8), dtype=np.int64)
    rc = _lib.load().kmc_debug_stamps(eng._ctx, st.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rc == 0, rc
    print('presim' if presim else 'sim', 'kernel %.3f ms' % eng.last_kernel_ms())
    tot = st.sum(axis=1).mean()
    for k in range(8):
        print('  %-34s %9.1f cycles/hop  %5.1f %%' % (names[k], st[:, k].mean() / H, 100 * st[:, k].mean() / tot))
    print('  %-34s %9.1f cycles/hop (memtime ticks; 100 MHz-based if constant clock)' % ('total', tot / H))
