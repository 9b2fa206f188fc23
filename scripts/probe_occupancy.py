"""How does the time of a hop grow with the number of replicas (waves) a CU carries?
Uses the generic LDS kernel on 32x32 lattices (about 8 KB of LDS each), so occupancy is set by registers.
    python scripts/probe_occupancy.py --build 4     # variant with amdgpu_waves_per_eu(4,4) -> kmc_amd/libkmc_hip_w4.so
    KMC_AMD_LIB=kmc_amd/libkmc_hip_w4.so python scripts/probe_occupancy.py
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if '--build' in sys.argv:
    from kmc_amd import build
    w = sys.argv[sys.argv.index('--build') + 1]
    out = os.path.join(ROOT, 'kmc_amd', 'libkmc_hip_w%s.so' % w)
    subprocess.check_call([build.hipcc()] + build.FLAGS + ['-DKMC_LDS_WAVES=' + w, '-o', out, build.SRC])
    print(out)
    sys.exit(0)
import numpy as np
import kmc_amd
t = np.load(os.path.join(ROOT, 'tests/golden/tables.npz'))
N = 32
H = 4096
for per_cu in (1, 4, 8, 12, 16):
    R = 256 * per_cu
    eng = kmc_amd.KMCEngine(R, N)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 1); eng.rates_full(); eng.sync()
    eng.run(H, presim=True, seed=3); eng.sync()
    ms = []
    for k in range(3):
        eng.run(H, presim=True, seed=4 + k); eng.sync(); ms.append(eng.last_kernel_ms())
    best = min(ms)
    print('%2d replicas per CU: %.3f ms per %d hops = %.3f us per hop-round, %.3e hops/s' %
          (per_cu, best, H, best * 1e3 / H, R * H / best * 1e3), flush=True)
    del eng
