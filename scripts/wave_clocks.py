"""Per-wave clocks and end times of hop_q64_kernel launches (build: python scripts/wave_clocks.py --build;
run on the GPU box: KMC_AMD_LIB=kmc_amd/libkmc_hip_clocks.so python scripts/wave_clocks.py [R])."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if '--build' in sys.argv:
    from kmc_amd import build
    out = os.path.join(ROOT, 'kmc_amd', 'libkmc_hip_clocks.so')
    subprocess.check_call([build.hipcc()] + build.FLAGS + ['-DKMC_STAMPS=2', '-o', out, build.SRC])
    print(out)
    sys.exit(0)
import ctypes as C
import numpy as np
import kmc_amd
from kmc_amd import _lib
t = np.load(os.path.join(ROOT, 'tests/golden/tables.npz'))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = 2048
for mode in (1, 2):
    eng = kmc_amd.KMCEngine(R, 64)
    eng.set_launch_groups(mode)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(0.5, 1); eng.rates_full()
    eng.run(H, presim=True, seed=3); eng.init_displacements()
    for _ in range(12): eng.run(H, presim=False, seed=3)
    span, g = eng.run_span_ms(8)
    st = np.zeros((R, 8), dtype=np.int64)
    assert _lib.load().kmc_debug_stamps(eng._ctx, st.ctypes.data_as(C.POINTER(C.c_int64))) == 0
    clk = st[:, 0] / st[:, 1] * 100.0                       # MHz
    dur = st[:, 1] / 100.0                                    # us
    t0 = (st[:, 2] - st[:, 2].min()) / 100.0
    t1 = (st[:, 3] - st[:, 2].min()) / 100.0
    print('groups %d: %.3f ms per run;  last run: wave clock %.0f MHz mean (%.0f - %.0f);  wave duration %.0f us mean, '
          'p5 %.0f, p50 %.0f, p95 %.0f, max %.0f;  starts within %.0f us;  ends: p5 %.0f p50 %.0f p95 %.0f max %.0f us'
          % (g, span / 8, clk.mean(), clk.min(), clk.max(), dur.mean(), *np.percentile(dur, [5, 50, 95]), dur.max(),
             t0.max(), *np.percentile(t1, [5, 50, 95]), t1.max()), flush=True)
    print('   prologue (state into LDS, sums): %.0f us mean, max %.0f;  write-back: %.0f us mean, max %.0f'
          % (st[:, 6].mean() / 100.0, st[:, 6].max() / 100.0, st[:, 7].mean() / 100.0, st[:, 7].max() / 100.0), flush=True)
    hw = st[:, 4]
    simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, st[:, 5] & 15
    slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = np.bincount(slot, minlength=1)
    per_simd = np.bincount(slot * 4 + simd)
    print('   waves per CU: min %d max %d over %d CUs;  per SIMD: min %d max %d;  duration by waves on the SIMD: %s'
          % (per_cu[per_cu > 0].min(), per_cu.max(), (per_cu > 0).sum(), per_simd[per_simd > 0].min(), per_simd.max(),
             {int(k): round(float(dur[per_simd[slot * 4 + simd] == k].mean())) for k in np.unique(per_simd[per_simd > 0])}), flush=True)
    wid = hw & 15
    key = slot * 4 + simd
    distinct = sum(len(set((wid[key == k] & 3).tolist())) == 4 for k in np.unique(key))
    print('   SIMDs whose 4 waves have 4 different (WAVE_ID & 3): %d of %d;  duration by WAVE_ID: %s' % (
        distinct, len(np.unique(key)), {int(w): round(float(dur[wid == w].mean())) for w in np.unique(wid)}), flush=True)
    spread = [dur[key == k].max() - dur[key == k].min() for k in np.unique(key)]
    print('   spread (max - min duration) inside a SIMD: mean %.0f us;  between SIMD means: std %.0f us;  between CU means: std %.0f us'
          % (np.mean(spread), np.std([dur[key == k].mean() for k in np.unique(key)]), np.std([dur[slot == k].mean() for k in np.unique(slot)])), flush=True)
    for x in range(8):
        m = xcc == x
        if m.any():
            print('   XCC %d: %4d waves, clock %.0f MHz, duration %.0f us' % (x, m.sum(), clk[m].mean(), dur[m].mean()))
    eng.close()
