"""Throughput of the other BASELINE configs (not bench lines; parity-test cases):
   python scripts/bench_configs.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmc_amd

t = np.load('tests/golden/tables.npz')


def run(R, nx, hops, conc=0.5, reps=3, label=''):
    eng = kmc_amd.KMCEngine(R, nx)
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.fill_random(conc, 11); eng.rates_full()
    eng.run(hops, presim=True, seed=3); eng.init_displacements(); eng.sync()
    eng.run(hops, presim=False, seed=3)
    for _ in range(reps):                       # back to back; GPU time = span of the launches' HIP events / runs
        eng.run(hops, presim=False, seed=3)
    span, groups = eng.run_span_ms(reps)
    ms = [span / reps]
    st = eng.status()
    out = dict(config=label, replicas=R, n=nx, hops_per_run=hops, launches_per_run=groups, kernel_ms=float(np.median(ms)),
               hops_per_s=R * hops / (np.median(ms) * 1e-3), n_exact=int(st['n_exact'].sum()),
               hops_done=int(st['hops_done'].sum()), lds_path=eng.uses_lds_path)
    print(json.dumps(out), flush=True)
    eng.close()


run(1, 30, 45000, label='config1: single 30x30 x=0.5')
run(256, 64, 4096, reps=10, label='config2: 256 x 64x64 x=0.5')
run(768, 64, 4096, reps=10, label='768 x 64x64 (one resident wave of replicas)')
run(4096, 64, 2048, reps=20, label='config5: 4096 x 64x64')
run(4608, 64, 2048, reps=20, label='config3 shape: 4608 x 64x64 (sweep is per-concentration)')
run(1, 1024, 20000, label='config4: single 1024x1024')
run(8, 1024, 20000, label='8 x 1024x1024')
run(64, 256, 20000, label='64 x 256x256')
