"""One-off volume check (not in the test suite): R replicas x H hops through a hop kernel, EVERY selection compared with the
C oracle farmed over the host threads (tests/test_gpu_boundary.py::test_ten_million_hops_against_the_oracle, ten times
the volume and on the headline kernel).      python scripts/soak_vs_oracle.py [R] [hops] [which] [nx]      (GPU box)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmc_amd
from oracle import c_oracle as CO
from oracle import kmc_oracle as O

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nx = int(sys.argv[4]) if len(sys.argv) > 4 else 64
t = np.load(os.path.join(ROOT, 'tests/golden/tables.npz'))
dense = O.build_hex_table(t['base512_normcons'])
rs = np.random.RandomState(R + H)
conc = np.linspace(0.2, 0.8, R)
occ = np.stack([(rs.rand(nx, nx) < c).astype(np.int8) for c in conc])
hp = H // 4
hs = H - hp
d_pre = np.stack([rs.rand(R, hp), np.ones((R, hp))], axis=-1)
d_sim = rs.rand(R, hs, 2)
t0 = time.time()
with kmc_amd.KMCEngine(R, nx) as eng:
    eng.set_base_table(t['base512_normcons'], kmc_amd.tables.dumbbell_rate())
    eng.set_hop_kernel(which=which)
    name = eng.hop_kernel_name
    eng.set_lattice(occ)
    eng.rates_full()
    g1 = eng.run(hp, presim=True, draws=d_pre, want_trace=True)
    eng.init_displacements()
    g2 = eng.run(hs, presim=False, draws=d_sim, want_trace=True)
    lat, tags, tm = eng.get_lattice(), eng.get_reference_tags(), eng.scalars()['time']
    st = eng.status()
print('%s: %d replicas x %d hops on the GPU in %.1f s (exact-path hops %d, stopped %d)' % (name, R, H, time.time() - t0, int(st['n_exact'].sum()), int((st['status'] != 0).sum())), flush=True)


def one(r):
    rep = CO.Replica(occ[r], dense)
    _, i1, _ = rep.run(d_pre[r], True)
    rep.init_tags()
    _, i2, _ = rep.run(d_sim[r], False)
    bad = int((g1[r] != i1).sum() + (g2[r] != i2).sum())
    ok = bool(np.array_equal(lat[r], rep.occ) and np.array_equal(tags[r], rep.tags))
    return bad, ok, abs(tm[r] - rep.time.value) / rep.time.value


t0 = time.time()
bad = 0
worst = 0.0
n_state_bad = 0
with ThreadPoolExecutor(max_workers=16) as pool:
    for k, (b, ok, e) in enumerate(pool.map(one, range(R))):
        bad += b
        n_state_bad += 0 if ok else 1
        worst = max(worst, e)
        if k % 128 == 127:
            print('  oracle: %d replicas done, %.0f s' % (k + 1, time.time() - t0), flush=True)
print('%d hops compared with the C oracle in %.0f s: %d selections differ, %d final lattices / tag arrays differ, clocks within %.2e'
      % (R * H, time.time() - t0, bad, n_state_bad, worst))
sys.exit(0 if bad == 0 and n_state_bad == 0 and worst < 1e-12 else 1)
