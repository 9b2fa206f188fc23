"""GPU parity AT the decision boundary of the selection, and in volume.

The reference picks ``searchsorted(cumsum(P) / sum(P), rho)`` (hex_LiCoO2_kmc.py:486-492): the first index whose
sequentially accumulated sum, divided by numpy's pairwise sum, reaches rho.  The engine certifies a hierarchical pick
and replays the reference literally when it cannot (kmc_exact_par.h).  A generic rho never tests
the replay's tie handling: here every draw is generated WITH THE ORACLE IN THE LOOP so that it sits exactly on, one
ulp below or one ulp above a value of the reference's own cdf -- the only inputs for which the stopping value c*, the
binade events and the "first such index" argument of the certification can be wrong.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle as CO          # noqa: E402
from oracle import kmc_oracle as O         # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
RTOL = 1e-12

KINDS = ('cdf[k]', 'below cdf[k]', 'above cdf[k]', 'cdf[k-1]', 'cdf[-1]', 'generic')


@pytest.fixture(scope='module')
def kmc():
    import kmc_amd
    return kmc_amd


@pytest.fixture(scope='module')
def base_tables():
    return np.load(os.path.join(GOLDEN, 'tables.npz'))


@pytest.fixture(scope='module')
def dense_tables(base_tables):
    return {'normcons': O.build_hex_table(base_tables['base512_normcons']),
            'master': O.build_hex_table(base_tables['base512_master'])}


def boundary_run(rep, n_hops, presim, rs):
    """n_hops hops of the C oracle, each with a draw placed on the reference's own cdf (:488-492).  Returns the draws
    (n_hops, 2), the selected indices and the kind of each draw."""
    nx, ny = rep.nx, rep.ny
    draws = np.empty((n_hops, 2))
    draws[:, 1] = rs.rand(n_hops) if not presim else 1.0
    idx = np.empty(n_hops, dtype=np.int32)
    kinds = np.empty(n_hops, dtype=np.int32)
    for h in range(n_hops):
        P4 = rep.P.reshape(nx, ny, 1, 6)
        cdf = np.cumsum(P4) / np.sum(P4)                       # the reference's own expressions (:486-489)
        pos = np.flatnonzero(rep.P.reshape(-1) > 0)
        k = int(pos[rs.randint(pos.size)])
        kind = h % len(KINDS)
        if kind == 0:
            rho = cdf[k]
        elif kind == 1:
            rho = np.nextafter(cdf[k], 0.0)
        elif kind == 2:
            rho = np.nextafter(cdf[k], 2.0)
            if rho > cdf[-1]:                                  # beyond the last entry: the reference's IndexError (B4),
                rho, kind = cdf[k], 0                          # covered by test_draw_above_the_cdf
        elif kind == 3:
            rho = cdf[k - 1] if k > 0 else cdf[k]
        elif kind == 4:
            rho = cdf[-1]
        else:
            rho = rs.rand() * cdf[-1]
        if not rho > 0.0:                                      # rho == 0 "hops" entry 0 whatever its rate (B4): its own
            rho, kind = cdf[k], 0                              # test, test_gpu_parity.py::test_selection_edges
        draws[h, 0] = rho
        done, sel, _ = rep.run(draws[h:h + 1], presim)
        assert done == 1
        # np.searchsorted of the reference itself, on the cdf the draw was placed on
        assert sel[0] == int(np.searchsorted(cdf, rho))
        idx[h] = sel[0]
        kinds[h] = kind
    return draws, idx, kinds


CASES = [
    # nx, ny, table, hop kernel (kmc_set_hop_kernel), hops presim, hops sim
    (64, 64, 'normcons', 5, 240, 480),
    (64, 64, 'master', 5, 240, 480),
    (30, 30, 'master', 5, 240, 480),
    (64, 64, 'master', 6, 240, 480),
    (30, 30, 'normcons', 6, 240, 480),
    (64, 64, 'normcons', 3, 240, 480),
    (64, 64, 'master', 3, 240, 480),
    (64, 64, 'normcons', 2, 120, 240),
    (64, 64, 'master', 1, 120, 240),
    (30, 30, 'normcons', 4, 240, 480),
    (30, 30, 'master', 4, 240, 480),
    (30, 30, 'master', 1, 120, 240),
    (70, 66, 'normcons', 0, 120, 240),
    (70, 66, 'master', 0, 120, 240),
    (70, 66, 'master', 1, 60, 120),
    (300, 260, 'normcons', 0, 36, 72),
    (300, 260, 'master', 0, 36, 72),
    (300, 260, 'master', 1, 18, 36),
]
KERNEL_NAMES = {(64, 5): 'lat', (30, 5): 'lat', (64, 6): 'lat2', (30, 6): 'lat2', (64, 3): 'q64', (64, 2): 'lds64', (64, 1): 'generic', (30, 4): 'qany', (30, 1): 'generic',
                (70, 0): 'big', (70, 1): 'global', (300, 0): 'big', (300, 1): 'global'}


@pytest.mark.parametrize('force', [0, 1, 2], ids=['certified', 'exact-long', 'exact-short'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: '%dx%d-%s-%s' % (c[0], c[1], c[2], KERNEL_NAMES[(c[0], c[3])]))
def test_draws_on_the_cdf(kmc, base_tables, dense_tables, case, force):
    nx, ny, table, which, hp, hs = case
    if force == 2 and nx <= 64:
        pytest.skip('the short replay form belongs to the large-lattice kernels')
    dense = dense_tables[table]
    rs = np.random.RandomState(nx * 131 + ny + which * 7 + (table == 'master'))
    conc = 0.5 if table == 'normcons' else 0.6
    occ = (rs.rand(nx, ny) < conc).astype(np.int8)
    rep = CO.Replica(occ, dense)
    d_pre, i_pre, k_pre = boundary_run(rep, hp, True, rs)
    rep.init_tags()
    d_sim, i_sim, k_sim = boundary_run(rep, hs, False, rs)
    n_boundary = int((k_pre != 5).sum() + (k_sim != 5).sum())
    with kmc.KMCEngine(1, nx, ny) as eng:
        eng.set_base_table(base_tables['base512_' + table], kmc.tables.dumbbell_rate())
        eng.set_hop_kernel(which=which)
        eng.set_debug(force_exact=force)
        eng.set_lattice(occ[None])
        eng.rates_full()
        g1 = eng.run(hp, presim=True, draws=d_pre[None], want_trace=True)[0]
        eng.init_displacements()
        g2 = eng.run(hs, presim=False, draws=d_sim[None], want_trace=True)[0]
        bad = np.flatnonzero(np.concatenate([g1, g2]) != np.concatenate([i_pre, i_sim]))
        assert bad.size == 0, 'first mismatch at hop %d (%s): got %d, reference %d' % (
            bad[0], KINDS[np.concatenate([k_pre, k_sim])[bad[0]]], np.concatenate([g1, g2])[bad[0]],
            np.concatenate([i_pre, i_sim])[bad[0]])
        assert np.array_equal(eng.get_lattice()[0], rep.occ)
        assert np.array_equal(eng.get_reference_tags()[0], rep.tags)
        assert np.array_equal(eng.get_rates()[0], rep.P)
        np.testing.assert_allclose(eng.scalars()['time'][0], rep.time.value, rtol=RTOL)
        st = eng.status()
        assert st['status'][0] == 0 and st['hops_done'][0] == hp + hs
        if force:
            assert st['n_exact'][0] == hp + hs
        else:
            # a draw on (or one ulp off) a cdf value can never be certified: each of them must have taken the replay
            assert n_boundary <= st['n_exact'][0] <= hp + hs


def test_draw_above_the_cdf(kmc, base_tables, dense_tables):
    """B4 on the master fold: rho one ulp above cdf[-1] is the reference's IndexError: the replica stops with status 1 at
    that hop, every kernel family.  (rho == 0: tests/test_gpu_parity.py::test_selection_edges.)"""
    dense = dense_tables['master']
    for (nx, ny, which) in [(64, 64, 5), (30, 30, 5), (64, 64, 6), (64, 64, 3), (30, 30, 4), (30, 30, 1), (70, 66, 0), (70, 66, 1)]:
        rs = np.random.RandomState(nx + which)
        occ = (rs.rand(nx, ny) < 0.5).astype(np.int8)
        rep = CO.Replica(occ, dense)
        draws = np.stack([rs.rand(8), np.ones(8)], axis=1)
        done, sel5, _ = rep.run(draws[:5], True)
        assert done == 5
        P4 = rep.P.reshape(nx, ny, 1, 6)                       # the state the sixth draw meets
        top = (np.cumsum(P4) / np.sum(P4))[-1]
        draws[5, 0] = np.nextafter(top, 2.0)
        done, _, _ = rep.run(draws[5:], True)
        assert done == 0
        sel = sel5
        with kmc.KMCEngine(1, nx, ny) as eng:
            eng.set_base_table(base_tables['base512_master'], kmc.tables.dumbbell_rate())
            eng.set_hop_kernel(which=which)
            eng.set_lattice(occ[None])
            eng.rates_full()
            try:
                got = eng.run(8, presim=True, draws=draws[None], want_trace=True)[0]
            except kmc.engine.ReplicaStopped:
                got = None
            st = eng.status()
            assert st['status'][0] == 1 and st['hops_done'][0] == 5, (nx, ny, which, st)
            if got is not None:
                assert np.array_equal(got[:5], sel[:5]) and (got[5:] == -1).all()
            assert np.array_equal(eng.get_lattice()[0], rep.occ)


def _oracle_replica(args):
    occ, dense, d_pre, d_sim = args
    rep = CO.Replica(occ, dense)
    _, i1, _ = rep.run(d_pre, True)
    rep.init_tags()
    _, i2, _ = rep.run(d_sim, False)
    return i1, i2, rep.occ, rep.tags, rep.time.value


@pytest.mark.parametrize('nx,R,hops', [(64, 256, 20000), (30, 256, 20000)], ids=['64x64', '30x30'])
def test_ten_million_hops_against_the_oracle(kmc, base_tables, dense_tables, nx, R, hops):
    """count mismatches against the CPU oracle over >= 1e7 hops and require 0.  Two ensembles of 256 replicas x
    20 000 hops (5.1e6 hops each; both parametrisations together 1.02e7), every hop compared: the C oracle is farmed
    over the box's host threads (ctypes releases the GIL), one replica per task, as bench.py's cpu_baseline does."""
    dense = dense_tables['normcons']
    rs = np.random.RandomState(nx)
    conc = np.linspace(0.2, 0.8, R)
    occ = np.stack([(rs.rand(nx, nx) < c).astype(np.int8) for c in conc])
    hp = hops // 4
    hs = hops - hp
    d_pre = np.stack([rs.rand(R, hp), np.ones((R, hp))], axis=-1)
    d_sim = rs.rand(R, hs, 2)
    with kmc.KMCEngine(R, nx) as eng:
        eng.set_base_table(base_tables['base512_normcons'], kmc.tables.dumbbell_rate())
        eng.set_lattice(occ)
        eng.rates_full()
        g1 = eng.run(hp, presim=True, draws=d_pre, want_trace=True)
        eng.init_displacements()
        g2 = eng.run(hs, presim=False, draws=d_sim, want_trace=True)
        lat, tags, t = eng.get_lattice(), eng.get_reference_tags(), eng.scalars()['time']
        assert not eng.status()['status'].any()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        out = list(pool.map(_oracle_replica, [(occ[r], dense, d_pre[r], d_sim[r]) for r in range(R)]))
    mismatches = 0
    for r in range(R):
        i1, i2, o, tg, tm = out[r]
        mismatches += int((g1[r] != i1).sum() + (g2[r] != i2).sum())
        assert np.array_equal(lat[r], o) and np.array_equal(tags[r], tg), r
        np.testing.assert_allclose(t[r], tm, rtol=RTOL)
    assert mismatches == 0, '%d of %d hops differ from the oracle' % (mismatches, R * hops)
