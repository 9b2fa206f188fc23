"""GPU parity: the HIP engine (through the C ABI) against the oracle.

Bars (north star): lattice state, keys and hop selections bit-exact; rates equal
to the reference table (<= 1e-12 relative, in practice bit-exact); clocks and
diffusion coefficients within 1e-12 relative.
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle as CO          # noqa: E402
from oracle import kmc_oracle as O         # noqa: E402
from oracle import philox as PH            # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, 'traj_*.npz')))
RTOL = 1e-12                               # floating-point bar of the north star


@pytest.fixture(scope='module')
def kmc():
    import kmc_amd
    return kmc_amd


@pytest.fixture(scope='module')
def tables():
    t = np.load(os.path.join(GOLDEN, 'tables.npz'))
    with open(os.path.join(GOLDEN, 'tables.json')) as f:
        info = json.load(f)
    return t, info


@pytest.fixture(scope='module')
def dense(tables):
    return O.build_hex_table(tables[0]['base512_normcons'])


@pytest.fixture(scope='module')
def dense_master(tables):
    return O.build_hex_table(tables[0]['base512_master'])


def make_engine(kmc, tables, R, nx, ny=None, mode='hex_ref', table='normcons'):
    eng = kmc.KMCEngine(R, nx, ny, window_mode=mode)
    eng.set_base_table(tables[0]['base512_' + table], kmc.tables.dumbbell_rate())
    return eng


# ---------------------------------------------------------------- row T
def test_table_fold_bit_exact(kmc, tables, dense):
    t, info = tables
    with make_engine(kmc, tables, 1, 8) as eng:
        got = eng.get_hex_table()
    assert hashlib.sha256(got.astype('<f8').tobytes()).hexdigest() == info['dense_sha256']
    assert np.array_equal(got[t['sample_rows']], t['sample_rates'])
    with kmc.KMCEngine(1, 8) as eng:
        eng.set_base_table(t['base512_master'], kmc.tables.dumbbell_rate())
        got = eng.get_hex_table()
        assert np.array_equal(got, O.build_hex_table(t['base512_master']))
        eng.set_hex_table(dense)
        assert np.array_equal(eng.get_hex_table(), dense)


def test_rate_generation(kmc, tables):
    """Row G: 512 rates from the cluster expansion, <= 1e-12 of the reference."""
    t, _ = tables
    with kmc.KMCEngine(1, 8) as eng:
        base, en = eng.generate_base_table(kmc.tables.LCE_ECI, 10 ** 13, 0.025)
    assert np.array_equal(en, t['base512_energies'])
    np.testing.assert_allclose(base, t['base512_normcons'], rtol=RTOL, atol=0)
    np.testing.assert_allclose(base, t['base512_regenerated'], rtol=RTOL, atol=0)


# ---------------------------------------------------------------- rows R1-R3
@pytest.mark.parametrize('shape', [(10, 10), (30, 30), (64, 64), (8, 12), (5, 7), (100, 37), (3, 3), (130, 70), (64, 30), (61, 64),
                                   (33, 16), (70, 5), (5, 100), (256, 256), (300, 260)])
def test_rates_full(kmc, tables, dense, shape):
    nx, ny = shape
    rs = np.random.RandomState(nx * 1000 + ny)
    R = 5
    occ = (rs.rand(R, nx, ny) < rs.rand(R, 1, 1)).astype(np.int8)
    occ[0] = 0
    occ[1] = 1
    with make_engine(kmc, tables, R, nx, ny) as eng:
        eng.set_lattice(occ)
        eng.rates_full()
        keys = eng.get_keys()
        P = eng.get_rates()
        assert np.array_equal(eng.get_lattice(), occ)
        for r in range(R):
            assert np.array_equal(keys[r], O.all_keys(occ[r]))
            assert np.array_equal(P[r], O.rates_full(occ[r], dense))
        # S1: np.sum with numpy's association
        g = eng.gamma_exact()
        for r in range(R):
            assert g[r] == float(np.sum(P[r].reshape(nx, ny, 1, 6)))
        # R1: single-site query
        for (i, j) in [(0, 0), (nx - 1, ny - 1), (nx // 2, 1)]:
            k, rates = eng.site_probability(2, i, j)
            assert k == O.site_key(occ[2], i, j)
            assert np.array_equal(rates, dense[k])


def test_rates_full_many_tiles(kmc, tables, dense):
    """Rows R1, R2 on an ensemble large enough for the 28-row tiles of rates_tile_kernel (>= 8192 of them), and on an
    ensemble of 30 x 30 lattices packed two to a wave: a sample of the replicas against the oracle."""
    for (R, nx, ny) in ((1500, 70, 66), (1031, 30, 30), (517, 9, 13)):
        rs = np.random.RandomState(R)
        occ = (rs.rand(R, nx, ny) < rs.rand(R, 1, 1)).astype(np.int8)
        with make_engine(kmc, tables, R, nx, ny) as eng:
            eng.set_lattice(occ)
            eng.rates_full()
            keys = eng.get_keys()
        for r in [0, 1, R - 2, R - 1] + list(rs.randint(0, R, 12)):
            assert np.array_equal(keys[r], O.all_keys(occ[r])), (R, nx, ny, r)


@pytest.mark.parametrize('mode', ['hex_ref', 'obj_ref', 'full'])
def test_refresh_window_standalone(kmc, tables, dense, mode):
    rs = np.random.RandomState(5)
    nx, ny, R = 12, 9, 6
    occ = (rs.rand(R, nx, ny) < 0.5).astype(np.int8)
    with make_engine(kmc, tables, R, nx, ny, mode) as eng:
        eng.set_lattice(occ)
        eng.rates_full()
        occ2 = occ.copy()
        se = np.zeros((R, 4), dtype=np.int32)
        P_ref = []
        for r in range(R):
            d = r % 6
            i, j = (0, 0) if r == 0 else (rs.randint(nx), rs.randint(ny))
            se[r] = (i, j, i + O.HOP_DIRS[d][0], j + O.HOP_DIRS[d][1])
            occ2[r, i, j] = 0
            occ2[r, se[r, 2] % nx, se[r, 3] % ny] = 1
            P = O.rates_full(occ[r], dense)
            O.rates_window(occ2[r], P, dense, (i, j), (se[r, 2], se[r, 3]), mode)
            P_ref.append(P)
        se[5, 0] = -1                     # skipped replica
        P_ref[5] = O.rates_full(occ[5], dense)
        eng.set_lattice(occ2)
        eng.set_keys(np.stack([O.all_keys(o) for o in occ]))
        eng.refresh_window(se)
        assert np.array_equal(eng.get_rates(), np.stack(P_ref))


# ---------------------------------------------------------------- hop loop vs golden
def _split_draws(g):
    nx, ny, conc = int(g['nx']), int(g['ny']), float(g['conc'])
    n_hops = int(nx * ny * conc)
    npre, nsim = int(g['n_presim']) * n_hops, int(g['n_sim']) * n_hops
    d = g['draws']
    d_pre = np.stack([d[:npre], np.ones(npre)], axis=1)
    d_sim = d[npre:].reshape(nsim, 2)
    return n_hops, d_pre, d_sim


@pytest.mark.parametrize('force_exact', [False, True], ids=['certified', 'exact'])
@pytest.mark.parametrize('path', TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_golden_trajectory(kmc, tables, dense, dense_master, path, force_exact):
    """Every trajectory generated by the reference's own text: the functional path on the normcons fold (10 x 10 ...
    64 x 64, where np.sum crosses three buffers), on the master_dictionary.p fold (BASELINE config 1: rates over 20
    decades), the hybrid OO-window scenario and the class's own text (Mc_Lattice.py:60-232)."""
    g = np.load(path)
    nx, ny, conc = int(g['nx']), int(g['ny']), float(g['conc'])
    mode = 'obj_ref' if int(g['oo_window']) else 'hex_ref'
    n_hops, d_pre, d_sim = _split_draws(g)
    table = str(g['table'])
    if table == 'master':
        dense = dense_master
    with make_engine(kmc, tables, 1, nx, ny, mode, table) as eng:
        eng.set_debug(force_exact=force_exact)
        eng.set_lattice(g['occ0'][None])
        eng.rates_full()
        assert np.array_equal(eng.get_rates()[0], g['P0'])
        idx1 = eng.run(len(d_pre), presim=True, draws=d_pre[None], want_trace=True)
        eng.init_displacements()
        # one kmc_step per call, like the reference's loop (:745-758)
        idx2, step_times = [], []
        for s in range(int(g['n_sim'])):
            idx2.append(eng.run(n_hops, presim=False, draws=d_sim[None, s * n_hops:(s + 1) * n_hops], want_trace=True))
            step_times.append(eng.scalars()['step_time'][0])
        idx = np.concatenate([idx1[0]] + [i[0] for i in idx2])
        assert np.array_equal(idx, g['idx'])
        assert np.array_equal(eng.get_lattice()[0], g['occ_final'])
        assert np.array_equal(eng.get_reference_tags()[0], g['tags_final'])
        assert np.array_equal(eng.get_rates()[0], g['P_final'])
        sc = eng.scalars()
        assert np.array_equal(sc['hist'][0], g['hist'].astype(np.int64))
        np.testing.assert_allclose(step_times, g['step_times'], rtol=RTOL)
        np.testing.assert_allclose(sc['time'][0], float(g['total_time']), rtol=RTOL)
        ds, dj, msd = eng.diffusion(conc)
        np.testing.assert_allclose(ds[0], float(g['d_star']), rtol=RTOL)
        np.testing.assert_allclose(dj[0], float(g['d_j']), rtol=RTOL)
        st = eng.status()
        assert st['status'][0] == 0 and st['hops_done'][0] == idx.size
        assert st['n_exact'][0] == (idx.size if force_exact else 0)
        bar = eng.barrier_stats()
        if not g['barriers'].size:
            pass
        elif force_exact:
            # exact path evaluates the reference's own (cancelling) expression
            np.testing.assert_allclose(bar['last'][0], g['barriers'][-1], rtol=RTOL)
        else:
            np.testing.assert_allclose(bar['last'][0], g['barriers'][-1], rtol=1e-6)
        stale = np.any(eng.get_rates()[0] != O.rates_full(g['occ_final'], dense), axis=-1)
        assert np.array_equal(stale, g['stale_mask'])
        np.testing.assert_allclose(sc['gamma'][0], np.sum(g['P_final']), rtol=RTOL)


@pytest.mark.parametrize('force_exact', [False, True], ids=['certified', 'exact'])
@pytest.mark.parametrize('name,which', [('n64_x05', 1), ('n64_x05', 2), ('n64_x05', 3), ('n30_x05_master', 1), ('n30_x05_master', 4),
                                        ('n30_x05', 1), ('n30_x05', 4), ('n8x12_x05', 4),
                                        ('n70x66_x04_master', 0), ('n70x66_x04_master', 1),
                                        ('n64_x05', 5), ('n30_x05_master', 5), ('n30_x05', 5), ('n8x12_x05', 5), ('n10_x03_master', 5),
                                        ('n64_x05', 6), ('n30_x05_master', 6), ('n8x12_x05', 6)],
                         ids=['n64-generic', 'n64-lds64', 'n64-q64', 'n30master-generic', 'n30master-q', 'n30-generic', 'n30-q', 'n8x12-q',
                              'n70x66master-big', 'n70x66master-global', 'n64-lat', 'n30master-lat', 'n30-lat', 'n8x12-lat',
                              'n10master-lat', 'n64-lat2', 'n30master-lat2', 'n8x12-lat2'])
def test_reference_trajectory_every_kernel_family(kmc, tables, dense, dense_master, name, which, force_exact):
    """The reference-generated trajectories at BASELINE's lattice sizes through EVERY hop kernel that can take them
    (kmc_set_hop_kernel), certified and with every hop forced through the exact replay: selections, lattice, tags,
    rate matrix bit for bit; clocks to 1e-12."""
    g = np.load(os.path.join(GOLDEN, 'traj_%s.npz' % name))
    nx, ny, conc = int(g['nx']), int(g['ny']), float(g['conc'])
    n_hops, d_pre, d_sim = _split_draws(g)
    with make_engine(kmc, tables, 1, nx, ny, 'hex_ref', str(g['table'])) as eng:
        eng.set_hop_kernel(which=which)
        eng.set_debug(force_exact=force_exact)
        eng.set_lattice(g['occ0'][None])
        eng.rates_full()
        idx1 = eng.run(len(d_pre), presim=True, draws=d_pre[None], want_trace=True)
        eng.init_displacements()
        idx2 = eng.run(len(d_sim), presim=False, draws=d_sim[None], want_trace=True)
        assert np.array_equal(np.concatenate([idx1[0], idx2[0]]), g['idx'])
        assert np.array_equal(eng.get_lattice()[0], g['occ_final'])
        assert np.array_equal(eng.get_reference_tags()[0], g['tags_final'])
        assert np.array_equal(eng.get_rates()[0], g['P_final'])
        np.testing.assert_allclose(eng.scalars()['time'][0], float(g['total_time']), rtol=RTOL)
        np.testing.assert_allclose(eng.gamma_exact()[0], np.sum(g['P_final']), rtol=0)     # np.sum's own bits (row S1)
        ds, dj, _ = eng.diffusion(conc)
        np.testing.assert_allclose([ds[0], dj[0]], [float(g['d_star']), float(g['d_j'])], rtol=RTOL)
        assert eng.status()['n_exact'][0] == (g['idx'].size if force_exact else 0)


# ---------------------------------------------------------------- hop loop vs C oracle
def _oracle_run(occ, dense, mode, draws_pre, draws_sim, conc):
    rep = CO.Replica(occ, dense, mode)
    _, i1, _ = rep.run(draws_pre, True)
    rep.init_tags()
    _, i2, _ = rep.run(draws_sim, False)
    return rep, np.concatenate([i1, i2])


@pytest.mark.parametrize('cfg', [
    # nx, ny, R, hops_pre, hops_sim, mode, margin_scale
    (64, 64, 6, 1500, 3000, 'hex_ref', 1.0),
    (64, 64, 3, 500, 1500, 'full', 1.0),
    (64, 64, 3, 500, 1500, 'obj_ref', 1.0),
    (30, 30, 8, 900, 2000, 'hex_ref', 1.0),
    (64, 33, 4, 400, 1200, 'hex_ref', 1.0),
    (5, 5, 4, 100, 400, 'hex_ref', 1.0),
    (6, 64, 4, 100, 600, 'full', 1.0),
    # the 16-per-CU layout at any size (hop_q64_kernel<., true>): a width that is no multiple of 16 (a window that wraps
    # touches three quarters), 61 ... 63 rows (the bitmap's third dword is read), the smallest lattice it takes
    (62, 35, 4, 600, 1500, 'hex_ref', 1.0),
    (63, 17, 3, 400, 1200, 'full', 1.0),
    (61, 34, 3, 400, 1200, 'obj_ref', 1.0),
    (8, 8, 4, 100, 500, 'hex_ref', 1.0),
    (9, 61, 3, 200, 800, 'obj_ref', 1.0),
    (40, 20, 3, 300, 900, 'full', 1.0),
    (35, 62, 3, 400, 1200, 'hex_ref', 3e7),
    (8, 64, 3, 200, 800, 'full', 1.0),            # a 7-row window on 8 rows
    (64, 8, 3, 200, 800, 'obj_ref', 1.0),         # a 6-column window on 8 columns
    (16, 16, 3, 150, 600, 'full', 1.0),
    (48, 33, 3, 400, 1200, 'hex_ref', 1.0),
    (33, 48, 3, 400, 1200, 'obj_ref', 1.0),
    # inflated margin: a few percent of the hops take the exact path, mixed in
    (30, 30, 4, 600, 1500, 'hex_ref', 3e7),
    (64, 64, 2, 100, 300, 'hex_ref', 1.5e6),
], ids=lambda c: '%dx%d-%s-m%g' % (c[0], c[1], c[5], c[6]))
def test_random_ensemble_vs_oracle(kmc, tables, dense, cfg):
    nx, ny, R, hp, hs, mode, mscale = cfg
    rs = np.random.RandomState(nx * 7 + ny + R)
    conc = np.linspace(0.15, 0.85, R)
    occ = np.stack([(rs.rand(nx, ny) < c).astype(np.int8) for c in conc])
    d_pre = np.stack([rs.rand(R, hp), np.ones((R, hp))], axis=-1)
    d_sim = rs.rand(R, hs, 2)
    oracle = [_oracle_run(occ[r], dense, mode, d_pre[r], d_sim[r], conc[r]) for r in range(R)]
    # 64 x 64 has two kernels (16 replicas per CU, the default, and 8): both; 8 ... 64 rows and columns: the 16-per-CU
    # layout (the default) and the any-size kernel with block sums in LDS
    # (5: the one-replica-per-CU kernel, kmc_hop_lat.h -- what a context of up to 256 replicas takes by default)
    for which in ((3, 2, 5, 6) if (nx, ny) == (64, 64) else ((4, 1, 5, 6) if min(nx, ny) >= 8 else (0,))):
        _ensemble_vs_oracle(kmc, tables, cfg, which, occ, d_pre, d_sim, oracle)


def _ensemble_vs_oracle(kmc, tables, cfg, which, occ, d_pre, d_sim, oracle):
    nx, ny, R, hp, hs, mode, mscale = cfg
    with make_engine(kmc, tables, R, nx, ny, mode) as eng:
        eng.set_hop_kernel(which=which)
        eng.set_debug(force_exact=False, margin_scale=mscale)
        eng.set_lattice(occ)
        eng.rates_full()
        i1 = eng.run(hp, presim=True, draws=d_pre, want_trace=True)
        eng.init_displacements()
        i2 = eng.run(hs, presim=False, draws=d_sim, want_trace=True)
        got_idx = np.concatenate([i1, i2], axis=1)
        lat, tags, P, sc, st = eng.get_lattice(), eng.get_reference_tags(), eng.get_rates(), eng.scalars(), eng.status()
        ds, dj, _ = eng.diffusion(occ.reshape(R, -1).mean(axis=1))
        gex = eng.gamma_exact()
    for r in range(R):
        rep, idx = oracle[r]
        assert np.array_equal(got_idx[r], idx), 'replica %d' % r
        assert np.array_equal(lat[r], rep.occ)
        assert np.array_equal(tags[r], rep.tags)
        assert np.array_equal(P[r], rep.P)
        assert np.array_equal(sc['hist'][r], rep.hist.astype(np.int64))
        np.testing.assert_allclose(sc['time'][r], rep.time.value, rtol=RTOL)
        assert gex[r] == float(np.sum(rep.P.reshape(nx, ny, 1, 6)))
        np.testing.assert_allclose(sc['gamma'][r], gex[r], rtol=RTOL)
        out = rep.diffusion(float(occ[r].mean()))
        np.testing.assert_allclose(ds[r], out[0], rtol=RTOL)
        np.testing.assert_allclose(dj[r], out[1], rtol=RTOL)
    assert not st['status'].any()
    if mscale > 1:
        assert st['n_exact'].sum() > 0, 'inflated margin should exercise the exact path'
        assert st['n_exact'].sum() < 0.5 * R * (hp + hs)
    else:
        assert st['n_exact'].sum() <= 1


# ---------------------------------------------------------------- 64 x 64 kernel vs any-size kernel
@pytest.mark.parametrize('mode', ['hex_ref', 'obj_ref', 'full'])
def test_lds64_kernel_equals_generic_kernel(kmc, tables, mode):
    """The 8-per-CU 64 x 64 kernel (kmc_hop_lds64.h) keeps every sum's association, so traces, state,
    gamma and clocks must be bit-identical to the any-size LDS kernel (which the oracle tests above
    pin), here over 10^6 hops, all concentrations, certified and mixed exact-path modes.  The default
    16-per-CU kernel (kmc_hop_q64.h) sums a row's sites in another fixed order: selections, lattices,
    keys, displacements and histograms are identical, clocks and gamma agree to rounding."""
    R, hp, hs, seed = 72, 4096 + 37, 10000, 0xC0FFEE
    conc = np.repeat(np.linspace(0.1, 0.9, 9), 8)
    out = []
    for which in (3, 2, 1, 5, 6):                # q64, lds64, generic, one-per-CU, two-per-CU
        with make_engine(kmc, tables, R, 64, 64, mode) as eng:
            eng.set_hop_kernel(which=which)
            eng.fill_random(conc, 99)
            eng.rates_full()
            i1 = eng.run(hp, presim=True, seed=seed, want_trace=True)
            eng.init_displacements()
            i2 = eng.run(hs, presim=False, seed=seed, want_trace=True)
            eng.set_debug(force_exact=False, margin_scale=2e6)
            i3 = eng.run(300, presim=False, seed=seed, want_trace=True)
            sc, st = eng.scalars(), eng.status()
            out.append((i1, i2, i3, eng.get_lattice(), eng.get_keys(), eng.get_displacements(), sc['hist'], st['status'],
                        sc['time'], sc['step_time'], sc['gamma'], st['n_exact'],
                        np.stack(list(eng.barrier_stats().values()))))
    q64, lds64, generic, lat, lat2 = out
    for a, b in zip(lds64, generic):
        assert np.array_equal(a, b)
    for a, b in zip(lat2, lat):                      # the two layouts of one kernel: the same sums, bit for bit
        assert np.array_equal(a, b)
    for other in (q64, lat):
        for k in range(8):                           # integer state: exact
            assert np.array_equal(other[k], generic[k])
        for k in (8, 9, 10):                         # clocks, gamma: a different (fixed) summation order
            np.testing.assert_allclose(other[k], generic[k], rtol=1e-13)
        # barrier statistics: the exact path evaluates the reference's cancelling expression (S5), and the
        # two kernels do not send the same hops through it
        np.testing.assert_allclose(other[12], generic[12], rtol=1e-5)
        assert other[11].sum() > 0 and not other[7].any()
    assert lds64[11].sum() > 0


@pytest.mark.parametrize('nx,ny,mode', [(30, 30, 'hex_ref'), (37, 53, 'obj_ref'), (61, 35, 'full'), (12, 9, 'hex_ref')],
                         ids=lambda v: str(v))
def test_any_size_q_kernel_equals_block_sum_kernel(kmc, tables, nx, ny, mode):
    """hop_q64_kernel<., ANY> (the 16-per-CU layout at any size, the default) against hop_lds_kernel (block sums in LDS, which
    the oracle tests pin) over 6 x 10^5 hops, all concentrations, certified and with an inflated margin: selections,
    lattices, keys, displacements and histograms identical; clocks and gamma agree to rounding (another fixed summation order)."""
    R, hp, hs, seed = 45, 1000 + 37, 12000, 0xFEED
    conc = np.repeat(np.linspace(0.1, 0.9, 9), 5)
    out = []
    for which in (4, 1, 5):
        with make_engine(kmc, tables, R, nx, ny, mode) as eng:
            eng.set_hop_kernel(which=which)
            eng.fill_random(conc, 5)
            eng.rates_full()
            i1 = eng.run(hp, presim=True, seed=seed, want_trace=True)
            eng.init_displacements()
            i2 = eng.run(hs, presim=False, seed=seed, want_trace=True)
            eng.set_debug(force_exact=False, margin_scale=5e6)
            i3 = eng.run(300, presim=False, seed=seed, want_trace=True)
            sc, st = eng.scalars(), eng.status()
            out.append((i1, i2, i3, eng.get_lattice(), eng.get_keys(), eng.get_displacements(), sc['hist'], st['status'],
                        sc['time'], sc['step_time'], sc['gamma'], st['n_exact']))
    q, blk, lat = out
    for other in (q, lat):
        if other is lat and min(nx, ny) < 8:
            continue
        for k in range(8):
            assert np.array_equal(other[k], blk[k])
        for k in (8, 9, 10):
            np.testing.assert_allclose(other[k], blk[k], rtol=1e-13)
        assert other[11].sum() > 0 and not other[7].any()


@pytest.mark.parametrize('nx,ny,kernels', [(64, 64, (3, 2, 0)), (30, 30, (4, 1, 0)), (37, 53, (4,)), (70, 66, (0, 1))],
                         ids=['64x64-q64-lds64', '30x30-q-blocksum', '37x53-q', '70x66-big-global'])
def test_dense_table_that_does_not_factor(kmc, tables, dense, nx, ny, kernels):
    """A user table (kmc_set_hex_table) need not depend on 13 key bits per direction only: the
    engine must notice and keep the dense gather; trajectories still match the oracle.  Every kernel family
    has a dense variant (template parameter COMPACT = false) or gathers from the dense table anyway."""
    rs = np.random.RandomState(5)
    dense2 = dense * (1.0 + 0.25 * rs.rand(*dense.shape))        # same zeros, key-dependent values
    R, hp, hs = 3, 700, 1300
    occ = np.stack([(rs.rand(nx, ny) < c).astype(np.int8) for c in (0.3, 0.5, 0.7)])
    d_pre = np.stack([rs.rand(R, hp), np.ones((R, hp))], axis=-1)
    d_sim = rs.rand(R, hs, 2)
    oracle = [_oracle_run(occ[r], dense2, 'hex_ref', d_pre[r], d_sim[r], None) for r in range(R)]
    for which in kernels:
        with kmc.KMCEngine(R, nx, ny) as eng:
            eng.set_hop_kernel(which=which)
            eng.set_base_table(tables[0]['base512_normcons'], kmc.tables.dumbbell_rate())
            assert eng.table_is_compact()
            eng.set_hex_table(dense2)
            assert not eng.table_is_compact()
            eng.set_lattice(occ)
            eng.rates_full()
            i1 = eng.run(hp, presim=True, draws=d_pre, want_trace=True)
            eng.init_displacements()
            i2 = eng.run(hs, presim=False, draws=d_sim, want_trace=True)
            got, lat, tags, sc = np.concatenate([i1, i2], axis=1), eng.get_lattice(), eng.get_reference_tags(), eng.scalars()
            eng.set_hex_table(dense)
            assert eng.table_is_compact()
        for r in range(R):
            rep, idx = oracle[r]
            assert np.array_equal(got[r], idx)
            assert np.array_equal(lat[r], rep.occ) and np.array_equal(tags[r], rep.tags)
            np.testing.assert_allclose(sc['time'][r], rep.time.value, rtol=RTOL)


# ---------------------------------------------------------------- RNG streams / fill
@pytest.mark.parametrize('nx,ny', [(30, 17), (64, 64), (70, 66)], ids=['30x17', '64x64-keys-in-registers', '70x66-keys-redrawn'])
def test_fill_random_matches_philox(kmc, tables, nx, ny):
    R = 7
    conc = np.array([0.0, 0.1, 0.33, 0.5, 0.77, 0.999, 1.0])
    with make_engine(kmc, tables, R, nx, ny) as eng:
        eng.fill_random(conc, 0xDEADBEEF12345)
        occ = eng.get_lattice()
    for r in range(R):
        assert occ[r].sum() == int(nx * ny * 1 * conc[r])
        assert np.array_equal(occ[r], PH.fill_lattice(0xDEADBEEF12345, r, nx, ny, conc[r]))


def test_philox_stream_run_matches_injected(kmc, tables, dense):
    """Production mode (draws = NULL) must consume exactly the documented Philox
    stream, continuing across kmc_run calls."""
    R, nx, seed = 5, 20, 0x1234567890ABCDEF
    with make_engine(kmc, tables, R, nx) as eng:
        eng.fill_random(0.4, 42)
        occ = eng.get_lattice()
        eng.rates_full()
        i1 = eng.run(300, presim=True, seed=seed, want_trace=True)
        eng.init_displacements()
        i2 = eng.run(500, presim=False, seed=seed, want_trace=True)
        i3 = eng.run(77, presim=False, seed=seed, want_trace=True)
        lat, tags, sc = eng.get_lattice(), eng.get_reference_tags(), eng.scalars()
    for r in range(R):
        d = PH.hop_draws(seed, r, 0, 877)
        rep = CO.Replica(occ[r], dense, 'hex_ref')
        _, j1, _ = rep.run(d[:300], True)
        rep.init_tags()
        _, j2, _ = rep.run(d[300:], False)
        assert np.array_equal(np.concatenate([i1[r], i2[r], i3[r]]), np.concatenate([j1, j2]))
        assert np.array_equal(lat[r], rep.occ) and np.array_equal(tags[r], rep.tags)
        np.testing.assert_allclose(sc['time'][r], rep.time.value, rtol=RTOL)


# ---------------------------------------------------------------- edges
@pytest.mark.parametrize('nx,ny,which', [(10, 10, 0), (64, 64, 3), (64, 64, 2), (70, 66, 0), (64, 64, 5), (10, 10, 4), (7, 9, 0), (64, 64, 6), (10, 10, 6)],
                         ids=['lat-10x10', 'q64', 'lds64', 'big-70x66', 'lat', 'q-10x10', 'lds-7x9', 'lat2', 'lat2-10x10'])
def test_selection_edges(kmc, tables, dense, nx, ny, which):
    """rho == 0 picks index 0 (searchsorted left); rho above cumsum[-1]/gamma is the
    reference's IndexError -> the replica stops with KMC_ST_OVERFLOW; a lattice
    with no possible hop stops with KMC_ST_ZERO_GAMMA.    Every hop-kernel family."""
    rs = np.random.RandomState(0)
    occ = (rs.rand(3, nx, ny) < 0.5).astype(np.int8)
    occ[2] = 1                                           # full lattice: gamma == 0
    occ[0, 0, 0] = 1                                     # entry 0 = site (0, 0) towards (1, 0): the first rho == 0 hop moves
    occ[0, 1, 0] = 1                                     # an atom onto an occupied site, the next ones leave a vacant site
    draws = rs.rand(3, 40, 2)
    draws[0, [10, 21, 22, 35], 0] = 0.0                  # entry 0 whatever its rate, several times in one chunk
    draws[1, 25, 0] = 1.0
    with make_engine(kmc, tables, 3, nx, ny) as eng:
        eng.set_hop_kernel(which=which)
        eng.set_lattice(occ)
        eng.rates_full()
        eng.init_displacements()
        idx = eng.run(40, presim=False, draws=draws, want_trace=True)
        st = eng.status()
        lat, tags, P = eng.get_lattice(), eng.get_reference_tags(), eng.get_rates()
    rep = CO.Replica(occ[0], dense)
    rep.init_tags()
    done, ref_idx, _ = rep.run(draws[0], False)
    assert done == 40 and (ref_idx[[10, 21, 22, 35]] == 0).all()
    assert np.array_equal(idx[0], ref_idx) and np.array_equal(lat[0], rep.occ)
    assert np.array_equal(tags[0], rep.tags) and np.array_equal(P[0], rep.P)
    rep1 = CO.Replica(occ[1], dense)
    rep1.init_tags()
    done1, ref1, _ = rep1.run(draws[1], False)
    if done1 < 40:                                       # cumsum[-1]/gamma < 1.0 for this state
        assert st['status'][1] == 1 and st['hops_done'][1] == done1
        assert np.array_equal(idx[1, :done1], ref1[:done1]) and (idx[1, done1:] == -1).all()
        assert np.array_equal(lat[1], rep1.occ) and np.array_equal(P[1], rep1.P)
    else:
        assert st['status'][1] == 0 and np.array_equal(idx[1], ref1)
    assert st['status'][2] == 2 and st['hops_done'][2] == 0 and (idx[2] == -1).all()
    assert st['status'][0] == 0


def test_errors_are_loud(kmc, tables):
    with kmc.KMCEngine(1, 10) as eng:
        with pytest.raises(kmc.KMCError, match='table'):
            eng.rates_full()
        eng.set_base_table(tables[0]['base512_normcons'], kmc.tables.dumbbell_rate())
        with pytest.raises(kmc.KMCError, match='rates_full'):
            eng.run(10)
        with pytest.raises(kmc.KMCError, match='0 or 1'):
            eng.set_lattice(np.full((1, 10, 10), -1, dtype=np.int8))
        with pytest.raises(kmc.KMCError, match='concentration is zero'):
            eng.diffusion(0.0)
    with pytest.raises(kmc.KMCError):
        kmc.KMCEngine(1, 2)
    with pytest.raises(kmc.KMCError, match='device'):
        kmc.KMCEngine(1, 10, device=99)


# ---------------------------------------------------------------- lattices beyond 64 x 64
@pytest.mark.parametrize('cfg', [
    # nx, ny, R, hops_pre, hops_sim, mode, force_exact, margin_scale
    (65, 65, 3, 300, 900, 'hex_ref', False, 1.0),
    (100, 37, 3, 300, 900, 'hex_ref', False, 1.0),
    (37, 100, 2, 200, 600, 'full', False, 1.0),
    (130, 70, 2, 200, 800, 'obj_ref', False, 1.0),
    (300, 260, 2, 200, 800, 'hex_ref', False, 1.0),
    (128, 520, 2, 200, 800, 'full', False, 1.0),
    (9, 1000, 2, 200, 800, 'hex_ref', False, 1.0),
    (70, 66, 2, 40, 120, 'hex_ref', True, 1.0),
    (300, 260, 1, 10, 50, 'hex_ref', True, 1.0),      # exact path over 76 tiles: binade crossings, events
    (90, 90, 2, 200, 600, 'hex_ref', False, 2e6),
    # force_exact = 2: every hop through the replay's short form (no np.sum pass; the large-lattice kernel's answer to a hop
    # it cannot certify), and with a margin inflated until the short form itself gives up now and then
    (70, 66, 2, 40, 160, 'hex_ref', 2, 1.0),
    (300, 260, 1, 10, 60, 'full', 2, 1.0),
    (130, 70, 1, 20, 100, 'hex_ref', 2, 3e9),
    (300, 260, 1, 100, 500, 'hex_ref', False, 2e6),
], ids=lambda c: '%dx%d-%s%s-m%g' % (c[0], c[1], c[5], {0: '', 1: '-exact', 2: '-exactshort'}[int(c[6])], c[7]))
@pytest.mark.parametrize('kernel', ['big', 'big-tree', 'global'])
def test_global_memory_kernel_vs_oracle(kmc, tables, dense, cfg, kernel):
    """Lattices beyond 64 x 64: hop_big_kernel (sum tree in L2, bitmaps in LDS; 'big' = with the per-site words and the
    palette of round 3, 'big-tree' = without) and the older hop_global_kernel (kept for lattices whose bitmaps do not
    fit LDS), all against the oracle."""
    nx, ny, R, hp, hs, mode, force, mscale = cfg
    rs = np.random.RandomState(nx + ny)
    conc = np.linspace(0.3, 0.7, R)
    occ = np.stack([(rs.rand(nx, ny) < c).astype(np.int8) for c in conc])
    d_pre = np.stack([rs.rand(R, hp), np.ones((R, hp))], axis=-1)
    d_sim = rs.rand(R, hs, 2)
    with make_engine(kmc, tables, R, nx, ny, mode) as eng:
        assert not eng.uses_lds_path
        eng.set_hop_kernel(which={'big': 0, 'big-tree': 2, 'global': 1}[kernel])
        assert eng.hop_kernel_name == {'big': 'hop_big_kernel<site words>', 'big-tree': 'hop_big_kernel', 'global': 'hop_global_kernel'}[kernel]
        eng.set_debug(force_exact=force, margin_scale=mscale)
        eng.set_lattice(occ)
        eng.rates_full()
        i1 = eng.run(hp, presim=True, draws=d_pre, want_trace=True)
        eng.init_displacements()
        i2 = eng.run(hs, presim=False, draws=d_sim, want_trace=True)
        got_idx = np.concatenate([i1, i2], axis=1)
        lat, tags, P, sc, st = eng.get_lattice(), eng.get_reference_tags(), eng.get_rates(), eng.scalars(), eng.status()
        ds, dj, _ = eng.diffusion(occ.reshape(R, -1).mean(axis=1))
    for r in range(R):
        rep, idx = _oracle_run(occ[r], dense, mode, d_pre[r], d_sim[r], conc[r])
        assert np.array_equal(got_idx[r], idx), 'replica %d' % r
        assert np.array_equal(lat[r], rep.occ) and np.array_equal(tags[r], rep.tags)
        assert np.array_equal(P[r], rep.P)
        assert np.array_equal(sc['hist'][r], rep.hist.astype(np.int64))
        np.testing.assert_allclose(sc['time'][r], rep.time.value, rtol=RTOL)
        np.testing.assert_allclose(sc['gamma'][r], np.sum(rep.P), rtol=RTOL)
        out = rep.diffusion(float(occ[r].mean()))
        np.testing.assert_allclose(ds[r], out[0], rtol=RTOL)
    assert not st['status'].any()
    if force:
        assert (st['n_exact'] == hp + hs).all()
    elif mscale > 1:
        assert 0 < st['n_exact'].sum() < 0.5 * R * (hp + hs)


def _random_hop_table(rs, all_li_row_zero=True):
    """A dense table of ~10^6 distinct rates that still describes hops: the rate of a direction is zero where its end point is
    occupied (key char 8, 4, 5, 9, 13, 12 for the six directions; char p is bit 17 - p)."""
    t = np.exp(rs.uniform(-20.0, 0.0, size=(1 << 18, 6)))
    keys = np.arange(1 << 18)
    for d, p in enumerate((8, 4, 5, 9, 13, 12)):
        t[((keys >> (17 - p)) & 1) == 1, d] = 0.0
    assert not t[0x3FFFF].any()
    if not all_li_row_zero:
        t[0x3FFFF] = 1e-3                    # never selected at x ~ 0.5 (a site with 18 Li neighbours), but no longer "vacant"
    return t


@pytest.mark.parametrize('nx,ny,R,hp,hs,zero', [(64, 64, 3, 200, 600, True), (30, 30, 3, 200, 600, True), (70, 66, 2, 100, 400, True),
                                               (6, 1100, 1, 100, 300, True), (64, 64, 2, 200, 400, False), (30, 30, 2, 200, 400, False),
                                               (70, 66, 1, 100, 300, False)],
                         ids=['64x64', '30x30', '70x66', '6x1100-rows-wider-than-the-pair-scan', '64x64-all-Li-row-not-zero',
                              '30x30-all-Li-row-not-zero', '70x66-all-Li-row-not-zero'])
def test_kernels_fall_back_when_their_table_forms_do_not_apply(kmc, tables, nx, ny, R, hp, hs, zero):
    """hop_lat_kernel and hop_big_kernel's per-site words read rates through a palette of at most 1024 distinct values, the
    latter scans a row's block sums two per lane (rows of at most 1024 sites), and hop_q64_kernel / hop_lat_kernel store a vacant
    site as the all-Li key, whose row is zero in any table that describes hops.  A table of ~10^6 distinct rates (which does not
    factor into the pair tables either: the dense gathers), a row of 1100 sites, and a table whose all-Li row is not zero must
    each take the kernels that do not need the form, by themselves, hop for hop like the oracle; asking for the others by
    number is an error."""
    rs = np.random.RandomState(nx * 7 + ny + int(zero))
    dense_t = O.build_hex_table(tables[0]['base512_normcons']) if ny > 1024 else _random_hop_table(rs, zero)
    conc = np.linspace(0.4, 0.6, R)
    occ = np.stack([(rs.rand(nx, ny) < c).astype(np.int8) for c in conc])
    d_pre = np.stack([rs.rand(R, hp), np.ones((R, hp))], axis=-1)
    d_sim = rs.rand(R, hs, 2)
    with kmc.KMCEngine(R, nx, ny, window_mode='hex_ref') as eng:
        eng.set_hex_table(dense_t)
        eng.set_lattice(occ)
        eng.rates_full()
        i1 = eng.run(hp, presim=True, draws=d_pre, want_trace=True)
        if zero:
            assert eng.hop_kernel_name in ('hop_q64_kernel', 'hop_q64_kernel<any>', 'hop_big_kernel')
        else:
            assert eng.hop_kernel_name in ('hop_lds64_kernel', 'hop_lds_kernel', 'hop_big_kernel')
        eng.init_displacements()
        i2 = eng.run(hs, presim=False, draws=d_sim, want_trace=True)
        got = np.concatenate([i1, i2], axis=1)
        lat, tags, st = eng.get_lattice(), eng.get_reference_tags(), eng.status()
        if eng.uses_lds_path and zero:
            eng.set_hop_kernel(which=5)
            with pytest.raises(kmc.KMCError, match='distinct rates'):
                eng.run(10, presim=True, seed=1)
        elif eng.uses_lds_path:
            for which in (3, 4, 5):
                with pytest.raises(kmc.KMCError, match='all-Li row'):
                    eng.set_hop_kernel(which=which)
    assert not st['status'].any()
    for r in range(R):
        rep, idx = _oracle_run(occ[r], dense_t, 'hex_ref', d_pre[r], d_sim[r], conc[r])
        assert np.array_equal(got[r], idx), 'replica %d' % r
        assert np.array_equal(lat[r], rep.occ) and np.array_equal(tags[r], rep.tags)


@pytest.mark.parametrize('nx,ny,table,hops,generic', [(130, 70, 'normcons', 6000, False), (200, 136, 'master', 3000, False),
                                                      (130, 70, 'normcons', 3000, True), (90, 150, 'master', 2000, True)],
                         ids=['130x70-normcons', '200x136-master', '130x70-normcons-global', '90x150-master-global'])
def test_exact_short_form_equals_long_form(kmc, tables, nx, ny, table, hops, generic):
    """The exact replay's short form (no np.sum pass: kmc_exact_par.h, gamma_hint) against its long form, every hop forced
    through either, over thousands of hops and three concentrations -- far more replays than the oracle can follow at this
    size.  One differing index would send the trajectories apart.  The master fold's rates span 20 decades: binade
    crossings and events all the way."""
    R = 3
    out = []
    for force in (1, 2, 0):                      # 0: the certified selection, which must agree with the replays as well
        with make_engine(kmc, tables, R, nx, ny, 'hex_ref', table) as eng:
            assert not eng.uses_lds_path
            eng.set_hop_kernel(generic=generic)          # hop_big_kernel, or hop_global_kernel (same replay, its own margin)
            eng.set_debug(force_exact=force)
            eng.fill_random([0.25, 0.5, 0.8], 21)
            eng.rates_full()
            eng.init_displacements()
            idx = eng.run(hops, presim=False, seed=0xABCDEF, want_trace=True)
            st = eng.status()
            assert not st['status'].any() and (force == 0 or (st['n_exact'] == hops).all())
            out.append((idx, eng.get_lattice(), eng.get_keys(), eng.get_displacements(), eng.scalars()['time']))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize('which', [0, 2], ids=['site-words', 'tree'])
def test_large_lattice_1024(kmc, tables, dense, which):
    """BASELINE config 4: one 1024 x 1024 lattice.  Hop-for-hop against the C oracle
    over 2 400 hops (the oracle's literal cumsum over 6.3 M entries per hop: ~20 s), then size-independent invariants
    on a longer Philox run.  Both forms of hop_big_kernel."""
    nx = 1024
    H = 2400 if which == 0 else 300
    rs = np.random.RandomState(1024)
    occ = (rs.rand(1, nx, nx) < 0.5).astype(np.int8)
    d_sim = rs.rand(1, H, 2)
    with make_engine(kmc, tables, 1, nx) as eng:
        eng.set_hop_kernel(which=which)
        eng.set_lattice(occ)
        eng.rates_full()
        eng.init_displacements()
        idx = eng.run(H, presim=False, draws=d_sim, want_trace=True)
        rep = CO.Replica(occ[0], dense)
        rep.init_tags()
        _, ref_idx, _ = rep.run(d_sim[0], False)
        assert np.array_equal(idx[0], ref_idx)
        assert np.array_equal(eng.get_lattice()[0], rep.occ)
        assert np.array_equal(eng.get_keys()[0], rep.keys)
        np.testing.assert_allclose(eng.scalars()['time'][0], rep.time.value, rtol=RTOL)
        # longer production-mode run: particle number conserved, every hop moved one
        # atom by one lattice vector, the rate state is reproducible from the keys
        n_li = int(occ.sum())
        eng.run(20000, presim=False, seed=7)
        lat = eng.get_lattice()[0]
        assert int(lat.sum()) == n_li
        tags = eng.get_displacements()[0]
        assert int((tags[..., 0] >= 0).sum()) == n_li and np.array_equal(tags[..., 0] >= 0, lat == 1)
        st = eng.status()
        assert st['status'][0] == 0 and st['hops_done'][0] == 20000 + H
        assert eng.scalars()['hist'][0].sum() == 20000 + H
        keys = eng.get_keys()[0]
        fresh = O.all_keys(lat)
        assert np.array_equal(keys == 0xFFFFFFFF, fresh == 0xFFFFFFFF)      # vacancy pattern always fresh
        assert (keys != fresh).mean() < 0.2                                  # stale keys exist but are a minority
        g = eng.gamma_exact()[0]
        np.testing.assert_allclose(eng.scalars()['gamma'][0], g, rtol=RTOL)


@pytest.mark.gpu
@pytest.mark.parametrize('nx,ny', [(64, 64), (30, 30), (70, 66)])
def test_per_replica_hop_counts(kmc, tables, nx, ny):
    """kmc_run_counts (the concentration sweep: int(n x) hops per replica and KMC step) leaves replica k
    exactly where kmc_run(counts[k]) leaves it: draws are keyed by (seed, replica, hop number), so the
    state after c hops does not depend on how many hops the neighbours run.  Counts straddle the
    64-hop chunks of the kernels (0, 1, 63, 64, 65) -- 64 x 64 is the 16-per-CU kernel, 30 x 30 the
    any-size LDS kernel, 70 x 66 the large-lattice kernel."""
    counts = np.array([0, 1, 63, 64, 65, 200, 129], dtype=np.int64)
    R, seed = len(counts), 77

    def prepared():
        eng = make_engine(kmc, tables, R, nx, ny)
        if (nx, ny) == (64, 64):
            eng.set_hop_kernel(which=3)
        eng.fill_random(np.linspace(0.3, 0.7, R), 5)
        eng.rates_full()
        eng.run(150, presim=True, seed=seed)
        eng.init_displacements()
        return eng

    def state(eng):
        sc, st = eng.scalars(), eng.status()
        return (eng.get_lattice(), eng.get_keys(), eng.get_displacements(), sc['time'], sc['step_time'], sc['gamma'],
                sc['hist'], st['hops_done'], st['status'])

    with prepared() as eng:
        eng.run_counts(counts, presim=False, seed=seed + 1)
        got = state(eng)
    for c in np.unique(counts):
        with prepared() as eng:
            if c:
                eng.run(int(c), presim=False, seed=seed + 1)
            want = state(eng)
        for k in np.nonzero(counts == c)[0]:
            for a, b in zip(got, want):
                if c == 0 and a is got[4]:
                    continue                 # step_time of a replica that did not run keeps the presimulation's
                assert np.array_equal(a[k], b[k]), (nx, ny, int(c), int(k))


@pytest.mark.gpu
def test_full_size_ensemble_two_kernels(kmc, tables, dense):
    """BASELINE's headline size -- 4096 replicas of 64 x 64, the bench configuration, one full round of
    16 workgroups per CU -- through the default 16-per-CU kernel and through the 8-per-CU kernel:
    every lattice, key, displacement tag and histogram identical, clocks equal to rounding, nobody
    stopped.  (The 8-per-CU kernel is pinned to the oracle by the trajectory tests above.)"""
    R, hp, hs, seed = 4096, 2048, 4096, 2024
    conc = np.tile(np.linspace(0.1, 0.9, 9), R // 9 + 1)[:R]
    out = []
    for which in (3, 2):
        with make_engine(kmc, tables, R, 64, 64) as eng:
            eng.set_hop_kernel(which=which)
            eng.fill_random(conc, 31)
            eng.rates_full()
            eng.run(hp, presim=True, seed=seed)
            eng.init_displacements()
            eng.run(hs, presim=False, seed=seed + 1)
            sc, st = eng.scalars(), eng.status()
            out.append((eng.get_lattice(), eng.get_keys(), eng.get_displacements(), sc['hist'], st['hops_done'], st['status'],
                        sc['time'], sc['gamma']))
    a, b = out
    for k in range(6):
        assert np.array_equal(a[k], b[k]), k
    assert not a[5].any() and (a[4] == hp + hs).all()
    np.testing.assert_allclose(a[6], b[6], rtol=1e-13)
    np.testing.assert_allclose(a[7], b[7], rtol=1e-13)
    # conservation: hops move atoms, they never create or destroy them
    assert np.array_equal(a[0].reshape(R, -1).sum(axis=1), (a[2].reshape(R, -1, 4)[:, :, 0] >= 0).sum(axis=1))
    # ... and the oracle at this size: three replicas (first, middle, last) replayed from the same Philox streams
    for r in (0, 2047, 4095):
        rep = CO.Replica(PH.fill_lattice(31, r, 64, 64, conc[r]), dense)
        rep.run(PH.hop_draws(seed, r, 0, hp), True, want_trace=False)
        rep.init_tags()
        rep.run(PH.hop_draws(seed + 1, r, hp, hs), False, want_trace=False)
        assert np.array_equal(a[0][r], rep.occ)
        t = a[2][r].astype(np.int64)
        assert np.array_equal(np.stack([t[..., 0], t[..., 1], 100 * t[..., 2] + t[..., 3]], axis=-1), rep.tags)
        assert np.array_equal(a[3][r], rep.hist.astype(np.int64))
        np.testing.assert_allclose(a[6][r], rep.time.value, rtol=RTOL)


# ---------------------------------------------------------------- launch groups
@pytest.mark.parametrize('n,which', [(64, 3), (64, 2), (30, 4), (30, 1)], ids=['q64', 'lds64', 'q-30x30', 'lds-30x30'])
def test_launch_groups_equal_one_launch(kmc, tables, n, which):
    """kmc_set_launch_groups: a run launched as two kernels over the two halves of the replicas, runs not
    joined in between, is the same computation replica by replica -- every output bit-identical to one launch
    per run; injected draws, traces, per-replica hop counts and the free-running Philox sequence.  Every LDS
    hop kernel."""
    R, seed = 77, 0xABCDEF                                   # odd: unequal halves (40 + 37)
    conc = np.tile(np.linspace(0.1, 0.9, 9), R // 9 + 1)[:R]
    rs = np.random.RandomState(5)
    d_inj = rs.rand(R, 150, 2)
    counts = rs.randint(0, 3000, size=R)
    out = []
    for mode in (1, 2):
        with make_engine(kmc, tables, R, n, n) as eng:
            eng.set_hop_kernel(which=which)
            eng.set_launch_groups(mode)
            eng.fill_random(conc, 17)
            eng.rates_full()
            eng.run(1000, presim=True, seed=seed)
            eng.init_displacements()
            for _ in range(5):                               # back to back: the groups run free
                eng.run(700, presim=False, seed=seed)
            t1 = eng.run(150, presim=False, draws=d_inj, want_trace=True)
            eng.run_counts(counts, presim=False, seed=seed)
            eng.run(333, presim=False, seed=seed)
            t2 = eng.run(64, presim=False, seed=seed, want_trace=True)
            ms, groups = eng.run_span_ms(3)
            assert groups == mode and ms > 0
            sc, st = eng.scalars(), eng.status()
            ds, dj, _ = eng.diffusion(conc)
            out.append((eng.get_lattice(), eng.get_keys(), eng.get_displacements(), sc['hist'], st['hops_done'],
                        st['status'], st['n_exact'], sc['time'], sc['gamma'], t1, t2, ds, dj, eng.barrier_stats()['sum']))
    for k, (a, b) in enumerate(zip(*out)):
        assert np.array_equal(a, b), k
    assert (out[0][4] == 1000 + 3500 + 150 + counts + 333 + 64).all() and not out[0][5].any()


def test_launch_group_controls_reject_bad_arguments(kmc, tables):
    with make_engine(kmc, tables, 16, 64, 64) as eng:
        with pytest.raises(kmc.engine.KMCError):
            eng.set_launch_groups(3)
        with pytest.raises(kmc.engine.KMCError):
            eng.run_span_ms(1)                              # nothing has run yet
        eng.fill_random(0.5, 1)
        eng.rates_full()
        eng.set_launch_groups(2)
        eng.run(10, presim=True, seed=1)
        ms, groups = eng.run_span_ms(1)
        assert groups == 2 and ms > 0 and eng.last_kernel_ms() >= ms * 0.999
        # the second group runs 400 x longer than the first: the time of the last launch is the span of BOTH
        # groups (first start to last end), not the first group's kernel
        eng.run_counts(np.array([10] * 8 + [4000] * 8), presim=True, seed=1)
        span, groups = eng.run_span_ms(1)
        g0 = eng.run_kernel_times(1)[-1]
        assert groups == 2 and span > 5 * g0
        assert eng.last_kernel_ms() >= span * 0.999
        with pytest.raises(kmc.engine.KMCError):
            eng.run_span_ms(3)                              # more runs than there were
        with pytest.raises(kmc.engine.KMCError):
            eng.run_span_ms(65)
        assert eng.status()['hops_done'].tolist() == [20] * 8 + [4010] * 8


def test_any_size_kernel_two_builds_agree(kmc, tables, dense):
    """The any-size LDS kernel exists in a 3- and a 4-waves-per-SIMD build (kmc_hop_lds.h); the engine takes the
    second for ensembles of more than 3072 small lattices.  Same source, so the same bits: 3200 replicas of 12 x 12
    (4-wave build) against the same replicas run 1600 at a time (3-wave build), and a sample against the oracle."""
    R, n, hp, hs, seed = 3200, 12, 300, 500, 77
    conc = np.tile(np.linspace(0.1, 0.9, 9), R // 9 + 1)[:R]
    def run(lo, hi):
        with make_engine(kmc, tables, hi - lo, n, n) as eng:
            eng.set_debug(force_exact=False, margin_scale=5e7)          # some hops take the exact path
            occ = np.stack([PH.fill_lattice(5, r, n, n, conc[r]) for r in range(lo, hi)])
            eng.set_lattice(occ)
            eng.rates_full()
            d = np.stack([PH.hop_draws(seed, r, 0, hp + hs) for r in range(lo, hi)])
            eng.run(hp, presim=True, draws=d[:, :hp])
            eng.init_displacements()
            eng.run(hs, presim=False, draws=d[:, hp:])
            sc = eng.scalars()
            assert eng.status()["n_exact"].sum() > 100
            return eng.get_lattice(), eng.get_keys(), eng.get_displacements(), sc['hist'], sc['time'], sc['gamma'], occ, d
    big = run(0, R)
    halves = [run(0, R // 2), run(R // 2, R)]
    for k in range(6):
        assert np.array_equal(big[k], np.concatenate([h[k] for h in halves])), k
    for r in (0, 1777, R - 1):
        rep = CO.Replica(big[6][r], dense, 'hex_ref')
        rep.run(big[7][r, :hp], True)
        rep.init_tags()
        rep.run(big[7][r, hp:], False)
        assert np.array_equal(big[0][r], rep.occ) and np.array_equal(big[3][r], rep.hist.astype(np.int64))
        np.testing.assert_allclose(big[4][r], rep.time.value, rtol=RTOL)


# ---------------------------------------------------------------- checkpoint / resume
@pytest.mark.parametrize('nx,ny,R,offset', [(64, 64, 5, 0), (30, 30, 4, 0), (70, 66, 2, 0), (64, 64, 3, 4096 * 5 + 17)],
                         ids=['q64', 'lds-30x30', 'big-70x66', 'q64-sharded-rank'])
def test_checkpoint_resume_continues_bit_for_bit(kmc, tables, tmp_path, nx, ny, R, offset):
    """The reference's state is the arrays kmc_step takes and returns (hex_LiCoO2_kmc.py:470, :539); saved with
    np.savez and restored into a NEW engine (plus each replica's position in the Philox stream), the run goes on
    exactly as the uninterrupted one: stale keys, displacement tags, clocks and histograms included.  The last case is
    a shard of a larger ensemble (run_ensemble always sets an offset): the checkpoint carries the replica offset the
    Philox streams are keyed by, and restore() into an engine created at offset 0 must put it back."""
    conc = np.linspace(0.3, 0.7, R)
    seed = 4242
    with make_engine(kmc, tables, R, nx, ny) as a:
        a.set_replica_offset(offset)
        a.fill_random(conc, 3)
        a.rates_full()
        a.run(300, presim=True, seed=seed)
        a.init_displacements()
        a.run(400, presim=False, seed=seed)
        path = str(tmp_path / 'state.npz')
        np.savez(path, **a.checkpoint())
        a.run(500, presim=False, seed=seed)
        a.run(77, presim=False, seed=seed)
        sc, st = a.scalars(), a.status()
        want = (a.get_lattice(), a.get_keys(), a.get_displacements(), sc['time'], sc['hist'], st['hops_done'],
                a.diffusion(conc)[0], a.gamma_exact())
    with make_engine(kmc, tables, R, nx, ny) as b:
        with np.load(path) as ck:
            b.restore(ck)
        assert b.replica_offset == offset
        b.run(500, presim=False, seed=seed)
        b.run(77, presim=False, seed=seed)
        sc, st = b.scalars(), b.status()
        got = (b.get_lattice(), b.get_keys(), b.get_displacements(), sc['time'], sc['hist'], st['hops_done'],
               b.diffusion(conc)[0], b.gamma_exact())
    for k, (x, y) in enumerate(zip(got, want)):
        assert np.array_equal(x, y), k
    assert (got[5] == 300 + 400 + 577).all()
    with make_engine(kmc, tables, R + 1, nx, ny) as c:
        with np.load(path) as ck, pytest.raises(ValueError):
            c.restore(ck)


@pytest.mark.parametrize('nx,ny,R,hops', [(64, 64, 24, 700), (30, 30, 24, 500), (70, 66, 6, 300), (4, 6, 6, 40)])
def test_replica_offset_makes_sharding_invisible(kmc, tables, nx, ny, R, hops):
    """kmc_set_replica_offset: two contexts of R/2 replicas with offsets 0 and R/2 reproduce one context of R bit
    for bit -- random fill, Philox draws, every kernel family (16-per-CU 64 x 64, any-size LDS, large-lattice,
    global-memory) -- so an ensemble does not depend on the number of GPUs it is sharded over."""
    conc = np.linspace(0.3, 0.7, R)

    def go(r0, r1, offset):
        with make_engine(kmc, tables, r1 - r0, nx, ny) as eng:
            eng.set_replica_offset(offset)
            eng.fill_random(conc[r0:r1], 77)
            occ0 = eng.get_lattice()
            eng.rates_full()
            eng.run(hops, presim=True, seed=78)
            eng.init_displacements()
            eng.run(hops, presim=False, seed=78)
            sc = eng.scalars()
            return occ0, eng.get_lattice(), eng.get_keys(), eng.get_displacements(), sc['time'], sc['hist'], sc['gamma']
    whole = go(0, R, 0)
    a, b = go(0, R // 2, 0), go(R // 2, R, R // 2)
    for k, (w, x, y) in enumerate(zip(whole, a, b)):
        assert np.array_equal(w, np.concatenate([x, y])), 'output %d differs' % k
    # ... and the offset is what does it
    c = go(R // 2, R, 0)
    assert not np.array_equal(whole[0][R // 2:], c[0])
    with make_engine(kmc, tables, 4, 8) as eng:
        with pytest.raises(kmc.engine.KMCError):
            eng.set_replica_offset(2 ** 32 - 2)
        with pytest.raises(kmc.engine.KMCError):
            eng.set_replica_offset(2 ** 64 - 1)         # offset + R wraps 64 bits
