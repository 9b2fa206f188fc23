"""C twin of the oracle against the Python oracle and the golden vectors."""
import glob
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import kmc_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, 'traj_*.npz')))


@pytest.fixture(scope='module')
def dense():
    t = np.load(os.path.join(GOLDEN, 'tables.npz'))
    return O.build_hex_table(t['base512_normcons'])


def test_table(dense):
    t = np.load(os.path.join(GOLDEN, 'tables.npz'))
    assert np.array_equal(CO.build_table(t['base512_normcons'], O.dumbbell_rate()), dense)


def test_numpy_sum():
    rs = np.random.RandomState(1)
    for n in (1, 7, 8, 127, 128, 129, 900, 5400, 8191, 8192, 8193, 24576, 100000, 6 * 1024 * 1024):
        a = rs.lognormal(0, 8, n) * (rs.rand(n) < 0.3)
        assert CO.numpy_sum(a) == float(np.sum(a)), n
        assert CO.numpy_sum(a) == float(np.sum(a.reshape(-1, 1, 1, 1))), n


@pytest.fixture(scope='module')
def dense_master():
    t = np.load(os.path.join(GOLDEN, 'tables.npz'))
    return O.build_hex_table(t['base512_master'])


@pytest.mark.parametrize('path', TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_trajectory(path, dense, dense_master):
    g = np.load(path)
    if str(g['table']) == 'master':
        dense = dense_master
    conc = float(g['conc'])
    mode = 'obj_ref' if int(g['oo_window']) else 'hex_ref'
    rep = CO.Replica(g['occ0'], dense, mode)
    assert np.array_equal(rep.P, g['P0'])
    n_hops = int(int(g['nx']) * int(g['ny']) * conc)
    npre, nsim = int(g['n_presim']) * n_hops, int(g['n_sim']) * n_hops
    draws = g['draws']
    d_pre = np.stack([draws[:npre], np.ones(npre)], axis=1)
    d_sim = draws[npre:].reshape(nsim, 2)
    done, idx1, gam1 = rep.run(d_pre, True)
    assert done == npre
    rep.init_tags()
    done, idx2, gam2 = rep.run(d_sim, False)
    assert done == nsim
    assert np.array_equal(np.concatenate([idx1, idx2]), g['idx'])
    assert np.array_equal(np.concatenate([gam1, gam2]), g['gammas'])
    assert np.array_equal(rep.occ, g['occ_final'])
    assert np.array_equal(rep.tags, g['tags_final'])
    assert np.array_equal(rep.P, g['P_final'])
    assert np.array_equal(rep.hist, g['hist'])
    np.testing.assert_allclose(rep.time.value, float(g['total_time']), rtol=1e-13)
    out = rep.diffusion(conc)
    np.testing.assert_allclose(out[0], float(g['d_star']), rtol=1e-12)
    np.testing.assert_allclose(out[1], float(g['d_j']), rtol=1e-12)
    if g['barriers'].size:
        np.testing.assert_allclose(rep.barrier.value, g['barriers'][-1], rtol=1e-12)
    # stale keys reproduce the stale rate matrix
    li = rep.keys != 0xFFFFFFFF
    assert np.array_equal(dense[rep.keys[li]], rep.P[li])
    assert not rep.P[~li].any()


def test_python_vs_c_random(dense):
    """Hop-for-hop agreement on a fresh random case, all three windows."""
    rs = np.random.RandomState(77)
    for mode in ('hex_ref', 'obj_ref', 'full'):
        occ = (rs.rand(12, 9) < 0.45).astype(np.uint8)
        draws = rs.rand(300, 2)
        rep = CO.Replica(occ, dense, mode)
        rep.init_tags()
        _, idx, gam = rep.run(draws, False)
        occ_p = occ.copy()
        P = O.rates_full(occ_p, dense)
        tags = O.init_tags(occ_p)
        trace = []
        O.kmc_step(occ_p, tags, P, dense, 300, iter(draws.ravel()), False, mode, trace=trace)
        assert np.array_equal([t[0] for t in trace], idx)
        assert np.array_equal([t[1] for t in trace], gam)
        assert np.array_equal(occ_p, rep.occ) and np.array_equal(tags, rep.tags)
        assert np.array_equal(P, rep.P)
        if mode == 'full':
            assert np.array_equal(P, O.rates_full(occ_p, dense))
