"""The oracle (oracle/kmc_oracle.py) against the vectors produced by running
the reference's own function text (oracle/gen_golden_from_reference.py)."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import kmc_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, 'traj_*.npz')))


@pytest.fixture(scope='module')
def tables():
    t = np.load(os.path.join(GOLDEN, 'tables.npz'))
    with open(os.path.join(GOLDEN, 'tables.json')) as f:
        info = json.load(f)
    return t, info


@pytest.fixture(scope='module')
def dense(tables):
    t, _ = tables
    return O.build_hex_table(t['base512_normcons'])


def test_dumbbell(tables):
    _, info = tables
    assert O.dumbbell_rate() == info['dumbbell'] == 0.0023195228302435697


def test_dense_table_bit_exact(tables, dense):
    t, info = tables
    assert hashlib.sha256(dense.astype('<f8').tobytes()).hexdigest() == info['dense_sha256']
    assert np.array_equal(dense[t['sample_rows']], t['sample_rates'])
    assert int(np.count_nonzero(dense)) == info['dense_nonzero']
    assert int(np.sum(~dense.any(axis=1))) == info['dense_allzero_rows']
    # six-fold symmetry self-check of the reference (hex_dict_builder.py:144-145)
    np.testing.assert_allclose(dense.sum(axis=0), info['dense_colsum'], rtol=1e-12)
    assert np.allclose(np.log(dense.sum(axis=0)), 26.66879082, atol=1e-7)


def test_master_table_restriction(tables):
    t, info = tables
    master = {int(k): float(v) for k, v in zip(t['master_keys'], t['master_values'])}
    b = O.base512_from_master(master)
    assert hashlib.sha256(b.astype('<f8').tobytes()).hexdigest() == info['base512_master_sha256']
    assert np.array_equal(b, t['base512_master'])


def test_rate_generation(tables):
    """Row G: 1e13*exp(-E/kT) from the 47-term cluster expansion reproduces the
    reference's own regeneration bit for bit and its shipped pickle to 1e-12."""
    t, _ = tables
    mine = O.generate_base512()
    assert np.array_equal(mine, t['base512_regenerated'])
    np.testing.assert_allclose(mine, t['base512_normcons'], rtol=1e-12, atol=0)
    e = np.array([O.excited_energy(O.base_key16(i)) for i in range(512)])
    assert np.array_equal(e, t['base512_energies'])


def test_excited_energy_guards():
    with pytest.raises(ValueError, match='endpoint not empty'):
        O.excited_energy('1111111111011111')
    with pytest.raises(ValueError, match='no ion to move'):
        O.excited_energy('1111101110011111')
    with pytest.raises(ValueError, match='nearest neighbour not empty'):
        O.excited_energy('1111111110111111')


@pytest.fixture(scope='module')
def dense_master(tables):
    """The fold of master_dictionary.p's restriction to the builder's 512 keys (BASELINE config 1's table)."""
    t, info = tables
    d = O.build_hex_table(t['base512_master'])
    assert hashlib.sha256(d.astype('<f8').tobytes()).hexdigest() == info['dense_master_sha256']
    return d


@pytest.mark.parametrize('path', TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_trajectory(path, dense, dense_master):
    g = np.load(path)
    if str(g['table']) == 'master':
        dense = dense_master
    nx, ny, conc = int(g['nx']), int(g['ny']), float(g['conc'])
    mode = 'obj_ref' if int(g['oo_window']) else 'hex_ref'
    occ = g['occ0'].copy()
    assert np.array_equal(O.rates_full(occ, dense), g['P0'])
    trace = []
    res = O.run_replica(occ, dense, conc, int(g['n_presim']), int(g['n_sim']),
                        iter(g['draws']), window_mode=mode, trace=trace)
    assert np.array_equal([t[0] for t in trace], g['idx'])
    assert np.array_equal([t[1] for t in trace], g['gammas'])
    assert np.array_equal(res['occ'], g['occ_final'])
    assert np.array_equal(res['tags'], g['tags_final'])
    assert np.array_equal(res['P'], g['P_final'])
    assert np.array_equal(res['hist'], g['hist'])
    assert res['total_time'] == float(g['total_time'])
    assert res['d_star'] == float(g['d_star'])
    assert res['d_j'] == float(g['d_j'])
    if g['barriers'].size:
        assert res['barrier'] == g['barriers'][-1]
    stale = np.any(res['P'] != O.rates_full(res['occ'], dense), axis=-1)
    assert np.array_equal(stale, g['stale_mask'])
    if 'class_clock' in g.files:
        # the class's text (Mc_Lattice.py:175-176) adds every hop's dt into one running sum; the functional
        # statements sum per step: the same numbers in another order
        np.testing.assert_allclose(res['total_time'], float(g['class_clock']), rtol=1e-13)
    mat, lst, dj = O.get_distance(res['tags'], [nx, ny, 1])
    assert np.array_equal(mat, g['dist_matrix'])
    assert np.array_equal(lst, g['dist_list'])
    assert dj == float(g['dj_len'])


def test_seeded_fill_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'traj_n10_x05.npz'))
    rs = np.random.RandomState(int(g['seed']))
    occ = O.fill_lattice_reference(10, 10, 0.5, rs.randint)
    assert np.array_equal(occ, g['occ0'])
    rows = ''.join(''.join(str(v) for v in r) + ' ' for r in occ).split()
    assert rows[0] == '0000011001' and rows[9] == '0001111110'


def test_numpy_sum_model():
    """np.sum association model against np.sum itself."""
    rs = np.random.RandomState(3)
    for n in (5, 100, 129, 5400, 8192, 8200, 24576, 30000):
        for _ in range(3):
            a = rs.lognormal(0, 8, n) * (rs.rand(n) < 0.3)
            assert O.numpy_sum_model(a) == float(np.sum(a)), n
            assert O.numpy_sum_model(a) == float(np.sum(a.reshape(-1, 1, 1))), n
