"""Result files of the reference's driver: the reader parses the reference's own output
files (tests/golden/ref_*: data the reference holds in its root, copied as fixtures), the writer produces
files the same reader -- and therefore any consumer of the reference's files -- parses to the same numbers."""
import os

import numpy as np

from kmc_amd import report

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_reads_reference_kmc_out():
    r = report.read_kmc_out(os.path.join(GOLDEN, 'ref_LiCoO2_kmc.out'))
    assert r['size'] == [0, 0, 0] and r['kb_t'] == 0.025 and r['presim'] == 100 and r['sim'] == 100
    assert r['total_time'] == 9.4329431057
    assert [p[0] for p in r['points']] == [0.1, 0.12, 0.14]
    np.testing.assert_array_equal(r['points'][0][1], [4.66504294e-20, 6.23843623e-20])
    np.testing.assert_array_equal(r['points'][2][2], [3.47251008e-19, 3.05338328e-20])
    assert r['table'].shape == (3, 5)
    np.testing.assert_array_equal(r['table'][1], [5.70521784e-20, 0.12, 2.81050318e-20, 8.14085611e-20, 2.41776400e-20])
    # the reference's own table is mean and np.std (ddof 0) of the per-iteration lines (:783-786)
    for (x, ds, dj), row in zip(r['points'], r['table']):
        np.testing.assert_allclose([np.average(ds), x, np.std(ds), np.average(dj), np.std(dj)], row, rtol=2e-8)


def test_reads_reference_header_only_file():
    r = report.read_kmc_out(os.path.join(GOLDEN, 'ref_0.5LiCoO2_kmc.out'))
    assert r['points'] == [] and r['table'].shape == (0, 5)


def test_kmc_out_round_trip_and_layout(tmp_path):
    ref_path = os.path.join(GOLDEN, 'ref_LiCoO2_kmc.out')
    ref = report.read_kmc_out(ref_path)
    p = tmp_path / 'LiCoO2_kmc.out'
    with report.KmcOut(str(p), ref['size'], ref['kb_t'], ref['presim'], ref['sim']) as out:
        for x, ds, dj in ref['points']:
            out.point(x, ds, dj)
        out.close(ref['total_time'])
    got = report.read_kmc_out(str(p))
    assert got['size'] == ref['size'] and got['presim'] == 100 and got['total_time'] == ref['total_time']
    for a, b in zip(got['points'], ref['points']):
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_allclose(got['table'], ref['table'], rtol=2e-8)
    # same text, line for line, down to the per-iteration lines (numbers there are printed like numpy 1.x did)
    mine, theirs = open(str(p)).read().splitlines(), open(ref_path).read().splitlines()
    assert mine[:14] == theirs[:14]
    assert mine[14:22] == theirs[14:22]                      # Total Time ... END ... table header
    assert len(mine) == len(theirs)


def test_data_run_csv_round_trip(tmp_path):
    ref_path = os.path.join(GOLDEN, 'ref_data_run_4_data')
    t = report.read_data_run(ref_path)
    assert t.shape == (44, 5) and t[0, 1] == 0.3 and t[-1, 0] == 1.25306662e-15
    p = tmp_path / 'data_run'
    report.write_data_run(str(p), t)
    theirs = [ln for ln in open(ref_path).read().splitlines() if ln.strip()]
    assert open(str(p)).read().splitlines() == theirs        # byte-identical rows
    np.testing.assert_array_equal(report.read_data_run(str(p)), t)


def test_data_out_and_sweep_writer(tmp_path):
    results = [dict(concentration=0.1, d_star=1.5e-10, d_j=2.5e-10, error_star=1e-11, error_j=2e-11, time=3.25e-7,
                    replicas=4, hops=100, hop_histogram=[1, 2, 3, 4, 5, 6]),
               dict(concentration=0.5, d_star=8e-10, d_j=2.7e-9, error_star=1e-10, error_j=3e-10, time=1.5e-6,
                    replicas=4, hops=100, hop_histogram=[6, 5, 4, 3, 2, 1])]
    per = {0.1: (np.array([1e-10, 2e-10]), np.array([2e-10, 3e-10])), 0.5: (np.array([8e-10]), np.array([2.7e-9]))}
    k, d, c = (str(tmp_path / n) for n in ('LiCoO2_kmc.out', 'LiCoO2_data.out', 'data_run_0_data'))
    table = report.write_sweep(results, 64, k, d, 0.025, 250, 750, per_replica=per, wall_seconds=1.5, data_run_path=c)
    assert open(d).readline() == report.DATA_HEADER
    cols = report.read_data_out(d)
    assert list(cols) == ['Concentration', 'Dimension', 'D_star', 'D_j', 'error_star', 'error_j', 'time']
    np.testing.assert_array_equal(cols['Dimension'], [64, 64])
    np.testing.assert_array_equal(cols['D_j'], [2.5e-10, 2.7e-9])      # repr precision: exact
    np.testing.assert_array_equal(cols['time'], [3.25e-7, 1.5e-6])
    got = report.read_kmc_out(k)
    assert got['size'] == [64, 64, 1] and got['presim'] == 250 and got['sim'] == 750 and got['total_time'] == 1.5
    np.testing.assert_array_equal(got['points'][0][1], per[0.1][0])
    np.testing.assert_array_equal(got['table'], table)                  # 8 digits suffice for these values
    np.testing.assert_array_equal(report.read_data_run(c), table)


def test_offset_score_and_barrier_summary():
    assert report.offset_score([4, 9, 4, 4, 3, 4]) == 1.5               # ((9+3)/2) / ((4+4+4+4)/4), :760-761
    np.testing.assert_array_equal(report.offset_score(np.array([[1, 1, 1, 1, 1, 1], [2, 1, 2, 2, 1, 2]])), [1.0, 0.5])
    st = dict(sum=np.array([1.0, 0.0, 3.0]), count=np.array([2.0, 0.0, 2.0]), min=np.array([0.3, 9e9, 0.4]),
              max=np.array([0.7, -9e9, 0.9]), last=np.zeros(3))
    assert report.barrier_summary(st) == (1.0, 0.9, 0.3)
