"""The reference's driver loops on the GPU: the batched self-check drivers
(kmc_amd.diagnostics, Mc_Lattice.py:243-302) and the sweep that writes the result files (kmc_amd.sweep,
hex_LiCoO2_kmc.py:650-818), each replayed replica by replica with the C oracle on the same Philox
lattices and draws.  Bar: integer observables exact, coefficients and clocks within 1e-12 relative."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle as CO          # noqa: E402
from oracle import kmc_oracle as O         # noqa: E402
from oracle import philox as PH            # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
RTOL = 1e-12
A2 = (2.8334 * 10 ** (-8)) ** 2


@pytest.fixture(scope='module')
def base512():
    return np.load(os.path.join(GOLDEN, 'tables.npz'))['base512_normcons']


@pytest.fixture(scope='module')
def dense(base512):
    return O.build_hex_table(base512)


def _replay(dense, fill_seed, run_seed, r, n, conc, pre_hops, sim_hops, mode):
    """One replica as the engine defines it: Philox lattice, Philox draws, presim then sim."""
    occ = PH.fill_lattice(fill_seed, r, n, n, conc)
    d = PH.hop_draws(run_seed, r, 0, pre_hops + sim_hops)
    rep = CO.Replica(occ, dense, mode)
    if pre_hops:
        done, _, _ = rep.run(d[:pre_hops], True)
        assert done == pre_hops
    rep.init_tags()
    if sim_hops:
        done, _, _ = rep.run(d[pre_hops:], False)
        assert done == sim_hops
    return rep


def _class_coeffs(rep, n_li, n):
    out = rep.diffusion(n_li / float(n * n))
    return A2 * out[2] / (4 * rep.time.value * n_li), A2 * out[3] / (4 * rep.time.value * n_li)


def test_dimension_check(base512, dense):
    from kmc_amd import diagnostics as D
    got = D.dimension_check(('base', base512), dimensions=[10, 23], averaging=3, presim_steps=3, sim_steps=5, seed=11)
    assert sorted(got) == [10, 23]
    for n in got:
        n_li = int(0.2 * n * n)
        for r in range(3):
            rep = _replay(dense, 11 + n, 11 + n, r, n, 0.2, 3 * n_li, 5 * n_li, 'obj_ref')
            np.testing.assert_allclose(got[n][r], _class_coeffs(rep, n_li, n)[1], rtol=RTOL)


def test_HH_check(base512, dense):
    from kmc_amd import diagnostics as D
    got = D.HH_check(('base', base512), dimensions=[10, 20], averaging=4, steps=6, seed=3)
    for n in got:
        n_li = int(0.2 * n * n)
        for r in range(4):
            occ = PH.fill_lattice(3 + n, r, n, n, 0.2)
            rep = CO.Replica(occ, dense, 'obj_ref')
            rep.run(PH.hop_draws(3 + n, r, 0, 6 * n_li), True)
            assert got[n][r] == rep.hist[1] / rep.hist[0]
    # the check itself: the table is isotropic, so over many hops the ratio tends to 1 (Mc_Lattice.py:261)
    # (loosely: hops come in back-and-forth pairs, so single lattices scatter by ~0.15)
    long_run = D.HH_check(('base', base512), dimensions=[30], averaging=128, steps=300, seed=5)[30]
    assert abs(long_run.mean() - 1) < 0.08


def test_counts_check(base512, dense):
    from kmc_amd import diagnostics as D
    n, x, steps = 12, 0.2, 6
    n_li = int(x * n * n)
    curve = D.counts_check(('base', base512), dimension=n, concentration=x, steps=steps, stride=2, replicas=2, seed=8)
    assert curve.shape == (3, 2)
    for r in range(2):
        for k in range(3):
            rep = _replay(dense, 8, 8, r, n, x, 0, 2 * (k + 1) * n_li, 'obj_ref')
            np.testing.assert_allclose(curve[k, r], _class_coeffs(rep, n_li, n)[1], rtol=RTOL)


def test_concentration_run(base512, dense):
    from kmc_amd import diagnostics as D
    n, xs = 14, [0.90, 0.94]
    got = D.concentration_run(('base', base512), n, concentrations=xs, averaging=3, presim_steps=2, sim_steps=3, seed=21)
    for k, x in enumerate(xs):
        n_li = int(x * n * n)
        ds, dj = [], []
        for a in range(3):
            rep = _replay(dense, 21, 21, 3 * k + a, n, x, 2 * n_li, 3 * n_li, 'obj_ref')
            s, j = _class_coeffs(rep, n_li, n)
            ds.append(s)
            dj.append(j)
        assert got[k][0] == x
        np.testing.assert_allclose(got[k][1:], [np.mean(dj), np.std(dj), np.mean(ds), np.std(ds)], rtol=1e-9)


def test_sweep_writes_reference_files(base512, dense, tmp_path, capsys):
    from kmc_amd import ensemble, report, sweep
    xs, avg, pre, sim, seed = [0.2, 0.5, 0.8], 3, 2, 3, 40
    res = sweep.run_sweep(('base', base512, None), [10, 16], xs, avg, pre, sim, seed=seed, out_dir=str(tmp_path))
    printed = capsys.readouterr().out
    assert 'concentration is: 0.2' in printed and 'average barrier = ' in printed and 'max, min barrir = ' in printed
    dat = report.read_data_out(str(tmp_path / 'LiCoO2_data.out'))
    out = report.read_kmc_out(str(tmp_path / 'LiCoO2_kmc.out'))
    assert out['size'] == [10, 10, 1] and out['presim'] == pre and out['sim'] == sim
    row = 0
    for n in (10, 16):
        for k, x in enumerate(xs):
            n_li = int(n * n * 1 * x)
            ds, dj, tt, hist = [], [], [], np.zeros(6)
            for a in range(avg):
                r = a * len(xs) + k                      # global replica id: iteration-major (ensemble.shard_replicas)
                sd = ensemble.derive_seed(seed, n)
                rep = _replay(dense, sd, sd + 1, r, n, x, pre * n_li, sim * n_li, 'hex_ref')
                o = rep.diffusion(x)
                ds.append(o[0])
                dj.append(o[1])
                tt.append(rep.time.value)
                hist += rep.hist
            got = res[n][k]
            assert got['concentration'] == x and got['replicas'] == avg
            assert got['hop_histogram'] == hist.astype(np.int64).tolist()
            np.testing.assert_allclose([got['d_star'], got['d_j'], got['time']], [np.mean(ds), np.mean(dj), np.mean(tt)],
                                       rtol=RTOL)
            np.testing.assert_allclose([got['error_star'], got['error_j']], [np.std(ds), np.std(dj)], rtol=1e-6)
            # the files carry the same numbers
            assert dat['Concentration'][row] == x and dat['Dimension'][row] == n
            assert dat['D_star'][row] == got['d_star'] and dat['D_j'][row] == got['d_j'] and dat['time'][row] == got['time']
            np.testing.assert_allclose(out['table'][row], [got['d_star'], x, got['error_star'], got['d_j'], got['error_j']],
                                       rtol=2e-8)
            row += 1
    # per-iteration lines: the reader keys points by concentration, so the LAST dimension's lines survive
    for k, x in enumerate(xs):
        np.testing.assert_allclose(np.mean(out['points'][k][1]), res[16][k]['d_star'], rtol=2e-8)


def test_temperature_sweep(tmp_path):
    """Next row 2: rates regenerated per kT on the device, then the same sweep."""
    from kmc_amd import ensemble, report, sweep
    kts, n, x, avg, pre, sim = [0.02, 0.035], 12, 0.4, 2, 2, 3
    res = sweep.temperature_sweep(kts, [n], [x], avg, pre, sim, seed=7, out_dir=str(tmp_path), echo=False)
    n_li = int(n * n * 1 * x)
    d_by_t = []
    for kt in kts:
        base = O.generate_base512(10 ** 13, kt)
        dense_t = O.build_hex_table(base, O.dumbbell_rate(10 ** 13, kt))
        ds = []
        for r in range(avg):
            sd = ensemble.derive_seed(7, n)
            rep = _replay(dense_t, sd, sd + 1, r, n, x, pre * n_li, sim * n_li, 'hex_ref')
            ds.append(rep.diffusion(x)[0])
        got = res[kt][n][0]
        # the device's exp may differ from libm's in the last place: selections are the same unless a draw falls
        # within that of a boundary (never in a run this short), coefficients agree to the table's 1e-12
        np.testing.assert_allclose(got['d_star'], np.mean(ds), rtol=1e-10)
        out = report.read_kmc_out(str(tmp_path / ('kbT_' + str(kt)) / 'LiCoO2_kmc.out'))
        assert out['kb_t'] == kt and out['size'] == [n, n, 1]
        d_by_t.append(got['d_star'])
    assert d_by_t[1] > d_by_t[0]                           # hotter lattice diffuses faster


def test_ensemble_is_the_same_at_any_rank_count(base512):
    """One seed, Philox keyed by the global replica id: the ranks of a 2- or 3-rank job (run here one after the
    other on the one GPU, their accumulators added as the all-reduce would) hold exactly the replicas of the
    1-rank job -- per-replica D*, D_J, clocks and histograms bit for bit, 7 replicas per concentration (does not
    divide by 2 or 3)."""
    import kmc_amd
    from kmc_amd import ensemble
    xs, rpp, n = [0.3, 0.5, 0.8], 7, 64
    factory = lambda R, nx, ny, wm, dev: kmc_amd.KMCEngine(R, nx, ny, wm, dev)    # noqa: E731
    table = ('base', base512, None)
    fin1, acc1, loc1 = ensemble.run_ensemble(factory, n, n, xs, rpp, 1, 2, seed=99, table=table, return_replicas=True)
    for world in (2, 3):
        acc, locs = 0, []
        for rank in range(world):
            _, a, loc = ensemble.run_ensemble(factory, n, n, xs, rpp, 1, 2, seed=99, table=table, rank=rank, world=world,
                                              return_replicas=True)
            acc = acc + a
            locs.append(loc)
        for key in ('conc', 'd_star', 'd_j', 'time', 'hist'):
            assert np.array_equal(np.concatenate([l[key] for l in locs]), loc1[key]), key
        assert np.array_equal(acc[:, [0, 6, 7, 8, 9, 10, 11, 12]], acc1[:, [0, 6, 7, 8, 9, 10, 11, 12]])   # counts: exact
        np.testing.assert_allclose(acc, acc1, rtol=1e-13)                                                # sums: order of addition


def test_dstar_against_the_published_files_master_table():
    """north_star: "D*(x) within reference error bars".  The reference's published D*(x) files (data_run_4_data,
    data_run_3_data; copies under tests/golden/) against this engine on the fold of master_dictionary.p, in the shape the
    reference's own LiCoO2_kmc.out header shows (10 x 10, 100 + 100 cycles).  What holds and is asserted: D* rises by
    decades with x on the master table, as in the files.  What does NOT hold (profiles/r2_dstar_vs_reference.md): every
    point lies above the files' mean + error, by 0.4 to 7.7 decades -- the files were not produced by the hex path at this
    revision (which this engine replays bit for bit, test_golden_trajectory) on either committed table.  Parity with the
    published D*(x) is UNPINNED; the offsets are printed as a record, not asserted: a test that passes
    only while parity is absent would fail the day it was reached."""
    import kmc_amd
    from kmc_amd import ensemble
    g = os.path.join(os.path.dirname(__file__), 'golden')
    t = np.load(os.path.join(g, 'tables.npz'))
    run4 = np.loadtxt(os.path.join(g, 'ref_data_run_4_data'), delimiter=',', skiprows=1)       # d_star, x, err, d_j, err_j
    ref = {round(float(r[1]), 2): (float(r[0]), float(r[2])) for r in run4}
    xs = [0.3, 0.5, 0.7, 0.9]
    factory = lambda R, nx, ny, wm, dev: kmc_amd.KMCEngine(R, nx, ny, wm, dev)    # noqa: E731
    fin, _ = ensemble.run_ensemble(factory, 10, 10, xs, 128, 100, 100, seed=ensemble.derive_seed(2025, 10),
                                   table=('base', t['base512_master'], None))
    d = {round(r['concentration'], 2): r['d_star'] for r in fin}
    assert d[0.9] > 1e2 * d[0.3] and d[0.9] > d[0.7] > d[0.5]          # the files: 6.4e-12 / 4.9e-19
    for x in xs:
        mean, err = ref[x]
        print('x = %.1f: D* = %.3e, published %.3e +- %.3e, log10 ratio %+.2f%s'
              % (x, d[x], mean, err, np.log10(d[x] / mean), '' if abs(d[x] - mean) > err else '  (inside the bar)'))


def test_rccl_allreduce_behind_the_c_abi(base512):
    """kmc_allreduce_stats: the one collective as an RCCL call on the engine's stream.  A 1-GPU box can only form a
    one-rank communicator (RCCL refuses two ranks on one device): unique id, communicator, the all-reduce (an
    identity here), twice with growing buffers, teardown, and the argument checks.  The N-rank arithmetic is
    covered by the gloo tests of tests/test_distributed_cpu.py; N ranks on N GPUs run it in bench.py --collective rccl."""
    import kmc_amd
    from kmc_amd import ensemble
    uid = kmc_amd.KMCEngine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with kmc_amd.KMCEngine(4, 8) as eng:
        comm = eng.comm_create(uid, 1, 0)
        rs = np.random.RandomState(3)
        for n in (13, 9 * 13, 5000):
            acc = rs.rand(n)
            assert np.array_equal(eng.allreduce_stats(comm, acc), acc)
        acc2 = rs.rand(9, 13)
        assert np.array_equal(ensemble.all_reduce_acc(acc2, rccl=(eng, comm)), acc2)
        with pytest.raises(kmc_amd.engine.KMCError):
            eng.allreduce_stats(None, acc)
        with pytest.raises(kmc_amd.engine.KMCError):
            eng.comm_create(uid, 2, 2)
        eng.comm_destroy(comm)


def test_bench_two_ranks_on_one_gpu_equal_one_rank():
    """The N > 1 host path of bench.py on the GPU box: `bench.py --gpus 2` with KMC_BENCH_BACKEND=gloo (two rank processes
    sharing the one GPU; the NCCL leg needs two devices).  The line must report n_gpus = 2 and carry the other scaling's
    numbers; and because the Philox streams are keyed by the global replica id, the 2-rank STRONG run (2048 replicas per
    rank) is the same ensemble as a 1-rank run of 4096: D* equal to 1e-12 (the all-reduce adds in another order)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ['--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-long-launch', '--replicas', '512']
    env = dict(os.environ, KMC_BENCH_BACKEND='gloo', KMC_BENCH_RENDEZVOUS_TIMEOUT='120')
    two = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--scaling', 'strong'] + common,
                         env=env, capture_output=True, text=True, timeout=420)
    assert two.returncode == 0, two.stderr[-2000:]
    line2 = json.loads(two.stdout.strip().splitlines()[-1])
    one = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--scaling', 'strong'] + common,
                         env=env, capture_output=True, text=True, timeout=420)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert line2['n_gpus'] == 2 and line2['scaling'] == 'strong' and line2['config']['replicas_total'] == 512
    assert line2['config']['replicas_rank0'] == 256
    assert line2['other_scaling']['scaling'] == 'weak' and line2['other_scaling']['replicas_total'] == 1024
    assert line2['value'] > 0 and line2['other_scaling']['value'] > 0
    assert abs(line2['d_star_mean'] - line1['d_star_mean']) <= 1e-12 * abs(line1['d_star_mean'])
    assert line2['stopped_replicas'] == 0 and line1['stopped_replicas'] == 0
