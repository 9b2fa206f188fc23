"""CPU tests of the host side: table loading, the C-ABI surface, ensemble maths."""
import io
import os
import pickle
import re

import numpy as np
import pytest

from kmc_amd import _lib, ensemble, tables
from oracle import kmc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def test_library_exports_every_declared_symbol():
    """libkmc_hip.so loads without a GPU and exports exactly what include/kmc_hip.h declares."""
    with open(os.path.join(ROOT, 'include', 'kmc_hip.h')) as f:
        header = f.read()
    declared = set(re.findall(r'^(?:int|const char\*)\s+(kmc_\w+)\s*\(', header, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)


def test_no_cpu_fallback():
    """Without a GPU the engine refuses to start; it never computes on the host."""
    import kmc_amd
    import ctypes
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    rc = lib.kmc_create(ctypes.byref(ctx), 0, 1, 10, 10, 0)
    if rc == 0:                      # a GPU is present: nothing to check here
        lib.kmc_destroy(ctx)
        pytest.skip('GPU present')
    assert rc == -2 and b'no CPU backend' in lib.kmc_last_error(None)
    with pytest.raises(kmc_amd.KMCError):
        kmc_amd.KMCEngine(1, 10)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'kmc_amd')):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r'^\s*(from|import)\s+oracle|#include.*oracle|libkmc_oracle|oracle[./]kmc_oracle\.py|c_oracle',
                                     src, flags=re.M), fn


def test_table_pickles_both_flavours():
    t = np.load(os.path.join(GOLDEN, 'tables.npz'))
    normcons = {tables.base_key16(i): float(t['base512_normcons'][i]) for i in range(512)}
    master = {int(k): float(v) for k, v in zip(t['master_keys'], t['master_values'])}
    for d, want in ((normcons, t['base512_normcons']), (master, t['base512_master'])):
        for proto in (2, 4):
            got = tables.base512_from_dict(tables.load_table_pickle(pickle.dumps(d, protocol=proto)))
            assert np.array_equal(got, want)
    assert tables.base_key16(0) == O.base_key16(0) == '0011010100001000'
    assert tables.base_key16(511) == '1111111110011111'
    assert tables.dumbbell_rate() == 0.0023195228302435697
    assert tables.LCE_ECI == O.LCE_ECI


def test_hex_dict_pickle_roundtrip():
    """hex_dict.p flavour: 2^18 x {18-char key -> ndarray(6)} through the restricted loader."""
    rs = np.random.RandomState(0)
    dense = rs.rand(1 << 18, 6)
    d = {format(k, '018b'): dense[k] for k in range(1 << 18)}
    blob = pickle.dumps(d, protocol=2)
    got = tables.dense_from_hex_dict(tables.load_table_pickle(blob))
    assert np.array_equal(got, dense)


def test_restricted_unpickler_refuses_code():
    class Evil:
        def __reduce__(self):
            return (os.system, ('echo pwned',))
    with pytest.raises(pickle.UnpicklingError):
        tables.load_table_pickle(pickle.dumps({'a': Evil()}))
    with pytest.raises(ValueError):
        tables.load_table_pickle(pickle.dumps([1, 2, 3]))
    with pytest.raises(KeyError):
        tables.base512_from_dict({'0011010100001000': 1.0})


def test_shard_and_statistics():
    conc, idx, offset = ensemble.shard_replicas([0.1, 0.5, 0.9], 512, rank=3, world=8)
    assert conc.size == 3 * 64 and np.bincount(idx).tolist() == [64, 64, 64] and offset == 3 * 64 * 3
    assert ensemble.shard_replicas([0.1], 10, 0, 4)[0].size == 2 and ensemble.shard_replicas([0.1], 10, 1, 4)[0].size == 3
    with pytest.raises(ValueError):
        ensemble.shard_replicas([0.1], 3, 0, 4)             # a rank without replicas
    rs = np.random.RandomState(0)
    ds, dj = rs.rand(192), rs.rand(192)
    acc = ensemble.accumulate(idx, 3, ds, dj, rs.rand(192), np.full(192, 100.0), np.ones((192, 6)))
    out = ensemble.finalize(acc, [0.1, 0.5, 0.9])
    for p in range(3):
        m = idx == p
        np.testing.assert_allclose(out[p]['d_star'], np.average(ds[m]), rtol=1e-13)
        np.testing.assert_allclose(out[p]['error_star'], np.std(ds[m]), rtol=1e-9)   # np.std, ddof 0 (:784)
        np.testing.assert_allclose(out[p]['error_j'], np.std(dj[m]), rtol=1e-9)
        assert out[p]['replicas'] == 64 and out[p]['hops'] == 6400


def test_hex_dict_writer_round_trip(tmp_path):
    """Next row 1: hex_dict.p for scripts that load it themselves; read back by the restricted loader."""
    import os
    from kmc_amd import tables
    from oracle import kmc_oracle as O
    base = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tables.npz'))['base512_normcons']
    dense = O.build_hex_table(base)
    p = str(tmp_path / 'hex_dict.p')
    tables.write_hex_dict(p, dense)
    kind, arr = tables.load_any(p)
    assert kind == 'dense' and np.array_equal(arr, dense)
    d = tables.load_table_pickle(p)
    assert len(d) == 1 << 18 and np.array_equal(d['111100110011001111'], dense[249039])


def test_default_table_lookup(tmp_path):
    """hex_dict.p, else normcons_dictionary.p, else master_dictionary.p, else the reference's own failure."""
    with pytest.raises(FileNotFoundError, match='hex_dict.p'):
        tables.find_default_table(str(tmp_path))
    for name in reversed(tables.DEFAULT_TABLE_FILES):
        (tmp_path / name).write_bytes(b'')
        assert os.path.basename(tables.find_default_table(str(tmp_path))) == name
    import kmc_amd.mc_lattice as m1
    import kmc_amd.Mc_Lattice as m2
    assert m1.mc_lattice is m2.mc_lattice
