"""Rate-table loading (row T).

The reference persists its tables as pickles of plain dicts:

* ``normcons_dictionary.p``: 512 x {16-char '0'/'1' str -> float}
  (writer hex_LiCoO2_kmc.py:351-365)
* ``master_dictionary.p``: 8192 x {16-digit int of 1 (Li) / 3 (Vac) -> float}
  (reader LiCoO2_kmc.py:491)
* ``hex_dict.p``: 2^18 x {18-char str -> ndarray(6)} (writer hex_dict_builder.py:147)

They are read with a restricted unpickler that refuses every global except the
numpy array reconstruction helpers ``hex_dict.p`` needs, so loading a table
never executes code from the file.  Only the *host-side bookkeeping* lives
here; the 2^18 x 6 fold itself runs on the GPU (kmc_set_base_table).
"""
import io
import math
import pickle

import numpy as np

KB_T = 0.025                       # hex_LiCoO2_kmc.py:27
PREFACTOR = 10 ** 13               # hex_LiCoO2_kmc.py:28
DUMBBELL_BARRIER = 0.900           # hex_dict_builder.py:29-30
BASE_FREE_POSITIONS = (0, 1, 4, 6, 8, 11, 13, 14, 15)   # hex_dict_builder.py:58-62

# the 47 effective cluster interactions of the local cluster expansion,
# hex_LiCoO2_kmc.py:125-126 (physical parameters, eV)
LCE_ECI = (
    0.5039840998, 0.0231779629, -0.0124155613, 0.0190676809, -0.0240978034,
    0.0359364321, -0.0322688382, 0.0189131728, 0.0059693908, -0.0107495262,
    0.0044709962, 0.0076906411, -0.0135407679, 0.0014011203, -0.0008762663,
    -0.0015711384, -0.0074600309, -0.0049897914, 0.0098191199, 0.0087838401,
    0.0032917204, 0.0075716889, 0.0020836078, -0.0043424798, -0.0107164342,
    -0.0079384991, -0.0047555242, 0.0061737865, 0.0100261837, 0.0016684245,
    0.0040444273, 0.007575093, -0.0146307688, -0.0101701531, 0.0011528037,
    -0.0028328131, -0.0020787791, 0.0038948159, 0.0131163879, 0.0048079312,
    0.0016672859, 0.0047770905, 0.0114784225, 0.0028142975, -0.002387814,
    -0.0109707522, -0.0032730688,
)


def dumbbell_rate(prefactor=PREFACTOR, kb_t=KB_T):
    """hex_dict_builder.py:29-30."""
    return prefactor * math.exp(-(DUMBBELL_BARRIER) / kb_t)


class _Restricted(pickle.Unpickler):
    _ALLOWED = {
        ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'),
        ('_codecs', 'encode'),          # py3-written protocol-2 arrays: str -> bytes, no side effects
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError('table pickle may not reference %s.%s' % (module, name))


def load_table_pickle(path_or_bytes):
    """dict from a reference-style table pickle (py2 or py3), nothing executed."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, 'rb') as f:
            data = f.read()
    d = _Restricted(io.BytesIO(data), encoding='latin1').load()
    if not isinstance(d, dict):
        raise ValueError('table pickle does not hold a dict')
    return d


def base_key16(index):
    """16-char key of base entry ``index``: forced '1' at 2,3,5,7,12 and '0' at
    9,10 (hex_LiCoO2_kmc.py:353-354), nine free chars MSB first."""
    chars = list('1111111110011111')
    for q, pos in enumerate(BASE_FREE_POSITIONS):
        chars[pos] = '1' if (index >> (8 - q)) & 1 else '0'
    return ''.join(chars)


def base512_from_dict(d):
    """512-vector from either dictionary flavour (str '0'/'1' keys or int 1/3
    keys); a superset table (master_dictionary.p, 8192 keys) is restricted to
    the keys with sites 2,3,7,12 = Li (hex_dict_builder.py:31, :58-62)."""
    if not d:
        raise ValueError('empty table')
    sample = next(iter(k for k in d if k != '3'))
    out = np.empty(512, dtype=np.float64)
    for i in range(512):
        k = base_key16(i)
        if not isinstance(sample, str):
            k = int(k.replace('0', '3'))
        try:
            out[i] = d[k]
        except KeyError:
            raise KeyError('table has no entry for local environment %r' % (k,)) from None
    return out


def dense_from_hex_dict(d):
    """(262144, 6) array from a hex_dict.p-style dict (row = int(key, 2))."""
    out = np.zeros((1 << 18, 6), dtype=np.float64)
    if len(d) != 1 << 18:
        raise ValueError('hex dict must have 2^18 entries, has %d' % len(d))
    for k, v in d.items():
        out[int(k, 2)] = v
    return out


def load_any(path):
    """('base', vec512) or ('dense', table) from any of the three pickles."""
    d = load_table_pickle(path)
    if len(d) >= 1 << 18:
        return 'dense', dense_from_hex_dict(d)
    return 'base', base512_from_dict(d)


DEFAULT_TABLE_FILES = ('hex_dict.p', 'normcons_dictionary.p', 'master_dictionary.p')


def find_default_table(directory='.'):
    """The table a reference script would have loaded by itself, looked up in ``directory`` (the working
    directory): ``hex_dict.p`` (read at import by hex_LiCoO2_kmc.py:636 / :644 and in the class body,
    Mc_Lattice.py:64), else ``normcons_dictionary.p`` (what hex_dict_builder.py:20 folds into it), else
    ``master_dictionary.p`` (LiCoO2_kmc.py:491; the table BASELINE's north star names).  Returns the path; raises
    the reference's own failure -- the missing hex_dict.p -- if none is there."""
    import errno
    import os
    for name in DEFAULT_TABLE_FILES:
        path = os.path.join(directory, name)
        if os.path.exists(path):
            return path
    raise FileNotFoundError(errno.ENOENT, 'No such file or directory (nor normcons_dictionary.p / master_dictionary.p '
                            'beside it; or call set_rate_table(...))', 'hex_dict.p')


def write_hex_dict(path, dense, protocol=2):
    """``hex_dict.p`` as hex_dict_builder.py:143-147 writes it: {18-char '0'/'1' key -> ndarray(6)}, 2^18
    entries, for scripts that ``pickle.load`` it themselves (Mc_Lattice.py:64).  ``dense`` is the folded
    table, e.g. ``KMCEngine.get_hex_table()``."""
    t = np.ascontiguousarray(dense, dtype=np.float64)
    if t.shape != (1 << 18, 6):
        raise ValueError('dense table must be (262144, 6)')
    d = {format(k, '018b'): t[k].copy() for k in range(1 << 18)}
    with open(path, 'wb') as f:
        pickle.dump(d, f, protocol)
