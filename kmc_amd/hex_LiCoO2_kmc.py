"""Drop-in for the functional hex path of the reference (hex_LiCoO2_kmc.py).

Same function names, argument order, return tuples, module globals and error
messages as the reference's per-step loop; the work runs on the MI355X through
the C-ABI engine (no CPU fallback).  Reference lines are cited per function.

Typical use mirrors hex_LiCoO2_kmc.py:695-768::

    import kmc_amd.hex_LiCoO2_kmc as kmc
    kmc.set_rate_table('master_dictionary.p')        # or normcons_dictionary.p / hex_dict.p; without this call the
                                                     # first use loads hex_dict.p (else normcons_dictionary.p, else
                                                     # master_dictionary.p) from the working directory, as the
                                                     # reference does at import (:636, :644)
    kmc.mc_lattice_size = [30, 30, 1]; kmc.Li_concentration = 0.5
    kmc.mc_lattice = lattice                         # float array of -1 (Li) / +1 (Vac)
    P = kmc.kmc_evolve(lattice, P, 0, 0, True)
    lattice, P = kmc.kmc_step(lattice, dist, P, True)

Module globals the reference reads inside its functions (hex_LiCoO2_kmc.py:422-428,
:437, :484, :534-535) are module globals here too: ``mc_lattice``,
``mc_lattice_size``, ``Li_concentration``, ``use_initialize_mode``,
``hop_histogram``.  Uniform draws come from numpy's global RandomState exactly
like the reference's np.random.uniform calls (:490, :530), so a seeded script
reproduces the reference's trajectory hop for hop.
"""
import sys

import numpy as np

from . import engine as _engine
from . import tables as _tables

# ---- the reference's module constants (hex_LiCoO2_kmc.py:25-30) -------------
atom_types = ([["Li", "Vac"], [-1, 1]])
kb_t = 0.025
prefactor = 10 ** (13)
lattice_constant = 2.8334 * 10 ** (-8)
hop_histogram = np.zeros(6)

# ---- globals the reference's functions read ---------------------------------------
mc_lattice = None
mc_lattice_size = [0, 0, 0]
Li_concentration = 0.0
use_initialize_mode = False
window_mode = 'hex_ref'          # the reference's own update window (:446-449)
device = 0

_table = None                    # ('base', vec512, dumbbell) | ('dense', array)
_engines = {}
_row_lookup = None
_host_table = None               # host copy of the dense table (kmc_evolve's window patch), read back once per table
# What the device of each engine holds after the last kmc_step, as the arrays that call returned: a caller that hands
# them back unchanged (the reference's loop, hex_LiCoO2_kmc.py:724-758, does exactly that) finds the lattice, the stale
# keys behind its rate matrix and the displacement tags already there, and the call uploads nothing but its draws.
# Compared by VALUE (a few hundred KB at 64 x 64), never by identity: an array edited in place is noticed.
_resident = {}


def set_rate_table(source, dumbbell=None):
    """Select the hop-rate table: a path to master_dictionary.p /
    normcons_dictionary.p / hex_dict.p, a dict loaded from one of them, a
    512-vector, or a dense (262144, 6) array."""
    global _table, _row_lookup, _host_table
    if isinstance(source, (str, bytes)):
        kind, arr = _tables.load_any(source)
    elif isinstance(source, dict):
        kind, arr = ('dense', _tables.dense_from_hex_dict(source)) if len(source) >= 1 << 18 \
            else ('base', _tables.base512_from_dict(source))
    else:
        arr = np.asarray(source, dtype=np.float64)
        kind = 'dense' if arr.ndim == 2 else 'base'
    if kind == 'base':
        _table = ('base', arr, _tables.dumbbell_rate(prefactor, kb_t) if dumbbell is None else float(dumbbell))
    else:
        _table = ('dense', arr)
    for e in _engines.values():
        e.close()
    _engines.clear()
    _resident.clear()
    _row_lookup = None
    _host_table = None


def _engine_for(nx, ny):
    if _table is None:
        # the reference loads hex_dict.p from the working directory when it is imported (:636, :644)
        set_rate_table(_tables.find_default_table('.'))
    key = (int(nx), int(ny), window_mode, device)
    eng = _engines.get(key)
    if eng is None:
        eng = _engine.KMCEngine(1, nx, ny, window_mode=window_mode, device=device)
        if _table[0] == 'base':
            eng.set_base_table(_table[1], _table[2])
        else:
            eng.set_hex_table(_table[1])
        _engines[key] = eng
    return eng


def hex_table():
    """The dense 2^18 x 6 table in use (device copy read back)."""
    return _engine_for(8, 8).get_hex_table()


def _occ(lattice):
    a = np.asarray(lattice)
    a = a.reshape(a.shape[0], a.shape[1])
    return (a == atom_types[1][0]).astype(np.int8)


def _keys_from_rates(eng, occ, P):
    """Stale 18-bit key of every site from a caller-supplied rate matrix: fresh
    key where the rates are fresh, otherwise any key whose table row equals the
    stored rates (rows with equal rates are interchangeable)."""
    global _row_lookup
    nx, ny = occ.shape
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(nx, ny, 6))
    eng.set_lattice(occ[None])
    eng.rates_full()
    keys = eng.get_keys()[0].copy()
    fresh = eng.get_rates()[0]
    stale = np.any(P != fresh, axis=-1)
    if stale.any():
        if _row_lookup is None:
            tab = eng.get_hex_table()
            _row_lookup = {}
            for k in range(tab.shape[0] - 1, -1, -1):
                _row_lookup[tab[k].tobytes()] = k
        for i, j in zip(*np.nonzero(stale)):
            if not P[i, j].any():
                keys[i, j] = _engine.VACANT_KEY
                continue
            k = _row_lookup.get(P[i, j].tobytes())
            if k is None:
                raise ValueError('hop_probability_matrix[%d][%d] is not a row of the rate table' % (i, j))
            keys[i, j] = k
    return keys


# ---- R1 -----------------------------------------------------------------------
def get_hex_probability(lattice_point):
    """hex_LiCoO2_kmc.py:418-428.  Reads the module-global ``mc_lattice`` like the
    reference does; returns the six hop rates of the site's 18-site environment."""
    i, j = int(lattice_point[0]), int(lattice_point[1])
    occ = _occ(mc_lattice)
    eng = _engine_for(*occ.shape)
    _resident.pop(id(eng), None)
    eng.set_lattice(occ[None])
    _, rates = eng.site_probability(0, i % occ.shape[0], j % occ.shape[1])
    return rates


# ---- R2 / R3 --------------------------------------------------------------------
def kmc_evolve(mc_lattice, hop_probability_matrix, startpoint, endpoint, use_initialize_mode):
    """hex_LiCoO2_kmc.py:432-467.  Initialise mode rebuilds the whole rate matrix;
    otherwise only the reference's window around (startpoint, endpoint) -- the
    end point is the UNWRAPPED one -- is re-evaluated and the rest of
    ``hop_probability_matrix`` is kept (and updated in place, as in the reference)."""
    occ = _occ(mc_lattice)
    nx, ny = occ.shape
    eng = _engine_for(nx, ny)
    if use_initialize_mode == True:                                       # noqa: E712
        _resident.pop(id(eng), None)
        eng.set_lattice(occ[None])
        eng.rates_full()
        return eng.get_rates()[0].reshape(nx, ny, 1, 6)
    # Only the window is re-evaluated (from the current lattice); every other
    # entry of the caller's matrix is left as it is -- stale or not.
    # On the device: the lattice goes up (nx * ny bytes), the sites are keyed, refresh_window_kernel re-keys the window's
    # sites with the selected window's geometry, the keys come back (4 bytes a site) and the window's <= 36 rows are taken
    # from the host copy of the table: no rate matrix is materialised or moved (48 bytes a site).
    global _host_table
    _resident.pop(id(eng), None)                   # the device no longer holds what the last kmc_step returned
    eng.set_lattice(occ[None])
    eng.rates_full()
    se = [int(startpoint[0]), int(startpoint[1]), int(endpoint[0]), int(endpoint[1])]
    eng.refresh_window([se])                       # R3 on the device (same sites, same keys)
    keys = eng.get_keys()[0]
    if _host_table is None:
        _host_table = eng.get_hex_table()
    P = np.asarray(hop_probability_matrix)
    Pv = P.reshape(nx, ny, 6)
    i_lo, i_hi, j_lo, j_hi = _window(startpoint, endpoint)
    for i in range(i_lo, i_hi):
        for j in range(j_lo, j_hi):
            k = keys[i % nx, j % ny]
            Pv[i % nx, j % ny] = _host_table[k] if k != _engine.VACANT_KEY else 0.0
    return hop_probability_matrix


def _window(startpoint, endpoint):
    """Half-open window of the selected update mode (hex_LiCoO2_kmc.py:446-449;
    Mc_Lattice.py:178; inclusive box for 'full')."""
    (i_s, j_s), (i_e, j_e) = (int(startpoint[0]), int(startpoint[1])), (int(endpoint[0]), int(endpoint[1]))
    if window_mode == 'hex_ref':
        return min(i_s, i_e) - 2, max(i_s, i_e) + 2, min(j_s, j_e) - 2, max(j_s, j_e) + 2
    if window_mode == 'obj_ref':
        return i_s - 2, i_s + 3, j_s - 2, j_s + 4
    return min(i_s, i_e) - 2, max(i_s, i_e) + 3, min(j_s, j_e) - 2, max(j_s, j_e) + 3


# ---- K (S1-S5, U1-U4) ---------------------------------------------------------------
def kmc_step(mc_lattice, input_distance_lattice, input_hop_probability_matrix, use_presimulation_mode):
    """hex_LiCoO2_kmc.py:470-542: int(N*N*1*x) hops.

    ``mc_lattice`` is updated in place AND returned (:518-520, :539); the distance
    lattice and the rate matrix are copies (:474-477).  Returns
    ``(mc_lattice, distance_lattice, time_step_per_kmcstep, hop_probability_matrix,
    hop_histogram, chosen_barrier)`` or ``(mc_lattice, hop_probability_matrix)`` in
    presimulation mode.
    """
    global hop_histogram
    lat = mc_lattice
    occ = _occ(lat)
    nx, ny = occ.shape
    n_hops = int(mc_lattice_size[0] * mc_lattice_size[1] * mc_lattice_size[2] * Li_concentration)   # :484
    eng = _engine_for(nx, ny)
    presim = bool(use_presimulation_mode)
    res = _resident.pop(id(eng), None)
    P_in = np.asarray(input_hop_probability_matrix)
    if not (res is not None and np.array_equal(res['occ'], occ) and res['P'].shape == P_in.shape and np.array_equal(res['P'], P_in)):
        # not what the last call left on the device: derive the stale keys from the caller's matrix, upload
        keys = _keys_from_rates(eng, occ, input_hop_probability_matrix)
        eng.set_keys(keys[None])
        res = None
    if not presim:
        d_in = np.asarray(input_distance_lattice)
        if not (res is not None and res.get('dist') is not None and res['dist'].shape == d_in.shape and np.array_equal(res['dist'], d_in)):
            d = d_in.reshape(nx, ny, 3).astype(np.int64)
            w_i = np.round(d[..., 2] / 100.0).astype(np.int64)            # :574 (valid for |w_j| <= 49)
            tags = np.stack([d[..., 0], d[..., 1], w_i, d[..., 2] - 100 * w_i], axis=-1).astype(np.int32)
            eng.set_displacements(tags[None])
    # the reference draws rho (:490) and, outside presimulation, the clock draw
    # (:530) from numpy's global stream, in this order, once per hop
    draws = np.random.uniform(0, 1, size=(n_hops, 1 if presim else 2))
    if presim:
        draws = np.concatenate([draws, np.ones((n_hops, 1))], axis=1)
    # the engine's running histogram and barrier count before this call: what the last call read back, while the state
    # it left is still the one on the device (two read-backs less per call)
    hist0 = res['hist'] if res is not None and 'hist' in res else eng.scalars(('hist',))['hist'][0].copy()
    bar0 = 0
    if not presim:            # (a presimulation step moves the count too and does not read it back: not cached across one)
        bar0 = res['bar_count'] if res is not None and res.get('bar_count') is not None else eng.barrier_stats()['count'][0]
    eng.run(n_hops, presim=presim, draws=draws[None])
    st = eng.status(('status',))
    if st['status'][0] & 1:
        raise IndexError('index %d is out of bounds for axis 0 with size %d'
                         % (nx * ny * 6, nx * ny * 6))               # the reference's failure at :494-506
    if st['status'][0] & 2:
        raise ZeroDivisionError('no hop possible: np.sum(hop_probability_matrix) == 0')
    new_occ = eng.get_lattice()[0]
    vals = np.where(new_occ == 1, atom_types[1][0], atom_types[1][1]).astype(np.asarray(lat).dtype)
    np.asarray(lat)[...] = vals.reshape(np.asarray(lat).shape)          # in place, like :518-520
    P = eng.get_rates()[0].reshape(nx, ny, 1, 6)
    sc = eng.scalars(('step_time', 'hist'))
    hop_histogram += (sc['hist'][0] - hist0)                           # module global (:30, :535)
    if presim:
        _resident[id(eng)] = dict(occ=new_occ, P=P.copy(), dist=None, hist=sc['hist'][0].copy(), bar_count=None)
        return (lat, P)
    t = eng.get_reference_tags()[0].reshape(nx, ny, 1, 3)
    dist = np.array(t, dtype=np.asarray(input_distance_lattice).dtype)
    # (the tags are resident only if the caller's dtype holds them exactly, which any integer or float dtype does)
    bar = eng.barrier_stats()
    _resident[id(eng)] = dict(occ=new_occ, P=P.copy(), dist=dist.copy(), hist=sc['hist'][0].copy(), bar_count=bar['count'][0])
    if bar['count'][0] == bar0:
        # every selected index was 0: the reference's chosen_barrier is unbound (:495, :539)
        raise UnboundLocalError("local variable 'chosen_barrier' referenced before assignment")
    return (lat, dist, sc['step_time'][0], P, hop_histogram, bar['last'][0])


# ---- D1 / D2 --------------------------------------------------------------------
def _tags_to_engine(input_lattice, size):
    nx, ny = int(size[0]), int(size[1])
    d = np.asarray(input_lattice).reshape(nx, ny, 3).astype(np.int64)
    w_i = np.round(d[..., 2] / 100.0).astype(np.int64)
    tags = np.stack([d[..., 0], d[..., 1], w_i, d[..., 2] - 100 * w_i], axis=-1).astype(np.int32)
    eng = _engine_for(nx, ny)
    eng.set_displacements(tags[None])
    return eng, d


def get_distance(input_lattice, mc_lattice_size, lattice_constant):
    """hex_LiCoO2_kmc.py:558-586 -> (matrix * a, list * a, d_j * a)."""
    nx, ny = int(mc_lattice_size[0]), int(mc_lattice_size[1])
    eng, d = _tags_to_engine(input_lattice, mc_lattice_size)
    mat = eng.distances(lattice_constant)[0]
    occupied = d[..., 0] >= 0
    lst = mat[occupied]                                   # C order, as the reference appends (:580-581)
    eng.set_time(1.0)
    _, _, msd = eng.diffusion(1.0, lattice_constant)
    s_i, s_j = msd[0, 1], msd[0, 2]
    d_j = np.sqrt(np.abs(s_i ** 2 + s_j ** 2 + s_i * s_j))
    return mat.reshape(nx, ny, 1), lst, d_j * lattice_constant


def get_diffusion_coefficient(distance_lattice, mc_lattice_size, time_diffusing, lattice_constant, Li_concentration):
    """hex_LiCoO2_kmc.py:589-603 -> (D*, list, D_J)."""
    number_of_atoms = mc_lattice_size[0] * mc_lattice_size[1] * mc_lattice_size[2] * Li_concentration
    if number_of_atoms == 0:
        sys.exit("concentration is zero, no atoms to move!")
    eng, d = _tags_to_engine(distance_lattice, mc_lattice_size)
    eng.set_time(float(time_diffusing))
    ds, dj, _ = eng.diffusion(float(Li_concentration), lattice_constant)
    lst = eng.distances(lattice_constant)[0][d[..., 0] >= 0]
    return ds[0], lst, dj[0]


# ---- G / table builders -------------------------------------------------------------
def dictionary_builder(path=None):
    """hex_LiCoO2_kmc.py:345-365: the 512-entry base table from the local cluster
    expansion at (prefactor, kb_t), computed on the device; optionally pickled in
    the reference's normcons_dictionary.p format.  Returns the dict."""
    import pickle
    eng = _engine.KMCEngine(1, 8, 8, device=device)
    try:
        base, _ = eng.generate_base_table(_tables.LCE_ECI, prefactor, kb_t)
    finally:
        eng.close()
    d = {_tables.base_key16(i): float(base[i]) for i in range(512)}
    if path is not None:
        with open(path, 'wb') as f:
            pickle.dump(d, f, protocol=2)
    return d


def close():
    for e in _engines.values():
        e.close()
    _engines.clear()
    _resident.clear()
