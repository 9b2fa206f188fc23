"""The reference's self-check drivers as batched ensembles.

``class mc_lattice`` carries four methods that are really scripts (Mc_Lattice.py:243-302): each builds
lattices one after the other, steps them in Python and prints a number per lattice.  Here the averaging
iterations of one size / concentration are replicas of ONE engine, so a check is a few kernel launches:

* ``dimension_check``  (:243-258)  D_J against lattice size, x = 0.2, 200 + 500 steps, 10 lattices per size;
* ``HH_check``         (:260-271)  hop_histogram[1] / hop_histogram[0] per lattice: 1 if the table is isotropic;
* ``counts_check``     (:273-282)  D_J after every KMC step of one long run (convergence with counts);
* ``concentration_run``(:284-302)  mean and np.std of D_J and D* for x = 0.90 ... 0.98.

Coefficients use the class's normalisation (Mc_Lattice.py:229-232: both over 4 t N_Li) and the class's
5 x 6 update window (``obj_ref``), like ``kmc_amd.Mc_Lattice``.  Uniforms come from the engine's Philox
stream (seeded), lattices from ``fill_random``; one "step" is N_Li = int(x nx ny) hops (:112-114, :200).
Each function returns its numbers and, with ``echo``, prints them in the reference's format.

One difference is deliberate: ``concentration_run`` of the reference re-uses ONE lattice for its ten
averaging iterations (each continues where the last stopped); here they are ten independent replicas.
"""
import numpy as np

from . import engine as _engine
from . import tables as _tables

LATTICE_CONSTANT = 2.8334 * 10 ** (-8)                       # Mc_Lattice.py:63


def _make(factory, table, R, n, device, window_mode):
    eng = (factory or _engine.KMCEngine)(R, n, n, window_mode, device)
    kind = table[0]
    if kind == 'base':
        eng.set_base_table(table[1], table[2] if len(table) > 2 and table[2] is not None else _tables.dumbbell_rate())
    else:
        eng.set_hex_table(table[1])
    return eng


def _check(eng):
    st = eng.status()
    if st['status'].any():
        raise RuntimeError('replicas stopped early: status %r' % np.unique(st['status']))


def class_coefficients(eng, number_of_li, lattice_constant=LATTICE_CONSTANT):
    """(d_star[R], d_j[R]) with the class's normalisation, Mc_Lattice.py:223-232."""
    n_sites = eng.nx * eng.ny
    _, _, msd = eng.diffusion(number_of_li / float(n_sites), lattice_constant)
    t = eng.scalars()['time']
    a2 = lattice_constant ** 2
    s_i, s_j = msd[:, 1], msd[:, 2]
    d_j = a2 * (s_i ** 2 + s_j ** 2 + s_i * s_j) / (2 * 2 * t * number_of_li)
    d_star = a2 * msd[:, 0] / (2 * 2 * t * number_of_li)
    return d_star, d_j


def dimension_check(table, dimensions=range(10, 60, 10), averaging=10, concentration=0.2, presim_steps=200,
                    sim_steps=500, seed=0, device=0, engine_factory=None, window_mode='obj_ref', echo=False):
    """{dimension: d_j[averaging]} (Mc_Lattice.py:243-258)."""
    out = {}
    for dimension in dimensions:
        n_li = int(concentration * (dimension * dimension))
        eng = _make(engine_factory, table, averaging, dimension, device, window_mode)
        try:
            eng.fill_random(concentration, seed + dimension)
            eng.rates_full()
            eng.run(n_li * presim_steps, presim=True, seed=seed + dimension)
            eng.init_displacements()
            eng.run(n_li * sim_steps, presim=False, seed=seed + dimension)
            _check(eng)
            out[dimension] = class_coefficients(eng, n_li)[1]
        finally:
            eng.close()
        if echo:
            print(dimension)
            for v in out[dimension]:
                print(v)
    return out


def HH_check(table, dimensions=range(10, 50, 10), averaging=10, concentration=0.2, steps=500, seed=0, device=0,
             engine_factory=None, window_mode='obj_ref', echo=False):
    """{dimension: hop_histogram[1] / hop_histogram[0] per lattice} (Mc_Lattice.py:260-271); the lattices stay
    in presimulation mode, as in the reference (no initialize_KMC)."""
    out = {}
    for dimension in dimensions:
        n_li = int(concentration * (dimension * dimension))
        eng = _make(engine_factory, table, averaging, dimension, device, window_mode)
        try:
            eng.fill_random(concentration, seed + dimension)
            eng.rates_full()
            eng.run(n_li * steps, presim=True, seed=seed + dimension)
            _check(eng)
            h = eng.scalars()['hist'].astype(np.float64)
        finally:
            eng.close()
        out[dimension] = h[:, 1] / h[:, 0]
        if echo:
            print(dimension)
            for v in out[dimension]:
                print(v)
            print('averaged HH: ' + str(out[dimension][-1]))          # the reference prints the LAST lattice's (:270)
    return out


def counts_check(table, dimension=50, concentration=0.2, steps=1000, stride=1, replicas=1, seed=0, device=0,
                 engine_factory=None, window_mode='obj_ref'):
    """d_j after every ``stride`` KMC steps, shape (steps // stride, replicas) (Mc_Lattice.py:273-282: one
    50 x 50 lattice, 100 000 steps, no presimulation; the scatter plot is the caller's)."""
    n_li = int(concentration * (dimension * dimension))
    eng = _make(engine_factory, table, replicas, dimension, device, window_mode)
    curve = []
    try:
        eng.fill_random(concentration, seed)
        eng.rates_full()
        eng.init_displacements()
        for _ in range(steps // stride):
            eng.run(n_li * stride, presim=False, seed=seed)
            curve.append(class_coefficients(eng, n_li)[1])
        _check(eng)
    finally:
        eng.close()
    return np.array(curve).reshape(-1, replicas)


def concentration_run(table, dimension, concentrations=np.array(range(90, 100, 2)) / 100., averaging=10,
                      presim_steps=500, sim_steps=750, seed=0, device=0, engine_factory=None,
                      window_mode='obj_ref', echo=False):
    """[(x, mean d_j, std d_j, mean d_star, std d_star)] (Mc_Lattice.py:284-302); all concentrations run in one
    engine, ``averaging`` replicas each, every replica taking its own N_Li hops per step."""
    conc = np.repeat(np.asarray(concentrations, dtype=np.float64), averaging)
    n_li = (conc * (dimension * dimension)).astype(np.int64)      # int(x * prod(dimensions)), :112
    eng = _make(engine_factory, table, conc.size, dimension, device, window_mode)
    try:
        eng.fill_random(conc, seed)
        eng.rates_full()
        eng.run_counts(n_li * presim_steps, presim=True, seed=seed)
        eng.init_displacements()
        eng.run_counts(n_li * sim_steps, presim=False, seed=seed)
        _check(eng)
        d_star, d_j = class_coefficients(eng, n_li)
    finally:
        eng.close()
    out = []
    for k, x in enumerate(np.asarray(concentrations, dtype=np.float64)):
        s = slice(k * averaging, (k + 1) * averaging)
        out.append((float(x), np.mean(d_j[s]), np.std(d_j[s]), np.mean(d_star[s]), np.std(d_star[s])))
        if echo:
            print(str(x) + ' ' + str(out[-1][1]) + ' ' + str(out[-1][2]))                          # :301
            print(str(x) + ' ' + str(out[-1][3]) + ' ' + str(out[-1][4]))                          # :302
    return out
