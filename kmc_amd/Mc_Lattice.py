"""Drop-in for the reference's object-oriented path: ``class mc_lattice``
(Mc_Lattice.py:60-232; mc_lattice.py / object_KMC.py are stubs of the same class).

Same constructor, methods and attributes::

    x = mc_lattice([N, N, 1]); x.fill_lattice(c)
    for z in range(n_presim): x.KMC_step()
    x.initialize_KMC()
    for z in range(n_sim): x.KMC_step()
    x.calculate_diffusion_coefficients(); x.d_star, x.d_j

The state lives on the GPU; ``lattice``, ``probability_matrix`` and
``distance_lattice`` are read back on access.  Numerics follow the functional
path hex_LiCoO2_kmc.py: the 18-site environment is read as
hex_LiCoO2_kmc.py:423-424 (the class's interior fast path transposes the
lattice, Mc_Lattice.py:95-100, SURVEY B2 -- not reproduced), the clock is the
sum of the per-hop waiting times (Mc_Lattice.py:175-176/203 adds a never-reset
running sum, SURVEY B3 -- not reproduced).  The update window is the class's
own 5 x 6 window around the start point (Mc_Lattice.py:178) unless
``window_mode`` says otherwise.  Diffusion coefficients use the class's own
normalisation (Mc_Lattice.py:229-232) by default, ``functional_normalisation``
switches to hex_LiCoO2_kmc.py:599-602.
"""
import random

import numpy as np

from . import engine as _engine
from . import tables as _tables


class mc_lattice:
    "general parameters of mc_lattice"
    atom_types = {'Li': '1', 'Vac': '0'}                   # Mc_Lattice.py:62
    lattice_constant = 2.8334 * 10 ** (-8)                 # Mc_Lattice.py:63
    rate_table = None                                      # class-level table, like hex_dict (:64)
    window_mode = 'obj_ref'
    functional_normalisation = False
    device = 0

    @classmethod
    def set_rate_table(cls, source, dumbbell=None):
        """Class-wide hop-rate table (the reference loads hex_dict.p in the class
        body, Mc_Lattice.py:64): a pickle path, a dict, a 512-vector or a dense array."""
        if isinstance(source, (str, bytes)):
            kind, arr = _tables.load_any(source)
        elif isinstance(source, dict):
            kind, arr = ('dense', _tables.dense_from_hex_dict(source)) if len(source) >= 1 << 18 \
                else ('base', _tables.base512_from_dict(source))
        else:
            arr = np.asarray(source, dtype=np.float64)
            kind = 'dense' if arr.ndim == 2 else 'base'
        cls.rate_table = (kind, arr, _tables.dumbbell_rate() if dumbbell is None else float(dumbbell))

    def __init__(self, mc_lattice_size):
        # Mc_Lattice.py:66-71
        self.dimensions = [mc_lattice_size[0], mc_lattice_size[1], mc_lattice_size[2]]
        if self.dimensions[2] != 1:
            raise ValueError('the hex path is two-dimensional: mc_lattice_size[2] must be 1')
        self.hop_histogram = np.zeros(6)
        self._eng = None
        self._hist_base = np.zeros(6, dtype=np.int64)
        self.use_presimulation_mode = True
        self.time_step_per_hop = 0
        self.time_step_per_kmcstep = 0
        self.total_time = 0

    # -- engine ------------------------------------------------------------------
    def _engine(self):
        if self._eng is None:
            if type(self).rate_table is None:
                # the reference's class body loads hex_dict.p from the working directory (Mc_Lattice.py:64)
                type(self).set_rate_table(_tables.find_default_table('.'))
            kind, arr, dumbbell = type(self).rate_table
            self._eng = _engine.KMCEngine(1, self.dimensions[0], self.dimensions[1],
                                          window_mode=self.window_mode, device=self.device)
            if kind == 'base':
                self._eng.set_base_table(arr, dumbbell)
            else:
                self._eng.set_hex_table(arr)
        return self._eng

    def close(self):
        if self._eng is not None:
            self._eng.close()
            self._eng = None

    # -- state views ----------------------------------------------------------------
    @property
    def lattice(self):
        occ = self._engine().get_lattice()[0]
        return np.where(occ == 1, self.atom_types['Li'], self.atom_types['Vac']).reshape(self.dimensions)

    @property
    def probability_matrix(self):
        return self._engine().get_rates()[0].reshape(self.dimensions + [6])

    @property
    def distance_lattice(self):
        return self._engine().get_reference_tags()[0].reshape(self.dimensions + [3])

    def get_hex_probability(self, lattice_point):
        """Mc_Lattice.py:73-110: six rates of an occupied site, zeros otherwise."""
        i, j = int(lattice_point[0]), int(lattice_point[1])
        eng = self._engine()
        if eng.get_lattice()[0][i, j] != 1:
            return np.zeros(6)
        return eng.site_probability(0, i, j)[1]

    # -- Mc_Lattice.py:112-138 ----------------------------------------------------------
    def fill_lattice(self, concentration):
        nx, ny, nz = self.dimensions
        self.number_of_li = int(concentration * np.prod(self.dimensions))
        occ = np.zeros((nx, ny), dtype=np.int8)
        Li_atoms = 0
        while Li_atoms < self.number_of_li:                # same draw sequence as :122-128
            i = np.random.randint(0, nx)
            j = np.random.randint(0, ny)
            np.random.randint(0, nz)
            if occ[i, j] == 0:
                occ[i, j] = 1
                Li_atoms += 1
        eng = self._engine()
        eng.set_lattice(occ[None])
        eng.rates_full()
        self.use_presimulation_mode = True

    def set_lattice(self, lattice):
        """Start from a given '1'/'0' (or 1/0) lattice instead of a random fill."""
        a = np.asarray(lattice).reshape(self.dimensions[0], self.dimensions[1])
        occ = (a.astype(str) == '1').astype(np.int8)
        self.number_of_li = int(occ.sum())
        eng = self._engine()
        eng.set_lattice(occ[None])
        eng.rates_full()
        self.use_presimulation_mode = True

    # -- Mc_Lattice.py:140-183 / :198-203 -------------------------------------------------
    def _run(self, n_hops):
        eng = self._engine()
        presim = bool(self.use_presimulation_mode)
        # rho = random.random() (:146); outside presimulation a second draw feeds the clock (:174)
        if presim:
            draws = np.array([[random.random(), 1.0] for _ in range(n_hops)]).reshape(n_hops, 2)
        else:
            draws = np.array([[random.random(), random.random()] for _ in range(n_hops)]).reshape(n_hops, 2)
        eng.run(n_hops, presim=presim, draws=draws[None])
        st = eng.status()
        if st['status'][0] & 1:
            n6 = 6 * self.dimensions[0] * self.dimensions[1]
            raise IndexError('index %d is out of bounds for axis 0 with size %d' % (n6, n6))
        if st['status'][0] & 2:
            raise ZeroDivisionError('no hop possible: np.sum(probability_matrix) == 0')
        sc = eng.scalars()
        self.hop_histogram = (sc['hist'][0] - self._hist_base).astype(np.float64)
        if not presim:
            self.time_step_per_kmcstep = sc['step_time'][0]
            self.total_time = sc['time'][0]

    def hop(self):
        """One hop (Mc_Lattice.py:140-183)."""
        self._run(1)

    def KMC_step(self):
        """number_of_li hops (Mc_Lattice.py:198-203)."""
        self._run(self.number_of_li)

    # -- Mc_Lattice.py:185-196 -----------------------------------------------------------
    def initialize_KMC(self):
        self._engine().init_displacements()
        self.use_presimulation_mode = False
        self.time_step_per_kmcstep = 0
        self.total_time = 0

    # -- Mc_Lattice.py:205-232 -----------------------------------------------------------
    def calculate_diffusion_coefficients(self):
        eng = self._engine()
        n_sites = self.dimensions[0] * self.dimensions[1]
        conc = self.number_of_li / float(n_sites)
        ds, dj, msd = eng.diffusion(conc, self.lattice_constant)
        a2 = self.lattice_constant ** 2
        if self.functional_normalisation:
            self.d_star, self.d_j = ds[0], dj[0]           # hex_LiCoO2_kmc.py:599-602
        else:
            # Mc_Lattice.py:223-232: both divided by 4 * t * N_Li
            s_i, s_j = msd[0, 1], msd[0, 2]
            self.d_j = a2 * (s_i ** 2 + s_j ** 2 + s_i * s_j) / (2 * 2 * self.total_time * self.number_of_li)
            self.d_star = a2 * msd[0, 0] / (2 * 2 * self.total_time * self.number_of_li)
        mat = eng.distances(self.lattice_constant)[0]
        self.distances_travelled_list = mat[eng.get_displacements()[0][..., 0] >= 0]

    def reset_histogram(self):
        """x.hop_histogram = np.zeros(6) of the reference's drivers (:258, :271)."""
        self._hist_base = self._engine().scalars()['hist'][0].copy()
        self.hop_histogram = np.zeros(6)
