"""ctypes binding of oracle/kmc_oracle_c.c -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
WINDOW_MODES = {'hex_ref': 0, 'obj_ref': 1, 'full': 2}


def build():
    subprocess.check_call(['make', '-s', '-C', HERE, 'libkmc_oracle.so'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, 'libkmc_oracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp, u8p, u32p, i64p, i32p = (C.POINTER(t) for t in
                                     (C.c_double, C.c_uint8, C.c_uint32, C.c_int64, C.c_int32))
        L.kmco_build_table.argtypes = [dp, C.c_double, dp]
        L.kmco_rates_full.argtypes = [u8p, C.c_int, C.c_int, dp, dp, u32p]
        L.kmco_numpy_sum.argtypes = [dp, C.c_long]
        L.kmco_numpy_sum.restype = C.c_double
        L.kmco_run.argtypes = [u8p, C.c_int, C.c_int, dp, dp, u32p, i64p, C.c_long, C.c_int,
                               C.c_int, dp, i32p, dp, dp, dp, dp]
        L.kmco_run.restype = C.c_long
        L.kmco_diffusion.argtypes = [i64p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, dp]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def build_table(base512, dumbbell):
    t = np.empty((1 << 18, 6))
    lib().kmco_build_table(_p(np.ascontiguousarray(base512, dtype=np.float64), C.c_double),
                           float(dumbbell), _p(t, C.c_double))
    return t


def numpy_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    return lib().kmco_numpy_sum(_p(a, C.c_double), a.size)


class Replica:
    """One lattice evolved by the C oracle (state arrays are numpy, in place)."""

    def __init__(self, occ, table, window_mode='hex_ref'):
        self.occ = np.ascontiguousarray(occ, dtype=np.uint8).copy()
        self.nx, self.ny = self.occ.shape
        self.table = np.ascontiguousarray(table, dtype=np.float64)
        self.P = np.zeros((self.nx, self.ny, 6))
        self.keys = np.zeros((self.nx, self.ny), dtype=np.uint32)
        self.tags = np.zeros((self.nx, self.ny, 3), dtype=np.int64)
        self.tags[..., 0] = -1
        self.hist = np.zeros(6)
        self.time = C.c_double(0.0)
        self.barrier = C.c_double(float('nan'))
        self.mode = WINDOW_MODES[window_mode]
        lib().kmco_rates_full(_p(self.occ, C.c_uint8), self.nx, self.ny, _p(self.table, C.c_double),
                              _p(self.P, C.c_double), _p(self.keys, C.c_uint32))

    def init_tags(self):
        self.tags[...] = 0
        self.tags[..., 0] = -1
        ii, jj = np.nonzero(self.occ)
        self.tags[ii, jj, 0] = ii
        self.tags[ii, jj, 1] = jj
        self.time = C.c_double(0.0)

    def run(self, draws, presim, want_trace=True):
        """draws: (n_hops, 2) rho / clock draw.  Returns (n_done, idx, gammas)."""
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        n = draws.shape[0]
        sel = np.zeros(n, dtype=np.int32) if want_trace else None
        gam = np.zeros(n) if want_trace else None
        done = lib().kmco_run(_p(self.occ, C.c_uint8), self.nx, self.ny, _p(self.table, C.c_double),
                              _p(self.P, C.c_double), _p(self.keys, C.c_uint32),
                              _p(self.tags, C.c_int64), n, int(presim), self.mode,
                              _p(draws, C.c_double), _p(sel, C.c_int32), _p(gam, C.c_double),
                              _p(self.hist, C.c_double), C.byref(self.time), C.byref(self.barrier))
        return done, sel, gam

    def diffusion(self, conc, lattice_constant=2.8334 * 10 ** (-8)):
        out = np.zeros(4)
        lib().kmco_diffusion(_p(self.tags, C.c_int64), self.nx, self.ny, self.time.value,
                             lattice_constant, conc, _p(out, C.c_double))
        return out
