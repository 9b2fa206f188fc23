"""kmc_amd -- MI355X-native hex-lattice kinetic Monte Carlo engine.

Drop-in for the per-hop loop of quentinstoyel/KMC (hex_LiCoO2_kmc.py /
Mc_Lattice.py): Python entry points with the reference's names over a C-ABI
library of hand-written HIP kernels (include/kmc_hip.h), loaded with ctypes.
"""
from .engine import KMCEngine, KMCError, ReplicaStopped, WINDOW_MODES, VACANT_KEY, LATTICE_CONSTANT  # noqa: F401
from . import tables  # noqa: F401

__all__ = ['KMCEngine', 'KMCError', 'ReplicaStopped', 'WINDOW_MODES', 'VACANT_KEY', 'LATTICE_CONSTANT', 'tables']
