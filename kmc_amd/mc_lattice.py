"""``mc_lattice`` -- the module name the reference's object-oriented scripts import (mc_lattice.py /
object_KMC.py are 34-line stubs of ``class mc_lattice``; Mc_Lattice.py:60-232 is the class itself).

    from kmc_amd.mc_lattice import mc_lattice
    x = mc_lattice([N, N, 1]); x.fill_lattice(0.5); x.KMC_step(); ...

The class lives in kmc_amd/Mc_Lattice.py; this module only gives it the stubs' module name.  Without
``mc_lattice.set_rate_table(...)`` the first use loads hex_dict.p (else normcons_dictionary.p, else
master_dictionary.p) from the working directory, as Mc_Lattice.py:64 does.
"""
from .Mc_Lattice import mc_lattice      # noqa: F401

__all__ = ['mc_lattice']
