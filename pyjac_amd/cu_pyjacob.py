"""Drop-in for pyJac's ``cu_pyjacob`` module (GPU batch path).

py_cuinit / py_cujac / py_cuclean with the argument order and SoA layout of
pyjac/pywrap/pyjacob_cuda_wrapper.pyx:13-34 and pyjac/pywrap/pyjacob.cu:84-188.
"""
from __future__ import annotations

from . import pyjacob as _pj


def use_mechanism(mech, therm=None, last_spec=None):
    return _pj.use_mechanism(mech, therm, last_spec)


def py_cuinit(num: int) -> int:
    return _pj._e().init(num)


def py_cuclean():
    _pj._e().cleanup()


def py_cujac(num, padded, pres, y, conc, fwd_rates, rev_rates, pres_mod, spec_rates, dy, jac):
    _pj._e().run(num, padded, pres, y, conc, fwd_rates, rev_rates, pres_mod, spec_rates, dy, jac)
