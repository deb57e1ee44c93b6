"""Drop-in for pyJac's generated ``pyjacob`` module (per-state C path).

Same function names, argument order and side effects as
pyjac/pywrap/pyjacob_wrapper.pyx:18-55; the arithmetic runs on the GPU through
libpyjac_hip.so.  pyJac compiles the mechanism into the module; here call
``use_mechanism(path)`` once (or set PYJAC_AMD_MECH) before the py_* functions.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib
from ._lib import check, dptr
from .evaluator import Evaluator

_ev = None


def use_mechanism(mech, therm=None, last_spec=None) -> Evaluator:
    global _ev
    _ev = mech if isinstance(mech, Evaluator) else Evaluator(mech, therm, last_spec)
    return _ev


def _e() -> Evaluator:
    global _ev
    if _ev is None:
        path = os.environ.get('PYJAC_AMD_MECH')
        if not path:
            raise _lib.PyjacError('no mechanism loaded: call pyjacob.use_mechanism(path)')
        use_mechanism(path)
    return _ev


def _f64(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
        raise TypeError('expected a contiguous float64 numpy array')
    return a


def py_dydt(t, pres, y, dy):
    check(_lib.lib().pj_dydt(_e()._h, t, pres, dptr(_f64(y)), dptr(_f64(dy))))


def py_eval_jacobian(t, pres, y, jac):
    check(_lib.lib().pj_eval_jacob(_e()._h, t, pres, dptr(_f64(y)), dptr(_f64(jac))))


def py_eval_rxn_rates(T, pres, C, fwd_rxn_rates, rev_rxn_rates):
    check(_lib.lib().pj_eval_rxn_rates(_e()._h, T, pres, dptr(_f64(C)), dptr(_f64(fwd_rxn_rates)),
                                       dptr(_f64(rev_rxn_rates))))


def py_eval_spec_rates(fwd_rxn_rates, rev_rxn_rates, pres_mod, sp_rates):
    # the wrapper aliases dy_N to the last element of sp_rates (pyjacob_wrapper.pyx:41)
    sp = _f64(sp_rates)
    last = ctypes.cast(sp.ctypes.data + 8 * (sp.shape[0] - 1), ctypes.POINTER(ctypes.c_double))
    check(_lib.lib().pj_eval_spec_rates(_e()._h, dptr(_f64(fwd_rxn_rates)), dptr(_f64(rev_rxn_rates)),
                                        dptr(_f64(pres_mod)), dptr(sp), last))


def py_get_rxn_pres_mod(T, pres, C, pres_mod):
    check(_lib.lib().pj_get_rxn_pres_mod(_e()._h, T, pres, dptr(_f64(C)), dptr(_f64(pres_mod))))


def py_eval_conc(T, pres, mass_frac, mw_avg, rho, conc):
    # mw_avg / rho are passed by value and discarded; y_N is written into
    # mass_frac[-1] (pyjacob_wrapper.pyx:49-55)
    mf = _f64(mass_frac)
    yN = ctypes.cast(mf.ctypes.data + 8 * (mf.shape[0] - 1), ctypes.POINTER(ctypes.c_double))
    a, b = ctypes.c_double(mw_avg), ctypes.c_double(rho)
    # the C function reads NSP-1 mass fractions; copy so the y_N write cannot race the read
    src = mf[:-1].copy() if mf.shape[0] >= _e().nsp else mf.copy()
    check(_lib.lib().pj_eval_conc(_e()._h, T, pres, dptr(src), yN, ctypes.byref(a), ctypes.byref(b),
                                  dptr(_f64(conc))))
