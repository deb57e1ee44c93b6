"""ctypes front end to the CPU oracle and (when built) the reference library.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (pyjac_amd) never imports
this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def _p(a):
    return a.ctypes.data_as(_dp)


def build(native: bool = False, quad: bool = False) -> str:
    """Compile oracle/pyjac_oracle.c (gcc) if missing or stale."""
    name = 'libpyjac_oracle_quad.so' if quad else 'libpyjac_oracle_native.so' if native else 'libpyjac_oracle.so'
    out = os.path.join(HERE, '_build', name)
    srcs = [os.path.join(HERE, 'pyjac_oracle.c')] + ([os.path.join(HERE, 'pyjac_oracle_quad.c')] if quad else [])
    if not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(['make', '-s', '-C', HERE, 'quad' if quad else 'native' if native else 'all'])
    return out


class Oracle:
    """Table-driven CPU restatement of pyJac's generated C."""

    def __init__(self, tables, native: bool = False):
        self.lib = ctypes.CDLL(build(native))
        L = self.lib
        L.pjo_create.restype = ctypes.c_void_p
        L.pjo_create.argtypes = [_ip, ctypes.c_long, _dp, ctypes.c_long]
        self.I = np.ascontiguousarray(tables.I, dtype=np.int32)
        self.D = np.ascontiguousarray(tables.D, dtype=np.float64)
        self.h = L.pjo_create(self.I.ctypes.data_as(_ip), self.I.size, _p(self.D), self.D.size)
        if not self.h:
            raise RuntimeError('oracle rejected the mechanism tables')
        self.h = ctypes.c_void_p(self.h)
        self.nsp, self.nrxn = tables.nsp, tables.nrxn
        self.nrev, self.npres = tables.nrev, tables.npres
        vp, d = ctypes.c_void_p, ctypes.c_double
        L.pjo_eval_conc.argtypes = [vp, d, d, _dp, _dp, _dp, _dp, _dp]
        L.pjo_eval_rxn_rates.argtypes = [vp, d, d, _dp, _dp, _dp]
        L.pjo_get_rxn_pres_mod.argtypes = [vp, d, d, _dp, _dp]
        L.pjo_eval_spec_rates.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
        L.pjo_dydt.argtypes = [vp, d, d, _dp, _dp]
        L.pjo_eval_jacob.argtypes = [vp, d, d, _dp, _dp]
        L.pjo_fd_jacob.argtypes = [vp, d, d, _dp, _dp]
        L.pjo_batch_jacob.argtypes = [vp, ctypes.c_long, _dp, _dp, _dp, ctypes.c_int]
        L.pjo_batch_dydt.argtypes = [vp, ctypes.c_long, _dp, _dp, _dp, ctypes.c_int]

    def __del__(self):
        try:
            self.lib.pjo_destroy(self.h)
        except Exception:
            pass

    # ---- per-state, same argument meaning as pyjacob.py_* ----
    def eval_all(self, pres: float, y: np.ndarray):
        """y = [T, Y_0..Y_{NSP-2}].  Returns dict of every intermediate array
        in the order the functional tester calls them (test.py:1299-1327)."""
        n = self.nsp
        y = np.ascontiguousarray(y, dtype=np.float64)
        T = float(y[0])
        conc = np.zeros(n)
        yN, mw, rho = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self.lib.pjo_eval_conc(self.h, T, pres, _p(y[1:].copy()), ctypes.byref(yN),
                               ctypes.byref(mw), ctypes.byref(rho), _p(conc))
        fwd = np.zeros(self.nrxn)
        rev = np.zeros(max(self.nrev, 1))
        pm = np.zeros(max(self.npres, 1))
        self.lib.pjo_eval_rxn_rates(self.h, T, pres, _p(conc), _p(fwd), _p(rev))
        self.lib.pjo_get_rxn_pres_mod(self.h, T, pres, _p(conc), _p(pm))
        sr = np.zeros(n)
        self.lib.pjo_eval_spec_rates(self.h, _p(fwd), _p(rev), _p(pm), _p(sr),
                                     ctypes.cast(sr.ctypes.data + 8 * (n - 1), _dp))
        dy = np.zeros(n)
        self.lib.pjo_dydt(self.h, 0.0, pres, _p(y), _p(dy))
        jac = np.zeros(n * n)
        self.lib.pjo_eval_jacob(self.h, 0.0, pres, _p(y), _p(jac))
        return dict(conc=conc, fwd=fwd, rev=rev, pres_mod=pm, spec_rates=sr, dydt=dy, jac=jac)

    def fd_jacob(self, pres: float, y: np.ndarray) -> np.ndarray:
        """fd_jacob.c:10-113 on the oracle's dydt."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        jac = np.zeros(self.nsp * self.nsp)
        self.lib.pjo_fd_jacob(self.h, 0.0, pres, _p(y), _p(jac))
        return jac

    # ---- batch (state-major) ----
    def batch_jacob(self, pres: np.ndarray, y_aos: np.ndarray, nthreads: int = 0) -> np.ndarray:
        num = pres.shape[0]
        y_aos = np.ascontiguousarray(y_aos, dtype=np.float64)
        pres = np.ascontiguousarray(pres, dtype=np.float64)
        jac = np.empty((num, self.nsp * self.nsp))
        self.lib.pjo_batch_jacob(self.h, num, _p(pres), _p(y_aos), _p(jac), nthreads)
        return jac

    def batch_dydt(self, pres: np.ndarray, y_aos: np.ndarray, nthreads: int = 0) -> np.ndarray:
        num = pres.shape[0]
        y_aos = np.ascontiguousarray(y_aos, dtype=np.float64)
        pres = np.ascontiguousarray(pres, dtype=np.float64)
        dy = np.empty((num, self.nsp))
        self.lib.pjo_batch_dydt(self.h, num, _p(pres), _p(y_aos), _p(dy), nthreads)
        return dy


class OracleQuad:
    """binary128 build of the oracle's text (oracle/pyjac_oracle_quad.c): the reference's formulas and
    constants with 113-bit intermediates, rounded to binary64 once at the end.  The "truth" that the
    rounding error of an evaluation order is measured against (tests/test_conditioning.py)."""

    def __init__(self, tables):
        self.lib = L = ctypes.CDLL(build(quad=True))
        L.pjq_create.restype = ctypes.c_void_p
        L.pjq_create.argtypes = [_ip, ctypes.c_long, _dp, ctypes.c_long]
        I = np.ascontiguousarray(tables.I, dtype=np.int32)
        D = np.ascontiguousarray(tables.D, dtype=np.float64)
        h = L.pjq_create(I.ctypes.data_as(_ip), I.size, _p(D), D.size)
        if not h:
            raise RuntimeError('oracle rejected the mechanism tables')
        self.h = ctypes.c_void_p(h)
        self.nsp, self.nrxn, self.nrev, self.npres = tables.nsp, tables.nrxn, tables.nrev, tables.npres
        vp = ctypes.c_void_p
        L.pjq_eval_all.argtypes = [vp, ctypes.c_double] + [_dp] * 8
        L.pjq_batch_jacob.argtypes = [vp, ctypes.c_long, _dp, _dp, _dp, ctypes.c_int]
        L.pjq_batch_dydt.argtypes = [vp, ctypes.c_long, _dp, _dp, _dp, ctypes.c_int]
        L.pjq_destroy.argtypes = [vp]

    def __del__(self):
        try:
            self.lib.pjq_destroy(self.h)
        except Exception:
            pass

    def set_sum_last_species(self, on: bool):
        self.lib.pjq_set_sum_last_species(int(on))

    def eval_all(self, pres: float, y: np.ndarray):
        n = self.nsp
        y = np.ascontiguousarray(y, dtype=np.float64)
        o = dict(conc=np.zeros(n), fwd=np.zeros(self.nrxn), rev=np.zeros(max(self.nrev, 1)),
                 pres_mod=np.zeros(max(self.npres, 1)), spec_rates=np.zeros(n), dydt=np.zeros(n), jac=np.zeros(n * n))
        self.lib.pjq_eval_all(self.h, float(pres), _p(y), *[_p(o[k]) for k in
                                                          ('conc', 'fwd', 'rev', 'pres_mod', 'spec_rates', 'dydt', 'jac')])
        return o

    def batch_jacob(self, pres: np.ndarray, y_aos: np.ndarray, nthreads: int = 0) -> np.ndarray:
        num = pres.shape[0]
        y_aos = np.ascontiguousarray(y_aos, dtype=np.float64)
        pres = np.ascontiguousarray(pres, dtype=np.float64)
        jac = np.empty((num, self.nsp * self.nsp))
        self.lib.pjq_batch_jacob(self.h, num, _p(pres), _p(y_aos), _p(jac), nthreads)
        return jac

    def batch_dydt(self, pres: np.ndarray, y_aos: np.ndarray, nthreads: int = 0) -> np.ndarray:
        num = pres.shape[0]
        y_aos = np.ascontiguousarray(y_aos, dtype=np.float64)
        pres = np.ascontiguousarray(pres, dtype=np.float64)
        dy = np.empty((num, self.nsp))
        self.lib.pjq_batch_dydt(self.h, num, _p(pres), _p(y_aos), _p(dy), nthreads)
        return dy


class Reference:
    """The reference's own generated C for one mechanism (oracle/_ref/*.so,
    built by oracle/build_ref.py in the container that has /root/reference)."""

    def __init__(self, name: str):
        path = os.path.join(HERE, '_ref', 'libpyjac_ref_%s.so' % name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = ctypes.CDLL(path)
        self.nsp = L.ref_nsp()
        self.nrxn = L.ref_fwd_rates()
        self.nrev = L.ref_rev_rates()
        self.npres = L.ref_pres_mod_rates()
        d = ctypes.c_double
        L.eval_conc.argtypes = [d, d, _dp, _dp, _dp, _dp, _dp]
        L.eval_rxn_rates.argtypes = [d, d, _dp, _dp, _dp]
        if self.npres:
            L.get_rxn_pres_mod.argtypes = [d, d, _dp, _dp]
        L.eval_spec_rates.argtypes = [_dp, _dp, _dp, _dp, _dp]
        L.dydt.argtypes = [d, d, _dp, _dp]
        L.eval_jacob.argtypes = [d, d, _dp, _dp]
        if hasattr(L, 'fd_eval_jacob'):
            L.fd_eval_jacob.argtypes = [d, d, _dp, _dp]
        L.ref_batch_jacob.argtypes = [ctypes.c_long, _dp, _dp, _dp, ctypes.c_int]
        L.ref_batch_dydt.argtypes = [ctypes.c_long, _dp, _dp, _dp, ctypes.c_int]

    @staticmethod
    def available(name: str) -> bool:
        return os.path.exists(os.path.join(HERE, '_ref', 'libpyjac_ref_%s.so' % name))

    def eval_all(self, pres: float, y: np.ndarray):
        n = self.nsp
        L = self.lib
        y = np.ascontiguousarray(y, dtype=np.float64)
        T = float(y[0])
        conc = np.zeros(n)
        yN, mw, rho = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        L.eval_conc(T, pres, _p(y[1:].copy()), ctypes.byref(yN), ctypes.byref(mw),
                    ctypes.byref(rho), _p(conc))
        fwd = np.zeros(self.nrxn)
        rev = np.zeros(max(self.nrev, 1))
        pm = np.zeros(max(self.npres, 1))
        L.eval_rxn_rates(T, pres, _p(conc), _p(fwd), _p(rev))
        if self.npres:
            L.get_rxn_pres_mod(T, pres, _p(conc), _p(pm))
        sr = np.zeros(n)
        L.eval_spec_rates(_p(fwd), _p(rev), _p(pm), _p(sr),
                          ctypes.cast(sr.ctypes.data + 8 * (n - 1), _dp))
        dy = np.zeros(n)
        L.dydt(0.0, pres, _p(y), _p(dy))
        jac = np.zeros(n * n)
        L.eval_jacob(0.0, pres, _p(y), _p(jac))
        return dict(conc=conc, fwd=fwd, rev=rev, pres_mod=pm, spec_rates=sr, dydt=dy, jac=jac)

    def fd_jacob(self, pres: float, y: np.ndarray):
        """The reference's own finite-difference arm (performance_tester/fd_jacob.c)."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        jac = np.zeros(self.nsp * self.nsp)
        self.lib.fd_eval_jacob(0.0, pres, _p(y), _p(jac))
        return jac

    def batch_jacob(self, pres, y_aos, nthreads: int = 0):
        num = pres.shape[0]
        y_aos = np.ascontiguousarray(y_aos, dtype=np.float64)
        pres = np.ascontiguousarray(pres, dtype=np.float64)
        jac = np.empty((num, self.nsp * self.nsp))
        self.lib.ref_batch_jacob(num, _p(pres), _p(y_aos), _p(jac), nthreads)
        return jac

    def batch_dydt(self, pres, y_aos, nthreads: int = 0):
        num = pres.shape[0]
        y_aos = np.ascontiguousarray(y_aos, dtype=np.float64)
        pres = np.ascontiguousarray(pres, dtype=np.float64)
        dy = np.empty((num, self.nsp))
        self.lib.ref_batch_dydt(num, _p(pres), _p(y_aos), _p(dy), nthreads)
        return dy
