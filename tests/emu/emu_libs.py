"""Session cache of CPU emulation builds of the state-per-lane kernels (test infrastructure only):
csrc/pj_rblk.hip compiled with g++ through hip_shim.h, one "thread" per workgroup."""
import ctypes
import os

import numpy as np

import build_emu

_dp = ctypes.POINTER(ctypes.c_double)
_cache = {}


def rblk_emu_lib(name, budget, tmp, **kw):
    """(Evaluator without attached kernels, ctypes library) of mechanism `name` at accumulator budget
    `budget`; `tmp`: a directory or pytest's tmp_path_factory.  One build per session and option set."""
    import pyjac_amd
    from conftest import MECHS, THERMS
    from pyjac_amd import _lib
    key = (name, budget, tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items())))
    if key not in _cache:
        d = str(tmp.mktemp('emu')) if hasattr(tmp, 'mktemp') else str(tmp)
        ev = pyjac_amd.Evaluator(MECHS[name], THERMS.get(name), specialize='off')
        hdr = os.path.join(d, '%s_q%d_%d.h' % (name, budget, len(_cache)))
        if kw.get('kcf'):
            # equilibrium constants from per-species factor columns: the header carries the rows
            from pyjac_amd.kcfactors import kc_factor_rows
            rows = kc_factor_rows(ev.tables)
            assert rows is not None, 'no factor rows for ' + name
            _lib.check(_lib.lib().pj_mech_set_kc_factors(ev._h, rows.ctypes.data_as(_dp), rows.size))
        _lib.check(_lib.lib().pj_mech_emit_rows_spec(ev._h, hdr.encode(), budget))
        so = build_emu.build_rblk(hdr, os.path.join(d, 'lib%s_q%d_%d.so' % (name, budget, len(_cache))), **kw)
        L = ctypes.CDLL(so)
        L.pj_spec_jacobian.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp,
                                       ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
        L.pj_spec_hash.restype = ctypes.c_ulonglong
        assert L.pj_spec_hash() == _lib.lib().pj_mech_spec_hash(ev._h)
        _cache[key] = (ev, L)
    return _cache[key]


def run_jacobian(L, nsp, pres, y_soa, sum_last=0, aos=False):
    """Jacobians of a batch through pj_spec_jacobian of an emulation library -> (n, nsp * nsp)."""
    n = pres.shape[0]
    if aos:
        y = np.ascontiguousarray(y_soa.T)
        jac = np.full((n, nsp * nsp), np.nan)
        rc = L.pj_spec_jacobian(n, pres.ctypes.data_as(_dp), y.ctypes.data_as(_dp), 1, nsp,
                                jac.ctypes.data_as(_dp), 1, nsp * nsp, sum_last, None)
        assert rc == 0
        return jac
    y = np.ascontiguousarray(y_soa)
    jac = np.full((nsp * nsp, n), np.nan)
    rc = L.pj_spec_jacobian(n, pres.ctypes.data_as(_dp), y.ctypes.data_as(_dp), n, 1,
                            jac.ctypes.data_as(_dp), n, 1, sum_last, None)
    assert rc == 0
    return jac.T
