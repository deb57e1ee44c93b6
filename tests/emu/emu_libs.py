"""Session cache of CPU emulation builds of the state-per-lane kernels (test infrastructure only):
csrc/pj_rblk.hip compiled with g++ through hip_shim.h, one "thread" per workgroup."""
import ctypes
import os

import numpy as np

import build_emu

_dp = ctypes.POINTER(ctypes.c_double)
_cache = {}
_pending = {}       # key -> (ev, so, future of the g++ build): started ahead of the tests that need them (prefetch)
_count = [0]
_pool = None


def _key(name, budget, kw):
    return (name, budget, tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items())))


def _prepare(name, budget, tmp, kw):
    """Evaluator + mechanism header (main thread: the C library is called here) -> (ev, hdr, so)."""
    import pyjac_amd
    from conftest import MECHS, THERMS
    from pyjac_amd import _lib
    d = str(tmp.mktemp('emu')) if hasattr(tmp, 'mktemp') else str(tmp)
    ev = pyjac_amd.Evaluator(MECHS[name], THERMS.get(name), specialize='off')
    n = _count[0]
    _count[0] += 1
    hdr = os.path.join(d, '%s_q%d_%d.h' % (name, budget, n))
    if kw.get('kcf'):
        # equilibrium constants from per-species factor columns: the header carries the rows
        from pyjac_amd.kcfactors import kc_factor_rows
        rows = kc_factor_rows(ev.tables)
        assert rows is not None, 'no factor rows for ' + name
        _lib.check(_lib.lib().pj_mech_set_kc_factors(ev._h, rows.ctypes.data_as(_dp), rows.size))
    _lib.check(_lib.lib().pj_mech_emit_rows_spec(ev._h, hdr.encode(), budget))
    return ev, hdr, os.path.join(d, 'lib%s_q%d_%d.so' % (name, budget, n))


def _load(ev, so):
    from pyjac_amd import _lib
    L = ctypes.CDLL(so)
    L.pj_spec_jacobian.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp,
                                   ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    L.pj_spec_hash.restype = ctypes.c_ulonglong
    assert L.pj_spec_hash() == _lib.lib().pj_mech_spec_hash(ev._h)
    return ev, L


def prefetch(name, budget, tmp, **kw):
    """Start the g++ build of an emulation library in a background thread (only compiler subprocesses run there); the
    test that asks for it later waits for the build instead of starting it.  For the two 53-species libraries, whose
    single translation units take 1 - 2 minutes on one core while the suite's other tests leave cores idle."""
    global _pool
    key = _key(name, budget, kw)
    if key in _cache or key in _pending:
        return
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=2)
    ev, hdr, so = _prepare(name, budget, tmp, kw)
    _pending[key] = (ev, so, _pool.submit(build_emu.build_rblk, hdr, so, **kw))


def rblk_emu_lib(name, budget, tmp, **kw):
    """(Evaluator without attached kernels, ctypes library) of mechanism `name` at accumulator budget
    `budget`; `tmp`: a directory or pytest's tmp_path_factory.  One build per session and option set."""
    key = _key(name, budget, kw)
    if key in _pending:
        ev, so, fut = _pending.pop(key)
        fut.result()
        _cache[key] = _load(ev, so)
    if key not in _cache:
        ev, hdr, so = _prepare(name, budget, tmp, kw)
        build_emu.build_rblk(hdr, so, **kw)
        _cache[key] = _load(ev, so)
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
