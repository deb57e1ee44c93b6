"""ctypes binding of libpyjac_hip.so (C ABI: include/pyjac_amd.h).

The HIP library is the only evaluation path.  If it has not been built, or no
GPU is usable, calls raise -- there is no CPU fallback in the product.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PYJAC_AMD_LIB: another build of the same library, for experiments)
LIB_PATH = os.environ.get('PYJAC_AMD_LIB') or os.path.join(_HERE, 'libpyjac_hip.so')

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p

LAYOUT_SOA, LAYOUT_AOS = 0, 1

# every symbol include/pyjac_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'pj_last_error': (ctypes.c_char_p, []),
    'pj_version': (ctypes.c_char_p, []),
    'pj_mech_create': (ctypes.c_int, [_ip, ctypes.c_long, _dp, ctypes.c_long, ctypes.POINTER(_vp)]),
    'pj_mech_load': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_vp)]),
    'pj_mech_destroy': (None, [_vp]),
    'pj_mech_nsp': (ctypes.c_int, [_vp]),
    'pj_mech_fwd_rates': (ctypes.c_int, [_vp]),
    'pj_mech_rev_rates': (ctypes.c_int, [_vp]),
    'pj_mech_pres_mod_rates': (ctypes.c_int, [_vp]),
    'pj_mech_set_sum_last_species': (ctypes.c_int, [_vp, ctypes.c_int]),
    'pj_mech_set_check_inputs': (ctypes.c_int, [_vp, ctypes.c_int]),
    'pj_mech_set_generic_kernel': (ctypes.c_int, [_vp, ctypes.c_int]),
    'pj_eval_state': (ctypes.c_int, [_vp, ctypes.c_double] + [_dp] * 8),
    'pj_mech_set_launch': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    'pj_mech_get_launch': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'pj_mech_spec_hash': (ctypes.c_ulonglong, [_vp]),
    'pj_mech_emit_spec': (ctypes.c_int, [_vp, ctypes.c_char_p]),
    'pj_mech_emit_rows_spec': (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int]),
    'pj_mech_emit_rblk_spec': (ctypes.c_int, [_vp, ctypes.c_char_p] + [ctypes.c_int] * 8 + [ctypes.c_double] * 2 +
                               [ctypes.POINTER(ctypes.c_int)]),
    'pj_mech_set_kc_factors': (ctypes.c_int, [_vp, _dp, ctypes.c_long]),
    'pj_mech_attach_spec': (ctypes.c_int, [_vp, ctypes.c_char_p]),
    'pj_mech_has_spec': (ctypes.c_int, [_vp]),
    'pj_mech_use_spec': (ctypes.c_int, [_vp, ctypes.c_int]),
    'pj_lu_factor_dev': (ctypes.c_int, [ctypes.c_int, ctypes.c_long, _vp, ctypes.c_int, ctypes.c_double, _vp, _vp, _vp]),
    'pj_lu_solve_dev': (ctypes.c_int, [ctypes.c_int, ctypes.c_long, _vp, _vp, _vp, _vp, ctypes.c_int, _vp]),
    'pj_newton_solve_dev': (ctypes.c_int, [ctypes.c_int, ctypes.c_long, _vp, ctypes.c_int, ctypes.c_double, _vp, _vp,
                                           ctypes.c_int, _vp, _vp, _vp]),
    'pj_mech_set_spec_launch': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int]),
    'pj_eval_jacobian_dev': (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp, ctypes.c_int, _vp,
                                            ctypes.c_int, _vp]),
    'pj_eval_jacobian_vec_dev': (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp, ctypes.c_int, _vp, _vp,
                                                ctypes.c_int, _vp]),
    'pj_eval_rates_dev': (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp, ctypes.c_int,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pj_eval_fd_jacobian_dev': (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp, _vp, ctypes.c_int, _vp]),
    'pj_time_jacobian_dev': (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp, ctypes.c_int, _vp,
                                            ctypes.c_int, _vp, ctypes.c_int, _dp]),
    'pj_init': (ctypes.c_int, [_vp, ctypes.c_int]),
    'pj_run': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int] + [_dp] * 9),
    'pj_cleanup': (ctypes.c_int, [_vp]),
    'pj_dydt': (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, _dp, _dp]),
    'pj_eval_jacob': (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, _dp, _dp]),
    'pj_eval_rxn_rates': (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, _dp, _dp, _dp]),
    'pj_eval_spec_rates': (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp]),
    'pj_get_rxn_pres_mod': (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, _dp, _dp]),
    'pj_eval_conc': (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, _dp, _dp, _dp, _dp, _dp]),
}

_lib = None


class PyjacError(RuntimeError):
    pass


def lib():
    """Load libpyjac_hip.so (fails loudly when the HIP extension is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PyjacError(
                'HIP extension %s is not built; run `python -c "import __graft_entry__ as g; '
                'g.build()"` (hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
        # One HIP runtime per process: the device buffers this package hands to the library are torch's, and torch
        # brings its own libamdhip64.  Loaded first, it also satisfies libpyjac_hip's dependency; loaded second (the
        # library before torch, as `build(); smoke()` in one process does), /opt/rocm's copy would be a second runtime
        # in the process and its hipGetDeviceCount fails once torch's has the device ("no HIP device available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass    # a torch-free consumer of the C ABI: the system runtime is then the only one
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc < 0:
        raise PyjacError('libpyjac_hip: %s (code %d)' % (lib().pj_last_error().decode(), rc))
    return rc


def dptr(a: np.ndarray):
    return a.ctypes.data_as(_dp)
