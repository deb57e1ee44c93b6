"""CPU (kernel sources compiled with g++ through tests/emu): columns whose species weighs what the LAST species weighs.

Entry (k, j) of the species block is (W_k / W_j)(P_k - w_j Q_k + S_kj), w_j = W_j / W_N, Q_k = P_k + QN_k.  For an isomer of
the last species (pyJac leaves the last listed species last when the file has no N2 / AR / HE,
pyjac/core/create_jacobian.py:3521-3542) w_j = 1 and the dense parts cancel EXACTLY; the reference forms a_i (1 - W_j / W_N)
per reaction (create_jacobian.py:341-489) and gets the remainder -- 1e-16 of the row scale -- right to 1e-9.  A kernel that
accumulates P_k and Q_k separately and subtracts at the end carries P_k's rounding error there: percents of the entry (found by
the random-mechanism sweep of round 6: sweep_r2, HCNO next to HOCN, 3 % off on entries of the HCNO column).  All four kernel
families (pj_rblk, pj_lane, k_tab, k_eval) therefore accumulate QN_k = sum nu gN and form (1 - w_j) P_k - w_j QN_k; this file
holds them to the north star's ENTRY-WISE rtol 1e-6 against the binary128 evaluation of the reference's formulas on
  * iso_n012  12 species, CH2 next to CH2(S) (last)         -- pj_lane, k_tab, k_eval, pj_rblk
  * sweep_r2  45 species, HCNO next to HOCN (last)          -- pj_rblk (its shipped geometry), k_tab, k_eval
(tests/golden/make_sweep_mechs.py)."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))
from conftest import GOLDEN, MECHS, rel_err_entries  # noqa: E402
from pyjac_amd import synth  # noqa: E402
from pyjac_amd.mechanism import read_mech  # noqa: E402
from pyjac_amd.tables import build_tables  # noqa: E402

ISO = os.path.join(GOLDEN, 'sweep', 'iso_n012.inp')
RTOL = 1e-6
_dp = ctypes.POINTER(ctypes.c_double)


def _case(name):
    from oracle.oracle import Oracle, OracleQuad
    mech = ISO if name == 'iso_n012' else MECHS[name]
    tab = build_tables(read_mech(mech))
    n = 48
    pres, y = synth.dist_b(n, tab.nsp, seed=663, Tlo=500, Thi=2600)
    y_aos = np.ascontiguousarray(y.T)
    truth = OracleQuad(tab).batch_jacob(pres, y_aos)
    ref = Oracle(tab).batch_jacob(pres, y_aos)
    # the case is what it claims to be: a column with w_j = 1 exactly, and the reference's own arithmetic is fine on it
    assert float(rel_err_entries(ref, truth).max()) < RTOL
    return mech, tab, pres, y, truth


def _check(label, jac, truth):
    r = float(rel_err_entries(jac, truth).max())
    print('%s: max entry-wise relative error vs binary128 %.3g' % (label, r))
    assert np.isfinite(jac).all() and r < RTOL, (label, r)


def test_the_isomer_mechanisms_have_a_unit_weight_ratio():
    for mech in (ISO, MECHS['sweep_r2']):
        m = read_mech(mech)
        mw = [s.mw for s in m.specs]
        assert any(abs(mw[j] / mw[-1] - 1.0) == 0.0 for j in range(len(mw) - 1)), mech
        assert m.species_names()[-1] not in ('N2', 'AR', 'HE')


@pytest.mark.parametrize('name', ['iso_n012', 'sweep_r2'])
def test_row_block_kernels_on_isomer_columns(name, tmp_path):
    from emu_libs import run_jacobian
    import build_emu
    import pyjac_amd
    from pyjac_amd import _lib
    from pyjac_amd.kcfactors import kc_factor_rows
    mech, tab, pres, y, truth = _case(name)
    ev = pyjac_amd.Evaluator(mech, specialize='off')
    rows = kc_factor_rows(ev.tables)
    assert rows is not None
    _lib.check(_lib.lib().pj_mech_set_kc_factors(ev._h, rows.ctypes.data_as(_dp), rows.size))
    hdr = str(tmp_path / (name + '.h'))
    _lib.check(_lib.lib().pj_mech_emit_rows_spec(ev._h, hdr.encode(), 56))
    so = build_emu.build_rblk(hdr, str(tmp_path / ('lib%s.so' % name)), kcf=1, halves=4, single=1, c_lds=0, only_rows=True)
    L = ctypes.CDLL(so)
    L.pj_spec_jacobian.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp, ctypes.c_long, ctypes.c_long,
                                   ctypes.c_int, ctypes.c_void_p]
    _check('pj_rblk ' + name, run_jacobian(L, ev.nsp, pres, y), truth)
    _check('pj_rblk (per-state layout) ' + name, run_jacobian(L, ev.nsp, pres, y, aos=True), truth)


def test_register_resident_kernel_on_isomer_columns(tmp_path):
    from test_lane_emu import _lane_emu, _p
    mech, tab, pres, y, truth = _case('iso_n012')
    ev, L = _lane_emu(mech, str(tmp_path), 'iso_n012')
    assert ev.spec_kind() == 'lane'
    n, nsp = pres.size, ev.nsp
    jac = np.full((nsp * nsp, n), np.nan)
    yc = np.ascontiguousarray(y)
    assert L.pj_spec_jacobian(n, _p(pres), _p(yc), n, 1, _p(jac), n, 1, 0, None) == 0
    _check('pj_lane iso_n012', jac.T, truth)


@pytest.mark.parametrize('name', ['iso_n012', 'sweep_r2'])
def test_table_driven_kernels_on_isomer_columns(name):
    from test_host_logic import _emu, _run_emu
    mech, tab, pres, y, truth = _case(name)
    n = pres.size
    # k_eval (a workgroup per state tile)
    out = _run_emu(tab, pres, y, 16 if tab.nsp <= 16 else 1, 256 if tab.nsp <= 16 else 64, tab.nsp > 16)
    _check('k_eval ' + name, out['jac'].reshape(n, -1) if tab.nsp > 16 else out['jac'].reshape(-1, n).T, truth)
    # k_tab + k_tab_fin (state per lane, accumulators in LDS)
    I = np.ascontiguousarray(tab.I, dtype=np.int32)
    D = np.ascontiguousarray(tab.D)
    jac = np.full(tab.nsp * tab.nsp * n, np.nan)
    info = (ctypes.c_int * 6)()
    P = lambda a: a.ctypes.data_as(_dp)
    rc = _emu().emu_tab_run(I.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_long(I.size), P(D), ctypes.c_long(D.size),
                            ctypes.c_long(n), P(pres), P(np.ascontiguousarray(y)), P(jac), 0, 0, ctypes.c_long(156 * 1024), info)
    assert rc == 0
    _check('k_tab ' + name, jac.reshape(-1, n).T, truth)
