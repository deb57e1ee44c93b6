"""CPU checks of the register-resident kernel (csrc/pj_lane.hip): the kernel source is compiled with
g++ through tests/emu/hip_shim.h (one lane per workgroup) and its three modes -- Jacobian blocks,
fused Jacobian-vector product, rate outputs -- are compared with the oracle; plus a small fuzz over
randomly generated mechanisms for both state-per-lane kernel families.  (The AoS transpose path
needs whole wavefronts and is covered by the GPU tests only.)"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))
from conftest import MECHS, THERMS, jac_scaled_err, mixed_err, rate_scales, thresholded_rel_err  # noqa: E402
import build_emu  # noqa: E402
import pyjac_amd  # noqa: E402
from pyjac_amd import _lib, synth  # noqa: E402

_dp = ctypes.POINTER(ctypes.c_double)
_p = lambda a: a.ctypes.data_as(_dp)


def _lane_emu(mech, tmp, tag, therm=None):
    ev = pyjac_amd.Evaluator(mech, therm, specialize='off')
    hdr = os.path.join(tmp, tag + '.h')
    _lib.check(_lib.lib().pj_mech_emit_spec(ev._h, hdr.encode()))
    so = os.path.join(tmp, 'liblane_%s.so' % tag)
    import glob
    import shutil
    cached = build_emu.cache_path(sorted(glob.glob(os.path.join(build_emu.CSRC, '*.h'))) +
                                  [os.path.join(build_emu.CSRC, 'pj_lane.hip'), os.path.join(build_emu.HERE, 'hip_shim.h'),
                                   os.path.abspath(__file__)], open(hdr, 'rb').read())
    if cached and os.path.exists(cached):
        shutil.copyfile(cached, so)
    else:
        subprocess.check_call(['g++', '-O1', '-std=c++17', '-fPIC', '-shared', '-x', 'c++', '-DPJL_HOST_EMU',
                               '-DPJL_BLOCK=1', '-DPJS_HEADER="%s"' % hdr, '-I', build_emu.HERE,
                               '-I', build_emu.CSRC, os.path.join(build_emu.CSRC, 'pj_lane.hip'), '-o', so])
        build_emu.cache_store(cached, so)
    L = ctypes.CDLL(so)
    cl, ci, vp = ctypes.c_long, ctypes.c_int, ctypes.c_void_p
    L.pj_spec_jacobian.argtypes = [cl, _dp, _dp, cl, cl, _dp, cl, cl, ci, vp]
    L.pj_spec_jacvec.argtypes = [cl, _dp, _dp, cl, cl, _dp, cl, cl, _dp, cl, cl, ci, vp]
    L.pj_spec_rates.argtypes = [cl, _dp, _dp, cl, cl] + [_dp] * 6 + [vp]
    return ev, L


def _check_all_modes(ev, L, orc, tab, n=40, seed=5):
    nsp = ev.nsp
    pres, y = synth.dist_b(n, nsp, seed=seed, Tlo=600, Thi=2600)
    y = np.ascontiguousarray(y)
    y_aos = np.ascontiguousarray(y.T)
    ref = orc.batch_jacob(pres, y_aos)
    # mode 0, SoA and (strided) AoS
    jac = np.full((nsp * nsp, n), np.nan)
    assert L.pj_spec_jacobian(n, _p(pres), _p(y), n, 1, _p(jac), n, 1, 0, None) == 0
    assert not np.isnan(jac).any() and jac_scaled_err(jac.T, ref, nsp) <= 1.0
    ja = np.full((n, nsp * nsp), np.nan)
    assert L.pj_spec_jacobian(n, _p(pres), _p(y_aos), 1, nsp, _p(ja), 1, nsp * nsp, 0, None) == 0
    assert np.array_equal(ja, jac.T)
    # mode 1: w = J v
    v = np.ascontiguousarray(np.random.default_rng(seed).standard_normal((nsp, n)))
    w = np.full((nsp, n), np.nan)
    assert L.pj_spec_jacvec(n, _p(pres), _p(y), n, 1, _p(v), n, 1, _p(w), n, 1, 0, None) == 0
    J = ref.reshape(n, nsp, nsp)                                     # [s][col][row]
    wref = np.einsum('scr,cs->sr', J, v)
    scale = np.einsum('scr,cs->sr', np.abs(J), np.abs(v)) + 1e-300
    assert (np.abs(w.T - wref) / scale).max() < 1e-9
    # mode 2: rate outputs
    outs = dict(conc=nsp, fwd=ev.n_fwd, rev=max(ev.n_rev, 1), pres_mod=max(ev.n_pres_mod, 1), spec_rates=nsp, dy=nsp)
    bufs = {k: np.zeros((r, n)) for k, r in outs.items()}
    assert L.pj_spec_rates(n, _p(pres), _p(y), n, 1, *[_p(bufs[k]) for k in outs], None) == 0
    o = [orc.eval_all(float(pres[s]), y_aos[s]) for s in range(n)]
    g = {k: np.array([x[k] for x in o]) for k in ('conc', 'fwd', 'rev', 'pres_mod', 'spec_rates', 'dydt')}
    for k, cols in (('conc', nsp), ('fwd', ev.n_fwd), ('rev', ev.n_rev), ('pres_mod', ev.n_pres_mod)):
        if cols:
            mx, _ = thresholded_rel_err(bufs[k].T[:, :cols], g[k][:, :cols])
            assert mx < 1e-9, (k, mx)
    gross, sdy = rate_scales(tab, pres, y_aos, g['conc'], g['fwd'], g['rev'], g['pres_mod'])
    assert mixed_err(bufs['spec_rates'].T, g['spec_rates'], gross[:, None]) <= 1.0
    assert mixed_err(bufs['dy'].T, g['dydt'], sdy) <= 1.0


# (fe_septherm: species with three different T_mid -- several pre-summed K_c groups per reaction)
@pytest.mark.parametrize('name', ['h2o2_n2', 'h2o2', 'synth_alltypes', 'fe_septherm'])
def test_lane_kernel_modes_vs_oracle(name, tmp_path, tables):
    from oracle.oracle import Oracle
    ev, L = _lane_emu(MECHS[name], str(tmp_path), name, THERMS.get(name))
    _check_all_modes(ev, L, Oracle(tables(name)), tables(name))


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_random_mechanisms_lane_and_rows(seed, tmp_path):
    """Random 9..13-species mechanisms (falloff, third bodies, PLOG, irreversible steps, duplicates):
    the oracle and the two state-per-lane kernel families are independent evaluations of the same
    tables and must agree."""
    from oracle.oracle import Oracle
    from pyjac_amd import synth_mech
    from pyjac_amd.mechanism import read_mech
    from pyjac_amd.tables import build_tables
    rng = np.random.default_rng(seed)
    nsp, nrxn = int(rng.integers(9, 14)), int(rng.integers(24, 48))
    txt = synth_mech.generate(nsp, nrxn, n_falloff=int(rng.integers(2, 6)), n_thd=int(rng.integers(2, 6)),
                              n_plog=int(rng.integers(0, 4)), n_irrev=int(rng.integers(1, 5)),
                              n_dup_pairs=int(rng.integers(0, 3)), seed=seed, title='fuzz %d' % seed)
    mech = os.path.join(str(tmp_path), 'fuzz%d.inp' % seed)
    open(mech, 'w').write(txt)
    tab = build_tables(read_mech(mech))
    orc = Oracle(tab)
    ev, L = _lane_emu(mech, str(tmp_path), 'fuzz%d' % seed)
    _check_all_modes(ev, L, orc, tab, n=24, seed=seed)
    # the row-block kernels on the same mechanism, fine partition
    hdr = os.path.join(str(tmp_path), 'rows%d.h' % seed)
    _lib.check(_lib.lib().pj_mech_emit_rows_spec(ev._h, hdr.encode(), 14))
    so = build_emu.build_rblk(hdr, os.path.join(str(tmp_path), 'librblk%d.so' % seed), blocks_per_part=3, rates_per_part=9)
    R = ctypes.CDLL(so)
    R.pj_spec_jacobian.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp, ctypes.c_long,
                                   ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    n = 24
    pres, y = synth.dist_b(n, ev.nsp, seed=seed, Tlo=600, Thi=2600)
    y = np.ascontiguousarray(y)
    jac = np.full((ev.nsp ** 2, n), np.nan)
    assert R.pj_spec_jacobian(n, _p(pres), _p(y), n, 1, _p(jac), n, 1, 0, None) == 0
    assert jac_scaled_err(jac.T, orc.batch_jacob(pres, np.ascontiguousarray(y.T)), ev.nsp) <= 1.0
