"""CPU checks of the state-per-lane row-block kernels (csrc/pj_rows.hip): the kernel source is
compiled with g++ through tests/emu/hip_shim.h (one thread per workgroup) and compared with the
oracle and the committed golden vectors.  Covers the rate-kernel / row-kernel split, the scratch
numbering, the row-block partition at several budgets and the K_c class handling across
rate-kernel ranges -- everything but the GPU's memory system."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))
from conftest import MECHS, jac_scaled_err, mixed_err, rate_scales, thresholded_rel_err  # noqa: E402
import build_rows_emu  # noqa: E402
import pyjac_amd  # noqa: E402
from pyjac_amd import _lib, synth  # noqa: E402

_dp = ctypes.POINTER(ctypes.c_double)


def _emu_lib(name, budget, tmp, **kw):
    ev = pyjac_amd.Evaluator(MECHS[name], specialize='off')
    hdr = os.path.join(tmp, '%s_%d.h' % (name, budget))
    _lib.check(_lib.lib().pj_mech_emit_rows_spec(ev._h, hdr.encode(), budget))
    so = build_rows_emu.build(hdr, os.path.join(tmp, 'lib%s_%d_%d.so' % (name, budget, len(kw.get('defines', ())))), **kw)
    L = ctypes.CDLL(so)
    L.pj_spec_jacobian.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp,
                                   ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    L.pj_spec_hash.restype = ctypes.c_ulonglong
    assert L.pj_spec_hash() == _lib.lib().pj_mech_spec_hash(ev._h)
    return ev, L


def _run(L, nsp, pres, y_soa, sum_last=0, aos=False):
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


@pytest.mark.parametrize('name,budget,kw', [
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, c_lds=1)),
    ('synth_alltypes', 200, dict(blocks_per_part=1, rates_per_part=1000)),
    ('h2o2', 24, dict(blocks_per_part=3, rates_per_part=10)),
    ('h2o2_n2', 12, dict(blocks_per_part=100, rates_per_part=5)),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40)),
    # row kernels rebuild c*k_r from c*k_f and K_c(T) instead of reading it from the scratch array
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, defines=('-DPJR_RECOMPUTE_KR=1',))),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40, defines=('-DPJR_RECOMPUTE_KR=1',))),
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7,
                                defines=('-DPJR_RECOMPUTE_KR=1', '-DPJR_RECOMPUTE_KF=1', '-DPJR_DUMMY=1'))),
    ('synth_srichb', 16, dict(blocks_per_part=2, rates_per_part=5)),
])
def test_rows_kernels_vs_oracle(name, budget, kw, tmp_path, tables):
    from oracle.oracle import Oracle
    ev, L = _emu_lib(name, budget, str(tmp_path), **kw)
    orc = Oracle(tables(name))
    n = 300                               # crosses a 256-state scratch tile
    pres, y = synth.dist_b(n, ev.nsp)
    ref = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
    for aos in (False, True):
        jac = _run(L, ev.nsp, pres, y, aos=aos)
        assert not np.isnan(jac).any()    # every entry written
        assert jac_scaled_err(jac, ref, ev.nsp) <= 1.0
    # the J_nplusone quirk switch (create_jacobian.py:2786-2818)
    orc.lib.pjo_set_sum_last_species(1)
    try:
        ref1 = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
    finally:
        orc.lib.pjo_set_sum_last_species(0)
    assert jac_scaled_err(_run(L, ev.nsp, pres, y, sum_last=1), ref1, ev.nsp) <= 1.0


@pytest.mark.parametrize('name,budget,kw', [
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, c_lds=1)),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40)),
    ('synth_srichb', 16, dict(blocks_per_part=2, rates_per_part=5)),
])
def test_rows_rate_outputs_vs_oracle(name, budget, kw, tmp_path, tables):
    """k_rates<true> + k_dy (pj_spec_rates of the row-block library): conc, fwd, rev, pres_mod,
    spec_rates summed over several rate kernels, dydt -- with every array requested and with dydt only
    (omega_k then lives in the library's scratch array)."""
    from oracle.oracle import Oracle
    ev, L = _emu_lib(name, budget, str(tmp_path), **kw)
    L.pj_spec_rates.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long] + [_dp] * 6 + [ctypes.c_void_p]
    tab = tables(name)
    orc = Oracle(tab)
    n, nsp = 300, ev.nsp
    pres, y = synth.dist_b(n, nsp, seed=9, Tlo=600, Thi=2600)
    y = np.ascontiguousarray(y)
    y_aos = np.ascontiguousarray(y.T)
    o = [orc.eval_all(float(pres[s]), y_aos[s]) for s in range(n)]
    g = {k: np.array([x[k] for x in o]) for k in ('conc', 'fwd', 'rev', 'pres_mod', 'spec_rates', 'dydt')}
    rows = dict(conc=nsp, fwd=ev.n_fwd, rev=max(ev.n_rev, 1), pres_mod=max(ev.n_pres_mod, 1), spec_rates=nsp, dy=nsp)
    bufs = {k: np.full((r, n), np.nan) for k, r in rows.items()}
    P = lambda a: a.ctypes.data_as(_dp)
    assert L.pj_spec_rates(n, P(pres), P(y), n, 1, *[P(bufs[k]) for k in rows], None) == 0
    for k, cols in (('conc', nsp), ('fwd', ev.n_fwd), ('rev', ev.n_rev), ('pres_mod', ev.n_pres_mod)):
        if cols:
            mx, _ = thresholded_rel_err(bufs[k].T[:, :cols], g[k][:, :cols])
            assert mx < 1e-9, (k, mx)
    gross, sdy = rate_scales(tab, pres, y_aos, g['conc'], g['fwd'], g['rev'], g['pres_mod'])
    assert mixed_err(bufs['spec_rates'].T, g['spec_rates'], gross[:, None]) <= 1.0
    assert mixed_err(bufs['dy'].T, g['dydt'], sdy) <= 1.0
    dy2 = np.full((nsp, n), np.nan)
    assert L.pj_spec_rates(n, P(pres), P(y), n, 1, None, None, None, None, None, P(dy2), None) == 0
    assert np.array_equal(dy2, bufs['dy'])


def test_rows_kernels_chunked_double_buffered(tmp_path, tables, monkeypatch):
    """Batches larger than the scratch chunk alternate between two scratch arrays (on the GPU:
    two streams, rate kernels of chunk c+1 next to the row kernels of chunk c)."""
    from oracle.oracle import Oracle
    monkeypatch.setenv('PJ_ROWS_CHUNK', '256')
    ev, L = _emu_lib('h2o2', 24, str(tmp_path), blocks_per_part=2, rates_per_part=10)
    n = 256 * 3 + 17
    pres, y = synth.dist_b(n, ev.nsp, seed=5)
    ref = Oracle(tables('h2o2')).batch_jacob(pres, np.ascontiguousarray(y.T))
    jac = _run(L, ev.nsp, pres, y)
    assert not np.isnan(jac).any() and jac_scaled_err(jac, ref, ev.nsp) <= 1.0


def test_rows_kernels_vs_reference_golden(tmp_path, golden):
    """53-species mechanism at the shipping budget against vectors from pyJac's generated C."""
    ev, L = _emu_lib('gri30_shaped', 64, str(tmp_path), blocks_per_part=8)
    g = golden('gri30_shaped')
    pres, y = g['pres'], np.ascontiguousarray(g['y'].T)
    jac = _run(L, ev.nsp, pres, y)
    assert jac_scaled_err(jac, g['jac'], ev.nsp) <= 1.0
    fro = np.linalg.norm(jac - g['jac']) / np.linalg.norm(g['jac'])
    assert fro < 1e-9


def _fused_emu_lib(name, budget, tmp):
    """csrc/pj_rows.hip as the single fused kernel (PJR_PART=3) with one lane and one wavefront per
    workgroup: the arms of the four wavefronts run one after the other."""
    import subprocess
    ev = pyjac_amd.Evaluator(MECHS[name], specialize='off')
    hdr = os.path.join(tmp, '%s_f%d.h' % (name, budget))
    _lib.check(_lib.lib().pj_mech_emit_rows_spec(ev._h, hdr.encode(), budget))
    so = os.path.join(tmp, 'lib%s_fused.so' % name)
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-fPIC', '-shared', '-x', 'c++', '-DPJR_HOST_EMU',
                           '-DPJR_PART=3', '-DPJR_WLANES=1', '-DPJR_NW=1', '-DPJS_HEADER="%s"' % hdr,
                           '-I', build_rows_emu.HERE, '-I', build_rows_emu.CSRC,
                           os.path.join(build_rows_emu.CSRC, 'pj_rows.hip'), '-o', so])
    L = ctypes.CDLL(so)
    L.pj_spec_jacobian.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp,
                                   ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    return ev, L


@pytest.mark.parametrize('name,budget', [('synth_alltypes', 16), ('h2o2', 24)])
def test_fused_kernel_vs_oracle(name, budget, tmp_path, tables):
    from oracle.oracle import Oracle
    ev, L = _fused_emu_lib(name, budget, str(tmp_path))
    orc = Oracle(tables(name))
    n = 37
    pres, y = synth.dist_b(n, ev.nsp)
    for sum_last in (0, 1):
        orc.lib.pjo_set_sum_last_species(sum_last)
        try:
            ref = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
        finally:
            orc.lib.pjo_set_sum_last_species(0)
        for aos in (False, True):
            jac = _run(L, ev.nsp, pres, y, sum_last=sum_last, aos=aos)
            assert not np.isnan(jac).any()
            assert jac_scaled_err(jac, ref, ev.nsp) <= 1.0


def _rblk_emu_lib(name, budget, tmp, **kw):
    """csrc/pj_rblk.hip (+ the rate-output kernels of pj_rows.hip) compiled for the host."""
    ev = pyjac_amd.Evaluator(MECHS[name], specialize='off')
    hdr = os.path.join(tmp, '%s_q%d.h' % (name, budget))
    _lib.check(_lib.lib().pj_mech_emit_rows_spec(ev._h, hdr.encode(), budget))
    so = build_rows_emu.build_rblk(hdr, os.path.join(tmp, 'lib%s_q%d.so' % (name, budget)), **kw)
    L = ctypes.CDLL(so)
    L.pj_spec_jacobian.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp,
                                   ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    L.pj_spec_hash.restype = ctypes.c_ulonglong
    assert L.pj_spec_hash() == _lib.lib().pj_mech_spec_hash(ev._h)
    return ev, L


@pytest.mark.parametrize('name,budget,kw', [
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, c_lds=1)),
    ('synth_alltypes', 200, dict(blocks_per_part=1, rates_per_part=1000)),
    ('h2o2', 24, dict(blocks_per_part=3, rates_per_part=10)),
    ('h2o2_n2', 12, dict(blocks_per_part=100, rates_per_part=5)),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40)),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40, defines=('-DPJQ_DEPTH=1',))),
    # SRI falloff (3 / 5 parameters, LOW / HIGH, collider) and Chebyshev reactions: evaluated by the pre-pass
    ('synth_srichb', 16, dict(blocks_per_part=2, rates_per_part=5)),
])
def test_rblk_kernels_vs_oracle(name, budget, kw, tmp_path, tables):
    """Row blocks that rebuild their rates (Arrhenius, K_c, third body, theta per visit), the falloff /
    PLOG pre-pass with its register ring, the d/dT column finished per block, energy-row partials handed
    from kernel to kernel: against the oracle, both layouts, with and without the J_nplusone quirk."""
    from oracle.oracle import Oracle
    ev, L = _rblk_emu_lib(name, budget, str(tmp_path), **kw)
    orc = Oracle(tables(name))
    n = 300                               # crosses a 256-state hand-over tile
    pres, y = synth.dist_b(n, ev.nsp)
    ref = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
    for aos in (False, True):
        jac = _run(L, ev.nsp, pres, y, aos=aos)
        assert not np.isnan(jac).any()    # every entry written
        assert jac_scaled_err(jac, ref, ev.nsp) <= 1.0
    orc.lib.pjo_set_sum_last_species(1)
    try:
        ref1 = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
    finally:
        orc.lib.pjo_set_sum_last_species(0)
    assert jac_scaled_err(_run(L, ev.nsp, pres, y, sum_last=1), ref1, ev.nsp) <= 1.0


def test_rblk_kernels_chunked(tmp_path, tables, monkeypatch):
    """Batches larger than the chunk run chunk by chunk through the hand-over arrays of the internal streams."""
    from oracle.oracle import Oracle
    monkeypatch.setenv('PJ_RBLK_CHUNK', '256')
    monkeypatch.setenv('PJ_RBLK_STREAMS', '2')
    ev, L = _rblk_emu_lib('synth_alltypes', 16, str(tmp_path), blocks_per_part=2, rates_per_part=10)
    n = 256 * 3 + 17
    pres, y = synth.dist_b(n, ev.nsp, seed=5)
    ref = Oracle(tables('synth_alltypes')).batch_jacob(pres, np.ascontiguousarray(y.T))
    jac = _run(L, ev.nsp, pres, y)
    assert not np.isnan(jac).any() and jac_scaled_err(jac, ref, ev.nsp) <= 1.0


def test_rblk_kernels_vs_reference_golden(tmp_path, golden):
    """53-species mechanism at the shipping budget against vectors from pyJac's generated C; the library's
    rate outputs (k_rates<true> + k_dy of pj_rows.hip, linked in) against the same vectors."""
    ev, L = _rblk_emu_lib('gri30_shaped', 56, str(tmp_path), blocks_per_part=13)
    g = golden('gri30_shaped')
    pres, y = g['pres'], np.ascontiguousarray(g['y'].T)
    jac = _run(L, ev.nsp, pres, y)
    assert jac_scaled_err(jac, g['jac'], ev.nsp) <= 1.0
    fro = np.linalg.norm(jac - g['jac']) / np.linalg.norm(g['jac'])
    assert fro < 1e-9
    mx, _ = thresholded_rel_err(jac, g['jac'])
    assert mx < 1e-4
    L.pj_spec_rates.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long] + [_dp] * 6 + [ctypes.c_void_p]
    n, nsp = pres.size, ev.nsp
    rows = dict(conc=nsp, fwd=ev.n_fwd, rev=max(ev.n_rev, 1), pres_mod=max(ev.n_pres_mod, 1), spec_rates=nsp, dy=nsp)
    bufs = {k: np.full((r, n), np.nan) for k, r in rows.items()}
    P = lambda a: a.ctypes.data_as(_dp)
    assert L.pj_spec_rates(n, P(pres), P(y), n, 1, *[P(bufs[k]) for k in rows], None) == 0
    for k, cols in (('conc', nsp), ('fwd', ev.n_fwd), ('rev', ev.n_rev), ('pres_mod', ev.n_pres_mod)):
        mx, _ = thresholded_rel_err(bufs[k].T[:, :cols], g[k][:, :cols])
        assert mx < 1e-9, (k, mx)


@pytest.mark.parametrize('name,budget,kw', [
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7)),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40)),
])
def test_rblk_fused_jacobian_vector_product(name, budget, kw, tmp_path, tables):
    """N2 for the row-block family: w = J v per state with the Jacobian consumed in registers
    (pj_spec_jacvec, PJQ_JV kernels) against the oracle's Jacobian times the same vectors, both layouts."""
    from oracle.oracle import Oracle
    ev, L = _rblk_emu_lib(name, budget, str(tmp_path), **kw)
    L.pj_spec_jacvec.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp, ctypes.c_long, ctypes.c_long,
                                 _dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    n, nsp = 300, ev.nsp
    pres, y = synth.dist_b(n, nsp, seed=12, Tlo=700, Thi=2500)
    v = np.random.default_rng(3).standard_normal((nsp, n))
    v[0] *= 100.0
    J = Oracle(tables(name)).batch_jacob(pres, np.ascontiguousarray(y.T)).reshape(n, nsp, nsp)   # [s][col][row]
    ref = np.einsum('scr,cs->sr', J, v)
    scale = np.einsum('scr,cs->sr', np.abs(J), np.abs(v)) + 1e-300
    P = lambda a: a.ctypes.data_as(_dp)
    ys, vs, ws = np.ascontiguousarray(y), np.ascontiguousarray(v), np.full((nsp, n), np.nan)
    assert L.pj_spec_jacvec(n, P(pres), P(ys), n, 1, P(vs), n, 1, P(ws), n, 1, 0, None) == 0
    assert (np.abs(ws.T - ref) / scale).max() < 1e-9
    ya, va, wa = np.ascontiguousarray(y.T), np.ascontiguousarray(v.T), np.full((n, nsp), np.nan)
    assert L.pj_spec_jacvec(n, P(pres), P(ya), 1, nsp, P(va), 1, nsp, P(wa), 1, nsp, 0, None) == 0
    assert (np.abs(wa - ref) / scale).max() < 1e-9
