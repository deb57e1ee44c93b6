"""CPU checks of the state-per-lane row-block kernels (csrc/pj_rblk.hip): the kernel source is compiled with
g++ through tests/emu/hip_shim.h (one thread per workgroup) and compared with the oracle and the committed
golden vectors.  Covers the pre-pass / row-kernel split, the hand-over numbering, the row-block partition at
several budgets, the rate-output kernels (one and several reaction ranges) and the fused w = J v -- everything
but the GPU's memory system."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))
from conftest import jac_scaled_err, mixed_err, rate_scales, thresholded_rel_err  # noqa: E402
from emu_libs import rblk_emu_lib, run_jacobian as _run  # noqa: E402
from pyjac_amd import synth  # noqa: E402

_dp = ctypes.POINTER(ctypes.c_double)


def _rblk_emu_lib(name, budget, tmp, **kw):
    return rblk_emu_lib(name, budget, tmp, **kw)


@pytest.mark.parametrize('name,budget,kw', [
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, c_lds=1)),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40)),
    ('synth_srichb', 16, dict(blocks_per_part=2, rates_per_part=5)),
    # N1: fractional stoichiometric coefficients, more than three molecules / species per side
    ('synth_fracnu', 16, dict(blocks_per_part=2, rates_per_part=6)),
    # the lean outputs' fast kernel (k_jvd's dydt build) in the GPU geometry: four lane groups on shared concentration columns,
    # K_c rows from LDS copies / from the table in global memory
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40, jvd=(4, 1, 1))),
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, jvd=(4, 1, 1, 1))),
])
def test_rate_outputs_vs_oracle(name, budget, kw, tmp_path, tables):
    """k_rate (pj_spec_rates of the row-block library): conc, fwd, rev, pres_mod, spec_rates handed from
    rate kernel to rate kernel, dydt -- with every array requested (the kernels with per-reaction outputs) and
    with dydt only (the lean kernels; omega_k then travels through the library's scratch array)."""
    from oracle.oracle import Oracle
    ev, L = _rblk_emu_lib(name, budget, str(tmp_path), **kw)
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
            assert not np.isnan(bufs[k][:cols]).any(), k
            mx, _ = thresholded_rel_err(bufs[k].T[:, :cols], g[k][:, :cols])
            assert mx < 1e-9, (k, mx)
    gross, sdy = rate_scales(tab, pres, y_aos, g['conc'], g['fwd'], g['rev'], g['pres_mod'])
    assert mixed_err(bufs['spec_rates'].T, g['spec_rates'], gross[:, None]) <= 1.0
    assert mixed_err(bufs['dy'].T, g['dydt'], sdy) <= 1.0
    # the lean outputs (conc / spec_rates / dydt): through k_jvd's dydt build (every reaction once, several lane groups; the
    # default where the library has one) and through k_rate's lean kernels
    L.pj_spec_ctx_rate_fast.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.pj_spec_ctx_rate_fast(None, -1) == 1
    for fast in (1, 0):
        assert L.pj_spec_ctx_rate_fast(None, fast) == fast
        dy2, sr2, c2 = np.full((nsp, n), np.nan), np.full((nsp, n), np.nan), np.full((nsp, n), np.nan)
        assert L.pj_spec_rates(n, P(pres), P(y), n, 1, P(c2), None, None, None, P(sr2), P(dy2), None) == 0
        assert not np.isnan(dy2).any() and not np.isnan(sr2).any() and not np.isnan(c2).any()
        if fast:
            mx, _ = thresholded_rel_err(c2.T, g['conc'])
            assert mx < 1e-9, mx
            assert mixed_err(sr2.T, g['spec_rates'], gross[:, None]) <= 1.0
            assert mixed_err(dy2.T, g['dydt'], sdy) <= 1.0
        else:
            assert np.array_equal(dy2, bufs['dy']) and np.array_equal(sr2, bufs['spec_rates']) and np.array_equal(c2, bufs['conc'])
        dy4 = np.full((nsp, n), np.nan)
        assert L.pj_spec_rates(n, P(pres), P(y), n, 1, None, None, None, None, None, P(dy4), None) == 0
        assert np.array_equal(dy4, dy2)
        # layouts: AoS states (y_si = 1, y_ss = NSP)
        dy3 = np.full((nsp, n), np.nan)
        assert L.pj_spec_rates(n, P(pres), P(y_aos), 1, nsp, None, None, None, None, None, P(dy3), None) == 0
        assert np.array_equal(dy3, dy2)
    L.pj_spec_ctx_rate_fast(None, 1)


@pytest.mark.parametrize('name,budget,kw', [
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, c_lds=1)),
    ('synth_alltypes', 200, dict(blocks_per_part=1, rates_per_part=1000)),
    ('h2o2', 24, dict(blocks_per_part=3, rates_per_part=10)),
    ('h2o2_n2', 12, dict(blocks_per_part=100, rates_per_part=5)),
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40)),
    # several lane groups per workgroup on the same states (each an OS thread in the emulation, a real barrier behind
    # __syncthreads): groups split the row blocks of a kernel, exchange the energy-row sums and share its columns
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40, halves=2)),
    # ... four lane groups over SEVERAL row kernels (buildable through PJ_RBLK_HALVES / what more than 120 species get), and a
    # factor-column build with another look-ahead depth of the hand-over ring than specbuild's default (PJQ_DEPTH = 3)
    ('synth_mid24', 40, dict(blocks_per_part=8, rates_per_part=40, halves=4, defines=('-DPJQ_COOP=1',))),
    ('synth_mid24', 40, dict(rates_per_part=40, kcf=1, halves=4, single=1, defines=('-DPJQ_DEPTH=1',))),
    # the energy-row terms a row block cannot see (enhanced colliders, a falloff collider, a species on both sides): summed
    # once per state by the pre-pass (PJQ_ECL, the default with several lane groups -- the cases above and below) -- here
    # with a two-group pre-pass (the 111-species geometry) and with one lane group
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40, halves=2, c_lds=1, pre_halves=2)),
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7, c_lds=1, ecl=1)),
    ('synth_srichb', 16, dict(blocks_per_part=2, rates_per_part=5, halves=2, c_lds=1, pre_halves=2)),
    # PJQ_FIN: the energy row finished by k_fin, a kernel of its own behind the row kernels (here: one row kernel whose
    # column sums travel through the hand-over array)
    ('synth_alltypes', 16, dict(rates_per_part=7, halves=4, single=1, fin=1, defines=('-DPJQ_ECOLS=1', '-DPJQ_COOP=1'))),
    # ... and PJQ_ECL with the factor columns: one kernel (the sums land in the EJ columns behind the prologue) and several
    ('synth_alltypes', 16, dict(rates_per_part=7, kcf=1, halves=4, single=1, ecl=1)),
    # ONE row kernel with polynomial K_c rows (the 111-species geometry: no LDS room for the finished column sums, which
    # travel through the hand-over array -- PJQ_ECOLS), four lane groups with a cooperative prologue (PJQ_COOP)
    ('synth_alltypes', 16, dict(rates_per_part=7, halves=4, single=1, defines=('-DPJQ_ECOLS=1', '-DPJQ_COOP=1'))),
    ('synth_srichb', 16, dict(rates_per_part=5, halves=2, single=1, c_lds=1, pre_halves=2,
                              defines=('-DPJQ_ECOLS=1', '-DPJQ_COOP=1', '-DPJQ_DEFER=1'))),
    # PJQ_DEFER: the rows of a block are stored during the visits of the next one
    ('synth_alltypes', 16, dict(rates_per_part=7, kcf=1, halves=4, single=1, defines=('-DPJQ_DEFER=1',))),
    # equilibrium constants from per-species factor columns (PJQ_KCF: cooperative prologue, products instead of a
    # polynomial + exp per visit): one group and several kernels; four groups and ONE kernel (the 53-species shape)
    ('synth_mid24', 40, dict(blocks_per_part=4, rates_per_part=40, kcf=1)),
    ('synth_alltypes', 16, dict(rates_per_part=7, kcf=1, halves=4, single=1)),
    ('h2o2_n2', 12, dict(blocks_per_part=2, rates_per_part=5, kcf=1, halves=2)),
    # ... two groups and ONE kernel (what mechanisms of up to 26 species get: specbuild.rblk_geometry)
    ('synth_mid24', 56, dict(rates_per_part=40, kcf=1, halves=2, single=1)),
    # ... with species of three different T_mid (range select per species instead of per K_c group)
    ('fe_septherm', 16, dict(rates_per_part=9, kcf=1, halves=4, single=1)),
    # ... with SRI / Chebyshev reactions (reversible hand-over visits take their 1 / K_c from the factors too)
    ('synth_srichb', 16, dict(rates_per_part=5, kcf=1, halves=4, single=1)),
    # SRI falloff (3 / 5 parameters, LOW / HIGH, collider) and Chebyshev reactions: evaluated by the pre-pass
    ('synth_srichb', 16, dict(blocks_per_part=2, rates_per_part=5)),
    # N1: fractional stoichiometric coefficients (pow), more than three molecules / species per side, also on
    # third-body and falloff reactions and with the last species as a reactant
    ('synth_fracnu', 16, dict(blocks_per_part=2, rates_per_part=6)),
    ('synth_fracnu', 200, dict(blocks_per_part=1, rates_per_part=1000, c_lds=1)),
    # species with three different T_mid (separate thermo database): several pre-summed K_c groups per reaction
    ('fe_septherm', 16, dict(blocks_per_part=2, rates_per_part=9)),
])
def test_rblk_kernels_vs_oracle(name, budget, kw, tmp_path, tables):
    """Row blocks that rebuild their rates (Arrhenius, K_c, third body, theta per visit), the falloff /
    PLOG pre-pass with its register ring, the d/dT column finished per block, energy-row partials handed
    from kernel to kernel: against the oracle, both layouts, with and without the J_nplusone quirk."""
    from oracle.oracle import Oracle
    ev, L = _rblk_emu_lib(name, budget, str(tmp_path), **kw)
    orc = Oracle(tables(name))
    n = 262                               # crosses a 256-state hand-over tile
    # (T from 300 K: every range of every species' NASA polynomials and K_c group is visited)
    pres, y = synth.dist_b(n, ev.nsp) if name != 'fe_septherm' else synth.dist_b(n, ev.nsp, seed=3, Tlo=300, Thi=2600)
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


def test_rblk_kernels_vs_reference_golden(tmp_path_factory, golden):
    """53-species mechanism at the shipping budget against vectors from pyJac's generated C; the library's
    rate outputs (k_rate, one kernel for the whole mechanism) against the same vectors."""
    ev, L = _rblk_emu_lib('gri30_shaped', 56, tmp_path_factory, blocks_per_part=13, c_lds=0)
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
    # ... and w = J v (k_jvd: every reaction once) against pyJac's own Jacobians times the same vectors
    L.pj_spec_jacvec.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp, ctypes.c_long, ctypes.c_long,
                                 _dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    L.pj_spec_ctx_row_jv.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.pj_spec_ctx_row_jv(None, 0) == 0
    v = np.random.default_rng(4).standard_normal((nsp, n))
    v[0] *= 100.0
    J = g['jac'].reshape(n, nsp, nsp)           # [s][col][row]
    ref = np.einsum('scr,cs->sr', J, v)
    scale = np.einsum('scr,cs->sr', np.abs(J), np.abs(v)) + 1e-300
    w = np.full((nsp, n), np.nan)
    assert L.pj_spec_jacvec(n, P(pres), P(y), n, 1, P(np.ascontiguousarray(v)), n, 1, P(w), n, 1, 0, None) == 0
    assert (np.abs(w.T - ref) / scale).max() < 1e-9


@pytest.mark.parametrize('name,budget,kw', [
    ('synth_alltypes', 16, dict(blocks_per_part=2, rates_per_part=7)),
    # one kernel, four lane groups, factor columns: v lives in an LDS column set, finished column sums fold into w_0
    ('synth_alltypes', 16, dict(rates_per_part=7, kcf=1, halves=4, single=1)),
    # two lane groups over several kernels: v in registers, column sums through the hand-over array
    ('synth_srichb', 16, dict(blocks_per_part=2, rates_per_part=5, halves=2, c_lds=1, pre_halves=2)),
    # ... with the pre-pass's column sums (PJQ_ECL) folded into w_0's share behind the prologue
    ('synth_alltypes', 16, dict(rates_per_part=7, kcf=1, halves=4, single=1, ecl=1)),
])
def test_rblk_fused_jacobian_vector_product(name, budget, kw, tmp_path, tables):
    """N2 for the row-block family: w = J v per state with the Jacobian consumed in registers
    (pj_spec_jacvec through the row kernels' PJQ_JV builds: pj_spec_ctx_row_jv) against the oracle's Jacobian times the
    same vectors, both layouts."""
    from oracle.oracle import Oracle
    ev, L = _rblk_emu_lib(name, budget, str(tmp_path), **kw)
    L.pj_spec_ctx_row_jv.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.pj_spec_ctx_row_jv(None, 1) == 1
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


@pytest.mark.parametrize('name,budget,kw', [
    # everything in registers; reaction ranges over several kernels (D_k and the scalars through `sr`)
    ('synth_alltypes', 16, dict(rates_per_part=7)),
    ('synth_srichb', 16, dict(rates_per_part=5)),
    ('synth_fracnu', 16, dict(rates_per_part=6)),
    ('synth_mid24', 40, dict(rates_per_part=1000, jvd=(1, 1, 0))),
    # two lane groups on the same states (every other reaction each), concentrations in LDS
    ('synth_mid24', 40, dict(rates_per_part=40, jvd=(2, 1, 0))),
    # the large-mechanism geometry: four lane groups, concentrations and vector in LDS columns
    ('synth_alltypes', 16, dict(rates_per_part=7, jvd=(4, 1, 1))),
    ('synth_mid24', 40, dict(rates_per_part=1000, jvd=(4, 1, 1))),
    # ... with the K_c rows read from the mechanism table instead of LDS copies
    ('synth_srichb', 16, dict(rates_per_part=1000, jvd=(4, 1, 1, 1))),
    ('synth_alltypes', 16, dict(rates_per_part=1000, jvd=(4, 1, 1, 1, 2))),
    ('synth_alltypes', 16, dict(rates_per_part=7, jvd=(1, 0, 0, 0, 2))),
])
@pytest.mark.parametrize('sum_last', [0, 1])
def test_jvd_directional_derivative_vs_oracle(name, budget, kw, sum_last, tmp_path, tables):
    """k_jvd (pj_spec_jacvec's default): w = J v with every reaction visited once -- d_i = the reaction's derivative row
    times the vector, scattered like q_i -- against the oracle's Jacobian times the same vectors, with and without the
    J_nplusone quirk (sum_last), both layouts, a batch that ends inside a workgroup."""
    from oracle.oracle import Oracle
    ev, L = _rblk_emu_lib(name, budget, str(tmp_path), only_jvd=True, **kw)
    L.pj_spec_jacvec.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, ctypes.c_long, _dp, ctypes.c_long, ctypes.c_long,
                                 _dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    L.pj_spec_ctx_row_jv.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.pj_spec_ctx_row_jv(None, 0) == 0
    n, nsp = 131, ev.nsp
    pres, y = synth.dist_b(n, nsp, seed=12, Tlo=400, Thi=2500)
    v = np.random.default_rng(3).standard_normal((nsp, n))
    v[0] *= 100.0
    orc = Oracle(tables(name))
    orc.lib.pjo_set_sum_last_species(sum_last)
    try:
        J = orc.batch_jacob(pres, np.ascontiguousarray(y.T)).reshape(n, nsp, nsp)   # [s][col][row]
    finally:
        orc.lib.pjo_set_sum_last_species(0)
    ref = np.einsum('scr,cs->sr', J, v)
    scale = np.einsum('scr,cs->sr', np.abs(J), np.abs(v)) + 1e-300
    P = lambda a: a.ctypes.data_as(_dp)
    ys, vs, ws = np.ascontiguousarray(y), np.ascontiguousarray(v), np.full((nsp, n), np.nan)
    assert L.pj_spec_jacvec(n, P(pres), P(ys), n, 1, P(vs), n, 1, P(ws), n, 1, sum_last, None) == 0
    assert (np.abs(ws.T - ref) / scale).max() < 1e-9
    ya, va, wa = np.ascontiguousarray(y.T), np.ascontiguousarray(v.T), np.full((n, nsp), np.nan)
    assert L.pj_spec_jacvec(n, P(pres), P(ya), 1, nsp, P(va), 1, nsp, P(wa), 1, nsp, sum_last, None) == 0
    assert (np.abs(wa - ref) / scale).max() < 1e-9
