"""Whose rounding error is it?  ("weak" #1.)

On the 53- / 111-species mechanisms a few Jacobian entries of the HIP kernels differ from pyJac's generated C
by more than the north star's entry-wise rtol 1e-6 (reference tester's metric: 4e-6 GRI-shaped, 2e-3
USC-shaped).  Those entries are ~1e-13 of their row / column scale: sums of terms 1e13 larger.  This test
settles the question with an extended-precision truth -- the oracle's text (pyJac's formulas, constants and
evaluation order) compiled in binary128 (oracle/pyjac_oracle_quad.c), rounded to binary64 at the end:

* the kernels' regrouped formulation (pj_rblk.hip, run here through the CPU emulation build) is within 1e-7
  of the truth on EVERY entry (measured 2e-9 / 2e-10): it meets rtol 1e-6 against the exact value;
* pyJac's own evaluation order (the oracle in binary64, and the golden vectors from pyJac's generated C) is
  what is off by up to 1e-6 / 2e-3 on exactly the entries where kernel and reference disagree.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))
from conftest import rel_err_entries, truth_report  # noqa: E402
from emu_libs import rblk_emu_lib, run_jacobian  # noqa: E402
from pyjac_amd import synth  # noqa: E402


def _ktab_emulated(tab, pres, y_soa):
    """The same formulation through the table-driven kernel k_tab, emulated thread by thread (tests/emu/emu.cpp):
    no per-mechanism compilation, so the 111-species case costs seconds instead of a two-minute emulation build."""
    import ctypes
    from test_host_logic import _emu
    dp = ctypes.POINTER(ctypes.c_double)
    n = pres.size
    I = np.ascontiguousarray(tab.I, dtype=np.int32)
    D = np.ascontiguousarray(tab.D)
    jac = np.full(tab.nsp * tab.nsp * n, np.nan)
    P = lambda a: a.ctypes.data_as(dp)
    rc = _emu().emu_tab_run(I.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_long(I.size), P(D),
                            ctypes.c_long(D.size), ctypes.c_long(n), P(np.ascontiguousarray(pres)),
                            P(np.ascontiguousarray(y_soa)), P(jac), 0, 0, ctypes.c_long(156 * 1024), None)
    assert rc == 0
    return jac.reshape(-1, n).T


def test_factor_form_of_the_equilibrium_constants_against_the_truth(tables, golden, tmp_path_factory):
    """The one-kernel geometry of the 53-species mechanism (csrc/pj_rblk.hip PJQ_KCF: 1 / K_c as a product of per-species
    factors exp(ln X_k) that the prologue evaluates once per state, T dlnK_c/dT as a sum of t_k, the energy row finished
    column by column from reaction enthalpies; four lane groups as four OS threads) against the binary128 truth: the
    factors carry eps |ln X_k| each, a few times the rounding error of the pre-summed polynomial -- every entry stays
    three orders of magnitude inside rtol 1e-6."""
    from oracle.oracle import Oracle, OracleQuad
    name = 'gri30_shaped'
    tab = tables(name)
    nsp = tab.nsp
    L = rblk_emu_lib(name, 48, tmp_path_factory, kcf=1, halves=4, single=1, c_lds=0, only_rows=True)[1]
    g = golden(name)
    pres, y = synth.dist_b(120, nsp, seed=11, Tlo=300, Thi=3000)
    pres = np.concatenate([g['pres'], pres])
    y = np.concatenate([g['y'].T, y], axis=1)
    y_aos = np.ascontiguousarray(y.T)
    truth = OracleQuad(tab).batch_jacob(pres, y_aos)
    orc = Oracle(tab).batch_jacob(pres, y_aos)
    emu = run_jacobian(L, nsp, pres, y)
    rep = truth_report(emu, orc, truth, nsp, label='%s (emulated pj_rblk, factor columns / 4 groups / 1 kernel, %d states)' % (name, pres.size))
    assert rep['test_vs_truth'] < 1e-7 and rep['test_over_1e6'] == 0
    if rep['n_bad']:
        assert rep['bad_explained'] and rep['bad_size_max'] < 1e-9


@pytest.mark.parametrize('name,n_random', [('gri30_shaped', 300), ('usc2_shaped', 48)])
def test_regrouped_formulation_is_closer_to_the_truth_than_the_reference(name, n_random, tables, golden, tmp_path_factory):
    from oracle.oracle import Oracle, OracleQuad
    tab = tables(name)
    nsp = tab.nsp
    # 53 species: the compiled row-block kernels through their CPU emulation build (shared with
    # tests/test_rblk_emu.py); 111 species: the table-driven kernel of the same formulation
    L = rblk_emu_lib(name, 56, tmp_path_factory, blocks_per_part=13, c_lds=0)[1] if nsp <= 64 else None
    g = golden(name)
    pres, y = synth.dist_b(n_random, nsp, seed=11, Tlo=800, Thi=2500)
    pres = np.concatenate([g['pres'], pres])
    y = np.concatenate([g['y'].T, y], axis=1)
    y_aos = np.ascontiguousarray(y.T)
    ng = g['pres'].size
    truth = OracleQuad(tab).batch_jacob(pres, y_aos)
    orc = Oracle(tab).batch_jacob(pres, y_aos)
    emu = run_jacobian(L, nsp, pres, y) if L is not None else _ktab_emulated(tab, pres, y)
    rep = truth_report(emu, orc, truth, nsp, label='%s (emulated %s vs oracle, %d states)' % (
        name, 'pj_rblk' if L is not None else 'k_tab', pres.size))
    # the kernels' formulation meets the entry-wise tolerance against the exact value, with room to spare
    assert rep['test_vs_truth'] < 1e-7
    assert rep['test_over_1e6'] == 0
    # where kernel and reference disagree, the reference's own rounding error is the whole difference
    if rep['n_bad']:
        assert rep['bad_explained'] and rep['bad_size_max'] < 1e-9
    # and the reference's order is what exceeds 1e-6 against the truth on the 111-species mechanism
    if name == 'usc2_shaped':
        assert rep['ref_vs_truth'] > 1e-6 and rep['ref_over_1e6'] >= rep['n_bad']
    # same statement for the committed vectors from pyJac's generated C (not only our restatement of it)
    r_gold = rel_err_entries(g['jac'], truth[:ng])
    r_emu = rel_err_entries(emu[:ng], truth[:ng])
    print('%s golden states: pyJac generated C vs truth %.3g, emulated kernel vs truth %.3g' % (name, r_gold.max(), r_emu.max()))
    assert r_emu.max() < 1e-7 and r_emu.max() <= r_gold.max()


def test_truth_agrees_with_binary64_oracle_where_well_conditioned(tables, golden):
    """The binary128 build is the same text as the oracle: on the H2 mechanism (no ill-conditioned entries)
    the two agree to rounding, rates and Jacobian."""
    from oracle.oracle import Oracle, OracleQuad
    tab = tables('h2o2_n2')
    g = golden('h2o2_n2')
    q, o = OracleQuad(tab), Oracle(tab)
    for s in (0, 40, 101):
        a, b = q.eval_all(float(g['pres'][s]), g['y'][s]), o.eval_all(float(g['pres'][s]), g['y'][s])
        for k in ('conc', 'fwd', 'rev', 'pres_mod'):
            assert np.allclose(a[k], b[k], rtol=1e-13, atol=0), k
        assert rel_err_entries(b['jac'], a['jac']).max() < 1e-9


def self_noise_report(test, ref, ref_other, label=''):
    """Evidence that needs no trust in the binary128 restatement: `ref` and `ref_other` are two builds of the SAME
    C that pyJac's generator emitted (reference flags vs -mfma -ffp-contract=fast, oracle/build_ref.py VARIANTS).
    d = kernel vs pyJac, s = pyJac vs pyJac, entry by entry under the reference tester's metric."""
    d, s = rel_err_entries(test, ref), rel_err_entries(ref_other, ref)
    bad = d > 1e-6
    repro = s < 1e-11           # entries pyJac reproduces under a change of compiler flags
    rep = dict(kernel_vs_ref=float(d.max()), self_noise=float(s.max()), n_bad=int(bad.sum()),
               self_over_1e6=int((s > 1e-6).sum()), repro_frac=float(repro.mean()),
               kernel_vs_ref_on_repro=float(d[repro].max()))
    if bad.any():
        rep.update(bad_with_self_over_1e9=float((s[bad] > 1e-9).mean()),
                   bad_self_over_d_median=float(np.median(s[bad] / d[bad])))
    print('%s kernel vs pyJac %.3g; pyJac vs pyJac (other flags) %.3g; on the %.1f %% of entries pyJac reproduces to '
          '1e-11: kernel vs pyJac %.3g; entries kernel-vs-pyJac > 1e-6: %d%s'
          % (label, rep['kernel_vs_ref'], rep['self_noise'], 100 * rep['repro_frac'], rep['kernel_vs_ref_on_repro'],
             rep['n_bad'], '' if not bad.any() else ' (%.1f %% of them differ between the two pyJac builds by > 1e-9, '
             'median self / kernel difference %.2f)' % (100 * rep['bad_with_self_over_1e9'], rep['bad_self_over_d_median'])))
    return rep


@pytest.mark.parametrize('name,n_random', [('gri30_shaped', 300), ('usc2_shaped', 48)])
def test_reference_disagrees_with_itself_where_the_kernels_disagree_with_it(name, n_random, tables, golden, tmp_path_factory):
    """pyJac's generated C compiled with -O3 -mfma -ffp-contract=fast differs from the same C compiled with the
    reference's flags by 7e-7 (GRI-shaped) / 9e-4 (USC-shaped) -- on the same entries, and by the same order, as
    the kernels differ from it.  -O0 against -O3 is bit-identical (x86-64 without -mfma never contracts)."""
    import json
    from oracle.oracle import Reference
    if not (Reference.available(name) and Reference.available(name + '_fma')):
        pytest.skip('oracle/_ref variants not built (no /root/reference here)')
    tab = tables(name)
    nsp = tab.nsp
    L = rblk_emu_lib(name, 56, tmp_path_factory, blocks_per_part=13, c_lds=0)[1] if nsp <= 64 else None
    g = golden(name)
    pres, y = synth.dist_b(n_random, nsp, seed=11, Tlo=800, Thi=2500)
    pres = np.concatenate([g['pres'], pres])
    y = np.concatenate([g['y'].T, y], axis=1)
    y_aos = np.ascontiguousarray(y.T)
    ref, ref_fma = Reference(name).batch_jacob(pres, y_aos), Reference(name + '_fma').batch_jacob(pres, y_aos)
    if Reference.available(name + '_O0'):
        assert np.array_equal(Reference(name + '_O0').batch_jacob(pres, y_aos), ref)
    emu = run_jacobian(L, nsp, pres, y) if L is not None else _ktab_emulated(tab, pres, y)
    rep = self_noise_report(emu, ref, ref_fma, label='%s (%d states):' % (name, pres.size))
    # 1. where pyJac is reproducible, the kernels meet the north star's entry-wise rtol 1e-6 against pyJac itself
    assert rep['repro_frac'] > 0.9 and rep['kernel_vs_ref_on_repro'] < 1e-6
    # 2. the kernels are not further from pyJac than 10x pyJac is from itself (MX_BIG of the GPU tests)
    assert rep['kernel_vs_ref'] < 10 * rep['self_noise']
    # 3. (almost) every entry on which kernel and pyJac differ by more than 1e-6 is one pyJac does not reproduce
    if rep['n_bad']:
        assert rep['bad_with_self_over_1e9'] > 0.95
    if name == 'usc2_shaped':
        assert rep['self_noise'] > 1e-6 and rep['self_over_1e6'] > 0       # pyJac misses rtol 1e-6 against itself
    # the committed fixture the GPU tests read (tests/golden/make_self_noise.py) was measured the same way
    fix = json.load(open(os.path.join(HERE, 'golden', 'self_noise.json')))[name]
    assert 0.1 * fix['self_noise'] < rep['self_noise'] < 10 * fix['self_noise']
