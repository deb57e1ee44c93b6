import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
MECHS = {
    'h2o2_n2': os.path.join(ROOT, 'pyjac_amd', 'data', 'h2o2_n2.inp'),
    'h2o2': os.path.join(GOLDEN, 'h2o2.inp'),
    'synth_alltypes': os.path.join(GOLDEN, 'synth_alltypes.inp'),
    'gri30_shaped': os.path.join(ROOT, 'pyjac_amd', 'data', 'gri30_shaped.inp'),
    'usc2_shaped': os.path.join(ROOT, 'pyjac_amd', 'data', 'usc2_shaped.inp'),
    'synth_mid24': os.path.join(GOLDEN, 'synth_mid24.inp'),     # 24 sp / 96 rxn incl. Troe, PLOG
    'synth_srichb': os.path.join(GOLDEN, 'synth_srichb.inp'),   # SRI falloff (3 / 5 parameters) + Chebyshev
    # fractional stoichiometric coefficients, more than three molecules / species on a reaction side
    'synth_fracnu': os.path.join(GOLDEN, 'synth_fracnu.inp'),
    # 72 species / 260 reactions, 200 of them irreversible: the second mechanism of the two-lane-group row kernels
    # (57..120 species), with row kernels that touch only a handful of K_c groups
    'synth_irrev72': os.path.join(GOLDEN, 'synth_irrev72.inp'),
    # front-end corners (tests/golden/make_frontend_mechs.py): the 28 H2/O2 reactions with a units keyword on the
    # REACTIONS line (mech_interpret.py:42-49, 135-159) ...
    'fe_kcal': os.path.join(GOLDEN, 'fe_kcal.inp'),
    'fe_kelvins': os.path.join(GOLDEN, 'fe_kelvins.inp'),
    'fe_kjoules': os.path.join(GOLDEN, 'fe_kjoules.inp'),
    'fe_joules': os.path.join(GOLDEN, 'fe_joules.inp'),
    'fe_evolts': os.path.join(GOLDEN, 'fe_evolts.inp'),
    # ... and with a separate thermodynamic database: plain THERMO header, common middle temperature 1200 K, cards
    # with / without their own, a foreign and a repeated card (mech_interpret.py:735-883): three different T_mid
    'fe_septherm': os.path.join(GOLDEN, 'fe_septherm.inp'),
}
# planner-geometry sweep (tests/golden/make_sweep_mechs.py): 9 mechanisms around the geometry thresholds of the row kernels
# (17 .. 140 species) + 5 seeded random ones; sweep_n054 / sweep_n121 have golden vectors of pyJac's generated C
import glob as _glob
MECHS.update({os.path.basename(_f)[:-4]: _f for _f in sorted(_glob.glob(os.path.join(GOLDEN, 'sweep', 'sweep_*.inp')))})
THERMS = {'fe_septherm': os.path.join(GOLDEN, 'fe_septherm.dat')}
FRONT_END = ('fe_kcal', 'fe_kelvins', 'fe_kjoules', 'fe_joules', 'fe_evolts', 'fe_septherm')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def thresholded_rel_err(test, ref):
    """Error metric of the reference's functional tester
    (pyjac/functional_tester/test.py:1446-1463): relative error on entries with
    |ref| > ||ref||_2 / 1e20, per state; returns (max, ||d||/||ref||)."""
    test = np.atleast_2d(test)
    ref = np.atleast_2d(ref)
    nrm = np.linalg.norm(ref, axis=1, keepdims=True)
    mask = np.abs(ref) > nrm / 1e20
    rel = np.zeros_like(ref)
    rel[mask] = np.abs(test - ref)[mask] / np.abs(ref)[mask]
    fro = np.linalg.norm(test - ref, axis=1) / np.maximum(nrm[:, 0], 1e-300)
    return float(rel.max()), float(fro.max())


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '_golden.npz'))
    return load


@pytest.fixture(scope='session')
def tables():
    from pyjac_amd.mechanism import read_mech
    from pyjac_amd.tables import build_tables
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = build_tables(read_mech(MECHS[name], THERMS.get(name)))
        return cache[name]
    return get


def rate_scales(tab, pres, y, conc, fwd, rev, pres_mod):
    """Cancellation scales for net rates.  Near equilibrium the species rates are
    differences of forward/reverse rates that agree to ~16 digits, so a pure
    relative error is meaningless (SURVEY.md 7.4 'Tolerance definition'); errors
    in omega_k / dY_k/dt / dT/dt are judged against the gross rate
    max_i |q_f,i|, |q_r,i| (times pres_mod) of the state, propagated to each row.
    Returns (gross[n], scale_dydt[n, NSP])."""
    from pyjac_amd import tables as T
    I, D = tab.I, tab.D
    ia = lambda j, cnt: I[I[16 + j]:I[16 + j] + cnt]
    da = lambda j, cnt: D[I[48 + j]:I[48 + j] + cnt]
    nsp, nrxn = tab.nsp, tab.nrxn
    rev_idx, pres_idx = ia(T.IA_REV_IDX, nrxn), ia(T.IA_PRES_IDX, nrxn)
    mw, tmid = da(T.DA_MW, nsp), da(T.DA_TMID, nsp)
    lo, hi = da(T.DA_LO, 7 * nsp).reshape(nsp, 7), da(T.DA_HI, 7 * nsp).reshape(nsp, 7)
    n = pres.size
    gross = np.zeros(n)
    for i in range(nrxn):
        c = pres_mod[:, pres_idx[i]] if pres_idx[i] >= 0 else 1.0
        gross = np.maximum(gross, np.abs(fwd[:, i] * c))
        if rev_idx[i] >= 0:
            gross = np.maximum(gross, np.abs(rev[:, rev_idx[i]] * c))
    rho = conc @ mw
    Tt = y[:, 0][:, None]
    a = np.where((Tt <= tmid[None, :])[:, :, None], lo[None], hi[None])
    RU = 8314.4621
    h = RU / mw * (a[..., 5] + Tt * (a[..., 0] + Tt * (a[..., 1] / 2 + Tt * (a[..., 2] / 3 + Tt * (a[..., 3] / 4 + a[..., 4] / 5 * Tt)))))
    cp = RU / mw * (a[..., 0] + Tt * (a[..., 1] + Tt * (a[..., 2] + Tt * (a[..., 3] + a[..., 4] * Tt))))
    Y = conc * mw / rho[:, None]
    cpavg = (Y * cp).sum(axis=1)
    sc = np.empty((n, nsp))
    sc[:, 1:] = gross[:, None] * mw[None, :-1] / rho[:, None]
    sc[:, 0] = gross * (np.abs(h) * mw).sum(axis=1) / (rho * cpavg)
    return gross, sc


def mixed_err(test, ref, scale, rtol=1e-6, ctol=1e-10):
    """max |test - ref| / (rtol |ref| + ctol scale): <= 1 passes."""
    return float((np.abs(test - ref) / (rtol * np.abs(ref) + ctol * scale + 1e-300)).max())


def jac_scaled_err(test, ref, nsp, rtol=1e-6, ctol=1e-12):
    """Entry-wise Jacobian check that is meaningful for cancellation-dominated
    entries: |d_kj| <= rtol |J_kj| + ctol max(max_j |J_k.|, max_k |J_.j|).
    Entries that are 1e-12 of their row/column scale are differences of terms
    ~1e12 times larger, so their relative error is bounded by the conditioning,
    not by the implementation (the reference's own builds differ
    by up to 6e-9 there).  test/ref: (n, nsp*nsp) column-major blocks.
    Returns max over all entries of |d| / tolerance (<= 1 passes)."""
    t = np.asarray(test).reshape(-1, nsp, nsp)
    r = np.asarray(ref).reshape(-1, nsp, nsp)
    a = np.abs(r)
    colmax = a.max(axis=2, keepdims=True)      # blocks are [col][row]
    rowmax = a.max(axis=1, keepdims=True)
    tol = rtol * a + ctol * np.maximum(rowmax, colmax) + 1e-300
    return float((np.abs(t - r) / tol).max())


def rel_err_entries(test, ref):
    """Per-entry relative error under the reference tester's mask (|ref| > ||ref||_2 / 1e20 per state,
    functional_tester/test.py:1446-1463); masked-out entries count as 0."""
    test, ref = np.atleast_2d(test), np.atleast_2d(ref)
    mask = np.abs(ref) > np.linalg.norm(ref, axis=1, keepdims=True) / 1e20
    rel = np.zeros_like(ref)
    rel[mask] = np.abs(test - ref)[mask] / np.abs(ref)[mask]
    return rel


def truth_report(test, ref, truth, nsp, label=''):
    """Where `test` (a HIP kernel or its CPU emulation) and `ref` (pyJac's generated C or the oracle that keeps
    its evaluation order) disagree by more than 1e-6 on a Jacobian entry, whose rounding error is it?
    `truth` is the binary128 evaluation of the reference's formulas (oracle/pyjac_oracle_quad.c) rounded to
    binary64.  Returns the figures DESIGN.md section 2 quotes and prints them (pytest -s / GPUTEST log)."""
    n = np.atleast_2d(ref).shape[0]
    r_tr, r_tt, r_rt = rel_err_entries(test, ref), rel_err_entries(test, truth), rel_err_entries(ref, truth)
    bad = r_tr > 1e-6
    rep = dict(test_vs_ref=float(r_tr.max()), test_vs_truth=float(r_tt.max()), ref_vs_truth=float(r_rt.max()),
               n_bad=int(bad.sum()), bad_per_state=float(bad.sum()) / n, ref_over_1e6=int((r_rt > 1e-6).sum()),
               test_over_1e6=int((r_tt > 1e-6).sum()))
    if bad.any():
        a = np.abs(np.atleast_2d(truth)).reshape(n, nsp, nsp)
        scale = np.maximum(a.max(axis=1, keepdims=True), a.max(axis=2, keepdims=True)) * np.ones_like(a)
        size = (a / (scale + 1e-300)).reshape(n, -1)[bad]
        rep.update(bad_test_vs_truth=float(r_tt[bad].max()), bad_ref_vs_truth_median=float(np.median(r_rt[bad])),
                   bad_size_max=float(size.max()), bad_size_median=float(np.median(size)),
                   bad_explained=bool((r_tt[bad] <= 1e-3 * r_tr[bad]).all()))
    print('%s entry-wise vs binary128 truth: kernel %.3g, reference %.3g (kernel vs reference %.3g); entries where '
          'kernel and reference differ by > 1e-6: %d (%.3g per state)%s'
          % (label, rep['test_vs_truth'], rep['ref_vs_truth'], rep['test_vs_ref'], rep['n_bad'], rep['bad_per_state'],
             '' if not bad.any() else '; on those kernel-vs-truth <= %.3g, reference-vs-truth median %.3g, |J| / row-column '
             'scale median %.3g max %.3g' % (rep['bad_test_vs_truth'], rep['bad_ref_vs_truth_median'],
                                              rep['bad_size_median'], rep['bad_size_max'])))
    return rep


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """The two CPU emulation builds of the 53-species row kernels (single g++ translation units of 1 - 2 minutes) start as
    soon as a test that needs them is known to run -- in the background, next to the tests in front of it."""
    ids = [i.nodeid for i in items]
    want_a = any('test_conditioning.py' in x or 'test_rblk_kernels_vs_reference_golden' in x for x in ids)
    want_b = any('test_factor_form_of_the_equilibrium_constants' in x for x in ids)
    if not (want_a or want_b) or config.option.collectonly:
        return
    import atexit
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
    try:
        import emu_libs
        d = tempfile.mkdtemp(prefix='pj_emu_prefetch_')
        atexit.register(shutil.rmtree, d, True)
        if want_b:
            emu_libs.prefetch('gri30_shaped', 48, d, kcf=1, halves=4, single=1, c_lds=0, only_rows=True)
        if want_a:
            emu_libs.prefetch('gri30_shaped', 56, d, blocks_per_part=13, c_lds=0)
    except Exception as ex:         # (the tests build what they need themselves)
        sys.stderr.write('emulation prefetch not started: %r\n' % (ex,))
