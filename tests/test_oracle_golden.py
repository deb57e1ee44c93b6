"""CPU: the oracle (oracle/pyjac_oracle.c) against golden vectors produced by the
reference's own generated C (tests/golden/make_golden.py), and -- when
/root/reference's build is present -- against that library live."""
import numpy as np
import pytest

from conftest import MECHS, thresholded_rel_err
from oracle.oracle import Oracle, Reference

KEYS = ('conc', 'fwd', 'rev', 'pres_mod', 'spec_rates', 'dydt', 'jac')


@pytest.mark.parametrize('name', ['h2o2_n2', 'h2o2', 'synth_alltypes', 'gri30_shaped', 'usc2_shaped', 'synth_mid24',
                                  'synth_srichb', 'synth_fracnu', 'synth_irrev72',
                                  # front-end corners: units keywords, separate thermo database (conftest.FRONT_END)
                                  'fe_kcal', 'fe_kelvins', 'fe_kjoules', 'fe_joules', 'fe_evolts', 'fe_septherm',
                                  # planner-geometry sweep (tests/golden/make_sweep_mechs.py): 54 and 121 species
                                  'sweep_n054', 'sweep_n121'])
def test_oracle_matches_reference_golden(name, golden, tables):
    g = golden(name)
    tab = tables(name)
    assert (tab.nsp, tab.nrxn, tab.nrev, tab.npres) == (int(g['nsp']), int(g['n_fwd']),
                                                        int(g['n_rev']), int(g['n_pres_mod']))
    o = Oracle(tab)
    for s in range(g['pres'].size):
        out = o.eval_all(float(g['pres'][s]), g['y'][s])
        for k in KEYS:
            mx, fro = thresholded_rel_err(out[k], g[k][s])
            # rates are bit-for-bit; the Jacobian differs by summation order only
            tol = 1e-13 if k != 'jac' else (5e-11 if tab.nsp <= 16 else 5e-9)
            assert mx <= tol, (name, s, k, mx)
            assert fro <= 1e-13


def test_oracle_writes_full_jacobian_block(tables):
    """pyJac needs a pre-zeroed jac (tester.c.in:27); the restatement must not."""
    tab = tables('h2o2_n2')
    o = Oracle(tab)
    import ctypes
    y = np.array([1500.0] + [0.1] * 9)
    jac = np.full(100, np.nan)
    o.lib.pjo_eval_jacob(o.h, 0.0, 101325.0, y.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                         jac.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert np.isfinite(jac).all()


@pytest.mark.parametrize('name', ['h2o2_n2', 'synth_alltypes', 'gri30_shaped', 'usc2_shaped', 'synth_mid24', 'synth_srichb',
                                  'synth_fracnu', 'synth_irrev72', 'sweep_n054', 'sweep_n121'])
def test_oracle_matches_reference_live(name, tables):
    if not Reference.available(name):
        pytest.skip('oracle/_ref not built (no /root/reference here)')
    tab = tables(name)
    o, r = Oracle(tab), Reference(name)
    rng = np.random.default_rng(11)
    n = 64 if tab.nsp <= 16 else 16
    T = rng.uniform(350, 3000, n)
    P = 101325 * 10 ** rng.uniform(-2, 2, n)
    Y = rng.uniform(0, 1, (n, tab.nsp)) ** 3 + 1e-9
    Y /= Y.sum(axis=1, keepdims=True)
    y = np.concatenate([T[:, None], Y[:, :-1]], axis=1)
    a, b = o.batch_jacob(P, y), r.batch_jacob(P, y)
    mx, fro = thresholded_rel_err(a, b)
    assert mx < (1e-9 if tab.nsp <= 16 else 1e-7) and fro < 1e-12
    mx, fro = thresholded_rel_err(o.batch_dydt(P, y), r.batch_dydt(P, y))
    assert mx < 1e-12


def test_jacobian_consistent_with_finite_differences(tables):
    """Species rows of the analytical Jacobian vs central differences of the
    oracle's own dydt (the reference's tester uses autodiff)."""
    tab = tables('h2o2_n2')
    o = Oracle(tab)
    rng = np.random.default_rng(3)
    Y = rng.uniform(0.05, 1, 10)
    Y /= Y.sum()
    y = np.concatenate([[1400.0], Y[:-1]])
    P = 2 * 101325.0
    jac = o.eval_all(P, y)['jac'].reshape(10, 10).T
    for j in range(1, 10):
        h = 1e-6 * max(abs(y[j]), 1e-3)
        yp, ym = y.copy(), y.copy()
        yp[j] += h
        ym[j] -= h
        fd = (o.eval_all(P, yp)['dydt'] - o.eval_all(P, ym)['dydt']) / (2 * h)
        scale = np.abs(jac[1:, j]).max()
        assert np.abs(fd[1:] - jac[1:, j]).max() < 1e-5 * scale


@pytest.mark.parametrize('name', ['h2o2_n2', 'synth_alltypes', 'synth_srichb'])
def test_fd_arm_oracle_pinned_to_reference(name, tables):
    """N3 pin: the restated finite-difference Jacobian (pyjac/performance_tester/fd_jacob.c:10-113)
    is bit-identical to the reference's own fd_jacob.c compiled into oracle/_ref."""
    if not Reference.available(name):
        pytest.skip('oracle/_ref not built (no /root/reference here)')
    tab = tables(name)
    o, r = Oracle(tab), Reference(name)
    rng = np.random.default_rng(5)
    for _ in range(50):
        T = rng.uniform(500, 2800)
        P = 101325 * 10 ** rng.uniform(-1.5, 1.5)
        Y = rng.uniform(0, 1, tab.nsp) ** 2 + 1e-7
        Y /= Y.sum()
        y = np.concatenate([[T], Y[:-1]])
        assert np.array_equal(o.fd_jacob(P, y.copy()), r.fd_jacob(P, y.copy()), equal_nan=True)
