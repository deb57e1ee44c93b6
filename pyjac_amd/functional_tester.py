"""Functional-test harness: the caller contract of pyJac's tester on the HIP path.

Reproduces, against a caller-supplied reference evaluator, what
pyjac/functional_tester/test.py does per state (call order and array sizes
``:1282-1327``, state normalisation ``:1254-1258``, species permutation
``:334-430``) and its error statistics (``:1429-1472``, summary ``:1582-1587``).
The reference's comparison arms (Cantera, Adept, TChem) are not available here;
any object exposing the six ``py_*`` functions can serve as the reference
(pyJac's own compiled ``pyjacob`` module, or -- in this repo's tests -- the CPU
oracle).  Nothing in this module imports that reference.
"""
from __future__ import annotations

import numpy as np


def normalise_states(Y: np.ndarray) -> np.ndarray:
    """test.py:1254-1258: divide by the sum, recompute the last species."""
    Y = Y / Y.sum(axis=1, keepdims=True)
    Y[:, -1] = 1.0 - Y[:, :-1].sum(axis=1)
    return Y


def jacobian_error_metrics(test_jacob: np.ndarray, jacob: np.ndarray) -> dict:
    """The statistics test.py:1429-1472 prints for one state (fractions, not %)."""
    out = {}
    nz = np.where(np.abs(test_jacob) > 1.e-30)[0]
    zero = np.where(test_jacob == 0.)[0]
    with np.errstate(divide='ignore', invalid='ignore'):
        err = np.abs((test_jacob[nz] - jacob[nz]) / jacob[nz])
    err = err[np.isfinite(err)]
    out['max_rel'] = float(err.max()) if err.size else 0.0
    out['l2_rel'] = float(np.linalg.norm(err))
    thr = np.where(np.abs(test_jacob) > np.linalg.norm(test_jacob) / 1.e20)[0]
    with np.errstate(divide='ignore', invalid='ignore'):
        e2 = np.abs((test_jacob[thr] - jacob[thr]) / jacob[thr])
    e2 = e2[np.isfinite(e2)]
    out['thr_max_rel'] = float(e2.max()) if e2.size else 0.0
    out['thr_l2_rel'] = float(np.linalg.norm(e2))
    out['norm_err'] = float(np.linalg.norm(test_jacob - jacob) / np.linalg.norm(jacob))
    out['zero_diff'] = float(np.linalg.norm(test_jacob[zero] - jacob[zero]))
    return out


def evaluate_state(mod, nsp, n_fwd, n_rev, n_pres_mod, T, P, Y):
    """One state through the per-state API in the tester's order (test.py:1299-1327).
    ``Y``: all NSP mass fractions in the module's internal species order."""
    mass_frac = np.array(Y, dtype=np.float64)
    conc = np.zeros(nsp)
    mod.py_eval_conc(T, P, mass_frac, 0.0, 0.0, conc)
    fwd = np.zeros(n_fwd)
    rev = np.zeros(max(n_rev, 1))
    mod.py_eval_rxn_rates(T, P, conc, fwd, rev)
    pm = np.zeros(max(n_pres_mod, 1))
    if n_pres_mod:
        mod.py_get_rxn_pres_mod(T, P, conc, pm)
    sr = np.zeros(nsp)
    mod.py_eval_spec_rates(fwd, rev, pm, sr)
    y = np.hstack((T, np.asarray(Y, dtype=np.float64)))     # tester passes NSP+1 entries
    dydt = np.zeros(nsp + 1)
    mod.py_dydt(0.0, P, y, dydt)
    jac = np.zeros(nsp * nsp)
    mod.py_eval_jacobian(0.0, P, y, jac)
    return dict(conc=conc, fwd=fwd, rev=rev, pres_mod=pm, spec_rates=sr, dydt=dydt[:nsp], jac=jac)


def run(test_mod, ref_mod, sizes, T, P, Y, fwd_spec_map=None, verbose=False):
    """Compare two evaluators over a set of states.

    sizes = (nsp, n_fwd, n_rev, n_pres_mod); Y (n, NSP) in the MECHANISM order;
    ``fwd_spec_map`` moves the last species to the end (Mechanism.fwd_spec_map),
    as the tester does for its inputs.  Returns per-state metric arrays and the
    tester's headline statistic."""
    nsp = sizes[0]
    Y = normalise_states(np.array(Y, dtype=np.float64))
    if fwd_spec_map is not None:
        Y = Y[:, fwd_spec_map]
    keys = ('max_rel', 'l2_rel', 'thr_max_rel', 'thr_l2_rel', 'norm_err', 'zero_diff')
    stats = {k: np.zeros(len(T)) for k in keys}
    for i in range(len(T)):
        a = evaluate_state(test_mod, *sizes, float(T[i]), float(P[i]), Y[i])
        b = evaluate_state(ref_mod, *sizes, float(T[i]), float(P[i]), Y[i])
        m = jacobian_error_metrics(a['jac'], b['jac'])
        for k in keys:
            stats[k][i] = m[k]
        if verbose:
            print('state %d: thresholded L2 rel err %.2e, max %.2e, norm err %.2e'
                  % (i, m['thr_l2_rel'], m['thr_max_rel'], m['norm_err']))
    stats['max_thr_l2_rel'] = float(stats['thr_l2_rel'].max())       # test.py:1582-1584
    stats['std_thr_l2_rel'] = float(stats['thr_l2_rel'].std())
    return stats
