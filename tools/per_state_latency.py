#!/usr/bin/env python3
"""Latency of the per-state host API (pyjacob.py_* through pj_dydt / pj_eval_jacob: blocking copies + one
launch per call) -- the price of keeping pyJac's per-state C call signature on a GPU path."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pyjac_amd import pyjacob
for mech in ('pyjac_amd/data/h2o2_n2.inp', 'pyjac_amd/data/gri30_shaped.inp'):
    ev = pyjacob.use_mechanism(os.path.join(ROOT, mech))
    pyjacob.cache_states(False)
    nsp = ev.nsp
    y = np.concatenate([[1500.0], np.full(nsp - 1, 1.0 / nsp)])
    dy, jac = np.zeros(nsp + 1), np.zeros(nsp * nsp)
    yy = np.concatenate([y, [0.0]])
    for _ in range(20):
        pyjacob.py_dydt(0.0, 101325.0, yy, dy); pyjacob.py_eval_jacobian(0.0, 101325.0, y, jac)
    t0 = time.perf_counter()
    for _ in range(300):
        pyjacob.py_dydt(0.0, 101325.0, yy, dy)
    t1 = time.perf_counter()
    for _ in range(300):
        pyjacob.py_eval_jacobian(0.0, 101325.0, y, jac)
    t2 = time.perf_counter()
    pyjacob.cache_states(True)
    conc, fwd, rev, pm, sr = np.zeros(nsp), np.zeros(ev.n_fwd), np.zeros(max(ev.n_rev, 1)), np.zeros(max(ev.n_pres_mod, 1)), np.zeros(nsp)
    def seq(i):
        yy[0] = y[0] = 1500.0 + i          # a new state every time: one evaluation + five cache hits
        mf = np.concatenate([y[1:], [0.0]])
        pyjacob.py_eval_conc(y[0], 101325.0, mf, 0.0, 0.0, conc)
        pyjacob.py_eval_rxn_rates(y[0], 101325.0, conc, fwd, rev)
        pyjacob.py_get_rxn_pres_mod(y[0], 101325.0, conc, pm)
        pyjacob.py_eval_spec_rates(fwd, rev, pm, sr)
        pyjacob.py_dydt(0.0, 101325.0, yy, dy)
        pyjacob.py_eval_jacobian(0.0, 101325.0, y, jac)
    for mode in (True, False):
        pyjacob.cache_states(mode)
        for i in range(10):
            seq(i)
        t3 = time.perf_counter()
        for i in range(100):
            seq(100 + i)
        print('%s: six-call sequence of the functional tester, one-state cache %s: %.1f us per state' % (
            os.path.basename(mech), 'on' if mode else 'off', (time.perf_counter() - t3) / 100 * 1e6))
    pyjacob.cache_states(False)
    print('%s: py_dydt %.1f us/call, py_eval_jacobian %.1f us/call' % (os.path.basename(mech), (t1 - t0) / 300 * 1e6, (t2 - t1) / 300 * 1e6))
