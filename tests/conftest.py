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
}


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
            cache[name] = build_tables(read_mech(MECHS[name]))
        return cache[name]
    return get
