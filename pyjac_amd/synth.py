"""Seeded synthetic state batches.

Dist-A "PaSR-tiled": rows of the reference's PaSR fixture
(data/h2_pasr_output.npy, columns t, T, P, Y x 10; layout
functional_tester/partially_stirred_reactor.py:715-742) drawn with replacement and
perturbed; Dist-B "uniform": the docs' recipe (docs/examples.rst:198-208).
States are returned in pyJac's batch (SoA) layout: y[(NSP), n] = [T; Y_0..Y_{NSP-2}].
"""
from __future__ import annotations

import os

import numpy as np

SEED = 20240901
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def pasr_states(nsp: int = 10):
    """The 1020 PaSR states, normalised as the functional tester does
    (functional_tester/test.py:1254-1258).  Returns (pres[n], Y[n, nsp], T[n])."""
    a = np.load(os.path.join(_DATA, 'h2_pasr_output.npy')).reshape(-1, 13)
    T, P, Y = a[:, 1].copy(), a[:, 2].copy(), a[:, 3:3 + nsp].copy()
    Y /= Y.sum(axis=1, keepdims=True)
    Y[:, -1] = 1.0 - Y[:, :-1].sum(axis=1)
    return P, Y, T


def dist_a(n: int, nsp: int = 10, seed: int = SEED):
    rng = np.random.default_rng(seed)
    P0, Y0, T0 = pasr_states(nsp)
    idx = rng.integers(0, T0.size, n)
    Y = Y0[idx] * np.exp(0.05 * rng.standard_normal((n, nsp)))
    Y /= Y.sum(axis=1, keepdims=True)
    T = T0[idx] * (1.0 + 0.02 * rng.uniform(-1, 1, n))
    P = 101325.0 * 10.0 ** rng.uniform(-0.3, 1.4, n)
    return P, _pack(T, Y)


def dist_b(n: int, nsp: int, seed: int = SEED, Tlo: float = 800.0, Thi: float = 2500.0):
    rng = np.random.default_rng(seed)
    T = rng.uniform(Tlo, Thi, n)
    P = np.maximum(rng.uniform(0.0, 25.0, n), 0.05) * 101325.0
    Y = rng.uniform(0.0, 1.0, (n, nsp))
    Y /= Y.sum(axis=1, keepdims=True)
    return P, _pack(T, Y)


def _pack(T, Y):
    """SoA state array (NSP, n): row 0 = T, rows 1.. = Y_0..Y_{NSP-2}."""
    n, nsp = Y.shape
    y = np.empty((nsp, n))
    y[0] = T
    y[1:] = Y[:, :-1].T
    return np.ascontiguousarray(y)
