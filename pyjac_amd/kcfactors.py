"""Equilibrium constants from per-species factors (host side of the `PJQ_KCF` row kernels, csrc/pj_rblk.hip).

pyJac evaluates ``K_c,i = (p_atm / R_u T)^{sum nu} exp(sum_k nu_ki (s_k/R - h_k/RT))`` per reaction from
pre-summed NASA polynomials: one 7-term polynomial and one ``exp`` per reversible reaction
(pyjac/core/rate_subs.py:660-809; ∂/∂T: create_jacobian.py:492-619).  A row-block kernel visits a reaction once per
species row it touches (3.6 times on average), so it pays that polynomial and that ``exp`` 3.6 times.  The same
number is a product over the reaction's species,

    1 / K_c,i = (p_atm / R_u)^{-sum nu} prod_k X_k^{-nu_ki},     X_k = exp(s_k/R - h_k/RT - ln T),

and ``X_k`` is a function of the STATE, not of the reaction: NSP exponentials per state instead of one per visit.
``X_k`` itself spans e^{+-200} at flame temperatures (products of three overflow), but any shift
``ln X_k -> ln X_k - sigma_k(T)`` with ``sum_k nu_ki sigma_k = 0`` for every reversible reaction cancels in every
product.  The vectors sigma with that property are the left null space of the stoichiometric matrix -- the element
compositions of an atom-conserving mechanism, and whatever else the mechanism conserves; it is taken from the
stoichiometry itself (SVD), so no composition data is needed and a mechanism that conserves nothing simply gets no
shift.  ``sigma_k(T) = sum_d B_kd lambda_d(T)`` with ``lambda_d`` least-squares fitted in the NASA basis
{1, ln T, T, T^2, T^3, T^4, -1/T}: the shifted ``ln X_k`` is again a 7-coefficient row per temperature range.

``kc_factor_rows`` returns those rows, or None when the scheme cannot be used for the mechanism (a fractional or
large stoichiometric coefficient on a reversible reaction, or factors whose products could overflow): the kernels
then keep the per-reaction polynomial form.
"""
from __future__ import annotations

import numpy as np

from . import tables as T

LNX_LIMIT = 200.0       # |shifted ln X_k| over the temperature window: three factors multiply before any divides
T_WINDOW = (200.0, 6000.0)
PARTIAL_LIMIT = 650.0   # |ln| of any partial product of a reaction's factors (double range: e^+-708)


def _basis(Tt):
    Tt = np.asarray(Tt, dtype=np.float64)
    return np.stack([np.ones_like(Tt), np.log(Tt), Tt, Tt ** 2, Tt ** 3, Tt ** 4, -1.0 / Tt], axis=-1)


def species_rows(tab):
    """Unshifted rows of ln X_k = s_k/R - h_k/RT - ln T in the K_c polynomial form of rate_subs.py:540-558:
    (T_mid[k], lo[k][7], hi[k][7]) with ln X = b0 + b1 ln T + b2 T + b3 T^2 + b4 T^3 + b5 T^4 - b6 / T."""
    I, D = tab.I, tab.D
    nsp = tab.nsp
    da = lambda j, cnt: D[I[48 + j]:I[48 + j] + cnt]
    tmid = da(T.DA_TMID, nsp).copy()

    def row(a):
        return np.stack([a[:, 6] - a[:, 0], a[:, 0] - 1.0, a[:, 1] / 2.0, a[:, 2] / 6.0, a[:, 3] / 12.0, a[:, 4] / 20.0,
                         a[:, 5]], axis=1)
    return tmid, row(da(T.DA_LO, 7 * nsp).reshape(nsp, 7)), row(da(T.DA_HI, 7 * nsp).reshape(nsp, 7))


def net_matrix(tab):
    """Net stoichiometric coefficients of the reversible reactions, (NSP, n_rev)."""
    I, D = tab.I, tab.D
    nrxn = tab.nrxn
    flags = I[I[16 + T.IA_FLAGS]:I[16 + T.IA_FLAGS] + nrxn]
    ptr = I[I[16 + T.IA_NET_PTR]:I[16 + T.IA_NET_PTR] + nrxn + 1]
    sp = I[I[16 + T.IA_NET_SP]:I[16 + T.IA_NET_SP] + int(ptr[-1])]
    nu = D[I[48 + T.DA_NET_NU]:I[48 + T.DA_NET_NU] + int(ptr[-1])]
    cols = []
    for i in range(nrxn):
        if not (int(flags[i]) & T.F_REV):
            continue
        c = np.zeros(tab.nsp)
        for q in range(int(ptr[i]), int(ptr[i + 1])):
            c[int(sp[q])] += nu[q]
        cols.append(c)
    return np.array(cols).T if cols else np.zeros((tab.nsp, 0))


def kc_factor_rows(tab, max_nu: int = 4):
    """(NSP, 15) array [T_mid, lo0..lo6, hi0..hi6] of the SHIFTED rows of ln X_k, or None (see module docstring).
    Species that take part in no reversible reaction get zero rows (X_k = 1, never read)."""
    nsp = tab.nsp
    N = net_matrix(tab)
    if N.shape[1] == 0:
        return None
    if np.any(N != np.round(N)) or np.abs(N).max() > max_nu:
        return None
    tmid, lo, hi = species_rows(tab)
    used = np.abs(N).sum(axis=1) > 0
    # left null space of N: sigma with sigma^T N = 0
    u, s, _ = np.linalg.svd(N, full_matrices=True)
    rank = int((s > 1e-10 * max(s.max(), 1.0)).sum())
    B = u[:, rank:]                                        # (NSP, d), orthonormal
    # ln X_k on a temperature grid, each species with its own range select
    Tg = np.exp(np.linspace(np.log(250.0), np.log(4500.0), 96))
    phi = _basis(Tg)                                       # (nt, 7)
    F = np.where(Tg[None, :] <= tmid[:, None], lo @ phi.T, hi @ phi.T)      # (NSP, nt)
    F[~used] = 0.0
    shift_lo, shift_hi = lo.copy(), hi.copy()
    if B.shape[1]:
        # species outside every reversible reaction carry no information about the shift
        Bu = B.copy()
        Bu[~used] = 0.0
        G, *_ = np.linalg.lstsq(Bu, F, rcond=None)         # lambda_d(T_t): (d, nt)
        scale = np.abs(phi).max(axis=0)
        Lc, *_ = np.linalg.lstsq(phi / scale, G.T, rcond=None)             # (7, d)
        Lc = Lc / scale[:, None]
        sh = B @ Lc.T                                      # (NSP, 7): coefficients of sigma_k(T)
        shift_lo, shift_hi = lo - sh, hi - sh
    shift_lo[~used] = 0.0
    shift_hi[~used] = 0.0
    # overflow guard over the whole window
    Tw = np.exp(np.linspace(np.log(T_WINDOW[0]), np.log(T_WINDOW[1]), 400))
    pw = _basis(Tw)
    Fw = np.where(Tw[None, :] <= tmid[:, None], shift_lo @ pw.T, shift_hi @ pw.T)
    if not np.isfinite(Fw).all() or np.abs(Fw).max() > LNX_LIMIT:
        return None
    # ... and of what the kernel actually forms: 1 / K_c,i = (p_atm / R_u)^(-sum nu) * prod X_k^(-nu_ki), multiplied up
    # factor by factor in the order of the reaction's net species, |nu| times each (pj_rblk.hip) -- no partial product may
    # leave the double range (overflow: inf * 0 = NaN further on; underflow: digits lost without a trace)
    I, D = tab.I, tab.D
    flags = I[I[16 + T.IA_FLAGS]:I[16 + T.IA_FLAGS] + tab.nrxn]
    ptr = I[I[16 + T.IA_NET_PTR]:I[16 + T.IA_NET_PTR] + tab.nrxn + 1]
    nsp_ = I[I[16 + T.IA_NET_SP]:I[16 + T.IA_NET_SP] + int(ptr[-1])]
    nnu = D[I[48 + T.DA_NET_NU]:I[48 + T.DA_NET_NU] + int(ptr[-1])]
    pref = D[I[48 + T.DA_KCPREF]:I[48 + T.DA_KCPREF] + tab.nrxn]
    for i in range(tab.nrxn):
        if not (int(flags[i]) & T.F_REV):
            continue
        part = np.full(Tw.shape, -np.log(pref[i]))
        worst = np.abs(part).max()
        for q in range(int(ptr[i]), int(ptr[i + 1])):
            for _ in range(int(abs(nnu[q]))):
                part = part - np.sign(nnu[q]) * Fw[int(nsp_[q])]
                worst = max(worst, np.abs(part).max())
        if worst > PARTIAL_LIMIT:
            return None
    # the shift must cancel in every reversible reaction (it does by construction; guards the SVD threshold)
    resid = np.abs((lo - shift_lo)[used].T @ N[used]).max() if used.any() else 0.0
    scale_c = np.abs(lo - shift_lo).max() + 1e-300
    if resid > 1e-9 * scale_c * max(1.0, np.abs(N).sum(axis=0).max()):
        return None
    return np.ascontiguousarray(np.concatenate([tmid[:, None], shift_lo, shift_hi], axis=1))
