"""Mechanism -> flat table blob ("mechanism as data").

pyJac prints a mechanism into unrolled source (pyjac/core/rate_subs.py,
pyjac/core/create_jacobian.py); here the same numbers are laid out as two flat
arrays (int32 ``I`` and float64 ``D``) that cross the C-ABI once at
mechanism-load time (include/pyjac_amd.h: ``pj_mech_create``) and are also what
the CPU oracle (oracle/pyjac_oracle.c) consumes.  Every value that the
reference computes in Python at generation time and prints with 17
significant digits is computed here with the same arithmetic in the same
order, so the tables hold bit-identical constants:

  * per-reaction pre-summed equilibrium-constant polynomials grouped by the
    species' T_mid, and the (PA/RU)^sum(nu) prefactor   rate_subs.py:540-558, 660-809
  * low/high-pressure ratio parameters (get_infs)        create_jacobian.py:622-655
  * Troe parameters as ``get_rxn_pres_mod`` prints them ('%.8e')  rate_subs.py:1187-1211
  * falloff beta difference as the Jacobian prints it ('%.4e')    create_jacobian.py:1167
  * PLOG pressure breakpoints as printed ('%.4e')        rate_subs.py:601-629
  * SRI parameters as each emitter prints them ('{:.6}' in get_rxn_pres_mod and the F_i factor,
    '{:.4}' in the dPr/dY_j term, '{:.16}' in the d/dT term)
                                                         rate_subs.py:1229-1256, create_jacobian.py:173-179, 249-266, 1194-1237
  * Chebyshev coefficients and reduced-variable constants as printed ('{:.8e}' in the rate,
    '{:.16e}' in the d/dT sum)                           rate_subs.py:149-251, create_jacobian.py:1532-1684

Blob layout (version 1).  ``I[0:HDR]`` is a header; ``I[16+j]`` is the offset in
``I`` of int array j, ``I[48+j]`` the offset in ``D`` of double array j.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .mechanism import Mechanism, PA, RU, get_nu

MAGIC = 0x314D4A50
VERSION = 2
HDR = 96

# reaction flag bits
F_REV, F_THD, F_PDEP, F_LOW, F_HIGH = 1, 2, 4, 8, 16
F_TROE, F_SRI, F_PLOG, F_TROE4, F_SRI5, F_HAS_EFF = 32, 64, 128, 256, 512, 1024
F_CHEB = 32768      # (2048 .. 16384 are derived by the device-table builder, csrc/pj_tables.h)

# int arrays
(IA_FLAGS, IA_REAC_PTR, IA_REAC_SP, IA_PROD_PTR, IA_PROD_SP, IA_NET_PTR,
 IA_NET_SP, IA_EFF_PTR, IA_EFF_SP, IA_PLOG_PTR, IA_KC_PTR, IA_PDEP_SP,
 IA_REV_IDX, IA_PRES_IDX, IA_SEEN, IA_CHEB_PTR) = range(16)
# double arrays
(DA_MW, DA_TMID, DA_LO, DA_HI, DA_A, DA_B, DA_E, DA_REAC_NU, DA_PROD_NU,
 DA_NET_NU, DA_EFF, DA_PD, DA_TROE, DA_SRI, DA_PLOG, DA_KCG, DA_KCPREF,
 DA_INFS, DA_TROE8, DA_PLOG4, DA_SRIQ, DA_CHEB) = range(22)
SRIQ_W = 16         # doubles per reaction in DA_SRIQ


def _r(fmt: str, x: float) -> float:
    """Value after a print/parse round trip with the reference's format."""
    return float(fmt.format(x))


@dataclass
class MechTables:
    I: np.ndarray
    D: np.ndarray
    nsp: int
    nrxn: int
    nrev: int
    npres: int
    species: list

    def save(self, path: str):
        """On-disk table file: little-endian [u64 nI][u64 nD][I int32][D f64]
        (readable by ``pj_mech_load``, include/pyjac_amd.h)."""
        with open(path, 'wb') as f:
            f.write(np.array([self.I.size, self.D.size], dtype='<u8').tobytes())
            f.write(self.I.astype('<i4').tobytes())
            f.write(self.D.astype('<f8').tobytes())

    @staticmethod
    def load(path: str) -> 'MechTables':
        with open(path, 'rb') as f:
            n = np.frombuffer(f.read(16), dtype='<u8')
            I = np.frombuffer(f.read(int(n[0]) * 4), dtype='<i4').copy()
            D = np.frombuffer(f.read(int(n[1]) * 8), dtype='<f8').copy()
        return MechTables(I, D, int(I[2]), int(I[3]), int(I[4]), int(I[5]), [])


def _kc_groups(mech: Mechanism, rx):
    """Pre-summed NASA coefficient groups for ln Kc (rate_subs.py:540-558, 660-809)."""
    specs = mech.specs

    def arrays(sp, nu, factor):
        def one(a):
            arr = [nu * factor, a[6], a[0], a[0] - 1.0, a[1] / 2.0, a[2] / 6.0,
                   a[3] / 12.0, a[4] / 20.0, a[5]]
            return [x * arr[0] for x in [arr[1] - arr[2]] + arr[3:]]
        return one(sp.lo), one(sp.hi)

    coeffs = {}
    sum_nu = 0

    def acc(sp, lo, hi):
        t = sp.Trange[1]
        if t not in coeffs:
            coeffs[t] = (lo, hi)
        else:
            coeffs[t] = ([lo[i] + coeffs[t][0][i] for i in range(7)],
                         [hi[i] + coeffs[t][1][i] for i in range(7)])

    for ip, psp in enumerate(rx.prod):
        if psp in rx.reac:
            nu = rx.prod_nu[ip] - rx.reac_nu[rx.reac.index(psp)]
        else:
            nu = rx.prod_nu[ip]
        if nu == 0:
            continue
        sum_nu += nu
        lo, hi = arrays(specs[psp], nu, 1.0)
        acc(specs[psp], lo, hi)
    for ir, rsp in enumerate(rx.reac):
        if rsp in rx.prod:
            continue
        nu = rx.reac_nu[ir]
        sum_nu -= nu
        lo, hi = arrays(specs[rsp], nu, -1.0)
        acc(specs[rsp], lo, hi)
    groups = [(t, lo, hi) for t, (lo, hi) in coeffs.items()]
    return groups, (PA / RU) ** sum_nu


def build_tables(mech: Mechanism) -> MechTables:
    nsp, nrxn = mech.nsp, len(mech.reacs)
    ia = [[] for _ in range(16)]
    da = [[] for _ in range(22)]

    for sp in mech.specs:
        da[DA_MW].append(sp.mw)
        da[DA_TMID].append(sp.Trange[1])
        da[DA_LO] += list(sp.lo)
        da[DA_HI] += list(sp.hi)

    seen = [0] * nsp
    rev_i = pres_i = 0
    for p in (IA_REAC_PTR, IA_PROD_PTR, IA_NET_PTR, IA_EFF_PTR, IA_PLOG_PTR, IA_KC_PTR, IA_CHEB_PTR):
        ia[p].append(0)
    for rx in mech.reacs:
        fl = 0
        if rx.rev:
            fl |= F_REV
        if rx.thd_body:
            fl |= F_THD
        if rx.pdep:
            fl |= F_PDEP
            if rx.low:
                fl |= F_LOW
            elif rx.high:
                fl |= F_HIGH
            else:
                raise ValueError('falloff reaction without LOW or HIGH parameters')
        if rx.troe:
            fl |= F_TROE
            if len(rx.troe_par) == 4 and rx.troe_par[3] != 0.0:
                fl |= F_TROE4
        if rx.sri:
            fl |= F_SRI
            if len(rx.sri_par) == 5:
                fl |= F_SRI5
        if rx.plog:
            fl |= F_PLOG
        if rx.cheb:
            fl |= F_CHEB
            if rx.plog or rx.pdep or rx.thd_body:
                raise ValueError('Chebyshev reaction combined with another pressure dependence')
        if rx.thd_body_eff:
            fl |= F_HAS_EFF
        ia[IA_FLAGS].append(fl)
        da[DA_A].append(rx.A)
        da[DA_B].append(rx.b)
        da[DA_E].append(rx.E)

        ia[IA_REAC_SP] += list(rx.reac)
        da[DA_REAC_NU] += [float(n) for n in rx.reac_nu]
        ia[IA_REAC_PTR].append(len(ia[IA_REAC_SP]))
        ia[IA_PROD_SP] += list(rx.prod)
        da[DA_PROD_NU] += [float(n) for n in rx.prod_nu]
        ia[IA_PROD_PTR].append(len(ia[IA_PROD_SP]))

        # net production list in the order eval_spec_rates visits species
        # (rate_subs.py:1396-1420: sorted set of participants)
        for k in sorted(set(rx.reac + rx.prod)):
            nu = get_nu(k, rx)
            if nu == 0:
                continue
            ia[IA_NET_SP].append(k)
            da[DA_NET_NU].append(float(nu))
            seen[k] = 1
        ia[IA_NET_PTR].append(len(ia[IA_NET_SP]))

        for k, a in rx.thd_body_eff:
            ia[IA_EFF_SP].append(k)
            da[DA_EFF].append(a)
        ia[IA_EFF_PTR].append(len(ia[IA_EFF_SP]))

        ia[IA_PDEP_SP].append(-1 if rx.pdep_sp is None else rx.pdep_sp)
        ia[IA_REV_IDX].append(rev_i if rx.rev else -1)
        rev_i += 1 if rx.rev else 0
        has_pm = rx.thd_body or rx.pdep
        ia[IA_PRES_IDX].append(pres_i if has_pm else -1)
        pres_i += 1 if has_pm else 0

        pd = rx.low if rx.low else (rx.high if rx.high else [0.0, 0.0, 0.0])
        da[DA_PD] += list(pd)
        tro = list(rx.troe_par) + [0.0] * (4 - len(rx.troe_par))
        da[DA_TROE] += tro[:4]
        sri = list(rx.sri_par) + [0.0] * (5 - len(rx.sri_par))
        da[DA_SRI] += sri[:5]
        if rx.sri:
            a, b, c = rx.sri_par[0], rx.sri_par[1], rx.sri_par[2]
            five = len(rx.sri_par) == 5
            d, e = (rx.sri_par[3], rx.sri_par[4]) if five else (1.0, 0.0)
            da[DA_SRIQ] += [_r('{:.6}', a), _r('{:.6}', b), _r('{:.6}', c), _r('{:.8e}', d), _r('{:.6}', e),
                            1.0 if (five and d != 1.0 and e != 0.0) else 0.0,
                            _r('{:.4}', a), _r('{:.4}', b), _r('{:.4}', c),
                            _r('{:.16}', a), _r('{:.16}', b), _r('{:.16}', c), _r('{:.16}', a * b),
                            _r('{:.16e}', 1.0 / c), _r('{:.16}', e) if (five and e != 0.0) else 0.0, 0.0]
        else:
            da[DA_SRIQ] += [0.0] * SRIQ_W
        if rx.cheb:
            n, m = rx.cheb_n_temp, rx.cheb_n_pres
            par = np.reshape(np.array(rx.cheb_par, dtype=float), (n, m))
            tsum = 1.0 / rx.cheb_tlim[0] + 1.0 / rx.cheb_tlim[1]
            tsub = 1.0 / rx.cheb_tlim[1] - 1.0 / rx.cheb_tlim[0]
            psum = math.log10(rx.cheb_plim[0]) + math.log10(rx.cheb_plim[1])
            psub = math.log10(rx.cheb_plim[1]) - math.log10(rx.cheb_plim[0])
            rec = [float(n), float(m), _r('{:.8e}', tsum), _r('{:.8e}', tsub), _r('{:.8e}', psum), _r('{:.8e}', psub),
                   _r('{:.16e}', tsum), _r('{:.16e}', tsub), _r('{:.16e}', psum), _r('{:.16e}', psub),
                   _r('{:.16e}', -2.0 * math.log(10) / tsub)]
            rec += [_r('{:.8e}', par[i, j]) for i in range(n) for j in range(m)]
            rec += [_r('{:.16e}', i * par[i, j]) for i in range(1, n) for j in range(m)]
            da[DA_CHEB] += rec
        ia[IA_CHEB_PTR].append(len(da[DA_CHEB]))
        if rx.troe:
            a, T3, T1 = rx.troe_par[0], rx.troe_par[1], rx.troe_par[2]
            T2 = tro[3]
            da[DA_TROE8] += [_r('{:.8e}', 1.0 - a), _r('{:.8e}', a),
                             _r('{:.8e}', abs(T3)) * (1 if T3 > 0 else -1),
                             _r('{:.8e}', abs(T1)) * (1 if T1 > 0 else -1),
                             _r('{:.8e}', abs(T2)) * (1 if T2 > 0 else -1)]
        else:
            da[DA_TROE8] += [0.0] * 5

        # get_infs (create_jacobian.py:622-655)
        if rx.pdep and rx.low:
            b0 = rx.low[1] - rx.b
            e0 = rx.low[2] - rx.E
            ar = rx.low[0] / rx.A
        elif rx.pdep and rx.high:
            b0 = rx.b - rx.high[1]
            e0 = rx.E - rx.high[2]
            ar = rx.A / rx.high[0]
        else:
            b0 = e0 = ar = 0.0
        da[DA_INFS] += [ar, b0, e0, _r('{:.4e}', b0)]

        if rx.plog:
            for pars in rx.plog_par:
                da[DA_PLOG] += list(pars)
                da[DA_PLOG4].append(_r('{:.4e}', pars[0]))
        ia[IA_PLOG_PTR].append(len(da[DA_PLOG4]))

        if rx.rev:
            groups, pref = _kc_groups(mech, rx)
            for t, lo, hi in groups:
                da[DA_KCG] += [t] + lo + hi
            da[DA_KCPREF].append(pref)
        else:
            da[DA_KCPREF].append(1.0)
        ia[IA_KC_PTR].append(len(da[DA_KCG]) // 15)

    ia[IA_SEEN] = seen

    I = [0] * HDR
    off = HDR
    for j, arr in enumerate(ia):
        I[16 + j] = off
        off += len(arr)
    body = []
    for arr in ia:
        body += [int(x) for x in arr]
    D = []
    for j, arr in enumerate(da):
        I[48 + j] = len(D)
        D += [float(x) for x in arr]
    I[0:14] = [MAGIC, VERSION, nsp, nrxn, rev_i, pres_i,
               len(ia[IA_REAC_SP]), len(ia[IA_PROD_SP]), len(ia[IA_EFF_SP]),
               len(da[DA_PLOG4]), len(da[DA_KCG]) // 15, len(ia[IA_NET_SP]),
               HDR + len(body), len(D)]
    Iarr = np.array(I + body, dtype=np.int32)
    Darr = np.array(D, dtype=np.float64)
    return MechTables(Iarr, Darr, nsp, nrxn, rev_i, pres_i, mech.species_names())
