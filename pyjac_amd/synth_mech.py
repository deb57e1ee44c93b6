"""Seeded generator of mechanism-shaped Chemkin files.

GRI-Mech 3.0 and USC-Mech II are named by BASELINE.json but are not in the
reference tree nor in this container (SURVEY.md "five facts" #2), so configs
3-5 run on synthetic mechanisms with the same *shape*: species / reaction
counts and a GRI-like mix of reaction types (Troe / Lindemann falloff,
third-body reactions with 5-7 enhanced colliders, duplicates, irreversible
steps, PLOG for the USC-shaped case).  Species names and element compositions
of the 53-species case are GRI-3.0's; NASA coefficients other than the
H2/O2/N2/AR cards and all rate parameters are synthetic.  Reactions are
element-balanced where a balancing product pair exists so equilibrium
constants stay moderate.  pyJac's generator accepts these files, so parity is
pinned against the reference exactly as for H2/O2 -- but results are for
"GRI-shaped synthetic", not for GRI-Mech 3.0.
"""
from __future__ import annotations

import itertools
import os

import numpy as np

from .mechanism import ELEM_WT, parse_mech

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')

# GRI-Mech 3.0 species set (names + compositions C, H, O, N, AR)
GRI_SPECIES = [
    ('H2', 0, 2, 0, 0, 0), ('H', 0, 1, 0, 0, 0), ('O', 0, 0, 1, 0, 0), ('O2', 0, 0, 2, 0, 0),
    ('OH', 0, 1, 1, 0, 0), ('H2O', 0, 2, 1, 0, 0), ('HO2', 0, 1, 2, 0, 0), ('H2O2', 0, 2, 2, 0, 0),
    ('C', 1, 0, 0, 0, 0), ('CH', 1, 1, 0, 0, 0), ('CH2', 1, 2, 0, 0, 0), ('CH2(S)', 1, 2, 0, 0, 0),
    ('CH3', 1, 3, 0, 0, 0), ('CH4', 1, 4, 0, 0, 0), ('CO', 1, 0, 1, 0, 0), ('CO2', 1, 0, 2, 0, 0),
    ('HCO', 1, 1, 1, 0, 0), ('CH2O', 1, 2, 1, 0, 0), ('CH2OH', 1, 3, 1, 0, 0), ('CH3O', 1, 3, 1, 0, 0),
    ('CH3OH', 1, 4, 1, 0, 0), ('C2H', 2, 1, 0, 0, 0), ('C2H2', 2, 2, 0, 0, 0), ('C2H3', 2, 3, 0, 0, 0),
    ('C2H4', 2, 4, 0, 0, 0), ('C2H5', 2, 5, 0, 0, 0), ('C2H6', 2, 6, 0, 0, 0), ('HCCO', 2, 1, 1, 0, 0),
    ('CH2CO', 2, 2, 1, 0, 0), ('HCCOH', 2, 2, 1, 0, 0), ('N', 0, 0, 0, 1, 0), ('NH', 0, 1, 0, 1, 0),
    ('NH2', 0, 2, 0, 1, 0), ('NH3', 0, 3, 0, 1, 0), ('NNH', 0, 1, 0, 2, 0), ('NO', 0, 0, 1, 1, 0),
    ('NO2', 0, 0, 2, 1, 0), ('N2O', 0, 0, 1, 2, 0), ('HNO', 0, 1, 1, 1, 0), ('CN', 1, 0, 0, 1, 0),
    ('HCN', 1, 1, 0, 1, 0), ('H2CN', 1, 2, 0, 1, 0), ('HCNN', 1, 1, 0, 2, 0), ('HCNO', 1, 1, 1, 1, 0),
    ('HOCN', 1, 1, 1, 1, 0), ('HNCO', 1, 1, 1, 1, 0), ('NCO', 1, 0, 1, 1, 0), ('N2', 0, 0, 0, 2, 0),
    ('AR', 0, 0, 0, 0, 1), ('C3H7', 3, 7, 0, 0, 0), ('C3H8', 3, 8, 0, 0, 0), ('CH2CHO', 2, 3, 1, 0, 0),
    ('CH3CHO', 2, 4, 1, 0, 0),
]
_EL = ('C', 'H', 'O', 'N', 'AR')


def _real_cards():
    """NASA cards of the H2/O2/N2/AR species (mechanism data shipped in data/)."""
    txt = open(os.path.join(_DATA, 'h2o2_n2.inp')).read()
    m = parse_mech(txt)
    return {s.name: s for s in m.specs}


def _species_pool(nsp: int, rng):
    pool = list(GRI_SPECIES)
    i = 0
    while len(pool) < nsp:
        c = int(rng.integers(3, 8))
        h = int(rng.integers(max(1, c - 2), 2 * c + 3))
        o = int(rng.integers(0, 3))
        pool.append(('X%03d' % i, c, h, o, 0, 0))
        i += 1
    return pool[:nsp]


def _thermo_card(name, comp, real, rng):
    if name in real:
        sp = real[name]
        lo, hi = list(sp.lo), list(sp.hi)
        Tr = sp.Trange
    else:
        nat = sum(comp)
        # heat capacity grows with atom count; formation enthalpy ~ atom-additive + noise
        a1 = 2.5 + 0.45 * (nat - 1) + rng.uniform(-0.3, 0.3)
        a2 = (1.5e-3 * nat) * rng.uniform(0.5, 1.5)
        lo = [a1, a2, -a2 * 4e-4 * rng.uniform(0.5, 1.5), a2 * 1e-7 * rng.uniform(0.2, 1.5),
              -a2 * 1.2e-11 * rng.uniform(0.2, 1.5), 0.0, 0.0]
        e_atom = dict(C=8.5e3, H=1.2e3, O=-9.0e3, N=3.0e3, AR=0.0)
        lo[5] = sum(e_atom[e] * c for e, c in zip(_EL, comp)) + rng.normal(0, 1500.0)
        lo[6] = rng.uniform(-2.0, 12.0)
        # high range: continuous cp and h at 1000 K with flatter curvature
        T = 1000.0
        cp = lo[0] + T * (lo[1] + T * (lo[2] + T * (lo[3] + lo[4] * T)))
        h = lo[5] + T * (lo[0] + T * (lo[1] / 2 + T * (lo[2] / 3 + T * (lo[3] / 4 + lo[4] / 5 * T))))
        s = lo[0] * np.log(T) + T * (lo[1] + T * (lo[2] / 2 + T * (lo[3] / 3 + lo[4] / 4 * T))) + lo[6]
        b2 = lo[1] * 0.45
        b3 = -b2 * 2.6e-4
        b4 = b2 * 3.0e-8
        b5 = -b2 * 1.5e-12
        b1 = cp - T * (b2 + T * (b3 + T * (b4 + b5 * T)))
        b6 = h - T * (b1 + T * (b2 / 2 + T * (b3 / 3 + T * (b4 / 4 + b5 / 5 * T))))
        b7 = s - (b1 * np.log(T) + T * (b2 + T * (b3 / 2 + T * (b4 / 3 + b5 / 4 * T))))
        hi = [b1, b2, b3, b4, b5, b6, b7]
        Tr = [200.0, 1000.0, 3500.0]
    el = ''
    for e, c in zip(_EL, comp):
        if c:
            el += '%-2s%3d' % (e, c)
    el = (el + ' ' * 20)[:20]
    l1 = '%-18s%-6s%s%s%10.3f%10.3f%8.2f' % (name, 'SYNTH', el, 'G', Tr[0], Tr[2], Tr[1])
    l1 = (l1 + ' ' * 80)[:79] + '1'

    def row(vals, idx):
        s_ = ''.join('%15.8E' % v for v in vals)
        return (s_ + ' ' * 80)[:79] + str(idx)
    return '\n'.join([l1, row(hi[0:5], 2), row(hi[5:7] + lo[0:3], 3), row(lo[3:7], 4)]), list(lo), list(hi)


def _balanced_products(pool, reac, rng, used):
    """A product pair (or single product) with the reactants' elements."""
    tot = np.sum([np.array(pool[i][1:]) for i in reac], axis=0)
    idx = list(range(len(pool)))
    rng.shuffle(idx)
    for i in idx[:60]:
        if np.array_equal(np.array(pool[i][1:]), tot) and (i,) != tuple(reac):
            return [i]
    for i, j in itertools.combinations(idx[:45], 2):
        if np.array_equal(np.array(pool[i][1:]) + np.array(pool[j][1:]), tot):
            key = (tuple(sorted(reac)), tuple(sorted((i, j))))
            if set((i, j)) != set(reac) and key not in used:
                used.add(key)
                return [i, j]
    return None


def generate(nsp: int = 53, nrxn: int = 325, n_falloff: int = 29, n_thd: int = 40,
             n_plog: int = 0, n_irrev: int = 12, n_dup_pairs: int = 5, seed: int = 20240901,
             title: str = 'GRI-Mech-3.0-shaped synthetic mechanism') -> str:
    rng = np.random.default_rng(seed)
    pool = _species_pool(nsp, rng)
    real = _real_cards()
    names = [p[0] for p in pool]
    inert = {names.index('AR'), names.index('N2')} if 'N2' in names else set()
    reactive = [i for i in range(nsp) if i not in inert]
    colliders = [n for n in ('H2', 'H2O', 'CH4', 'CO', 'CO2', 'C2H6', 'AR') if n in names]

    def arr():
        A = 10 ** rng.uniform(6, 14)
        b = 0.0 if rng.random() < 0.45 else round(float(rng.uniform(-1.8, 2.8)), 3)
        E = 0.0 if rng.random() < 0.3 else round(float(rng.uniform(-1500, 42000)), 1)
        return A, b, E

    def fmt(eq, A, b, E):
        return '%-48s %10.3E %8.3f %10.2f' % (eq, A, b, E)

    def effs():
        k = int(rng.integers(min(4, len(colliders)), min(7, len(colliders)) + 1))
        ch = list(rng.choice(colliders, size=k, replace=False))
        return ' '.join('%s/%.2f/' % (c, 0.0 if rng.random() < 0.08 else rng.uniform(0.4, 6.0)) for c in ch)

    lines = []
    used = set()
    count = 0
    cards, LO, HI = [], [], []
    for p_ in pool:
        txt, lo_, hi_ = _thermo_card(p_[0], p_[1:], real, rng)
        cards.append(txt); LO.append(lo_); HI.append(hi_)

    def smh(i, T):
        a = LO[i] if T <= 1000.0 else HI[i]
        return (a[0] * (np.log(T) - 1.0) + a[1] * T / 2 + a[2] * T ** 2 / 6 + a[3] * T ** 3 / 12 +
                a[4] * T ** 4 / 20 - a[5] / T + a[6])

    def sane(r, p):
        """|ln Kp| stays moderate over the temperature range of the benchmark states."""
        for T in (700.0, 1500.0, 2600.0):
            d = sum(smh(i, T) for i in p) - sum(smh(i, T) for i in r)
            if abs(d) > 30.0:
                return False
        return True

    def two_two():
        for _ in range(2000):
            r = list(rng.choice(reactive, size=2, replace=rng.random() < 0.08))
            p = _balanced_products(pool, r, rng, used)
            if p is not None and sane(r, p):
                return r, p
        raise RuntimeError('could not find a balanced, thermodynamically moderate reaction')

    def side(ix):
        out = []
        for i in sorted(set(ix), key=ix.index):
            c = ix.count(i)
            out.append(('%d' % c if c > 1 else '') + names[i])
        return '+'.join(out)

    # falloff: A + B (+M) <=> C (+M) (recombination), Troe / Lindemann, a few chemically activated
    for q in range(n_falloff):
        r, p = None, None
        for _ in range(300):
            r = list(rng.choice(reactive, size=2, replace=rng.random() < 0.1))
            tot = np.sum([np.array(pool[i][1:]) for i in r], axis=0)
            cands = [i for i in reactive if np.array_equal(np.array(pool[i][1:]), tot) and sane(r, [i])]
            if cands:
                p = [int(rng.choice(cands))]
                break
        if p is None:
            r, p = two_two()
        A, b, E = arr()
        chem_act = q % 9 == 8
        eq = '%s(+M)<=>%s(+M)' % (side(r), side(p))
        lines.append(fmt(eq, A * 1e-2, b * 0.3, max(E, 0.0) * 0.2))
        A0, b0, E0 = 10 ** rng.uniform(14, 20), round(float(rng.uniform(-4.8, -0.5)), 3), round(float(rng.uniform(0, 7000)), 1)
        if chem_act:
            lines.append('     HIGH / %10.3E %8.3f %10.2f /' % (10 ** rng.uniform(8, 12), round(float(rng.uniform(0, 1.5)), 3), E0))
        else:
            lines.append('     LOW  / %10.3E %8.3f %10.2f /' % (A0, b0, E0))
        kind = q % 7
        if kind != 6:
            a = rng.uniform(0.2, 0.95)
            t3, t1 = rng.uniform(50, 600), rng.uniform(800, 3500)
            if kind == 5:
                lines.append('     TROE/ %8.4f %9.2f %9.2f /' % (a, t3, t1))
            else:
                lines.append('     TROE/ %8.4f %9.2f %9.2f %9.2f /' % (a, t3, t1, rng.uniform(2000, 9000)))
        lines.append(effs())
        count += 1

    # third-body: A + B + M <=> C + M and dissociations
    for q in range(n_thd):
        r, p = two_two()
        if rng.random() < 0.5 and len(p) == 2:
            tot = np.sum([np.array(pool[i][1:]) for i in r], axis=0)
            cands = [i for i in reactive if np.array_equal(np.array(pool[i][1:]), tot) and sane(r, [i])]
            if cands:
                p = [int(rng.choice(cands))]
        A, b, E = arr()
        eq = '%s+M<=>%s+M' % (side(r), side(p))
        lines.append(fmt(eq, A * 1e3, min(b, 0.0) - 0.5, 0.0 if rng.random() < 0.6 else E * 0.3))
        if q % 8 != 7:
            lines.append(effs())
        count += 1

    # PLOG
    for q in range(n_plog):
        r, p = two_two()
        rev = rng.random() < 0.8
        A, b, E = arr()
        lines.append(fmt('%s%s%s' % (side(r), '<=>' if rev else '=>', side(p)), A, b if b else 0.5, abs(E) + 500.0))
        npz = int(rng.integers(3, 6))
        ps = sorted(10 ** rng.uniform(-2, 2, npz))
        for P in ps:
            lines.append('     PLOG / %10.4E %10.3E %8.3f %10.2f /' %
                         (P, A * 10 ** rng.uniform(-1, 1), (b if b else 0.5) + rng.uniform(-0.3, 0.3),
                          abs(E) + 500.0 + rng.uniform(0, 3000)))
        count += 1

    # duplicates (pairs with identical stoichiometry)
    for q in range(n_dup_pairs):
        r, p = two_two()
        eq = '%s<=>%s' % (side(r), side(p))
        for _ in range(2):
            A, b, E = arr()
            lines.append(fmt(eq, A, b, E))
            lines.append(' DUPLICATE')
            count += 1

    # irreversible steps, some with three products
    for q in range(n_irrev):
        r, p = two_two()
        if q % 3 == 0 and len(p) == 2:
            # split one product further when possible: A + B => C + D + E
            tot = np.array(pool[p[1]][1:])
            for i, j in itertools.combinations(reactive, 2):
                if np.array_equal(np.array(pool[i][1:]) + np.array(pool[j][1:]), tot):
                    p = [p[0], i, j]
                    break
        A, b, E = arr()
        lines.append(fmt('%s=>%s' % (side(r), side(p)), A, b, E))
        count += 1

    # the rest: reversible elementary steps
    while count < nrxn:
        r, p = two_two()
        A, b, E = arr()
        lines.append(fmt('%s<=>%s' % (side(r), side(p)), A, b, E))
        count += 1

    els = 'O  H  C  N  AR'
    head = ['! %s' % title,
            '! generated by pyjac_amd/synth_mech.py (seed %d): synthetic rate and thermo data, NOT a real mechanism' % seed,
            'ELEMENTS', els, 'END', 'SPECIES']
    for i in range(0, nsp, 6):
        head.append('  '.join('%-8s' % n for n in names[i:i + 6]))
    head += ['END', 'THERMO ALL', '   300.000  1000.000  5000.000']
    head += cards
    head += ['END', 'REACTIONS']
    return '\n'.join(head + lines + ['END', ''])


def write_default_mechanisms(outdir: str = _DATA):
    """GRI-3.0-shaped (53/325) and USC-II-shaped (111/784, PLOG) files."""
    with open(os.path.join(outdir, 'gri30_shaped.inp'), 'w') as f:
        f.write(generate(53, 325, 29, 40, 0, 12, 5, seed=20240901,
                         title='GRI-Mech-3.0-shaped synthetic mechanism (53 species, 325 reactions)'))
    with open(os.path.join(outdir, 'usc2_shaped.inp'), 'w') as f:
        f.write(generate(111, 784, 48, 60, 30, 30, 8, seed=20240902,
                         title='USC-Mech-II-shaped synthetic mechanism (111 species, 784 reactions, PLOG)'))


if __name__ == '__main__':
    write_default_mechanisms()
