"""Chemkin-format mechanism front end (host side, mechanism-load time).

This is the build's own reader for the inputs pyJac's ``read_mech`` accepts
(reference behaviour: pyjac/core/mech_interpret.py:56-883).  It produces the same
mechanism *model* pyJac's generator consumes -- species with NASA-7 data and
molecular weights, reactions in (kmol, m^3, s, K) units with activation
temperatures -- so that the tables built from it (pyjac_amd/tables.py) describe
exactly the arithmetic the reference's generated C performs.

Reference behaviours reproduced on purpose (each cited where it is done):
  * unit conversion of A / E             mech_interpret.py:438-452, 504-540, 649-652
  * Troe zero-parameter guard (1e-30)    mech_interpret.py:551-560
  * explicit REV -> two irreversible     mech_interpret.py:693-713
  * last species = first of N2/AR/HE     create_jacobian.py:3521-3563
  * element weights / RU / PA            chem_utilities.py:16-24, 51-99

Cantera input: .cti files through pyjac_amd/cti.py (its own reader of the format: no Cantera needed); .xml / a live
cantera.Solution (mech_interpret.py:886-1137) are not read.
"""
from __future__ import annotations

import copy
import math
import re
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

# chem_utilities.py:16-24
RU = 8314.4621          # J / (kmol K)
RU_JOUL = 8.3144621
PA = 101325.0

# chem_utilities.py:63-99 (values are data: standard atomic weights as the
# reference tabulates them; only the elements combustion mechanisms use).
ELEM_WT = {
    'h': 1.00794, 'he': 4.00260, 'li': 6.93900, 'be': 9.01220, 'b': 10.81100,
    'c': 12.0110, 'n': 14.00674, 'o': 15.99940, 'f': 18.99840, 'ne': 20.18300,
    'na': 22.98980, 'mg': 24.31200, 'al': 26.98150, 'si': 28.08600,
    'p': 30.97380, 's': 32.06400, 'cl': 35.45300, 'ar': 39.94800,
    'k': 39.10200, 'ca': 40.08000, 'fe': 55.84700, 'br': 79.90090,
    'kr': 83.80000, 'i': 126.90440, 'xe': 131.30000, 'd': 2.01410,
    'e': 5.48578e-4,
}

# mech_interpret.py:42-49
ACT_ENERGY_FACT = {
    'kelvins': 1.0,
    'evolts': 11595.,
    'cal/mole': 4.184 / RU_JOUL,
    'kcal/mole': 4184. / RU_JOUL,
    'joules/mole': 1. / RU_JOUL,
    'kjoules/mole': 1000.0 / RU_JOUL,
    'joules/kmole': 1. / (RU_JOUL * 1000.),
}


@dataclass
class Species:
    name: str
    elem: List[Tuple[str, int]] = field(default_factory=list)
    mw: float = 0.0
    lo: List[float] = field(default_factory=lambda: [0.0] * 7)
    hi: List[float] = field(default_factory=lambda: [0.0] * 7)
    Trange: List[float] = field(default_factory=lambda: [300.0, 1000.0, 5000.0])


@dataclass
class Reaction:
    rev: bool
    reac: list            # species names, later indices
    reac_nu: list
    prod: list
    prod_nu: list
    A: float
    b: float
    E: float              # activation temperature [K]
    rev_par: list = field(default_factory=list)
    dup: bool = False
    thd_body: bool = False
    thd_body_eff: list = field(default_factory=list)   # [(species, alpha)]
    pdep: bool = False
    pdep_sp: object = ''   # '' / None = (+M); else species
    low: list = field(default_factory=list)
    high: list = field(default_factory=list)
    troe: bool = False
    troe_par: list = field(default_factory=list)
    sri: bool = False
    sri_par: list = field(default_factory=list)
    plog: bool = False
    plog_par: list = field(default_factory=list)       # [[P, A, b, E], ...]
    cheb: bool = False
    cheb_n_temp: int = 0
    cheb_n_pres: int = 0
    cheb_par: list = field(default_factory=list)       # row-major [n_temp][n_pres], log10 k
    cheb_tlim: list = field(default_factory=list)      # [Tmin, Tmax] K
    cheb_plim: list = field(default_factory=list)      # [Pmin, Pmax] Pa


@dataclass
class Mechanism:
    elems: List[str]
    specs: List[Species]
    reacs: List[Reaction]
    # permutation applied to put the last species at the end
    # (utils.py:55-91): specs[i] = original[fwd_spec_map[i]]
    fwd_spec_map: List[int]
    back_spec_map: List[int]

    @property
    def nsp(self):
        return len(self.specs)

    @property
    def n_fwd(self):
        return len(self.reacs)

    @property
    def n_rev(self):
        return sum(1 for r in self.reacs if r.rev)

    @property
    def n_pres_mod(self):
        return sum(1 for r in self.reacs if r.thd_body or r.pdep)

    def species_names(self):
        return [s.name for s in self.specs]


# --------------------------------------------------------------------------
# parsing helpers
# --------------------------------------------------------------------------
_NUM = re.compile(r'^[+-]?(\d+\.?\d*|\.\d+)([eEdD][+-]?\d+)?$')


def _fl(tok: str) -> float:
    return float(tok.replace('d', 'e').replace('D', 'E'))


def _strip_pdep(side: str):
    """Find '(+M)' / '(+SPECIES)' in one side of a reaction string.

    Returns (side without it, pdep flag, pdep species or '' for M, thd flag).
    Parentheses that do not start with '+' belong to species names
    (mech_interpret.py:240-272).
    """
    pos = 0
    while True:
        i1 = side.find('(', pos)
        if i1 < 0:
            return side, False, '', False
        i2 = side.find(')', i1)
        if i2 < 0:
            return side, False, '', False
        inner = side[i1 + 1:i2].strip()
        if inner.startswith('+') and len(inner) > 1:
            sp = inner[1:].strip()
            rest = side[:i1] + side[i2 + 1:]
            if sp.lower() == 'm':
                return rest, True, '', True
            return rest, True, sp, False
        pos = i2 + 1


def _split_species(side: str, known: set):
    """Split 'A+2B+M' into [(name, nu)], third-body flag.

    A '+' that ends a species name (ions, e.g. 'CH+') is kept with the name
    when the joined token is a declared species (mech_interpret.py:276-293).
    """
    raw = [t.strip() for t in side.split('+')]
    toks = []
    i = 0
    while i < len(raw):
        t = raw[i]
        if t == '' and toks:
            toks[-1] += '+'
        elif t != '':
            toks.append(t)
        i += 1
    out_sp, out_nu, thd = [], [], False
    for t in toks:
        nu = 1
        if t[0].isdigit() or t[0] == '.':
            # leading stoichiometric coefficient unless the whole token is a
            # declared species (names may start with digits)
            if t not in known:
                j = 0
                while j < len(t) and not t[j].isalpha():
                    j += 1
                nus = t[:j]
                nu = float(nus) if '.' in nus else int(nus)
                t = t[j:].strip()
        if t.lower() == 'm':
            thd = True
            continue
        if t in out_sp:
            out_nu[out_sp.index(t)] += nu
        else:
            out_sp.append(t)
            out_nu.append(nu)
    return out_sp, out_nu, thd


def _split_fixed(s: str, n: int):
    return [s[i:i + n] for i in range(0, len(s), n)]


def _read_thermo_lines(lines: List[str], specs: List[Species], elem_wt: dict):
    """NASA-7 fixed-column THERMO block (mech_interpret.py:735-883)."""
    byname = {s.name: s for s in specs}
    it = iter(lines)
    T_ranges = [300.0, 1000.0, 5000.0]
    first = True
    for line in it:
        if not line.strip() or line.lstrip().startswith('!'):
            continue
        if first:
            first = False
            toks = line.split()
            if toks and _NUM.match(toks[0]):
                T_ranges = [_fl(t) for t in toks[:3]]
                continue
        if line[:3].lower() == 'end':
            break
        name = line[0:18].strip()
        if ' ' in name:
            name = name[:name.find(' ')]
        l2 = next(it)
        l3 = next(it)
        l4 = next(it)
        sp = byname.get(name)
        if sp is None or sp.mw:
            continue
        for es in _split_fixed(line[24:44], 5):
            e = es[0:2].strip()
            if e == '' or e == '0':
                continue
            cnt = es[2:].strip()
            if not cnt:
                continue
            num = int(float(cnt))
            sp.elem.append((e, num))
            sp.mw += num * elem_wt[e.lower()]
        tt = [_fl(t) for t in line[45:74].split()]
        T_low, T_high = tt[0], tt[1]
        T_com = tt[2] if len(tt) >= 3 else T_ranges[1]
        sp.Trange = [T_low, T_com, T_high]
        c2 = [_fl(c) for c in _split_fixed(l2[0:75], 15) if c.strip()]
        c3 = [_fl(c) for c in _split_fixed(l3[0:75], 15) if c.strip()]
        c4 = [_fl(c) for c in _split_fixed(l4[0:75], 15) if c.strip()]
        sp.hi = [c2[0], c2[1], c2[2], c2[3], c2[4], c3[0], c3[1]]
        sp.lo = [c3[2], c3[3], c3[4], c4[0], c4[1], c4[2], c4[3]]


def read_mech(mech_filename: str, therm_filename: Optional[str] = None,
              last_spec: Optional[str] = None) -> Mechanism:
    """Parse a Chemkin mechanism (and optional thermo database) -- or, by its extension, a Cantera .cti file
    (create_jacobian.py:3490-3493 dispatches the same way)."""
    with open(mech_filename, 'r') as f:
        text = f.read()
    if mech_filename.lower().endswith('.cti'):
        from .cti import parse_cti
        return parse_cti(text, last_spec)
    therm_text = None
    if therm_filename:
        with open(therm_filename, 'r') as f:
            therm_text = f.read()
    return parse_mech(text, therm_text, last_spec)


def parse_mech(text: str, therm_text: Optional[str] = None,
               last_spec: Optional[str] = None) -> Mechanism:
    elem_wt = dict(ELEM_WT)
    elems: List[str] = []
    specs: List[Species] = []
    reacs: List[Reaction] = []
    thermo_lines: List[str] = []

    key = ''
    units_E, units_A = 'cal/mole', 'moles'
    # state of the last reaction line, used by REV unit conversion exactly as
    # the reference does (mech_interpret.py:484-494 reads the loop locals)
    last_thd = last_pdep = False
    known: set = set()

    raw_lines = text.splitlines()
    idx = 0
    while idx < len(raw_lines):
        raw = raw_lines[idx]
        idx += 1
        if key == 'ther':
            if raw[:3].lower() == 'end':
                key = ''
                continue
            # REACTIONS may follow THERMO without an END in sloppy files
            if raw[:4].lower() == 'reac':
                key = ''
            else:
                thermo_lines.append(raw)
                continue
        if not raw.strip() or raw.lstrip().startswith('!'):
            continue
        line = raw.strip()
        c = line.find('!')
        if c > 0:
            line = line[:c].strip()
        head = line[:4].lower()
        if head == 'elem':
            key = 'elem'
            line = line.split(None, 1)[1] if len(line.split()) > 1 else ''
            if not line:
                continue
        elif head == 'spec':
            key = 'spec'
            line = line.split(None, 1)[1] if len(line.split()) > 1 else ''
            if not line:
                continue
        elif head == 'reac':
            key = 'reac'
            units_E, units_A = 'cal/mole', 'moles'
            for u in line.split()[1:]:
                ul = u.lower()
                if ul in ('moles', 'molecules'):
                    units_A = ul
                elif ul in ACT_ENERGY_FACT:
                    units_E = ul
                else:
                    raise ValueError('unsupported units on REACTIONS line: ' + u)
            if units_A == 'molecules':
                raise NotImplementedError('molecules units not supported')
            known = set(s.name for s in specs)
            continue
        elif head == 'ther':
            key = 'ther'
            continue
        elif line[:3].lower() == 'end':
            key = ''
            continue

        if key == 'elem':
            last_e = ''
            for tok in line.replace('/', ' ').split():
                if tok.lower() == 'end':
                    key = ''
                    break
                if tok.isalpha():
                    if tok not in elems:
                        elems.append(tok)
                    last_e = tok
                else:
                    elem_wt[last_e.lower()] = _fl(tok)
        elif key == 'spec':
            for tok in line.split():
                if tok.lower() == 'end':
                    key = ''
                    break
                if all(tok != s.name for s in specs):
                    specs.append(Species(tok))
        elif key == 'reac':
            if '=' in line:
                eq, sA, sb, sE = line.rsplit(None, 3)
                A, b, E = _fl(sA), _fl(sb), _fl(sE)
                eq = eq.replace(' ', '').replace('\t', '')
                if '<=>' in eq:
                    lhs, rhs = eq.split('<=>', 1)
                    rev = True
                elif '=>' in eq:
                    lhs, rhs = eq.split('=>', 1)
                    rev = False
                else:
                    lhs, rhs = eq.split('=', 1)
                    rev = True
                lhs, pd1, psp1, thd1 = _strip_pdep(lhs.strip())
                rhs, pd2, psp2, thd2 = _strip_pdep(rhs.strip())
                pdep = pd1 or pd2
                pdep_sp = psp1 or psp2
                r_sp, r_nu, t1 = _split_species(lhs, known)
                p_sp, p_nu, t2 = _split_species(rhs, known)
                thd = (t1 or t2 or thd1 or thd2) and not pdep
                for s in r_sp + p_sp:
                    if s not in known:
                        raise ValueError('reaction %d contains unknown species %s'
                                         % (len(reacs), s))
                E *= ACT_ENERGY_FACT[units_E]
                if units_A == 'moles':
                    order = sum(r_nu)
                    if thd:
                        A /= 1000. ** order
                    else:
                        A /= 1000. ** (order - 1.)
                rx = Reaction(rev, r_sp, r_nu, p_sp, p_nu, A, b, E)
                rx.thd_body = thd
                rx.pdep = pdep
                rx.pdep_sp = pdep_sp if pdep else ''
                reacs.append(rx)
                last_thd, last_pdep = thd, pdep
            else:
                rx = reacs[-1]
                aux = line[:3].lower()
                body = line.replace('/', ' ').replace(',', ' ').split()
                if aux == 'dup':
                    rx.dup = True
                elif aux == 'rev':
                    p1, p2, p3 = _fl(body[1]), _fl(body[2]), _fl(body[3])
                    p3 *= ACT_ENERGY_FACT[units_E]
                    if units_A == 'moles':
                        order = sum(rx.prod_nu)
                        if last_thd:
                            p1 /= 1000. ** order
                        else:
                            p1 /= 1000. ** (order - 1.)
                    if p1 != 0.0:
                        rx.rev_par = [p1, p2, p3]
                    else:
                        rx.rev = False
                elif aux == 'low':
                    p1, p2, p3 = _fl(body[1]), _fl(body[2]), _fl(body[3])
                    p3 *= ACT_ENERGY_FACT[units_E]
                    if units_A == 'moles':
                        p1 /= 1000. ** sum(rx.reac_nu)
                    rx.low = [p1, p2, p3]
                elif aux == 'hig':
                    p1, p2, p3 = _fl(body[1]), _fl(body[2]), _fl(body[3])
                    p3 *= ACT_ENERGY_FACT[units_E]
                    if units_A == 'moles':
                        p1 /= 1000. ** (sum(rx.reac_nu) - 2.)
                    rx.high = [p1, p2, p3]
                elif aux == 'tro':
                    rx.troe = True
                    p1, p2, p3 = _fl(body[1]), _fl(body[2]), _fl(body[3])
                    if p2 == 0:
                        p2 = 1e-30
                    if p3 == 0:
                        p3 = 1e-30
                    rx.troe_par = [p1, p2, p3]
                    if len(body) > 4:
                        rx.troe_par.append(_fl(body[4]))
                elif aux == 'sri':
                    rx.sri = True
                    rx.sri_par = [_fl(body[1]), _fl(body[2]), _fl(body[3])]
                    if len(body) > 4:
                        rx.sri_par += [_fl(body[4]), _fl(body[5])]
                elif aux == 'che':
                    # Chebyshev rate expression (mech_interpret.py:589-606): "CHEB / n m c.. /", then
                    # continuation lines of coefficients; not lumped in with the falloff reactions
                    if not rx.cheb:
                        rx.cheb = True
                        rx.pdep = False
                        rx.cheb_n_temp, rx.cheb_n_pres = int(_fl(body[1])), int(_fl(body[2]))
                        rx.cheb_par = [_fl(t) for t in body[3:]]
                    else:
                        rx.cheb_par += [_fl(t) for t in body[1:]]
                elif aux == 'pch':
                    rx.cheb_plim = [_fl(body[1]) * PA, _fl(body[2]) * PA]       # atm -> Pa
                    if len(body) > 5 and body[3].lower() == 'tcheb':
                        rx.cheb_tlim = [_fl(body[4]), _fl(body[5])]
                elif aux == 'tch':
                    rx.cheb_tlim = [_fl(body[1]), _fl(body[2])]
                    if len(body) > 5 and body[3].lower() == 'pcheb':
                        rx.cheb_plim = [_fl(body[4]) * PA, _fl(body[5]) * PA]
                elif aux == 'plo':
                    if not rx.plog:
                        rx.plog = True
                        rx.pdep = False
                        rx.plog_par = []
                    pars = [_fl(t) for t in body[1:5]]
                    pars[0] *= 101325.0
                    pars[3] *= ACT_ENERGY_FACT[units_E]
                    if units_A == 'moles':
                        pars[1] /= 1000. ** (sum(rx.reac_nu) - 1.)
                    rx.plog_par.append(pars)
                else:
                    for i in range(0, len(body) - 1, 2):
                        rx.thd_body_eff.append((body[i], _fl(body[i + 1])))

    # Chebyshev reactions: coefficient count, unit conversion of the leading coefficient
    # (mech_interpret.py:663-680)
    for idx, rx in enumerate(reacs):
        if not rx.cheb:
            continue
        if len(rx.cheb_par) != rx.cheb_n_temp * rx.cheb_n_pres:
            raise ValueError('incorrect number of CHEB coefficients in reaction %d' % idx)
        if len(rx.cheb_tlim) != 2 or len(rx.cheb_plim) != 2:
            raise ValueError('reaction %d: CHEB needs TCHEB and PCHEB limits' % idx)
        if rx.cheb_n_temp < 3 or rx.cheb_n_pres < 2:
            # the reference's emitters index dot_prod[2] / cheb_par[i, 1] unconditionally
            # (rate_subs.py:199-230, create_jacobian.py:1553-1585)
            raise ValueError('reaction %d: CHEB needs at least 3 x 2 coefficients' % idx)
        if units_A == 'moles':
            rx.cheb_par[0] += math.log10(0.001 ** (sum(rx.reac_nu) - 1.))

    # explicit REV -> pair of irreversible reactions (mech_interpret.py:693-713)
    out: List[Reaction] = []
    for rx in reacs:
        if rx.rev_par:
            fwd = rx
            bwd = copy.deepcopy(rx)
            bwd.A, bwd.b, bwd.E = rx.rev_par
            bwd.rev = False
            bwd.rev_par = []
            bwd.reac, bwd.reac_nu = list(rx.prod), list(rx.prod_nu)
            bwd.prod, bwd.prod_nu = list(rx.reac), list(rx.reac_nu)
            fwd.rev = False
            fwd.rev_par = []
            out += [fwd, bwd]
        else:
            out.append(rx)
    reacs = out

    if thermo_lines:
        _read_thermo_lines(thermo_lines, specs, elem_wt)
    if any(not s.mw for s in specs) and therm_text is not None:
        tl = therm_text.splitlines()
        start = 0
        for i, l in enumerate(tl):
            # (blank and comment lines in front of the THERMO keyword are skipped, mech_interpret.py:764-769)
            if not l.strip() or l.lstrip().startswith('!'):
                continue
            if 'thermo' in l.lower():
                start = i + 1
                break
        _read_thermo_lines(tl[start:], specs, elem_wt)
    return finish_mechanism(elems, specs, reacs, last_spec)


def finish_mechanism(elems: List[str], specs: List[Species], reacs: List[Reaction],
                     last_spec: Optional[str] = None) -> Mechanism:
    """Common tail of the front ends (Chemkin: parse_mech; Cantera .cti: pyjac_amd/cti.py): checks, the last-species
    rule, species names -> indices."""
    missing = [s.name for s in specs if not s.mw]
    if missing:
        raise ValueError('missing thermo data for ' + ', '.join(missing))
    if not specs or not reacs:
        raise ValueError('no species / reactions found')

    # ---- last species (create_jacobian.py:3503-3563) ----
    last = None
    if last_spec is not None:
        last = next((i for i, s in enumerate(specs)
                     if s.name.lower() == last_spec.lower().strip()), None)
    if last is None:
        for nm, wt in (('n2', ELEM_WT['n'] * 2.), ('ar', ELEM_WT['ar']),
                       ('he', ELEM_WT['he'])):
            m = next((i for i, s in enumerate(specs)
                      if s.name.lower() == nm and s.mw == wt), None)
            if m is not None:
                last = m
                break
    if last is None:
        last = len(specs) - 1
    n = len(specs)
    fwd_map = [i for i in range(n) if i != last] + [last]
    back_map = [0] * n
    for new, old in enumerate(fwd_map):
        back_map[old] = new
    specs = [specs[i] for i in fwd_map]

    # names -> indices (utils.py:250-277)
    smap = {s.name: i for i, s in enumerate(specs)}
    for rx in reacs:
        rx.reac = [smap[s] for s in rx.reac]
        rx.prod = [smap[s] for s in rx.prod]
        rx.thd_body_eff = [(smap[s], a) for s, a in rx.thd_body_eff]
        rx.pdep_sp = smap[rx.pdep_sp] if rx.pdep_sp != '' else None

    return Mechanism(elems, specs, reacs, fwd_map, back_map)


def get_nu(isp: int, rx: Reaction):
    """Net stoichiometric coefficient of species ``isp`` (utils.py:94-123)."""
    if isp in rx.prod and isp in rx.reac:
        return rx.prod_nu[rx.prod.index(isp)] - rx.reac_nu[rx.reac.index(isp)]
    if isp in rx.prod:
        return rx.prod_nu[rx.prod.index(isp)]
    if isp in rx.reac:
        return -rx.reac_nu[rx.reac.index(isp)]
    return 0
