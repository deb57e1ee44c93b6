"""Cantera ``.cti`` mechanism front end (host side, mechanism-load time).

pyJac reads Cantera input THROUGH Cantera (``read_mech_ct``, pyjac/core/mech_interpret.py:886-1137:
``ct.Solution(filename)`` and a walk over its species / reaction objects).  Cantera is not a dependency here: a
``.cti`` file is a Python script over a small vocabulary of directives (``units``, ``ideal_gas``, ``species``,
``reaction``, ``three_body_reaction``, ``falloff_reaction``, ``chemically_activated_reaction``, ``pdep_arrhenius``,
``chebyshev_reaction`` and their helpers ``NASA``, ``Arrhenius``, ``Troe``, ``SRI``, ``Lindemann`` ...), so this module
supplies that vocabulary itself, evaluates the file against it, and builds the same mechanism MODEL the Chemkin front
end builds (pyjac_amd/mechanism.py) -- (kmol, m^3, s, K) units, activation temperatures -- with the same conventions
``read_mech_ct`` applies:

  * a reaction type per directive, low / high limits of falloff and chemically activated reactions
    (mech_interpret.py:997-1070), Troe zero-parameter guard (1e-30; :1021-1031, :1055-1065), SRI parameters,
    PLOG rate lists (:1072-1088), Chebyshev limits and coefficients (:1090-1103), duplicates (:1127);
  * elementary reactions with a zero pre-exponential factor are dropped (:1109-1111);
  * third-body efficiencies in SPECIES order, a default efficiency other than 1 given to every species the
    directive does not list (:958-990); a falloff reaction whose collider is one named species, ``(+ AR)``, keeps
    it as ``pdep_sp``.

Differences from going through Cantera, stated rather than hidden: molecular weights come from the element table of
chem_utilities.py:63-99 (mechanism.ELEM_WT, as for Chemkin input), not from Cantera's own atomic weights (which
differ in the fifth digit and between Cantera versions); only NASA-7 two-range thermo (what ``read_mech_ct``
accepts, :942-949); quantities with explicit unit strings are understood for pressures and temperatures
(``(1.0, 'atm')``), rate coefficients are plain numbers in the file's ``units(...)``.

A .cti file IS a Python script (Cantera's own converter executes it as one): it is evaluated here with this module's
directives and a handful of builtins as its only names, which keeps honest files honest and is no sandbox -- read
files you would also hand to Cantera.
"""
from __future__ import annotations

import math
import re
from typing import List, Optional

from .mechanism import ACT_ENERGY_FACT, ELEM_WT, PA, Mechanism, Reaction, Species, finish_mechanism

_LEN = {'m': 1.0, 'cm': 1e-2, 'mm': 1e-3}
_QTY = {'kmol': 1.0, 'mol': 1e-3, 'molec': 1.0 / 6.02214129e26, 'molecule': 1.0 / 6.02214129e26}
_ENERGY = {'cal/mol': 'cal/mole', 'kcal/mol': 'kcal/mole', 'j/mol': 'joules/mole', 'kj/mol': 'kjoules/mole',
           'j/kmol': 'joules/kmole', 'k': 'kelvins', 'ev': 'evolts'}
_PRES = {'pa': 1.0, 'atm': PA, 'bar': 1e5, 'kpa': 1e3, 'mpa': 1e6, 'torr': PA / 760.0}


class _Arrhenius:
    def __init__(self, A=0.0, b=0.0, E=0.0, **kw):
        self.A, self.b, self.E = A, kw.get('n', b), E


class _Falloff:
    def __init__(self, kind, pars):
        self.kind, self.pars = kind, list(pars)


def _arr(k):
    """[A, b, E] of a rate given as a list / tuple or Arrhenius(...)."""
    if isinstance(k, _Arrhenius):
        return [k.A, k.b, k.E]
    if isinstance(k, (list, tuple)) and len(k) == 3:
        return [k[0], k[1], k[2]]
    raise ValueError('rate coefficient: expected [A, b, E] or Arrhenius(A, b, E), got %r' % (k,))


def _plain(x, what):
    if isinstance(x, (int, float)):
        return float(x)
    raise ValueError('%s: a plain number in the units of the file\'s units(...) directive is expected, got %r' % (what, x))


def _with_unit(x, table, what):
    """A number, or (number, 'unit')."""
    if isinstance(x, (list, tuple)) and len(x) == 2 and isinstance(x[1], str):
        u = x[1].strip().lower()
        if u not in table:
            raise ValueError('%s: unit %r not understood' % (what, x[1]))
        return float(x[0]) * table[u]
    return float(x)


_SIDE_PDEP = re.compile(r'\(\s*\+\s*([^)\s]+)\s*\)')


def _side(text: str, known: set):
    """One side of a .cti reaction equation -> (species, coefficients, third body M?, falloff collider or None).
    Species are separated by ' + ' (a name may itself contain '+': ions), a coefficient is a leading number."""
    pdep_sp = None
    m = _SIDE_PDEP.search(text)
    if m:
        pdep_sp = m.group(1)
        text = text[:m.start()] + text[m.end():]
    sp, nu, thd = [], [], False
    for tok in re.split(r'\s+\+\s+', text.strip()):
        tok = tok.strip()
        if not tok:
            continue
        if tok == 'M':
            thd = True
            continue
        parts = tok.split(None, 1)
        coeff, name = 1.0, tok
        if len(parts) == 2:
            try:
                coeff, name = float(parts[0]), parts[1].strip()
            except ValueError:
                pass
        if name not in known:
            raise ValueError('reaction equation %r contains unknown species %r' % (text, name))
        coeff = int(coeff) if float(coeff).is_integer() else coeff
        if name in sp:
            nu[sp.index(name)] += coeff
        else:
            sp.append(name)
            nu.append(coeff)
    return sp, nu, thd, pdep_sp


def _equation(eq: str, known: set):
    if '<=>' in eq:
        lhs, rhs, rev = *eq.split('<=>', 1), True
    elif '=>' in eq:
        lhs, rhs, rev = *eq.split('=>', 1), False
    elif '=' in eq:
        lhs, rhs, rev = *eq.split('=', 1), True
    else:
        raise ValueError('reaction equation without "=": %r' % eq)
    r_sp, r_nu, t1, p1 = _side(lhs, known)
    p_sp, p_nu, t2, p2 = _side(rhs, known)
    return rev, r_sp, r_nu, p_sp, p_nu, (t1 or t2), (p1 or p2)


def _efficiencies(text: str):
    out = []
    for tok in (text or '').replace(',', ' ').split():
        name, _, val = tok.rpartition(':')
        out.append((name.strip(), float(val)))
    return out


def parse_cti(text: str, last_spec: Optional[str] = None) -> Mechanism:
    """Evaluate a .cti file against this module's directives and build the mechanism model."""
    st = dict(length='m', quantity='kmol', act_energy='j/kmol', elems=None, order=None,
              species={}, reactions=[])

    def units(length='', time='', quantity='', act_energy='', energy='', mass='', pressure=''):
        if length:
            st['length'] = length.strip().lower()
        if quantity:
            st['quantity'] = quantity.strip().lower()
        if act_energy:
            st['act_energy'] = act_energy.strip().lower()
        if time and time.strip().lower() != 's':
            raise ValueError('units(time=%r): only seconds are supported' % time)

    def ideal_gas(name='', elements='', species='', reactions='all', **kw):
        if st['elems'] is None:         # (the first phase defines the mechanism; further ones re-list it)
            st['elems'] = elements.split()
            names = []
            for tok in species.replace(',', ' ').split():
                tok = tok.split(':')[-1].strip()        # 'file: A B C' prefixes are not followed
                if tok and tok not in names:
                    names.append(tok)
            st['order'] = names

    def NASA(Trange=None, coeffs=None, p0=None, **kw):
        rng = kw.get('range', Trange)
        if coeffs is None or len(coeffs) != 7:
            raise ValueError('NASA(): seven coefficients per range (the two-range NASA-7 form) are expected')
        return ('NASA', [float(rng[0]), float(rng[1])], [float(c) for c in coeffs])

    def species(name='', atoms='', thermo=None, **kw):
        if not isinstance(thermo, (list, tuple)) or len(thermo) != 2 or any(t[0] != 'NASA' for t in thermo):
            raise ValueError('species %s: unsupported thermo form (two NASA ranges are expected)' % name)
        lo, hi = sorted(thermo, key=lambda t: t[1][0])
        sp = Species(name)
        for tok in atoms.replace(',', ' ').split():
            el, _, cnt = tok.partition(':')
            sp.elem.append((el.strip(), int(float(cnt))))
        sp.mw = sum(ELEM_WT[el.lower()] * n for el, n in sp.elem)
        sp.lo, sp.hi = list(lo[2]), list(hi[2])
        sp.Trange = [lo[1][0], lo[1][1], hi[1][1]]
        st['species'][name] = sp

    def _options(options):
        opts = [options] if isinstance(options, str) else list(options or [])
        return any(o.strip().lower() == 'duplicate' for o in opts)

    def _add(kind, equation, **kw):
        st['reactions'].append((kind, equation, kw))

    def reaction(equation='', kf=None, id='', order='', options=(), **kw):
        _add('elementary', equation, kf=kf if kf is not None else kw.get('rate_coeff'), dup=_options(options))

    def three_body_reaction(equation='', kf=None, efficiencies='', id='', options=(), **kw):
        _add('three_body', equation, kf=kf if kf is not None else kw.get('rate_coeff'), eff=efficiencies, dup=_options(options))

    def falloff_reaction(equation='', kf=None, kf0=None, efficiencies='', falloff=None, id='', options=(), **kw):
        _add('falloff', equation, kf=kf, kf0=kf0, eff=efficiencies, falloff=falloff, dup=_options(options))

    def chemically_activated_reaction(equation='', kLow=None, kHigh=None, efficiencies='', falloff=None, id='',
                                      options=(), **kw):
        _add('chem_act', equation, kLow=kLow, kHigh=kHigh, eff=efficiencies, falloff=falloff, dup=_options(options))

    def pdep_arrhenius(equation='', *rates, **kw):
        _add('plog', equation, rates=rates, dup=_options(kw.get('options', ())))

    def chebyshev_reaction(equation='', Tmin=(300.0, 'K'), Tmax=(2500.0, 'K'), Pmin=(0.001, 'atm'), Pmax=(100.0, 'atm'),
                           coeffs=(), id='', options=(), **kw):
        _add('cheb', equation, Tmin=Tmin, Tmax=Tmax, Pmin=Pmin, Pmax=Pmax, coeffs=coeffs, dup=_options(options))

    ignore = lambda *a, **k: None
    env = dict(units=units, ideal_gas=ideal_gas, IdealGas=ideal_gas, NASA=NASA, species=species, reaction=reaction,
               three_body_reaction=three_body_reaction, falloff_reaction=falloff_reaction,
               chemically_activated_reaction=chemically_activated_reaction, pdep_arrhenius=pdep_arrhenius,
               chebyshev_reaction=chebyshev_reaction, Arrhenius=_Arrhenius,
               Troe=lambda A=0.0, T3=0.0, T1=0.0, T2=None: _Falloff('troe', [A, T3, T1] + ([T2] if T2 is not None else [])),
               SRI=lambda A=0.0, B=0.0, C=0.0, D=None, E=None: _Falloff('sri', [A, B, C] + ([D, E] if D is not None else [])),
               Lindemann=lambda: None, state=ignore, gas_transport=ignore, validate=ignore, element=ignore,
               OneAtm=PA, OneBar=1e5, __builtins__={'range': range, 'len': len, 'float': float, 'int': int, 'dict': dict,
                                                   'list': list, 'tuple': tuple, 'True': True, 'False': False, 'None': None})
    exec(compile(text, '<cti>', 'exec'), env)      # (a .cti file is a Python script by definition of the format)

    if st['order'] is None:
        raise ValueError('no ideal_gas(...) phase in the .cti file')
    unknown = [n for n in st['order'] if n not in st['species']]
    if unknown:
        raise ValueError('species without a species(...) entry: ' + ', '.join(unknown))
    specs: List[Species] = [st['species'][n] for n in st['order']]
    known = set(st['order'])
    if st['length'] not in _LEN or st['quantity'] not in _QTY or st['act_energy'] not in _ENERGY:
        raise ValueError('units(length=%r, quantity=%r, act_energy=%r): not understood'
                         % (st['length'], st['quantity'], st['act_energy']))
    cunit = _LEN[st['length']] ** 3 / _QTY[st['quantity']]      # one (length^3 / quantity) in m^3 / kmol
    efac = ACT_ENERGY_FACT[_ENERGY[st['act_energy']]]

    def conv(k, order):
        A, b, E = _arr(k)
        return [_plain(A, 'pre-exponential factor') * cunit ** (order - 1.0), _plain(b, 'temperature exponent'),
                _plain(E, 'activation energy') * efac]

    def third_bodies(rx, eff_text, default=1.0):
        eff = dict(_efficiencies(eff_text))
        for n in eff:
            if n not in known:
                raise ValueError('efficiency of unknown species %r' % n)
        for n in st['order']:                                  # (species order: mech_interpret.py:984-989)
            if n in eff:
                rx.thd_body_eff.append((n, eff[n]))
            elif default != 1.0:
                rx.thd_body_eff.append((n, default))

    def falloff_pars(rx, f):
        if f is None:
            return
        if f.kind == 'troe':
            rx.troe = True
            p = [float(x) for x in f.pars]
            if p[1] == 0:
                p[1] = 1e-30
            if p[2] == 0:
                p[2] = 1e-30
            rx.troe_par = p
        else:
            rx.sri = True
            rx.sri_par = [float(x) for x in f.pars]

    reacs: List[Reaction] = []
    for kind, eq, kw in st['reactions']:
        rev, r_sp, r_nu, p_sp, p_nu, thd_m, pdep_sp = _equation(eq, known)
        order = float(sum(r_nu))
        if kind == 'elementary':
            A, b, E = conv(kw['kf'], order)
            if A == 0.0:
                continue
            rx = Reaction(rev, r_sp, r_nu, p_sp, p_nu, A, b, E)
        elif kind == 'three_body':
            A, b, E = conv(kw['kf'], order + 1.0)
            rx = Reaction(rev, r_sp, r_nu, p_sp, p_nu, A, b, E)
            rx.thd_body = True
            third_bodies(rx, kw['eff'])
        elif kind in ('falloff', 'chem_act'):
            if kind == 'falloff':
                A, b, E = conv(kw['kf'], order)
                rx = Reaction(rev, r_sp, r_nu, p_sp, p_nu, A, b, E)
                rx.low = conv(kw['kf0'], order + 1.0)
            else:
                A, b, E = conv(kw['kLow'], order)
                rx = Reaction(rev, r_sp, r_nu, p_sp, p_nu, A, b, E)
                rx.high = conv(kw['kHigh'], order - 1.0)
            rx.pdep = True
            if pdep_sp is not None and pdep_sp != 'M':
                if pdep_sp not in known:
                    raise ValueError('falloff collider %r is not a species' % pdep_sp)
                rx.pdep_sp = pdep_sp
            else:
                third_bodies(rx, kw['eff'])
            falloff_pars(rx, kw['falloff'])
        elif kind == 'plog':
            rx = Reaction(rev, r_sp, r_nu, p_sp, p_nu, 0.0, 0.0, 0.0)
            rx.plog = True
            for r in kw['rates']:
                P = _with_unit(r[0], _PRES, 'pdep_arrhenius pressure')
                A, b, E = conv(list(r[1:4]), order)
                rx.plog_par.append([P, A, b, E])
        else:
            rx = Reaction(rev, r_sp, r_nu, p_sp, p_nu, 0.0, 0.0, 0.0)
            rx.cheb = True
            co = [[float(x) for x in row] for row in kw['coeffs']]
            rx.cheb_n_temp, rx.cheb_n_pres = len(co), len(co[0]) if co else 0
            if rx.cheb_n_temp < 3 or rx.cheb_n_pres < 2 or any(len(r) != rx.cheb_n_pres for r in co):
                raise ValueError('chebyshev_reaction %r: at least 3 x 2 coefficients in a full table are expected' % eq)
            rx.cheb_par = [x for row in co for x in row]
            rx.cheb_par[0] += math.log10(cunit ** (order - 1.0))
            rx.cheb_tlim = [_with_unit(kw['Tmin'], {'k': 1.0}, 'Tmin'), _with_unit(kw['Tmax'], {'k': 1.0}, 'Tmax')]
            rx.cheb_plim = [_with_unit(kw['Pmin'], _PRES, 'Pmin'), _with_unit(kw['Pmax'], _PRES, 'Pmax')]
        rx.dup = bool(kw.get('dup'))
        reacs.append(rx)

    return finish_mechanism(list(st['elems'] or []), specs, reacs, last_spec)
