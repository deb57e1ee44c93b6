"""Cantera .cti input (pyjac_amd/cti.py; the reference reads the format through Cantera, mech_interpret.py:886-1137): the
mechanisms of the Chemkin golden files written out as .cti text -- in SI units (exact round trip) and in the usual
cm / mol / cal units (unit conversion of every rate form) -- must give the tables of the Chemkin files; the reference's own
data/h2o2.cti, where the reference tree is present, must give the tables of the same mechanism in Chemkin form."""
import os

import numpy as np
import pytest

from conftest import MECHS, THERMS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _to_cti(mech, length, quantity, act_energy):
    """The model of pyjac_amd.mechanism as .cti text (test infrastructure: the inverse of pyjac_amd.cti.parse_cti)."""
    from pyjac_amd.mechanism import ACT_ENERGY_FACT
    from pyjac_amd.cti import _ENERGY, _LEN, _QTY
    cunit = _LEN[length] ** 3 / _QTY[quantity]
    efac = ACT_ENERGY_FACT[_ENERGY[act_energy]]
    names = mech.species_names()
    out = ["units(length='%s', time='s', quantity='%s', act_energy='%s')" % (length, quantity, act_energy),
           "ideal_gas(name='gas', elements='%s', species='''%s''', reactions='all')" % (' '.join(mech.elems), ' '.join(names))]
    for s in mech.specs:
        out.append("species(name=%r, atoms=%r, thermo=(NASA([%r, %r], %r), NASA([%r, %r], %r)))" % (
            s.name, ' '.join('%s:%d' % (e, n) for e, n in s.elem), s.Trange[0], s.Trange[1], list(s.lo), s.Trange[1], s.Trange[2],
            list(s.hi)))

    def side(sp, nu, extra):
        return ' + '.join(('%r %s' % (n, names[i])) if n != 1 else names[i] for i, n in zip(sp, nu)) + extra

    def k(par, order):
        return [par[0] / cunit ** (order - 1.0), par[1], par[2] / efac]

    for r in mech.reacs:
        order = float(sum(r.reac_nu))
        extra = ' + M' if r.thd_body else (' (+ %s)' % (names[r.pdep_sp] if r.pdep_sp is not None else 'M')) if r.pdep else ''
        eq = side(r.reac, r.reac_nu, extra) + (' <=> ' if r.rev else ' => ') + side(r.prod, r.prod_nu, extra)
        eff = ' '.join('%s:%r' % (names[i], a) for i, a in r.thd_body_eff)
        opt = ", options='duplicate'" if r.dup else ''
        fall = ''
        if r.troe:
            fall = ', falloff=Troe(A=%r, T3=%r, T1=%r%s)' % (r.troe_par[0], r.troe_par[1], r.troe_par[2],
                                                             (', T2=%r' % r.troe_par[3]) if len(r.troe_par) > 3 else '')
        elif r.sri:
            fall = ', falloff=SRI(A=%r, B=%r, C=%r%s)' % (r.sri_par[0], r.sri_par[1], r.sri_par[2],
                                                          (', D=%r, E=%r' % (r.sri_par[3], r.sri_par[4])) if len(r.sri_par) > 3 else '')
        if r.plog:
            rates = ', '.join('[(%r, "Pa"), %r, %r, %r]' % (p[0], *k(p[1:], order)) for p in r.plog_par)
            out.append('pdep_arrhenius(%r, %s%s)' % (eq, rates, opt))
        elif r.cheb:
            co = list(r.cheb_par)
            co[0] -= np.log10(cunit ** (order - 1.0))
            co = [float(x) for x in co]
            rows = [co[i * r.cheb_n_pres:(i + 1) * r.cheb_n_pres] for i in range(r.cheb_n_temp)]
            out.append('chebyshev_reaction(%r, Tmin=(%r, "K"), Tmax=(%r, "K"), Pmin=(%r, "Pa"), Pmax=(%r, "Pa"), coeffs=%r%s)' % (
                eq, r.cheb_tlim[0], r.cheb_tlim[1], r.cheb_plim[0], r.cheb_plim[1], rows, opt))
        elif r.pdep and r.high:
            out.append('chemically_activated_reaction(%r, kLow=%r, kHigh=%r, efficiencies=%r%s%s)' % (
                eq, k([r.A, r.b, r.E], order), k(r.high, order - 1.0), eff, fall, opt))
        elif r.pdep:
            out.append('falloff_reaction(%r, kf=%r, kf0=%r, efficiencies=%r%s%s)' % (
                eq, k([r.A, r.b, r.E], order), k(r.low, order + 1.0), eff, fall, opt))
        elif r.thd_body:
            out.append('three_body_reaction(%r, %r, efficiencies=%r%s)' % (eq, k([r.A, r.b, r.E], order + 1.0), eff, opt))
        else:
            out.append('reaction(%r, %r%s)' % (eq, k([r.A, r.b, r.E], order), opt))
    return '\n'.join(out) + '\n'


def _tables(mech):
    from pyjac_amd import tables
    return tables.build_tables(mech)


@pytest.mark.parametrize('name', ['h2o2', 'synth_alltypes', 'synth_srichb', 'synth_fracnu', 'synth_mid24', 'fe_septherm'])
@pytest.mark.parametrize('units', [('m', 'kmol', 'k'), ('cm', 'mol', 'cal/mol'), ('cm', 'mol', 'kj/mol')])
def test_cti_text_gives_the_tables_of_the_chemkin_file(name, units, tmp_path):
    from pyjac_amd import mechanism
    ref = mechanism.read_mech(MECHS[name], THERMS.get(name))
    if any(abs(sum(mechanism.ELEM_WT[e.lower()] * n for e, n in s.elem) - s.mw) > 0 for s in ref.specs):
        pytest.skip('element weights given in the Chemkin file: a .cti file cannot carry them')
    # (the Arrhenius triple on the reaction line of a PLOG / CHEB reaction: kept by the Chemkin reader, 0 0 0 through Cantera --
    # mech_interpret.py:1072-1095 -- and used by neither rate form)
    # (third-body efficiencies: in the order of the file through the Chemkin reader, in SPECIES order through Cantera,
    # mech_interpret.py:984-989)
    for r in ref.reacs:
        if r.plog or r.cheb:
            r.A = r.b = r.E = 0.0
        r.thd_body_eff = sorted(r.thd_body_eff)
    path = tmp_path / (name + '.cti')
    path.write_text(_to_cti(ref, *units))
    got = mechanism.read_mech(str(path))
    assert got.species_names() == ref.species_names() and got.n_fwd == ref.n_fwd and got.n_rev == ref.n_rev
    ta, tb = _tables(ref), _tables(got)
    assert np.array_equal(ta.I, tb.I)
    exact = units == ('m', 'kmol', 'k')
    scale = np.maximum(np.abs(ta.D), 1e-300)
    assert (np.abs(ta.D - tb.D) / scale).max() <= (0.0 if exact else 1e-13)


def test_reference_h2o2_cti_matches_its_chemkin_twin():
    """data/h2o2.cti of the reference tree (H2/O2 + AR + N2) against the shipped Chemkin file of the same mechanism."""
    cti = '/root/reference/data/h2o2.cti'
    if not os.path.exists(cti):
        pytest.skip('reference tree not present')
    from pyjac_amd import mechanism
    a = mechanism.read_mech(os.path.join(ROOT, 'pyjac_amd', 'data', 'h2o2_n2.inp'))
    b = mechanism.read_mech(cti)
    assert a.species_names() == b.species_names() and b.species_names()[-1] == 'N2'
    ta, tb = _tables(a), _tables(b)
    assert np.array_equal(ta.I, tb.I)
    assert (np.abs(ta.D - tb.D) / np.maximum(np.abs(ta.D), 1e-300)).max() < 1e-14


def test_cti_reader_conventions_and_errors():
    from pyjac_amd.cti import parse_cti
    head = """units(length='cm', time='s', quantity='mol', act_energy='cal/mol')
ideal_gas(name='g', elements='H O Ar', species='H2 O2 H2O AR', reactions='all')
"""
    th = "thermo=(NASA([300.0, 1000.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]), NASA([1000.0, 3000.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]))"
    sp = ''.join("species(name='%s', atoms='%s', %s)\n" % (n, a, th) for n, a in
                 (('H2', 'H:2'), ('O2', 'O:2'), ('H2O', 'H:2 O:1'), ('AR', 'Ar:1')))
    m = parse_cti(head + sp + """
reaction('2 H2 + O2 <=> 2 H2O', [1.0e12, 0.5, 1000.0])
reaction('H2 + O2 => H2O', [0.0, 0.0, 0.0])
three_body_reaction('H2 + M <=> H2 + M', Arrhenius(2.0e15, 0.0, 0.0), efficiencies='AR:0.5 H2O:6.0')
falloff_reaction('H2 + O2 (+ AR) <=> H2O (+ AR)', kf=[1.0e10, 0.0, 0.0], kf0=[1.0e16, 0.0, 0.0], falloff=Troe(A=0.5, T3=0.0, T1=100.0))
pdep_arrhenius('H2 + O2 <=> H2O', [(0.1, 'atm'), 1.0e10, 0.0, 0.0], [(10.0, 'atm'), 1.0e11, 0.0, 0.0])
""")
    assert m.species_names()[-1] == 'AR' and m.n_fwd == 4            # the zero-A reaction is dropped (mech_interpret.py:1109-1111)
    r0, r1, r2, r3 = m.reacs
    assert r0.reac_nu == [2, 1] and abs(r0.A - 1.0e12 * 1e-3 ** 2) < 1e-3 and abs(r0.E - 1000.0 * 4.184 / 8.3144621) < 1e-9
    assert r1.thd_body and abs(r1.A - 2.0e15 * 1e-3) < 1e-3
    assert sorted(r1.thd_body_eff) == sorted([(m.species_names().index('H2O'), 6.0), (m.species_names().index('AR'), 0.5)])
    assert r2.pdep and r2.pdep_sp == m.species_names().index('AR') and r2.troe_par[1] == 1e-30 and abs(r2.low[0] - 1.0e16 * 1e-6) < 1e-3
    assert r3.plog and abs(r3.plog_par[0][0] - 10132.5) < 1e-9 and abs(r3.plog_par[1][1] - 1.0e11 * 1e-3) < 1e-6
    with pytest.raises(ValueError):
        parse_cti(head + sp + "reaction('H2 + XX <=> H2O', [1.0, 0.0, 0.0])\n")
    with pytest.raises(ValueError):
        parse_cti(head + "species(name='H2', atoms='H:2', thermo=None)\n")


@pytest.mark.gpu
def test_cti_mechanism_on_the_gpu(tmp_path):
    """The evaluator built from .cti text gives the Jacobians of the evaluator built from the Chemkin file."""
    import torch
    import pyjac_amd
    from pyjac_amd import mechanism, synth
    inp = os.path.join(ROOT, 'pyjac_amd', 'data', 'h2o2_n2.inp')
    path = tmp_path / 'h2o2_n2.cti'
    path.write_text(_to_cti(mechanism.read_mech(inp), 'cm', 'mol', 'cal/mol'))
    a, b = pyjac_amd.Evaluator(inp), pyjac_amd.Evaluator(str(path))
    pres, y = synth.dist_a(777, a.nsp)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
    ja, jb = a.jacobian(d_p, d_y).cpu().numpy(), b.jacobian(d_p, d_y).cpu().numpy()
    assert np.isfinite(jb).all()
    assert (np.abs(ja - jb) <= 1e-10 * np.abs(ja) + 1e-13 * np.abs(ja).max(axis=0, keepdims=True)).all()
