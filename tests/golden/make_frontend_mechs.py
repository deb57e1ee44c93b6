#!/usr/bin/env python3
"""Author the front-end corner mechanisms (mechanism DATA, derived from tests/golden/h2o2.inp by this script):

  fe_kcal.inp / fe_kelvins.inp / fe_kjoules.inp / fe_joules.inp / fe_evolts.inp
      the same 28 reactions with a units keyword on the REACTIONS line (KCAL/MOLE, KELVINS, KJOULES/MOLE,
      JOULES/MOLE, EVOLTS) and the activation energies (also of the LOW line) converted to that unit --
      pyjac/core/mech_interpret.py:42-49, 135-159;
  fe_septherm.inp + fe_septherm.dat
      no THERMO block in the mechanism; a separate thermodynamic database with a plain `THERMO` header (not
      `THERMO ALL`), a common temperature line whose middle temperature is NOT 1000 K, cards in another order than
      the SPECIES list, cards without a third temperature (they take the common one), one card with its own, a card
      of a species the mechanism does not have (skipped) and a repeated card (the first one counts) --
      mech_interpret.py:735-883.  Species with different T_mid put several pre-summed K_c groups on one reaction
      (rate_subs.py:660-809).

Golden vectors for them come from pyJac's generated C through make_golden.py."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r'[+-]?(?:\d+\.?\d*|\.\d+)(?:[eEdD][+-]?\d+)?'
CAL_TO_K = 4.184 / 8.3144621


def split_h2o2():
    text = open(os.path.join(HERE, 'h2o2.inp')).read().splitlines()
    i_th = next(i for i, l in enumerate(text) if l.upper().startswith('THERMO'))
    i_re = next(i for i, l in enumerate(text) if l.upper().startswith('REACTIONS'))
    return text[:i_th], text[i_th:i_re], text[i_re:]


def convert_reactions(lines, keyword, factor):
    """E [cal/mole] -> the unit of `keyword` (factor = unit per cal/mole)."""
    out = ['REACTIONS ' + keyword]
    for l in lines[1:]:
        m = re.match(r'^(\s*\S.*?\S)\s+(%s)\s+(%s)\s+(%s)\s*$' % (NUM, NUM, NUM), l)
        low = re.match(r'^(\s*LOW\s*/\s*)(%s)\s+(%s)\s+(%s)(\s*/\s*)$' % (NUM, NUM, NUM), l, re.I)
        if low:
            out.append('%s%s %s %.16E%s' % (low.group(1), low.group(2), low.group(3), float(low.group(4)) * factor, low.group(5)))
        elif m and ('=' in m.group(1)):
            out.append('%-40s %s %s %.16E' % (m.group(1), m.group(2), m.group(3), float(m.group(4)) * factor))
        else:
            out.append(l)
    return out


def main():
    head, thermo, reac = split_h2o2()
    for name, kw, f in (('fe_kcal', 'KCAL/MOLE', 1e-3), ('fe_kelvins', 'KELVINS', CAL_TO_K),
                        ('fe_kjoules', 'KJOULES/MOLE', 4.184e-3), ('fe_joules', 'JOULES/MOLE', 4.184),
                        ('fe_evolts', 'EVOLTS', CAL_TO_K / 11595.0)):
        with open(os.path.join(HERE, name + '.inp'), 'w') as fh:
            fh.write('\n'.join(head + thermo + convert_reactions(reac, kw, f)) + '\n')
    # ---- separate thermo database ----
    cards = {}
    body = thermo[2:-1]
    for i in range(0, len(body), 4):
        cards[body[i][:18].split()[0]] = body[i:i + 4]

    def retemp(card, tlo, thi, tmid):
        l0 = card[0]
        t = '%10.3f%10.3f' % (tlo, thi) + ('%10.3f' % tmid if tmid else ' ' * 10)
        return [l0[:45] + t + l0[75:]] + card[1:]
    n2 = ['N2                121286N   2               G   300.000  5000.000  1000.000    1',
          ' 0.02926640E+02 0.14879768E-02-0.05684760E-05 0.10097038E-09-0.06753351E-13    2',
          '-0.09227977E+04 0.05980528E+02 0.03298677E+02 0.14082404E-02-0.03963222E-04    3',
          ' 0.05641515E-07-0.02444854E-10-0.10208999E+04 0.03950372E+02                   4']
    order = ['AR', 'H2O2', 'HO2', 'H2O', 'OH', 'H2', 'H', 'O2', 'O']
    db = ['! separate thermodynamic database for fe_septherm.inp (data derived from h2o2.inp by make_frontend_mechs.py)',
          'THERMO', '   300.000  1200.000  5000.000']
    for sp in order:
        if sp in ('H2O', 'HO2'):
            db += retemp(cards[sp], 200.0, 3500.0, None)          # no third temperature: the common 1200 K
        elif sp == 'OH':
            db += retemp(cards[sp], 200.0, 3500.0, 900.0)         # its own
        else:
            db += cards[sp]
        if sp == 'H2O':
            db += n2                                              # not in the mechanism: skipped
    db += retemp(cards['O'], 200.0, 3500.0, 1500.0)               # a second O card: ignored (the first one counts)
    db += ['END']
    with open(os.path.join(HERE, 'fe_septherm.dat'), 'w') as fh:
        fh.write('\n'.join(db) + '\n')
    with open(os.path.join(HERE, 'fe_septherm.inp'), 'w') as fh:
        fh.write('\n'.join(head + reac) + '\n')


if __name__ == '__main__':
    main()
