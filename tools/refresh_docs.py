#!/usr/bin/env python3
"""Regenerate the blocks of DESIGN.md and tests/test_gpu_parity.py that quote measured tolerances, from the files that hold
the measurements (tests/golden/self_noise.json) -- so that no tolerance in prose is typed by hand.
    python tools/refresh_docs.py          rewrite the blocks
    python tools/refresh_docs.py --check  exit 1 if a block is out of date (tests/test_host_logic.py runs this)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = (('h2o2_n2', 'H2/O2+N2'), ('gri30_shaped', 'GRI-shaped, 53 species'), ('usc2_shaped', 'USC-shaped, 111 species'),
         ('synth_irrev72', '72 species, mostly irreversible'))


def self_noise_block():
    j = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'self_noise.json')))
    rows = ['| mechanism | states | pyJac `-O3` vs pyJac `-O3 -mfma -ffp-contract=fast`, max thresholded relative difference | entries > 1e-6 per state | bound asserted on kernel vs pyJac (`MX_BIG` = 10 x) |',
            '|---|---|---|---|---|']
    for key, label in NAMES:
        if key in j:
            d = j[key]
            rows.append('| %s | %d | %.3g | %.3g | %s |' % (label, d['states'], d['self_noise'], d['entries_over_1e-6_per_state_fma'],
                                                          '%.3g' % (10 * d['self_noise']) if key != 'h2o2_n2' else 'rtol 1e-6 on every entry'))
    return '\n'.join(rows)


BLOCKS = {'self-noise': self_noise_block}


def refresh(path, check):
    txt = open(path).read()
    out = txt
    for name, fn in BLOCKS.items():
        pat = re.compile(r'(<!-- generated:%s:begin[^>]*-->\n)(.*?)(\n<!-- generated:%s:end -->)' % (name, name), re.S)
        if pat.search(out):
            out = pat.sub(lambda m: m.group(1) + fn() + m.group(3), out)
    if out != txt:
        if check:
            return False
        open(path, 'w').write(out)
    return True


def main():
    check = '--check' in sys.argv
    ok = all(refresh(os.path.join(ROOT, f), check) for f in ('DESIGN.md',))
    if not ok:
        sys.stderr.write('generated blocks out of date: run python tools/refresh_docs.py\n')
        sys.exit(1)


if __name__ == '__main__':
    main()
