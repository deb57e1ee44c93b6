#!/usr/bin/env python3
"""Register / LDS / scratch figures of every kernel in a built library (.so with embedded gfx950 code objects):
vgpr, agpr, sgpr counts, spill counts, scratch bytes per lane, LDS bytes -- from the code objects' metadata notes."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def resources(so):
    d = tempfile.mkdtemp(prefix='co_', dir='/tmp')
    fat = os.path.join(d, 'fatbin')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so])
    data = open(fat, 'rb').read()
    idx = [m.start() for m in re.finditer(b'\x7fELF', data)]
    out = []
    for n, i in enumerate(idx):
        elf = os.path.join(d, 'co%d.elf' % n)
        open(elf, 'wb').write(data[i:idx[n + 1] if n + 1 < len(idx) else len(data)])
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', elf], capture_output=True, text=True).stdout
        for blk in notes.split('- .agpr_count:')[1:]:
            blk = '.agpr_count:' + blk
            g = lambda k: (re.findall(re.escape(k) + r':\s+(\S+)', blk) or ['?'])[0]
            out.append(dict(co=n, name=g('.name'), vgpr=g('.vgpr_count'), agpr=g('.agpr_count'), sgpr=g('.sgpr_count'),
                            vgpr_spill=g('.vgpr_spill_count'), sgpr_spill=g('.sgpr_spill_count'),
                            scratch=g('.private_segment_fixed_size'), lds=g('.group_segment_fixed_size'), elf=elf))
    return out


if __name__ == '__main__':
    for r in resources(sys.argv[1]):
        print('%2d %-44s vgpr %s agpr %s sgpr %s  spills v %s s %s  scratch %s B  lds %s B' % (
            r['co'], r['name'][:44], r['vgpr'], r['agpr'], r['sgpr'], r['vgpr_spill'], r['sgpr_spill'], r['scratch'], r['lds']))
