#!/usr/bin/env python3
"""Where does a kernel touch scratch memory?  scratch_sites.py <lib.so|.o> [kernel-substring]
Per code object with a matching kernel: number of scratch loads / stores, how many use a run-time (SGPR) offset -- an array
with a variable index, which the optimiser could not keep in registers -- the offsets touched and where in the kernel."""
import collections, os, re, subprocess, sys, tempfile
LLVM = '/opt/rocm/lib/llvm/bin'
so, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'k_rblk')
d = tempfile.mkdtemp(prefix='ss_', dir='/tmp')
fat = os.path.join(d, 'fat')
subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so])
data = open(fat, 'rb').read()
idx = [m.start() for m in re.finditer(b'\x7fELF', data)]
for n, i in enumerate(idx):
    elf = os.path.join(d, 'co%d.elf' % n)
    open(elf, 'wb').write(data[i:idx[n + 1] if n + 1 < len(idx) else len(data)])
    notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', elf], capture_output=True, text=True).stdout
    if pat not in notes:
        continue
    L = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', elf], capture_output=True, text=True).stdout.split('\n')
    sc = [(j, l) for j, l in enumerate(L) if 'scratch_' in l]
    if not sc:
        print('co %d: no scratch access' % n)
        continue
    dyn = sum(1 for _, l in sc if re.search(r', s\d+', l.split('//')[0]))
    offs = collections.Counter()
    for _, l in sc:
        m = re.search(r'offset:(\d+)', l)
        offs[(int(m.group(1)) if m else 0, 'ld' if 'load' in l else 'st')] += 1
    where = collections.Counter((j * 10) // len(L) for j, _ in sc)
    print('co %d: %d lines, %d scratch accesses (%d with a run-time offset); by tenth of the kernel: %s' % (n, len(L), len(sc), dyn, sorted(where.items())))
    hot = sorted(offs.items(), key=lambda x: -x[1])[:8]
    print('   most used: ' + ', '.join('%s@%d x%d' % (k[1], k[0], v) for k, v in hot))
