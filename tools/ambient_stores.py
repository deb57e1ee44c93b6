#!/usr/bin/env python3
"""Does the store traffic of OTHER wavefronts slow the row kernels down?  The -DPJQ_NO_STORE build of the GRI-shaped library
(tools/rblk_variants.py build ... nostore:D=-DPJQ_NO_STORE) runs on one stream, alone and next to a stream of plain fill
kernels (pure HBM writes, a few registers per lane: they fit beside the row kernels' wavefronts) on another.
ambient_stores.py <mech> <n> <no-store library>"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pyjac_amd
from pyjac_amd import synth, _lib
mech, n, so = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ev = pyjac_amd.Evaluator(mech, specialize='off')
_lib.check(_lib.lib().pj_mech_attach_spec(ev._h, so.encode()))
pres, y = synth.dist_b(n, ev.nsp)
d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
jac = torch.empty((ev.nsp ** 2, n), dtype=torch.float64, device='cuda')
sink = torch.empty(2 * 1024 ** 3, dtype=torch.float64, device='cuda')        # 16 GB
L = pyjac_amd.LAYOUT_SOA
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
small = torch.rand(1 << 20, dtype=torch.float64, device='cuda')               # 8 MB: L2-resident
def ambient(kind):
    if kind == 'writes':
        sink.fill_(1.0)                     # 16 GB of HBM writes
    elif kind == 'reads':
        sink.sum()                          # 16 GB of HBM reads
    else:
        for _ in range(40):                 # arithmetic on an L2-resident array: issue slots, hardly any HBM traffic
            small.sin_()
def run(fill, reps=6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fill:
        with torch.cuda.stream(sb):
            f0.record()
            for _ in range(3 * reps):
                ambient(fill)
            f1.record()
    with torch.cuda.stream(sa):
        e0.record()
        for _ in range(reps):
            ev.jacobian(d_p, d_y, out=jac)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (f0.elapsed_time(f1) / (3 * reps) if fill else float('nan'))
with torch.cuda.stream(sa):
    ev.jacobian(d_p, d_y, out=jac)
torch.cuda.synchronize()
t0 = time.time(); sink.fill_(0.0); torch.cuda.synchronize(); t_fill = (time.time() - t0) * 1e3
print('fill alone: %.3f ms per 16 GB = %.2f TB/s' % (t_fill, 16 * 1.073741824 / t_fill))
for fill in (0, 'writes', 0, 'reads', 0, 'arithmetic', 'writes', 'reads'):
    a, b = run(fill)
    print('no-store row kernels: %.3f ms per step%s' % (a, (' next to ambient %s of %.3f ms each%s' % (
        fill, b, (' (%.2f TB/s)' % (16 * 1.073741824 / b)) if fill != 'arithmetic' else '')) if fill else ' alone'))
