#!/usr/bin/env python3
"""Phase cycles of a pj_rblk.hip library built with -DPJQ_TIMING (tools/rblk_variants.py build ... D=-DPJQ_TIMING):
per row kernel, mean cycles per wavefront in: prologue, Arrhenius visits, hand-over visits, output phase, epilogue."""
import ctypes, os, sys
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
L = pyjac_amd.LAYOUT_SOA
ev.time_jacobian(d_p, d_y, jac, 2, L, L)
ms = ev.time_jacobian(d_p, d_y, jac, 3, L, L)
print('%.3f ms per step' % ms)
lib = ctypes.CDLL(so)
names = ['prologue', 'visits', 'pre-visits', 'output', 'epilogue', 'ep:cp', 'ep:fence', 'ep:loads']
tot = np.zeros(8)
for part in range(64):
    buf = np.zeros((8, 1024, 8), dtype=np.int64)
    if lib.pj_spec_debug_timing(part, buf.ctypes.data_as(ctypes.c_void_p)) != 0:
        break
    # per lane group (several groups per workgroup: index = group; one group: index = wavefront of the workgroup)
    g = buf.mean(axis=1)                    # [phase][group]
    for q in range(8):
        if g[:, q].sum() > 0:
            print('   group/wave %d: ' % q + '  '.join('%s %7.0f' % (nm, v) for nm, v in zip(names, g[:, q])) + '   sum %8.0f' % g[:, q].sum())
    m = buf.reshape(8, -1).mean(axis=1)
    tot += m
    print('kernel %2d: ' % part + '  '.join('%s %7.0f' % (nm, v) for nm, v in zip(names, m)) + '   sum %8.0f' % m.sum())
print('all      : ' + '  '.join('%s %7.0f' % (nm, v) for nm, v in zip(names, tot)) + '   sum %8.0f' % tot.sum())
