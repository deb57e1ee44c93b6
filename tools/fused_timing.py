#!/usr/bin/env python3
"""Per-phase cycle split of k_fused (debug build of pj_rows.hip with -DPJR_TIMING)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pyjac_amd
from pyjac_amd import _lib, synth
mech, n, so = sys.argv[1], int(sys.argv[2]), os.path.abspath(sys.argv[3])
ev = pyjac_amd.Evaluator(mech, specialize='off')
_lib.check(_lib.lib().pj_mech_attach_spec(ev._h, so.encode()))
pres, y = synth.dist_b(n, ev.nsp)
d_p = torch.from_numpy(pres).cuda(); d_y = torch.from_numpy(y).cuda()
out = torch.empty(ev.nsp**2 * n, dtype=torch.float64, device='cuda')
S = pyjac_amd.LAYOUT_SOA
ev.time_jacobian(d_p, d_y, out, 1, S, S)
ms = ev.time_jacobian(d_p, d_y, out, 1, S, S)
L = ctypes.CDLL(so)
buf = np.zeros((8, 4, 512), dtype=np.int64)
rc = L.pj_spec_debug_timing(buf.ctypes.data_as(ctypes.c_void_p)); assert rc == 0
wgs = int((buf[4].sum(axis=0) > 0).sum())
tiles = -(-n // 64) / wgs
names = ['load', 'rates', 'barrier', 'dTcol', 'rows-tail', 'energy', 'rows-visits', 'rows-output']
print('ms', round(ms, 3), 'wgs', wgs, 'tiles/wg %.1f' % tiles)
tot = 0
for ph, nm in enumerate(names):
    per = buf[ph][:, :wgs] / tiles           # cycles per tile, [wave][wg]
    print('%8s  per-wave mean %s   all %9.0f' % (nm, np.round(per.mean(axis=1)).astype(int), per.mean()))
    tot += per.mean()
print('   total per tile %9.0f cycles (clock64 ticks)' % tot)
