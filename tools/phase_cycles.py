#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_eval (debug build libpyjac_hip_timing.so, -DPJ_TIMING)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pyjac_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'pyjac_amd', 'libpyjac_hip_timing.so')
import pyjac_amd
from pyjac_amd import synth
mech, n, ts, nt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ev = pyjac_amd.Evaluator(mech, specialize='off'); ev.set_launch(ts, nt)
pres, y = synth.dist_b(n, ev.nsp)
aos = ts < 16
d_p = torch.from_numpy(pres).cuda(); d_y = torch.from_numpy(np.ascontiguousarray(y.T) if aos else y).cuda()
out = torch.empty(ev.nsp**2 * n, dtype=torch.float64, device='cuda'); dbg = torch.zeros(640, dtype=torch.float64, device='cuda')
L = _lib.lib(); f = L.pj_debug_phase_cycles; f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_void_p]*2 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lay = 1 if aos else 0
for _ in range(2):
    _lib.check(f(ev._h, n, d_p.data_ptr(), d_y.data_ptr(), lay, out.data_ptr(), lay, dbg.data_ptr()))
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(64, 10)
tiles = -(-n // ts) / min(-(-n // ts), 256 * 8)
names = ['0a', 'zero', '0b', '0c', 'P2', 'scatter', 'fin1', 'fin2', 'energy', 'block']
print(ev.get_launch(), 'tiles/block %.1f' % tiles)
for i, nm in enumerate(names):
    print('%8s %10.0f cycles/tile  %5.1f%%' % (nm, d[:, i].mean() / tiles, 100 * d[:, i].mean() / d.mean(0).sum()))
print('   total %10.0f cycles/tile' % (d.mean(0).sum() / tiles))
