#!/usr/bin/env python3
"""Time every library under pyjac_amd/spec/variants/ (lane-kernel build variants
for the H2/O2+N2 mechanism) on the bench workload."""
import ctypes, glob, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pyjac_amd
from pyjac_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ev = pyjac_amd.Evaluator(os.path.join(ROOT, 'pyjac_amd', 'data', 'h2o2_n2.inp'), specialize='off')
pres, y = synth.dist_a(n, ev.nsp)
d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
out = torch.empty(ev.nsp * ev.nsp * n, dtype=torch.float64, device='cuda')
ref = None
for so in sorted(glob.glob(os.path.join(ROOT, 'pyjac_amd', 'spec', 'variants', '*.so'))):
    _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, so.encode()))
    ev.time_jacobian(d_p, d_y, out, 3)
    ms = min(ev.time_jacobian(d_p, d_y, out, 20) for _ in range(3))
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(ref, out))
    print(json.dumps(dict(variant=os.path.basename(so), ms=round(ms, 4), jac_per_s=round(n / ms * 1e3),
                          frac_hbm=round(n * 888 / ms / 1e6 / 8000, 4), same=same)), flush=True)
