#!/usr/bin/env python3
"""Exactly <steps> Jacobian evaluations of <n> synthetic states (for rocprofv3 PMC passes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pyjac_amd
from pyjac_amd import synth
mech, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else None
ev = pyjac_amd.Evaluator(mech, specialize='off')
if kind != 'table':
    assert ev.specialize(build=False, kind=kind), 'no prebuilt specialisation'
pres, y = (synth.dist_a if 'h2o2_n2' in mech else synth.dist_b)(n, ev.nsp)
d_p = torch.from_numpy(pres).cuda(); d_y = torch.from_numpy(y).cuda()
out = torch.empty((ev.nsp**2, n), dtype=torch.float64, device='cuda')
for _ in range(steps):
    ev.jacobian(d_p, d_y, out=out)
torch.cuda.synchronize()
print(ev.spec_kernel or 'k_eval', n, steps, bool(torch.isfinite(out[:, ::997]).all()))
