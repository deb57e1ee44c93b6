#!/usr/bin/env python3
"""Jacobians/s of every planner-sweep mechanism (tests/golden/sweep/*.inp + the three shipped ones) with its prebuilt
library: one line per mechanism -- species, reactions, kernel family, states, ms per launch, share of the 8 TB/s roofline
on the algorithmic bytes.  sweep_perf.py [states-budget in GB of Jacobian, default 6]"""
import glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pyjac_amd
from pyjac_amd import synth
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
mechs = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'sweep', '*.inp'))) + \
    [os.path.join(ROOT, 'pyjac_amd', 'data', f) for f in ('h2o2_n2.inp', 'gri30_shaped.inp', 'usc2_shaped.inp')]
for mech in mechs:
    ev = pyjac_amd.Evaluator(mech, specialize='off')
    if not ev.specialize(build=False):
        print('%-18s no prebuilt library' % os.path.basename(mech)); continue
    n = int(min(1 << 20, gb * 1e9 / (8 * ev.nsp ** 2))) // 4096 * 4096
    pres, y = (synth.dist_a if 'h2o2_n2' in mech else synth.dist_b)(n, ev.nsp)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
    jac = torch.empty((ev.nsp ** 2, n), dtype=torch.float64, device='cuda')
    L = pyjac_amd.LAYOUT_SOA
    ev.time_jacobian(d_p, d_y, jac, 2, L, L)
    ms = min(ev.time_jacobian(d_p, d_y, jac, 4, L, L) for _ in range(2))
    bj = ev.jacobian_bytes_per_state
    print('%-18s %3d sp %4d rxn  %-8s %8d states  %8.3f ms  %.3g Jac/s  frac %.3f  finite %s' % (
        os.path.basename(mech), ev.nsp, ev.n_fwd, ev.spec_kernel, n, ms, n / ms * 1e3, n * bj / ms / 1e6 / 8000,
        bool(torch.isfinite(jac[:, ::997]).all())), flush=True)
    ev.close(); del jac, d_p, d_y
