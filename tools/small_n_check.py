#!/usr/bin/env python3
"""Odd batch sizes through the attached row-block library (general kernels below one workgroup, the pair-store
kernels' shifted last workgroup above) against the oracle: small_n_check.py <mechanism name> [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import pyjac_amd
from conftest import MECHS, THERMS, jac_scaled_err
from pyjac_amd import synth
from oracle.oracle import Oracle
name = sys.argv[1]
ev = pyjac_amd.Evaluator(MECHS[name], THERMS.get(name))
orc = Oracle(ev.tables)
for n in [int(x) for x in sys.argv[2:]] or [1, 2, 7, 63, 64, 65, 127, 128, 129, 191, 1000, 65537]:
    pres, y = synth.dist_b(n, ev.nsp, seed=n)
    ref = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
    for lay in ('soa', 'aos'):
        if lay == 'soa':
            jac = ev.jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()).cpu().numpy().T
        else:
            ev.use_spec(2)
            jac = ev.jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(np.ascontiguousarray(y.T)).cuda(),
                              y_layout=pyjac_amd.LAYOUT_AOS, jac_layout=pyjac_amd.LAYOUT_AOS).cpu().numpy()
            ev.use_spec(1)
        print('%s %s n=%d %s: nan %d scaled err %.3g' % (name, ev.spec_kernel, n, lay, int(np.isnan(jac).sum()), jac_scaled_err(jac, ref, ev.nsp)), flush=True)
