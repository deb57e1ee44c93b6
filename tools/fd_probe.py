#!/usr/bin/env python3
"""Time of the finite-difference Jacobian arm (pj_eval_fd_jacobian_dev: NSP + 1 dydt passes, fd_jacob.c protocol)
next to the analytical Jacobian, per mechanism: fd_probe.py <mech> <n>"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pyjac_amd
from pyjac_amd import synth
mech, n = sys.argv[1], int(sys.argv[2])
ev = pyjac_amd.Evaluator(mech)
pres, y = (synth.dist_a if 'h2o2_n2' in mech else synth.dist_b)(n, ev.nsp)
d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
out = torch.empty((ev.nsp ** 2, n), dtype=torch.float64, device='cuda')
res = {}
for name, fn in (('analytical', lambda: ev.jacobian(d_p, d_y, out=out)), ('finite differences', lambda: ev.fd_jacobian(d_p, d_y, out=out))):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record(); torch.cuda.synchronize()
    res[name] = e0.elapsed_time(e1) / 3
print('%s (%s, %d species), %d states: analytical %.3f ms, finite differences %.3f ms (%.1fx)' % (
    os.path.basename(mech), ev.spec_kernel, ev.nsp, n, res['analytical'], res['finite differences'],
    res['finite differences'] / res['analytical']))
