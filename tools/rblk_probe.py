#!/usr/bin/env python3
"""GPU probe of the pj_rblk.hip libraries: parity against the oracle on small batches, then timing of
the GRI-shaped 1e6-state batch against the pj_rows.hip library (same process, HIP events)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pyjac_amd
from pyjac_amd import synth
from conftest import MECHS, jac_scaled_err, thresholded_rel_err
from oracle.oracle import Oracle

for name in sys.argv[1].split(','):
    ev = pyjac_amd.Evaluator(MECHS[name], specialize='off')
    if not ev.specialize(build=False, kind='rblk'):
        print(name, 'no rblk library'); continue
    n = 777
    pres, y = synth.dist_b(n, ev.nsp)
    orc = Oracle(ev.tables)
    for sum_last in (0, 1):
        ev.set_sum_last_species(sum_last) if hasattr(ev, 'set_sum_last_species') else pyjac_amd._lib.lib().pj_mech_set_sum_last_species(ev._h, sum_last)
        orc.lib.pjo_set_sum_last_species(sum_last)
        ref = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
        orc.lib.pjo_set_sum_last_species(0)
        jac = torch.full((ev.nsp ** 2, n), float('nan'), dtype=torch.float64, device='cuda')
        ev.jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda(), out=jac)
        torch.cuda.synchronize()
        got = jac.cpu().numpy().T
        mx, fro = thresholded_rel_err(got, ref)
        print('%-16s %s sum_last=%d nan=%d scaled_err=%.3g thresholded_max=%.3g fro=%.3g' % (
            name, ev.spec_kernel, sum_last, int(np.isnan(got).sum()), jac_scaled_err(got, ref, ev.nsp), mx, fro), flush=True)

if len(sys.argv) > 2:
    name, n = sys.argv[2], int(sys.argv[3])
    kinds = sys.argv[4].split(',') if len(sys.argv) > 4 else ['rows', 'rblk']
    pres, y = synth.dist_b(n, pyjac_amd.Evaluator(MECHS[name], specialize='off').nsp)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
    outs = {}
    for kind in kinds:
        ev = pyjac_amd.Evaluator(MECHS[name], specialize='off')
        if not ev.specialize(build=False, kind=kind):
            print(kind, 'missing'); continue
        jac = torch.empty((ev.nsp ** 2, n), dtype=torch.float64, device='cuda')
        L = pyjac_amd.LAYOUT_SOA
        ev.time_jacobian(d_p, d_y, jac, 2, L, L)
        ms = ev.time_jacobian(d_p, d_y, jac, 5, L, L)
        bj = ev.jacobian_bytes_per_state
        print('%s %s n=%d: %.3f ms  %.3g Jac/s  %.0f GB/s  frac %.3f' % (name, ev.spec_kernel, n, ms, n / ms * 1e3,
              n * bj / ms / 1e6, n * bj / ms / 1e6 / 8000), flush=True)
        outs[kind] = jac[:, ::4999].cpu().numpy().T
        del jac
    if len(outs) == 2:
        a, b = outs[kinds[0]], outs[kinds[1]]
        nsp = int(round(a.shape[1] ** 0.5))
        print('cross-check %s vs %s on strided sample: scaled_err %.3g' % (kinds[0], kinds[1], jac_scaled_err(b, a, nsp)))
