#!/usr/bin/env python3
"""Exactly <steps> evaluations of <n> synthetic states (for rocprofv3 passes):
one_step.py <mech> <n> <steps> [lane|rblk|table] [jac|aos|rates|dydt]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pyjac_amd
from pyjac_amd import synth
mech, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else None
what = sys.argv[5] if len(sys.argv) > 5 else 'jac'
ev = pyjac_amd.Evaluator(mech, specialize='off')
if os.environ.get('PJ_ONE_STEP_LIB'):       # an explicit library (a variant of tools/rblk_variants.py)
    from pyjac_amd import _lib
    _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, os.environ['PJ_ONE_STEP_LIB'].encode()))
    ev.attached_spec = os.environ['PJ_ONE_STEP_LIB']
elif kind != 'table':
    assert ev.specialize(build=False, kind=kind), 'no prebuilt specialisation'
pres, y = (synth.dist_a if 'h2o2_n2' in mech else synth.dist_b)(n, ev.nsp)
d_p = torch.from_numpy(pres).cuda(); d_y = torch.from_numpy(y).cuda()
if what == 'jac':
    out = torch.empty((ev.nsp**2, n), dtype=torch.float64, device='cuda')
    for _ in range(steps):
        ev.jacobian(d_p, d_y, out=out)
elif what == 'aos':
    # pyJac's per-state C layout (state-major NSP x NSP blocks), forced through the attached library
    ev.use_spec(2)
    out = torch.empty((n, ev.nsp**2), dtype=torch.float64, device='cuda')
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev.jacobian(d_p, d_y, out=out, jac_layout=pyjac_amd.LAYOUT_AOS)
    t0.record()
    for _ in range(steps):
        ev.jacobian(d_p, d_y, out=out, jac_layout=pyjac_amd.LAYOUT_AOS)
    t1.record(); torch.cuda.synchronize()
    print('aos: %.3f ms per step' % (t0.elapsed_time(t1) / steps))
    out = out.T
else:
    import ctypes
    from pyjac_amd import _lib
    rows = dict(conc=ev.nsp, fwd=ev.n_fwd, rev=max(ev.n_rev, 1), pres_mod=max(ev.n_pres_mod, 1), spec_rates=ev.nsp, dy=ev.nsp)
    bufs = {k: torch.empty((r, n), dtype=torch.float64, device='cuda') for k, r in rows.items()}
    p = (lambda k: bufs[k].data_ptr()) if what == 'rates' else (lambda k: bufs[k].data_ptr() if k == 'dy' else None)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(steps):
        _lib.check(_lib.lib().pj_eval_rates_dev(ev._h, n, d_p.data_ptr(), d_y.data_ptr(), 0, p('conc'), p('fwd'), p('rev'),
                                                p('pres_mod'), p('spec_rates'), p('dy'), st))
    out = bufs['dy']
torch.cuda.synchronize()
print(ev.spec_kernel or 'k_eval', n, steps, bool(torch.isfinite(out[:, ::997]).all()), 'library', os.path.basename(ev.attached_spec or '-'))
