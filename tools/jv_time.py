#!/usr/bin/env python3
"""ms per 1e6 fused w = J v products of variant libraries (tools/rblk_variants.py build ...):
jv_time.py <mech> <n> <tag> [tag ...]   ('rblk': the shipped library)"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import pyjac_amd
from pyjac_amd import synth, _lib
mech, n = sys.argv[1], int(sys.argv[2])
stem = os.path.splitext(os.path.basename(mech))[0]
ev0 = pyjac_amd.Evaluator(mech, specialize='off')
pres, y = synth.dist_b(n, ev0.nsp)
d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
d_v = torch.randn_like(d_y); d_w = torch.empty_like(d_y)
ref = None
for tag in sys.argv[3:]:
    ev = pyjac_amd.Evaluator(mech, specialize='off')
    if tag == 'rblk':      # whatever prebuilt library of that family exists (any build digest)
        pat = os.path.basename(ev.spec_path('rblk')).rsplit('_', 1)[0] + '_*.so'
        _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, sorted(glob.glob(os.path.join(ROOT, 'pyjac_amd', 'spec', pat)))[-1].encode()))
    else:
        _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, os.path.join(ROOT, 'pyjac_amd', 'spec', 'var', '%s_%s.so' % (stem, tag)).encode()))
    for _ in range(2):
        ev.jacobian_vec(d_p, d_y, d_v, out=d_w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ev.jacobian_vec(d_p, d_y, d_v, out=d_w)
    e1.record(); torch.cuda.synchronize()
    s = d_w[:, ::4999].cpu().numpy()
    if ref is None:
        ref = s
    print('%-12s %.3f ms  max rel diff vs first %.2g' % (tag, e0.elapsed_time(e1) / 5, np.abs(s - ref).max() / np.abs(ref).max()), flush=True)
