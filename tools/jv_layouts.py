import sys, torch, numpy as np
sys.path.insert(0, '.')
import pyjac_amd
from pyjac_amd import synth
for mech, n in (('pyjac_amd/data/gri30_shaped.inp', 1000000), ('pyjac_amd/data/usc2_shaped.inp', 200000)):
    ev = pyjac_amd.Evaluator(mech)
    pres, y = synth.dist_b(n, ev.nsp)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
    d_v = torch.randn_like(d_y)
    ya, va = d_y.T.contiguous(), d_v.T.contiguous()
    for lay, (yy, vv) in (('soa', (d_y, d_v)), ('aos', (ya, va))):
        L = pyjac_amd.LAYOUT_SOA if lay == 'soa' else pyjac_amd.LAYOUT_AOS
        w = ev.jacobian_vec(d_p, yy, vv, layout=L)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ev.jacobian_vec(d_p, yy, vv, layout=L, out=w)
        e1.record(); torch.cuda.synchronize()
        print(mech.split('/')[-1], lay, '%.3f ms' % (e0.elapsed_time(e1) / 5), flush=True)
