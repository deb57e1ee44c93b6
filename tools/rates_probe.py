import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import pyjac_amd, ctypes
from pyjac_amd import synth, _lib
for mech, n in (('pyjac_amd/data/gri30_shaped.inp', 200000), ('pyjac_amd/data/usc2_shaped.inp', 50000)):
    ev = pyjac_amd.Evaluator(mech)
    pres, y = synth.dist_b(n, ev.nsp)
    for lay, yy in ((pyjac_amd.LAYOUT_SOA, y), (pyjac_amd.LAYOUT_AOS, np.ascontiguousarray(y.T))):
        d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(yy).cuda()
        dy = torch.empty((ev.nsp, n), dtype=torch.float64, device='cuda')
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda: _lib.check(_lib.lib().pj_eval_rates_dev(ev._h, n, d_p.data_ptr(), d_y.data_ptr(), lay, None, None, None, None, None, dy.data_ptr(), st))
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(mech.split('/')[-1], 'layout', lay, 'dydt ms %.3f  states/s %.3g' % (ms, n / ms * 1e3))
