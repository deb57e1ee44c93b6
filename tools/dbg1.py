import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT,'tests'))
import pyjac_amd
from conftest import MECHS, thresholded_rel_err
g = np.load(os.path.join(ROOT,'tests/golden/h2o2_n2_golden.npz'))
ev = pyjac_amd.Evaluator(MECHS['h2o2_n2'])
pres = g['pres'].copy(); y = np.ascontiguousarray(g['y'].T)
n = pres.size
for ts in (16, 64, 8):
    ev.set_launch(ts if ts != 64 else 32, 256)
    d_p = torch.from_numpy(pres).cuda(); d_y = torch.from_numpy(y).cuda()
    r = ev.rates(d_p, d_y)
    torch.cuda.synchronize()
    for k in ('conc','fwd','rev','pres_mod','spec_rates','dydt'):
        got = r[k].cpu().numpy().T
        ref = g[k]
        rows = ref.shape[1]
        mx, fro = thresholded_rel_err(got[:, :rows], ref)
        print('dev ts', ts, k, '%.2e %.2e' % (mx, fro))
    if ts == 16:
        got = r['spec_rates'].cpu().numpy().T
        print(got[0]); print(g['spec_rates'][0])
        print(got[50]); print(g['spec_rates'][50])
