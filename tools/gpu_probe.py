#!/usr/bin/env python3
"""Quick on-box probe: kernel time of the Jacobian launch for a mechanism at
several tile mappings (HIP events via pj_time_jacobian_dev)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyjac_amd  # noqa: E402
from pyjac_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--mech', default=os.path.join(ROOT, 'pyjac_amd', 'data', 'h2o2_n2.inp'))
ap.add_argument('--n', type=int, default=1_000_000)
ap.add_argument('--ts', default='64,32,16,8')
ap.add_argument('--nt', default='256')
ap.add_argument('--layout', default='soa')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--spec', default='off')
a = ap.parse_args()

ev = pyjac_amd.Evaluator(a.mech, specialize=a.spec)
n = a.n
if ev.nsp == 10 and 'h2o2' in a.mech:
    pres, y = synth.dist_a(n, ev.nsp)
else:
    pres, y = synth.dist_b(n, ev.nsp)
d_p = torch.from_numpy(pres).cuda()
lay = pyjac_amd.LAYOUT_SOA if a.layout == 'soa' else pyjac_amd.LAYOUT_AOS
d_y = torch.from_numpy(y if lay == pyjac_amd.LAYOUT_SOA else np.ascontiguousarray(y.T)).cuda()
out = torch.empty(ev.nsp * ev.nsp * n, dtype=torch.float64, device='cuda')
for ts in [int(x) for x in a.ts.split(',')]:
    for nt in [int(x) for x in a.nt.split(',')]:
        try:
            ev.set_launch(ts, nt)
            ev.time_jacobian(d_p, d_y, out, 2, lay, lay)
            ms = ev.time_jacobian(d_p, d_y, out, a.iters, lay, lay)
            gbs = n * ev.jacobian_bytes_per_state / ms / 1e6
            print(json.dumps(dict(spec=ev.has_spec, mech=os.path.basename(a.mech), n=n, ts=ts, nt=nt, layout=a.layout,
                                  lds=ev.get_launch()['lds_bytes'], ms=round(ms, 4),
                                  jac_per_s=round(n / ms * 1e3), GBps=round(gbs, 1),
                                  frac_hbm=round(gbs / 8000, 4))), flush=True)
        except Exception as ex:
            print('ts', ts, 'nt', nt, 'failed:', ex, flush=True)
