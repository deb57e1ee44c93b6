#!/usr/bin/env python3
"""On-box check + timing of the pj_rows.hip specialisation against the table-driven kernel."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pyjac_amd
from pyjac_amd import synth
from conftest import jac_scaled_err
mech = sys.argv[1]; n = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ev = pyjac_amd.Evaluator(mech, specialize='auto')
assert ev.has_spec, 'no specialisation attached'
pres, y = synth.dist_b(n, ev.nsp)
d_p = torch.from_numpy(pres).cuda(); d_y = torch.from_numpy(y).cuda()
S = pyjac_amd.LAYOUT_SOA
nchk = min(n, 4096)
a = torch.full((ev.nsp**2 * nchk,), float('nan'), dtype=torch.float64, device='cuda')
ev.use_spec(True); ev.time_jacobian(d_p[:nchk].contiguous(), d_y[:, :nchk].contiguous(), a, 1, S, S)
b = torch.empty_like(a)
ev.use_spec(False); ev.time_jacobian(d_p[:nchk].contiguous(), d_y[:, :nchk].contiguous(), b, 1, S, S)
A = a.cpu().numpy().reshape(ev.nsp**2, nchk).T; B = b.cpu().numpy().reshape(ev.nsp**2, nchk).T
print('nan', int(np.isnan(A).sum()), 'rows-vs-table scaled err', jac_scaled_err(A, B, ev.nsp), flush=True)
out = torch.empty(ev.nsp**2 * n, dtype=torch.float64, device='cuda')
for use in (True, False):
    ev.use_spec(use)
    ev.time_jacobian(d_p, d_y, out, 1, S, S)
    ms = ev.time_jacobian(d_p, d_y, out, iters, S, S)
    gbs = n * ev.jacobian_bytes_per_state / ms / 1e6
    print(json.dumps(dict(spec=use, n=n, ms=round(ms, 3), jac_per_s=round(n / ms * 1e3), GBps=round(gbs, 1), frac=round(gbs / 8000, 4))), flush=True)
