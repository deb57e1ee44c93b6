#!/usr/bin/env python3
"""Build (CPU container) or time (GPU box) variants of the pj_rows.hip library.
  build: rows_variants.py build <mech> <name> [ENV=VAL ...]   -> gpurun_variants/<name>.so
  time:  rows_variants.py time <mech> <n> <iters> <name> [<name> ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pyjac_amd
from pyjac_amd import _lib
VDIR = os.path.join(ROOT, 'pyjac_amd', 'spec', 'variants')
if sys.argv[1] == 'build':
    mech, name = sys.argv[2], sys.argv[3]
    for kv in sys.argv[4:]:
        k, v = kv.split('=', 1); os.environ[k] = v
    os.makedirs(VDIR, exist_ok=True)
    ev = pyjac_amd.Evaluator(mech, specialize='off')
    ev._build_rows(os.path.join(VDIR, name + '.so'))
else:
    import numpy as np, torch
    from pyjac_amd import synth
    from conftest import jac_scaled_err
    mech, n, iters = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    ev = pyjac_amd.Evaluator(mech, specialize='off')
    pres, y = synth.dist_b(n, ev.nsp)
    d_p = torch.from_numpy(pres).cuda(); d_y = torch.from_numpy(y).cuda()
    S = pyjac_amd.LAYOUT_SOA
    nchk = min(n, 4096)
    pc, yc = d_p[:nchk].contiguous(), d_y[:, :nchk].contiguous()
    b = torch.empty(ev.nsp**2 * nchk, dtype=torch.float64, device='cuda')
    ev.time_jacobian(pc, yc, b, 1, S, S)
    B = b.cpu().numpy().reshape(ev.nsp**2, nchk).T
    out = torch.empty(ev.nsp**2 * n, dtype=torch.float64, device='cuda')
    for name in sys.argv[5:]:
        _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, os.path.join(VDIR, name + '.so').encode()))
        a = torch.full_like(b, float('nan'))
        ev.time_jacobian(pc, yc, a, 1, S, S)
        A = a.cpu().numpy().reshape(ev.nsp**2, nchk).T
        err = jac_scaled_err(A, B, ev.nsp)
        ev.time_jacobian(d_p, d_y, out, 1, S, S)
        ms = ev.time_jacobian(d_p, d_y, out, iters, S, S)
        gbs = n * ev.jacobian_bytes_per_state / ms / 1e6
        print(json.dumps(dict(variant=name, err=float('%.2g' % err), n=n, ms=round(ms, 3), jac_per_s=round(n / ms * 1e3), frac=round(gbs / 8000, 4))), flush=True)
