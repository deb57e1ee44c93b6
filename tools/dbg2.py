import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import pyjac_amd
from conftest import MECHS, thresholded_rel_err
from pyjac_amd import synth
from pyjac_amd.mechanism import read_mech
from pyjac_amd.tables import build_tables
from oracle.oracle import Oracle
for name in ('h2o2_n2','h2o2','synth_alltypes'):
    ev = pyjac_amd.Evaluator(MECHS[name]); assert ev.has_spec
    n=20000
    if name=='h2o2_n2': pres,y = synth.dist_a(n, ev.nsp)
    else:
        pres,y = synth.dist_b(n, ev.nsp, seed=31, Tlo=400, Thi=2800); pres = 101325*10**np.random.default_rng(8).uniform(-1.5,1.5,n)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
    spec = ev.jacobian(d_p,d_y).cpu().numpy().T
    ev.use_spec(False); gen = ev.jacobian(d_p,d_y).cpu().numpy().T
    ref = Oracle(ev.tables).batch_jacob(pres, np.ascontiguousarray(y.T), 16)
    print(name, 'spec-vs-oracle', thresholded_rel_err(spec,ref), 'generic-vs-oracle', thresholded_rel_err(gen,ref), flush=True)
