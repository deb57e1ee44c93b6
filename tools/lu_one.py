#!/usr/bin/env python3
"""Exactly <steps> batched LU factorisations of <n> random NSP x NSP blocks (for rocprofv3 PMC passes):
lu_one.py <nsp> <n> <steps> [soa]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pyjac_amd import linsolve
nsp, n, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
soa = len(sys.argv) > 4 and sys.argv[4] == 'soa'
g = torch.Generator(device='cuda').manual_seed(1)
a = torch.randn((n, nsp * nsp), dtype=torch.float64, device='cuda', generator=g)
a[:, ::nsp + 1] += 10.0 * nsp
if soa:
    a = a.T.contiguous()
for _ in range(steps):
    lu, perm = linsolve.lu_factor(a, layout=0 if soa else 1)
torch.cuda.synchronize()
print(nsp, n, steps, bool(torch.isfinite(lu[::997]).all()))
