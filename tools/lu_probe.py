#!/usr/bin/env python3
"""Time the batched LU / Newton-solve consumer (csrc/pj_lu.h) on random diagonally dominant blocks:
lu_probe.py [nsp:n ...]   -> ms per call, blocks/s, achieved GB/s on the algorithmic bytes
(factor: read + write 8 NSP^2 per block; fused solve: read 8 NSP^2 + 16 NSP)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pyjac_amd import linsolve

cases = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]] or [(10, 1000000), (24, 1000000), (53, 1000000), (64, 500000), (111, 200000)]
for nsp, n in cases:
    g = torch.Generator(device='cuda').manual_seed(1)
    a = torch.randn((n, nsp * nsp), dtype=torch.float64, device='cuda', generator=g)
    a[:, ::nsp + 1] += 10.0 * nsp
    b = torch.randn((n, nsp), dtype=torch.float64, device='cuda', generator=g)
    x = torch.empty_like(b)
    lu = torch.empty_like(a)

    def timed(f, reps=5):
        f(); torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            f()
        t1.record(); torch.cuda.synchronize()
        return t0.elapsed_time(t1) / reps
    import ctypes
    from pyjac_amd import _lib
    L = _lib.lib()
    perm = torch.empty((n, nsp), dtype=torch.int32, device='cuda')
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms_f = timed(lambda: _lib.check(L.pj_lu_factor_dev(nsp, n, a.data_ptr(), 1, 0.0, lu.data_ptr(), perm.data_ptr(), st())))
    ms_n = timed(lambda: linsolve.newton_solve(a, b, gamma=1e-3, out=x))
    ms_s = timed(lambda: linsolve.lu_solve(lu, perm, b, out=x))
    # the fused solve straight from the state-fastest batch layout (a: (NSP*NSP, n), b, x: (NSP, n))
    a_s, b_s = a.T.contiguous(), b.T.contiguous()
    x_s = torch.empty_like(b_s)
    ms_soa = timed(lambda: linsolve.newton_solve(a_s, b_s, gamma=1e-3, out=x_s, layout=0))
    del a_s, b_s, x_s
    bf, bn = 16 * nsp * nsp + 4 * nsp, 8 * nsp * nsp + 16 * nsp
    print('nsp %3d n %8d | factor %8.3f ms %.3g blocks/s %6.0f GB/s (%.3f of 8 TB/s) | fused Newton solve %8.3f ms %6.0f GB/s (%.3f) | solve from factors %8.3f ms | fused solve from the batch (SoA) layout %8.3f ms'
          % (nsp, n, ms_f, n / ms_f * 1e3, n * bf / ms_f / 1e6, n * bf / ms_f / 1e6 / 8000, ms_n, n * bn / ms_n / 1e6,
             n * bn / ms_n / 1e6 / 8000, ms_s, ms_soa), flush=True)
    del a, b, x, lu, perm
    torch.cuda.empty_cache()
