"""CPU stand-in for pyjac_amd.Evaluator, injected into bench.py by tests/test_bench_gloo.py
(PJ_BENCH_EVALUATOR=stub_evaluator:make) so that bench.main()'s own multi-rank branch runs over gloo in the
GPU-less container.  TEST INFRASTRUCTURE: Jacobians come from the CPU oracle; never a measurement."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class StubEvaluator:
    has_spec = True
    spec_kernel = 'stub (CPU oracle)'

    def __init__(self, mech):
        from oracle.oracle import Oracle
        from pyjac_amd.mechanism import read_mech
        from pyjac_amd.tables import build_tables
        self.tables = build_tables(read_mech(mech))
        self._o = Oracle(self.tables)
        self.nsp, self.n_fwd = self.tables.nsp, self.tables.nrxn
        # (tests/test_bench_gloo.py: one rank without its library -- the ranks must agree to stop)
        if os.environ.get('STUB_MISSING_RANK') == os.environ.get('RANK', '0'):
            self.has_spec = False

    @property
    def jacobian_bytes_per_state(self):
        return 8 * (self.nsp + 1) + 8 * self.nsp * self.nsp

    def get_launch(self):
        return dict(tile_states=64, threads=256, lds_bytes=0)

    def jacobian(self, pres, y, y_layout=0, out=None, jac_layout=0):
        y_aos = y.numpy().T if y_layout == 0 else y.numpy()
        j = self._o.batch_jacob(pres.numpy(), np.ascontiguousarray(y_aos))          # (n, nsp * nsp)
        t = torch.from_numpy(np.ascontiguousarray(j.T) if jac_layout == 0 else j)
        if out is None:
            return t
        out.copy_(t.reshape(out.shape))
        return out

    def time_jacobian(self, pres, y, out, iters, y_layout=0, jac_layout=0):
        t0 = time.perf_counter()
        for _ in range(iters):
            self.jacobian(pres, y, y_layout, out, jac_layout)
        return (time.perf_counter() - t0) / iters * 1e3


def make(mech):
    return StubEvaluator(mech)
