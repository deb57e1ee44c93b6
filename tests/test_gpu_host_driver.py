"""GPU: the host batch driver (pj_init / pj_run / pj_cleanup = pyjacob.cu:84-188's init / run / cleanup, and the per-state
pyjacob calls that go through the same code) works on a stream of its own and waits for THAT stream only.  Round 5 ended
pj_run with a device-wide synchronisation: every other stream of the process stalled with it."""
import time

import numpy as np
import pytest

from conftest import MECHS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['h2o2_n2', 'gri30_shaped'])
def test_pj_run_does_not_wait_for_other_streams(name):
    import torch
    import pyjac_amd
    from pyjac_amd import synth
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    ev = pyjac_amd.Evaluator(MECHS[name])
    n = 2048
    pres, y = synth.dist_b(n, ev.nsp, seed=3, Tlo=800, Thi=2400)
    y = np.ascontiguousarray(y)
    outs = dict(conc=np.zeros((ev.nsp, n)), fwd=np.zeros((ev.n_fwd, n)), rev=np.zeros((max(ev.n_rev, 1), n)),
                pres_mod=np.zeros((max(ev.n_pres_mod, 1), n)), spec_rates=np.zeros((ev.nsp, n)), dy=np.zeros((ev.nsp, n)),
                jac=np.zeros((ev.nsp * ev.nsp, n)))
    padded = ev.init(n)
    assert padded >= n

    def run():
        t0 = time.perf_counter()
        ev.run(n, padded, pres, y, outs['conc'], outs['fwd'], outs['rev'], outs['pres_mod'], outs['spec_rates'], outs['dy'], outs['jac'])
        return time.perf_counter() - t0
    run()
    quiet = min(run() for _ in range(3))
    ref = {k: v.copy() for k, v in outs.items()}
    # a second stream kept busy for much longer than a pj_run call takes
    side = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):       # (what a spin of 2e7 "cycles" lasts on this device)
        e0.record()
        torch.cuda._sleep(20_000_000)
        e1.record()
    side.synchronize()
    unit_s = max(e0.elapsed_time(e1) * 1e-3, 1e-4)
    busy_s = max(0.4, 20 * quiet)
    with torch.cuda.stream(side):
        for _ in range(int(busy_s / unit_s) + 1):
            torch.cuda._sleep(20_000_000)
    t = run()
    still_busy = not side.query()
    side.synchronize()
    print('%s: pj_run alone %.1f ms, next to a busy stream %.1f ms (that stream busy for >= %.0f ms, still busy afterwards: %s)'
          % (name, quiet * 1e3, t * 1e3, busy_s * 1e3, still_busy))
    assert still_busy, 'the side stream finished first: the test did not test anything'
    assert t < 0.5 * busy_s and t < 10 * quiet + 0.05, 'pj_run waited for a stream it has nothing to do with'
    for k in outs:
        assert np.array_equal(outs[k], ref[k]), k
    ev.cleanup()
