"""Performance-test harness on the HIP path (reference:
pyjac/performance_tester/performance_tester.py:213-508, tester.cu.in:1-168,
read_initial_conditions.cu:9-59).

* ``data.bin`` format: records of NSP+3 doubles ``[t, T, P, Y_0..Y_{NSP-1}]``
  (performance_tester.py:320-338), written from PaSR ``.npy`` arrays.
* ``read_initial_conditions``: the CUDA flavour's SoA host layout
  ``y_host[i + (j+1)*NUM]`` with ``apply_mask`` (last species moved to the end).
* ``speedtest``: the timed region of tester.cu.in:109-156 -- H2D of the states,
  the Jacobian kernel, D2H of the Jacobians, chunked by the ``padded`` capacity --
  printed as the reference's ``"N,milliseconds"`` line; additionally the
  kernel-only time (HIP events), which the reference cannot separate.

    python -m pyjac_amd.performance_tester --mech mech.inp --data data.bin --num 1000000
"""
from __future__ import annotations

import argparse
import time

import numpy as np


def write_data_bin(path: str, arrays) -> int:
    """Tile PaSR arrays (steps, particles, 3+NSP) into data.bin; returns #records."""
    data = None
    for a in arrays:
        a = np.asarray(a, dtype=np.float64)
        a = a.reshape(a.shape[0] * a.shape[1], a.shape[2]) if a.ndim == 3 else a
        data = a if data is None else np.vstack((data, a))
    data.tofile(path)
    return data.shape[0]


def read_initial_conditions(path: str, num: int, nsp: int, fwd_spec_map=None):
    """Returns (pres[num], y SoA (NSP, num) = [T; Y_0..Y_{NSP-2}] in internal order)."""
    rec = nsp + 3
    buf = np.fromfile(path, dtype=np.float64, count=num * rec)
    if buf.size != num * rec:
        raise ValueError('File (%s) is incorrectly formatted, %d doubles were expected but only %d '
                         'were read.' % (path, num * rec, buf.size))
    buf = buf.reshape(num, rec)
    Y = buf[:, 3:3 + nsp]
    if fwd_spec_map is not None:          # apply_mask, mech_auxiliary.py:188-206
        Y = Y[:, fwd_spec_map]
    y = np.empty((nsp, num))
    y[0] = buf[:, 1]
    y[1:] = Y[:, :-1].T
    return np.ascontiguousarray(buf[:, 2]), y


def speedtest(ev, pres: np.ndarray, y: np.ndarray, repeats: int = 1, quiet: bool = False):
    """ev: pyjac_amd.Evaluator.  Returns dict(end_to_end_ms, kernel_ms)."""
    import torch
    num = pres.size
    nsp = ev.nsp
    padded = ev.init(num)
    conc = np.zeros(nsp * min(num, padded))
    # --- end to end, as tester.cu.in:109-156 (H2D + kernel + D2H, chunked) ---
    jac = np.empty(nsp * nsp * min(num, padded))
    d = lambda r: np.zeros(max(r, 1) * min(num, padded))
    bufs = (d(nsp), d(ev.n_fwd), d(ev.n_rev), d(ev.n_pres_mod), d(nsp), d(nsp))
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        done = 0
        while done < num:
            nc = min(num - done, padded)
            yc = np.ascontiguousarray(y[:, done:done + nc]).ravel()
            ev.run(nc, padded, np.ascontiguousarray(pres[done:done + nc]), yc, *bufs, jac)
            done += nc
        ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None else min(best, ms)
    ev.cleanup()
    if not quiet:
        print('%d,%.15e' % (num, best))
    # --- kernel only ---
    nk = min(num, padded)
    d_p = torch.from_numpy(np.ascontiguousarray(pres[:nk])).cuda()
    d_y = torch.from_numpy(np.ascontiguousarray(y[:, :nk])).cuda()
    out = torch.empty(nsp * nsp * nk, dtype=torch.float64, device='cuda')
    ev.time_jacobian(d_p, d_y, out, 2)
    kms = ev.time_jacobian(d_p, d_y, out, 10) * (num / nk)
    return dict(end_to_end_ms=best, kernel_ms=kms, num=num, padded=padded)


def main():
    import pyjac_amd
    ap = argparse.ArgumentParser()
    ap.add_argument('--mech', required=True)
    ap.add_argument('--data', required=True, help='data.bin (performance_tester.py:320-338 format)')
    ap.add_argument('--num', type=int, required=True)
    ap.add_argument('--repeats', type=int, default=3)
    a = ap.parse_args()
    ev = pyjac_amd.Evaluator(a.mech, specialize='build')
    fmap = ev.mechanism.fwd_spec_map if ev.mechanism is not None else None
    pres, y = read_initial_conditions(a.data, a.num, ev.nsp, fmap)
    r = speedtest(ev, pres, y, a.repeats)
    print('kernel only: %d,%.15e' % (a.num, r['kernel_ms']))


if __name__ == '__main__':
    main()
