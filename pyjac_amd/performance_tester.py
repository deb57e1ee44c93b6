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

* ``sweep``: the reference driver's loop over batch sizes 1, 2, 4, ... N for its GPU arm
  (performance_tester.py:341-347 ``steplist``, :497-508), ``repeats`` runs each, analytical or
  finite-difference Jacobian (``finite_diffs``, :280-296), appended as ``"N,milliseconds"`` lines to
  ``cuda_nco_nosmem_{ajac|fd}_-1_output.txt`` -- the file the reference's plotting scripts read (:388-397).
  The reference's CPU arm sweeps OpenMP thread counts 1, 2, 4 ... ncpu at the full batch (:276-283); its
  counterpart here is bench.py's ``cpu_baseline`` (pyJac's generated C at 1, 2, 4 ... and all usable cores:
  ``cpu_baseline.thread_sweep``).

    python -m pyjac_amd.performance_tester --mech mech.inp --data data.bin --num 1000000
    python -m pyjac_amd.performance_tester --mech mech.inp --data data.bin --num 1000000 --sweep [--fd]
"""
from __future__ import annotations

import argparse
import os
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


def step_list(num_conditions: int):
    """Batch sizes of the reference's GPU sweep (performance_tester.py:341-347): powers of two below
    num_conditions, then num_conditions itself."""
    steps, step = [], 1
    while step < num_conditions:
        steps.append(step)
        step *= 2
    if step / 2 != num_conditions:
        steps.append(num_conditions)
    return steps


def output_name(fd: bool) -> str:
    """File name of the reference driver for lang=cuda, no cache optimisation, no shared memory, no thread
    count (performance_tester.py:388-397)."""
    return 'cuda_nco_nosmem_%s_-1_output.txt' % ('fd' if fd else 'ajac')


def sweep(ev, pres: np.ndarray, y: np.ndarray, repeats: int = 10, fd: bool = False, out_dir: str = None,
          steps=None, quiet: bool = False):
    """The reference driver's batch-size sweep on the HIP path: for N in 1, 2, 4, ... num, `repeats` runs of
    the timed region of tester.cu.in:109-156 (H2D, every kernel of pj_run -- or the finite-difference arm,
    fd_jacob.cu:23-96 --, D2H), one ``"N,milliseconds"`` line per run.  Returns [(N, [ms, ...]), ...]; with
    out_dir the lines are appended to the reference's output file name."""
    import torch
    num, nsp = pres.size, ev.nsp
    steps = list(steps) if steps is not None else step_list(num)
    padded = ev.init(num)
    cap = min(num, padded)
    jac = np.empty(nsp * nsp * cap)
    d = lambda r: np.zeros(max(r, 1) * cap)
    bufs = (d(nsp), d(ev.n_fwd), d(ev.n_rev), d(ev.n_pres_mod), d(nsp), d(nsp))
    fh = open(os.path.join(out_dir, output_name(fd)), 'a+') if out_dir else None
    res = []
    try:
        for n in steps:
            times = []
            for _ in range(repeats):
                t0 = time.perf_counter()
                done = 0
                while done < n:
                    nc = min(n - done, padded)
                    if fd:
                        # the reference's FD arm evaluates dydt NSP + 1 times per state on the device and copies
                        # the Jacobians back (fd_jacob.cu:23-96 inside tester.cu.in's loop)
                        d_p = torch.from_numpy(np.ascontiguousarray(pres[done:done + nc])).cuda()
                        d_y = torch.from_numpy(np.ascontiguousarray(y[:, done:done + nc])).cuda()
                        out = ev.fd_jacobian(d_p, d_y)
                        jac[:nsp * nsp * nc] = out.reshape(-1).cpu().numpy()
                    else:
                        yc = np.ascontiguousarray(y[:, done:done + nc]).ravel()
                        ev.run(nc, padded, np.ascontiguousarray(pres[done:done + nc]), yc, *bufs, jac)
                    done += nc
                ms = (time.perf_counter() - t0) * 1e3
                times.append(ms)
                line = '%d,%.15e' % (n, ms)
                if fh:
                    fh.write(line + '\n')
                if not quiet:
                    print(line)
            res.append((n, times))
    finally:
        ev.cleanup()
        if fh:
            fh.close()
    return res


def main():
    import pyjac_amd
    ap = argparse.ArgumentParser()
    ap.add_argument('--mech', required=True)
    ap.add_argument('--data', required=True, help='data.bin (performance_tester.py:320-338 format)')
    ap.add_argument('--num', type=int, required=True)
    ap.add_argument('--repeats', type=int, default=3)
    ap.add_argument('--sweep', action='store_true',
                    help='batch sizes 1, 2, 4, ... num, `repeats` runs each (performance_tester.py:341-347, 497-508)')
    ap.add_argument('--fd', action='store_true', help='the finite-difference arm (fd_jacob.cu) instead of the analytical Jacobian')
    ap.add_argument('--out-dir', default=None, help='append the lines to the reference\'s output file name there')
    a = ap.parse_args()
    ev = pyjac_amd.Evaluator(a.mech, specialize='build')
    fmap = ev.mechanism.fwd_spec_map if ev.mechanism is not None else None
    pres, y = read_initial_conditions(a.data, a.num, ev.nsp, fmap)
    if a.sweep:
        sweep(ev, pres, y, a.repeats, fd=a.fd, out_dir=a.out_dir)
        return
    r = speedtest(ev, pres, y, a.repeats)
    print('kernel only: %d,%.15e' % (a.num, r['kernel_ms']))


if __name__ == '__main__':
    main()
