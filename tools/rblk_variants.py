#!/usr/bin/env python3
"""Build (CPU box) or time (GPU box) variants of the pj_rblk.hip library of one mechanism.
  build: rblk_variants.py build <mech> tag:key=val,key=val,D=-DPJQ_X=1;-DPJQ_Y=2 ...
  time : rblk_variants.py time <mech> <n> [tags...]   (every variant found if no tags)"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pyjac_amd
VDIR = os.path.join(ROOT, 'pyjac_amd', 'spec', 'var')


def main():
    mode, mech = sys.argv[1], sys.argv[2]
    stem = os.path.splitext(os.path.basename(mech))[0]
    if mode == 'build':
        os.makedirs(VDIR, exist_ok=True)
        for spec in sys.argv[3:]:
            tag, _, rest = spec.partition(':')
            opts, defines = {}, ()
            for kv in filter(None, rest.split(',')):
                k, v = kv.split('=', 1)
                if k == 'D':
                    defines = tuple(v.split(';'))
                else:
                    opts[k] = int(v)
            ev = pyjac_amd.Evaluator(mech, specialize='off')
            so = os.path.join(VDIR, '%s_%s.so' % (stem, tag))
            t0 = time.time()
            ev._build_rblk(so, defines=defines, **opts)
            print('built %s in %.0f s' % (so, time.time() - t0), flush=True)
        return
    import numpy as np, torch
    from pyjac_amd import synth, _lib
    from conftest import jac_scaled_err
    n = int(sys.argv[3])
    tags = sys.argv[4:] or sorted(os.path.basename(p)[len(stem) + 1:-3] for p in glob.glob(os.path.join(VDIR, stem + '_*.so')))
    ev0 = pyjac_amd.Evaluator(mech, specialize='off')
    pres, y = synth.dist_b(n, ev0.nsp)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
    jac = torch.empty((ev0.nsp ** 2, n), dtype=torch.float64, device='cuda')
    L = pyjac_amd.LAYOUT_SOA
    ref = None
    for tag in tags:
        ev = pyjac_amd.Evaluator(mech, specialize='off')
        if tag in ('rows', 'rblk'):      # whatever prebuilt library of that family exists (any build digest)
            pat = os.path.basename(ev.spec_path(tag)).rsplit('_', 1)[0] + '_*.so'
            so = sorted(glob.glob(os.path.join(ROOT, 'pyjac_amd', 'spec', pat)))[-1]
        else:
            so = os.path.join(VDIR, '%s_%s.so' % (stem, tag))
        _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, so.encode()))
        jac.fill_(float('nan'))
        ev.time_jacobian(d_p, d_y, jac, 2, L, L)
        ms = min(ev.time_jacobian(d_p, d_y, jac, 4, L, L) for _ in range(2))
        bj = ev.jacobian_bytes_per_state
        sample = jac[:, ::4999].cpu().numpy().T
        if ref is None:
            ref = sample
        print('%-14s %8.3f ms  %.3g Jac/s  frac %.3f   vs first: %.2g  nan=%d' % (
            tag, ms, n / ms * 1e3, n * bj / ms / 1e6 / 8000, jac_scaled_err(sample, ref, ev0.nsp), int(np.isnan(sample).sum())), flush=True)
        ev.close()


if __name__ == '__main__':
    main()
