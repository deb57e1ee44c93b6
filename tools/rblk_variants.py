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
            from pyjac_amd import specbuild, _lib
            from pyjac_amd.kcfactors import kc_factor_rows
            specbuild.build_rblk(_lib.lib(), ev._h, ev.nsp, so, defines=defines, kcf_rows=kc_factor_rows(ev.tables), nkc=int(ev.tables.I[10]), nrxn=ev.n_fwd, **opts)
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
        if tag in ('tab', 'keval'):     # the no-compile paths: table-driven state-per-lane kernel / cooperative kernel
            ev.set_generic_kernel('k_tab' if tag == 'tab' else 'k_eval')
            so = None
        elif tag == 'rblk':      # whatever prebuilt library of that family exists (any build digest)
            pat = os.path.basename(ev.spec_path(tag)).rsplit('_', 1)[0] + '_*.so'
            so = sorted(glob.glob(os.path.join(ROOT, 'pyjac_amd', 'spec', pat)))[-1]
        else:
            so = os.path.join(VDIR, '%s_%s.so' % (stem, tag))
        if so:
            _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, so.encode()))
        jac.fill_(float('nan'))
        ev.time_jacobian(d_p, d_y, jac, 2, L, L)
        ms = min(ev.time_jacobian(d_p, d_y, jac, 4, L, L) for _ in range(2))
        bj = ev.jacobian_bytes_per_state
        sample = jac[:, ::4999].cpu().numpy().T
        if ref is None:
            ref = sample
        line = '%-14s %8.3f ms  %.3g Jac/s  frac %.3f   vs first: %.2g  nan=%d' % (
            tag, ms, n / ms * 1e3, n * bj / ms / 1e6 / 8000, jac_scaled_err(sample, ref, ev0.nsp), int(np.isnan(sample).sum()))
        if os.environ.get('PJ_VAR_RATES', '1') != '0' and so:
            line += ' | rates' + rates_ms(ev, d_p, d_y, n, torch)
        print(line, flush=True)
        ev.close()


_bufs = {}


def rates_ms(ev, d_p, d_y, n, torch):
    """ms per launch of the rate pass (pj_eval_rates_dev): every array, dydt only."""
    import ctypes
    from pyjac_amd import _lib
    rows = dict(conc=ev.nsp, fwd=ev.n_fwd, rev=max(ev.n_rev, 1), pres_mod=max(ev.n_pres_mod, 1), spec_rates=ev.nsp, dy=ev.nsp)
    if not _bufs:
        _bufs.update({k: torch.empty((r, n), dtype=torch.float64, device='cuda') for k, r in rows.items()})
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = ''
    for label, which in (('all', tuple(rows)), ('dydt', ('dy',))):
        p = lambda k: _bufs[k].data_ptr() if k in which else None
        run = lambda: _lib.check(_lib.lib().pj_eval_rates_dev(ev._h, n, d_p.data_ptr(), d_y.data_ptr(), 0, p('conc'), p('fwd'),
                                                              p('rev'), p('pres_mod'), p('spec_rates'), p('dy'), stream))
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        by = 8 * (ev.nsp + 1) + 8 * sum(rows[k] for k in which)
        out += ' %s %.3f ms (%.3f)' % (label, ms, n * by / ms / 1e6 / 8000)
    return out


if __name__ == '__main__':
    main()
