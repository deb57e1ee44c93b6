"""Planner-geometry sweep on the GPU.

pyJac handles any mechanism size by emitting more files (pyjac/core/create_jacobian.py:2213-2223, CParams.py:19-22); the
row-block kernels change GEOMETRY with the species count instead (pyjac_amd/specbuild.rblk_geometry):

    <= 53 species (with factor rows)  64 states x four lane groups, factor columns, ONE row kernel
    54 .. 56                          256 states, one lane group, several row kernels
    57 .. 120                         128 states x two lane groups, several row kernels
    > 120                             64 states x four lane groups, cooperative prologue, several row kernels (one on request)

The shipped mechanisms sit at 24, 53, 72 and 111 species.  tests/golden/make_sweep_mechs.py puts a small mechanism on each
side of every threshold (17, 54, 56, 57, 64, 65, 120, 121, 140 species) and adds five seeded random ones (20-60 species);
__graft_entry__.build() prebuilds their libraries.  Every one is run here through the C ABI against the CPU oracle: Jacobians
(SoA and per-state layout, batches that end inside a wavefront / a workgroup), the six rate arrays, w = J v (k_jvd) and the
finite-difference arm; two of them also against golden vectors of pyJac's own generated C (tests/golden/make_golden.py).

Tolerances: entry-wise rtol 1e-6 against the binary128 evaluation of the reference's formulas (BASELINE.json's rtol) on a
subset of states, and the scaled metric of conftest.jac_scaled_err (1e-6 |J| + 1e-12 of the row / column scale) + a relative
Frobenius error < 1e-9 per state against the oracle on all of them."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, jac_scaled_err, mixed_err, rate_scales, rel_err_entries, thresholded_rel_err

pytestmark = pytest.mark.gpu

SWEEP = {os.path.basename(f)[:-4]: f for f in sorted(glob.glob(os.path.join(GOLDEN, 'sweep', 'sweep_*.inp')))}
GEOMETRY = [k for k in SWEEP if k.startswith('sweep_n')]
RANDOM = [k for k in SWEEP if k.startswith('sweep_r')]
# (name, build-time environment, options) of every prebuilt library: __graft_entry__.spec_build_list
VARIANTS = [(k, {}, {}) for k in SWEEP] + [('sweep_n121', {'PJ_RBLK_WIDE_SINGLE_RXN': '1000'}, {})]
VIDS = [v[0] + ('-onekernel' if v[1] else '') for v in VARIANTS]
RTOL = 1e-6


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    torch.cuda.set_device(0)
    return torch


_tab = {}


def _tables(name):
    from pyjac_amd.mechanism import read_mech
    from pyjac_amd.tables import build_tables
    if name not in _tab:
        _tab[name] = build_tables(read_mech(SWEEP[name]))
    return _tab[name]


def _ev(name, env, opts, monkeypatch):
    """The prebuilt row-block library of a sweep mechanism, attached -- never compiled here: a missing library is a failure
    of __graft_entry__.build(), not something to paper over with ten minutes of hipcc on the GPU box."""
    import pyjac_amd
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ev = pyjac_amd.Evaluator(SWEEP[name], specialize='off')
    so = ev.spec_path('rblk', **opts)
    assert so and os.path.exists(so), 'library of %s %s missing (%s): run __graft_entry__.build()' % (name, env, so)
    assert ev.specialize(build=False, kind='rblk', **opts) and ev.spec_kernel == 'pj_rblk'
    ev.use_spec(2)
    return ev


@pytest.mark.parametrize('variant', VARIANTS, ids=VIDS)
@pytest.mark.parametrize('layout', ['soa', 'aos'])
def test_sweep_jacobians_vs_oracle(variant, layout, torch_cuda, monkeypatch):
    import pyjac_amd
    from oracle.oracle import Oracle, OracleQuad
    from pyjac_amd import synth
    torch = torch_cuda
    name, env, opts = variant
    ev = _ev(name, env, opts, monkeypatch)
    orc = Oracle(_tables(name))
    for n in (63, 257, 4099):
        pres, y = synth.dist_b(n, ev.nsp, seed=600 + n, Tlo=500, Thi=2600)
        d_p = torch.from_numpy(pres).cuda()
        L = pyjac_amd.LAYOUT_SOA if layout == 'soa' else pyjac_amd.LAYOUT_AOS
        d_y = torch.from_numpy(y if layout == 'soa' else np.ascontiguousarray(y.T)).cuda()
        out = torch.full((ev.nsp ** 2, n) if layout == 'soa' else (n, ev.nsp ** 2), float('nan'), dtype=torch.float64, device='cuda')
        jac = ev.jacobian(d_p, d_y, y_layout=L, out=out, jac_layout=L).cpu().numpy()
        if layout == 'soa':
            jac = jac.T
        assert np.isfinite(jac).all(), (name, layout, n, 'an entry was not written')
        ref = orc.batch_jacob(pres, np.ascontiguousarray(y.T))
        sc = jac_scaled_err(jac, ref, ev.nsp)
        mx, fro = thresholded_rel_err(jac, ref)
        print('%s %s n=%d: scaled %.3g, fro %.3g, thresholded max rel vs oracle %.3g' % (VIDS[VARIANTS.index(variant)], layout, n, sc, fro, mx))
        assert sc <= 1.0 and fro < 1e-9, (name, layout, n, sc, fro)
        if n == 63:
            # north star: every entry within rtol 1e-6 of the reference's formulas evaluated in binary128
            truth = OracleQuad(_tables(name)).batch_jacob(pres[:32], np.ascontiguousarray(y.T[:32]))
            r = float(rel_err_entries(jac[:32], truth).max())
            assert r < RTOL, (name, layout, r)
    ev.close()


@pytest.mark.parametrize('variant', VARIANTS, ids=VIDS)
def test_sweep_rate_outputs_vs_oracle(variant, torch_cuda, monkeypatch):
    from oracle.oracle import Oracle
    from pyjac_amd import synth
    torch = torch_cuda
    name, env, opts = variant
    ev = _ev(name, env, opts, monkeypatch)
    n = 257
    pres, y = synth.dist_b(n, ev.nsp, seed=14, Tlo=600, Thi=2600)
    y_aos = np.ascontiguousarray(y.T)
    orc = Oracle(_tables(name))
    o = [orc.eval_all(float(pres[s]), y_aos[s]) for s in range(n)]
    g = {k: np.array([x[k] for x in o]) for k in ('conc', 'fwd', 'rev', 'pres_mod', 'spec_rates', 'dydt')}
    gross, sdy = rate_scales(_tables(name), pres, y_aos, g['conc'], g['fwd'], g['rev'], g['pres_mod'])
    r = {k: v.cpu().numpy().T for k, v in ev.rates(torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()).items()}
    for k, rows in (('conc', ev.nsp), ('fwd', ev.n_fwd), ('rev', ev.n_rev), ('pres_mod', ev.n_pres_mod)):
        if rows:
            mx, _ = thresholded_rel_err(r[k][:, :rows], g[k][:, :rows])
            assert mx < 1e-9, (name, k, mx)
    assert mixed_err(r['spec_rates'], g['spec_rates'], gross[:, None]) <= 1.0, name
    assert mixed_err(r['dydt'], g['dydt'], sdy) <= 1.0, name
    ev.close()


@pytest.mark.parametrize('variant', VARIANTS, ids=VIDS)
def test_sweep_jacobian_vector_product(variant, torch_cuda, monkeypatch):
    """w = J v (k_jvd: every reaction once) with and without the J_nplusone quirk, a batch that ends inside a workgroup."""
    from oracle.oracle import Oracle
    from pyjac_amd import synth
    torch = torch_cuda
    name, env, opts = variant
    ev = _ev(name, env, opts, monkeypatch)
    n = 333
    pres, y = synth.dist_b(n, ev.nsp, seed=12, Tlo=700, Thi=2500)
    v = np.random.default_rng(3).standard_normal((ev.nsp, n))
    v[0] *= 100.0
    d_p, d_y, d_v = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(v).cuda()
    orc = Oracle(_tables(name))
    for on in (False, True):
        ev.set_sum_last_species(on)
        orc.lib.pjo_set_sum_last_species(int(on))
        try:
            J = orc.batch_jacob(pres, np.ascontiguousarray(y.T)).reshape(n, ev.nsp, ev.nsp)      # [s][col][row]
        finally:
            orc.lib.pjo_set_sum_last_species(0)
        w = ev.jacobian_vec(d_p, d_y, d_v).cpu().numpy().T
        ref = np.einsum('scr,cs->sr', J, v)
        scale = np.einsum('scr,cs->sr', np.abs(J), np.abs(v)) + 1e-300
        assert np.isfinite(w).all() and (np.abs(w - ref) / scale).max() < 1e-9, (name, on)
    ev.set_sum_last_species(False)
    ev.close()


@pytest.mark.parametrize('name', GEOMETRY + RANDOM[:2])
def test_sweep_finite_difference_arm(name, torch_cuda, monkeypatch):
    """The reference's FD arm (performance_tester/fd_jacob.c) on the rate kernels of every geometry: NSP + 1 k_rate passes."""
    from oracle.oracle import Oracle
    from pyjac_amd import synth
    torch = torch_cuda
    ev = _ev(name, {}, {}, monkeypatch)
    n = 24
    pres, y = synth.dist_b(n, ev.nsp, seed=5, Tlo=900, Thi=2200)
    fd = ev.fd_jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()).cpu().numpy().T.reshape(n, ev.nsp, ev.nsp)
    o = Oracle(_tables(name))
    ref = np.array([o.fd_jacob(float(pres[s]), y[:, s].copy()) for s in range(n)]).reshape(n, ev.nsp, ev.nsp)
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(fd), ok), (name, int((~np.isfinite(fd)).sum()), int((~ok).sum()))
    okcol = ok.all(axis=2, keepdims=True)
    colscale = np.abs(np.where(ok, ref, 0.0)).max(axis=2, keepdims=True) + 1e-300
    err = (np.abs(np.where(okcol, fd - ref, 0.0)) / colscale).max()
    print('%s FD arm vs oracle FD (column-scaled): %.3g' % (name, err))
    assert err < 1e-4
    ev.close()


@pytest.mark.parametrize('name', ['sweep_n054', 'sweep_n121'])
def test_sweep_vs_reference_golden(name, torch_cuda, monkeypatch):
    """Two of the sweep geometries (256 states / one lane group; 64 states / four lane groups beyond 120 species) against
    vectors of pyJac's OWN generated C (tests/golden/make_golden.py -> <name>_golden.npz)."""
    torch = torch_cuda
    g = np.load(os.path.join(GOLDEN, name + '_golden.npz'))
    ev = _ev(name, {}, {}, monkeypatch)
    assert ev.nsp == int(g['nsp']) and ev.n_fwd == int(g['n_fwd']) and ev.n_rev == int(g['n_rev']) and ev.n_pres_mod == int(g['n_pres_mod'])
    d_p = torch.from_numpy(g['pres'].copy()).cuda()
    d_y = torch.from_numpy(np.ascontiguousarray(g['y'].T)).cuda()
    jac = ev.jacobian(d_p, d_y).cpu().numpy().T
    sc = jac_scaled_err(jac, g['jac'], ev.nsp)
    mx, fro = thresholded_rel_err(jac, g['jac'])
    print('%s vs pyJac golden: scaled %.3g, fro %.3g, thresholded max rel %.3g' % (name, sc, fro, mx))
    assert sc <= 1.0 and fro < 1e-9
    r = {k: v.cpu().numpy().T for k, v in ev.rates(d_p, d_y).items()}
    gross, sdy = rate_scales(_tables(name), g['pres'], g['y'], g['conc'], g['fwd'], g['rev'], g['pres_mod'])
    for k, rows in (('conc', ev.nsp), ('fwd', ev.n_fwd), ('rev', ev.n_rev), ('pres_mod', ev.n_pres_mod)):
        mx, _ = thresholded_rel_err(r[k][:, :rows], g[k][:, :rows])
        assert mx < RTOL, (name, k, mx)
    assert mixed_err(r['spec_rates'], g['spec_rates'], gross[:, None]) <= 1.0
    assert mixed_err(r['dydt'], g['dydt'], sdy) <= 1.0
    ev.close()


@pytest.mark.parametrize('layout', ['soa', 'aos'])
def test_isomer_of_the_last_species_on_the_register_resident_kernel(layout, torch_cuda):
    """iso_n012 (CH2 next to CH2(S), which is last: W_j / W_N = 1, the dense parts of that column cancel exactly) through
    pj_lane and through the table-driven kernels: entry-wise rtol 1e-6 against the binary128 evaluation (tests/
    test_isomer_columns.py holds the same on the CPU emulation of all four kernel families; sweep_r2 above is the row-block case)."""
    import pyjac_amd
    from oracle.oracle import OracleQuad
    from pyjac_amd import synth
    from pyjac_amd.mechanism import read_mech
    from pyjac_amd.tables import build_tables
    torch = torch_cuda
    mech = os.path.join(GOLDEN, 'sweep', 'iso_n012.inp')
    ev = pyjac_amd.Evaluator(mech)
    assert ev.has_spec and ev.spec_kernel == 'pj_lane', 'pj_lane library of iso_n012 missing: run __graft_entry__.build()'
    n = 4099
    pres, y = synth.dist_b(n, ev.nsp, seed=663, Tlo=500, Thi=2600)
    truth = OracleQuad(build_tables(read_mech(mech))).batch_jacob(pres[:256], np.ascontiguousarray(y.T[:256]))
    L = pyjac_amd.LAYOUT_SOA if layout == 'soa' else pyjac_amd.LAYOUT_AOS
    d_p = torch.from_numpy(pres).cuda()
    d_y = torch.from_numpy(y if layout == 'soa' else np.ascontiguousarray(y.T)).cuda()
    for use in (2, 0):
        ev.use_spec(use)
        jac = ev.jacobian(d_p, d_y, y_layout=L, jac_layout=L).cpu().numpy()
        jac = jac.T if layout == 'soa' else jac
        r = float(rel_err_entries(jac[:256], truth).max())
        print('iso_n012 %s %s: max entry-wise relative error vs binary128 %.3g' % (layout, 'pj_lane' if use else 'table-driven', r))
        assert np.isfinite(jac).all() and r < RTOL, (layout, use, r)
