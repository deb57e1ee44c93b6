"""CPU: Chemkin front end, table builder, C-ABI surface, and the kernel phases
run through the thread-emulation harness (tests/emu) against the oracle."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import GOLDEN, jac_scaled_err, MECHS, ROOT, thresholded_rel_err
from oracle.oracle import Oracle
from pyjac_amd import synth
from pyjac_amd.mechanism import read_mech
from pyjac_amd.tables import MechTables, build_tables


def test_parser_h2o2_counts_and_last_species():
    m = read_mech(MECHS['h2o2'])
    assert (m.nsp, m.n_fwd, m.n_rev, m.n_pres_mod) == (9, 28, 28, 6)      # mechanism.h of the reference
    assert m.species_names()[-1] == 'AR'
    m = read_mech(MECHS['h2o2_n2'])
    assert m.nsp == 10 and m.species_names()[-1] == 'N2'
    assert m.fwd_spec_map == list(range(10))                             # N2 already last: identity maps


def test_parser_units_and_rev_split():
    m = read_mech(MECHS['synth_alltypes'])
    # explicit REV splits into two irreversible reactions (mech_interpret.py:693-713)
    assert m.n_fwd == 34 + 1
    r0 = m.reacs[0]                      # 2O+M<=>O2+M, A = 1.2e17 cm^6/mol^2/s -> /1000^2
    assert r0.thd_body and r0.A == pytest.approx(1.2e17 / 1000.0 ** 2)
    assert r0.E == 0.0
    r2 = m.reacs[2]                      # E in cal/mol -> activation temperature
    assert r2.E == pytest.approx(6260.0 * 4.184 / 8.3144621)


def test_last_species_moved_to_end(tmp_path):
    src = open(MECHS['h2o2_n2']).read().replace(
        'H2      H       O       O2      OH      H2O     HO2     H2O2     AR      N2',
        'H2      N2      H       O       O2      OH      H2O     HO2     H2O2     AR')
    p = tmp_path / 'perm.inp'
    p.write_text(src)
    m = read_mech(str(p))
    assert m.species_names() == ['H2', 'H', 'O', 'O2', 'OH', 'H2O', 'HO2', 'H2O2', 'AR', 'N2']
    assert m.fwd_spec_map == [0, 2, 3, 4, 5, 6, 7, 8, 9, 1]
    assert [m.back_spec_map[i] for i in m.fwd_spec_map] == list(range(10))


def test_tables_roundtrip(tmp_path, tables):
    t = tables('synth_alltypes')
    f = tmp_path / 'm.pjtab'
    t.save(str(f))
    u = MechTables.load(str(f))
    assert np.array_equal(t.I, u.I) and np.array_equal(t.D, u.D)


def test_cabi_exports_every_declared_symbol():
    from pyjac_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'pyjac_amd.h')).read()
    hdr = re.sub(r'#ifdef PJ_TIMING.*?#endif', '', hdr, flags=re.S)      # debug-build-only declarations
    declared = set(re.findall(r'\b(pj_[a-z_0-9]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name)
    assert b'gfx950' in L.pj_version()


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import pyjac_amd
    ev = pyjac_amd.Evaluator(MECHS['h2o2_n2'])
    assert (ev.nsp, ev.n_fwd, ev.n_rev, ev.n_pres_mod) == (10, 28, 28, 6)
    with pytest.raises(pyjac_amd.PyjacError):
        ev.init(64)
    from pyjac_amd import pyjacob
    pyjacob.use_mechanism(ev)
    with pytest.raises(pyjac_amd.PyjacError):
        pyjacob.py_dydt(0.0, 101325.0, np.ones(10), np.zeros(10))


def test_linsolve_argument_checks_and_no_cpu_fallback():
    """pyjac_amd.linsolve (batched LU consumer): shapes, dtypes and devices are refused before a raw pointer reaches
    the C ABI, and without a HIP device the entry points fail loudly (no host fallback)."""
    import ctypes as ct
    import torch
    from pyjac_amd import PyjacError, _lib, linsolve
    with pytest.raises(ValueError):
        linsolve.lu_factor(torch.zeros((4, 9), dtype=torch.float64))            # not on the device
    with pytest.raises(ValueError):
        linsolve._blocks(torch.zeros((4, 10), dtype=torch.float64))             # not a device tensor either
    L = _lib.lib()
    if not torch.cuda.is_available():
        buf = (ct.c_double * 16)()
        prm = (ct.c_int * 4)()
        rc = L.pj_lu_factor_dev(4, 1, ct.cast(buf, ct.c_void_p), 1, 0.0, ct.cast(buf, ct.c_void_p), ct.cast(prm, ct.c_void_p), None)
        assert rc == -2                                                          # PJ_ENODEV
        with pytest.raises(PyjacError):
            _lib.check(rc)
    assert L.pj_lu_factor_dev(141, 1, None, 1, 0.0, None, None, None) != 0          # bad arguments never reach a launch
    assert L.pj_lu_factor_dev(4, 0, None, 1, 0.0, None, None, None) == 0            # empty batch


def test_product_package_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'pyjac_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'(^|\n)\s*(from|import)\s+oracle', txt), f
                for needle in ('libpyjac_oracle', 'libpyjac_ref', 'pjo_', 'oracle.oracle', '_ref/'):
                    assert needle not in txt, (f, needle)


# ---- kernel phases through the emulation harness ----
_dp = ctypes.POINTER(ctypes.c_double)


def _emu():
    path = os.path.join(ROOT, 'tests', 'emu', '_build', 'libpj_emu.so')
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(path)


def _run_emu(tab, pres, y_soa, TS, NT, aos):
    n, nsp = pres.size, tab.nsp
    out = dict(jac=np.zeros(nsp * nsp * n), conc=np.zeros(nsp * n), fwd=np.zeros(tab.nrxn * n),
               rev=np.zeros(max(tab.nrev, 1) * n), pres_mod=np.zeros(max(tab.npres, 1) * n),
               spec_rates=np.zeros(nsp * n), dydt=np.zeros(nsp * n))
    I = np.ascontiguousarray(tab.I, dtype=np.int32)
    D = np.ascontiguousarray(tab.D)
    P = lambda a: a.ctypes.data_as(_dp)
    rc = _emu().emu_run(I.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_long(I.size), P(D),
                        ctypes.c_long(D.size), ctypes.c_long(n), P(pres), P(np.ascontiguousarray(y_soa)),
                        P(out['jac']), int(aos), P(out['conc']), P(out['fwd']), P(out['rev']),
                        P(out['pres_mod']), P(out['spec_rates']), P(out['dydt']), TS, NT, 0)
    assert rc == 0
    return out


@pytest.mark.parametrize('name,TS,NT,aos', [('h2o2_n2', 64, 256, False), ('h2o2', 16, 256, False),
                                            ('synth_alltypes', 4, 128, True),
                                            ('synth_alltypes', 64, 64, False),
                                            ('synth_alltypes', 1, 64, True),
                                            ('synth_srichb', 16, 256, False), ('synth_srichb', 1, 64, True),
                                            ('synth_fracnu', 16, 256, False), ('synth_fracnu', 1, 64, True)])
def test_kernel_phases_match_oracle(name, TS, NT, aos, tables):
    tab = tables(name)
    o = Oracle(tab)
    n = 97          # ragged: not a multiple of any tile size
    pres, y = synth.dist_b(n, tab.nsp, seed=5, Tlo=400, Thi=2800)
    pres = 101325 * 10 ** np.random.default_rng(2).uniform(-1.5, 1.5, n)
    out = _run_emu(tab, pres, y, TS, NT, aos)
    ref_j = o.batch_jacob(pres, np.ascontiguousarray(y.T))
    got_j = out['jac'].reshape(n, -1) if aos else out['jac'].reshape(-1, n).T
    mx, fro = thresholded_rel_err(got_j, ref_j)
    assert mx < 1e-8 and fro < 1e-12, (mx, fro)
    ref_d = o.batch_dydt(pres, np.ascontiguousarray(y.T))
    mx, _ = thresholded_rel_err(out['dydt'].reshape(-1, n).T, ref_d)
    assert mx < 1e-10


@pytest.mark.parametrize('name,lds_kb,aos,sum_last', [
    ('h2o2_n2', 156, False, 0), ('synth_alltypes', 156, True, 1),
    ('synth_alltypes', 50, False, 0),        # 64 states x 4 lane groups, 13 slots: rows split into column parts
    ('synth_srichb', 156, False, 0), ('synth_fracnu', 52, True, 0), ('synth_mid24', 156, False, 1),
    ('gri30_shaped', 156, False, 0),         # 128 states x 2 lane groups
    ('usc2_shaped', 156, False, 0),          # 64 states x 4 lane groups, hub rows split
])
def test_table_driven_lane_kernel_matches_oracle(name, lds_kb, aos, sum_last, tables):
    """k_tab / k_tab_fin (csrc/pj_tab.h: the state-per-lane kernel that needs no compilation; program built by
    pj_tabprog.cpp) thread by thread on the host: stage, row blocks with accumulators in "LDS", output phase,
    energy row from the finished species rows -- against the oracle, and entry-wise against the binary128 truth
    for the two large mechanisms (the regrouped formulation is held to rtol 1e-6 on every entry)."""
    tab = tables(name)
    n = 70 if tab.nsp <= 30 else 24
    pres, y = synth.dist_b(n, tab.nsp, seed=5, Tlo=500, Thi=2700)
    I = np.ascontiguousarray(tab.I, dtype=np.int32)
    D = np.ascontiguousarray(tab.D)
    jac = np.full(tab.nsp * tab.nsp * n, np.nan)
    info = (ctypes.c_int * 6)()
    P = lambda a: a.ctypes.data_as(_dp)
    rc = _emu().emu_tab_run(I.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_long(I.size), P(D),
                            ctypes.c_long(D.size), ctypes.c_long(n), P(pres), P(np.ascontiguousarray(y)), P(jac),
                            int(aos), sum_last, ctypes.c_long(lds_kb * 1024), info)
    assert rc == 0
    L, G, B = info[0], info[1], info[2]
    assert L * G == 256 and info[5] <= lds_kb * 1024 and tab.nsp * L * 8 + 256 * B * 8 + 4 * 576 * 8 == info[5]
    got = jac.reshape(n, -1) if aos else jac.reshape(-1, n).T
    assert not np.isnan(got).any()           # every entry written
    o = Oracle(tab)
    o.lib.pjo_set_sum_last_species(sum_last)
    try:
        ref = o.batch_jacob(pres, np.ascontiguousarray(y.T))
    finally:
        o.lib.pjo_set_sum_last_species(0)
    assert jac_scaled_err(got, ref, tab.nsp) <= 1.0
    _, fro = thresholded_rel_err(got, ref)
    assert fro < 1e-12
    if tab.nsp > 30 and not sum_last:
        from oracle.oracle import OracleQuad
        mx, _ = thresholded_rel_err(got, OracleQuad(tab).batch_jacob(pres, np.ascontiguousarray(y.T)))
        assert mx < 1e-7, mx


def test_generated_build_dir_matches_reference_callers(tmp_path):
    """pyjac_amd.pywrap.generate_wrapper: the build directory the reference's functional tester works from --
    mechanism.h / mechanism.cuh carrying what check_numbers / check_optimized parse
    (functional_tester/test.py:289-332, 352-364) and modules importable by bare name
    (test.py:432-437, 739-741)."""
    import re
    import subprocess
    import sys
    from pyjac_amd.pywrap import generate_wrapper
    from pyjac_amd.mechanism import read_mech
    mech_file = os.path.join(ROOT, 'tests', 'golden', 'synth_alltypes.inp')
    d = generate_wrapper(mech_file, str(tmp_path / 'out'))
    m = read_mech(mech_file)
    for fn in ('mechanism.h', 'mechanism.cuh'):
        n_spec = n_reac = last_spec = None
        for line in open(os.path.join(d, fn)).read().split('\n'):
            a = re.search(r'^#define NSP (\d+)$', line)
            b = re.search(r'^#define FWD_RATES (\d+)$', line)
            c = re.search(r'^//last_spec (\d+)$', line)
            n_spec = int(a.group(1)) if a else n_spec
            n_reac = int(b.group(1)) if b else n_reac
            last_spec = int(c.group(1)) if c else last_spec
        assert n_spec == len(m.specs) and n_reac == len(m.reacs) and last_spec == m.fwd_spec_map[-1]
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); "
            "p = __import__('pyjacob'); c = __import__('cu_pyjacob'); "
            "assert all(hasattr(p, f) for f in ('py_dydt', 'py_eval_jacobian', 'py_eval_rxn_rates', "
            "'py_eval_spec_rates', 'py_get_rxn_pres_mod', 'py_eval_conc')); "
            "assert all(hasattr(c, f) for f in ('py_cuinit', 'py_cujac', 'py_cuclean')); print('ok')" % (d, ROOT))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == 'ok', out.stderr[-1500:]


def test_blob_validation_refuses_bad_flag_combinations_and_versions(tables):
    """validate_blob (csrc/pj_tables.cpp): a Chebyshev reaction that is also PLOG / falloff / third body, or SRI
    parameters on something that is not a plain falloff, would make the kernels index the wrong table (the record
    field is shared): refused at load time, with the version mismatch of an old table file reported as such."""
    import pyjac_amd
    from pyjac_amd import _lib
    from pyjac_amd.tables import F_CHEB, F_PLOG, F_SRI, IA_FLAGS, MechTables
    tab = tables('synth_srichb')
    I = np.array(tab.I, dtype=np.int32)
    fo = I[16 + IA_FLAGS]
    cheb = [i for i in range(tab.nrxn) if I[fo + i] & F_CHEB]
    sri = [i for i in range(tab.nrxn) if I[fo + i] & F_SRI]
    assert cheb and sri
    for idx, extra, msg in ((cheb[0], F_PLOG, 'Chebyshev'), (sri[0], F_PLOG, 'SRI')):
        J = I.copy()
        J[fo + idx] |= extra
        with pytest.raises(_lib.PyjacError, match=msg):
            pyjac_amd.Evaluator(MechTables(J, tab.D.copy(), tab.nsp, tab.nrxn, tab.nrev, tab.npres, []), specialize='off')
    J = I.copy()
    J[1] = 1
    with pytest.raises(_lib.PyjacError, match='version 1'):
        pyjac_amd.Evaluator(MechTables(J, tab.D.copy(), tab.nsp, tab.nrxn, tab.nrev, tab.npres, []), specialize='off')


def test_one_kernel_library_keeps_nothing_in_scratch_memory():
    """Performance guard (DESIGN.md section 5c'): the pair-store row kernel of the shipped one-kernel library of the
    53-species mechanism must not use scratch memory -- a scratch reload sits behind every Jacobian store issued before
    it, and the kernel loses a third of its speed to a handful of spilled registers.  Read from the code object's
    metadata (no GPU needed); skipped where the library or the ROCm binutils are not there."""
    import pyjac_amd
    from pyjac_amd import specbuild
    ev = pyjac_amd.Evaluator(MECHS['gri30_shaped'], specialize='off')
    so = ev.spec_path('rblk')
    if not so or not os.path.exists(so):
        pytest.skip('no prebuilt row-block library of the 53-species mechanism')
    res = [r for r in specbuild.kernel_resources(so) if 'k_rblk' in r[0]]
    if not res:
        pytest.skip('llvm-objcopy / llvm-readelf not available')
    # code objects in link order: pair stores, general, w = J v (specbuild.build_rblk)
    name, spills, scratch, lds = res[0]
    assert scratch == 0, 'pair-store row kernel: %d bytes of scratch per lane (%d spilled registers)' % (scratch, spills)
    assert lds <= 160 * 1024


def test_rblk_geometry_fits_the_lds(tmp_path):
    """The geometry specbuild picks must fit the 160 KB of LDS in EVERY build of the library (round 4 admitted the
    one-kernel factor-column geometry up to 62 species; its w = J v build needs 48 NSP bytes per state: 53).  The Python
    model (specbuild.rblk_lds_bytes) mirrors SM_DOUBLES of csrc/pj_rblk.hip -- pinned here by compiling one row kernel of
    a small 56-species mechanism for gfx950 with the geometry it picks (the kernel static_asserts its LDS layout)."""
    import subprocess
    import pyjac_amd
    from pyjac_amd import specbuild, synth_mech, _lib
    from pyjac_amd.kcfactors import kc_factor_rows
    for nsp in range(8, 141):
        for kcf_ok in (False, True):
            block, halves, kcf, single, ecols, coop = specbuild.rblk_geometry(nsp, kcf_ok, nkc=0)
            assert kcf == 0 or kcf_ok
            for jv in (False, True):
                # (K_c rows: the planner cuts the kernels where they stop fitting -- checked with none here)
                assert specbuild.rblk_lds_bytes(nsp, block, halves, kcf, single, 0, jv=jv, ecols=bool(ecols), coop=bool(coop)) \
                    <= specbuild.LDS_BYTES, (nsp, kcf_ok, jv)
    assert specbuild.rblk_geometry(53, True)[2:4] == (1, 1) and specbuild.rblk_geometry(54, True)[2] == 0
    if not os.path.exists(specbuild._hipcc()):
        pytest.skip('no hipcc')
    inp = str(tmp_path / 'm56.inp')
    with open(inp, 'w') as f:
        f.write(synth_mech.generate(56, 60, 6, 8, 0, 4, 1, seed=5, title='56-species test mechanism'))
    ev = pyjac_amd.Evaluator(inp, specialize='off')
    rows = kc_factor_rows(ev.tables)
    L = _lib.lib()
    import ctypes as ct
    if rows is not None:
        _lib.check(L.pj_mech_set_kc_factors(ev._h, rows.ctypes.data_as(ct.POINTER(ct.c_double)), rows.size))
    block, halves, kcf, single, ecols, coop = specbuild.rblk_geometry(ev.nsp, rows is not None, int(ev.tables.I[10]))
    hdr = str(tmp_path / 'm56.h')
    counts = (ct.c_int * 5)()
    _lib.check(L.pj_mech_emit_rblk_spec(ev._h, hdr.encode(), 56, 13, block, halves, single, 256, 0, 0, 0.0, 0.0, counts))
    nker = counts[0]
    for jv in (0, 1):
        subprocess.check_call([specbuild._hipcc(), '--offload-arch=gfx950', '-O1', '-std=c++17', '-fPIC', '-c', '-fsyntax-only',
                               '-DPJS_HEADER="%s"' % hdr, '-I', specbuild.CSRC, '-DPJQ_SUMSETS=%d' % (0 if nker == 1 else 2 * halves),
                               '-DPJQ_SINGLE=%d' % int(nker == 1), '-DPJQ_ECL=%d' % int(halves > 1), '-DPJQ_BLOCK=%d' % block,
                               '-DPJQ_C_LDS=0', '-DPJQ_HALVES=%d' % halves, '-DPJQ_KCF=%d' % kcf, '-DPJQ_COOP=%d' % coop,
                               '-DPJQ_PART=2', '-DPJQ_ID=%d' % (nker - 1), '-DPJQ_PAIR=0', '-DPJQ_JV=%d' % jv,
                               os.path.join(specbuild.CSRC, 'pj_rblk.hip')])
    # k_jvd (w = J v, every reaction once): the geometry picked for this mechanism against the kernel's own static_assert
    geo = specbuild.jvd_geometry(ev.nsp, int(ev.tables.I[10]), 0)
    assert geo == (128, 4, 1, 1)
    subprocess.check_call([specbuild._hipcc(), '--offload-arch=gfx950', '-O1', '-std=c++17', '-fPIC', '-c', '-fsyntax-only',
                           '-DPJS_HEADER="%s"' % hdr, '-I', specbuild.CSRC, '-DPJQ_SUMSETS=0', '-DPJQ_SINGLE=1', '-DPJQ_ECL=0',
                           '-DPJQ_BLOCK=%d' % geo[0], '-DPJQ_HALVES=%d' % geo[1], '-DPJQ_C_LDS=%d' % geo[2], '-DPJQ_V_LDS=%d' % geo[3],
                           '-DPJQ_PART=5', '-DPJQ_ID=0', os.path.join(specbuild.CSRC, 'pj_rblk.hip')])


def test_sweep_mechanisms_cover_every_planner_geometry():
    """tests/golden/sweep/ (make_sweep_mechs.py): a mechanism on each side of every geometry threshold of the row kernels,
    and no geometry launches a workgroup of fewer than 256 threads (until round 5 more than 120 species meant 64 states x ONE
    lane group: three of a CU's four SIMDs idle).  pyJac scales by emitting more files and never refuses a size
    (create_jacobian.py:2213-2223).  Every file of the sweep has a prebuilt library in __graft_entry__.spec_build_list()."""
    import glob
    import pyjac_amd
    from pyjac_amd import specbuild
    from pyjac_amd.kcfactors import kc_factor_rows
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    built = {os.path.basename(j[0]) for j in g.spec_build_list() if j[1] == 'rblk'}
    seen = {}
    files = sorted(glob.glob(os.path.join(GOLDEN, 'sweep', 'sweep_*.inp')))
    assert len(files) >= 14
    for f in files:
        assert os.path.basename(f) in built
        ev = pyjac_amd.Evaluator(f, specialize='off')
        assert ev.spec_kind() == 'rblk'
        geo = specbuild.rblk_geometry(ev.nsp, kc_factor_rows(ev.tables) is not None, int(ev.tables.I[10]), ev.n_fwd)
        seen[ev.nsp] = geo[:3]
        assert geo[0] * geo[1] >= 256, (f, geo)
    assert seen[17] == (128, 2, 1) and seen[54] == seen[56] == (256, 1, 0) and seen[57] == seen[120] == (128, 2, 0)
    assert specbuild.rblk_geometry(26, True)[:4] == (128, 2, 1, 1) and specbuild.rblk_geometry(27, True)[:4] == (64, 4, 1, 1)
    assert seen[121] == seen[140] == (64, 4, 0), seen
    for nsp in range(8, 300):
        block, halves = specbuild.rblk_geometry(nsp, False, 0, 2000)[:2]
        if 8 * nsp * block <= 150 * 1024:
            assert block * halves >= 256, nsp
    # beyond 120 species: SEVERAL row kernels (a translation unit per kernel stays compilable; one kernel gains nothing)
    assert specbuild.rblk_geometry(140, False, 100, 120)[3] == 0 and specbuild.rblk_geometry(140, False, 100, 1200)[3] == 0


def test_prose_tolerances_are_generated_from_the_measurements():
    """DESIGN.md quotes pyJac's distance from itself (the bound of the GPU parity tests) in a block that tools/refresh_docs.py
    generates from tests/golden/self_noise.json: a hand-typed or stale number fails here."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'refresh_docs.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = open(os.path.join(ROOT, 'DESIGN.md')).read()
    import json
    sn = json.load(open(os.path.join(GOLDEN, 'self_noise.json')))
    assert '%.3g' % sn['usc2_shaped']['self_noise'] in txt and '%.3g' % (10 * sn['gri30_shaped']['self_noise']) in txt
    assert len(txt.encode()) <= 40 * 1024, 'DESIGN.md: current sections only, at most 40 KB (history: docs/history/)'


def test_jvd_geometry_model():
    """specbuild.jvd_geometry: columns + the K_c rows a kernel may have to stage (at most what the rate-kernel plan leaves
    room for) fit the LDS, for every size the row-block family serves."""
    from pyjac_amd import specbuild
    for nsp in range(8, 141):
        rate_block = 128 if nsp > 64 else 0          # (build_rblk: r_block if r_clds else 0)
        for nkc in (0, 50, 300, 800, 2000):
            geo = specbuild.jvd_geometry(nsp, nkc, rate_block)
            if geo is None:
                continue
            block, groups, c_lds, v_lds = geo
            plan_rows = (specbuild.LDS_BYTES - nsp * rate_block * 8 - 2048) // 128
            rows = min(nkc, plan_rows) if nkc else plan_rows
            lds = 8 * (max(rows, 1) * 16 + (c_lds + v_lds) * nsp * block + (4 * block if groups > 1 else 0))
            assert lds <= specbuild.LDS_BYTES, (nsp, nkc, geo)
            assert block * groups <= 512 and (block * groups < 512 or nsp <= 64)      # 256 registers per lane at 512 threads
    assert specbuild.jvd_geometry(53, 313, 0) == (128, 4, 1, 1) and specbuild.jvd_geometry(111, 749, 128) == (64, 4, 1, 1)
