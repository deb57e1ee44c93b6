"""Build recipes of the mechanism-specific kernel libraries (hipcc; the counterpart of pyJac's libgen,
pyjac/libgen/libgen.py:330-420, with the difference that what is compiled is one fixed source per kernel
family against a header of constexpr tables, not emitted code).

The text of this file, of the kernel sources and of pj_tables.{h,cpp} is part of a library's file name
(source_digest): a library built from other sources or with other flags is never attached by accident.
"""
from __future__ import annotations

import contextlib
import ctypes
import fcntl
import hashlib
import os
import shutil
import subprocess
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SPEC_DIR = os.path.join(HERE, 'spec')

STEM = {'lane': 'libpj_spec_%016x', 'rblk': 'libpj_rblk_%016x'}
SOURCES = {'lane': ('pj_lane.hip', 'pj_math.h'), 'rblk': ('pj_rblk.hip', 'pj_math.h', 'pj_rate_pre.inc')}
# environment overrides that shape a binary (experiments): part of the digest
ENV = ('PJ_RBLK_NO_RATE_FAST', 'PJ_LANE_FLAGS', 'PJ_RBLK_BUDGET', 'PJ_RBLK_FUSE', 'PJ_RBLK_BLOCK', 'PJ_RBLK_FLAGS', 'PJ_RBLK_DEFINES',
       'PJ_RBLK_PAIR_MODES', 'PJ_RBLK_HALVES', 'PJ_RBLK_HALF_COST', 'PJ_RBLK_RATE_GROUPS', 'PJ_RBLK_RATE_DEFINES',
       'PJ_RBLK_KCF', 'PJ_RBLK_SINGLE', 'PJ_RBLK_NO_JV', 'PJ_RBLK_ECL', 'PJ_RBLK_WIDE', 'PJ_RBLK_FIN',
       'PJ_RBLK_JVD_GEOMETRY', 'PJ_RBLK_JVD_KC_GLOBAL', 'PJ_RBLK_JVD_DEFINES', 'PJ_RBLK_ROW_JV', 'PJ_RBLK_WIDE_SINGLE_RXN')

# reciprocal instead of IEEE division sequences, contraction, no -0 special-casing; NO reassociation (it keeps
# every product of an accumulation chain live: +40 AGPRs, -5 %); measured on MI355X against -ffast-math and
# plain -O3
LANE_FLAGS = ('-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math -ffinite-math-only '
              '-mllvm -amdgpu-schedule-relaxed-occupancy=1')
RBLK_FLAGS = '-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math -mllvm -amdgpu-schedule-relaxed-occupancy=1'

RBLK_BUDGET = 56          # accumulator doubles per row block of pj_rblk.hip (4 dense + non-zero S per row)
RBLK_BUDGET_KCF = 56      # ... of the one-kernel builds with per-species factor columns.  These must not keep ANYTHING in scratch
                          # memory (a scratch reload sits behind every Jacobian store issued before it: GRI-shaped 8.5 ms with
                          # 92 bytes of scratch per lane, 5.9 ms with none; profiles/r04_rblk_gri_variants.txt); since the
                          # energy row is finished column by column they have none at 48 and at 56 (5.88 / 6.03 ms at 48
                          # depending on unrelated edits of the source, 5.88 at 56: register allocation noise)
RBLK_BUDGET_HALVES = 40   # ... of the two-lane-group builds (57..120 species: 110 energy-row sums per lane leave less
                          # room; USC-shaped 6.46 ms at 56, 6.35 at 48, 6.28 at 40: profiles/r03_rblk_energy_row_atomics.txt)
RBLK_FUSE = 13            # row blocks per kernel and lane group (at most)

_src_digest = {}


def source_digest(kind: str):
    """sha1 of everything but the mechanism and the options that shapes a library of `kind`; None when the
    kernel sources are not installed (a deployment that ships prebuilt libraries only)."""
    if kind not in _src_digest:
        d = hashlib.sha1()
        try:
            for f in SOURCES[kind] + ('pj_tables.h', 'pj_tables.cpp'):
                with open(os.path.join(CSRC, f), 'rb') as fh:
                    d.update(fh.read())
            if kind == 'rblk':      # the per-species equilibrium-constant factors shape the header
                with open(os.path.join(HERE, 'kcfactors.py'), 'rb') as fh:
                    d.update(fh.read())
            with open(os.path.abspath(__file__).replace('.pyc', '.py'), 'rb') as fh:
                d.update(fh.read())
            _src_digest[kind] = d.hexdigest()
        except OSError:
            _src_digest[kind] = None
    return _src_digest[kind]


def library_path(kind: str, mech_hash: int, opts: dict):
    """File name of the library of a mechanism: hash of the mechanism tables + digest of the sources, this
    file, the options and the PJ_* overrides.  Without the sources: the newest prebuilt library of the
    mechanism and kind (or None)."""
    stem = STEM[kind] % mech_hash
    src = source_digest(kind)
    if src is None:
        import glob
        found = sorted(glob.glob(os.path.join(SPEC_DIR, stem + '_*.so')), key=os.path.getmtime)
        return found[-1] if found else None
    d = hashlib.sha1(repr((kind, src, sorted((k, v) for k, v in opts.items() if v is not None),
                           [(e, os.environ.get(e)) for e in ENV if os.environ.get(e)])).encode())
    return os.path.join(SPEC_DIR, stem + '_' + d.hexdigest()[:10] + '.so')


def _hipcc():
    return os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@contextlib.contextmanager
def compiler_slot():
    """One of os.cpu_count() compiler tokens shared by every build process of this user on this machine (flock on files of a
    per-user directory): __graft_entry__.build() compiles the libraries of several mechanisms side by side, each with a
    thread pool of its own, and without a common bound 8 cores would face 40 hipcc processes of 2 - 5 GB each."""
    n = int(os.environ.get('PJ_BUILD_SLOTS', os.cpu_count() or 1))
    d = os.path.join(tempfile.gettempdir(), 'pj_build_slots_%d' % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    fds = []
    try:
        while True:
            for i in range(n):
                fd = os.open(os.path.join(d, 'slot%d' % i), os.O_CREAT | os.O_RDWR, 0o600)
                try:
                    fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
                    fds.append(fd)
                    break
                except OSError:
                    os.close(fd)
            if fds:
                break
            time.sleep(0.25)
        yield
    finally:
        for fd in fds:
            os.close(fd)        # (closing the descriptor drops the lock)


def compile_call(cmd):
    with compiler_slot():
        subprocess.check_call(cmd)


def kernel_resources(so: str):
    """[(kernel name, vgpr spills, scratch bytes per lane, LDS bytes)] of the gfx950 code objects embedded in a built
    library (llvm-objcopy + llvm-readelf of the ROCm toolchain; [] when they are not installed)."""
    import re
    import tempfile
    llvm = os.path.join(os.path.dirname(os.path.dirname(_hipcc())), 'lib', 'llvm', 'bin')
    if not os.path.exists(os.path.join(llvm, 'llvm-readelf')):
        return []
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, 'fatbin')
        if subprocess.call([os.path.join(llvm, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so],
                           stderr=subprocess.DEVNULL) or not os.path.exists(fat):
            return []
        data = open(fat, 'rb').read()
        idx = [m.start() for m in re.finditer(b'\x7fELF', data)]
        for n, i in enumerate(idx):
            elf = os.path.join(d, 'co%d.elf' % n)
            with open(elf, 'wb') as f:
                f.write(data[i:idx[n + 1] if n + 1 < len(idx) else len(data)])
            notes = subprocess.run([os.path.join(llvm, 'llvm-readelf'), '--notes', elf], capture_output=True, text=True).stdout
            for blk in notes.split('- .agpr_count:')[1:]:
                g = lambda k: (re.findall(re.escape(k) + r':\s+(\S+)', blk) or ['0'])[0]
                out.append((g('.name'), int(g('.vgpr_spill_count')), int(g('.private_segment_fixed_size')),
                            int(g('.group_segment_fixed_size'))))
    return out


def _finish(tmp_so, tmp_hdr, work, so):
    """Publish a finished build: header first, library last, both by rename (other ranks / processes never see
    a half-written file; every process builds in its own work directory)."""
    os.replace(tmp_hdr, so[:-3] + '.h')
    os.replace(tmp_so, so)
    if work:
        shutil.rmtree(work, ignore_errors=True)


def build_lane(L, handle, so: str):
    """csrc/pj_lane.hip: the whole Jacobian in one lane's registers (small mechanisms); seconds."""
    from ._lib import check
    os.makedirs(os.path.dirname(so), exist_ok=True)
    hdr = so[:-3] + '.%d.h' % os.getpid()
    check(L.pj_mech_emit_spec(handle, hdr.encode()))
    flags = os.environ.get('PJ_LANE_FLAGS', LANE_FLAGS).split()
    tmp = so + '.tmp.%d' % os.getpid()
    compile_call([_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC'] + flags +
                 ['-DPJS_HEADER="%s"' % hdr, '-I', CSRC, '-o', tmp, os.path.join(CSRC, 'pj_lane.hip')])
    _finish(tmp, hdr, None, so)


LDS_BYTES = 160 * 1024


def rblk_lds_bytes(nsp: int, block: int, halves: int, kcf: int, single: int, nkc_rows: int = 0, jv: bool = False,
                   ecols: bool = False, coop: bool = None, ecl: bool = True, last: bool = True) -> int:
    """LDS bytes of a k_rblk workgroup: the mirror of SM_DOUBLES in csrc/pj_rblk.hip (concentration columns | factor
    columns or the kernel's K_c polynomial rows | the finished column sums of the energy row (one kernel, several lane
    groups, room permitting) or the cooperative prologue's partial sums | the epilogue's exchange area)."""
    g = halves
    if coop is None:
        coop = bool(kcf) and g > 1
    cl = nsp * block
    xt = 4 * nsp * block if kcf else 0
    ltk = 0 if kcf else 16 * nkc_rows
    off_ej = cl + xt + ltk
    ej_lds = bool(single) and g > 1 and not ecols
    ej = ((nsp if jv else nsp - 1) * block) if ej_lds else 0
    pred = ((4 if kcf else 2) * g * block) if (coop and g > 1) else 0
    main = off_ej + max(ej, pred)
    nex = 0 if ecl else nsp - 1
    epi_size = (nex * (g - 1) * block + 6 * g * block) if (g > 1 and last) else 0
    epi = (0 if epi_size <= off_ej else main) + epi_size if epi_size else 0
    return 8 * max(main, epi)


WIDE_SINGLE_RXN = 0       # more than 120 species: ONE row kernel only up to this many reactions.  A kernel's compile time and
                          # register pressure grow with the row blocks a lane group runs through (USC-shaped, 784 reactions,
                          # 64 states x four lane groups: one kernel 8.9 ms and 36 minutes of hipcc, six kernels 5.0 ms), and
                          # where one kernel is compilable it gains nothing: 140 species / 120 reactions, 65 536 states: 2.132 ms
                          # as one kernel (655 s of hipcc), 2.128 ms as two (profiles/r06_n140_geometries.txt).  So: never by
                          # default; PJ_RBLK_WIDE_SINGLE_RXN=<n> asks for it (a test geometry)


def rblk_geometry(nsp: int, kcf_ok: bool, nkc: int = 0, nrxn: int = None):
    """(block, halves, kcf, single, ecols, coop) of the row kernels of a mechanism; nkc: its K_c groups (polynomial row
    pairs, 128 bytes each); nrxn: its reactions (None: unknown, as many as it takes).
    * Per-species equilibrium-constant factors available and ONE kernel's columns of 64 states fit the LDS (concentration
      + two 16-byte factor columns + the finished column sums / the vector of the w = J v build: 48 NSP bytes per state,
      NSP <= 53): 64 states per workgroup, four lane groups on them, ONE row kernel (PJQ_KCF, PJQ_SINGLE) -- and where
      128 states fit (NSP <= 26), 128 states and two lane groups (round 6).
    * Otherwise the concentration columns (8 NSP bytes per lane) + the K_c rows of the kernel's reactions must fit:
      256 states (up to 56 species), or 128 states and two lane groups (57 .. 120 species), several row kernels -- or,
      beyond 120 species (and on request: PJ_RBLK_WIDE=1), 64 states and FOUR lane groups with a cooperative prologue
      (64 states with one lane group -- the geometry this size had until round 5 -- is a 64-thread workgroup: three of a
      CU's four SIMDs idle), as ONE row kernel if every K_c row of the mechanism fits next to the columns and the mechanism
      is small enough for one translation unit (the column sums of the energy row then travel through the hand-over
      array: PJQ_ECOLS)."""
    env = os.environ.get
    fits = lambda **kw: rblk_lds_bytes(nsp, **kw) <= LDS_BYTES
    kcf_default = int(kcf_ok and fits(block=64, halves=4, kcf=1, single=1, jv=True))
    kcf = int(env('PJ_RBLK_KCF', kcf_default))
    if kcf and not kcf_ok:
        raise ValueError('PJ_RBLK_KCF=1: the mechanism has no per-species factor rows (pyjac_amd/kcfactors.py)')
    ecols, coop = 0, 0
    if kcf:
        # small mechanisms (up to 26 species: the columns of 128 states fit): 128 states x TWO lane groups.  A lane group's
        # prologue share, its per-state scalars and its epilogue are per-wavefront work that does not shrink with the
        # mechanism, and with four groups on 64 states it is done twice as often per state: 17 - 24 species, 1e6 states,
        # same box (profiles/r06_small_variants_m*.txt) 0.32 - 0.34 -> 0.36, 0.35 -> 0.38, 0.39 -> 0.45, 0.34 -> 0.39 - 0.40 of
        # the roofline (a larger accumulator budget on top is a lottery: 0.40 / 0.45 here, scratch memory and 0.34 there)
        small = fits(block=128, halves=2, kcf=1, single=1, jv=True)
        block = int(env('PJ_RBLK_BLOCK', 128 if small else 64))
        halves = int(env('PJ_RBLK_HALVES', 2 if small else 4))
        single = int(env('PJ_RBLK_SINGLE', 1))
        coop = int(halves > 1)
    else:
        wide_default = int(nsp * 128 * 8 > 120 * 1024)
        wide = int(env('PJ_RBLK_WIDE', wide_default)) and nsp * 256 * 8 > 112 * 1024
        if wide:
            block, halves, coop = 64, 4, 1
            small = env('PJ_RBLK_WIDE') is not None or nrxn is None or nrxn <= int(env('PJ_RBLK_WIDE_SINGLE_RXN', WIDE_SINGLE_RXN))
            single = int(small and fits(block=64, halves=4, kcf=0, single=1, nkc_rows=nkc, ecols=True, coop=True))
            ecols = 1      # (applied only if the library comes out with ONE row kernel: build_rblk)
        else:
            block = 256 if nsp * 256 * 8 <= 112 * 1024 else 128 if nsp * 128 * 8 <= 120 * 1024 else 64
            single = 0
            # 128 states per workgroup leave two SIMDs of a CU idle: the workgroup is then two groups of lanes on the
            # same states (shared concentration columns), each running its own row blocks (pj_rblk.hip)
            halves = 2 if block == 128 else 1
        block = int(env('PJ_RBLK_BLOCK', block))
        halves = int(env('PJ_RBLK_HALVES', halves))
        single = int(env('PJ_RBLK_SINGLE', single))
    return block, halves, kcf, single, ecols, coop


def jvd_kc_global_default(nsp: int) -> int:
    """k_jvd's K_c rows from the mechanism table in global memory instead of LDS copies: the 64-state / four-group geometry of
    the large mechanisms is bound by its LDS traffic (a third of it those rows) while its vector memory path idles, and
    without the copies one kernel covers the mechanism: 2e5 USC-shaped products 2.34 -> 1.72 ms; the 128-state geometry
    (two wavefronts per SIMD) is not: 1e6 GRI-shaped products 2.20 -> 2.26 ms (profiles/r05_jvd_variants.txt)."""
    return int(nsp > 64)


def jvd_geometry(nsp: int, nkc: int, rate_block_clds: int = 0):
    """(states per workgroup, lane groups, concentrations in LDS, vector in LDS) of k_jvd, or None if nothing fits.
    nkc: K_c groups of the mechanism (128 bytes each; a kernel stages those of its reaction range: at most what the
    rate-kernel plan of pj_tables.cpp leaves room for, rate_block_clds = the states per workgroup that plan reserves
    concentration columns for).  A lane holds D_k (NSP doubles); concentrations and scaled vector sit in LDS columns that
    four lane groups share, each taking every fourth reaction: 128 states per workgroup (512 threads, two wavefronts per
    SIMD, 256 registers per lane) up to 64 species if the columns fit, 64 states otherwise.  Measured, 1e6 GRI-shaped
    products (profiles/r05_jvd_variants.txt): 128 x 4 2.20 ms; 256 states, one group, vector in registers 2.45; everything in
    registers 2.54 (a third of the lane's values then live in AGPRs: 4.7 k v_accvgpr moves)."""
    env = os.environ.get('PJ_RBLK_JVD_GEOMETRY')        # "block,groups,conc_in_lds,vector_in_lds"
    if env:
        return tuple(int(x) for x in env.split(','))
    plan_rows = (LDS_BYTES - nsp * rate_block_clds * 8 - 2048) // 128       # (pj_tables.cpp: rlimit)
    rows = min(nkc, plan_rows) if nkc else plan_rows
    fits = lambda block, groups, cols: 8 * (max(rows, 1) * 16 + cols * nsp * block + (4 * block if groups > 1 else 0)) <= LDS_BYTES
    cands = ([(128, 4, 1, 1)] if nsp <= 64 else []) + [(64, 4, 1, 1)]
    for block, groups, c_lds, v_lds in cands:
        if fits(block, groups, c_lds + v_lds):
            return block, groups, c_lds, v_lds
    return None


def build_rblk(L, handle, nsp: int, so: str, budget: int = None, fuse: int = None, rates_per_part: int = None, defines=(),
               kcf_rows=None, nkc: int = 0, nrxn: int = None):
    """csrc/pj_rblk.hip: row-block kernels that rebuild the rates they need (+ a pre-pass for the falloff / PLOG
    reactions) and the one-pass rate-output kernels (k_rate: pj_spec_rates).  One translation unit per kernel,
    compiled in parallel; which row blocks / reactions a kernel takes is planned by the C side
    (pj_mech_emit_rblk_spec) and travels in the header.  rates_per_part: K_c groups per rate kernel at most.
    kcf_rows: per-species equilibrium-constant factor rows (kcfactors.kc_factor_rows) or None."""
    import ctypes as ct
    import numpy as np
    from ._lib import check
    os.makedirs(os.path.dirname(so), exist_ok=True)
    pid = os.getpid()
    hdr = so[:-3] + '.%d.h' % pid
    work = so[:-3] + '.%d.obj' % pid
    os.makedirs(work, exist_ok=True)
    fuse = int(fuse or os.environ.get('PJ_RBLK_FUSE', RBLK_FUSE))
    block, halves, kcf, single, ecols, coop = rblk_geometry(nsp, kcf_rows is not None, nkc, nrxn)
    if kcf_rows is not None:
        rows = np.ascontiguousarray(kcf_rows, dtype=np.float64)
        check(L.pj_mech_set_kc_factors(handle, rows.ctypes.data_as(ct.POINTER(ct.c_double)), rows.size))
    else:
        check(L.pj_mech_set_kc_factors(handle, None, 0))
    budget = int(budget or os.environ.get('PJ_RBLK_BUDGET', RBLK_BUDGET_KCF if kcf else RBLK_BUDGET_HALVES if halves == 2 else RBLK_BUDGET))
    c_lds = int(nsp > 64)
    # rate kernels: concentrations in registers up to 64 species (256 states per workgroup), in LDS columns beyond
    # (128 states, two lane groups)
    r_clds = int(nsp > 64)
    r_block, r_halves = (128, 2) if r_clds else (256, 1)
    # the pre-pass (falloff / PLOG / Chebyshev reactions once per state): its own geometry, 256 states or -- with the
    # concentrations in LDS -- 128 states and two lane groups
    p_block, p_halves = (128, 2) if c_lds else (256, 1)
    # balance of the lane groups: time of a visit / of a Jacobian entry of the output phase (0: the planner's defaults,
    # measured on the 111-species kernels).  One-kernel factor-column builds, GRI-shaped, -DPJQ_TIMING with the stores flowing
    # (round 6, profiles/r06_rblk_gri_phase_cycles_timing.txt): 420 - 435 cycles per visit, 94 per entry, and two lane groups
    # waiting 23 - 25 k of 231 k cycles for the other two under round 4's 0.206 : 0.0575; with the visits weighted 0.5 : 0.094 the
    # planner moves two blocks and the step goes 5.82 - 5.85 -> 5.71 ms on the same box (profiles/r06_gri_variants_g.txt)
    cv, ce = (float(x) for x in os.environ.get('PJ_RBLK_HALF_COST', '0.5,0.094' if kcf else '0,0').split(','))
    counts = (ctypes.c_int * 5)()
    check(L.pj_mech_emit_rblk_spec(handle, hdr.encode(), budget, fuse, block, halves, single, r_block, r_clds,
                                   int(rates_per_part or os.environ.get('PJ_RBLK_RATE_GROUPS', 0)), cv, ce, counts))
    nker, nrate, npre = counts[0], counts[1], counts[2]
    # several lane groups, polynomial K_c (the register-starved 111-species geometry): what a row block cannot see of its
    # column of the energy row is summed once per state by the pre-pass (PJQ_ECL), so the row kernels carry no long-lived
    # sums (USC-shaped -3 .. -7 %).  The one-kernel factor-column builds keep their 32 long-lived sums: with the
    # pre-pass's extra visits they are 6 % SLOWER (GRI-shaped 6.23 -> 6.62 ms, profiles/r05_gri_variants_d.txt)
    ecl = int(os.environ.get('PJ_RBLK_ECL', 1 if (halves > 1 and not kcf) else 0))
    # ... and the energy row is finished by a kernel of its own (PJQ_FIN: k_fin) when its column sums travel through the
    # hand-over array anyway (several row kernels, or one without LDS room for them)
    # (measured, round 5: NOT the default -- with the workgroup-scope fence the last kernel's epilogue is 95 k of 1 280 k cycles
    # per wavefront and step, and k_fin as written keeps 2.5 KB of scratch memory per lane: USC-shaped 4.98 -> 6.08 ms)
    fin = int(os.environ.get('PJ_RBLK_FIN', 0))
    if fin and not (ecl and halves > 1 and (nker > 1 or ecols)):
        raise ValueError('PJ_RBLK_FIN=1 needs PJQ_ECL, several lane groups and the column sums in the hand-over array')
    common = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', '-DPJS_HEADER="%s"' % hdr, '-I', CSRC,
              '-DPJQ_SUMSETS=%d' % (0 if (nker == 1 and not fin) else 2 * halves), '-DPJQ_SINGLE=%d' % int(nker == 1), '-DPJQ_ECL=%d' % ecl,
              '-DPJQ_FIN=%d' % fin] + (['-DPJQ_ECOLS=1'] if (ecols and nker == 1) else [])
    flags = os.environ.get('PJ_RBLK_FLAGS', RBLK_FLAGS).split()
    src = os.path.join(CSRC, 'pj_rblk.hip')
    # (the 111-species kernels are short of registers: without the one-visit look-ahead of the K_c rows and
    # concentrations they spill half as much, and spill reloads queue behind the Jacobian stores: -3 %)
    # (the one-kernel builds must not keep anything in scratch memory, see below: three hand-over visits in flight
    # instead of four leave the register allocator the dozen registers it is short of at budget 48 -- GRI-shaped:
    # 12 bytes of scratch per lane at four, none at three or two)
    rblk = common + flags + ['-DPJQ_BLOCK=%d' % block, '-DPJQ_C_LDS=%d' % c_lds, '-DPJQ_HALVES=%d' % halves, '-DPJQ_KCF=%d' % kcf,
                             '-DPJQ_COOP=%d' % coop] + \
        (['-DPJQ_CONC_AHEAD=0', '-DPJQ_KC_AHEAD=0'] if (halves == 2 and not kcf) else []) + \
        (['-DPJQ_DEPTH=3'] if kcf and not any('PJQ_DEPTH' in d for d in list(defines) + os.environ.get('PJ_RBLK_DEFINES', '').split()) else []) + \
        list(defines) + os.environ.get('PJ_RBLK_DEFINES', '').split() + [src]
    pre = common + flags + ['-DPJQ_BLOCK=%d' % p_block, '-DPJQ_C_LDS=%d' % c_lds, '-DPJQ_HALVES=%d' % p_halves] + \
        list(defines) + os.environ.get('PJ_RBLK_DEFINES', '').split() + [src]
    rate = common + flags + ['-DPJQ_BLOCK=%d' % r_block, '-DPJQ_C_LDS=%d' % r_clds, '-DPJQ_HALVES=%d' % r_halves] + \
        list(defines) + os.environ.get('PJ_RBLK_RATE_DEFINES', '').split() + [src]
    # w = J v (k_jvd: every reaction once, pj_rblk.hip): D_k in registers; concentrations and vector in registers too, or in
    # LDS columns (jvd_geometry)
    # (K_c rows from the mechanism table in global memory instead of LDS copies: no limit on a kernel's reaction range)
    jvd_kcg = int(os.environ.get('PJ_RBLK_JVD_KC_GLOBAL', jvd_kc_global_default(nsp)))
    jvd_geo = jvd_geometry(nsp, 1 if jvd_kcg else nkc, r_block if r_clds else 0)
    jvd = None if jvd_geo is None else common + flags + \
        ['-DPJQ_BLOCK=%d' % jvd_geo[0], '-DPJQ_HALVES=%d' % jvd_geo[1], '-DPJQ_C_LDS=%d' % jvd_geo[2], '-DPJQ_V_LDS=%d' % jvd_geo[3]] + \
        (['-DPJQ_JVD_KC_GLOBAL=1', '-DPJQ_JVD_SB=4', '-DPJQ_R0=0', '-DPJQ_R1=pjs::NRXN', '-DPJQ_FIRST=1', '-DPJQ_LAST=1'] if jvd_kcg else []) + \
        list(defines) + os.environ.get('PJ_RBLK_JVD_DEFINES', '').split() + [src]
    jobs = [(rblk + ['-DPJQ_PART=0'], 'qhost.o')]
    if npre or ecl:
        jobs.append((pre + ['-DPJQ_PART=1'], 'pre.o'))
    # each row kernel three times: with pair stores (SoA output, whole workgroups: the fast path), general,
    # and as w = J v (the Jacobian consumed in registers)
    pair_modes = [int(x) for x in os.environ.get('PJ_RBLK_PAIR_MODES', '1,0').split(',') if x.strip()]
    # ... and, on request (PJ_RBLK_ROW_JV=1) or when k_jvd's columns do not fit the LDS, as w = J v with the Jacobian consumed
    # in registers (rounds 2 - 4's fused product: 3.6 visits per reaction; k_jvd visits each once)
    no_jv = bool(os.environ.get('PJ_RBLK_NO_JV'))       # (experiments: no w = J v kernels at all)
    row_jv = not no_jv and (bool(os.environ.get('PJ_RBLK_ROW_JV')) or jvd is None)
    for i in range(nker):
        for pair in pair_modes:
            jobs.append((rblk + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % i, '-DPJQ_PAIR=%d' % pair], 'rblk%d_%d.o' % (i, pair)))
        if row_jv:
            jobs.append((rblk + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % i, '-DPJQ_PAIR=0', '-DPJQ_JV=1'], 'rblk%d_jv.o' % i))
    if fin:
        jobs.append((rblk + ['-DPJQ_PART=4'], 'fin.o'))
        if row_jv:
            jobs.append((rblk + ['-DPJQ_PART=4', '-DPJQ_JV=1'], 'fin_jv.o'))
    if jvd is not None and not no_jv:
        for i in range(1 if jvd_kcg else nrate):
            jobs.append((jvd + ['-DPJQ_PART=5', '-DPJQ_ID=%d' % i], 'jvd%d.o' % i))
    # ... and the same kernel without the vector as the lean rate kernel (conc / spec_rates / dydt), where ONE kernel covers the
    # mechanism (nothing then travels from kernel to kernel): the lean k_rate kernels stay in the library as the fallback
    if jvd is not None and (jvd_kcg or nrate == 1) and not os.environ.get('PJ_RBLK_NO_RATE_FAST'):
        jobs.append((jvd + ['-DPJQ_PART=5', '-DPJQ_ID=0', '-DPJQ_JVD_DYDT=1'], 'jvd_dydt.o'))
    for i in range(nrate):
        for full in (0, 1):
            jobs.append((rate + ['-DPJQ_PART=3', '-DPJQ_ID=%d' % i, '-DPJQ_FULL=%d' % full], 'rate%d_%d.o' % (i, full)))
    # longest first so the pool drains evenly
    jobs.sort(key=lambda j: 0 if (j[1].startswith('rblk') and nker == 1) else 1 if j[1].startswith('rate') else 2 if j[1].startswith('rblk') else 3)

    def run(job):
        compile_call(job[0] + ['-o', os.path.join(work, job[1])])
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    # A row kernel that keeps values in scratch memory reloads them behind its own Jacobian stores (one in-order vmcnt
    # queue on gfx9): the one-kernel builds lose a third of their speed to a handful of spilled registers
    # (profiles/r04_rblk_gri_variants.txt).  A pair-store row kernel of such a build that came out with scratch is
    # compiled once more with two hand-over visits in flight instead of three (twelve more registers).
    if kcf:
        redo = []
        for j in jobs:
            if j[1].startswith('rblk') and j[1].endswith('_1.o'):
                res = [r for r in kernel_resources(os.path.join(work, j[1])) if 'k_rblk' in r[0]]
                if res and res[0][2] > 0:
                    redo.append((j[0] + ['-UPJQ_DEPTH', '-DPJQ_DEPTH=2'], j[1]))
        if redo:
            with ThreadPoolExecutor(max_workers=len(redo)) as ex:
                list(ex.map(run, redo))
    tmp = so + '.tmp.%d' % pid
    subprocess.check_call([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] +
                          [os.path.join(work, j[1]) for j in jobs])
    # ... and if a row kernel still has scratch, say so
    for name, spills, scratch, lds in kernel_resources(tmp):
        if 'k_rblk' in name and scratch > 0 and kcf:
            import sys
            sys.stderr.write('pyjac_amd.specbuild: %s: a k_rblk kernel of this library uses %d bytes of scratch memory per '
                             'lane (%d spilled registers); a smaller accumulator budget (PJ_RBLK_BUDGET, now %d) or '
                             '-DPJQ_DEPTH=2 (PJ_RBLK_DEFINES) avoids it\n' % (os.path.basename(so), scratch, spills, budget))
    _finish(tmp, hdr, work, so)
