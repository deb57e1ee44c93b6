"""Mechanism handle over the C ABI (host arrays and torch device tensors)."""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib
from ._lib import LAYOUT_AOS, LAYOUT_SOA, check, dptr
from .mechanism import read_mech
from .tables import MechTables, build_tables


class Evaluator:
    """One mechanism on the current HIP device.

    Replaces "one compiled pyjacob module per mechanism"
    (pyjac/pywrap/pywrap_gen.py:66-128): the mechanism is loaded as tables.
    """

    def __init__(self, mech, therm: str = None, last_spec: str = None, specialize: str = 'auto'):
        """specialize: 'auto' attaches the prebuilt mechanism-specific kernels if
        pyjac_amd/spec/ holds them; 'build' also compiles them when missing (hipcc:
        seconds for H2-size mechanisms, about a minute on 8 cores for GRI-size);
        'off' never (table-driven kernel)."""
        if isinstance(mech, MechTables):
            self.tables = mech
            self.mechanism = None
        elif isinstance(mech, str) and mech.endswith('.pjtab'):
            self.tables = MechTables.load(mech)
            self.mechanism = None
        else:
            self.mechanism = read_mech(mech, therm, last_spec)
            self.tables = build_tables(self.mechanism)
        L = _lib.lib()
        I = np.ascontiguousarray(self.tables.I, dtype=np.int32)
        D = np.ascontiguousarray(self.tables.D, dtype=np.float64)
        h = ctypes.c_void_p()
        check(L.pj_mech_create(I.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), I.size,
                               dptr(D), D.size, ctypes.byref(h)))
        self._h = h
        self.nsp = L.pj_mech_nsp(h)
        self.n_fwd = L.pj_mech_fwd_rates(h)
        self.n_rev = L.pj_mech_rev_rates(h)
        self.n_pres_mod = L.pj_mech_pres_mod_rates(h)
        self.attached_spec = None
        if specialize != 'off':
            self.specialize(build=(specialize == 'build'))

    # ---- mechanism-specific state-per-lane kernels ----
    # csrc/pj_lane.hip: the whole Jacobian in one lane's registers (small mechanisms);
    # csrc/pj_rblk.hip: row-block kernels that rebuild the rates they need (the rest);
    # csrc/pj_rows.hip: its predecessor (rate kernel + row-block kernels through an HBM scratch array),
    # still the source of the rate-output kernels (pj_spec_rates) of the pj_rblk libraries
    SPEC_MAX_NSP, SPEC_MAX_RXN = 16, 64
    ROWS_BUDGET = 64          # accumulator doubles per row block (stays inside 256 VGPRs)
    ROWS_FUSE = 16            # row blocks per kernel
    ROWS_RATES_PER_PART = 128  # reactions per rate kernel (its coefficient tables sit in LDS)

    def spec_kind(self) -> str:
        """Which specialised kernel family specialize(build=True) compiles for this mechanism."""
        from .tables import F_CHEB, F_SRI, IA_FLAGS
        I = self.tables.I
        flags = I[I[16 + IA_FLAGS]:I[16 + IA_FLAGS] + self.n_fwd]
        # (pj_lane.hip carries no SRI / Chebyshev code: those mechanisms take the row-block family at any size)
        small = self.nsp <= self.SPEC_MAX_NSP and self.n_fwd <= self.SPEC_MAX_RXN
        return 'lane' if small and not any(int(f) & (F_SRI | F_CHEB) for f in flags) else 'rblk'

    _SPEC_STEM = {'lane': 'libpj_spec_%016x.so', 'rows': 'libpj_rows_%016x.so', 'fused': 'libpj_fused_%016x.so',
                  'rblk': 'libpj_rblk_%016x.so'}

    _SPEC_SOURCES = {'lane': ('pj_lane.hip', 'pj_math.h'), 'rows': ('pj_rows.hip', 'pj_rows_rate.inc', 'pj_rows_block.inc'),
                     'fused': ('pj_rows.hip', 'pj_rows_rate.inc', 'pj_rows_block.inc'),
                     'rblk': ('pj_rblk.hip', 'pj_math.h', 'pj_rows.hip', 'pj_rows_rate.inc')}
    _SPEC_ENV = ('PJ_LANE_FLAGS', 'PJ_ROWS_FLAGS', 'PJ_ROWS_RATES_FLAGS', 'PJ_ROWS_BUDGET', 'PJ_ROWS_FUSE',
                 'PJ_ROWS_RATES_PER_PART', 'PJ_ROWS_BLOCK', 'PJ_ROWS_RECOMPUTE_KR', 'PJ_RBLK_BUDGET', 'PJ_RBLK_FUSE',
                 'PJ_RBLK_BLOCK', 'PJ_RBLK_FLAGS', 'PJ_RBLK_DEFINES', 'PJ_RBLK_PAIR_MODES', 'PJ_RBLK_HALVES', 'PJ_RBLK_HALF_COST')

    def spec_path(self, kind: str = None, **opts) -> str:
        """File name of a specialised library: mechanism hash + a digest of everything else that shapes the
        binary (the kernel sources, pj_tables.{h,cpp}, this file with its flags, the build options and
        the PJ_* environment overrides), so a library built from other sources or with other options is
        never attached by accident."""
        import hashlib
        kind = kind or self.spec_kind()
        here = os.path.dirname(os.path.abspath(__file__))
        d = hashlib.sha1(repr((kind, sorted((k, v) for k, v in opts.items() if v is not None),
                               [(e, os.environ.get(e)) for e in self._SPEC_ENV if os.environ.get(e)])).encode())
        for f in self._SPEC_SOURCES[kind] + ('pj_tables.h', 'pj_tables.cpp'):
            with open(os.path.join(here, 'csrc', f), 'rb') as fh:
                d.update(fh.read())
        with open(os.path.abspath(__file__).replace('.pyc', '.py'), 'rb') as fh:
            d.update(fh.read())
        h = _lib.lib().pj_mech_spec_hash(self._h)
        stem = self._SPEC_STEM[kind] % h
        return os.path.join(here, 'spec', stem[:-3] + '_' + d.hexdigest()[:10] + '.so')

    def specialize(self, build: bool = False, kind: str = None, **rows_opts) -> bool:
        """Attach (and with build=True compile if missing) the mechanism-specific kernels.
        kind: 'lane' | 'rows' | 'fused' | None (whatever is there, else the default for the size).
        'fused' is the single-kernel variant of pj_rows.hip (one translation unit: minutes to
        compile for a 53-species mechanism; coefficient tables of all reactions must fit the LDS)."""
        L = _lib.lib()
        kinds = [kind] if kind else [self.spec_kind()] + [k for k in ('lane', 'rblk', 'fused', 'rows') if k != self.spec_kind()]
        so = next((self.spec_path(k, **rows_opts) for k in kinds if os.path.exists(self.spec_path(k, **rows_opts))), None)
        if so is None:
            if not build:
                return False
            so = self.spec_path(kinds[0], **rows_opts)
            if kinds[0] == 'lane':
                self._build_lane(so)
            elif kinds[0] == 'fused':
                self._build_fused(so, **rows_opts)
            elif kinds[0] == 'rblk':
                self._build_rblk(so, **rows_opts)
            else:
                self._build_rows(so, **rows_opts)
        check(L.pj_mech_attach_spec(self._h, so.encode()))
        self.attached_spec = so
        return True

    def _build_lane(self, so: str):
        import subprocess
        L = _lib.lib()
        here = os.path.dirname(os.path.abspath(__file__))
        os.makedirs(os.path.dirname(so), exist_ok=True)
        hdr = so[:-3] + '.h'
        check(L.pj_mech_emit_spec(self._h, hdr.encode()))
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        # reciprocal instead of IEEE division sequences, contraction, no NaN/Inf/-0 special-casing;
        # NO reassociation (it keeps every product of an accumulation chain live: +40 AGPRs, -5 %);
        # measured on MI355X against -ffast-math and plain -O3
        flags = os.environ.get('PJ_LANE_FLAGS',
                               '-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math '
                               '-ffinite-math-only -mllvm -amdgpu-schedule-relaxed-occupancy=1').split()
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC'] + flags +
                              ['-DPJS_HEADER="%s"' % hdr, '-I', os.path.join(here, 'csrc'),
                               '-o', so + '.tmp.%d' % os.getpid(), os.path.join(here, 'csrc', 'pj_lane.hip')])
        os.replace(so + '.tmp.%d' % os.getpid(), so)     # other ranks / processes never see a half-written library

    def _build_fused(self, so: str, budget: int = None, **_):
        """csrc/pj_rows.hip as ONE kernel (PJR_PART=3): a workgroup of 4 wavefronts per 64-state
        tile, scratch region per resident workgroup."""
        import subprocess
        L = _lib.lib()
        here = os.path.dirname(os.path.abspath(__file__))
        os.makedirs(os.path.dirname(so), exist_ok=True)
        hdr = so[:-3] + '.h'
        check(L.pj_mech_emit_rows_spec(self._h, hdr.encode(), int(budget or os.environ.get('PJ_ROWS_BUDGET', self.ROWS_BUDGET))))
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        flags = os.environ.get('PJ_ROWS_FLAGS',
                               '-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math').split()
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC'] + flags +
                              ['-DPJR_PART=3', '-DPJS_HEADER="%s"' % hdr, '-I', os.path.join(here, 'csrc'),
                               '-o', so + '.tmp.%d' % os.getpid(), os.path.join(here, 'csrc', 'pj_rows.hip')])
        os.replace(so + '.tmp.%d' % os.getpid(), so)

    def _build_rows(self, so: str, budget: int = None, fuse: int = None, rates_per_part: int = None):
        """One translation unit per kernel of csrc/pj_rows.hip, compiled in parallel."""
        import re
        import shutil
        import subprocess
        from concurrent.futures import ThreadPoolExecutor
        L = _lib.lib()
        here = os.path.dirname(os.path.abspath(__file__))
        os.makedirs(os.path.dirname(so), exist_ok=True)
        hdr = so[:-3] + '.h'
        budget = int(budget or os.environ.get('PJ_ROWS_BUDGET', self.ROWS_BUDGET))
        fuse = int(fuse or os.environ.get('PJ_ROWS_FUSE', self.ROWS_FUSE))
        rpp = int(rates_per_part or os.environ.get('PJ_ROWS_RATES_PER_PART', self.ROWS_RATES_PER_PART))
        check(L.pj_mech_emit_rows_spec(self._h, hdr.encode(), budget))
        nblk = int(re.search(r'NBLK = (\d+)', open(hdr).read()).group(1))
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        work = so[:-3] + '.obj'
        os.makedirs(work, exist_ok=True)
        # lanes per workgroup: the concentration columns (8 NSP bytes per lane) must fit the LDS
        block = 256 if self.nsp * 256 * 8 <= 150 * 1024 else 128 if self.nsp * 128 * 8 <= 150 * 1024 else 64
        block = int(os.environ.get('PJ_ROWS_BLOCK', block))
        # Row kernels rebuild c*k_r from c*k_f and K_c(T) instead of reading it back from the scratch
        # array (-17 % HBM bytes, +3..6 % on MI355X) when the K_c polynomial table fits the LDS next
        # to the concentration columns
        lt_sp = int(re.search(r'LT_SP = (\d+)', open(hdr).read()).group(1))
        recompute = int(os.environ.get('PJ_ROWS_RECOMPUTE_KR', int(lt_sp * 8 + self.nsp * block * 8 <= 156 * 1024)))
        base = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c',
                '-DPJS_HEADER="%s"' % hdr, '-DPJR_BLOCK=%d' % block, '-DPJR_C_LDS=%d' % int(self.nsp > 64),
                '-DPJR_RECOMPUTE_KR=%d' % recompute,
                '-I', os.path.join(here, 'csrc'), os.path.join(here, 'csrc', 'pj_rows.hip')]
        # no reassociation anywhere: it makes the compiler keep every product of an accumulation
        # chain live (AGPR traffic in the row kernels, 0.6 KB of spills per lane in the rate kernels)
        f_rates = os.environ.get('PJ_ROWS_RATES_FLAGS',
                                 '-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math '
                                 '-ffinite-math-only').split()
        f_rows = os.environ.get('PJ_ROWS_FLAGS',
                                '-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math').split()
        jobs = [(f_rows + ['-DPJR_PART=0'], 'host.o')]
        for i, r0 in enumerate(range(0, self.n_fwd, rpp)):
            jobs.append((f_rates + ['-DPJR_PART=1', '-DPJR_ID=%d' % i, '-DPJR_R0=%d' % r0,
                                    '-DPJR_R1=%d' % min(self.n_fwd, r0 + rpp)], 'rates%d.o' % i))
        for i, b0 in enumerate(range(0, nblk, fuse)):
            jobs.append((f_rows + ['-DPJR_PART=2', '-DPJR_ID=%d' % i, '-DPJR_B0=%d' % b0,
                                   '-DPJR_B1=%d' % min(nblk, b0 + fuse)], 'rows%d.o' % i))
        # longest first so the pool drains evenly
        jobs.sort(key=lambda j: 0 if j[1].startswith('rates') else 1)

        def run(job):
            subprocess.check_call(base + job[0] + ['-o', os.path.join(work, job[1])])
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            list(ex.map(run, jobs))
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so + '.tmp.%d' % os.getpid()] +
                              [os.path.join(work, j[1]) for j in jobs])
        os.replace(so + '.tmp.%d' % os.getpid(), so)
        shutil.rmtree(work, ignore_errors=True)

    RBLK_BUDGET = 56          # accumulator doubles per row block of pj_rblk.hip (4 dense + non-zero S per row)
    RBLK_FUSE = 13            # row blocks per kernel (at most)
    RBLK_FUSE_LARGE = 13      # ... for mechanisms whose concentration columns leave little LDS for the K_c rows

    def _build_rblk(self, so: str, budget: int = None, fuse: int = None, rates_per_part: int = None, defines=()):
        """csrc/pj_rblk.hip: row-block kernels that rebuild the rates they need (+ a pre-pass for the
        falloff / PLOG reactions), linked with the rate-output kernels of csrc/pj_rows.hip
        (-DPJR_RATES_LIB: pj_spec_rates).  One translation unit per kernel, compiled in parallel."""
        import re
        import shutil
        import subprocess
        from concurrent.futures import ThreadPoolExecutor
        L = _lib.lib()
        here = os.path.dirname(os.path.abspath(__file__))
        os.makedirs(os.path.dirname(so), exist_ok=True)
        hdr = so[:-3] + '.h'
        budget = int(budget or os.environ.get('PJ_RBLK_BUDGET', self.RBLK_BUDGET))
        fuse = int(fuse or os.environ.get('PJ_RBLK_FUSE', self.RBLK_FUSE if self.nsp <= 64 else self.RBLK_FUSE_LARGE))
        rpp = int(rates_per_part or os.environ.get('PJ_ROWS_RATES_PER_PART', self.ROWS_RATES_PER_PART))
        check(L.pj_mech_emit_rows_spec(self._h, hdr.encode(), budget))
        t = open(hdr).read()
        nblk = int(re.search(r'NBLK = (\d+)', t).group(1))
        npre = int(re.search(r'NPRE = (\d+)', t).group(1))
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        work = so[:-3] + '.obj'
        os.makedirs(work, exist_ok=True)
        # lanes per workgroup: the concentration columns (8 NSP bytes per lane) + the K_c table must fit the LDS
        # (a kernel stages only the K_c rows of its own reactions: at most 16 doubles per visit)
        block = 256 if self.nsp * 256 * 8 <= 112 * 1024 else 128 if self.nsp * 128 * 8 <= 120 * 1024 else 64
        block = int(os.environ.get('PJ_RBLK_BLOCK', block))
        c_lds = int(self.nsp > 64)
        common = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', '-DPJS_HEADER="%s"' % hdr,
                  '-I', os.path.join(here, 'csrc')]
        f_rows = os.environ.get('PJ_RBLK_FLAGS',
                                '-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math '
                                '-mllvm -amdgpu-schedule-relaxed-occupancy=1').split()
        f_rates = os.environ.get('PJ_ROWS_RATES_FLAGS',
                                 '-ffp-contract=fast -fno-math-errno -fno-signed-zeros -freciprocal-math '
                                 '-ffinite-math-only').split()
        # 128 states per workgroup leave two SIMDs of a CU idle: the workgroup is then two groups of lanes on
        # the same states (shared concentration columns), each running its own row blocks (pj_rblk.hip)
        halves = int(os.environ.get('PJ_RBLK_HALVES', 2 if block == 128 else 1))
        # (the 111-species kernels are short of registers: without the one-visit look-ahead of the K_c rows and
        # concentrations they spill half as much, and spill reloads queue behind the Jacobian stores: -3 %)
        rblk = common + f_rows + ['-DPJQ_BLOCK=%d' % block, '-DPJQ_C_LDS=%d' % c_lds, '-DPJQ_HALVES=%d' % halves] + \
            (['-DPJQ_CONC_AHEAD=0', '-DPJQ_KC_AHEAD=0'] if halves == 2 else []) + \
            list(defines) + os.environ.get('PJ_RBLK_DEFINES', '').split() + [os.path.join(here, 'csrc', 'pj_rblk.hip')]
        rows = common + ['-DPJR_BLOCK=%d' % (256 if self.nsp * 256 * 8 <= 150 * 1024 else 128),
                         '-DPJR_C_LDS=%d' % c_lds, '-DPJR_RATES_LIB', os.path.join(here, 'csrc', 'pj_rows.hip')]
        jobs = [(rblk + ['-DPJQ_PART=0'], 'qhost.o'), (rows + f_rows + ['-DPJR_PART=0'], 'rhost.o')]
        if npre:
            jobs.append((rblk + ['-DPJQ_PART=1'], 'pre.o'))
        # row kernels of (nearly) equal block counts, at most `fuse` blocks each
        nker = (nblk + fuse * halves - 1) // (fuse * halves)
        bounds = [nblk * i // nker for i in range(nker + 1)]

        def table(name):
            m = re.search(r'constexpr int %s\[\d+\]\[1\] = \{(.*?)\};\n' % name, t, re.S)
            return [int(x) for x in re.findall(r'\{(-?\d+),\}', m.group(1))]
        if halves == 2:
            # a kernel stages the K_c rows of all its blocks (128 bytes each) next to the concentration
            # columns: kernels are cut where the rows of one more block would not fit the LDS any more
            ri = [[int(x) for x in r.split(',') if x.strip()] for r in re.findall(
                r'\{([^{}]*)\}', re.search(r'constexpr int RI\[\d+\]\[\d+\] = \{(.*?)\};\n', t, re.S).group(1))]
            enum = re.search(r'enum \{ RI_FLAGS,(.*?)\};', open(os.path.join(here, 'csrc', 'pj_tables.h')).read(), re.S).group(1)
            names = ['RI_FLAGS'] + [x.strip() for x in enum.replace('\n', ' ').split(',') if x.strip()]
            kp, kc = names.index('RI_KC_PTR'), names.index('RI_KC_CNT')
            rxp, brx = table('BLK_RX_PTR'), table('BLK_RX')
            limit = (160 * 1024 - self.nsp * block * 8) // 128 - 2
            bounds, cur = [0], set()
            for b in range(nblk):
                g = set()
                for v in range(rxp[b], rxp[b + 1]):
                    r = ri[brx[v]]
                    if r[0] & 1:
                        g.update(range(r[kp], r[kp] + r[kc]))
                if b > bounds[-1] and (len(cur | g) > limit or b - bounds[-1] >= 2 * fuse):
                    bounds.append(b)
                    cur = set()
                cur |= g
            if nblk - bounds[-1] < 2 and len(bounds) > 1:        # a kernel needs a block per half
                bounds.pop()
            bounds.append(nblk)
            nker = len(bounds) - 1
        # two halves: the blocks of a kernel are cut where the halves' estimated times meet (a visit
        # ~0.24 us, a Jacobian entry of the output phase ~0.06 us: DESIGN.md section 5c)
        mids = []
        if halves == 2:
            rx, rw = table('BLK_RX_PTR'), table('BLK_ROW_PTR')
            cv, co = (float(x) for x in os.environ.get('PJ_RBLK_HALF_COST', '0.24,0.06').split(','))
            cost = [cv * (rx[b + 1] - rx[b]) + co * self.nsp * (rw[b + 1] - rw[b]) for b in range(nblk)]
            for i in range(nker):
                b0, b1 = bounds[i], bounds[i + 1]
                tot, acc, bm = sum(cost[b0:b1]), 0.0, b0 + 1
                for b in range(b0, b1 - 1):
                    acc += cost[b]
                    bm = b + 1
                    if acc >= 0.5 * tot:
                        if acc - 0.5 * tot > 0.5 * cost[b] and b > b0:
                            bm = b
                        break
                mids.append(min(max(bm, b0 + 1), b1 - 1))
        mid = lambda i: ['-DPJQ_BM=%d' % mids[i]] if halves == 2 else []
        # each row kernel twice: with pair stores (SoA output, whole workgroups: the fast path) and general
        pair_modes = [int(x) for x in os.environ.get('PJ_RBLK_PAIR_MODES', '1,0').split(',')]
        for i in range(nker):
            for pair in pair_modes:
                jobs.append((rblk + mid(i) + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % i, '-DPJQ_B0=%d' % bounds[i],
                                     '-DPJQ_B1=%d' % bounds[i + 1], '-DPJQ_FIRST=%d' % (i == 0),
                                     '-DPJQ_LAST=%d' % (i == nker - 1), '-DPJQ_PAIR=%d' % pair], 'rblk%d_%d.o' % (i, pair)))
            # ... and as w = J v (the Jacobian consumed in registers)
            jobs.append((rblk + mid(i) + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % i, '-DPJQ_B0=%d' % bounds[i],
                                 '-DPJQ_B1=%d' % bounds[i + 1], '-DPJQ_FIRST=%d' % (i == 0),
                                 '-DPJQ_LAST=%d' % (i == nker - 1), '-DPJQ_PAIR=0', '-DPJQ_JV=1'], 'rblk%d_jv.o' % i))
        for i, r0 in enumerate(range(0, self.n_fwd, rpp)):
            jobs.append((rows + f_rates + ['-DPJR_PART=1', '-DPJR_ID=%d' % i, '-DPJR_R0=%d' % r0,
                                           '-DPJR_R1=%d' % min(self.n_fwd, r0 + rpp)], 'rates%d.o' % i))
        # longest first so the pool drains evenly
        jobs.sort(key=lambda j: 0 if j[1].startswith('rates') else 1 if j[1].startswith('rblk') else 2)

        def run(job):
            subprocess.check_call(job[0] + ['-o', os.path.join(work, job[1])])
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            list(ex.map(run, jobs))
        tmp = so + '.tmp.%d' % os.getpid()
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] +
                              [os.path.join(work, j[1]) for j in jobs])
        os.replace(tmp, so)
        shutil.rmtree(work, ignore_errors=True)

    @property
    def spec_kernel(self) -> str:
        """'pj_lane' / 'pj_rows' for the attached specialisation, '' if none."""
        so = os.path.basename(self.attached_spec or '')
        return ('pj_lane' if so.startswith('libpj_spec_') else 'pj_rows' if so.startswith('libpj_rows_')
                else 'pj_rblk' if so.startswith('libpj_rblk_')
                else 'pj_fused' if so.startswith('libpj_fused_') else '')

    @property
    def has_spec(self) -> bool:
        return bool(_lib.lib().pj_mech_has_spec(self._h))

    def use_spec(self, on):
        """False/0: table-driven kernel; True/1: attached kernels for SoA Jacobians (default);
        2: attached kernels for every layout."""
        check(_lib.lib().pj_mech_use_spec(self._h, int(on)))

    def close(self):
        if getattr(self, '_h', None):
            _lib.lib().pj_mech_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tuning / options ----
    def set_launch(self, tile_states: int = 0, threads: int = 0):
        check(_lib.lib().pj_mech_set_launch(self._h, tile_states, threads))

    def get_launch(self):
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(_lib.lib().pj_mech_get_launch(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(tile_states=a.value, threads=b.value, lds_bytes=c.value)

    def set_sum_last_species(self, on: bool):
        check(_lib.lib().pj_mech_set_sum_last_species(self._h, int(on)))

    # ---- bytes per unit of work ----
    @property
    def jacobian_bytes_per_state(self) -> int:
        return 8 * (self.nsp + 1) + 8 * self.nsp * self.nsp

    @property
    def rates_bytes_per_state(self) -> int:
        return 8 * (self.nsp + 1) + 8 * self.nsp

    # ---- device tensors (torch is plumbing: memory + streams) ----
    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _chk(name, t, numel, device=None):
        """The C ABI takes raw pointers: a wrong size, dtype, device or a strided view would be an
        out-of-bounds access on the device, so refuse it here."""
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError('%s: expected a contiguous float64 CUDA tensor' % name)
        if t.numel() != numel:
            raise ValueError('%s: expected %d elements, got %d' % (name, numel, t.numel()))
        if device is not None and t.device != device:
            raise ValueError('%s: on %s, expected %s' % (name, t.device, device))

    def jacobian(self, pres, y, y_layout=LAYOUT_SOA, out=None, jac_layout=LAYOUT_SOA):
        """pres: (n,) cuda f64; y: SoA (NSP, n) or AoS (n, NSP) cuda f64.
        Returns jac as SoA (NSP*NSP, n) or AoS (n, NSP*NSP)."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        if out is None:
            shape = (self.nsp * self.nsp, n) if jac_layout == LAYOUT_SOA else (n, self.nsp * self.nsp)
            out = torch.empty(shape, dtype=torch.float64, device=pres.device)
        self._chk('out', out, n * self.nsp * self.nsp, pres.device)
        check(_lib.lib().pj_eval_jacobian_dev(self._h, n, pres.data_ptr(), y.data_ptr(), y_layout,
                                              out.data_ptr(), jac_layout, self._stream()))
        return out

    def jacobian_vec(self, pres, y, v, layout=LAYOUT_SOA, out=None):
        """w_s = J(Phi_s) v_s per state (the consumer of pyJac's sparse_multiplier fused into the
        Jacobian kernel when a pj_lane library is attached).  y, v, w: SoA (NSP, n) or AoS (n, NSP)."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        self._chk('v', v, n * self.nsp, pres.device)
        if out is None:
            out = torch.empty_like(v)
        self._chk('out', out, n * self.nsp, pres.device)
        check(_lib.lib().pj_eval_jacobian_vec_dev(self._h, n, pres.data_ptr(), y.data_ptr(), layout,
                                                  v.data_ptr(), out.data_ptr(), layout, self._stream()))
        return out

    def rates(self, pres, y, y_layout=LAYOUT_SOA, want=('conc', 'fwd', 'rev', 'pres_mod',
                                                          'spec_rates', 'dydt')):
        """All SoA outputs of pyjacob.cu's k_dydt pass as a dict of (rows, n) tensors."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        rows = dict(conc=self.nsp, fwd=self.n_fwd, rev=max(self.n_rev, 1),
                    pres_mod=max(self.n_pres_mod, 1), spec_rates=self.nsp, dydt=self.nsp)
        outs = {k: torch.zeros((rows[k], n), dtype=torch.float64, device=pres.device) for k in want}
        p = lambda k: outs[k].data_ptr() if k in outs else None
        check(_lib.lib().pj_eval_rates_dev(self._h, n, pres.data_ptr(), y.data_ptr(), y_layout,
                                           p('conc'), p('fwd'), p('rev'), p('pres_mod'),
                                           p('spec_rates'), p('dydt'), self._stream()))
        return outs

    def fd_jacobian(self, pres, y, out=None, jac_layout=LAYOUT_SOA):
        """Finite-difference Jacobian of dydt (the reference's comparison arm,
        performance_tester/fd_jacob.c); y SoA (NSP, n)."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        if out is None:
            shape = (self.nsp * self.nsp, n) if jac_layout == LAYOUT_SOA else (n, self.nsp * self.nsp)
            out = torch.empty(shape, dtype=torch.float64, device=pres.device)
        self._chk('out', out, n * self.nsp * self.nsp, pres.device)
        check(_lib.lib().pj_eval_fd_jacobian_dev(self._h, n, pres.data_ptr(), y.data_ptr(), out.data_ptr(),
                                                 jac_layout, self._stream()))
        return out

    def time_jacobian(self, pres, y, out, iters: int, y_layout=LAYOUT_SOA, jac_layout=LAYOUT_SOA):
        """Average kernel time (ms) over `iters` launches, HIP events on the
        launch stream (pj_time_jacobian_dev)."""
        self._chk('pres', pres, pres.numel())
        self._chk('y', y, pres.numel() * self.nsp, pres.device)
        self._chk('out', out, pres.numel() * self.nsp * self.nsp, pres.device)
        ms = ctypes.c_double()
        check(_lib.lib().pj_time_jacobian_dev(self._h, pres.numel(), pres.data_ptr(), y.data_ptr(),
                                              y_layout, out.data_ptr(), jac_layout, self._stream(),
                                              iters, ctypes.byref(ms)))
        return ms.value

    # ---- host batch driver: pyjacob.cu init / run / cleanup ----
    def init(self, num: int) -> int:
        return check(_lib.lib().pj_init(self._h, int(num)))

    def run(self, num, padded, pres, y, conc, fwd, rev, pres_mod, spec_rates, dy, jac):
        check(_lib.lib().pj_run(self._h, int(num), int(padded), dptr(pres), dptr(y), dptr(conc),
                                dptr(fwd), dptr(rev), dptr(pres_mod), dptr(spec_rates), dptr(dy),
                                dptr(jac)))

    def cleanup(self):
        check(_lib.lib().pj_cleanup(self._h))
