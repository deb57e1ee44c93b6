"""Compile csrc/pj_rblk.hip with g++ through tests/emu/hip_shim.h (one "thread" per workgroup)
so the CPU suite can run the state-per-lane row-block kernels against the oracle.
Test infrastructure only."""
import hashlib
import os
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'pyjac_amd', 'csrc')


def cache_path(files, extra: bytes):
    """Path of a library in the build cache (test infrastructure, like ccache), keyed by the CONTENT of `files` and of
    `extra` (options, generated header text) and the compiler version; None if the cache is off (PJ_EMU_CACHE=off)."""
    # (a per-user directory with mode 0700, not a world-writable predictable path under the system temp directory: the cached
    # libraries are loaded with ctypes.CDLL)
    default = os.path.join(os.path.expanduser('~'), '.cache', 'pyjac_amd', 'emu') if os.path.expanduser('~') not in ('', '/') or os.access('/root', os.W_OK) \
        else os.path.join(ROOT, '.emu_cache')
    cache_dir = os.environ.get('PJ_EMU_CACHE', default)
    if cache_dir == 'off':
        return None
    try:
        os.makedirs(cache_dir, mode=0o700, exist_ok=True)
        if os.stat(cache_dir).st_uid != os.getuid():
            return None             # somebody else's directory: no cache
    except OSError:
        return None
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, 'rb').read())
    h.update(extra)
    h.update(subprocess.run(['g++', '--version'], capture_output=True).stdout)
    return os.path.join(cache_dir, h.hexdigest()[:32] + '.so')


def _report(what, out):
    """One line per emulation library (pytest -s / PJ_EMU_VERBOSE=1): was g++ run or did the library come from the cache?  A
    warm-cache run never invokes the compiler, so compiler or flag breakage would otherwise stay hidden."""
    if os.environ.get('PJ_EMU_VERBOSE', '1') != '0':
        print('emu build cache %s (%s)' % (what, os.path.basename(out)))


def cache_store(cached, out):
    if not cached:
        return
    try:
        os.makedirs(os.path.dirname(cached), exist_ok=True)
        tmp = cached + '.%d.tmp' % os.getpid()
        shutil.copyfile(out, tmp)
        os.replace(tmp, cached)
    except OSError:
        pass


def build_rblk(hdr: str, out: str, blocks_per_part: int = 4, rates_per_part: int = 128, c_lds: int = 0,
               opt: str = '-O1', defines=(), halves: int = 1, kcf: int = 0, single: int = 0, ecl: int = None,
               pre_halves: int = 1, fin: int = None, jvd=(1, None, 0), only_jvd: bool = False, only_rows: bool = False) -> str:
    """csrc/pj_rblk.hip for the host: row blocks that rebuild their rates + falloff / PLOG pre-pass (k_pre,
    k_rblk, also as w = J v) and the rate-output kernels (k_rate, one per `rates_per_part` reactions, with and
    without the per-reaction outputs), the way specbuild.build_rblk links them.  halves: lane groups of the row kernels
    (each one OS thread in the emulation); kcf: equilibrium constants from the per-species factor columns (the header
    must carry the rows: pj_mech_set_kc_factors before it was emitted); single: one row kernel for every block;
    ecl: the energy-row terms a row block cannot see summed by the pre-pass (PJQ_ECL; default as specbuild: with several
    lane groups and polynomial K_c); pre_halves: lane groups of the pre-pass (2 needs c_lds); jvd: (lane groups,
    concentrations in LDS, vector in LDS[, K_c rows from global memory[, look-ahead depth]]) of k_jvd (w = J v, every reaction once); only_jvd: no other kernels; only_rows: the Jacobian path
    alone (pre-pass + row kernels: no w = J v, no rate kernels)."""
    t = open(hdr).read()
    # build cache: kernel sources, shim, this recipe, the mechanism header, every option -- a warm container re-runs the suite
    # without g++
    cached = cache_path([os.path.join(CSRC, f) for f in ('pj_rblk.hip', 'pj_math.h', 'pj_rate_pre.inc', 'pj_tables.h')] +
                        [os.path.join(HERE, 'hip_shim.h'), os.path.abspath(__file__)],
                        t.encode() + repr((blocks_per_part, rates_per_part, c_lds, opt, tuple(defines), halves, kcf, single, ecl,
                                           pre_halves, fin, tuple(jvd), only_jvd, only_rows)).encode())
    if cached and os.path.exists(cached):
        shutil.copyfile(cached, out)
        _report('hit', out)
        return out
    _report('miss: g++' if cached else 'off: g++', out)
    work = out + '.obj'
    os.makedirs(work, exist_ok=True)
    nblk = int(re.search(r'NBLK = (\d+)', t).group(1))
    nrxn = int(re.search(r'NRXN = (\d+)', t).group(1))
    npre = int(re.search(r'NPRE = (\d+)', t).group(1))
    common = ['g++', opt, '-std=c++17', '-fPIC', '-pthread', '-c', '-x', 'c++', '-DPJR_HOST_EMU', '-DPJS_HEADER="%s"' % hdr] + \
        list(defines) + ['-I', HERE, '-I', CSRC]
    if single:
        blocks_per_part = nblk
    starts = list(range(0, nblk, blocks_per_part))
    if ecl is None:
        ecl = 1 if (halves > 1 and not kcf) else 0
    if fin is None:         # (as specbuild: off unless asked for)
        fin = 0
    common += ['-DPJQ_SUMSETS=%d' % (0 if (len(starts) == 1 and not fin) else 2 * halves), '-DPJQ_SINGLE=%d' % int(len(starts) == 1),
               '-DPJQ_ECL=%d' % ecl, '-DPJQ_FIN=%d' % fin]
    base = common + ['-DPJQ_BLOCK=1', '-DPJQ_C_LDS=%d' % c_lds, os.path.join(CSRC, 'pj_rblk.hip')]
    rblk = base + ['-DPJQ_HALVES=%d' % halves, '-DPJQ_KCF=%d' % kcf]
    jobs = [(rblk + ['-DPJQ_PART=0'], 'qhost.o')]
    if npre or ecl:
        jobs.append((base + ['-DPJQ_PART=1', '-DPJQ_HALVES=%d' % pre_halves], 'pre.o'))
    for n, b0 in enumerate(starts):
        jobs.append((rblk + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % n, '-DPJQ_B0=%d' % b0,
                             '-DPJQ_B1=%d' % min(nblk, b0 + blocks_per_part), '-DPJQ_FIRST=%d' % (n == 0),
                             '-DPJQ_LAST=%d' % (n == len(starts) - 1), '-DPJQ_PAIR=0'], 'rblk%d.o' % n))
        jobs.append((rblk + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % n, '-DPJQ_B0=%d' % b0,
                             '-DPJQ_B1=%d' % min(nblk, b0 + blocks_per_part), '-DPJQ_FIRST=%d' % (n == 0),
                             '-DPJQ_LAST=%d' % (n == len(starts) - 1), '-DPJQ_PAIR=0', '-DPJQ_JV=1'], 'rblk%d_jv.o' % n))
    if fin:
        jobs.append((rblk + ['-DPJQ_PART=4', '-DPJQ_NKER=%d' % len(starts)], 'fin.o'))
        jobs.append((rblk + ['-DPJQ_PART=4', '-DPJQ_NKER=%d' % len(starts), '-DPJQ_JV=1'], 'fin_jv.o'))
    rstarts = list(range(0, nrxn, rates_per_part))
    for n, r0 in enumerate(rstarts):
        for full in (0, 1):
            jobs.append((base + ['-DPJQ_PART=3', '-DPJQ_ID=%d' % n, '-DPJQ_R0=%d' % r0, '-DPJQ_R1=%d' % min(nrxn, r0 + rates_per_part),
                                 '-DPJQ_FIRST=%d' % (n == 0), '-DPJQ_LAST=%d' % (n == len(rstarts) - 1),
                                 '-DPJQ_FULL=%d' % full], 'rate%d_%d.o' % (n, full)))

    if only_jvd:
        jobs = jobs[:1]
    jg, jc, jvl = jvd[:3]
    jkg = int(len(jvd) > 3 and jvd[3])
    jah = ['-DPJQ_JVD_AHEAD=%d' % jvd[4]] if len(jvd) > 4 else []
    jc = int(jg > 1 or jvl) if jc is None else jc
    jflags = common + ['-DPJQ_BLOCK=1', '-DPJQ_C_LDS=%d' % jc, '-DPJQ_HALVES=%d' % jg, '-DPJQ_V_LDS=%d' % jvl, '-DPJQ_JVD_KC_GLOBAL=%d' % jkg] + jah + [os.path.join(CSRC, 'pj_rblk.hip')]
    for n, r0 in enumerate(rstarts):
        jobs.append((jflags + ['-DPJQ_PART=5', '-DPJQ_ID=%d' % n, '-DPJQ_R0=%d' % r0, '-DPJQ_R1=%d' % min(nrxn, r0 + rates_per_part),
                               '-DPJQ_FIRST=%d' % (n == 0), '-DPJQ_LAST=%d' % (n == len(rstarts) - 1)], 'jvd%d.o' % n))

    # k_jvd's dydt build (the fast lean rate kernel): ONE kernel for the mechanism
    if not only_jvd:
        jobs.append((jflags + ['-DPJQ_PART=5', '-DPJQ_ID=0', '-DPJQ_R0=0', '-DPJQ_R1=%d' % nrxn, '-DPJQ_FIRST=1', '-DPJQ_LAST=1',
                               '-DPJQ_JVD_DYDT=1'], 'jvd_dydt.o'))
    if only_rows:
        jobs = [j for j in jobs if j[1] in ('qhost.o', 'pre.o', 'fin.o') or (j[1].startswith('rblk') and not j[1].endswith('_jv.o'))]

    def run(j):
        subprocess.check_call(j[0] + ['-o', os.path.join(work, j[1])])
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    subprocess.check_call(['g++', '-shared', '-pthread', '-o', out] + [os.path.join(work, j[1]) for j in jobs])
    cache_store(cached, out)
    return out
