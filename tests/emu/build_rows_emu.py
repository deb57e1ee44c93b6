"""Compile csrc/pj_rows.hip with g++ through tests/emu/hip_shim.h (one "thread" per workgroup)
so the CPU suite can run the state-per-lane row-block kernels against the oracle.
Test infrastructure only."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'pyjac_amd', 'csrc')


def build(hdr: str, out: str, blocks_per_part: int = 4, rates_per_part: int = 128, c_lds: int = 0,
          opt: str = '-O1', defines=()) -> str:
    work = out + '.obj'
    os.makedirs(work, exist_ok=True)
    t = open(hdr).read()
    nblk = int(re.search(r'NBLK = (\d+)', t).group(1))
    nrxn = int(re.search(r'NRXN = (\d+)', t).group(1))
    base = ['g++', opt, '-std=c++17', '-fPIC', '-c', '-x', 'c++', '-DPJR_HOST_EMU', '-DPJR_BLOCK=1',
            '-DPJR_C_LDS=%d' % c_lds, '-DPJS_HEADER="%s"' % hdr] + list(defines) + ['-I', HERE, '-I', CSRC,
            os.path.join(CSRC, 'pj_rows.hip')]
    jobs = [(['-DPJR_PART=0'], 'host.o')]
    for n, r0 in enumerate(range(0, nrxn, rates_per_part)):
        jobs.append((['-DPJR_PART=1', '-DPJR_ID=%d' % n, '-DPJR_R0=%d' % r0,
                      '-DPJR_R1=%d' % min(nrxn, r0 + rates_per_part)], 'rates%d.o' % n))
    for n, b0 in enumerate(range(0, nblk, blocks_per_part)):
        jobs.append((['-DPJR_PART=2', '-DPJR_ID=%d' % n, '-DPJR_B0=%d' % b0,
                      '-DPJR_B1=%d' % min(nblk, b0 + blocks_per_part)], 'rows%d.o' % n))

    def run(j):
        subprocess.check_call(base + j[0] + ['-o', os.path.join(work, j[1])])
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    subprocess.check_call(['g++', '-shared', '-o', out] + [os.path.join(work, j[1]) for j in jobs])
    return out


def build_rblk(hdr: str, out: str, blocks_per_part: int = 4, rates_per_part: int = 128, c_lds: int = 0,
               opt: str = '-O1', defines=()) -> str:
    """csrc/pj_rblk.hip (row blocks that rebuild their rates + falloff / PLOG pre-pass) together with
    the rate-output kernels of csrc/pj_rows.hip (-DPJR_RATES_LIB), the way Evaluator._build_rblk links them."""
    work = out + '.obj'
    os.makedirs(work, exist_ok=True)
    t = open(hdr).read()
    nblk = int(re.search(r'NBLK = (\d+)', t).group(1))
    nrxn = int(re.search(r'NRXN = (\d+)', t).group(1))
    npre = int(re.search(r'NPRE = (\d+)', t).group(1))
    common = ['g++', opt, '-std=c++17', '-fPIC', '-c', '-x', 'c++', '-DPJR_HOST_EMU', '-DPJS_HEADER="%s"' % hdr] + \
        list(defines) + ['-I', HERE, '-I', CSRC]
    rblk = common + ['-DPJQ_BLOCK=1', '-DPJQ_C_LDS=%d' % c_lds, os.path.join(CSRC, 'pj_rblk.hip')]
    rows = common + ['-DPJR_BLOCK=1', '-DPJR_C_LDS=%d' % c_lds, '-DPJR_RATES_LIB', os.path.join(CSRC, 'pj_rows.hip')]
    jobs = [(rblk + ['-DPJQ_PART=0'], 'qhost.o'), (rows + ['-DPJR_PART=0'], 'rhost.o')]
    if npre:
        jobs.append((rblk + ['-DPJQ_PART=1'], 'pre.o'))
    starts = list(range(0, nblk, blocks_per_part))
    for n, b0 in enumerate(starts):
        jobs.append((rblk + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % n, '-DPJQ_B0=%d' % b0,
                             '-DPJQ_B1=%d' % min(nblk, b0 + blocks_per_part), '-DPJQ_FIRST=%d' % (n == 0),
                             '-DPJQ_LAST=%d' % (n == len(starts) - 1), '-DPJQ_PAIR=0'], 'rblk%d.o' % n))
        jobs.append((rblk + ['-DPJQ_PART=2', '-DPJQ_ID=%d' % n, '-DPJQ_B0=%d' % b0,
                             '-DPJQ_B1=%d' % min(nblk, b0 + blocks_per_part), '-DPJQ_FIRST=%d' % (n == 0),
                             '-DPJQ_LAST=%d' % (n == len(starts) - 1), '-DPJQ_PAIR=0', '-DPJQ_JV=1'], 'rblk%d_jv.o' % n))
    for n, r0 in enumerate(range(0, nrxn, rates_per_part)):
        jobs.append((rows + ['-DPJR_PART=1', '-DPJR_ID=%d' % n, '-DPJR_R0=%d' % r0,
                             '-DPJR_R1=%d' % min(nrxn, r0 + rates_per_part)], 'rates%d.o' % n))

    def run(j):
        subprocess.check_call(j[0] + ['-o', os.path.join(work, j[1])])
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    subprocess.check_call(['g++', '-shared', '-o', out] + [os.path.join(work, j[1]) for j in jobs])
    return out
