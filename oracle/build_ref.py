#!/usr/bin/env python3
"""Build the REFERENCE's own generated-C evaluator for a mechanism (this
container only; needs /root/reference).

TEST INFRASTRUCTURE.  Runs pyJac's generator (python -m pyjac --lang c) on a
Chemkin file with its output in a scratch directory under /tmp, then compiles
the emitted C -- where it lies, never copied into the repo -- together with
oracle/ref_driver.c into ``oracle/_ref/libpyjac_ref_<name>.so``.  oracle/_ref/
is git-ignored; the built library travels to the GPU box with the snapshot
so bench.py can time "pyJac generated-C" on that box's host cores.

Flags follow pyjac/libgen/libgen.py:43-46 (-O3 -mtune=native; gnu99 instead of
c99 only so OpenMP pragmas and glibc prototypes resolve on current gcc).

VARIANTS: the same emitted C compiled with other floating-point settings
(``libpyjac_ref_<name>_O0.so``: -O0, no contraction; ``..._fma.so``: -O3 -mfma
-ffp-contract=fast).  tests/test_conditioning.py measures how far pyJac's
generated C is from ITSELF under them: the entries of the 53- / 111-species
Jacobians on which the HIP kernels differ from the reference by more than 1e-6
are entries on which the reference differs from the reference by the same
order (tests/golden/self_noise.json, written by tests/golden/make_self_noise.py).
"""
import glob
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('PYJAC_REFERENCE', '/root/reference')


DEFAULT_FLAGS = ('-O3', '-mtune=native')
VARIANTS = {
    '': DEFAULT_FLAGS,
    '_O0': ('-O0', '-ffp-contract=off'),
    '_fma': ('-O3', '-mtune=native', '-mfma', '-ffp-contract=fast'),
}


def build_ref(mech_path: str, name: str, last_spec: str = None, force: bool = False, variants=('',), therm_path: str = None) -> str:
    """Returns the path of the default-flags library; `variants`: keys of VARIANTS to build from the same
    generated sources (one generator run)."""
    path = lambda v: os.path.join(HERE, '_ref', 'libpyjac_ref_%s%s.so' % (name, v))
    newest_input = max(os.path.getmtime(p) for p in (mech_path, therm_path) if p)
    todo = [v for v in variants if force or not os.path.exists(path(v)) or os.path.getmtime(path(v)) < newest_input]
    if not todo:
        return path('')
    if not os.path.isdir(os.path.join(REF, 'pyjac')):
        raise RuntimeError('reference tree not present at %s' % REF)
    os.makedirs(os.path.join(HERE, '_ref'), exist_ok=True)
    work = tempfile.mkdtemp(prefix='pyjac_ref_%s_' % name, dir='/tmp')
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE='1')
    cmd = [sys.executable, '-W', 'ignore', '-m', 'pyjac', '--lang', 'c',
           '--input', os.path.abspath(mech_path), '-b', work]
    if last_spec:
        cmd += ['-ls', last_spec]
    if therm_path:
        cmd += ['--thermo', os.path.abspath(therm_path)]
    subprocess.check_call(cmd, env=env, cwd=work, stdout=subprocess.DEVNULL)
    srcs = [os.path.join(work, f) for f in
            ('chem_utils.c', 'dydt.c', 'spec_rates.c', 'rxn_rates.c',
             'rxn_rates_pres_mod.c', 'mechanism.c', 'mass_mole.c', 'jacob.c')
            if os.path.exists(os.path.join(work, f))]
    srcs += sorted(glob.glob(os.path.join(work, 'jacobs', '*.c')))
    srcs += sorted(glob.glob(os.path.join(work, 'rates', '*.c')))
    for v in todo:
        flags = list(VARIANTS[v])
        objs = []
        procs = []
        for s in srcs + [os.path.join(HERE, 'ref_driver.c')]:
            o = os.path.join(work, os.path.basename(s) + v + '.o')
            objs.append(o)
            procs.append(subprocess.Popen(
                ['gcc', '-std=gnu99'] + flags + ['-fPIC', '-fopenmp',
                 '-I', work, '-I', os.path.join(work, 'jacobs'),
                 '-I', os.path.join(work, 'rates'), '-c', s, '-o', o]))
            if len(procs) >= 6:
                for p in procs:
                    if p.wait():
                        raise RuntimeError('compile failed')
                procs = []
        for p in procs:
            if p.wait():
                raise RuntimeError('compile failed')
        # the reference's finite-difference arm, compiled where it lies with its entry point renamed
        fd_src = os.path.join(REF, 'pyjac', 'performance_tester', 'fd_jacob.c')
        if os.path.exists(fd_src):
            o = os.path.join(work, 'fd_jacob.c%s.o' % v)
            subprocess.check_call(['gcc', '-std=gnu99'] + flags + ['-fPIC', '-include', 'string.h',
                                   '-Deval_jacob=fd_eval_jacob', '-I', work, '-c', fd_src, '-o', o])
            objs.append(o)
        subprocess.check_call(['gcc', '-shared', '-fopenmp', '-o', path(v)] + objs + ['-lm'])
    import shutil
    shutil.rmtree(work, ignore_errors=True)
    return path('')


if __name__ == '__main__':
    mech = sys.argv[1]
    nm = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(os.path.basename(mech))[0]
    print(build_ref(mech, nm, force=True, variants=tuple(sys.argv[3].split(',')) if len(sys.argv) > 3 else ('',)))
