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
"""
import glob
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('PYJAC_REFERENCE', '/root/reference')


def build_ref(mech_path: str, name: str, last_spec: str = None, force: bool = False) -> str:
    out = os.path.join(HERE, '_ref', 'libpyjac_ref_%s.so' % name)
    if os.path.exists(out) and not force and \
            os.path.getmtime(out) >= os.path.getmtime(mech_path):
        return out
    if not os.path.isdir(os.path.join(REF, 'pyjac')):
        raise RuntimeError('reference tree not present at %s' % REF)
    os.makedirs(os.path.join(HERE, '_ref'), exist_ok=True)
    work = tempfile.mkdtemp(prefix='pyjac_ref_%s_' % name, dir='/tmp')
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE='1')
    cmd = [sys.executable, '-W', 'ignore', '-m', 'pyjac', '--lang', 'c',
           '--input', os.path.abspath(mech_path), '-b', work]
    if last_spec:
        cmd += ['-ls', last_spec]
    subprocess.check_call(cmd, env=env, cwd=work, stdout=subprocess.DEVNULL)
    srcs = [os.path.join(work, f) for f in
            ('chem_utils.c', 'dydt.c', 'spec_rates.c', 'rxn_rates.c',
             'rxn_rates_pres_mod.c', 'mechanism.c', 'mass_mole.c', 'jacob.c')
            if os.path.exists(os.path.join(work, f))]
    srcs += sorted(glob.glob(os.path.join(work, 'jacobs', '*.c')))
    srcs += sorted(glob.glob(os.path.join(work, 'rates', '*.c')))
    objs = []
    procs = []
    for s in srcs + [os.path.join(HERE, 'ref_driver.c')]:
        o = os.path.join(work, os.path.basename(s) + '.o')
        objs.append(o)
        procs.append(subprocess.Popen(
            ['gcc', '-std=gnu99', '-O3', '-mtune=native', '-fPIC', '-fopenmp',
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
        o = os.path.join(work, 'fd_jacob.c.o')
        subprocess.check_call(['gcc', '-std=gnu99', '-O3', '-mtune=native', '-fPIC', '-include', 'string.h',
                               '-Deval_jacob=fd_eval_jacob', '-I', work, '-c', fd_src, '-o', o])
        objs.append(o)
    subprocess.check_call(['gcc', '-shared', '-fopenmp', '-o', out] + objs + ['-lm'])
    return out


if __name__ == '__main__':
    mech = sys.argv[1]
    nm = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(os.path.basename(mech))[0]
    print(build_ref(mech, nm, force=True))
