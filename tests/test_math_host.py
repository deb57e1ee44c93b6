"""CPU: csrc/pj_math.h compiled for the host (tests/emu/math_check.cpp) -- the exponentials and the logarithm the state-per-lane
kernels use instead of the device library's (rate_subs.py's exp / log / log10 / pow calls go through them): maximum relative
error against long-double libm over the kernels' argument ranges."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_lean_exponentials_and_logarithm_are_accurate_to_an_ulp(tmp_path):
    exe = str(tmp_path / 'math_check')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-I', os.path.join(ROOT, 'pyjac_amd', 'csrc'),
                           '-o', exe, os.path.join(HERE, 'emu', 'math_check.cpp')])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    err = {l.split()[0]: float(l.split()[1]) for l in out.splitlines() if not l.startswith('edge')}
    print(out)
    # (2^-53 = 1.1e-16: half an ulp)
    assert err['exp_one'] < 2.0e-16 and err['exp_pair'] < 2.0e-16 and err['log_lean'] < 2.0e-16, err
    assert err['exp_tab'] < 3.0e-16, err              # table entry's half ulp + the final fused multiply-add's
    edge = [l for l in out.splitlines() if l.startswith('edge')][0].split()
    assert float(edge[1]) == 0.0 and edge[2] == 'inf' and float(edge[3]) == 1.0 and abs(float(edge[4]) + 690.7755278982137) < 1e-12
