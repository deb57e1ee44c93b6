#!/bin/bash
# round 6: shader clock and package power while (a) the GRI-shaped Jacobian step, (b) w = J v on the same batch (compute only),
# (c) the store pattern alone (tools/micro/store_rate) run for several seconds each: is the step clock- / power-limited?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
sample() { # label, seconds
  for i in $(seq 1 $2); do
    echo "== $1 $(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" ; sleep 0.5
  done
}
( sample idle 4 ) > $O/r06_power_idle.txt 2>&1
python - <<'PY' > $O/r06_power_run.log 2>&1 &
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import pyjac_amd
from pyjac_amd import synth
ev = pyjac_amd.Evaluator('pyjac_amd/data/gri30_shaped.inp')
n = 1000000
pres, y = synth.dist_b(n, ev.nsp)
d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
jac = torch.empty((ev.nsp ** 2, n), dtype=torch.float64, device='cuda')
v = torch.randn_like(d_y); w = torch.empty_like(d_y)
def phase(name, fn, secs):
    open('/tmp/phase', 'w').write(name)
    t0 = time.time(); k = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); k += 20
    print(name, 'ms per call %.3f' % ((time.time() - t0) / k * 1e3), flush=True)
phase('jacobian', lambda: ev.jacobian(d_p, d_y, out=jac), 8)
phase('jacvec', lambda: ev.jacobian_vec(d_p, d_y, v, out=w), 8)
phase('rates', lambda: ev.rates(d_p, d_y, want=('dydt',)), 6)
open('/tmp/phase', 'w').write('done')
PY
sleep 6   # import + setup
: > $O/r06_power_samples.txt
while [ "$(cat /tmp/phase 2>/dev/null)" != "done" ]; do
  echo "== $(cat /tmp/phase 2>/dev/null) $(date +%s.%N)" >> $O/r06_power_samples.txt
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $O/r06_power_samples.txt
  sleep 0.4
done
wait
# the store pattern alone, in a loop
( for i in 1 2 3 4 5 6; do tools/micro/store_rate > /dev/null 2>&1; done ) &
sleep 1
for i in 1 2 3 4 5 6; do echo "== stores $(date +%s.%N)" >> $O/r06_power_samples.txt; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $O/r06_power_samples.txt; sleep 0.4; done
wait
cat $O/r06_power_run.log | grep -v amdgpu
python - <<'PY'
import re, os, collections
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/'
ph = None; acc = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(O + 'r06_power_samples.txt'):
    if l.startswith('=='): ph = l.split()[1]; continue
    m = re.search(r'(sclk|mclk|fclk) clock level: \S+ \((\d+)Mhz\)', l)
    if m: acc[ph][m.group(1)].append(int(m.group(2)))
    m = re.search(r'Power \(W\): ([\d.]+)', l)
    if m: acc[ph]['W'].append(float(m.group(1)))
for ph, d in acc.items():
    print(ph, {k: (round(sum(v) / len(v)), min(v), max(v), len(v)) for k, v in d.items()})
PY
head -12 $O/r06_power_idle.txt
