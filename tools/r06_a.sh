#!/bin/bash
# round 6, session A: the GPU test suite with the planner-geometry sweep, the GRI-shaped phase map (-DPJQ_TIMING) and the
# no-store time (-DPJQ_NO_STORE) of the one-kernel build, a default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > $O/r06_pytest_gpu.log
tail -8 $O/r06_pytest_gpu.log
timeout 900 python -m pytest tests/test_gpu_sweep.py -q -m gpu -s 2>&1 | grep -E "^sweep|passed|failed|Error" | head -150 > $O/r06_sweep_detail.log
tail -3 $O/r06_sweep_detail.log
GRI=$R/pyjac_amd/data/gri30_shaped.inp
timeout 600 python tools/rblk_timing.py $GRI 1000000 $R/pyjac_amd/spec/var/gri30_shaped_timing.so > $O/r06_rblk_gri_phase_cycles.txt 2>&1
cat $O/r06_rblk_gri_phase_cycles.txt | grep -v amdgpu
PJ_VAR_RATES=0 timeout 600 python tools/rblk_variants.py time $GRI 1000000 rblk nostore timing > $O/r06_gri_variants_a.txt 2>&1
grep -v amdgpu $O/r06_gri_variants_a.txt
timeout 900 python bench.py --steps 50 --warmup 5 > $O/r06_bench_a.json 2> $O/r06_bench_a.err
python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'
try:
    j=json.loads(open(O+'r06_bench_a.json').read().strip().split('\n')[-1])
    print('value %.4g ms %.3f frac %.3f kernel_ms %.3f lib %s match %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'], j['roofline'].get('library'), j['roofline'].get('profile_matches_library')))
    for k,v in (j.get('also') or {}).items():
        print('   also', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('kernel_ms','frac','solve_ms','jacobian_ms','ms','products_per_s','jacobians_per_s','error')} if isinstance(v,dict) else v)
except Exception as e:
    print('ERR', e); print(open(O+'r06_bench_a.err').read()[-2000:])
PY
