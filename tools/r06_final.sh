#!/bin/bash
# round 6, final GPU session of a build: the GPU test suite, the driver-style bench lines, the LU probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r06_pytest_gpu.log
tail -5 $O/r06_pytest_gpu.log
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err
timeout 600 python bench.py --workload usc --no-cpu-baseline --no-also > $O/r06_bench_usc.json 2>> $O/r06_bench_default.err
timeout 600 python bench.py --workload h2 --no-cpu-baseline --no-also > $O/r06_bench_h2.json 2>> $O/r06_bench_default.err
python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'
for f in ('r06_bench_default.json','r06_bench_usc.json','r06_bench_h2.json'):
    try:
        j=json.loads(open(O+f).read().strip().split('\n')[-1])
        print(f, 'value %.4g ms %.3f frac %.3f kernel_ms %.3f lib %s match %s traffic %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'], j['roofline'].get('library'), j['roofline'].get('profile_matches_library'), j['roofline'].get('traffic')))
        for k,v in (j.get('also') or {}).items():
            print('   also', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('kernel_ms','frac','solve_ms','jacobian_ms','ms','products_per_s','jacobians_per_s','error')} if isinstance(v,dict) else v)
        if 'cpu_baseline' in j: print('   cpu', {k:v for k,v in j['cpu_baseline'].items() if k in ('value','cores','kind','one_thread','cpu')})
        if 'end_to_end' in j: print('   e2e', {k:(round(v,3) if isinstance(v,float) else v) for k,v in j['end_to_end'].items() if k!='note'})
    except Exception as e:
        print(f, 'ERR', e); print(open(O+'r06_bench_default.err').read()[-1500:])
PY
