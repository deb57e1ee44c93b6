#!/bin/bash
# round 5, final GPU session of a build: the GPU test suite, the driver-style bench lines, the LU probe and the
# MFMA lower-bound microbenchmark (tools/micro/lu_mfma_bound.hip)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r05_pytest_gpu.log
tail -5 $O/r05_pytest_gpu.log
timeout 900 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
timeout 600 python bench.py --workload usc --no-cpu-baseline --no-also > $O/r05_bench_usc.json 2>> $O/r05_bench_default.err
timeout 600 python bench.py --workload h2 --no-cpu-baseline --no-also > $O/r05_bench_h2.json 2>> $O/r05_bench_default.err
python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'
for f in ('r05_bench_default.json','r05_bench_usc.json','r05_bench_h2.json'):
    try:
        j=json.loads(open(O+f).read().strip().split('\n')[-1])
        print(f, 'value %.4g ms %.3f frac %.3f kernel_ms %.3f' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms']))
        for k,v in (j.get('also') or {}).items():
            print('   also', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('kernel_ms','frac','solve_ms','jacobian_ms','ms','products_per_s','jacobians_per_s')} if isinstance(v,dict) else v)
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python tools/lu_probe.py > $O/r05_lu_probe_full.txt 2>&1
grep -v amdgpu $O/r05_lu_probe_full.txt
timeout 300 tools/micro/lu_mfma_bound 1000000 > $O/r05_lu_mfma_bound.txt 2>&1
cat $O/r05_lu_mfma_bound.txt
