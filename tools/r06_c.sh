#!/bin/bash
# round 6, session C: staggered start of the first round of workgroups (PJ_RBLK_STAGGER) on the GRI- and USC-shaped row kernels,
# instruction-cache counters of the one-kernel GRI-shaped build
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
GRI=$R/pyjac_amd/data/gri30_shaped.inp; USC=$R/pyjac_amd/data/usc2_shaped.inp
: > $O/r06_stagger.txt
for s in 0 1 2 4 0 1 3 8; do
  echo "PJ_RBLK_STAGGER=$s" >> $O/r06_stagger.txt
  PJ_RBLK_STAGGER=$s PJ_VAR_RATES=0 timeout 600 python tools/rblk_variants.py time $GRI 1000000 stg 2>&1 | grep -v amdgpu >> $O/r06_stagger.txt
done
for s in 0 1 2 4 0 1; do
  echo "USC PJ_RBLK_STAGGER=$s" >> $O/r06_stagger.txt
  PJ_RBLK_STAGGER=$s PJ_VAR_RATES=0 timeout 600 python tools/rblk_variants.py time $USC 200000 stg 2>&1 | grep -v amdgpu >> $O/r06_stagger.txt
done
cat $O/r06_stagger.txt
cd /tmp
for s in 0 1; do
PJ_RBLK_STAGGER=$s PJ_ONE_STEP_LIB=$R/pyjac_amd/spec/var/gri30_shaped_stg.so timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_ic_$s --output-format csv -- python $R/tools/one_step.py $GRI 262144 2 rblk > $O/pmc_ic_$s.log 2>&1
python $R/tools/pmc_summarize.py $O/pmc_ic_$s > $O/r06_rblk_gri_icache_counters_stagger$s.json 2>&1
PJ_RBLK_STAGGER=$s PJ_ONE_STEP_LIB=$R/pyjac_amd/spec/var/gri30_shaped_stg.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc_sq_$s --output-format csv -- python $R/tools/one_step.py $GRI 262144 2 rblk > $O/pmc_sq_$s.log 2>&1
python $R/tools/pmc_summarize.py $O/pmc_sq_$s > $O/r06_rblk_gri_sq_counters_stagger$s.json 2>&1
rm -rf $O/pmc_ic_$s $O/pmc_sq_$s
done
python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'
for s in (0,1):
  for f in ('r06_rblk_gri_icache_counters_stagger%d.json'%s,'r06_rblk_gri_sq_counters_stagger%d.json'%s):
    try:
        d=json.load(open(O+f))
        for k,v in d.items():
            if 'rblk' in k: print(f, k, {c:round(x['mean']) for c,x in v.items()})
    except Exception as e: print(f,'ERR',e)
PY
# 140 species: ONE row kernel of 64 states x four lane groups against several (libraries of session A's build, if still there)
cd $R
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/r06_n140_geometries.txt
import glob, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import pyjac_amd
from pyjac_amd import synth, _lib
mech = 'tests/golden/sweep/sweep_n140.inp'
n = 65536
for so in sorted(glob.glob('pyjac_amd/spec/libpj_rblk_46880acad077734a_*.so')):
    ev = pyjac_amd.Evaluator(mech, specialize='off')
    _lib.check(_lib.lib().pj_mech_attach_spec(ev._h, so.encode()))
    pres, y = synth.dist_b(n, ev.nsp)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda()
    jac = torch.empty((ev.nsp ** 2, n), dtype=torch.float64, device='cuda')
    ev.time_jacobian(d_p, d_y, jac, 2, 0, 0)
    ms = ev.time_jacobian(d_p, d_y, jac, 5, 0, 0)
    print('%s: %.3f ms per %d states, frac %.3f' % (os.path.basename(so), ms, n, n * ev.jacobian_bytes_per_state / ms / 1e6 / 8000))
PY
