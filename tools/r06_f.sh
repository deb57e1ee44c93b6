#!/bin/bash
# round 6, session F: store cache-policy bits on the pair stores; VMEM-level counters of the GRI-shaped row kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
GRI=$R/pyjac_amd/data/gri30_shaped.inp
PJ_VAR_RATES=0 timeout 900 python tools/rblk_variants.py time $GRI 1000000 base3 st_sc1 st_sc0_sc1 st_sc1_nt st_sc0 base3 st_sc1 st_sc0_sc1 st_sc1_nt st_sc0 2>&1 | grep -v amdgpu > $O/r06_gri_variants_f.txt
cat $O/r06_gri_variants_f.txt
cd /tmp
LIB=$R/pyjac_amd/spec/var/gri30_shaped_base3.so
i=0
for set in "SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_NC_WRITE_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PJ_ONE_STEP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_v_$i --output-format csv -- python $R/tools/one_step.py $GRI 262144 2 rblk > $O/pmc_v_$i.log 2>&1
  python $R/tools/pmc_summarize.py $O/pmc_v_$i > $O/pmc_v_$i.json 2>&1
  python - <<PY
import json
try:
    d=json.load(open('$O/pmc_v_$i.json'))
    for k,v in d.items():
        if 'rblk' in k: print('$set'.split()[0], k, {c:round(x['mean']) for c,x in v.items()})
except Exception as e:
    print('set $i ERR', e); print(open('$O/pmc_v_$i.log').read()[-600:])
PY
  rm -rf $O/pmc_v_$i
done
