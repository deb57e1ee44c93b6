#!/bin/bash
# instruction-fetch counters of the row kernels: one lane group per workgroup (GRI-shaped) against two lane groups
# that run different code (USC-shaped)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp
for W in gri usc; do
  if [ $W = gri ]; then MECH=$R/pyjac_amd/data/gri30_shaped.inp; NP=262144; else MECH=$R/pyjac_amd/data/usc2_shaped.inp; NP=65536; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_ic_$W --output-format csv -- python $R/tools/one_step.py $MECH $NP 2 rblk > $O/pmc_ic_$W.log 2>&1
  python $R/tools/pmc_summarize.py $O/pmc_ic_$W > $O/r03_rblk_${W}_icache_counters.json 2>&1
  python - <<PY
import json
d=json.load(open('$O/r03_rblk_${W}_icache_counters.json'))
for k in d:
    if 'k_rblk' in k or 'k_pre' in k:
        m={c:v['mean'] for c,v in d[k].items()}
        print('$W', k[:40], {c: round(x) for c,x in m.items()}, 'hit rate %.4f' % (m['SQC_ICACHE_HITS']/max(m['SQC_ICACHE_REQ'],1)), 'ifetch_level/wave_cycles %.3f' % (m['SQ_IFETCH_LEVEL']/m['SQ_WAVE_CYCLES']))
PY
  rm -rf $O/pmc_ic_$W
done
