#!/bin/bash
# round-3 verification call: GPU suite, default bench line, AoS step time + kernel stats, LDS counters of the row kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-900 $O/bench_default.json; tail -3 $O/bench_default.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_aos --output-format csv -- python $R/tools/one_step.py $R/pyjac_amd/data/gri30_shaped.inp 1000000 5 rblk aos > $O/aos.log 2>&1
cat $O/aos.log | tail -2
cp $(ls $O/kt_aos/*/*kernel_stats.csv | head -1) $O/r03_rblk_gri_aos_kernel_stats.csv; head -8 $O/r03_rblk_gri_aos_kernel_stats.csv | cut -c1-160
rm -rf $O/kt_aos
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES -d $O/pmc_lds --output-format csv -- python $R/tools/one_step.py $R/pyjac_amd/data/gri30_shaped.inp 262144 2 rblk > $O/pmc_lds.log 2>&1
cd $R
python tools/pmc_summarize.py $O/pmc_lds > $O/r03_rblk_gri_lds_counters.json 2>&1
python - <<PY
import json
d=json.load(open('$O/r03_rblk_gri_lds_counters.json'))
for k in ('k_rblk','k_pre'):
    if k in d: print(k, {c:v['mean'] for c,v in d[k].items()})
PY
rm -rf $O/pmc_lds
