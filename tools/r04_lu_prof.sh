#!/bin/bash
# round 4: batched LU (csrc/pj_lu.h) -- parity tests, timings of every size class, per-kernel durations and SQ counters
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_lu.py -m gpu -x -q 2>&1 | tail -3
python tools/lu_probe.py 10:1000000 24:1000000 40:1000000 53:1000000 64:500000 111:200000 2>/dev/null > $O/r04_lu_probe.txt
cat $O/r04_lu_probe.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r04_kt_lu --output-format csv -- python $R/tools/lu_probe.py 53:1000000 > $O/r04_kt_lu.log 2>&1
cp $(ls $O/r04_kt_lu/*/*kernel_stats.csv | head -1) $O/r04_lu_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $O/r04_pmc_lu_a --output-format csv -- python $R/tools/lu_one.py 53 262144 2 > $O/r04_pmc_lu_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/r04_pmc_lu_b --output-format csv -- python $R/tools/lu_one.py 53 262144 2 > $O/r04_pmc_lu_b.log 2>&1
cd $R
python tools/pmc_summarize.py $O/r04_pmc_lu_a $O/r04_pmc_lu_b > $O/r04_lu_sq_counters.json 2>&1
head -5 $O/r04_lu_kernel_stats.csv | cut -c1-220
head -40 $O/r04_lu_sq_counters.json
rm -rf $O/r04_pmc_lu_a $O/r04_pmc_lu_b $O/r04_kt_lu
